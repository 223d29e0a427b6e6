"""Developer timing: the 128 x 128 forward by plan format over a few (angles, slices) shapes; CTPVAE_VARIANT_LIB for timing builds.
   python tools/time_fwd_formats.py [formats, comma separated]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib, phantoms
if os.environ.get("CTPVAE_VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CTPVAE_VARIANT_LIB"])
    _lib.torch_node = lambda: None
print("library:", _lib.LIB_PATH)
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
fmts = sys.argv[1].split(',') if len(sys.argv) > 1 else ["u16", "compact"]
def t_us(plan, x, out, n=50):
    plan.forward(x, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): plan.forward(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(r))
for A in (20, 90, 180):
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
    plans = {f: RotatePlan(theta, 128, 128, True, dev, plan_format=f) for f in fmts}
    for B in (16, 50, 100, 200, 400):
        x = torch.rand((B, 128, 128), device=dev)
        outs = {f: torch.empty((B, A, plans[f].PW), device=dev) for f in fmts}
        ts = {f: t_us(plans[f], x, outs[f]) for f in fmts}
        same = all(torch.equal(outs[f], outs[fmts[0]]) for f in fmts)
        print("A=%3d B=%3d  " % (A, B) + "  ".join("%s %7.2f us" % (f, ts[f]) for f in fmts) + ("  equal" if same else "  DIFFER"), flush=True)
