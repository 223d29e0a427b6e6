"""Developer timing: compact vs u16 forward plan over a grid of shapes (library's own launch choices)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib, phantoms
if os.environ.get("CTPVAE_VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CTPVAE_VARIANT_LIB"])
    _lib.torch_node = lambda: None
print("library:", _lib.LIB_PATH)
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
REP = 40
def t_us(plan, x, out):
    plan.forward(x, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP): plan.forward(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / REP)
    return float(np.median(r))
for A in (20, 90, 180):
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
    pc = RotatePlan(theta, 128, 128, True, dev, plan_format="compact")
    p16 = RotatePlan(theta, 128, 128, True, dev, plan_format="u16")
    pa = RotatePlan(theta, 128, 128, True, dev)        # plan_format="auto": decides per launch (RotatePlan.dense_plan)
    for B in (1, 2, 5, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64, 80, 100, 128, 160, 200, 256, 300, 400):
        x = torch.rand((B, 128, 128), device=dev)
        oc, o16, oa = (torch.empty((B, A, pc.PW), device=dev) for _ in range(3))
        tc, t16, ta = t_us(pc, x, oc), t_us(p16, x, o16), t_us(pa, x, oa)
        best = min(tc, t16)
        print("A=%3d B=%3d  compact %7.2f us   u16 %7.2f us   auto %7.2f us = %.3f x best (%s)  %s %s" % (
            A, B, tc, t16, ta, ta / best, pa.forward_kernel_name(B).replace("rotate_fwd_", "").replace("_kernel", ""),
            "equal" if torch.equal(oc, o16) and torch.equal(oa, o16) else "DIFFER",
            "<<" if tc < 0.97 * t16 else (">>" if tc > 1.03 * t16 else "")), flush=True)
