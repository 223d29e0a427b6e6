"""Developer timing (round 4): the u16 planned forward with its angles dealt to the XCDs (knob AFFINE = 1) against units dealt to
the XCDs (AFFINE = 0) and the library's own choice, HIP-graph replays; results compared bit for bit."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
def t_us(plan, x, out, n=100):
    plan.forward(x, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): plan.forward(x, out=out)
    for _ in range(2): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(r))
for A in (20, 45, 90, 180):
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
    p16 = RotatePlan(theta, 128, 128, True, dev, plan_format="u16")
    for B in (4, 8, 12, 16, 25, 32, 50, 100, 200, 400):
        x = torch.rand((B, 128, 128), device=dev)
        o = [torch.empty((B, A, p16.PW), device=dev) for _ in range(3)]
        t = []
        for k, v in enumerate((0, 1, None)):
            if v is not None: _lib.tune("AFFINE", v)
            t.append(t_us(p16, x, o[k], 100 if B * A < 20000 else 30))
            _lib.tune("*")
        same = torch.equal(o[0], o[1]) and torch.equal(o[0], o[2])
        print("A=%3d B=%3d  units->XCDs %7.2f us   angles->XCDs %7.2f us   library %7.2f us   %s" % (A, B, t[0], t[1], t[2], "equal" if same else "DIFFER"), flush=True)
