"""Developer timing: compact vs u16 forward plan over a grid of shapes (library's own launch choices)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
def t_us(plan, x, out):
    plan.forward(x, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(100): plan.forward(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / 100)
    return float(np.median(r))
for A in (20, 90, 180):
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
    pc = RotatePlan(theta, 128, 128, True, dev, plan_format="compact")
    p16 = RotatePlan(theta, 128, 128, True, dev, plan_format="u16")
    for B in (1, 2, 5, 10, 25, 50, 100, 200, 400):
        x = torch.rand((B, 128, 128), device=dev)
        oc, o16 = torch.empty((B, A, pc.PW), device=dev), torch.empty((B, A, pc.PW), device=dev)
        tc, t16 = t_us(pc, x, oc), t_us(p16, x, o16)
        print("A=%3d B=%3d  compact %7.2f us   u16 %7.2f us   %s %s" % (A, B, tc, t16, "equal" if torch.equal(oc, o16) else "DIFFER", "<<" if tc < 0.97 * t16 else (">>" if tc > 1.03 * t16 else "")), flush=True)
