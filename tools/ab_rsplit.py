"""Developer tool: the bilinear forward's row-split walks (knob BRSPLIT) against the plain unsorted tasks, one process, NaN-poisoned outputs bit-compared."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
shapes = ((50, 128, 20), (5, 128, 20), (1, 128, 20), (50, 128, 10), (50, 128, 30), (50, 128, 40), (50, 128, 60), (100, 128, 20), (76, 128, 20), (10, 128, 180), (20, 128, 90), (64, 64, 60), (256, 64, 20), (3, 100, 7), (7, 150, 33), (50, 128, 180))
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, N, A in shapes:
    theta = np.pi * np.arange(A) / A
    plan = RotatePlan(theta, N, N, True, d, interp="bilinear")
    x = torch.rand((B, N, N), device=d)
    out = torch.full((B, A, plan.PW), float('nan'), device=d); ref = torch.full_like(out, float('nan'))
    n = 100
    res = []
    for rep in range(3):
        with _lib.tuned("BRSPLIT", 0):
            t0 = graph_time(lambda: plan.forward(x, out=ref), n) * 1e6
        with _lib.tuned("BRSPLIT", 1):
            t1 = graph_time(lambda: plan.forward(x, out=out), n) * 1e6
        res.append((t0, t1))
    print(f"B={B} N={N} A={A}: plain {min(r[0] for r in res):.2f}  row-split {min(r[1] for r in res):.2f} us  {'equal' if torch.equal(out, ref) else 'DIFFER'}", flush=True)
