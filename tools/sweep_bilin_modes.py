"""Developer tool: the bilinear forward's launch rules against forced launches -- task groups per class (BW) x task form (sorted bands /
plain (angle, block) tasks / row-split walks) -- over a grid of shapes; prints the library's time, the best forced one and what it was
(profiles/r05_bilin_fwd_rules.txt).   python tools/sweep_bilin_modes.py [BxNxA ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd.forward_functions import RotatePlan
from ct_pvae_amd import _lib
from time_modes import graph_time
d = torch.device("cuda", 0)
shapes = [(1, 128, 20), (5, 128, 20), (12, 128, 20), (25, 128, 20), (50, 128, 20), (76, 128, 20), (100, 128, 20), (200, 128, 20), (400, 128, 20),
          (50, 128, 10), (50, 128, 30), (50, 128, 45), (50, 128, 60), (50, 128, 90), (20, 128, 90), (10, 128, 180), (50, 128, 180), (100, 128, 90),
          (64, 64, 60), (256, 64, 20), (16, 256, 45), (7, 150, 33)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
x0 = torch.rand((400, 128, 128), device=d)
for B, N, A in shapes:
    theta = np.pi * (np.arange(A) + 0.37) / A
    plan = RotatePlan(theta, N, N, True, d, interp="bilinear")
    x = x0[:B] if N == 128 else torch.rand((B, N, N), device=d)
    out = torch.empty((B, A, plan.PW), device=d)
    n = 100 if B * N * N * A < 2e8 else 20
    for _ in range(3): graph_time(lambda: plan.forward(x, out=out), n)
    lib = min(graph_time(lambda: plan.forward(x, out=out), n) for _ in range(3)) * 1e6
    res = []
    for G in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
        for name, bs, rs in (("sorted", 1, 0), ("plain", 0, 0), ("rsplit", 0, 1)):
            with _lib.tuned("BW", G), _lib.tuned("BSORT", bs), _lib.tuned("BRSPLIT", rs):
                t = min(graph_time(lambda: plan.forward(x, out=out), n) for _ in range(2)) * 1e6
            res.append((t, name, G))
    res.sort()
    print(f"B={B} N={N} A={A}: library {lib:.2f} us | best " + " ; ".join(f"{nm} G={G} {t:.2f}" for t, nm, G in res[:4]) + f" | library / best {lib / res[0][0]:.3f}", flush=True)
