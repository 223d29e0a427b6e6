"""The two-part piece list of the planned forward forced over (G1, G2, units1) at one shape, against the library's own choice:
    python tools/sweep_mixg.py B A"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
B, A = int(sys.argv[1]), int(sys.argv[2])
theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
plan = RotatePlan(theta, 128, 128, True, d, plan_format="u16")
x = torch.rand((B, 128, 128), device=d); out = torch.empty((B, A, plan.PW), device=d)
n = 100 if B * A < 20000 else 20
units = (B + 1) // 2
print(f"B={B} A={A}: library {graph_time(lambda: plan.forward(x, out=out), n) * 1e6:.2f} us", flush=True)
res = []
for g1 in (1, 2, 3, 4, 5):
    for g2 in range(g1, 13):
        for k in range(0, 8):
            u1 = min(units, (256 * k) // (2 * g1))
            with _lib.tuned("NS", 2), _lib.tuned("MIXG_G1", g1), _lib.tuned("MIXG_G2", g2), _lib.tuned("MIXG_U1", u1):
                t = graph_time(lambda: plan.forward(x, out=out), n) * 1e6
            res.append((t, g1, g2, u1))
            if u1 == units:
                break
res.sort()
for t, g1, g2, u1 in res[:12]:
    print(f"  G1={g1} G2={g2} units1={u1}: {t:.2f} us")
