"""Developer tool: a THIRD part of the planned forward's piece list forced behind a forced two-part cut (knobs MIXG_G3 / MIXG_U2 behind
MIXG_G1 / _G2 / _U1): G3 = G2 .. G2+3 x units2 = units1 .. units step 4, the six fastest, bit-compared with the library's own launch.

    python tools/sweep_mixg3.py B A G1 G2 units1          (profiles/r05_rounds.txt)"""
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
x = torch.rand((B, 128, 128), device=d); out = torch.empty((B, A, plan.PW), device=d); ref = torch.empty_like(out)
n = 20
units = (B + 1) // 2
for _ in range(2):
    t = graph_time(lambda: plan.forward(x, out=ref), n) * 1e6
print(f"B={B} A={A}: library {t:.2f} us", flush=True)
g1, g2, u1 = (int(v) for v in sys.argv[3:6])
res = []
for g3 in range(g2, g2 + 4):
    for u2 in range(u1, units + 1, 4):
        with _lib.tuned("NS", 2), _lib.tuned("MIXG_G1", g1), _lib.tuned("MIXG_G2", g2), _lib.tuned("MIXG_U1", u1), _lib.tuned("MIXG_G3", g3), _lib.tuned("MIXG_U2", u2):
            t = graph_time(lambda: plan.forward(x, out=out), n) * 1e6
        res.append((t, g3, u2, torch.equal(out, ref)))
res.sort()
for _ in range(2):
    t = graph_time(lambda: plan.forward(x, out=ref), n) * 1e6
print(f"B={B} A={A}: library again {t:.2f} us", flush=True)
with _lib.tuned("NS", 2), _lib.tuned("MIXG_G1", g1), _lib.tuned("MIXG_G2", g2), _lib.tuned("MIXG_U1", u1):
    t = graph_time(lambda: plan.forward(x, out=out), n) * 1e6
print(f"two-part forced {t:.2f} us")
for t, g3, u2, eq in res[:6]:
    print(f"  G1={g1} G2={g2} u1={u1}  G3={g3} u2={u2}: {t:.2f} us {'equal' if eq else 'DIFFER'}")
