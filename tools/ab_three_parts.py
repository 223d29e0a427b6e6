"""Developer tool: the planned forward with and without the third part of its piece list (knob MIXG_G3=0 switches it off) at the batch
sizes where the model takes one; one process, HIP-graph replays, best of three, NaN-poisoned outputs bit-compared (profiles/r05_rounds.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
plans = {}
for A in (90, 180):
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
    plans[A] = RotatePlan(theta, 128, 128, True, d, plan_format="u16")
x = torch.rand((1200, 128, 128), device=d)
for _ in range(6):
    graph_time(lambda: plans[180].forward(x[:400]), 20)
for A, S in ((180, 388), (180, 396), (180, 400), (180, 404), (180, 644), (180, 900), (90, 388), (90, 644), (90, 900)):
    plan = plans[A]
    out = torch.full((S, A, plan.PW), float('nan'), device=d); ref = torch.full_like(out, float('nan'))
    ts = []
    for rep in range(3):
        with _lib.tuned("MIXG_G3", 0):
            t2 = graph_time(lambda: plan.forward(x[:S], out=ref), 20) * 1e6
        t3 = graph_time(lambda: plan.forward(x[:S], out=out), 20) * 1e6
        ts.append((t2, t3))
    print(f"S={S} A={A}: two parts {min(t[0] for t in ts):.2f}  three {min(t[1] for t in ts):.2f} us  {'equal' if torch.equal(out, ref) else 'DIFFER'}", flush=True)
