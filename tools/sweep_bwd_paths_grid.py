"""Developer tool: the nearest tf_compat backward's DISPATCH (planned gather / stepped segment kernel, forward_functions.py
backward_uses_step_plan) against both paths forced, over batch sizes x angle counts at N x N (argument, default 128) (appended to profiles/r05_nearest_rules.txt)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for A in (10, 20, 45, 64, 90, 180):
    theta = np.pi * (np.arange(A) + 0.37) / A
    row = []
    for B in (40, 64, 80, 100, 128, 160, 200, 256, 400):
        lib = RotatePlan(theta, N, N, True, d)
        pl = RotatePlan(theta, N, N, True, d); pl.backward_uses_step_plan = lambda S: False; pl.backward_uses_plan = lambda S: True
        st = RotatePlan(theta, N, N, True, d); st.backward_uses_step_plan = lambda S: True; st.backward_uses_plan = lambda S: False
        g = torch.rand((B, A, lib.PW), device=d); gi = torch.empty((B, N, N), device=d)
        n = 50 if B * A <= 8000 else 15
        ts = []
        for p in (lib, pl, st):
            try:
                for _ in range(2): graph_time(lambda: p.backward(g, out=gi), n)
                ts.append(min(graph_time(lambda: p.backward(g, out=gi), n) for _ in range(3)) * 1e6)
            except Exception:
                ts.append(float('nan'))     # (no step plan below 128 x 128)
        best = np.nanmin(ts[1:])
        row.append(f"B={B}: {ts[0]:.1f} ({ts[1]:.1f} / {ts[2]:.1f}){'' if ts[0] <= best * 1.03 else ' <<'}")
    print(f"N={N} A={A}  library (planned / stepped) us:  " + "  ".join(row), flush=True)
