import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd.forward_functions import RotatePlan
from ct_pvae_amd import _lib
from time_modes import graph_time
d = torch.device('cuda', 0)
for N in (128, 160, 192, 256):
    for A in (20, 90):
        theta = np.pi * (np.arange(A) + 0.37) / A
        row = []
        for B in (4, 8, 16, 32, 64):
            lib = RotatePlan(theta, N, N, True, d)
            pl = RotatePlan(theta, N, N, True, d); pl.backward_uses_step_plan = lambda S: False; pl.backward_uses_plan = lambda S: True
            st = RotatePlan(theta, N, N, True, d); st.backward_uses_step_plan = lambda S: True; st.backward_uses_plan = lambda S: False
            g = torch.rand((B, A, lib.PW), device=d); gi = torch.empty((B, N, N), device=d)
            ts = []
            for p in (lib, pl, st, "stk"):
                try:
                    if p == "stk":
                        with _lib.tuned("SEG_PPT", 8):
                            for _ in range(2): graph_time(lambda: st.backward(g, out=gi), 30)
                            ts.append(min(graph_time(lambda: st.backward(g, out=gi), 30) for _ in range(3)) * 1e6)
                        continue
                    for _ in range(2): graph_time(lambda: p.backward(g, out=gi), 30)
                    ts.append(min(graph_time(lambda: p.backward(g, out=gi), 30) for _ in range(3)) * 1e6)
                except Exception as e:
                    ts.append(float('nan'))
            row.append(f"B={B}: {ts[0]:.1f} ({ts[1]:.1f} / {ts[2]:.1f} / {ts[3]:.1f})")
        print(f"N={N} A={A} library (planned / stepped entry / stepped kernel forced): " + "  ".join(row), flush=True)
