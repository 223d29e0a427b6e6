import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
A = 20
theta = np.pi * (np.arange(A) + 0.37) / A
for N in (128, 136, 144, 152, 160, 168, 176, 184, 192, 200):
    st = RotatePlan(theta, N, N, True, d); st.backward_uses_step_plan = lambda S: True; st.backward_uses_plan = lambda S: False
    B = 8
    g = torch.rand((B, A, st.PW), device=d); gi = torch.empty((B, N, N), device=d)
    for _ in range(2): graph_time(lambda: st.backward(g, out=gi), 30)
    t = min(graph_time(lambda: st.backward(g, out=gi), 30) for _ in range(3)) * 1e6
    print(f"N={N} PW={st.PW} py={st.py}: direct segment kernel {t:.1f} us  (step plan {'yes' if st._step_plan is not None else 'no'}, bwd plan {'yes' if st.planned[1] else 'no'})", flush=True)
