import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
for N in (32, 64, 100):
    for A in (20, 90):
        theta = np.pi * (np.arange(A) + 0.37) / A
        for B in (1, 16, 128):
            x = torch.rand((B, N, N), device=d)
            row = []
            for interp in ("nearest", "bilinear"):
                p = RotatePlan(theta, N, N, True, d, interp=interp)
                out = torch.empty((B, A, p.PW), device=d); g = torch.rand((B, A, p.PW), device=d); gi = torch.empty((B, N, N), device=d)
                ts = []
                for np_ in (-1, 1):
                    with _lib.tuned("NO_PLAN", np_):
                        for _ in range(2): graph_time(lambda: p.forward(x, out=out), 50)
                        tf = min(graph_time(lambda: p.forward(x, out=out), 50) for _ in range(3)) * 1e6
                        for _ in range(2): graph_time(lambda: p.backward(g, out=gi), 50)
                        tb = min(graph_time(lambda: p.backward(g, out=gi), 50) for _ in range(3)) * 1e6
                    ts.append((tf, tb))
                flag = " <<" if ts[1][0] < ts[0][0] * 0.97 or ts[1][1] < ts[0][1] * 0.97 else ""
                row.append(f"{interp} fwd {ts[0][0]:.1f} (direct {ts[1][0]:.1f}) bwd {ts[0][1]:.1f} (direct {ts[1][1]:.1f}){flag}")
            print(f"N={N} A={A} B={B}: " + " | ".join(row), flush=True)
