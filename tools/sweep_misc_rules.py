"""Developer tool: two more launch rules against forced launches -- the nearest exact adjoint (planned gather: slices per unit x waves) and the
stepped tf_compat backward (slice pairs per workgroup, STEP_NS) -- appended to profiles/r05_nearest_rules.txt."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
# nearest exact adjoint, 128^2
for B, A in ((5, 20), (25, 20), (50, 20), (100, 20), (50, 90), (50, 180), (400, 180)):
    theta = np.pi * (np.arange(A) + 0.37) / A
    p = RotatePlan(theta, 128, 128, True, d, backward="exact")
    g = torch.rand((B, A, p.PW), device=d); gi = torch.empty((B, 128, 128), device=d)
    n = 100 if B * A <= 4000 else 20
    for _ in range(3): graph_time(lambda: p.backward(g, out=gi), n)
    lib = min(graph_time(lambda: p.backward(g, out=gi), n) for _ in range(3)) * 1e6
    res = []
    for ns in (1, 2):
        for w in (2, 4, 8, 16):
            with _lib.tuned("BNS", ns), _lib.tuned("BW", w):
                t = min(graph_time(lambda: p.backward(g, out=gi), n) for _ in range(2)) * 1e6
            res.append((t, f"BNS={ns} BW={w}"))
    res.sort()
    print(f"nearest exact B={B} A={A}: library {lib:.2f} us | best " + " ; ".join(f"{nm} {t:.2f}" for t, nm in res[:3]) + f" | library / best {lib / res[0][0]:.3f}", flush=True)
# stepped backward: slice pairs per workgroup
for B, N, A in ((400, 128, 20), (200, 128, 20), (400, 128, 180), (32, 512, 90), (8, 512, 90), (16, 256, 45)):
    theta = np.pi * (np.arange(A) + 0.37) / A
    p = RotatePlan(theta, N, N, True, d)
    g = torch.rand((B, A, p.PW), device=d); gi = torch.empty((B, N, N), device=d)
    n = 20
    for _ in range(3): graph_time(lambda: p.backward(g, out=gi), n)
    lib = min(graph_time(lambda: p.backward(g, out=gi), n) for _ in range(3)) * 1e6
    row = []
    for ns in (2, 4):
        with _lib.tuned("STEP_NS", ns):
            t = min(graph_time(lambda: p.backward(g, out=gi), n) for _ in range(2)) * 1e6
        row.append(f"STEP_NS={ns} {t:.2f}")
    print(f"tf_compat backward B={B} N={N} A={A}: library {lib:.2f} us | " + " ; ".join(row), flush=True)
