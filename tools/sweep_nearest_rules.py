"""Developer tool: the nearest pair's launch rules against forced launches over a grid of batch sizes and angle counts (128 x 128):
planned forward (u16 and step-coded plans x slices per unit NS x task groups G; the library's `auto` format and shape first) and
tf_compat backward (planned: NS x waves; stepped / segment as the library dispatches).  Prints library time, best forced, ratio
(profiles/r05_nearest_rules.txt).   python tools/sweep_nearest_rules.py [BxA ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
shapes = [(1, 20), (5, 20), (12, 20), (25, 20), (50, 20), (76, 20), (100, 20), (200, 20), (50, 10), (50, 45), (50, 90), (20, 90), (10, 180), (25, 180), (50, 180), (100, 90)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
N = int(os.environ.get('N', 128))     # N=100 python tools/sweep_nearest_rules.py ...: another image size
x0 = torch.rand((400, N, N), device=d)
for B, A in shapes:
    theta = np.pi * (np.arange(A) + 0.37) / A
    plans = {f: RotatePlan(theta, N, N, True, d, plan_format=f) for f in ("auto", "u16", "compact")}
    x = x0[:B]
    out = torch.empty((B, A, plans["auto"].PW), device=d)
    n = 100 if B * A <= 4000 else 30
    for _ in range(3): graph_time(lambda: plans["auto"].forward(x, out=out), n)
    lib = min(graph_time(lambda: plans["auto"].forward(x, out=out), n) for _ in range(3)) * 1e6
    res = []
    for f in ("u16", "compact"):
        for ns in (1, 2):
            if B < ns: continue
            for G in (1, 2, 3, 4, 5, 6, 8, 10, 12):
                with _lib.tuned("NS", ns), _lib.tuned("G", G):
                    try:
                        t = min(graph_time(lambda: plans[f].forward(x, out=out), n) for _ in range(2)) * 1e6
                    except Exception:
                        continue
                res.append((t, f"{f} NS={ns} G={G}"))
    res.sort()
    print(f"fwd N={N} B={B} A={A}: library ({'compact' if plans['auto']._compact else 'u16'}) {lib:.2f} us | best " + " ; ".join(f"{nm} {t:.2f}" for t, nm in res[:3]) + f" | library / best {lib / res[0][0]:.3f}", flush=True)
    g = torch.rand((B, A, plans["auto"].PW), device=d); gi = torch.empty((B, N, N), device=d)
    p = plans["auto"]
    for _ in range(3): graph_time(lambda: p.backward(g, out=gi), n)
    lib = min(graph_time(lambda: p.backward(g, out=gi), n) for _ in range(3)) * 1e6
    res = []
    for ns in (1, 2):
        for w in (2, 4, 8, 16):
            with _lib.tuned("BNS", ns), _lib.tuned("BW", w):
                t = min(graph_time(lambda: p.backward(g, out=gi), n) for _ in range(2)) * 1e6
            res.append((t, f"BNS={ns} BW={w}"))
    res.sort()
    print(f"bwd N={N} B={B} A={A}: library {lib:.2f} us | best " + " ; ".join(f"{nm} {t:.2f}" for t, nm in res[:3]) + f" | library / best {lib / res[0][0]:.3f}", flush=True)
