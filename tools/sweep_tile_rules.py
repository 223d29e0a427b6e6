"""Developer tool: the tile forwards' launch rules (slices larger than LDS) against forced launches: nearest (plan formats x TILED_NS x TILED_G) and
bilinear (BNS x BW x sorted bands / plain tasks) at a few batch sizes; library time, best forced, ratio (profiles/r05_tile_rules.txt)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
shapes = [(2, 512, 90), (4, 512, 90), (8, 512, 90), (16, 512, 90), (32, 512, 90), (8, 512, 20), (16, 256, 45), (64, 256, 20)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, N, A in shapes:
    theta = np.pi * (np.arange(A) + 0.37) / A
    x = torch.rand((B, N, N), device=d)
    n = 10
    # nearest
    p = RotatePlan(theta, N, N, True, d)
    out = torch.empty((B, A, p.PW), device=d)
    for _ in range(3): graph_time(lambda: p.forward(x, out=out), n)
    lib = min(graph_time(lambda: p.forward(x, out=out), n) for _ in range(3)) * 1e6
    res = []
    for ns in (1, 2, 4):
        for G in (1, 2, 3, 4, 6, 8):
            with _lib.tuned("TILED_NS", ns), _lib.tuned("TILED_G", G):
                try:
                    t = min(graph_time(lambda: p.forward(x, out=out), n) for _ in range(2)) * 1e6
                except Exception:
                    continue
            res.append((t, f"TILED_NS={ns} TILED_G={G}"))
    res.sort()
    print(f"nearest  B={B} N={N} A={A}: library {lib:.1f} us | best " + " ; ".join(f"{nm} {t:.1f}" for t, nm in res[:3]) + f" | library / best {lib / res[0][0]:.3f}", flush=True)
    # bilinear
    p = RotatePlan(theta, N, N, True, d, interp="bilinear")
    for _ in range(3): graph_time(lambda: p.forward(x, out=out), n)
    lib = min(graph_time(lambda: p.forward(x, out=out), n) for _ in range(3)) * 1e6
    res = []
    for ns in (2, 4):
        for G in (1, 2, 3, 4, 6, 8):
            for bs in (0, 1):
                with _lib.tuned("BNS", ns), _lib.tuned("BW", G), _lib.tuned("BSORT", bs):
                    try:
                        t = min(graph_time(lambda: p.forward(x, out=out), n) for _ in range(2)) * 1e6
                    except Exception:
                        continue
                res.append((t, f"BNS={ns} BW={G} {'sorted' if bs else 'plain'}"))
    res.sort()
    print(f"bilinear B={B} N={N} A={A}: library {lib:.1f} us | best " + " ; ".join(f"{nm} {t:.1f}" for t, nm in res[:3]) + f" | library / best {lib / res[0][0]:.3f}", flush=True)
