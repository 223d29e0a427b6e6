"""Developer tool: the bilinear tf_compat backward over slices per cell (knob SEG_NS) x rows per lane (SEG_PPT: 1 = 16 waves, 2 = 8 waves, 4 = 4 waves per
workgroup of 64 x 16 pixels) against the library's own choice, bit-compared (profiles/r05_bilin_bwd_sweep.txt)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
shapes = ((1, 128, 20), (5, 128, 20), (12, 128, 20), (25, 128, 20), (50, 128, 20), (76, 128, 20), (100, 128, 20), (200, 128, 20), (50, 128, 10), (50, 128, 45),
          (50, 128, 90), (20, 128, 90), (10, 128, 180), (50, 128, 180), (400, 128, 180), (64, 64, 60), (16, 256, 45), (8, 512, 90), (32, 512, 90))
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, N, A in shapes:
    theta = np.pi * np.arange(A) / A
    plan = RotatePlan(theta, N, N, True, d, interp="bilinear", backward=os.environ.get("BWD", "tf_compat"))   # BWD=exact: the true transpose
    gs = torch.rand((B, A, plan.PW), device=d)
    ref = plan.backward(gs)
    n = 100 if B * N * N * A < 3e8 else 10
    for _ in range(3): graph_time(lambda: plan.backward(gs), n)
    res = []
    for ns in (1, 2, 4):
        for ppt in (1, 2, 4):
            with _lib.tuned("SEG_NS", ns), _lib.tuned("SEG_PPT", ppt):
                out = plan.backward(gs)
                t = min(graph_time(lambda: plan.backward(gs), n) for _ in range(3)) * 1e6
            res.append((t, f"ns={ns} ppt={ppt}"))
            print(f"B={B} N={N} A={A} ns={ns} ppt={ppt}: {t:.2f} us {'equal' if torch.equal(out, ref) else 'DIFFER'}", flush=True)
    t = min(graph_time(lambda: plan.backward(gs), n) for _ in range(3)) * 1e6
    print(f"B={B} N={N} A={A} library: {t:.2f} us", flush=True)
    best = min(res)
    print(f"== B={B} N={N} A={A}: library {t:.2f} us | best forced {best[1]} {best[0]:.2f} | library / best {t / best[0]:.3f}", flush=True)
