"""Developer tool: the bilinear tf_compat backward over slices per cell (knob SEG_NS) x rows per lane (SEG_PPT: 1 = 16 waves, 2 = 8 waves, 4 = 4 waves per
workgroup of 64 x 16 pixels) against the library's own choice, bit-compared (profiles/r05_bilin_bwd_sweep.txt)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
for B, N, A in ((50, 128, 20), (1, 128, 20), (5, 128, 20), (12, 128, 20), (25, 128, 20), (50, 128, 180), (400, 128, 180), (100, 128, 20), (32, 512, 90)):
    theta = np.pi * np.arange(A) / A
    plan = RotatePlan(theta, N, N, True, d, interp="bilinear", backward=os.environ.get("BWD", "tf_compat"))   # BWD=exact: the true transpose
    gs = torch.rand((B, A, plan.PW), device=d)
    ref = plan.backward(gs)
    n = 100 if B * N * N * A < 3e8 else 10
    for _ in range(3): graph_time(lambda: plan.backward(gs), n)
    for ns in (1, 2, 4):
        for ppt in (1, 2, 4):
            with _lib.tuned("SEG_NS", ns), _lib.tuned("SEG_PPT", ppt):
                out = plan.backward(gs)
                t = min(graph_time(lambda: plan.backward(gs), n) for _ in range(3)) * 1e6
            print(f"B={B} N={N} A={A} ns={ns} ppt={ppt}: {t:.2f} us {'equal' if torch.equal(out, ref) else 'DIFFER'}", flush=True)
    t = min(graph_time(lambda: plan.backward(gs), n) for _ in range(3)) * 1e6
    print(f"B={B} N={N} A={A} library: {t:.2f} us", flush=True)
