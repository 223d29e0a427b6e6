"""Launch shapes of the bilinear forward (rotate_fwd_bilin_kernel): slices per cell (BNS), task groups per class (BW), waves per
workgroup (WAVES).   python tools/sweep_bilin.py [B N A]"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from ct_pvae_amd.forward_functions import RotatePlan  # noqa: E402
from ct_pvae_amd import _lib, phantoms  # noqa: E402
from time_modes import graph_time  # noqa: E402

B, N, A = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (50, 128, 20)
dev = torch.device("cuda", 0)
theta = np.pi * np.arange(A) / A
plan = RotatePlan(theta, N, N, True, dev, interp="bilinear")
x = torch.rand((B, N, N), device=dev)
out = torch.empty((B, A, plan.PW), device=dev)
n = 100 if B * N * N * A < 2e8 else 20
print(f"B={B} N={N} A={A}: default {graph_time(lambda: plan.forward(x, out=out), n) * 1e6:.2f} us")
for ns in ((1, 2) if N <= 128 else (2, 4)):
    for G in (1, 2, 3, 4, 5, 6, 8, 10, 12):
        row = []
        for waves in (2, 4, 6, 8, 12, 16):
            with _lib.tuned("BNS", ns), _lib.tuned("BW", G), _lib.tuned("WAVES", waves):
                row.append(graph_time(lambda: plan.forward(x, out=out), n) * 1e6)
        print(f"  BNS={ns} G={G:2d}  waves 2/4/6/8/12/16: " + " ".join(f"{t:7.2f}" for t in row), flush=True)
