"""Developer tool: forward / backward times of the four modes over odd image shapes (non-square, ragged widths, tiny, larger than LDS), normalised to
ns per (slice x angle x image pixel) -- a quick way to spot a shape that falls off its kernels' fast paths.   python tools/scan_shapes.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
shapes = [(128, 128), (128, 64), (64, 128), (100, 200), (200, 100), (96, 96), (33, 129), (129, 33), (256, 64), (64, 256), (192, 192), (256, 256), (300, 330), (40, 100)]
for H, W in shapes:
    for B, A in ((16, 20), (64, 45)):
        theta = np.pi * (np.arange(A) + 0.37) / A
        row = []
        for interp in ("nearest", "bilinear"):
            p = RotatePlan(theta, H, W, True, d, interp=interp)
            x = torch.rand((B, H, W), device=d); out = torch.empty((B, A, p.PW), device=d)
            g = torch.rand((B, A, p.PW), device=d); gi = torch.empty((B, H, W), device=d)
            n = 20
            for _ in range(2): graph_time(lambda: p.forward(x, out=out), n)
            tf = min(graph_time(lambda: p.forward(x, out=out), n) for _ in range(3)) * 1e6
            for _ in range(2): graph_time(lambda: p.backward(g, out=gi), n)
            tb = min(graph_time(lambda: p.backward(g, out=gi), n) for _ in range(3)) * 1e6
            k = B * A * H * W / 1e3
            row.append(f"{interp} fwd {tf:7.1f} us ({tf / k:6.3f}) bwd {tb:7.1f} us ({tb / k:6.3f})")
        print(f"{H:3d}x{W:3d} B={B:2d} A={A:2d} PW={p.PW:3d}: " + " | ".join(row), flush=True)
