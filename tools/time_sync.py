"""Developer tool: what the closing bracket of a short timed region costs -- torch.cuda.synchronize() alone against an event
spin (query loop) followed by synchronize -- on regions of 2 replays of a 10-step graph (bench.py --steps 20)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ct_pvae_amd import phantoms  # noqa: E402
from ct_pvae_amd.forward_functions import RotatePlan  # noqa: E402

d = torch.device("cuda", 0)
theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, 20)]
plan = RotatePlan(theta, 128, 128, True, d)
x, g = torch.rand((50, 128, 128), device=d), torch.rand((50, 20, 184), device=d)
sino, gimg = torch.empty((50, 20, 184), device=d), torch.empty((50, 128, 128), device=d)


def step():
    plan.forward(x, out=sino)
    plan.backward(g, out=gimg)


for _ in range(20):
    step()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for _ in range(10):
        step()
graph.replay()
torch.cuda.synchronize()
ev = torch.cuda.Event()
for k in (2, 50):
    for mode in ("sync", "spin"):
        ts = []
        for _ in range(300):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                graph.replay()
            if mode == "spin":
                ev.record()
                while not ev.query():
                    pass
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"{k * 10} steps, {mode}: median {np.median(ts) * 1e6 / (k * 10):.3f} us/step, region {np.median(ts) * 1e6:.1f} us")
