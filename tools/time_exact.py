"""Developer tool: the exact-transpose backward at the headline shape -- planned gather vs the atomic scatter vs tf_compat."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ct_pvae_amd import phantoms
from ct_pvae_amd.forward_functions import RotatePlan
d = torch.device("cuda", 0)
for B, A in ((50, 20), (50, 180), (10, 20)):
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
    g = torch.rand((B, A, 184), device=d)
    out = torch.empty((B, 128, 128), device=d)
    for name, kw in (("tf_compat planned", dict()), ("exact planned gather", dict(backward="exact")),
                     ("exact atomic scatter", dict(backward="exact", use_plan=False))):
        plan = RotatePlan(theta, 128, 128, True, d, **kw)
        for _ in range(5):
            plan.backward(g, out=out)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(50):
                plan.backward(g, out=out)
        graph.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            graph.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} A={A} {name}: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us")
