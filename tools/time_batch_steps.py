"""Forward and backward launch time of the rotate projector at 128 x 128 against the batch size (50 ... 600), nearest and bilinear,
20 and 180 angles: where a launch steps from one round of workgroups to the next (profiles/r05_batch_steps.txt)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
for interp in ("nearest", "bilinear"):
    for A in (20, 180):
        theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
        plan = RotatePlan(theta, 128, 128, True, d, interp=interp)
        line = f"{interp} A={A}:"
        for B in (50, 100, 150, 200, 256, 300, 350, 400, 512, 600):
            g = torch.randn((B, A, plan.PW), device=d); gx = torch.empty((B, 128, 128), device=d)
            x = torch.rand((B, 128, 128), device=d); out = torch.empty((B, A, plan.PW), device=d)
            n = 50 if B * A < 20000 else 10
            tb = graph_time(lambda: plan.backward(g, out=gx), n) * 1e6
            tf = graph_time(lambda: plan.forward(x, out=out), n) * 1e6
            line += f"  B={B} f {tf:.1f} b {tb:.1f}"
        print(line, flush=True)
