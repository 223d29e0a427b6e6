"""Developer tool: how often do RANDOM angle sets (not the reference's pi k / A grids) overflow the step-coded plans?
A step code holds moves of 0 / 1 cell per canvas row; within ~1e-3 rad of an axis fp32 rounding can move a tap by two."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ct_pvae_amd.forward_functions import RotatePlan
d = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for N, A in ((128, 180), (512, 90)):
    fwd = bwd = 0
    for _ in range(n):
        theta = rng.uniform(0.0, 2 * np.pi, A)
        p = RotatePlan(theta, N, N, True, d, plan_format="compact")
        fwd += int((p._tplan is None) if p.tiled else (not p.compact))
        bwd += int(p._step_plan is None) if hasattr(p, "_step_plan") else 0
    print(f"{N}x{N}, {A} random angles, {n} sets: step-coded forward plan overflowed {fwd}, step plan of the adjoint {bwd}")
grid = RotatePlan(np.pi * np.arange(90) / 90, 512, 512, True, d)
print("the reference's grid, 512x512 x 90:", "tile plan", grid._tplan is not None, "step plan", grid._step_plan is not None)
