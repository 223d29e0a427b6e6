"""Where does the host time of project_tf_fast(...).backward() go?  (cProfile, run on the GPU box)"""
import cProfile, pstats, sys, io
import numpy as np, torch
sys.path.insert(0, '.')
from ct_pvae_amd import phantoms
from ct_pvae_amd.forward_functions import project_tf_fast
dev = torch.device('cuda', 0)
theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, 20)]
x4 = torch.rand((50, 128, 128, 1), device=dev, requires_grad=True)
g4 = torch.rand((50, 20, 184, 1), device=dev)
def step():
    x4.grad = None
    out = project_tf_fast(x4, theta, pad=True, dim=2, integrate_vae=True)
    out.backward(g4)
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3500])
