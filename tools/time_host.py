"""Developer check: is the fwd+adj loop GPU-bound or host-bound?  (enqueue time vs completion time)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import phantoms
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
B, A = int(sys.argv[1]) if len(sys.argv) > 1 else 50, 20
theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)]
plan = RotatePlan(theta, 128, 128, True, dev)
x = torch.rand((B, 128, 128), device=dev); g = torch.rand((B, A, plan.PW), device=dev)
sino = torch.empty((B, A, plan.PW), device=dev); gimg = torch.empty_like(x)
for _ in range(50): plan.forward(x, out=sino); plan.backward(g, out=gimg)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n): plan.forward(x, out=sino); plan.backward(g, out=gimg)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: enqueue {1e6 * (t1 - t0) / n:.2f} us/step, complete {1e6 * (t2 - t0) / n:.2f} us/step")
