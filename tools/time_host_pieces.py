"""Developer check: where does the host time of one RotatePlan.forward call go?"""
import os, sys, timeit, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import phantoms, forward_functions as ff
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, 20)]
plan = RotatePlan(theta, 128, 128, True, dev)
x = torch.rand((50, 128, 128), device=dev); sino = torch.empty((50, 20, plan.PW), device=dev)
plan.forward(x, out=sino); torch.cuda.synchronize()
def t(f, n=100000): return timeit.timeit(f, number=n) / n * 1e6
print("current_device     %.2f us" % t(ff._current_device))
print("_check (cached)    %.2f us" % t(lambda: plan._check(x, (128, 128), "img")))
print("_stream_ptr        %.2f us" % t(lambda: ff._stream_ptr(0)))
print("data_ptr x3        %.2f us" % t(lambda: (x.data_ptr(), sino.data_ptr(), plan._fwd_plan.data_ptr())))
print("_tile_workspace    %.2f us" % t(lambda: plan._tile_workspace(50)))
n = 20000
torch.cuda.synchronize(); t0 = timeit.default_timer()
for _ in range(n): plan.forward(x, out=sino)
t1 = timeit.default_timer(); torch.cuda.synchronize()
print("forward (enqueue)  %.2f us" % ((t1 - t0) / n * 1e6))
lib, sp = plan._lib, ff._stream_ptr(0)
args = (x.data_ptr(), 50, 128, 128, plan.PH, plan.PW, plan.A, plan._fwd_plan.data_ptr(), sino.data_ptr(), sp)
torch.cuda.synchronize(); t0 = timeit.default_timer()
for _ in range(n): lib.ctpvae_rotate_fwd_planned_f32(*args)
t1 = timeit.default_timer(); torch.cuda.synchronize()
print("ctypes call alone  %.2f us" % ((t1 - t0) / n * 1e6))
