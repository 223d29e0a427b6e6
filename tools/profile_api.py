"""Developer tool: where the host time of the drop-in API goes.  cProfile of N calls of
project_tf_fast(...).backward() and calculate_log_prob_M_given_R(...).backward() against the raw RotatePlan pair."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ct_pvae_amd as cp  # noqa: E402
from ct_pvae_amd import phantoms  # noqa: E402
from ct_pvae_amd.forward_functions import RotatePlan  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
d = torch.device("cuda", 0)
theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, 20)]
x4 = torch.rand((B, 128, 128, 1), device=d, requires_grad=True)
g4 = torch.rand((B, 20, 184, 1), device=d)
plan = RotatePlan(theta, 128, 128, True, d)
x3, g3 = x4.detach()[..., 0].contiguous(), g4[..., 0].contiguous()
sino, gimg = torch.empty((B, 20, 184), device=d), torch.empty((B, 128, 128), device=d)
mask = torch.full((B, 20), 0.05, device=d)
meas = torch.rand((B, 20, 184), device=d)
pnm = torch.tensor(1e4, device=d)
w = torch.ones(B, device=d)


def raw():
    plan.forward(x3, out=sino)
    plan.backward(g3, out=gimg)


def api():
    x4.grad = None
    cp.project_tf_fast(x4, theta, pad=True, dim=2, integrate_vae=True).backward(g4)


def api_lp():
    x4.grad = None
    lp = cp.calculate_log_prob_M_given_R(x4, mask, meas, pnm, 1e-7, theta=theta, pad=True)
    lp.sum(dim=(1, 2, 3)).backward(w)


def timeit(fn, name):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt * 1e6:.1f} us per call, {B * 20 / dt / 1e6:.2f} M projections/s", flush=True)


for fn, name in ((raw, "raw RotatePlan fwd+bwd"), (api, "project_tf_fast(...).backward()"),
                 (api_lp, "calculate_log_prob_M_given_R(...).sum.backward()")):
    timeit(fn, name)
from ct_pvae_amd import forward_functions as ff  # noqa: E402
ff.USE_CPP_NODE = False
timeit(api, "project_tf_fast(...).backward(), Python autograd node")
with torch.autograd.set_multithreading_enabled(False):
    timeit(api, "project_tf_fast(...).backward(), Python autograd node, single-threaded autograd engine")
ff.USE_CPP_NODE = True
with torch.autograd.set_multithreading_enabled(False):
    for fn, name in ((api, "project_tf_fast(...).backward(), single-threaded autograd engine"),
                     (api_lp, "calculate_log_prob_M_given_R(...).sum.backward(), single-threaded autograd engine")):
        timeit(fn, name)
for fn in (api, api_lp):
    with torch.autograd.set_multithreading_enabled(False):
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        pr.disable()
    print("=" * 30, fn.__name__, "(single-threaded engine: the backward's Python is visible)")
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
