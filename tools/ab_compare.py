"""Developer tool: the hot launches timed on ONE box with one of two builds of the library, so that a before / after pair does
not ride on the difference between two boxes or two clock states (seen this round: up to 10 % between gpurun calls, and a
cold first measurement reads ~15 % slow).

    CTPVAE_VARIANT_LIB=tools/libctpvae_radon_<tag>.bin python tools/ab_compare.py      (an older build: same C ABI)
    python tools/ab_compare.py                                                          (the in-tree library)

Each figure: HIP-graph replay of N launches between two events, after three untimed warm-up rounds."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ct_pvae_amd import _lib

if os.environ.get("CTPVAE_VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CTPVAE_VARIANT_LIB"])
    _lib.torch_node = lambda: None      # the C++ autograd node binds the in-tree library: not used here
from ct_pvae_amd import phantoms
from ct_pvae_amd.forward_functions import RotatePlan

d = torch.device("cuda", 0)


def timed(body, n):
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(best))


print("library:", _lib.LIB_PATH, flush=True)
for B, A in ((50, 20), (50, 180), (400, 20), (400, 180)):
    theta = phantoms.dense_theta(180)[:: 180 // A]
    plan = RotatePlan(theta, 128, 128, True, d)
    x = torch.rand((B, 128, 128), device=d)
    gs = torch.randn((B, A, plan.PW), device=d)
    out, gx = torch.empty_like(gs), torch.empty_like(x)
    n = 200 if B * A < 20000 else 40
    print(f"128x128 B={B} A={A}: fwd {timed(lambda: plan.forward(x, out=out), n):.2f} us, adj {timed(lambda: plan.backward(gs, out=gx), n):.2f} us",
          flush=True)

# the training call: 10 objects, 20 of 180 angles, fused likelihood + per-object sums, scaled adjoint
theta = phantoms.dense_theta(180)
dense = RotatePlan(theta, 128, 128, True, d)
S = 10
x = torch.rand((S, 128, 128), device=d)
mask, meas = torch.full((S, 180), 0.05, device=d), torch.rand((S, 180, 184), device=d)
pnm = torch.tensor(1e4, device=d)
sub = torch.from_numpy(np.random.default_rng(0).permutation(180)[:20].astype(np.int32))   # host-resident: rides in the launch arguments
w = torch.ones(S, device=d)
sums, dlp = dense.forward_loglik_sums(x, mask, meas, pnm, 1e-7, angles_i=sub, dense_inputs=True)
gx = torch.empty_like(x)
print(f"training call S=10, 20 of 180: fwd + likelihood + sums {timed(lambda: dense.forward_loglik_sums(x, mask, meas, pnm, 1e-7, angles_i=sub, dense_inputs=True), 100):.2f} us, "
      f"adj {timed(lambda: dense.backward(dlp, out=gx, scale=w, angles_i=sub), 100):.2f} us", flush=True)

# config 5
theta = np.pi * np.arange(90) / 90
plan = RotatePlan(theta, 512, 512, True, d)
B = 32
x = torch.rand((B, 512, 512), device=d)
mask = torch.full((B, 90), 1.0 / 90, device=d)
meas = torch.rand((B, 90, plan.PW), device=d) * 3
up = torch.full((B,), -1.0 / B, device=d)
dlp = plan.forward_loglik_sums(x, mask, meas, pnm, 1e-7)[1]
gx = torch.empty_like(x)
out = torch.empty((B, 90, plan.PW), device=d)
print(f"512x512 B=32 A=90: fwd + likelihood + sums {timed(lambda: plan.forward_loglik_sums(x, mask, meas, pnm, 1e-7), 20):.1f} us, "
      f"plain fwd {timed(lambda: plan.forward(x, out=out), 20):.1f} us, adj {timed(lambda: plan.backward(dlp, out=gx, scale=up), 20):.1f} us", flush=True)
