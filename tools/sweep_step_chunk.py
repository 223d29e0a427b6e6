"""Developer timing: the stepped segment backward at 32 x 512 x 512 x 90 angles over its angle-chunk knob (SEG_CHUNK)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
def timed(body, n=20):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): body()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [g.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)
theta = np.pi * np.arange(90) / 90
plan = RotatePlan(theta, 512, 512, True, dev)
B = 32
g_ = torch.rand((B, 90, plan.PW), device=dev); gx = torch.empty((B, 512, 512), device=dev)
def eager(body, n=20):
    body(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [body() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
up = torch.full((B,), -1.0 / B, device=dev)
for _ in range(3): timed(lambda: plan.backward(g_, out=gx))   # clocks up
for ch in (-1, 48, 30, -1, 48, 30):
    _lib.tune("SEG_CHUNK", ch)
    print("SEG_CHUNK", ch, "graph %.1f us, eager %.1f us, eager scaled %.1f us" % (timed(lambda: plan.backward(g_, out=gx)), eager(lambda: plan.backward(g_, out=gx)), eager(lambda: plan.backward(g_, out=gx, scale=up))), flush=True)
for ch in (45, 32, 24, 23, 18, 16):
    _lib.tune("SEG_CHUNK", ch)
    print("SEG_CHUNK", ch, "bwd %.1f us" % timed(lambda: plan.backward(g_, out=gx)), flush=True)
