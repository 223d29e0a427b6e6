"""Developer timing: the 512 x 512 tile forward + reduce pass (with the log-likelihood epilogue, as config 5 runs it) against
the reduce pass's workgroup size (knob REDUCE_WAVES: waves of 64 bins per workgroup)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
if os.environ.get("CTPVAE_VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CTPVAE_VARIANT_LIB"])
    _lib.torch_node = lambda: None
print("library:", _lib.LIB_PATH)
from ct_pvae_amd.forward_functions import RotatePlan
d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
theta = np.pi * np.arange(90) / 90
plan = RotatePlan(theta, 512, 512, True, d)
x = torch.rand((B, 512, 512), device=d)
mask = torch.rand((B, 90), device=d) * 0.1 + 0.01
meas = torch.rand((B, 90, plan.PW), device=d)
pnm = torch.tensor([1e4], device=d)
def step():
    return plan.forward_loglik(x, mask, meas, pnm, 1.2e-7, with_dlp=True)
def timed(n=20):
    step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): step()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(r))
ref = [t.clone() for t in step()]
for rnd in range(2):
    for w in (0, 16, 6):
        if w: _lib.tune("REDUCE_WAVES", w)
        t = timed()
        same = all(torch.equal(a, b) for a, b in zip(step(), ref))
        _lib.tune("*")
        print(f"round {rnd} B={B} forward + reduce + log-lik, reduce waves {w or 'default':>7}: {t:7.2f} us  {'equal' if same else 'DIFFER'}", flush=True)
