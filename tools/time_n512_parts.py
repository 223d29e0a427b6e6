"""Developer timing: config 5's forward + log-likelihood + per-object sums (tile kernel + reduce pass + ordered sum) and its
adjoint, per variant library:  CTPVAE_VARIANT_LIB=tools/libctpvae_radon_<tag>.bin python tools/time_n512_parts.py [B]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
if os.environ.get("CTPVAE_VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CTPVAE_VARIANT_LIB"])
    _lib.torch_node = lambda: None
from ct_pvae_amd.forward_functions import RotatePlan
d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
theta = np.pi * np.arange(90) / 90
plan = RotatePlan(theta, 512, 512, True, d)
x = torch.rand((B, 512, 512), device=d)
mask = torch.rand((B, 90), device=d) * 0.1 + 0.01
meas = torch.rand((B, 90, plan.PW), device=d)
pnm = torch.tensor([1e4], device=d)
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    r = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(r))
out = torch.empty((B, 90, plan.PW), device=d)
t_sums = timed(lambda: plan.forward_loglik_sums(x, mask, meas, pnm, 1.2e-7))
t_plain = timed(lambda: plan.forward(x, out=out))
ref = plan.forward_loglik_sums(x, mask, meas, pnm, 1.2e-7)
print(f"{os.path.basename(_lib.LIB_PATH):40s} B={B}: forward + log-lik + sums {t_sums:7.2f} us   plain forward (tiles + reduce) {t_plain:7.2f} us   "
      f"checksum {float(ref[0].double().sum()):.6e} {float(ref[1].double().abs().sum()):.6e}", flush=True)
