"""Developer timing (round 4): per-object sums added inside the projector launch (knob FOLD_SUMS = 1) against the default's second launch: the training call (S objects, 20 of 180 angles) and config 5 (32 x 512 x 512, 90 angles)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
d = torch.device('cuda', 0)
def timed(body, n):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): body()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(r))
pnm = torch.tensor(1e4, device=d)
dense = RotatePlan(phantoms.dense_theta(180), 128, 128, True, d)
for S in (2, 10, 50):
    x = torch.rand((S, 128, 128), device=d)
    mask, meas = torch.full((S, 180), 0.05, device=d), torch.rand((S, 180, 184), device=d)
    sub = torch.from_numpy(np.random.default_rng(0).permutation(180)[:20].astype(np.int32))
    for rnd in range(2):
        for v in (0, 1):
            _lib.tune("FOLD_SUMS", 1 - v)
            t = timed(lambda: dense.forward_loglik_sums(x, mask, meas, pnm, 1e-7, angles_i=sub, dense_inputs=True), 100)
            _lib.tune("*")
            print(f"training call S={S:2d}, 20 of 180: forward + likelihood + sums, {'two launches' if v else 'one launch  '}: {t:6.2f} us", flush=True)
theta = np.pi * np.arange(90) / 90
plan = RotatePlan(theta, 512, 512, True, d)
x = torch.rand((32, 512, 512), device=d)
mask = torch.full((32, 90), 1.0 / 90, device=d)
meas = torch.rand((32, 90, plan.PW), device=d) * 3
for rnd in range(2):
    for v in (0, 1):
        _lib.tune("FOLD_SUMS", 1 - v)
        t = timed(lambda: plan.forward_loglik_sums(x, mask, meas, pnm, 1e-7), 20)
        _lib.tune("*")
        print(f"config 5 (32 x 512 x 512, 90 angles): forward + reduce + likelihood + sums, {'four launches ' if v else 'three launches'}: {t:6.1f} us", flush=True)
