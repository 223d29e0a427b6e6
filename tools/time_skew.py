"""Developer experiment ("LDS bank conflicts"): the u16 planned forward with the simulated lane skew
(knob SKEW0: row pitch == 0 mod 32, no mirroring, lane l delayed by round(alpha l) rows) against the current layout, on an
angle set inside the skew's favourable range (within 45 degrees of the row direction) and on the dataset's 180 angles.
Results must be equal bit for bit either way."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
def t_us(plan, x, out):
    plan.forward(x, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(100): plan.forward(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / 100)
    return float(np.median(r))
fav = np.deg2rad(np.concatenate([np.arange(1, 45, 2.2), 180 - np.arange(1, 45, 2.2)]))      # 40 angles, all favourable
sets = {"40 angles within 45 deg of the row direction": fav, "180 dense angles": np.pi * np.arange(180) / 180}
for name, theta in sets.items():
    for B in (50, 400):
        x = torch.rand((B, 128, 128), device=dev)
        _lib.tune("SKEW0", -1); p0 = RotatePlan(theta, 128, 128, True, dev, plan_format="u16")
        _lib.tune("SKEW0", 1); p1 = RotatePlan(theta, 128, 128, True, dev, plan_format="u16")
        o0, o1 = torch.empty((B, len(theta), p0.PW), device=dev), torch.empty((B, len(theta), p0.PW), device=dev)
        _lib.tune("SKEW0", -1); t0 = t_us(p0, x, o0)
        _lib.tune("SKEW0", 1); t1 = t_us(p1, x, o1)
        _lib.tune("SKEW0", -1)
        print(f"{name}, B={B}: current {t0:.2f} us, skewed {t1:.2f} us, equal={torch.equal(o0, o1)}", flush=True)
