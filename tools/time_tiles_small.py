"""Developer timing: 128 x 128 slices through the TILE kernels (64 x 128 half-slice tiles, four slices per LDS cell, b128 gathers +
reduce pass; knob TILED_FORCE=1) against the whole-slice kernels (two slices per cell, b64 gathers)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
def t_us(plan, x, out, n=30):
    plan.forward(x, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): plan.forward(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(r))
for A in (90, 180):
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
    whole = RotatePlan(theta, 128, 128, True, dev)
    for B in (64, 128, 256, 400, 512):
        x = torch.rand((B, 128, 128), device=dev)
        o1, o2 = torch.empty((B, A, whole.PW), device=dev), torch.empty((B, A, whole.PW), device=dev)
        tw = t_us(whole, x, o1)
        res = []
        for G in (0, 1, 2, 3):
            with _lib.tuned("TILED_FORCE", 1):
                tiles = RotatePlan(theta, 128, 128, True, dev)
                assert tiles.tiled and tiles._tplan is not None
                if G: _lib.tune("TILED_G", G)
                res.append(t_us(tiles, x, o2))
                _lib.tune("TILED_G")
        print("A=%3d B=%3d  whole-slice %7.2f us   tiles (G auto,1,2,3) %s   max rel diff %.2e" % (A, B, tw, " ".join("%7.2f" % r for r in res), float(((o1 - o2).abs().max() / o1.abs().max()))), flush=True)
