"""Developer sweep: the ray-driven back-projector over runs per ray (SIDDON_BWD_CHUNKS) and threads, 50 x 180 x 184 -> 184^2."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ct_pvae_amd as cp
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.recon import siddon_backproject
d = torch.device("cuda", 0)
theta = phantoms.dense_theta(180)
sino = cp.create_sinograms(torch.from_numpy(phantoms.foam_batch(50, 128, seed=0, supersample=2)).to(d), theta, pad=True)
ref = None
for k in (2, 4, 8, 16, 32, 64):
    for th in (0, 512, 1024):
        _lib.tune("SIDDON_BWD_CHUNKS", k)
        _lib.tune("SIDDON_BWD_THREADS", th if th else -1)
        out = siddon_backproject(sino, theta)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            siddon_backproject(sino, theta)
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref = out
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f"chunks {k:2d} threads {th or 'auto':>4}: {e0.elapsed_time(e1) / 5:.3f} ms  (max rel diff vs chunks=2: {err:.1e})")
