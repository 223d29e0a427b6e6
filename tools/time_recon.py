"""Developer tool: timings of the setup-path kernels (TomoPy-style projector, its transpose, SIRT, Poisson sampler) at the
training set's size -- 50 slices, 180 angles, 184 bins, 184 x 184 reconstruction grid -- from HIP events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ct_pvae_amd as cp  # noqa: E402
from ct_pvae_amd import phantoms  # noqa: E402
from ct_pvae_amd.create_masks import poisson_measure  # noqa: E402
from ct_pvae_amd.recon import recon, siddon_backproject  # noqa: E402

d = torch.device("cuda", 0)
theta = phantoms.dense_theta(180)
imgs = torch.from_numpy(phantoms.foam_batch(50, 128, seed=0, supersample=2)).to(d)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


sino = cp.create_sinograms(imgs, theta, pad=True)
masks = torch.zeros((50, 180), device=d)
masks[:, ::9] = 1 / 20
sparse = sino * masks[..., None]
print(f"create_sinograms 50 x 180 x 184 (object grid 128^2): {timed(lambda: cp.create_sinograms(imgs, theta, pad=True)):.3f} ms")
print(f"siddon_backproject dense  -> 184^2: {timed(lambda: siddon_backproject(sino, theta)):.3f} ms")
print(f"siddon_backproject sparse (20 of 180 angles) -> 184^2: {timed(lambda: siddon_backproject(sparse, theta)):.3f} ms")
print(f"siddon_backproject dense  -> 128^2: {timed(lambda: siddon_backproject(sino, theta, 128, 128)):.3f} ms")
print(f"recon sirt num_iter=1: {timed(lambda: recon(sino, theta, sinogram_order=True, algorithm='sirt'), 5):.3f} ms")
print(f"recon sirt num_iter=20: {timed(lambda: recon(sino, theta, sinogram_order=True, algorithm='sirt', num_iter=20), 2):.3f} ms")
import warnings
warnings.simplefilter("ignore")
print(f"recon tv (stand-in) num_iter=1: {timed(lambda: recon(sino, theta, sinogram_order=True, algorithm='tv'), 5):.3f} ms")
print(f"recon tv (stand-in) num_iter=20: {timed(lambda: recon(sino, theta, sinogram_order=True, algorithm='tv', num_iter=20), 2):.3f} ms")
print(f"recon gridrec (parzen): {timed(lambda: recon(sino, theta, sinogram_order=True, algorithm='gridrec'), 5):.3f} ms")
print(f"poisson_measure 50 x 180 x 184, pnm 1e4: {timed(lambda: poisson_measure(sino, masks, 1e4, 0)):.3f} ms")
dense_mask = torch.full((50, 180), 0.05, device=d)
print(f"poisson_measure, every angle measured:   {timed(lambda: poisson_measure(sino, dense_mask, 1e4, 0)):.3f} ms")
