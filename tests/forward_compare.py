"""The reference's only check of its projectors, on the MI355X: ctvae/tomopy_forward_compare.py without the plots.

That script builds two 128 x 128 foam phantoms, runs project_tf_fast, project_tf_low_mem and tomopy.project at 100 angles,
prints the three wall times (:51-67), then reconstructs both sinograms with tomopy.recon(algorithm='sirt') and shows
their difference (:91-110).  Same calls here through the drop-in functions (first call = cold: tables, plans, kernels'
first launch; second = warm), with the agreement the reference only eyeballs printed as numbers, and the CPU restatement
(oracle/) of the TomoPy-style projector timed beside them as the script's `tomopy time` stand-in."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ct_pvae_amd as cp  # noqa: E402
from ct_pvae_amd import phantoms  # noqa: E402
from ct_pvae_amd.recon import recon  # noqa: E402
from oracle import radon_oracle as orc  # noqa: E402  (the checker: this script lives under tests/)

d = torch.device("cuda", 0)
theta = np.linspace(0, np.pi, 100, endpoint=False)
foam = phantoms.foam_batch(2, 128, seed=0, supersample=2)                 # xdesign is not installable: seeded stand-in
phantom = torch.from_numpy(np.transpose(foam, (1, 2, 0)).astype(np.float64)).to(d)   # [X][Y][2], float64 like xdesign's


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.time()
    out = fn()
    torch.cuda.synchronize()
    return out, time.time() - t0


for label in ("cold", "warm"):
    proj_fast, t_fast = wall(lambda: cp.project_tf_fast(phantom, theta, pad=True))
    proj_low, t_low = wall(lambda: cp.project_tf_low_mem(phantom, theta, pad=True))
    proj_tomo, t_tomo = wall(lambda: cp.create_sinograms(phantom.permute(2, 0, 1).float(), theta, pad=True).permute(1, 2, 0))
    print(f"[{label}] fast time: {t_fast:.6f} seconds | low memory time: {t_low:.6f} seconds | tomopy-style time: {t_tomo:.6f} seconds")
t0 = time.time()
cpu = orc.siddon_project(foam, theta, pad=True)
print(f"CPU restatement of tomopy.project, one thread: {time.time() - t0:.4f} seconds")
print("shapes", tuple(proj_fast.shape), tuple(proj_low.shape), tuple(proj_tomo.shape))


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


print(f"nearest (fast) vs bilinear (low memory) sinogram: max rel diff {rel(proj_fast, proj_low):.3e}")
print(f"rotate-and-sum (fast) vs ray-driven (tomopy-style): max rel diff {rel(proj_fast.float(), proj_tomo):.3e}; "
      f"bilinear vs ray-driven {rel(proj_low.float(), proj_tomo):.3e}")
print(f"tomopy-style on the GPU vs its CPU restatement: equal = "
      f"{np.array_equal(proj_tomo.permute(2, 0, 1).cpu().numpy(), np.swapaxes(cpu, 0, 1))}")
recon0 = recon(proj_fast.float().permute(0, 2, 1).contiguous(), theta, center=None, algorithm="sirt", sinogram_order=False, num_iter=50)
recon1 = recon(proj_tomo.permute(0, 2, 1).contiguous(), theta, center=None, algorithm="sirt", sinogram_order=False, num_iter=50)
lo = recon0.shape[1] // 2 - 64
c0, c1 = recon0[:, lo:lo + 128, lo:lo + 128], recon1[:, lo:lo + 128, lo:lo + 128]
truth = torch.from_numpy(foam).to(d)
print(f"SIRT (50 iterations) from the fast sinogram vs from the tomopy-style sinogram: rms difference {float(((c0 - c1) ** 2).mean().sqrt()):.4f}; "
      f"rms error vs the phantom {float(((c0 - truth) ** 2).mean().sqrt()):.4f} / {float(((c1 - truth) ** 2).mean().sqrt()):.4f}")
