"""Dose masks and sparse noisy measurements, on the device -- the step that feeds the projector's training loop.

Mirror of ``create_all_masks`` (ctvae/create_masks.py:16-103): same arguments, same files (``all_masks.npy``,
``all_proj_samples.npy`` under ``save_path``), same mask rules --
  * toy masks: rows [1,0],[0,1],[1,0],[0,1] tiled (:35-42);
  * random: the first ``num_sparse_angles`` of a shuffle of 0..A-1 (:51);
  * uniform: indices ``0, s, 2s, ...`` with ``s = ceil(A / num_sparse_angles)``, modulo A (:53-59);
  each mask row = one-hot sum / num_sparse_angles (:60-61);
  * measurements: ``Poisson(sino * mask * pnm) / pnm`` (:94-95), or the masked sinogram itself for real data (:83-84).
The masks' random draws come from a numpy generator (seed 0) and the Poisson counts from the library's counter-based
sampler (ctpvae_poisson_measure_f32, csrc/poisson.hip: Philox4x32-10 per element, fully specified, so a seed gives the
same counts on any device and the CPU oracle reproduces them), not from TensorFlow's generator: the distribution is the
reference's, the bits are not (nothing in the reference pins them).
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib

__all__ = ["create_all_masks", "sparse_angles", "poisson_measure"]


def poisson_measure(sino, masks, poisson_noise_multiplier, seed=0):
    """Poisson(max(sino, 0) * masks[..., None] * pnm) / pnm in ONE launch (ctvae/create_masks.py:80-103): sino [n][A][P]
    and masks [n][A] float32 on a HIP device.  There is no CPU path."""
    if sino.device.type != "cuda":
        raise _lib.RadonLibraryError(f"sinograms live on {sino.device}: the Poisson sampler runs on a HIP device only; "
                                     "there is no CPU path")
    lib = _lib.load()
    sino = sino.to(torch.float32).contiguous()
    masks = masks.to(device=sino.device, dtype=torch.float32).contiguous()
    if sino.dim() != 3 or tuple(masks.shape) != tuple(sino.shape[:2]):
        raise ValueError(f"need sinograms [n][A][P] and masks [n][A] (got {tuple(sino.shape)}, {tuple(masks.shape)})")
    out = torch.empty_like(sino)
    if sino.numel() == 0:
        return out
    from .forward_functions import _stream_ptr
    with torch.cuda.device(sino.device):
        _lib.check(lib.ctpvae_poisson_measure_f32(sino.data_ptr(), masks.data_ptr(), sino.shape[0], sino.shape[1], sino.shape[2],
                                                  ctypes.c_float(poisson_noise_multiplier), int(seed) & (2 ** 64 - 1),
                                                  out.data_ptr(), _stream_ptr()), "poisson_measure")
    return out


def sparse_angles(num_angles, num_sparse_angles, random=False, rng=None):
    """Indices of the angles one example is measured at (ctvae/create_masks.py:50-59)."""
    if random:
        rng = rng if rng is not None else np.random.default_rng(0)
        return rng.permutation(num_angles)[:num_sparse_angles].astype(np.int64)
    spacing = math.ceil(num_angles / num_sparse_angles)
    return (np.arange(0, spacing * num_sparse_angles, spacing) % num_angles).astype(np.int64)


def create_all_masks(x_train_sinograms=None, num_angles=None, save_path=None, poisson_noise_multiplier=1e3,
                     num_sparse_angles=10, random=False, reg=float(np.finfo(np.float32).eps), real_data=False,
                     train=False, truncate_dataset=100, toy_masks=False, device=None, seed=0, **kwargs):
    """Returns (all_masks [n][A], all_proj_samples [n][A][P]) as float32 tensors on `device` (default: the sinograms'
    device, or cuda:0 for numpy input).  ``train=False`` loads the two files instead of drawing them."""
    if device is None:
        device = x_train_sinograms.device if isinstance(x_train_sinograms, torch.Tensor) else torch.device("cuda", 0)
    device = torch.device(device)
    if not train:
        masks = np.load(os.path.join(save_path, "all_masks.npy"))
        samples = np.load(os.path.join(save_path, "all_proj_samples.npy"))
        return (torch.from_numpy(np.asarray(masks, np.float32)).to(device),
                torch.from_numpy(np.asarray(samples, np.float32)).to(device))

    sino = torch.as_tensor(x_train_sinograms)[:truncate_dataset].to(device=device, dtype=torch.float32)
    n = sino.shape[0]
    num_angles = int(num_angles if num_angles is not None else sino.shape[1])
    if sino.shape[1] != num_angles:
        raise ValueError(f"sinograms have {sino.shape[1]} angles, num_angles says {num_angles}")
    if toy_masks:
        if num_angles != 2:
            raise ValueError("toy masks are defined for two angles")
        base = np.array([[1, 0], [0, 1], [1, 0], [0, 1]], np.float32)
        masks_np = np.tile(base, (n // 4, 1))
        if len(masks_np) != n:
            raise ValueError("toy masks need a multiple of 4 examples")
    else:
        rng = np.random.default_rng(seed)
        masks_np = np.zeros((n, num_angles), np.float32)
        for k in range(n):
            np.add.at(masks_np[k], sparse_angles(num_angles, num_sparse_angles, random, rng), 1.0)   # one-hot sum
        masks_np /= num_sparse_angles
    masks = torch.from_numpy(masks_np).to(device)
    if real_data:
        samples = sino.clamp_min(0) * masks[..., None]          # :32 negatives to zero, :82-84 the masked sinogram itself
    else:
        samples = poisson_measure(sino, masks, poisson_noise_multiplier, seed)   # clamp, mask, pnm, draw, / pnm: one kernel
    if save_path is not None:
        os.makedirs(save_path, exist_ok=True)
        np.save(os.path.join(save_path, "all_masks.npy"), masks.cpu().numpy())
        np.save(os.path.join(save_path, "all_proj_samples.npy"), samples.cpu().numpy())
    return masks, samples
