"""On-disk formats and reconstruction metrics of CT_PVAE, so that datasets written by the reference's
scripts can be consumed and results scored the reference's way.  numpy / scipy host code.

    dataset_<name>/x_train_sinograms.npy      float  [n][angles][num_proj_pix]     scripts/images_to_sinograms.py:74
    dataset_<name>/dataset_parameters.npy     object [theta, num_proj_pix] (pickle)  scripts/images_to_sinograms.py:75-76
    dataset_<name>/x_size.npy, y_size.npy     ints                                   scripts/images_to_sinograms.py:78-79
    <save_path>/all_masks.npy, all_proj_samples.npy, all_input_encode.npy, reconstruction_final.npy
                                              ctvae/create_masks.py:70,101; ctvae/helper_functions.py:523; ctvae/main_ct_vae.py:452
"""
import os

import numpy as np
from scipy.ndimage import uniform_filter

__all__ = ["get_sinograms", "save_sinograms", "images_to_sinograms", "crop", "compare"]


def get_sinograms(save_path):
    """ctvae/helper_functions.py:50-56: (x_train_sinograms, theta, num_proj_pix)."""
    theta, num_proj_pix = np.load(os.path.join(save_path, "dataset_parameters.npy"), allow_pickle=True)
    x_train_sinograms = np.load(os.path.join(save_path, "x_train_sinograms.npy"))
    return x_train_sinograms, np.asarray(theta), int(num_proj_pix)


def save_sinograms(save_path, x_train_sinograms, theta, x_size, y_size):
    """The files scripts/images_to_sinograms.py:74-79 writes (np.object is gone from NumPy: plain `object`)."""
    os.makedirs(save_path, exist_ok=True)
    x_train_sinograms = np.asarray(x_train_sinograms)
    np.save(os.path.join(save_path, "x_train_sinograms.npy"), x_train_sinograms)
    params = np.empty(2, dtype=object)
    params[0], params[1] = np.asarray(theta), x_train_sinograms.shape[-1]
    np.save(os.path.join(save_path, "dataset_parameters.npy"), params, allow_pickle=True)
    np.save(os.path.join(save_path, "x_size.npy"), x_size)
    np.save(os.path.join(save_path, "y_size.npy"), y_size)


def images_to_sinograms(x_train_imgs, save_path, theta=None, pad=True, batch=256):
    """scripts/images_to_sinograms.py:61-79 with the TomoPy-style projector on the GPU: images [n][X][Y] in [0, 1] ->
    sinograms [n][angles][P], negatives clamped to 0 (:72), written in the reference's layout."""
    import torch

    from .helper_functions import create_sinograms
    theta = np.linspace(0, np.pi, 180, endpoint=False) if theta is None else np.asarray(theta)
    imgs = np.asarray(x_train_imgs, dtype=np.float32)
    out = []
    for k in range(0, imgs.shape[0], batch):
        s = create_sinograms(torch.from_numpy(imgs[k:k + batch]).cuda(), theta, pad=pad)
        out.append(s.clamp_min_(0).cpu().numpy())
    sino = np.concatenate(out, axis=0)
    save_sinograms(save_path, sino, theta, imgs.shape[1], imgs.shape[2])
    return sino


def crop(img_2d, final_x, final_y, ignore_dim_0=False):
    """ctvae/helper_functions.py:420-430 -- the package's one `crop` (arrays and tensors, any leading axes; ignore_dim_0 is
    accepted for the reference's signature: the last two axes are cropped either way)."""
    x, y = img_2d.shape[-2:]
    rx, ry = final_x % 2, final_y % 2
    return img_2d[..., x // 2 - final_x // 2:x // 2 + final_x // 2 + rx, y // 2 - final_y // 2:y // 2 + final_y // 2 + ry]


def _ssim(im1, im2, data_range, win_size=None):
    """skimage.metrics.structural_similarity (0.18 defaults: uniform window 7, K1 0.01, K2 0.03, sample covariance)."""
    im1, im2 = np.asarray(im1, np.float64), np.asarray(im2, np.float64)
    win = 7 if win_size is None else win_size
    if min(im1.shape) < win:
        raise ValueError("win_size exceeds image extent")
    npix = win ** im1.ndim
    cov_norm = npix / (npix - 1)
    ux, uy = uniform_filter(im1, size=win), uniform_filter(im2, size=win)
    uxx, uyy, uxy = uniform_filter(im1 * im1, size=win), uniform_filter(im2 * im2, size=win), uniform_filter(im1 * im2, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[tuple(slice(pad, n - pad) for n in s.shape)].mean())


def compare(recon0, recon1, verbose=True):
    """ctvae/helper_functions.py:394-418: (MSE, SSIM, PSNR) of recon1 against recon0."""
    recon0, recon1 = np.asarray(recon0, np.float64), np.asarray(recon1, np.float64)
    mse = float(np.mean((recon0 - recon1) ** 2))
    small = min(recon0.shape)
    win = (small if small % 2 else small - 1) if small < 7 else None
    rng = float(recon0.max() - recon0.min())
    ssim = _ssim(recon0, recon1, rng, win)
    psnr = float(10 * np.log10(rng * rng / mse))
    if verbose:
        print("MSE: {:.8f}, SSIM: {:.3f}, PSNR: {:.3f}".format(mse, ssim, psnr))
    return mse, ssim, psnr
