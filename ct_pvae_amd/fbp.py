"""Drop-in replacement for ctvae/fbp_tensorflow.py of vganapati/CT_PVAE: iradon (filtered back-projection).

    iradon(sinogram, theta, x_size, y_size, filter_1d)          ctvae/fbp_tensorflow.py:14-75

float64 throughout, as the reference runs it.  The Fourier-domain filter becomes a circular convolution with
Re(ifft(filter_1d)) inside one HIP kernel (see csrc/fbp.hip)."""
import numpy as np
import torch

from . import _lib
from .forward_functions import _stream_ptr

__all__ = ["iradon"]


def iradon(sinogram, theta, x_size, y_size, filter_1d):
    """sinogram [batch][angles][num_proj_pix] -> reconstruction [batch][x_size][y_size] (float64)."""
    lib = _lib.load()
    if not isinstance(sinogram, torch.Tensor) or sinogram.device.type != "cuda":
        raise _lib.RadonLibraryError("iradon expects a sinogram tensor on a HIP device; there is no CPU path")
    if sinogram.dim() != 3:
        raise ValueError(f"sinogram must be batch x angles x num_proj_pix (got {tuple(sinogram.shape)})")
    num_angles = len(theta)
    B, A, P = sinogram.shape
    if num_angles != A:
        # same exception type and wording as ctvae/fbp_tensorflow.py:43-45
        raise ValueError("The given ``theta`` does not match the number of projections in ``radon_image``.")
    dev = sinogram.device
    filt = np.asarray(filter_1d.detach().cpu() if isinstance(filter_1d, torch.Tensor) else filter_1d)
    filt = filt.reshape(-1)
    if filt.shape[0] != P:
        raise ValueError(f"filter_1d must hold num_proj_pix={P} values (got {filt.shape[0]})")
    hker = torch.from_numpy(np.ascontiguousarray(np.fft.ifft(filt.astype(np.complex128)).real)).to(dev)
    th = torch.as_tensor(theta).detach().to(device=dev, dtype=torch.float64)
    cos_t, sin_t = torch.cos(th).contiguous(), torch.sin(th).contiguous()
    sino = sinogram.to(torch.float64).contiguous()
    filtered = torch.empty_like(sino)
    recon = torch.empty((B, int(x_size), int(y_size)), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.ctpvae_fbp_filter_f64(sino.data_ptr(), B * A, P, hker.data_ptr(), filtered.data_ptr(),
                                             _stream_ptr()), "fbp_filter")
        _lib.check(lib.ctpvae_fbp_backproject_f64(filtered.data_ptr(), B, A, P, cos_t.data_ptr(), sin_t.data_ptr(),
                                                  int(x_size), int(y_size), recon.data_ptr(), _stream_ptr()),
                   "fbp_backproject")
    return recon
