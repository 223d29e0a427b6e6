"""Drop-in replacement for ctvae/fbp_tensorflow.py of vganapati/CT_PVAE: iradon (filtered back-projection).

    iradon(sinogram, theta, x_size, y_size, filter_1d)          ctvae/fbp_tensorflow.py:14-75

float64 throughout, as the reference runs it.  The Fourier-domain filter becomes a circular convolution with
Re(ifft(filter_1d)) inside one HIP kernel (see csrc/fbp.hip)."""
import numpy as np
import torch

from . import forward_functions as _fwd  # noqa: E402  (NaN-poisoned outputs in test sessions)

from . import _lib
from .forward_functions import _stream_ptr

__all__ = ["iradon", "iradon_all", "ramp_filter"]


_CACHE = {}
_CACHE_MAX = 32


def _cached(key, make):
    """Small keyed store for the per-filter / per-angle-set device tables (the filter kernel Re(ifft(filter_1d)) and
    cos/sin of theta): the reference recomputes them inside every call; here a repeated call costs two launches."""
    val = _CACHE.get(key)
    if val is None:
        val = make()
        if len(_CACHE) >= _CACHE_MAX:
            _CACHE.pop(next(iter(_CACHE)))
        _CACHE[key] = val
    return val


def iradon(sinogram, theta, x_size, y_size, filter_1d, *, tomopy_geometry=False):
    """sinogram [batch][angles][num_proj_pix] -> reconstruction [batch][x_size][y_size] (float64).

    Differentiable with respect to the sinogram (like the reference's TF-op version; nothing in the reference takes that
    gradient).

    tomopy_geometry (keyword-only extension): sample the sinogram on tomopy's ray-driven grid -- pixel centres at half-
    integers, detector bin d at d - (P - 1) / 2 -- instead of the reference iradon's (pixel i at i - X / 2, sample k at
    k - P / 2): the right geometry for sinograms made by create_sinogram / tomopy.project."""
    _lib.load()
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
    hker = _cached(("hker", filt.tobytes(), str(filt.dtype), str(dev)), lambda: torch.from_numpy(
        np.ascontiguousarray(np.fft.ifft(filt.astype(np.complex128)).real)).to(dev))
    if isinstance(theta, torch.Tensor) and theta.device.type == "cuda":
        th = theta.detach().to(device=dev, dtype=torch.float64)
        cos_t, sin_t = torch.cos(th).contiguous(), torch.sin(th).contiguous()
    else:   # host-resident angle set: the tables are made once
        th_np = np.ascontiguousarray(np.asarray(theta.detach().cpu() if isinstance(theta, torch.Tensor) else theta,
                                                dtype=np.float64))

        def make():
            th = torch.from_numpy(th_np).to(dev)
            return torch.cos(th).contiguous(), torch.sin(th).contiguous()

        cos_t, sin_t = _cached(("trig", th_np.tobytes(), str(dev)), make)
    X, Y = int(x_size), int(y_size)
    geom = ((X - 1) / 2.0, (Y - 1) / 2.0, (P - 1) / 2.0) if tomopy_geometry else (X / 2.0, Y / 2.0, P / 2.0)
    if sinogram.requires_grad and torch.is_grad_enabled():
        # the transposed filter: hker reversed, hker[(P - n) % P]
        hker_t = _cached(("hker_t", filt.tobytes(), str(filt.dtype), str(dev)), lambda: torch.roll(torch.flip(hker, (0,)), 1, 0).contiguous())
        return _IRadon.apply(sinogram, hker, hker_t, cos_t, sin_t, X, Y, geom)
    return _iradon_forward(sinogram, hker, cos_t, sin_t, X, Y, geom)


def _iradon_forward(sinogram, hker, cos_t, sin_t, X, Y, geom):
    lib = _lib.load()
    B, A, P = sinogram.shape
    dev = sinogram.device
    sino = sinogram.to(torch.float64).contiguous()
    filtered = _fwd._new_output(sino.shape, sino.dtype, sino.device)
    recon = _fwd._new_output((B, X, Y), torch.float64, dev)
    with torch.cuda.device(dev):
        _lib.check(lib.ctpvae_fbp_filter_f64(sino.data_ptr(), B * A, P, hker.data_ptr(), filtered.data_ptr(),
                                             _stream_ptr()), "fbp_filter")
        _lib.check(lib.ctpvae_fbp_backproject_geom_f64(filtered.data_ptr(), B, A, P, cos_t.data_ptr(), sin_t.data_ptr(), X, Y,
                                                       geom[0], geom[1], geom[2], recon.data_ptr(), _stream_ptr()), "fbp_backproject")
    return recon


class _IRadon(torch.autograd.Function):
    """iradon with its gradient with respect to the sinogram (the reference's iradon is TF ops, hence differentiable):
    backward = transposed back-projection (ctpvae_fbp_backproject_bwd_f64, ordered fp64 sums) then the transposed filter
    (the same circular convolution with the kernel reversed)."""

    @staticmethod
    def forward(ctx, sinogram, hker, hker_t, cos_t, sin_t, X, Y, geom):
        ctx.save_for_backward(hker_t, cos_t, sin_t)
        ctx.shape, ctx.geom, ctx.in_dtype = (tuple(sinogram.shape), X, Y), geom, sinogram.dtype
        return _iradon_forward(sinogram, hker, cos_t, sin_t, X, Y, geom)

    @staticmethod
    def backward(ctx, grecon):
        lib = _lib.load()
        hker_t, cos_t, sin_t = ctx.saved_tensors
        (B, A, P), X, Y = ctx.shape
        g = grecon.to(torch.float64).contiguous()
        gfilt = _fwd._new_output((B, A, P), torch.float64, g.device)
        gsino = _fwd._new_output(gfilt.shape, gfilt.dtype, gfilt.device)
        with torch.cuda.device(g.device):
            _lib.check(lib.ctpvae_fbp_backproject_bwd_f64(g.data_ptr(), B, A, P, cos_t.data_ptr(), sin_t.data_ptr(), X, Y,
                                                          ctx.geom[0], ctx.geom[1], ctx.geom[2], gfilt.data_ptr(), _stream_ptr()),
                       "fbp_backproject_bwd")
            _lib.check(lib.ctpvae_fbp_filter_f64(gfilt.data_ptr(), B * A, P, hker_t.data_ptr(), gsino.data_ptr(), _stream_ptr()),
                       "fbp_filter")
        return gsino.to(ctx.in_dtype), None, None, None, None, None, None, None


def ramp_filter(P):
    """The ramp filter of skimage.transform.radon_transform._get_fourier_filter(P, 'ramp') (scikit-image 0.18, squeezed)
    -- the filter the reference's own iradon call sites pass (ctvae/main_ct_vae.py:181-190, commented) -- for any even P:
    2 Re(fft(h)) of the band-limited ramp's kernel h[0] = 1/4, h[k] = -1 / (pi n)^2 at odd circular distance
    n = min(k, P - k), 0 elsewhere.  skimage builds the same array when P / 2 is even (its sizes are powers of two:
    P = 184 for 128 x 128 objects); for odd P / 2 (P = 94 for 64 x 64) its index arithmetic mis-places half the taps
    and the reconstruction comes out ~8x too large, so the distances are written out here."""
    k = np.arange(P)
    n = np.minimum(k, P - k)
    h = np.where(n % 2 == 1, -1.0 / (np.pi * np.maximum(n, 1)) ** 2, 0.0)
    h[0] = 0.25
    return 2 * np.real(np.fft.fft(h))


def iradon_all(all_proj_samples, all_masks, num_proj_pix, theta, algorithms, sqrt_reg, x_size, y_size, save_path=None,
               train=False, **kwargs):
    """Initial reconstructions that feed the encoder (ctvae/helper_functions.py:477-529): one channel per entry of
    `algorithms` -- tomopy.recon(proj_sample_expand, theta, center=None, sinogram_order=True, algorithm=...) cropped to
    x_size x y_size (:503-505) -- plus the un-filtered back-projection of the dose mask (algorithm='fbp',
    filter_name='none', :514-515).  The reconstructions run on the GPU (ct_pvae_amd/recon.py: 'fbp', 'sirt' on the
    TomoPy-style operator pair, 'gridrec' = libtomo's gridrec.c on csrc/gridrec.hip, 'tv' as a flagged
    Chambolle-Pock stand-in).
    Returns [n][x_size][y_size][len(algorithms) + 1] float32 on the sinograms' device and, like the reference, writes /
    reads ``all_input_encode.npy`` under `save_path`."""
    import os
    from .recon import crop, recon
    if not train:
        arr = np.load(os.path.join(save_path, "all_input_encode.npy"))
        dev = all_proj_samples.device if isinstance(all_proj_samples, torch.Tensor) else torch.device("cuda", 0)
        return torch.from_numpy(np.asarray(arr, np.float32)).to(dev)
    P = int(num_proj_pix)
    mask_expand = all_masks[..., None].expand(-1, -1, P)
    expand = torch.where(mask_expand > sqrt_reg, all_proj_samples / mask_expand.clamp_min(1e-30), all_proj_samples)
    chans = [crop(recon(expand.contiguous(), theta, center=None, sinogram_order=True, algorithm=alg), x_size, y_size,
                  ignore_dim_0=True) for alg in algorithms]
    chans.append(crop(recon(mask_expand.contiguous(), theta, center=None, sinogram_order=True, algorithm="fbp",
                            filter_name="none"), x_size, y_size, ignore_dim_0=True))
    out = torch.stack(chans, dim=-1).to(torch.float32)
    if save_path is not None:
        os.makedirs(save_path, exist_ok=True)
        np.save(os.path.join(save_path, "all_input_encode.npy"), out.cpu().numpy())
    return out
