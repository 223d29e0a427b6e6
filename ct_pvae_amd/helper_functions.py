"""Drop-in replacements for the hot-path pieces of ctvae/helper_functions.py of vganapati/CT_PVAE.

    create_sinogram(img, theta, pad=True)                       ctvae/helper_functions.py:33-38
    calculate_log_prob_M_given_R(output_sample, mask, proj_sample, poisson_noise_multiplier, sqrt_reg,
                                 theta=None, angles_i=None, pad=True)   ctvae/helper_functions.py:336-368
"""
import ctypes

import numpy as np
import torch

from . import forward_functions as _fwd  # noqa: E402  (NaN-poisoned outputs in test sessions)

from . import _lib, forward_functions
from .forward_functions import _cached_plan, _current_device, _stream_ptr, as_angle_index, project_tf_fast

__all__ = ["create_sinogram", "create_sinograms", "calculate_log_prob_M_given_R", "gaussian_poisson_log_prob"]


# ---------------------------------------------------------------------------------------------------------
# a7: create_sinogram -> tomopy.project(phantom[None], theta, center=None, emission=True, pad=pad,
#                                       sinogram_order=False), squeezed to [angles][dx]
# ---------------------------------------------------------------------------------------------------------
_SIDDON_TABLES = {}
_SIDDON_TABLES_MAX = 16


def _siddon_tables(theta, device):
    """(sin, cos, quadrant) device tables of an angle set.  Keyed by value and kept (a script projects image after image
    with one theta, scripts/images_to_sinograms.py:62-66): a repeated call uploads nothing, which is also what makes
    create_sinogram(s) capturable into a HIP graph after its first call."""
    lib = _lib.load()
    th = np.ascontiguousarray(np.asarray(theta.detach().cpu() if isinstance(theta, torch.Tensor) else theta,
                                         dtype=np.float32))  # tomopy: dtype.as_float32(theta)
    if th.ndim != 1 or th.size == 0:
        raise ValueError(f"theta must be a non-empty 1-D array (got shape {th.shape})")
    key = (th.tobytes(), str(device))
    hit = _SIDDON_TABLES.get(key)
    if hit is not None:
        return hit
    dt = th.size
    sin_t, cos_t = np.empty(dt, np.float32), np.empty(dt, np.float32)
    quad = np.empty(dt, np.int32)
    _lib.check(lib.ctpvae_siddon_tables_f32(th.ctypes.data, dt, sin_t.ctypes.data, cos_t.ctypes.data,
                                            quad.ctypes.data), "siddon_tables")
    tables = (torch.from_numpy(sin_t).to(device), torch.from_numpy(cos_t).to(device), torch.from_numpy(quad).to(device))
    if len(_SIDDON_TABLES) >= _SIDDON_TABLES_MAX:
        _SIDDON_TABLES.pop(next(iter(_SIDDON_TABLES)))
    _SIDDON_TABLES[key] = tables
    return tables


def _siddon_forward(obj, tables, dx, meas=None, rn2=None, out=None):
    """A obj: [oy][ox][oz] -> [oy][dt][dx] through ctpvae_siddon_fwd_ws_f32 (>= 3 slices: one walk of a ray serves 4 or 8 slices
    interleaved in a workspace; fewer: the LDS kernels).  With meas [oy][dt][dx] and rn2 [dt][dx]: SIRT's update factor
    (meas - A obj) / rn2 instead of the ray-sums."""
    lib = _lib.load()
    sin_t, cos_t, quad = tables
    oy, ox, oz = obj.shape
    dt = sin_t.numel()
    if out is None:
        out = _fwd._new_output((oy, dt, dx), torch.float32, obj.device)
    need = lib.ctpvae_siddon_fwd_workspace_bytes(oy, ox, oz)
    _lib.check(need, "siddon_fwd_workspace_bytes")
    ws = torch.empty(int(need), dtype=torch.uint8, device=obj.device) if need else None
    # center=None -> dx / 2 (tomopy.sim.project.get_center)
    _lib.check(lib.ctpvae_siddon_fwd_ws_f32(obj.data_ptr(), oy, ox, oz, sin_t.data_ptr(), cos_t.data_ptr(), quad.data_ptr(), dt, dx,
                                            ctypes.c_float(dx / 2.0), meas.data_ptr() if meas is not None else None,
                                            rn2.data_ptr() if rn2 is not None else None, ws.data_ptr() if ws is not None else None,
                                            out.data_ptr(), _stream_ptr()), "siddon_fwd")
    return out


def create_sinograms(imgs, theta, pad=True, device=None):
    """Batched create_sinogram: imgs [S][X][Y] -> [S][angles][dx] (fp32), TomoPy's ray-driven projector."""
    lib = _lib.load()
    as_numpy = not isinstance(imgs, torch.Tensor)
    t = torch.as_tensor(np.asarray(imgs, dtype=np.float32)) if as_numpy else imgs
    if t.dim() != 3:
        raise ValueError(f"expected slices x X x Y (got shape {tuple(t.shape)})")
    if t.device.type != "cuda":
        if device is None:
            if not torch.cuda.is_available():
                raise _lib.RadonLibraryError("create_sinogram needs a HIP device; there is no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        t = t.to(device)
    t = t.to(torch.float32).contiguous()
    oy, ox, oz = t.shape
    dx = lib.ctpvae_siddon_dx(ox, oz, 1 if pad else 0)
    tables = _siddon_tables(theta, t.device)
    if oy == 0:
        return np.empty((0, tables[0].numel(), dx), np.float32) if as_numpy else t.new_empty((0, tables[0].numel(), dx))
    with torch.cuda.device(t.device):
        data = _siddon_forward(t, tables, dx)
    return data.cpu().numpy() if as_numpy else data


def create_sinogram(img, theta, pad=True):
    """img [X][Y] -> sinogram [angles][dx]; numpy in -> numpy out, tensor in -> tensor out."""
    if isinstance(img, torch.Tensor):
        return create_sinograms(img[None], theta, pad=pad)[0]
    return create_sinograms(np.asarray(img)[None], theta, pad=pad)[0]


# ---------------------------------------------------------------------------------------------------------
# a8: log-likelihood of the measured sparse sinogram given a reconstruction
# ---------------------------------------------------------------------------------------------------------
class _GaussianPoissonLogProb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, proj, mask, x, pnm, eps):
        lib = _lib.load()
        for name, t in (("proj", proj), ("mask", mask), ("proj_sample", x), ("pnm", pnm)):
            if t.dtype is not torch.float32 or not t.is_contiguous():
                raise TypeError(f"{name} must be contiguous float32 (got {t.dtype}, contiguous={t.is_contiguous()})")
        B, A, P = proj.shape
        out = _fwd._new_output(proj.shape, proj.dtype, proj.device)
        with torch.cuda.device(proj.device):
            _lib.check(lib.ctpvae_loglik_fwd_f32(proj.data_ptr(), mask.data_ptr(), x.data_ptr(), B, A, P,
                                                 pnm.data_ptr(), ctypes.c_float(eps), out.data_ptr(),
                                                 _stream_ptr()), "loglik_fwd")
        ctx.save_for_backward(proj, mask, x, pnm)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        proj, mask, x, pnm = ctx.saved_tensors
        B, A, P = proj.shape
        gout = gout.contiguous()
        gproj = _fwd._new_output(proj.shape, proj.dtype, proj.device)
        gpnm = _fwd._new_output((), torch.float32, proj.device) if ctx.needs_input_grad[3] else None
        with torch.cuda.device(proj.device):
            _lib.check(lib.ctpvae_loglik_bwd_f32(proj.data_ptr(), mask.data_ptr(), x.data_ptr(), gout.data_ptr(),
                                                 B, A, P, pnm.data_ptr(), ctypes.c_float(ctx.eps),
                                                 gproj.data_ptr(), gpnm.data_ptr() if gpnm is not None else None,
                                                 _stream_ptr()), "loglik_bwd")
        return gproj, None, None, (gpnm.reshape(pnm.shape) if gpnm is not None else None), None


def gaussian_poisson_log_prob(proj, mask, proj_sample, poisson_noise_multiplier, sqrt_reg):
    """Normal(loc=proj*mask, scale=sqrt_reg + sqrt(loc/pnm + sqrt_reg)).log_prob(proj_sample), elementwise.

    proj, proj_sample [B][A][P]; mask [B][A]; poisson_noise_multiplier a python number or a 0-d/1-element tensor
    (it may require grad: --train_pnm)."""
    dev = proj.device
    if dev.type != "cuda":
        raise _lib.RadonLibraryError(f"proj lives on {dev}: the log-likelihood runs on a HIP device only; there is no CPU path")
    if proj.dim() != 3 or tuple(proj_sample.shape) != tuple(proj.shape) or tuple(mask.shape) != tuple(proj.shape[:2]):
        raise ValueError(f"need proj [B][A][P], mask [B][A], proj_sample [B][A][P] (got {tuple(proj.shape)}, "
                         f"{tuple(mask.shape)}, {tuple(proj_sample.shape)})")
    if mask.device != dev or proj_sample.device != dev:
        raise ValueError("proj, mask and proj_sample must live on the same device")
    if proj.numel() == 0:                         # an empty batch has an empty log-likelihood
        return proj * 0.0
    pnm = poisson_noise_multiplier
    if not isinstance(pnm, torch.Tensor):
        pnm = torch.tensor(float(pnm), dtype=torch.float32, device=dev)
    if pnm.numel() != 1:
        raise ValueError("poisson_noise_multiplier must be a number or a one-element tensor")
    pnm = pnm.to(device=dev, dtype=torch.float32)
    if not proj.dtype.is_floating_point:
        raise TypeError(f"proj must be floating point (got {proj.dtype})")
    # the kernels read and write 4-byte floats: every operand is made float32 here (a float64 / half projection from
    # project_tf_fast's dtype-preserving return would otherwise be read with the wrong element size); the result goes
    # back in the caller's dtype
    out = _GaussianPoissonLogProb.apply(proj.to(torch.float32).contiguous(), mask.to(torch.float32).contiguous(),
                                        proj_sample.to(torch.float32).contiguous(), pnm, float(sqrt_reg))
    return out if proj.dtype == torch.float32 else out.to(proj.dtype)


class _ProjectLogLik(torch.autograd.Function):
    """a2 + a8 in one launch: planned / tiled forward with the log-likelihood epilogue.

    Backward, when only the reconstruction needs a gradient: the epilogue also stored d lp / d sino, and the upstream
    gradient of a per-object sum (what find_loss_vae_unsup takes, ctvae/helper_functions.py:305-312: autograd hands it
    over as an expanded tensor, stride 0 over angles and bins) rides the projector's backward as a per-slice factor
    -- ONE launch, no [B][A][P] cotangent in HBM.  Any other upstream gradient multiplies dlp elementwise first; a
    trainable pnm (--train_pnm) takes the two-step backward (ctpvae_loglik_bwd_f32, which also reduces d/d pnm).

    angles_i (int32 device vector or None): the step's angle subset of the DENSE plan; mask / x are then the dense
    [B][A] / [B][A][P] arrays (dense_inputs) and nothing is gathered, rebuilt or re-planned per step."""

    @staticmethod
    def forward(ctx, sample, plan, mask, x, pnm, eps, angles_i=None):
        # sample: the caller's [B][X][Y][1] tensor (float32, contiguous); re-laid out here, as views, so that the
        # backward pass is this one node (no select_backward: a zero fill and a copy of the whole batch)
        slices = sample.view(sample.shape[0], sample.shape[1], sample.shape[2])
        ctx.plan, ctx.eps, ctx.angles_i = plan, eps, angles_i
        dense = angles_i is not None
        ctx.fused_bwd = ctx.needs_input_grad[0] and not ctx.needs_input_grad[4] and plan.supports_scale
        if ctx.fused_bwd:
            _, lp, dlp = plan.forward_loglik(slices, mask, x, pnm, eps, with_dlp=True, angles_i=angles_i, dense_inputs=dense)
            ctx.save_for_backward(dlp)
        else:
            sino, lp = plan.forward_loglik(slices, mask, x, pnm, eps, angles_i=angles_i, dense_inputs=dense)
            if dense:      # the two-step backward reads compact operands
                idx = angles_i.to(mask.device).long()
                mask, x = mask.index_select(1, idx).contiguous(), x.index_select(1, idx).contiguous()
            ctx.save_for_backward(sino, mask, x, pnm)
        return lp.unsqueeze(-1)

    @staticmethod
    def backward(ctx, gout):
        ai = ctx.angles_i
        gout = gout.squeeze(-1)
        if ctx.fused_bwd:
            dlp, = ctx.saved_tensors
            if gout.stride(1) == 0 and gout.stride(2) == 0:
                gimg = ctx.plan.backward(dlp, scale=gout[:, 0, 0], angles_i=ai)
            else:
                gimg = ctx.plan.backward(gout * dlp, angles_i=ai)
            return gimg.unsqueeze(-1), None, None, None, None, None, None
        lib = _lib.load()
        sino, mask, x, pnm = ctx.saved_tensors
        B, A, P = sino.shape
        gout = gout.contiguous()
        gproj = _fwd._new_output(sino.shape, sino.dtype, sino.device)
        gpnm = _fwd._new_output((), torch.float32, sino.device) if ctx.needs_input_grad[4] else None
        with torch.cuda.device(sino.device):
            _lib.check(lib.ctpvae_loglik_bwd_f32(sino.data_ptr(), mask.data_ptr(), x.data_ptr(), gout.data_ptr(),
                                                 B, A, P, pnm.data_ptr(), ctypes.c_float(ctx.eps),
                                                 gproj.data_ptr(), gpnm.data_ptr() if gpnm is not None else None,
                                                 _stream_ptr()), "loglik_bwd")
            gimg = ctx.plan.backward(gproj, angles_i=ai).unsqueeze(-1) if ctx.needs_input_grad[0] else None
        return gimg, None, None, None, (gpnm.reshape(pnm.shape) if gpnm is not None else None), None, None


class _ProjectLogLikSums(torch.autograd.Function):
    """a2 + a8 + the per-object reduce_sum of ctvae/helper_functions.py:305-312 as ONE node: forward = the projector launch
    that reduces the log-probabilities itself (plan.forward_loglik_sums) and stores only d lp / d ray-sum; backward = the
    projector's backward with the upstream gradient of the sums as its per-slice factor.  Reconstruction gradient only
    (a fixed pnm): a trainable pnm takes the two-step path."""

    @staticmethod
    def forward(ctx, sample, plan, mask, x, pnm, eps, angles_i=None):
        slices = sample.view(sample.shape[0], sample.shape[1], sample.shape[2])
        sums, dlp = plan.forward_loglik_sums(slices, mask, x, pnm, eps, angles_i=angles_i, dense_inputs=angles_i is not None,
                                             with_dlp=ctx.needs_input_grad[0])
        ctx.plan, ctx.angles_i = plan, angles_i
        if dlp is not None:
            ctx.save_for_backward(dlp)
        return sums

    @staticmethod
    def backward(ctx, gout):
        dlp, = ctx.saved_tensors
        scale = gout if gout.dtype is torch.float32 else gout.to(torch.float32)
        gimg = ctx.plan.backward(dlp, scale=scale, angles_i=ctx.angles_i)
        return gimg.unsqueeze(-1), None, None, None, None, None, None


class _ObjectSums(torch.autograd.Function):
    """lp [B][A][P][1] -> [B] in the library's fixed order (ctpvae_loglik_object_sums_f32); backward = a broadcast."""

    @staticmethod
    def forward(ctx, lp4, partition):
        lib = _lib.load()
        lp = lp4.reshape(lp4.shape[0], lp4.shape[1], lp4.shape[2]).to(torch.float32).contiguous()
        ctx.shape, ctx.dtype = tuple(lp4.shape), lp4.dtype
        out = _fwd._new_output((lp.shape[0],), torch.float32, lp.device)
        if lp.shape[0]:
            with torch.cuda.device(lp.device):
                _lib.check(lib.ctpvae_loglik_object_sums_f32(lp.data_ptr(), lp.shape[0], lp.shape[1], lp.shape[2], partition,
                                                             out.data_ptr(), _stream_ptr()), "loglik_object_sums")
        return out

    @staticmethod
    def backward(ctx, gout):
        return gout.to(ctx.dtype).view(-1, 1, 1, 1).expand(ctx.shape), None


def calculate_log_prob_M_given_R(output_sample, mask, proj_sample, poisson_noise_multiplier, sqrt_reg,
                                 theta=None, angles_i=None, pad=True, *, reduce=None):
    """ctvae/helper_functions.py:336-368.  output_sample [B][X][Y][1], mask [B][angles], proj_sample
    [B][angles][P]; returns the log-probabilities [B][angles_used][P][1].

    reduce="per_object" (keyword-only extension; default None = the reference's return value): returns the [B] per-object
    sums over angles and bins that find_loss_vae_unsup takes next (ctvae/helper_functions.py:305-312), reduced inside the
    projector launch on a compact plan -- the [B][A][P] sinogram and log-probabilities never reach HBM -- and
    always in the library's fixed summation order (oracle.loglik_object_sums), whichever path computes them.

    When the geometry takes the planned or the tiled forward (nearest) the projection and the log-probability are
    one launch (the same numbers, bit for bit, as project_tf_fast followed by gaussian_poisson_log_prob).

    angles_i (the step's random angle subset, :350-357): the plan, the transform tables and the gather plan are built
    ONCE for the whole `theta` (host-resident: on the host, so they are the bits the CPU oracle computes) and the kernels
    take `angles_i` as an index operand -- nothing is gathered or rebuilt per step, and mask / proj_sample are read at
    the selected angles by the kernel itself."""
    if reduce not in (None, "per_object"):
        raise ValueError(f"reduce must be None or 'per_object' (got {reduce!r})")
    x = output_sample
    fast = (isinstance(x, torch.Tensor) and x.dim() == 4 and x.shape[3] == 1 and x.device.type == "cuda"
            and x.dtype == torch.float32 and x.shape[0] > 0)
    if fast:
        if not x.is_contiguous():
            x = x.contiguous()
        slices = x       # (shapes below: [B][X][Y][1])
        plan = _cached_plan(theta, x.shape[1], x.shape[2], pad, x.device, "nearest", "tf_compat")
        if plan.planned[0] or plan.tiled:
            sel = None
            if angles_i is not None:
                sel = as_angle_index(angles_i, x.device, keep_host=True)   # host indices ride the launch arguments
                if not plan.planned[0] or sel.numel() > plan.MAX_SEL:     # tiled geometry: gather here, tables in subset()
                    idx = sel.to(x.device).long()
                    mask, proj_sample = mask.index_select(1, idx), proj_sample.index_select(1, idx)
                    plan, sel = plan.subset(sel), None
            n_in = plan.A
            pnm = poisson_noise_multiplier
            if not isinstance(pnm, torch.Tensor):
                pnm = torch.tensor(float(pnm), dtype=torch.float32, device=x.device)
            elif pnm.device != x.device or pnm.dtype is not torch.float32:
                pnm = pnm.to(device=x.device, dtype=torch.float32)
            if (tuple(mask.shape) != (slices.shape[0], n_in) or tuple(proj_sample.shape) != (slices.shape[0], n_in, plan.PW)
                    or mask.device != x.device or proj_sample.device != x.device or pnm.numel() != 1):
                raise ValueError(f"need mask [B][A] and proj_sample [B][A][P] = [{slices.shape[0]}][{n_in}][{plan.PW}] on "
                                 f"{x.device} (got {tuple(mask.shape)}, {tuple(proj_sample.shape)})")
            if mask.dtype is not torch.float32 or not mask.is_contiguous():
                mask = mask.to(torch.float32).contiguous()
            if proj_sample.dtype is not torch.float32 or not proj_sample.is_contiguous():
                proj_sample = proj_sample.to(torch.float32).contiguous()
            if reduce == "per_object":
                with torch.cuda.device(x.device):
                    if not pnm.requires_grad and plan.supports_scale:
                        return _ProjectLogLikSums.apply(x, plan, mask, proj_sample, pnm, float(sqrt_reg), sel)
                    lp4 = _ProjectLogLik.apply(x, plan, mask, proj_sample, pnm, float(sqrt_reg), sel)
                    return _ObjectSums.apply(lp4, 0 if plan.planned[0] else 1)
            if x.device.index == _current_device():
                if (forward_functions.USE_CPP_NODE and plan.planned[0] and x.requires_grad and not pnm.requires_grad
                        and torch.is_grad_enabled()):
                    out = plan.loglik_vae_cpp(x, mask, proj_sample, pnm, float(sqrt_reg), sel)
                    if out is not None:
                        return out
                return _ProjectLogLik.apply(x, plan, mask, proj_sample, pnm, float(sqrt_reg), sel)
            with torch.cuda.device(x.device):
                return _ProjectLogLik.apply(x, plan, mask, proj_sample, pnm, float(sqrt_reg), sel)
    if angles_i is not None:
        angles_i = torch.as_tensor(angles_i, device=output_sample.device).long()
        theta = torch.as_tensor(theta, device=output_sample.device)[angles_i].to(torch.float32)
        mask = mask[:, angles_i]
        proj_sample = proj_sample[:, angles_i]
    proj = project_tf_fast(output_sample, theta, pad=pad, dim=2, integrate_vae=True)
    logp = gaussian_poisson_log_prob(proj[..., 0], mask, proj_sample, poisson_noise_multiplier, sqrt_reg)
    if reduce == "per_object":
        return _ObjectSums.apply(logp.unsqueeze(-1), 0)
    return logp.unsqueeze(-1)
