"""Drop-in replacements for ctvae/forward_functions.py of vganapati/CT_PVAE on MI355X.

Same names, argument order, defaults and tensor layouts as the reference:

    pad_phantom(phantom, dim=3, integrate_vae=False)                        ctvae/forward_functions.py:18-46
    project_tf_low_mem(phantom, theta, pad=False)                            ctvae/forward_functions.py:49-78
    project_tf_fast(phantom, theta, pad=False, dim=3, integrate_vae=False)   ctvae/forward_functions.py:80-123

Inputs and outputs are torch tensors on a HIP device; the work is done by hand-written gfx950 kernels behind
the C ABI of include/ctpvae_radon.h (no TensorFlow, no per-angle image copies, the zero padding is never
materialised).  Both projectors are differentiable; keyword-only extensions select behaviour the reference
gets implicitly from TensorFlow:

    interp   = "nearest" | "bilinear"   tfa.image.rotate's interpolation (the reference's fast path uses
                                        the tfa default "nearest", the low-memory path asks for "bilinear")
    backward = "tf_compat" | "exact"    "tf_compat" is what tf.GradientTape computes for the reference
                                        (ctvae/main_ct_vae.py:471-481: the incoming gradient is re-sampled with
                                        the inverted transform); "exact" is the true transpose of the forward.
"""
import math
import os

import ctypes
import weakref

import numpy as np
import torch

from . import _lib

__all__ = ["pad_phantom", "project_tf_fast", "project_tf_low_mem", "num_proj_pix", "pad_amounts", "RotatePlan",
           "rotate_tables", "as_angle_index"]

_current_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)
_INTERP = {"nearest": _lib.NEAREST, "bilinear": _lib.BILINEAR}
_BACKWARD = {"tf_compat": _lib.BWD_TF_COMPAT, "exact": _lib.BWD_EXACT}


# ---------------------------------------------------------------------------------------------------------
# a1: size rule of pad_phantom (host arithmetic, ctvae/forward_functions.py:29-36)
# ---------------------------------------------------------------------------------------------------------

# Test aid: outputs the kernels must fill completely come NaN-poisoned instead of uninitialised (tests/conftest.py switches it
# on).  torch's caching allocator hands a freshly freed block straight back, so a launch that SKIPS part of its output would
# otherwise be "checked" against the previous, correct result still lying in that memory.
POISON_OUTPUTS = bool(os.environ.get("CTPVAE_POISON_OUTPUTS"))


def _new_output(shape, dtype, device):
    if POISON_OUTPUTS:
        return torch.full(tuple(shape), float("nan"), dtype=dtype, device=device)
    return torch.empty(tuple(shape), dtype=dtype, device=device)


def num_proj_pix(img_size_x, img_size_y):
    """P = int(ceil((sqrt(float64(Nx^2 + Ny^2)) + 2) / 2) * 2)."""
    return int(math.ceil((math.sqrt(float(img_size_x ** 2 + img_size_y ** 2)) + 2.0) / 2.0) * 2)


def pad_amounts(n, P):
    """(lo, hi) with lo = (P-n)//2 and the odd remainder on the high side."""
    lo = (P - n) // 2
    return lo, lo + (P - n) % 2


def pad_phantom(phantom, dim=3, integrate_vae=False):
    """Zero-pad the two spatial axes to P x P (materialised, like the reference's tf.pad).

    The projectors below do NOT call this: they fold the padding into their bounds test.
    """
    if integrate_vae:
        nx, ny = phantom.shape[1], phantom.shape[2]
    else:
        nx, ny = phantom.shape[0], phantom.shape[1]
    P = num_proj_pix(nx, ny)
    (x_lo, x_hi), (y_lo, y_hi) = pad_amounts(nx, P), pad_amounts(ny, P)
    # torch.nn.functional.pad lists the LAST axis first
    if integrate_vae:
        pads = (0, 0, y_lo, y_hi, x_lo, x_hi)
    elif dim == 3:
        pads = (0, 0, y_lo, y_hi, x_lo, x_hi)
    elif dim == 2:
        pads = (y_lo, y_hi, x_lo, x_hi)
    else:
        raise ValueError(f"pad_phantom: dim must be 2 or 3 (got {dim})")
    return torch.nn.functional.pad(phantom, pads, mode="constant", value=0.0)


# ---------------------------------------------------------------------------------------------------------
# Plan: geometry + transform tables on the device + bound C entry points
# ---------------------------------------------------------------------------------------------------------
_TABLE_CACHE = {}
_TABLE_CACHE_MAX = 64


def _stream_ptr(device_index=None):
    """Raw hipStream_t of torch's current stream (the fast private accessor when it exists: the public
    torch.cuda.current_stream() costs ~10 us per call, more than the kernels it precedes)."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if device_index is None else device_index)
    except AttributeError:
        return torch.cuda.current_stream().cuda_stream


def _theta_to_device(theta, device):
    """Returns (theta_dev fp32 [A], cache_key or None).  `theta` is anything 1-D with a length: a list, a numpy
    array (the reference's scripts) or a tensor (the training loop gathers it on the device,
    ctvae/helper_functions.py:355)."""
    if isinstance(theta, torch.Tensor):
        if theta.dim() != 1:
            raise ValueError(f"theta must be 1-D (got shape {tuple(theta.shape)})")
        if theta.device.type == "cuda":
            return theta.detach().to(device=device, dtype=torch.float32).contiguous(), None
        host = theta.detach().cpu().numpy()
    else:
        host = np.asarray(theta)
    if host.ndim != 1:
        raise ValueError(f"theta must be 1-D (got shape {host.shape})")
    host32 = np.ascontiguousarray(host.astype(np.float32))  # tfa converts angles to float32
    return host32, host32.tobytes()


def rotate_tables(theta, H, W, device):
    """Device tables (T8, Tinv8), each [A][8] fp32, for rotating an H x W canvas by -theta (a3/a4).

    A host-resident angle set (list / numpy / CPU tensor -- the dataset's theta, ctvae/main_ct_vae.py:152) is evaluated
    ON THE HOST (ctpvae_rotate_transforms_host_f32) and uploaded: the kernels then see the very bits any other host code
    computes from the same expressions, cached by value.  A device-resident theta costs one tiny kernel launch
    (device libm: may differ from the host table in rare 1-ulp cases)."""
    lib = _lib.load()
    th, key = _theta_to_device(theta, device)
    if key is not None:
        key = (key, H, W, str(device))
        hit = _TABLE_CACHE.get(key)
        if hit is not None:
            return hit
        A = th.size
        if A == 0:
            raise ValueError("theta is empty")
        host = np.empty((2, A, 8), dtype=np.float32)
        _lib.check(lib.ctpvae_rotate_transforms_host_f32(th.ctypes.data, A, H, W, host[0].ctypes.data, host[1].ctypes.data),
                   "rotate_transforms_host")
        tables = torch.from_numpy(host).to(device)
        out = (tables[0], tables[1])
        if len(_TABLE_CACHE) >= _TABLE_CACHE_MAX:
            _TABLE_CACHE.pop(next(iter(_TABLE_CACHE)))
        _TABLE_CACHE[key] = out
        return out
    A = th.numel()
    if A == 0:
        raise ValueError("theta is empty")
    tables = torch.empty((2, A, 8), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.ctpvae_rotate_transforms_f32(th.data_ptr(), A, H, W, tables[0].data_ptr(),
                                                    tables[1].data_ptr(), _stream_ptr()), "rotate_transforms")
    return tables[0], tables[1]


def as_angle_index(angles_i, device, keep_host=False):
    """The per-step angle subset (ctvae/helper_functions.py:350-357) as the int32 vector the kernels take: on `device`, or --
    keep_host -- left in HOST memory when that is where the caller has it (a list, a numpy array, a CPU tensor: the training
    loop draws its subset on the host, ctvae/helper_functions.py:104-107).  Kernels that accept host indices carry them in
    their launch arguments: no upload, nothing on the stream in front of the projector."""
    t = torch.as_tensor(angles_i) if not isinstance(angles_i, torch.Tensor) else angles_i
    if t.dim() != 1 or t.numel() == 0:
        raise ValueError(f"angles_i must be a non-empty 1-D index vector (got shape {tuple(t.shape)})")
    if t.dtype.is_floating_point or t.dtype is torch.bool:
        raise TypeError(f"angles_i must hold integers (got {t.dtype})")
    if keep_host and t.device.type == "cpu":
        return t if (t.dtype is torch.int32 and t.is_contiguous()) else t.to(torch.int32).contiguous()
    if t.device != device or t.dtype is not torch.int32 or not t.is_contiguous():
        t = t.to(device=device, dtype=torch.int32).contiguous()
    return t


class RotatePlan:
    """Geometry and tables of one rotate-and-sum projector: slices [S][H][W] -> sinograms [S][A][PW].

    `forward` / `backward` are the raw operator pair (no autograd bookkeeping); `apply` is differentiable."""

    MAX_SEL = 256   # angles per subset launch of a dense plan (kSelRounds * 64 in rotate_plan.hip)
    # Forward plan format (nearest).  "u16": 2 B per sample (csrc/rotate_plan.hip); "compact": first tap + 2 bits per row,
    # 1/3 B per sample (csrc/rotate_cplan.hip) -- the same bits either way.  "auto" takes what was measured faster on the
    # MI355X (round 3, tools/time_compact_shapes.py, tools/time_sel.py): the compact plan for many-angle plans, whose 17 MB
    # of u16 taps at the dataset's 180 angles overflow an XCD's 4 MB L2 (B=400 A=180: 145 vs 155 us; a 20-angle subset of the
    # dense plan, the training call: 7.8 vs 8.9 us), the u16 plan for few angles, where its taps are L2-resident anyway and
    # its shorter per-task prologue wins (B=50 A=20: 7.3 vs 7.65 us).
    COMPACT_MIN_ANGLES = 64
    # Round 4: "auto" decides PER LAUNCH.  A many-angle plan keeps the compact form for what only it has (host-resident angle
    # subsets, per-object sums inside the launch) and builds the u16 plan beside it -- on the first dense launch -- because that one
    # wins the dense shapes of tools/time_compact_shapes.py (profiles/r04_time_compact_shapes.txt: A = 180: B = 50 20.6 vs 21.4 us,
    # B = 200 73 vs 78; A = 90, B = 400 79 vs 87), since round 4 also between 16 and 32 slices, where its angles are dealt to the
    # XCDs (A = 180, B = 16: 10.5 vs 12.2 us; the compact plan won that window in round 3: 12.1 vs 17.1).  Batch sizes inside
    # COMPACT_DENSE_WINDOW (first, last) keep the compact plan for dense launches:
    # Round 5: near-ties go to the format that moves fewer bytes.  At 180 angles the u16 plan streams 92 MB per launch through
    # the fabric (9.3x the algorithmic bytes, profiles/r04_angles180_traffic_pmc.json), the compact one 28.5 MB (2.9x); where
    # the two are within 3 % in time (profiles/r05_time_compact_shapes.txt: B = 40: 19.3 vs 19.6 us; B = 50: 22.2 vs 21.3 and
    # B = 32: 16.2 vs 14.6 are not) the compact plan runs -- on one GPU the same time, under eight ranks' traffic the cheaper one.
    COMPACT_DENSE_WINDOW = (36, 44)

    def __init__(self, theta, H, W, pad, device, interp="nearest", backward="tf_compat", use_plan=True, _tables=None,
                 plan_format="auto"):
        if plan_format not in ("auto", "compact", "u16"):
            raise ValueError(f"plan_format must be 'auto', 'compact' or 'u16' (got {plan_format!r})")
        if interp not in _INTERP:
            raise ValueError(f"interp must be one of {sorted(_INTERP)} (got {interp!r})")
        if backward not in _BACKWARD:
            raise ValueError(f"backward must be one of {sorted(_BACKWARD)} (got {backward!r})")
        self.H, self.W = int(H), int(W)
        if pad:
            P = num_proj_pix(self.H, self.W)
            self.PH = self.PW = P
            self.py, self.px = pad_amounts(self.H, P)[0], pad_amounts(self.W, P)[0]
        else:
            self.PH, self.PW, self.py, self.px = self.H, self.W, 0, 0
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RadonLibraryError(
                f"the projector runs on a HIP device only (got a tensor on {self.device}); there is no CPU path")
        self.interp, self.mode = _INTERP[interp], _BACKWARD[backward]
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._interp_name, self._backward_name, self._pad = interp, backward, bool(pad)
        self.T8, self.Tinv8 = _tables if _tables is not None else rotate_tables(theta, self.PH, self.PW, self.device)
        self._tdev = self.T8.device
        self.A = self.T8.shape[0]
        self._lib = _lib.load()
        # Gather plans (NEAREST): tap indices computed once for this geometry, shared by every slice of every call.
        # The forward plan is built now, the backward plan on the first backward.
        self._fwd_plan = self._bwd_plan = self._exact_plan = self._bwd4_plan = None
        self._want_bwd_plan = False
        # angle subsets of the backward (the training call): a plan whose layout lets a launch select angles, built on
        # the first subset backward
        self._want_bwd4 = bool(use_plan and self.interp == _lib.NEAREST and self.mode == _lib.BWD_TF_COMPAT)
        # exact transpose (nearest): a deterministic gather through an inverse plan
        self._want_exact_plan = bool(use_plan and self.mode == _lib.BWD_EXACT and self.interp == _lib.NEAREST)
        self._use_tiles = bool(use_plan)   # slices larger than LDS: tiled forward (workspace grown on demand)
        self._tile_ws = None
        self._compact = False      # the forward plan is a compact (step-coded) one: ctpvae_rotate_fwd_compact_f32 runs it
        self._part_ws = {}         # forward_loglik_sums: partial sums + arrival counters per (slices, angles, partition)
        self._u16_plan = None      # "auto", many angles: the u16 plan of the same geometry for dense launches (built on demand)
        self._auto_dense_u16 = False
        if use_plan:
            geo = (self.H, self.W, self.PH, self.PW, self.A, self.interp)
            want_compact = plan_format == "compact" or (plan_format == "auto" and self.A >= self.COMPACT_MIN_ANGLES)
            if want_compact and self._lib.ctpvae_rotate_cplan_supported(*geo):
                self._fwd_plan = self._build_compact_plan()
                self._compact = self._fwd_plan is not None
            if self._fwd_plan is None and self._lib.ctpvae_rotate_plan_supported(*geo, 0):
                self._fwd_plan = self._build_plan(0)
            self._auto_dense_u16 = bool(plan_format == "auto" and self._compact and self._lib.ctpvae_rotate_plan_supported(*geo, 0))
            if self._auto_dense_u16:
                # built now, not on the first dense launch: a first launch inside a caller's HIP-graph capture would
                # have captured the plan-build kernels and taken the plan's memory from the graph's pool.  A many-angle "auto"
                # plan therefore holds BOTH forms: 2.35 MB compact + 17 MB u16 at 180 angles.
                self._u16_plan = self._build_plan(0)
            self._want_bwd_plan = bool(self.mode == _lib.BWD_TF_COMPAT and
                                       self._lib.ctpvae_rotate_plan_supported(*geo, 1))
        # slices larger than LDS: the tiled forward walks compact TILE plans when they can be built (csrc/rotate.hip)
        self._tplan = None
        if (use_plan and self._fwd_plan is None and self.interp == _lib.NEAREST and plan_format != "u16"
                and self.py > 0 and self.px > 0):
            self._tplan = self._build_tile_plan()
        # ... and their backward takes three index operations per tap from a step plan instead of five (csrc/rotate.hip); so do
        # large batches of slices that fit LDS, where it beats the planned backward (backward_uses_step_plan)
        self._step_plan = None
        if (use_plan and self.interp == _lib.NEAREST and self.mode == _lib.BWD_TF_COMPAT and self.py > 0 and self.px > 0
                and self.A <= 65535 and self.H * self.W >= 128 * 128):
            self._step_plan = self._build_step_plan()
        if self._want_exact_plan:
            # built here, not on the first backward: reading its overflow word synchronises the stream, which must not
            # happen inside a caller's HIP-graph capture
            self._build_exact_plan()
        # exact transpose through an inverse plan of SUMMED WEIGHTS (round 5): the bilinear forward's at every size, the nearest
        # forward's where the byte plan above does not hold the geometry (512 x 512: the scatter there is 6 G global atomics)
        self._exact_bilin_plan = None
        if (use_plan and self.mode == _lib.BWD_EXACT and self.H <= 65535
                and (self.interp == _lib.BILINEAR or self._exact_plan is None)):
            self._build_exact_bilinear_plan()

    def _build_plan(self, which):
        nbytes = self._lib.ctpvae_rotate_plan_bytes(self.H, self.W, self.PH, self.PW, self.A, which)
        _lib.check(nbytes, "rotate_plan_bytes")
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctpvae_rotate_plan_build_f32(
                self.T8.data_ptr(), self.Tinv8.data_ptr(), self.A, self.H, self.W, self.PH, self.PW, self.py, self.px,
                buf.data_ptr() if which == 0 else None, buf.data_ptr() if which == 1 else None, _stream_ptr()),
                "rotate_plan_build")
        return buf

    def _build_compact_plan(self):
        """Step-coded forward plan (first tap + 2 bits per row: csrc/rotate_cplan.hip), or None when some ray's steps do not
        fit the code (unpadded canvases at oblique angles, rows that are not a rotation): the u16 plan is built then.
        Reading the overflow word synchronises -- here, at construction, never inside a caller's graph capture."""
        nbytes = self._lib.ctpvae_rotate_cplan_bytes(self.H, self.W, self.PH, self.PW, self.A)
        _lib.check(nbytes, "rotate_cplan_bytes")
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctpvae_rotate_cplan_build_f32(self.T8.data_ptr(), self.A, self.H, self.W, self.PH, self.PW,
                                                               self.py, self.px, buf.data_ptr(), _stream_ptr()),
                       "rotate_cplan_build")
            over = self._lib.ctpvae_rotate_cplan_overflowed(buf.data_ptr(), self.H, self.W, self.PH, self.PW, self.A,
                                                            _stream_ptr())
        _lib.check(over, "rotate_cplan_overflowed")
        return buf if over == 0 else None

    def _build_tile_plan(self):
        """Compact plans of the tiles of a slice that does not fit LDS (None: not a tiled geometry, or a ray whose steps do
        not fit the code -- the direct tiled kernel then)."""
        nbytes = self._lib.ctpvae_rotate_tplan_bytes(self.H, self.W, self.PH, self.PW, self.A)
        _lib.check(nbytes, "rotate_tplan_bytes")
        if nbytes == 0:
            return None
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctpvae_rotate_tplan_build_f32(self.T8.data_ptr(), self.A, self.H, self.W, self.PH, self.PW, self.py,
                                                               self.px, buf.data_ptr(), _stream_ptr()), "rotate_tplan_build")
            over = self._lib.ctpvae_rotate_tplan_overflowed(buf.data_ptr(), self.H, self.W, self.PH, self.PW, self.A, _stream_ptr())
        _lib.check(over, "rotate_tplan_overflowed")
        return buf if over == 0 else None

    def _get_bwd4_plan(self):
        """The angle-selecting backward plan (one dword = the four rows a lane owns at one angle), or None when the geometry
        does not fit byte taps -- the segment kernel (ctpvae_rotate_bwd_sel_scaled_f32) serves subsets then."""
        if self._bwd4_plan is None and self._want_bwd4:
            self._want_bwd4 = False
            nbytes = self._lib.ctpvae_rotate_bwd4_plan_bytes(self.H, self.W, self.PH, self.PW, self.A)
            _lib.check(nbytes, "rotate_bwd4_plan_bytes")
            if nbytes > 0:
                buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
                with torch.cuda.device(self.device):
                    _lib.check(self._lib.ctpvae_rotate_bwd4_plan_build_f32(self.Tinv8.data_ptr(), self.A, self.H, self.W, self.PH,
                                                                           self.PW, self.py, self.px, buf.data_ptr(),
                                                                           _stream_ptr()), "rotate_bwd4_plan_build")
                self._bwd4_plan = buf
        return self._bwd4_plan

    def _build_exact_plan(self):
        """Inverse plan of the nearest forward (<= 2 hitting bins per angle and pixel).  Falls back to the scatter kernel
        (atomics) when the geometry does not fit byte taps or the rows are not a rotation."""
        self._want_exact_plan = False
        nbytes = self._lib.ctpvae_rotate_exact_plan_bytes(self.H, self.W, self.PH, self.PW, self.A)
        _lib.check(nbytes, "rotate_exact_plan_bytes")
        if nbytes == 0:
            return
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctpvae_rotate_exact_plan_build_f32(self.T8.data_ptr(), self.Tinv8.data_ptr(), self.A, self.H,
                                                                    self.W, self.PH, self.PW, self.py, self.px, buf.data_ptr(),
                                                                    _stream_ptr()), "rotate_exact_plan_build")
            over = self._lib.ctpvae_rotate_exact_plan_overflowed(buf.data_ptr(), self.H, self.W, self.PH, self.PW, self.A,
                                                                 _stream_ptr())
        _lib.check(over, "rotate_exact_plan_overflowed")
        if over == 0:
            self._exact_plan = buf

    def _build_exact_bilinear_plan(self):
        """Inverse plan of the bilinear forward: per (angle, pixel) the first of <= 3 contributing bins and the pixel's summed
        weight in each -- 16 bytes per (angle, pixel) (5.2 MB at 128 x 128 x 20 angles, 377 MB at 512 x 512 x 90).  Falls back
        to the scatter kernel (atomics) when the rows are not a rotation (the build says so)."""
        nbytes = self._lib.ctpvae_rotate_exact_wplan_bytes(self.H, self.W, self.A)
        _lib.check(nbytes, "rotate_exact_wplan_bytes")
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctpvae_rotate_exact_wplan_build_f32(
                self.T8.data_ptr(), self.Tinv8.data_ptr(), self.A, self.H, self.W, self.PH, self.PW, self.py, self.px,
                self.interp, buf.data_ptr(), _stream_ptr()), "rotate_exact_wplan_build")
            over = self._lib.ctpvae_rotate_exact_wplan_overflowed(buf.data_ptr(), self.H, self.W, self.A, _stream_ptr())
        _lib.check(over, "rotate_exact_wplan_overflowed")
        if over == 0:
            self._exact_bilin_plan = buf

    @property
    def planned(self):
        """(forward uses a plan, backward uses / will use a plan)"""
        return self._fwd_plan is not None, self._want_bwd_plan

    @property
    def compact(self):
        """True if the forward plan is the compact (step-coded) form."""
        return self._compact

    def dense_plan(self, S):
        """(plan buffer, is_compact) for a DENSE forward launch over S slices (all plan angles, no in-launch sums): the plan
        format "auto" measured faster at this shape -- see COMPACT_DENSE_WINDOW.  Same bits either way."""
        if self._auto_dense_u16 and not (self.COMPACT_DENSE_WINDOW[0] <= S <= self.COMPACT_DENSE_WINDOW[1]):
            if self._u16_plan is None:
                self._u16_plan = self._build_plan(0)
            return self._u16_plan, False
        return self._fwd_plan, self._compact

    def forward_kernel_name(self, S):
        if self._fwd_plan is None:
            return "rotate_fwd_fast_kernel"
        return "rotate_fwd_compact_kernel" if self.dense_plan(S)[1] else "rotate_fwd_planned_kernel"

    def _run_compact(self, img_ptr, S, out_ptr, angles_i=None, n=0, mask=None, meas=None, dense=0, pnm=None, eps=0.0,
                     lp_ptr=None, dlp_ptr=None, part_ptr=None, sum_ptr=None):
        return self._lib.ctpvae_rotate_fwd_compact_f32(
            img_ptr, S, self.H, self.W, self.PH, self.PW, self.A, self._fwd_plan.data_ptr(),
            angles_i.data_ptr() if angles_i is not None else None, n,
            1 if (angles_i is not None and angles_i.device.type == "cpu") else 0, mask.data_ptr() if mask is not None else None,
            meas.data_ptr() if meas is not None else None, dense, pnm.data_ptr() if pnm is not None else None,
            ctypes.c_float(eps), out_ptr, lp_ptr, dlp_ptr, part_ptr, sum_ptr, _stream_ptr(self._dev_index))

    def forward_loglik_sums(self, img, mask, meas, pnm, eps, angles_i=None, dense_inputs=False, with_dlp=True):
        """Per-object log-likelihood sums (ctvae/helper_functions.py:305-312: reduce_sum of the log-probabilities over
        angles and bins) -> (sums [S], d lp / d ray-sum [S][n][PW] or None), in the library's fixed order
        (ctpvae_loglik_object_sums_f32 / oracle.loglik_object_sums).  On a compact plan the reduction happens INSIDE the
        projector launch: neither the sinogram nor the log-probabilities are written to HBM, only dlp (the
        backward's operand) and one partial per (object, angle, 64-bin task).  Other planned / tiled geometries write the
        log-probabilities and reduce them with the same-order kernel."""
        if angles_i is not None and not self.sel_supported(angles_i.numel()):
            if dense_inputs:
                idx = self._sel_dev(angles_i).long()
                mask, meas = mask.index_select(1, idx).contiguous(), meas.index_select(1, idx).contiguous()
            return self.subset(angles_i).forward_loglik_sums(img, mask, meas, pnm, eps, with_dlp=with_dlp)
        with torch.cuda.device(self._dev_index):
            S = img.shape[0]
            sums = _new_output((S,), torch.float32, self._tdev)
            ws = self._tile_workspace(S) if angles_i is None else None
            if ws is not None:      # tiled geometry: the reduce pass of the tiled forward reduces the log-probabilities too
                self._check(img, (self.H, self.W), "img")
                self._check(meas, (self.A, self.PW), "meas")
                self._check(mask, (self.A,), "mask")
                if meas.shape[0] != S or mask.shape[0] != S or pnm.numel() != 1 or pnm.dtype != torch.float32 or pnm.device != img.device:
                    raise ValueError("mask [S][A], meas [S][A][PW] and a one-element float32 pnm on the same device are needed")
                dlp = _new_output((S, self.A, self.PW), torch.float32, self._tdev) if with_dlp else None
                part = self._part_workspace(S, self.A, 1)
                rc = self._lib.ctpvae_rotate_fwd_tiled_compact_f32(
                    img.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.py, self.px, self.T8.data_ptr(), self.A,
                    self._tplan.data_ptr() if self._tplan is not None else None, ws.data_ptr(), mask.data_ptr(), meas.data_ptr(),
                    pnm.data_ptr(), ctypes.c_float(eps), None, None, dlp.data_ptr() if dlp is not None else None,
                    part.data_ptr(), sums.data_ptr(), _stream_ptr(self._dev_index))
                if rc:
                    _lib.check(rc, "rotate_fwd_tiled_compact")
                return sums, dlp
            if not self._compact:
                res = self._forward_loglik(img, mask, meas, pnm, eps, with_dlp=with_dlp, angles_i=angles_i,
                                           dense_inputs=dense_inputs)
                lp = res[1]
                partition = 0 if self._fwd_plan is not None else 1
                _lib.check(self._lib.ctpvae_loglik_object_sums_f32(lp.data_ptr(), S, lp.shape[1], self.PW, partition,
                                                                   sums.data_ptr(), _stream_ptr(self._dev_index)),
                           "loglik_object_sums")
                return sums, (res[2] if with_dlp else None)
            self._check(img, (self.H, self.W), "img")
            n = self.A if angles_i is None else self._check_sel(angles_i)
            n_in = self.A if (angles_i is None or dense_inputs) else n
            self._check(meas, (n_in, self.PW), "meas")
            self._check(mask, (n_in,), "mask")
            if meas.shape[0] != S or mask.shape[0] != S or pnm.numel() != 1 or pnm.dtype != torch.float32 or pnm.device != img.device:
                raise ValueError("mask [S][A], meas [S][A][PW] and a one-element float32 pnm on the same device are needed")
            dlp = _new_output((S, n, self.PW), torch.float32, self._tdev) if with_dlp else None
            part = self._part_workspace(S, n, 0)
            rc = self._run_compact(img.data_ptr(), S, None, angles_i, n if angles_i is not None else 0, mask, meas,
                                   1 if dense_inputs else 0, pnm, eps, None, dlp.data_ptr() if dlp is not None else None,
                                   part.data_ptr(), sums.data_ptr())
            if rc:
                _lib.check(rc, "rotate_fwd_compact")
            return sums, dlp

    def _part_workspace(self, S, n, partition):
        """The partial sums + arrival counters of forward_loglik_sums (ctpvae_loglik_part_floats): kept per (S, n) -- the counters
        are zeroed once, here, and every launch leaves them zero -- and used by one launch at a time (this plan's launches go to
        one stream, like its tile workspace)."""
        ws = self._part_ws.get((S, n, partition))
        if ws is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the first per-object-sum call of a shape allocates and zeroes its workspace: run it once before "
                                   "capturing a HIP graph (the zero fill must not be captured: the counters are zeroed ONCE)")
            nfl = self._lib.ctpvae_loglik_part_floats(S, n, self.PW, partition)
            _lib.check(nfl, "loglik_part_floats")
            if len(self._part_ws) >= 8:
                self._part_ws.clear()
            ws = self._part_ws[(S, n, partition)] = torch.zeros((int(nfl),), dtype=torch.float32, device=self._tdev)
        return ws

    def _sel_dev(self, angles_i):
        """The subset as a device vector (kernels without a host-index form; torch gathers)."""
        return angles_i if angles_i.device == self._tdev else angles_i.to(self._tdev)

    def subset(self, angles_i):
        """A plan over rows `angles_i` of this one's tables (two small gathers, no plan kernels): the fallback for angle
        subsets on paths whose kernels take no angle-index operand (tiled / bilinear forward, exact backward)."""
        idx = as_angle_index(angles_i, self._tdev).long()
        return RotatePlan(None, self.H, self.W, self._pad, self.device, interp=self._interp_name,
                          backward=self._backward_name, use_plan=False if self._fwd_plan is not None else self._use_tiles,
                          _tables=(self.T8.index_select(0, idx), self.Tinv8.index_select(0, idx)))

    def sel_supported(self, n, forward=True):
        """True if a launch over `n` selected angles runs on THIS (dense) plan with an angle-index operand."""
        if forward:
            return self._fwd_plan is not None and 1 <= n <= self.MAX_SEL
        return self.supports_scale and n >= 1

    # The library launches on the calling thread's current HIP device: the wrappers below make that this plan's device (the
    # common case -- it already is -- costs one cheap query).
    def forward(self, img, out=None, angles_i=None):
        """slices [S][H][W] -> sinograms [S][A][PW] (raw operator, no autograd bookkeeping).
        angles_i: int32 device vector of plan angles -> sinograms [S][len(angles_i)][PW] of those angles, in that order."""
        n = self.A if angles_i is None else angles_i.numel()
        if img.dim() == 3 and img.shape[0] == 0 and tuple(img.shape[1:]) == (self.H, self.W):   # empty batch -> empty batch
            return out if out is not None else img.new_empty((0, n, self.PW))
        if angles_i is not None and not self.sel_supported(n):
            return self.subset(angles_i).forward(img, out)
        if _current_device() == self._dev_index:
            return self._forward(img, out, angles_i)
        with torch.cuda.device(self._dev_index):
            return self._forward(img, out, angles_i)

    def forward_f64(self, img):
        """float64 slices [S][H][W] -> float64 sinograms [S][A][PW]: fp32 coordinates and weights, float64 taps, products and
        row sum -- TensorFlow's arithmetic for a float64 image (ctvae/tomopy_forward_compare.py:52,56 projects xdesign's float64
        phantoms).  A correctness-first kernel (ctpvae_rotate_fwd_f64); the float32 paths are untouched."""
        if (img.dim() != 3 or tuple(img.shape[1:]) != (self.H, self.W) or img.dtype is not torch.float64 or not img.is_contiguous()
                or img.device != self._tdev):
            raise ValueError(f"img must be a contiguous float64 tensor [S][{self.H}][{self.W}] on {self._tdev} "
                             f"(got {tuple(img.shape)}, {img.dtype}, {img.device})")
        S = img.shape[0]
        out = _new_output((S, self.A, self.PW), torch.float64, img.device)
        if S == 0:
            return out
        with torch.cuda.device(self._dev_index):
            _lib.check(self._lib.ctpvae_rotate_fwd_f64(img.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.py, self.px,
                                                       self.T8.data_ptr(), self.A, self.interp, out.data_ptr(),
                                                       _stream_ptr(self._dev_index)), "rotate_fwd_f64")
        return out

    def forward_loglik(self, img, mask, meas, pnm, eps, out=None, out_lp=None, out_dlp=None, with_dlp=False,
                       angles_i=None, dense_inputs=False):
        """Forward with the log-likelihood epilogue (one launch): returns (sino, lp), both [S][A][PW];
        lp = Normal(loc = sino * mask, scale = eps + sqrt(loc / pnm + eps)).log_prob(meas).  Planned and tiled geometries.
        with_dlp: also returns d lp / d sino (third value), which `backward(dlp, scale=...)` turns into the image
        gradient without an elementwise pass.
        angles_i (planned geometries): int32 device vector of plan angles; outputs are [S][len(angles_i)][PW].  With
        dense_inputs, mask [S][A] and meas [S][A][PW] are the DENSE arrays and the kernel reads them at the selected
        angles (the reference's tf.gather of both, ctvae/helper_functions.py:356-357, costs no launch)."""
        if angles_i is not None and not self.sel_supported(angles_i.numel()):
            if dense_inputs:
                idx = self._sel_dev(angles_i).long()
                mask, meas = mask.index_select(1, idx).contiguous(), meas.index_select(1, idx).contiguous()
            return self.subset(angles_i).forward_loglik(img, mask, meas, pnm, eps, out, out_lp, out_dlp, with_dlp)
        if _current_device() == self._dev_index:
            return self._forward_loglik(img, mask, meas, pnm, eps, out, out_lp, out_dlp, with_dlp, angles_i, dense_inputs)
        with torch.cuda.device(self._dev_index):
            return self._forward_loglik(img, mask, meas, pnm, eps, out, out_lp, out_dlp, with_dlp, angles_i, dense_inputs)

    def backward(self, gsino, out=None, scale=None, angles_i=None):
        """cotangents [S][A][PW] -> gradient images [S][H][W] (the mode chosen at construction).
        scale: optional float32 device tensor of S per-slice factors (any stride, 0 included: an expanded scalar)
        applied in the kernel's store -- nearest / tf_compat only (`supports_scale`).
        angles_i: the cotangents are [S][len(angles_i)][PW], row k belonging to plan angle angles_i[k]."""
        n = self.A if angles_i is None else angles_i.numel()
        if gsino.dim() == 3 and gsino.shape[0] == 0 and tuple(gsino.shape[1:]) == (n, self.PW):
            return out if out is not None else gsino.new_empty((0, self.H, self.W))
        if angles_i is not None and not self.sel_supported(n, forward=False):
            return self.subset(angles_i).backward(gsino, out, scale)
        if _current_device() == self._dev_index:
            return self._backward(gsino, out, scale, angles_i)
        with torch.cuda.device(self._dev_index):
            return self._backward(gsino, out, scale, angles_i)

    @property
    def supports_scale(self):
        return self.interp == _lib.NEAREST and self.mode == _lib.BWD_TF_COMPAT

    def _check(self, t, shape_tail, what):
        """Operand checks before a launch: the kernels index by these shapes and would read out of bounds otherwise.
        Every call checks -- the tests cost about a microsecond.  (Round 2 remembered tensor OBJECTS that had passed;
        `t.data = other`, `t.resize_()` or `t.set_()` change dtype, shape or storage under an unchanged object, and a
        kernel would then have indexed a buffer of the wrong size.)"""
        if (t.shape[1:] == shape_tail and t.dtype is torch.float32 and t.is_contiguous() and t.device == self._tdev
                and t.shape[0] > 0):
            return
        raise ValueError(f"{what} must be a contiguous float32 tensor [S]{list(shape_tail)} on {self.T8.device} "
                         f"(got {tuple(t.shape)}, {t.dtype}, {t.device}, contiguous={t.is_contiguous()})")

    def _tile_workspace(self, S):
        """Device workspace of the tiled forward for S slices, or None when this geometry is not tiled."""
        if not self._use_tiles or self._fwd_plan is not None:
            return None
        need = self._lib.ctpvae_rotate_fwd_tiled_workspace_bytes(S, self.H, self.W, self.PH, self.PW, self.A, self.interp)
        _lib.check(need, "rotate_fwd_tiled_workspace_bytes")
        if need == 0:
            self._use_tiles = False
            return None
        if self._tile_ws is None or self._tile_ws.numel() < need:
            self._tile_ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        return self._tile_ws

    @property
    def tiled(self):
        """True if the forward cuts slices into LDS-sized tiles (slices larger than LDS, nearest)."""
        return self._tile_workspace(1) is not None

    def _check_sel(self, angles_i):
        host = angles_i.device.type == "cpu"
        if (angles_i.dtype is not torch.int32 or angles_i.dim() != 1 or not angles_i.is_contiguous()
                or not (host or angles_i.device == self._tdev)):
            raise ValueError(f"angles_i must be a contiguous int32 vector on {self._tdev} or in host memory (see "
                             f"as_angle_index; got {tuple(angles_i.shape)}, {angles_i.dtype}, {angles_i.device})")
        if host and angles_i.numel():
            # host-resident indices are checked like the reference's tf.gather would (it raises on an index outside the
            # angle list); device-resident ones cannot be read without a sync and are clamped by the kernels
            lo, hi = int(angles_i.min()), int(angles_i.max())
            if lo < 0 or hi >= self.A:
                raise ValueError(f"angles_i holds indices outside this plan's {self.A} angles (min {lo}, max {hi})")
        return angles_i.numel()

    def _forward(self, img, out=None, angles_i=None):
        self._check(img, (self.H, self.W), "img")
        S = img.shape[0]
        n = self.A if angles_i is None else self._check_sel(angles_i)
        if out is None:
            out = _new_output((S, n, self.PW), torch.float32, img.device)
        else:
            self._check(out, (n, self.PW), "out")
            if out.shape[0] != S:
                raise ValueError(f"out holds {out.shape[0]} sinograms for {S} slices")
        fplan, compact = (self._fwd_plan, self._compact) if (angles_i is not None or self._fwd_plan is None) else self.dense_plan(S)
        if compact:
            rc = self._run_compact(img.data_ptr(), S, out.data_ptr(), angles_i, n if angles_i is not None else 0)
            if rc:
                _lib.check(rc, "rotate_fwd_compact")
            return out
        if angles_i is not None:
            angles_i = self._sel_dev(angles_i)
            rc = self._lib.ctpvae_rotate_fwd_planned_sel_f32(img.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.A,
                                                             self._fwd_plan.data_ptr(), angles_i.data_ptr(), n,
                                                             out.data_ptr(), _stream_ptr(self._dev_index))
            if rc:
                _lib.check(rc, "rotate_fwd_planned_sel")
            return out
        ws = self._tile_workspace(S)
        if fplan is not None:
            rc = self._lib.ctpvae_rotate_fwd_planned_f32(img.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.A,
                                                         fplan.data_ptr(), out.data_ptr(), _stream_ptr(self._dev_index))
        elif ws is not None and self._tplan is not None:
            rc = self._lib.ctpvae_rotate_fwd_tiled_compact_f32(img.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.py,
                                                               self.px, self.T8.data_ptr(), self.A, self._tplan.data_ptr(),
                                                               ws.data_ptr(), None, None, None, ctypes.c_float(0.0),
                                                               out.data_ptr(), None, None, None, None, _stream_ptr(self._dev_index))
        elif ws is not None:
            rc = self._lib.ctpvae_rotate_fwd_tiled_interp_f32(img.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.py,
                                                              self.px, self.T8.data_ptr(), self.A, self.interp, ws.data_ptr(),
                                                              out.data_ptr(), _stream_ptr(self._dev_index))
        else:
            rc = self._lib.ctpvae_rotate_fwd_f32(img.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.py,
                                                 self.px, self.T8.data_ptr(), self.A, self.interp, out.data_ptr(),
                                                 _stream_ptr(self._dev_index))
        if rc:
            _lib.check(rc, "rotate_fwd")
        return out

    def _forward_loglik(self, img, mask, meas, pnm, eps, out=None, out_lp=None, out_dlp=None, with_dlp=False,
                        angles_i=None, dense_inputs=False):
        self._check(img, (self.H, self.W), "img")
        S = img.shape[0]
        ws = self._tile_workspace(S)
        if (self._fwd_plan is None and ws is None) or self.interp != _lib.NEAREST:
            raise ValueError("forward_loglik needs a planned or tiled forward (nearest)")
        n = self.A if angles_i is None else self._check_sel(angles_i)
        n_in = self.A if (angles_i is None or dense_inputs) else n     # angle rows of mask / meas
        self._check(meas, (n_in, self.PW), "meas")
        self._check(mask, (n_in,), "mask")
        if meas.shape[0] != S or mask.shape[0] != S or pnm.numel() != 1 or pnm.dtype != torch.float32 or pnm.device != img.device:
            raise ValueError("mask [S][A], meas [S][A][PW] and a one-element float32 pnm on the same device are needed")
        if out is None:
            out = _new_output((S, n, self.PW), torch.float32, img.device)
        else:
            self._check(out, (n, self.PW), "out")
        if out_lp is None:
            out_lp = _new_output(out.shape, out.dtype, out.device)
        else:
            self._check(out_lp, (n, self.PW), "out_lp")
        if out.shape[0] != S or out_lp.shape[0] != S:
            raise ValueError("out / out_lp must hold one sinogram per slice")
        if with_dlp or out_dlp is not None:
            if out_dlp is None:
                out_dlp = _new_output(out.shape, out.dtype, out.device)
            else:
                self._check(out_dlp, (n, self.PW), "out_dlp")
                if out_dlp.shape[0] != S:
                    raise ValueError("out_dlp must hold one sinogram per slice")
        dlp_ptr = out_dlp.data_ptr() if out_dlp is not None else None
        fplan, compact = (self._fwd_plan, self._compact) if (angles_i is not None or self._fwd_plan is None) else self.dense_plan(S)
        if compact:
            rc = self._run_compact(img.data_ptr(), S, out.data_ptr(), angles_i, n if angles_i is not None else 0, mask, meas,
                                   1 if dense_inputs else 0, pnm, eps, out_lp.data_ptr(), dlp_ptr)
        elif angles_i is not None:
            angles_i = self._sel_dev(angles_i)
            rc = self._lib.ctpvae_rotate_fwd_planned_loglik_sel_f32(
                img.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.A, self._fwd_plan.data_ptr(), angles_i.data_ptr(), n,
                mask.data_ptr(), meas.data_ptr(), 1 if dense_inputs else 0, pnm.data_ptr(), ctypes.c_float(eps), out.data_ptr(),
                out_lp.data_ptr(), dlp_ptr, _stream_ptr(self._dev_index))
        elif fplan is not None:
            rc = self._lib.ctpvae_rotate_fwd_planned_loglik_f32(
                img.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.A, fplan.data_ptr(), mask.data_ptr(),
                meas.data_ptr(), pnm.data_ptr(), ctypes.c_float(eps), out.data_ptr(), out_lp.data_ptr(), dlp_ptr,
                _stream_ptr(self._dev_index))
        elif self._tplan is not None:
            rc = self._lib.ctpvae_rotate_fwd_tiled_compact_f32(
                img.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.py, self.px, self.T8.data_ptr(), self.A,
                self._tplan.data_ptr(), ws.data_ptr(), mask.data_ptr(), meas.data_ptr(), pnm.data_ptr(), ctypes.c_float(eps),
                out.data_ptr(), out_lp.data_ptr(), dlp_ptr, None, None, _stream_ptr(self._dev_index))
        else:
            rc = self._lib.ctpvae_rotate_fwd_tiled_loglik_f32(
                img.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.py, self.px, self.T8.data_ptr(), self.A,
                ws.data_ptr(), mask.data_ptr(), meas.data_ptr(), pnm.data_ptr(), ctypes.c_float(eps), out.data_ptr(),
                out_lp.data_ptr(), dlp_ptr, _stream_ptr(self._dev_index))
        if rc:
            _lib.check(rc, "rotate_fwd_planned_loglik")
        return (out, out_lp, out_dlp) if out_dlp is not None else (out, out_lp)

    def _build_step_plan(self):
        """Step plan of the segment backward (slices too large for the planned one: 512 x 512), or None when the geometry does not
        fit the code (the build says so).  Reading the overflow word synchronises -- at construction, never inside a capture."""
        nbytes = self._lib.ctpvae_rotate_bwd_step_plan_bytes(self.H, self.W, self.A)
        _lib.check(nbytes, "rotate_bwd_step_plan_bytes")
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctpvae_rotate_bwd_step_plan_build_f32(self.Tinv8.data_ptr(), self.A, self.H, self.W, self.PH, self.PW,
                                                                       self.py, self.px, buf.data_ptr(), _stream_ptr()),
                       "rotate_bwd_step_plan_build")
            over = self._lib.ctpvae_rotate_bwd_step_plan_overflowed(buf.data_ptr(), self.H, self.W, self.A, _stream_ptr())
        _lib.check(over, "rotate_bwd_step_plan_overflowed")
        return buf if over == 0 else None

    def backward_uses_step_plan(self, S):
        """The segment backward over the step plan (ctpvae_rotate_bwd_stepped_scaled_f32; below 512 tiles of 64 x 32 that entry
        point runs the direct segment kernel itself): always for slices too large for the planned backward; for slices that fit
        LDS where _segments_win says so -- round 3, 128 x 128, planned vs this entry point:
        B=400 A=180 142.7 vs 111.9 us, B=400 A=20 30.0 vs 20.0, B=200 A=90 47.0 vs 39.4, B=160 A=180 62.4 vs 58.9; the planned
        kernel keeps B=128 A=180 (42.5 vs 49.7) and everything smaller at many angles.  The three paths give the same bits."""
        if self._step_plan is None:
            return False
        return (not self._want_bwd_plan) or self._segments_win(S)

    def _segments_win(self, S):
        """Where the segment kernels beat the planned gather at sizes that fit LDS (round 5, both paths forced over batch sizes x angle
        counts x image sizes, tools/sweep_bwd_paths_grid.py, profiles/r05_nearest_rules.txt).  128 x 128: from 200 slices on; from 80 at
        <= 24 angles; from 128 at <= 45 (round 3's rule -- 160, or 80 at <= 64 angles -- sent 80 .. 160 slices x 45 .. 64 angles to the
        segments: 100 x 64 angles 20.8 us against the planned gather's 14.7).  LARGER slices (160 x 160: the planned tiles stage detector
        rows of 230 bins): always -- 50 x 160^2 x 180 angles 42.5 us against 77.9.  SMALLER ones (no step plan below 128 x 128: the
        direct segment kernel): only long launches at few angles -- 128 x 100^2 x 45 angles had run 47.2 us there against 12.5 planned."""
        area = self.H * self.W
        if self.py <= 0 or self.px <= 0:
            # an unpadded canvas: pixels map off the canvas at most angles, the segment kernels' tiles leave their all-inside loop
            # (200 x 128^2 x 20 angles: 30.3 us there, 11 us on the planned gather)
            return False
        if area > 128 * 128:
            return True
        if area < 128 * 128:
            return S >= 200 and self.A <= 24 and area >= 64 * 64
        return S >= 200 or (S >= 80 and self.A <= 24) or (S >= 128 and self.A <= 45)

    def backward_uses_plan(self, S):
        """The planned gather backward: slices that fit LDS, except where the segment kernels are faster (see above; without a
        step plan -- unpadded canvases -- the direct segment kernel still takes the shapes of _segments_win)."""
        return self._want_bwd_plan and not self.backward_uses_step_plan(S) and not self._segments_win(S)

    def backward_kernel_name(self, S):
        if self.backward_uses_step_plan(S):
            return "rotate_bwd_stepped_kernel"
        return "rotate_bwd_planned_kernel" if self.backward_uses_plan(S) else "rotate_bwd_tfcompat_seg_kernel"

    def _backward(self, gsino, out=None, scale=None, angles_i=None):
        n = self.A if angles_i is None else self._check_sel(angles_i)
        self._check(gsino, (n, self.PW), "gsino")
        S = gsino.shape[0]
        sc_ptr, sc_stride = None, 0
        if scale is not None:
            if not self.supports_scale:
                raise ValueError("a per-slice scale needs interp='nearest' and backward='tf_compat'")
            if (scale.dim() != 1 or scale.shape[0] != S or scale.dtype is not torch.float32 or scale.device != self._tdev):
                raise ValueError(f"scale must be a float32 tensor of {S} per-slice factors on {self._tdev} "
                                 f"(got {tuple(scale.shape)}, {scale.dtype}, {scale.device})")
            sc_ptr, sc_stride = scale.data_ptr(), scale.stride(0)
        if out is None:
            out = _new_output((S, self.H, self.W), torch.float32, gsino.device)
        else:
            self._check(out, (self.H, self.W), "out")
            if out.shape[0] != S:
                raise ValueError(f"out holds {out.shape[0]} slices for {S} sinograms")
        if angles_i is not None and self._get_bwd4_plan() is not None:
            if angles_i.device.type == "cpu" and n > self.MAX_SEL:
                angles_i = self._sel_dev(angles_i)     # host indices ride the launch arguments: at most MAX_SEL of them
            rc = self._lib.ctpvae_rotate_bwd_planned_sel_scaled_f32(gsino.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.A,
                                                                    self._bwd4_plan.data_ptr(), angles_i.data_ptr(), n,
                                                                    1 if angles_i.device.type == "cpu" else 0,
                                                                    sc_ptr, sc_stride, out.data_ptr(), _stream_ptr(self._dev_index))
            if rc:
                _lib.check(rc, "rotate_bwd_planned_sel")
            return out
        if angles_i is not None:
            # the segment kernel reads its table rows through the index vector: nothing is gathered, no plan is built
            angles_i = self._sel_dev(angles_i)
            rc = self._lib.ctpvae_rotate_bwd_sel_scaled_f32(gsino.data_ptr(), S, self.A, self.PH, self.PW, self.Tinv8.data_ptr(),
                                                            angles_i.data_ptr(), n, self.H, self.W, self.py, self.px,
                                                            sc_ptr, sc_stride, out.data_ptr(), _stream_ptr(self._dev_index))
            if rc:
                _lib.check(rc, "rotate_bwd_sel")
            return out
        if self._want_exact_plan:            # first exact backward of a nearest plan: build the inverse plan
            self._build_exact_plan()
        if self._exact_bilin_plan is not None:
            rc = self._lib.ctpvae_rotate_bwd_exact_wplan_f32(gsino.data_ptr(), S, self.A, self.PH, self.PW,
                                                                        self.Tinv8.data_ptr(), self.H, self.W, self.py, self.px,
                                                                        self._exact_bilin_plan.data_ptr(), out.data_ptr(),
                                                                        _stream_ptr(self._dev_index))
        elif self._exact_plan is not None:
            rc = self._lib.ctpvae_rotate_bwd_exact_planned_f32(gsino.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.A,
                                                               self._exact_plan.data_ptr(), out.data_ptr(),
                                                               _stream_ptr(self._dev_index))
        elif self.backward_uses_plan(S):
            if self._bwd_plan is None:
                self._bwd_plan = self._build_plan(1)
            rc = self._lib.ctpvae_rotate_bwd_planned_scaled_f32(gsino.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.A,
                                                                self._bwd_plan.data_ptr(), sc_ptr, sc_stride, out.data_ptr(),
                                                                _stream_ptr(self._dev_index))
        elif self.backward_uses_step_plan(S):
            rc = self._lib.ctpvae_rotate_bwd_stepped_scaled_f32(gsino.data_ptr(), S, self.A, self.PH, self.PW, self.Tinv8.data_ptr(),
                                                                self.H, self.W, self.py, self.px, self._step_plan.data_ptr(),
                                                                sc_ptr, sc_stride, out.data_ptr(), _stream_ptr(self._dev_index))
        else:
            tab = self.Tinv8 if self.mode == _lib.BWD_TF_COMPAT else self.T8
            rc = self._lib.ctpvae_rotate_bwd_scaled_f32(gsino.data_ptr(), S, self.A, self.PH, self.PW, tab.data_ptr(),
                                                        self.interp, self.mode, self.H, self.W, self.py, self.px,
                                                        sc_ptr, sc_stride, out.data_ptr(), _stream_ptr(self._dev_index))
        if rc:
            _lib.check(rc, "rotate_bwd")
        return out

    # The training layout [B][X][Y][1] <-> [B][A][P][1] has the memory layout of [B][X][Y] <-> [B][A][P]: the two methods
    # below launch on the caller's 4-D tensors directly (no reshape / unsqueeze views, one check) -- the drop-in API's
    # fast path for planned float32 geometries.  None = not applicable, take the general path.
    def forward_vae(self, x4):
        if (self._fwd_plan is None or x4.dtype is not torch.float32 or not x4.is_contiguous() or x4.shape[0] == 0
                or x4.device != self._tdev or x4.shape[1] != self.H or x4.shape[2] != self.W
                or _current_device() != self._dev_index):
            return None
        S = x4.shape[0]
        out = _new_output((S, self.A, self.PW, 1), torch.float32, self._tdev)
        fplan, compact = self.dense_plan(S)
        if compact:
            rc = self._run_compact(x4.data_ptr(), S, out.data_ptr())
        else:
            rc = self._lib.ctpvae_rotate_fwd_planned_f32(x4.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.A,
                                                         fplan.data_ptr(), out.data_ptr(), _stream_ptr(self._dev_index))
        if rc:
            _lib.check(rc, "rotate_fwd")
        return out

    def backward_vae(self, g4):
        S = g4.shape[0]
        if (g4.dtype is not torch.float32 or not g4.is_contiguous() or S == 0 or g4.device != self._tdev
                or g4.shape[1] != self.A or g4.shape[2] != self.PW or _current_device() != self._dev_index
                or self._want_exact_plan or self._exact_plan is not None or not self.backward_uses_plan(S)):
            return None
        if self._bwd_plan is None:
            self._bwd_plan = self._build_plan(1)
        out = _new_output((S, self.H, self.W, 1), torch.float32, self._tdev)
        rc = self._lib.ctpvae_rotate_bwd_planned_scaled_f32(g4.data_ptr(), S, self.H, self.W, self.PH, self.PW, self.A,
                                                            self._bwd_plan.data_ptr(), None, 0, out.data_ptr(),
                                                            _stream_ptr(self._dev_index))
        if rc:
            _lib.check(rc, "rotate_bwd")
        return out

    def project_vae_cpp(self, x4):
        """The training layout through the C++ autograd node (csrc/torch_node.cpp): the same two C-ABI calls as
        forward_vae / backward_vae, without the Python of torch.autograd.Function around them.  None = not applicable."""
        node = _lib.torch_node()
        S = x4.shape[0]
        if (node is None or self._fwd_plan is None or x4.dtype is not torch.float32 or not x4.is_contiguous() or S == 0
                or x4.device != self._tdev or self._want_exact_plan or self._exact_plan is not None
                or not self.backward_uses_plan(S)):
            return None
        if self._bwd_plan is None:
            self._bwd_plan = self._build_plan(1)
        fplan, compact = self.dense_plan(S)
        return node.rotate_vae(x4, fplan, self._bwd_plan, self.H, self.W, self.PH, self.PW, self.A,
                               _stream_ptr(self._dev_index), int(compact))

    def loglik_vae_cpp(self, x4, mask, meas, pnm, eps, angles_i):
        """calculate_log_prob_M_given_R through the C++ autograd node (one-launch forward that stores d lp / d sino, scaled
        backward): the calls _ProjectLogLik's fused branch makes.  The caller has checked x4 / mask / meas / pnm (float32,
        contiguous, this device).  None = not applicable."""
        node = _lib.torch_node()
        if node is None or self._fwd_plan is None or not self.supports_scale:
            return None
        use_plan, bplan = 0, self.Tinv8
        if angles_i is None:
            if self.backward_uses_step_plan(x4.shape[0]):
                use_plan, bplan = 3, self._step_plan   # 3: the segment backward over the step plan
            elif self.backward_uses_plan(x4.shape[0]):
                if self._bwd_plan is None:
                    self._bwd_plan = self._build_plan(1)
                use_plan, bplan = 1, self._bwd_plan
        else:
            self._check_sel(angles_i)
            if self._get_bwd4_plan() is not None:      # 2: the angle-selecting planned backward
                use_plan, bplan = 2, self._bwd4_plan
            if angles_i.device.type == "cpu" and not (self._compact and use_plan == 2):
                angles_i = self._sel_dev(angles_i)     # the u16 / segment kernels read the subset from device memory
        return node.rotate_loglik(x4, self._fwd_plan, bplan, self.Tinv8, mask, meas, pnm,
                                  angles_i, eps, [self.H, self.W, self.PH, self.PW, self.A, self.py, self.px, int(use_plan),
                                                  int(angles_i is not None), _stream_ptr(self._dev_index), int(self._compact)])

    def apply(self, img):
        """Differentiable projection of slices [S][H][W] -> [S][A][PW]."""
        return _RotateProject.apply(img, self, 0)


_LAYOUT_SLICES, _LAYOUT_VAE, _LAYOUT_DIM3, _LAYOUT_DIM2 = 0, 1, 2, 3
USE_CPP_NODE = True     # tests and tools/profile_api.py switch this off to compare the two host paths


class _RotateProject(torch.autograd.Function):
    """ONE autograd node from the caller's tensor, in the reference's layout, to the result in the reference's layout
    (ctvae/forward_functions.py:102-121): the [..., 0] / permute / unsqueeze re-layouts happen inside, as views, so that
    the backward pass runs one function -- no select_backward (a zero fill and a copy of the whole batch) around it."""

    @staticmethod
    def forward(ctx, phantom, plan, layout):
        ctx.plan, ctx.layout, ctx.in_dtype = plan, layout, phantom.dtype
        if layout == _LAYOUT_VAE:            # [B][X][Y][1]
            out = plan.forward_vae(phantom)
            if out is not None:
                return out
            x = phantom.reshape(phantom.shape[0], phantom.shape[1], phantom.shape[2])
        elif layout == _LAYOUT_DIM3:         # [X][Y][Z]
            x = phantom.permute(2, 0, 1)
        elif layout == _LAYOUT_DIM2:         # [X][Y]
            x = phantom[None]
        else:
            x = phantom
        if x.dtype is torch.float64:         # the reference's float64 callers: fp32 coordinates and weights, float64 sums
            sino = plan.forward_f64(x.contiguous())
        else:
            if x.dtype is not torch.float32:
                x = x.to(torch.float32)
            if not x.is_contiguous():
                x = x.contiguous()
            sino = plan.forward(x)           # [S][A][PW]
        if layout == _LAYOUT_VAE:
            out = sino.unsqueeze(-1)         # batch x angles x P x 1   (ctvae/forward_functions.py:116-121)
        elif layout in (_LAYOUT_DIM3, _LAYOUT_DIM2):
            out = sino.permute(1, 2, 0)      # angles x P x Z          (ctvae/forward_functions.py:111-114)
        else:
            out = sino
        return out if ctx.in_dtype is torch.float32 else out.to(ctx.in_dtype)

    @staticmethod
    def backward(ctx, gout):
        layout, plan = ctx.layout, ctx.plan
        if layout == _LAYOUT_VAE:
            if ctx.in_dtype is torch.float32:
                gimg = plan.backward_vae(gout)
                if gimg is not None:
                    return gimg, None, None
            g = gout.reshape(gout.shape[0], gout.shape[1], gout.shape[2])
        elif layout in (_LAYOUT_DIM3, _LAYOUT_DIM2):
            g = gout.permute(2, 0, 1)
        else:
            g = gout
        if g.dtype is not torch.float32:
            g = g.to(torch.float32)
        if not g.is_contiguous():
            g = g.contiguous()
        gimg = plan.backward(g)              # [S][H][W]
        if layout == _LAYOUT_VAE:
            gimg = gimg.unsqueeze(-1)
        elif layout == _LAYOUT_DIM3:
            gimg = gimg.permute(1, 2, 0)
        elif layout == _LAYOUT_DIM2:
            gimg = gimg[0]
        return (gimg if ctx.in_dtype is torch.float32 else gimg.to(ctx.in_dtype)), None, None


_PLAN_CACHE = {}
_PLAN_CACHE_MAX = 16


_DEV_THETA_PLANS = {}   # id(theta tensor) -> (weakref, version, {geometry key: plan})
_HOST_THETA_PLANS = {}  # id(theta ndarray) -> (weakref, bytes snapshot, {geometry key: plan})


def _cached_plan(theta, H, W, pad, device, interp, backward):
    """RotatePlan for this call.  A host-resident angle set (list / numpy / CPU tensor) is keyed by value, so the
    scripts that project with a fixed theta build their tables and gather plans once.  A device-resident theta cannot
    be read without synchronising; it is keyed by the tensor OBJECT (and its in-place version counter), so a caller that
    keeps projecting with the same theta tensor -- the dataset's angle list held on the device -- also builds once; a
    fresh tensor per call (the reference's per-step tf.gather of theta) rebuilds: pass `angles_i` to
    calculate_log_prob_M_given_R instead, which selects rows of ONE dense plan."""
    if isinstance(theta, torch.Tensor) and theta.device.type == "cuda":
        geo = (H, W, bool(pad), str(device), interp, backward)
        ent = _DEV_THETA_PLANS.get(id(theta))
        if ent is not None and ent[0]() is theta and ent[1] == theta._version:
            plan = ent[2].get(geo)
            if plan is not None:
                return plan
        else:
            if len(_DEV_THETA_PLANS) >= _PLAN_CACHE_MAX:      # drop dead entries, then the oldest
                for k in [k for k, e in _DEV_THETA_PLANS.items() if e[0]() is None]:
                    del _DEV_THETA_PLANS[k]
                while len(_DEV_THETA_PLANS) >= _PLAN_CACHE_MAX:
                    _DEV_THETA_PLANS.pop(next(iter(_DEV_THETA_PLANS)))
            ent = _DEV_THETA_PLANS[id(theta)] = (weakref.ref(theta), theta._version, {})
        plan = ent[2][geo] = RotatePlan(theta, H, W, pad, device, interp=interp, backward=backward)
        return plan
    if type(theta) is np.ndarray:
        # the same array object as last time (a script's or a trainer's fixed theta): compare its bytes with the snapshot
        # taken when the plan was built (an in-place edit is seen) -- no conversion, no hashing of a new key
        ent = _HOST_THETA_PLANS.get(id(theta))
        if ent is not None and ent[0]() is theta and ent[1] == theta.tobytes():
            plan = ent[2].get((H, W, pad, device, interp, backward))
            if plan is not None:
                return plan
    host = np.ascontiguousarray(np.asarray(theta.detach().cpu() if isinstance(theta, torch.Tensor) else theta,
                                           dtype=np.float32))
    key = (host.tobytes(), H, W, bool(pad), str(device), interp, backward)
    plan = _PLAN_CACHE.get(key)
    if type(theta) is np.ndarray and plan is not None:
        ent = _HOST_THETA_PLANS.get(id(theta))
        snap = theta.tobytes()
        if ent is None or ent[0]() is not theta or ent[1] != snap:
            if len(_HOST_THETA_PLANS) >= 4 * _PLAN_CACHE_MAX:
                _HOST_THETA_PLANS.clear()
            ent = _HOST_THETA_PLANS[id(theta)] = (weakref.ref(theta), snap, {})
        ent[2][(H, W, pad, device, interp, backward)] = plan
    if plan is None:
        plan = RotatePlan(host, H, W, pad, device, interp=interp, backward=backward)
        if len(_PLAN_CACHE) >= _PLAN_CACHE_MAX:
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
        _PLAN_CACHE[key] = plan
    return plan


def _project(phantom, theta, pad, dim, integrate_vae, interp, backward):
    if not isinstance(phantom, torch.Tensor):
        raise TypeError(f"phantom must be a torch.Tensor on a HIP device (got {type(phantom).__name__})")
    if phantom.device.type != "cuda":
        raise _lib.RadonLibraryError(
            f"phantom lives on {phantom.device}: the projector runs on a HIP device only; there is no CPU path")
    if not phantom.dtype.is_floating_point:
        raise TypeError(f"phantom must be floating point (got {phantom.dtype})")
    if integrate_vae:
        if phantom.dim() != 4 or phantom.shape[3] != 1:
            raise ValueError("integrate_vae=True expects batch_size x img_size_x x img_size_y x 1 "
                             f"(got {tuple(phantom.shape)})")
        layout, H, W = _LAYOUT_VAE, phantom.shape[1], phantom.shape[2]
    elif dim == 3:
        if phantom.dim() != 3:
            raise ValueError(f"dim=3 expects img_size_x x img_size_y x img_size_z (got {tuple(phantom.shape)})")
        layout, H, W = _LAYOUT_DIM3, phantom.shape[0], phantom.shape[1]
    elif dim == 2:
        if phantom.dim() != 2:
            raise ValueError(f"dim=2 expects img_size_x x img_size_y (got {tuple(phantom.shape)})")
        layout, H, W = _LAYOUT_DIM2, phantom.shape[0], phantom.shape[1]
    else:
        raise ValueError(f"dim must be 2 or 3 (got {dim})")
    dev = phantom.device
    plan = _cached_plan(theta, H, W, pad, dev, interp, backward)
    if dev.index == _current_device():
        if layout == _LAYOUT_VAE and USE_CPP_NODE:
            out = plan.project_vae_cpp(phantom)
            if out is not None:
                return out
        return _RotateProject.apply(phantom, plan, layout)
    with torch.cuda.device(dev):
        return _RotateProject.apply(phantom, plan, layout)


class _SiddonProject(torch.autograd.Function):
    """model="siddon": TomoPy's ray-driven projector (tomopy.project, ctvae/helper_functions.py:33-38 -- intersection
    lengths) in project_tf_fast's layouts, differentiable: the backward is its exact transpose (libtomo fbp.c's
    accumulation on the object grid, ct_pvae_amd/recon.py).  One node per call, like _RotateProject."""

    @staticmethod
    def forward(ctx, phantom, theta, pad, layout):
        from .helper_functions import create_sinograms
        ctx.theta, ctx.layout, ctx.in_dtype = theta, layout, phantom.dtype
        if layout == _LAYOUT_VAE:
            x = phantom.reshape(phantom.shape[0], phantom.shape[1], phantom.shape[2])
        elif layout == _LAYOUT_DIM3:
            x = phantom.permute(2, 0, 1)
        else:
            x = phantom[None]
        ctx.grid = (x.shape[1], x.shape[2])
        sino = create_sinograms(x, theta, pad=pad)           # [S][A][dx] (float32, contiguous inside)
        out = sino.unsqueeze(-1) if layout == _LAYOUT_VAE else sino.permute(1, 2, 0)
        return out if ctx.in_dtype is torch.float32 else out.to(ctx.in_dtype)

    @staticmethod
    def backward(ctx, gout):
        from .recon import siddon_backproject
        g = gout.reshape(gout.shape[0], gout.shape[1], gout.shape[2]) if ctx.layout == _LAYOUT_VAE else gout.permute(2, 0, 1)
        gimg = siddon_backproject(g, ctx.theta, ctx.grid[0], ctx.grid[1])          # [S][X][Y]
        if ctx.layout == _LAYOUT_VAE:
            gimg = gimg.unsqueeze(-1)
        elif ctx.layout == _LAYOUT_DIM3:
            gimg = gimg.permute(1, 2, 0)
        else:
            gimg = gimg[0]
        return (gimg if ctx.in_dtype is torch.float32 else gimg.to(ctx.in_dtype)), None, None, None


def _project_siddon(phantom, theta, pad, dim, integrate_vae):
    if not isinstance(phantom, torch.Tensor):
        raise TypeError(f"phantom must be a torch.Tensor on a HIP device (got {type(phantom).__name__})")
    if phantom.device.type != "cuda":
        raise _lib.RadonLibraryError(
            f"phantom lives on {phantom.device}: the projector runs on a HIP device only; there is no CPU path")
    if not phantom.dtype.is_floating_point:
        raise TypeError(f"phantom must be floating point (got {phantom.dtype})")
    if integrate_vae:
        if phantom.dim() != 4 or phantom.shape[3] != 1:
            raise ValueError("integrate_vae=True expects batch_size x img_size_x x img_size_y x 1 "
                             f"(got {tuple(phantom.shape)})")
        layout = _LAYOUT_VAE
    elif dim in (2, 3):
        if phantom.dim() != dim:
            raise ValueError(f"dim={dim} expects a {dim}-D phantom (got {tuple(phantom.shape)})")
        layout = _LAYOUT_DIM3 if dim == 3 else _LAYOUT_DIM2
    else:
        raise ValueError(f"dim must be 2 or 3 (got {dim})")
    theta_host = np.ascontiguousarray(np.asarray(theta.detach().cpu() if isinstance(theta, torch.Tensor) else theta,
                                                 dtype=np.float32))
    with torch.cuda.device(phantom.device):
        return _SiddonProject.apply(phantom, theta_host, bool(pad), layout)


def project_tf_fast(phantom, theta, pad=False, dim=3, integrate_vae=False, *, interp="nearest",
                    backward="tf_compat", model="rotate"):
    """Vectorised Radon forward, ctvae/forward_functions.py:80-123.

    phantom: img_size_x x img_size_y x img_size_z (dim=3), img_size_x x img_size_y (dim=2), or
    batch_size x img_size_x x img_size_y x 1 (integrate_vae=True).  Returns angles x P x Z, angles x P x 1, or
    batch_size x angles x P x 1.  Every slice is rotated by -theta (nearest neighbour, zero fill) and summed
    over image rows.

    model="siddon" (keyword-only extension): the same layouts through TomoPy's ray-driven projector -- what
    the reference's data were MADE with (scripts/images_to_sinograms.py:62-66) -- differentiable through its exact
    transpose; `interp` and `backward` do not apply to it."""
    if model == "siddon":
        return _project_siddon(phantom, theta, pad, dim, integrate_vae)
    if model != "rotate":
        raise ValueError(f"model must be 'rotate' or 'siddon' (got {model!r})")
    return _project(phantom, theta, pad, dim, integrate_vae, interp, backward)


def project_tf_low_mem(phantom, theta, pad=False, *, interp="bilinear", backward="tf_compat"):
    """Per-angle Radon forward, ctvae/forward_functions.py:49-78: img_size_x x img_size_y x img_size_z ->
    angles x img_size_y x img_size_z, bilinear interpolation."""
    return _project(phantom, theta, pad, 3, False, interp, backward)
