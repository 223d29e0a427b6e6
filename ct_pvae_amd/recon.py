"""Initial reconstructions on the GPU: the `tomopy.recon` calls of CT_PVAE's setup path.

The reference makes the encoder's input channels with TomoPy on the CPU (ctvae/helper_functions.py:477-529):

    tomopy.recon(proj_sample_expand, theta, center=None, sinogram_order=True, algorithm=algorithm)       # :503, per algorithm
    tomopy.recon(mask_expand, theta, center=None, sinogram_order=True, algorithm='fbp', filter_name='none')   # :514

with `algorithm` one of 'gridrec' (the default, ctvae/main_ct_vae.py:111-112), 'sirt', 'tv', 'fbp' (README.md:80,221).
Here `recon` takes the same arguments and runs on the hand-written kernels behind include/ctpvae_radon.h:

    'fbp'      libtomo's fbp.c: the ray-driven back-projection of the sinogram = the TRANSPOSE of tomopy.project
               (ctpvae_siddon_bwd_f32), after an optional row filter; 'none' (tomopy's default filter_name) is what the
               reference asks for the mask channel.  [3P-recalled: TomoPy 1.11.0]
    'sirt'     libtomo's sirt.c update rule on the same operator pair (ctpvae_siddon_fwd_f32 / _bwd_f32 / _rownorm_f32):
               recon += (A^T ((data - A recon) / sum_dist2)) / sum_dist, tomopy's defaults num_iter=1, init 1e-6.
    'gridrec'  libtomo's gridrec.c (round 3): zero-padded 1-D FFTs of the projections, filter x centre phase, convolution onto
               a Cartesian frequency grid with the prolate-spheroidal window, 2-D FFT, window correction -- csrc/gridrec.hip
               (hand-written LDS FFTs, the convolution as a deterministic gather), restated in oracle/gridrec_oracle.c.
               filter_name defaults to tomopy's per-algorithm default 'parzen' (algorithm.py _get_algorithm_kwargs); 'none',
               'shepp', 'cosine', 'hann', 'hamming', 'ramlak', 'butterworth' are built too.  [3P-recalled: TomoPy 1.11.0]
               (The round-2 stand-in -- float64 ramp-filtered back-projection on the same grid -- stays reachable under its own
               name: algorithm='fbp', filter_name='ramp'; it is NOT a tomopy filter name.)
    'tv'       STAND-IN, flagged: total-variation regularised reconstruction on the same operator pair by the diagonally
               preconditioned Chambolle-Pock iteration (Pock & Chambolle 2011: step sizes from the operator's own row and
               column sums -- sirt.c's sum_dist2-free weights -- so nothing has to be tuned), reg_par[0] = the TV weight.
               libtomo's tv.c (also a primal-dual TV scheme) is NOT restated: same family, not TomoPy's numbers.

Grid and centre follow tomopy: num_gridx = num_gridy = detector width, center = width / 2; callers crop
(ctvae/helper_functions.py:420-430, `crop`).  There is no CPU path."""
import ctypes

import numpy as np
import torch

from . import forward_functions as _fwd  # noqa: E402  (NaN-poisoned outputs in test sessions)

from . import _lib
from .dataset_io import compare, crop
from .fbp import iradon, ramp_filter
from .forward_functions import _stream_ptr
from .helper_functions import _siddon_forward, _siddon_tables

__all__ = ["recon", "siddon_backproject", "crop", "evaluate_sinogram", "ALGORITHMS", "GRIDREC_FILTERS"]

ALGORITHMS = ("fbp", "sirt", "gridrec", "tv")
GRIDREC_FILTERS = {"none": 0, "shepp": 1, "cosine": 2, "hann": 3, "hamming": 4, "ramlak": 5, "parzen": 6, "butterworth": 7}
# tomopy/recon/algorithm.py _get_algorithm_kwargs [3P-recalled]: the default filter_name is per algorithm
_DEFAULT_FILTER = {"gridrec": "parzen", "fbp": "none"}
_GRIDREC_TABLES = {}


def _gridrec(data, theta, gx, gy, filter_name, filter_par):
    """data [oy][dt][dx] float32 on the device -> [oy][gx][gy] (tomopy.recon(algorithm='gridrec'), center = dx / 2)."""
    lib = _lib.load()
    oy, dt, dx = data.shape
    if filter_name not in GRIDREC_FILTERS:
        raise ValueError(f"recon: gridrec filter_name must be one of {sorted(GRIDREC_FILTERS)} (got {filter_name!r})")
    th = np.ascontiguousarray(np.asarray(theta, dtype=np.float32))
    par = np.ascontiguousarray(np.asarray([0.5, 8.0] if filter_par is None else filter_par, dtype=np.float32))
    key = (th.tobytes(), dx, filter_name, par.tobytes(), str(data.device))
    tab = _GRIDREC_TABLES.get(key)
    if tab is None:
        nbytes = lib.ctpvae_gridrec_tables_bytes(dt, dx)
        _lib.check(nbytes, "gridrec_tables_bytes")
        host = np.empty(int(nbytes), dtype=np.uint8)
        _lib.check(lib.ctpvae_gridrec_tables_host_f32(dt, dx, ctypes.c_float(dx / 2.0), th.ctypes.data, GRIDREC_FILTERS[filter_name],
                                                      par.ctypes.data, host.ctypes.data), "gridrec_tables")
        tab = torch.from_numpy(host).to(data.device)
        if len(_GRIDREC_TABLES) >= 16:
            _GRIDREC_TABLES.pop(next(iter(_GRIDREC_TABLES)))
        _GRIDREC_TABLES[key] = tab
    need = lib.ctpvae_gridrec_workspace_bytes(oy, dt, dx)
    _lib.check(need, "gridrec_workspace_bytes")
    ws = torch.empty(int(need), dtype=torch.uint8, device=data.device)
    out = _fwd._new_output((oy, gx, gy), torch.float32, data.device)
    _lib.check(lib.ctpvae_gridrec_f32(data.data_ptr(), oy, dt, dx, tab.data_ptr(), gx, gy, ws.data_ptr(), out.data_ptr(),
                                      _stream_ptr()), "gridrec")
    return out


def _as_device_f32(t, what):
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise _lib.RadonLibraryError(f"{what} must be a tensor on a HIP device; there is no CPU path")
    return t.to(torch.float32).contiguous()


def _project(x, tables, dx):
    """A x: [oy][gx][gy] -> [oy][dt][dx] (tomopy.project on the reconstruction grid, center = dx / 2)."""
    return _siddon_forward(x, tables, dx)


_BP_WORKSPACES = {}          # (tables' device pointer, grid, angles, dx, slices) -> (tables, workspace); a few entries


def _bp_workspace(tables, oy, gx, gy, dt, dx, device):
    """The back-projector's workspace with its geometry part (ray table, slow-path flags) filled in.  Kept per geometry --
    iradon_all reconstructs the same stack with several algorithms (ctvae/helper_functions.py:489-516) --; calls that share an
    entry must be on one stream (the workspace also holds the call's scratch image)."""
    lib = _lib.load()
    sin_t, cos_t, quad = tables
    key = (sin_t.data_ptr(), int(gx), int(gy), int(dt), int(dx), int(oy), str(device))
    hit = _BP_WORKSPACES.get(key)
    if hit is not None and hit[0] is tables:
        return hit[1]
    need = lib.ctpvae_siddon_bwd_workspace_bytes(oy, gx, gy, dt, dx)
    _lib.check(need, "siddon_bwd_workspace_bytes")
    ws = torch.empty(int(need), dtype=torch.uint8, device=device)
    _lib.check(lib.ctpvae_siddon_bwd_prepare_f32(gx, gy, sin_t.data_ptr(), cos_t.data_ptr(), quad.data_ptr(), dt, dx,
                                                 ctypes.c_float(dx / 2.0), ws.data_ptr(), _stream_ptr()), "siddon_bwd_prepare")
    if len(_BP_WORKSPACES) >= 4:
        _BP_WORKSPACES.pop(next(iter(_BP_WORKSPACES)))
    _BP_WORKSPACES[key] = (tables, ws)
    return ws


def _backproject(data, tables, gx, gy, ws=None, colsum=None, out=None):
    """A^T y: [oy][dt][dx] -> [oy][gx][gy]; with `colsum` and `out`: out += A^T y / colsum where colsum != 0 (SIRT's update)."""
    lib = _lib.load()
    sin_t, cos_t, quad = tables
    oy, dt, dx = data.shape
    if ws is None:
        ws = _bp_workspace(tables, oy, gx, gy, dt, dx, data.device)
    if out is None:
        out = _fwd._new_output((oy, gx, gy), torch.float32, data.device)
    _lib.check(lib.ctpvae_siddon_bwd_prepared_f32(data.data_ptr(), oy, gx, gy, sin_t.data_ptr(), cos_t.data_ptr(), quad.data_ptr(),
                                                  dt, dx, ctypes.c_float(dx / 2.0), ws.data_ptr(),
                                                  colsum.data_ptr() if colsum is not None else None, out.data_ptr(),
                                                  _stream_ptr()), "siddon_bwd")
    return out


def siddon_backproject(data, theta, num_gridx=None, num_gridy=None):
    """The transpose of create_sinograms: data [slices][angles][dx] -> [slices][num_gridx][num_gridy] (default: dx x dx),
    i.e. tomopy.recon(data, theta, center=None, sinogram_order=True, algorithm='fbp', filter_name='none') without its 1e-6."""
    data = _as_device_f32(data, "data")
    if data.dim() != 3:
        raise ValueError(f"expected slices x angles x dx (got {tuple(data.shape)})")
    if data.shape[1] != len(theta):
        raise ValueError("The given ``theta`` does not match the number of projections in ``data``.")
    dx = data.shape[2]
    gx, gy = int(num_gridx or dx), int(num_gridy or dx)
    if data.shape[0] == 0:
        return data.new_empty((0, gx, gy))
    with torch.cuda.device(data.device):
        return _backproject(data, _siddon_tables(theta, data.device), gx, gy)


def _sirt(data, tables, gx, gy, num_iter, init):
    """libtomo sirt.c: per iteration ONE forward launch (its store is the ray's update factor (data - A x) / sum dist^2) and ONE
    back-projector launch (its store is x += A^T upd / sum_dist); the row and column weights are geometry, computed once."""
    lib = _lib.load()
    sin_t, cos_t, quad = tables
    oy, dt, dx = data.shape
    rn2 = _fwd._new_output((dt, dx), torch.float32, data.device)
    _lib.check(lib.ctpvae_siddon_rownorm_f32(gx, gy, sin_t.data_ptr(), cos_t.data_ptr(), quad.data_ptr(), dt, dx,
                                             ctypes.c_float(dx / 2.0), rn2.data_ptr(), _stream_ptr()), "siddon_rownorm")
    ws = _bp_workspace(tables, oy, gx, gy, dt, dx, data.device)
    colsum = _backproject(torch.ones((1, dt, dx), dtype=torch.float32, device=data.device), tables, gx, gy, ws=ws)[0]   # sum_dist
    x = init.contiguous().clone()
    upd = _fwd._new_output(data.shape, data.dtype, data.device)
    for _ in range(int(num_iter)):
        _siddon_forward(x, tables, dx, meas=data, rn2=rn2, out=upd)
        _backproject(upd, tables, gx, gy, ws=ws, colsum=colsum, out=x)
    return x


_TV_WARNED = False


def _tv(data, tables, gx, gy, num_iter, init, lam):
    """min_x 1/2 |A x - b|^2 + lam TV(x) by preconditioned Chambolle-Pock (Pock & Chambolle, ICCV 2011, alpha = 1):
    dual steps 1 / (row sums of |K|), primal steps 1 / (column sums of |K|) for K = (A; grad).  STAND-IN for tomopy's 'tv'
    (see the module docstring; warns once).  Round 4: an iteration is TWO projector launches -- the forward stores the data
    term's dual step, the back-projector's store does the TV dual step, the divergence, the primal step and the over-relaxation
    (ctpvae_siddon_fwd_ws_tv_dual_f32 / ctpvae_siddon_bwd_tv_primal_f32) -- and equals oracle.tv_standin bit for bit."""
    global _TV_WARNED
    if not _TV_WARNED:
        import warnings
        warnings.warn("recon(algorithm='tv') is a STAND-IN: total-variation reconstruction by preconditioned Chambolle-Pock on the "
                      "TomoPy-style projector pair, not libtomo's tv.c -- the same family, not TomoPy's numbers", stacklevel=3)
        _TV_WARNED = True
    lib = _lib.load()
    sin_t, cos_t, quad = tables
    oy, dt, dx = data.shape
    dev = data.device
    ones_img = torch.ones((1, gx, gy), dtype=torch.float32, device=dev)
    rowsum = _project(ones_img, tables, dx)[0]                                          # sum_n dist[n] of every ray
    ws = _bp_workspace(tables, oy, gx, gy, dt, dx, dev)
    colsum = _backproject(torch.ones((1, dt, dx), dtype=torch.float32, device=dev), tables, gx, gy)[0]
    sigma_a = torch.where(rowsum > 0, 1.0 / rowsum.clamp_min(1e-30), torch.zeros_like(rowsum)).contiguous()
    tau = (1.0 / (colsum + 4.0)).contiguous()                                           # |grad| has column sums <= 4, row sums 2
    need = lib.ctpvae_siddon_fwd_workspace_bytes(oy, gx, gy)
    _lib.check(need, "siddon_fwd_workspace_bytes")
    fws = torch.empty(int(need), dtype=torch.uint8, device=dev) if need else None
    x = init.contiguous().clone()
    xbar, xbar2 = x.clone(), _fwd._new_output(x.shape, x.dtype, x.device)
    p = torch.zeros_like(data)
    qx, qy, qx2, qy2 = torch.zeros_like(x), torch.zeros_like(x), _fwd._new_output(x.shape, x.dtype, x.device), _fwd._new_output(x.shape, x.dtype, x.device)
    center, sp = ctypes.c_float(dx / 2.0), _stream_ptr()
    for _ in range(int(num_iter)):
        _lib.check(lib.ctpvae_siddon_fwd_ws_tv_dual_f32(xbar.data_ptr(), oy, gx, gy, sin_t.data_ptr(), cos_t.data_ptr(), quad.data_ptr(),
                                                        dt, dx, center, data.data_ptr(), sigma_a.data_ptr(),
                                                        fws.data_ptr() if fws is not None else None, p.data_ptr(), sp), "siddon_fwd_tv_dual")
        _lib.check(lib.ctpvae_siddon_bwd_tv_primal_f32(p.data_ptr(), oy, gx, gy, sin_t.data_ptr(), cos_t.data_ptr(), quad.data_ptr(), dt, dx,
                                                       center, ws.data_ptr(), tau.data_ptr(), ctypes.c_float(lam), x.data_ptr(),
                                                       xbar.data_ptr(), xbar2.data_ptr(), qx.data_ptr(), qy.data_ptr(), qx2.data_ptr(),
                                                       qy2.data_ptr(), sp), "siddon_bwd_tv_primal")
        xbar, xbar2, qx, qx2, qy, qy2 = xbar2, xbar, qx2, qx, qy2, qy
    return x


def recon(tomo, theta, center=None, sinogram_order=False, algorithm=None, init_recon=None, num_gridx=None, num_gridy=None,
          num_iter=1, filter_name=None, filter_par=None, reg_par=None, **kwargs):
    """tomopy.recon's call shape for the algorithms above.  tomo: [angles][slices][dx] (sinogram_order=False) or
    [slices][angles][dx] (True), a float tensor on a HIP device.  Returns [slices][num_gridx][num_gridy] float32.
    filter_name None = tomopy's default for the algorithm ('parzen' for gridrec, 'none' for fbp)."""
    if algorithm not in ALGORITHMS:
        raise ValueError(f"recon: unknown algorithm {algorithm!r}; available: {ALGORITHMS}")
    if filter_name is None:
        filter_name = _DEFAULT_FILTER.get(algorithm, "none")
    data = _as_device_f32(tomo, "tomo")
    if data.dim() != 3:
        raise ValueError(f"tomo must be 3-D (got {tuple(data.shape)})")
    if not sinogram_order:
        data = data.permute(1, 0, 2).contiguous()
    oy, dt, dx = data.shape
    if dt != len(theta):
        raise ValueError("The given ``theta`` does not match the number of projections in ``tomo``.")
    if center is not None and float(center) != dx / 2.0:
        raise NotImplementedError("recon: only center=None (the detector's middle, as every reference call passes) is built")
    gx, gy = int(num_gridx or dx), int(num_gridy or dx)
    if oy == 0:
        return data.new_empty((0, gx, gy))
    with torch.cuda.device(data.device):
        if algorithm == "gridrec":
            return _gridrec(data, theta, gx, gy, filter_name, filter_par)
        if algorithm == "fbp" and filter_name == "ramp":
            # round 2's gridrec stand-in under its own name: float64 ramp-filtered back-projection on tomopy's grid (an
            # extension: 'ramp' is not a tomopy filter name)
            return iradon(data.to(torch.float64), np.asarray(theta, dtype=np.float64), gx, gy, ramp_filter(dx),
                          tomopy_geometry=True).to(torch.float32)
        tables = _siddon_tables(theta, data.device)
        if algorithm == "fbp":
            if filter_name != "none":
                raise NotImplementedError("recon: 'fbp' is built with filter_name='none' (the reference's only use of it) and "
                                          "the extension 'ramp'")
            out = _backproject(data, tables, gx, gy)
            return out if init_recon is None else out + init_recon
        init = torch.full((oy, gx, gy), 1e-6, dtype=torch.float32, device=data.device) if init_recon is None else \
            _as_device_f32(init_recon, "init_recon")
        if algorithm == "tv":
            lam = float(np.asarray(reg_par if reg_par is not None else 1.0, dtype=np.float64).reshape(-1)[0])   # tomopy: ones(10)
            if not lam > 0:
                raise ValueError("recon: reg_par[0] (the TV weight) must be positive")
            return _tv(data, tables, gx, gy, num_iter, init, lam)
        return _sirt(data, tables, gx, gy, num_iter, init)


def evaluate_sinogram(actual_sinogram, computed_sinogram, partial_noisy_sinogram, mask, theta, final_x, final_y,
                      algorithm="sirt", verbose=True):
    """ctvae/helper_functions.py:433-475: reconstruct the actual, the predicted and the masked noisy sinogram ([angles][P]
    each) with `algorithm`, crop to final_x x final_y and compare the last two against the first -- (MSE, SSIM, PSNR) twice,
    then the three reconstructions.  Sinograms and mask may be numpy arrays or tensors; the reconstructions run on the GPU."""
    dev = torch.device("cuda", torch.cuda.current_device())

    def to_dev(a):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))
        return t.to(device=dev, dtype=torch.float32)

    theta = np.asarray(theta, dtype=np.float32)
    mask_h = mask.detach().cpu().numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)
    used = mask_h > 0
    recon0 = recon(to_dev(actual_sinogram)[:, None, :], theta, center=None, algorithm=algorithm, sinogram_order=False)[0]
    recon1 = recon(to_dev(computed_sinogram)[:, None, :], theta, center=None, algorithm=algorithm, sinogram_order=False)[0]
    noisy = to_dev(partial_noisy_sinogram)[torch.from_numpy(used).to(dev)] / to_dev(mask_h[used])[:, None]
    recon2 = recon(noisy[:, None, :], theta[used], center=None, algorithm=algorithm, sinogram_order=False)[0]
    recon0, recon1, recon2 = (crop(r, final_x, final_y).cpu().numpy() for r in (recon0, recon1, recon2))
    if verbose:
        print("Predicted")
    predicted_err = list(compare(recon0, recon1, verbose=verbose))
    if verbose:
        print("Noisy")
    noisy_err = list(compare(recon0, recon2, verbose=verbose))
    return predicted_err, noisy_err, recon0, recon1, recon2
