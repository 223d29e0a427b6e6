"""ctypes binding of oracle/radon_oracle.c -- the CPU restatement of CT_PVAE's Radon hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
ct_pvae_amd/.  PARITY UNPINNED (see the header of radon_oracle.c): nothing the reference ships pins the
third-party projectors numerically beyond the 2x2 toy case and analytic identities.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTPVAE_ORACLE_LIB") or os.path.join(_HERE, "libradon_oracle.so")   # (override: the sanitizer build)
NEAREST, BILINEAR = 0, 1

_lib = None
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i = ctypes.c_int


def build(force=False):
    """Compile the restatement with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, "radon_oracle.c"), os.path.join(_HERE, "gridrec_oracle.c")]
    if "CTPVAE_ORACLE_LIB" in os.environ:
        return LIB_PATH
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        L.oracle_num_proj_pix.restype = _i
        L.oracle_num_proj_pix.argtypes = [_i, _i]
        L.oracle_pad_amounts.argtypes = [_i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]
        L.oracle_pad_phantom.argtypes = [_f32p, _i, _i, _i, _i, _i, _i, _i, _f32p]
        L.oracle_rotate_transforms.argtypes = [_f32p, _i, _i, _i, _f32p]
        L.oracle_invert_transforms.argtypes = [_f32p, _i, _f32p]
        L.oracle_rotate_fwd.argtypes = [_f32p, _i, _i, _i, _i, _i, _i, _i, _f32p, _i, _i, _f32p]
        L.oracle_rotate_fwd_f64.argtypes = [_f64p, _i, _i, _i, _i, _i, _i, _i, _f32p, _i, _i, _f64p]
        L.oracle_rotate_fwd_tiled.argtypes = [_f32p, _i, _i, _i, _i, _i, _i, _i, _f32p, _i, _i, _i, _f32p]
        L.oracle_rotate_fwd_tiled.restype = _i
        L.oracle_rotate_fwd_tiled_interp.argtypes = [_f32p, _i, _i, _i, _i, _i, _i, _i, _f32p, _i, _i, _i, _i, _f32p]
        L.oracle_rotate_fwd_tiled_interp.restype = _i
        L.oracle_rotate_bwd_tfcompat.argtypes = [_f32p, _i, _i, _i, _i, _f32p, _i, _i, _i, _i, _i, _f32p]
        L.oracle_rotate_bwd_exact.argtypes = [_f32p, _i, _i, _i, _i, _f32p, _i, _i, _i, _i, _i, _f32p]
        L.oracle_siddon_dx.restype = _i
        L.oracle_siddon_dx.argtypes = [_i, _i, _i]
        L.oracle_siddon_project.argtypes = [_f32p, _i, _i, _i, _f32p, _i, _i, ctypes.c_float, _f32p]
        L.oracle_fbp_filter.argtypes = [_f64p, _i, _i, _f64p, ctypes.c_void_p, _f64p]
        L.oracle_fbp_backproject.argtypes = [_f64p, _i, _i, _i, _f64p, _i, _i, _f64p]
        L.oracle_iradon.argtypes = [_f64p, _i, _i, _i, _f64p, _i, _i, _f64p, ctypes.c_void_p, _f64p]
        L.oracle_loglik.argtypes = [_f32p, _f32p, _f32p, _i, _i, _i, ctypes.c_float, ctypes.c_float, _f32p]
        L.oracle_siddon_backproject.argtypes = [_f32p, _i, _i, _i, _f32p, ctypes.c_float, _i, _i, _f32p]
        L.oracle_sirt.argtypes = [_f32p, _i, _i, _i, _f32p, ctypes.c_float, _i, _i, _i, _f32p]
        L.oracle_philox4x32_10.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_poisson_count.restype = ctypes.c_double
        L.oracle_poisson_count.argtypes = [ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64]
        L.oracle_poisson_measure.argtypes = [_f32p, _f32p, _i, _i, _i, ctypes.c_float, ctypes.c_uint64, _f32p]
        L.oracle_gridrec_pdim.restype = _i
        L.oracle_gridrec_pdim.argtypes = [_i]
        L.oracle_gridrec.restype = _i
        L.oracle_gridrec.argtypes = [_f32p, _i, _i, _i, ctypes.c_float, _f32p, _i, _i, _i, _f32p, _f32p]
        L.oracle_gridrec_pswf_tables.argtypes = [_i, _f32p, _f32p]
        L.oracle_gridrec_filter.restype = ctypes.c_float
        L.oracle_gridrec_filter.argtypes = [_i, ctypes.c_float, _i, _f32p]
        for name in ("oracle_series_log_public", "oracle_series_exp_neg_public"):
            getattr(L, name).restype = ctypes.c_double
            getattr(L, name).argtypes = [ctypes.c_double]
        _lib = L
    return _lib


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- a1 -----------------------------------------------------------------------------------------------
def num_proj_pix(nx, ny):
    return lib().oracle_num_proj_pix(nx, ny)


def pad_amounts(n, P):
    lo, hi = _i(), _i()
    lib().oracle_pad_amounts(n, P, ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value


class Geometry:
    """Canvas geometry of pad_phantom + rotate for slices of H x W (pad=True: square P x P canvas)."""

    def __init__(self, H, W, pad):
        self.H, self.W = H, W
        if pad:
            P = num_proj_pix(H, W)
            self.PH = self.PW = P
            self.py, self.px = pad_amounts(H, P)[0], pad_amounts(W, P)[0]
        else:
            self.PH, self.PW, self.py, self.px = H, W, 0, 0


def pad_phantom(img, geom):
    img = _c32(img)
    out = np.empty((img.shape[0], geom.PH, geom.PW), np.float32)
    lib().oracle_pad_phantom(img, img.shape[0], geom.H, geom.W, geom.py, geom.px, geom.PH, geom.PW, out)
    return out


# ---- a3 / a4 tables -----------------------------------------------------------------------------------
def rotate_transforms(theta, H, W):
    """Rows tfa.image.rotate(images, -theta) builds for an H x W image.  theta: radians, cast to fp32 first."""
    ang = _c32(-_c32(theta))
    T = np.empty((ang.size, 8), np.float32)
    lib().oracle_rotate_transforms(ang, ang.size, H, W, T)
    return T


def invert_transforms(T8):
    T8 = _c32(T8)
    out = np.empty_like(T8)
    lib().oracle_invert_transforms(T8, T8.shape[0], out)
    return out


# ---- a2 / a5 / a4 -------------------------------------------------------------------------------------
def rotate_fwd(img, geom, T8, interp=NEAREST):
    img, T8 = _c32(img), _c32(T8)
    S, A = img.shape[0], T8.shape[0]
    sino = np.empty((S, A, geom.PW), np.float32)
    lib().oracle_rotate_fwd(img, S, geom.H, geom.W, geom.PH, geom.PW, geom.py, geom.px, T8, A, interp, sino)
    return sino


def tile_shape(H, W, interp=NEAREST):
    """The tile shape the build's tiled forward uses for H x W slices (include/ctpvae_radon.h: the shape fixes the association
    of the fp32 row sum, so it is part of the contract) -- RESTATED here, not read from the library under test:
    64 wide (or W), th = ceil(H / ceil(H / R)) tall: equal rows of tiles, R = 128 (nearest), 96 (bilinear)."""
    r = 128 if interp == NEAREST else 96
    return (-(-H // -(-H // r)), min(W, 64))


def rotate_fwd_f64(img, geom, T8, interp=NEAREST):
    """rotate_fwd on float64 pixels: fp32 coordinates and weights, float64 products and row sum (TF's T = double)."""
    img = np.ascontiguousarray(img, dtype=np.float64)
    T8 = _c32(T8)
    S, A = img.shape[0], T8.shape[0]
    sino = np.empty((S, A, geom.PW), np.float64)
    lib().oracle_rotate_fwd_f64(img, S, geom.H, geom.W, geom.PH, geom.PW, geom.py, geom.px, T8, A, interp, sino)
    return sino


def rotate_fwd_tiled(img, geom, T8, tile=(128, 128), interp=NEAREST):
    """Same samples as rotate_fwd, summed tile by tile (the association the build uses for slices > LDS; a bilinear sample
    belongs to the tile of its floor tap)."""
    img, T8 = _c32(img), _c32(T8)
    S, A = img.shape[0], T8.shape[0]
    sino = np.empty((S, A, geom.PW), np.float32)
    rc = lib().oracle_rotate_fwd_tiled_interp(img, S, geom.H, geom.W, geom.PH, geom.PW, geom.py, geom.px, T8, A, int(interp),
                                              int(tile[0]), int(tile[1]), sino)
    if rc:
        raise ValueError("rotate_fwd_tiled: bad tile size")
    return sino


def rotate_bwd_tfcompat(gsino, geom, Tinv8, interp=NEAREST):
    gsino, Tinv8 = _c32(gsino), _c32(Tinv8)
    S, A = gsino.shape[0], gsino.shape[1]
    gimg = np.empty((S, geom.H, geom.W), np.float32)
    lib().oracle_rotate_bwd_tfcompat(gsino, S, A, geom.PH, geom.PW, Tinv8, interp, geom.H, geom.W, geom.py, geom.px,
                                     gimg)
    return gimg


def rotate_bwd_exact(gsino, geom, T8, interp=NEAREST):
    gsino, T8 = _c32(gsino), _c32(T8)
    S, A = gsino.shape[0], gsino.shape[1]
    gimg = np.empty((S, geom.H, geom.W), np.float32)
    lib().oracle_rotate_bwd_exact(gsino, S, A, geom.PH, geom.PW, T8, interp, geom.H, geom.W, geom.py, geom.px, gimg)
    return gimg


def project_tf_fast(phantom, theta, pad=False, dim=3, integrate_vae=False, interp=NEAREST):
    """Reference layouts of ctvae/forward_functions.py:80-123 on numpy arrays."""
    phantom = np.asarray(phantom, dtype=np.float32)
    if integrate_vae:
        slices = phantom[..., 0]
    elif dim == 3:
        slices = np.transpose(phantom, (2, 0, 1))
    else:
        slices = phantom[None]
    geom = Geometry(slices.shape[1], slices.shape[2], pad)
    sino = rotate_fwd(slices, geom, rotate_transforms(theta, geom.PH, geom.PW), interp)
    return sino[..., None] if integrate_vae else np.transpose(sino, (1, 2, 0))


# ---- a7 -----------------------------------------------------------------------------------------------
def siddon_project(obj, theta, pad=True):
    """tomopy.project(obj, theta, center=None, emission=True, pad=pad, sinogram_order=False) -> [dt][oy][dx]."""
    obj, theta = _c32(obj), _c32(theta)
    oy, ox, oz = obj.shape
    dx = lib().oracle_siddon_dx(ox, oz, 1 if pad else 0)
    data = np.empty((oy, theta.size, dx), np.float32)
    lib().oracle_siddon_project(obj, oy, ox, oz, theta, theta.size, dx, dx / 2.0, data)
    return np.swapaxes(data, 0, 1).copy()


def create_sinogram(img, theta, pad=True):
    """ctvae/helper_functions.py:33-38."""
    return np.squeeze(siddon_project(np.asarray(img)[None], theta, pad=pad), axis=1)


def siddon_backproject(data, theta, ngridx=None, ngridy=None, init=0.0):
    """tomopy.recon(data, theta, center=None, sinogram_order=True, algorithm='fbp', filter_name='none'): data [oy][dt][dx]
    (one sinogram per slice) -> [oy][ngridx][ngridy], grid = detector width by default, added to `init`."""
    data, theta = _c32(data), _c32(theta)
    oy, dt, dx = data.shape
    gx, gy = int(ngridx or dx), int(ngridy or dx)
    recon = np.full((oy, gx, gy), init, np.float32)
    lib().oracle_siddon_backproject(data, oy, dt, dx, theta, dx / 2.0, gx, gy, recon)
    return recon


def sirt(data, theta, num_iter=1, ngridx=None, ngridy=None, init=1e-6):
    """tomopy.recon(data, theta, center=None, sinogram_order=True, algorithm='sirt', num_iter=num_iter)."""
    data, theta = _c32(data), _c32(theta)
    oy, dt, dx = data.shape
    gx, gy = int(ngridx or dx), int(ngridy or dx)
    recon = np.full((oy, gx, gy), init, np.float32)
    lib().oracle_sirt(data, oy, dt, dx, theta, dx / 2.0, gx, gy, int(num_iter), recon)
    return recon


# ---- a6 -----------------------------------------------------------------------------------------------
def iradon(sinogram, theta, x_size, y_size, filter_1d):
    sino, theta = _c64(sinogram), _c64(theta)
    B, A, P = sino.shape
    filt = np.asarray(filter_1d).reshape(-1)
    fre = _c64(filt.real)
    fim = _c64(filt.imag) if np.iscomplexobj(filt) else None
    recon = np.empty((B, x_size, y_size), np.float64)
    lib().oracle_iradon(sino, B, A, P, theta, x_size, y_size, fre,
                        fim.ctypes.data if fim is not None else None, recon)
    return recon


# ---- a8 -----------------------------------------------------------------------------------------------
def loglik(proj, mask, x, pnm, eps):
    proj, mask, x = _c32(proj), _c32(mask), _c32(x)
    B, A, P = proj.shape
    out = np.empty_like(proj)
    lib().oracle_loglik(proj, mask, x, B, A, P, pnm, eps, out)
    return out


GRIDREC_FILTERS = {"none": 0, "shepp": 1, "cosine": 2, "hann": 3, "hamming": 4, "ramlak": 5, "parzen": 6, "butterworth": 7}


def gridrec(data, theta, filter_name="parzen", filter_par=(0.5, 8.0), ngridx=None, ngridy=None, center=None):
    """tomopy.recon(data, theta, center=None, sinogram_order=True, algorithm='gridrec') [3P-recalled: TomoPy 1.11.0
    gridrec.c + algorithm.py defaults: filter 'parzen', grid = detector width, center = width / 2]; data [dy][dt][dx]."""
    data, theta = _c32(data), _c32(theta)
    dy, dt, dx = data.shape
    gx, gy = int(ngridx or dx), int(ngridy or dx)
    out = np.empty((dy, gx, gy), np.float32)
    par = _c32(np.asarray(filter_par, np.float32))
    rc = lib().oracle_gridrec(data, dy, dt, dx, float(dx / 2.0 if center is None else center), theta, gx, gy,
                              GRIDREC_FILTERS[filter_name], par, out)
    if rc:
        raise ValueError("oracle_gridrec: the grid must not exceed the padded detector width")
    return out


def gridrec_pswf_tables(dx):
    pdim = lib().oracle_gridrec_pdim(dx)
    wtbl, winv = np.empty(513, np.float32), np.empty(pdim - 1, np.float32)
    lib().oracle_gridrec_pswf_tables(pdim // 2 - 1, wtbl, winv)
    return wtbl, winv


def tv_standin(data, theta, num_iter=1, lam=1.0, init=1e-6, ngridx=None, ngridy=None):
    """The build's STAND-IN for tomopy.recon(algorithm='tv') -- NOT libtomo's tv.c (ct_pvae_amd/recon.py says so): total-variation
    reconstruction min_x 1/2 |A x - b|^2 + lam TV(x) by the diagonally preconditioned Chambolle-Pock iteration (Pock & Chambolle
    2011, alpha = 1) on K = (A; grad), A = tomopy.project restated above.  This function IS the statement of the iteration the HIP
    epilogues compute (csrc/siddon.hip siddon_fwd_store mode 2, TvPrimal), operation by operation in fp32:
        sigma = 1 / rowsum(A) (0 where a ray misses the grid);  tau = 1 / (colsum(A) + 4)
        p     <- (p + sigma (A xbar - b)) / (1 + sigma)
        q'    = q + 0.5 grad xbar;  q <- q' / (max(sqrt(qx'^2 + qy'^2), lam) / lam)         grad: forward differences, 0 at the far edges
        x_new = x - tau (A^T p - div q);  div q = ((qx[i][j] - qx[i-1][j]) + qy[i][j]) - qy[i][j-1], missing terms left out
        xbar  <- 2 x_new - x;  x <- x_new
    data [oy][dt][dx] -> [oy][gx][gy]."""
    data, theta = _c32(data), _c32(theta)
    oy, dt, dx = data.shape
    gx, gy = int(ngridx or dx), int(ngridy or dx)
    f32 = np.float32
    rowsum = _project_grid(np.ones((1, gx, gy), f32), theta, dx)[0]
    colsum = siddon_backproject(np.ones((1, dt, dx), f32), theta, gx, gy)[0]
    with np.errstate(divide="ignore"):
        sigma = np.where(rowsum > 0, f32(1.0) / np.maximum(rowsum, f32(1e-30)), f32(0.0)).astype(f32)
    tau = (f32(1.0) / (colsum + f32(4.0))).astype(f32)
    lam = f32(lam)
    x = np.full((oy, gx, gy), init, f32) if np.isscalar(init) else _c32(init).copy()
    xbar = x.copy()
    p = np.zeros_like(data)
    qx, qy = np.zeros_like(x), np.zeros_like(x)
    for _ in range(int(num_iter)):
        sim = _project_grid(xbar, theta, dx)
        p = ((p + sigma * (sim - data)) / (f32(1.0) + sigma)).astype(f32)
        gxu, gyu = np.zeros_like(xbar), np.zeros_like(xbar)
        gxu[:, :-1] = xbar[:, 1:] - xbar[:, :-1]
        gyu[:, :, :-1] = xbar[:, :, 1:] - xbar[:, :, :-1]
        ax, ay = qx + f32(0.5) * gxu, qy + f32(0.5) * gyu
        nrm = np.maximum(np.sqrt(ax * ax + ay * ay), lam) / lam
        qx, qy = (ax / nrm).astype(f32), (ay / nrm).astype(f32)
        dv = np.zeros_like(x)
        dv[:, :-1] += qx[:, :-1]
        dv[:, 1:] -= qx[:, :-1]
        dv[:, :, :-1] += qy[:, :, :-1]
        dv[:, :, 1:] -= qy[:, :, :-1]
        x_new = (x - tau * (siddon_backproject(p, theta, gx, gy) - dv)).astype(f32)
        xbar = (f32(2.0) * x_new - x).astype(f32)
        x = x_new
    return x


def _project_grid(obj, theta, dx):
    """tomopy.project of a [oy][gx][gy] grid onto a dx-wide detector with center = dx / 2 (the projector inside sirt / tv)."""
    obj, theta = _c32(obj), _c32(theta)
    oy, ox, oz = obj.shape
    out = np.empty((oy, theta.size, dx), np.float32)
    lib().oracle_siddon_project(obj, oy, ox, oz, theta, theta.size, dx, dx / 2.0, out)
    return out


def loglik_task_bins(PW, partition=0):
    """The 64-lane tasks a detector row is cut into (list of 64 bin numbers each; a bin outside [0, PW) = an idle lane).
    partition 0: the planned projector kernels' (angle, bin block) tasks -- block k = the two 32-bin bands mirrored about the
    detector centre c = PW // 2: lanes 0..31 = bins c - 32 (k + 1) + l, lanes 32..63 = bins c + 32 k + (l - 32).
    partition 1: the tiled reduce pass's contiguous 64-bin blocks."""
    if partition == 0:
        c = PW // 2
        nblk = (PW - c + 31) // 32
        return [np.concatenate([c - 32 * (k + 1) + np.arange(32), c + 32 * k + np.arange(32)]) for k in range(nblk)]
    return [64 * k + np.arange(64) for k in range((PW + 63) // 64)]


def _butterfly64(v):
    lanes = np.arange(64)
    v = v.astype(np.float32)
    for off in (32, 16, 8, 4, 2, 1):
        v = (v + v[lanes ^ off]).astype(np.float32)
    return v[0]


def loglik_object_sums(lp, partition=0):
    """Per-object log-likelihood sums, reduce_sum over angles and bins of lp [S][A][PW] (ctvae/helper_functions.py:305-306) --
    TensorFlow does not fix the order of that sum; the build does, so that every path gives the same bits:
      * a task's 64 values (idle lanes: +0.0) are added by the xor butterfly -- for off in 32, 16, 8, 4, 2, 1:
        v[l] = v[l] + v[l ^ off] in fp32;
      * an angle's task sums are added one by one in ascending order, starting from +0.0  ->  S_a;
      * the object's sum is ((0 + B_0) + B_1) + ..., B_g = the same butterfly over S_(64 g) .. S_(64 g + 63) (angles past A: +0.0)."""
    lp = _c32(lp)
    S, A, PW = lp.shape
    tasks = loglik_task_bins(PW, partition)
    out = np.zeros(S, np.float32)
    for s in range(S):
        sa = np.zeros(((A + 63) // 64) * 64, np.float32)
        for a in range(A):
            acc = np.float32(0.0)
            for bins in tasks:
                ok = (bins >= 0) & (bins < PW)
                acc = np.float32(acc + _butterfly64(np.where(ok, lp[s, a, np.clip(bins, 0, PW - 1)], np.float32(0.0))))
            sa[a] = acc
        tot = np.float32(0.0)
        for g0 in range(0, A, 64):
            tot = np.float32(tot + _butterfly64(sa[g0:g0 + 64]))
        out[s] = tot
    return out


# ---- f2 -----------------------------------------------------------------------------------------------
def philox4x32_10(counter4, key2):
    c, k = np.asarray(counter4, np.uint32), np.asarray(key2, np.uint32)
    out = np.zeros(4, np.uint32)
    lib().oracle_philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def poisson_count(lam, element, seed):
    return lib().oracle_poisson_count(float(lam), int(element), int(seed))


def poisson_measure(sino, mask, pnm, seed):
    """Poisson(max(sino, 0) * mask * pnm) / pnm with the build's specified sampler (ctvae/create_masks.py:80-103)."""
    sino, mask = _c32(sino), _c32(mask)
    S, A, P = sino.shape
    out = np.empty_like(sino)
    lib().oracle_poisson_measure(sino, mask, S, A, P, pnm, int(seed), out)
    return out
