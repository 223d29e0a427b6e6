"""Literal numpy restatement of the reference's TensorFlow op CHAIN (pad -> repeat -> transform -> reduce_sum and
its registered gradients), used to check that the fused formulas in oracle/radon_oracle.c mean the same thing.
numpy float32 arithmetic is IEEE, unfused, so the results must agree bit for bit."""
import numpy as np

f32 = np.float32


def round_half_away(v):
    # std::round
    t = np.trunc(v)
    d = v - t
    return np.where(np.abs(d) >= f32(0.5), t + np.copysign(f32(1), v), t).astype(np.float32)


def read_fill(img, iy, ix):
    """img [H][W]; integer index arrays; zero outside (fill_mode CONSTANT, fill_value 0)."""
    H, W = img.shape
    ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
    out = np.zeros(iy.shape, np.float32)
    out[ok] = img[iy[ok], ix[ok]]
    return out


def transform_image(img, t8, interp):
    """ImageProjectiveTransformV3 of one H x W image with one flat transform (same output size)."""
    H, W = img.shape
    t = np.asarray(t8, np.float32)
    oy, ox = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    x = (t[0] * ox + t[1] * oy) + t[2]
    y = (t[3] * ox + t[4] * oy) + t[5]
    if interp == 0:
        return read_fill(img, round_half_away(y).astype(np.int64), round_half_away(x).astype(np.int64))
    yf, xf = np.floor(y), np.floor(x)
    yc, xc = yf + f32(1), xf + f32(1)
    iyf, ixf, iyc, ixc = (a.astype(np.int64) for a in (yf, xf, yc, xc))
    v_yf = (xc - x) * read_fill(img, iyf, ixf) + (x - xf) * read_fill(img, iyf, ixc)
    v_yc = (xc - x) * read_fill(img, iyc, ixf) + (x - xf) * read_fill(img, iyc, ixc)
    return ((yc - y) * v_yf + (y - yf) * v_yc).astype(np.float32)


def seq_sum(arr, axis):
    """Sequential fp32 sum along `axis` (index order)."""
    arr = np.moveaxis(arr.astype(np.float32), axis, 0)
    acc = np.zeros(arr.shape[1:], np.float32)
    for k in range(arr.shape[0]):
        acc = acc + arr[k]
    return acc


def project_chain(canvas, T8, interp):
    """canvas [S][PH][PW] (already padded) -> [S][A][PW]: rotate every copy, reduce_sum over rows."""
    S, A = canvas.shape[0], T8.shape[0]
    out = np.empty((S, A, canvas.shape[2]), np.float32)
    for s in range(S):
        for a in range(A):
            out[s, a] = seq_sum(transform_image(canvas[s], T8[a], interp), 0)
    return out


def project_chain_grad(gsino, Tinv8, interp, PH):
    """Registered gradients of the chain, up to (not including) the crop: [S][A][PW] -> [S][PH][PW]."""
    S, A, PW = gsino.shape
    out = np.empty((S, PH, PW), np.float32)
    for s in range(S):
        per_angle = [transform_image(np.broadcast_to(gsino[s, a], (PH, PW)).copy(), Tinv8[a], interp)
                     for a in range(A)]
        out[s] = seq_sum(np.stack(per_angle), 0)
    return out
