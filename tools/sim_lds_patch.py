"""Developer study: LDS passes of the nearest forward's gathers when the 32 lanes of a ds_read_b64 group (16 of a
ds_read_b128 group) hold a 2-D PATCH of samples -- RJ neighbouring rays x RI consecutive canvas rows -- instead of 32
neighbouring rays at one row, counted with the gfx950 rule (slot = cell index mod LANES, identical addresses broadcast,
passes = largest number of distinct addresses on one slot).

    python tools/sim_lds_patch.py [angles] [lanes]      (128 x 128 slice, P = 184; lanes = 32 (b64) or 16 (b128))

For every angle the best of the candidate (RJ x RI, pitch mod LANES, mirror) layouts is also reported ("best of K classes"):
a workgroup stages its slice once per layout class, so only a few classes are affordable."""
import sys

import numpy as np

N = 128
P = int(np.ceil((np.sqrt(np.float64(2 * N * N)) + 2) / 2) * 2)
pad = (P - N) // 2
A = int(sys.argv[1]) if len(sys.argv) > 1 else 180
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 32
theta = np.pi * np.arange(A) / A


def transforms(theta, H, W):
    ang = (-theta).astype(np.float32)
    c, s = np.cos(ang.astype(np.float64)).astype(np.float32), np.sin(ang.astype(np.float64)).astype(np.float32)
    w1, h1 = np.float32(W - 1), np.float32(H - 1)
    xo = (w1 - (c * w1 - s * h1)) / np.float32(2)
    yo = (h1 - (s * w1 + c * h1)) / np.float32(2)
    return np.stack([c, -s, xo, s, c, yo], axis=1).astype(np.float32)


T = transforms(theta, P, P)
jj, ii = np.meshgrid(np.arange(P, dtype=np.float32), np.arange(P, dtype=np.float32), indexing="ij")   # [bin j][row i]


def taps(a):
    t0, t1, t2, t3, t4, t5 = [np.float32(v) for v in T[a]]
    x = (t0 * jj + t1 * ii) + t2
    y = (t3 * jj + t4 * ii) + t5
    rnd = lambda v: np.where(v >= 0, np.floor(v + np.float32(0.5)), np.ceil(v - np.float32(0.5))).astype(np.int64)
    ix, iy = rnd(x) - pad, rnd(y) - pad
    ok = (ix >= 0) & (ix < N) & (iy >= 0) & (iy < N)
    return ix, iy, ok


def passes(ix, iy, ok, rj, ri, pitch, mirror):
    """sum over live patches of the passes one lane group needs; also the number of live patches"""
    xx = np.where(mirror, N - 1 - ix, ix)
    addr = np.where(ok, iy * pitch + xx, -1)
    pj, pi_ = -(-P // rj) * rj, -(-P // ri) * ri
    full = np.full((pj, pi_), -1, np.int64)
    full[:P, :P] = addr
    pt = full.reshape(pj // rj, rj, pi_ // ri, ri).transpose(0, 2, 1, 3).reshape(-1, rj * ri)
    pt = pt[(pt >= 0).any(1)]
    pt = np.sort(pt, axis=1)
    dup = np.zeros_like(pt, bool)
    dup[:, 1:] = pt[:, 1:] == pt[:, :-1]
    slot = np.where((pt < 0) | dup, -1, pt % LANES)
    cnt = (slot[:, :, None] == np.arange(LANES)[None, None, :]).sum(1)
    return int(cnt.max(1).sum()), pt.shape[0]


shapes = [(LANES, 1), (LANES // 2, 2), (LANES // 4, 4), (LANES // 8, 8), (LANES // 16, 16)]
if LANES == 32:
    shapes.append((1, 32))
pitches = sorted(set([1, LANES - 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, LANES // 2 + 1, LANES // 4, LANES // 4 + 1]))
cands = [(rj, ri, p) for (rj, ri) in shapes for p in pitches]
res = np.zeros((A, len(cands), 2))
live = np.zeros((A, len(shapes)))
for a in range(A):
    ix, iy, ok = taps(a)
    for k, (rj, ri, p) in enumerate(cands):
        for m in (0, 1):
            c, n = passes(ix, iy, ok, rj, ri, N + 32 + p if False else (N // LANES + 1) * LANES + p, bool(m))
            res[a, k, m] = c
        live[a, shapes.index((rj, ri))] = n
# the current scheme: 32 x 1, pitch == 1, mirror class by sign
plus = (T[:, 0] >= 0) == (T[:, 3] >= 0)
k_cur = cands.index((LANES, 1, 1))
cur = np.where(plus, res[:, k_cur, 0], res[:, k_cur, 1])
n_cur = live[:, 0]
print("lanes per group %d, %d angles" % (LANES, A))
print("current (%dx1, pitch 1, mirrored class): passes per live group %.3f" % (LANES, cur.sum() / n_cur.sum()))
flat = res.reshape(A, -1)                       # candidate c = (k, m)
nlive = np.repeat(live[:, [shapes.index((rj, ri)) for (rj, ri, p) in cands]], 2, axis=1)
# passes are compared per SAMPLE GROUP of LANES lanes, so normalise each candidate by its own number of live groups
ratio = flat / nlive
print("best single layout per angle: mean passes %.3f" % ratio.min(1).mean())
# greedy choice of K layout classes
chosen = []
best = np.full(A, np.inf)
for K in range(1, 7):
    gains = [np.minimum(best, ratio[:, c]).mean() for c in range(ratio.shape[1])]
    c = int(np.argmin(gains))
    chosen.append(c)
    best = np.minimum(best, ratio[:, c])
    k, m = divmod(c, 2)
    print("K=%d  + %2dx%-2d pitch %2d mirror %d  -> mean passes per group %.3f" % (K, cands[k][0], cands[k][1], cands[k][2], m, best.mean()))
print("per shape (best pitch / mirror per angle):")
for (rj, ri) in shapes:
    idx = [2 * k + m for k, (a_, b_, p) in enumerate(cands) if (a_, b_) == (rj, ri) for m in (0, 1)]
    print("  %2dx%-2d  %.3f" % (rj, ri, ratio[:, idx].min(1).mean()))
