"""Developer study (round 4): rows the 512 x 512 tile forward GATHERS against the rows its rays hold, for the task shapes of
rotate_fwd_tile_compact_kernel, counted from the geometry (64 x 96 tiles, 128 ray slots per (tile, angle), six rows per walk
step, a wave = four 16-slot bands).

    python tools/sim_tile_tasks.py [angles] [N]

"sorted"  = round 3: bands of a (tile, mirror class, step sign) sorted by length, four consecutive bands per task;
"paired"  = round 4: a lane walks TWO rays back to back -- slot s of band q, then slot s of band nq - 1 - q of the same angle
            (the mirror image of slot 15 - s: lengths of a trapezoid's rising and falling side add up to a constant) --, the
            band pairs sorted by their longest lane and dealt four to a task."""
import sys

import numpy as np

A = int(sys.argv[1]) if len(sys.argv) > 1 else 90
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
P = int(np.ceil((np.sqrt(np.float64(2 * N * N)) + 2) / 2) * 2)
pad = (P - N) // 2
theta = np.pi * np.arange(A) / A
TW, TH, RPG = 64, 96, 6
radius = np.float32(0.5) * np.float32(np.sqrt(np.float32(TW * TW + TH * TH))) + np.float32(3.0)
NB = (int(np.ceil(2.0 * radius)) + 2 + 63) // 64 * 64
NQ16 = NB // 16


def transforms(theta, H, W):
    ang = (-theta).astype(np.float32)
    c, s = np.cos(ang.astype(np.float64)).astype(np.float32), np.sin(ang.astype(np.float64)).astype(np.float32)
    w1, h1 = np.float32(W - 1), np.float32(H - 1)
    xo = (w1 - (c * w1 - s * h1)) / np.float32(2)
    yo = (h1 - (s * w1 + c * h1)) / np.float32(2)
    return np.stack([c, -s, xo, s, c, yo], axis=1).astype(np.float32)


T = transforms(theta, P, P)
rnd = lambda v: np.where(v >= 0, np.floor(v + np.float32(0.5)), np.ceil(v - np.float32(0.5))).astype(np.int64)
irow = np.arange(P, dtype=np.float32)[None, :]


def lengths(a, y0, x0, h, w):
    """live rows of the NB ray slots of tile (y0, x0, h, w) at angle a"""
    t0, t1, t2, t3, t4, t5 = [np.float32(v) for v in T[a]]
    cx = np.float32(pad + x0) + np.float32(0.5) * np.float32(w - 1)
    cy = np.float32(pad + y0) + np.float32(0.5) * np.float32(h - 1)
    jc = t0 * (cx - t2) + t3 * (cy - t5)
    j0 = int(np.floor(jc - radius))
    js = j0 + np.arange(NB)
    j = js.astype(np.float32)[:, None]
    x = (t0 * j + t1 * irow) + t2
    y = (t3 * j + t4 * irow) + t5
    ix, iy = rnd(x) - pad - x0, rnd(y) - pad - y0
    ok = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h) & ((js >= 0) & (js < P))[:, None]
    return ok.sum(1)


def groups(n):
    return -(-n // RPG)


tiles = [(ty * TH, tx * TW, min(TH, N - ty * TH), min(TW, N - tx * TW)) for ty in range(-(-N // TH)) for tx in range(-(-N // TW))]
live = 0
g_block = g_sorted = g_paired = g_paired_ideal = 0
for (y0, x0, h, w) in tiles:
    per_class = {}
    for a in range(A):
        t = T[a]
        cls = (int((t[0] >= 0) == (t[3] >= 0)), int(t[4] < 0))
        ln = lengths(a, y0, x0, h, w)
        live += int(ln.sum())
        # round 3 before the sort: (angle, 64-slot block) tasks of mirrored 32-slot runs
        for blk in range(NB // 64):
            sl = np.r_[ln[blk * 32:blk * 32 + 32], ln[NB - 32 * (blk + 1):NB - 32 * blk]]
            g_block += groups(int(sl.max())) * 64
        per_class.setdefault(cls, []).append(ln)
    for cls, lst in per_class.items():
        # sorted bands
        bands = sorted((groups(int(ln[16 * q:16 * q + 16].max())) for ln in lst for q in range(NQ16)), reverse=True)
        bands += [0] * (-len(bands) % 4)
        g_sorted += sum(bands[i] for i in range(0, len(bands), 4)) * 64
        # paired bands: per lane groups(lenA) + groups(lenB)
        units = []
        for ln in lst:
            for q in range(NQ16 // 2):
                ga = -(-ln[16 * q:16 * q + 16] // RPG)
                gb = -(-ln[16 * (NQ16 - 1 - q):16 * (NQ16 - q)] // RPG)
                units.append(ga + gb)
        units.sort(key=lambda u: -int(u.max()))
        units += [np.zeros(16, np.int64)] * (-len(units) % 4)
        for i in range(0, len(units), 4):
            g_paired += int(units[i].max()) * 64
            g_paired_ideal += int(sum(u.sum() for u in units[i:i + 4]))
print(f"{N} x {N}, {A} angles, {len(tiles)} tiles, {NB} slots per (tile, angle): live rows {live}")
for name, g in (("(angle, block) tasks", g_block), ("sorted bands", g_sorted), ("paired bands, sorted", g_paired),
                ("  (their lanes' own groups)", g_paired_ideal)):
    print(f"{name:32s} gathered rows {g * RPG:10d} = {g * RPG / live:.3f} x live")

# where the paired scheme's remaining waste sits: inside a unit (its 16 lanes differ) or between the four units of a task
intra = inter = own = 0
best_any = 0
for (y0, x0, h, w) in tiles[:: max(1, len(tiles) // 12)]:
    per_class = {}
    for a in range(A):
        t = T[a]
        cls = (int((t[0] >= 0) == (t[3] >= 0)), int(t[4] < 0))
        per_class.setdefault(cls, []).append(lengths(a, y0, x0, h, w))
    for cls, lst in per_class.items():
        units = []
        for ln in lst:
            for q in range(NQ16 // 2):
                units.append(-(-ln[16 * q:16 * q + 16] // RPG) + -(-ln[16 * (NQ16 - 1 - q):16 * (NQ16 - q)] // RPG))
        units.sort(key=lambda u: -int(u.max()))
        for i in range(0, len(units), 4):
            grp = units[i:i + 4]
            m = int(grp[0].max())
            own += int(sum(u.sum() for u in grp))
            intra += int(sum(16 * int(u.max()) for u in grp))
            inter += 64 * m if len(grp) == 4 else 16 * len(grp) * m
print(f"sampled tiles: lanes' own groups {own}, + inside-unit spread {intra / own:.3f} x, + between units of a task {inter / own:.3f} x")
