"""Developer study (round 4): LDS passes of the 512 x 512 tile forward's gathers for three task shapes, counted from the
geometry with the gfx950 rule for ds_read_b128 (a hardware group = 16 lanes over 16 sixteen-byte slots; passes = the largest
number of distinct cells on one slot; parked lanes sit on their border cell and take part).

    python tools/sim_tile_pairs.py [angles] [tile stride]

"sorted" = round 3 (a 16-slot band per hardware group, bands sorted by length, four per task);
"mirror" = a lane walks slot s of band q, then slot s of band nq - 1 - q;
"fold"   = per half of an angle's slot range the short rays are paired with the long ones by two pointers (flat-top rays
           ride alone), lanes of a unit hold consecutive A rays (ascending) and consecutive B rays (descending)."""
import sys

import numpy as np

A = int(sys.argv[1]) if len(sys.argv) > 1 else 90
STRIDE = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = 512
P = int(np.ceil((np.sqrt(np.float64(2 * N * N)) + 2) / 2) * 2)
pad = (P - N) // 2
theta = np.pi * np.arange(A) / A
TW, TH, RPG, PITCH = 64, 96, 6, 65
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
irow = np.arange(-1, P + 1, dtype=np.float32)[None, :]


def rays(a, y0, x0, h, w):
    """per slot: (entry border cell, cells of the live rows, exit border cell) in the class's staged image"""
    t0, t1, t2, t3, t4, t5 = [np.float32(v) for v in T[a]]
    plus = (t0 >= 0) == (t3 >= 0)
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
    ok[:, 0] = ok[:, -1] = False
    cxx, cyy = np.clip(ix, -1, w), np.clip(iy, -1, h)
    xx = cxx if plus else w - 1 - cxx
    cell = 1 + (cyy + 1) * PITCH + xx
    out = []
    for s in range(NB):
        if not ok[s].any():
            out.append((0, np.zeros(0, np.int64), 0))
            continue
        f = int(ok[s].argmax())
        n = int(ok[s].sum())
        out.append((int(cell[s, f - 1]), cell[s, f:f + n], int(cell[s, f + n])))
    return out


def lane_stream(ra, rb):
    """cells a lane taps: ray A padded to whole groups (parked on its exit cell), then ray B; returns (cells, park cell)"""
    parts = []
    park = 0
    for r in (ra, rb):
        if r is None:
            continue
        e, c, x = r
        g = -(-len(c) // RPG)
        parts.append(c)
        parts.append(np.full(g * RPG - len(c), x, np.int64))
        park = x
    return (np.concatenate(parts) if parts else np.zeros(0, np.int64)), park


def unit_passes(lanes, steps):
    """lanes: list of <= 16 (cells, park); passes of this hardware group over `steps` rows"""
    M = np.zeros((16, steps), np.int64)
    for k, (c, park) in enumerate(lanes):
        M[k, :] = park
        M[k, :len(c)] = c[:steps]
    tot = 0
    for s in range(steps):
        u = np.unique(M[:, s])
        tot += np.bincount(u % 16, minlength=16).max()
    return tot


def tasks_cost(units):
    """units: list of (lanes, groups); sorted by groups, four per task; returns (passes, gathered rows x 16-lane groups)"""
    units = sorted(units, key=lambda u: -u[1])
    passes = rows = 0
    for i in range(0, len(units), 4):
        ng = units[i][1]
        for lanes, _ in units[i:i + 4]:
            passes += unit_passes(lanes, ng * RPG)
        rows += 4 * ng * RPG
    return passes, rows


def fold_half(rl, order):
    """two-pointer pairing over the slots `order` (ascending length expected): returns lanes [(A, B or None)]"""
    g = [-(-len(rl[s][1]) // RPG) for s in order]
    gmax = max(g) if g else 0
    lanes = []
    i, j = 0, len(order) - 1
    singles, pairs = [], []
    while i <= j:
        if i < j and g[i] + g[j] <= gmax:
            pairs.append((order[i], order[j]))
            i += 1
            j -= 1
        else:
            singles.append(order[j])
            j -= 1
    return [(s, None) for s in singles] + pairs


tiles = [(ty * TH, tx * TW, min(TH, N - ty * TH), min(TW, N - tx * TW)) for ty in range(-(-N // TH)) for tx in range(-(-N // TW))]
res = {k: [0, 0] for k in ("sorted", "mirror", "fold")}
live = 0
for (y0, x0, h, w) in tiles[::STRIDE]:
    per = {k: {} for k in res}
    for a in range(A):
        t = T[a]
        cls = (int((t[0] >= 0) == (t[3] >= 0)), int(t[4] < 0))
        rl = rays(a, y0, x0, h, w)
        live += sum(len(r[1]) for r in rl)
        for k in per:
            per[k].setdefault(cls, [])
        for q in range(NQ16):
            lanes = [lane_stream(rl[16 * q + k], None) for k in range(16)]
            per["sorted"][cls].append((lanes, max(len(c) for c, _ in lanes) // RPG))
        for q in range(NQ16 // 2):
            lanes = [lane_stream(rl[16 * q + k], rl[16 * (NQ16 - 1 - q) + k]) for k in range(16)]
            per["mirror"][cls].append((lanes, max(len(c) for c, _ in lanes) // RPG))
        half = NB // 2
        for order in (list(range(0, half)), list(range(NB - 1, half - 1, -1))):
            lp = fold_half(rl, order)
            for u0 in range(0, len(lp), 16):
                lanes = [lane_stream(rl[sa], None if sb is None else rl[sb]) for sa, sb in lp[u0:u0 + 16]]
                per["fold"][cls].append((lanes, max(len(c) for c, _ in lanes) // RPG))
    for k in res:
        for cls, units in per[k].items():
            p, r = tasks_cost(units)
            res[k][0] += p
            res[k][1] += r
    print("tile", (y0, x0, h, w), {k: (v[0], v[1]) for k, v in res.items()}, flush=True)
print(f"live rows {live}")
for k, (p, r) in res.items():
    print(f"{k:7s} passes {p:9d}  group-rows {r:9d} = {r * 16 / live:.3f} x live   passes per group-row {p / r:.3f}")
