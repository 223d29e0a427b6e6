"""Developer study (round 4): LDS passes of the step-coded forward's gathers when a lane may be DELAYED by whole rows and
the mirror class of an angle is free -- both are decisions of the plan builder alone: a delayed ray parks on the border
cell it enters the image from (its tap of row first - 1, a zero) and its first codes are "stay"; the kernel's walk does
not change.

    python tools/sim_lds_skew.py slice [angles]       128 x 128 slice pairs: ds_read_b64, 32 lanes x 32 eight-byte slots
    python tools/sim_lds_skew.py tile  [angles]       64 x 96 tiles of 512 x 512, four slices: ds_read_b128, 16 lanes x 16 slots

Counted with the gfx950 rule: a lane group's pass count = the largest number of DISTINCT cells on one slot (cell mod
lanes; identical cells broadcast).  Layouts: "cur" = today's classes (lanes step by |c| + |s| slots, rows by |c| - |s|);
"flip" = the other mirror class (lanes step by |c| - |s|, rows by |c| + |s|), where a delay of alpha rows per lane moves
a lane's slot by alpha (|c| + |s|) -- alpha is chosen so that consecutive lanes sit one slot apart."""
import sys

import numpy as np

mode = sys.argv[1] if len(sys.argv) > 1 else "slice"
A = int(sys.argv[2]) if len(sys.argv) > 2 else (180 if mode == "slice" else 90)
N = 128 if mode == "slice" else 512
P = int(np.ceil((np.sqrt(np.float64(2 * N * N)) + 2) / 2) * 2)
pad = (P - N) // 2
theta = np.pi * np.arange(A) / A
LANES = 32 if mode == "slice" else 16
ROWS_PER_GROUP = 6
LAYOUTS = {"flip": (True,), "cur": (False,), "both": (False, True)}[sys.argv[3] if len(sys.argv) > 3 else "flip"]


def transforms(theta, H, W):
    ang = (-theta).astype(np.float32)
    c, s = np.cos(ang.astype(np.float64)).astype(np.float32), np.sin(ang.astype(np.float64)).astype(np.float32)
    w1, h1 = np.float32(W - 1), np.float32(H - 1)
    xo = (w1 - (c * w1 - s * h1)) / np.float32(2)
    yo = (h1 - (s * w1 + c * h1)) / np.float32(2)
    return np.stack([c, -s, xo, s, c, yo], axis=1).astype(np.float32)


T = transforms(theta, P, P)
rnd = lambda v: np.where(v >= 0, np.floor(v + np.float32(0.5)), np.ceil(v - np.float32(0.5))).astype(np.int64)


def ray_taps(a, js, y0, x0, h, w):
    """taps of rays js at canvas rows -1 .. P: (ix, iy) relative to the core rect, live mask"""
    t0, t1, t2, t3, t4, t5 = [np.float32(v) for v in T[a]]
    i = np.arange(-1, P + 1, dtype=np.float32)[None, :]
    j = js.astype(np.float32)[:, None]
    x = (t0 * j + t1 * i) + t2
    y = (t3 * j + t4 * i) + t5
    ix, iy = rnd(x) - pad - x0, rnd(y) - pad - y0
    ok = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
    ok[:, 0] = ok[:, -1] = False
    return ix, iy, ok


def band_cost(a, js, rect, pitch, mirror, delays):
    """LDS passes of one lane group walking rays js (consecutive bins), lane k delayed by delays[k] rows; returns
    (per-step pass counts as an array over steps 0 .. maxlen-1, lengths incl. delay)"""
    y0, x0, h, w = rect
    ix, iy, ok = ray_taps(a, js, y0, x0, h, w)
    n = len(js)
    has = ok.any(1)
    first = np.where(has, ok.argmax(1), 0)             # index into the -1 .. P axis
    cnt = ok.sum(1)
    total = np.where(has, delays + cnt, 0)
    L = int(total.max()) if has.any() else 0
    if L == 0:
        return np.zeros(0, np.int64), total
    steps = np.arange(L)[None, :]
    # row index (into the -1..P axis) each lane taps at each step: parked before entry on row first-1, after exit on row last+1
    rel = steps - delays[:, None]                      # ray-relative row
    rowi = first[:, None] + np.clip(rel, -1, cnt[:, None])
    rowi = np.clip(rowi, 0, P + 1)
    ar = np.arange(n)[:, None]
    cx, cy = ix[ar, rowi], iy[ar, rowi]
    cx = np.clip(cx, -1, w)                            # (border ring; taps of parked rows lie in it for a rotation)
    cy = np.clip(cy, -1, h)
    xx = np.where(mirror, w - 1 - cx, cx)
    cell = 1 + (cy + 1) * pitch + xx
    cell = np.where(has[:, None], cell, 0)             # rays that miss the core sit on the guard cell
    passes = np.zeros(L, np.int64)
    for s in range(L):
        u = np.unique(cell[:, s])
        passes[s] = np.bincount(u % LANES, minlength=LANES).max()
    return passes, total


def alpha_for(a, layout):
    c, s = abs(float(T[a][0])), abs(float(T[a][3]))
    if layout == "cur":
        return 0.0
    lane_step, row_step = c - s, c + s                 # slots per lane / per row in the flipped layout (up to signs)
    return lane_step, row_step


def best_delays(a, js, rect, pitch, plus, n, layouts=(True,)):
    """try the layouts (flipped mirror class or today's) x target steps +1 / -1: lane k starts at canvas row
    round(beta k) + const <= its first live row, beta = (target - slots per lane) / slots per row; returns the cheapest
    (passes, delays relative to the ray's own entry, flipped)"""
    t = T[a]
    y0, x0, h, w = rect
    ix, iy, ok = ray_taps(a, js, y0, x0, h, w)
    has = ok.any(1)
    first = np.where(has, ok.argmax(1), 0).astype(np.int64)
    lanes = np.arange(n, dtype=np.float64)
    best = None
    for flipped in layouts:
        mirror = not_plus_mirror(plus, flipped)
        ls = float(t[3]) - float(t[0]) if mirror else float(t[0]) + float(t[3])
        rs = float(t[4]) - float(t[1]) if mirror else float(t[1]) + float(t[4])
        cands = [None]
        if abs(rs) > 0.2:
            cands += [(tgt - ls) / rs for tgt in (1.0, -1.0)]
        for beta in cands:
            if beta is None:
                d = np.zeros(n, np.int64)
            else:
                st = np.round(beta * lanes).astype(np.int64)
                c0 = (first - st)[has].min() if has.any() else 0
                d = np.where(has, first - (st + c0), 0)
            p, tot = band_cost(a, js, rect, pitch, mirror, d)
            cost = groups_cost(p)
            if best is None or cost < best[0]:
                best = (cost, d, flipped)
    return best


def not_plus_mirror(plus, flipped):
    # today: class 1 (plus) is staged as it is, class 0 column-mirrored; flipped: the other way round
    return plus if flipped else (not plus)


def groups_cost(p):
    ng = -(-len(p) // ROWS_PER_GROUP)
    return int(p.sum()) + (ng * ROWS_PER_GROUP - len(p))   # the padding rows of the last group: parked lanes, ~1 pass


def run():
    if mode == "slice":
        rect = (0, 0, N, N)
        pitch = 129
        c0 = P >> 1
        nJB = (P - c0 + 31) // 32
        bands = []
        for jb in range(nJB):
            bands.append(np.arange(c0 - 32 * (jb + 1), c0 - 32 * jb))
            bands.append(np.arange(c0 + 32 * jb, c0 + 32 * (jb + 1)))
        rects = [rect]
    else:
        pitch = 65
        rects = [(ty * 96, tx * 64, min(96, N - ty * 96), 64) for ty in (0, 2, 5) for tx in (0, 3, 7)]
    tot = {"cur": [0, 0, 0], "flip": [0, 0, 0]}
    per_angle = []
    for a in range(A):
        t = T[a]
        plus = (t[0] >= 0) == (t[3] >= 0)
        row = [np.degrees(theta[a])]
        acc = {"cur": [0, 0, 0], "flip": [0, 0, 0]}
        for rect in rects:
            if mode == "tile":
                y0, x0, h, w = rect
                cx, cy = pad + x0 + 0.5 * (w - 1), pad + y0 + 0.5 * (h - 1)
                radius = 0.5 * np.sqrt(64 * 64 + 96 * 96) + 3.0
                jc = float(t[0]) * (cx - float(t[2])) + float(t[3]) * (cy - float(t[5]))
                j0 = int(np.floor(jc - radius))
                bands = [np.arange(j0 + 16 * q, j0 + 16 * (q + 1)) for q in range(8)]
            for js in bands:
                js = js[(js >= 0) & (js < P)] if mode == "tile" else js
                if len(js) == 0:
                    continue
                jsv = np.clip(js, 0, P - 1)
                n = len(js)
                p, total = band_cost(a, jsv, rect, pitch, not_plus_mirror(plus, False), np.zeros(n, np.int64))
                if len(p) == 0:
                    continue
                acc["cur"][0] += groups_cost(p)
                acc["cur"][1] += -(-len(p) // ROWS_PER_GROUP)
                acc["cur"][2] += int(total.sum())
                cost, d, fl = best_delays(a, jsv, rect, pitch, plus, n, LAYOUTS)
                p2, total2 = band_cost(a, jsv, rect, pitch, not_plus_mirror(plus, fl), d)
                acc["flip"][0] += groups_cost(p2)
                acc["flip"][1] += -(-len(p2) // ROWS_PER_GROUP)
                acc["flip"][2] += int((total2 - d * (total2 > 0)).sum())
        for k in tot:
            for q in range(3):
                tot[k][q] += acc[k][q]
        per_angle.append((row[0], acc["cur"][0], acc["cur"][1], acc["flip"][0], acc["flip"][1]))
    print(f"mode {mode}: {A} angles, lanes per group {LANES}")
    for k in ("cur", "flip"):
        c, g, live = tot[k]
        print(f"{k:5s} passes {c:9d}  band-groups {g:7d}  passes per gathered row {c / (g * ROWS_PER_GROUP):.3f}  live taps {live}")
    print("angle: cur passes, groups | flip+skew passes, groups")
    for r in per_angle[:: max(1, A // 45)]:
        print("%6.1f  %7d %5d | %7d %5d   %.2f" % (r[0], r[1], r[2], r[3], r[4], r[3] / max(r[1], 1)))


run()
