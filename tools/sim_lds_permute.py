"""Developer study: would a per-(angle, bin block) LANE PERMUTATION remove the planned forward's LDS bank conflicts?
A ds_read_b64 is served in two groups of 32 lanes over 32 eight-byte slots; the 64 rays of a task are free to sit on any
lane.  For each (angle, block) the rays are dealt to the two groups so that, at a reference row, every group holds at most
one ray per slot residue (mod 32); the cost is then counted over ALL rows of the walk with the gfx950 rule.

    python tools/sim_lds_permute.py [angles]      (128 x 128 slice, P = 184, rays aligned at their own first row)"""
import sys
import numpy as np

N = 128
P = int(np.ceil((np.sqrt(np.float64(2 * N * N)) + 2) / 2) * 2)
pad = (P - N) // 2
A = int(sys.argv[1]) if len(sys.argv) > 1 else 20
theta = np.pi * np.arange(0, 180, 180 // A)[:A] / 180


def transforms(theta, H, W):
    ang = (-theta).astype(np.float32)
    c, s = np.cos(ang.astype(np.float64)).astype(np.float32), np.sin(ang.astype(np.float64)).astype(np.float32)
    w1, h1 = np.float32(W - 1), np.float32(H - 1)
    xo = (w1 - (c * w1 - s * h1)) / np.float32(2)
    yo = (h1 - (s * w1 + c * h1)) / np.float32(2)
    z = np.zeros_like(c)
    return np.stack([c, -s, xo, s, c, yo, z, z], axis=1).astype(np.float32)


T = transforms(theta, P, P)
nJB = (P - (P >> 1) + 31) // 32
rows = np.arange(P, dtype=np.float32)
pitch = 129


def lane_to_bin(jb, lane):
    c = P >> 1
    return np.where(lane < 32, c - 32 * (jb + 1) + lane, c + 32 * jb + (lane - 32))


def xy_for(a, j):
    t0, t1, t2, t3, t4, t5 = [np.float32(v) for v in T[a][:6]]
    x = (t0 * np.float32(j) + t1 * rows) + t2
    y = (t3 * np.float32(j) + t4 * rows) + t5
    rnd = lambda v: np.where(v >= 0, np.floor(v + np.float32(0.5)), np.ceil(v - np.float32(0.5))).astype(np.int64)
    ix, iy = rnd(x) - pad, rnd(y) - pad
    return ix, iy, (ix >= 0) & (ix < N) & (iy >= 0) & (iy < N)


def walk(a, jb):
    """slot values v[lane][step] of the block's rays, each aligned at its own first live row; dead cell = one zero cell"""
    js = lane_to_bin(jb, np.arange(64))
    t = T[a]
    mirror = not ((t[0] >= 0) == (t[3] >= 0))
    V, live = [], []
    for l in range(64):
        if not (0 <= js[l] < P):
            V.append(None); continue
        ix, iy, ok = xy_for(a, js[l])
        if not ok.any():
            V.append(None); continue
        f, la = ok.argmax(), P - 1 - ok[::-1].argmax()
        xx = np.where(mirror, N - 1 - ix, ix)
        V.append(np.where(ok, iy * pitch + xx, -1)[f:la + 1])
    n = max([len(v) for v in V if v is not None], default=0)
    if n == 0:
        return None
    n = (n + 5) // 6 * 6
    out = np.full((64, n), -1, np.int64)
    for l, v in enumerate(V):
        if v is not None:
            out[l, :len(v)] = v
    return out


def cycles(v64):
    """v64 [64][steps] in LANE order -> LDS cycles (2 groups of 32 lanes; -1 = the shared zero cell)"""
    c = 0
    for n in range(v64.shape[1]):
        for h in (slice(0, 32), slice(32, 64)):
            vv = np.unique(v64[h, n])
            vv = np.where(vv < 0, N * pitch, vv)
            c += np.bincount(vv % 32, minlength=32).max()
    return c


def deal(v64, ref):
    """rays -> lanes: per slot residue at step `ref`, the first ray goes to group 0, the second to group 1, the rest fill up"""
    res = np.where(v64[:, ref] >= 0, v64[:, ref] % 32, -1)
    g0, g1, rest = [], [], []
    seen0, seen1 = set(), set()
    for l in range(64):
        r = res[l]
        if r < 0:
            rest.append(l)
        elif r not in seen0 and len(g0) < 32:
            g0.append(l); seen0.add(r)
        elif r not in seen1 and len(g1) < 32:
            g1.append(l); seen1.add(r)
        else:
            rest.append(l)
    for l in rest:
        (g0 if len(g0) < 32 else g1).append(l)
    return np.array(g0 + g1)


tot = {"current": 0, "dealt at mid row": 0, "best of 5 reference rows": 0}
instr = 0
per_angle = []
for a in range(A):
    ca = [0, 0, 0]
    for jb in range(nJB):
        v = walk(a, jb)
        if v is None:
            continue
        instr += v.shape[1]
        c0 = cycles(v)
        refs = [v.shape[1] // 2] + [int(v.shape[1] * f) for f in (0.2, 0.35, 0.65, 0.8)]
        cs = [cycles(v[deal(v, r)]) for r in refs]
        ca[0] += c0; ca[1] += cs[0]; ca[2] += min(cs)
    per_angle.append(ca)
    tot["current"] += ca[0]; tot["dealt at mid row"] += ca[1]; tot["best of 5 reference rows"] += ca[2]
print("A", A, "gather instructions", instr)
for k, c in tot.items():
    print("%-26s cycles %7d  = %.2f per instruction (2.00 = conflict-free)" % (k, c, c / instr))
print("per angle (deg: current, dealt, best-of-5):")
for a, ca in enumerate(per_angle):
    print("  %5.1f: %6d %6d %6d" % (np.degrees(theta[a]), *ca))
