"""Developer timing: the 512 x 512 tile forward against the tile height (knob TILED_TH; the plan is built and used under it).
Round 5: the knob is compiled into TIMING builds only -- bash tools/build_variant.sh th -DCTPVAE_TUNE_TILED_TH, then
CTPVAE_VARIANT_LIB=tools/libctpvae_radon_th.bin python tools/time_tile_heights.py (the product library ignores it).
512 rows in 96-row tiles (rounds 1-3) are five full rows of tiles and a 32-row remainder: 768 workgroups of unequal cost on
256 CUs; 86 rows: 768 equal ones; 128 rows (the default since round 4): 512 equal ones.
   python tools/time_tile_heights.py [slices] [N] [heights, comma separated]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
theta = np.pi * np.arange(90) / 90
x = torch.rand((B, N, N), device=d)
def timed(plan, out, n=20):
    plan.forward(x, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): plan.forward(x, out=out)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(r))
heights = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [128, 103, 96, 88, 86, 80, 64]
for rnd in range(2):
    for th in heights:
        with _lib.tuned("TILED_TH", th):
            plan = RotatePlan(theta, N, N, True, d)
            out = torch.empty((B, 90, plan.PW), device=d)
            t = timed(plan, out)
            ref = plan.forward(x).double()
        print(f"round {rnd} B={B} {N}x{N} tile height {th:3d}: {t:7.2f} us  (sum {float(ref.sum()):.6e})", flush=True)
