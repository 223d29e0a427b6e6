"""Developer sweep: the tiled forward (slices larger than LDS) over CTPVAE_TUNE_NS / _G at one size (graph replays)."""
import itertools, os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
B, A, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), (int(sys.argv[2]) if len(sys.argv) > 2 else 90), (int(sys.argv[3]) if len(sys.argv) > 3 else 512)
plan = RotatePlan(np.pi * np.arange(A) / A, N, N, True, dev)
assert plan.tiled
x = torch.rand((B, N, N), device=dev); out = torch.empty((B, A, plan.PW), device=dev)
def t_us():
    plan.forward(x, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): plan.forward(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / 20)
    return float(np.median(r))
print("library choice: %.1f us (tile kernel + reduce)" % t_us())
res = []
for ns, G in itertools.product((1, 2, 4), (1, 2, 3, 4, 6, 8)):
    if ns > B: continue
    _lib.tune("TILED_NS", ns); _lib.tune("TILED_G", G)
    res.append((t_us(), ns, G))
for ns in (1, 2, 4):
    best = sorted(r for r in res if r[1] == ns)[:2]
    print("NS=%d best:" % ns, ", ".join("G=%d %.1f us" % (G, t) for t, _, G in best))
