"""Developer sweep: the TomoPy-style projector over CTPVAE_TUNE_SIDDON_THREADS / _PPB at one size (graph replays)."""
import itertools, os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.helper_functions import create_sinograms
dev = torch.device('cuda', 0)
B, A = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 180)
x = torch.rand((B, 128, 128), device=dev); theta = phantoms.dense_theta(A)
def t_us():
    create_sinograms(x, theta); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): create_sinograms(x, theta)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / 50)
    return float(np.median(r))
print("library choice: %.1f us" % t_us())
res = []
for th, ppb in itertools.product((256, 512, 1024), (1, 2, 3, 4, 6, 8, 12, 16, 32)):
    if ppb > A: continue
    _lib.tune("SIDDON_THREADS", th); _lib.tune("SIDDON_PPB", ppb)
    res.append((t_us(), th, ppb))
for t, th, ppb in sorted(res)[:6]:
    print("threads=%4d angles/workgroup=%2d: %.1f us" % (th, ppb, t))
