"""Developer sweep: planned forward at the headline size over the launch knobs (CTPVAE_TUNE_NS / _G / _WAVES), timed from
HIP-graph replays of 200 launches; prints the library's own choice (no knobs) first."""
import itertools, os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
B, A = (int(sys.argv[1]) if len(sys.argv) > 1 else 50), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
FMT = sys.argv[3] if len(sys.argv) > 3 else "auto"     # u16 | compact | auto
plan = RotatePlan(theta, 128, 128, True, dev, plan_format=FMT)
print("B", B, "A", A, "plan format", FMT, "->", "compact" if plan._compact else "u16")
x = torch.rand((B, 128, 128), device=dev); out = torch.empty((B, A, plan.PW), device=dev)
REP = 200 if B * A <= 4000 else 30
def t_us():
    plan.forward(x, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP): plan.forward(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / REP)
    return float(np.median(r))
print("library choice: %.2f us" % t_us())
res = []
grid = itertools.product((2,), (1, 2, 3, 4, 5, 6, 7, 8, 10, 12), (16,)) if os.environ.get('SWEEP_FAST') else itertools.product((1, 2), (1, 2, 3, 4, 5, 6, 8, 10), (8, 12, 16))
for ns, G, w in grid:
    _lib.tune("NS", ns); _lib.tune("G", G); _lib.tune("WAVES", w)
    try:
        res.append((t_us(), ns, G, w))
    except Exception as e:
        print("skip", ns, G, w, type(e).__name__)
for t, ns, G, w in sorted(res)[:int(os.environ.get('SWEEP_TOP', 12))]:
    print("NS=%d G=%2d waves=%2d: %.2f us" % (ns, G, w, t))
