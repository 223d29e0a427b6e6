"""Developer sweep: the two bit-identical nearest backward paths at one size over their launch knobs (planned:
CTPVAE_TUNE_BNS / _BW; segment: CTPVAE_TUNE_SEG_NS / _SEG_PPT / _SEG_CHUNK), from HIP-graph replays of 200 launches;
the library's own choices are printed first."""
import itertools, os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
B, A = (int(sys.argv[1]) if len(sys.argv) > 1 else 50), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 128
theta = np.pi * np.arange(A) / A
auto = RotatePlan(theta, N, N, True, dev)
g = torch.rand((B, A, auto.PW), device=dev); out = torch.empty((B, N, N), device=dev)
def t_us(plan):
    plan.backward(g, out=out); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(200): plan.backward(g, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / 200)
    return float(np.median(r))
has_plan = auto.planned[1]
planned = RotatePlan(theta, N, N, True, dev); planned.backward_uses_plan = lambda S: True
seg = RotatePlan(theta, N, N, True, dev); seg.backward_uses_plan = lambda S: False
print("library: %.2f us (%s);  planned default %s us, segment default %.2f us" % (
    t_us(auto), "planned" if auto.backward_uses_plan(B) else "segment", ("%.2f" % t_us(planned)) if has_plan else "n/a", t_us(seg)))
res = []
for ns, w in (itertools.product((1, 2), (2, 4, 8, 16)) if has_plan else ()):
    _lib.tune("BNS", ns); _lib.tune("BW", w)
    res.append((t_us(planned), "planned NS=%d waves=%2d" % (ns, w)))
for k in ("BNS", "BW"): _lib.tune(k)
for ns, ppt, ch in itertools.product((1, 2), (4, 8), (24, 48, 96)):
    _lib.tune("SEG_NS", ns); _lib.tune("SEG_PPT", ppt); _lib.tune("SEG_CHUNK", ch)
    res.append((t_us(seg), "segment NS=%d ppt=%d chunk=%2d" % (ns, ppt, ch)))
for t, name in sorted(res)[:8]:
    print("%-32s %.2f us" % (name, t))
