"""Developer timing: compact (step-coded) forward plan against the u16 plan at one shape, HIP-graph replays of 200 launches.
usage: python tools/time_compact.py B A [sweep]"""
import itertools, os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
B, A = (int(sys.argv[1]) if len(sys.argv) > 1 else 50), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
x = torch.rand((B, 128, 128), device=dev)
def t_us(plan, out):
    plan.forward(x, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(200): plan.forward(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / 200)
    return float(np.median(r))
pc = RotatePlan(theta, 128, 128, True, dev, plan_format="compact")
p16 = RotatePlan(theta, 128, 128, True, dev, plan_format="u16")
assert pc.compact and not p16.compact
oc, o16 = torch.empty((B, A, pc.PW), device=dev), torch.empty((B, A, pc.PW), device=dev)
print("B=%d A=%d  compact %.2f us   u16 %.2f us   equal=%s   plan bytes %d vs %d" % (
    B, A, t_us(pc, oc), t_us(p16, o16), torch.equal(oc, o16), pc._fwd_plan.numel(), p16._fwd_plan.numel()), flush=True)
if len(sys.argv) > 3:
    res = []
    for ns, G, w in itertools.product((1, 2), (1, 2, 3, 4, 5, 6, 8, 10, 12), (8, 12, 16)):
        _lib.tune("NS", ns); _lib.tune("G", G); _lib.tune("WAVES", w)
        res.append((t_us(pc, oc), ns, G, w))
    for t, ns, G, w in sorted(res)[:10]:
        print("  compact NS=%d G=%2d waves=%2d: %.2f us" % (ns, G, w, t), flush=True)
