"""Developer timing: the stepped backward at B x N x N x 90 angles with one slice pair per workgroup (STEP_NS=2, rounds 3) and
two pairs sharing the address arithmetic (STEP_NS=4, round 4), over the angle chunk; results compared bit for bit.
   python tools/time_step_ns.py [B] [N] [angles]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
if os.environ.get("CTPVAE_VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CTPVAE_VARIANT_LIB"])
    _lib.torch_node = lambda: None
print("library:", _lib.LIB_PATH)
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
A = int(sys.argv[3]) if len(sys.argv) > 3 else 90
def timed(body, n=20):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): body()
    for _ in range(2): g.replay()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(r))
theta = np.pi * np.arange(A) / A
plan = RotatePlan(theta, N, N, True, dev)
g_ = torch.rand((B, A, plan.PW), device=dev); gx = torch.empty((B, N, N), device=dev)
up = torch.rand((B,), device=dev)
with _lib.tuned("STEP_NS", 2):
    ref = plan.backward(g_, scale=up).clone()
for rnd in range(2):
    for ns, ch, kb in ((2, -1, 0), (4, -1, 0), (4, 24, 0), (4, 18, 0), (2, 24, 0), (2, -1, 40), (4, -1, 53)):
        if ns: _lib.tune("STEP_NS", ns)
        _lib.tune("SEG_CHUNK", ch)
        if kb: _lib.tune("STEP_LDS_KB", kb)
        t = timed(lambda: plan.backward(g_, out=gx, scale=up))
        same = torch.equal(plan.backward(g_, scale=up), ref)
        _lib.tune("*")
        print(f"round {rnd} B={B} {N}x{N} A={A} STEP_NS={ns or 'default'} SEG_CHUNK={ch:3d} LDS>={kb:2d} KB: {t:7.2f} us  {'equal' if same else 'DIFFER'}", flush=True)
