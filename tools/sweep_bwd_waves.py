"""Developer sweep: the planned backward (slice pairs) over its waves per workgroup (knob BW: a tile is 64 columns x 2 BW rows), at
a few batch sizes -- which tile height fills 256 CUs best?   python tools/sweep_bwd_waves.py [angles] [N]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
if os.environ.get("CTPVAE_VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CTPVAE_VARIANT_LIB"])
    _lib.torch_node = lambda: None
print("library:", _lib.LIB_PATH)
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
A = int(sys.argv[1]) if len(sys.argv) > 1 else 180
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
theta = np.pi * np.arange(A) / A
plan = RotatePlan(theta, N, N, True, dev); plan.backward_uses_plan = lambda S: True; plan.backward_uses_step_plan = lambda S: False
def t_us(g, out, n=100):
    plan.backward(g, out=out); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): plan.backward(g, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(r))
for B in (25, 32, 40, 50, 64, 80, 100, 128, 160):
    g = torch.rand((B, A, plan.PW), device=dev); out = torch.empty((B, N, N), device=dev)
    t_us(g, out)
    ref = plan.backward(g).clone()
    row = ["default %6.2f" % t_us(g, out)]
    for w in (4, 6, 7, 8, 9, 10, 11, 13, 16):
        with _lib.tuned("BW", w):
            t = t_us(g, out)
            ok = torch.equal(plan.backward(g), ref)
        units = (B + 1) // 2; tiles = 2 * -(-N // (2 * w))
        row.append("%2d:%6.2f%s(%d)" % (w, t, "" if ok else "!", units * tiles))
    print("A=%3d B=%3d  " % (A, B) + "  ".join(row), flush=True)
