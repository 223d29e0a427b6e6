"""Developer timing of the 512x512 x 90-angle projector pair (BASELINE config 5) -- not part of the product."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
if os.environ.get('CTPVAE_VARIANT_LIB'): _lib.LIB_PATH = os.environ['CTPVAE_VARIANT_LIB']
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
theta = np.pi * np.arange(90) / 90
plan = RotatePlan(theta, 512, 512, True, dev)
x = torch.rand((B, 512, 512), device=dev); g = torch.rand((B, 90, plan.PW), device=dev)
out = torch.empty((B, 90, plan.PW), device=dev); gx = torch.empty_like(x)
print(f"N=512 A=90 B={B} tiled={plan.tiled} fwd %.0f us bwd %.0f us" % (timeit(lambda: plan.forward(x, out=out)), timeit(lambda: plan.backward(g, out=gx))))
if len(sys.argv) > 2:
    for G in (1, 2, 3, 4, 6, 8):
        _lib.tune("TILED_G", G)
        print(f"  G={G}: fwd %.0f us" % timeit(lambda: plan.forward(x, out=out)))
if len(sys.argv) > 3:
    _lib.tune("TILED_G")
    for ns in (1, 2, 4):
        _lib.tune("TILED_NS", ns)
        for G in (1, 2):
            _lib.tune("TILED_G", G)
            print(f"  NS={ns} G={G}: fwd %.0f us" % timeit(lambda: plan.forward(x, out=out)))
