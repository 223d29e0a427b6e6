"""Developer timing: the tiled forward (512 x 512, 90 angles) through compact tile plans against the direct tiled kernel."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
theta = np.pi * np.arange(90) / 90
pc = RotatePlan(theta, 512, 512, True, dev)
pd = RotatePlan(theta, 512, 512, True, dev, plan_format="u16")
assert pc._tplan is not None and pd._tplan is None
print("tile plan bytes", pc._tplan.numel())
for B in (4, 8, 16, 32, 64):
    x = torch.rand((B, 512, 512), device=dev); out = torch.empty((B, 90, pc.PW), device=dev); o2 = torch.empty_like(out)
    tc, td = timeit(lambda: pc.forward(x, out=out)), timeit(lambda: pd.forward(x, out=o2))
    _lib.tune("TILED_SORT", 0); o3 = torch.empty_like(out)
    tu = timeit(lambda: pc.forward(x, out=o3)); _lib.tune("TILED_SORT")
    print(f"B={B}: compact tile plans, sorted 16-slot bands {tc:.0f} us, (angle, block) tasks {tu:.0f} us, direct tiled {td:.0f} us, "
          f"equal={torch.equal(out, o2) and torch.equal(out, o3)}", flush=True)
    if len(sys.argv) > 1:
        for G in (1, 2, 3, 4):
            _lib.tune("TILED_G", G)
            print(f"   G={G}: compact {timeit(lambda: pc.forward(x, out=out)):.0f} us  direct {timeit(lambda: pd.forward(x, out=o2)):.0f} us", flush=True)
        _lib.tune("TILED_G")
