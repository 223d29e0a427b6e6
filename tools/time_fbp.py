"""Developer timing of iradon (fp64 FBP) -- not part of the product."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import fbp, phantoms
dev = torch.device('cuda', 0)
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
th = phantoms.dense_theta(180)
sino = torch.rand((50, 180, 184), device=dev, dtype=torch.float64)
filt = fbp.ramp_filter(184)
print("iradon B=50 A=180 -> 128x128: %.0f us" % timeit(lambda: fbp.iradon(sino, th, 128, 128, filt)))
print("iradon B=5 A=180 -> 128x128: %.0f us" % timeit(lambda: fbp.iradon(sino[:5], th, 128, 128, filt)))
