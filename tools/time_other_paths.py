import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from ct_pvae_amd.forward_functions import RotatePlan
from ct_pvae_amd import helper_functions as hf, fbp, phantoms
dev = torch.device('cuda', 0)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
theta = np.pi * np.arange(90) / 90
for interp in ("nearest", "bilinear"):
    plan = RotatePlan(theta, 512, 512, True, dev, interp=interp)
    x = torch.rand((8, 512, 512), device=dev); g = torch.rand((8, 90, plan.PW), device=dev)
    print("N=512 A=90 B=8", interp, "planned", plan.planned, "fwd %.0f us" % timeit(lambda: plan.forward(x)), "bwd %.0f us" % timeit(lambda: plan.backward(g)))
# siddon: config 1 (single 128x128 phantom, 180 angles) and a 50-batch
img = torch.from_numpy(phantoms.foam_batch(50, 128, seed=0, supersample=2)).to(dev)
th = phantoms.dense_theta(180)
print("siddon B=1 A=180: %.0f us" % timeit(lambda: hf.create_sinograms(img[:1], th)), " B=50 A=180: %.0f us" % timeit(lambda: hf.create_sinograms(img, th)), " B=50 A=20: %.0f us" % timeit(lambda: hf.create_sinograms(img, th[::9])))
sino = torch.rand((50, 180, 184), device=dev, dtype=torch.float64)
filt = np.abs(np.fft.fftfreq(184)) * 2
print("iradon B=50 A=180 -> 128x128: %.0f us" % timeit(lambda: fbp.iradon(sino, th, 128, 128, filt), 5))
for mode in ("exact",):
    plan = RotatePlan(th[::9], 128, 128, True, dev, backward=mode)
    g = torch.rand((50, 20, 184), device=dev)
    print("exact bwd B=50 A=20: %.0f us" % timeit(lambda: plan.backward(g)))
plan = RotatePlan(th[::9], 128, 128, True, dev, interp="bilinear")
x = torch.rand((50, 128, 128), device=dev); g = torch.rand((50, 20, 184), device=dev)
print("bilinear B=50 A=20: fwd %.0f us bwd %.0f us" % (timeit(lambda: plan.forward(x)), timeit(lambda: plan.backward(g))))
