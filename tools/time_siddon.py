"""Developer timing of the TomoPy-style projector (create_sinograms) -- not part of the product."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import helper_functions as hf, phantoms
dev = torch.device('cuda', 0)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
img = torch.from_numpy(phantoms.foam_batch(50, 128, seed=0, supersample=2)).to(dev)
th = phantoms.dense_theta(180)
for threads in (256, 512, 1024):
    os.environ["CTPVAE_TUNE_SIDDON_THREADS"] = str(threads)
    for ppb in (0, 6, 12, 24, 45):
        if ppb: os.environ["CTPVAE_TUNE_SIDDON_PPB"] = str(ppb)
        else: os.environ.pop("CTPVAE_TUNE_SIDDON_PPB", None)
        print(f"threads={threads} ppb={ppb or 'auto'}: B=1 A=180 %.0f us | B=50 A=180 %.0f us | B=50 A=20 %.0f us" % (
            timeit(lambda: hf.create_sinograms(img[:1], th)), timeit(lambda: hf.create_sinograms(img, th)),
            timeit(lambda: hf.create_sinograms(img, th[::9]))))
