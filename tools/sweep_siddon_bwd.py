"""Developer sweep: the pixel-driven back-projector over slices per workgroup (SIDDON_BWD_NS) and angles per LDS chunk
(SIDDON_BWD_CHUNKS), and the forward over slices per walk (SIDDON_NS), 50 x 180 x 184 (-> 184^2)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ct_pvae_amd as cp
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.recon import siddon_backproject
d = torch.device("cuda", 0)
theta = phantoms.dense_theta(180)
imgs = torch.from_numpy(phantoms.foam_batch(50, 128, seed=0, supersample=2)).to(d)
sino = cp.create_sinograms(imgs, theta, pad=True)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = siddon_backproject(sino, theta)
for ns in (1, 2, 4, 8):
    for ch in (4, 7, 15, 30):
        _lib.tune("SIDDON_BWD_NS", ns); _lib.tune("SIDDON_BWD_CHUNKS", ch)
        out = siddon_backproject(sino, theta)
        print(f"back-projector NS {ns} chunk {ch:2d}: {timed(lambda: siddon_backproject(sino, theta)):.3f} ms  equal={torch.equal(out, ref)}", flush=True)
_lib.tune("*")
big = torch.rand((50, 184, 184), device=d)
for ns in (1, 2, 4, 8):
    _lib.tune("SIDDON_NS", ns)
    t128 = timed(lambda: cp.create_sinograms(imgs, theta, pad=True))
    t184 = timed(lambda: cp.create_sinograms(big, theta, pad=False)) if ns != 2 else float("nan")
    print(f"forward SIDDON_NS {ns}: 128^2 object {t128:.3f} ms, 184^2 grid {t184:.3f} ms", flush=True)
_lib.tune("*")
