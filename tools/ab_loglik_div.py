"""Developer tool: the log-probability kernel of the in-tree library (IEEE quotients by a shared refined reciprocal, loglik_math.h) against a
variant built with the compiler's own division sequence (CTPVAE_VARIANT_LIB=tools/libctpvae_radon_<tag>.bin): 2^26 samples over the ranges
the model meets (ray-sums 0 .. 1e3, counts 0 .. 1e6, pnm 1 .. 1e5, eps 1e-8 .. 1e-2, masks 0 / 1) -- how many log-probabilities differ in any bit."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
d = torch.device('cuda', 0)
new = _lib.load()
old = ctypes.CDLL(os.path.abspath(os.environ["CTPVAE_VARIANT_LIB"]))
sig = _lib._SIGNATURES["ctpvae_loglik_fwd_f32"] if hasattr(_lib, "_SIGNATURES") else None
fn_old = old.ctpvae_loglik_fwd_f32
fn_old.restype = ctypes.c_int
fn_old.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
g = torch.Generator(device=d); g.manual_seed(1)
total = bad = 0
worst = 0.0
for rep in range(16):
    B, A, P = 64, 128, 512
    scale = 10.0 ** torch.empty((B, A, 1), device=d).uniform_(-3, 3, generator=g)
    proj = torch.rand((B, A, P), device=d, generator=g) * scale
    mask = (torch.rand((B, A), device=d, generator=g) < 0.8).float()
    pnm = torch.tensor([10.0 ** (rep % 6)], device=d)
    x = torch.poisson((proj * mask[:, :, None] * pnm).clamp(max=1e7)) / pnm * (torch.rand((B, A, P), device=d, generator=g) < 0.9)
    eps = 10.0 ** (-8 + rep % 7)
    a = torch.full_like(proj, float('nan')); b = torch.full_like(proj, float('nan'))
    _lib.check(new.ctpvae_loglik_fwd_f32(proj.data_ptr(), mask.data_ptr(), x.data_ptr(), B, A, P, pnm.data_ptr(), eps, a.data_ptr(), None), "new")
    assert fn_old(proj.data_ptr(), mask.data_ptr(), x.data_ptr(), B, A, P, pnm.data_ptr(), eps, b.data_ptr(), None) == 0
    torch.cuda.synchronize()
    ne = (a.view(torch.int32) != b.view(torch.int32))
    total += a.numel(); bad += int(ne.sum())
    if ne.any(): worst = max(worst, float(((a - b).abs() / b.abs().clamp_min(1e-30))[ne].max()))
print(f"{total} log-probabilities, {bad} differ in some bit from the compiler's division sequence (largest relative difference {worst:.3g})")
