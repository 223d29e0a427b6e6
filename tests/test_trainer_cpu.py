"""CPU tests of the trainer's host pieces (nets, distributions, angle stream).  The projector itself needs the GPU
(tests/test_gpu_trainer.py)."""
import math

import numpy as np
import torch

from ct_pvae_amd import trainer as tr


def test_positive_range_matches_reference_formula():
    # ctvae/helper_functions.py:198-201
    x = torch.tensor([-3.0, 0.0, 0.999, 1.0, 2.5, 40.0])
    got = tr.positive_range(x)
    xm = x.numpy().astype(np.float64) - 1
    want = np.where(xm < 0, np.exp(np.clip(xm, -1e10, 10)) + tr.EPS32, xm + 1)
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-6)
    assert (got > 0).all()


def test_net_shapes_follow_the_reference_architecture():
    fm = [int(20 * 1.1 ** i) for i in range(3)]                 # --nfm 20 --nfmm 1.1 --nb 3
    enc = tr.EncodeNet(2, fm, 2, 4, 2, 2, 4)
    dec = tr.DecodeNet(enc.channels, 2, 1, 4, 2, 2, 4)
    x = torch.randn(3, 2, 128, 128)
    skips = enc(x)
    assert [tuple(s.shape) for s in skips] == [(3, 4, 128, 128), (3, 40, 64, 64), (3, 44, 32, 32), (3, 48, 16, 16)]
    lat = [s.chunk(2, dim=1)[0] for s in skips]
    a, b = dec(lat)
    assert a.shape == b.shape == (3, 1, 128, 128)
    n = sum(p.numel() for p in enc.parameters()) + sum(p.numel() for p in dec.parameters())
    assert 3e5 < n < 3e6                                          # SURVEY: ~0.7 M parameters
    # odd sizes: periodic padding keeps ceil(n / stride)
    skips = tr.EncodeNet(2, [8, 8], 2, 4, 2, 1, 4)(torch.randn(1, 2, 37, 50))
    assert tuple(skips[1].shape[-2:]) == (19, 25) and tuple(skips[2].shape[-2:]) == (10, 13)


def test_truncated_normal():
    from scipy.stats import truncnorm
    loc, scale = torch.tensor([0.3, -0.2, 2.0]), torch.tensor([0.5, 1.0, 0.1])
    d = tr.TruncatedNormal(loc, scale, 0.0, 1e10)
    x = torch.tensor([0.1, 0.7, 2.05])
    a = (0 - loc.numpy()) / scale.numpy()
    want = truncnorm.logpdf(x.numpy(), a, np.inf, loc=loc.numpy(), scale=scale.numpy())
    np.testing.assert_allclose(d.log_prob(x).numpy(), want, rtol=1e-4, atol=1e-5)
    torch.manual_seed(0)
    big = tr.TruncatedNormal(torch.full((20000,), 0.3), torch.full((20000,), 0.5))
    s = big.rsample()
    assert (s >= 0).all() and abs(s.mean().item() - truncnorm.mean(-0.6, np.inf, loc=0.3, scale=0.5)) < 0.02
    loc = torch.tensor([0.5], requires_grad=True)
    tr.TruncatedNormal(loc, torch.tensor([0.3])).rsample().sum().backward()
    assert loc.grad is not None and torch.isfinite(loc.grad).all()


def test_kl_and_angle_stream():
    loc, scale = torch.tensor([0.0, 1.0]), torch.tensor([1.0, 2.0])
    want = torch.distributions.kl_divergence(torch.distributions.Normal(loc, scale), torch.distributions.Normal(0.0, 1.0))
    np.testing.assert_allclose(tr.kl_normal_std(loc, scale).numpy(), want.numpy(), rtol=1e-6)
    st = tr.AngleStream(180, 20, seed=1)
    seen = np.concatenate([st.next() for _ in range(9)])         # 180 draws = one pass of the shuffled stream
    assert sorted(seen.tolist()) == list(range(180))
    assert len(set(st.next().tolist())) == 20


def test_ramp_filter_and_args():
    from ct_pvae_amd.fbp import ramp_filter
    f = ramp_filter(184)
    assert f.shape == (184,) and abs(f[0]) < 1e-2 and np.argmax(f) in (92, 91, 93)
    a = tr.get_args("--nsa 20 --td 50 -b 5 --ns 2 --api 20 --pnm 1e4 --pnm_start 1e3 --random --normal -i 1000 --train".split())
    assert (a.nsa, a.td, a.batch_size, a.ns, a.api, a.pnm, a.pnm_start, a.random, a.num_iter) == (20, 50, 5, 2, 20, 1e4, 1e3, True, 1000)
    assert math.isclose(math.exp(math.log(a.pnm / a.pnm_start) / a.num_iter) ** 1000, 10.0, rel_tol=1e-9)


def test_periodic_pad_and_maxout_match_their_torch_definitions():
    """The trainer's two launch-saving autograd functions against F.pad(mode='circular') and chunk + torch.maximum:
    same values, same gradients (ctvae/models.py:219-263 periodic padding, :330-341 maxout)."""
    import torch.nn.functional as F
    from ct_pvae_amd.trainer import _Maxout, _PeriodicPad
    torch.manual_seed(0)
    x = torch.randn(2, 3, 7, 9, dtype=torch.float64, requires_grad=True)
    for pads in [(1, 1, 1, 1), (2, 1, 1, 2), (3, 2, 0, 1)]:
        a, b = _PeriodicPad.apply(x, pads), F.pad(x, list(pads), mode="circular")
        assert torch.equal(a, b)
        w = torch.randn_like(a)
        ga, = torch.autograd.grad((a * w).sum(), x)
        gb, = torch.autograd.grad((b * w).sum(), x)
        assert torch.allclose(ga, gb, atol=1e-13)
    y = torch.randn(2, 8, 5, 5, dtype=torch.float64, requires_grad=True)
    m1, m2 = _Maxout.apply(y), torch.maximum(*y.chunk(2, 1))
    assert torch.equal(m1, m2)
    w = torch.randn_like(m1)
    g1, = torch.autograd.grad((m1 * w).sum(), y)
    g2, = torch.autograd.grad((m2 * w).sum(), y)
    assert torch.equal(g1, g2)
