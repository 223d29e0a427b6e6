"""Parity of the HIP path (through the C ABI) with the CPU oracle, on a real MI355X.

Bars (north_star: <=1e-5 relative fp32 vs. the reference projector on identical inputs):
  * rotate forward and TensorFlow-compatible backward: every lane adds in the oracle's order and the index
    arithmetic is unfused fp32, so NEAREST must be BIT-EXACT (no flipped sample possible) and BILINEAR as well;
  * exact backward (atomics, order not fixed) and everything compared through a tolerance: max|diff| <= 1e-5 * max|ref|;
  * siddon: bit-exact (same expressions, correctly rounded / and sqrt);
  * iradon (fp64): 1e-10 relative (convolution vs. DFT ordering)."""
import os

import numpy as np
import pytest
import torch

import ct_pvae_amd as cp
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan, rotate_tables

pytestmark = pytest.mark.gpu

REL = 1e-5
ROTATE_CASES = ["rotate_toy", "rotate_rand8", "rotate_rect_nopad", "rotate_rect_pad", "rotate_foam128_a20"]


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def rel_err(got, want):
    want = np.asarray(want, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


def nan_out(shape, device, dtype=torch.float32):
    """An output buffer for a raw launch: NaN, so that an element the launch skips cannot pass on recycled memory."""
    return torch.full(tuple(shape) if not isinstance(shape, int) else (shape,), float("nan"), dtype=dtype, device=device)


def to_np(t):
    return t.detach().cpu().numpy()


def oT(oracle, theta, plan):
    """The ORACLE's transform table of the plan's geometry (oracle/radon_oracle.c:65-79): parity tests hand the oracle its
    own tables, never the kernel's -- for a host-resident theta the product builds the same bits on the host."""
    return oracle.rotate_transforms(np.asarray(theta, dtype=np.float32), plan.PH, plan.PW)


def oTinv(oracle, theta, plan):
    return oracle.invert_transforms(oT(oracle, theta, plan))


def test_extension_is_loaded_and_sees_the_gpu():
    lib = _lib.load()
    assert os.path.samefile(_lib.LIB_PATH, os.path.join(os.path.dirname(cp.__file__), "libctpvae_radon.so"))
    assert lib.ctpvae_device_count() >= 1
    name = torch.cuda.get_device_properties(0).gcnArchName
    assert "gfx950" in name, name


def test_device_tables_match_oracle(oracle):
    """Host-resident theta: the product's tables are built on the HOST and are the oracle's bits exactly.
    Device-resident theta: built by the device kernel (device libm): every entry within 1 ulp of the host table."""
    theta = np.concatenate([phantoms.dense_theta(180), [-1.0, 4.0, 0.3]])
    for H, W in ((184, 184), (16, 12), (2, 2), (728, 728)):
        T, Tinv = rotate_tables(theta, H, W, dev())
        T0 = oracle.rotate_transforms(theta, H, W)
        np.testing.assert_array_equal(to_np(T), T0)
        np.testing.assert_array_equal(to_np(Tinv), oracle.invert_transforms(T0))
        T2, Tinv2 = rotate_tables(torch.from_numpy(theta.astype(np.float32)).to(dev()), H, W, dev())
        T2 = to_np(T2)
        # cos / sin (entries 0, 1, 3, 4) within one ulp; the offsets follow from them
        trig = [0, 1, 3, 4]
        assert (np.abs(T2[:, trig] - T0[:, trig]) <= np.spacing(np.abs(T0[:, trig]))).all()
        assert np.abs(T2 - T0).max() <= 2e-5 * max(H, W) / 184
        np.testing.assert_array_equal(to_np(Tinv2), oracle.invert_transforms(T2))   # same inverse arithmetic on either


def test_device_theta_path_flips_are_reported(oracle):
    """A theta that exists only on the device gets its table from the device's cos/sin.  The projector is then the same
    operator on a table that may differ by 1 ulp in a few entries -- a nearest-neighbour projector is discontinuous in the
    table, so the honest statement is (max rel-err, number of differing ray-sums), not a bare tolerance: with the kernel's
    own table the oracle agrees bit for bit; against the oracle's table only a small fraction of ray-sums may differ."""
    d = dev()
    foam = phantoms.foam_batch(2, 128, seed=0, supersample=2)
    theta = phantoms.dense_theta(180)
    plan = RotatePlan(torch.from_numpy(theta.astype(np.float32)).to(d), 128, 128, True, d)
    got = to_np(plan.forward(torch.from_numpy(foam).to(d)))
    geom = oracle.Geometry(128, 128, True)
    np.testing.assert_array_equal(got, oracle.rotate_fwd(foam, geom, to_np(plan.T8), 0))
    want = oracle.rotate_fwd(foam, geom, oT(oracle, theta, plan), 0)
    differing = int((got != want).sum())
    err = rel_err(got, want)
    print(f"device-theta tables: {int((to_np(plan.T8) != oT(oracle, theta, plan)).sum())} of {plan.T8.numel()} entries differ; "
          f"{differing} of {got.size} ray-sums differ, max rel-err {err:.2e}")
    assert differing <= 2e-3 * got.size and err <= 2e-2


@pytest.mark.parametrize("name", ROTATE_CASES)
@pytest.mark.parametrize("interp,use_plan", [("nearest", True), ("nearest", False), ("bilinear", False)])
def test_rotate_against_oracle_and_golden(oracle, golden_dir, name, interp, use_plan):
    """use_plan=True: the gather-plan kernels (csrc/rotate_plan.hip); False: the direct kernels (csrc/rotate.hip)."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    img, pad, theta, g = z["img"], bool(z["pad"]), z["theta"], z["g"]
    code = 0 if interp == "nearest" else 1
    plan = RotatePlan(theta, img.shape[1], img.shape[2], pad, dev(), interp=interp, backward="tf_compat",
                      use_plan=use_plan)
    assert plan.planned == (use_plan, use_plan)
    geom = oracle.Geometry(img.shape[1], img.shape[2], pad)
    T, Tinv = oT(oracle, theta, plan), oTinv(oracle, theta, plan)   # the oracle's own tables
    np.testing.assert_array_equal(to_np(plan.T8), T)                 # ... which the host-built product tables equal
    np.testing.assert_array_equal(to_np(plan.Tinv8), Tinv)
    x = torch.from_numpy(img).to(dev())
    gt = torch.from_numpy(g).to(dev())

    fwd = to_np(plan.forward(x))
    np.testing.assert_array_equal(fwd, oracle.rotate_fwd(img, geom, T, code))       # bit-exact
    assert rel_err(fwd, z[f"fwd_{interp}"]) <= REL                                  # golden (host-built tables)

    bwd = to_np(plan.backward(gt))
    np.testing.assert_array_equal(bwd, oracle.rotate_bwd_tfcompat(g, geom, Tinv, code))
    assert rel_err(bwd, z[f"bwd_tfcompat_{interp}"]) <= REL

    plan_x = RotatePlan(theta, img.shape[1], img.shape[2], pad, dev(), interp=interp, backward="exact")
    bx = to_np(plan_x.backward(gt))
    assert rel_err(bx, oracle.rotate_bwd_exact(g, geom, T, code)) <= REL
    assert rel_err(bx, z[f"bwd_exact_{interp}"]) <= REL


@pytest.mark.parametrize("shape,kw", [((4, 10, 7, 1), dict(integrate_vae=True)), ((10, 7, 3), dict(dim=3)),
                                      ((128, 128), dict(dim=2)), ((50, 128, 128, 1), dict(dim=2, integrate_vae=True))])
def test_pad_phantom_on_the_device(oracle, shape, kw):
    """a1: the materialising pad (ctvae/forward_functions.py:18-46) on device tensors, all three layouts, against the
    oracle's; and projecting the padded tensor WITHOUT padding again equals projecting the original with pad=True."""
    d = dev()
    rng = np.random.default_rng(1)
    x = rng.random(shape, dtype=np.float32)
    out = cp.pad_phantom(torch.from_numpy(x).to(d), **kw)
    assert out.device.type == "cuda"
    if kw.get("integrate_vae"):
        slices, got = x[..., 0], to_np(out)[..., 0]
    elif kw.get("dim") == 3:
        slices, got = np.transpose(x, (2, 0, 1)), np.transpose(to_np(out), (2, 0, 1))
    else:
        slices, got = x[None], to_np(out)[None]
    geom = oracle.Geometry(slices.shape[1], slices.shape[2], True)
    np.testing.assert_array_equal(got, oracle.pad_phantom(slices, geom))
    if slices.shape[1] == slices.shape[2]:          # square: the padded canvas is what pad=True projects
        theta = np.array([0.0, 0.4, 1.3, 2.2])
        a = cp.project_tf_fast(torch.from_numpy(x).to(d), theta, pad=True, **kw)
        b = cp.project_tf_fast(out, theta, pad=False, **kw)
        assert torch.equal(a, b)


def test_toy_known_answers_through_the_public_api():
    # scripts/images_to_sinograms.py:54-59 / ctvae/toy_mcmc_v2_functions.py:41
    x = torch.from_numpy(phantoms.toy_images()).to(dev())
    theta = np.array([0, np.pi / 2])
    for b, want in enumerate(([[.4, .6], [.7, .3]], [[.4, .6], [.3, .7]])):
        out = cp.project_tf_fast(x[b], theta, pad=False, dim=2, integrate_vae=False)
        assert out.shape == (2, 2, 1)
        np.testing.assert_allclose(to_np(out)[..., 0], want, atol=2e-7)
    out = cp.project_tf_fast(x[..., None], theta, pad=False, dim=2, integrate_vae=True)
    assert out.shape == (2, 2, 2, 1)
    np.testing.assert_allclose(to_np(out)[0, :, :, 0], [[.4, .6], [.7, .3]], atol=2e-7)
    sid = cp.create_sinogram(phantoms.toy_images()[0], theta, pad=False)
    np.testing.assert_allclose(sid, [[.4, .6], [.7, .3]], atol=2e-7)


def test_public_layouts_match_the_reference(oracle):
    rng = np.random.default_rng(0)
    theta = rng.uniform(0, np.pi, 6)
    d = dev()
    # (i) integrate_vae: [B][X][Y][1] -> [B][A][P][1]          ctvae/helper_functions.py:359
    x = rng.random((3, 12, 12, 1), dtype=np.float32)
    got = cp.project_tf_fast(torch.from_numpy(x).to(d), theta, pad=True, dim=2, integrate_vae=True)
    want = oracle.project_tf_fast(x, theta, pad=True, dim=2, integrate_vae=True)
    assert got.shape == want.shape == (3, 6, 20, 1) and rel_err(to_np(got), want) <= REL
    # (ii) dim=3: [X][Y][Z] -> [A][P][Z]                        ctvae/tomopy_forward_compare.py:52
    x = rng.random((12, 10, 2), dtype=np.float32)
    got = cp.project_tf_fast(torch.from_numpy(x).to(d), theta, pad=True)
    want = oracle.project_tf_fast(x, theta, pad=True)
    assert got.shape == want.shape and rel_err(to_np(got), want) <= REL
    # (iii) dim=2: [X][Y] -> [A][P][1]                          ctvae/main_ct_vae.py:523-524
    got = cp.project_tf_fast(torch.from_numpy(x[..., 0]).to(d), theta, pad=True, dim=2)
    want = oracle.project_tf_fast(x[..., 0], theta, pad=True, dim=2)
    assert got.shape == want.shape and rel_err(to_np(got), want) <= REL
    # low_mem: bilinear, [X][Y][Z] -> [A][Y][Z]                 ctvae/tomopy_forward_compare.py:56
    got = cp.project_tf_low_mem(torch.from_numpy(x).to(d), theta, pad=True)
    want = oracle.project_tf_fast(x, theta, pad=True, interp=1)
    assert got.shape == want.shape and rel_err(to_np(got), want) <= REL
    # float64 pixels (the compare script feeds xdesign's float64) come back as float64
    got64 = cp.project_tf_low_mem(torch.from_numpy(x.astype(np.float64)).to(d), theta, pad=True)
    assert got64.dtype == torch.float64 and rel_err(to_np(got64), want) <= REL
    # a theta tensor living on the device (the training loop) gives the same answer
    got_t = cp.project_tf_fast(torch.from_numpy(x).to(d), torch.from_numpy(theta.astype(np.float32)).to(d), pad=True)
    assert rel_err(to_np(got_t), oracle.project_tf_fast(x, theta, pad=True)) <= REL


@pytest.mark.parametrize("backward", ["tf_compat", "exact"])
def test_autograd_matches_the_raw_backward(oracle, backward):
    rng = np.random.default_rng(1)
    d = dev()
    theta = rng.uniform(0, np.pi, 5)
    x = torch.from_numpy(rng.random((2, 16, 16, 1), dtype=np.float32)).to(d).requires_grad_(True)
    out = cp.project_tf_fast(x, theta, pad=True, dim=2, integrate_vae=True, backward=backward)
    g = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32)).to(d)
    (out * g).sum().backward()
    geom = oracle.Geometry(16, 16, True)
    T = oracle.rotate_transforms(theta, geom.PH, geom.PW)
    if backward == "tf_compat":
        want = oracle.rotate_bwd_tfcompat(to_np(g)[..., 0], geom, oracle.invert_transforms(T), 0)
    else:
        want = oracle.rotate_bwd_exact(to_np(g)[..., 0], geom, T, 0)
    assert x.grad.shape == x.shape
    assert rel_err(to_np(x.grad)[..., 0], want) <= REL


def test_exact_backward_is_the_transpose_at_full_size():
    """Size-independent property at BASELINE's headline shape (B=50, 128x128, 20 angles): <Ax, g> == <x, A^T g>."""
    d = dev()
    rng = np.random.default_rng(2)
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, 20)]
    for interp in ("nearest", "bilinear"):
        plan = RotatePlan(theta, 128, 128, True, d, interp=interp, backward="exact")
        x = torch.from_numpy(rng.standard_normal((50, 128, 128)).astype(np.float32)).to(d)
        g = torch.from_numpy(rng.standard_normal((50, 20, 184)).astype(np.float32)).to(d)
        lhs = (plan.forward(x).double() * g.double()).sum().item()
        rhs = (x.double() * plan.backward(g).double()).sum().item()
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)


@pytest.mark.parametrize("use_plan", [True, False])
def test_full_size_properties(oracle, use_plan):
    """BASELINE config 2 (B=50, N=128, 20 sparse angles) and the 180-angle evaluation set: linearity, slice
    independence, axis-aligned analytic answers, and a sampled comparison with the oracle."""
    d = dev()
    foam = phantoms.foam_batch(50, 128, seed=0, supersample=2)
    x = torch.from_numpy(foam).to(d)
    theta180 = phantoms.dense_theta(180)
    plan = RotatePlan(theta180, 128, 128, True, d, use_plan=use_plan)
    assert plan.planned == (use_plan, use_plan)
    s = plan.forward(x)
    assert s.shape == (50, 180, 184)
    sn = to_np(s)
    # theta = 0: column sums; theta = pi/2 (index 90): reversed row sums (padded by 28 zeros each side)
    np.testing.assert_allclose(sn[:, 0, 28:156], foam.sum(axis=1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(sn[:, 90, 28:156], foam.sum(axis=2)[:, ::-1], rtol=1e-5, atol=1e-5)
    assert np.all(sn[:, 0, :28] == 0) and np.all(sn[:, 0, 156:] == 0)
    # slice independence and linearity
    np.testing.assert_array_equal(to_np(plan.forward(x[7:8])), sn[7:8])
    y = torch.roll(x, 1, 0)
    lin = to_np(plan.forward(x + 2 * y))
    assert rel_err(lin, sn + 2 * np.roll(sn, 1, 0)) <= REL
    # three whole objects against the oracle, bit for bit
    geom = oracle.Geometry(128, 128, True)
    np.testing.assert_array_equal(sn[[0, 23, 49]], oracle.rotate_fwd(foam[[0, 23, 49]], geom, oT(oracle, theta180, plan), 0))
    g = torch.from_numpy(np.random.default_rng(3).standard_normal((50, 180, 184)).astype(np.float32)).to(d)
    b = to_np(plan.backward(g))
    np.testing.assert_array_equal(b[[5, 31]], oracle.rotate_bwd_tfcompat(to_np(g)[[5, 31]], geom, oTinv(oracle, theta180, plan), 0))


@pytest.mark.parametrize("shape,pad,A,S", [((128, 128), True, 20, 5), ((128, 128), True, 7, 1), ((40, 100), True, 33, 3),
                                          ((65, 31), False, 9, 2), ((2, 2), False, 2, 2), ((130, 70), True, 181, 2)])
def test_planned_equals_direct(oracle, shape, pad, A, S):
    """The gather-plan kernels and the direct kernels are the same operator, bit for bit, on ragged shapes, odd angle
    counts (the plans pack rows by 8 and angles by 8), unpadded canvases (negative-tie rounding) and tiny batches."""
    d = dev()
    rng = np.random.default_rng(A)
    theta = rng.uniform(-1.0, 4.0, A)
    x = torch.from_numpy(rng.standard_normal((S,) + shape).astype(np.float32)).to(d)
    pa = RotatePlan(theta, shape[0], shape[1], pad, d, use_plan=True)
    pb = RotatePlan(theta, shape[0], shape[1], pad, d, use_plan=False)
    assert pa.planned == (True, True) and pb.planned == (False, False)
    fa, fb = pa.forward(x), pb.forward(x)
    assert torch.equal(fa, fb)
    g = torch.from_numpy(rng.standard_normal(tuple(fa.shape)).astype(np.float32)).to(d)
    assert torch.equal(pa.backward(g), pb.backward(g))
    geom = oracle.Geometry(shape[0], shape[1], pad)
    np.testing.assert_array_equal(to_np(fa[:1]), oracle.rotate_fwd(to_np(x[:1]), geom, oT(oracle, theta, pa), 0))
    np.testing.assert_array_equal(to_np(pa.backward(g)[:1]), oracle.rotate_bwd_tfcompat(to_np(g[:1]), geom, oTinv(oracle, theta, pa), 0))


def test_index_divisions_by_multiplication_equal_divisions(oracle):
    """The planned kernels map block numbers to (unit, class, group / tile) and task numbers to (bin block, angle) with
    multiplications by host- / plan-made reciprocal words where the operands are small (always, in practice); knob NO_MAGIC takes
    the divisions: the same bits, forward and backward, on shapes with partial octets of units, one-angle classes, one-tile units."""
    d = dev()
    rng = np.random.default_rng(77)
    for (H, W), pad, A, S in (((128, 128), True, 20, 53), ((40, 100), True, 33, 11), ((64, 64), False, 1, 9), ((128, 128), True, 180, 21)):
        theta = rng.uniform(-1.0, 4.0, A)
        x = torch.from_numpy(rng.standard_normal((S, H, W)).astype(np.float32)).to(d)
        plan = RotatePlan(theta, H, W, pad, d, plan_format="u16")
        f = plan.forward(x)
        g = torch.from_numpy(rng.standard_normal(tuple(f.shape)).astype(np.float32)).to(d)
        b = plan.backward(g)
        with _lib.tuned("NO_MAGIC", 1):
            assert torch.equal(plan.forward(x), f) and torch.equal(plan.backward(g), b), (H, W, A, S)
        geom = oracle.Geometry(H, W, pad)
        np.testing.assert_array_equal(to_np(f[-1:]), oracle.rotate_fwd(to_np(x[-1:]), geom, oT(oracle, theta, plan), 0))


@pytest.mark.parametrize("A,S", [(20, 300), (20, 301), (90, 301), (7, 700), (90, 388)])
def test_two_part_piece_lists_equal_one_cut(oracle, A, S):
    """Launches of more than one round of workgroups cut their first units into coarse pieces and the rest into finer ones (round
    5; a still finer third part where it fills the last round, S = 388 x 90 angles): which piece carries a ray does not touch its
    sum.  The library's own cut, one cut for the whole launch (knob MIXG=0) and
    forced cuts -- part boundaries off the octets of units, a second part of one unit, an empty first part, odd batches --
    give the same bits, and the oracle's."""
    d = dev()
    rng = np.random.default_rng(A + S)
    theta = rng.uniform(-1.0, 4.0, A)
    x = torch.from_numpy(rng.standard_normal((S, 128, 128)).astype(np.float32)).to(d)
    plan = RotatePlan(theta, 128, 128, True, d, plan_format="u16")
    auto = plan.forward(x)
    with _lib.tuned("MIXG", 0):
        one = plan.forward(x)
    assert torch.equal(auto, one)
    units = (S + 1) // 2
    for g2, u1 in ((5, 128), (3, units - 1), (2, 0), (7, 13), (4, units)):
        with _lib.tuned("NS", 2), _lib.tuned("MIXG_G2", g2), _lib.tuned("MIXG_U1", u1):
            assert torch.equal(plan.forward(x), one), (g2, u1)
    # ... and a third part, finer again, behind the second: boundaries off the octets, an empty second part, a third of one unit
    for g1, g2, u1, g3, u2 in ((1, 3, 64, 4, 101), (2, 3, 13, 5, 13), (1, 2, 0, 7, units - 1), (1, 3, 128, 4, 140), (3, 2, 7, 1, 90)):
        with _lib.tuned("NS", 2), _lib.tuned("MIXG_G1", g1), _lib.tuned("MIXG_G2", g2), _lib.tuned("MIXG_U1", u1), \
                _lib.tuned("MIXG_G3", g3), _lib.tuned("MIXG_U2", u2):
            assert torch.equal(plan.forward(x), one), (g1, g2, u1, g3, u2)
    geom = oracle.Geometry(128, 128, True)
    for k in (0, S // 2, S - 1):
        np.testing.assert_array_equal(to_np(auto[k:k + 1]), oracle.rotate_fwd(to_np(x[k:k + 1]), geom, oT(oracle, theta, plan), 0))


@pytest.mark.parametrize("shape,pad,A,S", [((128, 128), True, 20, 5), ((128, 128), True, 180, 51), ((40, 100), True, 33, 3),
                                          ((65, 31), False, 9, 1), ((2, 2), False, 2, 2), ((128, 128), True, 20, 17)])
def test_paired_slices_equal_single_slices(oracle, shape, pad, A, S):
    """The planned forward runs one or two slices per workgroup (two: float2-interleaved in LDS behind one index stream);
    both forms are the same operator bit for bit -- odd batches (half-empty last pair), one slice, ragged shapes."""
    d = dev()
    rng = np.random.default_rng(A + S)
    theta = rng.uniform(-1.0, 4.0, A)
    x = torch.from_numpy(rng.standard_normal((S,) + shape).astype(np.float32)).to(d)
    plan = RotatePlan(theta, shape[0], shape[1], pad, d)
    assert plan.planned[0]
    with _lib.tuned("NS", 1):
        one = plan.forward(x)
        _lib.tune("NS", 2)
        two = plan.forward(x)
    auto = plan.forward(x)
    assert torch.equal(one, two) and torch.equal(one, auto)
    geom = oracle.Geometry(shape[0], shape[1], pad)
    np.testing.assert_array_equal(to_np(two[-1:]), oracle.rotate_fwd(to_np(x[-1:]), geom, oT(oracle, theta, plan), 0))


@pytest.mark.parametrize("shape,pad,A,S", [((128, 128), True, 20, 5), ((128, 128), True, 180, 7), ((40, 100), True, 33, 3),
                                          ((65, 31), False, 9, 1), ((2, 2), False, 2, 2), ((128, 128), True, 70, 40),
                                          ((128, 128), True, 16, 50), ((128, 128), True, 17, 9), ((64, 72), True, 32, 4)])
def test_paired_backward_equals_single(oracle, shape, pad, A, S):
    """The planned backward runs one or two slices per workgroup (two: cotangent rows fetched together and interleaved
    as float2 behind one index stream, 32-angle chunks); both forms are the same operator bit for bit.  (A <= 32: the
    instantiation that requests both index groups up front -- 16, 17 and 32 angles are its edges, 33 the other kernel's.)"""
    d = dev()
    rng = np.random.default_rng(A * 7 + S)
    theta = rng.uniform(-1.0, 4.0, A)
    plan = RotatePlan(theta, shape[0], shape[1], pad, d)
    assert plan.planned[1]
    g = torch.from_numpy(rng.standard_normal((S, A, plan.PW)).astype(np.float32)).to(d)
    with _lib.tuned("BNS", 1):
        one = plan.backward(g)
        _lib.tune("BNS", 2)
        two = plan.backward(g)
    auto = plan.backward(g)
    assert torch.equal(one, two) and torch.equal(one, auto)
    geom = oracle.Geometry(shape[0], shape[1], pad)
    np.testing.assert_array_equal(to_np(two[-1:]), oracle.rotate_bwd_tfcompat(to_np(g[-1:]), geom, oTinv(oracle, theta, plan), 0))


@pytest.mark.parametrize("shape,pad,A,S", [((128, 128), True, 20, 5), ((192, 192), True, 100, 3), ((40, 100), False, 33, 4),
                                          ((300, 200), True, 7, 2), ((128, 128), True, 20, 33)])
def test_paired_segment_backward_equals_single(oracle, shape, pad, A, S):
    """The direct (segment) backward runs one or two slices per workgroup (two: segments interleaved as float2, one
    coordinate / address / ds_read_b64 per tap for both); same operator bit for bit -- padded (all-inside fast loop),
    unpadded (zero-fill classes), more angles than one chunk, odd batches."""
    d = dev()
    rng = np.random.default_rng(A * 3 + S)
    theta = rng.uniform(-1.0, 4.0, A)
    plan = RotatePlan(theta, shape[0], shape[1], pad, d, use_plan=False)
    g = torch.from_numpy(rng.standard_normal((S, A, plan.PW)).astype(np.float32)).to(d)
    with _lib.tuned("SEG_NS", 1):
        one = plan.backward(g)
        _lib.tune("SEG_NS", 2)
        two = plan.backward(g)
    auto = plan.backward(g)
    assert torch.equal(one, two) and torch.equal(one, auto)
    geom = oracle.Geometry(shape[0], shape[1], pad)
    np.testing.assert_array_equal(to_np(two[-1:]), oracle.rotate_bwd_tfcompat(to_np(g[-1:]), geom, oTinv(oracle, theta, plan), 0))


def test_mixed_planned_forward_direct_backward(oracle):
    """192x192: the slice still fits the forward plan's LDS image (148 KiB), but P = 274 bins do not fit the backward
    plan's byte taps -- forward planned, backward direct, both bit-exact."""
    d = dev()
    rng = np.random.default_rng(8)
    img = rng.random((2, 192, 192), dtype=np.float32)
    theta = rng.uniform(0, np.pi, 5)
    plan = RotatePlan(theta, 192, 192, True, d)
    assert plan.PW == 274 and plan.planned == (True, False)
    geom = oracle.Geometry(192, 192, True)
    np.testing.assert_array_equal(to_np(plan.forward(torch.from_numpy(img).to(d))), oracle.rotate_fwd(img, geom, oT(oracle, theta, plan), 0))
    g = rng.standard_normal((2, 5, 274)).astype(np.float32)
    np.testing.assert_array_equal(to_np(plan.backward(torch.from_numpy(g).to(d))),
                                  oracle.rotate_bwd_tfcompat(g, geom, oTinv(oracle, theta, plan), 0))


def test_inputs_in_other_forms(oracle):
    """theta as a python list / CPU tensor; a non-contiguous phantom; a single slice; many angles for one tiny image."""
    d = dev()
    rng = np.random.default_rng(9)
    x = rng.random((12, 10, 4), dtype=np.float32)
    theta = [0.1, 0.9, 2.0]
    want = oracle.project_tf_fast(x, np.array(theta), pad=True)
    got = cp.project_tf_fast(torch.from_numpy(x).to(d), theta, pad=True)
    assert rel_err(to_np(got), want) <= REL
    got = cp.project_tf_fast(torch.from_numpy(x).to(d), torch.tensor(theta, dtype=torch.float64), pad=True)
    assert rel_err(to_np(got), want) <= REL
    xt = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).to(d).permute(1, 0, 2)   # non-contiguous view
    assert not xt.is_contiguous()
    assert rel_err(to_np(cp.project_tf_fast(xt, theta, pad=True)), want) <= REL
    th = np.linspace(0, 2 * np.pi, 300)
    one = rng.random((1, 4, 4, 1), dtype=np.float32)
    got = cp.project_tf_fast(torch.from_numpy(one).to(d), th, pad=True, dim=2, integrate_vae=True)
    assert rel_err(to_np(got), oracle.project_tf_fast(one, th, pad=True, dim=2, integrate_vae=True)) <= REL


def test_large_image_is_tiled(oracle):
    """512x512 (BASELINE config 5) does not fit LDS: the forward cuts the slice into 96x64 tiles, every tile staged
    once for all angles.  Same taps as the whole-slice kernels; the sum is associated tile by tile, which the oracle
    restates (rotate_fwd_tiled) -- bit-exact against that, within REL of the row-sequential sum, and bit-exact against
    the row-sequential oracle when tiling is switched off (generic kernel)."""
    d = dev()
    rng = np.random.default_rng(4)
    img = rng.random((3, 512, 512), dtype=np.float32)
    theta = np.array([0.0, 0.4, np.pi / 4, np.pi / 2, 2.0, 3.0])
    plan = RotatePlan(theta, 512, 512, True, d)
    assert plan.PW == 728 and plan.planned[0] is False and plan.tiled
    geom = oracle.Geometry(512, 512, True)
    x = torch.from_numpy(img).to(d)
    got = to_np(plan.forward(x))
    np.testing.assert_array_equal(got[[0, 2]], oracle.rotate_fwd_tiled(img[[0, 2]], geom, oT(oracle, theta, plan), oracle.tile_shape(geom.H, geom.W)))
    seq = oracle.rotate_fwd(img[:1], geom, oT(oracle, theta, plan), 0)
    assert rel_err(got[:1], seq) <= REL
    # a slice's sinogram does not depend on its batch (1, 2 and 3 slices take 1, 2 and 4 slices per workgroup)
    assert torch.equal(plan.forward(x[1:2])[0], plan.forward(x)[1]) and torch.equal(plan.forward(x[1:3])[0], plan.forward(x)[1])
    untiled = RotatePlan(theta, 512, 512, True, d, use_plan=False)
    assert not untiled.tiled
    np.testing.assert_array_equal(to_np(untiled.forward(x[:1])), seq)
    g = rng.standard_normal((1, 6, 728)).astype(np.float32)
    np.testing.assert_array_equal(to_np(plan.backward(torch.from_numpy(g).to(d))),
                                  oracle.rotate_bwd_tfcompat(g, geom, oTinv(oracle, theta, plan), 0))


@pytest.mark.parametrize("shape,pad,A,S", [((300, 200), True, 7, 2), ((129, 385), False, 5, 1), ((256, 256), True, 33, 4),
                                          ((200, 200), True, 400, 3), ((200, 200), True, 1, 5)])
def test_tiled_forward_ragged(oracle, shape, pad, A, S):
    """Edge tiles smaller than 96x64, unpadded canvases (negative-tie rounding at the canvas edge), odd angle counts, one
    angle (an empty bank class), and more angles than fit behind a 4-slice tile in LDS (the kernel then scans the
    transform table instead of reading its LDS copy and class list)."""
    d = dev()
    rng = np.random.default_rng(A)
    theta = rng.uniform(-1.0, 4.0, A)
    img = rng.standard_normal((S,) + shape).astype(np.float32)
    plan = RotatePlan(theta, shape[0], shape[1], pad, d)
    assert plan.tiled
    geom = oracle.Geometry(shape[0], shape[1], pad)
    np.testing.assert_array_equal(to_np(plan.forward(torch.from_numpy(img).to(d))),
                                  oracle.rotate_fwd_tiled(img, geom, oT(oracle, theta, plan), oracle.tile_shape(geom.H, geom.W)))


def test_siddon_against_oracle_and_golden(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "siddon.npz"))
    for case, pad in (("toy", False), ("rand", True), ("rect", True), ("foam", True)):
        img, theta = z[case + "_img"], z[case + "_theta"]
        got = cp.create_sinograms(img, theta, pad=pad)             # [S][A][dx]
        want = np.swapaxes(z[case + "_out"], 0, 1)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(got[0], cp.create_sinogram(img[0], theta, pad=pad))
    # dense 180-angle dataset geometry, 4 foam images (scripts/images_to_sinograms.py:34,61-68)
    foam = phantoms.foam_batch(4, 128, seed=1, supersample=2)
    theta = phantoms.dense_theta(180)
    got = cp.create_sinograms(foam, theta, pad=True)
    assert got.shape == (4, 180, 184)
    np.testing.assert_array_equal(got, np.swapaxes(oracle.siddon_project(foam, theta, pad=True), 0, 1))


def test_paired_siddon_equals_single():
    """create_sinograms walks every ray once for two slices (interleaved in LDS) when the call has slices to pair; same
    numbers as one slice per workgroup, bit for bit, odd batches included."""
    foam = phantoms.foam_batch(5, 128, seed=2, supersample=2)
    theta = phantoms.dense_theta(180)[::7]
    with _lib.tuned("SIDDON_NS", 1):
        one = cp.create_sinograms(foam, theta, pad=True)
        _lib.tune("SIDDON_NS", 2)
        two = cp.create_sinograms(foam, theta, pad=True)
    auto = cp.create_sinograms(foam, theta, pad=True)
    np.testing.assert_array_equal(one, two)
    np.testing.assert_array_equal(one, auto)
    np.testing.assert_array_equal(one[4], cp.create_sinogram(foam[4], theta, pad=True))


@pytest.mark.parametrize("S,shape", [(3, (128, 128)), (5, (40, 57)), (8, (64, 64)), (9, (184, 184)), (17, (33, 47))])
def test_many_slices_per_walk_equal_the_lds_kernels(oracle, S, shape):
    """Round 3: with >= 3 slices create_sinograms interleaves 4 or 8 slices per pixel in a workspace and one walk of a ray
    serves them all from L2 (ctpvae_siddon_fwd_ws_f32) -- the same bits as one slice per workgroup in LDS and as the oracle,
    whatever the group size (ragged last groups, grids whose slice pairs do not fit LDS, odd rectangles); and its SIRT store
    (meas - A x) / rn2 equals that expression on the ray-sums."""
    from ct_pvae_amd.helper_functions import _siddon_forward, _siddon_tables
    d = dev()
    rng = np.random.default_rng(S)
    img = rng.random((S,) + shape, dtype=np.float32)
    theta = rng.uniform(0.0, np.pi, 11)
    theta[:2] = [0.0, np.pi / 2]
    res = {}
    for ns in (1, 4, 8, None):
        _lib.tune("SIDDON_NS", *([ns] if ns else []))
        res[ns] = cp.create_sinograms(img, theta, pad=True)
    _lib.tune("SIDDON_NS")
    for ns in (4, 8, None):
        np.testing.assert_array_equal(res[1], res[ns])
    np.testing.assert_array_equal(res[None], np.swapaxes(oracle.siddon_project(img, theta, pad=True), 0, 1))
    t = torch.from_numpy(img).to(d)
    tables = _siddon_tables(theta, d)
    sino = torch.from_numpy(res[None]).to(d)
    meas = torch.from_numpy(rng.random(res[None].shape, dtype=np.float32)).to(d)
    rn2 = torch.from_numpy(rng.random(res[None].shape[1:], dtype=np.float32)).to(d)
    rn2[::3, ::5] = 0.0
    upd = _siddon_forward(t, tables, sino.shape[2], meas=meas, rn2=rn2)
    want = torch.where(rn2 != 0, (meas - sino) / torch.where(rn2 != 0, rn2, torch.ones_like(rn2)), torch.zeros_like(sino))
    assert torch.equal(upd, want)


def test_iradon_against_oracle_and_golden(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "iradon.npz"))
    d = dev()
    got = cp.iradon(torch.from_numpy(z["sino"]).to(d), z["theta"], int(z["x_size"]), int(z["y_size"]), z["filt"])
    assert got.dtype == torch.float64
    assert rel_err(to_np(got), z["recon"]) <= 1e-10
    with pytest.raises(ValueError, match="does not match the number of"):
        cp.iradon(torch.from_numpy(z["sino"]).to(d), z["theta"][:-1], 4, 4, z["filt"])
    # full size: 128x128 from 180 x 184
    rng = np.random.default_rng(0)
    sino = rng.random((2, 180, 184))
    theta = phantoms.dense_theta(180)
    filt = np.abs(np.fft.fftfreq(184)) * 2
    got = cp.iradon(torch.from_numpy(sino).to(d), theta, 128, 128, filt)
    assert rel_err(to_np(got), oracle.iradon(sino, theta, 128, 128, filt)) <= 1e-10
    # 17 sinograms: two per thread in the back-projection (the last thread group ragged), same sums
    sino17 = rng.random((17, 24, 184))
    theta24 = np.sort(rng.uniform(0.0, np.pi, 24))
    got = cp.iradon(torch.from_numpy(sino17).to(d), theta24, 128, 128, filt)
    assert rel_err(to_np(got), oracle.iradon(sino17, theta24, 128, 128, filt)) <= 1e-10
    np.testing.assert_array_equal(to_np(got[16]), to_np(cp.iradon(torch.from_numpy(sino17[16:]).to(d), theta24, 128, 128, filt)[0]))


def test_loglik_against_oracle_golden_and_autograd(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "loglik.npz"))
    d = dev()
    proj = torch.from_numpy(z["proj"]).to(d).requires_grad_(True)
    mask, x = torch.from_numpy(z["mask"]).to(d), torch.from_numpy(z["x"]).to(d)
    pnm = torch.tensor(float(z["pnm"]), device=d, requires_grad=True)
    eps = float(z["eps"])
    out = cp.gaussian_poisson_log_prob(proj, mask, x, pnm, eps)
    assert rel_err(to_np(out), z["out"]) <= 2e-6
    out.sum().backward()
    # reference formula in torch (float64, CPU) -> autograd
    p64 = torch.from_numpy(z["proj"]).double().requires_grad_(True)
    n64 = torch.tensor(float(z["pnm"]), dtype=torch.float64, requires_grad=True)
    loc = p64 * torch.from_numpy(z["mask"]).double()[..., None]
    scale = eps + torch.sqrt(loc / n64 + eps)
    torch.distributions.Normal(loc, scale).log_prob(torch.from_numpy(z["x"]).double()).sum().backward()
    assert rel_err(to_np(proj.grad), p64.grad.numpy()) <= 1e-4
    assert abs(pnm.grad.item() - n64.grad.item()) <= 1e-3 * abs(n64.grad.item())


def test_calculate_log_prob_M_given_R(oracle):
    """The differentiated caller of the path, ctvae/helper_functions.py:336-368, with a random angle subset."""
    rng = np.random.default_rng(5)
    d = dev()
    B, N, A_all, api = 3, 16, 12, 5
    theta = np.linspace(0, np.pi, A_all, endpoint=False)
    recon = rng.random((B, N, N, 1), dtype=np.float32)
    mask = np.zeros((B, A_all), np.float32)
    mask[:, ::2] = 1 / 6
    P = cp.num_proj_pix(N, N)
    meas = rng.random((B, A_all, P), dtype=np.float32)
    angles_i = rng.permutation(A_all)[:api]
    eps = float(np.finfo(np.float32).eps)
    proj = oracle.project_tf_fast(recon, theta[angles_i], pad=True, dim=2, integrate_vae=True)[..., 0]
    want = oracle.loglik(proj, mask[:, angles_i], meas[:, angles_i], 1e3, eps)
    # host theta (the dataset's angle list): host-built tables, ray-sums equal to the oracle's bit for bit; what is left
    # is the device's logf against the host's (<= 2 ulp of log(scale), |log(scale)| < 8: <= 2e-6 absolute, against
    # |lp| of order 1..1e7) -- well inside the north star's 1e-5
    for th, ai in ((theta, angles_i), (theta, torch.from_numpy(angles_i).to(d)),
                   (torch.from_numpy(theta.astype(np.float32)).to(d), torch.from_numpy(angles_i).to(d))):
        got = cp.calculate_log_prob_M_given_R(torch.from_numpy(recon).to(d), torch.from_numpy(mask).to(d),
                                              torch.from_numpy(meas).to(d), 1e3, eps, theta=th, angles_i=ai, pad=True)
        assert got.shape == (B, api, P, 1)
        assert rel_err(to_np(got)[..., 0], want) <= REL


@pytest.mark.parametrize("B,N,A", [(5, 128, 20), (50, 128, 20), (3, 40, 7), (60, 64, 90), (3, 256, 6)])
def test_fused_projector_loglik_equals_two_steps(B, N, A):
    """calculate_log_prob_M_given_R in one launch (planned forward + log-likelihood epilogue) against the
    two-step path it replaces (project_tf_fast, then gaussian_poisson_log_prob): same values bit for bit, same
    gradients w.r.t. the reconstruction and a trainable pnm (also through slice pairs, and through the tiled forward's
    reduce pass: the last two cases)."""
    from ct_pvae_amd.helper_functions import gaussian_poisson_log_prob
    d = dev()
    rng = np.random.default_rng(B + N)
    theta = torch.from_numpy(np.linspace(0, np.pi, A, endpoint=False).astype(np.float32)).to(d)
    P = cp.num_proj_pix(N, N)
    mask = torch.full((B, A), 1.0 / A, device=d)
    meas = torch.from_numpy(rng.random((B, A, P), dtype=np.float32) * 3).to(d)
    eps = float(np.finfo(np.float32).eps)
    up = torch.from_numpy(rng.standard_normal((B, A, P, 1)).astype(np.float32)).to(d)
    outs = []
    for fused in (True, False):
        x = torch.from_numpy(rng_img(B, N)).to(d).requires_grad_(True)
        pnm = torch.tensor(1e3, device=d, requires_grad=True)
        if fused:
            lp = cp.calculate_log_prob_M_given_R(x, mask, meas, pnm, eps, theta=theta, pad=True)
        else:
            proj = cp.project_tf_fast(x, theta, pad=True, dim=2, integrate_vae=True)
            lp = gaussian_poisson_log_prob(proj[..., 0], mask, meas, pnm, eps).unsqueeze(-1)
        (lp * up).sum().backward()
        outs.append((lp.detach(), x.grad.detach(), pnm.grad.detach()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert abs(float(outs[0][2]) - float(outs[1][2])) <= 1e-4 * abs(float(outs[1][2])) + 1e-12   # atomics order


@pytest.mark.parametrize("B,N,A", [(5, 128, 20), (51, 128, 20), (81, 48, 12), (3, 256, 6), (17, 256, 6)])
@pytest.mark.parametrize("upstream", ["per_object", "scalar", "elementwise"])
def test_fused_loglik_backward_is_one_launch_and_matches_two_steps(B, N, A, upstream):
    """Backward half of SURVEY 8 f1: with only the reconstruction differentiated, the fused forward stores d lp / d sino
    and the projector's backward applies the upstream gradient of a per-object sum as a per-slice factor in its store
    (planned pairs / singles, segment kernel pairs / singles with odd batches).  Against the two-step path (projector,
    elementwise log-likelihood, their two backward kernels): <= 1e-5 of the largest gradient (the factor multiplies
    after the sum over angles instead of before it)."""
    from ct_pvae_amd.helper_functions import gaussian_poisson_log_prob
    d = dev()
    rng = np.random.default_rng(7 * B + N)
    theta = torch.from_numpy(np.linspace(0, np.pi, A, endpoint=False).astype(np.float32)).to(d)
    P = cp.num_proj_pix(N, N)
    mask = torch.from_numpy((rng.random((B, A), dtype=np.float32) + 0.5) / A).to(d)
    meas = torch.from_numpy(rng.random((B, A, P), dtype=np.float32) * 3).to(d)
    eps = float(np.finfo(np.float32).eps)
    w = torch.from_numpy(rng.standard_normal(B).astype(np.float32)).to(d)
    up = torch.from_numpy(rng.standard_normal((B, A, P, 1)).astype(np.float32)).to(d)

    def loss(lp):
        if upstream == "per_object":
            return (lp.sum(dim=(1, 2, 3)) * w).sum()
        if upstream == "scalar":
            return lp.sum() * 0.37
        return (lp * up).sum()

    grads = []
    for fused in (True, False):
        x = torch.from_numpy(rng_img(B, N)).to(d).requires_grad_(True)
        if fused:
            lp = cp.calculate_log_prob_M_given_R(x, mask, meas, 1e3, eps, theta=theta, pad=True)
            assert "ProjectLogLik" in lp.grad_fn.name() or "RotateLogLik" in lp.grad_fn.name()   # ONE autograd node for the call
        else:
            proj = cp.project_tf_fast(x, theta, pad=True, dim=2, integrate_vae=True)
            lp = gaussian_poisson_log_prob(proj[..., 0], mask, meas, 1e3, eps).unsqueeze(-1)
        loss(lp).backward()
        grads.append(x.grad.detach())
    assert torch.isfinite(grads[0]).all()
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * float(grads[1].abs().max())


def test_random_fused_likelihood_cases():
    """Seeded random (batch, size, angles, upstream gradient kind) cases of calculate_log_prob_M_given_R's one-launch
    paths against the two-step path: log-probabilities bit for bit, gradients to 1e-5 (CTPVAE_FUZZ_SEED / _CASES: more)."""
    from ct_pvae_amd.helper_functions import gaussian_poisson_log_prob
    d = dev()
    rng = np.random.default_rng(int(os.environ.get("CTPVAE_FUZZ_SEED", 404)))
    for case in range(int(os.environ.get("CTPVAE_FUZZ_CASES", 8))):
        B, N, A = int(rng.integers(1, 91)), int(rng.integers(8, 141)), int(rng.integers(1, 31))
        if case % 4 == 3:
            B, N = int(rng.integers(1, 7)), int(rng.integers(200, 260))              # tiled forward, segment backward
        kind = ("per_object", "scalar", "elementwise")[case % 3]
        theta = torch.from_numpy(rng.uniform(0, np.pi, A).astype(np.float32)).to(d)
        P = cp.num_proj_pix(N, N)
        mask = torch.from_numpy(((rng.random((B, A)) > 0.3) * (rng.random((B, A)) + 0.2) / A).astype(np.float32)).to(d)
        meas = torch.from_numpy((rng.random((B, A, P)) * 3).astype(np.float32)).to(d)
        w = torch.from_numpy(rng.standard_normal(B).astype(np.float32)).to(d)
        up = torch.from_numpy(rng.standard_normal((B, A, P, 1)).astype(np.float32)).to(d)
        img = rng.random((B, N, N, 1), dtype=np.float32)
        res = []
        for fused in (True, False):
            x = torch.from_numpy(img).to(d).requires_grad_(True)
            if fused:
                lp = cp.calculate_log_prob_M_given_R(x, mask, meas, 1e3, 1e-7, theta=theta, pad=True)
            else:
                proj = cp.project_tf_fast(x, theta, pad=True, dim=2, integrate_vae=True)
                lp = gaussian_poisson_log_prob(proj[..., 0], mask, meas, 1e3, 1e-7).unsqueeze(-1)
            loss = {"per_object": lambda: (lp.sum(dim=(1, 2, 3)) * w).sum(), "scalar": lambda: lp.sum() * -0.2,
                    "elementwise": lambda: (lp * up).sum()}[kind]()
            loss.backward()
            res.append((lp.detach(), x.grad.detach()))
        tag = f"case {case}: B={B} N={N} A={A} {kind}"
        assert torch.equal(res[0][0], res[1][0]), tag
        assert torch.isfinite(res[0][1]).all(), tag
        assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-5 * float(res[1][1].abs().max()) + 1e-30, tag


@pytest.mark.parametrize("N,A_plan", [(128, 180), (40, 37)])
def test_angle_subsets_of_a_dense_plan(oracle, N, A_plan):
    """The training loop projects `api` random angles of the dataset's theta per step (ctvae/helper_functions.py:350-357).
    ONE dense plan + an angle-index operand gives, bit for bit, what a plan built for the gathered theta gives -- forward,
    fused likelihood (dense mask / measurements read at the selected angles) and the TF-compatible backward; any order,
    repeats, one angle, more than one ballot round, all angles; bad indices are clamped, never followed."""
    d = dev()
    rng = np.random.default_rng(N + A_plan)
    theta = rng.uniform(-0.5, 3.5, A_plan) if N != 128 else phantoms.dense_theta(A_plan)
    S = 5
    img = rng.random((S, N, N), dtype=np.float32)
    x = torch.from_numpy(img).to(d)
    plan = RotatePlan(theta, N, N, True, d)
    assert plan.planned == (True, True)
    geom = oracle.Geometry(N, N, True)
    T = oT(oracle, theta, plan)
    Tinv = oracle.invert_transforms(T)
    dense_sino = plan.forward(x)
    mask_d = torch.from_numpy(((rng.random((S, A_plan)) > 0.2) * (rng.random((S, A_plan)) + 0.2) / 20).astype(np.float32)).to(d)
    meas_d = torch.from_numpy((rng.random((S, A_plan, plan.PW)) * 3).astype(np.float32)).to(d)
    pnm = torch.tensor(1e3, device=d)
    subsets = [np.sort(rng.choice(A_plan, 20, replace=False)), rng.permutation(A_plan)[:20], np.array([A_plan - 1]),
               rng.integers(0, A_plan, 64), rng.integers(0, A_plan, 65), rng.permutation(A_plan), rng.integers(0, A_plan, 256),
               np.array([3, 3, 3, 0])]
    for sub in subsets:
        idx = cp.as_angle_index(sub, d)
        assert idx.dtype == torch.int32
        n = len(sub)
        got = plan.forward(x, angles_i=idx)
        assert tuple(got.shape) == (S, n, plan.PW)
        assert torch.equal(got, dense_sino[:, torch.from_numpy(np.asarray(sub)).to(d).long()])
        np.testing.assert_array_equal(to_np(got[[0, S - 1]]), oracle.rotate_fwd(img[[0, S - 1]], geom, T[sub], 0))
        g = rng.standard_normal((S, n, plan.PW)).astype(np.float32)
        gb = to_np(plan.backward(torch.from_numpy(g).to(d), angles_i=idx))
        np.testing.assert_array_equal(gb[[0, S - 1]], oracle.rotate_bwd_tfcompat(g[[0, S - 1]], geom, Tinv[sub], 0))
        sc = torch.linspace(-1, 2, S, device=d)
        sub_plan = RotatePlan(np.asarray(theta)[sub], N, N, True, d)
        assert torch.equal(plan.backward(torch.from_numpy(g).to(d), scale=sc, angles_i=idx),
                           sub_plan.backward(torch.from_numpy(g).to(d), scale=sc))
        # fused likelihood: dense operands read through the index == compact operands on a plan of the gathered theta
        li = torch.from_numpy(np.asarray(sub)).to(d).long()
        a = plan.forward_loglik(x, mask_d, meas_d, pnm, 1e-7, with_dlp=True, angles_i=idx, dense_inputs=True)
        b = sub_plan.forward_loglik(x, mask_d[:, li].contiguous(), meas_d[:, li].contiguous(), pnm, 1e-7, with_dlp=True)
        c = plan.forward_loglik(x, mask_d[:, li].contiguous(), meas_d[:, li].contiguous(), pnm, 1e-7, with_dlp=True, angles_i=idx)
        for u, v, w in zip(a, b, c):
            assert torch.equal(u, v) and torch.equal(u, w)
    # out-of-range indices are clamped into the plan
    bad = torch.tensor([-7, 0, A_plan + 1000], dtype=torch.int32, device=d)
    ok = torch.tensor([0, 0, A_plan - 1], dtype=torch.int32, device=d)
    assert torch.equal(plan.forward(x, angles_i=bad), plan.forward(x, angles_i=ok))
    g3 = torch.from_numpy(rng.standard_normal((S, 3, plan.PW)).astype(np.float32)).to(d)
    assert torch.equal(plan.backward(g3, angles_i=bad), plan.backward(g3, angles_i=ok))
    # more angles than one launch selects: the gathered-table fallback, same numbers
    many = rng.integers(0, A_plan, 300)
    got = plan.forward(x[:2], angles_i=cp.as_angle_index(many, d))
    np.testing.assert_array_equal(to_np(got), oracle.rotate_fwd(img[:2], geom, T[many], 0))


def test_training_call_builds_one_plan_and_matches_the_two_step_path(oracle, monkeypatch):
    """calculate_log_prob_M_given_R(theta=<the dataset's angles>, angles_i=<this step's subset>) over several steps:
    ONE RotatePlan is constructed (no table or plan kernel after the first step), and every step's log-probabilities and
    image gradients equal the reference's sequence -- gather theta / mask / proj_sample, project, log_prob -- to the bit
    (gradients to 1e-5: the per-object factor multiplies after the sum over angles)."""
    from ct_pvae_amd import forward_functions as ff
    from ct_pvae_amd.helper_functions import gaussian_poisson_log_prob
    d = dev()
    built = []
    real_init = ff.RotatePlan.__init__

    def counting_init(self, *a, **k):
        built.append(1)
        return real_init(self, *a, **k)

    rng = np.random.default_rng(5)
    theta = phantoms.dense_theta(180) + 1e-4          # an angle set no other test has cached
    B, N, P = 5, 128, 184
    mask = torch.from_numpy(((rng.random((B, 180)) > 0.5) / 20).astype(np.float32)).to(d)
    meas = torch.from_numpy((rng.random((B, 180, P)) * 3).astype(np.float32)).to(d)
    geom = oracle.Geometry(N, N, True)
    T = oracle.rotate_transforms(theta.astype(np.float32), P, P)
    monkeypatch.setattr(ff.RotatePlan, "__init__", counting_init)
    for step in range(4):
        sub = rng.permutation(180)[:20]
        img = rng.random((B, N, N, 1), dtype=np.float32)
        x = torch.from_numpy(img).to(d).requires_grad_(True)
        lp = cp.calculate_log_prob_M_given_R(x, mask, meas, 1e4, 1e-7, theta=theta, angles_i=sub, pad=True)
        assert tuple(lp.shape) == (B, 20, P, 1)
        lp.sum(dim=(1, 2, 3)).mul(torch.arange(1, B + 1, device=d)).sum().backward()
        # the reference's sequence, on the oracle's sinogram
        sino = oracle.rotate_fwd(img[..., 0], geom, T[sub], 0)
        want = oracle.loglik(sino, to_np(mask)[:, sub], to_np(meas)[:, sub], 1e4, 1e-7)
        assert rel_err(to_np(lp)[..., 0], want) <= 1e-5
        x2 = torch.from_numpy(img).to(d).requires_grad_(True)
        li = torch.from_numpy(sub).to(d)
        proj = cp.project_tf_fast(x2, theta[sub], pad=True, dim=2, integrate_vae=True)
        assert torch.equal(proj[..., 0].detach(), torch.from_numpy(sino).to(d))
        lp2 = gaussian_poisson_log_prob(proj[..., 0], mask[:, li], meas[:, li], 1e4, 1e-7)
        assert torch.equal(lp2.detach(), lp[..., 0].detach())
        lp2.sum(dim=(1, 2)).mul(torch.arange(1, B + 1, device=d)).sum().backward()
        assert float((x.grad - x2.grad).abs().max()) <= 1e-5 * float(x2.grad.abs().max())
    # 1 dense plan for the training call + 4 gathered-theta plans of the two-step comparison
    assert len(built) == 5, built


def test_config5_full_size_likelihood_and_adjoint(oracle):
    """BASELINE config 5 at its full configuration: 512 x 512 objects (P = 728), 90 angles, the Poisson-noise forward
    model at pnm = 1e4, B = 8 -- the tiled forward (two groups of four interleaved slices) with the log-likelihood in its
    reduce pass, and the one-launch backward (paired segment kernel with the per-object factor):
      * ray-sums of two objects bit-exact vs oracle.rotate_fwd_tiled (96 x 64 tiles) on the ORACLE's tables;
      * log-probabilities vs oracle.loglik on those ray-sums <= 1e-5 of the largest (only logf differs: <= 2 ulp);
      * d lp / d ray-sum vs the float64 derivative of the same expression <= 1e-5;
      * the image gradient = w_b * oracle.rotate_bwd_tfcompat(dlp_b), bit for bit (one fp32 multiply in the store);
      * the same numbers through calculate_log_prob_M_given_R(...).backward()."""
    d = dev()
    rng = np.random.default_rng(55)
    B, N, A, nsa, pnm_v = 8, 512, 90, 20, 1e4
    eps = float(np.finfo(np.float32).eps)
    theta = np.pi * np.arange(A) / A
    img = phantoms.foam_batch(B, N, seed=5, supersample=1)
    x = torch.from_numpy(img).to(d)
    plan = RotatePlan(theta, N, N, True, d)
    assert plan.tiled and plan.PW == 728 and plan.supports_scale
    # dose masks 1/nsa on nsa random angles per object, Poisson(proj * mask * pnm) / pnm (ctvae/create_masks.py:45-63,:94-95)
    mask = np.zeros((B, A), np.float32)
    for b in range(B):
        mask[b, rng.choice(A, nsa, replace=False)] = 1.0 / nsa
    clean = to_np(plan.forward(x))
    meas = (rng.poisson(np.maximum(clean, 0) * mask[..., None] * pnm_v) / pnm_v).astype(np.float32)
    mask_t, meas_t = torch.from_numpy(mask).to(d), torch.from_numpy(meas).to(d)
    pnm = torch.tensor(pnm_v, dtype=torch.float32, device=d)
    sino, lp, dlp = plan.forward_loglik(x, mask_t, meas_t, pnm, eps, with_dlp=True)
    geom = oracle.Geometry(N, N, True)
    T = oT(oracle, theta, plan)
    pick = [1, 6]                                                   # one object of each four-slice group
    want_sino = oracle.rotate_fwd_tiled(img[pick], geom, T, oracle.tile_shape(geom.H, geom.W))
    np.testing.assert_array_equal(to_np(sino)[pick], want_sino)
    assert torch.equal(sino, plan.forward(x))                       # the epilogue does not change the ray-sums
    want_lp = oracle.loglik(want_sino, mask[pick], meas[pick], pnm_v, eps)
    assert rel_err(to_np(lp)[pick], want_lp) <= REL
    # d lp / d sino in float64 from the oracle's ray-sums
    p64 = torch.from_numpy(want_sino.astype(np.float64)).requires_grad_(True)
    loc = p64 * torch.from_numpy(mask[pick].astype(np.float64))[..., None]
    scale = eps + torch.sqrt(loc / pnm_v + eps)
    torch.distributions.Normal(loc, scale).log_prob(torch.from_numpy(meas[pick].astype(np.float64))).sum().backward()
    assert rel_err(to_np(dlp)[pick], p64.grad.numpy()) <= REL
    # backward: per-object factor in the store
    w = rng.standard_normal(B).astype(np.float32)
    gimg = to_np(plan.backward(dlp, scale=torch.from_numpy(w).to(d)))
    want_g = oracle.rotate_bwd_tfcompat(to_np(dlp)[pick], geom, oracle.invert_transforms(T), 0)
    np.testing.assert_array_equal(gimg[pick], w[pick, None, None] * want_g)
    # the public caller, autograd end to end
    xa = torch.from_numpy(img[..., None]).to(d).requires_grad_(True)
    lpa = cp.calculate_log_prob_M_given_R(xa, mask_t, meas_t, pnm_v, eps, theta=theta, pad=True)
    assert torch.equal(lpa[..., 0].detach(), lp)
    (lpa.sum(dim=(1, 2, 3)) * torch.from_numpy(w).to(d)).sum().backward()
    assert torch.equal(xa.grad[..., 0], torch.from_numpy(gimg).to(d))


def test_hip_against_the_independent_resampler_fixture(golden_dir):
    """The HIP projectors against tests/golden/gridsample_crosscheck.npz -- PyTorch's CPU grid_sample computing the same
    rotate-and-sum, a second implementation that owes nothing to oracle/radon_oracle.c (tests/test_oracle.py holds the
    oracle's side of the same comparison): bilinear within 1e-5 everywhere; nearest bit-equal on all but the 18 ray-sums
    (0.05 %) whose coordinates sit on a rounding tie; the exact adjoint within 5e-5."""
    d = dev()
    z = np.load(os.path.join(golden_dir, "gridsample_crosscheck.npz"))
    img = phantoms.foam_batch(1, 128, seed=int(z["seed"]), supersample=2)
    x = torch.from_numpy(img).to(d)
    theta = z["theta"]
    bil = to_np(RotatePlan(theta, 128, 128, True, d, interp="bilinear").forward(x))
    assert rel_err(bil, z["fwd_bilinear"]) <= REL
    for use_plan in (True, False):
        near = to_np(RotatePlan(theta, 128, 128, True, d, use_plan=use_plan).forward(x))
        assert int((near != z["fwd_nearest"]).sum()) == int(z["nearest_differing_ray_sums"]) == 18
        assert rel_err(near, z["fwd_nearest"]) <= 5e-3
    g = torch.from_numpy(np.random.default_rng(int(z["g_seed"])).standard_normal((1, 180, 184)).astype(np.float32)).to(d)
    grad = to_np(RotatePlan(theta, 128, 128, True, d, interp="bilinear", backward="exact").backward(g))
    assert rel_err(grad, z["grad_bilinear"]) <= 5e-5


def test_poisson_sampler_kernel_equals_its_cpu_twin(oracle, tmp_path):
    """Poisson(max(sino, 0) * mask * pnm) / pnm as one HIP kernel (ctvae/create_masks.py:80-103).  The sampler
    is counter-based and fully specified, so the CPU twin reproduces EVERY count -- rates from 0 through the algorithm
    switch at 10 to 1e6, negative inputs, zero masks; the launch shape does not matter (grid-stride over elements)."""
    from ct_pvae_amd.create_masks import create_all_masks, poisson_measure
    d = dev()
    rng = np.random.default_rng(12)
    S, A, P = 7, 12, 184
    rates = np.concatenate([[0.0, 1e-3, 0.5, 9.99, 10.0, 10.01, 1e2, 1e4, 6e4, 1e6], 10 ** rng.uniform(-3, 6, 200)])
    sino = rng.choice(rates, size=(S, A, P)).astype(np.float32)
    sino[0, 0, :20] = -rng.random(20).astype(np.float32)                     # negatives are clamped to zero first
    mask = (rng.random((S, A)) > 0.3).astype(np.float32) * rng.choice([1.0, 0.5, 1 / 20], size=(S, A)).astype(np.float32)
    for pnm, seed in ((1.0, 0), (1e4, 2 ** 40 + 17)):
        got = to_np(poisson_measure(torch.from_numpy(sino).to(d), torch.from_numpy(mask).to(d), pnm, seed))
        np.testing.assert_array_equal(got, oracle.poisson_measure(sino, mask, pnm, seed))
    # distribution on the device itself: 400k draws at one rate per decade
    for lam in (0.3, 4.0, 25.0, 1e3, 5e4):
        x = torch.full((1, 1, 400000), lam, device=d)
        k = to_np(poisson_measure(x, torch.ones((1, 1), device=d), 1.0, 5))[0, 0].astype(np.float64)
        assert abs(k.mean() - lam) <= 5 * np.sqrt(lam / k.size) and abs(k.var() / lam - 1) <= 0.02
    # through create_all_masks: files written, measured angles only, counts are integers / pnm
    sino_t = torch.from_numpy(np.abs(sino)).to(d)
    masks, samples = create_all_masks(sino_t, A, save_path=str(tmp_path), poisson_noise_multiplier=1e3, num_sparse_angles=4,
                                      random=True, train=True, truncate_dataset=5)
    assert tuple(masks.shape) == (5, A) and tuple(samples.shape) == (5, A, P) and samples.device == sino_t.device
    np.testing.assert_array_equal(to_np(samples), oracle.poisson_measure(np.abs(sino[:5]), to_np(masks), 1e3, 0))
    assert float(samples[masks == 0].abs().max()) == 0.0
    m2, s2 = create_all_masks(None, A, save_path=str(tmp_path), train=False, device=d)
    assert torch.equal(m2, masks) and torch.equal(s2, samples)


@pytest.mark.parametrize("shape,A,S,pad", [((128, 128), 180, 4, True), ((184, 184), 60, 3, False), ((40, 57), 23, 1, True),
                                          ((2, 2), 2, 2, False), ((300, 300), 12, 2, True), ((64, 64), 16, 600, True),
                                          ((33, 47), 8, 3, True), ((33, 33), 4, 2, True), ((5, 3), 6, 2, True)])
def test_siddon_backprojector_is_the_transpose(oracle, shape, A, S, pad):
    """The back-projector (libtomo fbp.c's accumulation = the transpose of project.c), pixel-driven since round 3: a lane asks
    the two rays per angle that can cross its pixel for their segment in it, with libtomo's own fp32 expressions and in
    libtomo's order -- BIT-EQUAL to the oracle's ray-driven accumulation on even grids (random data of both signs, random
    angles, theta = 0 and pi / 2 exactly, whole angles unmeasured, trimmed crossings along the outline, corner-cutting slivers
    whose midpoint rounds into a neighbour); <A x, y> = <x, A^T y> with the GPU forward; bit-reproducible from run to run.  ODD
    grids under the even padded detector at theta = 0 / pi / 2 put rays ON grid lines, where libtomo's merge zig-zags: those
    rays are walked as libtomo walks them by a second kernel and added first -- another order, <= 1e-5."""
    from ct_pvae_amd.recon import siddon_backproject
    d = dev()
    rng = np.random.default_rng(shape[0] + A)
    theta = rng.uniform(-1.0, 7.0, A)
    theta[: min(A, 2)] = [0.0, np.pi / 2][: min(A, 2)]
    img = rng.random((S,) + shape, dtype=np.float32)
    sino = cp.create_sinograms(torch.from_numpy(img).to(d), theta, pad=pad)        # [S][A][dx]
    dx = sino.shape[2]
    y = rng.standard_normal((S, A, dx)).astype(np.float32)
    y[:, ::3] = 0.0                                                                 # whole angles unmeasured
    yt = torch.from_numpy(y).to(d)
    # the transpose lives on the OBJECT grid here (center = dx / 2 either way)
    got = siddon_backproject(yt, theta, shape[0], shape[1])
    again = siddon_backproject(yt, theta, shape[0], shape[1])
    assert torch.equal(got, again)
    n_chk = min(S, 3)
    want = np.zeros((n_chk,) + shape, np.float32)
    oracle.lib().oracle_siddon_backproject(np.ascontiguousarray(y[:n_chk]), n_chk, A, dx, theta.astype(np.float32), dx / 2.0,
                                           shape[0], shape[1], want)
    if shape[0] % 2 == 0 and shape[1] % 2 == 0:
        np.testing.assert_array_equal(to_np(got)[:n_chk], want)
    else:
        assert rel_err(to_np(got)[:n_chk], want) <= REL
    lhs = float((to_np(sino).astype(np.float64) * y).sum())
    rhs = float((img.astype(np.float64) * to_np(got)).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0), (lhs, rhs)


def test_sirt_and_the_encoder_channels(oracle, tmp_path):
    """tomopy.recon's algorithms behind iradon_all (ctvae/helper_functions.py:477-529): 'sirt' against the oracle's
    restatement of libtomo's sirt.c (1 iteration = tomopy's default, and 5), 'fbp' / filter 'none' against fbp.c's, the
    gridrec reconstructing the phantom in the same orientation, and iradon_all stacking [algorithms..., mask channel] with the
    reference's crop.  (Round 3: a SIRT iteration is two launches -- the forward's store is the ray's update factor, the
    pixel-driven back-projector's store is the update -- and equals the oracle's sirt.c restatement BIT FOR BIT at 1 iteration.)"""
    from ct_pvae_amd.recon import crop, recon
    d = dev()
    N, A = 64, 45
    theta = np.pi * np.arange(A) / A
    img = phantoms.foam_batch(3, N, seed=4, supersample=2)
    sino = cp.create_sinograms(torch.from_numpy(img).to(d), theta, pad=True)        # [3][A][94]
    P = sino.shape[2]
    host = to_np(sino)
    for it in (1, 5):
        got = to_np(recon(sino, theta, sinogram_order=True, algorithm="sirt", num_iter=it))
        want = oracle.sirt(host, theta, num_iter=it)
        assert got.shape == want.shape == (3, P, P)
        assert rel_err(got, want) <= (1e-5 if it == 1 else 5e-5), it
        if it == 1:
            np.testing.assert_array_equal(got, want)
    bp = to_np(recon(sino, theta, sinogram_order=True, algorithm="fbp", filter_name="none"))
    assert rel_err(bp, oracle.siddon_backproject(host, theta)) <= REL
    # projection order (sinogram_order=False: [angles][slices][dx]) is the same reconstruction
    assert torch.equal(recon(sino.permute(1, 0, 2), theta, algorithm="fbp"), torch.from_numpy(bp).to(d))
    # many SIRT iterations and the gridrec stand-in both reconstruct the phantom, in the same orientation
    truth = img
    s100 = to_np(crop(recon(sino, theta, sinogram_order=True, algorithm="sirt", num_iter=100), N, N, ignore_dim_0=True))
    grid = to_np(crop(recon(sino, theta, sinogram_order=True, algorithm="gridrec"), N, N, ignore_dim_0=True))
    base = float((truth ** 2).mean())
    assert float(((s100 - truth) ** 2).mean()) < 0.1 * base and float(((grid - truth) ** 2).mean()) < 0.1 * base
    assert float(((grid - truth[:, ::-1]) ** 2).mean()) > 2 * float(((grid - truth) ** 2).mean())     # a flip would show
    # the TV stand-in (flagged: preconditioned Chambolle-Pock on the same operator pair, not tomopy's tv.c): converges to a
    # reconstruction at least as good as SIRT's on noisy data, and its default single iteration is finite
    noisy = sino + 0.5 * torch.randn(sino.shape, device=d, generator=torch.Generator(device=d).manual_seed(0))
    tv = to_np(crop(recon(noisy, theta, sinogram_order=True, algorithm="tv", num_iter=300, reg_par=[0.05]), N, N, ignore_dim_0=True))
    s_noisy = to_np(crop(recon(noisy, theta, sinogram_order=True, algorithm="sirt", num_iter=300), N, N, ignore_dim_0=True))
    assert float(((tv - truth) ** 2).mean()) < 0.1 * base
    assert float(((tv - truth) ** 2).mean()) <= 1.1 * float(((s_noisy - truth) ** 2).mean())
    assert torch.isfinite(recon(sino, theta, sinogram_order=True, algorithm="tv")).all()
    with pytest.raises(ValueError):
        recon(sino, theta, sinogram_order=True, algorithm="art")
    # iradon_all: README.md:221's algorithm list
    masks = torch.zeros((3, A), device=d)
    masks[:, ::5] = 1.0 / 9
    samples = sino * masks[..., None]
    enc = cp.iradon_all(samples, masks, P, theta, ["sirt", "tv", "fbp", "gridrec"], 1e-7, N, N, save_path=str(tmp_path), train=True)
    assert tuple(enc.shape) == (3, N, N, 5)
    used = masks[0] > 0
    sparse = (samples / masks[..., None].clamp_min(1e-30))[:, used]
    want_sirt = crop(recon(sparse.contiguous(), theta[to_np(used)], sinogram_order=True, algorithm="sirt"), N, N, ignore_dim_0=True)
    # (unmeasured angles are zero rows of the dose-normalised sinogram: they still constrain SIRT, so the dense call differs
    # from reconstructing the measured angles alone -- the reference makes the dense call, :489-503)
    assert not torch.allclose(enc[..., 0], want_sirt)
    dense = torch.where(masks[..., None].expand(-1, -1, P) > 1e-7, samples / masks[..., None].clamp_min(1e-30), samples)
    assert torch.equal(enc[..., 0], crop(recon(dense.contiguous(), theta, sinogram_order=True, algorithm="sirt"), N, N, ignore_dim_0=True))
    mask_chan = oracle.siddon_backproject(to_np(masks[..., None].expand(-1, -1, P).contiguous()), theta)
    lo = P // 2 - N // 2
    assert rel_err(to_np(enc[..., 4]), mask_chan[:, lo:lo + N, lo:lo + N]) <= REL
    again = cp.iradon_all(None, masks, P, theta, ["sirt", "tv", "fbp", "gridrec"], 1e-7, N, N, save_path=str(tmp_path), train=False)
    assert torch.equal(again.to(d), enc)


@pytest.mark.parametrize("shape,pad,A,S", [((128, 128), True, 20, 50), ((128, 128), True, 180, 3), ((40, 100), True, 33, 3),
                                          ((65, 31), False, 9, 1), ((2, 2), False, 2, 2), ((130, 70), True, 70, 17)])
def test_exact_transpose_is_a_deterministic_gather(oracle, shape, pad, A, S):
    """backward='exact' (nearest): the inverse plan (<= 2 hitting bins per angle and pixel, in the scatter's order) makes the
    true transpose a gather -- BIT-EQUAL to the oracle's in-order scatter (oracle_rotate_bwd_exact), identical from run to
    run, single slices and pairs, and still the adjoint of the forward."""
    d = dev()
    rng = np.random.default_rng(A * 11 + S)
    theta = rng.uniform(-1.0, 4.0, A)
    theta[: min(A, 3)] = [0.0, np.pi / 2, np.pi / 4][: min(A, 3)]
    plan = RotatePlan(theta, shape[0], shape[1], pad, d, backward="exact")
    g = rng.standard_normal((S, A, plan.PW)).astype(np.float32)
    gt = torch.from_numpy(g).to(d)
    got = plan.backward(gt)
    assert plan._exact_plan is not None                      # the planned gather ran, not the atomic scatter
    geom = oracle.Geometry(shape[0], shape[1], pad)
    n = min(S, 3)
    np.testing.assert_array_equal(to_np(got)[:n], oracle.rotate_bwd_exact(g[:n], geom, oT(oracle, theta, plan), 0))
    assert torch.equal(got, plan.backward(gt))
    _lib.tune("BNS", 1)
    one = plan.backward(gt)
    _lib.tune("BNS", 2)
    two = plan.backward(gt)
    assert torch.equal(one, got) and torch.equal(two, got)
    # the scatter kernel (atomics) agrees to rounding, and <A x, y> = <x, A^T y>
    scatter = RotatePlan(theta, shape[0], shape[1], pad, d, backward="exact", use_plan=False).backward(gt)
    assert rel_err(to_np(scatter), to_np(got)) <= REL
    x = rng.standard_normal((S,) + shape).astype(np.float32)
    lhs = float((to_np(plan.forward(torch.from_numpy(x).to(d))).astype(np.float64) * g).sum())
    rhs = float((x.astype(np.float64) * to_np(got)).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))


def test_angle_subsets_on_geometries_without_an_index_operand(oracle):
    """angles_i on paths whose kernels take no angle-index operand -- a slice larger than LDS (tiled forward), bilinear
    interpolation, the exact backward -- falls back to gathered table rows (RotatePlan.subset): same numbers as a plan built
    for the gathered theta, through the raw operators and through calculate_log_prob_M_given_R."""
    from ct_pvae_amd.helper_functions import gaussian_poisson_log_prob
    d = dev()
    rng = np.random.default_rng(9)
    N, A = 230, 24
    theta = rng.uniform(0, np.pi, A)
    sub = rng.permutation(A)[:7]
    idx = cp.as_angle_index(sub, d)
    img = rng.random((3, N, N), dtype=np.float32)
    x = torch.from_numpy(img).to(d)
    geom = oracle.Geometry(N, N, True)
    big = RotatePlan(theta, N, N, True, d)
    assert big.tiled and not big.sel_supported(7)
    T = oT(oracle, theta, big)
    np.testing.assert_array_equal(to_np(big.forward(x, angles_i=idx)), oracle.rotate_fwd_tiled(img, geom, T[sub], oracle.tile_shape(geom.H, geom.W)))
    g = rng.standard_normal((3, 7, big.PW)).astype(np.float32)
    np.testing.assert_array_equal(to_np(big.backward(torch.from_numpy(g).to(d), angles_i=idx)),
                                  oracle.rotate_bwd_tfcompat(g, geom, oracle.invert_transforms(T)[sub], 0))
    small = rng.random((2, 40, 40), dtype=np.float32)
    geo40 = oracle.Geometry(40, 40, True)
    for kw, ref_b in ((dict(interp="bilinear"), lambda gg, TT: oracle.rotate_bwd_tfcompat(gg, geo40, oracle.invert_transforms(TT), 1)),
                      (dict(backward="exact"), lambda gg, TT: oracle.rotate_bwd_exact(gg, geo40, TT, 0))):
        plan = RotatePlan(theta, 40, 40, True, d, **kw)
        T40 = oT(oracle, theta, plan)
        code = 1 if "interp" in kw else 0
        got = to_np(plan.forward(torch.from_numpy(small).to(d), angles_i=idx))
        if code:
            np.testing.assert_array_equal(got, oracle.rotate_fwd(small, geo40, T40[sub], code))
        g40 = rng.standard_normal((2, 7, plan.PW)).astype(np.float32)
        gb = to_np(plan.backward(torch.from_numpy(g40).to(d), angles_i=idx))
        assert rel_err(gb, ref_b(g40, T40[sub])) <= REL
    # the training call on the tiled geometry
    mask = torch.from_numpy(((rng.random((3, A)) > 0.3) / 7).astype(np.float32)).to(d)
    meas = torch.from_numpy((rng.random((3, A, big.PW)) * 3).astype(np.float32)).to(d)
    xa = torch.from_numpy(img[..., None]).to(d).requires_grad_(True)
    lp = cp.calculate_log_prob_M_given_R(xa, mask, meas, 1e3, 1e-7, theta=theta, angles_i=sub, pad=True)
    lp.sum().backward()
    xb = torch.from_numpy(img[..., None]).to(d).requires_grad_(True)
    li = torch.from_numpy(sub).to(d)
    proj = cp.project_tf_fast(xb, theta[sub], pad=True, dim=2, integrate_vae=True)
    lp2 = gaussian_poisson_log_prob(proj[..., 0], mask[:, li], meas[:, li], 1e3, 1e-7)
    lp2.sum().backward()
    assert torch.equal(lp[..., 0].detach(), lp2.detach())
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * float(xb.grad.abs().max())


def test_round2_setup_path_against_golden(golden_dir):
    """The HIP set-up-path kernels against tests/golden/round2_setup_path.npz directly: Poisson counts bit-exact, back-
    projection and SIRT to 1e-5 / 5e-5 of the largest value."""
    from ct_pvae_amd.create_masks import poisson_measure
    from ct_pvae_amd.recon import recon, siddon_backproject
    d = dev()
    z = np.load(os.path.join(golden_dir, "round2_setup_path.npz"))
    got = poisson_measure(torch.from_numpy(z["p_sino"]).to(d), torch.from_numpy(z["p_mask"]).to(d), float(z["p_pnm"]),
                          int(z["p_seed"]))
    np.testing.assert_array_equal(to_np(got), z["p_out"])
    data, theta = torch.from_numpy(z["r_data"]).to(d), z["r_theta"]
    np.testing.assert_array_equal(to_np(cp.create_sinograms(torch.from_numpy(z["r_img"]).to(d), theta, pad=True)), z["r_data"])
    assert rel_err(to_np(siddon_backproject(data, theta)), z["r_backproject"]) <= REL
    assert rel_err(to_np(siddon_backproject(data, theta, 24, 24)), z["r_backproject_obj"]) <= REL
    assert rel_err(to_np(recon(data, theta, sinogram_order=True, algorithm="sirt")), z["r_sirt1"]) <= REL
    assert rel_err(to_np(recon(data, theta, sinogram_order=True, algorithm="sirt", num_iter=7)), z["r_sirt7"]) <= 5e-5


def test_backward_scale_operand_checks():
    d = dev()
    theta = np.linspace(0, np.pi, 6, endpoint=False)
    plan = cp.RotatePlan(theta, 32, 32, True, d)
    g = torch.randn(4, 6, plan.PW, device=d)
    base = plan.backward(g)
    sc = torch.tensor([2.0, -1.0, 0.5, 0.0], device=d)
    assert torch.equal(plan.backward(g, scale=sc), base * sc.view(4, 1, 1))
    assert torch.equal(plan.backward(g, scale=torch.tensor(3.0, device=d).expand(4)), base * 3.0)
    with pytest.raises(ValueError):
        plan.backward(g, scale=sc[:3])
    with pytest.raises(ValueError):
        plan.backward(g, scale=sc.double())
    bil = cp.RotatePlan(theta, 32, 32, True, d, interp="bilinear")
    with pytest.raises(ValueError):
        bil.backward(g, scale=sc)


def rng_img(B, N):
    return np.random.default_rng(1234).random((B, N, N, 1), dtype=np.float32)


@pytest.mark.parametrize("use_plan", [True, False])
def test_more_slices_than_a_grid_dimension(oracle, use_plan):
    """70,000 tiny slices: past the 65,535 limit of a grid's y / z dimension, which the direct kernels index slices
    with (they fall back to their generic forms), and a big 1024 x 1024 slice (P = 1452: tiled forward, segment
    backward, no byte-sized backward plan) -- the two ends of the size range, bit-exact against the oracle."""
    d = dev()
    rng = np.random.default_rng(3)
    S, N, A = 70000, 8, 3
    theta = np.array([0.3, 1.1, 2.5])
    img = rng.standard_normal((S, N, N)).astype(np.float32)
    plan = RotatePlan(theta, N, N, True, d, use_plan=use_plan)
    geom = oracle.Geometry(N, N, True)
    got = to_np(plan.forward(torch.from_numpy(img).to(d)))
    np.testing.assert_array_equal(got, oracle.rotate_fwd(img, geom, oT(oracle, theta, plan), 0))
    g = rng.standard_normal(got.shape).astype(np.float32)
    np.testing.assert_array_equal(to_np(plan.backward(torch.from_numpy(g).to(d))),
                                  oracle.rotate_bwd_tfcompat(g, geom, oTinv(oracle, theta, plan), 0))
    if use_plan:
        big = rng.standard_normal((1, 1024, 1024)).astype(np.float32)
        plan = RotatePlan(theta, 1024, 1024, True, d)
        geom = oracle.Geometry(1024, 1024, True)
        assert plan.tiled and geom.PW == 1452
        np.testing.assert_array_equal(to_np(plan.forward(torch.from_numpy(big).to(d))),
                                      oracle.rotate_fwd_tiled(big, geom, oT(oracle, theta, plan), oracle.tile_shape(geom.H, geom.W)))
        gb = rng.standard_normal((1, 3, 1452)).astype(np.float32)
        np.testing.assert_array_equal(to_np(plan.backward(torch.from_numpy(gb).to(d))),
                                      oracle.rotate_bwd_tfcompat(gb, geom, oTinv(oracle, theta, plan), 0))


def test_long_batches_are_launched_in_chunks():
    """Entry points whose kernels index slices with a grid dimension split a long batch into launches of at most
    65,535 slices.  With the limit lowered to 5 (CTPVAE_TUNE_MAX_SLICES) every such path -- direct and tiled forward
    (with the likelihood epilogue's per-slice operands), segment / bilinear / exact backward (with the per-slice
    factor), the TomoPy-style projector and the FBP -- must give, bit for bit, what one launch gives."""
    from ct_pvae_amd import fbp
    d = dev()
    rng = np.random.default_rng(11)
    S, A = 13, 4
    theta = rng.uniform(0, np.pi, A)

    def run_all():
        out = []
        for (H, W), kw in (((40, 60), dict(use_plan=False)), ((220, 190), {}), ((40, 60), dict(interp="bilinear")),
                           ((24, 24), dict(backward="exact"))):
            plan = RotatePlan(theta, H, W, True, d, **kw)
            x = torch.from_numpy(np.random.default_rng(H).standard_normal((S, H, W)).astype(np.float32)).to(d)
            g = torch.from_numpy(np.random.default_rng(W).standard_normal((S, A, plan.PW)).astype(np.float32)).to(d)
            out += [plan.forward(x), plan.backward(g)]
            if plan.supports_scale:
                sc = torch.linspace(-1, 2, S, device=d)
                out.append(plan.backward(g, scale=sc))
            if plan.tiled:
                mask = torch.rand((S, A), device=d) + 0.5
                meas = torch.rand((S, A, plan.PW), device=d) * 3
                out += list(plan.forward_loglik(x.abs(), mask, meas, torch.tensor(1e3, device=d), 1e-7, with_dlp=True))
        imgs = torch.rand((S, 20, 28), device=d)
        out.append(cp.create_sinograms(imgs, theta))
        sino = torch.rand((S, A, 30), device=d, dtype=torch.float64)
        out.append(fbp.iradon(sino, theta, 20, 20, fbp.ramp_filter(30)))
        torch.cuda.synchronize()
        return out

    torch.manual_seed(0)
    whole = run_all()
    with _lib.tuned("MAX_SLICES", 5):
        torch.manual_seed(0)
        chunked = run_all()
    assert len(whole) == len(chunked) >= 14
    for k, (a, b) in enumerate(zip(whole, chunked)):
        if k == 12:      # the exact-transpose backward adds with float atomics: equal up to the order of the adds
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)
        else:
            assert torch.equal(a, b), k


def test_empty_batch_is_an_empty_result():
    """A batch of zero objects projects to zero sinograms (what the TensorFlow op chain gives), forward and backward,
    through the raw operator, the public function and the likelihood caller -- no launch, no error."""
    d = dev()
    theta = np.linspace(0, np.pi, 6, endpoint=False)
    plan = RotatePlan(theta, 32, 32, True, d)
    assert tuple(plan.forward(torch.empty((0, 32, 32), device=d)).shape) == (0, 6, plan.PW)
    assert tuple(plan.backward(torch.empty((0, 6, plan.PW), device=d)).shape) == (0, 32, 32)
    x = torch.empty((0, 32, 32, 1), device=d, requires_grad=True)
    out = cp.project_tf_fast(x, theta, pad=True, dim=2, integrate_vae=True)
    assert tuple(out.shape) == (0, 6, plan.PW, 1)
    out.sum().backward()
    assert tuple(x.grad.shape) == (0, 32, 32, 1)
    lp = cp.calculate_log_prob_M_given_R(x, torch.empty((0, 6), device=d), torch.empty((0, 6, plan.PW), device=d), 1e3, 1e-7,
                                         theta=theta, pad=True)
    assert tuple(lp.shape) == (0, 6, plan.PW, 1)


def test_bad_shapes_raise():
    d = dev()
    with pytest.raises(ValueError):
        cp.project_tf_fast(torch.zeros(2, 8, 8, 3, device=d), np.array([0.0]), integrate_vae=True)
    with pytest.raises(ValueError):
        cp.project_tf_fast(torch.zeros(8, 8, device=d), np.array([0.0]), dim=3)
    with pytest.raises(ValueError):
        cp.project_tf_fast(torch.zeros(8, 8, device=d), np.zeros((2, 2)), dim=2)
    with pytest.raises(ValueError):
        cp.project_tf_fast(torch.zeros(8, 8, device=d), np.zeros(0), dim=2)


def test_bench_under_torchrun_two_ranks():
    """bench.py launched exactly as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), rehearsed
    on this box's single GPU: both ranks bind device 0 and the group runs over gloo.  One JSON line from rank 0, whole-job
    value, the contract's keys."""
    import json
    import socket
    import subprocess
    import sys
    from tests.conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CTPVAE_REHEARSE_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30",
           "--warmup", "5", "--grad-allreduce"]      # config 4's collective rides along (2.8 MB bucket per step)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["grad_allreduce_bytes_per_step"] == 4 * 711164
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d
    assert d["n_gpus"] == 2 and d["steps"] == 30 and d["scaling"] == "weak" and d["vs_baseline"] is None
    # whole-job value: 2 ranks x 50 objects x 20 angles per step
    assert abs(d["value"] - 2 * 50 * 20 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torch.distributed.run around it (how the driver runs N = 1): the parent starts the
    ranks itself, as children, before it touches the GPU; on this one-GPU box they rehearse on device 0 over gloo.  rc 0,
    one JSON line from rank 0, n_gpus = 2, whole-job value."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                          "--min-ms", "5"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["repeats"] >= 1 and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 50 * 20 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # world > 1: the projector-only value AND the same steps with the data-parallel trainer's one collective, both reported
    assert d["value_with_grad_allreduce"] > 0 and d["allreduce_us"] > 0 and d["rccl_ranks"] == 2
    assert d["value_with_grad_allreduce"] <= d["value"] * 1.05 and "cache-warm" in d["config"]["inputs"]


def test_random_geometries_against_the_oracle(oracle):
    """Seeded random small geometries (odd sizes, 1-pixel slices, unpadded canvases, 1..40 angles, any angle values):
    planned or direct, whichever the geometry takes, forward and tf_compat backward are bit-exact against the oracle; the
    exact backward stays the transpose."""
    d = dev()
    # CTPVAE_FUZZ_SEED / CTPVAE_FUZZ_CASES: a longer run over other seeds (then with batches large enough for the
    # slice-pair kernels, S >= 16 / 32), e.g. CTPVAE_FUZZ_SEED=5 CTPVAE_FUZZ_CASES=200 pytest -m gpu -k random
    fuzz = "CTPVAE_FUZZ_SEED" in os.environ
    rng = np.random.default_rng(int(os.environ.get("CTPVAE_FUZZ_SEED", 20261004)))
    for case in range(int(os.environ.get("CTPVAE_FUZZ_CASES", 24))):
        H, W = int(rng.integers(1, 150)), int(rng.integers(1, 150))
        if case < 4:
            H, W = [(1, 1), (1, 77), (93, 1), (3, 2)][case]
        pad, A, S = bool(rng.integers(0, 2)), int(rng.integers(1, 41)), int(rng.integers(1, 7))
        if fuzz and case % 2:
            S = int(rng.integers(16, 90))
        theta = rng.uniform(-2 * np.pi, 2 * np.pi, A)
        if case % 3 == 0:
            theta[: min(A, 4)] = [0.0, np.pi / 2, np.pi, -np.pi / 2][: min(A, 4)]      # exact ties in the rounding
        img = rng.standard_normal((S, H, W)).astype(np.float32)
        geom = oracle.Geometry(H, W, pad)
        for use_plan in (True, False):
            plan = RotatePlan(theta, H, W, pad, d, use_plan=use_plan)
            got = to_np(plan.forward(torch.from_numpy(img).to(d)))
            want = oracle.rotate_fwd(img, geom, oT(oracle, theta, plan), 0)
            np.testing.assert_array_equal(got, want, err_msg=f"fwd case {case}: {H}x{W} pad={pad} A={A} S={S} plan={use_plan}")
            g = rng.standard_normal(got.shape).astype(np.float32)
            gb = to_np(plan.backward(torch.from_numpy(g).to(d)))
            np.testing.assert_array_equal(gb, oracle.rotate_bwd_tfcompat(g, geom, oTinv(oracle, theta, plan), 0),
                                          err_msg=f"bwd case {case}: {H}x{W} pad={pad} A={A} S={S} plan={use_plan}")
        # round 4: the launch shapes a large plan would take -- angles dealt to the XCDs, both plan formats -- forced on this one
        for fmt in ("u16", "compact"):
            pf = RotatePlan(theta, H, W, pad, d, plan_format=fmt)
            if pf.planned[0]:
                with _lib.tuned("AFFINE", 1):
                    np.testing.assert_array_equal(to_np(pf.forward(torch.from_numpy(img).to(d))), want,
                                                  err_msg=f"affine / {fmt} case {case}: {H}x{W} pad={pad} A={A} S={S}")
        ex = RotatePlan(theta, H, W, pad, d, backward="exact")
        gx = to_np(ex.backward(torch.from_numpy(g).to(d)))
        lhs = float((to_np(ex.forward(torch.from_numpy(img).to(d))).astype(np.float64) * g).sum())
        rhs = float((gx.astype(np.float64) * img).sum())
        assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (case, lhs, rhs)
        if ex._exact_plan is not None:          # the planned gather: the oracle's in-order scatter, bit for bit
            np.testing.assert_array_equal(gx, oracle.rotate_bwd_exact(g, geom, oT(oracle, theta, ex), 0),
                                          err_msg=f"exact case {case}: {H}x{W} pad={pad} A={A} S={S}")
        # a random angle subset of the same (dense) plan, with repeats, through the index operand
        dense = RotatePlan(theta, H, W, pad, d)
        sub = rng.integers(0, A, int(rng.integers(1, 2 * A + 1)))
        idx = cp.as_angle_index(sub, d)
        T = oT(oracle, theta, dense)
        np.testing.assert_array_equal(to_np(dense.forward(torch.from_numpy(img).to(d), angles_i=idx)),
                                      oracle.rotate_fwd(img, geom, T[sub], 0), err_msg=f"sel fwd case {case}")
        gs = rng.standard_normal((S, len(sub), geom.PW)).astype(np.float32)
        np.testing.assert_array_equal(to_np(dense.backward(torch.from_numpy(gs).to(d), angles_i=idx)),
                                      oracle.rotate_bwd_tfcompat(gs, geom, oracle.invert_transforms(T)[sub], 0),
                                      err_msg=f"sel bwd case {case}")


def test_tiled_forward_against_golden(golden_dir):
    """The committed tiled vector (220 x 190, the smallest kind of slice the HIP path tiles)."""
    z = np.load(os.path.join(golden_dir, "rotate_tiled.npz"))
    d = dev()
    plan = RotatePlan(z["theta"], 220, 190, True, d)
    assert plan.tiled
    np.testing.assert_array_equal(to_np(plan.T8), z["T8"])          # host-built table == the committed (oracle-made) one
    np.testing.assert_array_equal(to_np(plan.forward(torch.from_numpy(z["img"]).to(d))), z["fwd_tiled_96x64"])


def test_random_siddon_and_tiled_geometries(oracle):
    """Seeded random cases for the two other projectors: the TomoPy-style one (odd / rectangular grids, padded and not,
    angles of every quadrant incl. exactly axis-aligned ones, where the bisection of the crossing lists falls back to
    libtomo's scan) and the tiled forward (random slice shapes larger than LDS)."""
    d = dev()
    rng = np.random.default_rng(int(os.environ.get("CTPVAE_FUZZ_SEED", 77)))
    n_cases = int(os.environ.get("CTPVAE_FUZZ_CASES", 10))
    for case in range(n_cases):
        ox, oz = int(rng.integers(2, 97)), int(rng.integers(2, 97))
        pad, A, S = bool(rng.integers(0, 2)), int(rng.integers(1, 13)), int(rng.integers(1, 4))
        theta = rng.uniform(-7.0, 7.0, A)
        theta[: min(A, 3)] = [0.0, np.pi / 2, np.pi][: min(A, 3)]
        img = rng.random((S, ox, oz), dtype=np.float32)
        got = cp.create_sinograms(img, theta, pad=pad)
        np.testing.assert_array_equal(got, np.swapaxes(oracle.siddon_project(img, theta, pad=pad), 0, 1),
                                      err_msg=f"siddon case {case}: {ox}x{oz} pad={pad} A={A} S={S}")
    for case in range(max(4, n_cases // 4)):
        H, W = int(rng.integers(150, 330)), int(rng.integers(210, 330))
        pad, A, S = bool(rng.integers(0, 2)), int(rng.integers(1, 9)), int(rng.integers(1, 6))
        if "CTPVAE_FUZZ_SEED" in os.environ and case % 2:
            S = int(rng.integers(16, 24))          # the segment backward's slice pairs
        theta = rng.uniform(-np.pi, np.pi, A)
        img = rng.standard_normal((S, H, W)).astype(np.float32)
        plan = RotatePlan(theta, H, W, pad, d)
        assert plan.tiled or "CTPVAE_FUZZ_SEED" in os.environ      # (the default seed's four shapes are all tiled)
        geom = oracle.Geometry(H, W, pad)
        want = (oracle.rotate_fwd_tiled(img, geom, oT(oracle, theta, plan), oracle.tile_shape(geom.H, geom.W)) if plan.tiled
                else oracle.rotate_fwd(img, geom, oT(oracle, theta, plan), 0))
        np.testing.assert_array_equal(to_np(plan.forward(torch.from_numpy(img).to(d))), want,
                                      err_msg=f"tiled case {case}: {H}x{W} pad={pad} A={A} S={S} tiled={plan.tiled}")
        g = rng.standard_normal((S, A, geom.PW)).astype(np.float32)
        np.testing.assert_array_equal(to_np(plan.backward(torch.from_numpy(g).to(d))),
                                      oracle.rotate_bwd_tfcompat(g, geom, oTinv(oracle, theta, plan), 0),
                                      err_msg=f"segment bwd case {case}: {H}x{W} pad={pad} A={A} S={S}")
    # the pixel-driven transpose on random grids (rectangular, tiny, larger than a workgroup's tile, detector wider or
    # narrower than the grid), random angles plus the special ones and their fp32 neighbours, signed and sparse data:
    # bit-equal to the oracle's ray-driven accumulation on even grids; odd grids (rays ON grid lines at 0 / pi / 2: a second
    # kernel, another order) to 1e-5; and SIRT, two fused launches per iteration, against the oracle's sirt.c restatement
    from ct_pvae_amd.recon import recon, siddon_backproject
    for case in range(n_cases):
        even = case % 3 != 2
        gx, gy = (int(rng.integers(1, 80)) * 2, int(rng.integers(1, 80)) * 2) if even else (int(rng.integers(2, 120)), int(rng.integers(2, 120)))
        dx, A, S = int(rng.integers(2, 200)), int(rng.integers(1, 40)), int(rng.integers(1, 12))
        theta = rng.uniform(-7.0, 7.0, A)
        special = np.array([0.0, np.pi / 2, np.pi / 4, 3 * np.pi / 4, np.pi, np.nextafter(np.float32(np.pi / 4), np.float32(1)),
                            np.arctan(0.5), np.arctan(2.0), -np.pi / 2])
        theta[: min(A, special.size)] = rng.permutation(special)[: min(A, special.size)]
        y = rng.standard_normal((S, A, dx)).astype(np.float32)
        y[:, rng.random(A) < 0.3] = 0.0
        got = to_np(siddon_backproject(torch.from_numpy(y).to(d), theta, gx, gy))
        want = np.zeros((S, gx, gy), np.float32)
        oracle.lib().oracle_siddon_backproject(y, S, A, dx, theta.astype(np.float32), dx / 2.0, gx, gy, want)
        msg = f"back-projector case {case}: grid {gx}x{gy} dx={dx} A={A} S={S}"
        if even and dx % 2 == 0:     # (an odd detector over an even grid also puts its rays on grid lines)
            np.testing.assert_array_equal(got, want, err_msg=msg)
        else:
            assert rel_err(got, want) <= REL, msg
    for case in range(max(3, n_cases // 4)):
        n, A, S = int(rng.integers(4, 40)) * 2, int(rng.integers(2, 30)), int(rng.integers(1, 10))
        theta = np.sort(rng.uniform(0.0, np.pi, A))
        sino = cp.create_sinograms(rng.random((S, n, n), dtype=np.float32), theta, pad=True)
        for it in (1, 3):
            got = to_np(recon(torch.from_numpy(sino).to(d), theta, sinogram_order=True, algorithm="sirt", num_iter=it))
            want = oracle.sirt(sino, theta, num_iter=it)
            assert rel_err(got, want) <= (1e-5 if it == 1 else 5e-5), (case, n, A, S, it)


@pytest.mark.parametrize("shape,A,S", [((512, 512), 23, 32), ((300, 330), 9, 21), ((260, 257), 64, 17)])
def test_step_plan_backward_equals_the_direct_kernel(oracle, shape, A, S):
    """Round 3: the backward of slices too large for the planned kernels reads a STEP PLAN -- per (angle, row octet, column) the
    first row's tap relative to the tile's segment and seven "the tap steps" bits -- instead of computing five index operations per
    tap.  Same taps, same order: bit-equal to the direct segment kernel (NO_PLAN) and to the oracle, with the per-slice factor of
    the fused likelihood backward, ragged shapes, odd batches; an unpadded canvas has no step plan and runs the direct kernel."""
    d = dev()
    rng = np.random.default_rng(shape[0] + A)
    theta = rng.uniform(-np.pi, np.pi, A)
    plan = RotatePlan(theta, shape[0], shape[1], True, d)
    assert plan._step_plan is not None and not plan.backward_uses_plan(S)
    g = rng.standard_normal((S, A, plan.PW)).astype(np.float32)
    gt = torch.from_numpy(g).to(d)
    scale = torch.from_numpy(rng.uniform(0.5, 2.0, S).astype(np.float32)).to(d)
    got, got_s = plan.backward(gt), plan.backward(gt, scale=scale)
    with _lib.tuned("NO_PLAN", 1):
        direct, direct_s = plan.backward(gt), plan.backward(gt, scale=scale)
    assert torch.equal(got, direct) and torch.equal(got_s, direct_s)
    # round 4: one slice pair per workgroup, or two pairs sharing the plan word and the address arithmetic (large launches);
    # batches that are no multiple of four, angle counts that end a chunk of 30 in a partial trip
    for ns in (2, 4):
        with _lib.tuned("STEP_NS", ns):
            assert torch.equal(plan.backward(gt), direct) and torch.equal(plan.backward(gt, scale=scale), direct_s), ns
            assert torch.equal(plan.backward(gt[:S - 2]), direct[:S - 2]), ns
    n_chk = 3
    geom = oracle.Geometry(shape[0], shape[1], True)
    np.testing.assert_array_equal(to_np(got[:n_chk]), oracle.rotate_bwd_tfcompat(g[:n_chk], geom, oTinv(oracle, theta, plan), 0))
    with _lib.tuned("STEP_NS", 4):
        np.testing.assert_array_equal(to_np(plan.backward(gt)[S - n_chk:]),
                                      oracle.rotate_bwd_tfcompat(g[S - n_chk:], geom, oTinv(oracle, theta, plan), 0))
    assert RotatePlan(theta, shape[0], shape[1], False, d)._step_plan is None


def test_random_step_plan_geometries(oracle):
    """Seeded random shapes (ragged in both directions), angle sets of every quadrant incl. axis-aligned ones, batch sizes just
    large enough for the stepped kernel: the step-plan backward against the direct segment kernel, bit for bit
    (CTPVAE_FUZZ_SEED / _CASES: more)."""
    d, orc = dev(), oracle
    rng = np.random.default_rng(int(os.environ.get("CTPVAE_FUZZ_SEED", 909)))
    n_cases, no_plan = int(os.environ.get("CTPVAE_FUZZ_CASES", 6)), 0
    for case in range(n_cases):
        H, W, A = int(rng.integers(128, 300)), int(rng.integers(128, 300)), int(rng.integers(1, 50))
        tiles = -(-W // 64) * -(-H // 32)
        S = 2 * (-(-512 // tiles)) + int(rng.integers(0, 4))          # >= 512 workgroups of slice pairs (odd batches too)
        theta = rng.uniform(-7.0, 7.0, A)
        theta[: min(A, 4)] = [0.0, np.pi / 2, np.pi, -np.pi / 2][: min(A, 4)]
        plan = RotatePlan(theta, H, W, True, d)
        g = torch.from_numpy(rng.standard_normal((S, A, plan.PW)).astype(np.float32)).to(d)
        tag = f"case {case}: {H}x{W} A={A} S={S}"
        if plan._step_plan is None:
            # the plan's overflow word: within a few 1e-3 rad of 90 / 270 degrees |t1| is 1 - 1e-5 and fp32 rounding can move a
            # tap by TWO bins between two rows (found by seed 32: 264 x 278, -4.71696 rad); the direct kernel then serves the
            # geometry -- legitimately, but it must stay the exception
            # and its result is still checked: the fallback against the oracle (two slices are enough)
            no_plan += 1
            geom = orc.Geometry(H, W, True)
            np.testing.assert_array_equal(to_np(plan.backward(g[:2])),
                                          orc.rotate_bwd_tfcompat(to_np(g[:2]), geom, oTinv(orc, theta, plan), 0), err_msg=tag)
            continue
        plan.backward_uses_step_plan = lambda S: True          # (shapes the planned backward serves would take it at this S)
        with _lib.tuned("STEP_NS", (2, 4)[case % 2]):          # one slice pair per workgroup / two
            got = plan.backward(g)
        with _lib.tuned("NO_PLAN", 1):
            ref = plan.backward(g)
        assert torch.equal(got, ref), tag
    assert no_plan <= max(1, n_cases // 20), f"{no_plan} of {n_cases} random geometries did not fit the step plan"


def test_step_plan_holds_two_bin_steps_near_a_right_angle():
    """Within ~1e-3 rad of 90 / 270 degrees |t1| = 1 - 1e-5 and fp32 rounding moves a column's tap by TWO bins between two rows
    now and then (seeded soak, round 3: 264 x 278 at -4.71696 rad).  The eight-byte plan words hold such steps; a plan of
    one-step words overflowed there -- for all of its angles."""
    d = dev()
    theta = np.array([-4.716961354375347, 0.3, np.pi / 2 + 2.0e-3, 3 * np.pi / 2 - 1.5e-3, 1.2])
    plan = RotatePlan(theta, 264, 278, True, d)
    assert plan._step_plan is not None
    g = torch.from_numpy(np.random.default_rng(1).standard_normal((25, len(theta), plan.PW)).astype(np.float32)).to(d)
    plan.backward_uses_step_plan = lambda S: True
    got = plan.backward(g)
    with _lib.tuned("STEP_NS", 4):
        got4 = plan.backward(g)
    with _lib.tuned("NO_PLAN", 1):
        ref = plan.backward(g)
    assert torch.equal(got, ref) and torch.equal(got4, ref)


def test_large_batches_of_small_slices_take_the_step_plan_too():
    """From 160 slices on (80 at <= 64 angles) the backward of 128 x 128 slices runs the stepped segment kernel instead of the
    planned gather (B=400 x 180 angles: 143 -> 112 us); the three kernels give the same bits, through the raw call and through
    the autograd API with the fused likelihood's per-slice factor."""
    d = dev()
    theta = phantoms.dense_theta(180)[::2]
    plan = RotatePlan(theta, 128, 128, True, d)
    assert plan.backward_kernel_name(50) == "rotate_bwd_planned_kernel" and plan.backward_kernel_name(200) == "rotate_bwd_stepped_kernel"
    rng = np.random.default_rng(3)
    g = torch.from_numpy(rng.standard_normal((200, 90, plan.PW)).astype(np.float32)).to(d)
    scale = torch.from_numpy(rng.uniform(0.5, 2.0, 200).astype(np.float32)).to(d)
    stepped, stepped_s = plan.backward(g), plan.backward(g, scale=scale)
    forced = RotatePlan(theta, 128, 128, True, d)
    forced.backward_uses_step_plan = lambda S: False
    forced.backward_uses_plan = lambda S: True
    assert torch.equal(stepped, forced.backward(g)) and torch.equal(stepped_s, forced.backward(g, scale=scale))
    with _lib.tuned("SEG_PPT", 4):                  # the direct segment kernel through the same entry point
        assert torch.equal(stepped, plan.backward(g))
    x = torch.from_numpy(rng.random((200, 128, 128, 1), dtype=np.float32)).to(d).requires_grad_(True)
    cp.project_tf_fast(x, theta, pad=True, dim=2, integrate_vae=True).backward(g[..., None])
    assert torch.equal(x.grad[..., 0], stepped)


@pytest.mark.parametrize("shift", [0.3, -1.25, 2.0])
def test_ray_driven_pair_with_a_shifted_rotation_centre(oracle, shift):
    """The C ABI takes tomopy's `center`; the reference only ever passes None (dx / 2), which is all the Python wrappers do --
    so the shifted centre (libtomo's `mov`, incl. its +0.01 nudge off whole pixels) is checked through the entry points
    themselves: forward (8 slices per walk) and the pixel-driven back-projector against the oracle, bit for bit."""
    import ctypes
    from ct_pvae_amd.helper_functions import _siddon_tables
    d = dev()
    lib = _lib.load()
    rng = np.random.default_rng(7)
    S, n, dx, A = 9, 64, 94, 17
    theta = np.sort(rng.uniform(0.0, np.pi, A)).astype(np.float32)
    center = dx / 2.0 + shift
    obj = rng.random((S, n, n), dtype=np.float32)
    sin_t, cos_t, quad = _siddon_tables(theta, d)
    sp = torch.cuda.current_stream().cuda_stream
    t = torch.from_numpy(obj).to(d)
    sino = nan_out((S, A, dx), d)
    fws = torch.empty(int(lib.ctpvae_siddon_fwd_workspace_bytes(S, n, n)), dtype=torch.uint8, device=d)
    _lib.check(lib.ctpvae_siddon_fwd_ws_f32(t.data_ptr(), S, n, n, sin_t.data_ptr(), cos_t.data_ptr(), quad.data_ptr(), A, dx,
                                            ctypes.c_float(center), None, None, fws.data_ptr(), sino.data_ptr(), sp), "siddon_fwd_ws")
    want = np.empty((S, A, dx), np.float32)
    oracle.lib().oracle_siddon_project(obj, S, n, n, theta, A, dx, center, want)
    np.testing.assert_array_equal(to_np(sino), want)
    y = rng.standard_normal((S, A, dx)).astype(np.float32)
    yt = torch.from_numpy(y).to(d)
    ws = torch.empty(int(lib.ctpvae_siddon_bwd_workspace_bytes(S, n, n, A, dx)), dtype=torch.uint8, device=d)
    rec = nan_out((S, n, n), d)
    _lib.check(lib.ctpvae_siddon_bwd_f32(yt.data_ptr(), S, n, n, sin_t.data_ptr(), cos_t.data_ptr(), quad.data_ptr(), A, dx,
                                         ctypes.c_float(center), ws.data_ptr(), rec.data_ptr(), sp), "siddon_bwd")
    want_r = np.zeros((S, n, n), np.float32)
    oracle.lib().oracle_siddon_backproject(y, S, A, dx, theta, center, n, n, want_r)
    if float(shift).is_integer():      # whole-pixel shifts put the half-pixel rays back on the grid's lines only through mov's nudge
        assert rel_err(to_np(rec), want_r) <= REL
    else:
        np.testing.assert_array_equal(to_np(rec), want_r)


def test_launches_are_graph_capturable():
    """The library allocates nothing and never synchronises, so a caller can capture its launches into a HIP graph
    (torch.cuda.graph) and replay them: same results as eager launches."""
    d = dev()
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, 20)]
    plan = RotatePlan(theta, 128, 128, True, d)
    x = torch.rand((6, 128, 128), device=d)
    g = torch.rand((6, 20, 184), device=d)
    sino, gimg = nan_out((6, 20, 184), d), nan_out(x.shape, d)

    def step():
        plan.forward(x, out=sino)
        plan.backward(g, out=gimg)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                  # warm-up off the default stream (plans, one-time attributes)
        step()
    torch.cuda.synchronize()
    want_s, want_g = sino.clone(), gimg.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    sino.zero_()
    gimg.zero_()
    x2 = torch.rand_like(x)
    x.copy_(x2)                                     # new input in the captured buffers
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(gimg, want_g)                # same cotangent -> same gradient image
    assert torch.equal(sino, plan.forward(x2))      # the replay projected the new input
    assert not torch.equal(sino, want_s)


def test_bench_single_process_contract():
    """`python bench.py` as the driver runs it at N = 1 (few steps here): one JSON line with the contract's keys, the
    roofline and cpu_baseline objects, traffic from the committed PMC passes for the default workload."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "10"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 10 and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["traffic"] is not None and r["traffic"] > 4.0e6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 1e7                      # tens of millions of projections per second on an MI355X
    # round 5: all four (interp x backward) modes and the cold-input figure ride in the driver's line
    m = d["modes"]
    assert {"nearest_tf_compat", "nearest_exact", "bilinear_tf_compat", "bilinear_exact"} <= set(m)
    for k in ("nearest_tf_compat", "nearest_exact", "bilinear_tf_compat", "bilinear_exact"):
        assert 1.0 < m[k]["fwd_us"] < 100.0 and 1.0 < m[k]["bwd_us"] < 100.0 and m[k]["projections_per_s"] > 5e6, (k, m[k])
        assert abs(m[k]["hbm_frac"]["fwd"] - 4.0 * 50 * (128 * 128 + 20 * 184) / (m[k]["fwd_us"] * 1e-6) / 8e12) < 1e-9
    assert m["bilinear_exact"]["bwd_us"] < 40.0          # the scatter kernel took 367 us (profiles/r05_modes_baseline_*)
    assert 0.5 * d["value"] < d["cold"]["value"] <= 1.1 * d["value"] and d["cold"]["ms_per_step"] > 0


def test_bench_strong_scaling_two_ranks_and_projection():
    """BASELINE configs[3] fixes the batch: `--total-batch` shares ONE batch between the ranks (sharding.shard_range) and says
    "strong"; two ranks rehearsed on this box's one GPU over gloo.  `--project-scaling`: the 1 / 2 / 4 / 8-rank shares of a
    fixed batch timed on one GPU and the speed-ups they imply."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total-batch", "101", "--steps", "20",
                          "--warmup", "5", "--min-ms", "5"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["objects_per_gpu"] == 51
    assert abs(d["value"] - 101 * 20 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--project-scaling", "--total-batch", "96", "--angles", "20"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert [d["shares"][k]["objects_per_rank"] for k in ("1", "2", "4", "8")] == [96, 48, 24, 12]
    assert d["shares"]["1"]["projected_speedup"] == 1.0 and 1.0 < d["value"] <= 8.5 and d["scaling"].startswith("strong")


def test_raw_operator_checks_its_operands():
    """RotatePlan.forward / backward / forward_loglik launch kernels that index by the plan's shapes: a tensor of another
    shape, dtype, device or layout is refused on the host (ValueError), never handed to a kernel."""
    d = dev()
    plan = RotatePlan(np.linspace(0, 3, 5), 40, 60, True, d)
    x = torch.rand((2, 40, 60), device=d)
    g = torch.rand((2, 5, plan.PW), device=d)
    plan.forward(x), plan.backward(g)
    for bad in (torch.rand((2, 60, 40), device=d), torch.rand((2, 40, 60), device=d, dtype=torch.float64),
                torch.rand((2, 40, 60)), torch.rand((2, 40, 120), device=d)[:, :, ::2], torch.rand((40, 60), device=d),
                torch.rand((0, 41, 60), device=d)):
        with pytest.raises(ValueError):
            plan.forward(bad)
    with pytest.raises(ValueError):
        plan.forward(x, out=torch.empty((3, 5, plan.PW), device=d))
    with pytest.raises(ValueError):
        plan.backward(torch.rand((2, 5, plan.PW + 1), device=d))
    with pytest.raises(ValueError):
        plan.backward(g, out=torch.empty((2, 60, 40), device=d))
    with pytest.raises(ValueError):
        plan.forward_loglik(x, torch.rand((2, 4), device=d), g, torch.tensor(1e3, device=d), 1e-7)


def test_likelihood_entry_points_check_their_operands():
    d = dev()
    theta = torch.linspace(0, 3, 6, device=d)
    x = torch.rand((2, 32, 32, 1), device=d)
    P = cp.num_proj_pix(32, 32)
    with pytest.raises(ValueError):
        cp.calculate_log_prob_M_given_R(x, torch.rand((2, 5), device=d), torch.rand((2, 6, P), device=d), 1e3, 1e-7, theta=theta)
    with pytest.raises(ValueError):
        cp.calculate_log_prob_M_given_R(x, torch.rand((2, 6), device=d), torch.rand((2, 6, P + 1), device=d), 1e3, 1e-7, theta=theta)
    with pytest.raises(ValueError):
        cp.gaussian_poisson_log_prob(torch.rand((2, 6, P), device=d), torch.rand((2, 6), device=d),
                                     torch.rand((2, 5, P), device=d), 1e3, 1e-7)
    ok = cp.calculate_log_prob_M_given_R(x, torch.rand((2, 6), device=d), torch.rand((2, 6, P), device=d), 1e3, 1e-7, theta=theta)
    assert ok.shape == (2, 6, P, 1)


def test_cpp_autograd_node_and_python_node_are_the_same_calls(oracle, torch_node):
    """csrc/torch_node.cpp (the training layout's C++ autograd node) and the Python node make the same two C-ABI calls:
    forward and gradient are bit-equal between them and to the oracle; inputs the C++ node does not take (float64,
    non-contiguous, no-plan geometries) fall through to the Python node."""
    from ct_pvae_amd import forward_functions as ff
    d = dev()
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, 20)]
    img = phantoms.foam_batch(5, 128, seed=3, supersample=2)
    g = np.random.default_rng(4).standard_normal((5, 20, 184, 1)).astype(np.float32)
    res = {}
    for use_cpp in (True, False):
        ff.USE_CPP_NODE = use_cpp
        try:
            x = torch.from_numpy(img[..., None]).to(d).requires_grad_(True)
            out = cp.project_tf_fast(x, theta, pad=True, dim=2, integrate_vae=True)
            assert ("RotateVae" in out.grad_fn.name()) == use_cpp, out.grad_fn.name()
            (out * torch.from_numpy(g).to(d)).sum().backward()
            res[use_cpp] = (to_np(out), to_np(x.grad))
        finally:
            ff.USE_CPP_NODE = True
    assert np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1])
    geom = oracle.Geometry(128, 128, True)
    T = oracle.rotate_transforms(theta, geom.PH, geom.PW)
    assert np.array_equal(res[True][0][..., 0], oracle.rotate_fwd(img, geom, T, oracle.NEAREST))
    assert np.array_equal(res[True][1][..., 0],
                          oracle.rotate_bwd_tfcompat(g[..., 0], geom, oracle.invert_transforms(T), oracle.NEAREST))
    # no grad wanted: no node is recorded, same numbers
    y = cp.project_tf_fast(torch.from_numpy(img[..., None]).to(d), theta, pad=True, dim=2, integrate_vae=True)
    assert y.grad_fn is None and np.array_equal(to_np(y), res[True][0])
    # float64 and non-contiguous inputs take the Python node
    x64 = torch.from_numpy(img[..., None].astype(np.float64)).to(d).requires_grad_(True)
    o64 = cp.project_tf_fast(x64, theta, pad=True, dim=2, integrate_vae=True)
    assert o64.dtype is torch.float64 and "RotateVae" not in o64.grad_fn.name()
    assert np.array_equal(to_np(o64).astype(np.float32), res[True][0])
    xt = torch.from_numpy(np.ascontiguousarray(img.transpose(0, 2, 1))[..., None]).to(d).transpose(1, 2).requires_grad_(True)
    ot = cp.project_tf_fast(xt, theta, pad=True, dim=2, integrate_vae=True)
    assert "RotateVae" not in ot.grad_fn.name() and np.array_equal(to_np(ot), res[True][0])
    # double backward is not defined for either node; a second first-order backward through a retained graph is
    x = torch.from_numpy(img[..., None]).to(d).requires_grad_(True)
    out = cp.project_tf_fast(x, theta, pad=True, dim=2, integrate_vae=True)
    gt = torch.from_numpy(g).to(d)
    g1, = torch.autograd.grad((out * gt).sum(), x, retain_graph=True)
    g2, = torch.autograd.grad((out * gt).sum(), x)
    assert torch.equal(g1, g2) and np.array_equal(to_np(g1), res[True][1])


@pytest.mark.parametrize("B,subset,upstream", [(5, False, "sum"), (5, True, "sum"), (6, True, "full"), (4, False, "full"),
                                               (5, False, "scalar"), (80, False, "sum"), (168, False, "sum"), (168, False, "scalar")])
def test_cpp_loglik_node_matches_the_python_node(oracle, torch_node, B, subset, upstream):
    """calculate_log_prob_M_given_R through csrc/torch_node.cpp's RotateLogLik against _ProjectLogLik (same C-ABI calls):
    value and gradient bit-equal -- dense plan + angle subset or compact, the three upstream-gradient kinds (per-object sum:
    the scaled backward; arbitrary; fully expanded scalar), the large batch whose backward is the segment kernel, and the larger
    one (168 slices: 672 tiles) whose backward is the STEPPED kernel over the step plan -- the node's mode 3."""
    from ct_pvae_amd import forward_functions as ff
    d = dev()
    rng = np.random.default_rng(B * 7 + subset)
    dense = phantoms.dense_theta(180)
    theta = dense if subset else dense[phantoms.sparse_angle_indices(180, 20)]
    sub = np.sort(rng.permutation(180)[:20]).astype(np.int32) if subset else None
    A = len(theta)
    img = phantoms.foam_batch(B, 128, seed=B, supersample=1)
    mask = torch.from_numpy((rng.random((B, A)) * 0.1 + 0.01).astype(np.float32)).to(d)
    meas = torch.from_numpy(rng.random((B, A, 184)).astype(np.float32)).to(d)
    n = 20
    w = torch.from_numpy(rng.standard_normal(B).astype(np.float32)).to(d)
    gfull = torch.from_numpy(rng.standard_normal((B, n, 184, 1)).astype(np.float32)).to(d)
    res = {}
    for use_cpp in (True, False):
        ff.USE_CPP_NODE = use_cpp
        try:
            x = torch.from_numpy(img[..., None]).to(d).requires_grad_(True)
            lp = cp.calculate_log_prob_M_given_R(x, mask, meas, 1e4, 1.2e-7, theta=theta, angles_i=sub, pad=True)
            assert lp.shape == (B, n, 184, 1) and ("RotateLogLik" in lp.grad_fn.name()) == use_cpp, lp.grad_fn.name()
            if upstream == "sum":
                lp.sum(dim=(1, 2, 3)).backward(w)
            elif upstream == "scalar":
                lp.sum().backward()
            else:
                lp.backward(gfull)
            res[use_cpp] = (to_np(lp), to_np(x.grad))
        finally:
            ff.USE_CPP_NODE = True
    assert np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1])
    assert np.isfinite(res[True][1]).all() and np.abs(res[True][1]).max() > 0
    # against the oracle's projection + log-likelihood
    geom = oracle.Geometry(128, 128, True)
    T = oracle.rotate_transforms(theta, geom.PH, geom.PW)
    m, y = to_np(mask), to_np(meas)
    if subset:
        T, m, y = T[sub], m[:, sub], y[:, sub]
    want = oracle.loglik(oracle.rotate_fwd(img, geom, T, oracle.NEAREST), m, y, 1e4, 1.2e-7)
    assert rel_err(res[True][0][..., 0], want) <= REL
    # no gradient wanted / trainable pnm: the Python node (which skips dlp, or reduces d/d pnm)
    lp = cp.calculate_log_prob_M_given_R(torch.from_numpy(img[..., None]).to(d), mask, meas, 1e4, 1.2e-7, theta=theta,
                                         angles_i=sub, pad=True)
    assert lp.grad_fn is None and np.array_equal(to_np(lp), res[True][0])
    pnm = torch.tensor(1e4, device=d, requires_grad=True)
    x = torch.from_numpy(img[..., None]).to(d).requires_grad_(True)
    lp = cp.calculate_log_prob_M_given_R(x, mask, meas, pnm, 1.2e-7, theta=theta, angles_i=sub, pad=True)
    assert "RotateLogLik" not in lp.grad_fn.name()


@pytest.mark.parametrize("layout", ["vae", "dim3", "dim2"])
def test_project_tf_fast_siddon_model_is_differentiable(oracle, layout):
    """model="siddon" (SURVEY 8b's keyword-only extension): project_tf_fast's three layouts through the TomoPy-style
    projector -- forward bit-equal to the oracle's restatement of project.c, backward = its transpose (oracle's
    restatement, 1e-5; <A x, y> = <x, A^T y>), float64 in -> float64 out like the reference's xdesign phantoms."""
    d = dev()
    rng = np.random.default_rng(11)
    theta = rng.uniform(0, np.pi, 13)
    img = rng.random((3, 40, 40), dtype=np.float32)
    want = np.swapaxes(oracle.siddon_project(img, theta, pad=True), 0, 1)           # [S][A][dx]
    dx = want.shape[2]
    gy = rng.standard_normal(want.shape).astype(np.float32)
    gwant = np.zeros_like(img)
    oracle.lib().oracle_siddon_backproject(np.ascontiguousarray(gy), 3, len(theta), dx, theta.astype(np.float32), dx / 2.0, 40, 40, gwant)
    if layout == "vae":
        x = torch.from_numpy(img[..., None]).to(d).requires_grad_(True)
        out = cp.project_tf_fast(x, theta, pad=True, dim=2, integrate_vae=True, model="siddon")
        assert out.shape == (3, 13, dx, 1)
        got, g = to_np(out)[..., 0], torch.from_numpy(gy[..., None]).to(d)
        out.backward(g)
        ggot = to_np(x.grad)[..., 0]
    elif layout == "dim3":
        x = torch.from_numpy(np.transpose(img, (1, 2, 0)).astype(np.float64)).to(d).requires_grad_(True)
        out = cp.project_tf_fast(x, theta, pad=True, model="siddon")
        assert out.shape == (13, dx, 3) and out.dtype is torch.float64
        got = np.transpose(to_np(out), (2, 0, 1)).astype(np.float32)
        out.backward(torch.from_numpy(np.transpose(gy, (1, 2, 0)).astype(np.float64)).to(d))
        ggot = np.transpose(to_np(x.grad), (2, 0, 1))
    else:
        x = torch.from_numpy(img[0]).to(d).requires_grad_(True)
        out = cp.project_tf_fast(x, theta, pad=True, dim=2, model="siddon")
        assert out.shape == (13, dx, 1)
        got, want, gwant = to_np(out)[None, ..., 0], want[:1], gwant[:1]
        out.backward(torch.from_numpy(gy[0][..., None]).to(d))
        ggot = to_np(x.grad)[None]
        gy = gy[:1]
    np.testing.assert_array_equal(got, want)
    assert rel_err(ggot, gwant) <= REL
    lhs, rhs = float((want.astype(np.float64) * gy).sum()), float((img[:len(ggot)].astype(np.float64) * ggot).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)
    with pytest.raises(ValueError, match="model must be"):
        cp.project_tf_fast(x, theta, pad=True, dim=x.dim() if layout != "vae" else 2, integrate_vae=layout == "vae", model="fan")


@pytest.mark.parametrize("tomopy_geometry", [False, True])
def test_iradon_gradient_is_the_transpose_of_the_oracle_operator(oracle, tomopy_geometry):
    """iradon is linear in the sinogram and (like the reference's TF-op version) differentiable: the gradient must be M^T g
    for the matrix M the ORACLE's iradon defines -- built column by column from unit sinograms at a small size (a complex
    filter_1d, so the transposed filter's kernel is NOT symmetric; angles beyond the detector, so the clamped edge bins
    collect) -- and <iradon(s), g> = <s, grad> at full size, float32 in -> float32 gradient out."""
    d = dev()
    rng = np.random.default_rng(5)
    A, P, X, Y = 5, 12, 9, 8                       # 9 x 8 pixels reach beyond 12 bins on the diagonal
    theta = rng.uniform(0, np.pi, A)
    filt = rng.standard_normal(P) + 1j * rng.standard_normal(P)
    s = rng.standard_normal((2, A, P))
    g = rng.standard_normal((2, X, Y))
    st = torch.from_numpy(s).to(d).requires_grad_(True)
    out = cp.iradon(st, theta, X, Y, filt, tomopy_geometry=tomopy_geometry)
    out.backward(torch.from_numpy(g).to(d))
    if not tomopy_geometry:                        # the oracle restates the reference's geometry
        assert rel_err(to_np(out), oracle.iradon(s, theta, X, Y, filt)) <= 1e-10
        M = oracle.iradon(np.eye(A * P).reshape(A * P, A, P), theta, X, Y, filt).reshape(A * P, X * Y)   # rows: M^T
        want = (M @ g.reshape(2, X * Y).T).T.reshape(2, A, P)
        assert rel_err(to_np(st.grad), want) <= 1e-10
    lhs, rhs = float((to_np(out) * g).sum()), float((s * to_np(st.grad)).sum())
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), 1.0)
    # full size, float32 sinogram, the ramp: adjointness in fp64 accumulators
    theta = phantoms.dense_theta(180)[::9]
    filt = np.abs(np.fft.fftfreq(184)) * 2
    s32 = torch.from_numpy(rng.random((3, 20, 184)).astype(np.float32)).to(d).requires_grad_(True)
    g = rng.standard_normal((3, 128, 128))
    out = cp.iradon(s32, theta, 128, 128, filt, tomopy_geometry=tomopy_geometry)
    out.backward(torch.from_numpy(g).to(d))
    assert out.dtype is torch.float64 and s32.grad.dtype is torch.float32
    lhs = float((to_np(out) * g).sum())
    rhs = float((to_np(s32).astype(np.float64) * to_np(s32.grad).astype(np.float64)).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0)       # the gradient was rounded to float32
    # no gradient wanted: no graph
    assert cp.iradon(s32.detach(), theta, 128, 128, filt).grad_fn is None


# ---- round 3: compact (step-coded) forward plans, csrc/rotate_cplan.hip ---------------------------------------------------
@pytest.mark.parametrize("shape,A,S", [((128, 128), 20, 50), ((128, 128), 180, 5), ((64, 64), 7, 3), ((33, 47), 12, 4),
                                       ((150, 150), 9, 2), ((6, 5), 5, 3), ((128, 128), 20, 1)])
def test_compact_plan_equals_u16_plan_and_oracle(oracle, shape, A, S):
    """The compact plan stores the u16 plan's taps as first tap + 2 bits per row: same taps, same row order -> the same
    bits, at every launch shape (pairs, singles, few or many task groups)."""
    d = dev()
    rng = np.random.default_rng(A * 1000 + S)
    img = rng.random((S,) + shape, dtype=np.float32)
    theta = rng.uniform(-4.0, 7.0, A).astype(np.float32) if A != 180 else phantoms.dense_theta(180)
    plan = RotatePlan(theta, shape[0], shape[1], True, d, plan_format="compact")
    assert plan.compact and plan.planned[0]
    x = torch.from_numpy(img).to(d)
    got = plan.forward(x)
    want = oracle.rotate_fwd(img, oracle.Geometry(shape[0], shape[1], True), oT(oracle, theta, plan), 0)
    np.testing.assert_array_equal(to_np(got), want)
    plan16 = RotatePlan(theta, shape[0], shape[1], True, d, plan_format="u16")
    assert not plan16.compact and plan16.planned[0]
    assert torch.equal(plan16.forward(x), got)
    assert RotatePlan(theta, shape[0], shape[1], True, d).compact == (A >= RotatePlan.COMPACT_MIN_ANGLES)   # "auto"
    _lib.tune("NO_COMPACT", 1)
    assert not RotatePlan(theta, shape[0], shape[1], True, d, plan_format="compact").compact   # the knob wins
    _lib.tune("NO_COMPACT", -1)
    for ns, G, w in ((1, 1, 4), (2, 3, 16), (1, 7, 8), (2, 12, 5)):
        _lib.tune("NS", ns), _lib.tune("G", G), _lib.tune("WAVES", w)
        assert torch.equal(plan.forward(x), got), (ns, G, w)


def test_compact_plan_falls_back_where_the_code_does_not_fit(oracle):
    """An unpadded canvas at an oblique angle has rays that are still inside the slice at the canvas' last row: no border cell
    to step onto -> the plan's overflow word is raised and the u16 plan is used; results are the oracle's either way."""
    d = dev()
    rng = np.random.default_rng(3)
    img = rng.random((3, 40, 40), dtype=np.float32)
    theta = np.array([0.0, 0.4, 0.79, 1.3, np.pi / 2, 2.5], np.float32)
    plan = RotatePlan(theta, 40, 40, False, d, plan_format="compact")
    assert plan.planned[0] and not plan.compact
    np.testing.assert_array_equal(to_np(plan.forward(torch.from_numpy(img).to(d))),
                                  oracle.rotate_fwd(img, oracle.Geometry(40, 40, False), oT(oracle, theta, plan), 0))
    # axis-aligned angles only: every ray leaves through the canvas edge with the slice -> nothing to step onto either,
    # unless the ray is exactly as long as its block's walk; whichever the builder decides, the sums are the oracle's
    theta2 = np.array([0.0, np.pi / 2, np.pi], np.float32)
    plan2 = RotatePlan(theta2, 40, 40, False, d, plan_format="compact")
    np.testing.assert_array_equal(to_np(plan2.forward(torch.from_numpy(img).to(d))),
                                  oracle.rotate_fwd(img, oracle.Geometry(40, 40, False), oT(oracle, theta2, plan2), 0))


def test_compact_plan_subsets_and_likelihood(oracle):
    """Angle subsets of a dense compact plan and the fused likelihood epilogue against the oracle."""
    d = dev()
    rng = np.random.default_rng(11)
    S, N = 6, 128
    img = rng.random((S, N, N), dtype=np.float32)
    theta = phantoms.dense_theta(180)
    plan = RotatePlan(theta, N, N, True, d)
    assert plan.compact
    geom = oracle.Geometry(N, N, True)
    T = oT(oracle, theta, plan)
    x = torch.from_numpy(img).to(d)
    for n in (1, 20, 64, 65, 200):
        sub = rng.integers(0, 180, n).astype(np.int32)
        ai = torch.from_numpy(sub).to(d)
        np.testing.assert_array_equal(to_np(plan.forward(x, angles_i=ai)), oracle.rotate_fwd(img, geom, T[sub], 0))
    sub = rng.permutation(180)[:20].astype(np.int32)
    ai = torch.from_numpy(sub).to(d)
    mask = rng.uniform(0.01, 0.1, (S, 180)).astype(np.float32)
    meas = rng.random((S, 180, plan.PW), dtype=np.float32)
    pnm = torch.tensor([1e4], device=d)
    sino, lp, dlp = plan.forward_loglik(x, torch.from_numpy(mask).to(d), torch.from_numpy(meas).to(d), pnm, 1.2e-7,
                                        with_dlp=True, angles_i=ai, dense_inputs=True)
    want = oracle.rotate_fwd(img, geom, T[sub], 0)
    np.testing.assert_array_equal(to_np(sino), want)
    assert rel_err(to_np(lp), oracle.loglik(want, mask[:, sub], meas[:, sub], 1e4, 1.2e-7)) <= REL
    p16 = RotatePlan(theta, N, N, True, d, plan_format="u16")
    s16, lp16, dlp16 = p16.forward_loglik(x, torch.from_numpy(mask).to(d), torch.from_numpy(meas).to(d), pnm, 1.2e-7,
                                          with_dlp=True, angles_i=ai, dense_inputs=True)
    assert torch.equal(s16, sino) and torch.equal(lp16, lp) and torch.equal(dlp16, dlp)


# ---- round 4: the absolute-LDS-address guard is a host-side error, not a device trap ------------------------------------------
def test_static_lds_in_an_absolute_addressing_kernel_is_a_host_error():
    """The planned / compact kernels address LDS absolutely and need their dynamic array at address 0, i.e. NO static LDS.  The
    library checks the code object's static LDS size on the host before a kernel's first launch (hipFuncGetAttributes); the knob
    FAKE_STATIC_LDS stubs a non-zero answer: every such entry point then fails with a message through ctpvae_last_error() --
    nothing is launched, nothing aborts -- and works again once the stub is gone."""
    d = dev()
    rng = np.random.default_rng(2)
    theta = phantoms.dense_theta(180)[::9]
    x = torch.from_numpy(rng.random((4, 128, 128), dtype=np.float32)).to(d)
    plans = {fmt: RotatePlan(theta, 128, 128, True, d, plan_format=fmt) for fmt in ("u16", "compact")}
    ref = {fmt: p.forward(x) for fmt, p in plans.items()}
    g = torch.from_numpy(rng.standard_normal((4, 20, 184)).astype(np.float32)).to(d)
    gref = plans["u16"].backward(g)
    sub = torch.from_numpy(np.array([3, 1, 7], np.int32))
    gsub = plans["u16"].backward(g[:, :3].contiguous(), angles_i=sub)
    big = RotatePlan(np.pi * np.arange(6) / 6, 300, 260, True, d)
    xb = torch.from_numpy(rng.random((2, 300, 260), dtype=np.float32)).to(d)
    bref = big.forward(xb)
    with _lib.tuned("FAKE_STATIC_LDS", 16):
        for call in (lambda: plans["u16"].forward(x), lambda: plans["compact"].forward(x), lambda: plans["u16"].backward(g),
                     lambda: plans["u16"].backward(g[:, :3].contiguous(), angles_i=sub), lambda: big.forward(xb)):
            with pytest.raises(_lib.RadonLibraryError, match="static LDS"):
                call()
    assert torch.equal(plans["u16"].forward(x), ref["u16"]) and torch.equal(plans["compact"].forward(x), ref["compact"])
    assert torch.equal(plans["u16"].backward(g), gref) and torch.equal(big.forward(xb), bref)
    assert torch.equal(plans["u16"].backward(g[:, :3].contiguous(), angles_i=sub), gsub)


# ---- round 4: the u16 planned forward with its ANGLES dealt to the XCDs ---------------------------------------------------------
@pytest.mark.parametrize("A,S", [(180, 16), (180, 31), (90, 9), (20, 50), (7, 3)])
def test_planned_forward_with_angles_dealt_to_the_xcds(oracle, A, S):
    """Which workgroup walks which (angle, bin block) task is a launch-shape decision: dealing a class's angles to task groups
    (knob AFFINE = 1: every XCD sees one eighth of the plan) or its task list round-robin (AFFINE = 0) gives the same bits, with
    and without the likelihood epilogue, odd batches and angle counts below the group count included."""
    d = dev()
    rng = np.random.default_rng(A * 100 + S)
    img = rng.random((S, 128, 128), dtype=np.float32)
    theta = np.sort(rng.uniform(0, np.pi, A)).astype(np.float32)
    plan = RotatePlan(theta, 128, 128, True, d, plan_format="u16")
    x = torch.from_numpy(img).to(d)
    mask = torch.from_numpy(rng.uniform(0.01, 0.1, (S, A)).astype(np.float32)).to(d)
    meas = torch.from_numpy(rng.random((S, A, plan.PW), dtype=np.float32)).to(d)
    pnm = torch.tensor([1e4], device=d)
    res = {}
    for aff in (0, 1):
        with _lib.tuned("AFFINE", aff):
            res[aff] = (plan.forward(x),) + tuple(plan.forward_loglik(x, mask, meas, pnm, 1.2e-7, with_dlp=True))
            for G in (4, 8):
                _lib.tune("G", G)
                assert torch.equal(plan.forward(x), res[aff][0]), (aff, G)
            _lib.tune("G")
    assert all(torch.equal(u, v) for u, v in zip(res[0], res[1]))
    n = min(S, 3)
    np.testing.assert_array_equal(to_np(res[1][0][:n]), oracle.rotate_fwd(img[:n], oracle.Geometry(128, 128, True), oT(oracle, theta, plan), 0))


# ---- round 3: per-object log-likelihood sums inside the projector launch -------------------------------------
@pytest.mark.parametrize("fmt,subset", [("compact", True), ("compact", False), ("u16", True), ("u16", False)])
def test_per_object_loglik_sums_are_the_ordered_sum_of_the_two_step_path(oracle, fmt, subset):
    """reduce_sum over angles and bins (ctvae/helper_functions.py:305-312) in the library's fixed order: the fused epilogue of
    the compact kernel (partials per task + ordered pass), the standalone same-order kernel and the oracle's statement of the
    order give the same bits on the two-step path's log-probabilities."""
    d = dev()
    rng = np.random.default_rng(5)
    S, N = 7, 128
    img = rng.random((S, N, N), dtype=np.float32)
    theta = phantoms.dense_theta(180)
    plan = RotatePlan(theta, N, N, True, d, plan_format=fmt)
    x = torch.from_numpy(img).to(d)
    mask = torch.from_numpy(rng.uniform(0.01, 0.1, (S, 180)).astype(np.float32)).to(d)
    meas = torch.from_numpy(rng.random((S, 180, plan.PW), dtype=np.float32)).to(d)
    pnm = torch.tensor([1e4], device=d)
    ai = torch.from_numpy(rng.permutation(180)[:20].astype(np.int32)).to(d) if subset else None
    sino, lp, dlp = plan.forward_loglik(x, mask, meas, pnm, 1.2e-7, with_dlp=True, angles_i=ai, dense_inputs=subset)
    sums, dlp2 = plan.forward_loglik_sums(x, mask, meas, pnm, 1.2e-7, angles_i=ai, dense_inputs=subset)
    want = oracle.loglik_object_sums(to_np(lp), 0)
    np.testing.assert_array_equal(to_np(sums), want)
    assert torch.equal(dlp2, dlp)
    lib = _lib.load()
    out = nan_out(S, d)
    assert lib.ctpvae_loglik_object_sums_f32(lp.data_ptr(), S, lp.shape[1], plan.PW, 0, out.data_ptr(), None) == 0
    np.testing.assert_array_equal(to_np(out), want)
    assert lib.ctpvae_loglik_object_sums_f32(lp.data_ptr(), S, lp.shape[1], plan.PW, 1, out.data_ptr(), None) == 0
    np.testing.assert_array_equal(to_np(out), oracle.loglik_object_sums(to_np(lp), 1))


def test_object_sums_of_many_angles_and_host_subsets_past_the_argument_list(oracle):
    """Round-3 ADVICE: (i) the same-order reduction of a stored [S][A][P] array takes any number of angles (it used to refuse
    A x tasks-per-row > 16384: a many-angle tiled geometry that trained through lp.sum() then failed) -- the order adds angles
    in groups of 64, so the kernel keeps one group's task sums in LDS at a time; (ii) a HOST-resident angle subset longer than
    the 256 indices a launch's arguments hold is uploaded and runs on the device-index form of the planned backward."""
    d = dev()
    rng = np.random.default_rng(21)
    lib = _lib.load()
    lp = torch.from_numpy(rng.standard_normal((2, 600, 2048)).astype(np.float32)).to(d)       # 600 x 32 tasks = 19200
    out = nan_out(2, d)
    for part in (0, 1):
        assert lib.ctpvae_loglik_object_sums_f32(lp.data_ptr(), 2, 600, 2048, part, out.data_ptr(), None) == 0
        np.testing.assert_array_equal(to_np(out), oracle.loglik_object_sums(to_np(lp), part))
    theta = phantoms.dense_theta(180)
    plan = RotatePlan(theta, 128, 128, True, d)
    sub = rng.integers(0, 180, 300).astype(np.int32)                                        # repeats allowed
    g = torch.from_numpy(rng.standard_normal((3, 300, plan.PW)).astype(np.float32)).to(d)
    host, devi = plan.backward(g, angles_i=torch.from_numpy(sub)), plan.backward(g, angles_i=torch.from_numpy(sub).to(d))
    assert torch.equal(host, devi)
    geom = oracle.Geometry(128, 128, True)
    want = oracle.rotate_bwd_tfcompat(to_np(g), geom, oTinv(oracle, theta, plan)[sub], 0)
    assert rel_err(to_np(host), want) <= REL       # (300 angles incl. repeats: the gather adds them in index order, as the oracle does)


@pytest.mark.parametrize("subset", [True, False])
def test_calculate_log_prob_reduce_per_object(oracle, subset):
    """The drop-in call with reduce='per_object': values = the ordered sum of the unreduced call's log-probabilities, and its
    gradient = the unreduced call's gradient under the same per-object weights (one scaled backward launch either way)."""
    d = dev()
    rng = np.random.default_rng(8)
    B, N = 6, 128
    theta = phantoms.dense_theta(180)
    x0 = torch.from_numpy(rng.random((B, N, N, 1), dtype=np.float32)).to(d)
    mask = torch.from_numpy(rng.uniform(0.01, 0.1, (B, 180)).astype(np.float32)).to(d)
    meas = torch.from_numpy(rng.random((B, 180, 184), dtype=np.float32)).to(d)
    sub = rng.permutation(180)[:20] if subset else None
    w = torch.from_numpy(rng.standard_normal(B).astype(np.float32)).to(d)
    xa = x0.clone().requires_grad_(True)
    lp = cp.calculate_log_prob_M_given_R(xa, mask, meas, 1e4, 1.2e-7, theta=theta, angles_i=sub, pad=True)
    (lp.sum(dim=(1, 2, 3)) * w).sum().backward()
    xb = x0.clone().requires_grad_(True)
    sums = cp.calculate_log_prob_M_given_R(xb, mask, meas, 1e4, 1.2e-7, theta=theta, angles_i=sub, pad=True, reduce="per_object")
    assert sums.shape == (B,)
    np.testing.assert_array_equal(to_np(sums), oracle.loglik_object_sums(to_np(lp)[..., 0], 0))
    (sums * w).sum().backward()
    assert torch.equal(xa.grad, xb.grad)
    with pytest.raises(ValueError, match="reduce must be"):
        cp.calculate_log_prob_M_given_R(xb, mask, meas, 1e4, 1.2e-7, theta=theta, reduce="mean")


@pytest.mark.parametrize("shape,A,S,n", [((128, 128), 180, 10, 20), ((128, 128), 180, 1, 70), ((33, 47), 24, 5, 9),
                                         ((128, 128), 180, 50, 180), ((6, 5), 7, 3, 40)])
def test_angle_selecting_planned_backward(oracle, shape, A, S, n):
    """The bwd4 plan (one dword = the four rows a lane owns at one angle): a planned backward that takes any subset of the
    plan's angles (any order, repeats) -- the bits of the oracle on the gathered table and of the segment kernel."""
    d = dev()
    rng = np.random.default_rng(A + n)
    theta = phantoms.dense_theta(A) if A == 180 else rng.uniform(-3, 3, A).astype(np.float32)
    plan = RotatePlan(theta, shape[0], shape[1], True, d)
    sub = rng.integers(0, A, n).astype(np.int32)
    g = rng.standard_normal((S, n, plan.PW)).astype(np.float32)
    gt, ai = torch.from_numpy(g).to(d), torch.from_numpy(sub).to(d)
    scale = torch.from_numpy(rng.standard_normal(S).astype(np.float32)).to(d)
    got = plan.backward(gt, angles_i=ai)
    assert plan._bwd4_plan is not None
    want = oracle.rotate_bwd_tfcompat(g, oracle.Geometry(shape[0], shape[1], True), oTinv(oracle, theta, plan)[sub], 0)
    np.testing.assert_array_equal(to_np(got), want)
    for ns, bw in ((1, 1), (2, 4), (1, 4), (2, 2)):
        _lib.tune("BNS", ns), _lib.tune("BW", bw)
        assert torch.equal(plan.backward(gt, angles_i=ai), got), (ns, bw)
    _lib.tune("*")
    got_s = plan.backward(gt, angles_i=ai, scale=scale)
    plan._bwd4_plan, plan._want_bwd4 = None, False          # the segment kernel
    assert torch.equal(plan.backward(gt, angles_i=ai), got)
    assert torch.equal(plan.backward(gt, angles_i=ai, scale=scale), got_s)


def test_host_resident_angle_subsets_ride_the_launch_arguments(oracle, torch_node):
    """A subset that lives in host memory (the trainer draws it on the host) is carried in the kernel arguments of the compact
    forward and of the angle-selecting backward: same bits as the device-index launches, no upload; an index outside the
    angle list raises like the reference's tf.gather (device-resident indices cannot be read and are clamped)."""
    from ct_pvae_amd.forward_functions import as_angle_index
    import ct_pvae_amd.forward_functions as ff
    d = dev()
    rng = np.random.default_rng(21)
    S, N = 5, 128
    theta = phantoms.dense_theta(180)
    plan = RotatePlan(theta, N, N, True, d)
    assert plan.compact
    x = torch.from_numpy(rng.random((S, N, N), dtype=np.float32)).to(d)
    for n in (1, 20, 100, 256):
        sub = rng.integers(0, 180, n).astype(np.int32)
        host = as_angle_index(sub, d, keep_host=True)
        assert host.device.type == "cpu"
        on_dev = as_angle_index(sub, d)
        f_h, f_d = plan.forward(x, angles_i=host), plan.forward(x, angles_i=on_dev)
        assert torch.equal(f_h, f_d)
        g = torch.from_numpy(rng.standard_normal((S, n, plan.PW)).astype(np.float32)).to(d)
        assert torch.equal(plan.backward(g, angles_i=host), plan.backward(g, angles_i=on_dev))
    with pytest.raises(ValueError, match="outside this plan"):
        plan.forward(x, angles_i=as_angle_index(np.array([3, 180]), d, keep_host=True))
    with pytest.raises(ValueError, match="outside this plan"):
        plan.forward(x, angles_i=as_angle_index(np.array([-1, 2]), d, keep_host=True))
    # the public call: numpy subset (host path) == device tensor subset, values and gradients, both autograd nodes
    B = 4
    x0 = torch.from_numpy(rng.random((B, N, N, 1), dtype=np.float32)).to(d)
    mask = torch.from_numpy(rng.uniform(0.01, 0.1, (B, 180)).astype(np.float32)).to(d)
    meas = torch.from_numpy(rng.random((B, 180, 184), dtype=np.float32)).to(d)
    sub = rng.permutation(180)[:20]
    res = []
    for use_cpp in (True, False):
        ff.USE_CPP_NODE = use_cpp
        try:
            for idx in (sub, torch.from_numpy(sub).to(d)):
                for red in (None, "per_object"):
                    xa = x0.clone().requires_grad_(True)
                    lp = cp.calculate_log_prob_M_given_R(xa, mask, meas, 1e4, 1.2e-7, theta=theta, angles_i=idx, pad=True, reduce=red)
                    (lp.sum() if red is None else lp.sum()).backward()
                    res.append((red, lp.detach().clone(), xa.grad.clone()))
        finally:
            ff.USE_CPP_NODE = True
    for red, lp, gx in res:
        ref = next(r for r in res if r[0] == red)
        assert torch.equal(lp, ref[1]) and torch.equal(gx, ref[2])


# ---- round 3: gridrec, csrc/gridrec.hip against oracle/gridrec_oracle.c -----------------------------------------
@pytest.mark.parametrize("dy,dt,dx,grid,filt", [(50, 180, 184, None, "parzen"), (3, 20, 184, None, "ramlak"), (1, 7, 30, (24, 30), "shepp"),
                                                (4, 33, 94, None, "butterworth"), (2, 12, 16, None, "none"), (5, 45, 300, (256, 200), "hann"),
                                                (17, 12, 30, None, "cosine"), (43, 9, 16, None, "hamming"),      # 4 / 5 slice pairs per thread, ragged
                                                (3, 30, 128, None, "parzen"), (2, 11, 64, (64, 40), "shepp")])     # grid == padded row (round-3 ADVICE)
def test_gridrec_against_the_oracle(oracle, dy, dt, dx, grid, filt):
    """tomopy.recon(algorithm='gridrec') on the GPU: same tables (built on the host), same butterflies, the convolution gathered
    in gridrec.c's order -> the oracle's reconstruction to fp32 rounding (<= 1e-5 of its largest value; in practice the bits)."""
    d = dev()
    rng = np.random.default_rng(dy * 1000 + dt)
    if dx == 184:
        foam = phantoms.foam_batch(dy, 128, seed=1, supersample=2)
        theta = phantoms.dense_theta(180)[:dt] if dt < 180 else phantoms.dense_theta(180)
        data = np.ascontiguousarray(oracle.siddon_project(foam, theta.astype(np.float32), pad=True).transpose(1, 0, 2))
    else:
        theta = np.sort(rng.uniform(0, np.pi, dt)).astype(np.float32)
        data = rng.random((dy, dt, dx), dtype=np.float32)
    gx, gy = grid if grid else (dx, dx)
    want = oracle.gridrec(data, theta, filter_name=filt, ngridx=gx, ngridy=gy)
    from ct_pvae_amd.recon import recon
    got = recon(torch.from_numpy(data).to(d), theta, center=None, sinogram_order=True, algorithm="gridrec", filter_name=filt,
                num_gridx=gx, num_gridy=gy)
    assert got.shape == (dy, gx, gy)
    err = rel_err(to_np(got), want)
    print(f"gridrec {dy}x{dt}x{dx} {filt}: max rel-err {err:.2e}, {int((to_np(got) != want).sum())} of {want.size} values differ")
    assert err <= REL
    again = recon(torch.from_numpy(data).to(d), theta, center=None, sinogram_order=True, algorithm="gridrec", filter_name=filt,
                  num_gridx=gx, num_gridy=gy)
    assert torch.equal(again, got)                     # deterministic: a gather, no atomics
    if dx == 184 and dy == 50:
        # the default filter is tomopy's per-algorithm default, and projection order [angles][slices][dx] is accepted
        dflt = recon(torch.from_numpy(np.ascontiguousarray(data.transpose(1, 0, 2))).to(d), theta, algorithm="gridrec")
        assert torch.equal(dflt, got)
        # sanity pin: the independently written ramp-filtered back-projection of the same sinograms agrees after the affine
        # fit that the recalled normalisation and the dropped zero frequency call for (see tests/test_oracle.py)
        fbp = to_np(recon(torch.from_numpy(data).to(d), theta, sinogram_order=True, algorithm="fbp", filter_name="ramp"))[0].astype(np.float64)
        rl = to_np(recon(torch.from_numpy(data[:1]).to(d), theta, sinogram_order=True, algorithm="gridrec", filter_name="ramlak"))[0].astype(np.float64)
        A = np.stack([fbp.ravel(), np.ones(fbp.size)], 1)
        (a, b), *_ = np.linalg.lstsq(A, rl.ravel(), rcond=None)
        res = np.linalg.norm(rl - a * fbp - b) / np.linalg.norm(fbp)
        print(f"gridrec(ramlak) vs ramp-filtered back-projection: gain {a:.3f}, offset {b:.4f}, residual {res:.3f}")
        assert 1.0 < a < 1.3 and res < 0.25          # (a flipped, transposed or mis-centred reconstruction reads ~1)


def test_evaluate_sinogram_wrapper(oracle):
    """ctvae/helper_functions.py:433-475: three reconstructions, crop, two comparisons."""
    d = dev()
    foam = phantoms.foam_batch(1, 128, seed=2, supersample=2)
    theta = phantoms.dense_theta(180).astype(np.float32)
    sino = oracle.siddon_project(foam, theta, pad=True)[:, 0, :]          # [angles][P]
    rng = np.random.default_rng(0)
    mask = np.zeros(180, np.float32)
    mask[::9] = 1.0 / 20
    noisy = (sino * mask[:, None] * (1 + 0.02 * rng.standard_normal(sino.shape))).astype(np.float32)
    for alg in ("gridrec", "sirt"):
        pe, ne, r0, r1, r2 = cp.evaluate_sinogram(sino, sino * 0.98, noisy, torch.from_numpy(mask), theta, 128, 128, algorithm=alg,
                                                  verbose=False)
        assert r0.shape == r1.shape == r2.shape == (128, 128) and len(pe) == len(ne) == 3
        assert np.isfinite(pe + ne).all() and pe[1] > 0.9       # a 2 % rescaled sinogram reconstructs to nearly the same image
        if alg == "gridrec":
            assert pe[0] < ne[0] and pe[2] > ne[2]              # ... closer than the 20-angle noisy one


def test_operand_checks_see_a_tensor_whose_storage_was_swapped():
    """the operand checks must not trust a tensor OBJECT that passed before -- `t.data = other` puts a buffer of
    another size (or dtype) under the same object, and the kernels index by the plan's shapes."""
    d = dev()
    plan = RotatePlan(phantoms.dense_theta(180)[::9], 128, 128, True, d)
    x = torch.rand((4, 128, 128), device=d)
    out = nan_out((4, 20, plan.PW), d)
    plan.forward(x, out=out)
    x.data = torch.rand((4, 64, 64), device=d)             # same object, a quarter of the storage
    with pytest.raises(ValueError, match="img must be"):
        plan.forward(x, out=out)
    x.data = torch.rand((4, 128, 128), device=d, dtype=torch.float64)
    with pytest.raises(ValueError, match="img must be"):
        plan.forward(x, out=out)
    x.data = torch.rand((4, 128, 128), device=d)
    out.data = torch.empty((4, 10, plan.PW), device=d)     # the OUTPUT shrank: a launch would write past its end
    with pytest.raises(ValueError, match="out must be"):
        plan.forward(x, out=out)
    g = torch.rand((4, 20, plan.PW), device=d)
    gi = torch.empty((4, 128, 128), device=d)
    plan.backward(g, out=gi)
    gi.resize_(4, 64, 64)
    with pytest.raises(ValueError, match="out must be"):
        plan.backward(g, out=gi)


def test_bench_cold_inputs():
    """`bench.py --cold` cycles 96 distinct batches (more than the Infinity Cache) and says so in its config."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cold", "--steps", "192", "--warmup", "5", "--min-ms", "5",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["inputs"].startswith("cold: 96 distinct batches") and d["steps"] == 192 and d["value"] > 0


def test_hip_ray_driven_fbp_and_gridrec_against_the_scikit_image_fixture(golden_dir):
    """The HIP kernels -- no oracle in this test -- against scikit-image 0.18.3's radon / iradon (tests/golden/
    skimage_crosscheck.npz; bounds and what they mean: tests/test_oracle.py, same fixture)."""
    from tests.test_oracle import _skimage_mapped
    from ct_pvae_amd.recon import recon
    d = dev()
    z = np.load(os.path.join(golden_dir, "skimage_crosscheck.npz"))
    img, theta = z["img"], np.deg2rad(z["theta_deg"])
    sino = cp.create_sinograms(torch.from_numpy(img[None]).to(d), theta, pad=True)          # [1][A][184]
    ours = to_np(sino)[0].astype(np.float64)
    sk = z["sk_sino"].T.astype(np.float64)
    m = _skimage_mapped(ours, theta)
    l2, worst = np.linalg.norm(m - sk) / np.linalg.norm(sk), np.abs(m - sk).max() / np.abs(sk).max()
    assert l2 < 0.008 and worst < 0.05, (l2, worst)
    rec = to_np(cp.iradon(sino.to(torch.float64), theta, 128, 128, z["ramp184"]))[0]
    assert np.linalg.norm(rec - img) / np.linalg.norm(img) < 0.2
    assert np.linalg.norm(rec - z["sk_rec"]) / np.linalg.norm(z["sk_rec"]) < 0.18
    g = to_np(recon(sino, theta, sinogram_order=True, algorithm="gridrec", filter_name="ramlak"))[0][28:156, 28:156].astype(np.float64)
    A = np.stack([img.ravel().astype(np.float64), np.ones(img.size)], 1)
    (a, b), *_ = np.linalg.lstsq(A, g.ravel(), rcond=None)
    assert 1.05 < a < 1.2 and np.linalg.norm((g - b) / a - z["sk_rec"]) / np.linalg.norm(z["sk_rec"]) < 0.12
    print(f"HIP vs scikit-image: radon L2 {l2:.4f} / worst {worst:.4f}; gridrec gain {a:.3f}")


# ---- round 3: the tiled forward through compact tile plans ----------------------------------------------------------------------
@pytest.mark.parametrize("shape,A,S", [((512, 512), 90, 5), ((300, 260), 17, 3), ((512, 512), 12, 1), ((257, 400), 9, 6)])
def test_tiled_forward_through_compact_tile_plans(oracle, shape, A, S):
    """Slices larger than LDS: the compact tile plans (first tap inside the tile + 2 bits per row, per tile / angle / ray slot)
    give the partial sums of the direct tiled kernel -- the tiled oracle's bits -- for 4, 2 and 1 slices per workgroup."""
    d = dev()
    rng = np.random.default_rng(shape[0] + A)
    img = rng.random((S,) + shape, dtype=np.float32)
    theta = rng.uniform(-3.0, 6.0, A).astype(np.float32)
    theta[:2] = [0.0, np.pi / 2]
    plan = RotatePlan(theta, shape[0], shape[1], True, d)
    assert plan.tiled and plan._tplan is not None
    x = torch.from_numpy(img).to(d)
    got = plan.forward(x)
    want = oracle.rotate_fwd_tiled(img, oracle.Geometry(shape[0], shape[1], True), oT(oracle, theta, plan), tile=oracle.tile_shape(*shape))
    np.testing.assert_array_equal(to_np(got), want)
    direct = RotatePlan(theta, shape[0], shape[1], True, d, plan_format="u16")     # the direct tiled kernel
    assert direct.tiled and direct._tplan is None
    assert torch.equal(direct.forward(x), got)
    for ns, G in ((1, 1), (2, 2), (4, 3), (4, 1)):
        _lib.tune("TILED_NS", ns), _lib.tune("TILED_G", G)
        assert torch.equal(plan.forward(x), got), (ns, G)
        # tasks = four 16-slot bands of the plan's sorted list (the default) against (angle, 64-slot block) tasks: which rays
        # ride in one wave changes no ray's sum
        with _lib.tuned("TILED_SORT", 0):
            assert torch.equal(plan.forward(x), got), (ns, G, "unsorted tasks")
    _lib.tune("*")
    # round 4 (built, measured equal in time, kept behind a knob): lanes that walk TWO rays back to back -- band u, then band
    # nq - 1 - u of the same angle -- through their own plan sections
    with _lib.tuned("TILED_PAIR", 1):
        pairs = RotatePlan(theta, shape[0], shape[1], True, d)
        assert pairs._tplan is not None and pairs._tplan.numel() > plan._tplan.numel()
        for ns in (4, 2, 1):
            _lib.tune("TILED_NS", ns)
            assert torch.equal(pairs.forward(x), got), (ns, "paired bands")
    _lib.tune("*")
    mask = torch.from_numpy(rng.uniform(0.01, 0.1, (S, A)).astype(np.float32)).to(d)
    meas = torch.from_numpy(rng.random((S, A, plan.PW), dtype=np.float32)).to(d)
    pnm = torch.tensor([1e4], device=d)
    a3, b3 = plan.forward_loglik(x, mask, meas, pnm, 1.2e-7, with_dlp=True), direct.forward_loglik(x, mask, meas, pnm, 1.2e-7, with_dlp=True)
    assert all(torch.equal(u, v) for u, v in zip(a3, b3))
    # round 4: the tile shape is a rule of the library (equal rows of tiles, at most 128 tall), reported by ctpvae_rotate_tile_shape.
    # Round 5: other heights exist in timing builds only (-DCTPVAE_TUNE_TILED_TH, tools/time_tile_heights.py) -- a switch that
    # changes result bits is not part of the product library, which ignores the knob:
    with _lib.tuned("TILED_TH", 96):
        assert _lib.tile_shape(*shape) == (-(-shape[0] // -(-shape[0] // 128)), 64)      # ceil(H / ceil(H / 128)) whatever the knob
        assert torch.equal(RotatePlan(theta, shape[0], shape[1], True, d).forward(x), got)


@pytest.mark.parametrize("fmt", ["auto", "u16"])
def test_per_object_loglik_sums_of_a_tiled_geometry(oracle, fmt):
    """Config 5's geometry (512 x 512, tiled): the reduce pass of the tiled forward reduces the log-probabilities per object too
    (partition 1 of the fixed order: contiguous 64-bin blocks) -- the ordered sum of the unreduced launch's log-probabilities, bit
    for bit, and the same d lp / d ray-sum; through the public call with reduce='per_object' as well."""
    d = dev()
    rng = np.random.default_rng(2)
    S, N, A = 5, 512, 12
    theta = (np.pi * np.arange(A) / A).astype(np.float32)
    plan = RotatePlan(theta, N, N, True, d, plan_format=fmt)
    assert plan.tiled and (plan._tplan is not None) == (fmt == "auto")
    x = torch.from_numpy(rng.random((S, N, N), dtype=np.float32)).to(d)
    mask = torch.from_numpy(rng.uniform(0.01, 0.1, (S, A)).astype(np.float32)).to(d)
    meas = torch.from_numpy(rng.random((S, A, plan.PW), dtype=np.float32)).to(d)
    pnm = torch.tensor([1e4], device=d)
    sino, lp, dlp = plan.forward_loglik(x, mask, meas, pnm, 1.2e-7, with_dlp=True)
    sums, dlp2 = plan.forward_loglik_sums(x, mask, meas, pnm, 1.2e-7)
    np.testing.assert_array_equal(to_np(sums), oracle.loglik_object_sums(to_np(lp), 1))
    assert torch.equal(dlp2, dlp)
    if fmt == "auto":
        xa = x[..., None].clone().requires_grad_(True)
        got = cp.calculate_log_prob_M_given_R(xa, mask, meas, pnm, 1.2e-7, theta=theta, pad=True, reduce="per_object")
        assert torch.equal(got, sums)
        got.sum().backward()
        assert torch.isfinite(xa.grad).all() and xa.grad.abs().max() > 0


def test_round3_operators_against_golden(golden_dir):
    """The HIP gridrec and the same-order object sums against tests/golden/round3.npz directly (no oracle in this test)."""
    from ct_pvae_amd.recon import recon
    d = dev()
    z = np.load(os.path.join(golden_dir, "round3.npz"))
    data, theta = torch.from_numpy(z["g_data"]).to(d), z["g_theta"]
    np.testing.assert_array_equal(to_np(cp.create_sinograms(torch.from_numpy(z["g_img"]).to(d), theta, pad=True)), z["g_data"])
    assert rel_err(to_np(recon(data, theta, sinogram_order=True, algorithm="gridrec")), z["g_parzen"]) <= REL
    assert rel_err(to_np(recon(data, theta, sinogram_order=True, algorithm="gridrec", filter_name="ramlak", num_gridx=40,
                               num_gridy=44)), z["g_ramlak_40x44"]) <= REL
    lib = _lib.load()
    lp = torch.from_numpy(z["s_lp"]).to(d)
    out = nan_out(lp.shape[0], d)
    for part, key in ((0, "s_sums_bands"), (1, "s_sums_blocks")):
        assert lib.ctpvae_loglik_object_sums_f32(lp.data_ptr(), lp.shape[0], lp.shape[1], lp.shape[2], part, out.data_ptr(), None) == 0
        np.testing.assert_array_equal(to_np(out), z[key])


# ---- round 4: the ordered sum of the partials inside the projector launch ------------------------------------------------------
@pytest.mark.parametrize("shape,A,S,sub", [((128, 128), 180, 7, True), ((128, 128), 180, 50, False), ((128, 128), 20, 1, False),
                                           ((300, 260), 17, 11, False), ((512, 512), 9, 5, False)])
def test_per_object_sums_inside_the_launch(oracle, shape, A, S, sub):
    """Knob FOLD_SUMS = 1 (round 4; built, measured slower than the second launch, kept for the record): forward_loglik_sums as
    ONE projector launch (+ the tiled geometry's reduce pass) -- the workgroup that finishes a slice last adds the slice's
    partials in the library's fixed order.  The reader fixes the order, so the sums are the bits of the default's second launch
    and of the oracle's statement of the order; launch after launch (the arrival counters are left zero), for odd batches, angle
    subsets, and batches cut into chunks."""
    d = dev()
    rng = np.random.default_rng(S * 7 + A)
    H, W = shape
    img = rng.random((S, H, W), dtype=np.float32)
    theta = phantoms.dense_theta(180)[:: 180 // A][:A] if A in (20, 180) else np.sort(rng.uniform(0, np.pi, A)).astype(np.float32)
    plan = RotatePlan(theta, H, W, True, d, plan_format="compact")
    x = torch.from_numpy(img).to(d)
    mask = torch.from_numpy(rng.uniform(0.01, 0.1, (S, A)).astype(np.float32)).to(d)
    meas = torch.from_numpy(rng.random((S, A, plan.PW), dtype=np.float32) * 3).to(d)
    pnm = torch.tensor([1e4], device=d)
    ai = torch.from_numpy(rng.permutation(A)[:20].astype(np.int32)) if sub else None       # host-resident subset
    kw = dict(angles_i=ai, dense_inputs=sub)
    _, lp, dlp = plan.forward_loglik(x, mask, meas, pnm, 1.2e-7, with_dlp=True, **kw)
    want = oracle.loglik_object_sums(to_np(lp), 1 if plan.tiled else 0)
    two, _ = plan.forward_loglik_sums(x, mask, meas, pnm, 1.2e-7, **kw)            # the default: a second, tiny launch
    np.testing.assert_array_equal(to_np(two), want)
    with _lib.tuned("FOLD_SUMS", 1):
        for rep in range(3):                      # the counters come back to zero
            sums, dlp2 = plan.forward_loglik_sums(x, mask, meas, pnm, 1.2e-7, **kw)
            np.testing.assert_array_equal(to_np(sums), want, err_msg=f"launch {rep}")
            assert torch.equal(dlp2, dlp)
        if not sub:
            _lib.tune("MAX_SLICES", 4 if plan.tiled else 3)     # the batch in chunks (tiled: whole slice quads)
            chunked, _ = plan.forward_loglik_sums(x, mask, meas, pnm, 1.2e-7)
            np.testing.assert_array_equal(to_np(chunked), want)
    ws = next(iter(plan._part_ws.values()))
    assert int(ws[-S:].view(torch.int32).abs().sum()) == 0          # every arrival counter is back at zero


# ---- round 4: the TV stand-in as two projector launches per iteration, against its restatement ----------------------------------
@pytest.mark.parametrize("oy,N,dt,iters,lam", [(3, 32, 24, 4, 0.05), (9, 48, 30, 3, 1.0), (1, 20, 9, 5, 0.3)])
def test_tv_standin_iteration_against_its_restatement(oracle, oy, N, dt, iters, lam):
    """recon(algorithm='tv') is a flagged STAND-IN for tomopy's tv.c (preconditioned Chambolle-Pock on the TomoPy-style projector
    pair).  Since round 4 an iteration is two projector launches whose stores do the dual and primal steps; oracle.tv_standin
    states the same iteration operation by operation -- the kernels give its values (<= 1e-5; in practice the bits), it still
    warns that it is not TomoPy's algorithm, and it still denoises (lower TV than the back-projection it starts from)."""
    import importlib
    import warnings
    recon_mod = importlib.import_module("ct_pvae_amd.recon")      # (the package exports the function under the same name)
    d = dev()
    rng = np.random.default_rng(oy + N)
    img = phantoms.foam_batch(oy, N, seed=3, supersample=2)
    theta = np.sort(rng.uniform(0, np.pi, dt)).astype(np.float32)
    data = np.ascontiguousarray(oracle.siddon_project(img, theta, pad=True).transpose(1, 0, 2))
    data = (data + 0.05 * rng.standard_normal(data.shape)).astype(np.float32)
    want = oracle.tv_standin(data, theta, num_iter=iters, lam=lam)
    recon_mod._TV_WARNED = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = recon_mod.recon(torch.from_numpy(data).to(d), theta, sinogram_order=True, algorithm="tv", num_iter=iters,
                              reg_par=np.array([lam]))
    assert any("STAND-IN" in str(x.message) for x in w)
    err = rel_err(to_np(got), want)
    print(f"tv stand-in {oy}x{N} {iters} iterations: max rel-err {err:.2e}, {int((to_np(got) != want).sum())} of {want.size} values differ")
    assert err <= REL
    again = recon_mod.recon(torch.from_numpy(data).to(d), theta, sinogram_order=True, algorithm="tv", num_iter=iters, reg_par=np.array([lam]))
    assert torch.equal(again, got)


# ---- round 4: the dispatch matrix, enumerated ------------------------------------------------------------------------------------
def _matrix_variants(tiled):
    """(forward format, subset kind, epilogue / node, backward path) -- every combination the host code can reach for a geometry"""
    fmts = ("tile_plan", "tile_direct") if tiled else ("u16", "compact")
    out = []
    for fmt in fmts:
        for sel in ("dense", "device", "host"):
            for epi in ("plain", "lp_py", "lp_cpp", "sums"):
                if tiled and epi == "lp_cpp":
                    continue                       # the C++ node serves planned geometries only
                bwds = ("stepped", "segment") if tiled else (("planned", "stepped", "segment") if sel == "dense" else ("bwd4", "seg_sel"))
                for bwd in bwds:
                    out.append((fmt, sel, epi, bwd))
    return out


@pytest.mark.parametrize("shape,A,S", [((120, 140), 11, 5), ((512, 512), 5, 3)])
def test_dispatch_matrix_against_the_oracle(oracle, torch_node, shape, A, S):
    """Every reachable (forward plan format x angle-subset kind x epilogue x autograd node x backward path) combination of the
    nearest / tf_compat operator pair on one geometry that fits LDS and one that does not, each forced explicitly, against the
    oracle: ray-sums bit for bit, log-probabilities <= 1e-5, per-object sums = the oracle's ordered sum of the kernel's
    log-probabilities, image gradients = (per-object weight) x the oracle's back-projection of the kernel's d lp / d ray-sum, bit
    for bit -- and therefore all combinations equal to each other.  (Round 3's two dispatch bugs -- a plan of the wrong layout
    handed to the C++ likelihood node in step-plan mode; a step plan overflowing near 90 degrees -- were found by random soaks;
    this is the test that would have failed.)"""
    from ct_pvae_amd import helper_functions as hf
    from ct_pvae_amd.forward_functions import _RotateProject, _LAYOUT_VAE
    d = dev()
    H, W = shape
    tiled = H * W > 200 * 200
    rng = np.random.default_rng(H + A)
    img = rng.random((S, H, W), dtype=np.float32)
    theta = np.sort(rng.uniform(0.0, np.pi, A)).astype(np.float32)
    theta[0] = 0.0
    n_sub = 4
    sub = np.array([A - 1, 2, 0, 2][:n_sub], np.int32)               # any order, a repeat
    geom = oracle.Geometry(H, W, True)
    mk = {"u16": dict(plan_format="u16"), "compact": dict(plan_format="compact"), "tile_plan": {}, "tile_direct": dict(plan_format="u16")}
    x0 = torch.from_numpy(img[..., None]).to(d)
    pnm, eps = torch.tensor([1e4], device=d), 1.2e-7
    w = torch.from_numpy(rng.uniform(0.5, 2.0, S).astype(np.float32)).to(d)
    ref = {}

    def oracle_for(sel, plan):
        key = sel is not None
        if key not in ref:
            T = oT(oracle, theta, plan)
            Ts = T if sel is None else T[sub]
            sino = (oracle.rotate_fwd_tiled(img, geom, Ts, tile=oracle.tile_shape(geom.H, geom.W)) if tiled else oracle.rotate_fwd(img, geom, Ts, 0))
            n = Ts.shape[0]
            mask = rng.uniform(0.01, 0.1, (S, A)).astype(np.float32)
            meas = rng.random((S, A, plan.PW), dtype=np.float32) * 3
            g = rng.standard_normal((S, n, plan.PW)).astype(np.float32)
            msub, xsub = (mask, meas) if sel is None else (mask[:, sub], meas[:, sub])
            lp = oracle.loglik(sino, msub, xsub, 1e4, eps)
            Tinv = oTinv(oracle, theta, plan)
            Tinv = Tinv if sel is None else Tinv[sub]
            ref[key] = dict(sino=sino, lp=lp, mask=mask, meas=meas, g=g, Tinv=Tinv,
                            gimg_plain=oracle.rotate_bwd_tfcompat(g, geom, Tinv, 0))
        return ref[key]

    seen = set()
    for fmt, selk, epi, bwd in _matrix_variants(tiled):
        _lib.tune("*")
        plan = RotatePlan(theta, H, W, True, d, **mk[fmt])
        if tiled:
            assert plan.tiled and (plan._tplan is not None) == (fmt == "tile_plan")
        else:
            assert plan.planned[0] and plan.compact == (fmt == "compact")
        sel = None if selk == "dense" else (torch.from_numpy(sub).to(d) if selk == "device" else torch.from_numpy(sub))
        R = oracle_for(sel, plan)
        # force the backward path
        if bwd == "planned":
            plan.backward_uses_step_plan, plan.backward_uses_plan = (lambda S_: False), (lambda S_: True)
        elif bwd == "stepped":
            assert plan._step_plan is not None
            plan.backward_uses_step_plan, plan.backward_uses_plan = (lambda S_: True), (lambda S_: False)
        elif bwd == "segment":
            plan._step_plan = None
            plan.backward_uses_step_plan, plan.backward_uses_plan = (lambda S_: False), (lambda S_: False)
        elif bwd == "seg_sel":
            plan._want_bwd4, plan._bwd4_plan = False, None
        elif bwd == "bwd4":
            assert plan._get_bwd4_plan() is not None
        tag = (fmt, selk, epi, bwd)
        mask, meas = torch.from_numpy(R["mask"]).to(d), torch.from_numpy(R["meas"]).to(d)
        run_plan, run_sel, run_mask, run_meas = plan, sel, mask, meas
        if sel is not None and tiled:            # what calculate_log_prob_M_given_R does for a tiled geometry: gather, subset()
            idx = torch.from_numpy(sub).to(d).long()
            run_plan, run_sel = plan.subset(sel), None
            run_mask, run_meas = mask.index_select(1, idx).contiguous(), meas.index_select(1, idx).contiguous()
            if bwd == "segment":
                run_plan._step_plan = None
        x = x0.clone().requires_grad_(True)
        if epi == "plain":
            g = torch.from_numpy(R["g"]).to(d)
            if sel is None:
                out = plan.project_vae_cpp(x) if bwd == "planned" else None          # the C++ node (planned backward only) ...
                node = "cpp" if out is not None else "py"
                if out is None:
                    out = _RotateProject.apply(x, plan, _LAYOUT_VAE)                  # ... else the Python node
                out.backward(g[..., None])
                sino, gimg = out.detach()[..., 0], x.grad[..., 0]
            else:
                node = "raw"
                sino = run_plan.forward(x0[..., 0], angles_i=run_sel)
                gimg = run_plan.backward(g, angles_i=run_sel)
            np.testing.assert_array_equal(to_np(sino), R["sino"], err_msg=str(tag))
            np.testing.assert_array_equal(to_np(gimg), R["gimg_plain"], err_msg=str(tag))
            seen.add(tag + (node,))
            continue
        # the kernel's own d lp / d ray-sum (the backward's operand) from the raw call of THIS plan: the gradient reference
        _, lp_k, dlp_k = run_plan.forward_loglik(x0[..., 0], run_mask, run_meas, pnm, eps, with_dlp=True, angles_i=run_sel,
                                                 dense_inputs=run_sel is not None)
        assert rel_err(to_np(lp_k), R["lp"]) <= REL, tag
        want_g = to_np(w)[:, None, None] * oracle.rotate_bwd_tfcompat(to_np(dlp_k), geom, R["Tinv"], 0)
        if epi == "sums":
            sums = hf._ProjectLogLikSums.apply(x, run_plan, run_mask, run_meas, pnm, eps, run_sel)
            np.testing.assert_array_equal(to_np(sums), oracle.loglik_object_sums(to_np(lp_k), 1 if tiled else 0), err_msg=str(tag))
            (sums * w).sum().backward()
        else:
            if epi == "lp_cpp":
                lp4 = run_plan.loglik_vae_cpp(x, run_mask, run_meas, pnm, eps, run_sel)
                assert lp4 is not None, tag
            else:
                lp4 = hf._ProjectLogLik.apply(x, run_plan, run_mask, run_meas, pnm, eps, run_sel)
            assert torch.equal(lp4.detach()[..., 0], lp_k), tag
            (lp4.sum(dim=(1, 2, 3)) * w).sum().backward()                            # per-object weights: the scaled backward
        np.testing.assert_array_equal(to_np(x.grad[..., 0]), want_g.astype(np.float32), err_msg=str(tag))
        seen.add(tag)
    _lib.tune("*")
    assert len(seen) >= len(_matrix_variants(tiled))
