"""The BILINEAR quadrants of the rotate projector (project_tf_low_mem, ctvae/forward_functions.py:69-77; round 5) against the
CPU oracle, on a real MI355X, to the depth the nearest path is tested: random geometries, unpadded canvases, axis-aligned
angles, every slices-per-cell variant, tiles, and the old direct kernels as a second implementation.

Bars: forward and TensorFlow-compatible backward BIT-EXACT (same expression per sample, same order of the sums); the exact
adjoint <= 1e-5 of the oracle's scatter, equal bits run to run, and still the transpose."""
import os

import numpy as np
import pytest
import torch

from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan

pytestmark = pytest.mark.gpu
REL = 1e-5


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def to_np(t):
    return t.detach().cpu().numpy()


def rel_err(got, want):
    want = np.asarray(want, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


def oT(oracle, theta, plan):
    return oracle.rotate_transforms(np.asarray(theta, dtype=np.float32), plan.PH, plan.PW)


def fuzz_cases(rng, n):
    """(H, W, pad, theta, S): odd sizes, one-pixel slices, unpadded canvases, axis-aligned angles, 1..7 slices."""
    fixed = [(1, 1), (1, 77), (93, 1), (3, 2), (2, 2), (128, 128), (129, 127), (64, 160)]
    for case in range(n):
        H, W = fixed[case] if case < len(fixed) else (int(rng.integers(1, 150)), int(rng.integers(1, 150)))
        pad, A, S = bool(rng.integers(0, 2)), int(rng.integers(1, 25)), int(rng.integers(1, 8))
        theta = rng.uniform(-2 * np.pi, 2 * np.pi, A)
        if case % 3 == 0:
            theta[: min(A, 4)] = [0.0, np.pi / 2, np.pi, -np.pi / 2][: min(A, 4)]
        yield case, H, W, pad, theta, S


def test_bilinear_forward_random_geometries(oracle):
    d = dev()
    rng = np.random.default_rng(int(os.environ.get("CTPVAE_FUZZ_SEED", 20261005)))
    for case, H, W, pad, theta, S in fuzz_cases(rng, int(os.environ.get("CTPVAE_FUZZ_CASES", 28))):
        img = rng.standard_normal((S, H, W)).astype(np.float32)
        geom = oracle.Geometry(H, W, pad)
        plan = RotatePlan(theta, H, W, pad, d, interp="bilinear")
        want = oracle.rotate_fwd(img, geom, oT(oracle, theta, plan), 1)
        x = torch.from_numpy(img).to(d)
        msg = f"case {case}: {H}x{W} pad={pad} A={len(theta)} S={S}"
        np.testing.assert_array_equal(to_np(plan.forward(x)), want, err_msg="fwd " + msg)
        for ns in (1, 2, 4):                       # slices per LDS cell, forced
            with _lib.tuned("BNS", ns):
                np.testing.assert_array_equal(to_np(plan.forward(x)), want, err_msg=f"fwd BNS={ns} " + msg)
                for bsort in (0, 1):               # ... with (angle, block) tasks and with length-sorted band tasks
                    with _lib.tuned("BSORT", bsort):
                        np.testing.assert_array_equal(to_np(plan.forward(x)), want, err_msg=f"fwd BNS={ns} BSORT={bsort} " + msg)
                for rsplit in (0, 1):              # ... (angle, block) tasks of 64 rays, and of 32 rays with four rows of a ray per step
                    with _lib.tuned("BSORT", 0), _lib.tuned("BRSPLIT", rsplit):
                        np.testing.assert_array_equal(to_np(plan.forward(x)), want, err_msg=f"fwd BNS={ns} BRSPLIT={rsplit} " + msg)
        with _lib.tuned("NO_PLAN", 1):             # round 1's direct kernel: a second implementation of the same sums
            np.testing.assert_array_equal(to_np(plan.forward(x)), want, err_msg="fwd direct " + msg)


def test_bilinear_slice_independence_and_launch_shapes(oracle):
    """A slice's sinogram does not depend on what it is batched with (1 / 2 / 3 / 5 / 9 slices: singles, pairs, quads and their
    half-empty remainders), nor on the number of task groups or waves."""
    d = dev()
    rng = np.random.default_rng(3)
    theta = phantoms.dense_theta(180)[::12]
    img = rng.random((9, 128, 128)).astype(np.float32)
    geom = oracle.Geometry(128, 128, True)
    plan = RotatePlan(theta, 128, 128, True, d, interp="bilinear")
    want = oracle.rotate_fwd(img, geom, oT(oracle, theta, plan), 1)
    x = torch.from_numpy(img).to(d)
    for S in (1, 2, 3, 5, 9):
        np.testing.assert_array_equal(to_np(plan.forward(x[:S].contiguous())), want[:S], err_msg=f"S={S}")
    for G, waves in ((1, 16), (3, 5), (7, 2), (2, 1)):
        for bsort in (0, 1):   # (sorted bands: band b of every angle belongs to task group b mod G)
            with _lib.tuned("BW", G), _lib.tuned("WAVES", waves), _lib.tuned("BSORT", bsort):
                np.testing.assert_array_equal(to_np(plan.forward(x)), want, err_msg=f"G={G} waves={waves} BSORT={bsort}")
        for rsplit in (0, 1):  # (row-split walks: 32-ray tasks, lanes 32-63 two rows further on)
            with _lib.tuned("BW", G), _lib.tuned("WAVES", waves), _lib.tuned("BSORT", 0), _lib.tuned("BRSPLIT", rsplit):
                np.testing.assert_array_equal(to_np(plan.forward(x)), want, err_msg=f"G={G} waves={waves} BRSPLIT={rsplit}")


@pytest.mark.parametrize("H,W,pad", [(65, 40, False), (66, 31, False), (67, 128, False), (3, 50, False), (2, 2, False), (33, 47, True), (128, 128, True)])
def test_bilinear_row_split_walks(oracle, H, W, pad):
    """Row-split walks (launches of few tasks per workgroup: lanes 32-63 of a wave carry the rays of lanes 0-31 two rows further on,
    the sums kept in ray-row order through v_permlane32_swap) against the plain walk and the oracle: canvas heights that leave 0, 1,
    2 and 3 rows behind the steps of four, canvases shorter than one step, 1 / 2 / 4 slices per cell, axis-aligned angles."""
    d = dev()
    rng = np.random.default_rng(H * 1000 + W)
    theta = np.concatenate([rng.uniform(-np.pi, np.pi, 5), [0.0, np.pi / 2, np.pi / 4]])
    img = rng.standard_normal((5, H, W)).astype(np.float32)
    geom = oracle.Geometry(H, W, pad)
    plan = RotatePlan(theta, H, W, pad, d, interp="bilinear")
    want = oracle.rotate_fwd(img, geom, oT(oracle, theta, plan), 1)
    x = torch.from_numpy(img).to(d)
    for ns in (1, 2, 4):
        for G in (-1, 1, 3):
            with _lib.tuned("BNS", ns), _lib.tuned("BW", G), _lib.tuned("BSORT", 0), _lib.tuned("BRSPLIT", 1):
                np.testing.assert_array_equal(to_np(plan.forward(x)), want, err_msg=f"BNS={ns} G={G}")


def test_bilinear_forward_at_many_angles(oracle):
    """180 angles at 128 x 128: the launch the library itself runs with length-sorted band tasks (several chunks of bands per
    workgroup, interiors of many rows) -- and forced onto (angle, block) tasks -- against the oracle, every bit."""
    d = dev()
    rng = np.random.default_rng(180)
    theta = phantoms.dense_theta(180)
    img = rng.standard_normal((7, 128, 128)).astype(np.float32)
    plan = RotatePlan(theta, 128, 128, True, d, interp="bilinear")
    want = oracle.rotate_fwd(img, oracle.Geometry(128, 128, True), oT(oracle, theta, plan), 1)
    x = torch.from_numpy(img).to(d)
    np.testing.assert_array_equal(to_np(plan.forward(x)), want)
    for bsort, G in ((0, -1), (1, 1), (1, 5), (0, 3)):
        with _lib.tuned("BSORT", bsort), _lib.tuned("BW", G):
            np.testing.assert_array_equal(to_np(plan.forward(x)), want, err_msg=f"BSORT={bsort} G={G}")


@pytest.mark.parametrize("S,A", [(151, 180), (300, 20), (257, 45)])
def test_bilinear_forward_long_launches(oracle, S, A):
    """Batches whose workgroups come in several rounds: the task groups the library picks for whole rounds (and their snake deal /
    sorted bands) against one group per class and against the oracle on a few slices -- every output NaN before the launch."""
    d = dev()
    rng = np.random.default_rng(S + A)
    theta = phantoms.dense_theta(180)[:: 180 // A][:A]
    img = rng.standard_normal((S, 128, 128)).astype(np.float32)
    plan = RotatePlan(theta, 128, 128, True, d, interp="bilinear")
    x = torch.from_numpy(img).to(d)
    got = plan.forward(x)
    assert not bool(torch.isnan(got).any())
    with _lib.tuned("BW", 1):
        assert torch.equal(plan.forward(x), got)
    pick = [0, S // 2, S - 1]
    np.testing.assert_array_equal(to_np(got[pick]), oracle.rotate_fwd(img[pick], oracle.Geometry(128, 128, True), oT(oracle, theta, plan), 1))


@pytest.mark.parametrize("H,W,S,A", [(512, 512, 3, 6), (300, 200, 5, 4), (190, 260, 2, 5)])
def test_bilinear_tiles_against_the_tiled_oracle(oracle, H, W, S, A):
    """Slices larger than LDS: tiles with a one-pixel halo; a sample belongs to the tile of its floor tap, the tiles' partial
    sums are added in ascending order -- oracle_rotate_fwd_tiled(interp = 1) with the shape the library reports."""
    d = dev()
    rng = np.random.default_rng(H + W)
    theta = np.concatenate([[0.0, np.pi / 2], rng.uniform(0, np.pi, A - 2)])
    img = rng.random((S, H, W)).astype(np.float32)
    geom = oracle.Geometry(H, W, True)
    plan = RotatePlan(theta, H, W, True, d, interp="bilinear")
    shape = oracle.tile_shape(H, W, 1)         # the checker's own statement of the rule ...
    assert plan.tiled and _lib.tile_shape(H, W, 1) == shape == (-(-H // -(-H // 96)), 64)   # ... which the library reports too
    T = oT(oracle, theta, plan)
    got = to_np(plan.forward(torch.from_numpy(img).to(d)))
    np.testing.assert_array_equal(got, oracle.rotate_fwd_tiled(img, geom, T, shape, interp=1))
    assert rel_err(got, oracle.rotate_fwd(img, geom, T, 1)) <= REL
    for ns in (1, 2):
        with _lib.tuned("BNS", ns):
            np.testing.assert_array_equal(to_np(plan.forward(torch.from_numpy(img).to(d))), got, err_msg=f"BNS={ns}")


def test_bilinear_forced_tiles_on_small_unpadded_slices(oracle):
    """TILED_FORCE cuts slices that fit LDS into tiles: ragged edge tiles, one-row / one-column tiles, unpadded canvases."""
    d = dev()
    rng = np.random.default_rng(11)
    for H, W, pad in ((150, 100, True), (97, 65, False), (130, 129, False), (64, 64, True), (1, 200, True)):
        theta = rng.uniform(-np.pi, np.pi, 5)
        img = rng.standard_normal((3, H, W)).astype(np.float32)
        geom = oracle.Geometry(H, W, pad)
        with _lib.tuned("TILED_FORCE", 1):
            plan = RotatePlan(theta, H, W, pad, d, interp="bilinear")
            shape = oracle.tile_shape(H, W, 1)
            assert _lib.tile_shape(H, W, 1) == shape and plan.tiled
            got = to_np(plan.forward(torch.from_numpy(img).to(d)))
            for bsort in (0, 1):                   # both task forms of the tile kernel
                with _lib.tuned("BSORT", bsort):
                    np.testing.assert_array_equal(to_np(plan.forward(torch.from_numpy(img).to(d))), got, err_msg=f"BSORT={bsort}")
        np.testing.assert_array_equal(got, oracle.rotate_fwd_tiled(img, geom, oT(oracle, theta, plan), shape, interp=1),
                                      err_msg=f"{H}x{W} pad={pad} tiles {shape}")


def oTinv(oracle, theta, plan):
    return oracle.invert_transforms(oT(oracle, theta, plan))


def test_bilinear_tfcompat_backward_random_geometries(oracle):
    """What tf.GradientTape computes for the bilinear projector (a4 with interpolation BILINEAR): bit-exact, for every
    slices-per-cell / rows-per-lane variant of the segment kernel and for round 1's whole-row kernel."""
    d = dev()
    rng = np.random.default_rng(int(os.environ.get("CTPVAE_FUZZ_SEED", 20261006)))
    for case, H, W, pad, theta, S in fuzz_cases(rng, int(os.environ.get("CTPVAE_FUZZ_CASES", 28))):
        geom = oracle.Geometry(H, W, pad)
        plan = RotatePlan(theta, H, W, pad, d, interp="bilinear")
        g = rng.standard_normal((S, len(theta), geom.PW)).astype(np.float32)
        want = oracle.rotate_bwd_tfcompat(g, geom, oTinv(oracle, theta, plan), 1)
        gt = torch.from_numpy(g).to(d)
        msg = f"case {case}: {H}x{W} pad={pad} A={len(theta)} S={S}"
        np.testing.assert_array_equal(to_np(plan.backward(gt)), want, err_msg="bwd " + msg)
        for ns, ppt in ((1, 4), (1, 8), (2, 4), (2, 8), (4, 4), (1, 1), (2, 1), (4, 1), (1, 2), (2, 2), (4, 2)):   # (1 / 2 rows per lane: 16 / 8 waves per tile)
            with _lib.tuned("SEG_NS", ns), _lib.tuned("SEG_PPT", ppt):
                np.testing.assert_array_equal(to_np(plan.backward(gt)), want, err_msg=f"bwd SEG_NS={ns} SEG_PPT={ppt} " + msg)
        with _lib.tuned("SEG_CHUNK", 3):           # several chunks of angles
            np.testing.assert_array_equal(to_np(plan.backward(gt)), want, err_msg="bwd SEG_CHUNK=3 " + msg)
        with _lib.tuned("NO_PLAN", 1):
            np.testing.assert_array_equal(to_np(plan.backward(gt)), want, err_msg="bwd whole rows " + msg)


@pytest.mark.parametrize("H,S,A", [(128, 9, 20), (512, 3, 6)])
def test_bilinear_tfcompat_backward_full_sizes(oracle, H, S, A):
    d = dev()
    rng = np.random.default_rng(H)
    theta = phantoms.dense_theta(180)[:: 180 // A][:A]
    geom = oracle.Geometry(H, H, True)
    plan = RotatePlan(theta, H, H, True, d, interp="bilinear")
    g = rng.standard_normal((S, A, geom.PW)).astype(np.float32)
    want = oracle.rotate_bwd_tfcompat(g, geom, oTinv(oracle, theta, plan), 1)
    gt = torch.from_numpy(g).to(d)
    np.testing.assert_array_equal(to_np(plan.backward(gt)), want)
    for n in (1, 2, 3, 5):                         # slice independence: singles, pairs, quads and their remainders
        if n <= S:
            np.testing.assert_array_equal(to_np(plan.backward(gt[:n].contiguous())), want[:n], err_msg=f"S={n}")


def test_bilinear_exact_adjoint_random_geometries(oracle):
    """The true transpose of the bilinear forward as a deterministic gather (inverse plan of summed weights): within 1e-5 of the
    oracle's in-order scatter, the same bits from every slices-per-cell / rows-per-lane variant and from run to run, and
    <Ax, g> = <x, A^T g>."""
    d = dev()
    rng = np.random.default_rng(int(os.environ.get("CTPVAE_FUZZ_SEED", 20261007)))
    for case, H, W, pad, theta, S in fuzz_cases(rng, int(os.environ.get("CTPVAE_FUZZ_CASES", 28))):
        geom = oracle.Geometry(H, W, pad)
        plan = RotatePlan(theta, H, W, pad, d, interp="bilinear", backward="exact")
        assert plan._exact_bilin_plan is not None, "a rotation always fits three bins"
        img = rng.standard_normal((S, H, W)).astype(np.float32)
        g = rng.standard_normal((S, len(theta), geom.PW)).astype(np.float32)
        want = oracle.rotate_bwd_exact(g, geom, oT(oracle, theta, plan), 1)
        gt = torch.from_numpy(g).to(d)
        msg = f"case {case}: {H}x{W} pad={pad} A={len(theta)} S={S}"
        got = plan.backward(gt)
        assert rel_err(to_np(got), want) <= REL, msg
        assert torch.equal(got, plan.backward(gt)), "run to run " + msg
        for ns, ppt in ((1, 4), (1, 8), (2, 4), (2, 8), (4, 4), (1, 1), (2, 1), (4, 1), (1, 2), (2, 2), (4, 2)):   # (1 / 2 rows per lane: 16 / 8 waves per tile)
            with _lib.tuned("SEG_NS", ns), _lib.tuned("SEG_PPT", ppt):
                assert torch.equal(got, plan.backward(gt)), f"SEG_NS={ns} SEG_PPT={ppt} " + msg
        with _lib.tuned("SEG_CHUNK", 3):
            assert torch.equal(got, plan.backward(gt)), "SEG_CHUNK=3 " + msg
        lhs = float((to_np(plan.forward(torch.from_numpy(img).to(d))).astype(np.float64) * g).sum())
        rhs = float((to_np(got).astype(np.float64) * img).sum())
        assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (msg, lhs, rhs)


@pytest.mark.parametrize("H,S,A", [(128, 50, 20), (512, 4, 12)])
def test_bilinear_exact_adjoint_full_sizes(oracle, H, S, A):
    """Full-size: the transpose property at the headline shape and at 512 x 512; the oracle's scatter on a sample of slices."""
    d = dev()
    rng = np.random.default_rng(H + 1)
    theta = phantoms.dense_theta(180)[:: 180 // A][:A]
    geom = oracle.Geometry(H, H, True)
    plan = RotatePlan(theta, H, H, True, d, interp="bilinear", backward="exact")
    assert plan._exact_bilin_plan is not None
    img = rng.random((S, H, H)).astype(np.float32)
    g = rng.standard_normal((S, A, geom.PW)).astype(np.float32)
    x, gt = torch.from_numpy(img).to(d), torch.from_numpy(g).to(d)
    got = plan.backward(gt)
    assert torch.equal(got, plan.backward(gt))
    lhs = float((to_np(plan.forward(x)).astype(np.float64) * g).sum())
    rhs = float((to_np(got).astype(np.float64) * img).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs), (lhs, rhs)
    n = min(S, 3)
    assert rel_err(to_np(got[:n]), oracle.rotate_bwd_exact(g[:n], geom, oT(oracle, theta, plan), 1)) <= REL


def test_bilinear_exact_plan_refuses_a_non_rotation(oracle):
    """A table row that is not a rotation (a 2x zoom: a pixel's samples span five bins) raises the plan's overflow word; the
    backward then keeps the scatter kernel and still agrees with the oracle."""
    d = dev()
    H = W = 40
    geom = oracle.Geometry(H, W, True)
    theta = np.array([0.3, 1.1])
    T = oracle.rotate_transforms(theta.astype(np.float32), geom.PH, geom.PW).copy()
    c = (geom.PW - 1) / 2
    T[1, :6] = [0.5, 0.0, c * 0.5, 0.0, 0.5, c * 0.5]          # sample positions at half pitch
    Tinv = oracle.invert_transforms(T)
    tabs = (torch.from_numpy(T).to(d), torch.from_numpy(Tinv).to(d))
    plan = RotatePlan(None, H, W, True, d, interp="bilinear", backward="exact", _tables=tabs)
    assert plan._exact_bilin_plan is None
    g = np.random.default_rng(0).standard_normal((2, 2, geom.PW)).astype(np.float32)
    got = to_np(plan.backward(torch.from_numpy(g).to(d)))
    assert rel_err(got, oracle.rotate_bwd_exact(g, geom, T, 1)) <= REL


@pytest.mark.parametrize("interp,code", [("bilinear", 1), ("nearest", 0)])
def test_float64_pixels_are_projected_in_float64(oracle, interp, code):
    """ctvae/tomopy_forward_compare.py:52,56 hands xdesign's float64 phantoms to project_tf_fast and project_tf_low_mem;
    TensorFlow interpolates and sums in the image type with fp32 coordinates and weights (oracle_rotate_fwd_f64): the same
    double operations in the same order -- equal bits; and the float32 call of the same data is a different (narrower) sum."""
    import ct_pvae_amd as cp
    d = dev()
    rng = np.random.default_rng(64)
    for H, W, Z, pad, A in ((128, 128, 2, True, 100), (33, 57, 3, False, 7), (200, 180, 1, True, 5)):
        theta = np.linspace(0.0, np.pi, A, endpoint=False)
        x = rng.random((H, W, Z))                                     # float64, the reference's [X][Y][Z] layout
        geom = oracle.Geometry(H, W, pad)
        T = oracle.rotate_transforms(theta.astype(np.float32), geom.PH, geom.PW)
        want = oracle.rotate_fwd_f64(np.ascontiguousarray(x.transpose(2, 0, 1)), geom, T, code).transpose(1, 2, 0)
        fn = cp.project_tf_low_mem if interp == "bilinear" else cp.project_tf_fast
        got = fn(torch.from_numpy(x).to(d), theta, pad=pad) if interp == "bilinear" else fn(torch.from_numpy(x).to(d), theta, pad=pad, dim=3)
        assert got.dtype == torch.float64 and tuple(got.shape) == (A, geom.PW, Z)
        np.testing.assert_array_equal(to_np(got), want)
        got32 = fn(torch.from_numpy(x.astype(np.float32)).to(d), theta, pad=pad)
        assert got32.dtype == torch.float32 and rel_err(to_np(got32), want) <= REL


def test_nearest_exact_adjoint_without_atomics_at_512(oracle):
    """The byte plan of the nearest exact adjoint holds detectors of <= 255 bins; larger geometries (512 x 512: 728 bins) took
    the scatter kernel -- 6 G global atomics at config 5's shape.  They now gather through the summed-weights plan (a pixel is
    the tap of at most two samples of an angle): <= 1e-5 of the oracle's scatter, equal bits run to run, still the transpose."""
    d = dev()
    rng = np.random.default_rng(12)
    H, S, A = 512, 3, 6
    theta = np.concatenate([[0.0, np.pi / 2], rng.uniform(0, np.pi, A - 2)])
    geom = oracle.Geometry(H, H, True)
    plan = RotatePlan(theta, H, H, True, d, interp="nearest", backward="exact")
    assert plan._exact_plan is None and plan._exact_bilin_plan is not None
    img = rng.random((S, H, H)).astype(np.float32)
    g = rng.standard_normal((S, A, geom.PW)).astype(np.float32)
    gt = torch.from_numpy(g).to(d)
    got = plan.backward(gt)
    assert torch.equal(got, plan.backward(gt))
    assert rel_err(to_np(got), oracle.rotate_bwd_exact(g, geom, oT(oracle, theta, plan), 0)) <= REL
    lhs = float((to_np(plan.forward(torch.from_numpy(img).to(d))).astype(np.float64) * g).sum())
    rhs = float((to_np(got).astype(np.float64) * img).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs), (lhs, rhs)
    small = RotatePlan(theta, 128, 128, True, d, interp="nearest", backward="exact")      # the byte plan keeps what it holds
    assert small._exact_plan is not None and small._exact_bilin_plan is None


@pytest.mark.parametrize("H", [96, 512])
def test_bilinear_dispatch_rows_angle_subsets_and_autograd(oracle, H):
    """The bilinear rows of the dispatch matrix the host code can reach: dense / device-resident subset / host-resident subset,
    forward and both backward modes, on a geometry that fits LDS and on 512 x 512 (tiles); and the drop-in call
    project_tf_low_mem(...).backward() in the reference's [X][Y][Z] layout."""
    import ct_pvae_amd as cp
    d = dev()
    rng = np.random.default_rng(H)
    S, A = 3, 9
    theta = rng.uniform(0, np.pi, A)
    geom = oracle.Geometry(H, H, True)
    img = rng.random((S, H, H)).astype(np.float32)
    x = torch.from_numpy(img).to(d)
    sub = np.array([7, 0, 3, 3, 8])
    tiled = H == 512
    for back in ("tf_compat", "exact"):
        plan = RotatePlan(theta, H, H, True, d, interp="bilinear", backward=back)
        T = oT(oracle, theta, plan)
        assert plan.tiled == tiled
        fwd = (lambda im, TT: oracle.rotate_fwd_tiled(im, geom, TT, oracle.tile_shape(H, H, 1), interp=1)) if tiled else \
              (lambda im, TT: oracle.rotate_fwd(im, geom, TT, 1))
        np.testing.assert_array_equal(to_np(plan.forward(x)), fwd(img, T), err_msg=f"dense {back}")
        for where in ("device", "host"):
            idx = cp.as_angle_index(sub, d, keep_host=(where == "host"))
            np.testing.assert_array_equal(to_np(plan.forward(x, angles_i=idx)), fwd(img, T[sub]), err_msg=f"{where} subset fwd {back}")
            g = rng.standard_normal((S, len(sub), geom.PW)).astype(np.float32)
            got = to_np(plan.backward(torch.from_numpy(g).to(d), angles_i=idx))
            if back == "tf_compat":
                np.testing.assert_array_equal(got, oracle.rotate_bwd_tfcompat(g, geom, oracle.invert_transforms(T)[sub], 1))
            else:
                assert rel_err(got, oracle.rotate_bwd_exact(g, geom, T[sub], 1)) <= REL
    # the drop-in call: [X][Y][Z] in, [A][P][Z] out, gradient through tf_compat (what tf.GradientTape computes)
    xz = torch.from_numpy(np.ascontiguousarray(img.transpose(1, 2, 0))).to(d).requires_grad_(True)
    out = cp.project_tf_low_mem(xz, theta, pad=True)
    gz = rng.standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.from_numpy(gz).to(d))
    T = oracle.rotate_transforms(theta.astype(np.float32), geom.PH, geom.PW)
    want_g = oracle.rotate_bwd_tfcompat(np.ascontiguousarray(gz.transpose(2, 0, 1)), geom, oracle.invert_transforms(T), 1)
    np.testing.assert_array_equal(to_np(xz.grad), want_g.transpose(1, 2, 0))


def test_bilinear_very_many_angles_fall_back_correctly(oracle):
    """The bilinear kernels keep their angles' transform rows and class list in LDS (36 bytes per angle): a call with several
    thousand angles does not fit and takes round 1's kernels (whole slices) / reports "not tiled" (tile geometries: the
    global-memory kernel) -- same sums, bit for bit."""
    d = dev()
    rng = np.random.default_rng(9)
    A = 4300
    theta = rng.uniform(0, np.pi, A)
    for H, W, force in ((40, 40, False), (64, 64, True)):
        img = rng.random((1, H, W)).astype(np.float32)
        geom = oracle.Geometry(H, W, True)
        with _lib.tuned("TILED_FORCE", 1 if force else -1):
            plan = RotatePlan(theta, H, W, True, d, interp="bilinear")
            assert not plan.tiled
            got = to_np(plan.forward(torch.from_numpy(img).to(d)))
        np.testing.assert_array_equal(got, oracle.rotate_fwd(img, geom, oT(oracle, theta, plan), 1))
    with _lib.tuned("TILED_FORCE", 1):          # ... while a few hundred angles are tiled
        assert RotatePlan(theta[:300], 64, 64, True, d, interp="bilinear").tiled
