"""The oracle reproduces the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

ROTATE_CASES = ["rotate_toy", "rotate_rand8", "rotate_rect_nopad", "rotate_rect_pad", "rotate_foam128_a20"]


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name", ROTATE_CASES)
def test_rotate_golden(oracle, golden_dir, name):
    z = load(golden_dir, name)
    img, pad = z["img"], bool(z["pad"])
    geom = oracle.Geometry(img.shape[1], img.shape[2], pad)
    T = oracle.rotate_transforms(z["theta"], geom.PH, geom.PW)
    np.testing.assert_array_equal(T, z["T8"])
    Tinv = oracle.invert_transforms(T)
    np.testing.assert_array_equal(Tinv, z["Tinv8"])
    for tag, interp in (("nearest", 0), ("bilinear", 1)):
        np.testing.assert_array_equal(oracle.rotate_fwd(img, geom, T, interp), z[f"fwd_{tag}"])
        np.testing.assert_array_equal(oracle.rotate_bwd_tfcompat(z["g"], geom, Tinv, interp), z[f"bwd_tfcompat_{tag}"])
        np.testing.assert_array_equal(oracle.rotate_bwd_exact(z["g"], geom, T, interp), z[f"bwd_exact_{tag}"])


def test_toy_golden_holds_the_reference_known_answers(golden_dir):
    z = load(golden_dir, "rotate_toy")
    np.testing.assert_allclose(z["fwd_nearest"], [[[.4, .6], [.7, .3]], [[.4, .6], [.3, .7]]], atol=2e-7)
    s = load(golden_dir, "siddon")
    np.testing.assert_allclose(np.swapaxes(s["toy_out"], 0, 1), [[[.4, .6], [.7, .3]], [[.4, .6], [.3, .7]]], atol=2e-7)


def test_siddon_golden(oracle, golden_dir):
    z = load(golden_dir, "siddon")
    for case, pad in (("toy", False), ("rand", True), ("rect", True), ("foam", True)):
        np.testing.assert_array_equal(oracle.siddon_project(z[case + "_img"], z[case + "_theta"], pad=pad),
                                      z[case + "_out"])


def test_iradon_and_loglik_golden(oracle, golden_dir):
    z = load(golden_dir, "iradon")
    got = oracle.iradon(z["sino"], z["theta"], int(z["x_size"]), int(z["y_size"]), z["filt"])
    np.testing.assert_allclose(got, z["recon"], rtol=1e-12, atol=1e-14)
    z = load(golden_dir, "loglik")
    np.testing.assert_array_equal(oracle.loglik(z["proj"], z["mask"], z["x"], float(z["pnm"]), float(z["eps"])),
                                  z["out"])


def test_tiled_summation_golden(oracle, golden_dir):
    """The tile-blocked association of the row sum (what the HIP path uses for slices larger than LDS): pinned for two
    tile shapes; whatever the tiles, the taps are those of the row-wise sum, so the three agree to fp32 rounding."""
    z = load(golden_dir, "rotate_tiled")
    geom = oracle.Geometry(220, 190, True)
    np.testing.assert_array_equal(oracle.rotate_transforms(z["theta"], geom.PH, geom.PW), z["T8"])
    for tile, key in (((96, 64), "fwd_tiled_96x64"), ((50, 40), "fwd_tiled_50x40")):
        np.testing.assert_array_equal(oracle.rotate_fwd_tiled(z["img"], geom, z["T8"], tile), z[key])
        assert np.abs(z[key] - z["fwd_rowwise"]).max() <= 2e-6 * np.abs(z["fwd_rowwise"]).max()
    # one tile that covers the slice IS the row-wise sum
    np.testing.assert_array_equal(oracle.rotate_fwd_tiled(z["img"], geom, z["T8"], (220, 190)), z["fwd_rowwise"])
    with pytest.raises(ValueError):
        oracle.rotate_fwd_tiled(z["img"], geom, z["T8"], (0, 64))


def test_round2_setup_path_golden(oracle, golden_dir):
    """The set-up path's operators (f2: Poisson measurements with the specified counter-based sampler; f3: ray-driven
    back-projection and SIRT) against their committed vectors: a change of the sampler's specification or of the walk shows
    here, on the CPU."""
    z = load(golden_dir, "round2_setup_path")
    np.testing.assert_array_equal(oracle.poisson_measure(z["p_sino"], z["p_mask"], float(z["p_pnm"]), int(z["p_seed"])),
                                  z["p_out"])
    assert z["p_out"][0, 0, 0] == 0.0 and z["p_out"][0, 0, 1] == 0.0 and (z["p_out"][0, 1] == 0).all()
    data, theta = z["r_data"], z["r_theta"]
    np.testing.assert_array_equal(np.swapaxes(oracle.siddon_project(z["r_img"], theta, pad=True), 0, 1), data)
    np.testing.assert_array_equal(oracle.siddon_backproject(data, theta), z["r_backproject"])
    np.testing.assert_array_equal(oracle.siddon_backproject(data, theta, 24, 24), z["r_backproject_obj"])
    np.testing.assert_array_equal(oracle.sirt(data, theta, 1), z["r_sirt1"])
    np.testing.assert_array_equal(oracle.sirt(data, theta, 7), z["r_sirt7"])
    # <A x, y> = <x, A^T y> on the stored vectors
    lhs = float((data.astype(np.float64) ** 2).sum())
    rhs = float((z["r_img"].astype(np.float64) * z["r_backproject_obj"]).sum())
    assert abs(lhs - rhs) <= 1e-5 * lhs


def test_round3_golden(oracle, golden_dir):
    """gridrec (f3) and the per-object log-likelihood sums' fixed order (f1) against their committed vectors."""
    z = load(golden_dir, "round3")
    np.testing.assert_array_equal(np.swapaxes(oracle.siddon_project(z["g_img"], z["g_theta"], pad=True), 0, 1), z["g_data"])
    np.testing.assert_array_equal(oracle.gridrec(z["g_data"], z["g_theta"]), z["g_parzen"])
    np.testing.assert_array_equal(oracle.gridrec(z["g_data"], z["g_theta"], filter_name="ramlak", ngridx=40, ngridy=44),
                                  z["g_ramlak_40x44"])
    np.testing.assert_array_equal(oracle.loglik_object_sums(z["s_lp"], 0), z["s_sums_bands"])
    np.testing.assert_array_equal(oracle.loglik_object_sums(z["s_lp"], 1), z["s_sums_blocks"])
    want = z["s_lp"].astype(np.float64).sum(axis=(1, 2))
    assert np.abs(z["s_sums_bands"] - want).max() <= 2e-6 * np.abs(want).max()
