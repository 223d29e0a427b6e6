"""On-disk formats and metrics: files round-trip in the reference's layout; compare() reproduces the
numbers scikit-image 0.18.3 gives (golden values generated with it, tests/golden/make_compare_golden.md)."""
import os

import numpy as np

from ct_pvae_amd import dataset_io as io


def test_dataset_files_round_trip(tmp_path):
    theta = np.linspace(0, np.pi, 180, endpoint=False)
    sino = np.random.default_rng(0).random((3, 180, 184)).astype(np.float32)
    io.save_sinograms(str(tmp_path), sino, theta, 128, 128)
    assert sorted(os.listdir(tmp_path)) == ["dataset_parameters.npy", "x_size.npy", "x_train_sinograms.npy", "y_size.npy"]
    raw = np.load(tmp_path / "dataset_parameters.npy", allow_pickle=True)       # what the reference's loader sees
    assert raw.dtype == object and raw.shape == (2,) and raw[1] == 184
    s2, th2, p2 = io.get_sinograms(str(tmp_path))
    np.testing.assert_array_equal(s2, sino)
    np.testing.assert_array_equal(th2, theta)
    assert p2 == 184 and int(np.load(tmp_path / "x_size.npy")) == 128


def test_crop_matches_reference_rule():
    img = np.arange(184 * 184).reshape(184, 184)
    assert io.crop(img, 128, 128).shape == (128, 128) and io.crop(img, 128, 128)[0, 0] == img[28, 28]
    assert io.crop(img[None].repeat(2, 0), 127, 5, ignore_dim_0=True).shape == (2, 127, 5)
    assert io.crop(img, 127, 5)[0, 0] == img[92 - 63, 92 - 2]


def test_compare_matches_scikit_image(golden_dir):
    z = np.load(os.path.join(golden_dir, "compare_skimage018.npz"))
    for k in "abc":
        mse, ssim, psnr = io.compare(z[k + "_r0"], z[k + "_r1"], verbose=False)
        np.testing.assert_allclose([mse, ssim, psnr], [z[k + "_mse"], z[k + "_ssim"], z[k + "_psnr"]], rtol=1e-9)
