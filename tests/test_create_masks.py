"""create_all_masks: the mask rules and the measurement model of ctvae/create_masks.py (host logic; torch on CPU)."""
import numpy as np
import pytest
import torch

from ct_pvae_amd.create_masks import create_all_masks, sparse_angles


def test_uniform_masks_follow_the_reference_rule():
    # ctvae/create_masks.py:53-59: spacing ceil(180 / 20) = 9 -> 0, 9, ..., 171; value 1 / nsa
    idx = sparse_angles(180, 20)
    np.testing.assert_array_equal(idx, np.arange(0, 180, 9))
    # spacing * nsa > A wraps around and may hit an angle twice: the one-hot SUM then gives 2 / nsa there
    idx = sparse_angles(10, 4)          # spacing 3 -> 0, 3, 6, 9
    np.testing.assert_array_equal(idx, [0, 3, 6, 9])
    idx = sparse_angles(10, 6)          # spacing 2 -> 0, 2, 4, 6, 8, 10 % 10 = 0
    np.testing.assert_array_equal(idx, [0, 2, 4, 6, 8, 0])
    sino = torch.ones((3, 10, 5))
    masks, samples = create_all_masks(sino, 10, num_sparse_angles=6, train=True, real_data=True, device="cpu")
    np.testing.assert_allclose(masks[0].numpy(), np.array([2, 0, 1, 0, 1, 0, 1, 0, 1, 0]) / 6.0, rtol=1e-6)
    assert torch.equal(samples, sino * masks[..., None])          # real data: the masked sinogram itself
    assert float(masks.sum(1).min()) == pytest.approx(1.0, rel=1e-6)   # the dose is the same for every example


def test_random_masks_and_files(tmp_path):
    rng = np.random.default_rng(0)
    sino = torch.from_numpy(rng.random((40, 30, 16)).astype(np.float32) * 50 - 1.0)     # some negatives: clamped to 0
    masks, samples = create_all_masks(sino, 30, save_path=str(tmp_path), poisson_noise_multiplier=1e2,
                                      num_sparse_angles=5, random=True, train=True, truncate_dataset=32, device="cpu",
                                      real_data=True)
    assert masks.shape == (32, 30) and samples.shape == (32, 30, 16)
    assert ((masks > 0).sum(1) == 5).all() and torch.allclose(masks.sum(1), torch.ones(32))
    assert len({tuple(m.nonzero().flatten().tolist()) for m in masks}) > 1          # not all the same subset
    assert float(samples.min()) >= 0.0 and float(samples[masks == 0].abs().max()) == 0.0
    # train=False reads back exactly what train=True wrote
    m2, s2 = create_all_masks(None, 30, save_path=str(tmp_path), train=False, device="cpu")
    assert torch.equal(m2, masks) and torch.equal(s2, samples)


def test_the_poisson_draw_has_no_cpu_path():
    from ct_pvae_amd import _lib
    with pytest.raises(_lib.RadonLibraryError, match="no CPU path"):
        create_all_masks(torch.ones((4, 10, 5)), 10, num_sparse_angles=2, train=True, device="cpu")


def test_toy_masks():
    sino = torch.rand((8, 2, 2))
    masks, _ = create_all_masks(sino, 2, train=True, toy_masks=True, real_data=True, device="cpu")
    np.testing.assert_array_equal(masks.numpy(), np.tile([[1, 0], [0, 1], [1, 0], [0, 1]], (2, 1)))
    with pytest.raises(ValueError):
        create_all_masks(torch.rand((6, 2, 2)), 2, train=True, toy_masks=True, device="cpu")
