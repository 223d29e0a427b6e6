"""Generates tests/golden/gridsample_crosscheck.npz: the rotate-and-sum projector of the reference computed by a SECOND,
independent implementation -- PyTorch's CPU affine_grid / grid_sample (oracle/torch_gridsample.py) -- on one seeded 128 x 128
foam phantom at the dataset's 180 angles (scripts/images_to_sinograms.py:34), bilinear and nearest, plus the autograd
gradient of the bilinear projector for a seeded cotangent.

    python tests/golden/make_gridsample_crosscheck.py

Why: the reference holds no fixtures and TensorFlow / TomoPy cannot be installed here (PARITY UNPINNED), so the CPU
restatement oracle/radon_oracle.c is pinned by what CAN be had in this container: a framework resampler written by other
people.  tests/test_oracle.py compares the restatement with this file and asserts the agreement recorded below; the
outputs are data (inputs are regenerated from the seed), not code.  Template: ctvae/tomopy_forward_compare.py:51-67.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ct_pvae_amd import phantoms  # noqa: E402
from oracle import radon_oracle as orc  # noqa: E402
from oracle import torch_gridsample as tg  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gridsample_crosscheck.npz")


def main():
    torch.set_num_threads(1)      # one summation order, whatever the host
    img = phantoms.foam_batch(1, 128, seed=0, supersample=2)
    theta = phantoms.dense_theta(180)
    g = np.random.default_rng(2).standard_normal((1, 180, 184)).astype(np.float32)
    x = torch.from_numpy(img)
    bil, grad_bil = tg.fwd_and_grad(img, theta, g, pad=True, mode="bilinear")
    near = tg.rotate_and_sum(x, theta, pad=True, mode="nearest").numpy()
    # agreement with the restatement at the time of generation (the test asserts these bounds, it does not re-derive them)
    geom = orc.Geometry(128, 128, True)
    T = orc.rotate_transforms(theta, geom.PH, geom.PW)
    o_bil, o_near = orc.rotate_fwd(img, geom, T, orc.BILINEAR), orc.rotate_fwd(img, geom, T, orc.NEAREST)
    o_grad = orc.rotate_bwd_exact(g, geom, T, orc.BILINEAR)
    stats = dict(bilinear_max_rel_err=float(np.abs(bil - o_bil).max() / np.abs(o_bil).max()),
                 nearest_differing_ray_sums=int((near != o_near).sum()),
                 nearest_max_rel_err=float(np.abs(near - o_near).max() / np.abs(o_near).max()),
                 exact_adjoint_max_rel_err=float(np.abs(grad_bil - o_grad).max() / np.abs(o_grad).max()))
    print(stats)
    np.savez_compressed(OUT, seed=np.array(0), theta=theta, fwd_bilinear=bil, fwd_nearest=near, g_seed=np.array(2),
                        grad_bilinear=grad_bil, torch_version=np.array(torch.__version__),
                        **{k: np.array(v) for k, v in stats.items()})


if __name__ == "__main__":
    main()
