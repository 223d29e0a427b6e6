"""Regenerates tests/golden/*.npz from the CPU restatement (oracle/radon_oracle.c).

    python tests/golden/make_golden.py

The reference ships no numeric fixtures for this path and TensorFlow / TomoPy cannot be installed here, so these
vectors are the restatement's own outputs on seeded inputs (PARITY UNPINNED, see oracle/radon_oracle.c).  They
guard the oracle against regressions and give the HIP path a fixed target on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ct_pvae_amd import phantoms  # noqa: E402
from oracle import radon_oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NEAREST, BILINEAR = 0, 1


def rotate_case(name, img, theta, pad, seed):
    rng = np.random.default_rng(seed)
    geom = orc.Geometry(img.shape[1], img.shape[2], pad)
    T = orc.rotate_transforms(theta, geom.PH, geom.PW)
    Tinv = orc.invert_transforms(T)
    g = rng.standard_normal((img.shape[0], len(theta), geom.PW)).astype(np.float32)
    out = dict(img=img, theta=np.asarray(theta, np.float64), pad=np.array(pad), T8=T, Tinv8=Tinv, g=g)
    for tag, interp in (("nearest", NEAREST), ("bilinear", BILINEAR)):
        out[f"fwd_{tag}"] = orc.rotate_fwd(img, geom, T, interp)
        out[f"bwd_tfcompat_{tag}"] = orc.rotate_bwd_tfcompat(g, geom, Tinv, interp)
        out[f"bwd_exact_{tag}"] = orc.rotate_bwd_exact(g, geom, T, interp)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def tiled_case():
    """rotate_fwd_tiled (the association the build uses for slices larger than LDS) on a 220 x 190 slice -- the
    smallest kind that the HIP path tiles -- with coarse pixel values so that the file stays small."""
    rng = np.random.default_rng(6)
    img = (np.round(rng.random((1, 220, 190)) * 16) / 16).astype(np.float32)
    theta = np.array([0.0, 0.3, np.pi / 4, 1.9, np.pi / 2, 2.8])
    geom = orc.Geometry(220, 190, True)
    T = orc.rotate_transforms(theta, geom.PH, geom.PW)
    np.savez_compressed(os.path.join(OUT, "rotate_tiled.npz"), img=img, theta=theta, T8=T,
                        fwd_tiled_96x64=orc.rotate_fwd_tiled(img, geom, T, (96, 64)),
                        fwd_tiled_50x40=orc.rotate_fwd_tiled(img, geom, T, (50, 40)),
                        fwd_rowwise=orc.rotate_fwd(img, geom, T, NEAREST))


def round2_case():
    """Round-2 set-up-path operators: the Poisson measurement model with the build's specified sampler (f2), the ray-driven
    back-projection and SIRT (f3) -- small shapes, fixed seeds."""
    rng = np.random.default_rng(21)
    sino = (10 ** rng.uniform(-3, 5, size=(2, 5, 24))).astype(np.float32)
    sino[0, 0, :4] = [-1.0, 0.0, 9.99, 10.0]
    mask = np.array([[1, 0, 0.5, 0.05, 1], [0.05, 0.05, 0, 1, 0.5]], np.float32)
    out = dict(p_sino=sino, p_mask=mask, p_pnm=np.array(1e2, np.float32), p_seed=np.array(2 ** 40 + 17, np.uint64),
               p_out=orc.poisson_measure(sino, mask, 1e2, 2 ** 40 + 17))
    img = phantoms.foam_batch(2, 24, seed=9, supersample=2)
    theta = np.array([0.0, 0.35, 0.9, np.pi / 2, 2.0, 2.6, 3.3, 4.4, 5.9])
    data = np.ascontiguousarray(np.swapaxes(orc.siddon_project(img, theta, pad=True), 0, 1))       # [2][9][36]
    out.update(r_img=img, r_theta=theta, r_data=data, r_backproject=orc.siddon_backproject(data, theta),
               r_backproject_obj=orc.siddon_backproject(data, theta, 24, 24), r_sirt1=orc.sirt(data, theta, 1),
               r_sirt7=orc.sirt(data, theta, 7))
    np.savez_compressed(os.path.join(OUT, "round2_setup_path.npz"), **out)


def round3_case():
    """Round-3 operators: gridrec (f3; tomopy's default parzen filter and the plain ramlak, an odd slice count, a grid smaller
    than the detector) on the ray-driven sinograms of small foams, and the per-object log-likelihood sums in the library's
    fixed order (f1), both partitions."""
    img = phantoms.foam_batch(3, 32, seed=4, supersample=2)
    theta = np.linspace(0, np.pi, 24, endpoint=False)
    data = np.ascontiguousarray(np.swapaxes(orc.siddon_project(img, theta, pad=True), 0, 1))        # [3][24][48]
    out = dict(g_img=img, g_theta=theta, g_data=data, g_parzen=orc.gridrec(data, theta),
               g_ramlak_40x44=orc.gridrec(data, theta, filter_name="ramlak", ngridx=40, ngridy=44))
    rng = np.random.default_rng(33)
    lp = (-6.0 * rng.random((4, 70, 184))).astype(np.float32)
    out.update(s_lp=lp, s_sums_bands=orc.loglik_object_sums(lp, 0), s_sums_blocks=orc.loglik_object_sums(lp, 1))
    np.savez_compressed(os.path.join(OUT, "round3.npz"), **out)


def main():
    orc.build(force=True)
    if sys.argv[1:] == ["round3"]:
        round3_case()
        return
    if sys.argv[1:] == ["tiled"]:
        tiled_case()
        return
    if sys.argv[1:] == ["round2"]:
        round2_case()
        return
    round2_case()
    round3_case()
    tiled_case()
    rng = np.random.default_rng(0)
    # the reference's 2x2 toy set (scripts/create_toy_images.py:36-40), no padding
    rotate_case("rotate_toy", phantoms.toy_images(), np.array([0, np.pi / 2]), False, 1)
    rotate_case("rotate_rand8", rng.standard_normal((2, 8, 8)).astype(np.float32), rng.uniform(0, np.pi, 5), True, 2)
    rotate_case("rotate_rect_nopad", rng.random((3, 16, 12), dtype=np.float32), rng.uniform(0, np.pi, 7), False, 3)
    rotate_case("rotate_rect_pad", rng.random((2, 10, 15), dtype=np.float32), rng.uniform(-np.pi, 2 * np.pi, 6), True, 4)
    theta20 = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, 20)]
    foam = phantoms.foam_batch(1, 128, seed=0)
    rotate_case("rotate_foam128_a20", foam, theta20, True, 5)

    # siddon (tomopy.project) cases
    sid = {}
    sid["toy_img"] = phantoms.toy_images()
    sid["toy_theta"] = np.array([0, np.pi / 2])
    sid["toy_out"] = orc.siddon_project(sid["toy_img"], sid["toy_theta"], pad=False)
    sid["rand_img"] = rng.random((2, 16, 16), dtype=np.float32)
    sid["rand_theta"] = rng.uniform(0, 2 * np.pi, 9)
    sid["rand_out"] = orc.siddon_project(sid["rand_img"], sid["rand_theta"], pad=True)
    sid["rect_img"] = rng.random((1, 12, 20), dtype=np.float32)
    sid["rect_theta"] = np.linspace(0, np.pi, 8, endpoint=False)
    sid["rect_out"] = orc.siddon_project(sid["rect_img"], sid["rect_theta"], pad=True)
    sid["foam_img"] = foam
    sid["foam_theta"] = theta20
    sid["foam_out"] = orc.siddon_project(foam, theta20, pad=True)
    np.savez_compressed(os.path.join(OUT, "siddon.npz"), **sid)

    # iradon
    B, A, P, X, Y = 2, 12, 34, 20, 22
    fbp = dict(sino=rng.random((B, A, P)), theta=np.linspace(0, np.pi, A, endpoint=False),
               filt=np.abs(np.fft.fftfreq(P)) * 2, x_size=np.array(X), y_size=np.array(Y))
    fbp["recon"] = orc.iradon(fbp["sino"], fbp["theta"], X, Y, fbp["filt"])
    np.savez_compressed(os.path.join(OUT, "iradon.npz"), **fbp)

    # log-likelihood epilogue
    B, A, P = 3, 4, 23
    proj = (rng.random((B, A, P), dtype=np.float32) * 60).astype(np.float32)
    mask = np.full((B, A), 1 / 20, np.float32)
    x = (rng.poisson(proj * mask[..., None] * 1e4) / 1e4).astype(np.float32)
    eps = float(np.finfo(np.float32).eps)
    np.savez_compressed(os.path.join(OUT, "loglik.npz"), proj=proj, mask=mask, x=x, pnm=np.array(1e4, np.float32),
                        eps=np.array(eps, np.float32), out=orc.loglik(proj, mask, x, 1e4, eps))
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
