"""The P-VAE training step with the HIP projector as physics decoder (README quick-check recipe, scaled down)."""
import math
import os

import pytest
import torch

from ct_pvae_amd import trainer as tr

pytestmark = pytest.mark.gpu


def test_quick_check_recipe_trains():
    # README.md:80: --nsa 20 --td 50 -b 5 --ns 2 --api 20 --pnm 1e4 --pnm_start 1e3 --random --normal
    # (without --pnm_start: the reference's annealing multiplies pnm by a factor > 1 every iteration, which by itself
    # raises the loss -- checked separately below)
    args = tr.get_args("--nsa 20 --td 10 -b 5 --ns 2 --api 20 --pnm 1e4 --random --normal -i 60 --lr 1e-3 --train".split())
    t = tr.PVAETrainer(args, torch.device("cuda", 0))
    assert t.P == 184 and t.x_size == 128 and tuple(t.proj_samples.shape) == (10, 180, 184)
    assert tuple(t.input_encode.shape) == (10, 2, 128, 128)
    assert abs(t.masks.sum(dim=1) - 1).max() < 1e-6 and int((t.masks[0] > 0).sum()) == 20
    before = [p.detach().clone() for p in t.params[:3]]
    losses, _ = t.train()
    assert len(losses) == 60 and all(math.isfinite(x) for x in losses)
    assert any(not torch.equal(a, b) for a, b in zip(before, t.params[:3]))      # the optimiser moved the nets
    assert sum(losses[-10:]) < sum(losses[:10])                                   # and the ELBO improves
    # gradients flow through the projector into the decoder
    t.opt.zero_grad(set_to_none=True)
    ps, m, ie = t._batch()
    loss_vec, _, _, recon = tr.find_loss_vae_unsup(ps, m, ie, t.enc, t.dec, t.pnm, t.sqrt_reg, 1.0, 1.0, num_samples=1,
                                                   theta=t.theta, angles_i=torch.arange(0, 180, 9, device=t.dev))
    loss_vec.mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in t.dec.parameters())
    assert recon.shape == (5, 1, 128, 128)


def test_pnm_annealing_follows_the_reference_quirk():
    # ctvae/main_ct_vae.py:146-149,352,392: the variable holds the FINAL pnm and is multiplied by factor**iter
    args = tr.get_args("--nsa 20 --td 6 -b 3 --ns 1 --api 10 --pnm 1e4 --pnm_start 1e3 --normal -i 10 --train".split())
    t = tr.PVAETrainer(args, torch.device("cuda", 0))
    assert math.isclose(t.pnm_anneal ** 10, 10.0, rel_tol=1e-9)
    t.train()
    assert math.isclose(float(t.pnm) * t.pnm_anneal ** (t.iter - 1), 1e4 * 10 ** 0.9, rel_tol=1e-5)


def test_checkpoint_roundtrip(tmp_path):
    args = tr.get_args("--nsa 20 --td 6 -b 3 --ns 1 --api 10 --pnm 1e4 --normal -i 2 --train".split())
    t = tr.PVAETrainer(args, torch.device("cuda", 0))
    t.train()
    path = str(tmp_path / "training_checkpoints" / "ckpt-1.pt")
    t.save(path, [1.0, 2.0])
    t2 = tr.PVAETrainer(args, torch.device("cuda", 0))
    t2.restore(path)
    assert t2.iter == 2
    for a, b in zip(t.params, t2.params):
        assert torch.equal(a, b)


def _dp_worker(rank, world, port, out_dir):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)                      # both ranks share the box's one GPU; gloo carries the gradients
    dist.init_process_group("gloo")
    args = tr.get_args("--nsa 20 --td 8 -b 4 --ns 1 --api 10 --pnm 1e4 --normal -i 3 --train".split())
    t = tr.PVAETrainer(args, torch.device("cuda", 0))
    assert (t.world, t.rank) == (world, rank)
    ps, _, _ = t._batch()
    assert ps.shape[0] == 2                       # global batch 4 -> 2 objects per rank
    losses, _ = t.train()
    flat = torch.cat([p.detach().reshape(-1) for p in t.params]).cpu()
    torch.save({"flat": flat, "losses": losses}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_ranks_stay_in_lockstep(tmp_path):
    """2 ranks (gloo), each on its shard of the batch: after one flat-bucket gradient all-reduce per step the
    replicas hold bit-identical parameters and report the same global loss."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(a["flat"], b["flat"])
    assert a["losses"] == b["losses"] and all(math.isfinite(x) for x in a["losses"])


@pytest.mark.gpu
def test_setup_chain_on_the_device(tmp_path):
    """The steps either side of the projector in the reference's setup (create_all_masks -> iradon_all), on the GPU:
    files written like the reference's, reloaded bit for bit, encoder input = [n][X][Y][algorithms + mask]."""
    import numpy as np
    from ct_pvae_amd import create_all_masks, iradon_all, phantoms
    from ct_pvae_amd.helper_functions import create_sinograms
    d = torch.device("cuda", 0)
    theta = phantoms.dense_theta(180)
    imgs = torch.from_numpy(phantoms.foam_batch(4, 128, seed=3, supersample=2)).to(d)
    sino = create_sinograms(imgs, theta, pad=True)
    masks, samples = create_all_masks(sino, 180, save_path=str(tmp_path), poisson_noise_multiplier=1e3,
                                      num_sparse_angles=20, train=True, truncate_dataset=4)
    assert masks.device.type == "cuda" and samples.shape == (4, 180, 184)
    enc = iradon_all(samples, masks, 184, theta, ["gridrec"], float(np.finfo(np.float32).eps), 128, 128,
                     save_path=str(tmp_path), train=True)
    assert enc.shape == (4, 128, 128, 2) and enc.dtype == torch.float32 and torch.isfinite(enc).all()
    # the gridrec channel resembles the phantom (sparse, noisy: loose), the mask channel is the same for uniform masks
    c = np.corrcoef(enc[0, ..., 0].cpu().numpy().ravel(), imgs[0].cpu().numpy().ravel())[0, 1]
    assert c > 0.6
    assert torch.allclose(enc[0, ..., 1], enc[3, ..., 1])
    again = iradon_all(samples, masks, 184, theta, ["gridrec"], 1e-7, 128, 128, save_path=str(tmp_path), train=False)
    assert torch.equal(again, enc)
    with pytest.raises(ValueError):
        iradon_all(samples, masks, 184, theta, ["art"], 1e-7, 128, 128, train=True)


def test_algorithms_flag_sets_the_encoder_channels():
    """--algorithms (ctvae/main_ct_vae.py:111-112): one encoder input channel per initial reconstruction, plus the mask's
    back-projection -- the README's list (README.md:221); an unknown name is refused."""
    assert tr.get_args([]).algorithms == ["gridrec"]                 # the reference's default
    args = tr.get_args("--nsa 20 --td 6 -b 3 --ns 2 --api 10 --pnm 1e4 --random --normal -i 2 --train "
                       "--algorithms sirt gridrec".split())
    t = tr.PVAETrainer(args, torch.device("cuda", 0))
    assert tuple(t.input_encode.shape) == (6, 3, 128, 128)
    assert math.isfinite(t.train_step())
    bad = tr.get_args("--nsa 20 --td 6 -b 3 --train --algorithms art".split())
    with pytest.raises(ValueError):
        tr.PVAETrainer(bad, torch.device("cuda", 0))


def test_reference_dataset_folder_in_and_reconstruction_file_out(tmp_path):
    """SURVEY 8 f4 end to end: a dataset folder in the layout scripts/images_to_sinograms.py writes (made here with the
    GPU TomoPy-style projector) is what --input_path trains from, and final_evaluation leaves loss_final.npy and
    reconstruction_final.npy in the layout bin/final_merit.py reads and compare() scores."""
    import numpy as np
    from ct_pvae_amd import dataset_io, phantoms
    imgs = phantoms.foam_batch(6, 64, seed=3, supersample=2)
    ds = str(tmp_path / "dataset_foam")
    theta = np.linspace(0, np.pi, 60, endpoint=False)
    sino = dataset_io.images_to_sinograms(imgs, ds, theta=theta)
    assert sino.shape == (6, 60, 94) and sorted(os.listdir(ds)) == ["dataset_parameters.npy", "x_size.npy",
                                                                    "x_train_sinograms.npy", "y_size.npy"]
    out = str(tmp_path / "run")
    args = tr.get_args(f"--input_path {ds} --save_path {out} --nsa 10 --td 6 -b 3 --ns 2 --api 10 --pnm 1e4 --random "
                       "--normal -i 3 --train".split())
    t = tr.PVAETrainer(args, torch.device("cuda", 0))
    assert t.P == 94 and t.x_size == 64 and t.num_angles == 60 and t.truth is None
    assert tuple(t.proj_samples.shape) == (6, 60, 94) and tuple(t.input_encode.shape) == (6, 2, 64, 64)
    losses, _ = t.train()
    assert len(losses) == 3 and all(math.isfinite(v) for v in losses)
    loss_final, recon = t.final_evaluation(out)
    assert loss_final.shape == (2,) and recon.shape == (6, 64, 64, 1) and np.isfinite(recon).all() and (recon >= 0).all()
    saved = np.load(os.path.join(out, "reconstruction_final.npy"))
    assert np.array_equal(saved, recon) and os.path.exists(os.path.join(out, "loss_final.npy"))
    mse, ssim, psnr = dataset_io.compare(imgs[0], saved[0, ..., 0], verbose=False)
    assert np.isfinite([mse, ssim, psnr]).all()


def test_toy_recipe(tmp_path):
    """README.md:199 (the toy problem): 2x2 objects, two angles, no padding, the two-angle toy masks, one angle per
    example -- from a dataset folder made as scripts/create_toy_images.py + images_to_sinograms.py --toy make it."""
    import numpy as np
    from ct_pvae_amd import dataset_io, phantoms
    imgs = np.tile(np.repeat(phantoms.toy_images(), 2, axis=0), (4, 1, 1))          # [16][2][2]
    ds = str(tmp_path / "dataset_toy_discrete2")
    sino = dataset_io.images_to_sinograms(imgs, ds, theta=np.array([0.0, np.pi / 2]), pad=False)
    assert sino.shape == (16, 2, 2) and np.allclose(sino[0], [[0.4, 0.6], [0.7, 0.3]], atol=1e-6)
    args = tr.get_args(f"--input_path {ds} -b 4 --pnm 10000 -i 5 --td 16 --train --nsa 1 --ik 2 --il 5 --ks 2 --nb 3 "
                       "--api 2 --se 1 --no_pad --ns 10 --pnm_start 1000 --normal --toy_masks".split())
    t = tr.PVAETrainer(args, torch.device("cuda", 0))
    assert t.P == 2 and t.x_size == 2 and not t.pad
    assert torch.equal(t.masks[:4].cpu(), torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [0.0, 1.0]]))
    losses, _ = t.train()
    assert len(losses) == 5 and all(math.isfinite(v) for v in losses)
    _, recon = t.final_evaluation()
    assert recon.shape == (16, 2, 2, 1)


def test_train_then_restore_and_evaluate_from_the_saved_files(tmp_path):
    """The reference's two-phase use (README.md:80 with --train, then --restore --ulc without it): training leaves the
    setup arrays and a checkpoint under --save_path; a second process-equivalent reads all of them back instead of
    drawing new masks and noise, restores the nets and evaluates."""
    import numpy as np
    out = str(tmp_path / "run")
    base = f"--save_path {out} --nsa 10 --td 6 -b 3 --ns 2 --api 10 --pnm 1e4 --random --normal -i 4 --n_pixel 64 --num_angles 60"
    tr.main((base + " --train").split())
    for name in ("all_masks.npy", "all_proj_samples.npy", "all_input_encode.npy", "train_loss_vec.npy",
                 "reconstruction_final.npy", "loss_final.npy"):
        assert os.path.exists(os.path.join(out, name)), name
    again = tr.PVAETrainer(tr.get_args((base + " --restore").split()), torch.device("cuda", 0))  # reloads, does not redraw
    saved = torch.from_numpy(np.load(os.path.join(out, "all_proj_samples.npy"))).cuda()
    assert torch.equal(again.proj_samples, saved)
    again.restore(again.latest_checkpoint())
    assert again.iter == 4 and again.latest_checkpoint().endswith("ckpt-3.pt")
    loss_final, recon = again.final_evaluation()
    assert recon.shape == (6, 64, 64, 1) and np.isfinite(loss_final).all()


def test_beta_branch_of_the_elbo():
    """Without --normal the reference's ELBO uses Beta latents, a Beta(0.5, 0.5) prior and a Beta output distribution
    (ctvae/helper_functions.py:247-252, :275-285; ctvae/main_ct_vae.py:369-372 -- its argparse default): the harness runs
    it through the same HIP physics decoder, reconstructions stay inside (0, 1), losses are finite and gradients flow."""
    args = tr.get_args("--nsa 20 --td 6 -b 3 --ns 2 --api 10 --pnm 1e4 --random -i 3 --train".split())
    assert args.use_normal is False
    t = tr.PVAETrainer(args, torch.device("cuda", 0))
    before = [p.detach().clone() for p in t.params[:3]]
    losses = [t.train_step() for _ in range(3)]
    assert all(math.isfinite(v) for v in losses)
    assert any(not torch.equal(a, b) for a, b in zip(before, t.params[:3]))
    loss_final, recon = t.final_evaluation(None)
    assert recon.shape == (6, 128, 128, 1) and (recon > 0).all() and (recon < 1).all()
    mse, _ = t.evaluate()
    assert math.isfinite(mse)
