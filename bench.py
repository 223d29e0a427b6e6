#!/usr/bin/env python3
"""bench.py -- projections/sec (fwd+adj) of the Radon projector on synthetic 128x128 foam, 20 sparse angles.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: rotate-and-sum forward (nearest, the reference's
project_tf_fast) followed by the backward TensorFlow runs for it (tf_compat), on B=50 objects per GPU that are
already resident in HBM (BASELINE.json configs[1]).  The batch shards over ranks with no data-path collective
(weak scaling: 50 objects per GPU).  Rank 0 prints ONE JSON line; see DESIGN.md section "Measurement".
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ct_pvae_amd import phantoms, sharding  # noqa: E402
from ct_pvae_amd.forward_functions import RotatePlan, project_tf_fast  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak (spec)
HBM_COPY_GBS = 6290.0        # same guide: measured float4 copy ceiling
B_PER_GPU, N_PIX, A_SPARSE = 50, 128, 20


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--angles", type=int, default=A_SPARSE, help="20 (headline) or 180 (dense evaluation set)")
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="objects per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--min-ms", type=float, default=50.0,
                    help="the K-step timed region is repeated until this much time has been measured in all; the median "
                         "region is reported (steps stays K)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the all-cores CPU baseline (0: the cores this process may run on, at most 16 -- the "
                         "CPU share of a one-GPU box)")
    ap.add_argument("--mode", choices=["projector", "train", "siddon", "n512"], default="projector",
                    help="projector: the headline fwd+adj metric, BASELINE config 2 (default; --angles 180 = config 4's "
                         "per-GPU share); train: config 3, P-VAE steps/s; siddon: config 1, TomoPy-style forward; "
                         "n512: config 5, 512x512 x 90 angles fwd + log-likelihood + adj")
    ap.add_argument("--graph-steps", type=int, default=100,
                    help="steps captured per HIP graph in the timed loop (projector mode; the K mod this remaining steps are launched from Python)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every step from Python instead of replaying HIP graphs of --graph-steps steps (projector mode)")
    ap.add_argument("--grad-allreduce", action="store_true",
                    help="config 4 (batch=400 over 8 GPUs, 180 angles): every step also sums one flat fp32 bucket of the "
                         "P-VAE's 711,164 gradients (2.8 MB) over the ranks -- the data-parallel trainer's only collective")
    ap.add_argument("--plan-format", choices=["auto", "compact", "u16"], default="auto",
                    help="forward gather plan of the projector mode (auto: u16 taps below 64 angles, step codes from there)")
    ap.add_argument("--cold", action="store_true",
                    help="cycle 96 distinct input batches (objects + cotangents, 385 MB at the default shape: more than the "
                         "256 MB Infinity Cache) instead of re-projecting ONE resident batch; config.inputs says which")
    ap.add_argument("--total-batch", type=int, default=0,
                    help="STRONG scaling (BASELINE configs[3]: a fixed batch of 400 across the ranks): the ranks share this many "
                         "objects (sharding.shard_range), `scaling` reads \"strong\"; 0 = weak scaling, --batch objects per rank")
    ap.add_argument("--project-scaling", action="store_true",
                    help="one GPU: time the per-rank shares a fixed batch of --total-batch (default 400) objects at --angles would "
                         "leave each of 1 / 2 / 4 / 8 ranks and print the implied strong-scaling speed-ups (a projection from "
                         "single-GPU timings, not a multi-GPU measurement)")
    ap.add_argument("--no-modes", action="store_true", help="skip the four (interp x backward) modes and the cold figure")
    ap.add_argument("--n512-batch", type=int, default=32,
                    help="objects per GPU in --mode n512 (32: 768 tile workgroups = 3 full rounds on 256 CUs; 8: 192)")
    return ap.parse_args()


def self_launch(n_gpus):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed.run around it: start the N ranks as CHILDREN
    (python -m torch.distributed.run ... bench.py <same arguments>), before this process has touched the GPU -- a process
    that has initialised HIP must never exec or fork GPU work -- pass their output through and exit with their code.
    On a box with fewer than N GPUs the ranks rehearse on device 0 over gloo (CTPVAE_REHEARSE_ONE_GPU, sharding.py)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if torch.cuda.device_count() < n_gpus:        # counting devices does not initialise the GPU
        env["CTPVAE_REHEARSE_ONE_GPU"] = "1"
        print(f"[bench] {torch.cuda.device_count()} GPU(s) visible for --gpus {n_gpus}: rehearsing all ranks on device 0 "
              "over gloo (not a scaling measurement)", file=sys.stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def dist_setup(n_gpus):
    world, rank, local = sharding.init_from_env()      # one process per GPU; "nccl" = RCCL
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}")
    return world, rank, local


def barrier_sync(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def capture_graph(body):
    """Capture `body`'s launches into a HIP graph and replay it once (untimed upload); None if capture is not possible
    here, in which case the caller launches from Python.  thread_local capture mode: with RCCL initialised, its watchdog
    thread may query events while this thread captures, which the default global mode treats as an error."""
    torch.cuda.synchronize()
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            body()
        graph.replay()
        torch.cuda.synchronize()
        return graph
    except Exception as exc:        # noqa: BLE001 -- any capture failure only costs the graph, never the measurement
        print(f"[bench] HIP graph capture unavailable ({type(exc).__name__}: {exc}); launching from Python", file=sys.stderr)
        torch.cuda.synchronize()
        return None


def close_timed_region(t0, world):
    """Closing bracket of a timed region: this rank's K steps are complete (synchronize) -> read the clock -> barrier +
    synchronize -> MAX over ranks.  The clock is read before the closing barrier so that the collective's own latency
    (tens of microseconds over 8 ranks, against 14 us steps) is not billed to the steps; the maximum over ranks still
    spans from the common start to the slowest rank's last kernel."""
    torch.cuda.synchronize()
    local = time.perf_counter() - t0
    barrier_sync(world)
    return max_over_ranks(local, world)


def max_over_ranks(seconds, world):
    return sharding.max_over_ranks(seconds)


def cpu_baseline(imgs, theta, g, threads=0):
    """CPU figures for the SAME fwd+adj pair on a bounded sample of the workload (about 25 s in all), after the template
    of ctvae/tomopy_forward_compare.py:51-67 (three timings side by side):
      * the CPU restatement (oracle/, kind 'port'), ONE thread -- `value`: what the reference's per-image Python loop gets
        from a one-slice call (scripts/images_to_sinograms.py:62-66; TomoPy parallelises over slices only);
      * the same restatement, one object per thread on `cores_all` threads -- the best case of a multi-slice stack;
      * PyTorch's CPU grid_sample rotate-and-sum with autograd (oracle/torch_gridsample.py) -- the framework-op
        implementation comparable to the reference's TensorFlow graph on a CPU;
      * the TomoPy-style (siddon) forward the reference uses on the CPU for dataset generation, one thread."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import radon_oracle as orc
    from oracle import torch_gridsample as tg
    orc.build()
    geom = orc.Geometry(imgs.shape[1], imgs.shape[2], True)
    T = orc.rotate_transforms(theta, geom.PH, geom.PW)
    Tinv = orc.invert_transforms(T)
    A, B = len(theta), imgs.shape[0]

    def one_object(k):
        orc.rotate_fwd(imgs[k:k + 1], geom, T, orc.NEAREST)
        orc.rotate_bwd_tfcompat(g[k:k + 1], geom, Tinv, orc.NEAREST)

    n_obj, t_rot = 0, 0.0
    while t_rot < 6.0:
        t0 = time.perf_counter()
        one_object(n_obj % B)
        t_rot += time.perf_counter() - t0
        n_obj += 1
    # all cores: ctypes releases the GIL, so threads run the C restatement in parallel; one object per thread per round
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores_all = threads if threads > 0 else max(1, min(avail, 16))
    rounds = max(1, int(round(5.0 / (t_rot / n_obj))))         # ~5 s: every thread does `rounds` objects
    with ThreadPoolExecutor(cores_all) as pool:
        list(pool.map(one_object, range(cores_all)))            # threads up, caches warm
        t0 = time.perf_counter()
        list(pool.map(lambda w: [one_object((w + r) % B) for r in range(rounds)], range(cores_all)))
        t_all = time.perf_counter() - t0
    # torch-CPU: nearest forward + autograd gradient (the true transpose of grid_sample's forward; TF's gradient is the
    # re-sampling tf_compat restates -- same amount of work), default intra-op threads
    torch.set_num_threads(cores_all)          # the same cores as the all-cores figure (the default is every host core)
    nt, t_torch, n_torch = torch.get_num_threads(), 0.0, 0
    tg.fwd_and_grad(imgs[:1], theta, g[:1], pad=True, mode="nearest")
    while t_torch < 5.0:
        k = n_torch % B
        t0 = time.perf_counter()
        tg.fwd_and_grad(imgs[k:k + 1], theta, g[k:k + 1], pad=True, mode="nearest")
        t_torch += time.perf_counter() - t0
        n_torch += 1
    n_sid, t_sid = 0, 0.0
    while t_sid < 4.0:
        k = n_sid % B
        t0 = time.perf_counter()
        orc.siddon_project(imgs[k:k + 1], theta, pad=True)
        t_sid += time.perf_counter() - t0
        n_sid += 1
    return {
        "value": n_obj * A / t_rot, "unit": "projections/s (fwd+adj)", "cores": 1, "kind": "port",
        "sample": f"{n_obj} objects x {A} angles, rotate nearest fwd + tf_compat bwd, oracle/radon_oracle.c -O2, "
                  f"1 thread, {t_rot:.1f} s",
        "all_cores": {"value": cores_all * rounds * A / t_all, "unit": "projections/s (fwd+adj)", "cores": cores_all,
                      "kind": "port", "sample": f"{cores_all} threads x {rounds} objects x {A} angles, one object per "
                                                f"thread, same restatement, {t_all:.1f} s"},
        "torch_cpu": {"value": n_torch * A / t_torch, "unit": "projections/s (fwd+adj)", "cores": nt, "kind": "port",
                      "sample": f"{n_torch} objects x {A} angles, torch {torch.__version__} CPU affine_grid + grid_sample "
                                f"(nearest) + autograd backward, {nt} intra-op threads, {t_torch:.1f} s"},
        "siddon_fwd_projections_per_s": n_sid * A / t_sid,
        "siddon_sample": f"{n_sid} objects x {A} angles, TomoPy project.c restatement (forward only), 1 thread, "
                         f"{t_sid:.1f} s",
        "host_cores": os.cpu_count(), "cores_available": avail,
    }


def train_mode(args, world, rank, dev):
    """BASELINE config 3: main_ct_vae.py --nsa 20 --td 50 -b 5 --ns 2 --api 20 (README.md:80), HIP projector decoder."""
    from ct_pvae_amd import trainer as tr
    targs = tr.get_args(f"--nsa 20 --td 50 -b {5 * world} --ns 2 --api 20 --pnm 1e4 --pnm_start 1e3 --random --normal "
                        f"-i {args.steps + args.warmup} --train --miopen_find".split())
    t = tr.PVAETrainer(targs, dev)
    for _ in range(args.warmup):
        t.train_step(sync=False)
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t.train_step(sync=False)
    elapsed = close_timed_region(t0, world)
    # Projector share: EXACTLY what a step does with the projector, everything per-step included -- a fresh 20-angle
    # subset drawn on the host and uploaded, calculate_log_prob_M_given_R on the ns * b = 10 sample-objects against the
    # trainer's own dense 180-angle plan (angle-index operand: no table, plan or gather kernel per step), the per-object
    # sums and backward() -- timed alone on the host clock with the GPU drained at both ends.
    from ct_pvae_amd.helper_functions import calculate_log_prob_M_given_R
    nobj = targs.ns * (targs.batch_size // world)
    xs = torch.rand((nobj, 128, 128, 1), device=dev, requires_grad=True)
    mask10, meas10 = t.masks[:nobj].contiguous(), t.proj_samples[:nobj].contiguous()
    w10 = torch.full((nobj,), -1e-5, device=dev)

    def projector_part():
        ai = t._to_device(t.angles.next().astype(np.int32))
        xs.grad = None
        lp = calculate_log_prob_M_given_R(xs, mask10, meas10, t.pnm, t.sqrt_reg, theta=t.theta_host, angles_i=ai, pad=t.pad)
        with torch.autograd.set_multithreading_enabled(False):
            lp.sum(dim=(1, 2, 3)).backward(w10)

    for _ in range(20):
        projector_part()
    torch.cuda.synchronize()
    tp = time.perf_counter()
    for _ in range(200):
        projector_part()
    torch.cuda.synchronize()
    proj_s = (time.perf_counter() - tp) / 200
    if rank == 0:
        print(json.dumps({"metric": "P-VAE training steps/sec (main_ct_vae.py --nsa 20 --td 50 -b 5 --ns 2 --api 20)",
                          "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "P-VAE step: encoder + 2 samples x (decoder, TruncatedNormal sample, HIP "
                                                 "projector fwd, log-likelihood) + backward + Adam; 5 objects/GPU, 20 of 180 angles"},
                          "projector_ms_per_step": proj_s * 1e3, "projector_share": proj_s / (elapsed / args.steps),
                          "projector_what": "host time of the step's whole projector call (angle-subset upload, "
                                            "calculate_log_prob_M_given_R fwd on the dense plan, per-object sums, backward), "
                                            "timed alone; includes all per-step set-up"}))


LDS_PEAK_TAPS = 150e12 / 4.0     # MI355X_MICROARCH.md: ~150 TB/s aggregate for conflict-free ds_read_b64 / b128 = 3.75e13 fp32 taps/s


def committed_counters(tag, kernel_substr, samples_per_launch, kernel_seconds):
    """What this run cannot observe itself, from the COMMITTED counter files of the same command (separate rocprofv3 --pmc
    passes, tools/collect_r04.sh): (traffic bytes per launch of the kernel, its source, the `lds` object of the roofline).
    traffic = 2 * FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction).  lds: busy_frac = LDS-array cycles per CU / kernel
    cycles; conflict_frac = bank-conflict cycles / LDS-array cycles; taps_per_s = algorithmic samples of this launch / its live
    duration; bound_taps_per_s = the conflict-free gather rate of the chip; frac = their ratio ("vs the LDS bound")."""
    import glob
    traffic = source = None
    lds = None
    try:
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{tag}traffic_pmc.json")))[-1]
        pmc = json.load(open(f))
        traffic = max((v for k, v in pmc["kernels"].items() if kernel_substr in k),
                      key=lambda v: v.get("dispatches", 0))["traffic_bytes_per_launch"]
        source = f"profiles/{os.path.basename(f)}"
    except Exception:
        pass
    try:
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_sq_{tag or 'a20_'}counters.json")))[-1]
        sq = json.load(open(f))
        c = max((v for k, v in sq.items() if kernel_substr in k and isinstance(v, dict)), key=lambda v: v.get("SQ_INSTS_LDS", 0))
        lds = {"busy_frac": c["SQ_LDS_IDX_ACTIVE"] / (8.0 * c["SQ_BUSY_CYCLES"]),       # 256 CUs / 32 shader engines
               "conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
               "lds_instructions_per_launch": c["SQ_INSTS_LDS"], "lds_array_cycles_per_launch": c["SQ_LDS_IDX_ACTIVE"],
               "taps_per_s": samples_per_launch / kernel_seconds, "bound_taps_per_s": LDS_PEAK_TAPS,
               "frac": samples_per_launch / kernel_seconds / LDS_PEAK_TAPS,
               "source": f"profiles/{os.path.basename(f)} (counters: committed passes of this command; taps_per_s: this run)"}
    except Exception:
        pass
    return traffic, source, lds


def _time_loop(fn, steps, warmup, world):
    for _ in range(warmup):
        fn()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    return close_timed_region(t0, world)


def siddon_mode(args, world, rank, dev):
    """BASELINE config 1: scripts/images_to_sinograms.py's inner call -- tomopy.project on one 128x128 slice, 180 angles
    (one slice per call as the script does, and 50 slices per call), next to the CPU restatement on one host thread."""
    from ct_pvae_amd.helper_functions import create_sinograms
    imgs = phantoms.foam_batch(50, N_PIX, seed=rank, supersample=2)
    x = torch.from_numpy(imgs).to(dev)
    theta = phantoms.dense_theta(180)
    t1 = _time_loop(lambda: create_sinograms(x[:1], theta), args.steps, args.warmup, world) / args.steps
    t50 = _time_loop(lambda: create_sinograms(x, theta), max(args.steps // 10, 5), 3, world) / max(args.steps // 10, 5)
    if rank != 0:
        return
    from oracle import radon_oracle as orc
    orc.build()
    n, tc = 0, 0.0
    while tc < 8.0:
        t0 = time.perf_counter()
        orc.siddon_project(imgs[n % 50:n % 50 + 1], theta, pad=True)
        tc += time.perf_counter() - t0
        n += 1
    P = orc.num_proj_pix(N_PIX, N_PIX)
    print(json.dumps({"metric": "projections/sec, TomoPy-style forward (create_sinogram), 128x128 foam, 180 angles",
                      "value": world * 50 * 180 / t50, "unit": "projections/s", "n_gpus": world, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": t50 * 1e3, "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "create_sinograms: 50 slices x 180 angles x 184 bins per call (siddon_fwd_packed_kernel: 8 slices per ray walk)"},
                      "single_slice_call_us": t1 * 1e6, "single_slice_projections_per_s": 180 / t1,
                      "ray_sums_per_s_per_gpu": 50 * 180 * P / t50,
                      "cpu_baseline": {"value": n * 180 / tc, "unit": "projections/s", "cores": 1, "kind": "port",
                                       "sample": f"{n} slices x 180 angles, oracle/radon_oracle.c (TomoPy project.c "
                                                 f"restatement), 1 thread, {tc:.1f} s"}}))


def n512_mode(args, world, rank, dev):
    """BASELINE config 5: 512x512 phantoms, 90 angles, Poisson-noise forward model (pnm 1e4): forward + Gaussian-Poisson
    log-likelihood + backward.  The slice (1 MiB) does not fit LDS: the forward cuts it into 64x96 tiles, each staged once
    for all angles with 4 slices interleaved per workgroup; the backward stages an 80-bin cotangent segment per angle
    and pixel tile."""
    B, N, A = args.n512_batch, 512, 90   # B per GPU chosen to fill the chip with whole rounds of workgroups
    theta = np.pi * np.arange(A) / A
    plan = RotatePlan(theta, N, N, True, dev)
    x = torch.rand((B, N, N), device=dev)
    mask = torch.full((B, A), 1.0 / A, device=dev)
    from ct_pvae_amd.create_masks import poisson_measure
    meas = poisson_measure(plan.forward(x), mask, 1e4, seed=rank)      # the Poisson-noise forward model, on the device
    pnm = torch.tensor(1e4, device=dev)
    eps = float(np.finfo(np.float32).eps)
    sino = torch.empty((B, A, plan.PW), device=dev)
    lp, dlp, gx = torch.empty_like(sino), torch.empty_like(sino), torch.empty_like(x)
    up = torch.full((B,), -1.0 / B, device=dev)      # upstream gradient of the per-object sums (the trainer's mean loss)

    def step():
        # projection + log-likelihood, reduced to the per-object sums the loss takes inside the tiled forward's reduce pass
        # (only d lp / d projection and one partial per object, angle and 64 bins leave it), then the backward with the
        # upstream gradient of those sums applied in its own store (both halves): no elementwise pass, no
        # [B][A][P] sinogram or log-probabilities in HBM
        sums, dlp_ = plan.forward_loglik_sums(x, mask, meas, pnm, eps)
        plan.backward(dlp_, out=gx, scale=up)

    steps = max(args.steps // 10, 10)
    el = _time_loop(step, steps, 3, world)

    def ev_time(fn, n=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / n

    t_fwd = ev_time(lambda: plan.forward_loglik_sums(x, mask, meas, pnm, eps))
    dlp = plan.forward_loglik_sums(x, mask, meas, pnm, eps)[1]
    t_bwd = ev_time(lambda: plan.backward(dlp, out=gx, scale=up))
    if rank == 0:
        bytes_step = 8.0 * B * (N * N + A * plan.PW)
        bytes_fwd = 4.0 * B * (N * N + A * plan.PW)
        committed = B == 32
        traffic, traffic_source, lds = committed_counters("n512_", "rotate_fwd_tile_compact_kernel", B * A * plan.PW * plan.PW,
                                                          t_fwd) if committed else (None, None, None)
        traffic_all = None
        if traffic is not None:   # the forward is three launches: tile kernel + reduce pass + ordered sum
            try:
                pmc = json.load(open(os.path.join(ROOT, traffic_source)))["kernels"]
                traffic_all = sum(v["traffic_bytes_per_launch"] for k, v in pmc.items()
                                  if any(n_ in k for n_ in ("rotate_fwd_tile_compact_kernel", "rotate_tile_reduce_kernel<2>", "loglik_sum_partials_kernel")))
            except Exception:
                pass
        print(json.dumps({"metric": "projections/sec (fwd + log-lik + adj), 512x512, 90 angles, pnm 1e4",
                          "value": world * B * A * steps / el, "unit": "projections/s", "n_gpus": world, "steps": steps,
                          "warmup": 3, "ms_per_step": el / steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"batch={B}/GPU 512x512, 90 angles, P={plan.PW}, nearest fwd + "
                                                 "fused Gaussian-Poisson log-likelihood reduced per object in the launch + its derivative, tf_compat adj with the upstream per-object factor (tiled fwd through compact tile plans, 4 slices per workgroup; segment-staged adj)",
                                     "forward_plan": "compact tile plans" if plan._tplan is not None else "direct tiled kernel"},
                          "hbm_fraction_whole_step": bytes_step / (el / steps) / 1e9 / HBM_PEAK_GBS,
                          "roofline": {"bound": "hbm", "kernel": "rotate_fwd_tile_compact_kernel + rotate_tile_reduce_kernel<loglik, per-object sums>",
                                       "achieved": bytes_fwd / t_fwd / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": bytes_fwd / t_fwd / 1e9 / HBM_PEAK_GBS, "traffic": traffic_all,
                                       "traffic_tile_kernel_alone": traffic, "traffic_source": traffic_source, "lds": lds,
                                       "algorithmic_bytes_per_launch": bytes_fwd,
                                       "kernel_us": {"tiled_fwd_plus_reduce_loglik": t_fwd * 1e6, "segment_adj_scaled": t_bwd * 1e6},
                                       "note": "bound by the LDS gathers (ds_read_b128 of four interleaved slices, 2-way bank conflicts at oblique angles), not by HBM (DESIGN.md section 9)"}}))


def _event_graph_seconds(fn, n):
    """Average seconds per call of fn over n calls replayed from ONE HIP graph between two events (the method of
    `roofline.achieved`), median of 5."""
    fn()
    graph = capture_graph(lambda: [fn() for _ in range(n)])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    runs = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0.record()
        if graph is not None:
            graph.replay()
        else:
            for _ in range(n):
                fn()
        e1.record()
        torch.cuda.synchronize()
        runs.append(e0.elapsed_time(e1) * 1e-3 / n)
    return float(np.median(runs))


def four_modes(theta, B, N, A, dev, x, g, n=200):
    """all four (interpolation x backward) modes of the rotate projector at the headline shape -- forward and
    backward launch time (HIP events around a graph of n launches of ONE kernel), projections/s of the pair, fraction of the
    HBM roofline of each launch's algorithmic bytes.  `nearest_tf_compat` is the headline pair (`value`)."""
    out = {}
    for interp in ("nearest", "bilinear"):
        for back in ("tf_compat", "exact"):
            plan = RotatePlan(theta, N, N, True, dev, interp=interp, backward=back)
            sino = torch.empty((B, A, plan.PW), dtype=torch.float32, device=dev)
            gimg = torch.empty((B, N, N), dtype=torch.float32, device=dev)
            tf = _event_graph_seconds(lambda: plan.forward(x, out=sino), n)
            tb = _event_graph_seconds(lambda: plan.backward(g, out=gimg), n)
            bytes_dir = 4.0 * B * (N * N + A * plan.PW)
            out[f"{interp}_{back}"] = {"fwd_us": tf * 1e6, "bwd_us": tb * 1e6, "projections_per_s": B * A / (tf + tb),
                                       "hbm_frac": {"fwd": bytes_dir / tf / 1e9 / HBM_PEAK_GBS, "bwd": bytes_dir / tb / 1e9 / HBM_PEAK_GBS}}
    out["what"] = (f"B={B} {N}x{N} foam, {A} angles, 1 GPU; per-launch times from one HIP event pair around a graph of {n} launches of "
                   "one kernel; exact = the true transpose (nearest: planned gather; bilinear: inverse plan of summed weights, no "
                   "atomics); hbm_frac = 4 B (N^2 + A P) bytes / launch time / 8 TB/s")
    return out


def cold_figure(plan, x, g, sino, gimg, B, A, n_cold=96):
    """The headline pair with its inputs coming from HBM: n_cold distinct batches (more than the 256 MB Infinity Cache) walked by
    one HIP graph, the median of repeated replays."""
    xb, gb = [x], [g]
    for k in range(1, n_cold):
        xb.append(torch.roll(x, shifts=(k, 3 * k), dims=(1, 2)) * (1.0 - 0.002 * k))
        gb.append(torch.roll(g, shifts=k, dims=2).contiguous())

    def walk():
        for i in range(n_cold):
            plan.forward(xb[i], out=sino)
            plan.backward(gb[i], out=gimg)

    walk()
    graph = capture_graph(walk)
    regions = []
    while (sum(regions) < 0.05 and len(regions) < 1000) or len(regions) < 3:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        (graph.replay if graph is not None else walk)()
        torch.cuda.synchronize()
        regions.append(time.perf_counter() - t0)
    sec = float(np.median(regions))
    return {"value": B * A * n_cold / sec, "ms_per_step": sec / n_cold * 1e3, "repeats": len(regions),
            "what": f"{n_cold} distinct batches ({n_cold * (x.numel() + g.numel()) * 4 / 1e6:.0f} MB > the 256 MB Infinity Cache) walked "
                    "by one HIP graph per region: every step's objects and cotangents come from HBM"}


def project_scaling_mode(args, dev):
    """Strong scaling of BASELINE configs[3] projected from ONE GPU: the fixed batch splits into equal shares (the path has no
    data-path collective), so N ranks take the time of ONE share of total / N objects.  Timed here: each share's
    forward + backward (graph replay, HIP events); printed: the implied speed-ups t(total) / t(total / N)."""
    total, A, N = (args.total_batch or 400), args.angles, N_PIX
    theta_dense = phantoms.dense_theta(180)
    theta = theta_dense[phantoms.sparse_angle_indices(180, A)] if A < 180 else theta_dense
    plan = RotatePlan(theta, N, N, True, dev, interp="nearest", backward="tf_compat", plan_format=args.plan_format)
    rng = np.random.default_rng(0)
    shares = {}
    for ranks in (1, 2, 4, 8):
        lo, hi = sharding.shard_range(total, 0, ranks)      # rank 0 holds the largest share
        b = hi - lo
        x = torch.from_numpy(rng.random((b, N, N), dtype=np.float32)).to(dev)
        g = torch.from_numpy(rng.standard_normal((b, A, plan.PW)).astype(np.float32)).to(dev)
        sino, gimg = torch.empty_like(g), torch.empty_like(x)
        n = 100 if b * A <= 20000 else 30
        tf = _event_graph_seconds(lambda: plan.forward(x, out=sino), n)
        tb = _event_graph_seconds(lambda: plan.backward(g, out=gimg), n)
        shares[str(ranks)] = {"objects_per_rank": b, "fwd_us": tf * 1e6, "bwd_us": tb * 1e6, "step_us": (tf + tb) * 1e6,
                              "forward_kernel": plan.forward_kernel_name(b), "backward_kernel": plan.backward_kernel_name(b)}
    t1 = shares["1"]["step_us"]
    for ranks in ("1", "2", "4", "8"):
        shares[ranks]["projected_speedup"] = t1 / shares[ranks]["step_us"]
        shares[ranks]["projected_projections_per_s"] = total * A / (shares[ranks]["step_us"] * 1e-6)
    limit = min(("2", "4", "8"), key=lambda r: shares[r]["projected_speedup"] / int(r))
    print(json.dumps({"metric": "projected strong scaling of the rotate nearest fwd + tf_compat adj pair (single-GPU timings)",
                      "value": shares["8"]["projected_speedup"], "unit": "x at 8 GPUs (projected)", "n_gpus": 1, "scaling": "strong (projected)",
                      "higher_is_better": True, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": f"total batch {total} x {N}x{N}, {A} angles, split over 1 / 2 / 4 / 8 ranks "
                                             "(sharding.shard_range; no data-path collective)"},
                      "shares": shares, "least_efficient_share": limit,
                      "note": "a projection, NOT a multi-GPU measurement: N ranks each run one share concurrently; the data-parallel "
                              "trainer adds one ~3 MB gradient all-reduce per step (not included)"}))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world, rank, local = dist_setup(args.gpus)
    dev = torch.device("cuda", local)
    if args.mode == "train":
        return train_mode(args, world, rank, dev)
    if args.mode == "siddon":
        return siddon_mode(args, world, rank, dev)
    if args.mode == "n512":
        return n512_mode(args, world, rank, dev)
    if args.project_scaling:
        if world > 1:
            raise SystemExit("--project-scaling is a single-GPU projection")
        return project_scaling_mode(args, dev)
    B, N, A = args.batch, N_PIX, args.angles
    strong = args.total_batch > 0
    if strong:      # a FIXED batch shared by the ranks: rank r projects objects [lo, hi) of it
        lo, hi = sharding.shard_range(args.total_batch, rank, world)
        B = hi - lo
        if B == 0:
            raise SystemExit(f"--total-batch {args.total_batch} leaves rank {rank} of {world} without an object")

    theta_dense = phantoms.dense_theta(180)
    theta = theta_dense[phantoms.sparse_angle_indices(180, A)] if A < 180 else theta_dense
    imgs = phantoms.foam_batch(B, N, seed=rank, supersample=2)
    plan = RotatePlan(theta, N, N, True, dev, interp="nearest", backward="tf_compat", plan_format=args.plan_format)
    P = plan.PW
    g_host = np.random.default_rng(1000 + rank).standard_normal((B, A, P)).astype(np.float32)
    x = torch.from_numpy(imgs).to(dev)
    g = torch.from_numpy(g_host).to(dev)
    sino = torch.empty((B, A, P), dtype=torch.float32, device=dev)
    gimg = torch.empty((B, N, N), dtype=torch.float32, device=dev)
    # Cache state of the inputs.  Default ("cache-warm"): every step re-projects the SAME resident batch -- 3.3 MB of objects
    # and 0.7 MB of cotangents that stay in the 256 MB Infinity Cache between steps.  --cold cycles n_cold distinct batches
    # (the first is the foam batch, the others are it rolled and rescaled: same statistics, distinct memory) whose total
    # exceeds the Infinity Cache, so every step's inputs come from HBM.
    n_cold = 96 if args.cold else 1
    x_bank, g_bank = [x], [g]
    for k in range(1, n_cold):
        x_bank.append(torch.roll(x, shifts=(k, 3 * k), dims=(1, 2)) * (1.0 - 0.002 * k))
        g_bank.append(torch.roll(g, shifts=k, dims=2).contiguous())
    cursor = [0]

    bucket = torch.zeros(711164, dtype=torch.float32, device=dev) if (args.grad_allreduce or world > 1) else None

    def projector_step():
        i = cursor[0]
        cursor[0] = (i + 1) % n_cold
        plan.forward(x_bank[i], out=sino)
        plan.backward(g_bank[i], out=gimg)

    def step():
        projector_step()
        if args.grad_allreduce and world > 1:
            torch.distributed.all_reduce(bucket)                 # RCCL over xGMI: the bucket is already flat

    for _ in range(max(args.warmup, n_cold if args.cold else 0)):
        step()
    # The two launches of a step take ~6 us each -- about what one Python call of them costs the host -- so the timed
    # loop replays HIP graphs of up to --graph-steps steps (two kernel nodes per step: the same launches in the same order) and
    # launches only the remaining steps from Python: the GPU runs back to back whatever the host's speed.  A replay itself
    # costs ~4 us of stream time (measured, round 3: 12.76 us per step with graphs of 10 steps, 12.37 us with 50 or 100,
    # 12.28 us with 250).  Not with the RCCL bucket.
    chunk = n_cold if args.cold else max(1, min(args.graph_steps, args.steps))   # (cold: one graph walks all the distinct batches once)
    graph = None
    cursor[0] = 0
    if not args.no_graph and not args.grad_allreduce and args.steps >= chunk:
        graph = capture_graph(lambda: [step() for _ in range(chunk)])
        cursor[0] = 0
    n_replay, n_eager = divmod(args.steps, chunk) if graph is not None else (0, args.steps)

    def timed_region():
        """EXACTLY K steps between barrier + synchronize on both sides; the maximum over ranks."""
        barrier_sync(world)
        t0 = time.perf_counter()
        for _ in range(n_replay):
            graph.replay()
        for _ in range(n_eager):
            step()
        return close_timed_region(t0, world)

    # One region of K = 20 steps lasts 0.3 ms: too short to hang a headline on.  The region is therefore repeated until
    # --min-ms of measured time has accumulated (every rank takes the same decision: the region times are already
    # maxima over ranks) and the MEDIAN region is reported; `steps` stays K, `repeats` says how many regions were timed.
    regions = [timed_region()]
    while sum(regions) * 1e3 < args.min_ms and len(regions) < 10000:
        regions.append(timed_region())
    elapsed = float(np.median(regions))

    # ---- multi-GPU: the same K steps WITH the data-parallel trainer's one collective ------------------------------------
    # north_star: "batches of objects shard across the GPUs of one node with RCCL all-reduce of VAE gradients".  The projector
    # itself needs no collective (value above); a data-parallel training step also sums ONE flat fp32 bucket of the P-VAE's
    # 711,164 gradients (2.8 MB) over the ranks.  Reported beside `value`, never instead of it: every step = the projector
    # pair (replayed from a one-step HIP graph) followed by the bucket's all-reduce on the same stream.
    with_allreduce = None
    if world > 1 and not args.grad_allreduce:
        import torch.distributed as dist
        cursor[0] = 0
        g1 = None if (args.no_graph or args.cold) else capture_graph(projector_step)
        cursor[0] = 0

        def ar_region():
            barrier_sync(world)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                (g1.replay if g1 is not None else projector_step)()
                dist.all_reduce(bucket)
            return close_timed_region(t0, world)

        ar_region()
        ar_regions = [ar_region()]
        while sum(ar_regions) * 1e3 < args.min_ms and len(ar_regions) < 10000:
            ar_regions.append(ar_region())
        ev_ar = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        barrier_sync(world)
        for e0, e1 in ev_ar:
            e0.record()
            dist.all_reduce(bucket)
            e1.record()
        torch.cuda.synchronize()
        with_allreduce = {"seconds": float(np.median(ar_regions)), "repeats": len(ar_regions),
                          "allreduce_us": float(np.median([e0.elapsed_time(e1) * 1e3 for e0, e1 in ev_ar])),
                          "backend": dist.get_backend(), "ranks": dist.get_world_size()}

    # ---- per-kernel durations, HIP events on the launch stream (torch's current stream) -------------------
    # One event pair brackets n_ev back-to-back launches of ONE kernel: the average duration of a launch in a stream of
    # them.  (An event pair per launch does not work on this stack: two events with nothing between them read ~5 us
    # apart, and subtracting that under-reads the kernel.)  rocprofv3's per-dispatch average (profiles/) is the span
    # first-wave-start -> last-wave-end of one dispatch, which can overlap its neighbours' ramp and drain: it reads a few
    # per cent off these (8.4 / 6.7 us against 8.1 / 6.8 us here), whose sum matches the measured step.
    n_ev = min(max(args.steps, 50), 400)

    def avg_launch_seconds(fn):
        """The n_ev launches are captured once into a HIP graph and replayed between the two events, so the figure is
        the GPU's back-to-back launch duration whatever the host's speed (a Python call of a 6-7 us kernel costs about
        as much on the host: timed from a Python loop the shorter kernel reads as host time on a slow box)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()

        def batch():
            for _ in range(n_ev):
                fn()

        replay = None if args.no_graph else capture_graph(batch)
        runs = []
        for _ in range(5):
            torch.cuda.synchronize()
            e0.record()
            (replay.replay if replay is not None else batch)()
            e1.record()
            torch.cuda.synchronize()
            runs.append(e0.elapsed_time(e1) * 1e-3 / n_ev)
        return float(np.median(runs))

    def cycling(fn_of_index):
        state = [0]

        def fn():
            i = state[0]
            state[0] = (i + 1) % n_cold
            fn_of_index(i)
        return fn

    t_fwd = avg_launch_seconds(cycling(lambda i: plan.forward(x_bank[i], out=sino)))
    t_bwd = avg_launch_seconds(cycling(lambda i: plan.backward(g_bank[i], out=gimg)))

    # spread of single steps (median and p10/p90): one event pair per fwd+adj step, 200 steps; each figure
    # carries the ~1.5 us of its own event pair, so read it as a distribution, not as `ms_per_step`
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    torch.cuda.synchronize()
    for e0, e1 in ev:
        e0.record()
        step()
        e1.record()
    torch.cuda.synchronize()
    step_us = np.array([e0.elapsed_time(e1) * 1e3 for e0, e1 in ev])

    # ---- the same step through the public autograd API (secondary) -----------------------------------------
    # project_tf_fast(...).backward() and calculate_log_prob_M_given_R(...).backward() -- what the reference's callers
    # use (ctvae/helper_functions.py:359) -- against the raw operator pair launched from Python, at this batch and at the
    # training batch (5).  Host-bound: PyTorch's autograd machinery around ONE custom node costs ~25 us, and its default
    # multi-threaded engine adds a ~50 us hand-off to the device thread per backward() call, which a training step pays
    # once for its whole graph -- `single_thread_engine` is the same call under
    # torch.autograd.set_multithreading_enabled(False), as ct_pvae_amd/trainer.py runs its backward.
    from ct_pvae_amd.helper_functions import calculate_log_prob_M_given_R
    n_api = min(max(args.steps, 50), 300)

    def host_rate(fn, n_obj):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(n_api):
            fn()
        torch.cuda.synchronize()
        return n_obj * A * n_api / (time.perf_counter() - ta)

    api = {"unit": "projections/s", "what": "fwd+bwd through torch.autograd, launched from Python, 1 GPU; raw = "
                                            "RotatePlan.forward/backward launched from Python with preallocated outputs; "
                                            "raw_loglik = forward_loglik + the caller's per-object sum + scaled backward, "
                                            "likewise; times_raw = how many times the raw launches' time each call takes"}
    eps32 = float(np.finfo(np.float32).eps)
    for nb in sorted({B, 5}):
        x4 = x[:nb, :, :, None].clone().requires_grad_(True)
        g4 = g[:nb, :, :, None].contiguous()
        xs, gs = x[:nb].contiguous(), g[:nb].contiguous()
        so, go = torch.empty_like(gs), torch.empty_like(xs)
        mask = torch.full((nb, A), 1.0 / A, device=dev)
        meas = torch.rand((nb, A, P), device=dev)
        pnm = torch.tensor(1e4, device=dev)
        w = torch.full((nb,), -1.0 / nb, device=dev)

        def raw_step():
            plan.forward(xs, out=so)
            plan.backward(gs, out=go)

        lpo, dlpo = torch.empty_like(gs), torch.empty_like(gs)

        def raw_lik_step():      # the launches calculate_log_prob_M_given_R(...).sum(...).backward(w) ends in, from Python
            plan.forward_loglik(xs, mask, meas, pnm, eps32, out=so, out_lp=lpo, out_dlp=dlpo)
            lpo.sum(dim=(1, 2))
            plan.backward(dlpo, out=go, scale=w)

        def api_step():
            x4.grad = None
            project_tf_fast(x4, theta, pad=True, dim=2, integrate_vae=True).backward(g4)

        def lik_step():
            x4.grad = None
            calculate_log_prob_M_given_R(x4, mask, meas, pnm, eps32, theta=theta, pad=True).sum(dim=(1, 2, 3)).backward(w)

        ent = {"raw": host_rate(raw_step, nb), "raw_loglik": host_rate(raw_lik_step, nb),
               "project_tf_fast": host_rate(api_step, nb),
               "calculate_log_prob_M_given_R": host_rate(lik_step, nb)}
        with torch.autograd.set_multithreading_enabled(False):
            ent["single_thread_engine"] = {"project_tf_fast": host_rate(api_step, nb),
                                           "calculate_log_prob_M_given_R": host_rate(lik_step, nb)}
        ent["times_raw"] = {"project_tf_fast": ent["raw"] / ent["project_tf_fast"],
                            "calculate_log_prob_M_given_R": ent["raw_loglik"] / ent["calculate_log_prob_M_given_R"],
                            "single_thread_engine": {
                                "project_tf_fast": ent["raw"] / ent["single_thread_engine"]["project_tf_fast"],
                                "calculate_log_prob_M_given_R":
                                    ent["raw_loglik"] / ent["single_thread_engine"]["calculate_log_prob_M_given_R"]}}
        api[f"batch_{nb}"] = ent
    api["value"] = api[f"batch_{B}"]["project_tf_fast"]
    from ct_pvae_amd import _lib as _cl
    api["autograd_node"] = "C++ (ct_pvae_amd/csrc/torch_node.cpp)" if _cl.torch_node() is not None else "Python"

    if rank != 0:
        return
    bytes_dir = 4.0 * B * (N * N + A * P)                 # one direction: read once + write once (fp32)
    fwd_name = plan.forward_kernel_name(B)
    bwd_name = plan.backward_kernel_name(B)
    dom = (fwd_name, t_fwd) if t_fwd >= t_bwd else (bwd_name, t_bwd)
    achieved = bytes_dir / dom[1] / 1e9
    # HBM-side bytes per launch of the dominant kernel: a COMMITTED measurement (separate rocprofv3 --pmc passes of this
    # command, tools/collect_profiles.sh -> profiles/rNN_traffic_pmc.json), not something this run can observe itself
    traffic, traffic_source, lds = None, None, None
    if (B, N) == (50, N_PIX) and A in (20, 180) and args.plan_format == "auto" and not args.cold:
        # committed for the two workloads BASELINE names at this size: the headline (tag "") and config 4's per-GPU share ("angles180_")
        traffic, traffic_source, lds = committed_counters("" if A == 20 else "angles180_", dom[0],
                                                          B * A * (P * P if dom[0] == fwd_name else N * N), dom[1])
        _, _, lds_fwd = committed_counters("" if A == 20 else "angles180_", fwd_name, B * A * P * P, t_fwd)
        if lds_fwd is not None and dom[0] != fwd_name:
            lds = dict(lds or {}, forward=lds_fwd)
    from ct_pvae_amd import _lib
    tune_env = {k: v for k, v in os.environ.items() if k.startswith("CTPVAE_")}
    total_objects = args.total_batch if strong else world * B
    proj_per_s = total_objects * A * args.steps / elapsed
    # the four (interp x backward) modes and the cold-input figure beside the headline (one GPU, the headline shape)
    modes, cold = None, None
    if world == 1 and not args.no_modes and not args.cold and not strong:
        modes = four_modes(theta, B, N, A, dev, x, g)
        cold = cold_figure(plan, x, g, sino, gimg, B, A)
    out = {
        "metric": "projections/sec (fwd+adj) 128x128 foam, 20 angles; fraction of HBM roofline",
        "value": proj_per_s, "unit": "projections/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "repeats": len(regions),
        "region_ms": {"min": min(regions) * 1e3, "median": elapsed * 1e3, "max": max(regions) * 1e3,
                      "what": f"{len(regions)} timed regions of exactly {args.steps} steps each (barrier + synchronize on both "
                              "sides, max over ranks); value and ms_per_step are from the median region"},
        "developer_knobs": {"library_knobs_set": _lib.load().ctpvae_tune_active(), "env": tune_env},
        "config": {"workload": (f"total batch {args.total_batch} shared by {world} GPU(s) ({B} on rank 0), " if strong else f"batch={B}/GPU ")
                               + f"{N}x{N} foam, {A} angles, P={P}, rotate nearest fwd + tf_compat adj",
                   "objects_per_gpu": B, "n_pixel": N, "angles": A, "num_proj_pix": P, "parallelism": f"batch-shard x{world}",
                   "inputs": (f"cold: {n_cold} distinct batches cycled, {n_cold * (x.numel() + g.numel()) * 4 / 1e6:.0f} MB > the "
                              "256 MB Infinity Cache" if args.cold else
                              "cache-warm: ONE resident batch re-projected every step (3.3 MB of objects + 0.7 MB of cotangents "
                              "stay in the Infinity Cache; --cold cycles 96 distinct batches)"),
                   "forward_plan": "compact (2 bits per row)" if plan.dense_plan(B)[1] else "u16 taps",
                   "grad_allreduce_bytes_per_step": 4 * 711164 if args.grad_allreduce else 0,
                   "launch": (f"{n_replay} replays of a HIP graph of {chunk} steps + {n_eager} steps launched from Python"
                              if graph is not None else "every step launched from Python")},
        "ray_sums_per_s_per_gpu": proj_per_s * P / world,
        "hbm_fraction_whole_step": (2 * bytes_dir / (elapsed / args.steps)) / 1e9 / HBM_PEAK_GBS,
        "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy": achieved / HBM_COPY_GBS,
                     "traffic": traffic, "traffic_source": traffic_source, "lds": lds, "algorithmic_bytes_per_launch": bytes_dir,
                     "kernel_us": {"rotate_fwd": t_fwd * 1e6, "rotate_bwd_tfcompat": t_bwd * 1e6},
                     "note": "object lives in LDS for all angles; the launch is bound by dispatch + L2->CU bytes (indices, fills) + per-task LDS latency chains, so the HBM fraction is small by construction (DESIGN.md section 6)"},
        "samples_per_s": {"fwd": B * A * P * P / t_fwd, "bwd": B * A * N * N / t_bwd},
        "step_us_event_pairs": {"p10": float(np.percentile(step_us, 10)), "median": float(np.median(step_us)),
                                "p90": float(np.percentile(step_us, 90)), "n": len(step_us)},
        "api": api,
    }
    if modes is not None:
        out["modes"] = modes
        out["cold"] = cold
    if with_allreduce is not None:
        out["value_with_grad_allreduce"] = world * B * A * args.steps / with_allreduce["seconds"]
        out["ms_per_step_with_grad_allreduce"] = with_allreduce["seconds"] / args.steps * 1e3
        out["allreduce_us"] = with_allreduce["allreduce_us"]
        out["rccl_ranks"] = with_allreduce["ranks"]
        out["collective_backend"] = with_allreduce["backend"] + (" (rehearsal on one GPU over gloo: not a scaling measurement)"
                                                                 if os.environ.get("CTPVAE_REHEARSE_ONE_GPU") else " = RCCL over xGMI")
        out["grad_allreduce"] = {"bytes_per_step": 4 * 711164, "repeats": with_allreduce["repeats"],
                                 "what": "the same K steps, each followed by ONE all-reduce of the P-VAE's flat fp32 gradient bucket "
                                         "(711,164 floats) over all ranks on the projector's stream -- the data-parallel trainer's "
                                         "only collective (ct_pvae_amd/sharding.py); `value` is the projector pair alone"}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(imgs, theta, g_host, args.cpu_threads)
    print(json.dumps(out))


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()
