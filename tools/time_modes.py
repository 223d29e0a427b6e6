"""The four (interp x backward) modes of the rotate projector, one HIP event pair around a graph of n
back-to-back launches of ONE kernel each -- the method of bench.py's roofline figure.

    python tools/time_modes.py [--shapes headline,a180,n512] [--n 200]

Prints one line per (shape, mode): forward / backward microseconds per launch, projections/s of the pair, and the
fraction of the 8 TB/s HBM roofline the algorithmic bytes of a launch amount to."""
import argparse
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ct_pvae_amd.forward_functions import RotatePlan  # noqa: E402
from ct_pvae_amd import phantoms  # noqa: E402

SHAPES = {"headline": (50, 128, 20), "a180": (50, 128, 180), "b400": (400, 128, 180), "n512": (32, 512, 90),
          "n512b8": (8, 512, 90), "b5": (5, 128, 20)}
MODES = [("nearest", "tf_compat"), ("nearest", "exact"), ("bilinear", "tf_compat"), ("bilinear", "exact")]


def graph_time(fn, n):
    """Average seconds per call of fn over n calls replayed from one HIP graph (5 repeats, median)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    runs = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        runs.append(e0.elapsed_time(e1) * 1e-3 / n)
    return float(np.median(runs))


def time_modes(B, N, A, dev, n=200, modes=MODES, theta=None):
    if theta is None:
        dense = phantoms.dense_theta(180)
        theta = dense[phantoms.sparse_angle_indices(180, A)] if A < 180 and 180 % A == 0 else np.pi * np.arange(A) / A
    out = {}
    x = torch.rand((B, N, N), device=dev)
    for interp, back in modes:
        plan = RotatePlan(theta, N, N, True, dev, interp=interp, backward=back)
        g = torch.randn((B, A, plan.PW), device=dev)
        sino = torch.empty((B, A, plan.PW), device=dev)
        gimg = torch.empty((B, N, N), device=dev)
        nn = n if B * N * N * A < 5e8 else max(10, n // 10)
        tf = graph_time(lambda: plan.forward(x, out=sino), nn)
        tb = graph_time(lambda: plan.backward(g, out=gimg), nn)
        bytes_dir = 4.0 * B * (N * N + A * plan.PW)
        out[f"{interp}_{back}"] = {"fwd_us": tf * 1e6, "bwd_us": tb * 1e6, "projections_per_s": B * A / (tf + tb),
                                   "hbm_frac": {"fwd": bytes_dir / tf / 8e12, "bwd": bytes_dir / tb / 8e12}}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="headline,a180,n512")
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--modes", default="")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    modes = MODES if not args.modes else [tuple(m.split("+")) for m in args.modes.split(",")]
    res = {}
    for name in args.shapes.split(","):
        B, N, A = SHAPES[name]
        r = res[name] = time_modes(B, N, A, dev, args.n, modes)
        for k, v in r.items():
            print(f"{name:9s} B={B:<4d} N={N:<4d} A={A:<4d} {k:20s} fwd {v['fwd_us']:9.2f} us  bwd {v['bwd_us']:9.2f} us  "
                  f"{v['projections_per_s'] / 1e6:8.2f} M proj/s  hbm fwd {100 * v['hbm_frac']['fwd']:.2f} % bwd {100 * v['hbm_frac']['bwd']:.2f} %",
                  flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)
