"""A/B of the bilinear forward on one box: in-tree library, or CTPVAE_VARIANT_LIB=tools/libctpvae_radon_<tag>.bin."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib
if os.environ.get("CTPVAE_VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CTPVAE_VARIANT_LIB"]); _lib.torch_node = lambda: None
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
d = torch.device('cuda', 0)
shapes = ((50, 128, 20), (5, 128, 20), (50, 128, 180), (100, 128, 20), (400, 128, 180), (200, 128, 90), (32, 512, 90), (8, 512, 90), (64, 64, 60), (16, 256, 90))
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, N, A in shapes:
    theta = np.pi * np.arange(A) / A
    plan = RotatePlan(theta, N, N, True, d, interp="bilinear")
    x = torch.rand((B, N, N), device=d); out = torch.empty((B, A, plan.PW), device=d)
    n = 100 if B * N * N * A < 5e8 else 10
    t = [graph_time(lambda: plan.forward(x, out=out), n) * 1e6 for _ in range(3)]
    print(os.path.basename(_lib.LIB_PATH), f"B={B} N={N} A={A}:", " ".join(f"{v:.2f}" for v in t), "us", flush=True)
