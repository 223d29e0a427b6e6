"""Developer timing: the 512 x 512 tile forward with its three task shapes -- (angle, 64-slot block) tasks (TILED_SORT=0), sorted
single bands (TILED_PAIR=0), sorted band pairs (default, round 4) -- HIP-graph replays, alternating, after a warm-up."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
if os.environ.get("CTPVAE_VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CTPVAE_VARIANT_LIB"])
    _lib.torch_node = lambda: None
print("library:", _lib.LIB_PATH)
from ct_pvae_amd.forward_functions import RotatePlan
d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
theta = np.pi * np.arange(90) / 90
_lib.tune("TILED_PAIR", 1)      # the plan with the pair sections
plan = RotatePlan(theta, 512, 512, True, d)
_lib.tune("*")
x = torch.rand((B, 512, 512), device=d)
out = torch.empty((B, 90, plan.PW), device=d)
def timed(n=20):
    plan.forward(x, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): plan.forward(x, out=out)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(r))
ref = plan.forward(x).clone()
for rnd in range(3):
    for name, knobs in (("pairs", {"TILED_PAIR": 1}), ("sorted bands (default)", {}),
                        ("sorted bands, round-3 workgroup order", {"TILED_XCD": 0}), ("blocks", {"TILED_SORT": 0}),
                        ("sorted bands, 12 waves", {"TILED_WAVES": 12}), ("sorted bands, 8 waves", {"TILED_WAVES": 8})):
        for k, v in knobs.items(): _lib.tune(k, v)
        t = timed()
        same = torch.equal(plan.forward(x), ref)
        _lib.tune("*")
        print(f"round {rnd} B={B} forward + reduce, {name:38s}: {t:7.2f} us  {'equal' if same else 'DIFFER'}", flush=True)
