"""The u16 planned forward at 128 x 128 over batch sizes around whole rounds of workgroups: the two-part piece list (coarse task groups for whole
rounds, finer ones behind them; knob MIXG, default) against one cut for the whole launch (MIXG=0), bit-compared; and the
task-group count G swept for both.   python tools/time_rounds.py [A ...]   (CTPVAE_VARIANT_LIB for other builds)"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib, phantoms
if os.environ.get("CTPVAE_VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CTPVAE_VARIANT_LIB"]); _lib.torch_node = lambda: None
from ct_pvae_amd.forward_functions import RotatePlan
from time_modes import graph_time
dev = torch.device('cuda', 0)
angles = [int(a) for a in sys.argv[1:]] or [20, 180]
sweep = os.environ.get("SWEEP_G", "")
for A in angles:
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
    plan = RotatePlan(theta, 128, 128, True, dev, plan_format="u16")
    for B in (50, 100, 150, 200, 256, 300, 350, 400, 450, 512, 600):
        x = torch.rand((B, 128, 128), device=dev)
        o0, o1 = torch.empty((B, A, plan.PW), device=dev), torch.empty((B, A, plan.PW), device=dev)
        n = 100 if B * A < 20000 else 20
        with _lib.tuned("MIXG", 0):
            t0 = graph_time(lambda: plan.forward(x, out=o0), n) * 1e6
        t1 = graph_time(lambda: plan.forward(x, out=o1), n) * 1e6
        line = f"A={A:3d} B={B:3d}  one cut {t0:7.2f} us   two-part list {t1:7.2f} us   {'equal' if torch.equal(o0, o1) else 'DIFFER'}"
        if sweep:
            for G in (1, 2, 3, 4, 5, 6, 8):
                with _lib.tuned("G", G), _lib.tuned("NS", 2):
                    with _lib.tuned("MIXG", 0):
                        a = graph_time(lambda: plan.forward(x, out=o0), n) * 1e6
                    b = graph_time(lambda: plan.forward(x, out=o1), n) * 1e6
                line += f" | G={G} {a:.1f}/{b:.1f}"
        print(line, flush=True)
