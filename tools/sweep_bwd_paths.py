"""Developer sweep: the three bit-identical nearest / tf_compat backward paths of 128 x 128 slices -- planned gather, stepped segment
kernel (64 x 32 tiles over the step plan), direct segment kernel -- and what RotatePlan picks, over batch and angle count
(HIP-graph replays, one event pair per 200 launches)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ct_pvae_amd import phantoms, _lib
from ct_pvae_amd.forward_functions import RotatePlan
d = torch.device("cuda", 0)


def timeit(fn, n=200):
    fn(); torch.cuda.synchronize()
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n // 20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n // 20 * 20) * 1e3


print("us per launch:      planned  stepped   direct   chosen")
for A in (20, 45, 90, 180):
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
    for B in (50, 80, 128, 160, 200, 256, 320, 400):
        auto = RotatePlan(theta, 128, 128, True, d)
        g = torch.randn((B, A, auto.PW), device=d)
        out = torch.empty((B, 128, 128), device=d)
        ref = auto.backward(g).clone()
        planned = RotatePlan(theta, 128, 128, True, d)
        planned.backward_uses_step_plan = lambda S: False
        planned.backward_uses_plan = lambda S: True
        seg = RotatePlan(theta, 128, 128, True, d)
        seg.backward_uses_step_plan = lambda S: True
        t_p = timeit(lambda: planned.backward(g, out=out)); ok = torch.equal(out, ref)
        _lib.tune("SEG_PPT", 8); t_s = timeit(lambda: seg.backward(g, out=out)); ok = ok and torch.equal(out, ref)
        _lib.tune("SEG_PPT", 4); t_d = timeit(lambda: seg.backward(g, out=out)); ok = ok and torch.equal(out, ref)
        _lib.tune("SEG_PPT")
        t_a = timeit(lambda: auto.backward(g, out=out))
        print(f"B={B:3d} A={A:3d}:  {t_p:8.2f} {t_s:8.2f} {t_d:8.2f}   {t_a:7.2f} ({auto.backward_kernel_name(B).replace('rotate_bwd_', '').replace('_kernel', '')})"
              f"{'' if ok else '  NOT EQUAL'}", flush=True)
