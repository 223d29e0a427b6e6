"""Developer tool: the training loop's projector call, alone, for a kernel trace.

    cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -o t -- python3 <repo>/tools/trace_training_call.py [steps]

Runs `steps` (default 50) iterations of what find_loss_vae_unsup does with the projector (ctvae/helper_functions.py:336-368
under ctvae/main_ct_vae.py:471-481): calculate_log_prob_M_given_R on ns * b = 10 objects with a fresh random 20-angle subset
of the dataset's 180 angles per step, then backward of the per-object sums.  With ONE dense plan and the angle-index
operand the trace must show the plan kernels (rotate_cplan_kernel, rotate_cplan_class_list_kernel, rotate_bwd4_plan_kernel) ONCE,
whatever the number of steps, and per step exactly: one rotate_fwd_compact_kernel (projection + log-likelihood + per-task
partial sums), one loglik_sum_partials_kernel, one rotate_bwd_planned_sel_kernel -- no at::native::reduce_kernel over the
log-probabilities, no per-step copy of the angle subset (profiles/r03_training_call_kernel_stats.csv)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ct_pvae_amd as cp  # noqa: E402
from ct_pvae_amd import phantoms  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
d = torch.device("cuda", 0)
rng = np.random.default_rng(0)
theta = phantoms.dense_theta(180).astype(np.float32)
B, N, P = 10, 128, 184
mask = torch.from_numpy(((rng.random((B, 180)) > 0.5) / 20).astype(np.float32)).to(d)
meas = torch.from_numpy((rng.random((B, 180, P)) * 3).astype(np.float32)).to(d)
x = torch.rand((B, N, N, 1), device=d, requires_grad=True)
pnm = torch.tensor(1e4, device=d)
subsets = [rng.permutation(180)[:20].astype(np.int32) for _ in range(steps)]      # host-resident, as the trainer draws them
w = torch.full((B,), -1.0 / B, device=d)          # the upstream gradient of the per-object sums (the trainer's mean loss)
torch.cuda.synchronize()
for ai in subsets:
    x.grad = None
    lp = cp.calculate_log_prob_M_given_R(x, mask, meas, pnm, 1e-7, theta=theta, angles_i=ai, pad=True, reduce="per_object")
    lp.backward(w)
torch.cuda.synchronize()
print("done", steps)
