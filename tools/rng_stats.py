import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import phantoms
from ct_pvae_amd.forward_functions import RotatePlan
for A in (20, 180):
    theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
    plan = RotatePlan(theta, 128, 128, True, torch.device('cuda', 0))
    buf = plan._fwd_plan.cpu().numpy()
    PW, nJB, PWpad = 184, 3, 192
    al = lambda x: (x + 255) // 256 * 256
    off_first = al(A * 4 + 2 * (A + 1) * 4)
    off_rng = al(off_first + A * PWpad * 4)
    rng = buf[off_rng:off_rng + A * nJB * 8].view(np.int32).reshape(A, nJB, 2)
    first, last = rng[..., 0], 0x7f7f7f7f - rng[..., 1]
    ng = np.where(last >= first, last - first + 1, 0)
    print(f"A={A}: groups per task mean {ng.mean():.2f} (central block {ng[:,0].mean():.2f}, next {ng[:,1].mean():.2f}, outer {ng[:,2].mean():.2f}), NG=23")
