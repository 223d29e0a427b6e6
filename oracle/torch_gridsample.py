"""A SECOND, independent CPU implementation of the reference's rotate-and-sum projector, on PyTorch's own resampler.

TEST INFRASTRUCTURE ONLY (like radon_oracle.c): imported by tests/, tests/golden/make_gridsample_crosscheck.py and
bench.py's cpu_baseline leg, never by ct_pvae_amd/.

What the reference computes (ctvae/forward_functions.py:92-121): pad -> tfa.image.rotate(imgs, -theta) -> reduce_sum over
image rows.  Here the rotation is torch.nn.functional.affine_grid + grid_sample (align_corners=True, zeros padding) on
the CPU: a framework resampler like TensorFlow's, written by other people, with its own coordinate arithmetic
(normalised coordinates, un-normalised again inside the sampler; nearest = round-half-to-even).  It is therefore NOT
bit-comparable with TensorFlow or with oracle/radon_oracle.c -- what it pins is everything a mis-remembered convention
would break: the sense of the rotation, the centre ((P-1)/2), which axis is summed, the pad rule, zero fill, and
bilinear weights to fp32 rounding.  tests/test_oracle.py holds the measured agreement (max rel-err for bilinear, number
of differing samples for nearest) against the committed fixture.

The same code is BASELINE.md section 5's "torch-CPU" baseline: the framework-op implementation comparable to the TF graph
(template: ctvae/tomopy_forward_compare.py:51-67 times project_tf_fast / project_tf_low_mem / tomopy side by side).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def num_proj_pix(nx, ny):
    """ctvae/forward_functions.py:29-30"""
    return int(math.ceil((math.sqrt(float(nx * nx + ny * ny)) + 2.0) / 2.0) * 2)


def pad_phantom(img):
    """[S][X][Y] -> [S][P][P], zeros, odd remainder on the high side (ctvae/forward_functions.py:32-45)."""
    nx, ny = img.shape[1], img.shape[2]
    P = num_proj_pix(nx, ny)
    xl, yl = (P - nx) // 2, (P - ny) // 2
    return F.pad(img, (yl, P - ny - yl, xl, P - nx - xl))


def rotate_and_sum(img, theta, pad=True, mode="bilinear"):
    """img [S][X][Y] float32 tensor (CPU), theta [A] radians -> sinograms [S][A][P] (differentiable).

    Output pixel (x, y) of angle a reads the input at  centre + R(-theta_a) (x - cx, y - cy)  -- the transform row
    tfa.image.rotate(imgs, -theta) builds (cos, -sin, x_off, sin, cos, y_off) -- then rows are summed (axis 1 of the
    reference's [A][P][P][B] tensor = image rows)."""
    x = pad_phantom(img) if pad else img
    S, PH, PW = x.shape
    ang = -torch.as_tensor(np.asarray(theta, dtype=np.float32))
    c, s = torch.cos(ang), torch.sin(ang)
    A = ang.numel()
    # normalised coordinates (align_corners=True: -1 and +1 are the centres of the corner pixels); for a non-square canvas
    # the x and y scales differ: x_in_n = c x_n - s (PH-1)/(PW-1) y_n ;  y_in_n = s (PW-1)/(PH-1) x_n + c y_n
    rx = (PH - 1) / (PW - 1) if PW > 1 else 0.0
    ry = (PW - 1) / (PH - 1) if PH > 1 else 0.0
    mat = torch.zeros((A, 2, 3), dtype=torch.float32)
    mat[:, 0, 0], mat[:, 0, 1] = c, -s * rx
    mat[:, 1, 0], mat[:, 1, 1] = s * ry, c
    grid = F.affine_grid(mat, (A, S, PH, PW), align_corners=True)             # [A][PH][PW][2]
    rot = F.grid_sample(x[None].expand(A, S, PH, PW), grid, mode=mode, padding_mode="zeros", align_corners=True)
    return rot.sum(dim=2).permute(1, 0, 2)                                     # [S][A][PW]


def fwd_and_grad(img, theta, g, pad=True, mode="bilinear"):
    """Forward and the autograd gradient w.r.t. the image for cotangent g [S][A][P] (numpy in, numpy out)."""
    x = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).requires_grad_(True)
    sino = rotate_and_sum(x, theta, pad=pad, mode=mode)
    sino.backward(torch.from_numpy(np.ascontiguousarray(g, dtype=np.float32)))
    return sino.detach().numpy(), x.grad.numpy()
