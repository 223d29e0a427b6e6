"""ct_pvae_amd -- MI355X-native Radon forward/back-projector behind CT_PVAE's physics-decoder call signatures.

Host code is Python + PyTorch-ROCm (device memory, streams, autograd); the operators are hand-written gfx950 HIP
kernels reached through the C ABI of include/ctpvae_radon.h.  There is no CPU path.
"""
from .forward_functions import (RotatePlan, as_angle_index, num_proj_pix, pad_amounts, pad_phantom,  # noqa: F401
                                project_tf_fast, project_tf_low_mem, rotate_tables)
from .helper_functions import (calculate_log_prob_M_given_R, create_sinogram, create_sinograms,  # noqa: F401
                               gaussian_poisson_log_prob)
from .create_masks import create_all_masks  # noqa: F401
from .fbp import iradon, iradon_all  # noqa: F401
from .recon import crop, evaluate_sinogram, recon, siddon_backproject  # noqa: F401

__version__ = "0.2.0"
