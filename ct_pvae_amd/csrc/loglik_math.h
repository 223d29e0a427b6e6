// loglik_math.h -- the Gaussian-approximated Poisson log-probability of one sinogram sample (a8,
// ctvae/helper_functions.py:360-368), shared by the elementwise kernel (loglik.hip) and the planned forward's fused
// epilogue (rotate_plan.hip) so that both evaluate the same fp32 expression.
#pragma once
#include <hip/hip_runtime.h>

namespace ctpvae {

constexpr float kHalfLog2Pi = 0.91893853320467274178f;

// tfd.Normal(loc = proj * mask, scale = eps + sqrt(loc / pnm + eps)).log_prob(x), in tfp's form
//   -0.5 * (x / scale - loc / scale)^2 - (0.5 * log(2 pi) + log(scale))
__device__ __forceinline__ float gaussian_poisson_logp(float proj, float m, float x, float pnm, float eps)
{
    const float loc = proj * m;
    const float scale = eps + sqrtf(loc / pnm + eps);
    const float z = x / scale - loc / scale;
    return -0.5f * (z * z) - (kHalfLog2Pi + logf(scale));
}

// what a projector kernel needs to write log-probabilities next to its ray-sums (lp == nullptr: no epilogue)
struct LogLikEpilogue {
    const float *mask, *meas, *pnm;   // [S][A], [S][A][PW], one value
    float eps;
    float *lp;                        // [S][A][PW]
};

}  // namespace ctpvae
