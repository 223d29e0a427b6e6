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

// d logp / d proj (the mask factor included) and d logp / d pnm of the same sample: what the backward multiplies the
// upstream gradient by.  One expression for loglik_bwd_kernel and for the projector epilogue that stores dlp.
__device__ __forceinline__ float gaussian_poisson_dlogp(float proj, float m, float x, float pnm, float eps, float &dpnm)
{
    const float loc = proj * m;
    const float root = sqrtf(loc / pnm + eps);
    const float scale = eps + root;
    const float z = (x - loc) / scale;
    const float dscale = (z * z - 1.0f) / scale;       // d logp / d scale
    const float dscale_du = 0.5f / root;                // d scale / d (loc/pnm + eps)
    dpnm = dscale * dscale_du * (-loc / (pnm * pnm));
    return (z / scale + dscale * dscale_du / pnm) * m;
}

// what a projector kernel needs to write log-probabilities next to its ray-sums (lp == nullptr: no epilogue)
struct LogLikEpilogue {
    const float *mask, *meas, *pnm;   // [S][A], [S][A][PW], one value
    float eps;
    float *lp;                        // [S][A][PW]
    float *dlp;                       // [S][A][PW] d lp / d ray-sum for the backward, or nullptr
    // angle-subset launches only: mask / meas are the DENSE [S][A_plan] / [S][A_plan][PW] arrays, indexed by the plan
    // angle (the caller's gather mask[:, angles_i], proj_sample[:, angles_i] -- ctvae/helper_functions.py:356-357 --
    // folded into the load); 0: they are compact like the outputs
    int dense = 0;
    // per-object sums ("reduction to per-object log-lik", ctvae/helper_functions.py:305-312): when `part` is set
    // the kernel writes ONE partial sum per 64-lane task into part[(s * A_out + k) * tasks_per_row + task] (lanes added by
    // the xor butterfly 32, 16, 8, 4, 2, 1; lanes without a bin add +0.0f) and lp / the ray-sum store become optional; a
    // second, tiny launch adds a slice's partials in the fixed order of object_sum_of_parts below.
    float *part = nullptr;

    // o: offset of the ray-sum in the outputs; om / sa: offsets of its measured sample and its mask entry
    __device__ __forceinline__ void write(size_t o, size_t om, size_t sa, float raysum) const
    {
        write_loaded(o, mask[sa], meas[om], *pnm, raysum);
    }
    __device__ __forceinline__ void write_loaded(size_t o, float m, float x, float pnm_v, float raysum) const
    {
        lp[o] = gaussian_poisson_logp(raysum, m, x, pnm_v, eps);
        if (dlp) {
            float unused;
            dlp[o] = gaussian_poisson_dlogp(raysum, m, x, pnm_v, eps, unused);
        }
    }
    // the same, returning the log-probability; lp (and dlp) are stored only where a buffer was given
    __device__ __forceinline__ float eval(size_t o, size_t om, size_t sa, float raysum) const
    {
        return eval_loaded(o, mask[sa], meas[om], *pnm, raysum);
    }
    // ... with the operands already in registers: a kernel requests them BEFORE its long phase (the walk, the sum over tiles)
    // so that their round trip to memory is not paid after it -- the stores below keep the compiler from moving the loads up
    // by itself (nothing tells it that the buffers are distinct)
    __device__ __forceinline__ float eval_loaded(size_t o, float m, float x, float pnm_v, float raysum) const
    {
        const float v = gaussian_poisson_logp(raysum, m, x, pnm_v, eps);
        if (lp) lp[o] = v;
        if (dlp) {
            float unused;
            dlp[o] = gaussian_poisson_dlogp(raysum, m, x, pnm_v, eps, unused);
        }
        return v;
    }
};

// The fixed order of a 64-lane task's partial sum: xor butterfly, every lane ends with the same value (a + b == b + a bit
// for bit, so both partners of a step compute the same sum).  oracle/radon_oracle.py loglik_object_sums restates it.
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// The fixed order of the per-object sum over task sums part[a][k] (A angles x tpr tasks per angle):
//   S_a   = ((0 + part[a][0]) + part[a][1]) + ...            one angle's tasks, ascending
//   total = ((0 + B_0) + B_1) + ...,  B_g = wave_sum over lanes l of S_(64 g + l)  (angles past A add +0.0f)
// -- every level is either a short sequential sum or the xor butterfly, so a wave computes it in a few hundred cycles whatever
// A is (a plain ascending sum over all A * tpr task sums took 40 us at 90 angles x 12 tasks: 1080 dependent adds).
__device__ __forceinline__ float object_sum_of_parts(const float *__restrict__ part, int A, int tpr, int lane)
{
    float total = 0.0f;
    for (int a0 = 0; a0 < A; a0 += 64) {
        const int a = a0 + lane;
        float sa = 0.0f;
        if (a < A)
            for (int k = 0; k < tpr; k += 4) {   // four loads in flight, added in order (one at a time: tpr round trips)
                const float *p = part + (size_t)a * tpr + k;
                const int n = tpr - k;
                const float v0 = p[0], v1 = p[n > 1 ? 1 : 0], v2 = p[n > 2 ? 2 : 0], v3 = p[n > 3 ? 3 : 0];
                sa += v0;
                if (n > 1) sa += v1;
                if (n > 2) sa += v2;
                if (n > 3) sa += v3;
            }
        total += wave_sum(sa);
    }
    return total;
}
// lp_sum[s] of the partial sums a fused epilogue wrote (LogLikEpilogue::part): one wave per slice
[[maybe_unused]] static __global__ __launch_bounds__(64) void loglik_sum_partials_kernel(const float *__restrict__ part, int S, int A,
                                                                                       int tpr, float *__restrict__ out)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const float total = object_sum_of_parts(part + (size_t)s * A * tpr, A, tpr, lane);
    if (lane == 0) out[s] = total;
}

}  // namespace ctpvae
