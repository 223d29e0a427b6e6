// loglik.hip -- Gaussian-approximated Poisson log-likelihood of a sinogram (a8,
// ctvae/helper_functions.py:360-368) and its backward.  Elementwise over [B][A][P]; HBM-bound.
#include <algorithm>

#include "common.h"
#include "loglik_math.h"
#include "rotate_plan.h"

namespace ctpvae {

__global__ __launch_bounds__(256) void loglik_fwd_kernel(const float *__restrict__ proj,
                                                         const float *__restrict__ mask,
                                                         const float *__restrict__ x, long long n, int P,
                                                         const float *__restrict__ pnm_p, float eps,
                                                         float *__restrict__ out)
{
    const float pnm = *pnm_p;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        out[k] = gaussian_poisson_logp(proj[k], mask[k / P], x[k], pnm, eps);
    }
}

__global__ __launch_bounds__(256) void loglik_bwd_kernel(const float *__restrict__ proj,
                                                         const float *__restrict__ mask,
                                                         const float *__restrict__ x,
                                                         const float *__restrict__ gout, long long n, int P,
                                                         const float *__restrict__ pnm_p, float eps,
                                                         float *__restrict__ gproj, float *__restrict__ gpnm)
{
    __shared__ float red[4];
    const float pnm = *pnm_p, inv_pnm = 1.0f / pnm;
    float gp_local = 0.0f;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        float dpnm;
        const float g = gout[k];
        gproj[k] = g * gaussian_poisson_dlogp(gaussian_poisson_terms(proj[k], mask[k / P], pnm, eps), mask[k / P], x[k], inv_pnm, dpnm);
        gp_local += g * dpnm;
    }
    if (gpnm) {
        for (int off = 32; off > 0; off >>= 1) gp_local += __shfl_down(gp_local, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gp_local;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(gpnm, red[0] + red[1] + red[2] + red[3]);
    }
}

// Per-object sums of a stored log-probability array in the fixed order of the fused epilogues (LogLikEpilogue::part):
// a slice's [A][PW] values are cut into 64-lane tasks -- partition 0: the planned kernels' (angle, bin block) tasks, two
// 32-bin bands mirrored about the detector centre (lane_to_bin); partition 1: the tiled reduce pass's contiguous 64-bin
// blocks -- each task is added by the xor butterfly (lanes without a bin add +0.0f), an angle's task sums are added in
// ascending order, and the angle sums by object_sum_of_parts' rule (loglik_math.h).  One workgroup per slice; the task sums
// wait in LDS.
__global__ __launch_bounds__(256) void loglik_object_sums_kernel(const float *__restrict__ lp, int A, int PW, int partition,
                                                                 int tasks_per_row, float *__restrict__ out)
{
    extern __shared__ float part[];   // the task sums of 64 angles: object_sum_of_parts' rule adds the angles in groups of 64
    const int s = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const float *row0 = lp + (size_t)s * A * PW;
    float total = 0.0f;
    for (int a0 = 0; a0 < A; a0 += 64) {
        const int na = min(64, A - a0), NT = na * tasks_per_row;
        for (int t = wave; t < NT; t += nwaves) {
            const int a = t / tasks_per_row, jb = t - a * tasks_per_row;
            const int j = partition == 0 ? lane_to_bin(PW, jb, lane) : 64 * jb + lane;
            const float v = (unsigned)j < (unsigned)PW ? row0[(size_t)(a0 + a) * PW + j] : 0.0f;
            const float tot = wave_sum(v);
            if (lane == 0) part[t] = tot;
        }
        __syncthreads();
        if (wave == 0) total += object_sum_of_parts(part, na, tasks_per_row, lane);   // (one group: 0 + B_g)
        __syncthreads();
    }
    if (threadIdx.x == 0) out[s] = total;
}

}  // namespace ctpvae

using namespace ctpvae;

extern "C" {

int ctpvae_loglik_fwd_f32(const float *proj_dev, const float *mask_dev, const float *x_dev, int B, int A, int P,
                          const float *pnm_dev, float eps, float *out_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(proj_dev && mask_dev && x_dev && pnm_dev && out_dev, "loglik_fwd: null pointer");
    CTPVAE_REQUIRE(B > 0 && A > 0 && P > 0, "loglik_fwd: sizes must be positive (B=%d A=%d P=%d)", B, A, P);
    const long long n = (long long)B * A * P;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(loglik_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, proj_dev, mask_dev, x_dev, n,
                       P, pnm_dev, eps, out_dev);
    CTPVAE_LAUNCH_CHECK("loglik_fwd_kernel");
    return CTPVAE_OK;
}

int ctpvae_loglik_bwd_f32(const float *proj_dev, const float *mask_dev, const float *x_dev, const float *gout_dev,
                          int B, int A, int P, const float *pnm_dev, float eps, float *gproj_dev, float *gpnm_dev,
                          ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(proj_dev && mask_dev && x_dev && gout_dev && pnm_dev && gproj_dev, "loglik_bwd: null pointer");
    CTPVAE_REQUIRE(B > 0 && A > 0 && P > 0, "loglik_bwd: sizes must be positive (B=%d A=%d P=%d)", B, A, P);
    const long long n = (long long)B * A * P;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (gpnm_dev) CTPVAE_HIP(hipMemsetAsync(gpnm_dev, 0, sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(loglik_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, proj_dev, mask_dev, x_dev,
                       gout_dev, n, P, pnm_dev, eps, gproj_dev, gpnm_dev);
    CTPVAE_LAUNCH_CHECK("loglik_bwd_kernel");
    return CTPVAE_OK;
}

int ctpvae_loglik_tasks_per_row(int PW, int partition)
{
    if (PW <= 0 || (partition != 0 && partition != 1)) return fail(CTPVAE_EINVAL, "loglik_tasks_per_row: bad arguments");
    return partition == 0 ? num_bin_blocks(PW) : ceil_div(PW, 64);
}

long long ctpvae_loglik_part_floats(int S, int n_angles, int PW, int partition)
{
    if (S <= 0 || n_angles <= 0 || PW <= 0 || (partition != 0 && partition != 1))
        return fail(CTPVAE_EINVAL, "loglik_part_floats: bad arguments");
    const int tpr = partition == 0 ? num_bin_blocks(PW) : ceil_div(PW, 64);
    return (long long)S * n_angles * tpr + S;   // the partial sums, then one arrival counter per slice
}

int ctpvae_loglik_object_sums_f32(const float *lp_dev, int S, int A, int PW, int partition, float *out_dev,
                                  ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(lp_dev && out_dev, "loglik_object_sums: null pointer");
    CTPVAE_REQUIRE(S > 0 && A > 0 && PW > 0 && (partition == 0 || partition == 1),
                   "loglik_object_sums: bad arguments (S=%d A=%d PW=%d partition=%d)", S, A, PW, partition);
    const int tpr = partition == 0 ? num_bin_blocks(PW) : ceil_div(PW, 64);
    const long long NT = 64ll * tpr;                       // task sums of one group of 64 angles (any number of angles)
    CTPVAE_REQUIRE(NT * 4 <= 64 * 1024, "loglik_object_sums: rows of at most 16384 bins (got %d)", PW);
    for (int s0 = 0; s0 < S; s0 += 65535) {
        const int n = std::min(65535, S - s0);
        hipLaunchKernelGGL(loglik_object_sums_kernel, dim3(n), dim3(256), (size_t)NT * 4, (hipStream_t)stream,
                           lp_dev + (size_t)s0 * A * PW, A, PW, partition, tpr, out_dev + s0);
        CTPVAE_LAUNCH_CHECK("loglik_object_sums_kernel");
    }
    return CTPVAE_OK;
}

}  // extern "C"
