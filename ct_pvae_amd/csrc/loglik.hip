// loglik.hip -- Gaussian-approximated Poisson log-likelihood of a sinogram (a8,
// ctvae/helper_functions.py:360-368) and its backward.  Elementwise over [B][A][P]; HBM-bound.
#include "common.h"
#include "loglik_math.h"

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
    const float pnm = *pnm_p;
    float gp_local = 0.0f;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        float dpnm;
        const float g = gout[k];
        gproj[k] = g * gaussian_poisson_dlogp(proj[k], mask[k / P], x[k], pnm, eps, dpnm);
        gp_local += g * dpnm;
    }
    if (gpnm) {
        for (int off = 32; off > 0; off >>= 1) gp_local += __shfl_down(gp_local, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gp_local;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(gpnm, red[0] + red[1] + red[2] + red[3]);
    }
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

}  // extern "C"
