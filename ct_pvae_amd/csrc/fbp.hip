// fbp.hip -- filtered back-projection (a6, ctvae/fbp_tensorflow.py:39-74) in float64.
//
// The reference filters with fft -> multiply by filter_1d -> ifft -> real part.  For a real
// sinogram that is a circular convolution with hker = Re(ifft(filter_1d)), which the host computes
// once (P values).  P = 184 here, so the O(P^2) convolution per row is 34k fp64 FMAs-worth of work
// out of LDS and needs no FFT library, no complex intermediate and no second pass over HBM.
#include "common.h"

namespace ctpvae {

__global__ __launch_bounds__(256) void fbp_filter_kernel(const double *__restrict__ sino, int R, int P,
                                                         const double *__restrict__ hker,
                                                         double *__restrict__ out)
{
    extern __shared__ double lds_d[];  // [P] row, [P] kernel
    double *row = lds_d, *hk = lds_d + P;
    const int r = blockIdx.x;
    for (int n = threadIdx.x; n < P; n += blockDim.x) {
        row[n] = sino[(size_t)r * P + n];
        hk[n] = hker[n];
    }
    __syncthreads();
    for (int n = threadIdx.x; n < P; n += blockDim.x) {
        double acc = 0.0;
        for (int m = 0; m < P; ++m) {
            int k = n - m;
            k += (k < 0) ? P : 0;
            acc += row[m] * hk[k];
        }
        out[(size_t)r * P + n] = acc;
    }
}

// tfp.math.interp_regular_1d_grid, fill_value='constant_extension' (tensorflow-probability 0.14.0)
__device__ __forceinline__ double interp_regular_1d(const double *__restrict__ y_ref, int ny, double x,
                                                    double x_min, double x_max)
{
    const double top = (double)(ny - 1);
    // tfp: (x - x_min) / (x_max - x_min) * (ny - 1).  iradon's grid spans exactly ny - 1 (x_max - x_min == top, both
    // exact in fp64), so the scale factor is one fp64 constant (1.0 here) instead of an fp64 division per tap; the
    // result moves by at most 1 ulp of idx, i.e. ~1e-16 of the interpolated value (the interpolant is continuous).
    const double idx_unclipped = (x - x_min) * (top / (x_max - x_min));
    double idx = idx_unclipped;
    idx = idx < 0.0 ? 0.0 : idx;
    idx = idx > top ? top : idx;
    double below = floor(idx);
    const double above = fmin(below + 1.0, top);
    below = fmax(above - 1.0, 0.0);
    const double t = idx - below;
    double y = t * y_ref[(int)above] + (1.0 - t) * y_ref[(int)below];
    if (idx_unclipped < 0.0) y = y_ref[0];
    if (idx_unclipped > top) y = y_ref[ny - 1];
    return y;
}

__global__ __launch_bounds__(256) void fbp_backproject_kernel(const double *__restrict__ filt, int B, int A,
                                                              int P, const double *__restrict__ cos_t,
                                                              const double *__restrict__ sin_t, int X, int Y,
                                                              double x0, double y0, double t0,
                                                              double *__restrict__ recon)
{
    // pixel (i, j) sits at (i - x0, j - y0); detector sample k at k - t0.  The reference's iradon
    // (ctvae/fbp_tensorflow.py:52-70): x0 = X / 2, y0 = Y / 2, t0 = P / 2.
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= X * Y) return;
    const int i = p / Y, j = p - i * Y;
    const double xpr = (double)i - x0, ypr = (double)j - y0;
    const double x_min = 0.0 - t0, x_max = (double)(P - 1) - t0;
    const double *f = filt + (size_t)b * A * P;
    double acc = 0.0;
    for (int a = 0; a < A; ++a) {
        const double t = ypr * cos_t[a] - xpr * sin_t[a];
        acc += interp_regular_1d(f + (size_t)a * P, P, t, x_min, x_max);
    }
    recon[(size_t)b * X * Y + p] = acc * 3.14159265358979323846 / (2.0 * A);
}

}  // namespace ctpvae

using namespace ctpvae;

extern "C" {

int ctpvae_fbp_filter_f64(const double *sino_dev, int R, int P, const double *hker_dev, double *out_dev,
                          ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(sino_dev && hker_dev && out_dev, "fbp_filter: null pointer");
    CTPVAE_REQUIRE(R > 0 && P > 0, "fbp_filter: sizes must be positive (R=%d P=%d)", R, P);
    const size_t shmem = (size_t)2 * P * sizeof(double);
    CTPVAE_REQUIRE(shmem <= 64 * 1024, "fbp_filter: %d detector bins do not fit LDS", P);
    hipLaunchKernelGGL(fbp_filter_kernel, dim3(R), dim3(256), shmem, (hipStream_t)stream, sino_dev, R, P, hker_dev,
                       out_dev);
    CTPVAE_LAUNCH_CHECK("fbp_filter_kernel");
    return CTPVAE_OK;
}

int ctpvae_fbp_backproject_f64(const double *filt_dev, int B, int A, int P, const double *cos_dev,
                               const double *sin_dev, int X, int Y, double *recon_dev, ctpvae_stream_t stream)
{
    return ctpvae_fbp_backproject_geom_f64(filt_dev, B, A, P, cos_dev, sin_dev, X, Y, X / 2.0, Y / 2.0, P / 2.0, recon_dev, stream);
}

int ctpvae_fbp_backproject_geom_f64(const double *filt_dev, int B, int A, int P, const double *cos_dev,
                                    const double *sin_dev, int X, int Y, double x0, double y0, double t0,
                                    double *recon_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(filt_dev && cos_dev && sin_dev && recon_dev, "fbp_backproject: null pointer");
    CTPVAE_REQUIRE(B > 0 && A > 0 && P > 1 && X > 0 && Y > 0,
                   "fbp_backproject: bad sizes (B=%d A=%d P=%d X=%d Y=%d)", B, A, P, X, Y);
    const int chunk = max_slices_per_launch();
    for (int b0 = 0; b0 < B; b0 += chunk) {   // sinograms are indexed with a grid dimension: longer batches go in chunks
        const int n = B - b0 < chunk ? B - b0 : chunk;
        hipLaunchKernelGGL(fbp_backproject_kernel, dim3(ceil_div(X * Y, 256), n), dim3(256), 0, (hipStream_t)stream,
                           filt_dev + (size_t)b0 * A * P, n, A, P, cos_dev, sin_dev, X, Y, x0, y0, t0,
                           recon_dev + (size_t)b0 * X * Y);
        CTPVAE_LAUNCH_CHECK("fbp_backproject_kernel");
    }
    return CTPVAE_OK;
}

}  // extern "C"
