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

// Back-projection by tfp.math.interp_regular_1d_grid, fill_value='constant_extension' (tensorflow-probability 0.14.0):
//   idx = (x - x_min) / (x_max - x_min) * (ny - 1), clipped to [0, ny - 1]; above = min(floor(idx) + 1, ny - 1), below = max(above - 1, 0);
//   y = t * y_ref[above] + (1 - t) * y_ref[below] with t = idx - below; y_ref[0] / y_ref[ny - 1] beyond the grid.
// iradon's grid spans exactly ny - 1 (x_max - x_min == top, both exact in fp64), so the scale factor is one fp64 constant
// (1.0 here) instead of an fp64 division per tap; the result moves by at most 1 ulp of idx, i.e. ~1e-16 of the
// interpolated value (the interpolant is continuous).
// NB sinograms per thread: where a pixel falls on the detector at an angle, and with what weights it interpolates, is
// geometry -- the fp64 index arithmetic (the kernel's cost) is done once for all of them; every sinogram's sum is the
// same expression in the same order as with one sinogram per thread.
template <int NB>
__global__ __launch_bounds__(256) void fbp_backproject_kernel(const double *__restrict__ filt, int B, int A,
                                                              int P, const double *__restrict__ cos_t,
                                                              const double *__restrict__ sin_t, int X, int Y,
                                                              double x0, double y0, double t0,
                                                              double *__restrict__ recon)
{
    // pixel (i, j) sits at (i - x0, j - y0); detector sample k at k - t0.  The reference's iradon
    // (ctvae/fbp_tensorflow.py:52-70): x0 = X / 2, y0 = Y / 2, t0 = P / 2.
    const int b0 = blockIdx.y * NB;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= X * Y) return;
    const int i = p / Y, j = p - i * Y;
    const double xpr = (double)i - x0, ypr = (double)j - y0;
    const double x_min = 0.0 - t0, x_max = (double)(P - 1) - t0, top = (double)(P - 1);
    const double scale = top / (x_max - x_min);
    const size_t bstride = (size_t)A * P;
    const double *f[NB];                                   // a ragged last group re-reads its first sinogram (never stored)
#pragma unroll
    for (int n = 0; n < NB; ++n) f[n] = filt + (size_t)(b0 + (b0 + n < B ? n : 0)) * bstride;
    double acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[n] = 0.0;
    for (int a = 0; a < A; ++a) {
        const double t = ypr * cos_t[a] - xpr * sin_t[a];
        const double idx_unclipped = (t - x_min) * scale;
        double idx = idx_unclipped;
        idx = idx < 0.0 ? 0.0 : idx;
        idx = idx > top ? top : idx;
        double below = floor(idx);
        const double above = fmin(below + 1.0, top);
        below = fmax(above - 1.0, 0.0);
        const double tt = idx - below;
        const int ka = (int)above, kb = (int)below;
        const bool lo = idx_unclipped < 0.0, hi = idx_unclipped > top;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const double *row = f[n] + (size_t)a * P;
            double y = tt * row[ka] + (1.0 - tt) * row[kb];
            if (lo) y = row[0];
            if (hi) y = row[P - 1];
            acc[n] += y;
        }
    }
#pragma unroll
    for (int n = 0; n < NB; ++n)
        if (b0 + n < B) recon[(size_t)(b0 + n) * X * Y + p] = acc[n] * 3.14159265358979323846 / (2.0 * A);
}

// The transpose of fbp_backproject_kernel (the gradient of iradon with respect to the filtered sinogram):
//     gfilt[b][a][k] = pi / (2A) * sum over pixels p = (i, j), ascending, of w_k(t(i, j, a)) * g[b][p],
// w_k = the weight interp_regular_1d gives y_ref[k] (linear interpolation on the clamped index: bins 0 and P - 1 collect
// everything beyond the detector).  One thread per (angle, bin) walks ALL pixels and evaluates its own weight -- 2 of P
// threads have a non-zero one at a pixel -- so every output is one ordered fp64 sum: deterministic, no atomics.  A
// set-up-path kernel (the reference never differentiates iradon): A * P * X * Y weight evaluations, shared by the kSlices
// slices of a thread.
constexpr int kFbpBwdSlices = 8;
__global__ __launch_bounds__(64) void fbp_backproject_bwd_kernel(const double *__restrict__ g, int B, int A, int P,
                                                                const double *__restrict__ cos_t,
                                                                const double *__restrict__ sin_t, int X, int Y, double x0,
                                                                double y0, double t0, double *__restrict__ gfilt)
{
    const int a = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    const int b0 = blockIdx.z * kFbpBwdSlices;
    const int nb = min(kFbpBwdSlices, B - b0);
    const double c = cos_t[a], s = sin_t[a], top = (double)(P - 1);
    const double x_min = 0.0 - t0, x_max = top - t0;
    const double scale = top / (x_max - x_min);
    double acc[kFbpBwdSlices];
#pragma unroll
    for (int n = 0; n < kFbpBwdSlices; ++n) acc[n] = 0.0;
    const double kd = (double)k;
    for (int i = 0; i < X; ++i) {
        const double xs = ((double)i - x0) * s;
        for (int j = 0; j < Y; ++j) {
            const double t = ((double)j - y0) * c - xs;
            double idx = (t - x_min) * scale;              // the forward's expression, bit for bit
            idx = idx < 0.0 ? 0.0 : idx;
            idx = idx > top ? top : idx;
            double below = floor(idx);
            const double above = fmin(below + 1.0, top);
            below = fmax(above - 1.0, 0.0);
            const double tt = idx - below;
            const double w = (kd == above ? tt : 0.0) + (kd == below ? 1.0 - tt : 0.0);
            if (w != 0.0 && k < P) {
                const double *gp = g + ((size_t)b0 * X + i) * Y + j;
#pragma unroll
                for (int n = 0; n < kFbpBwdSlices; ++n)
                    if (n < nb) acc[n] += w * gp[(size_t)n * X * Y];
            }
        }
    }
    if (k < P) {
#pragma unroll
        for (int n = 0; n < kFbpBwdSlices; ++n)
            if (n < nb) gfilt[((size_t)(b0 + n) * A + a) * P + k] = acc[n] * 3.14159265358979323846 / (2.0 * A);
    }
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
        // two sinograms per thread (measured at 50 x 180 x 184 -> 128 x 128: one 264 us per iradon call, two 233 us, five
        // 360 us -- the per-lane fp64 gathers from L2, not the index arithmetic, bound the kernel, and fewer, fatter threads hide them worse)
        const int cells = ceil_div(X * Y, 256);
        const int nb = n >= 2 && (long long)cells * ceil_div(n, 2) >= 512 ? 2 : 1;
        const dim3 grid(cells, ceil_div(n, nb));
        const double *f0 = filt_dev + (size_t)b0 * A * P;
        double *r0 = recon_dev + (size_t)b0 * X * Y;
        if (nb == 2)
            hipLaunchKernelGGL(fbp_backproject_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, f0, n, A, P, cos_dev, sin_dev, X, Y,
                               x0, y0, t0, r0);
        else
            hipLaunchKernelGGL(fbp_backproject_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, f0, n, A, P, cos_dev, sin_dev, X, Y,
                               x0, y0, t0, r0);
        CTPVAE_LAUNCH_CHECK("fbp_backproject_kernel");
    }
    return CTPVAE_OK;
}

int ctpvae_fbp_backproject_bwd_f64(const double *grecon_dev, int B, int A, int P, const double *cos_dev,
                                   const double *sin_dev, int X, int Y, double x0, double y0, double t0,
                                   double *gfilt_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(grecon_dev && cos_dev && sin_dev && gfilt_dev, "fbp_backproject_bwd: null pointer");
    CTPVAE_REQUIRE(B > 0 && A > 0 && P > 1 && X > 0 && Y > 0,
                   "fbp_backproject_bwd: bad sizes (B=%d A=%d P=%d X=%d Y=%d)", B, A, P, X, Y);
    CTPVAE_REQUIRE(A <= 65535 && ceil_div(B, kFbpBwdSlices) <= 65535, "fbp_backproject_bwd: A=%d or B=%d exceeds the grid", A, B);
    hipLaunchKernelGGL(fbp_backproject_bwd_kernel, dim3(ceil_div(P, 64), A, ceil_div(B, kFbpBwdSlices)), dim3(64), 0,
                       (hipStream_t)stream, grecon_dev, B, A, P, cos_dev, sin_dev, X, Y, x0, y0, t0, gfilt_dev);
    CTPVAE_LAUNCH_CHECK("fbp_backproject_bwd_kernel");
    return CTPVAE_OK;
}

}  // extern "C"
