// rotate.hip -- rotate-and-sum Radon forward (a2/a5) and its backward passes (a4, K2x) for gfx950.
//
// Index arithmetic follows TensorFlow's ImageProjectiveTransformV3 exactly:
//   x_in = (t0*x + t1*y) + t2 ; y_in = (t3*x + t4*y) + t5     (separate fp32 multiplies and adds;
//   this file is compiled with -ffp-contract=off), NEAREST = round half away from zero, BILINEAR =
//   floor + four zero-filled taps.  The zero-padded canvas of pad_phantom is never materialised: a
//   tap is live only if it lands in the H x W core that sits at (py, px) of the PH x PW canvas.
//
// Work decomposition (MI355X): one workgroup = one slice x one group of angles.  The slice is staged
// once into LDS (row pitch W+1 so that a wave walking a column at theta ~ 90 deg does not hit one
// bank) and every lane owns one ray (angle a, detector bin j), walking the canvas rows i = 0..PH-1 in
// order -- the same summation order as reduce_sum(axis=1) on the reference, so results are
// reproducible bit for bit against the CPU restatement.
#include "common.h"

namespace ctpvae {

struct RotGeom {
    int S, H, W, PH, PW, py, px, A;
};

__device__ __forceinline__ float round_half_away(float v) { return __builtin_roundf(v); }

// ---- forward ---------------------------------------------------------------------------------
template <bool USE_LDS>
__device__ __forceinline__ float core_read(const float *__restrict__ im, const float *lds, int H, int W,
                                           int pitch, int r, int c)
{
    if ((unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W)
        return USE_LDS ? lds[r * pitch + c] : im[(size_t)r * W + c];
    return 0.0f;
}

template <int INTERP, bool USE_LDS>
__global__ __launch_bounds__(256) void rotate_fwd_kernel(const float *__restrict__ img, RotGeom g,
                                                         const float *__restrict__ T8, int a_per_blk,
                                                         float *__restrict__ sino)
{
    extern __shared__ float lds[];
    const int s = blockIdx.y;
    const int a0 = blockIdx.x * a_per_blk;
    const int na = min(a_per_blk, g.A - a0);
    const float *im = img + (size_t)s * g.H * g.W;
    const int pitch = g.W + 1;

    if (USE_LDS) {
        for (int p = threadIdx.x; p < g.H * g.W; p += blockDim.x) {
            const int r = p / g.W, c = p - r * g.W;
            lds[r * pitch + c] = im[p];
        }
        __syncthreads();
    }

    for (int ray = threadIdx.x; ray < na * g.PW; ray += blockDim.x) {
        const int al = ray / g.PW;
        const int j = ray - al * g.PW;
        const int a = a0 + al;
        const float *t = T8 + 8 * a;
        const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
        const float xj = t0 * (float)j, yj = t3 * (float)j;
        float acc = 0.0f;
        for (int i = 0; i < g.PH; ++i) {
            const float fi = (float)i;
            const float x = (xj + t1 * fi) + t2;
            const float y = (yj + t4 * fi) + t5;
            float v;
            if (INTERP == CTPVAE_NEAREST) {
                const int ix = (int)round_half_away(x) - g.px;
                const int iy = (int)round_half_away(y) - g.py;
                v = core_read<USE_LDS>(im, lds, g.H, g.W, pitch, iy, ix);
            } else {
                const float yf = floorf(y), xf = floorf(x);
                const float yc = yf + 1.0f, xc = xf + 1.0f;
                const int ix0 = (int)xf - g.px, iy0 = (int)yf - g.py;
                const int ix1 = (int)xc - g.px, iy1 = (int)yc - g.py;
                const float v_yf = (xc - x) * core_read<USE_LDS>(im, lds, g.H, g.W, pitch, iy0, ix0) +
                                   (x - xf) * core_read<USE_LDS>(im, lds, g.H, g.W, pitch, iy0, ix1);
                const float v_yc = (xc - x) * core_read<USE_LDS>(im, lds, g.H, g.W, pitch, iy1, ix0) +
                                   (x - xf) * core_read<USE_LDS>(im, lds, g.H, g.W, pitch, iy1, ix1);
                v = (yc - y) * v_yf + (y - yf) * v_yc;
            }
            acc += v;
        }
        sino[((size_t)s * g.A + a) * g.PW + j] = acc;
    }
}

// ---- backward, TensorFlow-compatible (gather) -------------------------------------------------
// G_a[y][x] = sample(row-broadcast image of g[a][:], Tinv_a(x, y)); gimg = crop(sum_a G_a).
__device__ __forceinline__ float bcast_read(const float *grow, int PH, int PW, int iy, int ix)
{
    return ((unsigned)iy < (unsigned)PH && (unsigned)ix < (unsigned)PW) ? grow[ix] : 0.0f;
}

template <int INTERP>
__global__ __launch_bounds__(256) void rotate_bwd_tfcompat_kernel(const float *__restrict__ gsino, RotGeom g,
                                                                  const float *__restrict__ Tinv8,
                                                                  int chunk_a, int px_per_thread,
                                                                  float *__restrict__ gimg)
{
    extern __shared__ float lds[];  // [chunk_a][PW] cotangent rows, then [chunk_a][8] transforms
    float *lds_t = lds + (size_t)chunk_a * g.PW;
    const int s = blockIdx.y;
    const int npix = g.H * g.W;
    const int base = blockIdx.x * blockDim.x * px_per_thread;
    constexpr int kMaxPpt = 8;
    float acc[kMaxPpt];
#pragma unroll
    for (int k = 0; k < kMaxPpt; ++k) acc[k] = 0.0f;

    for (int ac = 0; ac < g.A; ac += chunk_a) {
        const int na = min(chunk_a, g.A - ac);
        __syncthreads();
        const float *src = gsino + ((size_t)s * g.A + ac) * g.PW;
        for (int p = threadIdx.x; p < na * g.PW; p += blockDim.x) lds[p] = src[p];
        for (int p = threadIdx.x; p < na * 8; p += blockDim.x) lds_t[p] = Tinv8[(size_t)ac * 8 + p];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kMaxPpt; ++k) {
            if (k >= px_per_thread) break;
            const int p = base + k * blockDim.x + threadIdx.x;
            if (p >= npix) continue;
            const int r = p / g.W, c = p - r * g.W;
            const float fx = (float)(c + g.px), fy = (float)(r + g.py);
            float sum = acc[k];
            for (int al = 0; al < na; ++al) {
                const float *t = lds_t + 8 * al;
                const float *grow = lds + (size_t)al * g.PW;
                const float x = (t[0] * fx + t[1] * fy) + t[2];
                const float y = (t[3] * fx + t[4] * fy) + t[5];
                float v;
                if (INTERP == CTPVAE_NEAREST) {
                    v = bcast_read(grow, g.PH, g.PW, (int)round_half_away(y), (int)round_half_away(x));
                } else {
                    const float yf = floorf(y), xf = floorf(x);
                    const float yc = yf + 1.0f, xc = xf + 1.0f;
                    const float v_yf = (xc - x) * bcast_read(grow, g.PH, g.PW, (int)yf, (int)xf) +
                                       (x - xf) * bcast_read(grow, g.PH, g.PW, (int)yf, (int)xc);
                    const float v_yc = (xc - x) * bcast_read(grow, g.PH, g.PW, (int)yc, (int)xf) +
                                       (x - xf) * bcast_read(grow, g.PH, g.PW, (int)yc, (int)xc);
                    v = (yc - y) * v_yf + (y - yf) * v_yc;
                }
                sum += v;
            }
            acc[k] = sum;
        }
    }
#pragma unroll
    for (int k = 0; k < kMaxPpt; ++k) {
        if (k >= px_per_thread) break;
        const int p = base + k * blockDim.x + threadIdx.x;
        if (p < npix) gimg[(size_t)s * npix + p] = acc[k];
    }
}

// ---- backward, exact transpose (scatter) -------------------------------------------------------
// Mirrors the forward's decomposition; every ray adds its cotangent into an LDS copy of the slice
// (ds_add_f32), and the workgroup then adds its tile into gimg (global_atomic_add_f32; gimg is zeroed
// first).  Summation order is not fixed, so results agree with the CPU restatement to rounding only.
template <bool USE_LDS>
__device__ __forceinline__ void core_add(float *gim, float *lds, int H, int W, int pitch, int r, int c, float v)
{
    if ((unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W) {
        if (USE_LDS)
            atomicAdd(&lds[r * pitch + c], v);
        else
            atomicAdd(&gim[(size_t)r * W + c], v);
    }
}

template <int INTERP, bool USE_LDS>
__global__ __launch_bounds__(256) void rotate_bwd_exact_kernel(const float *__restrict__ gsino, RotGeom g,
                                                               const float *__restrict__ T8, int a_per_blk,
                                                               float *__restrict__ gimg)
{
    extern __shared__ float lds[];
    const int s = blockIdx.y;
    const int a0 = blockIdx.x * a_per_blk;
    const int na = min(a_per_blk, g.A - a0);
    float *gim = gimg + (size_t)s * g.H * g.W;
    const int pitch = g.W + 1;

    if (USE_LDS) {
        for (int p = threadIdx.x; p < g.H * pitch; p += blockDim.x) lds[p] = 0.0f;
        __syncthreads();
    }
    for (int ray = threadIdx.x; ray < na * g.PW; ray += blockDim.x) {
        const int al = ray / g.PW;
        const int j = ray - al * g.PW;
        const int a = a0 + al;
        const float *t = T8 + 8 * a;
        const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
        const float xj = t0 * (float)j, yj = t3 * (float)j;
        const float gv = gsino[((size_t)s * g.A + a) * g.PW + j];
        for (int i = 0; i < g.PH; ++i) {
            const float fi = (float)i;
            const float x = (xj + t1 * fi) + t2;
            const float y = (yj + t4 * fi) + t5;
            if (INTERP == CTPVAE_NEAREST) {
                core_add<USE_LDS>(gim, lds, g.H, g.W, pitch, (int)round_half_away(y) - g.py,
                                  (int)round_half_away(x) - g.px, gv);
            } else {
                const float yf = floorf(y), xf = floorf(x);
                const float yc = yf + 1.0f, xc = xf + 1.0f;
                const int ix0 = (int)xf - g.px, iy0 = (int)yf - g.py;
                const int ix1 = (int)xc - g.px, iy1 = (int)yc - g.py;
                core_add<USE_LDS>(gim, lds, g.H, g.W, pitch, iy0, ix0, (yc - y) * ((xc - x) * gv));
                core_add<USE_LDS>(gim, lds, g.H, g.W, pitch, iy0, ix1, (yc - y) * ((x - xf) * gv));
                core_add<USE_LDS>(gim, lds, g.H, g.W, pitch, iy1, ix0, (y - yf) * ((xc - x) * gv));
                core_add<USE_LDS>(gim, lds, g.H, g.W, pitch, iy1, ix1, (y - yf) * ((x - xf) * gv));
            }
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (int p = threadIdx.x; p < g.H * g.W; p += blockDim.x) {
            const int r = p / g.W, c = p - r * g.W;
            const float v = lds[r * pitch + c];
            if (v != 0.0f) atomicAdd(&gim[p], v);
        }
    }
}

static int check_geom(const char *who, int S, int H, int W, int PH, int PW, int py, int px, int A, int interp)
{
    CTPVAE_REQUIRE(S > 0 && H > 0 && W > 0 && A > 0, "%s: sizes must be positive (S=%d H=%d W=%d A=%d)", who,
                   S, H, W, A);
    CTPVAE_REQUIRE(py >= 0 && px >= 0 && PH >= H + py && PW >= W + px,
                   "%s: the %dx%d slice at (%d,%d) does not fit the %dx%d canvas", who, H, W, py, px, PH, PW);
    CTPVAE_REQUIRE(interp == CTPVAE_NEAREST || interp == CTPVAE_BILINEAR, "%s: unknown interpolation %d", who,
                   interp);
    CTPVAE_REQUIRE(S <= 65535, "%s: at most 65535 slices per call (got %d)", who, S);
    CTPVAE_REQUIRE((long long)PH * PW < (1ll << 24), "%s: canvas too large for fp32 index arithmetic", who);
    return CTPVAE_OK;
}

// angles per workgroup: enough workgroups to cover the chip, and a whole number of 256-thread
// passes over the group's rays where possible
static int pick_angles_per_block(int S, int A, int PW)
{
    const int target_wgs = 256 * 2;
    int apb = A;
    while (apb > 1 && (long long)S * ceil_div(A, apb) < target_wgs) apb = (apb + 1) / 2;
    (void)PW;
    return apb;
}

}  // namespace ctpvae

using namespace ctpvae;

extern "C" {

int ctpvae_rotate_fwd_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                          const float *T8_dev, int A, int interp, float *sino_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(img_dev && T8_dev && sino_dev, "rotate_fwd: null pointer");
    if (int rc = check_geom("rotate_fwd", S, H, W, PH, PW, py, px, A, interp)) return rc;
    const RotGeom g{S, H, W, PH, PW, py, px, A};
    const size_t lds_bytes = (size_t)H * (W + 1) * sizeof(float);
    const bool use_lds = lds_bytes <= (size_t)kMaxLdsBytes;
    const int apb = pick_angles_per_block(S, A, PW);
    const dim3 grid(ceil_div(A, apb), S), block(256);
    auto launch = [&](auto kernel, size_t shmem) -> int {
        if (shmem > 64 * 1024)
            CTPVAE_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)shmem));
        hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)stream, img_dev, g, T8_dev, apb, sino_dev);
        CTPVAE_LAUNCH_CHECK("rotate_fwd_kernel");
        return CTPVAE_OK;
    };
    if (interp == CTPVAE_NEAREST)
        return use_lds ? launch(rotate_fwd_kernel<CTPVAE_NEAREST, true>, lds_bytes)
                       : launch(rotate_fwd_kernel<CTPVAE_NEAREST, false>, 0);
    return use_lds ? launch(rotate_fwd_kernel<CTPVAE_BILINEAR, true>, lds_bytes)
                   : launch(rotate_fwd_kernel<CTPVAE_BILINEAR, false>, 0);
}

int ctpvae_rotate_bwd_f32(const float *gsino_dev, int S, int A, int PH, int PW, const float *T8_dev, int interp,
                          int mode, int H, int W, int py, int px, float *gimg_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(gsino_dev && T8_dev && gimg_dev, "rotate_bwd: null pointer");
    if (int rc = check_geom("rotate_bwd", S, H, W, PH, PW, py, px, A, interp)) return rc;
    CTPVAE_REQUIRE(mode == CTPVAE_BWD_TF_COMPAT || mode == CTPVAE_BWD_EXACT, "rotate_bwd: unknown mode %d", mode);
    const RotGeom g{S, H, W, PH, PW, py, px, A};
    if (mode == CTPVAE_BWD_TF_COMPAT) {
        // cotangent rows staged in LDS in chunks of angles (<= 32 KiB per chunk)
        int chunk_a = (32 * 1024) / ((PW + 8) * (int)sizeof(float));
        if (chunk_a < 1) chunk_a = 1;
        if (chunk_a > A) chunk_a = A;
        const size_t shmem = (size_t)chunk_a * (PW + 8) * sizeof(float);
        CTPVAE_REQUIRE(shmem <= (size_t)kMaxLdsBytes, "rotate_bwd: detector of %d bins does not fit LDS", PW);
        const int npix = H * W;
        int ppt = 8;
        while (ppt > 1 && (long long)S * ceil_div(npix, 256 * ppt) < 512) ppt /= 2;
        const dim3 grid(ceil_div(npix, 256 * ppt), S), block(256);
        auto launch = [&](auto kernel) -> int {
            if (shmem > 64 * 1024)
                CTPVAE_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)shmem));
            hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)stream, gsino_dev, g, T8_dev, chunk_a, ppt,
                               gimg_dev);
            CTPVAE_LAUNCH_CHECK("rotate_bwd_tfcompat_kernel");
            return CTPVAE_OK;
        };
        return interp == CTPVAE_NEAREST ? launch(rotate_bwd_tfcompat_kernel<CTPVAE_NEAREST>)
                                        : launch(rotate_bwd_tfcompat_kernel<CTPVAE_BILINEAR>);
    }
    // exact transpose
    CTPVAE_HIP(hipMemsetAsync(gimg_dev, 0, (size_t)S * H * W * sizeof(float), (hipStream_t)stream));
    const size_t lds_bytes = (size_t)H * (W + 1) * sizeof(float);
    const bool use_lds = lds_bytes <= (size_t)kMaxLdsBytes;
    const int apb = pick_angles_per_block(S, A, PW);
    const dim3 grid(ceil_div(A, apb), S), block(256);
    auto launch = [&](auto kernel, size_t shmem) -> int {
        if (shmem > 64 * 1024)
            CTPVAE_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)shmem));
        hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)stream, gsino_dev, g, T8_dev, apb, gimg_dev);
        CTPVAE_LAUNCH_CHECK("rotate_bwd_exact_kernel");
        return CTPVAE_OK;
    };
    if (interp == CTPVAE_NEAREST)
        return use_lds ? launch(rotate_bwd_exact_kernel<CTPVAE_NEAREST, true>, lds_bytes)
                       : launch(rotate_bwd_exact_kernel<CTPVAE_NEAREST, false>, 0);
    return use_lds ? launch(rotate_bwd_exact_kernel<CTPVAE_BILINEAR, true>, lds_bytes)
                   : launch(rotate_bwd_exact_kernel<CTPVAE_BILINEAR, false>, 0);
}

}  // extern "C"
