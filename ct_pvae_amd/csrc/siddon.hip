// siddon.hip -- TomoPy-style ray-driven projector (a7): exact ray / pixel-grid intersection lengths.
//
// Follows libtomo's project(): for every (angle p, detector bin d) the ray's crossings with the
// horizontal grid lines (list "a") and with the vertical grid lines (list "b") are merged by x,
// consecutive crossings give a segment length and, from the segment midpoint, a pixel.  libtomo
// materialises both lists and the merged list; here one lane owns one ray and performs the merge on
// the fly with two cursors, evaluating exactly the same fp32 expressions, so no per-ray arrays exist
// and the slice is read from LDS.  Compiled with -ffp-contract=off; '/' and sqrtf are correctly
// rounded (hipcc default), so the result equals the CPU restatement bit for bit.
#include <cmath>

#include "common.h"

namespace ctpvae {

struct SidGeom {
    int oy, ox, oz, dt, dx;
    float mov;
};

template <bool USE_LDS>
__global__ __launch_bounds__(256) void siddon_fwd_kernel(const float *__restrict__ obj, SidGeom g,
                                                         const float *__restrict__ sin_t,
                                                         const float *__restrict__ cos_t,
                                                         const int *__restrict__ quad_t, int p_per_blk,
                                                         float *__restrict__ data)
{
    extern __shared__ float lds[];
    const int s = blockIdx.y;
    const int p0 = blockIdx.x * p_per_blk;
    const int np = min(p_per_blk, g.dt - p0);
    const float *model_g = obj + (size_t)s * g.ox * g.oz;
    const int pitch = g.oz + 1;
    if (USE_LDS) {
        for (int q = threadIdx.x; q < g.ox * g.oz; q += blockDim.x) {
            const int r = q / g.oz, c = q - r * g.oz;
            lds[r * pitch + c] = model_g[q];
        }
        __syncthreads();
    }
    const int ox = g.ox, oz = g.oz;
    const float gx0 = -ox * 0.5f, gy0 = -oz * 0.5f;  // gridx[n] = gx0 + n, gridy[n] = gy0 + n
    const float gx_gt = gx0 + 0.01f, gx_le = (gx0 + ox) - 0.01f;
    const float gy_gt = gy0 + 0.01f, gy_le = (gy0 + oz) - 0.01f;
    const float hx = ox * 0.5f, hz = oz * 0.5f;

    for (int ray = threadIdx.x; ray < np * g.dx; ray += blockDim.x) {
        const int pl = ray / g.dx;
        const int d = ray - pl * g.dx;
        const int p = p0 + pl;
        const float sin_p = sin_t[p], cos_p = cos_t[p];
        const int quadrant = quad_t[p];
        const float xi = (float)(-ox - oz);
        const float yi = (1 - g.dx) / 2.0f + d + g.mov;
        const float srcx = xi * cos_p - yi * sin_p, srcy = xi * sin_p + yi * cos_p;
        const float detx = -xi * cos_p - yi * sin_p, dety = -xi * sin_p + yi * cos_p;
        const float slope = (srcy - dety) / (srcx - detx);
        const float islope = (srcx - detx) / (srcy - dety);

        // list a: crossings with y = gridy[n], x = coordx(n); the kept n form one contiguous run
        int a_lo = 0, a_cnt = 0;
        for (int n = 0; n <= oz; ++n) {
            const float cx = islope * ((gy0 + n) - srcy) + srcx;
            if (cx >= gx_gt && cx <= gx_le) {
                if (a_cnt == 0) a_lo = n;
                ++a_cnt;
            }
        }
        // list b: crossings with x = gridx[n], y = coordy(n)
        int b_lo = 0, b_cnt = 0;
        for (int n = 0; n <= ox; ++n) {
            const float cy = slope * ((gx0 + n) - srcx) + srcy;
            if (cy >= gy_gt && cy <= gy_le) {
                if (b_cnt == 0) b_lo = n;
                ++b_cnt;
            }
        }
        const int csize = a_cnt + b_cnt;
        float acc = 0.0f;
        int i = 0, j = 0;
        float px_prev = 0.0f, py_prev = 0.0f;
        for (int k = 0; k < csize; ++k) {
            // head of list a (ascending n in quadrant 1, descending otherwise) and of list b
            const int an = a_lo + (quadrant ? i : (a_cnt - 1 - i));
            const float a_y = gy0 + an;
            const float a_x = islope * (a_y - srcy) + srcx;
            const int bn = b_lo + j;
            const float b_x = gx0 + bn;
            const float b_y = slope * (b_x - srcx) + srcy;
            bool take_a;
            if (i < a_cnt && j < b_cnt)
                take_a = a_x < b_x;
            else
                take_a = i < a_cnt;
            const float cx = take_a ? a_x : b_x;
            const float cy = take_a ? a_y : b_y;
            i += take_a ? 1 : 0;
            j += take_a ? 0 : 1;
            if (k > 0) {
                const float diffx = cx - px_prev, diffy = cy - py_prev;
                const float dist = sqrtf(diffx * diffx + diffy * diffy);
                const float midx = (cx + px_prev) * 0.5f, midy = (cy + py_prev) * 0.5f;
                const float x1 = midx + hx, x2 = midy + hz;
                const int i1 = (int)x1, i2 = (int)x2;
                const int indx = i1 - (i1 > x1), indy = i2 - (i2 > x2);
                // libtomo reads model[indy + indx*oz] unchecked; midpoints lie strictly inside the grid
                const int ix = min(max(indx, 0), ox - 1), iy = min(max(indy, 0), oz - 1);
                const float m = USE_LDS ? lds[ix * pitch + iy] : model_g[(size_t)ix * oz + iy];
                acc += m * dist;
            }
            px_prev = cx;
            py_prev = cy;
        }
        data[((size_t)s * g.dt + p) * g.dx + d] = acc;
    }
}

}  // namespace ctpvae

using namespace ctpvae;

extern "C" {

int ctpvae_siddon_dx(int ox, int oz, int pad)
{
    CTPVAE_REQUIRE(ox > 0 && oz > 0, "siddon_dx: sizes must be positive");
    if (!pad) return ox;
    return (int)(std::ceil((std::sqrt((double)ox * ox + (double)oz * oz) + 2.0) / 2.0) * 2.0);
}

int ctpvae_siddon_tables_f32(const float *theta, int dt, float *sin_out, float *cos_out, int *quadrant_out)
{
    CTPVAE_REQUIRE(theta && sin_out && cos_out && quadrant_out, "siddon_tables: null pointer");
    CTPVAE_REQUIRE(dt > 0, "siddon_tables: need at least one angle");
    const double kPi = 3.14159265358979323846;
    for (int p = 0; p < dt; ++p) {
        const float theta_p = std::fmod(theta[p], 2.0f * (float)kPi);
        // libtomo's calc_quadrant (integer-scaled angle; the offset and the bounds are doubles there)
        const int32_t ipi_c = 340870420;
        int32_t theta_i = (int32_t)(theta_p * (float)ipi_c);
        theta_i += (theta_i < 0) ? (2.0f * kPi * ipi_c) : 0;
        quadrant_out[p] = ((theta_i >= 0 && theta_i < 0.5f * kPi * ipi_c) ||
                           (theta_i >= 1.0f * kPi * ipi_c && theta_i < 1.5f * kPi * ipi_c))
                              ? 1 : 0;
        sin_out[p] = std::sin(theta_p);
        cos_out[p] = std::cos(theta_p);
    }
    return CTPVAE_OK;
}

int ctpvae_siddon_fwd_f32(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev,
                          const float *cos_dev, const int *quad_dev, int dt, int dx, float center,
                          float *data_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(obj_dev && sin_dev && cos_dev && quad_dev && data_dev, "siddon_fwd: null pointer");
    CTPVAE_REQUIRE(oy > 0 && ox > 0 && oz > 0 && dt > 0 && dx > 0,
                   "siddon_fwd: sizes must be positive (oy=%d ox=%d oz=%d dt=%d dx=%d)", oy, ox, oz, dt, dx);
    CTPVAE_REQUIRE(oy <= 65535, "siddon_fwd: at most 65535 slices per call (got %d)", oy);
    // utils.c preprocessing(): detector shift
    float mov = ((float)dx - 1) * 0.5f - center;
    if (mov - std::floor(mov) < 0.01f) mov += 0.01f;
    mov += 0.5f;
    const SidGeom g{oy, ox, oz, dt, dx, mov};
    const size_t lds_bytes = (size_t)ox * (oz + 1) * sizeof(float);
    const bool use_lds = lds_bytes <= (size_t)kMaxLdsBytes;
    int ppb = dt;
    while (ppb > 1 && (long long)oy * ceil_div(dt, ppb) < 512) ppb = (ppb + 1) / 2;
    const dim3 grid(ceil_div(dt, ppb), oy), block(256);
    auto launch = [&](auto kernel, size_t shmem) -> int {
        if (shmem > 64 * 1024)
            CTPVAE_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)shmem));
        hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)stream, obj_dev, g, sin_dev, cos_dev, quad_dev,
                           ppb, data_dev);
        CTPVAE_LAUNCH_CHECK("siddon_fwd_kernel");
        return CTPVAE_OK;
    };
    return use_lds ? launch(siddon_fwd_kernel<true>, lds_bytes) : launch(siddon_fwd_kernel<false>, 0);
}

}  // extern "C"
