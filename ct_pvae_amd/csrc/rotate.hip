// rotate.hip -- rotate-and-sum Radon forward (a2/a5) and its backward passes (a4, K2x) for gfx950.
//
// Index arithmetic follows TensorFlow's ImageProjectiveTransformV3 exactly:
//   x_in = (t0*x + t1*y) + t2 ; y_in = (t3*x + t4*y) + t5     (separate fp32 multiplies and adds;
//   this file is compiled with -ffp-contract=off), NEAREST = round half away from zero, BILINEAR =
//   floor + four zero-filled taps.  The zero-padded canvas of pad_phantom is never materialised: a
//   tap is live only if it lands in the H x W core that sits at (py, px) of the PH x PW canvas.
//
// Kernels in this file (the gather-plan kernels, which serve batched NEAREST projection of slices that fit LDS, are
// in rotate_plan.hip):
//   rotate_fwd_fast_kernel        direct forward out of a zero-bordered LDS copy of the slice: bilinear, unpadded
//                                 canvases, and -- TILED -- slices larger than LDS (512 x 512), cut into 64 x 96 tiles
//                                 that are each staged once for all angles, 4 slices interleaved per LDS pixel;
//   rotate_tile_reduce_kernel     adds the tiles' partial sinograms in tile order (+ the log-likelihood epilogue);
//   rotate_bwd_tfcompat_seg_kernel  direct NEAREST backward: an 80-bin cotangent segment per angle and pixel tile;
//   rotate_bwd_tfcompat_fast_kernel bilinear backward (whole cotangent rows in LDS);
//   rotate_fwd_kernel, rotate_bwd_tfcompat_kernel, rotate_bwd_exact_kernel   generic fallbacks, exact transpose.
// Every lane owns one ray (angle a, detector bin j) and walks canvas rows in ascending order -- the summation order of
// reduce_sum(axis=1) as the CPU restatement fixes it -- or, tiled, the rows inside each tile and then the tiles in
// order (oracle_rotate_fwd_tiled); results are reproducible bit for bit against the restatement either way.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "common.h"
#include "lds_stage.h"
#include "loglik_math.h"
#include "rotate_plan.h"
#include "cplan_walk.h"
#include "rotate_dev.h"

namespace ctpvae {


// ---- forward ---------------------------------------------------------------------------------
template <bool USE_LDS, typename T>
__device__ __forceinline__ T core_read(const T *__restrict__ im, const T *lds, int H, int W, int pitch, int r, int c)
{
    if ((unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W)
        return USE_LDS ? lds[r * pitch + c] : im[(size_t)r * W + c];
    return T(0);
}

// T = float: the generic fallback of the fp32 path.  T = double (round 5): the reference's float64 callers
// (ctvae/tomopy_forward_compare.py:52,56 hands xdesign's float64 phantoms to both projectors): TensorFlow computes the
// coordinates and the weights in fp32 whatever the image type, casts each weight to T and multiplies, adds and row-sums in T
// (ImageProjectiveTransformV3's bilinear_interpolation: static_cast<T>(x_ceil - x) * value) -- exactly this loop with T = double.
template <typename T, int INTERP, bool USE_LDS>
__global__ __launch_bounds__(256) void rotate_fwd_kernel(const T *__restrict__ img, RotGeom g,
                                                         const float *__restrict__ T8, int a_per_blk,
                                                         T *__restrict__ sino)
{
    extern __shared__ float lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    const int s = blockIdx.y;
    const int a0 = blockIdx.x * a_per_blk;
    const int na = min(a_per_blk, g.A - a0);
    const T *im = img + (size_t)s * g.H * g.W;
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
        T acc = T(0);
        for (int i = 0; i < g.PH; ++i) {
            const float fi = (float)i;
            const float x = (xj + t1 * fi) + t2;
            const float y = (yj + t4 * fi) + t5;
            T v;
            if (INTERP == CTPVAE_NEAREST) {
                const int ix = (int)round_half_away(x) - g.px;
                const int iy = (int)round_half_away(y) - g.py;
                v = core_read<USE_LDS>(im, lds, g.H, g.W, pitch, iy, ix);
            } else {
                const float yf = floorf(y), xf = floorf(x);
                const float yc = yf + 1.0f, xc = xf + 1.0f;
                const int ix0 = (int)xf - g.px, iy0 = (int)yf - g.py;
                const int ix1 = (int)xc - g.px, iy1 = (int)yc - g.py;
                const T v_yf = (T)(xc - x) * core_read<USE_LDS>(im, lds, g.H, g.W, pitch, iy0, ix0) +
                               (T)(x - xf) * core_read<USE_LDS>(im, lds, g.H, g.W, pitch, iy0, ix1);
                const T v_yc = (T)(xc - x) * core_read<USE_LDS>(im, lds, g.H, g.W, pitch, iy1, ix0) +
                               (T)(x - xf) * core_read<USE_LDS>(im, lds, g.H, g.W, pitch, iy1, ix1);
                v = (T)(yc - y) * v_yf + (T)(y - yf) * v_yc;
            }
            acc += v;
        }
        sino[((size_t)s * g.A + a) * g.PW + j] = acc;
    }
}

// ---- forward, fast path: zero-bordered LDS image, clipped row range, no per-sample bounds test ----
//
// The slice is staged with a zero border of BORDER pixels (1 for NEAREST, 2 for BILINEAR).  A sample's
// integer tap is clamped onto that border with one v_med3_i32 per coordinate, so a tap outside the core
// reads an exact 0 without a compare/select.  Each lane first clips its ray against the core
// (conservatively, in floats), then all lanes of a wave walk the same NUMBER of rows from their own first
// row -- rows outside the clipped range only ever contribute +0.0f, and a lane whose range is short is
// shifted so that it stays inside [0, PH): every visited row is a legitimate term of the sum, taken in
// ascending order, so the result is still bit-identical to the oracle.
//
// Rounding: v_cvt_rpi_i32_f32 is floor(x + 0.5) evaluated exactly (tools/probe_rpi.hip).  It differs
// from round-half-away-from-zero only at negative ties; of those only x == -0.5 can reach a live
// pixel, and only when the canvas has no padding on that side (TIE_FIX).
//
// What the measurements say (tools/tune_rotate.hip, tools/probe_valu.hip, profiles/): at the reference's
// batch sizes there are only ~12 waves of rays per CU, one wave issues a VALU op every ~4.4 cycles, and the
// SIMD saturates near 0.5 op/cycle for this mix -- the loop is VALU-issue bound, LDS bandwidth and HBM are
// idle.  So the per-sample instruction count is what is minimised here: packed fp32 for the coordinates of
// TWO consecutive rows at once, one asm block per pair (hipcc pads every asm statement and every packed-op
// consumer with s_nop), gathers software-pipelined two groups deep with one s_waitcnt per group.

// Two consecutive rows (i, i+1) of one ray, NEAREST: LDS byte addresses (absolute: the LDS base is folded
// into off4) of the two clamped taps.
//   (x_i, x_i+1) = ((xj, xj) + (t1, t1) * (i, i+1)) + (t2, t2), same for y with (yj, t4, t5): v_pk_mul_f32 and
//   v_pk_add_f32 round every lane like the scalar ops, so this is the reference's expression bit for bit;
//   then v_cvt_rpi, clamp onto the zero border, row * pitch4 + col * 4 + off4.
// A packed op cannot forward its result to the next instruction on gfx950 (the compiler pads with s_nop);
// the x and y chains are interleaved so every consumer is at least one instruction behind its producer.
// v60..v63 are scratch inside the block.
template <int SHIFT>   // log2 of the bytes per LDS pixel: 2, or 3 / 4 when 2 / 4 slices are interleaved
__device__ __forceinline__ void nearest_pair_addr(f32x2 &fi, f32x2 basex, f32x2 basey, f32x2 stepx, f32x2 stepy,
                                                  f32x2 shiftx, f32x2 shifty, int xlo, int xhi, int ylo, int yhi,
                                                  int pitch4, int off4, int &addr0, int &addr1)
{
    asm("v_pk_mul_f32 v[60:61], %[sx], %[fi]\n\t"
        "v_pk_mul_f32 v[62:63], %[sy], %[fi]\n\t"
        "v_pk_add_f32 v[60:61], %[bx], v[60:61]\n\t"
        "v_pk_add_f32 v[62:63], %[by], v[62:63]\n\t"
        "v_pk_add_f32 v[60:61], v[60:61], %[hx]\n\t"
        "v_pk_add_f32 v[62:63], v[62:63], %[hy]\n\t"
        "v_pk_add_f32 %[fi], %[fi], 2.0 op_sel_hi:[1,0]\n\t"
        "v_cvt_rpi_i32_f32 %[a0], v60\n\t"
        "v_cvt_rpi_i32_f32 %[a1], v61\n\t"
        "v_cvt_rpi_i32_f32 v62, v62\n\t"
        "v_cvt_rpi_i32_f32 v63, v63\n\t"
        "v_med3_i32 %[a0], %[a0], %[xlo], %[xhi]\n\t"
        "v_med3_i32 %[a1], %[a1], %[xlo], %[xhi]\n\t"
        "v_med3_i32 v62, v62, %[ylo], %[yhi]\n\t"
        "v_med3_i32 v63, v63, %[ylo], %[yhi]\n\t"
        "v_mad_i32_i24 v62, v62, %[p4], %[o4]\n\t"
        "v_mad_i32_i24 v63, v63, %[p4], %[o4]\n\t"
        "v_lshl_add_u32 %[a0], %[a0], %[sh], v62\n\t"
        "v_lshl_add_u32 %[a1], %[a1], %[sh], v63"
        : [fi] "+v"(fi), [a0] "=&v"(addr0), [a1] "=&v"(addr1)
        : [sx] "v"(stepx), [sy] "v"(stepy), [bx] "v"(basex), [by] "v"(basey), [hx] "v"(shiftx), [hy] "v"(shifty),
          [xlo] "v"(xlo), [xhi] "v"(xhi), [ylo] "v"(ylo), [yhi] "v"(yhi), [p4] "s"(pitch4), [o4] "v"(off4),
          [sh] "i"(SHIFT)
        : "v60", "v61", "v62", "v63");
}

#ifdef CTPVAE_TUNE_STAMPS
__device__ long long g_stamps[8 * 65536];
#define CTPVAE_STAMP(slot)                                                                                   \
    do {                                                                                                     \
        long long t_;                                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
        if ((threadIdx.x & 63) == 0)                                                                         \
            g_stamps[8 * ((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) + (slot)] = t_; \
    } while (0)
#else
#define CTPVAE_STAMP(slot)
#endif

// Slices that do not fit LDS (512 x 512 = 1 MiB) are cut into tiles.  A tile is an ordinary slice that sits at its own
// (py, px) of the SAME canvas, read through the slice's row stride -- the transform, the rounding and therefore every
// tap index are unchanged -- and the projector is linear, so the sinogram is the sum of the tiles' sinograms.  Each
// tile is staged ONCE and serves all angles.  A ray of angle a can hit tile t only if its bin lies within `radius` of
// the bin the tile's centre projects to, so a tile owns nb ray slots per angle starting at tile_first_bin(); its
// partial sums go to a workspace [S][tiles][A][nb] and rotate_tile_reduce_kernel adds them in ascending tile order:
//     sino[s][a][j] = ((0 + p_0) + p_1) + ... ,   p_t = sum over canvas rows, ascending, of the taps inside tile t.
// (A different association of the same terms than the row-sequential sum of a slice that fits LDS; the CPU
// restatement has the same tiled mode, oracle_rotate_fwd_tiled.)

template <int INTERP, bool TIE_FIX, bool TILED, int NS = 1>
__global__ __launch_bounds__(1024) void rotate_fwd_fast_kernel(const float *__restrict__ img, RotGeom gfull, TileSpec ts,
                                                               const float *__restrict__ T8, int rays_per_blk,
                                                               float *__restrict__ sino)
{
    constexpr int BORDER = (INTERP == CTPVAE_NEAREST) ? 1 : 2;
    constexpr bool PAIRS = (INTERP == CTPVAE_NEAREST) && !TIE_FIX;   // the asm pair path
    // NS > 1 (tiled launches): NS slices interleaved per LDS pixel share every address computation
    static_assert(NS == 1 || (TILED && PAIRS && (NS == 2 || NS == 4)), "interleaved slices: tiled NEAREST only");
    constexpr int SHIFT = NS == 1 ? 2 : (NS == 2 ? 3 : 4);
    typedef typename PixVec<NS>::type vec_t;
    extern __shared__ float lds[];
    CTPVAE_STAMP(0);
    // g: the geometry this workgroup works in -- the slice itself, or one tile of it as a slice of its own
    RotGeom g = gfull;
    int s = blockIdx.y;
    const float *im;
    float tile_cx = 0.0f, tile_cy = 0.0f;
    if (TILED) {
        const int nt = ts.ntx * ts.nty, v = blockIdx.y;
        s = (v / nt) * NS;   // first slice of this workgroup's group of NS
        int y0, x0;
        tile_rect(gfull, ts, v % nt, y0, x0, g.H, g.W);
        g.py = gfull.py + y0;
        g.px = gfull.px + x0;
        im = img + ((size_t)s * gfull.H + y0) * gfull.W + x0;
        tile_cx = (float)g.px + 0.5f * (float)(g.W - 1);
        tile_cy = (float)g.py + 0.5f * (float)(g.H - 1);
    } else {
        im = img + (size_t)s * g.H * g.W;
    }
    const int nb = TILED ? ts.nb : g.PW;   // ray slots per angle
    const int wb = g.W + 2 * BORDER;
    const int hb = g.H + 2 * BORDER;
    // Row pitch == +1 or -1 (mod 32), chosen per workgroup from the direction its rays' lanes walk: consecutive
    // detector bins step by (t0, t3) pixels, i.e. by t3*pitch + t0 dwords ~ +-t3 + t0 banks.  Picking the sign that
    // makes the two terms add keeps |step| in [1, 1.42] banks per lane: at most 2 lanes of a 32-lane group share a
    // bank at any angle.
    // A tiled launch deals whole classes to workgroups instead: blockIdx.x = 2 * group + class.
    int pitch;
    if (TILED) {
        pitch = pitch_for(wb, (blockIdx.x & 1) != 0);
    } else {
        const int ray_mid = min(blockIdx.x * rays_per_blk + rays_per_blk / 2, g.A * g.PW - 1);
        const float *tm = T8 + 8 * (ray_mid / g.PW);
        const bool same_sign = (tm[0] >= 0.0f) == (tm[3] >= 0.0f);
        pitch = pitch_for(wb, same_sign);
    }

#ifndef CTPVAE_TUNE_NOFILL
    // Stage the slice (16-byte loads, conflict-free ds_write_b32: lds_stage.h), then its zero border.  With NS > 1 the
    // LDS pixel (r, c) is NS consecutive floats, one per slice of the group (a short last group repeats its last slice).
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
        if constexpr (NS == 1) {
            stage_rows(lds + BORDER * pitch + BORDER, im, g.H, g.W, gfull.W, pitch, false, lane, wave, nwaves);
        } else {
            const float *srcs[NS];
#pragma unroll
            for (int n = 0; n < NS; ++n) srcs[n] = im + (size_t)(min(s + n, gfull.S - 1) - s) * gfull.H * gfull.W;
            stage_rows_interleaved<NS>(lds + (BORDER * pitch + BORDER) * NS, srcs, g.H, g.W, gfull.W, pitch, false, lane, wave, nwaves);
        }
        for (int p = threadIdx.x; p < 2 * BORDER * pitch; p += blockDim.x) {
            const int r = p / pitch, c = p - r * pitch;
#pragma unroll
            for (int n = 0; n < NS; ++n) lds[((r < BORDER ? r : hb - 2 * BORDER + r) * pitch + c) * NS + n] = 0.0f;
        }
        for (int p = threadIdx.x; p < g.H * (pitch - g.W); p += blockDim.x) {
            const int r = p / (pitch - g.W), c = p - r * (pitch - g.W);
#pragma unroll
            for (int n = 0; n < NS; ++n) lds[((r + BORDER) * pitch + (c < BORDER ? c : g.W + c)) * NS + n] = 0.0f;
        }
    }
#endif
    CTPVAE_STAMP(1);

    const int nrays = g.A * nb;
    const int ray0 = TILED ? 0 : blockIdx.x * rays_per_blk;
    const int ray_end = TILED ? nrays : min(ray0 + rays_per_blk, nrays);
    // clamp bounds in canvas coordinates and the matching LDS offset
    const int xlo = g.px - BORDER, xhi = g.px + g.W + BORDER - 1 - (INTERP == CTPVAE_BILINEAR ? 1 : 0);
    const int ylo = g.py - BORDER, yhi = g.py + g.H + BORDER - 1 - (INTERP == CTPVAE_BILINEAR ? 1 : 0);
    // clamp bounds pinned in VGPRs (v_med3_i32 takes them as is), byte pitch, absolute byte offset of canvas (0, 0)
    const int xlo_v = pin_vgpr(xlo), xhi_v = pin_vgpr(xhi), ylo_v = pin_vgpr(ylo), yhi_v = pin_vgpr(yhi);
    const int pitch4 = pitch * 4 * NS;
    const int off4_v = pin_vgpr(-(ylo * pitch + xlo) * 4 * NS + (int)(uintptr_t)(lds_cptr)lds);

    // Tiled launches keep, behind the tile: a copy of T8 (float offset rays_per_blk; 0: none, for very many angles) and
    // then the ascending list of the angles of this workgroup's bank class ([0] = their count) and a task counter.
    const int t8_lds_off = TILED ? rays_per_blk : 0;
    int *cls_list = reinterpret_cast<int *>(lds + t8_lds_off + 8 * g.A);
    if (TILED && t8_lds_off > 0) {
        for (int p = threadIdx.x; p < 8 * g.A; p += blockDim.x) lds[t8_lds_off + p] = T8[p];
        if (threadIdx.x < 64) {   // wave 0: classes of 64 angles at a time from one vector load + ballot
            const int lane = threadIdx.x, cls = blockIdx.x & 1;
            int n = 0;
            for (int a0 = 0; a0 < g.A; a0 += 64) {
                const float *tm = T8 + 8 * min(a0 + lane, g.A - 1);
                const bool in_cls = a0 + lane < g.A && ((((tm[0] >= 0.0f) == (tm[3] >= 0.0f)) ? 1 : 0) == cls);
                const unsigned long long m = __ballot(in_cls);
                if (in_cls) cls_list[1 + n + __popcll(m & ((1ull << lane) - 1ull))] = a0 + lane;
                n += __popcll(m);
            }
            if (lane == 0) {
                cls_list[0] = n;
                cls_list[1 + g.A] = 0;   // the task counter
            }
        }
    }

    // per-ray setup: transform row, conservative row range through the core, wave-uniform trip count
    struct Ray {
        float t1, t2, t4, t5, xj, yj;
        int ray, ilo, kmax;
        bool live;
    };
    auto setup = [&](int first_ray, int tiled_a = 0) -> Ray {
        Ray q;
        q.ray = first_ray;
        q.live = q.ray < ray_end;
        const int rr = q.live ? q.ray : ray_end - 1;
        const int a = TILED ? tiled_a : rr / nb;
        int j = rr - a * nb;
        // Tiled: the transform rows were copied behind the tile and are read with ds_read (an explicit LDS address:
        // a global / flat load here would make every task wait, through vmcnt, for the previous task's stores).
        float t6[6];
        if (TILED && t8_lds_off > 0) {
            const int ad = (t8_lds_off + 8 * a) * 4 + (int)(uintptr_t)(lds_cptr)lds;
            const f32x4 u = lds_abs_vec<4>(ad);
            const f32x2 w = lds_abs_vec<2>(ad + 16);
            t6[0] = u.x; t6[1] = u.y; t6[2] = u.z; t6[3] = u.w; t6[4] = w.x; t6[5] = w.y;
        } else {
            const float *t = T8 + 8 * a;
#pragma unroll
            for (int e = 0; e < 6; ++e) t6[e] = t[e];
        }
        if (TILED) {
            q.live = q.live && j < ts.span;                    // slots past the span pass the tile by (partial sum +0) and ...
            j += tile_first_bin(t6, tile_cx, tile_cy, ts.radius);
            q.live = q.live && (unsigned)j < (unsigned)g.PW;   // ... slots off the detector: neither is ever read back
        }
        const float t0 = t6[0], t3 = t6[3];
        q.t1 = t6[1]; q.t2 = t6[2]; q.t4 = t6[4]; q.t5 = t6[5];
        q.xj = t0 * (float)j;
        q.yj = t3 * (float)j;
        // margins of >= 1.5 px around the live zone; fp error of this estimate is far below that
        float lo = 0.0f, hi = (float)g.PH;
        clip_rows(q.xj + q.t2, q.t1, (float)(g.px - 2), (float)(g.px + g.W + 1), lo, hi);
        clip_rows(q.yj + q.t5, q.t4, (float)(g.py - 2), (float)(g.py + g.H + 1), lo, hi);
        lo = fminf(fmaxf(lo, 0.0f), (float)g.PH);
        hi = fminf(fmaxf(hi, -1.0f), (float)g.PH);
        const int ilo = max((int)floorf(lo) - 1, 0);
        const int ihi = min((int)ceilf(hi) + 2, g.PH);
        const int cnt = q.live ? max(ihi - ilo, 0) : 0;
#ifdef CTPVAE_TUNE_KMAX
        q.kmax = min(wave_max_nonneg(cnt) * CTPVAE_TUNE_KMAX, g.PH);  // timing only
#else
        q.kmax = wave_max_nonneg(cnt);            // wave-uniform trip count (SGPR)
#endif
        q.ilo = max(min(ilo, g.PH - q.kmax), 0);  // keep ilo + kmax <= PH: only legitimate rows are visited
        return q;
    };

    // one ray per lane: walk its rows, store its sum
    auto walk = [&](const Ray &q) {
        const float t1 = q.t1, t2 = q.t2, t4 = q.t4, t5 = q.t5, xj = q.xj, yj = q.yj;
        const int ray = q.ray, ilo = q.ilo, kmax = q.kmax;
        const bool live = q.live;
        CTPVAE_STAMP(3);

        vec_t acc = 0.0f;
        int k = 0;
#ifdef CTPVAE_TUNE_NOLOOP
        k = kmax;
#endif
        if constexpr (PAIRS) {
            // ---- NEAREST: two rows per asm block, groups of U gathers, two groups in flight --------------------
            constexpr int U = 6;
            const f32x2 basex = {xj, xj}, basey = {yj, yj}, stepx = {t1, t1}, stepy = {t4, t4};
            const f32x2 shiftx = {t2, t2}, shifty = {t5, t5};
            f32x2 fi = {(float)ilo, (float)ilo + 1.0f};
            const int nblk = (kmax - k) / U;
            if (nblk > 0) {
                vec_t bx[U], by[U];   // ping-pong groups: X holds even blocks, Y odd blocks
                auto issue = [&](vec_t (&buf)[U]) {
#pragma unroll
                    for (int u = 0; u < U; u += 2) {
                        int a0, a1;
                        nearest_pair_addr<SHIFT>(fi, basex, basey, stepx, stepy, shiftx, shifty, xlo_v, xhi_v, ylo_v, yhi_v,
                                          pitch4, off4_v, a0, a1);
#ifdef CTPVAE_TUNE_NOLDS
                        buf[u] = __int_as_float(a0 & 0x3fffff);
                        buf[u + 1] = __int_as_float(a1 & 0x3fffff);
#else
                        buf[u] = lds_abs_vec<NS>(a0);
                        buf[u + 1] = lds_abs_vec<NS>(a1);
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the adds of the older group behind these gathers
                };
                auto consume = [&](vec_t (&buf)[U], bool newer_in_flight) {
                    (void)newer_in_flight;   // hipcc counts the in-order LDS returns itself: lgkmcnt(2U-1 .. U)
#pragma unroll
                    for (int u = 0; u < U; ++u) acc += buf[u];
                    __builtin_amdgcn_sched_barrier(0);
                };
                issue(bx);
                int b = 1;
                for (; b + 1 < nblk; b += 2) {
                    issue(by);
                    consume(bx, true);
                    issue(bx);
                    consume(by, true);
                }
                if (b < nblk) {
                    issue(by);
                    consume(bx, true);
                    consume(by, false);
                } else {
                    consume(bx, false);
                }
                k += nblk * U;
            }
            // remainder rows, one at a time (fi.x is the next row)
            float fr = fi.x;
            for (; k < kmax; ++k) {
                const float x = (xj + t1 * fr) + t2, y = (yj + t4 * fr) + t5;
                fr += 1.0f;
                const int idx = __mul24(med3i(cvt_rpi(y), ylo_v, yhi_v), pitch4) + (med3i(cvt_rpi(x), xlo_v, xhi_v) << SHIFT);
                acc += lds_abs_vec<NS>(idx + off4_v);
            }
        } else {
            // ---- BILINEAR, and NEAREST on an unpadded canvas (TIE_FIX): one row at a time, U in flight ------------
            constexpr int U = (INTERP == CTPVAE_NEAREST) ? 4 : 2;
            float fr = (float)ilo;
            auto sample = [&]() -> float {
                const float x = (xj + t1 * fr) + t2, y = (yj + t4 * fr) + t5;
                fr += 1.0f;
                if (INTERP == CTPVAE_NEAREST) {
                    // x == -0.5 must round to -1 (dead): steer it below the clamp range instead of to pixel 0
                    const float xs = (x == -0.5f) ? -1.0f : x, ys = (y == -0.5f) ? -1.0f : y;
                    const int idx = __mul24(med3i(cvt_rpi(ys), ylo_v, yhi_v), pitch4) + (med3i(cvt_rpi(xs), xlo_v, xhi_v) << 2);
                    return lds_abs(idx + off4_v);
                } else {
                    const float xf = floorf(x), yf = floorf(y);
                    const float xc = xf + 1.0f, yc = yf + 1.0f;
                    const int ad = __mul24(med3i(cvt_flr(y), ylo_v, yhi_v), pitch4) + (med3i(cvt_flr(x), xlo_v, xhi_v) << 2) + off4_v;
                    const float v00 = lds_abs(ad), v01 = lds_abs(ad + 4);
                    const float v10 = lds_abs(ad + pitch4), v11 = lds_abs(ad + pitch4 + 4);
                    const float v_yf = (xc - x) * v00 + (x - xf) * v01;
                    const float v_yc = (xc - x) * v10 + (x - xf) * v11;
                    return (yc - y) * v_yf + (y - yf) * v_yc;
                }
            };
            for (; k + U <= kmax; k += U) {
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = sample();
#pragma unroll
                for (int u = 0; u < U; ++u) acc += v[u];
            }
            for (; k < kmax; ++k) acc += sample();
        }
        CTPVAE_STAMP(4);
#ifdef CTPVAE_TUNE_STAMPS
        if ((threadIdx.x & 63) == 0) g_stamps[8 * ((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) + 6] = kmax;
#endif
        if constexpr (NS == 1) {
            if (live) {
                if constexpr (TILED) {
                    const int nt = ts.ntx * ts.nty;
                    sino[partial_index(blockIdx.y / nt, nt, blockIdx.y % nt, nrays, ray)] = acc;
                } else {
                    sino[(size_t)blockIdx.y * nrays + ray] = acc;
                }
            }
        } else if (live) {
            // partial sums of slices s .. s + NS - 1, tile t (partial_index; slices past the batch hold copies of the last one)
            const int nt = ts.ntx * ts.nty, t = blockIdx.y % nt;
            *reinterpret_cast<typename PixVec<NS>::type *>(sino + partial_index(s, nt, t, nrays, ray)) = acc;
        }
    };

    if (TILED) {
        // (angle, 64-slot block) tasks of this workgroup's bank class: task m = (m / nbk)-th angle of the class list,
        // block m % nbk; group gi of G takes m = gi, gi + G, ..., dealt to its waves round-robin.
        const int cls = blockIdx.x & 1, gi = blockIdx.x >> 1, G = gridDim.x >> 1;
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
        const int nbk = nb >> 6;
        CTPVAE_STAMP(1);
        __syncthreads();
        CTPVAE_STAMP(2);
        auto task = [&](int a, int blk) {
            CTPVAE_STAMP(7);
            // A wave takes two MIRRORED 32-slot runs of the angle's slot range: a tile's chord lengths are symmetric
            // about its centre's bin, so both runs have the same trip count (the outermost waves walk short rays only,
            // instead of every wave carrying a piece of the longest ones).
            const int slot = lane < 32 ? blk * 32 + lane : nb - 32 * (blk + 1) + (lane - 32);
            walk(setup(a * nb + slot, a));
        };
        if (t8_lds_off > 0) {
            // Tasks are handed out dynamically (an LDS counter), the long ones first: m counts blocks from the
            // innermost (longest rays) outwards, angles within a block; group gi of G takes every G-th task.
            const int ncls = __builtin_amdgcn_readfirstlane(cls_list[0]);
            int *next_task = cls_list + 1 + g.A;
            const int ntask = ncls * nbk;
            for (;;) {
                int m = 0;
                if (lane == 0) m = atomicAdd(next_task, 1);
                m = __builtin_amdgcn_readfirstlane(m) * G + gi;
                if (m >= ntask) break;
                const int bi = m / ncls, ai = m - bi * ncls;
                task(__builtin_amdgcn_readfirstlane(cls_list[1 + ai]), nbk - 1 - bi);
            }
        } else {   // no room for the list: scan the angles (a scalar load per angle)
            int mg = 0, mw = 0;
            for (int a = 0; a < g.A; ++a) {
                const float *tm = T8 + 8 * a;
                if ((((tm[0] >= 0.0f) == (tm[3] >= 0.0f)) ? 1 : 0) != cls) continue;
                for (int blk = 0; blk < nbk; ++blk) {
                    const bool mine = mg == gi && mw == wave;
                    if (++mg == G) {
                        mg = 0;
                        if (++mw == nwaves) mw = 0;
                    }
                    if (mine) task(a, blk);
                }
            }
        }
    } else {
        Ray q = setup(ray0 + threadIdx.x);   // runs while the staging loads are in flight
        CTPVAE_STAMP(2);
        __syncthreads();   // (an unconditional barrier: a barrier inside the ray loop makes hipcc wait lgkmcnt(0) in it)
        for (int rbase = ray0; rbase < ray_end; rbase += blockDim.x) {
            if (rbase != ray0) q = setup(rbase + threadIdx.x);
            walk(q);
        }
    }
    CTPVAE_STAMP(5);
}

// sino[s][a][j] = sum over tiles, ascending, of the partial sums of the tiles whose slot range holds bin j.
// Workgroup = up to 1024 bins of one angle for kReduceSlices consecutive slices; every wave owns 64 consecutive bins: it
// lists, 64 tiles at a time and in ascending order, the tiles whose slot range touches its bins (about a quarter of them:
// ballot + popcount into a per-wave LDS list) -- the list depends on the angle and the bins only, so it is built ONCE and
// serves all the workgroup's slices -- then adds their partial sums tile by tile, the slices' loads in flight together.
// (One slice per workgroup made this pass wave-launch bound: 34.5 k waves of a dozen loads each at B=32, 27 us.)
#ifdef CTPVAE_TUNE_REDUCE_S
constexpr int kReduceSlices = CTPVAE_TUNE_REDUCE_S;   // timing builds only
#else
constexpr int kReduceSlices = 8;
#endif
// EPI 1: also write the log-probability of the measured sample under every ray-sum (loglik_math.h); EPI 2: the log-probabilities
// are REDUCED -- one partial sum per (slice, angle, 64-bin block) into epi.part (LogLikEpilogue; partition 1 of
// ctpvae_loglik_object_sums_f32), d lp / d ray-sum stored, ray-sums and log-probabilities only where buffers were given.
template <int EPI>
__device__ __forceinline__ void tile_reduce_wave(const float *__restrict__ partial, const RotGeom &g, const TileSpec &ts,
                                                 const float *__restrict__ T8, float *__restrict__ sino, const LogLikEpilogue &epi)
{
    __shared__ int list_tile[16][64], list_first[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = (blockIdx.x * (blockDim.x >> 6) + wave) * 64, j = j0 + lane, a = blockIdx.y;
    const int s0 = blockIdx.z * kReduceSlices;
    if (j0 >= g.PW) return;                         // (whole waves only: no barrier below)
    const float *t = T8 + 8 * a;
    const int nt = ts.ntx * ts.nty;
    const size_t tstride = (size_t)g.A * ts.nb;     // tile stride, in rays (a ray = the four partial sums of a slice quad)
    const size_t qstride = (size_t)nt * tstride;    // slice-quad stride
    constexpr int kQuads = kReduceSlices / kPartialQuad;
    static_assert(kReduceSlices % kPartialQuad == 0, "the reduce pass takes whole slice quads");
    const f32x4 *pa[kQuads];
#pragma unroll
    for (int q = 0; q < kQuads; ++q)               // quads past the batch re-read the last one, never stored
        pa[q] = reinterpret_cast<const f32x4 *>(partial) + (size_t)min(s0 / kPartialQuad + q, (g.S - 1) / kPartialQuad) * qstride +
                (size_t)a * ts.nb;
    float acc[kReduceSlices];
#pragma unroll
    for (int q = 0; q < kReduceSlices; ++q) acc[q] = 0.0f;
    // the epilogue's operands are requested now and arrive under the sum over tiles (loaded in the epilogue they cost each
    // slice a round trip of its own behind the previous slice's stores: 27.5 us per launch against 17.7 us without epilogue)
    [[maybe_unused]] float ex[kReduceSlices], em[kReduceSlices], epnm = 0.0f, einv = 0.0f;
    if constexpr (EPI != 0) {
        epnm = *epi.pnm, einv = 1.0f / epnm;   // the derivative multiplies by the reciprocal (loglik_math.h)
#pragma unroll
        for (int q = 0; q < kReduceSlices; ++q) {
            const size_t sa = (size_t)min(s0 + q, g.S - 1) * g.A + a;
            em[q] = epi.mask[sa];
            ex[q] = epi.meas[sa * g.PW + min(j, g.PW - 1)];
        }
    }
    for (int base = 0; base < nt; base += 64) {
        const int tile = base + lane;
        bool rel = false;
        int fb = 0;
        if (tile < nt) {
            int y0, x0, h, w;
            tile_rect(g, ts, tile, y0, x0, h, w);
            const float cx = (float)(g.px + x0) + 0.5f * (float)(w - 1), cy = (float)(g.py + y0) + 0.5f * (float)(h - 1);
            fb = tile_first_bin(t, cx, cy, ts.radius);   // the same expression as the tile kernel's
            rel = fb <= j0 + 63 && fb + ts.span > j0;
        }
        const unsigned long long m = __ballot(rel);
        const int n = __popcll(m);
        if (rel) {
            const int pos = __popcll(m & ((1ull << lane) - 1ull));
            list_tile[wave][pos] = tile;
            list_first[wave][pos] = fb;
        }
        __builtin_amdgcn_wave_barrier();              // the list is this wave's own: LDS writes are in order
        // tiles in ascending order; loads unconditional (slot clamped).  The loads of tile i + 2 are requested before tile i is
        // added: one tile at a time the pass was a chain of ~12 dependent round trips to memory per wave.
#ifdef CTPVAE_TUNE_REDUCE_D
        constexpr int D = CTPVAE_TUNE_REDUCE_D;
#else
        constexpr int D = 3;
#endif
        f32x4 v[D][kQuads];
        bool ok[D];
        auto issue = [&](int i, f32x4 (&dst)[kQuads], bool &okd) {
            const int slot = j - list_first[wave][i];
            okd = j < g.PW && (unsigned)slot < (unsigned)ts.span;
            const size_t off = (size_t)list_tile[wave][i] * tstride + (okd ? slot : 0);
#pragma unroll
            for (int q = 0; q < kQuads; ++q) dst[q] = pa[q][off];
        };
#pragma unroll
        for (int d = 0; d < D - 1; ++d)
            if (d < n) issue(d, v[d], ok[d]);
        for (int i = 0; i < n; i += D) {
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (i + d < n) {   // wave-uniform
                    if (i + d + D - 1 < n) issue(i + d + D - 1, v[(d + D - 1) % D], ok[(d + D - 1) % D]);
#pragma unroll
                    for (int q = 0; q < kReduceSlices; ++q)   // + 0.0f leaves the sum unchanged
                        acc[q] += ok[d] ? v[d][q / kPartialQuad][q % kPartialQuad] : 0.0f;
                }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (EPI == 2) {
        const int tpr = (g.PW + 63) >> 6;
#pragma unroll
        for (int q = 0; q < kReduceSlices; ++q) {
            const int s = s0 + q;
            if (s < g.S) {   // wave-uniform
                float lpv = 0.0f;
                if (j < g.PW) {
                    const size_t o = ((size_t)s * g.A + a) * g.PW + j;
                    if (sino) sino[o] = acc[q];
                    lpv = epi.eval_loaded(o, em[q], ex[q], epnm, einv, acc[q]);
                }
                const float tot = wave_sum(lpv);
                if (lane == 0) epi.store_part(((size_t)s * g.A + a) * tpr + (j0 >> 6), tot);
            }
        }
        return;
    }
    if (j >= g.PW) return;
#pragma unroll
    for (int q = 0; q < kReduceSlices; ++q) {
        const int s = s0 + q;
        if (s < g.S) {
            const size_t o = ((size_t)s * g.A + a) * g.PW + j;
            sino[o] = acc[q];
            if constexpr (EPI == 1) epi.write_loaded(o, em[q], ex[q], epnm, einv, acc[q]);
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(1024) void rotate_tile_reduce_kernel(const float *__restrict__ partial, RotGeom g, TileSpec ts,
                                                                  const float *__restrict__ T8, float *__restrict__ sino,
                                                                  LogLikEpilogue epi)
{
    tile_reduce_wave<EPI>(partial, g, ts, T8, sino, epi);
    if constexpr (EPI == 2) {
        // round 4: the workgroup that finishes a group of kReduceSlices slices last (all angles, all bin blocks) adds their
        // partials in the fixed order -- no loglik_sum_partials_kernel launch behind the pass
        __shared__ int last_flag;
        if (epi.sum != nullptr && arrived_last(epi.arrive + blockIdx.z * kReduceSlices, gridDim.x * gridDim.y, &last_flag)) {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, tpr = (g.PW + 63) >> 6;
            for (int q = wave; q < kReduceSlices; q += nwaves) {
                const int sl = blockIdx.z * kReduceSlices + q;
                if (sl < g.S) {
                    const float total = object_sum_of_parts<true>(epi.part + (size_t)sl * g.A * tpr, g.A, tpr, lane);
                    if (lane == 0) epi.sum[sl] = total;
                }
            }
        }
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

// ---- backward, TensorFlow-compatible, fast path ---------------------------------------------------
//
// One wave owns 64 consecutive columns of one image row at a time; a thread keeps PPT pixels of ONE
// column (rows r0, r0+RS, ...) in registers, so t0*x and t3*x are computed once per angle and thread.
// The cotangent rows of a chunk of angles sit in LDS with zero cells on both sides; a tap that TensorFlow
// would zero-fill is steered to a zero cell with one select, so the gather needs no branch.  Transform
// rows are wave-uniform and come through scalar loads.  Every pixel adds its angles in ascending order
// (bit-identical to the oracle).
//
// NEAREST: a tap is live iff -0.5 < x' < PW-0.5 and -0.5 < y' < PH-0.5 (round half away from zero lands in
// [0, P)); inside that interval v_cvt_rpi_i32_f32 equals the reference rounding exactly.
template <int INTERP, int PPT>
__global__ __launch_bounds__(256) void rotate_bwd_tfcompat_fast_kernel(const float *__restrict__ gsino, RotGeom g,
                                                                       const float *__restrict__ Tinv8, int chunk_a,
                                                                       float *__restrict__ gimg)
{
    constexpr int ZL = (INTERP == CTPVAE_NEAREST) ? 0 : 2;   // zero cells in front of a cotangent row
    constexpr int ZR = (INTERP == CTPVAE_NEAREST) ? 1 : 2;   // ... and behind it
    extern __shared__ float lds[];
    const int pitchg = g.PW + ZL + ZR;
    const int s = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int c = blockIdx.x * 64 + lane;                     // column of this thread
    const int r0 = blockIdx.y * (nwaves * PPT) + wave;        // first row; rows step by nwaves
    const bool col_ok = c < g.W;
    const float fx = (float)(c + g.px);
    const float x_hi = (float)g.PW - 0.5f, y_hi = (float)g.PH - 0.5f;

    float acc[PPT];
    float fy[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        acc[k] = 0.0f;
        fy[k] = (float)(r0 + k * nwaves + g.py);
    }

    for (int ac = 0; ac < g.A; ac += chunk_a) {
        const int na = min(chunk_a, g.A - ac);
        __syncthreads();
        const float *src = gsino + ((size_t)s * g.A + ac) * g.PW;
        for (int p = threadIdx.x; p < na * pitchg; p += blockDim.x) {
            const int al = p / pitchg, q = p - al * pitchg - ZL;
            lds[p] = ((unsigned)q < (unsigned)g.PW) ? src[(size_t)al * g.PW + q] : 0.0f;
        }
        __syncthreads();
        for (int al = 0; al < na; ++al) {
            const float *t = Tinv8 + 8 * (size_t)(ac + al);   // wave-uniform: scalar loads
            const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
            const float *grow = lds + al * pitchg + ZL;
            const float xa = t0 * fx, ya = t3 * fx;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const float x = (xa + t1 * fy[k]) + t2;
                const float y = (ya + t4 * fy[k]) + t5;
                if (INTERP == CTPVAE_NEAREST) {
                    const bool ok = (x > -0.5f) & (x < x_hi) & (y > -0.5f) & (y < y_hi);
                    const int ixr = cvt_rpi(x);                // unconditional: no branch around the gather
                    const int ix = ok ? ixr : g.PW;            // g.PW is the zero cell
                    acc[k] += grow[ix];
                } else {
                    const float xf = floorf(x), yf = floorf(y);
                    const float xc = xf + 1.0f, yc = yf + 1.0f;
                    const int ix = med3i(cvt_flr(x), -2, g.PW);
                    const float h = (xc - x) * grow[ix] + (x - xf) * grow[ix + 1];
                    // row taps: yf and yf+1 must lie in [0, PH)
                    const float v_yf = ((yf >= 0.0f) & (yf < (float)g.PH)) ? h : 0.0f;
                    const float v_yc = ((yc >= 0.0f) & (yc < (float)g.PH)) ? h : 0.0f;
                    acc[k] += (yc - y) * v_yf + (y - yf) * v_yc;
                }
            }
        }
    }
    if (col_ok) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int r = r0 + k * nwaves;
            if (r < g.H) gimg[((size_t)s * g.H + r) * g.W + c] = acc[k];
        }
    }
}

// ---- backward, TensorFlow-compatible, NEAREST: cotangent SEGMENTS in LDS ---------------------------------------
//
// A 64-column x 32-row pixel tile reads, for one angle, only the bins its rectangle projects to: at most
// sqrt(64^2 + 32^2) = 71.6 bins for a rotation.  So the workgroup stages an 80-bin segment per angle (320 B instead of
// the whole detector row: 2.9 KiB at 728 bins) -- all angles of a launch in one chunk, three barriers in total -- and
// classifies every angle once from the tile's corners:
//   0  the whole tile maps inside the canvas (always, on a padded canvas): the tap needs no bounds test, and the
//      second coordinate is not needed at all -- 6 VALU ops per tap;
//   1  some pixel may map outside: the reference's zero fill, by steering the tap to the segment's zero cell;
//   2  the segment would not hold the span (the table row is not a rotation): bounds-tested reads from global memory.
// Every pixel adds its angles in ascending order (bit-identical to the oracle), whatever the class.
constexpr int kSegBins = 80, kSegPitch = kSegBins + 1;   // cell kSegBins of every segment is 0.0f
template <int PPT, int NS>   // NS = 2: two slices per workgroup, segments interleaved as float2 -- one coordinate, one
                             // convert, one address and one ds_read_b64 per tap serve both slices
__global__ __launch_bounds__(256) void rotate_bwd_tfcompat_seg_kernel(const float *__restrict__ gsino, RotGeom g,
                                                                      const float *__restrict__ Tinv8, int chunk_a,
                                                                      SliceScale scale, float *__restrict__ gimg,
                                                                      const int *__restrict__ sel, int a_plan)
{
    // sel != nullptr: cotangent row k of the g.A rows belongs to row sel[k] of a DENSE table of a_plan angles (the
    // training loop's per-step angle subset, ctvae/helper_functions.py:350-357); a bad index cannot leave the table
    auto table_row = [&](int k) { return sel ? min(max(sel[k], 0), a_plan - 1) : k; };
    typedef typename PixVec<NS>::type vec_t;
    constexpr int SHIFT = NS == 1 ? 2 : 3;
    // [chunk_a][kSegPitch] segment cells (NS floats each), then per angle: (first bin, class) ints and (t0, t1, t2,
    // segment byte base) for the all-inside fast loop, which reads all it needs per angle with one broadcast ds_read_b128
    extern __shared__ float lds[];
    int *meta = reinterpret_cast<int *>(lds + chunk_a * kSegPitch * NS);
    f32x4 *meta4 = reinterpret_cast<f32x4 *>(lds + ((chunk_a * (kSegPitch * NS + 2) + 3) & ~3));
    const int s = blockIdx.z * NS;
    const bool has2 = NS == 2 && s + 1 < g.S;     // an odd batch ends with a half-empty pair (slice s read twice)
    const float k0 = scale.at(s), k1 = has2 ? scale.at(s + 1) : 1.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * (nwaves * PPT) + wave;   // rows r0, r0 + nwaves, ...
    // (lanes and rows of a ragged tile that hang over the image repeat its last column / row: never stored, and inside the rectangle)
    const float fx = (float)(min(c, g.W - 1) + g.px);
    // The tile's rectangle in canvas coordinates, CLIPPED to the image (round 5): the whole 64-column rectangle of a ragged tile reaches
    // over the image and, on a padded canvas, over the canvas -- the angle then left the all-inside class and the workgroup the fast
    // loop: every width that is no multiple of 64 ran 2.5 x slower (136 .. 184 and 200 px: 14 us at 8 slices x 20 angles; 128 / 192: 5.6)
    const float X0 = (float)(blockIdx.x * 64 + g.px), X1 = X0 + (float)min(63, g.W - 1 - (int)blockIdx.x * 64);
    const float Y0 = (float)(blockIdx.y * (nwaves * PPT) + g.py), Y1 = Y0 + (float)min(nwaves * PPT - 1, g.H - 1 - (int)blockIdx.y * (nwaves * PPT));
    const float x_hi = (float)g.PW - 0.5f, y_hi = (float)g.PH - 0.5f;
    const int lds_base = (int)(uintptr_t)(lds_cptr)lds;

    vec_t acc[PPT];
    float fy[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        acc[k] = 0.0f;
        fy[k] = (float)(min(r0 + k * nwaves, g.H - 1) + g.py);
    }

    for (int ac = 0; ac < g.A; ac += chunk_a) {
        const int na = min(chunk_a, g.A - ac);
        if (ac > 0) __syncthreads();
        int any_outside = 0;
        for (int al = threadIdx.x; al < na; al += blockDim.x) {
            const float *t = Tinv8 + 8 * (size_t)table_row(ac + al);
            const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
            const float xa = (t0 * X0 + t1 * Y0) + t2, xb = (t0 * X1 + t1 * Y0) + t2;
            const float xc = (t0 * X0 + t1 * Y1) + t2, xd = (t0 * X1 + t1 * Y1) + t2;
            const float ya = (t3 * X0 + t4 * Y0) + t5, yb = (t3 * X1 + t4 * Y0) + t5;
            const float yc = (t3 * X0 + t4 * Y1) + t5, yd = (t3 * X1 + t4 * Y1) + t5;
            const float xmin = fminf(fminf(xa, xb), fminf(xc, xd)), xmax = fmaxf(fmaxf(xa, xb), fmaxf(xc, xd));
            const float ymin = fminf(fminf(ya, yb), fminf(yc, yd)), ymax = fmaxf(fmaxf(ya, yb), fmaxf(yc, yd));
            int cls = 0, first = 0;
            if (!(xmax - xmin <= (float)(kSegBins - 6)) || !(fabsf(xmin) < 1.0e6f)) {
                cls = 2;   // (also NaN / absurd rows)
            } else {
                first = (int)floorf(xmin) - 2;
                if (!(xmin > 0.5f && xmax < x_hi - 1.0f && ymin > 0.5f && ymax < y_hi - 1.0f)) cls = 1;
            }
            meta[2 * al] = first;
            meta[2 * al + 1] = cls;
            meta4[al] = f32x4{t0, t1, t2, __int_as_float((al * kSegPitch - first) * 4 * NS + lds_base)};
            any_outside |= cls;
        }
        const bool all_inside = __syncthreads_or(any_outside) == 0;   // (also the barrier behind the table)
        const float *src = gsino + ((size_t)s * g.A + ac) * g.PW;
        const size_t src2 = has2 ? (size_t)g.A * g.PW : 0;      // offset of the pair's second slice
        {
            constexpr int U = 8 / NS;   // cells in flight per thread; loads unconditional (clamped), the select comes after
            const int ncell = na * kSegPitch;
            for (int p0 = threadIdx.x; p0 < ncell; p0 += U * blockDim.x) {
                vec_t v[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = min(p0 + u * (int)blockDim.x, ncell - 1);
                    const int al = p / kSegPitch, q = p - al * kSegPitch;
                    const int j = meta[2 * al] + q;
                    ok[u] = q < kSegBins && (unsigned)j < (unsigned)g.PW;
                    const float *cell = src + al * g.PW + min(max(j, 0), g.PW - 1);
                    if constexpr (NS == 1) {
                        v[u] = cell[0];
                    } else {
                        v[u].x = cell[0];
                        v[u].y = cell[src2];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + u * (int)blockDim.x;
                    if (p < ncell) reinterpret_cast<vec_t *>(lds)[p] = ok[u] ? v[u] : vec_t(0.0f);
                }
            }
        }
        __syncthreads();

        if (all_inside) {
            // Every angle of the chunk maps the whole tile inside the canvas (a padded canvas always does): no class
            // test, no scalar loads -- the angle's three coefficients and segment base arrive with one broadcast
            // ds_read_b128, fetched one angle ahead, and the gathers of angle al are consumed under those of al + 1.
            auto taps = [&](const f32x4 m, vec_t (&v)[PPT]) {
                const float xa = m.x * fx;
                const int k4 = __float_as_int(m.w);
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const float x = (xa + m.y * fy[k]) + m.z;
                    int addr;   // (round(x) - first) * cell bytes + segment base: one convert, one shift-add
                    asm("v_cvt_rpi_i32_f32 %0, %1\n\tv_lshl_add_u32 %0, %0, %3, %2" : "=&v"(addr) : "v"(x), "v"(k4), "i"(SHIFT));
                    v[k] = lds_abs_vec<NS>(addr);
                }
            };
            vec_t va[PPT], vb[PPT];
            f32x4 m = meta4[0];
            f32x4 mn = meta4[min(1, na - 1)];
            taps(m, va);
            for (int al = 1; al + 1 < na; al += 2) {      // angles al (-> vb) and al + 1 (-> va)
                m = mn;
                mn = meta4[al + 1];
                taps(m, vb);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < PPT; ++k) acc[k] += va[k];
                m = mn;
                mn = meta4[min(al + 2, na - 1)];
                taps(m, va);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < PPT; ++k) acc[k] += vb[k];
            }
            if ((na & 1) == 0) {                           // even count: one angle (na - 1) is still to be gathered
                taps(mn, vb);
#pragma unroll
                for (int k = 0; k < PPT; ++k) acc[k] += va[k];
#pragma unroll
                for (int k = 0; k < PPT; ++k) acc[k] += vb[k];
            } else {
#pragma unroll
                for (int k = 0; k < PPT; ++k) acc[k] += va[k];
            }
        } else
        for (int al = 0; al < na; ++al) {
            const float *t = Tinv8 + 8 * (size_t)table_row(ac + al);   // wave-uniform: scalar loads
            const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
            const int first = __builtin_amdgcn_readfirstlane(meta[2 * al]);
            const int cls = __builtin_amdgcn_readfirstlane(meta[2 * al + 1]);
            const float xa = t0 * fx, ya = t3 * fx;
            const vec_t *seg = reinterpret_cast<const vec_t *>(lds) + al * kSegPitch;
            const float *grow = src + (size_t)al * g.PW;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const float x = (xa + t1 * fy[k]) + t2;
                const float y = (ya + t4 * fy[k]) + t5;
                // class 0 needs no test; classes 1 and 2 apply the reference's zero fill
                const bool ok = cls == 0 || ((x > -0.5f) & (x < x_hi) & (y > -0.5f) & (y < y_hi));
                const int ix = ok ? cvt_rpi(x) : 0;
                if (cls != 2) {
                    acc[k] += seg[ok ? ix - first : kSegBins];
                } else if constexpr (NS == 1) {
                    acc[k] += ok ? grow[ix] : 0.0f;
                } else {
                    vec_t gv;
                    gv.x = ok ? grow[ix] : 0.0f;
                    gv.y = ok ? grow[src2 + ix] : 0.0f;
                    acc[k] += gv;
                }
            }
        }
    }
    if (c < g.W) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int r = r0 + k * nwaves;
            if (r < g.H) {
                if constexpr (NS == 1) {
                    gimg[((size_t)s * g.H + r) * g.W + c] = k0 * acc[k];
                } else {
                    gimg[((size_t)s * g.H + r) * g.W + c] = k0 * acc[k].x;
                    if (has2) gimg[((size_t)(s + 1) * g.H + r) * g.W + c] = k1 * acc[k].y;
                }
            }
        }
    }
}

// ---- the segment backward over a STEP PLAN (round 3) ---------------------------------------------------------------------
// The kernel above spends five index operations per tap (multiply, two adds, round, shift-add) on x_in = (t0 x + t1 y) + t2.
// Down a column of pixels x_in moves by t1 per row, |t1| <= 1, so the rounded tap stays or steps by one, always the same way
// for an angle: a lane that owns EIGHT CONSECUTIVE rows of a column needs its first row's tap and seven bits.  The plan holds,
// per (angle, row octet, column), one u16: bits 0..6 the first row's tap relative to the segment the tile stages for that
// angle (the same `first` the kernel computes), bits 7..13 "the tap steps" for rows 1..7 -- written by a kernel that evaluates
// the reference arithmetic exactly as the kernel above does, so the taps, their order and the sums are the same bits.  Per
// tap: one bit-field extract and one multiply-add onto the running LDS address.  Geometries the code cannot hold (a pixel
// that samples outside the canvas at some angle -- unpadded canvases --, a row that is not a rotation) raise the plan's
// overflow word and the caller keeps the kernel above.
// (Round 3, second half) The word became EIGHT BYTES: byte 0 the first row's tap, bytes 1..7 the byte offsets of rows 1..7's
// taps from a per-angle base (8 B cells: offset = 8 x steps so far for an angle whose taps step up the bins; 8 x (14 - steps) from
// a base 112 B lower for one that steps down, so that every offset is added; a row may step twice), and a tap's address is ONE SDWA add of a byte
// onto that base: 163 -> 120 vector instructions per six angles.  Measured on one box: 101 -> 96 us at 400 x 128 x 128 x 180
// angles, 59.9 -> 57.6 us at 32 x 512 x 512 x 90 -- a quarter fewer instructions bought 3-5 %: the kernel now waits on its LDS
// gathers and staging as much as on issue.  23.6 MB for 512 x 512 x 90 angles (5.9 MB as u16 words); the workgroups of one tile
// column run on one XCD (blockIdx.x == XCD), whose L2 holds that column's 3 MB of the plan.
constexpr int kStepRows = 8;           // rows per lane = rows per plan word
constexpr int kStepTileRows = 32;      // four waves of eight rows: the tile whose segment `first` the plan is relative to
constexpr int kStepCell = 8;           // bytes of an LDS cell the plan's offsets are in: a float2 (slice pairs only)
constexpr int kStepMax = 2 * (kStepRows - 1);   // steps a column's tap may take over a word's eight rows: up to TWO per row --
                                                // |t1| <= 1 moves it by one, but within ~1e-3 rad of 90 / 270 degrees |t1| = 1 - 1e-5
                                                // and fp32 rounding makes it two now and then (a seeded soak found 264 x 278 at
                                                // -4.71696 rad; a plan of one-step words overflowed there for ALL its angles)

struct StepLayout {
    int H8, Wpad;
    long long off_flag, bytes;
};
static StepLayout step_layout(int H, int W, int A)
{
    StepLayout L;
    L.H8 = ceil_div(H, kStepTileRows) * (kStepTileRows / kStepRows);
    L.Wpad = ceil_div(W, 64) * 64;
    L.off_flag = (long long)A * L.H8 * L.Wpad * 8;
    L.off_flag = (L.off_flag + 255) / 256 * 256;
    L.bytes = L.off_flag + 256;
    return L;
}

// the segment a 64 x 32 tile stages for an angle: its first bin, or "cannot" (the tile spans more bins than a segment
// holds) -- ONE definition for the plan builder and the kernel.  (The whole rectangle counts, also where a ragged tile hangs
// over the image: those pixels are never stored, but their taps must not decide the segment differently in the two places.)
__device__ __forceinline__ bool step_segment_first(const float *__restrict__ t, float X0, float Y0, int &first)
{
    const float X1 = X0 + 63.0f, Y1 = Y0 + (float)(kStepTileRows - 1);
    const float t0 = t[0], t1 = t[1], t2 = t[2];
    const float xa = (t0 * X0 + t1 * Y0) + t2, xb = (t0 * X1 + t1 * Y0) + t2;
    const float xc = (t0 * X0 + t1 * Y1) + t2, xd = (t0 * X1 + t1 * Y1) + t2;
    const float xmin = fminf(fminf(xa, xb), fminf(xc, xd)), xmax = fmaxf(fmaxf(xa, xb), fmaxf(xc, xd));
    first = 0;
    if (!(xmax - xmin <= (float)(kSegBins - 6)) || !(fabsf(xmin) < 1.0e6f)) return false;
    first = (int)floorf(xmin) - 2;
    return true;
}

// one lane per (column, row octet, angle)
__global__ __launch_bounds__(64) void rotate_bwd_step_plan_kernel(RotGeom g, const float *__restrict__ Tinv8, StepLayout L,
                                                                 char *__restrict__ plan)
{
    const int c = blockIdx.x * 64 + threadIdx.x, rg = blockIdx.y, a = blockIdx.z;
    const float *t = Tinv8 + 8 * (size_t)a;
    int first;
    const bool ok_tile = step_segment_first(t, (float)(blockIdx.x * 64 + g.px),
                                            (float)((rg / (kStepTileRows / kStepRows)) * kStepTileRows + g.py), first);
    const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
    const int sigma = t1 > 0.0f ? 1 : t1 < 0.0f ? -1 : 0;
    const float fx = (float)(c + g.px);
    const float xa = t0 * fx, ya = t3 * fx;
    const float x_hi = (float)g.PW - 0.5f, y_hi = (float)g.PH - 0.5f;
    bool ok = true;
    unsigned long long word = 0;
    int prev = 0, steps = 0;
    for (int k = 0; k < kStepRows; ++k) {
        const int r = rg * kStepRows + k;
        const float fy = (float)(r + g.py);
        const float x = (xa + t1 * fy) + t2, y = (ya + t4 * fy) + t5;
        const int tap = cvt_rpi(x);
        const bool live = c < g.W && r < g.H;      // pixels a ragged tile hangs over the image with are never stored
        // a live pixel must sample INSIDE the canvas (else the reference's zero fill applies: the direct kernel's business),
        // away from the one tie where v_cvt_rpi and std::round part, inside its tile's segment
        if (live) ok = ok && ok_tile && x > 0.5f && x < x_hi - 1.0f && y > 0.5f && y < y_hi - 1.0f;
        if (k == 0) {
            const int rel = tap - first;
            if (live) ok = ok && rel >= 0 && rel < kSegBins;
            word = (unsigned long long)((unsigned)rel & 127u);
        } else {
            const int dlt = tap - prev;
            const int ds = dlt * sigma;     // steps of this row: 0, 1 or (rarely) 2, the angle's way
            if (live) ok = ok && (dlt == 0 || ds == 1 || ds == 2) && tap - first >= 0 && tap - first < kSegBins;
            steps += min(max(ds, 0), 2);
            // byte offset of row k's tap from the angle's base (see rotate_bwd_stepped_kernel: kStepMax cells below the first
            // tap for a down-stepping angle)
            const int off = kStepCell * (sigma < 0 ? kStepMax - steps : steps);
            word |= (unsigned long long)(unsigned)off << (8 * k);
        }
        prev = tap;
    }
    reinterpret_cast<unsigned long long *>(plan)[((size_t)a * L.H8 + rg) * L.Wpad + c] = word;
    if (!ok) *reinterpret_cast<int *>(plan + L.off_flag) = 1;
}

// NS = 2: a slice pair per workgroup.  NS = 4 (round 4): TWO pairs -- the second pair's segments in a second array of float2
// cells kStepPairGap bytes above the first's, read with the SAME address registers (the gap rides in the ds_read's offset
// field): the plan word, its unpacking and the eight address adds of an angle serve four slices instead of two, 26 instead
// of 2 x 18 vector instructions per angle and lane.  Same taps, same order of the angles: same bits.
#ifdef CTPVAE_TUNE_STEP_CHUNK4
constexpr int kStepChunk4 = CTPVAE_TUNE_STEP_CHUNK4;
#else
constexpr int kStepChunk4 = 30;
#endif
//                                       // angles per staged chunk of the two-pair kernel
constexpr int kStepPairGap = kStepChunk4 * kSegPitch * kStepCell;      // bytes between the pairs' arrays (ds_read offset: < 64 KB)
template <int NS>
__global__ __launch_bounds__(256) void rotate_bwd_stepped_kernel(const float *__restrict__ gsino, RotGeom g,
                                                                 const float *__restrict__ Tinv8, int chunk_a, StepLayout L,
                                                                 const char *__restrict__ plan, SliceScale scale,
                                                                 float *__restrict__ gimg)
{
    static_assert(NS == 2 || NS == 4, "the step plan's offsets are in float2 cells: slices ride as (half-empty) pairs");
    constexpr int NP = NS / 2;           // pairs per workgroup
    constexpr int PPT = kStepRows;
    // [chunk_a][kSegPitch] segment cells (a float2 each) per pair, then per angle (first bin) and (segment byte base as the
    // plan's offsets count from it, the first row's distance above it)
    extern __shared__ float lds[];
    const int cell_floats = NP == 1 ? chunk_a * kSegPitch * 2 : kStepPairGap / 4 + chunk_a * kSegPitch * 2;
    int *first_s = reinterpret_cast<int *>(lds + cell_floats);
    const int s = blockIdx.z * NS;
    const int nlive = min(NS, g.S - s);          // slices past the batch re-read slice s and are never stored
    float kscale[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) kscale[n] = n < nlive ? scale.at(s + n) : 1.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int rg = blockIdx.y * (kStepTileRows / kStepRows) + wave;     // this lane's rows: rg * 8 .. rg * 8 + 7
    const float X0 = (float)(blockIdx.x * 64 + g.px), Y0 = (float)(blockIdx.y * kStepTileRows + g.py);
    const int lds_base = (int)(uintptr_t)(lds_cptr)lds;
    const uint2 *pl = reinterpret_cast<const uint2 *>(plan) + (size_t)rg * L.Wpad + c;
    const size_t astride = (size_t)L.H8 * L.Wpad;

    f32x2 acc[NP][PPT];
#pragma unroll
    for (int h = 0; h < NP; ++h)
#pragma unroll
        for (int k = 0; k < PPT; ++k) acc[h][k] = 0.0f;

    for (int ac = 0; ac < g.A; ac += chunk_a) {
        const int na = min(chunk_a, g.A - ac);
        if (ac > 0) __syncthreads();
        for (int al = threadIdx.x; al < na; al += blockDim.x) {
            const float *t = Tinv8 + 8 * (size_t)(ac + al);
            int first;
            step_segment_first(t, X0, Y0, first);
            first_s[al] = first;
        }
        // which of the chunk's angles step DOWN the bins (their offsets count from a base kStepMax cells lower): one bit per
        // angle in a scalar register pair.  (Round 4: this and the segment's base were an LDS word per angle, read in the walk;
        // LDS reads return in order, so waiting for that word drained the sixteen gathers in flight, twice per angle pair.)
        const unsigned long long down = __ballot(lane < na && Tinv8[8 * (size_t)(ac + min(lane, na - 1)) + 1] < 0.0f);
        __syncthreads();
        const float *src = gsino + ((size_t)s * g.A + ac) * g.PW;
        size_t soff[NS];
#pragma unroll
        for (int n = 0; n < NS; ++n) soff[n] = n < nlive ? (size_t)n * g.A * g.PW : 0;
#ifndef CTPVAE_TUNE_STEP_NOSTAGE   // (timing builds only)
        {
            // ALL of a chunk's cells are requested before the first is written (round 4): ceil(30 x 81 / 256) = 10 per thread with two
            // pairs, 2 x 8 with one -- two cells per pass made a chunk five dependent round trips to memory, with every
            // workgroup of the CU waiting at the same time
#ifdef CTPVAE_TUNE_STEP_U
            constexpr int U = CTPVAE_TUNE_STEP_U;
#else
            constexpr int U = NS == 4 ? 10 : 8;
#endif
            const int ncell = na * kSegPitch;
            for (int p0 = threadIdx.x; p0 < ncell; p0 += U * blockDim.x) {
                f32x2 v[U][NP];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = min(p0 + u * (int)blockDim.x, ncell - 1);
                    const int al = p / kSegPitch, q = p - al * kSegPitch;
                    const int j = first_s[al] + q;
                    ok[u] = q < kSegBins && (unsigned)j < (unsigned)g.PW;
                    const float *cell = src + al * g.PW + min(max(j, 0), g.PW - 1);
#pragma unroll
                    for (int h = 0; h < NP; ++h) {
                        v[u][h].x = cell[soff[2 * h]];
                        v[u][h].y = cell[soff[2 * h + 1]];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + u * (int)blockDim.x;
                    if (p < ncell) {
#pragma unroll
                        for (int h = 0; h < NP; ++h)
                            reinterpret_cast<f32x2 *>(lds + h * (kStepPairGap / 4))[p] = ok[u] ? v[u][h] : f32x2(0.0f);
                    }
                }
            }
        }
#endif
        __syncthreads();

        // per angle: the plan word (fetched kAhead angles ahead), the segment base (scalar arithmetic), eight gathers per pair
        // whose addresses are one SDWA add of a plan byte apart from the base
        auto addr = [&](const uint2 w, const int ang, int (&a)[PPT]) {
            static_assert(PPT == 8, "eight rows per plan word");
            // (segment base as the plan's offsets count from it, what the first row's tap lies above it)
            const int drop = ((down >> ang) & 1ull) ? kStepCell * kStepMax : 0;
            const int seg = ang * (kSegPitch * kStepCell) + lds_base - drop;
            int t, base;
            asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(t) : "v"(w.x));
            base = t + seg;
            a[0] = base + drop;
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a[1]) : "v"(base), "v"(w.x));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a[2]) : "v"(base), "v"(w.x));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a[3]) : "v"(base), "v"(w.x));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a[4]) : "v"(base), "v"(w.y));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a[5]) : "v"(base), "v"(w.y));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a[6]) : "v"(base), "v"(w.y));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a[7]) : "v"(base), "v"(w.y));
        };
        auto gather = [&](const int (&a)[PPT], auto h_tag, f32x2 (&v)[PPT]) {
            constexpr int OFF = decltype(h_tag)::value * kStepPairGap;
#pragma unroll
            for (int k = 0; k < PPT; ++k) v[k] = lds_abs_vec<2>(a[k] + OFF);
        };
        auto add = [&](f32x2 (&sum)[PPT], const f32x2 (&v)[PPT]) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) sum[k] += v[k];
        };
        // plan words are fetched kAhead angles ahead of their use (an L2 round trip is several angles long)
#ifdef CTPVAE_TUNE_STEP_AHEAD
        constexpr int kAhead = CTPVAE_TUNE_STEP_AHEAD;
#else
        constexpr int kAhead = 6;
#endif
        const uint2 *pa = pl + (size_t)ac * astride;
        uint2 wq[kAhead];
        f32x2 va[NP][PPT], vb[NP][PPT];
        int a[PPT];
        // (a finer pipeline for two pairs -- angle n + 1's first pair requested as soon as angle n's first pair is added, 8 to 16
        // gathers in flight all the time -- needed a wave-uniform branch per angle for the chunk's tail and ran 57.9 us against
        // this loop's 47.5: profiles/r04_step_pairs.txt)
        auto taps = [&](const uint2 w, const int ang, f32x2 (&v)[NP][PPT]) {
            addr(w, ang, a);
            gather(a, std::integral_constant<int, 0>{}, v[0]);
            if constexpr (NP == 2) gather(a, std::integral_constant<int, 1>{}, v[1]);
        };
        auto add_all = [&](const f32x2 (&v)[NP][PPT]) {
#pragma unroll
            for (int h = 0; h < NP; ++h) add(acc[h], v[h]);
        };
#ifndef CTPVAE_TUNE_STEP_NOWALK   // (timing builds only)
#pragma unroll
        for (int q = 0; q < kAhead; ++q) wq[q] = pa[(size_t)min(q, na - 1) * astride];
        int al = 0;
        for (; al + kAhead <= na; al += kAhead) {      // kAhead angles per trip, two register sets of gathers in flight
#pragma unroll
            for (int q = 0; q < kAhead; q += 2) {
                const uint2 w0 = wq[q], w1 = wq[q + 1];
                wq[q] = pa[(size_t)min(al + kAhead + q, na - 1) * astride];
                wq[q + 1] = pa[(size_t)min(al + kAhead + q + 1, na - 1) * astride];
#ifdef CTPVAE_TUNE_STEP_SINGLE   // timing only: one angle's gathers in flight (fewer registers, a full wait per angle)
                if constexpr (NP == 2) {
                    taps(w0, al + q, va);
                    __builtin_amdgcn_sched_barrier(0);
                    add_all(va);
                    __builtin_amdgcn_sched_barrier(0);
                    taps(w1, al + q + 1, va);
                    __builtin_amdgcn_sched_barrier(0);
                    add_all(va);
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
#endif
                taps(w0, al + q, va);
                taps(w1, al + q + 1, vb);
                __builtin_amdgcn_sched_barrier(0);
                add_all(va);
                add_all(vb);
            }
        }
        for (int q = 0; al < na; ++al, ++q) {          // the chunk's last angles (wq holds their words in order)
            uint2 w = wq[0];
#pragma unroll
            for (int j = 1; j < kAhead; ++j) w = q == j ? wq[j] : w;
            taps(w, al, va);
            add_all(va);
        }
#endif
    }
    if (c < g.W) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int r = rg * kStepRows + k;
            if (r < g.H) {
#pragma unroll
                for (int n = 0; n < NS; ++n)
                    if (n < nlive) gimg[((size_t)(s + n) * g.H + r) * g.W + c] = kscale[n] * ((n & 1) ? acc[n >> 1][k].y : acc[n >> 1][k].x);
            }
        }
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

// Tiling of a slice that does not fit LDS: 64 x 96 tiles, whatever the batch (so a slice's sinogram does not depend
// on what it is batched with).  NS slices of the batch share a workgroup, interleaved per LDS pixel, so that one
// address computation (the kernel is VALU-bound on it) serves NS taps: 4 slices fill LDS (149 KiB with the zero
// border at the +-1 (mod 32) pitch, one 16-wave workgroup per CU), 2 or 1 leave room for 2 workgroups per CU.
// ntx == 0: no tiling needed / possible.
static int pick_tile_ns(int S, bool tie_fix)
{
    int ns = S >= 3 ? 4 : (S == 2 ? 2 : 1);
    {
        const int v = knob(kKnobTiledNs);
        if (v == 1 || v == 2 || v == 4) ns = v;
    }
    return tie_fix ? 1 : ns;   // the negative-tie fix of an unpadded canvas is not in the interleaved (asm) path
}
constexpr int kTileRowsMax = 128;
// Bilinear tiles (round 5, rotate_bilin.hip): 16-byte cells (four slices) at a pitch of 81 cells leave room for 96 + 3 rows.  A
// sample's 2 x 2 footprint belongs to the tile of its FLOOR tap (the tile stages a one-pixel halo below and to the right), so a
// tile's rays lie within half the diagonal of (tw + 1) x (th + 1) of its centre.
constexpr int kTileRowsMaxBilinear = 96;
TileSpec pick_tiles(int H, int W, int interp)
{
    TileSpec ts{};
    if (interp != CTPVAE_NEAREST) {
        if (bilin_fwd_whole_geometry(H, W) && knob(kKnobTiledForce) != 1) return ts;
        ts.tw = std::min(W, 64);
        ts.th = ceil_div(H, ceil_div(H, kTileRowsMaxBilinear));
        ts.ntx = ceil_div(W, ts.tw);
        ts.nty = ceil_div(H, ts.th);
        const float diag = sqrtf((float)((ts.tw + 1) * (ts.tw + 1) + (ts.th + 1) * (ts.th + 1)));
        ts.radius = 0.5f * diag + 3.0f;
        ts.span = (int)ceilf(2.0f * ts.radius) + 2;
        ts.nb = (ts.span + 63) / 64 * 64;
        return ts;
    }
    const int wb = W + 2;
    const size_t whole = (size_t)(H + 2) * std::max(pitch_for(wb, true), pitch_for(wb, false)) * sizeof(float);
    if (whole <= (size_t)kMaxLdsBytes && knob(kKnobTiledForce) != 1) return ts;   // (TILED_FORCE = 1: timing, tiles for slices that fit)
    ts.tw = std::min(W, 64);
    // EQUAL rows of tiles, as tall as four interleaved slices allow in LDS (128 rows = 143 KB with the step table), round 4.
    // Rounds 1-3 cut 96-row tiles: 512 rows were five rows of tiles and a 32-row remainder -- 768 workgroups of unequal cost,
    // three per CU, a launch as long as three FULL tiles.  Six rows of 86 are the same 768 workgroups at 0.9 of that (forward
    // + reduce 103.7 -> 97.4 us); four rows of 128 are 512 workgroups, two per CU: a fill, a third of the per-ray set-up and a
    // third of the partial sums less (91.9 us; profiles/r04_tile_heights.txt).  The shape is part of the result (association
    // of the sum): a function of (H, W) alone, reported by ctpvae_rotate_tile_shape.
    ts.th = ceil_div(H, ceil_div(H, kTileRowsMax));
#ifdef CTPVAE_TUNE_TILED_TH   // timing builds only (tools/build_variant.sh ... -DCTPVAE_TUNE_TILED_TH): another tile height CHANGES the
    {                         // association of the tiled sum, i.e. result bits -- not a switch of the product library (round 5)
        const int v = knob(kKnobTiledTh);
        if (v >= 16 && v <= kTileRowsMax) ts.th = std::min(H, v);
    }
#endif
    ts.ntx = ceil_div(W, ts.tw);
    ts.nty = ceil_div(H, ts.th);
    const float diag = sqrtf((float)(ts.tw * ts.tw + ts.th * ts.th));
    ts.radius = 0.5f * diag + 3.0f;
    ts.span = (int)ceilf(2.0f * ts.radius) + 2;
    ts.nb = (ts.span + 63) / 64 * 64;
    return ts;
}
static size_t tile_lds_bytes(const TileSpec &ts, int ns)
{
    const int wb = ts.tw + 2;
    return (size_t)(ts.th + 2) * std::max(pitch_for(wb, true), pitch_for(wb, false)) * sizeof(float) * ns;
}


// ---- tiled forward through COMPACT tile plans (round 3) --------------------------------------------------------------------
// The tiled kernel above is VALU-bound on TensorFlow's index arithmetic (23 ops per two rows of four slices).
// A tile is an ordinary slice at its own (py, px) of the canvas, so the compact plan of rotate_cplan.hip -- first tap + 2 bits
// per row, decoded through a step table in LDS -- applies tile by tile: a plan section per (tile, angle, ray slot) written once
// per geometry by rotate_tplan_kernel with the reference arithmetic, walked by rotate_fwd_tile_compact_kernel with
// cplan_walk.h's cwalk.  Same taps, same row order inside a tile, same slot layout of the partial sums => the SAME partial
// sums as the kernel above, bit for bit; rotate_tile_reduce_kernel is unchanged.
struct TLayout {
    int nt, nb, nbk, nq16, NQ, pitch, cells, maxT;
    int nu, NQP, maxTP;   // paired tasks (round 4): band pairs per (tile, angle), code chunks of a lane's two rays, tasks per (tile, class)
    long long off_cls, off_ng, off_ng16, off_tcount, off_tasks, off_start, off_codes, off_flag, bytes;
    long long off_ngu, off_ptcount, off_ptasks, off_pstart, off_pcodes;
};
static TLayout t_layout(const TileSpec &ts, int A)
{
    TLayout L;
    L.nt = ts.ntx * ts.nty;
    L.nb = ts.nb;
    L.nbk = ts.nb >> 6;
    L.nq16 = ts.nb >> 4;
    L.NQ = ceil_div((int)ceilf(sqrtf((float)(ts.tw * ts.tw + ts.th * ts.th))) + 4, kRowsPerChunk);   // rows of a ray inside a tile
    L.pitch = pitch_mod32_is_1(ts.tw + 1);
    L.cells = 1 + (ts.th + 2) * L.pitch + 1;
    L.maxT = (A * L.nq16 + 3) / 4 + 2;                                        // sorted tasks per (tile, class): two sign groups
    auto up = [](long long v) { return (v + 255) / 256 * 256; };
    L.off_cls = 0;                                                            // [A] class words (cplan_class_word)
    L.off_ng = up((long long)A * 4);                                          // [nt][A][nbk] row groups of a 64-slot block
    L.off_ng16 = up(L.off_ng + (long long)L.nt * A * L.nbk * 4);              // [nt][A][nq16] row groups of a 16-slot band
    L.off_tcount = up(L.off_ng16 + (long long)L.nt * A * L.nq16 * 4);         // [nt][2] sorted tasks of a (tile, class)
    L.off_tasks = up(L.off_tcount + (long long)L.nt * 2 * 4);                 // [nt][2][maxT] uint4: four band words
    L.off_start = up(L.off_tasks + (long long)L.nt * 2 * L.maxT * 16);        // [nt][A][nb] first cell | live << 31
    L.off_codes = up(L.off_start + (long long)L.nt * A * L.nb * 4);           // [nt][A][NQ][nb] uint4
    L.off_flag = L.off_codes + (long long)L.nt * A * L.NQ * L.nb * 16;
    // paired tasks: unit u of an angle = band u and band nq16 - 1 - u, walked back to back by the same 16 lanes
    L.nu = L.nq16 / 2;
    L.NQP = 2 * L.NQ;                                                         // both rays' whole groups: <= 2 x 8 NQ groups
    L.maxTP = (A * L.nu + 3) / 4 + 2;
    L.off_ngu = up(L.off_flag + 256);                                         // [nt][A][nu] row groups of a unit's longest lane
    L.off_ptcount = up(L.off_ngu + (long long)L.nt * A * L.nu * 4);           // [nt][2]
    L.off_ptasks = up(L.off_ptcount + (long long)L.nt * 2 * 4);               // [nt][2][maxTP] uint4: four unit words
    L.off_pstart = up(L.off_ptasks + (long long)L.nt * 2 * L.maxTP * 16);     // [nt][A][nu][16] uint2: ray A's / ray B's first cell
    L.off_pcodes = up(L.off_pstart + (long long)L.nt * A * L.nu * 16 * 8);    // [nt][A][nu][NQP][16] uint4
    L.bytes = L.off_pcodes + (long long)L.nt * A * L.nu * L.NQP * 16 * 16;
    // Built and measured (round 4, profiles/r04_tile_pairs.txt): the pairs cut the kernel's LDS instructions by 12 % and its
    // LDS-array cycles by 6 % (a hardware group whose lanes sit in two rays conflicts more) for 12 % more vector instructions
    // and 128 registers -- and the launch takes the same 88 us.  The sections above exist only while the knob TILED_PAIR = 1
    // is set (when the plan is built AND when it is used); the default plan ends behind its flag word.
    if (knob(kKnobTiledPair) != 1) L.bytes = L.off_flag + 256;
    return L;
}
static size_t t_lds_bytes(const TLayout &L, int A, int ns) { return (size_t)kLutBytes + (size_t)L.cells * 4 * ns + ((size_t)A + 2) * 4; }

// one wave per (64-slot block, angle, tile): the mirrored 32-slot runs of the tiled kernel's tasks.  Every ray's codes end with
// its step onto the zero border (then "stay"), whatever the rows its task walks: a ray may ride in a task longer than its own
// block's (the sorted tasks below).
__global__ __launch_bounds__(64) void rotate_tplan_kernel(RotGeom gfull, TileSpec ts, const float *__restrict__ T8, TLayout L,
                                                          char *__restrict__ plan)
{
    const int blk = blockIdx.x, a = blockIdx.y, t = blockIdx.z, lane = threadIdx.x;
    int y0, x0, h, w;
    tile_rect(gfull, ts, t, y0, x0, h, w);
    const PlanGeom g{h, w, gfull.PH, gfull.PW, gfull.py + y0, gfull.px + x0, gfull.A};
    const float *t6 = T8 + 8 * a;
    const float tile_cx = (float)g.px + 0.5f * (float)(g.W - 1), tile_cy = (float)g.py + 0.5f * (float)(g.H - 1);
    const int slot = lane < 32 ? blk * 32 + lane : L.nb - 32 * (blk + 1) + (lane - 32);
    const int j = tile_first_bin(t6, tile_cx, tile_cy, ts.radius) + slot;
    const bool valid = (unsigned)j < (unsigned)g.PW;
    int *cls = reinterpret_cast<int *>(plan + L.off_cls);
    int *ngt = reinterpret_cast<int *>(plan + L.off_ng);
    int *ng16 = reinterpret_cast<int *>(plan + L.off_ng16);
    unsigned *start = reinterpret_cast<unsigned *>(plan + L.off_start);
    uint4 *codes = reinterpret_cast<uint4 *>(plan + L.off_codes);
    if (t == 0 && blk == 0 && lane == 0) cls[a] = cplan_class_word(t6);
    const RayScan rs = cplan_scan_ray(g, t6, j, valid, L.pitch);
    int n16 = rs.n;   // the longest ray of this lane's 16-slot band (a quarter of the wave holds one band)
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) n16 = max(n16, __shfl_xor(n16, off, 64));
    const int ng = (wave_max_i(rs.n) + kRowsPerGroup - 1) / kRowsPerGroup;
    const size_t ta = (size_t)t * gfull.A + a;
    if (lane == 0) ngt[ta * L.nbk + blk] = ng;
    if ((lane & 15) == 0) ng16[ta * L.nq16 + (slot >> 4)] = (n16 + kRowsPerGroup - 1) / kRowsPerGroup;
    // stored (and read back by the reduce pass) only inside the span; a ray past it that touches the tile would contradict
    // the span's derivation: the plan is then flagged unusable
    start[ta * L.nb + slot] = (unsigned)rs.start | (valid && slot < ts.span ? 0x80000000u : 0u);
    bool bad = cplan_encode_ray(g, t6, j, rs, kRowsPerChunk * L.NQ, L.pitch, L.NQ, codes + ta * L.NQ * L.nb + slot, (size_t)L.nb);
    bad = bad || kRowsPerGroup * ng > kRowsPerChunk * L.NQ || ng > 127 || (slot >= ts.span && rs.n > 0);
    if (__any(bad) && lane == 0) atomicOr(reinterpret_cast<int *>(plan + L.off_flag), 1);
}

// PAIRED TASKS (round 4).  Even with sorted bands a third of the gathered rows add zeros (1.30 x the rays' own rows,
// tools/sim_tile_tasks.py): the 16 rays of a band on a trapezoid's flank differ by up to 22 rows and the hardware group walks
// the longest.  The flank on the other side of the tile falls as this one rises, so a lane walks slot s of band u and THEN slot
// s of band nq16 - 1 - u (the mirror image of slot 15 - s): the two lengths add up to about the same for all 16 lanes (1.16 x).
// Per lane: ray A's first cell and whole groups gA, ray B's first cell; a code stream = A's codes (2 gA bytes), then B's.
// One wave per (64-slot block, angle, tile) as rotate_tplan_kernel: lanes 0..31 hold the A rays of two units, lanes 32..63
// their B rays (lane ^ 48 is the partner).
__global__ __launch_bounds__(64) void rotate_tplan_pairs_kernel(RotGeom gfull, TileSpec ts, const float *__restrict__ T8, TLayout L,
                                                                char *__restrict__ plan)
{
    const int blk = blockIdx.x, a = blockIdx.y, t = blockIdx.z, lane = threadIdx.x;
    int y0, x0, h, w;
    tile_rect(gfull, ts, t, y0, x0, h, w);
    const PlanGeom g{h, w, gfull.PH, gfull.PW, gfull.py + y0, gfull.px + x0, gfull.A};
    const float *t6 = T8 + 8 * a;
    const float tile_cx = (float)g.px + 0.5f * (float)(g.W - 1), tile_cy = (float)g.py + 0.5f * (float)(g.H - 1);
    const int slot = lane < 32 ? blk * 32 + lane : L.nb - 32 * (blk + 1) + (lane - 32);
    const int j = tile_first_bin(t6, tile_cx, tile_cy, ts.radius) + slot;
    const bool valid = (unsigned)j < (unsigned)g.PW;
    const RayScan rs = cplan_scan_ray(g, t6, j, valid, L.pitch);
    const bool isA = lane < 32;
    const int ul = isA ? lane : (lane ^ 48), u = 2 * blk + (ul >> 4), k = ul & 15;
    const int gown = (rs.n + kRowsPerGroup - 1) / kRowsPerGroup, gother = __shfl_xor(gown, 48, 64);
    const int gA = isA ? gown : gother, gB = isA ? gother : gown;
    extern __shared__ unsigned char stream[];   // [32 unit lanes][NQP * 16] code bytes
    const int sbytes = L.NQP * 16;
    for (int b = lane; b < 32 * sbytes / 4; b += 64) reinterpret_cast<unsigned *>(stream)[b] = 0u;
    __syncthreads();
    bool bad = 2 * (gA + gB) > sbytes || gA + gB > 127;
    if (!bad) {
        unsigned char *mine = stream + ul * sbytes + (isA ? 0 : 2 * gA);
        bad = cplan_encode_ray_bytes(g, t6, j, rs, L.pitch, [&](int b, unsigned v) { mine[b] = (unsigned char)v; });
    }
    __syncthreads();
    const size_t unit = ((size_t)t * gfull.A + a) * L.nu + u;
    unsigned *pstart = reinterpret_cast<unsigned *>(plan + L.off_pstart) + (unit * 16 + k) * 2;
    if (isA)
        pstart[0] = (unsigned)rs.start | ((unsigned)gA << 16) | (valid && slot < ts.span ? 0x80000000u : 0u);
    else
        pstart[1] = (unsigned)rs.start | (valid && slot < ts.span ? 0x80000000u : 0u);
    int tot = gA + gB;   // the same on both partner lanes
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) tot = max(tot, __shfl_xor(tot, off, 64));
    if (isA && k == 0) reinterpret_cast<int *>(plan + L.off_ngu)[unit] = tot;
    if (isA) {
        uint4 *out = reinterpret_cast<uint4 *>(plan + L.off_pcodes) + unit * L.NQP * 16 + k;
        for (int q = 0; q < L.NQP; ++q) out[(size_t)q * 16] = *reinterpret_cast<const uint4 *>(stream + ul * sbytes + 16 * q);
    }
    if (__any(bad) && lane == 0) atomicOr(reinterpret_cast<int *>(plan + L.off_flag), 1);   // (exactly when rotate_tplan_kernel flags)
}

// SORTED TASKS (round 3).  A wave walks its 64 ray slots for as many rows as its longest ray; with (angle, 64-slot block) tasks
// the rays of a 64 x 96 tile cut obliquely have triangular length profiles and a third of the gathers add zeros (1.54 x the
// rows of the rays themselves, counted from the geometry at 512 x 512, 90 angles).  A ds_read_b128 is served in four hardware
// groups of 16 lanes, and a group wants 16 consecutive rays (see the kernel below) -- but nothing ties the four groups of a
// wave to one angle.  So the unit becomes a 16-slot BAND of one angle, the bands of a (tile, mirror class) are sorted by their
// length in row groups (descending, the sign of the step first: a wave's walk adds or subtracts its table offsets), and a
// task is four consecutive bands of that order: 1.24 x.  Which rays ride together changes nothing in any ray's sum.
// Task word: angle | band << 16 | the band's row groups << 20 | sigma < 0 << 28 | 1 << 31 (0: an empty quarter).
// One wave per (tile, class); the sort is a stable counting sort over keys (sign, 127 - groups).
// (round 4: the same sort over the PAIRED units -- nq units per angle, lengths at off_len, lists at off_tcount / off_tasks)
// nq_used: units >= nq_used of an angle hold no slot inside the tile's span (TileSpec::span) -- nothing to walk, nothing stored,
// no task.
__global__ __launch_bounds__(64) void rotate_tplan_tasks_kernel(int A, TLayout L, char *__restrict__ plan, int nq, long long off_len,
                                                                long long off_tcount, long long off_tasks, int maxT, int nq_used)
{
    const int t = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
    const int *cls = reinterpret_cast<const int *>(plan + L.off_cls);
    const int *ng16 = reinterpret_cast<const int *>(plan + off_len) + (size_t)t * A * nq;
    unsigned *tasks = reinterpret_cast<unsigned *>(plan + off_tasks) + ((size_t)t * 2 + c) * maxT * 4;
    int *tcount = reinterpret_cast<int *>(plan + off_tcount) + t * 2 + c;
    __shared__ int hist[256], base[256];
    for (int k = lane; k < 256; k += 64) hist[k] = 0;
    __syncthreads();
    const int U = A * nq;
    auto key_of = [&](int u, bool &in) -> int {
        in = false;
        if (u >= U) return 0;
        const int a = u / nq, w = cls[a];
        in = (w & 1) == c && u - a * nq < nq_used;
        return ((w >> 1) & 1) * 128 + (127 - min(ng16[u], 127));
    };
    for (int u0 = 0; u0 < U; u0 += 64) {
        bool in;
        const int k = key_of(u0 + lane, in);
        if (in) atomicAdd(&hist[k], 1);
    }
    __syncthreads();
    if (lane == 0) {   // positions: each sign group padded to whole tasks
        int pos = 0;
        for (int sg = 0; sg < 2; ++sg) {
            for (int k = 0; k < 128; ++k) {
                base[sg * 128 + k] = pos;
                pos += hist[sg * 128 + k];
            }
            pos = (pos + 3) & ~3;
        }
        *tcount = pos >> 2;
    }
    __syncthreads();
    for (int u0 = 0; u0 < U; u0 += 64) {   // stable: units in ascending order, lanes of one key ranked by lane
        bool in;
        const int u = u0 + lane, k = key_of(u, in);
        unsigned long long todo = __ballot(in);
        int pos = -1;
        while (todo) {
            const int l0 = __ffsll((long long)todo) - 1;
            const int k0 = __shfl(k, l0, 64);
            const unsigned long long same = __ballot(in && k == k0) & todo;
            const int b = base[k0];
            if (in && k == k0) pos = b + (int)__popcll(same & ((1ull << lane) - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (lane == l0) base[k0] = b + (int)__popcll(same);
            __builtin_amdgcn_wave_barrier();
            todo &= ~same;
        }
        if (in) {   // (a task's row groups are read from its first, longest band's word)
            const int a = u / nq, band = u - a * nq;
            tasks[pos] = (unsigned)a | ((unsigned)band << 16) | ((unsigned)(127 - (k & 127)) << 20) | ((unsigned)(k >> 7) << 28) |
                         0x80000000u;
        }
    }
}

// Workgroup = (tile of NS slices, mirror class, task group), launched like the tiled kernel above: blockIdx.x = 2 * group +
// class, blockIdx.y = slice group * tiles + tile.  Stages the tile with its zero border (class 0 column-mirrored), then its
// waves take tasks from an LDS counter, long ones first, each prepared while the previous one is walked.  SORTED: a task is
// four 16-slot bands of the plan's sorted list (rotate_tplan_tasks_kernel), one per quarter of the wave; otherwise (knob
// TILED_SORT = 0, kept for measurement) an (angle, 64-slot block) pair.
#ifndef CTPVAE_TILE_PERMUTE
#define CTPVAE_TILE_PERMUTE 1
#endif
constexpr bool kTilePermuteLanes = CTPVAE_TILE_PERMUTE != 0;
// MODE: 0 = (angle, 64-slot block) tasks (knob TILED_SORT = 0, kept for measurement); 1 = four sorted 16-slot bands
// (round 3, the default); 2 = four sorted band PAIRS, every lane walking two rays back to back (round 4; knob TILED_PAIR = 1,
// set while the plan is built and used: measured equal in time, see t_layout).
template <int NS, int MODE>
__global__ __launch_bounds__(1024) void rotate_fwd_tile_compact_kernel(const float *__restrict__ img, RotGeom gfull, TileSpec ts,
                                                                       TLayout L, const char *__restrict__ plan,
                                                                       float *__restrict__ partial, int xcd_order)
{
    constexpr bool SORTED = MODE != 0, PAIRED = MODE == 2;
    typedef typename SliceVec<NS>::type vec_t;
    extern __shared__ float lds[];
    float *image = lds + kLutBytes / 4;
    // Which (tile, slice group, class) a workgroup takes.  Workgroups are dispatched in the order of their linear number, round
    // robin over the 8 XCDs; round 3 numbered them slice group-major, so the workgroups that walk one tile's plan section (one
    // per slice group) ran on one XCD but ROUNDS apart, and every one of them pulled the section (0.6 MB per tile) through the
    // fabric again: 207 MB of traffic per launch at 32 x 512 x 512 x 90 angles, five times the algorithmic bytes.  Round 4
    // (`xcd_order`: one task group, tiles in whole octets): linear number = octet of tiles | slice group | class | tile % 8 --
    // the 2 x groups workgroups of a tile sit on ONE XCD at the SAME time and share its plan section in that XCD's L2.
    const int nt = L.nt, A = gfull.A, G = gridDim.x >> 1;
    int v = blockIdx.y, cls = blockIdx.x & 1;
    const int gi = blockIdx.x >> 1;
    if (xcd_order) {
        const int groups = gridDim.y / nt, lin = blockIdx.y * 2 + blockIdx.x;   // (G == 1)
        const int per_octet = 16 * groups, o = lin / per_octet, r = lin - o * per_octet;
        cls = (r >> 3) & 1;
        v = (r >> 4) * nt + o * 8 + (r & 7);
    }
    const int t = v % nt, s = (v / nt) * NS;
    int y0, x0, h, w;
    tile_rect(gfull, ts, t, y0, x0, h, w);
    const float *im = img + ((size_t)s * gfull.H + y0) * gfull.W + x0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int *clsw = reinterpret_cast<const int *>(plan + L.off_cls);
    const int *ngt = reinterpret_cast<const int *>(plan + L.off_ng) + (size_t)t * A * L.nbk;
    const uint4 *tasks = PAIRED ? reinterpret_cast<const uint4 *>(plan + L.off_ptasks) + ((size_t)t * 2 + cls) * L.maxTP
                                : reinterpret_cast<const uint4 *>(plan + L.off_tasks) + ((size_t)t * 2 + cls) * L.maxT;
    const unsigned *start = reinterpret_cast<const unsigned *>(plan + L.off_start) + (size_t)t * A * L.nb;
    const uint4 *codes = reinterpret_cast<const uint4 *>(plan + L.off_codes) + (size_t)t * A * L.NQ * L.nb;
    const uint2 *pstart = reinterpret_cast<const uint2 *>(plan + L.off_pstart) + (size_t)t * A * L.nu * 16;
    const uint4 *pcodes = reinterpret_cast<const uint4 *>(plan + L.off_pcodes) + (size_t)t * A * L.nu * L.NQP * 16;
    // behind the image: the ascending list of this class's angles ([0] = their count; unsorted tasks only), then the task counter
    int *cls_list = reinterpret_cast<int *>(image + (size_t)L.cells * NS);
    if (threadIdx.x < 64) {
        int n = 0;
        if constexpr (!SORTED)
            for (int a0 = 0; a0 < A; a0 += 64) {
                const bool in_cls = a0 + lane < A && (clsw[min(a0 + lane, A - 1)] & 1) == cls;
                const unsigned long long m = __ballot(in_cls);
                if (in_cls) cls_list[1 + n + __popcll(m & ((1ull << lane) - 1ull))] = a0 + lane;
                n += __popcll(m);
            }
        if (lane == 0) {
            cls_list[0] = n;
            cls_list[1 + A] = nwaves;   // the task counter: every wave's first task is its own number
        }
    }
    // which slot of a 32-slot run a lane walks: a ds_read_b128 is served in four groups of 16 lanes that are NOT runs of
    // consecutive lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, and the same + 32), and only lanes of one group conflict:
    // giving a group 16 CONSECUTIVE rays keeps its taps within 15 (|cos| + |sin|) <= 21 sixteen-byte slots (a 2-way wrap at
    // worst) instead of the 38 that lanes 0..27 of a linear assignment span (3-way)
    int li = lane & 31;
    if constexpr (NS == 4 && kTilePermuteLanes)
        li = li < 4 ? li : li < 12 ? li + 12 : li < 16 ? li - 8 : li < 20 ? li + 8 : li < 28 ? li - 12 : li;
    const int quarter = (lane >> 5) * 2 + (li >> 4);   // SORTED: the hardware group = the task's band this lane rides in
    const int ntask_sorted =
        SORTED ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const int *>(plan + (PAIRED ? L.off_ptcount : L.off_tcount))[t * 2 + cls]) : 0;
    cplan_init_lut<NS>(lds, L.pitch);
    cplan_zero_border<NS>(image, h, w, L.pitch);
    float *core = image + (size_t)(1 + L.pitch) * NS;
#ifndef CTPVAE_TUNE_TILE_NOFILL   // (timing builds only: the walks over whatever LDS holds)
    {
        const float *srcs[NS];
#pragma unroll
        for (int n = 0; n < NS; ++n) srcs[n] = im + (size_t)(min(s + n, gfull.S - 1) - s) * gfull.H * gfull.W;
        stage_unit<NS>(core, srcs, h, w, gfull.W, L.pitch, cls == 0, lane, wave, nwaves);
    }
#endif
    __syncthreads();
    const int ncls = __builtin_amdgcn_readfirstlane(cls_list[0]);
    int *next_task = cls_list + 1 + A;
    const int ntask = SORTED ? ntask_sorted : ncls * L.nbk;
    const size_t st = PAIRED ? (size_t)16 : (size_t)L.nb;
    const int NQW = PAIRED ? L.NQP : L.NQ;   // code chunks of a lane's stream
    struct Task {
        bool valid, neg, live;
        int ray, ng, adr;   // ray = angle * nb + slot (per lane)
        const uint4 *p;   // the lane's code chunk 2 (chunk 1 = p[-st] is fetched when the walk starts: four registers less per waiting task)
        uint4 c0;
        // PAIRED: `adr` holds both rays' first cells (A | B << 16) until the walk starts; pk = (groups of ray A) | (slot distance
        // to ray B) << 8 | ray B live << 31 -- packed: a prepared task waits in registers while the current one is walked
        unsigned pk;
    };
    auto prepare = [&](int m) -> Task {
        Task q;
        q.valid = m < ntask;
        q.neg = q.live = false;
        q.ray = q.ng = 0;
        q.adr = PAIRED ? 0 : kLutBytes;   // (PAIRED: cell 0 = the guard cell, a zero)
        q.pk = 127u;                      // never switches
        q.p = (PAIRED ? pcodes : codes) + 2 * st;   // (idle lanes read some ray's chunks; their codes stay zero)
        q.c0 = uint4{0u, 0u, 0u, 0u};
        if (q.valid) {   // wave-uniform
            int a, slot;
            bool mine = true;
            if constexpr (SORTED) {
                const uint4 d = tasks[m];   // sorted longest first
                const unsigned wq = quarter == 0 ? d.x : quarter == 1 ? d.y : quarter == 2 ? d.z : d.w;
                mine = (wq >> 31) != 0;   // the last task of a sign group may have empty quarters
                a = wq & 0xffffu;
                slot = (int)((wq >> 16) & 15u) * 16 + (li & 15);   // (PAIRED: unit * 16 + lane of the unit = ray A's slot)
                q.ng = __builtin_amdgcn_readfirstlane((d.x >> 20) & 127u);
                q.neg = __builtin_amdgcn_readfirstlane((d.x >> 28) & 1u) != 0;
            } else {
                // m counts 64-slot blocks from the innermost (longest rays) outwards, the class's angles within a block
                const int bi = m / ncls, ai = m - bi * ncls, blk = L.nbk - 1 - bi;
                a = __builtin_amdgcn_readfirstlane(cls_list[1 + ai]);
                q.neg = (clsw[a] >> 1) != 0;
                q.ng = __builtin_amdgcn_readfirstlane(ngt[a * L.nbk + blk]);
                slot = lane < 32 ? blk * 32 + li : L.nb - 32 * (blk + 1) + li;
            }
            if (mine) {
                q.ray = a * L.nb + slot;
                q.pk |= 0x40000000u;   // this lane holds a ray: its code chunk 1 exists
                if constexpr (PAIRED) {
                    const int u = slot >> 4, ul = (a * L.nu + u) * 16 + (slot & 15);
                    const uint2 sw = pstart[ul];
                    const unsigned gA = (sw.x >> 16) & 127u;
                    q.live = (sw.x >> 31) != 0;
                    q.adr = (int)((sw.x & 0xffffu) | (sw.y << 16));
                    q.pk = gA | ((unsigned)(16 * (L.nq16 - 1 - 2 * u)) << 8) | (sw.y & 0x80000000u) | 0x40000000u;
                    const uint4 *p = pcodes + (size_t)(a * L.nu + u) * L.NQP * 16 + (slot & 15);
                    q.c0 = p[0];
                    q.p = p + 2 * st;
                } else {
                    const unsigned sw = start[q.ray];
                    q.live = (sw >> 31) != 0;
                    q.adr = kLutBytes + (int)(sw & 0x7fffffffu) * (4 * NS);
                    const uint4 *p = codes + (size_t)a * L.NQ * L.nb + slot;
                    q.c0 = p[0];
                    q.p = p + 2 * st;
                }
            }
        }
        return q;
    };
    Task cur = prepare(wave * G + gi);
    while (cur.valid) {
        int m = 0;
        if (lane == 0) m = atomicAdd(next_task, 1);
        const Task nxt = prepare(__builtin_amdgcn_readfirstlane(m) * G + gi);
        vec_t acc = 0.0f;
        [[maybe_unused]] vec_t accA = 0.0f;
#ifdef CTPVAE_TUNE_TILE_NOWALK   // timing builds only (tools/): what the kernel costs without its walks
        const int ng = 0;
#else
        const int ng = __builtin_amdgcn_readfirstlane(cur.ng);
#endif
        const bool neg = __builtin_amdgcn_readfirstlane((int)cur.neg) != 0;
        [[maybe_unused]] const int gsw = (int)(cur.pk & 127u);
        uint4 c1 = uint4{0u, 0u, 0u, 0u};
        if (NQW > 1 && ng > 5 && (cur.pk & 0x40000000u)) c1 = cur.p[-(ptrdiff_t)st];   // wanted from the walk's sixth group on
        if (ng > 0) {
            if constexpr (PAIRED) {
                const int adrB = kLutBytes + (int)((unsigned)cur.adr >> 16) * (4 * NS);
                const int adrA = gsw ? kLutBytes + (int)((unsigned)cur.adr & 0xffffu) * (4 * NS) : adrB;   // no groups of its own: ray B first
                acc = neg ? cwalk<NS, true, true>(adrA, ng, (lane & 31) << 3, cur.c0, c1, cur.p, st, NQW, gsw, adrB, &accA)
                          : cwalk<NS, false, true>(adrA, ng, (lane & 31) << 3, cur.c0, c1, cur.p, st, NQW, gsw, adrB, &accA);
            } else
                acc = neg ? cwalk<NS, true>(cur.adr, ng, (lane & 31) << 3, cur.c0, c1, cur.p, st, NQW)
                          : cwalk<NS, false>(cur.adr, ng, (lane & 31) << 3, cur.c0, c1, cur.p, st, NQW);
        }
        const size_t nrays = (size_t)A * L.nb;
        if constexpr (PAIRED) {
            const bool switched = gsw < ng;          // (gsw = 0: ray A holds no rows, its sum is the zero accA starts from)
            const vec_t zero = 0.0f;
            if (cur.live) *reinterpret_cast<vec_t *>(partial + partial_index(s, nt, t, nrays, (size_t)cur.ray)) = switched ? accA : acc;
            if ((cur.pk >> 31) != 0)
                *reinterpret_cast<vec_t *>(partial + partial_index(s, nt, t, nrays, (size_t)cur.ray + ((cur.pk >> 8) & 0xffffu))) =
                    switched ? acc : zero;
        } else if (cur.live) {
            // (partial_index: one NS-wide store; slices past the batch hold copies of the last one, the workspace has room)
            *reinterpret_cast<vec_t *>(partial + partial_index(s, nt, t, nrays, (size_t)cur.ray)) = acc;
        }
        cur = nxt;
    }
}

int launch_tile_reduce(const float *workspace_dev, const RotGeom &g, const TileSpec &ts, const float *T8_dev, float *sino_dev,
                       const LogLikEpilogue &epi, ctpvae_stream_t stream)
{
    int rwaves = std::min(16, ceil_div(g.PW, 64));   // waves per workgroup: 64 bins each
    if (knob(kKnobReduceWaves) > 0) rwaves = std::min(16, knob(kKnobReduceWaves));
    const dim3 rgrid(ceil_div(g.PW, 64 * rwaves), g.A, ceil_div(g.S, kReduceSlices)), rblock(64 * rwaves);
    if (epi.part)
        hipLaunchKernelGGL(rotate_tile_reduce_kernel<2>, rgrid, rblock, 0, (hipStream_t)stream, workspace_dev, g, ts, T8_dev, sino_dev, epi);
    else if (epi.lp)
        hipLaunchKernelGGL(rotate_tile_reduce_kernel<1>, rgrid, rblock, 0, (hipStream_t)stream, workspace_dev, g, ts, T8_dev, sino_dev, epi);
    else
        hipLaunchKernelGGL(rotate_tile_reduce_kernel<0>, rgrid, rblock, 0, (hipStream_t)stream, workspace_dev, g, ts, T8_dev, sino_dev, epi);
    CTPVAE_LAUNCH_CHECK("rotate_tile_reduce_kernel");
    return CTPVAE_OK;
}

}  // namespace ctpvae

using namespace ctpvae;

// The direct kernels index slices with a grid dimension (<= 65535): the entry points below hand them a longer batch in
// chunks, back to back on the caller's stream (slices are independent: same results as one launch).

extern "C" {

static int rotate_fwd_one(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                          const float *T8_dev, int A, int interp, float *sino_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(img_dev && T8_dev && sino_dev, "rotate_fwd: null pointer");
    if (int rc = check_geom("rotate_fwd", S, H, W, PH, PW, py, px, A, interp)) return rc;
    const RotGeom g{S, H, W, PH, PW, py, px, A};

    // bilinear (round 5): slices interleaved per LDS cell behind one coordinate-and-weight computation (rotate_bilin.hip)
    if (interp == CTPVAE_BILINEAR && knob(kKnobForceGeneric) < 0 && knob(kKnobNoPlan) < 0 && bilin_fwd_whole_ok(H, W, A))
        return bilin_fwd_whole(img_dev, S, H, W, PH, PW, py, px, T8_dev, A, sino_dev, stream);
    // fast path: the zero-bordered slice must fit LDS (row pitch == +-1 mod 32, see the kernel)
    const int border = interp == CTPVAE_NEAREST ? 1 : 2;
    const int wb = W + 2 * border;
    const size_t fast_lds = (size_t)(H + 2 * border) * std::max(pitch_for(wb, true), pitch_for(wb, false)) * sizeof(float);
    if (fast_lds <= (size_t)kMaxLdsBytes && knob(kKnobForceGeneric) < 0) {
        // rays per workgroup: whole waves, <= 1024, about one workgroup per CU when the launch is small (every
        // workgroup re-stages its slice, and a wave needs its CU's LDS bandwidth more than it needs neighbours)
        const long long nrays = (long long)A * PW;
        const int wgs_per_cu = fast_lds * 2 <= (size_t)kMaxLdsBytes ? 2 : 1;
        long long rpb = (nrays * S + 256 * wgs_per_cu - 1) / (256 * wgs_per_cu);
        rpb = std::max<long long>(64, std::min<long long>(rpb, std::min<long long>(nrays, 1024)));
        rpb = (rpb + 63) / 64 * 64;
#ifdef CTPVAE_TUNE_STAMPS
        if (const char *e = getenv("CTPVAE_TUNE_RPB")) rpb = atoi(e);
#endif
        const int block = (int)rpb;
        const dim3 grid((unsigned)((nrays + rpb - 1) / rpb), S);
        const bool tie_fix = (px == 0 || py == 0);
        auto launch = [&](auto kernel) -> int {
            static std::atomic<unsigned long long> attr_set{0};   // per kernel instantiation: devices done
            CTPVAE_SET_MAX_LDS_ONCE(kernel, attr_set);
            hipLaunchKernelGGL(kernel, grid, dim3(block), fast_lds, (hipStream_t)stream, img_dev, g, TileSpec{}, T8_dev,
                               (int)rpb, sino_dev);
            CTPVAE_LAUNCH_CHECK("rotate_fwd_fast_kernel");
            return CTPVAE_OK;
        };
        if (interp == CTPVAE_NEAREST)
            return tie_fix ? launch(rotate_fwd_fast_kernel<CTPVAE_NEAREST, true, false>)
                           : launch(rotate_fwd_fast_kernel<CTPVAE_NEAREST, false, false>);
        return launch(rotate_fwd_fast_kernel<CTPVAE_BILINEAR, false, false>);
    }

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
        return use_lds ? launch(rotate_fwd_kernel<float, CTPVAE_NEAREST, true>, lds_bytes)
                       : launch(rotate_fwd_kernel<float, CTPVAE_NEAREST, false>, 0);
    return use_lds ? launch(rotate_fwd_kernel<float, CTPVAE_BILINEAR, true>, lds_bytes)
                   : launch(rotate_fwd_kernel<float, CTPVAE_BILINEAR, false>, 0);
}

int ctpvae_rotate_fwd_f64(const double *img_dev, int S, int H, int W, int PH, int PW, int py, int px, const float *T8_dev, int A,
                          int interp, double *sino_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(img_dev && T8_dev && sino_dev && S > 0 && H > 0 && W > 0 && A > 0 && PW > 0, "rotate_fwd_f64: null pointer or empty sizes");
    if (int rc = check_geom("rotate_fwd_f64", 1, H, W, PH, PW, py, px, A, interp)) return rc;
    const size_t lds_bytes = (size_t)H * (W + 1) * sizeof(double);
    const bool use_lds = lds_bytes <= (size_t)kMaxLdsBytes;
    return for_slice_chunks(S, max_slices_per_launch(), [&](int s0, int n) {
        const RotGeom g{n, H, W, PH, PW, py, px, A};
        const int apb = pick_angles_per_block(n, A, PW);
        const dim3 grid(ceil_div(A, apb), n), block(256);
        const double *im = img_dev + (size_t)s0 * H * W;
        double *so = sino_dev + (size_t)s0 * A * PW;
        auto launch = [&](auto kernel, size_t shmem) -> int {
            if (shmem > 64 * 1024)
                CTPVAE_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)stream, im, g, T8_dev, apb, so);
            CTPVAE_LAUNCH_CHECK("rotate_fwd_kernel<double>");
            return CTPVAE_OK;
        };
        if (interp == CTPVAE_NEAREST)
            return use_lds ? launch(rotate_fwd_kernel<double, CTPVAE_NEAREST, true>, lds_bytes)
                           : launch(rotate_fwd_kernel<double, CTPVAE_NEAREST, false>, 0);
        return use_lds ? launch(rotate_fwd_kernel<double, CTPVAE_BILINEAR, true>, lds_bytes)
                       : launch(rotate_fwd_kernel<double, CTPVAE_BILINEAR, false>, 0);
    });
}

int ctpvae_rotate_fwd_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                          const float *T8_dev, int A, int interp, float *sino_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(img_dev && T8_dev && sino_dev && S > 0 && H > 0 && W > 0 && A > 0 && PW > 0, "rotate_fwd: null pointer or empty sizes");
    return for_slice_chunks(S, max_slices_per_launch(), [&](int s0, int n) {
        return rotate_fwd_one(img_dev + (size_t)s0 * H * W, n, H, W, PH, PW, py, px, T8_dev, A, interp,
                              sino_dev + (size_t)s0 * A * PW, stream);
    });
}

int ctpvae_rotate_tile_shape(int H, int W, int interp, int *tile_h, int *tile_w)
{
    if (H <= 0 || W <= 0 || !tile_h || !tile_w) return fail(CTPVAE_EINVAL, "rotate_tile_shape: bad sizes / null pointer");
    const TileSpec ts = pick_tiles(H, W, interp);
    *tile_h = ts.ntx ? ts.th : 0;
    *tile_w = ts.ntx ? ts.tw : 0;
    return ts.ntx ? 1 : 0;
}

long long ctpvae_rotate_fwd_tiled_workspace_bytes(int S, int H, int W, int PH, int PW, int A, int interp)
{
    if (S <= 0 || H <= 0 || W <= 0 || PH < H || PW < W || A <= 0) return fail(CTPVAE_EINVAL, "rotate_fwd_tiled_workspace_bytes: bad sizes");
    const TileSpec ts = pick_tiles(H, W, interp);
    if (ts.ntx == 0 || knob(kKnobForceGeneric) >= 0) return 0;
    // (bilinear tiles keep their angles' tables in LDS: thousands of angles do not fit -- 0 = use ctpvae_rotate_fwd_f32, the
    // global-memory kernel, for such a call)
    if (interp != CTPVAE_NEAREST && !bilin_fwd_tiles_ok(ts, A)) return 0;
    const long long quads = (S + kPartialQuad - 1) / kPartialQuad;   // partial_index: whole slice quads
    return quads * kPartialQuad * ts.ntx * ts.nty * A * ts.nb * (long long)sizeof(float);
}

static int launch_fwd_tiled_one(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px, const float *T8_dev,
                               int A, void *workspace_dev, float *sino_dev, const LogLikEpilogue &epi, ctpvae_stream_t stream,
                               const void *tplan_dev);

// tile workgroups are indexed with a grid dimension too: at most 65535 / tiles slices per launch, the workspace reused by
// the chunks (they run one after the other on the stream)
static int launch_fwd_tiled(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px, const float *T8_dev,
                           int A, void *workspace_dev, float *sino_dev, const LogLikEpilogue &epi, ctpvae_stream_t stream,
                           const void *tplan_dev = nullptr)
{
    CTPVAE_REQUIRE(img_dev && T8_dev && workspace_dev && (sino_dev || epi.part) && S > 0 && H > 0 && W > 0 && A > 0 && PW > 0,
                   "rotate_fwd_tiled: null pointer or empty sizes");
    const TileSpec ts = pick_tiles(H, W, CTPVAE_NEAREST);
    const int nt = std::max(1, ts.ntx * ts.nty);
    const int chunk = std::max(4, std::min(max_slices_per_launch(), 65535 / nt) / 4 * 4);
    return for_slice_chunks(S, chunk, [&](int s0, int n) {
        LogLikEpilogue e = epi;
        if (e.lp || e.part) {
            e.mask += (size_t)s0 * A;
            e.meas += (size_t)s0 * A * PW;
            if (e.lp) e.lp += (size_t)s0 * A * PW;
            if (e.dlp) e.dlp += (size_t)s0 * A * PW;
            if (e.part) e.part += (size_t)s0 * A * ((PW + 63) >> 6);
            if (e.sum) e.sum += s0, e.arrive += s0;
        }
        return launch_fwd_tiled_one(img_dev + (size_t)s0 * H * W, n, H, W, PH, PW, py, px, T8_dev, A, workspace_dev,
                                    sino_dev ? sino_dev + (size_t)s0 * A * PW : nullptr, e, stream, tplan_dev);
    });
}

static int launch_fwd_tiled_one(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px, const float *T8_dev,
                               int A, void *workspace_dev, float *sino_dev, const LogLikEpilogue &epi, ctpvae_stream_t stream,
                               const void *tplan_dev)
{
    CTPVAE_REQUIRE(img_dev && T8_dev && workspace_dev && (sino_dev || epi.part), "rotate_fwd_tiled: null pointer");
    if (int rc = check_geom("rotate_fwd_tiled", S, H, W, PH, PW, py, px, A, CTPVAE_NEAREST)) return rc;
    const RotGeom g{S, H, W, PH, PW, py, px, A};
    const bool tie_fix = (px == 0 || py == 0);
    int ns = pick_tile_ns(S, tie_fix);
    const TileSpec ts = pick_tiles(H, W, CTPVAE_NEAREST);
    CTPVAE_REQUIRE(ts.ntx > 0, "rotate_fwd_tiled: a %dx%d slice fits LDS whole; call ctpvae_rotate_fwd_f32", H, W);
    // the direct kernel's rows are 97 floats apart (two bank classes of one buffer; the compact plans' are 65): four slices of a
    // 128-row tile do not fit its LDS, two do -- same tiles, same partial sums
    while (!tplan_dev && ns > 1 && tile_lds_bytes(ts, ns) > (size_t)kMaxLdsBytes) ns >>= 1;
    CTPVAE_REQUIRE(tplan_dev || tile_lds_bytes(ts, ns) <= (size_t)kMaxLdsBytes, "rotate_fwd_tiled: a %d-row tile does not fit LDS", ts.th);
    const int nt = ts.ntx * ts.nty, groups = ceil_div(S, ns);
    CTPVAE_REQUIRE((long long)groups * nt <= 65535, "rotate_fwd_tiled: at most 65535 tiles per call (got %lld)", (long long)groups * nt);
    const size_t lds_bytes = tile_lds_bytes(ts, ns);
    // a copy of the transform rows behind the tile, when it fits
    const size_t t8_need = (size_t)A * 8 * sizeof(float) + ((size_t)A + 2) * sizeof(int);   // + class list + task counter
    const size_t t8_bytes = lds_bytes + t8_need <= (size_t)kMaxLdsBytes ? t8_need : 0;
    // Task groups per bank class: as many as keep tiles x 2 classes x groups within ONE round of the 256 CUs (rounding
    // up instead put 264 workgroups on the chip, a second round for 8 of them: 20.9 vs 16.6 us at 4 x 256 x 256, 90
    // angles) and leave every workgroup at least 4 waves of tasks (2-wave workgroups stage their tile too slowly:
    // 20.6 vs 13.3 us at 20 angles) -- tools/sweep_tiled.py.
    const int tasks = A * (ts.nb / 64);
    int G = std::max(1, 256 / (2 * groups * nt));
    G = std::min(G, std::max(1, tasks / 8));
    if (knob(kKnobTiledG) > 0) G = knob(kKnobTiledG);
    int waves = std::min(16, std::max(1, ceil_div(tasks, 2 * G)));
    if (knob(kKnobTiledWaves) > 0) waves = std::min(16, knob(kKnobTiledWaves));
    auto launch = [&](auto kernel) -> int {
        static std::atomic<unsigned long long> attr_set{0};   // per kernel instantiation: devices done
        CTPVAE_SET_MAX_LDS_ONCE(kernel, attr_set);
        hipLaunchKernelGGL(kernel, dim3(2 * G, groups * nt), dim3(64 * waves), lds_bytes + t8_bytes, (hipStream_t)stream,
                           img_dev, g, ts, T8_dev, t8_bytes ? (int)(lds_bytes / sizeof(float)) : 0, (float *)workspace_dev);
        CTPVAE_LAUNCH_CHECK("rotate_fwd_fast_kernel (tiled)");
        return CTPVAE_OK;
    };
    // compact tile plans (ctpvae_rotate_tplan_build_f32): the same partial sums without per-sample index arithmetic
    auto launch_compact = [&](auto kernel) -> int {
        static std::atomic<unsigned long long> attr_set{0}, abs_ok{0};
        CTPVAE_REQUIRE_NO_STATIC_LDS(kernel, "rotate_fwd_tile_compact_kernel", abs_ok);   // the step table sits at LDS address 0
        CTPVAE_SET_MAX_LDS_ONCE(kernel, attr_set);
        const TLayout TL = t_layout(ts, A);
        const int xcd_order = (G == 1 && nt % 8 == 0 && knob(kKnobTiledXcd) != 0) ? 1 : 0;
        hipLaunchKernelGGL(kernel, dim3(2 * G, groups * nt), dim3(64 * waves), t_lds_bytes(TL, A, ns), (hipStream_t)stream, img_dev, g,
                           ts, TL, (const char *)tplan_dev, (float *)workspace_dev, xcd_order);
        CTPVAE_LAUNCH_CHECK("rotate_fwd_tile_compact_kernel");
        return CTPVAE_OK;
    };
    int rc;
    if (tplan_dev && !tie_fix && knob(kKnobTiledSort) != 0 && knob(kKnobTiledPair) == 1)   // tasks = four sorted band pairs
        rc = ns == 4 ? launch_compact(rotate_fwd_tile_compact_kernel<4, 2>)
                     : (ns == 2 ? launch_compact(rotate_fwd_tile_compact_kernel<2, 2>)
                                : launch_compact(rotate_fwd_tile_compact_kernel<1, 2>));
    else if (tplan_dev && !tie_fix && knob(kKnobTiledSort) != 0)   // tasks = four sorted 16-slot bands (rotate_tplan_tasks_kernel)
        rc = ns == 4 ? launch_compact(rotate_fwd_tile_compact_kernel<4, 1>)
                     : (ns == 2 ? launch_compact(rotate_fwd_tile_compact_kernel<2, 1>)
                                : launch_compact(rotate_fwd_tile_compact_kernel<1, 1>));
    else if (tplan_dev && !tie_fix)
        rc = ns == 4 ? launch_compact(rotate_fwd_tile_compact_kernel<4, 0>)
                     : (ns == 2 ? launch_compact(rotate_fwd_tile_compact_kernel<2, 0>)
                                : launch_compact(rotate_fwd_tile_compact_kernel<1, 0>));
    else if (tie_fix)
        rc = launch(rotate_fwd_fast_kernel<CTPVAE_NEAREST, true, true, 1>);
    else if (ns == 4)
        rc = launch(rotate_fwd_fast_kernel<CTPVAE_NEAREST, false, true, 4>);
    else if (ns == 2)
        rc = launch(rotate_fwd_fast_kernel<CTPVAE_NEAREST, false, true, 2>);
    else
        rc = launch(rotate_fwd_fast_kernel<CTPVAE_NEAREST, false, true, 1>);
    if (rc) return rc;
    return launch_tile_reduce((const float *)workspace_dev, g, ts, T8_dev, sino_dev, epi, stream);
}

int ctpvae_rotate_fwd_tiled_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                                const float *T8_dev, int A, void *workspace_dev, float *sino_dev, ctpvae_stream_t stream)
{
    return launch_fwd_tiled(img_dev, S, H, W, PH, PW, py, px, T8_dev, A, workspace_dev, sino_dev, LogLikEpilogue{}, stream);
}

int ctpvae_rotate_fwd_tiled_interp_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px, const float *T8_dev,
                                       int A, int interp, void *workspace_dev, float *sino_dev, ctpvae_stream_t stream)
{
    if (interp == CTPVAE_NEAREST)
        return ctpvae_rotate_fwd_tiled_f32(img_dev, S, H, W, PH, PW, py, px, T8_dev, A, workspace_dev, sino_dev, stream);
    CTPVAE_REQUIRE(img_dev && T8_dev && workspace_dev && sino_dev && S > 0 && H > 0 && W > 0 && A > 0 && PW > 0,
                   "rotate_fwd_tiled: null pointer or empty sizes");
    if (int rc = check_geom("rotate_fwd_tiled", 1, H, W, PH, PW, py, px, A, interp)) return rc;
    const TileSpec ts = pick_tiles(H, W, interp);
    CTPVAE_REQUIRE(ts.ntx > 0, "rotate_fwd_tiled: a %dx%d slice fits LDS whole; call ctpvae_rotate_fwd_f32", H, W);
    CTPVAE_REQUIRE(knob(kKnobForceGeneric) < 0, "rotate_fwd_tiled: not available with FORCE_GENERIC");
    CTPVAE_REQUIRE(bilin_fwd_tiles_ok(ts, A), "rotate_fwd_tiled: the tables of %d angles do not fit LDS beside a tile (the workspace "
                   "size reads 0 for such a call: use ctpvae_rotate_fwd_f32)", A);
    const int nt = ts.ntx * ts.nty;
    const int chunk = std::max(4, std::min(max_slices_per_launch(), 65535 / nt) / 4 * 4);
    return for_slice_chunks(S, chunk, [&](int s0, int n) {
        if (int rc = bilin_fwd_tiles(img_dev + (size_t)s0 * H * W, n, H, W, PH, PW, py, px, T8_dev, A, ts, (float *)workspace_dev, stream))
            return rc;
        const RotGeom g{n, H, W, PH, PW, py, px, A};
        return launch_tile_reduce((const float *)workspace_dev, g, ts, T8_dev, sino_dev + (size_t)s0 * A * PW, LogLikEpilogue{}, stream);
    });
}

int ctpvae_rotate_fwd_tiled_loglik_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                                       const float *T8_dev, int A, void *workspace_dev, const float *mask_dev,
                                       const float *meas_dev, const float *pnm_dev, float eps, float *sino_dev,
                                       float *lp_dev, float *dlp_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(mask_dev && meas_dev && pnm_dev && lp_dev, "rotate_fwd_tiled_loglik: null pointer");
    return launch_fwd_tiled(img_dev, S, H, W, PH, PW, py, px, T8_dev, A, workspace_dev, sino_dev,
                            LogLikEpilogue{mask_dev, meas_dev, pnm_dev, eps, lp_dev, dlp_dev}, stream);
}

// ---- compact tile plans -----------------------------------------------------------------------------------------------------
long long ctpvae_rotate_tplan_bytes(int H, int W, int PH, int PW, int A)
{
    if (H <= 0 || W <= 0 || PH < H || PW < W || A <= 0) return fail(CTPVAE_EINVAL, "rotate_tplan_bytes: bad sizes");
    const TileSpec ts = pick_tiles(H, W, CTPVAE_NEAREST);
    if (ts.ntx == 0 || knob(kKnobForceGeneric) >= 0 || knob(kKnobNoCompact) >= 0 || knob(kKnobNoPlan) >= 0) return 0;
    const TLayout L = t_layout(ts, A);
    if (t_lds_bytes(L, A, 4) > (size_t)kMaxLdsBytes || 12ll * L.pitch * 4 > 32767) return 0;
    return L.bytes;
}

int ctpvae_rotate_tplan_build_f32(const float *T8_dev, int A, int H, int W, int PH, int PW, int py, int px, void *tplan_dev,
                                  ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(T8_dev && tplan_dev, "rotate_tplan_build: null pointer");
    if (int rc = check_geom("rotate_tplan_build", 1, H, W, PH, PW, py, px, A, CTPVAE_NEAREST)) return rc;
    const TileSpec ts = pick_tiles(H, W, CTPVAE_NEAREST);
    CTPVAE_REQUIRE(ts.ntx > 0, "rotate_tplan_build: a %dx%d slice fits LDS whole (ctpvae_rotate_cplan_build_f32)", H, W);
    const TLayout L = t_layout(ts, A);
    CTPVAE_REQUIRE(A <= 65535 && L.nt <= 65535, "rotate_tplan_build: at most 65535 angles and tiles");
    const RotGeom g{1, H, W, PH, PW, py, px, A};
    CTPVAE_HIP(hipMemsetAsync((char *)tplan_dev + L.off_flag, 0, 256, (hipStream_t)stream));
    CTPVAE_HIP(hipMemsetAsync((char *)tplan_dev + L.off_tasks, 0, (size_t)L.nt * 2 * L.maxT * 16, (hipStream_t)stream));   // empty quarters
    hipLaunchKernelGGL(rotate_tplan_kernel, dim3(L.nbk, A, L.nt), dim3(64), 0, (hipStream_t)stream, g, ts, T8_dev, L, (char *)tplan_dev);
    CTPVAE_LAUNCH_CHECK("rotate_tplan_kernel");
    hipLaunchKernelGGL(rotate_tplan_tasks_kernel, dim3(L.nt, 2), dim3(64), 0, (hipStream_t)stream, A, L, (char *)tplan_dev, L.nq16,
                       L.off_ng16, L.off_tcount, L.off_tasks, L.maxT, ceil_div(ts.span, 16));
    CTPVAE_LAUNCH_CHECK("rotate_tplan_tasks_kernel");
    if (knob(kKnobTiledPair) != 1) return CTPVAE_OK;
    // paired units (round 4, knob TILED_PAIR = 1): lane streams, then the same sort over them
    CTPVAE_HIP(hipMemsetAsync((char *)tplan_dev + L.off_ptasks, 0, (size_t)L.nt * 2 * L.maxTP * 16, (hipStream_t)stream));
    hipLaunchKernelGGL(rotate_tplan_pairs_kernel, dim3(L.nbk, A, L.nt), dim3(64), (size_t)32 * L.NQP * 16, (hipStream_t)stream, g, ts, T8_dev, L, (char *)tplan_dev);
    CTPVAE_LAUNCH_CHECK("rotate_tplan_pairs_kernel");
    hipLaunchKernelGGL(rotate_tplan_tasks_kernel, dim3(L.nt, 2), dim3(64), 0, (hipStream_t)stream, A, L, (char *)tplan_dev, L.nu,
                       L.off_ngu, L.off_ptcount, L.off_ptasks, L.maxTP, L.nu);
    CTPVAE_LAUNCH_CHECK("rotate_tplan_tasks_kernel (pairs)");
    return CTPVAE_OK;
}

// 1 if some ray's steps do not fit the code: keep ctpvae_rotate_fwd_tiled_f32.  SYNCHRONISES the stream.
int ctpvae_rotate_tplan_overflowed(const void *tplan_dev, int H, int W, int PH, int PW, int A, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(tplan_dev && H > 0 && W > 0 && A > 0 && PH >= H && PW >= W, "rotate_tplan_overflowed: bad arguments");
    const TileSpec ts = pick_tiles(H, W, CTPVAE_NEAREST);
    CTPVAE_REQUIRE(ts.ntx > 0, "rotate_tplan_overflowed: not a tiled geometry");
    const TLayout L = t_layout(ts, A);
    int flag = 0;
    CTPVAE_HIP(hipMemcpyAsync(&flag, (const char *)tplan_dev + L.off_flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    CTPVAE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return flag ? 1 : 0;
}

int ctpvae_rotate_fwd_tiled_compact_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                                        const float *T8_dev, int A, const void *tplan_dev, void *workspace_dev,
                                        const float *mask_dev, const float *meas_dev, const float *pnm_dev, float eps,
                                        float *sino_dev, float *lp_dev, float *dlp_dev, float *lp_part_dev, float *lp_sum_dev,
                                        ctpvae_stream_t stream)
{
    const bool red = lp_sum_dev != nullptr, lik = lp_dev != nullptr || red;
    CTPVAE_REQUIRE(!lik || (mask_dev && meas_dev && pnm_dev), "rotate_fwd_tiled_compact: the likelihood epilogue needs mask, meas and pnm");
    CTPVAE_REQUIRE(lik || dlp_dev == nullptr, "rotate_fwd_tiled_compact: dlp without lp");
    CTPVAE_REQUIRE(!red || lp_part_dev, "rotate_fwd_tiled_compact: per-object sums need the partial-sum workspace");
    LogLikEpilogue epi = lik ? LogLikEpilogue{mask_dev, meas_dev, pnm_dev, eps, lp_dev, dlp_dev, 0, red ? lp_part_dev : nullptr}
                             : LogLikEpilogue{};
    const bool fold = red && knob(kKnobFoldSums) == 1;   // round 4 (measured negative, see ctpvae_rotate_fwd_compact_f32): the reduce pass's last workgroup per slice group adds the partials
    if (fold) {
        epi.sum = lp_sum_dev;
        epi.arrive = reinterpret_cast<unsigned *>(lp_part_dev + (size_t)S * A * ((PW + 63) >> 6));   // ctpvae_loglik_part_floats
    }
    if (int rc = launch_fwd_tiled(img_dev, S, H, W, PH, PW, py, px, T8_dev, A, workspace_dev, sino_dev, epi, stream, tplan_dev)) return rc;
    if (red && !fold) {   // a slice's partials (angle, 64-bin block) in ascending order: ctpvae_loglik_object_sums_f32, partition 1
        hipLaunchKernelGGL(loglik_sum_partials_kernel, dim3(S), dim3(64), 0, (hipStream_t)stream, lp_part_dev, S, A, (PW + 63) >> 6,
                           lp_sum_dev);
        CTPVAE_LAUNCH_CHECK("loglik_sum_partials_kernel");
    }
    return CTPVAE_OK;
}

int ctpvae_rotate_bwd_f32(const float *gsino_dev, int S, int A, int PH, int PW, const float *T8_dev, int interp,
                          int mode, int H, int W, int py, int px, float *gimg_dev, ctpvae_stream_t stream)
{
    return ctpvae_rotate_bwd_scaled_f32(gsino_dev, S, A, PH, PW, T8_dev, interp, mode, H, W, py, px, nullptr, 0, gimg_dev, stream);
}

static int rotate_bwd_one(const float *gsino_dev, int S, int A, int PH, int PW, const float *T8_dev, int interp,
                          int mode, int H, int W, int py, int px, const float *scale_dev, long long scale_stride,
                          float *gimg_dev, ctpvae_stream_t stream, const int *sel_dev = nullptr, int A_plan = 0);

int ctpvae_rotate_bwd_scaled_f32(const float *gsino_dev, int S, int A, int PH, int PW, const float *T8_dev, int interp,
                                 int mode, int H, int W, int py, int px, const float *scale_dev, long long scale_stride,
                                 float *gimg_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(gsino_dev && T8_dev && gimg_dev && S > 0 && H > 0 && W > 0 && A > 0 && PW > 0, "rotate_bwd: null pointer or empty sizes");
    return for_slice_chunks(S, max_slices_per_launch(), [&](int s0, int n) {
        return rotate_bwd_one(gsino_dev + (size_t)s0 * A * PW, n, A, PH, PW, T8_dev, interp, mode, H, W, py, px,
                              scale_dev ? scale_dev + (long long)s0 * scale_stride : nullptr, scale_stride,
                              gimg_dev + (size_t)s0 * H * W, stream);
    });
}

// ---- step plan of the segment backward (slices too large for the planned backward) ----
long long ctpvae_rotate_bwd_step_plan_bytes(int H, int W, int A)
{
    if (H <= 0 || W <= 0 || A <= 0) return fail(CTPVAE_EINVAL, "rotate_bwd_step_plan_bytes: bad sizes");
    return step_layout(H, W, A).bytes;
}

int ctpvae_rotate_bwd_step_plan_build_f32(const float *Tinv8_dev, int A, int H, int W, int PH, int PW, int py, int px, void *plan_dev,
                                          ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(Tinv8_dev && plan_dev, "rotate_bwd_step_plan_build: null pointer");
    if (int rc = check_geom("rotate_bwd_step_plan_build", 1, H, W, PH, PW, py, px, A, CTPVAE_NEAREST)) return rc;
    CTPVAE_REQUIRE(A <= 65535, "rotate_bwd_step_plan_build: at most 65535 angles (got %d)", A);
    const StepLayout L = step_layout(H, W, A);
    const RotGeom g{1, H, W, PH, PW, py, px, A};
    CTPVAE_HIP(hipMemsetAsync((char *)plan_dev + L.off_flag, 0, 256, (hipStream_t)stream));
    hipLaunchKernelGGL(rotate_bwd_step_plan_kernel, dim3(L.Wpad / 64, L.H8, A), dim3(64), 0, (hipStream_t)stream, g, Tinv8_dev, L,
                       (char *)plan_dev);
    CTPVAE_LAUNCH_CHECK("rotate_bwd_step_plan_kernel");
    return CTPVAE_OK;
}

int ctpvae_rotate_bwd_step_plan_overflowed(const void *plan_dev, int H, int W, int A, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(plan_dev && H > 0 && W > 0 && A > 0, "rotate_bwd_step_plan_overflowed: bad arguments");
    int flag = 0;
    CTPVAE_HIP(hipMemcpyAsync(&flag, (const char *)plan_dev + step_layout(H, W, A).off_flag, sizeof(int), hipMemcpyDeviceToHost,
                              (hipStream_t)stream));
    CTPVAE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return flag != 0;
}

int ctpvae_rotate_bwd_stepped_scaled_f32(const float *gsino_dev, int S, int A, int PH, int PW, const float *Tinv8_dev, int H, int W,
                                         int py, int px, const void *step_plan_dev, const float *scale_dev, long long scale_stride,
                                         float *gimg_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(gsino_dev && Tinv8_dev && gimg_dev && step_plan_dev && S > 0 && H > 0 && W > 0 && A > 0 && PW > 0,
                   "rotate_bwd_stepped: null pointer or empty sizes");
    if (int rc = check_geom("rotate_bwd_stepped", S, H, W, PH, PW, py, px, A, CTPVAE_NEAREST)) return rc;
    const StepLayout L = step_layout(H, W, A);
    // (>= 4: a MAX_SLICES knob of 1..3 must not make a chunk of 0 slices -- the chunk loop would never advance)
    return for_slice_chunks(S, std::max(4, std::min(65532, max_slices_per_launch() / 4 * 4)), [&](int s0, int n) {
        const float *gs = gsino_dev + (size_t)s0 * A * PW;
        const float *sc = scale_dev ? scale_dev + (long long)s0 * scale_stride : nullptr;
        float *gi = gimg_dev + (size_t)s0 * H * W;
        // the plan is for slice PAIRS (a lone or last odd slice rides as a half-empty pair) in 64 x 32 tiles, which small launches
        // trade for 64 x 16 ones -- those keep the direct kernel
        const int pairs = ceil_div(n, 2);
        // (SEG_PPT = 8 / 4: a developer's way to force the 64 x 32 stepped tiles / the direct kernel whatever the launch size)
        const long long tiles = (long long)ceil_div(W, 64) * ceil_div(H, kStepTileRows);
        const bool big = knob(kKnobSegPpt) >= 0 ? knob(kKnobSegPpt) == 8 : pairs * tiles >= 512;
        if (knob(kKnobForceGeneric) >= 0 || knob(kKnobNoPlan) >= 0 || !big)
            return rotate_bwd_one(gs, n, A, PH, PW, Tinv8_dev, CTPVAE_NEAREST, CTPVAE_BWD_TF_COMPAT, H, W, py, px, sc, scale_stride, gi,
                                  stream);
        const RotGeom g{n, H, W, PH, PW, py, px, A};
        // two pairs per workgroup (round 4) while that leaves every CU three workgroups (knob STEP_NS = 2 / 4 forces).  Measured
        // (tools/time_step_ns.py, two pairs against one): 1024 workgroups 47.1 vs 57.1 us (32 x 512 x 512 x 90 angles), 800: 88.9 vs
        // 99.0 (400 x 128 x 128 x 180); 512: 35.0 vs 33.3 (16 x 512 x 512), 400: 65.7 vs 62.6 (200 x 128 x 128 x 180)
        const int ns = knob(kKnobStepNs) == 2 ? 2 : (knob(kKnobStepNs) == 4 || (long long)ceil_div(n, 4) * tiles >= 768) ? 4 : 2;
        // angles per staged chunk (knob SEG_CHUNK; measured with warm clocks at 32 x 512 x 512 x 90 angles, tools/sweep_step_chunk.py:
        // 48 + 42 angles 56.5-57.0 us, 45 + 45 56.8, 3 x 30 57.6, 4 x 23 57.4, 5 x 18 62.3 -- flat down to chunks of 23)
        const int max_chunk = knob(kKnobSegChunk) > 0 ? std::min(64, knob(kKnobSegChunk)) : 48;   // (<= 64: one mask bit per angle)
        const int chunk_a = std::min(A, ns == 4 ? std::min(max_chunk, kStepChunk4) : max_chunk);
        const size_t cells = ns == 4 ? (size_t)kStepPairGap + (size_t)chunk_a * kSegPitch * kStepCell : (size_t)chunk_a * kSegPitch * kStepCell;
        size_t shmem = cells + (size_t)chunk_a * sizeof(int) + 16;
        if (knob(kKnobStepLdsKb) > 0) shmem = std::max(shmem, (size_t)knob(kKnobStepLdsKb) * 1024);   // timing: fewer workgroups per CU
        const dim3 grid(ceil_div(W, 64), ceil_div(H, kStepTileRows), ceil_div(n, ns)), block(256);
        if (ns == 4)
            hipLaunchKernelGGL(rotate_bwd_stepped_kernel<4>, grid, block, shmem, (hipStream_t)stream, gs, g, Tinv8_dev, chunk_a, L,
                               (const char *)step_plan_dev, SliceScale{sc, scale_stride}, gi);
        else
            hipLaunchKernelGGL(rotate_bwd_stepped_kernel<2>, grid, block, shmem, (hipStream_t)stream, gs, g, Tinv8_dev, chunk_a, L,
                               (const char *)step_plan_dev, SliceScale{sc, scale_stride}, gi);
        CTPVAE_LAUNCH_CHECK("rotate_bwd_stepped_kernel");
        return CTPVAE_OK;
    });
}

int ctpvae_rotate_bwd_sel_scaled_f32(const float *gsino_dev, int S, int A_plan, int PH, int PW, const float *Tinv8_dev,
                                     const int *angle_idx_dev, int n_idx, int H, int W, int py, int px,
                                     const float *scale_dev, long long scale_stride, float *gimg_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(gsino_dev && Tinv8_dev && angle_idx_dev && gimg_dev && S > 0 && H > 0 && W > 0 && PW > 0,
                   "rotate_bwd_sel: null pointer or empty sizes");
    CTPVAE_REQUIRE(A_plan > 0 && n_idx > 0, "rotate_bwd_sel: need angles (plan %d, subset %d)", A_plan, n_idx);
    CTPVAE_REQUIRE(knob(kKnobForceGeneric) < 0, "rotate_bwd_sel: not available with FORCE_GENERIC");
    return for_slice_chunks(S, max_slices_per_launch(), [&](int s0, int n) {
        return rotate_bwd_one(gsino_dev + (size_t)s0 * n_idx * PW, n, n_idx, PH, PW, Tinv8_dev, CTPVAE_NEAREST,
                              CTPVAE_BWD_TF_COMPAT, H, W, py, px,
                              scale_dev ? scale_dev + (long long)s0 * scale_stride : nullptr, scale_stride,
                              gimg_dev + (size_t)s0 * H * W, stream, angle_idx_dev, A_plan);
    });
}

static int rotate_bwd_one(const float *gsino_dev, int S, int A, int PH, int PW, const float *T8_dev, int interp,
                          int mode, int H, int W, int py, int px, const float *scale_dev, long long scale_stride,
                          float *gimg_dev, ctpvae_stream_t stream, const int *sel_dev, int A_plan)
{
    CTPVAE_REQUIRE(gsino_dev && T8_dev && gimg_dev, "rotate_bwd: null pointer");
    // only the NEAREST / TF_COMPAT segment kernel applies the per-slice factor in its store
    CTPVAE_REQUIRE(scale_dev == nullptr || (mode == CTPVAE_BWD_TF_COMPAT && interp == CTPVAE_NEAREST && S <= 65535 &&
                                            knob(kKnobForceGeneric) < 0),
                   "rotate_bwd_scaled: a per-slice scale needs interp=NEAREST, mode=TF_COMPAT and S <= 65535");
    if (int rc = check_geom("rotate_bwd", S, H, W, PH, PW, py, px, A, interp)) return rc;
    CTPVAE_REQUIRE(mode == CTPVAE_BWD_TF_COMPAT || mode == CTPVAE_BWD_EXACT, "rotate_bwd: unknown mode %d", mode);
    const RotGeom g{S, H, W, PH, PW, py, px, A};
    if (mode == CTPVAE_BWD_TF_COMPAT && interp == CTPVAE_NEAREST && knob(kKnobForceGeneric) < 0 && S <= 65535) {
        // 64-column x 32-row tiles, an 80-bin cotangent segment per angle in LDS (<= 31 KiB per chunk of angles)
        // two slices per workgroup (shared coordinates and addresses, ds_read_b64) once the launch can spare the
        // workgroups: 16 slices, or pairs that still make 256 tiles of 64 x 16 (tools/sweep_bwd.py: 512 x 512 x 90 angles,
        // 8 slices 33 -> 25 us, 12 slices 46 -> 32 us; 256 x 256, 4 slices x 20 angles stays single)
        int ns = (S >= 16 || (S >= 2 && (long long)ceil_div(S, 2) * ceil_div(W, 64) * ceil_div(H, 16) >= 256)) ? 2 : 1;
        if (knob(kKnobSegNs) >= 0) ns = (knob(kKnobSegNs) == 2 && S >= 2) ? 2 : 1;
        const int units = ceil_div(S, ns);
        int chunk_a = std::min(A, ns == 2 ? 48 : 96);   // ~32 KiB of LDS either way: five 4-wave workgroups per CU
        if (knob(kKnobSegChunk) >= 0) chunk_a = std::max(1, std::min(A, std::min(knob(kKnobSegChunk), ns == 2 ? 90 : 180)));
        const size_t shmem = (size_t)chunk_a * (kSegPitch * sizeof(float) * ns + 2 * sizeof(int) + 4 * sizeof(float)) + 16;
        // 8 rows per lane (64 x 32 tiles); 4 (64 x 16) when that is what it takes to put ~2 workgroups on every CU
        int ppt = (long long)units * ceil_div(W, 64) * ceil_div(H, 32) >= 512 ? 8 : 4;
        if (knob(kKnobSegPpt) >= 0) ppt = knob(kKnobSegPpt) == 4 ? 4 : 8;
        const dim3 grid(ceil_div(W, 64), ceil_div(H, 4 * ppt), units), block(256);
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)stream, gsino_dev, g, T8_dev, chunk_a,
                               SliceScale{scale_dev, scale_stride}, gimg_dev, sel_dev, A_plan);
        };
        if (ns == 2) {
            if (ppt == 8) launch(rotate_bwd_tfcompat_seg_kernel<8, 2>); else launch(rotate_bwd_tfcompat_seg_kernel<4, 2>);
        } else {
            if (ppt == 8) launch(rotate_bwd_tfcompat_seg_kernel<8, 1>); else launch(rotate_bwd_tfcompat_seg_kernel<4, 1>);
        }
        CTPVAE_LAUNCH_CHECK("rotate_bwd_tfcompat_seg_kernel");
        return CTPVAE_OK;
    }
    CTPVAE_REQUIRE(sel_dev == nullptr, "rotate_bwd_sel: only the NEAREST / TF_COMPAT segment kernel takes an angle subset (S <= 65535)");
    if (mode == CTPVAE_BWD_TF_COMPAT && knob(kKnobForceGeneric) < 0 && knob(kKnobNoPlan) < 0)   // round 5: segments, slices per cell
        return bilin_bwd_tfcompat(gsino_dev, S, A, PH, PW, T8_dev, H, W, py, px, gimg_dev, stream);
    if (mode == CTPVAE_BWD_TF_COMPAT && knob(kKnobForceGeneric) < 0 && S <= 65535) {
        // BILINEAR: 64-column x (4 waves x PPT rows) tiles; whole cotangent rows in LDS, <= 48 KiB per chunk
        constexpr int kPpt = 8;
        const int pitchg = PW + 4;
        int chunk_a = (48 * 1024) / (pitchg * (int)sizeof(float));
        CTPVAE_REQUIRE(chunk_a >= 1, "rotate_bwd: a detector row of %d bins does not fit LDS", PW);
        chunk_a = std::min(chunk_a, A);
        const size_t shmem = (size_t)chunk_a * pitchg * sizeof(float);
        const dim3 grid(ceil_div(W, 64), ceil_div(H, 4 * kPpt), S), block(256);
        hipLaunchKernelGGL((rotate_bwd_tfcompat_fast_kernel<CTPVAE_BILINEAR, kPpt>), grid, block, shmem, (hipStream_t)stream,
                           gsino_dev, g, T8_dev, chunk_a, gimg_dev);
        CTPVAE_LAUNCH_CHECK("rotate_bwd_tfcompat_fast_kernel");
        return CTPVAE_OK;
    }
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
