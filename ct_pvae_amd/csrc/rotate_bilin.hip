// rotate_bilin.hip -- the BILINEAR quadrants of the rotate-and-sum projector for gfx950 (round 5): forward
// (project_tf_low_mem, ctvae/forward_functions.py:69-77), TensorFlow-compatible backward, exact adjoint.
//
// A bilinear sample of TensorFlow's ImageProjectiveTransformV3 (a3) is
//     x = (t0 j + t1 i) + t2, y = (t3 j + t4 i) + t5            (unfused fp32; this file is compiled -ffp-contract=off)
//     xf = floor(x), xc = xf + 1, yf = floor(y), yc = yf + 1
//     v = (yc - y) * ((xc - x) * I[yf][xf] + (x - xf) * I[yf][xc]) + (y - yf) * ((xc - x) * I[yc][xf] + (x - xf) * I[yc][xc])
// with every tap zero-filled outside the canvas.  Coordinates, the four weights and the tap address depend on (angle, row, bin)
// only -- not on the slice -- so the kernels here compute them ONCE per sample for NS slices interleaved per LDS cell (the fact
// the nearest path exploits with slice pairs and quads): ~14 vector instructions of index / weight arithmetic per sample
// whatever NS, plus 10 unfused fp32 operations per slice for the blend (packed two slices at a time: v_pk_mul_f32 /
// v_pk_add_f32 round each half like the scalar operation).
//
// Where x >= 0 the weights come from v_fract_f32: x - floor(x) is exact there, and (floor(x) + 1) - x and 1 - fract(x) round the
// same real number (tools/probe_bilin.hip: 0 of 4 M values differ); below zero they differ, so a canvas without padding
// (px == 0 or py == 0: a sample at x in (-1, 0) is live there) takes the literal expressions (PADDED = false).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "lds_stage.h"
#include "loglik_math.h"
#include "rotate_plan.h"
#include "rotate_dev.h"
#include "tune_stamps.h"

namespace ctpvae {

// ---- forward ---------------------------------------------------------------------------------------------------------------
//
// LDS image of one unit (a whole slice, or one tile of a slice larger than LDS), NS slices interleaved per cell:
//     rows 0, 1      zero                      (row 1 = canvas row Y0 - 1: the zero fill above the slice; other tiles never own
//                                               a sample whose floor tap lies above them)
//     rows 2 .. h+1  the unit's rows
//     row  h+2       the HALO: slice row y0 + h (a sample's 2 x 2 footprint belongs to the unit of its floor tap), zero at the
//                    slice's last row
//     cols 0 .. w-1  the unit's columns, col w the halo column; col -1 of row r is col pitch-1 of row r-1 and is ZERO
// (column-MIRRORED for the angles whose lane step and row step have opposite signs, like the planned kernels: the pitch is
// == +1 (mod 32 cells; mod 16 for 16-byte cells) for both classes, and a sample's pair of cells is then (xc, xf) instead of
// (xf, xc) -- the two products are added in the other order, which fp32 addition does not notice).
// A sample is OWNED by the unit iff its floor tap lies in [Y0 - first_row, Y0 + h) x [X0 - first_col, X0 + w); every other
// sample reads the all-zero 2 x 2 block at cell (0, 0) and adds +0.  For a whole slice that is TensorFlow's zero fill; for tiles
// it makes the sinogram the sum of the tiles' partial sums (rotate_tile_reduce_kernel adds them in tile order:
// oracle_rotate_fwd_tiled(interp = 1) restates that association).
template <int NS> struct BilinCell { static constexpr int kBytes = 4 * NS; static constexpr int kShift = NS == 1 ? 2 : (NS == 2 ? 3 : 4); };

__host__ __device__ inline int bilin_pitch(int w, bool tiled, int ns)
{
    const int wb = w + (tiled ? 2 : 1);          // + halo column (+ a zero column of its own when the halo can hold data)
    const int m = ns == 4 ? 16 : 32;             // cells per bank sweep of one hardware lane group
    return wb + ((1 - (wb % m)) + m) % m;        // smallest pitch >= wb with pitch == 1 (mod m)
}
__host__ __device__ inline size_t bilin_lds_cells(int h, int w, bool tiled, int ns) { return (size_t)(h + 3) * bilin_pitch(w, tiled, ns); }

struct BilinRay {
    float t1, t2, t4, t5, xj, yj;
    int ray, ilo, kmax;   // kmax: rows walked in pairs (even; row-split walks: in fours)
    int ka, kb;           // wave-uniform, even (fours): on rows [ka, kb) of the walk EVERY live lane's sample is owned by the unit
    bool live;
    int tail;             // rows behind the pairs (an odd canvas height walked whole: 1; row-split walks: up to 3)
    bool none;            // the ray misses the unit: whatever its lane walks on the interior, its sum is zero
};

template <int NS, bool TILED, bool PADDED, bool SORTED, bool RS = false>
__global__ __launch_bounds__(1024) void rotate_fwd_bilin_kernel(const float *__restrict__ img, RotGeom gfull, TileSpec ts,
                                                                const float *__restrict__ T8, int t8_lds_off, float *__restrict__ out)
{
    typedef typename PixVec<NS>::type vec_t;
    constexpr int SHIFT = BilinCell<NS>::kShift, CELL = BilinCell<NS>::kBytes;
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int cls = blockIdx.x & 1, gi = blockIdx.x >> 1, G = gridDim.x >> 1;
    const bool mirror = cls != 0;
    // the unit: a whole slice (nt = 1) or tile t of the slice, for slices s .. s + NS - 1
    const int nt = TILED ? ts.ntx * ts.nty : 1;
    const int t = TILED ? blockIdx.y % nt : 0;
    const int s = (TILED ? blockIdx.y / nt : blockIdx.y) * NS;
    RotGeom g = gfull;
    int y0 = 0, x0 = 0;
    if (TILED) tile_rect(gfull, ts, t, y0, x0, g.H, g.W);
    g.py = gfull.py + y0;
    g.px = gfull.px + x0;
    const int h = g.H, w = g.W;
    const int lowy = y0 == 0 ? 1 : 0, lowx = x0 == 0 ? 1 : 0;            // the first row / column of units owns the floor taps at -1
    const bool halo_y = y0 + h < gfull.H, halo_x = x0 + w < gfull.W;     // ... and the others stage a halo of real pixels
    const int pitch = bilin_pitch(w, TILED, NS);
    const float *srcs[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) srcs[n] = img + ((size_t)min(s + n, gfull.S - 1) * gfull.H + y0) * gfull.W + x0;

    CTPVAE_PSTAMP(0);
    // ---- stage ---------------------------------------------------------------------------------------------------------------
    // (the transform rows this thread copies to LDS and the first 64 angles' class tests are requested BEFORE the unit's rows: behind
    // them each was another round trip to memory in front of the barrier -- loads return in order, the rows' wait covers them)
    const float t8_first = (int)threadIdx.x < 8 * g.A ? T8[threadIdx.x] : 0.0f;
    float cls_t0 = 0.0f, cls_t3 = 0.0f;
    if (threadIdx.x < 64) {
        const float *tm = T8 + 8 * min(lane, g.A - 1);
        cls_t0 = tm[0];
        cls_t3 = tm[3];
    }
    {
        vec_t *cells = reinterpret_cast<vec_t *>(lds);
        // LDS column of canvas column X0 + c (c = -1 .. w), relative to the row's first cell
        auto col_of = [&](int c) { return mirror ? w - 1 - c : c; };
        // (the general forms: with the lean form of stage_unit the headline shape measured 1-2 % slower -- this kernel is bound
        // by its walks, and its register allocation is the one thing the prologue can still spoil)
        if constexpr (NS == 1)
            stage_rows(lds + 2 * pitch, srcs[0], h, w, gfull.W, pitch, mirror, lane, wave, nwaves);
        else
            stage_rows_interleaved<NS>(lds + (size_t)2 * pitch * NS, srcs, h, w, gfull.W, pitch, mirror, lane, wave, nwaves);
        // rows 0 and 1 (but the last cell of row 1: it is "column -1" of row 2, written below)
        for (int p = threadIdx.x; p < 2 * pitch - 1; p += blockDim.x) cells[p] = vec_t(0.0f);
        // the halo row
        for (int c = threadIdx.x; c < w; c += blockDim.x) {
            vec_t v = vec_t(0.0f);
            if (halo_y) {
#pragma unroll
                for (int n = 0; n < NS; ++n) {
                    const float pv = srcs[n][(size_t)h * gfull.W + c];
                    if constexpr (NS == 1) v = pv; else v[n] = pv;
                }
            }
            cells[(h + 2) * pitch + col_of(c)] = v;
        }
        // columns -1 (zero) and w (halo) of rows 2 .. h + 2
        for (int p = threadIdx.x; p < 2 * (h + 1); p += blockDim.x) {
            const int r = p >> 1, right = p & 1;                       // tile row r (r = h: the halo row)
            vec_t v = vec_t(0.0f);
            if (right && halo_x && (r < h || halo_y)) {
#pragma unroll
                for (int n = 0; n < NS; ++n) {
                    const float pv = srcs[n][(size_t)r * gfull.W + w];
                    if constexpr (NS == 1) v = pv; else v[n] = pv;
                }
            }
            // (a whole slice's pitch is w + 1: columns -1 and w share cells, and both are zero)
            cells[(r + 2) * pitch + col_of(right ? w : -1)] = v;
        }
    }

    // behind the image: a copy of the transform rows, the ascending list of this class's angles ([0] = their count), a task counter
    int *cls_list = reinterpret_cast<int *>(lds + t8_lds_off + 8 * g.A);
    if ((int)threadIdx.x < 8 * g.A) lds[t8_lds_off + threadIdx.x] = t8_first;
    for (int p = threadIdx.x + blockDim.x; p < 8 * g.A; p += blockDim.x) lds[t8_lds_off + p] = T8[p];
    if (threadIdx.x < 64) {
        int n = 0;
        for (int a0 = 0; a0 < g.A; a0 += 64) {
            const float *tm = T8 + 8 * min(a0 + lane, g.A - 1);
            const float c0 = a0 == 0 ? cls_t0 : tm[0], c3 = a0 == 0 ? cls_t3 : tm[3];
            const bool in_cls = a0 + lane < g.A && ((((c0 >= 0.0f) == (c3 >= 0.0f)) ? 0 : 1) == cls);
            const unsigned long long m = __ballot(in_cls);
            if (in_cls) cls_list[1 + n + __popcll(m & ((1ull << lane) - 1ull))] = a0 + lane;
            n += __popcll(m);
        }
        if (lane == 0) {
            cls_list[0] = n;
            cls_list[1 + g.A] = SORTED ? 0 : (int)(blockDim.x >> 6);   // the task counter (unsorted: the waves' first tasks are fixed)
        }
    }

    CTPVAE_PSTAMP(1);
    const int nb = TILED ? ts.nb : ((g.PW + 63) & ~63);   // ray slots per angle
    const int span = TILED ? ts.span : g.PW;
    const float tile_cx = (float)g.px + 0.5f * (float)(w - 1), tile_cy = (float)g.py + 0.5f * (float)(h - 1);
    const int lds_base = (int)(uintptr_t)(lds_cptr)lds;
    // ownership window and address constants (canvas coordinates): cy = iy - ylo in [0, ny), LDS row = iy - g.py + 2
    const int xlo = g.px - lowx, nx = w + lowx, ylo = g.py - lowy, ny = h + lowy;
    const int pitchB = pitch * CELL;
    // byte address of the sample's LOW cell: row (iy - g.py + 2), column (ix - g.px), or mirrored (w - 2 - (ix - g.px))
    const int off_rows = (2 - g.py) * pitchB + lds_base;
    const int off_cols = mirror ? (w - 2 + g.px) * CELL : -g.px * CELL;
    int off_v = pin_vgpr(off_rows + off_cols);   // (not const: a const int is not captured by the nested generic lambdas)

    // one lane's ray: transform row of its angle, bin, conservative row range [ilo, ilo + cnt) through the unit
    auto prepare = [&](int a, int slot, int &ilo, int &cnt) -> BilinRay {
        BilinRay q;
        const int ad = (t8_lds_off + 8 * a) * 4 + lds_base;
        const f32x4 u = lds_abs_vec<4>(ad);
        const f32x2 v2 = lds_abs_vec<2>(ad + 16);
        const float t6[6] = {u.x, u.y, u.z, u.w, v2.x, v2.y};
        int j = slot;
        q.live = slot < span;
        if (TILED) j += tile_first_bin(t6, tile_cx, tile_cy, ts.radius);
        q.live = q.live && (unsigned)j < (unsigned)g.PW;
        q.ray = a * nb + slot;
        if (!TILED) q.ray = a * g.PW + j;
        q.t1 = t6[1]; q.t2 = t6[2]; q.t4 = t6[4]; q.t5 = t6[5];
        q.xj = t6[0] * (float)j;
        q.yj = t6[3] * (float)j;
        float lo = 0.0f, hi = (float)g.PH;
        clip_rows(q.xj + q.t2, q.t1, (float)(g.px - 2), (float)(g.px + w + 1), lo, hi);
        clip_rows(q.yj + q.t5, q.t4, (float)(g.py - 2), (float)(g.py + h + 1), lo, hi);
        lo = fminf(fmaxf(lo, 0.0f), (float)g.PH);
        hi = fminf(fmaxf(hi, -1.0f), (float)g.PH);
        ilo = max((int)floorf(lo) - 1, 0);
        const int ihi = min((int)ceilf(hi) + 2, g.PH);
        cnt = q.live ? max(ihi - ilo, 0) : 0;
        return q;
    };
    // ... and the wave's common trip count: every lane walks kmax rows (+ the tail row) from its own first row
    auto setup = [&](int a, int slot) -> BilinRay {
        int ilo, cnt;
        BilinRay q = prepare(a, slot, ilo, cnt);
        const int need = wave_max_nonneg(cnt);                  // wave-uniform trip count (SGPR)
        constexpr int RM = RS ? 3 : 1;                          // rows per step of the walk - 1
        q.kmax = min((need + RM) & ~RM, g.PH & ~RM);            // whole row pairs (fours) ...
        q.tail = need > q.kmax ? g.PH - q.kmax : 0;             // ... and the last rows of a canvas walked whole alone
        q.ilo = max(min(ilo, g.PH - q.kmax - q.tail), 0);       // only legitimate rows are visited
        // The INTERIOR of the walk: rows on which the floor tap of every live lane's sample lies inside the unit's ownership
        // window need no ownership test (12 -> 6 vector instructions for floor, test and address; see walk).  Per lane from the
        // ray's line, conservatively: the window shrunk by 0.01 px (the walk's own coordinates are fp32 sums, off by ~1e-5),
        // one row given up at either end; per wave the intersection over its live lanes, in rows of the walk (every lane counts
        // from its own first row), cut to whole row pairs.
        float lo = 0.0f, hi = (float)g.PH;
        auto inside = [&](float base, float slope, float L, float U) {
            if (fabsf(slope) < 1e-6f) {
                if (base < L || base > U) hi = -1.0f;
            } else {
                const float inv = __builtin_amdgcn_rcpf(slope);
                const float i1 = (L - base) * inv, i2 = (U - base) * inv;
                lo = fmaxf(lo, fminf(i1, i2));
                hi = fminf(hi, fmaxf(i1, i2));
            }
        };
        inside(q.xj + q.t2, q.t1, (float)xlo + 0.01f, (float)(xlo + nx) - 0.01f);
        inside(q.yj + q.t5, q.t4, (float)ylo + 0.01f, (float)(ylo + ny) - 0.01f);
        const int ia = (int)ceilf(lo) + 1, ib = (int)floorf(fmaxf(hi, -1.0f));        // rows [ia, ib) are inside for sure
        int ka_l = 0, kb_l = q.kmax;
        q.none = cnt == 0;
        if (cnt > 0) {   // (a lane without rows walks whatever it walks: its sum is not stored, or stored as zero)
            ka_l = min(max(ia - q.ilo, 0), q.kmax);
            kb_l = max(min(ib - q.ilo, q.kmax), ka_l);
        }
        q.ka = (wave_max_nonneg(ka_l) + RM) & ~RM;
        q.kb = (q.kmax - wave_max_nonneg(q.kmax - kb_l)) & ~RM;
        if (q.kb <= q.ka) q.ka = q.kb = q.kmax;                  // no interior: the whole walk with tests
        return q;
    };

    auto walk = [&](const BilinRay &q, auto mirror_tag) {
        constexpr bool MIRROR = decltype(mirror_tag)::value;
        const f32x2 basex = {q.xj, q.xj}, basey = {q.yj, q.yj}, stepx = {q.t1, q.t1}, stepy = {q.t4, q.t4};
        const f32x2 shiftx = {q.t2, q.t2}, shifty = {q.t5, q.t5};
        // (row-split walks: lanes 32-63 carry the SAME rays as lanes 0-31, two rows further on -- four rows of a ray per step;
        // the sums are kept, and stored, by lanes 32-63)
        const float fi0 = (float)(q.ilo + (RS ? (lane >> 5) * 2 : 0));
        f32x2 fi = {fi0, fi0 + 1.0f};
        vec_t acc = vec_t(0.0f);
        const int offv = off_v;     // (named here: an outer variable used only inside an asm operand of a nested lambda is not captured)
        struct Pair {               // two consecutive rows of one ray
            f32x2 wl, wh, wy0, wy1;   // weights of the low / high cell of a pair, of the floor / ceil row
            vec_t tp[2][4];           // taps: [row][low cell of the floor row, high, low cell of the ceil row, high]
        };
        // One row: floor taps -> ownership test -> byte address of the low cell of the floor row (the all-zero block at LDS
        // address 0 for a sample this unit does not own), as ONE asm statement of 12 vector instructions (the select is a
        // v_cndmask; hipcc's own code branches around the address arithmetic with s_and_saveexec); then the four cells with
        // plain ds_read_b32 / _b64 / _b128 at immediate offsets -- VOLATILE loads: hipcc pairs ordinary ones into ds_read2_b64,
        // which tools/probe_bilin.hip measured at 7.1 ns per cell pair against 4.1 for two ds_read_b64.  (The loads stay C++:
        // issued inside the asm statement they were invisible to hipcc's s_waitcnt insertion, and hipcc copied the pending tap
        // registers on a loop edge before the hand-written wait -- stale taps, found by the fuzz test.)
        // On the walk's INTERIOR (setup: every live lane's sample is owned) the test falls away: 6 instructions.
        auto row_taps = [&](float x, float y, vec_t (&tp)[4], auto test_tag) {
            int ix, iy, tmp, ad, ad2;
            if constexpr (!decltype(test_tag)::value) {
                asm volatile("s_nop 0\n\t"
                             "v_cvt_flr_i32_f32 %[ix], %[x]\n\t"
                             "v_cvt_flr_i32_f32 %[iy], %[y]\n\t"
                             "v_mad_i32_i24 %[iy], %[iy], %[pB], %[off]\n\t"
                             "v_mad_i32_i24 %[ad], %[ix], %[cs], %[iy]\n\t"
                             "v_add_u32 %[ad2], %[pB], %[ad]"
                             : [ix] "=&v"(ix), [iy] "=&v"(iy), [ad] "=&v"(ad), [ad2] "=&v"(ad2)
                             : [x] "v"(x), [y] "v"(y), [pB] "s"(pitchB), [off] "v"(offv), [cs] "n"(MIRROR ? -CELL : CELL));
                (void)tmp;
            } else
            asm volatile("s_nop 0\n\t"   // x / y may come straight out of a packed op, whose result the next instruction cannot read
                         "v_cvt_flr_i32_f32 %[ix], %[x]\n\t"
                         "v_cvt_flr_i32_f32 %[iy], %[y]\n\t"
                         "v_subrev_u32 %[tmp], %[xlo], %[ix]\n\t"
                         "v_subrev_u32 %[ad2], %[ylo], %[iy]\n\t"
                         "v_cmp_gt_u32 vcc, %[nx], %[tmp]\n\t"
                         "v_mad_i32_i24 %[iy], %[iy], %[pB], %[off]\n\t"
                         "v_cndmask_b32 %[tmp], -1, %[ad2], vcc\n\t"      // column not owned: row index 0xffffffff fails the next test
                         "v_mad_i32_i24 %[ix], %[ix], %[cs], %[iy]\n\t"
                         "v_cmp_gt_u32 vcc, %[ny], %[tmp]\n\t"
                         "v_cndmask_b32 %[ad], 0, %[ix], vcc\n\t"
                         "v_add_u32 %[ad2], %[pB], %[ad]"
                         : [ix] "=&v"(ix), [iy] "=&v"(iy), [tmp] "=&v"(tmp), [ad] "=&v"(ad), [ad2] "=&v"(ad2)
                         : [x] "v"(x), [y] "v"(y), [xlo] "s"(xlo), [nx] "s"(nx), [ylo] "s"(ylo), [ny] "s"(ny), [pB] "s"(pitchB),
                           [off] "v"(offv), [cs] "n"(MIRROR ? -CELL : CELL)
                         : "vcc");
            typedef const volatile __attribute__((address_space(3))) vec_t *vptr;
            tp[0] = *(vptr)(uintptr_t)(unsigned)ad;
            tp[1] = *(vptr)(uintptr_t)(unsigned)(ad + CELL);
            tp[2] = *(vptr)(uintptr_t)(unsigned)ad2;
            tp[3] = *(vptr)(uintptr_t)(unsigned)(ad2 + CELL);
        };
        auto issue = [&](Pair &P, auto test_tag) {
            const f32x2 x = (basex + stepx * fi) + shiftx;
            const f32x2 y = (basey + stepy * fi) + shifty;
            fi += RS ? 4.0f : 2.0f;
            f32x2 wx0, wx1;
            if constexpr (PADDED) {
                wx1 = f32x2{__builtin_amdgcn_fractf(x.x), __builtin_amdgcn_fractf(x.y)};
                P.wy1 = f32x2{__builtin_amdgcn_fractf(y.x), __builtin_amdgcn_fractf(y.y)};
                wx0 = 1.0f - wx1;
                P.wy0 = 1.0f - P.wy1;
            } else {
                const f32x2 xf = {floorf(x.x), floorf(x.y)}, yf = {floorf(y.x), floorf(y.y)};
                const f32x2 xc = xf + 1.0f, yc = yf + 1.0f;
                wx0 = xc - x; wx1 = x - xf;
                P.wy0 = yc - y; P.wy1 = y - yf;
            }
            P.wl = MIRROR ? wx1 : wx0;
            P.wh = MIRROR ? wx0 : wx1;
            row_taps(x.x, y.x, P.tp[0], test_tag);
            row_taps(x.y, y.y, P.tp[1], test_tag);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto consume = [&](const Pair &P) {
            vec_t val[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float wl = r ? P.wl.y : P.wl.x, wh = r ? P.wh.y : P.wh.x;
                const float w0 = r ? P.wy0.y : P.wy0.x, w1 = r ? P.wy1.y : P.wy1.x;
                const vec_t v_yf = wl * P.tp[r][0] + wh * P.tp[r][1];
                const vec_t v_yc = wl * P.tp[r][2] + wh * P.tp[r][3];
                val[r] = w0 * v_yf + w1 * v_yc;
                if constexpr (!RS) acc += val[r];
            }
            if constexpr (RS) {
                // the ray's four rows in order, in lanes 32-63: the two of lanes 0-31, then their own.  v_permlane32_swap: lanes 32-63 of
                // its first operand change places with lanes 0-31 of its second -- the first operand is a dead tap register (its upper
                // half receives the other half's value), the second keeps its upper half, which is all that is read of it afterwards.
                // What lanes 0-31 accumulate is never stored.  (Inline: hipcc's builtin returns the first operand's register for BOTH
                // results on ROCm 7.2 and copies the second operand first.  Two wait states between a vector write and the swap.)
                vec_t oth[2];
                if constexpr (NS == 1) {
                    oth[0] = P.tp[0][0], oth[1] = P.tp[1][0];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3"
                                 : "+v"(oth[0]), "+v"(oth[1]), "+v"(val[0]), "+v"(val[1]));
                } else if constexpr (NS == 2) {
                    oth[0] = P.tp[0][0], oth[1] = P.tp[1][0];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\t"
                                 "v_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7"
                                 : "+v"(oth[0].x), "+v"(oth[0].y), "+v"(oth[1].x), "+v"(oth[1].y), "+v"(val[0].x), "+v"(val[0].y), "+v"(val[1].x), "+v"(val[1].y));
                } else {
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        oth[r] = P.tp[r][0];
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\t"
                                     "v_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7"
                                     : "+v"(oth[r].x), "+v"(oth[r].y), "+v"(oth[r].z), "+v"(oth[r].w), "+v"(val[r].x), "+v"(val[r].y), "+v"(val[r].z), "+v"(val[r].w));
                    }
                }
                acc += oth[0];
                acc += oth[1];
                acc += val[0];
                acc += val[1];
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // three stretches of row pairs: [0, ka) and [kb, kmax) with the ownership test, [ka, kb) -- the interior -- without
        Pair A;
        [[maybe_unused]] Pair B;
        auto stretch = [&](int npairs, auto test_tag) {
            if constexpr (NS == 4) {
                // four slices per cell: one pair in flight (two would be 64 registers of taps; the workgroup's other waves cover
                // the LDS latency -- the loop is bound by vector instructions either way)
                for (int b = 0; b < npairs; ++b) {
                    issue(A, test_tag);
                    consume(A);
                }
            } else if (npairs > 0) {
                issue(A, test_tag);
                int b = 1;
                for (; b + 1 < npairs; b += 2) {
                    issue(B, test_tag);
                    consume(A);
                    issue(A, test_tag);
                    consume(B);
                }
                if (b < npairs) {
                    issue(B, test_tag);
                    consume(A);
                    consume(B);
                } else {
                    consume(A);
                }
            }
        };
        constexpr int RSH = RS ? 2 : 1;
        stretch(q.ka >> RSH, std::true_type{});
        stretch((q.kb - q.ka) >> RSH, std::false_type{});
        stretch((q.kmax - q.kb) >> RSH, std::true_type{});
        for (int tr = 0; tr < q.tail; ++tr) {   // an odd canvas height walked whole: its last row (row-split walks: up to three, every lane all of them)
            const float fr = (float)(q.ilo + q.kmax + tr);
            const float x = (q.xj + q.t1 * fr) + q.t2, y = (q.yj + q.t4 * fr) + q.t5;
            const float xf = floorf(x), yf = floorf(y), xc = xf + 1.0f, yc = yf + 1.0f;
            const int ixr = cvt_flr(x), iyr = cvt_flr(y);
            const bool own = (unsigned)(ixr - xlo) < (unsigned)nx && (unsigned)(iyr - ylo) < (unsigned)ny;
            int ad = __mul24(iyr, pitchB) + off_rows;
            ad = MIRROR ? ad - (ixr << SHIFT) + off_cols : ad + (ixr << SHIFT) + off_cols;
            ad = own ? ad : lds_base;
            const float wl = MIRROR ? x - xf : xc - x, wh = MIRROR ? xc - x : x - xf;
            const vec_t v_yf = wl * lds_abs_vec<NS>(ad) + wh * lds_abs_vec<NS>(ad + CELL);
            const vec_t v_yc = wl * lds_abs_vec<NS>(ad + pitchB) + wh * lds_abs_vec<NS>(ad + pitchB + CELL);
            acc += (yc - y) * v_yf + (y - yf) * v_yc;
        }
        if (q.none) acc = vec_t(0.0f);
        if (q.live && (!RS || lane >= 32)) {
            if constexpr (TILED) {
                const size_t nrays = (size_t)g.A * nb;
                float *dst = out + partial_index(s, nt, t, nrays, (size_t)q.ray);
                if constexpr (NS == 1) *dst = acc; else *reinterpret_cast<vec_t *>(dst) = acc;
            } else {
#pragma unroll
                for (int n = 0; n < NS; ++n)
                    if (s + n < gfull.S) {
                        float av;
                        if constexpr (NS == 1) av = acc; else av = acc[n];
                        out[(size_t)(s + n) * g.A * g.PW + q.ray] = av;
                    }
            }
        }
    };

    // ---- tasks ------------------------------------------------------------------------------------------------------------------
    // Lanes of a wave walk in lockstep: all 64 take the trip count of the longest ray.  A unit's chord profile over its ray slots
    // is a trapezoid (tile) or worse (a square slice seen at 45 degrees), so a wave of 64 NEIGHBOURING slots spends 29 % (whole
    // 128 x 128 slice) to 44 % (64 x 86 tile) of its lane-rows on rays that have left the unit (tools/sim_bilin_tiles.py).
    // SORTED mode: the workgroup's (angle, band) pairs -- a band = 4 consecutive slots of one angle -- are sorted by length and
    // dealt 16 at a time: 0.88 of the walked lane-rows are then live for tiles (1.3x fewer wave-rows for a slice at 180 angles).
    // The hardware serves a ds_read_b32 / _b64 in two groups of 32 consecutive lanes and a ds_read_b128 in four groups of 16
    // lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32): whole bands sit inside one group; the extra conflicts
    // between a group's bands hide under the vector instructions (LDS array ~50 % busy).  Measured on one box at 32 x 512 x 512 x
    // 90 angles: mirrored 32-slot runs of one angle per wave 436 us, 16-slot bands 347, 8-slot 325, 4-slot 311-318.
    // The list is built HERE, per workgroup, for <= 2048 bands at a time (a chunk of the class's angles): every wave counts the
    // rows of its share of bands, a counting sort in LDS orders them, longest first, and the waves take tasks from an LDS counter.
    // Rays are independent: which task carries a ray does not touch its sum.  The G workgroups of a (unit, class) own DISJOINT
    // bands -- band b of every angle belongs to workgroup b mod G: each samples every angle's profile evenly, none needs
    // another's list.
    // UNSORTED mode (few tasks per wave: the headline shape has 6 per workgroup of 8 waves): the launch ends with its longest
    // ray whatever the other lanes do, and the sort's barriers would only delay the start (measured: B = 50 x 20 angles 24.5 us
    // sorted against 21.8; B = 100 x 20: 36.8 against 35.2).  Then: (angle, 64-slot block) tasks over all the class's angles,
    // every G-th to this workgroup, two MIRRORED 32-slot runs per wave (equal chords: one trip count serves both), the
    // innermost blocks first.
#ifdef CTPVAE_TUNE_BILIN_BSH
    constexpr int BSH = CTPVAE_TUNE_BILIN_BSH;   // timing builds: 8- or 16-slot bands
#else
    constexpr int BSH = 2;
#endif
    constexpr int BAND = 1 << BSH, PER = 64 / BAND;
    const int nbk = nb >> 6, nbands = nb >> BSH;
    __syncthreads();                                               // the image, the transform rows and the class list are staged
    CTPVAE_PSTAMP(2);
    const int ncls = __builtin_amdgcn_readfirstlane(cls_list[0]);
    int *next_task = cls_list + 1 + g.A;
    unsigned char *bcnt = reinterpret_cast<unsigned char *>(next_task + 1);          // [<= 2048] rows of a band
    int *hist = reinterpret_cast<int *>(bcnt + 2048);                                // [256] -> start offsets, descending
    unsigned short *order = reinterpret_cast<unsigned short *>(hist + 256);          // [<= 2048] entries li * nmine + k
    // lane -> (band of the task, slot within the band)
    int sub, kin;
    if constexpr (NS == 4) {
        const int l32 = lane & 31;
        const int grp = (lane >> 5) * 2 + ((l32 >= 4 && l32 < 12) || (l32 >= 16 && l32 < 20) || l32 >= 28 ? 1 : 0);
        const int k16 = (grp & 1) ? (l32 < 12 ? l32 - 4 : (l32 < 20 ? l32 - 8 : l32 - 16)) : (l32 < 4 ? l32 : (l32 < 16 ? l32 - 8 : l32 - 12));
        sub = grp * (16 >> BSH) + (k16 >> BSH);
        kin = k16 & (BAND - 1);
    } else {
        sub = lane >> BSH;
        kin = lane & (BAND - 1);
    }
    const int nmine = max(0, (nbands - gi + G - 1) / G);           // this workgroup's bands of an angle: gi, gi + G, ...
    const int nkg = (nmine + PER - 1) / PER;                       // ... in groups of 16
    const int CH = max(1, 2048 / max(nmine, 1));                   // angles per chunk
    // (SORTED is the host's choice from the launch shape: about two tasks or more per wave; a workgroup with no band of its own
    // has nothing to sort)
    if (SORTED && nmine == 0) return;
    for (int c0 = 0; c0 < ncls; c0 += (SORTED ? CH : ncls)) {
        const int nch = SORTED ? min(CH, ncls - c0) : ncls, E = nch * nmine;
        if constexpr (SORTED) {
            __syncthreads();                                           // (every wave has left the previous chunk's lists)
            for (int p = threadIdx.x; p < 256; p += blockDim.x) hist[p] = 0;
            if (threadIdx.x == 0) *next_task = 0;
            for (int it = wave; it < nch * nkg; it += nwaves) {        // (angle, 16 of this workgroup's bands) dealt to the waves
                const int li = it / nkg, k = (it - li * nkg) * PER + (lane >> BSH);
                const int a = __builtin_amdgcn_readfirstlane(cls_list[1 + c0 + li]);
                int ilo, cnt;
                (void)prepare(a, k < nmine ? ((gi + k * G) << BSH) + (lane & (BAND - 1)) : nb, ilo, cnt);
                cnt = min(cnt, 255);
                cnt = max(cnt, __builtin_amdgcn_update_dpp(0, cnt, 0x111, 0xf, 0xf, false));   // row_shr:1, 2: the max of a band
                cnt = max(cnt, __builtin_amdgcn_update_dpp(0, cnt, 0x112, 0xf, 0xf, false));   // of 4 lanes ends in its last lane
                if constexpr (BSH >= 3) cnt = max(cnt, __builtin_amdgcn_update_dpp(0, cnt, 0x114, 0xf, 0xf, false));
                if constexpr (BSH >= 4) cnt = max(cnt, __builtin_amdgcn_update_dpp(0, cnt, 0x118, 0xf, 0xf, false));
                if ((lane & (BAND - 1)) == BAND - 1 && k < nmine) bcnt[li * nmine + k] = (unsigned char)cnt;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < E; e += blockDim.x) atomicAdd(&hist[bcnt[e]], 1);
            __syncthreads();
            if (wave == 0) {   // hist[c] <- number of bands LONGER than c (their start in the descending order); 4 lengths per lane
                const int cc = 255 - 4 * lane;                    // this lane's lengths cc, cc - 1, cc - 2, cc - 3 (descending)
                const int h0 = hist[cc], h1 = hist[cc - 1], h2 = hist[cc - 2], h3 = hist[cc - 3];
                int run = h0 + h1 + h2 + h3, incl = run;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int up = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += up;
                }
                const int base = incl - run;
                hist[cc] = base;
                hist[cc - 1] = base + h0;
                hist[cc - 2] = base + h0 + h1;
                hist[cc - 3] = base + h0 + h1 + h2;
            }
            __syncthreads();
            // (bands no ray of which meets the unit sort last and are still walked -- zero rows: their rays' sums must be stored)
            for (int e = threadIdx.x; e < E; e += blockDim.x) order[atomicAdd(&hist[bcnt[e]], 1)] = (unsigned short)e;
            __syncthreads();
        }
        const int ntask = SORTED ? (E + PER - 1) / PER : nch * nbk * (RS ? 2 : 1);
        // UNSORTED: a wave's FIRST task is fixed, dealt to the SIMDs like a snake -- waves w and w + 4 share a SIMD (HW_ID of the
        // stamped launches: tools/stamp_rounds.hip), the tasks come longest first, and this kernel is bound by the vector unit: two
        // waves of one SIMD take turns.  In arrival order the headline shape's six tasks fell 2 long + 2 short on two SIMDs and
        // one short each on the other two; the snake gives the long ones a SIMD each and pairs the short ones.
        bool first = !SORTED;
        for (;;) {
            int m = 0;
            if (first) {
                const int lo = wave & ~3, hi = min(lo + 3, nwaves - 1);      // the wave's quad of SIMDs (a last quad may be short)
                m = (wave & 4) ? lo + (hi - wave) : wave;                     // odd quads run backwards: a bijection of 0 .. nwaves - 1
            } else {
                if (lane == 0) m = atomicAdd(next_task, 1);
                m = __builtin_amdgcn_readfirstlane(m);
            }
            first = false;
            int a, slot;
            if constexpr (SORTED) {
                if (m >= ntask) break;
                const int idx = PER * m + sub;
                const int e = order[min(idx, E - 1)];
                const int li = e / nmine, k = e - li * nmine;
                a = cls_list[1 + c0 + li];
                slot = idx < E ? ((gi + k * G) << BSH) + kin : nb;   // (a last task's missing bands: slots past the span, dead)
            } else {
                m = m * G + gi;
                if (m >= ntask) break;
                const int bi = m / ncls, ai = m - bi * ncls, blk = nbk - 1 - bi;
                a = __builtin_amdgcn_readfirstlane(cls_list[1 + ai]);
                if constexpr (RS) {   // 32-slot runs, both halves of the wave on the same one; the innermost first, either side in turn
                    const int pr = bi >> 1;
                    slot = ((bi & 1) ? nbk + pr : nbk - 1 - pr) * 32 + (lane & 31);
                } else
                slot = lane < 32 ? blk * 32 + lane : nb - 32 * (blk + 1) + (lane - 32);
            }
            const BilinRay q = setup(a, slot);
            if (mirror) walk(q, std::true_type{}); else walk(q, std::false_type{});
        }
    }
    CTPVAE_PSTAMP(3);
}

// ---- backward, TensorFlow-compatible ------------------------------------------------------------------------------------------
//
// TensorFlow's registered gradient of ImageProjectiveTransformV3 (a4) samples the row-broadcast image of g[a][:] with the
// inverted transform and the same interpolation:
//     x = (t0 X + t1 Y) + t2, y = (t3 X + t4 Y) + t5 for pixel (X, Y) of the canvas,  h = (xc - x) g[xf] + (x - xf) g[xc]
//     G_a = (yc - y) * (row yf on the canvas ? h : 0) + (y - yf) * (row yc on the canvas ? h : 0),  gimg = sum over angles, ascending
// (both rows of the broadcast image hold the same h; a tap off the detector is zero-filled).  Round 1's kernel kept whole
// cotangent rows in LDS and one slice per workgroup (~27 vector operations per pixel and angle).  Here, as in the nearest
// segment kernel (rotate.hip): a 64-column x (4 waves x PPT rows) pixel tile stages, per angle, only the <= 80 bins its
// rectangle projects to -- cells off the detector staged as zeros, so no tap needs a bounds test --, NS slices interleaved per
// cell; coordinates, the four weights and the tap address are computed once per (pixel, angle) for all NS slices (12 vector
// instructions + 7 per slice, packed two slices at a time).  Every angle is classified once from the tile's corners:
//   0  every pixel of the tile maps inside the canvas with both taps (always, on a padded canvas): weights from v_fract_f32
//      (x, y >= 0: exact), no row test;
//   1  some pixel may map outside: the literal floor expressions and the reference's zero fill of rows off the canvas;
//   2  the segment would not hold the span (the table row is not a rotation): bounds-tested reads from global memory.
// Every pixel adds its angles in ascending order: bit-identical to the oracle whatever the class.
constexpr int kBSegBins = 80, kBSegPitch = kBSegBins + 1;
template <int PPT, int NS>
__global__ __launch_bounds__(PPT == 1 ? 1024 : (PPT == 2 ? 512 : 256)) void rotate_bwd_bilin_seg_kernel(const float *__restrict__ gsino, RotGeom g,
                                                                   const float *__restrict__ Tinv8, int chunk_a,
                                                                   float *__restrict__ gimg)
{
    typedef typename PixVec<NS>::type vec_t;
    constexpr int SHIFT = BilinCell<NS>::kShift, CELL = BilinCell<NS>::kBytes;
    // [chunk_a][kBSegPitch] cells, then per angle (first bin, class) ints, then per angle eight floats (t0, t1, t2, segment byte
    // base, t3, t4, t5, 0): the fast loop reads them with two broadcast ds_read_b128
    extern __shared__ float lds[];
    int *meta = reinterpret_cast<int *>(lds + chunk_a * kBSegPitch * NS);
    f32x4 *meta8 = reinterpret_cast<f32x4 *>(lds + ((chunk_a * (kBSegPitch * NS + 2) + 3) & ~3));
    const int s = blockIdx.z * NS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * (nwaves * PPT) + wave;   // rows r0, r0 + nwaves, ...
    // (ragged tiles: the rectangle clipped to the image, the lanes / rows past it repeating its last column / row -- rotate.hip's segment kernel)
    const float fx = (float)(min(c, g.W - 1) + g.px);
    const float X0 = (float)(blockIdx.x * 64 + g.px), X1 = X0 + (float)min(63, g.W - 1 - (int)blockIdx.x * 64);
    const float Y0 = (float)(blockIdx.y * (nwaves * PPT) + g.py), Y1 = Y0 + (float)min(nwaves * PPT - 1, g.H - 1 - (int)blockIdx.y * (nwaves * PPT));
    const float x_hi = (float)g.PW - 0.5f, y_hi = (float)g.PH - 0.5f;
    const int lds_base = (int)(uintptr_t)(lds_cptr)lds;
    size_t soff[NS];                                   // slices past the batch re-read the last one (never stored)
#pragma unroll
    for (int n = 0; n < NS; ++n) soff[n] = (size_t)(min(s + n, g.S - 1) - s) * g.A * g.PW;

    vec_t acc[PPT];
    float fy[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        acc[k] = vec_t(0.0f);
        fy[k] = (float)(min(r0 + k * nwaves, g.H - 1) + g.py);
    }

    for (int ac = 0; ac < g.A; ac += chunk_a) {
        const int na = min(chunk_a, g.A - ac);
        if (ac > 0) __syncthreads();
        int any_outside = 0;
        for (int al = threadIdx.x; al < na; al += blockDim.x) {
            const float *t = Tinv8 + 8 * (size_t)(ac + al);
            const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
            const float xa = (t0 * X0 + t1 * Y0) + t2, xb = (t0 * X1 + t1 * Y0) + t2;
            const float xc = (t0 * X0 + t1 * Y1) + t2, xd = (t0 * X1 + t1 * Y1) + t2;
            const float ya = (t3 * X0 + t4 * Y0) + t5, yb = (t3 * X1 + t4 * Y0) + t5;
            const float yc = (t3 * X0 + t4 * Y1) + t5, yd = (t3 * X1 + t4 * Y1) + t5;
            const float xmin = fminf(fminf(xa, xb), fminf(xc, xd)), xmax = fmaxf(fmaxf(xa, xb), fmaxf(xc, xd));
            const float ymin = fminf(fminf(ya, yb), fminf(yc, yd)), ymax = fmaxf(fmaxf(ya, yb), fmaxf(yc, yd));
            int cls = 0, first = 0;
            if (!(xmax - xmin <= (float)(kBSegBins - 6)) || !(fabsf(xmin) < 1.0e6f)) {
                cls = 2;   // (also NaN / absurd rows)
            } else {
                first = (int)floorf(xmin) - 2;     // taps floor(x), floor(x) + 1 of every pixel lie in [first + 1, first + 78]
                if (!(xmin > 0.5f && xmax < x_hi - 1.0f && ymin > 0.5f && ymax < y_hi - 1.0f)) cls = 1;
            }
            meta[2 * al] = first;
            meta[2 * al + 1] = cls;
            meta8[2 * al] = f32x4{t0, t1, t2, __int_as_float((al * kBSegPitch - first) * CELL + lds_base)};
            meta8[2 * al + 1] = f32x4{t3, t4, t5, 0.0f};
            any_outside |= cls;
        }
        const bool all_inside = __syncthreads_or(any_outside) == 0;   // (also the barrier behind the table)
        const float *src = gsino + ((size_t)s * g.A + ac) * g.PW;
        {
            constexpr int U = NS == 4 ? 2 : 8 / NS;   // cells in flight per thread; loads unconditional (clamped), the select comes after
            const int ncell = na * kBSegPitch;
            for (int p0 = threadIdx.x; p0 < ncell; p0 += U * blockDim.x) {
                vec_t v[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = min(p0 + u * (int)blockDim.x, ncell - 1);
                    const int al = p / kBSegPitch, q = p - al * kBSegPitch;
                    const int j = meta[2 * al] + q;
                    ok[u] = q < kBSegBins && (unsigned)j < (unsigned)g.PW;
                    const float *cell = src + al * g.PW + min(max(j, 0), g.PW - 1);
#pragma unroll
                    for (int n = 0; n < NS; ++n) {
                        if constexpr (NS == 1) v[u] = cell[0]; else v[u][n] = cell[soff[n]];
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

        typedef const volatile __attribute__((address_space(3))) vec_t *vptr;   // (volatile: no ds_read2_b64 pairing, see the forward)
        if (all_inside) {
            // class 0 throughout (a padded canvas always): two broadcast reads per angle, then per pixel 12 shared instructions
            // and two cells
            for (int al = 0; al < na; ++al) {
                const f32x4 m = meta8[2 * al], n4 = meta8[2 * al + 1];
                const float xa = m.x * fx, ya = n4.x * fx;
                const int k4 = __float_as_int(m.w);
                vec_t g0[PPT], g1[PPT];
                float wx1[PPT], wy1[PPT];
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const float x = (xa + m.y * fy[k]) + m.z;
                    const float y = (ya + n4.y * fy[k]) + n4.z;
                    wx1[k] = __builtin_amdgcn_fractf(x);
                    wy1[k] = __builtin_amdgcn_fractf(y);
                    int addr;   // (floor(x) - first) * cell bytes + segment base: one convert, one shift-add
                    asm("v_cvt_flr_i32_f32 %0, %1\n\tv_lshl_add_u32 %0, %0, %3, %2" : "=&v"(addr) : "v"(x), "v"(k4), "i"(SHIFT));
                    g0[k] = *(vptr)(uintptr_t)(unsigned)addr;
                    g1[k] = *(vptr)(uintptr_t)(unsigned)(addr + CELL);
                }
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const float wx0 = 1.0f - wx1[k], wy0 = 1.0f - wy1[k];
                    const vec_t h = wx0 * g0[k] + wx1[k] * g1[k];
                    acc[k] += wy0 * h + wy1[k] * h;
                }
            }
        } else
        for (int al = 0; al < na; ++al) {
            const float *t = Tinv8 + 8 * (size_t)(ac + al);   // wave-uniform: scalar loads
            const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
            const int first = __builtin_amdgcn_readfirstlane(meta[2 * al]);
            const int cls = __builtin_amdgcn_readfirstlane(meta[2 * al + 1]);
            const float xa = t0 * fx, ya = t3 * fx;
            const vec_t *seg = reinterpret_cast<const vec_t *>(lds) + al * kBSegPitch;
            const float *grow = src + (size_t)al * g.PW;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const float x = (xa + t1 * fy[k]) + t2;
                const float y = (ya + t4 * fy[k]) + t5;
                const float xf = floorf(x), yf = floorf(y), xc = xf + 1.0f, yc = yf + 1.0f;
                vec_t a0, a1;
                if (cls != 2) {     // the segment holds both taps; cells off the detector are zeros
                    const int ix = cvt_flr(x) - first;
                    a0 = seg[ix];
                    a1 = seg[ix + 1];
                } else {
                    const bool ok0 = xf >= 0.0f && xf < (float)g.PW, ok1 = xc >= 0.0f && xc < (float)g.PW;
                    const int i0 = ok0 ? (int)xf : 0, i1 = ok1 ? (int)xc : 0;
#pragma unroll
                    for (int n = 0; n < NS; ++n) {
                        const float p0 = ok0 ? grow[soff[n] + i0] : 0.0f, p1 = ok1 ? grow[soff[n] + i1] : 0.0f;
                        if constexpr (NS == 1) { a0 = p0; a1 = p1; } else { a0[n] = p0; a1[n] = p1; }
                    }
                }
                const vec_t h = (xc - x) * a0 + (x - xf) * a1;
                const vec_t v_yf = (yf >= 0.0f && yf < (float)g.PH) ? h : vec_t(0.0f);
                const vec_t v_yc = (yc >= 0.0f && yc < (float)g.PH) ? h : vec_t(0.0f);
                acc[k] += (yc - y) * v_yf + (y - yf) * v_yc;
            }
        }
    }
    if (c < g.W) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int r = r0 + k * nwaves;
            if (r < g.H) {
#pragma unroll
                for (int n = 0; n < NS; ++n)
                    if (s + n < g.S) {
                        float av;
                        if constexpr (NS == 1) av = acc[k]; else av = acc[k][n];
                        gimg[((size_t)(s + n) * g.H + r) * g.W + c] = av;
                    }
            }
        }
    }
}

// ---- backward, exact transpose -------------------------------------------------------------------------------------------------
//
// The forward adds, for every canvas sample (a, i, j), four weighted taps into sino[a][j]; its transpose adds
// w(a, i, j -> pixel) * g[a][j] into each of the four pixels.  Round 1 did that as a scatter: four atomic adds per sample into an
// LDS image (global atomics at 512 x 512), order not fixed.  Here it is a GATHER through an inverse plan: for a rotation the
// samples that touch a pixel lie within sqrt(2) of its inverse-rotated position, i.e. in at most THREE consecutive detector
// bins, so the plan stores, per (angle, pixel), the first of those bins and the pixel's summed weight in each of the three --
//     W_k = sum over canvas rows i, ascending, of wy * wx of sample (a, i, first + k),
// wy, wx the forward's own fp32 weights ((yc - y) or (y - yf), (xc - x) or (x - xf)) evaluated from the forward's own
// coordinates -- and the backward is
//     gimg[pixel] = sum over angles, ascending, of ((W_0 g[first] + W_1 g[first + 1]) + W_2 g[first + 2]):
// no atomics at any size, equal bits run to run, <= 1e-5 of the oracle's in-order scatter (the weights are summed before
// they meet g, the scatter multiplies first), and the transpose of the forward to fp32 rounding.  16 bytes per (angle, pixel):
// 5.2 MB for 128 x 128 x 20 angles, 377 MB for 512 x 512 x 90.  The builder scans the 5 x 5 samples around the pixel's inverse
// position and raises the overflow word when the touching bins do not fit three (the rows are not a rotation): the caller then
// keeps the scatter kernel.
constexpr int kXSegBins = 80, kXSegPitch = kXSegBins + 1;

template <int INTERP>
__global__ __launch_bounds__(64) void rotate_exact_bilin_plan_kernel(RotGeom g, const float *__restrict__ T8,
                                                                     const float *__restrict__ Tinv8, float4 *__restrict__ plan,
                                                                     int *__restrict__ overflow)
{
    const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y, a = blockIdx.z;
    if (c >= g.W) return;
    const float *ti = Tinv8 + 8 * a, *t = T8 + 8 * a;
    const int X = c + g.px, Y = r + g.py;
    const float fx = (float)X, fy = (float)Y;
    const float j0 = (ti[0] * fx + ti[1] * fy) + ti[2], i0 = (ti[3] * fx + ti[4] * fy) + ti[5];
    int jc = 0, ic = 0;
    if (fabsf(j0) < 1.0e7f && fabsf(i0) < 1.0e7f) {
        jc = (int)__builtin_roundf(j0);
        ic = (int)__builtin_roundf(i0);
    }
    float wb[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    unsigned touched = 0;
    for (int i = ic - 2; i <= ic + 2; ++i) {
        if ((unsigned)i >= (unsigned)g.PH) continue;
#pragma unroll
        for (int dj = 0; dj < 5; ++dj) {
            const int j = jc - 2 + dj;
            if ((unsigned)j >= (unsigned)g.PW) continue;
            const float x = (t[0] * (float)j + t[1] * (float)i) + t[2];
            const float y = (t[3] * (float)j + t[4] * (float)i) + t[5];
            if (!(fabsf(x) < 1.0e7f && fabsf(y) < 1.0e7f)) continue;
            if constexpr (INTERP == CTPVAE_NEAREST) {
                // the nearest forward's one tap (round half away from zero): weight 1; a pixel is the tap of at most two samples
                if ((int)__builtin_roundf(x) != X || (int)__builtin_roundf(y) != Y) continue;
                wb[dj] += 1.0f;
            } else {
                const float xf = floorf(x), yf = floorf(y), xc = xf + 1.0f, yc = yf + 1.0f;
                const int ixf = (int)xf, iyf = (int)yf;
                float wx, wy;
                if (ixf == X) wx = xc - x; else if (ixf + 1 == X) wx = x - xf; else continue;
                if (iyf == Y) wy = yc - y; else if (iyf + 1 == Y) wy = y - yf; else continue;
                wb[dj] += wy * wx;
            }
            touched |= 1u << dj;
        }
    }
    int lo = 2, hi = 2;
    if (touched) {
        lo = __ffs(touched) - 1;
        hi = 31 - __clz(touched);
    }
    if (hi - lo > 2) atomicOr(overflow, 1);
    float w3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = 0.0f;
#pragma unroll
        for (int dj = 0; dj < 5; ++dj) v = (dj == lo + k) ? wb[dj] : v;
        w3[k] = v;
    }
    plan[((size_t)a * g.H + r) * g.W + c] = make_float4(w3[0], w3[1], w3[2], __int_as_float(jc - 2 + lo));
}

template <int PPT, int NS>
__global__ __launch_bounds__(PPT == 1 ? 1024 : (PPT == 2 ? 512 : 256)) void rotate_bwd_exact_bilin_kernel(const float *__restrict__ gsino, RotGeom g,
                                                                     const float *__restrict__ Tinv8, int chunk_a,
                                                                     const float4 *__restrict__ plan, float *__restrict__ gimg)
{
    typedef typename PixVec<NS>::type vec_t;
    constexpr int SHIFT = BilinCell<NS>::kShift, CELL = BilinCell<NS>::kBytes;
    // [chunk_a][kXSegPitch] cells (cell kXSegBins of every segment is zero), then per angle (first bin, class) ints
    extern __shared__ float lds[];
    int *meta = reinterpret_cast<int *>(lds + chunk_a * kXSegPitch * NS);
    const int s = blockIdx.z * NS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int c = blockIdx.x * 64 + lane, cl = min(c, g.W - 1);
    const int r0 = blockIdx.y * (nwaves * PPT) + wave;   // rows r0, r0 + nwaves, ...
    const float X0 = (float)(blockIdx.x * 64 + g.px), X1 = X0 + (float)min(63, g.W - 1 - (int)blockIdx.x * 64);
    const float Y0 = (float)(blockIdx.y * (nwaves * PPT) + g.py), Y1 = Y0 + (float)min(nwaves * PPT - 1, g.H - 1 - (int)blockIdx.y * (nwaves * PPT));
    const int lds_base = (int)(uintptr_t)(lds_cptr)lds;
    size_t soff[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) soff[n] = (size_t)(min(s + n, g.S - 1) - s) * g.A * g.PW;
    vec_t acc[PPT];
    size_t prow[PPT];                                    // plan index of (angle 0, this lane's k-th pixel)
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        acc[k] = vec_t(0.0f);
        prow[k] = (size_t)min(r0 + k * nwaves, g.H - 1) * g.W + cl;
    }
    const size_t pstride = (size_t)g.H * g.W;

    for (int ac = 0; ac < g.A; ac += chunk_a) {
        const int na = min(chunk_a, g.A - ac);
        if (ac > 0) __syncthreads();
        for (int al = threadIdx.x; al < na; al += blockDim.x) {
            const float *t = Tinv8 + 8 * (size_t)(ac + al);
            const float t0 = t[0], t1 = t[1], t2 = t[2];
            const float xa = (t0 * X0 + t1 * Y0) + t2, xb = (t0 * X1 + t1 * Y0) + t2;
            const float xc = (t0 * X0 + t1 * Y1) + t2, xd = (t0 * X1 + t1 * Y1) + t2;
            const float xmin = fminf(fminf(xa, xb), fminf(xc, xd)), xmax = fmaxf(fmaxf(xa, xb), fmaxf(xc, xd));
            int cls = 0, first = 0;
            // a pixel's three bins lie in [floor(j0) - 1, floor(j0) + 3], j0 its inverse position: the segment holds them all
            if (!(xmax - xmin <= (float)(kXSegBins - 8)) || !(fabsf(xmin) < 1.0e6f)) cls = 2;
            else first = (int)floorf(xmin) - 2;
            meta[2 * al] = first;
            meta[2 * al + 1] = cls;
        }
        __syncthreads();
        const float *src = gsino + ((size_t)s * g.A + ac) * g.PW;
        {
            constexpr int U = NS == 4 ? 2 : 8 / NS;
            const int ncell = na * kXSegPitch;
            for (int p0 = threadIdx.x; p0 < ncell; p0 += U * blockDim.x) {
                vec_t v[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = min(p0 + u * (int)blockDim.x, ncell - 1);
                    const int al = p / kXSegPitch, q = p - al * kXSegPitch;
                    const int j = meta[2 * al] + q;
                    ok[u] = q < kXSegBins && (unsigned)j < (unsigned)g.PW && meta[2 * al + 1] == 0;
                    const float *cell = src + al * g.PW + min(max(j, 0), g.PW - 1);
#pragma unroll
                    for (int n = 0; n < NS; ++n) {
                        if constexpr (NS == 1) v[u] = cell[0]; else v[u][n] = cell[soff[n]];
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

        typedef const volatile __attribute__((address_space(3))) vec_t *vptr;
        const float4 *pl = plan + (size_t)ac * pstride;
        for (int al = 0; al < na; ++al) {
            const int first = __builtin_amdgcn_readfirstlane(meta[2 * al]);
            const int cls = __builtin_amdgcn_readfirstlane(meta[2 * al + 1]);
            float4 w[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) w[k] = pl[(size_t)al * pstride + prow[k]];
            if (cls == 0) {
                const int base = (al * kXSegPitch - first) * CELL + lds_base;
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    // the plan's first bin relative to the segment, clamped into it (a bin the segment does not hold -- never, for
                    // a rotation -- would otherwise leave the workgroup's LDS)
                    const int rel = min(max(__float_as_int(w[k].w) - first, 0), kXSegBins - 3);
                    const int addr = ((rel + first) << SHIFT) + base;
                    const vec_t g0 = *(vptr)(uintptr_t)(unsigned)addr;
                    const vec_t g1 = *(vptr)(uintptr_t)(unsigned)(addr + CELL);
                    const vec_t g2 = *(vptr)(uintptr_t)(unsigned)(addr + 2 * CELL);
                    acc[k] += (w[k].x * g0 + w[k].y * g1) + w[k].z * g2;
                }
            } else {   // the tile's span does not fit a segment (not a rotation): bounds-tested reads from global memory
                const float *grow = src + (size_t)al * g.PW;
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int j = __float_as_int(w[k].w);
                    vec_t gv[3];
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const bool ok = (unsigned)(j + e) < (unsigned)g.PW;
#pragma unroll
                        for (int n = 0; n < NS; ++n) {
                            const float pv = ok ? grow[soff[n] + (ok ? j + e : 0)] : 0.0f;
                            if constexpr (NS == 1) gv[e] = pv; else gv[e][n] = pv;
                        }
                    }
                    acc[k] += (w[k].x * gv[0] + w[k].y * gv[1]) + w[k].z * gv[2];
                }
            }
        }
    }
    if (c < g.W) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int r = r0 + k * nwaves;
            if (r < g.H) {
#pragma unroll
                for (int n = 0; n < NS; ++n)
                    if (s + n < g.S) {
                        float av;
                        if constexpr (NS == 1) av = acc[k]; else av = acc[k][n];
                        gimg[((size_t)(s + n) * g.H + r) * g.W + c] = av;
                    }
            }
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------
// behind the image: transform rows, class list + task counter, and the band lists of one chunk of angles (lengths of <= 2048
// bands, the counting sort's 256 offsets, the sorted band numbers)
static size_t bilin_extra_bytes(int A)
{
    return (size_t)A * 8 * sizeof(float) + ((size_t)A + 2) * sizeof(int) + 16 + 2048 + 256 * sizeof(int) + 2048 * sizeof(short);
}
static size_t bilin_img_bytes(int h, int w, bool tiled, int ns) { return (bilin_lds_cells(h, w, tiled, ns) * 4 * ns + 15) & ~(size_t)15; }

// the tile kernel takes this many angles (their transform rows, class list and band lists sit in LDS beside the tile)
bool bilin_fwd_tiles_ok(const TileSpec &ts, int A) { return bilin_img_bytes(ts.th, ts.tw, true, 1) + bilin_extra_bytes(A) <= (size_t)kMaxLdsBytes; }
bool bilin_fwd_whole_geometry(int H, int W) { return bilin_img_bytes(H, W, false, 1) + kBilinLdsReserve <= (size_t)kMaxLdsBytes; }
bool bilin_fwd_whole_ok(int H, int W, int A)
{
    return bilin_fwd_whole_geometry(H, W) && bilin_img_bytes(H, W, false, 1) + bilin_extra_bytes(A) <= (size_t)kMaxLdsBytes;
}

template <int NS, bool TILED>
static int launch_bilin_fwd(const float *img_dev, const RotGeom &g, const TileSpec &ts, const float *T8_dev, float *out_dev,
                            ctpvae_stream_t stream)
{
    const int th = TILED ? ts.th : g.H, tw = TILED ? ts.tw : g.W, nt = TILED ? ts.ntx * ts.nty : 1;
    const size_t img_bytes = bilin_img_bytes(th, tw, TILED, NS), extra = bilin_extra_bytes(g.A);
    CTPVAE_REQUIRE(img_bytes + extra <= (size_t)kMaxLdsBytes, "rotate_fwd (bilinear): a %d x %d unit of %d slices and %d angles does not fit LDS",
                   th, tw, NS, g.A);
    const int groups = ceil_div(g.S, NS), units = groups * nt;
    CTPVAE_REQUIRE(units <= 65535, "rotate_fwd (bilinear): at most 65535 units per launch (got %d)", units);
    const int nb = TILED ? ts.nb : ((g.PW + 63) & ~63);
    const int tasks = g.A * (nb / 64);
    // Task groups per class.  A workgroup of this kernel runs for tens to hundreds of microseconds behind ~6 us of fill and band
    // sort, so the cut is chosen for WHOLE ROUNDS of workgroups on 256 CUs (round 5, second pass; profiles/r05_rounds.txt is the
    // study): rounds(G) x (6 us + the class's tasks / G at ~1.3 us each for a 184-row canvas).  One group per class used to be
    // taken from 128 units on -- 75 pairs x 2 classes = 150 workgroups on 256 CUs (B = 150 x 180 angles: 320 us; with G = 5,
    // 750 workgroups in 3 rounds); 150 pairs = 300 workgroups = two rounds for 1.17 rounds of work (635 us; G = 5: 6 rounds of a
    // fifth).  Near-ties go to fewer groups (every group stages the unit again).
    // (Tiles: the whole-slice constants over-split them -- a tile's walk is ~150 rows, not the canvas's 728: 8 x 512^2 x 90 measured 133 us
    // with G = 4 against 110 -- so they get the rule with their own constants, below.)
    int G = std::min(std::max(1, 256 / (2 * units)), std::max(1, tasks / 8));
    if (TILED) {
        // (tiles, second look -- tools/sweep_tile_rules.py, profiles/r05_tile_rules.txt: the same rounds rule with a tile's own walk
        // length and ~20 us of fill and band sorts per workgroup (12 us took four groups at 8 x 512^2 x 90: 123.5 us against 108.7 with one);
        // 16 x 512^2 x 90 angles: one group 202 us in 1.5 rounds, two 182)
        const double t_task = 1.3 * (double)(ts.th + ts.tw) / 184.0 * (NS == 4 ? 1.25 : 1.0), tc = std::max(1, tasks / 2);
        double best = 0.0;
        for (int c = 1; c <= std::min(16, std::max(1, tasks / 8)); ++c) {
            const double t = std::ceil(2.0 * units * c / 256.0) * (20.0 + t_task * tc / c);
            if (best == 0.0 || t < best * 0.97) best = t, G = c;
        }
    } else {
        const double t_task = 1.3 * (double)g.PH / 184.0 * (NS == 4 ? 1.25 : 1.0), tc = std::max(1, tasks / 2);
        double best = 0.0;
        // (up to a group per four tasks: with row-split walks -- below -- the small launches keep gaining up to 16 groups: 1 .. 12 x 128^2 x
        // 20 angles 14.7 us at 7 groups, 12.3 at 16; tools/sweep_bilin_modes.py, profiles/r05_bilin_fwd_rules.txt)
        for (int c = 1; c <= std::min(16, std::max(1, tasks / 4)); ++c) {
            const double t = std::ceil(2.0 * units * c / 256.0) * (6.0 + t_task * tc / c);
            if (best == 0.0 || t < best * 0.97) best = t, G = c;
        }
    }
    if (knob(kKnobBw) > 0) G = knob(kKnobBw);
    // >= 8 waves: a workgroup's fill is shared by its waves, and two waves per SIMD issue LDS reads and waits under each other's
    // vector instructions (tools/sweep_bilin.py, B = 50 x 128 x 128 x 20 angles, G = 5: 27.1 / 24.3 / 22.1 / 21.8 us at 4 / 6 / 8 / 16)
    int waves = std::min(16, std::max(8, ceil_div(tasks, 2 * G)));
    const bool padded = g.px >= 1 && g.py >= 1;
    // length-sorted band tasks from ~22 (angle, 64-slot block) tasks per workgroup on (the classes hold about half the angles each);
    // measured, B x 128^2 x angles, sorted against plain: 50 x 30 (9 tasks per workgroup) 28.5 / 23.6 us, 50 x 60 (18) 35.9 / 35.9,
    // 50 x 90 (27) 45.1 / 50.6, 100 x 45 (34) 62.3 / 65.0, 50 x 120 (36) 55.5 / 65.5, 10 x 180 (17) 36.8 / 34.0
    // (tiles from ~12 on: their chord profiles are the worse ones -- 8 x 512^2 x 20 angles 46.0 us plain, 42.2 sorted)
    bool sorted = (long long)g.A * (nb / 64) >= (TILED ? 24ll : 44ll) * G;
    if (knob(kKnobBsort) >= 0) sorted = knob(kKnobBsort) != 0;
    // ROW-SPLIT walks for the unsorted launches of whole slices (few tasks per workgroup: the headline shape has 6 on a CU of 4 SIMDs, and
    // a lone wave issues one instruction of ANY kind per ~4.4 cycles -- its LDS reads, waits and scalar steps are not hidden under anything):
    // a task is a run of 32 slots, lanes 32-63 walk the same rays two rows further on, so twice the tasks of half the length
    // -- up to ~14 tasks per workgroup (tools/ab_rsplit.py, plain / row-split in us: 50 x 128^2 x 20 angles [6 tasks per workgroup] 20.97 /
    // 19.05, 5 x 128^2 x 20 [4] 17.2 / 15.5, 3 x 100^2 x 7 16.9 / 12.5, 256 x 64^2 x 20 [10] 25.5 / 22.0, 50 x 128^2 x 30 [9] 23.4 / 23.7,
    // 76 x 128^2 x 20 [10] 26.9 / 26.8, 50 x 128^2 x 45 [13.5] 32.2 / 30.7; 100 x 128^2 x 20 [15] 31.9 / 34.1, 50 x 128^2 x 60 [18] 35.6 / 38.4: with many tasks per SIMD the
    // plain walk's longer stretches and half as many task set-ups win)
    bool rsplit = !TILED && !sorted && tasks <= 28 * G;
    if (knob(kKnobBrsplit) >= 0) rsplit = !TILED && !sorted && knob(kKnobBrsplit) != 0;
    if (rsplit) waves = std::min(16, std::max(8, ceil_div(2 * tasks, 2 * G)));
    if (knob(kKnobWaves) > 0) waves = std::min(16, knob(kKnobWaves));
#ifdef CTPVAE_TUNE_STAMPS
    g_pshape[0] = units, g_pshape[1] = 2 * G, g_pshape[2] = waves, g_pshape[3] = NS, g_pshape[4] = 0, g_pshape[5] = units, g_pshape[6] = 2 * G, g_pshape[7] = units, g_pshape[8] = 2 * G;
#endif
    auto launch = [&](auto kernel) -> int {
        static std::atomic<unsigned long long> attr_set{0}, abs_ok{0};
        CTPVAE_REQUIRE_NO_STATIC_LDS(kernel, "rotate_fwd_bilin_kernel", abs_ok);   // the all-zero block sits at LDS address 0
        CTPVAE_SET_MAX_LDS_ONCE(kernel, attr_set);
        hipLaunchKernelGGL(kernel, dim3(2 * G, units), dim3(64 * waves), img_bytes + extra, (hipStream_t)stream, img_dev, g, ts, T8_dev,
                           (int)(img_bytes / sizeof(float)), out_dev);
        CTPVAE_LAUNCH_CHECK("rotate_fwd_bilin_kernel");
        return CTPVAE_OK;
    };
    if (sorted) return padded ? launch(rotate_fwd_bilin_kernel<NS, TILED, true, true>) : launch(rotate_fwd_bilin_kernel<NS, TILED, false, true>);
    if constexpr (!TILED)
        if (rsplit) return padded ? launch(rotate_fwd_bilin_kernel<NS, false, true, false, true>) : launch(rotate_fwd_bilin_kernel<NS, false, false, false, true>);
    return padded ? launch(rotate_fwd_bilin_kernel<NS, TILED, true, false>) : launch(rotate_fwd_bilin_kernel<NS, TILED, false, false>);
}

// slices per LDS cell: as many as fit beside the transform copy (every one shares the sample's index instructions)
static int bilin_fwd_ns(int S, int h, int w, bool tiled, int A)
{
    int ns = S >= 3 ? 4 : (S == 2 ? 2 : 1);
    if (knob(kKnobBns) == 1 || knob(kKnobBns) == 2 || knob(kKnobBns) == 4) ns = knob(kKnobBns);
    while (ns > 1 && bilin_img_bytes(h, w, tiled, ns) + bilin_extra_bytes(A) > (size_t)kMaxLdsBytes) ns >>= 1;
    return ns;
}

// whole slices in LDS: [S][H][W] -> [S][A][PW]
int bilin_fwd_whole(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px, const float *T8_dev, int A,
                    float *sino_dev, ctpvae_stream_t stream)
{
    const int ns = bilin_fwd_ns(S, H, W, false, A);
    const TileSpec none{};
    return for_slice_chunks(S, std::max(4, std::min(max_slices_per_launch(), 65532) / 4 * 4), [&](int s0, int n) {
        const RotGeom g{n, H, W, PH, PW, py, px, A};
        const float *im = img_dev + (size_t)s0 * H * W;
        float *so = sino_dev + (size_t)s0 * A * PW;
        if (ns == 4) return launch_bilin_fwd<4, false>(im, g, none, T8_dev, so, stream);
        if (ns == 2) return launch_bilin_fwd<2, false>(im, g, none, T8_dev, so, stream);
        return launch_bilin_fwd<1, false>(im, g, none, T8_dev, so, stream);
    });
}

// tiles of a slice larger than LDS: partial sums [S / 4][tiles][A][nb][4] into the workspace; the caller runs the reduce pass
int bilin_fwd_tiles(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px, const float *T8_dev, int A,
                    const TileSpec &ts, float *workspace_dev, ctpvae_stream_t stream)
{
    const RotGeom g{S, H, W, PH, PW, py, px, A};
    const int ns = bilin_fwd_ns(S, ts.th, ts.tw, true, A);
    if (ns == 4) return launch_bilin_fwd<4, true>(img_dev, g, ts, T8_dev, workspace_dev, stream);
    if (ns == 2) return launch_bilin_fwd<2, true>(img_dev, g, ts, T8_dev, workspace_dev, stream);
    return launch_bilin_fwd<1, true>(img_dev, g, ts, T8_dev, workspace_dev, stream);
}

// cotangents [S][A][PW] -> gradient images [S][H][W], TensorFlow-compatible, bilinear
int bilin_bwd_tfcompat(const float *gsino_dev, int S, int A, int PH, int PW, const float *Tinv8_dev, int H, int W, int py, int px,
                       float *gimg_dev, ctpvae_stream_t stream)
{
    return for_slice_chunks(S, std::max(4, std::min(max_slices_per_launch(), 65532) / 4 * 4), [&](int s0, int n) {
        const RotGeom g{n, H, W, PH, PW, py, px, A};
        const float *gs = gsino_dev + (size_t)s0 * A * PW;
        float *gi = gimg_dev + (size_t)s0 * H * W;
        // Slices per cell: the most sharing that still leaves ~200 workgroups.  Rows per lane: TWO -- eight waves on a workgroup's
        // 64 x 16 pixels and its segments (four waves of four rows, round 5's first form, left one or two waves on a SIMD, and a lone
        // wave issues one instruction of any kind per ~4.4 cycles: tools/sweep_bilin_bwd.py, B x 128^2 x angles, us with 4 / 2 / 1
        // rows per lane: 50 x 20 [four slices per cell] 10.3 / 8.6 / 8.6, 50 x 180 77.6 / 59.9 / 53.7, 100 x 20 14.1 / 12.8 / 14.3,
        // 400 x 180 286 / 271 / -, 32 x 512^2 x 90 163.8 / 160.0 / 208; 25 x 20 [pairs] 9.0 / 7.1 / 7.1, 12 x 20 [singles] 7.8 / 6.8 /
        // 6.9) -- ONE row, sixteen waves, for launches of one round of workgroups at many angles.
        const long long tiles4 = (long long)ceil_div(W, 64) * ceil_div(H, 16);
        int ns = n >= 3 ? 4 : (n == 2 ? 2 : 1), ppt = 2;
        while (ns > 1 && ceil_div(n, ns) * tiles4 < 160) ns >>= 1;   // (20 x 128^2 x 90 angles: pairs, 160 workgroups, 20.5 us; singles, 320: 27)
        if (ceil_div(n, ns) * tiles4 <= 256 && A >= 32) ppt = 1;
        if (knob(kKnobSegNs) == 1 || knob(kKnobSegNs) == 2 || knob(kKnobSegNs) == 4) ns = std::min(knob(kKnobSegNs), n >= 3 ? 4 : n);
        if (ns == 3) ns = 2;
        if (knob(kKnobSegPpt) == 1 || knob(kKnobSegPpt) == 2 || knob(kKnobSegPpt) == 4 || knob(kKnobSegPpt) == 8) ppt = knob(kKnobSegPpt);
        if (ns == 4 && ppt == 8) ppt = 4;         // (eight rows of four slices: the taps of an angle alone are 64 registers)
        // ~48 KiB of segments per chunk of angles: three 4-wave workgroups per CU
        int chunk_a = std::max(1, std::min(A, (48 * 1024) / (kBSegPitch * 4 * ns + 40)));
        if (knob(kKnobSegChunk) > 0) chunk_a = std::max(1, std::min(A, std::min(knob(kKnobSegChunk), chunk_a)));
        const size_t shmem = (((size_t)chunk_a * (kBSegPitch * ns + 2) + 3) & ~(size_t)3) * 4 + (size_t)chunk_a * 32 + 16;
        // (two rows per lane: EIGHT waves on the same 64 x 16 pixels and the same segments)
        const dim3 grid(ceil_div(W, 64), ceil_div(H, ppt <= 2 ? 16 : 4 * ppt), ceil_div(n, ns)), block(ppt == 1 ? 1024 : (ppt == 2 ? 512 : 256));
        auto launch = [&](auto kernel) -> int {
            hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)stream, gs, g, Tinv8_dev, chunk_a, gi);
            CTPVAE_LAUNCH_CHECK("rotate_bwd_bilin_seg_kernel");
            return CTPVAE_OK;
        };
        if (ppt == 1) return ns == 4 ? launch(rotate_bwd_bilin_seg_kernel<1, 4>) : (ns == 2 ? launch(rotate_bwd_bilin_seg_kernel<1, 2>) : launch(rotate_bwd_bilin_seg_kernel<1, 1>));
        if (ppt == 2) return ns == 4 ? launch(rotate_bwd_bilin_seg_kernel<2, 4>) : (ns == 2 ? launch(rotate_bwd_bilin_seg_kernel<2, 2>) : launch(rotate_bwd_bilin_seg_kernel<2, 1>));
        if (ns == 4) return launch(rotate_bwd_bilin_seg_kernel<4, 4>);
        if (ns == 2) return ppt == 8 ? launch(rotate_bwd_bilin_seg_kernel<8, 2>) : launch(rotate_bwd_bilin_seg_kernel<4, 2>);
        return ppt == 8 ? launch(rotate_bwd_bilin_seg_kernel<8, 1>) : launch(rotate_bwd_bilin_seg_kernel<4, 1>);
    });
}

// ---- exact bilinear adjoint through the inverse plan ----
static size_t exact_bilin_plan_floats(int H, int W, int A) { return (size_t)A * H * W * 4; }

int exact_bilin_plan_build(const float *T8_dev, const float *Tinv8_dev, int A, int H, int W, int PH, int PW, int py, int px,
                           int interp, void *plan_dev, ctpvae_stream_t stream)
{
    const RotGeom g{1, H, W, PH, PW, py, px, A};
    int *flag = reinterpret_cast<int *>(reinterpret_cast<float *>(plan_dev) + exact_bilin_plan_floats(H, W, A));
    CTPVAE_HIP(hipMemsetAsync(flag, 0, 256, (hipStream_t)stream));
    for (int a0 = 0; a0 < A; a0 += 65535) {
        const int na = std::min(65535, A - a0);
        RotGeom gc = g;
        gc.A = na;
        if (interp == CTPVAE_NEAREST)
            hipLaunchKernelGGL(rotate_exact_bilin_plan_kernel<CTPVAE_NEAREST>, dim3(ceil_div(W, 64), H, na), dim3(64), 0,
                               (hipStream_t)stream, gc, T8_dev + 8 * (size_t)a0, Tinv8_dev + 8 * (size_t)a0,
                               reinterpret_cast<float4 *>(plan_dev) + (size_t)a0 * H * W, flag);
        else
            hipLaunchKernelGGL(rotate_exact_bilin_plan_kernel<CTPVAE_BILINEAR>, dim3(ceil_div(W, 64), H, na), dim3(64), 0,
                               (hipStream_t)stream, gc, T8_dev + 8 * (size_t)a0, Tinv8_dev + 8 * (size_t)a0,
                               reinterpret_cast<float4 *>(plan_dev) + (size_t)a0 * H * W, flag);
        CTPVAE_LAUNCH_CHECK("rotate_exact_bilin_plan_kernel");
    }
    return CTPVAE_OK;
}

int exact_bilin_bwd(const float *gsino_dev, int S, int A, int PH, int PW, const float *Tinv8_dev, int H, int W, int py, int px,
                    const void *plan_dev, float *gimg_dev, ctpvae_stream_t stream)
{
    return for_slice_chunks(S, std::max(4, std::min(max_slices_per_launch(), 65532) / 4 * 4), [&](int s0, int n) {
        const RotGeom g{n, H, W, PH, PW, py, px, A};
        const float *gs = gsino_dev + (size_t)s0 * A * PW;
        float *gi = gimg_dev + (size_t)s0 * H * W;
        // Every slice of a cell shares the 16-byte plan word of its (angle, pixel): four slices per cell from ~160 workgroups on, pairs
        // below.  Rows per lane: the kernel waits for its plan words, so the more waves the better while the plan is cache-resident
        // -- ONE row per lane (sixteen waves on 64 x 16 pixels) for a launch of one round of workgroups, TWO (eight waves) above;
        // four rows (round 5's first form) where the plan streams from memory (512 x 512 x 90 angles: 377 MB).
        // tools/sweep_bilin_bwd.py with BWD=exact, us at 4 / 2 / 1 rows per lane: 50 x 128^2 x 20 angles [four slices] 14.0 / 10.3 / 8.9,
        // 25 x 20 [pairs] 13.7 / 7.9 / 7.5, 5 x 20 [pairs] 13.5 / 7.6 / 7.2, 50 x 180 [four] 131 / 97 / 80, 100 x 20 17.1 / 14.0 / 15.3,
        // 400 x 180 399 / 323 / -, 32 x 512^2 x 90 302 / 340 / 566.
        const long long tiles4 = (long long)ceil_div(W, 64) * ceil_div(H, 16);
        int ns = n >= 3 ? 4 : (n == 2 ? 2 : 1), ppt = 2;
        // (second look, library against the best forced launch over 19 shapes and other image sizes: sixteen-wave workgroups only while
        // they make ONE round -- 50 x 160^2 x 90 angles, 390 of them: 87 us against 57 with eight waves; pairs in 400 sixteen-wave
        // workgroups were 6 % ahead at 50 x 128^2 x 180 and are not worth a rule of their own -- and the streaming plan's four rows per
        // lane pay from ~700 workgroups on -- 8 x 512^2 x 90 (512): 92.9 us with four rows, 82.0 with two)
        while (ns > 2 && ceil_div(n, ns) * tiles4 < 160) ns >>= 1;   // (50 x 100^2 x 20 angles: quads in 182 workgroups 8.8 us, pairs in 350: 10.1)
        if (ceil_div(n, ns) * tiles4 <= 256) ppt = 1;
        if ((long long)A * H * W * 16 > (128ll << 20)) ppt = ceil_div(n, ns) * tiles4 >= 700 ? 4 : 2;   // (one row: 145 us at 8 x 512^2 x 90; 12 slices, 768 workgroups: 99 us with four rows, 109 with two)
        if (knob(kKnobSegNs) == 1 || knob(kKnobSegNs) == 2 || knob(kKnobSegNs) == 4) ns = std::min(knob(kKnobSegNs), n >= 3 ? 4 : n);
        if (ns == 3) ns = 2;
        if (knob(kKnobSegPpt) == 1 || knob(kKnobSegPpt) == 2 || knob(kKnobSegPpt) == 4 || knob(kKnobSegPpt) == 8) ppt = knob(kKnobSegPpt);
        if (ns == 4 && ppt == 8) ppt = 4;
        int chunk_a = std::max(1, std::min(A, (48 * 1024) / (kXSegPitch * 4 * ns + 8)));
        if (knob(kKnobSegChunk) > 0) chunk_a = std::max(1, std::min(A, std::min(knob(kKnobSegChunk), chunk_a)));
        const size_t shmem = (size_t)chunk_a * (kXSegPitch * ns * 4 + 8) + 16;
        const dim3 grid(ceil_div(W, 64), ceil_div(H, ppt <= 2 ? 16 : 4 * ppt), ceil_div(n, ns)), block(ppt == 1 ? 1024 : (ppt == 2 ? 512 : 256));
        auto launch = [&](auto kernel) -> int {
            hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)stream, gs, g, Tinv8_dev, chunk_a,
                               reinterpret_cast<const float4 *>(plan_dev), gi);
            CTPVAE_LAUNCH_CHECK("rotate_bwd_exact_bilin_kernel");
            return CTPVAE_OK;
        };
        if (ppt == 1) return ns == 4 ? launch(rotate_bwd_exact_bilin_kernel<1, 4>) : (ns == 2 ? launch(rotate_bwd_exact_bilin_kernel<1, 2>) : launch(rotate_bwd_exact_bilin_kernel<1, 1>));
        if (ppt == 2) return ns == 4 ? launch(rotate_bwd_exact_bilin_kernel<2, 4>) : (ns == 2 ? launch(rotate_bwd_exact_bilin_kernel<2, 2>) : launch(rotate_bwd_exact_bilin_kernel<2, 1>));
        if (ns == 4) return launch(rotate_bwd_exact_bilin_kernel<4, 4>);
        if (ns == 2) return ppt == 8 ? launch(rotate_bwd_exact_bilin_kernel<8, 2>) : launch(rotate_bwd_exact_bilin_kernel<4, 2>);
        return ppt == 8 ? launch(rotate_bwd_exact_bilin_kernel<8, 1>) : launch(rotate_bwd_exact_bilin_kernel<4, 1>);
    });
}

}  // namespace ctpvae

using namespace ctpvae;

extern "C" {

long long ctpvae_rotate_exact_wplan_bytes(int H, int W, int A)
{
    if (H <= 0 || W <= 0 || A <= 0) return fail(CTPVAE_EINVAL, "rotate_exact_wplan_bytes: bad sizes");
    return (long long)(exact_bilin_plan_floats(H, W, A) * sizeof(float)) + 256;
}

int ctpvae_rotate_exact_wplan_build_f32(const float *T8_dev, const float *Tinv8_dev, int A, int H, int W, int PH, int PW,
                                        int py, int px, int interp, void *plan_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(T8_dev && Tinv8_dev && plan_dev, "rotate_exact_wplan_build: null pointer");
    CTPVAE_REQUIRE(interp == CTPVAE_NEAREST || interp == CTPVAE_BILINEAR, "rotate_exact_wplan_build: unknown interpolation %d", interp);
    if (int rc = check_plan_geom("rotate_exact_wplan_build", H, W, PH, PW, py, px, A)) return rc;
    CTPVAE_REQUIRE(H <= 65535, "rotate_exact_wplan_build: at most 65535 rows (got %d)", H);
    return exact_bilin_plan_build(T8_dev, Tinv8_dev, A, H, W, PH, PW, py, px, interp, plan_dev, stream);
}

int ctpvae_rotate_exact_wplan_overflowed(const void *plan_dev, int H, int W, int A, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(plan_dev && H > 0 && W > 0 && A > 0, "rotate_exact_wplan_overflowed: bad arguments");
    int flag = 0;
    CTPVAE_HIP(hipMemcpyAsync(&flag, reinterpret_cast<const float *>(plan_dev) + exact_bilin_plan_floats(H, W, A), sizeof(int),
                              hipMemcpyDeviceToHost, (hipStream_t)stream));
    CTPVAE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return flag != 0;
}

int ctpvae_rotate_bwd_exact_wplan_f32(const float *gsino_dev, int S, int A, int PH, int PW, const float *Tinv8_dev, int H,
                                                 int W, int py, int px, const void *plan_dev, float *gimg_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(gsino_dev && Tinv8_dev && plan_dev && gimg_dev && S > 0, "rotate_bwd_exact_wplan: null pointer or empty batch");
    if (int rc = check_plan_geom("rotate_bwd_exact_wplan", H, W, PH, PW, py, px, A)) return rc;
    return exact_bilin_bwd(gsino_dev, S, A, PH, PW, Tinv8_dev, H, W, py, px, plan_dev, gimg_dev, stream);
}

}  // extern "C"
