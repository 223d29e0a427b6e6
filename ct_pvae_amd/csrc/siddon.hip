// siddon.hip -- TomoPy-style ray-driven projector (a7): exact ray / pixel-grid intersection lengths.
//
// Follows libtomo's project(): for every (angle p, detector bin d) the ray's crossings with the
// horizontal grid lines (list "a") and with the vertical grid lines (list "b") are merged by x,
// consecutive crossings give a segment length and, from the segment midpoint, a pixel.  libtomo
// materialises both lists and the merged list; here one lane owns one ray and performs the merge on
// the fly with two cursors, evaluating exactly the same fp32 expressions, so no per-ray arrays exist
// and the slice is read from LDS.  Compiled with -ffp-contract=off; '/' and sqrtf are correctly
// rounded (hipcc default), so the result equals the CPU restatement bit for bit.
//
// In this file (tomopy.project / tomopy.recon(algorithm = 'fbp' | 'sirt') behind create_sinogram and iradon_all,
// ctvae/helper_functions.py:33-38,489-516):
//   siddon_fwd_kernel            the walk, one or two slices per workgroup in LDS (one- and two-slice calls)
//   siddon_fwd_packed_kernel     the same walk for 4 / 8 slices interleaved per pixel in global memory (batches); optional
//                                SIRT store (meas - A x) / sum dist^2
//   siddon_bwd_gather_kernel     the transpose as a pixel-driven gather with libtomo's arithmetic and order: bit-equal to the
//                                ray-driven accumulation (+ ray table, slow-path flags, the degenerate rays' ray-driven pass);
//                                optional SIRT store x += A^T upd / sum dist
//   siddon_rownorm_kernel        sum dist^2 of every ray
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "lds_stage.h"

namespace ctpvae {

struct SidGeom {
    int oy, ox, oz, dt, dx;
    float mov;
};

// One ray of libtomo's project(): calc_coords -> trim_coords -> sort_intersections -> calc_dist, as a walk with two cursors
// (no per-ray arrays).  `segment(ix, iy, dist)` is called for every segment n = 0 .. csize-2 in libtomo's order with the pixel
// its midpoint falls in and its length -- the forward projector adds model[pixel] * dist, the back-projector adds
// data * dist into the pixel, SIRT's row norm adds dist * dist: the SAME fp32 expressions in all three, so the
// back-projector is the forward's transpose by construction.
struct SidRayLine {
    float srcx, srcy, slope, islope;
};
__device__ __forceinline__ SidRayLine siddon_ray_line(const SidGeom &g, float sin_p, float cos_p, int d)
{
    const float xi = (float)(-g.ox - g.oz);
    const float yi = (1 - g.dx) / 2.0f + d + g.mov;
    const float srcx = xi * cos_p - yi * sin_p, srcy = xi * sin_p + yi * cos_p;
    const float detx = -xi * cos_p - yi * sin_p, dety = -xi * sin_p + yi * cos_p;
    return {srcx, srcy, (srcy - dety) / (srcx - detx), (srcx - detx) / (srcy - dety)};
}
// A ray that lies ON a grid line of a direction it is (numerically) parallel to: with a slope of ~1e7 the crossings with the
// coinciding line family land at erratic positions and libtomo's merge gives long zig-zag segments whose midpoints revisit
// pixels (odd grids under the even padded detector, theta = pi/2 or 0 exactly).  The pixel-driven back-projector cannot
// predict those; such rays are walked as libtomo walks them (siddon_bwd_degenerate_kernel).
__device__ __forceinline__ bool siddon_ray_is_degenerate(const SidGeom &g, const SidRayLine &r)
{
    bool deg = false;
    if (!(fabsf(r.slope) <= 1.0e6f)) {          // parallel to the x = gridx[n] lines
        const float f = r.srcx - (-g.ox * 0.5f);
        deg = deg || fabsf(f - rintf(f)) < 1.0e-3f;
    }
    if (!(fabsf(r.islope) <= 1.0e6f)) {         // parallel to the y = gridy[n] lines
        const float f = r.srcy - (-g.oz * 0.5f);
        deg = deg || fabsf(f - rintf(f)) < 1.0e-3f;
    }
    return deg;
}

template <class F>
__device__ __forceinline__ void siddon_walk_ray(const SidGeom &g, float sin_p, float cos_p, int quadrant, int d, F &&segment)
{
    const int ox = g.ox, oz = g.oz;
    const float gx0 = -ox * 0.5f, gy0 = -oz * 0.5f;  // gridx[n] = gx0 + n, gridy[n] = gy0 + n
    const float gx_gt = gx0 + 0.01f, gx_le = (gx0 + ox) - 0.01f;
    const float gy_gt = gy0 + 0.01f, gy_le = (gy0 + oz) - 0.01f;
    const float hx = ox * 0.5f, hz = oz * 0.5f;
    const float xi = (float)(-ox - oz);
    const float yi = (1 - g.dx) / 2.0f + d + g.mov;
    const float srcx = xi * cos_p - yi * sin_p, srcy = xi * sin_p + yi * cos_p;
    const float detx = -xi * cos_p - yi * sin_p, dety = -xi * sin_p + yi * cos_p;
    const float slope = (srcy - dety) / (srcx - detx);
    const float islope = (srcx - detx) / (srcy - dety);

    // list a: crossings with y = gridy[n], x = coordx(n) = islope * (gridy[n] - srcy) + srcx, kept iff
    // gx_gt <= x <= gx_le; list b: crossings with x = gridx[n], y = coordy(n), kept iff gy_gt <= y <= gy_le.
    // Every fp32 step of coord(n) is monotone in n, so the kept n form one contiguous run whose ends are found by
    // bisection on the SAME expression (libtomo scans all n; same set).  A non-finite slope (a ray exactly along
    // a grid direction) keeps the scan.
    int a_lo = 0, a_cnt = 0, b_lo = 0, b_cnt = 0;
    auto kept_run = [](float sl, float g0, float src_u, float src_v, float lo, float hi, int N, int &first, int &cnt) {
        auto coord = [&](int n) { return sl * ((g0 + n) - src_u) + src_v; };
        first = 0;
        cnt = 0;
        if (!(fabsf(sl) <= 3.0e38f)) {   // inf / NaN
            for (int n = 0; n <= N; ++n) {
                const float c = coord(n);
                if (c >= lo && c <= hi) {
                    if (cnt == 0) first = n;
                    ++cnt;
                }
            }
            return;
        }
        const bool inc = coord(0) <= coord(N);
        // smallest n whose coordinate has entered [lo, hi] from its low side, smallest n that has left it
        int l = 0, r = N + 1;      // first n with  (inc ? c >= lo : c <= hi)
        while (l < r) {
            const int m = (l + r) >> 1;
            const float c = coord(m);
            if (inc ? c >= lo : c <= hi) r = m; else l = m + 1;
        }
        const int n_in = l;
        l = n_in, r = N + 1;       // first n >= n_in with (inc ? c > hi : c < lo)
        while (l < r) {
            const int m = (l + r) >> 1;
            const float c = coord(m);
            if (inc ? c > hi : c < lo) r = m; else l = m + 1;
        }
        first = n_in;
        cnt = l - n_in;
    };
    kept_run(islope, gy0, srcy, srcx, gx_gt, gx_le, oz, a_lo, a_cnt);
    kept_run(slope, gx0, srcx, srcy, gy_gt, gy_le, ox, b_lo, b_cnt);
    const int csize = a_cnt + b_cnt;
    const int k_begin = 0, k_end = csize, ia0 = 0;
    const int ib0 = k_begin - ia0;
    // The merge of libtomo's two sorted lists, with two cursors.  List a runs over its kept n upwards in
    // quadrant 1 and downwards otherwise; gridy[n] = gy0 + n is exact in fp32, so a running +-1.0f gives the same
    // values as int -> float.  An exhausted list shows +inf as its key: "a_key < b_key" then reproduces
    // sort_intersections' choice (a first only if strictly smaller; the other list once one has run out).
    const float kInf = __builtin_inff();
    const float da = quadrant ? 1.0f : -1.0f;
    float a_y = gy0 + (float)(quadrant ? a_lo + ia0 : a_lo + a_cnt - 1 - ia0);
    float a_x = islope * (a_y - srcy) + srcx;
    int a_rem = a_cnt - ia0;
    float a_key = a_rem > 0 ? a_x : kInf;
    float b_x = gx0 + (float)(b_lo + ib0);
    float b_y = slope * (b_x - srcx) + srcy;
    int b_rem = b_cnt - ib0;
    float b_key = b_rem > 0 ? b_x : kInf;
    float px_prev = 0.0f, py_prev = 0.0f;
    for (int k = k_begin; k < k_end; ++k) {
        const bool take_a = a_key < b_key;
        const float cx = take_a ? a_x : b_x;
        const float cy = take_a ? a_y : b_y;
        {   // advance the list that was taken (selects, not a branch: the lanes of a wave disagree all the time)
            const float na_y = a_y + da, na_x = islope * (na_y - srcy) + srcx;
            const float nb_x = b_x + 1.0f, nb_y = slope * (nb_x - srcx) + srcy;
            const int na_rem = a_rem - 1, nb_rem = b_rem - 1;
            const float na_key = na_rem > 0 ? na_x : kInf, nb_key = nb_rem > 0 ? nb_x : kInf;
            a_y = take_a ? na_y : a_y;
            a_x = take_a ? na_x : a_x;
            a_key = take_a ? na_key : a_key;
            a_rem = take_a ? na_rem : a_rem;
            b_x = take_a ? b_x : nb_x;
            b_y = take_a ? b_y : nb_y;
            b_key = take_a ? b_key : nb_key;
            b_rem = take_a ? b_rem : nb_rem;
        }
        if (k > k_begin) {
            const float diffx = cx - px_prev, diffy = cy - py_prev;
            const float dist = sqrtf(diffx * diffx + diffy * diffy);
            const float midx = (cx + px_prev) * 0.5f, midy = (cy + py_prev) * 0.5f;
            // libtomo: i1 = (int)x1; indx = i1 - (i1 > x1)  ==  floor(x1)
            const int indx = (int)floorf(midx + hx), indy = (int)floorf(midy + hz);
            // libtomo reads model[indy + indx*oz] unchecked; midpoints lie strictly inside the grid
            segment(min(max(indx, 0), ox - 1), min(max(indy, 0), oz - 1), dist);
        }
        px_prev = cx;
        py_prev = cy;
    }
}

// NS = 2: two slices per workgroup, interleaved as float2 in LDS -- the crossings, segment lengths and pixel indices of
// a ray depend on the geometry only, so one walk serves both slices (the loop is VALU-bound on exactly that arithmetic).
// What a forward kernel stores for ray-sum `sim` at output element o (ray r = its (angle, bin)):
//   mode 0: sim                                                      (tomopy.project, helper_functions.py:33-38)
//   mode 1: (meas - sim) / w where w = sum dist^2 != 0, else 0       (SIRT's update factor, libtomo sirt.c)
//   mode 2: (data + w (sim - meas)) / (1 + w), data updated in place (round 4: the dual step of the TV stand-in's preconditioned
//           Chambolle-Pock iteration, prox of 1/2 |. - b|^2's conjugate with step w = 1 / (the ray's sum of dist); recon.py _tv)
__device__ __forceinline__ float siddon_fwd_store(int mode, float sim, const float *__restrict__ meas, const float *__restrict__ w_ray,
                                                  const float *data, size_t o, size_t r)
{
    if (mode == 0) return sim;
    const float w = w_ray[r];
    if (mode == 1) return w != 0.0f ? (meas[o] - sim) / w : 0.0f;
    return (data[o] + w * (sim - meas[o])) / (1.0f + w);
}
typedef float sid_f32x2 __attribute__((ext_vector_type(2)));
template <int NS> struct SidVec { typedef float type; };
template <> struct SidVec<2> { typedef sid_f32x2 type; };
// meas != NULL (SIRT, libtomo sirt.c): instead of the ray-sum `sim` the kernel stores the ray's update factor
// upd = (meas - sim) / rn2 where rn2 = sum dist^2 != 0, else 0 -- what the back-projector then spreads over the ray.
template <bool USE_LDS, int NS>
__global__ __launch_bounds__(1024) void siddon_fwd_kernel(const float *__restrict__ obj, SidGeom g,
                                                         const float *__restrict__ sin_t,
                                                         const float *__restrict__ cos_t,
                                                         const int *__restrict__ quad_t, int p_per_blk,
                                                         const float *__restrict__ meas, const float *__restrict__ rn2,
                                                         int mode, float *__restrict__ data)
{
    typedef typename SidVec<NS>::type vec_t;
    static_assert(NS == 1 || USE_LDS, "paired slices live in LDS");
    extern __shared__ float lds[];
    const int s = blockIdx.y * NS;
    const bool has2 = NS == 2 && s + 1 < g.oy;     // an odd batch ends with a half-empty pair
    const int p0 = blockIdx.x * p_per_blk;
    const int np = min(p_per_blk, g.dt - p0);
    const float *model_g = obj + (size_t)s * g.ox * g.oz;
    const int pitch = g.oz + ((1 - (g.oz & 31)) & 31);   // == 1 (mod 32): conflict-free staging, see lds_stage.h
    if (USE_LDS) {
        if constexpr (NS == 1) {
            stage_rows(lds, model_g, g.ox, g.oz, g.oz, pitch, false, threadIdx.x & 63, threadIdx.x >> 6, blockDim.x >> 6);
        } else {
            const float *srcs[2] = {model_g, model_g + (has2 ? (size_t)g.ox * g.oz : 0)};
            stage_rows_interleaved<2>(lds, srcs, g.ox, g.oz, g.oz, pitch, false, threadIdx.x & 63, threadIdx.x >> 6,
                                      blockDim.x >> 6);
        }
        __syncthreads();
    }
    const int oz = g.oz;
    for (int ray = threadIdx.x; ray < np * g.dx; ray += blockDim.x) {
        const int pl = ray / g.dx;
        const int d = ray - pl * g.dx;
        const int p = p0 + pl;
        vec_t acc = 0.0f;
        siddon_walk_ray(g, sin_t[p], cos_t[p], quad_t[p], d, [&](int ix, int iy, float dist) {
            vec_t m;
            if constexpr (NS == 1)
                m = USE_LDS ? lds[ix * pitch + iy] : model_g[(size_t)ix * oz + iy];
            else
                m = reinterpret_cast<const vec_t *>(lds)[ix * pitch + iy];
            acc += m * dist;
        });
        auto store = [&](int sl, float sim) {
            const size_t o = ((size_t)sl * g.dt + p) * g.dx + d;
            data[o] = siddon_fwd_store(mode, sim, meas, rn2, data, o, (size_t)p * g.dx + d);
        };
        if constexpr (NS == 1) {
            store(s, acc);
        } else {
            store(s, acc.x);
            if (has2) store(s + 1, acc.y);
        }
    }
}

// Many slices per walk, from global memory: a grid too large for a PAIR of slices in LDS (184 x 184: the reconstruction grid of
// the training set, SIRT's forward) walks every ray once per slice with the kernel above.  Here NS = 4 or 8 slices are first
// interleaved per pixel ([group][pixel][NS], siddon_pack_kernel) and a lane fetches all of a pixel's values with one or two
// 16-byte loads from L2 / L1; the product of a segment is accumulated one step late, so the load has a whole step of the walk
// (~60 VALU ops) to arrive.  Same walk, same order of the sum: the same bits.
template <int NS>
__global__ __launch_bounds__(256) void siddon_pack_kernel(const float *__restrict__ obj, int oy, int npix, float *__restrict__ packed)
{
    const int pix = blockIdx.x * blockDim.x + threadIdx.x, grp = blockIdx.y;
    if (pix >= npix) return;
    float v[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = grp * NS + k < oy ? obj[(size_t)(grp * NS + k) * npix + pix] : 0.0f;
    float4 *out = reinterpret_cast<float4 *>(packed + ((size_t)grp * npix + pix) * NS);
#pragma unroll
    for (int k = 0; k < NS / 4; ++k) out[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
}

template <int NS>
__global__ __launch_bounds__(256) void siddon_fwd_packed_kernel(const float *__restrict__ packed, SidGeom g,
                                                               const float *__restrict__ sin_t, const float *__restrict__ cos_t,
                                                               const int *__restrict__ quad_t, const float *__restrict__ meas,
                                                               const float *__restrict__ rn2, int mode, float *__restrict__ data)
{
    const int grp = blockIdx.y, s0 = grp * NS;
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= g.dt * g.dx) return;
    const int p = ray / g.dx, d = ray - p * g.dx;
    const int oz = g.oz;
    const float4 *img = reinterpret_cast<const float4 *>(packed + (size_t)grp * g.ox * g.oz * NS);
    float acc[NS], pm[NS], pd = 0.0f;
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0f, pm[k] = 0.0f;
    siddon_walk_ray(g, sin_t[p], cos_t[p], quad_t[p], d, [&](int ix, int iy, float dist) {
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] += pm[k] * pd;       // the previous segment (0 + 0 * 0 the first time)
        const float4 *q = img + (size_t)(ix * oz + iy) * (NS / 4);
#pragma unroll
        for (int k = 0; k < NS / 4; ++k) {
            const float4 v = q[k];
            pm[4 * k] = v.x, pm[4 * k + 1] = v.y, pm[4 * k + 2] = v.z, pm[4 * k + 3] = v.w;
        }
        pd = dist;
    });
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const float sim = acc[k] + pm[k] * pd;
        if (s0 + k >= g.oy) break;
        const size_t o = ((size_t)(s0 + k) * g.dt + p) * g.dx + d;
        data[o] = siddon_fwd_store(mode, sim, meas, rn2, data, o, (size_t)p * g.dx + d);
    }
}

// ---- back-projector: the transpose of the forward, pixel-driven ----------------------------------------------------
// recon[s][pixel] = sum over rays of data[s][p][d] * dist(p, d, pixel) -- what libtomo's fbp.c accumulates
// (recon[indi[n]] += data[ind_data] * dist[n]) and the A^T of sirt.c's update -- as a GATHER: a lane owns a pixel and, per
// angle, asks the (at most two) rays that can cross it for "your segments in my pixel".  The answer is found with libtomo's
// own fp32 expressions: the ray's crossings with the four grid lines around the pixel (coordx / coordy of utils.c), of which
// the LATER of {left line, the horizontal line the ray enters through} is the pixel's own segment's first point and the
// EARLIER of {right line, the other horizontal line} its second -- exactly the consecutive pair sort_intersections' merge
// by x produces; the segment belongs to the pixel its midpoint falls in (calc_dist's floor), so a pixel takes it only if
// that pixel is itself.  Two things need more than that, and both are properties of the GEOMETRY, so a flag per (pixel,
// angle) is computed once (siddon_gather_flags_kernel) and sends the lane down an exact but slower path:
//   * trim_coords drops crossings within 0.01 of the grid's outline: a segment then reaches back / on to the next kept
//     crossing, pixel by pixel (rays grazing the outline);
//   * where a ray passes within a few ulps of a grid corner, the corner-cutting sliver between its two crossings there can
//     be credited to a NEIGHBOUR of the pixel it lies in, because fl(mid + half) rounds up onto the grid line: the pixel
//     then also evaluates the own segments of its neighbours at ix - 1 and / or iy - 1 and takes those whose midpoint's
//     pixel it is.
// Same points, same sqrtf, and a pixel's terms arrive in libtomo's order (angles ascending, rays ascending, along the ray):
// the result EQUALS the ray-driven accumulation bit for bit (tests: assert_array_equal against the oracle).  No atomics,
// no barriers between rays, no partial images: deterministic.
//
// Launch: workgroup = (64 columns x R rows of pixels, NS slices); per chunk of angles the rays a tile can meet (a run of
// <= kGatherSeg detector bins per angle) are staged in LDS: the ray's line (from the table siddon_ray_table_kernel wrote
// once per geometry) and the NS slices' data interleaved, so one ds_read_b128 pair serves eight slices.  Angles whose staged
// data are all zero are skipped (sparse sinograms, dose masks).  VALU-bound: ~55 ops per (pixel, ray), shared by NS slices.
constexpr int kGatherRows = 8;     // pixel rows (= waves) per workgroup
constexpr int kGatherSeg = 68;     // rays staged per angle: 63 |cos| + 7 |sin| + 3 <= 66.4
constexpr int kGatherMaxCh = 15;   // angles per LDS chunk (two staging rounds of the 512 threads)
// the staged data of slice k sit in their own PLANE of kGatherPlane floats: the 64 lanes of a wave read neighbouring (or the
// same) rays, i.e. neighbouring words of one plane -- conflict-free -- where an interleaved [ray][slice] layout put them 32 bytes
// apart (8-way conflicts: SQ_LDS_BANK_CONFLICT was 72 % of the kernel's LDS cycles)
constexpr int kGatherPlane = kGatherMaxCh * kGatherSeg;

struct GatherGeo {
    float gx0, gy0, gx_gt, gx_le, gy_gt, gy_le, hx, hz;
    int ox, oz;
};
__device__ __forceinline__ GatherGeo gather_geo(const SidGeom &g)
{
    GatherGeo G;
    G.ox = g.ox, G.oz = g.oz;
    G.gx0 = -g.ox * 0.5f, G.gy0 = -g.oz * 0.5f;
    G.gx_gt = G.gx0 + 0.01f, G.gx_le = (G.gx0 + g.ox) - 0.01f;
    G.gy_gt = G.gy0 + 0.01f, G.gy_le = (G.gy0 + g.oz) - 0.01f;
    G.hx = g.ox * 0.5f, G.hz = g.oz * 0.5f;
    return G;
}

// The two rays that can cross a pixel are the ones that bracket its centre on the detector: ray d has yi = yi0 + d, the
// centre (cxp, cyp) projects to cyp * cos - cxp * sin; any other ray is at least one detector pitch away, further than half
// a pixel's diagonal.
__device__ __forceinline__ int gather_first_ray(float cxp, float cyp, float sin_p, float cos_p, float yi0)
{
    return (int)floorf((cyp * cos_p - cxp * sin_p) - yi0);
}

struct GatherPt {
    float x, y;
    bool is_a, kept;
};

// The pieces both paths share.  A crossing is an a-point (with y = gridy[n]) or a b-point (with x = gridx[m]).
struct GatherRay {
    const GatherGeo &G;
    float srcx, srcy, slope, islope;
    bool up, a_ok, b_ok;
    __device__ __forceinline__ GatherRay(const GatherGeo &G_, const SidRayLine &r, bool up_)
        : G(G_), srcx(r.srcx), srcy(r.srcy), slope(r.slope), islope(r.islope), up(up_),
          a_ok(fabsf(r.islope) <= 3.0e38f), b_ok(fabsf(r.slope) <= 3.0e38f) {}   // an infinite slope empties that list
    __device__ __forceinline__ float a_x(float y) const { return islope * (y - srcy) + srcx; }
    __device__ __forceinline__ float b_y(float x) const { return slope * (x - srcx) + srcy; }
    __device__ __forceinline__ bool a_kept(float x) const { return a_ok && x >= G.gx_gt && x <= G.gx_le; }
    __device__ __forceinline__ bool b_kept(float y) const { return b_ok && y >= G.gy_gt && y <= G.gy_le; }
    // sort_intersections takes a first only if strictly smaller: of two candidates the a-point is the LATER one iff !(ax < bx)
    __device__ __forceinline__ GatherPt entry_of(int cx, int cy) const
    {
        const float ay = G.gy0 + (float)(up ? cy : cy + 1), ax = a_x(ay), bx = G.gx0 + (float)cx, by = b_y(bx);
        const bool is_a = a_ok && (!b_ok || !(ax < bx));
        return {is_a ? ax : bx, is_a ? ay : by, is_a, is_a ? a_kept(ax) : b_kept(by)};
    }
    __device__ __forceinline__ GatherPt exit_of(int cx, int cy) const
    {
        const float ay = G.gy0 + (float)(up ? cy + 1 : cy), ax = a_x(ay), bx = G.gx0 + (float)(cx + 1), by = b_y(bx);
        const bool is_a = a_ok && (!b_ok || ax < bx);
        return {is_a ? ax : bx, is_a ? ay : by, is_a, is_a ? a_kept(ax) : b_kept(by)};
    }
    // calc_dist: the segment's length, and: is it a segment (points in merge order) whose midpoint's pixel is (ix, iy)?
    __device__ __forceinline__ bool owned(float ex, float ey, float xx, float xy, int ix, int iy, float &dist) const
    {
        const float diffx = xx - ex, diffy = xy - ey;
        dist = sqrtf(diffx * diffx + diffy * diffy);
        const float midx = (xx + ex) * 0.5f, midy = (xy + ey) * 0.5f;
        const int indx = (int)floorf(midx + G.hx), indy = (int)floorf(midy + G.hz);
        return !(xx < ex) && min(max(indx, 0), G.ox - 1) == ix && min(max(indy, 0), G.oz - 1) == iy;
    }
    // sort_intersections' order: by x; at equal x a b-point goes first ("a first only if strictly smaller"); a-points among
    // themselves in the a-list's traversal order (upwards when the ray climbs)
    __device__ __forceinline__ bool before(const GatherPt &p, const GatherPt &q) const
    {
        if (p.x != q.x) return p.x < q.x;
        if (p.is_a != q.is_a) return !p.is_a;
        return p.is_a && (up ? p.y < q.y : p.y > q.y);
    }
    // how close (in x along a horizontal line, in y along a vertical one) does the ray pass to corner (gx, gy)?
    __device__ __forceinline__ float corner_gap(float gx, float gy) const { return fminf(fabsf(a_x(gy) - gx), fabsf(b_y(gx) - gy)); }
    // cheap superset of "the fast path may be wrong for pixel (ix, iy)": a trimmed point, or a corner within tau
    __device__ __forceinline__ bool maybe_slow(int ix, int iy, float tau) const
    {
        const GatherPt e = entry_of(ix, iy), x = exit_of(ix, iy);
        const float gxL = G.gx0 + (float)ix, gxR = G.gx0 + (float)(ix + 1), gyB = G.gy0 + (float)iy, gyT = G.gy0 + (float)(iy + 1);
        const float gap = fminf(fminf(corner_gap(gxL, gyB), corner_gap(gxL, gyT)), fminf(corner_gap(gxR, gyB), corner_gap(gxR, gyT)));
        return !(e.kept && x.kept) || gap < tau;
    }
    // geometry flag: does the slow path credit pixel (ix, iy) with anything but what the fast path gives it?
    __device__ __forceinline__ bool needs_slow(int ix, int iy, float tau) const
    {
        if (!maybe_slow(ix, iy, tau)) return false;
        float fd = -1.0f, sd = -1.0f;
        const bool f = own_segment(ix, iy, fd);
        int n = 0;
        all_segments(ix, iy, tau, [&](float dist) {
            if (n == 0) sd = dist;
            ++n;
        });
        return n != (f ? 1 : 0) || (f && sd != fd);
    }
    // fast path (flag clear): the pixel's own segment, no trimming, no neighbours.  (A ray along the rows -- sin = 0 exactly --
    // has an infinite islope and no a-points: its segments run from one vertical line to the next.)
    __device__ __forceinline__ bool own_segment(int ix, int iy, float &dist) const
    {
        const float gxL = G.gx0 + (float)ix, gxR = G.gx0 + (float)(ix + 1), gyB = G.gy0 + (float)iy, gyT = G.gy0 + (float)(iy + 1);
        const float yin = up ? gyB : gyT, yout = up ? gyT : gyB;
        const float xin = a_x(yin), xout = a_x(yout), ybL = b_y(gxL), ybR = b_y(gxR);
        const bool ea = a_ok && !(xin < gxL), xa = a_ok && xout < gxR;
        return owned(ea ? xin : gxL, ea ? yin : ybL, xa ? xout : gxR, xa ? yout : ybR, ix, iy, dist);
    }
    // slow path: every segment libtomo credits to (ix, iy), in the order of the merge.  Its own segment is extended over
    // trimmed crossings (a merged segment's midpoint lies on the stretch it spans, so the pixel that owns it finds it as its
    // own).  At a corner the ray passes within tau of, the sliver between its two crossings there is the own segment of ONE
    // of the four pixels around the corner (which one is decided by fp32 comparisons that need not agree with the geometry)
    // and is credited to the pixel its midpoint rounds into -- any of the four: so the own segments of the neighbours
    // that share a close corner are evaluated too (un-extended: a sliver lies between two KEPT crossings) and taken where
    // the midpoint's pixel is this one.  A pixel the ray misses yields its two points in the wrong order, possibly at
    // equal x: `before` is the full merge order.  At most a few segments; they are added in merge order.
    template <class F> __device__ __forceinline__ void all_segments(int ix, int iy, float tau, F &&add) const
    {
        const float gxL = G.gx0 + (float)ix, gxR = G.gx0 + (float)(ix + 1), gyB = G.gy0 + (float)iy, gyT = G.gy0 + (float)(iy + 1);
        const bool both = a_ok && b_ok;       // (a sliver has an a-point and a b-point: none when a list is empty)
        const bool c00 = both && !(corner_gap(gxL, gyB) >= tau), c01 = both && !(corner_gap(gxL, gyT) >= tau),
                   c10 = both && !(corner_gap(gxR, gyB) >= tau), c11 = both && !(corner_gap(gxR, gyT) >= tau);
        // up to three credited segments, kept sorted by (x, tie) = the merge order of their first points
        float kx[3], kt[3], kd[3];
        int n = 0;
        auto take = [&](const GatherPt &pe, float dist) {
            // equal x: a b-point first, then a-points in the a-list's order
            const float tie = pe.is_a ? (up ? pe.y : -pe.y) : -3.0e38f;
            int pos = n;
            for (int q = n - 1; q >= 0; --q)
                if (pe.x < kx[q] || (pe.x == kx[q] && tie < kt[q])) pos = q;
            for (int q = 2; q > 0; --q)
                if (q > pos) kx[q] = kx[q - 1], kt[q] = kt[q - 1], kd[q] = kd[q - 1];
            if (pos < 3) kx[pos] = pe.x, kt[pos] = tie, kd[pos] = dist;
            n = min(n + 1, 3);
        };
        {   // own, extended over trimmed points
            int cx = ix, cy = iy;
            GatherPt pe = entry_of(cx, cy), px = exit_of(cx, cy);
            bool alive = true;
            while (alive && !pe.kept) {      // trim_coords dropped it: the segment starts at the previous kept crossing
                if (pe.is_a) cy += up ? -1 : 1; else cx -= 1;
                alive = cx >= 0 && cy >= 0 && cy < G.oz;
                if (alive) pe = entry_of(cx, cy);
            }
            cx = ix, cy = iy;
            while (alive && !px.kept) {
                if (px.is_a) cy += up ? 1 : -1; else cx += 1;
                alive = cx < G.ox && cy >= 0 && cy < G.oz;
                if (alive) px = exit_of(cx, cy);
            }
            float dist;
            if (alive && before(pe, px) && owned(pe.x, pe.y, px.x, px.y, ix, iy, dist)) take(pe, dist);
        }
        if (c00 || c01 || c10 || c11) {
            for (int c = 0; c < 9; ++c) {
                const int ddx = c % 3 - 1, ddy = c / 3 - 1;
                const bool want = (ddx <= 0 && ddy <= 0 && c00) || (ddx <= 0 && ddy >= 0 && c01) || (ddx >= 0 && ddy <= 0 && c10) ||
                                  (ddx >= 0 && ddy >= 0 && c11);
                const int cx = ix + ddx, cy = iy + ddy;
                if (!want || c == 4 || cx < 0 || cy < 0 || cx >= G.ox || cy >= G.oz) continue;
                const GatherPt pe = entry_of(cx, cy), px = exit_of(cx, cy);
                float dist;
                if (pe.kept && px.kept && before(pe, px) && owned(pe.x, pe.y, px.x, px.y, ix, iy, dist)) take(pe, dist);
            }
        }
        if (n > 0) add(kd[0]);
        if (n > 1) add(kd[1]);
        if (n > 2) add(kd[2]);
    }
};

// table[p][d] = the ray's line; degenerate rays get srcx = NaN (the gather skips them); degen_angle[p] = has such a ray
__global__ __launch_bounds__(256) void siddon_ray_table_kernel(SidGeom g, const float *__restrict__ sin_t,
                                                              const float *__restrict__ cos_t, float4 *__restrict__ table,
                                                              int *__restrict__ degen_angle)
{
    const int p = blockIdx.x;
    const float sin_p = sin_t[p], cos_p = cos_t[p];
    int any = 0;
    for (int d = threadIdx.x; d < g.dx; d += blockDim.x) {
        SidRayLine r = siddon_ray_line(g, sin_p, cos_p, d);
        if (siddon_ray_is_degenerate(g, r)) {
            r.srcx = __builtin_nanf("");
            any = 1;
        }
        table[(size_t)p * g.dx + d] = make_float4(r.srcx, r.srcy, r.slope, r.islope);
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) degen_angle[p] = any;
}

// flags[w][pixel] bit b: pixel needs the slow path at angle 32 w + b (for either of its two rays) -- geometry only
__global__ __launch_bounds__(256) void siddon_gather_flags_kernel(SidGeom g, const float *__restrict__ sin_t,
                                                                 const float *__restrict__ cos_t, const int *__restrict__ quad_t,
                                                                 const float4 *__restrict__ table, float tau,
                                                                 unsigned *__restrict__ flags)
{
    const int npix = g.ox * g.oz;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (pix >= npix) return;
    const int ix = pix / g.oz, iy = pix - ix * g.oz;
    const GatherGeo G = gather_geo(g);
    const float yi0 = (1 - g.dx) / 2.0f + g.mov;
    const float cxp = (G.gx0 + (float)ix) + 0.5f, cyp = (G.gy0 + (float)iy) + 0.5f;
    unsigned word = 0;
    for (int b = 0; b < 32; ++b) {
        const int p = 32 * w + b;
        if (p >= g.dt) break;
        const float sin_p = sin_t[p], cos_p = cos_t[p];
        const bool up = quad_t[p] != 0;
        const int d1 = gather_first_ray(cxp, cyp, sin_p, cos_p, yi0);
        bool slow = false;
        for (int c = 0; c < 2; ++c) {
            const int d = d1 + c;
            if (d < 0 || d >= g.dx) continue;
            const float4 l4 = table[(size_t)p * g.dx + d];
            if (l4.x == l4.x) slow = slow || GatherRay(G, SidRayLine{l4.x, l4.y, l4.z, l4.w}, up).needs_slow(ix, iy, tau);
        }
        word |= (slow ? 1u : 0u) << b;
    }
    flags[(size_t)w * npix + pix] = word;
}

// The degenerate rays, walked as libtomo walks them, into D[s] (zeroed here; left untouched when the geometry has none):
// one lane per ray, even rays then odd rays (two rays of one parity are two detector pitches apart and never share a pixel;
// a degenerate ray's zig-zag stays on its own grid line).
__global__ __launch_bounds__(256) void siddon_bwd_degenerate_kernel(const float *__restrict__ data, SidGeom g,
                                                                   const float *__restrict__ sin_t,
                                                                   const float *__restrict__ cos_t,
                                                                   const int *__restrict__ quad_t,
                                                                   const int *__restrict__ degen_angle, float *__restrict__ D)
{
    int any = 0;
    for (int p = threadIdx.x; p < g.dt; p += blockDim.x) any |= degen_angle[p];
    if (!__syncthreads_or(any)) return;
    const int s = blockIdx.x, npix = g.ox * g.oz;
    float *img = D + (size_t)s * npix;
    for (int t = threadIdx.x; t < npix; t += blockDim.x) img[t] = 0.0f;
    __syncthreads();
    for (int p = 0; p < g.dt; ++p) {
        if (!degen_angle[p]) continue;
        const float sin_p = sin_t[p], cos_p = cos_t[p];
        const int quadrant = quad_t[p];
        const float *row = data + ((size_t)s * g.dt + p) * g.dx;
        for (int par = 0; par < 2; ++par) {
            for (int d = 2 * threadIdx.x + par; d < g.dx; d += 2 * blockDim.x) {
                const float v = row[d];
                if (v != 0.0f && siddon_ray_is_degenerate(g, siddon_ray_line(g, sin_p, cos_p, d)))
                    siddon_walk_ray(g, sin_p, cos_p, quadrant, d,
                                    [&](int ix, int iy, float dist) { img[ix * g.oz + iy] += v * dist; });
            }
            __syncthreads();
        }
    }
}

template <int NS> struct GatherAcc { float v[NS]; };

// Round 4: the primal step of the TV stand-in (recon.py _tv: preconditioned Chambolle-Pock on K = (A; grad)) as the back-projector's
// store.  All arrays [oy][ox][oz]; xbar / q are read at neighbouring pixels, so they ping-pong (in != out); x is updated in place.
//   grad u (i, j) = (u[i+1][j] - u[i][j], u[i][j+1] - u[i][j]), zero at the far edges
//   q' = q + 0.5 grad xbar;  q_new = q' / (max(|q'|, lam) / lam)                            (dual step, projection onto |q| <= lam)
//   div q (i, j) = ((qx[i][j] - qx[i-1][j]) + qy[i][j]) - qy[i][j-1], terms outside the image left out, in that order
//   x_new = x - tau (A^T p - div q_new);  xbar_new = 2 x_new - x
// A lane recomputes q_new at its own pixel and at the two neighbours its divergence needs (from the OLD xbar and q: no race).
struct TvPrimal {
    const float *tau;                 // [ox][oz]: 1 / (column sums of A + 4)
    float lam;
    float *x;
    const float *xbar_in, *qx_in, *qy_in;
    float *xbar_out, *qx_out, *qy_out;
};
__device__ __forceinline__ void tv_dual_q(const TvPrimal &t, const float *xb, const float *qx, const float *qy, int ox, int oz, int i,
                                          int j, float &nx, float &ny)
{
    const size_t c = (size_t)i * oz + j;
    const float u = xb[c];
    const float gx = i + 1 < ox ? xb[c + oz] - u : 0.0f, gy = j + 1 < oz ? xb[c + 1] - u : 0.0f;
    const float ax = qx[c] + 0.5f * gx, ay = qy[c] + 0.5f * gy;
    const float nrm = fmaxf(sqrtf(ax * ax + ay * ay), t.lam) / t.lam;
    nx = ax / nrm;
    ny = ay / nrm;
}

// EPI 0: recon = A^T data.   EPI 1 (SIRT): recon += (A^T data) / colsum where colsum != 0 (libtomo sirt.c's last loop).
template <int NS, int EPI>
__global__ __launch_bounds__(kGatherRows * 64) void siddon_bwd_gather_kernel(
    const float *__restrict__ data, SidGeom g, const float *__restrict__ sin_t, const float *__restrict__ cos_t,
    const int *__restrict__ quad_t, const float4 *__restrict__ table, const unsigned *__restrict__ flags,
    const int *__restrict__ degen_angle, const float *__restrict__ D, const float *__restrict__ colsum, int CH, float tau,
    float *__restrict__ recon, TvPrimal tv)
{
    extern __shared__ float lds[];
    // LDS: lines [CH][SEG] float4 | vals [NS][kGatherPlane] | seg_lo [CH] | live [CH] | anyD
    float4 *lines = reinterpret_cast<float4 *>(lds);
    float *vals = lds + (size_t)CH * kGatherSeg * 4;
    int *seg_lo = reinterpret_cast<int *>(vals + (size_t)NS * kGatherPlane);
    int *live = seg_lo + CH;
    int *any_d = live + CH;
    const int tiles_y = (g.oz + 63) / 64;
    const int ty = blockIdx.x % tiles_y, tx = blockIdx.x / tiles_y;
    const int ix0 = tx * kGatherRows, iy0 = ty * 64;
    const int s0 = blockIdx.y * NS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ix = ix0 + wave, iy = iy0 + lane;
    const bool mine = ix < g.ox && iy < g.oz;
    const int npix = g.ox * g.oz;
    const GatherGeo G = gather_geo(g);
    const float yi0 = (1 - g.dx) / 2.0f + g.mov;         // yi of ray d is yi0 + d
    // pixel centre (libtomo's grid coordinates); its detector coordinate at angle p is  cyp * cos - cxp * sin
    const float cxp = (G.gx0 + (float)ix) + 0.5f, cyp = (G.gy0 + (float)iy) + 0.5f;
    if (threadIdx.x == 0) *any_d = 0;
    __syncthreads();
    {
        int any = 0;
        for (int p = threadIdx.x; p < g.dt; p += blockDim.x) any |= degen_angle[p];
        if (any) *any_d = 1;
    }
    __syncthreads();
    float acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = (*any_d && mine && s0 + k < g.oy) ? D[(size_t)(s0 + k) * npix + ix * g.oz + iy] : 0.0f;

    // the tile's corners (clipped tiles: the full rectangle -- a superset)
    const float cx_lo = (G.gx0 + (float)ix0) + 0.5f, cx_hi = (G.gx0 + (float)(ix0 + kGatherRows - 1)) + 0.5f;
    const float cy_lo = (G.gy0 + (float)iy0) + 0.5f, cy_hi = (G.gy0 + (float)(iy0 + 63)) + 0.5f;
    unsigned word = 0;
    for (int p0 = 0; p0 < g.dt; p0 += CH) {
        const int np = min(CH, g.dt - p0);
        __syncthreads();          // the previous chunk has been consumed
        for (int a = threadIdx.x; a < np; a += blockDim.x) live[a] = 0;
        __syncthreads();
        for (int e = threadIdx.x; e < np * kGatherSeg; e += blockDim.x) {
            const int a = e / kGatherSeg, r = e - a * kGatherSeg;
            const int p = p0 + a;
            const float sin_p = sin_t[p], cos_p = cos_t[p];
            // every term is monotone in cxp and in cyp: the extremes sit at the corners, with the lanes' own rounding
            const float c00 = cy_lo * cos_p - cx_lo * sin_p, c01 = cy_hi * cos_p - cx_lo * sin_p;
            const float c10 = cy_lo * cos_p - cx_hi * sin_p, c11 = cy_hi * cos_p - cx_hi * sin_p;
            const int lo = (int)floorf(fminf(fminf(c00, c01), fminf(c10, c11)) - yi0);
            if (r == 0) seg_lo[a] = lo;
            const int d = lo + r;
            float4 line = make_float4(__builtin_nanf(""), 0.0f, 0.0f, 0.0f);
            float v[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) v[k] = 0.0f;
            if (d >= 0 && d < g.dx) {
                line = table[(size_t)p * g.dx + d];
#pragma unroll
                for (int k = 0; k < NS; ++k)
                    if (s0 + k < g.oy) v[k] = data[((size_t)(s0 + k) * g.dt + p) * g.dx + d];
            }
            lines[e] = line;
            bool nz = false;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                vals[k * kGatherPlane + e] = v[k];
                nz = nz || v[k] != 0.0f;
            }
            if (nz && line.x == line.x) live[a] = 1;
        }
        __syncthreads();
        for (int a = 0; a < np; ++a) {
            const int p = p0 + a;
            if (a == 0 || (p & 31) == 0) word = mine ? flags[(size_t)(p >> 5) * npix + ix * g.oz + iy] : 0u;
            if (!live[a]) continue;
            const float sin_p = sin_t[p], cos_p = cos_t[p];
            const bool up = quad_t[p] != 0;
#ifdef CTPVAE_TUNE_GATHER_NOSLOW
            const bool slow = false;
#else
            const bool slow = (word >> (p & 31)) & 1u;
#endif
            const int r1 = min(max(gather_first_ray(cxp, cyp, sin_p, cos_p, yi0) - seg_lo[a], 0), kGatherSeg - 2);
#pragma unroll
            for (int c = 0; c < 2; ++c) {       // the two rays that bracket the pixel's centre, ascending
                const int e = a * kGatherSeg + r1 + c;
                const float4 l4 = lines[e];
                if (!(mine && l4.x == l4.x)) continue;
                const GatherRay R(G, SidRayLine{l4.x, l4.y, l4.z, l4.w}, up);
                auto add = [&](float dist) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) acc[k] += vals[k * kGatherPlane + e] * dist;
                };
                if (slow) {
                    R.all_segments(ix, iy, tau, add);
                } else {
                    float dist;
                    if (R.own_segment(ix, iy, dist)) add(dist);
                }
            }
        }
    }
    if (!mine) return;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (s0 + k >= g.oy) break;
        float *out = recon + (size_t)(s0 + k) * npix + ix * g.oz + iy;
        if constexpr (EPI == 0) {
            *out = acc[k];
        } else if constexpr (EPI == 1) {
            const float cs = colsum[ix * g.oz + iy];
            if (cs != 0.0f) *out += acc[k] / cs;
        } else {   // EPI 2: the TV stand-in's primal step (TvPrimal above); `recon` is not written
            const size_t so = (size_t)(s0 + k) * npix, c = (size_t)ix * g.oz + iy;
            const float *xb = tv.xbar_in + so, *qxi = tv.qx_in + so, *qyi = tv.qy_in + so;
            float qx0, qy0, qxu = 0.0f, qyl = 0.0f, unused;
            tv_dual_q(tv, xb, qxi, qyi, g.ox, g.oz, ix, iy, qx0, qy0);
            if (ix >= 1) tv_dual_q(tv, xb, qxi, qyi, g.ox, g.oz, ix - 1, iy, qxu, unused);
            if (iy >= 1) tv_dual_q(tv, xb, qxi, qyi, g.ox, g.oz, ix, iy - 1, unused, qyl);
            float dv = 0.0f;
            if (ix + 1 < g.ox) dv += qx0;
            if (ix >= 1) dv -= qxu;
            if (iy + 1 < g.oz) dv += qy0;
            if (iy >= 1) dv -= qyl;
            const float xo = tv.x[so + c];
            const float xn = xo - tv.tau[c] * (acc[k] - dv);
            tv.x[so + c] = xn;
            tv.xbar_out[so + c] = 2.0f * xn - xo;
            tv.qx_out[so + c] = qx0;
            tv.qy_out[so + c] = qy0;
        }
    }
}

// SIRT's row weights (libtomo sirt.c: sum_dist2 = sum of dist[n]^2 over a ray's segments): geometry only, [dt][dx]
__global__ __launch_bounds__(256) void siddon_rownorm_kernel(SidGeom g, const float *__restrict__ sin_t,
                                                            const float *__restrict__ cos_t, const int *__restrict__ quad_t,
                                                            float *__restrict__ rn2)
{
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= g.dt * g.dx) return;
    const int p = ray / g.dx, d = ray - p * g.dx;
    float acc = 0.0f;
    siddon_walk_ray(g, sin_t[p], cos_t[p], quad_t[p], d, [&](int, int, float dist) { acc += dist * dist; });
    rn2[ray] = acc;
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

static int siddon_fwd_one(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev,
                          const float *cos_dev, const int *quad_dev, int dt, int dx, float center, const float *meas_dev,
                          const float *rn2_dev, int mode, float *data_dev, ctpvae_stream_t stream);
static float siddon_mov(int dx, float center);

static int siddon_fwd_chunks(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                             const int *quad_dev, int dt, int dx, float center, const float *meas_dev, const float *rn2_dev,
                             int mode, float *data_dev, ctpvae_stream_t stream)
{
    const int chunk = std::max(2, max_slices_per_launch() / 2 * 2);   // even: whole slice pairs per chunk
    for (int s0 = 0; s0 < oy; s0 += chunk) {
        const int n = std::min(chunk, oy - s0);
        if (int rc = siddon_fwd_one(obj_dev + (size_t)s0 * ox * oz, n, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center,
                                    meas_dev ? meas_dev + (size_t)s0 * dt * dx : nullptr, rn2_dev, mode,
                                    data_dev + (size_t)s0 * dt * dx, stream))
            return rc;
    }
    return CTPVAE_OK;
}

// slices are indexed with a grid dimension (<= 65535): a longer stack goes in chunks, back to back on the stream
int ctpvae_siddon_fwd_f32(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev,
                          const float *cos_dev, const int *quad_dev, int dt, int dx, float center,
                          float *data_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(obj_dev && data_dev && oy > 0 && ox > 0 && oz > 0 && dt > 0 && dx > 0, "siddon_fwd: null pointer or empty sizes");
    return siddon_fwd_chunks(obj_dev, oy, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center, nullptr, nullptr, 0, data_dev, stream);
}

// slices per walk of the packed forward: 0 = keep the LDS kernels (few slices, or a grid whose PAIRS fit LDS and ... see below)
static int siddon_packed_ns(int oy, int ox, int oz)
{
    if (knob(kKnobSiddonNs) == 1 || knob(kKnobSiddonNs) == 2) return 0;       // the knob asks for an LDS kernel
    if (knob(kKnobSiddonNs) == 4 || knob(kKnobSiddonNs) == 8) return knob(kKnobSiddonNs);
    return oy >= 6 ? 8 : oy >= 3 ? 4 : 0;
}

long long ctpvae_siddon_fwd_workspace_bytes(int oy, int ox, int oz)
{
    if (oy <= 0 || ox <= 0 || oz <= 0) return fail(CTPVAE_EINVAL, "siddon_fwd_workspace_bytes: bad sizes");
    const int ns = siddon_packed_ns(oy, ox, oz);
    return ns ? (long long)ceil_div(oy, ns) * ns * ox * oz * (long long)sizeof(float) : 0;
}

}  // extern "C"

template <int NS>
static int siddon_fwd_packed(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                             const int *quad_dev, int dt, int dx, float center, const float *meas_dev, const float *rn2_dev,
                             int mode, float *packed, float *data_dev, hipStream_t stream)
{
    const int npix = ox * oz, groups = ceil_div(oy, NS);
    CTPVAE_REQUIRE(groups <= 65535, "siddon_fwd: at most %d slices per call with a workspace (got %d)", 65535 * NS, oy);
    hipLaunchKernelGGL(siddon_pack_kernel<NS>, dim3(ceil_div(npix, 256), groups), dim3(256), 0, stream, obj_dev, oy, npix, packed);
    CTPVAE_LAUNCH_CHECK("siddon_pack_kernel");
    const SidGeom g{oy, ox, oz, dt, dx, siddon_mov(dx, center)};
    hipLaunchKernelGGL(siddon_fwd_packed_kernel<NS>, dim3(ceil_div(dt * dx, 256), groups), dim3(256), 0, stream, packed, g, sin_dev,
                       cos_dev, quad_dev, meas_dev, rn2_dev, mode, data_dev);
    CTPVAE_LAUNCH_CHECK("siddon_fwd_packed_kernel");
    return CTPVAE_OK;
}

extern "C" {

static int siddon_fwd_ws(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                         const int *quad_dev, int dt, int dx, float center, const float *meas_dev, const float *rn2_dev, int mode,
                         void *workspace_dev, float *data_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(obj_dev && data_dev && sin_dev && cos_dev && quad_dev && oy > 0 && ox > 0 && oz > 0 && dt > 0 && dx > 0,
                   "siddon_fwd: null pointer or empty sizes");
    CTPVAE_REQUIRE((meas_dev == nullptr) == (rn2_dev == nullptr), "siddon_fwd: meas and the per-ray weights go together");
    int ns = siddon_packed_ns(oy, ox, oz);
    // A launch of few waves takes as long as ONE walk -- 55-70 us with eight slices behind a ray (packed, objects through the L2),
    // 43-47 us with the slice pair in LDS: up to ~750 waves of pairs the LDS kernels are the faster ones (16 x 128^2 x 20 angles: 47.4
    // against 67.6 us, 24 x 20: 47.4 against 67.6; 32 x 20, 920 waves of pairs: 75.6 against 66-69; tools/sweep_siddon_ns.py)
    if (ns != 0 && knob(kKnobSiddonNs) < 0 && 2 * (size_t)ox * (oz + ((1 - (oz & 31)) & 31)) * sizeof(float) <= (size_t)kMaxLdsBytes &&
        (long long)ceil_div(oy, 2) * dt * dx <= 750ll * 64)
        ns = 0;
    if (ns == 0)
        return siddon_fwd_chunks(obj_dev, oy, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center, meas_dev, rn2_dev, mode, data_dev, stream);
    CTPVAE_REQUIRE(workspace_dev, "siddon_fwd: %d slices need the workspace", oy);
    CTPVAE_REQUIRE(((uintptr_t)workspace_dev & 15) == 0, "siddon_fwd: the workspace must be 16-byte aligned");
    return ns == 8 ? siddon_fwd_packed<8>(obj_dev, oy, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center, meas_dev, rn2_dev, mode,
                                          (float *)workspace_dev, data_dev, (hipStream_t)stream)
                   : siddon_fwd_packed<4>(obj_dev, oy, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center, meas_dev, rn2_dev, mode,
                                          (float *)workspace_dev, data_dev, (hipStream_t)stream);
}

int ctpvae_siddon_fwd_ws_f32(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                             const int *quad_dev, int dt, int dx, float center, const float *meas_dev, const float *rn2_dev,
                             void *workspace_dev, float *data_dev, ctpvae_stream_t stream)
{
    return siddon_fwd_ws(obj_dev, oy, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center, meas_dev, rn2_dev, meas_dev ? 1 : 0,
                         workspace_dev, data_dev, stream);
}

// Round 4, the TV stand-in's dual step as the projector's own store (recon.py _tv; tomopy.recon(algorithm='tv') is NOT restated):
//   p <- (p + sigma (A xbar - b)) / (1 + sigma),  sigma_dev [dt][dx] the per-ray step, p_dev [oy][dt][dx] updated in place.
int ctpvae_siddon_fwd_ws_tv_dual_f32(const float *xbar_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                                     const int *quad_dev, int dt, int dx, float center, const float *meas_dev,
                                     const float *sigma_dev, void *workspace_dev, float *p_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(meas_dev && sigma_dev, "siddon_fwd_tv_dual: null pointer");
    return siddon_fwd_ws(xbar_dev, oy, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center, meas_dev, sigma_dev, 2, workspace_dev,
                         p_dev, stream);
}

int ctpvae_siddon_fwd_resid_f32(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                                const int *quad_dev, int dt, int dx, float center, const float *meas_dev,
                                const float *rn2_dev, float *upd_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(obj_dev && upd_dev && meas_dev && rn2_dev && oy > 0 && ox > 0 && oz > 0 && dt > 0 && dx > 0,
                   "siddon_fwd_resid: null pointer or empty sizes");
    return siddon_fwd_chunks(obj_dev, oy, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center, meas_dev, rn2_dev, 1, upd_dev, stream);
}

static int siddon_fwd_one(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev,
                          const float *cos_dev, const int *quad_dev, int dt, int dx, float center, const float *meas_dev,
                          const float *rn2_dev, int mode, float *data_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(obj_dev && sin_dev && cos_dev && quad_dev && data_dev, "siddon_fwd: null pointer");
    CTPVAE_REQUIRE(oy > 0 && ox > 0 && oz > 0 && dt > 0 && dx > 0,
                   "siddon_fwd: sizes must be positive (oy=%d ox=%d oz=%d dt=%d dx=%d)", oy, ox, oz, dt, dx);
    CTPVAE_REQUIRE(oy <= 65535, "siddon_fwd: at most 65535 slices per call (got %d)", oy);
    const SidGeom g{oy, ox, oz, dt, dx, siddon_mov(dx, center)};
    const size_t lds_one = (size_t)ox * (oz + ((1 - (oz & 31)) & 31)) * sizeof(float);
    const bool use_lds = lds_one <= (size_t)kMaxLdsBytes;
    // two slices per workgroup when the pair fits LDS and the call has slices to pair
    // (... and up to ~500 waves single slices: more workgroups of the same walk -- 8 x 128^2 x 20 angles 43.8 us single, 47.1 paired)
    int ns = (oy >= 2 && 2 * lds_one <= (size_t)kMaxLdsBytes && (long long)oy * dt * dx > 500ll * 64) ? 2 : 1;
    if (knob(kKnobSiddonNs) >= 0) ns = (knob(kKnobSiddonNs) == 2 && oy >= 2 && 2 * lds_one <= (size_t)kMaxLdsBytes) ? 2 : 1;
    const int units = ceil_div(oy, ns);
    const size_t lds_bytes = lds_one * ns;
    int ppb = dt;
    while (ppb > 1 && (long long)units * ceil_div(dt, ppb) < 512) ppb = (ppb + 1) / 2;
    // the kernel is VALU-bound and every workgroup holds its slice(s) in LDS: 16 waves per workgroup keep 8 (4 with a
    // pair) waves on every SIMD (4 waves per workgroup left 2 per SIMD -- one wave issues a VALU op every ~4.4 cycles)
    int threads = std::min(1024, ceil_div(ppb * dx, 64) * 64);
    if (knob(kKnobSiddonThreads) > 0) threads = std::max(64, std::min(1024, knob(kKnobSiddonThreads) / 64 * 64));
    if (knob(kKnobSiddonPpb) > 0) ppb = std::min(dt, knob(kKnobSiddonPpb));
    const dim3 grid(ceil_div(dt, ppb), units), block(threads);
    auto launch = [&](auto kernel, size_t shmem) -> int {
        if (shmem > 64 * 1024)
            CTPVAE_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)shmem));
        hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)stream, obj_dev, g, sin_dev, cos_dev, quad_dev,
                           ppb, meas_dev, rn2_dev, mode, data_dev);
        CTPVAE_LAUNCH_CHECK("siddon_fwd_kernel");
        return CTPVAE_OK;
    };
    if (!use_lds) return launch(siddon_fwd_kernel<false, 1>, 0);
    return ns == 2 ? launch(siddon_fwd_kernel<true, 2>, lds_bytes) : launch(siddon_fwd_kernel<true, 1>, lds_bytes);
}

// ---- back-projector (transpose) and SIRT row weights -----------------------------------------------------------
static float siddon_mov(int dx, float center)
{
    float mov = ((float)dx - 1) * 0.5f - center;   // utils.c preprocessing(): detector shift
    if (mov - std::floor(mov) < 0.01f) mov += 0.01f;
    return mov + 0.5f;
}

}  // extern "C"

// workspace of the back-projector: the ray table [dt][dx] float4, degen_angle [dt] int, the slow-path flags
// [ceil(dt / 32)][ox][oz] u32, D [oy][ox][oz] float (the
// degenerate rays' image; written only when the geometry has such rays)
struct SidWorkspace {
    long long off_table, off_degen, off_flags, off_d, bytes;
};
static SidWorkspace siddon_workspace(int oy, int ox, int oz, int dt, int dx)
{
    auto up = [](long long v) { return (v + 255) / 256 * 256; };
    SidWorkspace w;
    w.off_table = 0;
    w.off_degen = up((long long)dt * dx * 16);
    w.off_flags = up(w.off_degen + (long long)dt * 4);
    w.off_d = up(w.off_flags + (long long)ceil_div(dt, 32) * ox * oz * 4);
    w.bytes = w.off_d + (long long)oy * ox * oz * 4;
    return w;
}

extern "C" {

long long ctpvae_siddon_bwd_workspace_bytes(int oy, int ox, int oz, int dt, int dx)
{
    if (oy <= 0 || ox <= 0 || oz <= 0 || dt <= 0 || dx <= 0) return fail(CTPVAE_EINVAL, "siddon_bwd_workspace_bytes: bad sizes");
    return siddon_workspace(oy, ox, oz, dt, dx).bytes;
}

// fl(mid + half) can round up onto a grid line when mid is within half an ulp of it: two ulps of the largest coordinate
static float siddon_tau(int ox, int oz)
{
    const float big = (float)std::max(ox, oz);
    return 2.0f * (std::nextafter(big, 2.0f * big) - big);
}

int ctpvae_siddon_bwd_prepare_f32(int ox, int oz, const float *sin_dev, const float *cos_dev, const int *quad_dev, int dt, int dx,
                                  float center, void *workspace_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(sin_dev && cos_dev && quad_dev && workspace_dev, "siddon_bwd_prepare: null pointer");
    CTPVAE_REQUIRE(ox > 0 && oz > 0 && dt > 0 && dx > 0, "siddon_bwd_prepare: sizes must be positive");
    CTPVAE_REQUIRE(((uintptr_t)workspace_dev & 255) == 0, "siddon_bwd_prepare: the workspace must be 256-byte aligned");
    const SidWorkspace w = siddon_workspace(1, ox, oz, dt, dx);
    const SidGeom g{1, ox, oz, dt, dx, siddon_mov(dx, center)};
    hipLaunchKernelGGL(siddon_ray_table_kernel, dim3(dt), dim3(256), 0, (hipStream_t)stream, g, sin_dev, cos_dev,
                       (float4 *)((char *)workspace_dev + w.off_table), (int *)((char *)workspace_dev + w.off_degen));
    CTPVAE_LAUNCH_CHECK("siddon_ray_table_kernel");
    hipLaunchKernelGGL(siddon_gather_flags_kernel, dim3(ceil_div(ox * oz, 256), ceil_div(dt, 32)), dim3(256), 0, (hipStream_t)stream, g,
                       sin_dev, cos_dev, quad_dev, (const float4 *)((char *)workspace_dev + w.off_table), siddon_tau(ox, oz),
                       (unsigned *)((char *)workspace_dev + w.off_flags));
    CTPVAE_LAUNCH_CHECK("siddon_gather_flags_kernel");
    return CTPVAE_OK;
}

}  // extern "C"

template <int NS>
static int siddon_gather_launch(const float *data, const SidGeom &g, const float *sin_dev, const float *cos_dev,
                                const int *quad_dev, const float4 *table, const unsigned *flags, const int *degen,
                                const float *D, const float *colsum, float *recon, hipStream_t stream, const TvPrimal *tv = nullptr)
{
    // angles per LDS chunk: two staging rounds of the 512 threads, ~50 KB with eight slices -> three workgroups per CU
    int CH = std::max(1, std::min(g.dt, kGatherMaxCh));
    if (knob(kKnobSiddonBwdChunks) > 0) CH = std::max(1, std::min(CH, knob(kKnobSiddonBwdChunks)));
    const size_t shmem = (size_t)CH * kGatherSeg * 16 + (size_t)NS * kGatherPlane * 4 + (size_t)CH * 8 + 16;
    const float tau = siddon_tau(g.ox, g.oz);
    const dim3 grid(ceil_div(g.ox, kGatherRows) * ceil_div(g.oz, 64), ceil_div(g.oy, NS)), block(kGatherRows * 64);
    auto launch = [&](auto kernel) -> int {
        static std::atomic<unsigned long long> attr_set{0};
        if (shmem > 64 * 1024) CTPVAE_SET_MAX_LDS_ONCE(kernel, attr_set);
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, data, g, sin_dev, cos_dev, quad_dev, table, flags, degen, D, colsum,
                           CH, tau, recon, tv ? *tv : TvPrimal{});
        CTPVAE_LAUNCH_CHECK("siddon_bwd_gather_kernel");
        return CTPVAE_OK;
    };
    if (tv) return launch(siddon_bwd_gather_kernel<NS, 2>);
    return colsum ? launch(siddon_bwd_gather_kernel<NS, 1>) : launch(siddon_bwd_gather_kernel<NS, 0>);
}

extern "C" {

static int siddon_bwd_prepared(const float *data_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                               const int *quad_dev, int dt, int dx, float center, const void *workspace_dev,
                               const float *colsum_dev, float *recon_dev, ctpvae_stream_t stream, const TvPrimal *tv)
{
    CTPVAE_REQUIRE(data_dev && sin_dev && cos_dev && quad_dev && (recon_dev || tv) && workspace_dev, "siddon_bwd: null pointer");
    CTPVAE_REQUIRE(oy > 0 && ox > 0 && oz > 0 && dt > 0 && dx > 0,
                   "siddon_bwd: sizes must be positive (oy=%d ox=%d oz=%d dt=%d dx=%d)", oy, ox, oz, dt, dx);
    CTPVAE_REQUIRE(((uintptr_t)workspace_dev & 255) == 0, "siddon_bwd: the workspace must be 256-byte aligned");
    const SidWorkspace w = siddon_workspace(oy, ox, oz, dt, dx);
    const float4 *table = (const float4 *)((const char *)workspace_dev + w.off_table);
    const int *degen = (const int *)((const char *)workspace_dev + w.off_degen);
    const unsigned *flags = (const unsigned *)((const char *)workspace_dev + w.off_flags);
    float *D = (float *)((char *)workspace_dev + w.off_d);
    const long long npix = (long long)ox * oz;
    // slices per workgroup: the walk of a (pixel, ray) is shared by all of them
    int ns = oy >= 8 ? 8 : oy >= 4 ? 4 : oy >= 2 ? 2 : 1;
    if (knob(kKnobSiddonBwdNs) > 0) ns = knob(kKnobSiddonBwdNs) >= 8 ? 8 : knob(kKnobSiddonBwdNs) >= 4 ? 4 : knob(kKnobSiddonBwdNs) >= 2 ? 2 : 1;
    const int chunk = std::max(ns, std::min(65535, max_slices_per_launch()) / ns * ns);
    for (int s0 = 0; s0 < oy; s0 += chunk) {
        SidGeom g{std::min(chunk, oy - s0), ox, oz, dt, dx, siddon_mov(dx, center)};
        const float *data = data_dev + (size_t)s0 * dt * dx;
        float *Ds = D + (size_t)s0 * npix, *recon = recon_dev ? recon_dev + (size_t)s0 * npix : nullptr;
        TvPrimal tvs{};
        if (tv) {   // this chunk's slices
            const size_t so = (size_t)s0 * npix;
            tvs = TvPrimal{tv->tau, tv->lam, tv->x + so, tv->xbar_in + so, tv->qx_in + so, tv->qy_in + so, tv->xbar_out + so,
                           tv->qx_out + so, tv->qy_out + so};
        }
        const TvPrimal *tvp = tv ? &tvs : nullptr;
        hipLaunchKernelGGL(siddon_bwd_degenerate_kernel, dim3(g.oy), dim3(256), 0, (hipStream_t)stream, data, g, sin_dev, cos_dev,
                           quad_dev, degen, Ds);
        CTPVAE_LAUNCH_CHECK("siddon_bwd_degenerate_kernel");
        int rc;
        switch (ns) {
        case 8: rc = siddon_gather_launch<8>(data, g, sin_dev, cos_dev, quad_dev, table, flags, degen, Ds, colsum_dev, recon, (hipStream_t)stream, tvp); break;
        case 4: rc = siddon_gather_launch<4>(data, g, sin_dev, cos_dev, quad_dev, table, flags, degen, Ds, colsum_dev, recon, (hipStream_t)stream, tvp); break;
        case 2: rc = siddon_gather_launch<2>(data, g, sin_dev, cos_dev, quad_dev, table, flags, degen, Ds, colsum_dev, recon, (hipStream_t)stream, tvp); break;
        default: rc = siddon_gather_launch<1>(data, g, sin_dev, cos_dev, quad_dev, table, flags, degen, Ds, colsum_dev, recon, (hipStream_t)stream, tvp); break;
        }
        if (rc) return rc;
    }
    return CTPVAE_OK;
}

int ctpvae_siddon_bwd_prepared_f32(const float *data_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                                   const int *quad_dev, int dt, int dx, float center, const void *workspace_dev,
                                   const float *colsum_dev, float *recon_dev, ctpvae_stream_t stream)
{
    return siddon_bwd_prepared(data_dev, oy, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center, workspace_dev, colsum_dev, recon_dev,
                               stream, nullptr);
}

// Round 4: the TV stand-in's primal step as the back-projector's store (TvPrimal above; recon.py _tv).  p_dev [oy][dt][dx] the dual
// variable of the data term; x updated in place; xbar / qx / qy read from *_in and written to *_out (distinct buffers).
int ctpvae_siddon_bwd_tv_primal_f32(const float *p_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                                    const int *quad_dev, int dt, int dx, float center, const void *workspace_dev,
                                    const float *tau_dev, float lam, float *x_dev, const float *xbar_in_dev, float *xbar_out_dev,
                                    const float *qx_in_dev, const float *qy_in_dev, float *qx_out_dev, float *qy_out_dev,
                                    ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(tau_dev && x_dev && xbar_in_dev && xbar_out_dev && qx_in_dev && qy_in_dev && qx_out_dev && qy_out_dev,
                   "siddon_bwd_tv_primal: null pointer");
    CTPVAE_REQUIRE(lam > 0.0f, "siddon_bwd_tv_primal: the TV weight must be positive");
    CTPVAE_REQUIRE(xbar_in_dev != xbar_out_dev && qx_in_dev != qx_out_dev && qy_in_dev != qy_out_dev,
                   "siddon_bwd_tv_primal: xbar and q are read at neighbouring pixels -- in and out must be distinct buffers");
    const TvPrimal tv{tau_dev, lam, x_dev, xbar_in_dev, qx_in_dev, qy_in_dev, xbar_out_dev, qx_out_dev, qy_out_dev};
    return siddon_bwd_prepared(p_dev, oy, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center, workspace_dev, nullptr, nullptr, stream, &tv);
}

int ctpvae_siddon_bwd_f32(const float *data_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                          const int *quad_dev, int dt, int dx, float center, void *workspace_dev, float *recon_dev,
                          ctpvae_stream_t stream)
{
    if (int rc = ctpvae_siddon_bwd_prepare_f32(ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center, workspace_dev, stream)) return rc;
    return ctpvae_siddon_bwd_prepared_f32(data_dev, oy, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center, workspace_dev, nullptr,
                                          recon_dev, stream);
}

int ctpvae_siddon_rownorm_f32(int ox, int oz, const float *sin_dev, const float *cos_dev, const int *quad_dev, int dt, int dx,
                              float center, float *rn2_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(sin_dev && cos_dev && quad_dev && rn2_dev, "siddon_rownorm: null pointer");
    CTPVAE_REQUIRE(ox > 0 && oz > 0 && dt > 0 && dx > 0, "siddon_rownorm: sizes must be positive");
    const SidGeom g{1, ox, oz, dt, dx, siddon_mov(dx, center)};
    hipLaunchKernelGGL(siddon_rownorm_kernel, dim3(ceil_div(dt * dx, 256)), dim3(256), 0, (hipStream_t)stream, g, sin_dev,
                       cos_dev, quad_dev, rn2_dev);
    CTPVAE_LAUNCH_CHECK("siddon_rownorm_kernel");
    return CTPVAE_OK;
}

}  // extern "C"
