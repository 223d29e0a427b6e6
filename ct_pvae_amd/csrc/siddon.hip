// siddon.hip -- TomoPy-style ray-driven projector (a7): exact ray / pixel-grid intersection lengths.
//
// Follows libtomo's project(): for every (angle p, detector bin d) the ray's crossings with the
// horizontal grid lines (list "a") and with the vertical grid lines (list "b") are merged by x,
// consecutive crossings give a segment length and, from the segment midpoint, a pixel.  libtomo
// materialises both lists and the merged list; here one lane owns one ray and performs the merge on
// the fly with two cursors, evaluating exactly the same fp32 expressions, so no per-ray arrays exist
// and the slice is read from LDS.  Compiled with -ffp-contract=off; '/' and sqrtf are correctly
// rounded (hipcc default), so the result equals the CPU restatement bit for bit.
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
// CHUNKED: only the segments [nseg * chunk / nchunks, nseg * (chunk + 1) / nchunks) of the ray are visited -- the walk starts
// in the middle of the merge, at the cursor pair found by a merge-path bisection on the same two key sequences (both are
// monotone: the a-list's fp32 expression in its traversal order, the b-list's grid lines), so every visited segment has
// exactly the points, length and pixel the whole walk gives it.
template <bool CHUNKED = false, class F>
__device__ __forceinline__ void siddon_walk_ray(const SidGeom &g, float sin_p, float cos_p, int quadrant, int d, F &&segment,
                                                int chunk = 0, int nchunks = 1)
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
    int k_begin = 0, k_end = csize, ia0 = 0;
    if constexpr (CHUNKED) {
        const int nseg = max(csize - 1, 0);
        // Runs of one ray may share a pixel only where they touch -- the back-projector's phase rule -- as long as the ray
        // meets a pixel in consecutive segments only.  That fails for a ray that lies ON a grid line of a direction it is
        // (numerically) parallel to: with a slope of ~1e7 the crossings with the coinciding line family land at erratic
        // positions and libtomo's merge gives long zig-zag segments whose midpoints revisit pixels (odd grids under the even
        // padded detector, theta = pi/2 or 0 exactly); and with fewer segments than runs, empty runs break the even / odd
        // alternation.  Such a ray is walked whole by ITS FIRST RUN's lane: one lane, sequential adds, nothing to collide
        // with (same-parity rays stay two detector pitches away).
        bool whole = nseg < nchunks;
        if (!(fabsf(slope) <= 1.0e6f)) {          // parallel to the x = gridx[n] lines
            const float f = srcx - gx0;
            whole = whole || fabsf(f - rintf(f)) < 1.0e-3f;
        }
        if (!(fabsf(islope) <= 1.0e6f)) {         // parallel to the y = gridy[n] lines
            const float f = srcy - gy0;
            whole = whole || fabsf(f - rintf(f)) < 1.0e-3f;
        }
        if (whole && chunk != 0) return;
        const int s0 = whole ? 0 : (int)((long long)nseg * chunk / nchunks);
        const int s1 = whole ? nseg : (int)((long long)nseg * (chunk + 1) / nchunks);
        if (s1 <= s0) return;
        k_begin = s0;
        k_end = s1 + 1;
        // ia0 = how many a-elements are among the first k_begin merged points ("a first only if strictly smaller")
        int lo = max(0, k_begin - b_cnt), hi = min(k_begin, a_cnt);
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const float ay = gy0 + (float)(quadrant ? a_lo + mid : a_lo + a_cnt - 1 - mid);
            const float akey = islope * (ay - srcy) + srcx;
            const float bkey = gx0 + (float)(b_lo + (k_begin - 1 - mid));
            if (akey < bkey) lo = mid + 1; else hi = mid;
        }
        ia0 = lo;
    }
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
typedef float sid_f32x2 __attribute__((ext_vector_type(2)));
template <int NS> struct SidVec { typedef float type; };
template <> struct SidVec<2> { typedef sid_f32x2 type; };
template <bool USE_LDS, int NS>
__global__ __launch_bounds__(1024) void siddon_fwd_kernel(const float *__restrict__ obj, SidGeom g,
                                                         const float *__restrict__ sin_t,
                                                         const float *__restrict__ cos_t,
                                                         const int *__restrict__ quad_t, int p_per_blk,
                                                         float *__restrict__ data)
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
        if constexpr (NS == 1) {
            data[((size_t)s * g.dt + p) * g.dx + d] = acc;
        } else {
            data[((size_t)s * g.dt + p) * g.dx + d] = acc.x;
            if (has2) data[((size_t)(s + 1) * g.dt + p) * g.dx + d] = acc.y;
        }
    }
}

// Back-projector: the transpose of the forward, recon[s][pixel] = sum over rays of data[s][p][d] * dist(p, d, pixel) -- what
// libtomo's fbp.c accumulates (recon[indi[n]] += data[ind_data] * dist[n]) and the A^T of sirt.c's update.  Atomic-free
// and bit-reproducible: a workgroup owns one slice and a group of angles and adds into its own image (LDS when the slice
// fits, the partial image in global memory otherwise) with plain read-add-writes, made conflict-free by construction:
//   * two rays of the SAME parity of d are two detector pitches apart, further than a pixel's diagonal: they never add
//     into the same pixel;
//   * every ray is cut into kChunks runs of consecutive segments (merge-path start, see siddon_walk_ray), one lane each, so
//     that a phase fills the workgroup instead of 92 lanes; a ray meets a pixel in consecutive segments only, so two runs
//     of one ray can share a pixel only where they touch: even-numbered and odd-numbered runs go in separate phases.
// Four phases per angle (ray parity x run parity), a barrier after each: every pixel receives its terms in a fixed order
// (angles ascending; even rays before odd; even runs before odd) whatever the launch; angle groups write partial images
// that siddon_reduce_groups_kernel adds in ascending group order.  Rays whose datum is 0 are skipped (x + 0 * dist == x): the
// sparse sinograms and dose masks this is fed (ctvae/helper_functions.py:489-516) are zero at most angles.
template <bool USE_LDS>
__global__ __launch_bounds__(1024) void siddon_bwd_kernel(const float *__restrict__ data, SidGeom g,
                                                         const float *__restrict__ sin_t, const float *__restrict__ cos_t,
                                                         const int *__restrict__ quad_t, int p_per_grp, int n_grp,
                                                         int kChunks, float *__restrict__ partial)
{
    extern __shared__ float lds[];
    const int s = blockIdx.y, grp = blockIdx.x;
    const int p0 = grp * p_per_grp;
    const int np = min(p_per_grp, g.dt - p0);
    const int npix = g.ox * g.oz;
    float *out = partial + ((size_t)s * n_grp + grp) * npix;
    const int pitch = USE_LDS ? g.oz + ((1 - (g.oz & 31)) & 31) : g.oz;
    float *img = USE_LDS ? lds : out;
    for (int t = threadIdx.x; t < g.ox * pitch; t += blockDim.x) img[t] = 0.0f;
    __syncthreads();
    const float *row = data + ((size_t)s * g.dt + p0) * g.dx;
    const int KH = kChunks / 2;              // runs per ray and phase (kChunks is even)
    const int half_rays = (g.dx + 1) >> 1;
    for (int pl = 0; pl < np; ++pl, row += g.dx) {
        const int p = p0 + pl;
        const float sin_p = sin_t[p], cos_p = cos_t[p];
        const int quadrant = quad_t[p];
#pragma unroll 1
        for (int phase = 0; phase < 4; ++phase) {
            const int par = phase >> 1, cpar = phase & 1;
            for (int t = threadIdx.x; t < half_rays * KH; t += blockDim.x) {
                const int rr = t / KH, d = 2 * rr + par;
                if (d >= g.dx) continue;
                const float v = row[d];
                if (v != 0.0f)
                    siddon_walk_ray<true>(g, sin_p, cos_p, quadrant, d,
                                          [&](int ix, int iy, float dist) { img[ix * pitch + iy] += v * dist; },
                                          2 * (t - rr * KH) + cpar, kChunks);
            }
            __syncthreads();   // (workgroup-scope release / acquire of the image, LDS or global)
        }
    }
    if (USE_LDS)
        for (int t = threadIdx.x; t < npix; t += blockDim.x) out[t] = lds[(t / g.oz) * pitch + (t % g.oz)];
}

// recon[s][pix] = sum over angle groups, ascending
__global__ __launch_bounds__(256) void siddon_reduce_groups_kernel(const float *__restrict__ partial, int n_grp, long long npix,
                                                                  long long total, float *__restrict__ recon)
{
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long s = e / npix, q = e - s * npix;
        float acc = 0.0f;
        for (int gq = 0; gq < n_grp; ++gq) acc += partial[(s * n_grp + gq) * npix + q];
        recon[e] = acc;
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
                          const float *cos_dev, const int *quad_dev, int dt, int dx, float center,
                          float *data_dev, ctpvae_stream_t stream);
static float siddon_mov(int dx, float center);

// slices are indexed with a grid dimension (<= 65535): a longer stack goes in chunks, back to back on the stream
int ctpvae_siddon_fwd_f32(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev,
                          const float *cos_dev, const int *quad_dev, int dt, int dx, float center,
                          float *data_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(obj_dev && data_dev && oy > 0 && ox > 0 && oz > 0 && dt > 0 && dx > 0, "siddon_fwd: null pointer or empty sizes");
    const int chunk = std::max(2, max_slices_per_launch() / 2 * 2);   // even: whole slice pairs per chunk
    for (int s0 = 0; s0 < oy; s0 += chunk) {
        const int n = std::min(chunk, oy - s0);
        if (int rc = siddon_fwd_one(obj_dev + (size_t)s0 * ox * oz, n, ox, oz, sin_dev, cos_dev, quad_dev, dt, dx, center,
                                    data_dev + (size_t)s0 * dt * dx, stream))
            return rc;
    }
    return CTPVAE_OK;
}

static int siddon_fwd_one(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev,
                          const float *cos_dev, const int *quad_dev, int dt, int dx, float center,
                          float *data_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(obj_dev && sin_dev && cos_dev && quad_dev && data_dev, "siddon_fwd: null pointer");
    CTPVAE_REQUIRE(oy > 0 && ox > 0 && oz > 0 && dt > 0 && dx > 0,
                   "siddon_fwd: sizes must be positive (oy=%d ox=%d oz=%d dt=%d dx=%d)", oy, ox, oz, dt, dx);
    CTPVAE_REQUIRE(oy <= 65535, "siddon_fwd: at most 65535 slices per call (got %d)", oy);
    const SidGeom g{oy, ox, oz, dt, dx, siddon_mov(dx, center)};
    const size_t lds_one = (size_t)ox * (oz + ((1 - (oz & 31)) & 31)) * sizeof(float);
    const bool use_lds = lds_one <= (size_t)kMaxLdsBytes;
    // two slices per workgroup when the pair fits LDS and the call has slices to pair
    int ns = (oy >= 2 && 2 * lds_one <= (size_t)kMaxLdsBytes) ? 2 : 1;
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
                           ppb, data_dev);
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

// angle groups per slice: enough workgroups to fill the chip (~512), at least 4 angles each
static int siddon_bwd_groups(int oy, int dt)
{
    const int want = ceil_div(512, std::max(1, oy));
    return std::max(1, std::min(want, ceil_div(dt, 4)));
}

long long ctpvae_siddon_bwd_workspace_bytes(int oy, int ox, int oz, int dt)
{
    if (oy <= 0 || ox <= 0 || oz <= 0 || dt <= 0) return fail(CTPVAE_EINVAL, "siddon_bwd_workspace_bytes: bad sizes");
    const int n_grp = siddon_bwd_groups(std::min(oy, 65535), dt);
    return n_grp > 1 ? (long long)std::min(oy, 65535) * n_grp * ox * oz * (long long)sizeof(float) : 0;
}

int ctpvae_siddon_bwd_f32(const float *data_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                          const int *quad_dev, int dt, int dx, float center, void *workspace_dev, float *recon_dev,
                          ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(data_dev && sin_dev && cos_dev && quad_dev && recon_dev, "siddon_bwd: null pointer");
    CTPVAE_REQUIRE(oy > 0 && ox > 0 && oz > 0 && dt > 0 && dx > 0,
                   "siddon_bwd: sizes must be positive (oy=%d ox=%d oz=%d dt=%d dx=%d)", oy, ox, oz, dt, dx);
    const int chunk = std::min(65535, max_slices_per_launch());
    const SidGeom g0{0, ox, oz, dt, dx, siddon_mov(dx, center)};
    const size_t lds_bytes = (size_t)ox * (oz + ((1 - (oz & 31)) & 31)) * sizeof(float);
    const bool use_lds = lds_bytes <= (size_t)kMaxLdsBytes;
    // runs per ray: 16 (tools/time_recon.py sweep) -- more shorten a lane's chain of dependent read-add-writes but repeat the
    // ray set-up (two bisections + the merge-path search); never more runs than a short ray has segments to share out
    int kChunks = 16;
    if (knob(kKnobSiddonBwdChunks) > 0) kChunks = std::max(2, std::min(64, knob(kKnobSiddonBwdChunks) / 2 * 2));
    int threads = std::min(1024, ceil_div(ceil_div(dx, 2) * (kChunks / 2), 64) * 64);   // one lane per (ray, run) of a phase
    if (knob(kKnobSiddonBwdThreads) > 0) threads = std::max(64, std::min(1024, knob(kKnobSiddonBwdThreads) / 64 * 64));
    const long long npix = (long long)ox * oz;
    for (int s0 = 0; s0 < oy; s0 += chunk) {
        const int n = std::min(chunk, oy - s0);
        const int n_grp = siddon_bwd_groups(std::min(oy, 65535), dt);   // one rule for every chunk: the workspace was sized with it
        CTPVAE_REQUIRE(n_grp == 1 || workspace_dev, "siddon_bwd: %d angle groups need the workspace", n_grp);
        const int p_per_grp = ceil_div(dt, n_grp);
        const int groups = ceil_div(dt, p_per_grp);
        SidGeom g = g0;
        g.oy = n;
        float *partial = groups > 1 ? (float *)workspace_dev : recon_dev + (size_t)s0 * npix;
        const float *data = data_dev + (size_t)s0 * dt * dx;
        auto launch = [&](auto kernel, size_t shmem) -> int {
            static std::atomic<unsigned long long> attr_set{0};
            if (shmem > 64 * 1024) CTPVAE_SET_MAX_LDS_ONCE(kernel, attr_set);
            hipLaunchKernelGGL(kernel, dim3(groups, n), dim3(threads), shmem, (hipStream_t)stream, data, g, sin_dev, cos_dev,
                               quad_dev, p_per_grp, groups, kChunks, partial);
            CTPVAE_LAUNCH_CHECK("siddon_bwd_kernel");
            return CTPVAE_OK;
        };
        if (int rc = use_lds ? launch(siddon_bwd_kernel<true>, lds_bytes) : launch(siddon_bwd_kernel<false>, 0)) return rc;
        if (groups > 1) {
            const long long total = (long long)n * npix;
            const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 256ll * 32);
            hipLaunchKernelGGL(siddon_reduce_groups_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, partial, groups, npix,
                               total, recon_dev + (size_t)s0 * npix);
            CTPVAE_LAUNCH_CHECK("siddon_reduce_groups_kernel");
        }
    }
    return CTPVAE_OK;
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
