// rotate_cplan.hip -- COMPACT (step-coded) gather plans for the NEAREST rotate-and-sum forward (gfx950).
//
// rotate_plan.hip stores every tap of a geometry as a u16 LDS index: 2 B per sample, 1.9 MB at 20 angles and 17 MB at the
// dataset's 180 -- more than an XCD's 4 MB L2, so at many angles the per-slice kernel is bound by the index stream it
// pulls through L2 (nine times its algorithmic bytes at A = 180, profiles/r03_angles180_u16_traffic_pmc.json).  But along a ray the tap
// moves, from one canvas row to the next, by one of FOUR cell deltas: x_in and y_in (ctvae/forward_functions.py:113 ->
// tfa.image.rotate -> ImageProjectiveTransformV3) are monotone in the row number with slope |t1|, |t4| <= 1,
// so round(x_in) and round(y_in) each stay or step by one, always the same way for an angle.  A ray is therefore
//     its first live canvas row's tap  +  2 bits per row  (bit 0: the column steps, bit 1: the row steps),
// and its live rows are one interval.  The plan kernel below still evaluates the reference arithmetic EXACTLY (unfused
// fp32, std::round, zero fill) -- it only stores the differences: 1/3 B per sample, 0.3 MB at 20 angles, 2.3 MB at 180.
//
// Decoding: one code BYTE holds the steps of THREE rows (6 bits) and indexes a 64-entry table in LDS whose entry holds the
// cumulative byte offsets (i16) after one, two and three of those steps; an SDWA add with a sign-extended word select both
// extracts an offset and adds it to the ray's running address.  Per three taps: 1 SDWA move (table address), 1 ds_read_b64
// (table entry), 3 SDWA adds -- against 3 SDWA shifts for u16 taps.  The table is kept in 32 copies, one per lane of a
// half-wave (entry e of copy l at byte e * 256 + l * 8): whatever entries the 32 lanes of a ds_read_b64 group ask for,
// they sit in 32 different bank pairs, so a table read is always one LDS pass.  (Round 3, measured: with four rows per
// byte and ONE 2 KB table the lanes' different entries collided -- 5.3 LDS cycles per table read, 64 % more LDS cycles
// than the u16 kernel, which made the kernel LDS-bound and 3-10 % slower than the u16 plan at every large shape;
// 256 entries x 32 copies would not fit beside a slice pair, 64 x 32 x 8 B = 16 KB does.)
//
// Rows behind a ray's last live row must add +0.0f.  The slice is staged with a ONE-CELL ZERO BORDER (a zero row above and
// below, the spare column(s) of the odd row pitch as the gutter between rows): the step that leaves the core lands on a
// border cell, and all later codes of the ray are 0 (stay).  Rays of a 64-bin block start at their own first live row
// (as in rotate_plan.hip), all walk 6 * ng rows, ng = the block's longest ray in groups of six (two code bytes).
//
// Same taps, same ascending row order => bit-identical to rotate_plan.hip, rotate.hip and the oracle.  A geometry
// whose steps do not fit the code (rows that are not a rotation, a ray still inside the core at the canvas' last row,
// rounding ties that make a coordinate jump by two) raises the plan's overflow word; the caller then keeps the u16 plan.
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "common.h"
#include "lds_stage.h"
#include "loglik_math.h"
#include "rotate_plan.h"
#include "cplan_walk.h"

namespace ctpvae {

constexpr int kCSelRounds = 4;    // angle subsets hold <= 64 * kCSelRounds angles

struct CLayout {
    int nJB, PWpad, NQ, pitch, cells;
    long long off_cls, off_clist, off_ng, off_start, off_codes, off_flag, bytes;
};

static CLayout c_layout(const PlanGeom &g)
{
    CLayout L;
    L.nJB = num_bin_blocks(g.PW);
    L.PWpad = L.nJB * 64;
    L.NQ = ceil_div(g.PH, kRowsPerChunk);           // uint4 code chunks per ray: 48 rows each
    L.pitch = pitch_mod32_is_1(g.W + 1);            // >= W + 1: at least one gutter column
    L.cells = 1 + (g.H + 2) * L.pitch + 1;          // guard cell, border row, H rows, border row, guard
    auto up = [](long long v) { return (v + 255) / 256 * 256; };
    L.off_cls = 0;                                   // per angle: bit 0 = mirror class (see rotate_plan.hip), bit 1 = sigma < 0
    L.off_clist = (long long)g.A * 4;                // two lists of (count, angles...)
    L.off_ng = up(L.off_clist + 2ll * (g.A + 1) * 4);          // [A][nJB]: row groups (of 6) a task walks
    L.off_start = up(L.off_ng + (long long)g.A * L.nJB * 4);   // [A][PWpad]: first tap's cell in the bordered image
    L.off_codes = up(L.off_start + (long long)g.A * L.PWpad * 4);
    L.off_flag = L.off_codes + (long long)g.A * L.NQ * L.PWpad * 16;
    L.bytes = L.off_flag + 256;
    return L;
}
static size_t c_lds_bytes(const CLayout &L, int ns) { return (size_t)kLutBytes + (size_t)L.cells * 4 * ns + 16; }
static bool cplan_fits(const PlanGeom &g, int ns = 1)
{
    const CLayout L = c_layout(g);
    // three steps of at most one row each must fit the table's i16 byte offsets
    return c_lds_bytes(L, ns) <= (size_t)kMaxLdsBytes && 12ll * L.pitch * ns <= 32767;
}

// ---- plan builder ------------------------------------------------------------------------------------------------
// one wave per (bin block, angle)
__global__ __launch_bounds__(64) void rotate_cplan_kernel(PlanGeom g, const float *__restrict__ T8, CLayout L,
                                                          char *__restrict__ plan)
{
    const int a = blockIdx.y, jb = blockIdx.x, lane = threadIdx.x;
    const int j = lane_to_bin(g.PW, jb, lane);
    const float *t = T8 + 8 * a;
    int *cls = reinterpret_cast<int *>(plan + L.off_cls);
    int *ngt = reinterpret_cast<int *>(plan + L.off_ng);
    unsigned *start = reinterpret_cast<unsigned *>(plan + L.off_start);
    uint4 *codes = reinterpret_cast<uint4 *>(plan + L.off_codes);
    int *flag = reinterpret_cast<int *>(plan + L.off_flag);
    if (jb == 0 && lane == 0) cls[a] = cplan_class_word(t);
    const RayScan rs = cplan_scan_ray(g, t, j, (unsigned)j < (unsigned)g.PW, L.pitch);
    const int ng = (wave_max_i(rs.n) + kRowsPerGroup - 1) / kRowsPerGroup;
    if (lane == 0) ngt[a * L.nJB + jb] = ng;
    start[(size_t)a * L.PWpad + jb * 64 + lane] = (unsigned)rs.start;
    const bool bad = cplan_encode_ray(g, t, j, rs, kRowsPerGroup * ng, L.pitch, L.NQ,
                                      codes + (size_t)a * L.NQ * L.PWpad + jb * 64 + lane, (size_t)L.PWpad);
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

// clist[c] = (count, the angles of mirror class c in ascending order)
__global__ __launch_bounds__(64) void rotate_cplan_class_list_kernel(int A, CLayout L, char *__restrict__ plan)
{
    const int *cls = reinterpret_cast<const int *>(plan + L.off_cls);
    int *clist = reinterpret_cast<int *>(plan + L.off_clist);
    const int lane = threadIdx.x;
    for (int c = 0; c < 2; ++c) {
        int *list = clist + c * (A + 1);
        int n = 0;
        for (int a0 = 0; a0 < A; a0 += 64) {
            const bool in = a0 + lane < A && (cls[a0 + lane] & 1) == c;
            const unsigned long long m = __ballot(in);
            if (in) list[1 + n + __popcll(m & ((1ull << lane) - 1ull))] = a0 + lane;
            n += __popcll(m);
        }
        if (lane == 0) list[0] = n;
    }
}

// ---- executing a compact plan ---------------------------------------------------------------------------------------
// Forward.  Workgroup = (slice or slice pair, mirror class, task group), placed as in rotate_fwd_planned_kernel; stages the
// unit with its zero border, builds the step table, then its waves take (angle, bin block) tasks.
// EPI: 0 = ray-sums only; 1 = + log-probabilities (and d lp / d ray-sum) of the measured samples; 2 = the
// log-probabilities are REDUCED: one partial sum per task into epi.part (see LogLikEpilogue), d lp / d ray-sum stored for the
// backward, the ray-sum and log-probability stores only where buffers were given.
// SELM: 0 = all plan angles; 1 = the launch projects a subset of the plan's angles (see rotate_fwd_planned_kernel), read
// from device memory (sel); 2 = the subset arrived in host memory and travels in the kernel arguments (selh).
template <int NS, int EPI, int SELM>
__global__ __launch_bounds__(1024) void rotate_fwd_compact_kernel(const float *__restrict__ img, PlanGeom g, CLayout L,
                                                                  const char *__restrict__ plan, int wgs_per_slice, int g_S,
                                                                  float *__restrict__ sino, LogLikEpilogue epi,
                                                                  const int *__restrict__ sel, int n_sel, SelHost selh)
{
    constexpr bool SEL = SELM != 0;
    typedef typename SliceVec<NS>::type vec_t;
    extern __shared__ float lds[];
    float *image = lds + kLutBytes / 4;   // cell c of slice n: image[c * NS + n]
    const int units = (g_S + NS - 1) / NS;
    int u, wg;
    {
        const int per8 = 8 * wgs_per_slice, octet = blockIdx.x / per8, rem = blockIdx.x - octet * per8;
        if ((octet + 1) * 8 <= units) {
            wg = rem >> 3;
            u = octet * 8 + (rem & 7);
        } else {   // the last, partial octet is laid out unit-major
            u = octet * 8 + rem / wgs_per_slice;
            wg = rem % wgs_per_slice;
        }
    }
    const int s = u * NS;
    const bool has2 = NS == 2 && s + 1 < g_S;     // an odd batch ends with a half-empty pair (slice s staged twice)
    const int c = wg & 1, gi = wg >> 1, G = wgs_per_slice >> 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const float *im = img + (size_t)s * g.H * g.W;

    const int *cls = reinterpret_cast<const int *>(plan + L.off_cls);
    const int *clist = reinterpret_cast<const int *>(plan + L.off_clist) + c * (g.A + 1);
    const int *ngt = reinterpret_cast<const int *>(plan + L.off_ng);
    const unsigned *start = reinterpret_cast<const unsigned *>(plan + L.off_start);
    const uint4 *codes = reinterpret_cast<const uint4 *>(plan + L.off_codes);
    const int A_out = SEL ? n_sel : g.A;   // rows of an output sinogram
    int sel_a[kCSelRounds], sel_rank[kCSelRounds], sel_cum[kCSelRounds + 1], sel_neg[kCSelRounds];
    int ncls;
    if constexpr (SEL) {
        sel_cum[0] = 0;
#pragma unroll
        for (int r = 0; r < kCSelRounds; ++r) {
            const int k = 64 * r + lane;
            int a = 0, cl = -1, w = 0;
            if (k < n_sel) {
                a = min(max(SELM == 2 ? selh.get(k) : sel[k], 0), g.A - 1);   // a bad index cannot leave the plan
                w = cls[a];
                cl = w & 1;
            }
            const unsigned long long m = __ballot(cl == c);
            sel_a[r] = a;
            sel_neg[r] = w >> 1;
            sel_rank[r] = cl == c ? (int)__popcll(m & ((1ull << lane) - 1ull)) : -1;
            sel_cum[r + 1] = sel_cum[r] + (int)__popcll(m);
        }
        ncls = sel_cum[kCSelRounds];
    } else {
        ncls = clist[0];
    }
    const int ntask = ncls * L.nJB;
    const size_t st = (size_t)L.PWpad;
    struct Task {
        bool valid, neg;
        int a, k, j, jb, ng, adr;   // plan angle, output row, bin, bin block, row groups, first tap's LDS byte address
        const uint4 *p;         // the ray's chunk 2
        uint4 c0, c1;
        float em[NS], ex[NS];   // EPI: mask entry and measured sample of the task's outputs, requested with the task
    };
    [[maybe_unused]] float epnm = 0.0f, einv = 0.0f;
    if constexpr (EPI != 0) epnm = *epi.pnm, einv = 1.0f / epnm;   // the derivative multiplies by the reciprocal (loglik_math.h)
    auto prepare = [&](int m) -> Task {
        Task t;
        t.valid = m < ntask;
        t.neg = false;
        t.a = t.k = t.j = t.jb = t.ng = 0;
        t.adr = kLutBytes;
        t.p = codes;
        t.c0 = t.c1 = uint4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int n = 0; n < NS; ++n) t.em[n] = t.ex[n] = 0.0f;
        if (t.valid) {   // wave-uniform
            const int jb = m / ncls, ai = m - jb * ncls;
            t.jb = jb;
            if constexpr (SEL) {
#pragma unroll
                for (int r = 0; r < kCSelRounds; ++r)
                    if (ai >= sel_cum[r] && ai < sel_cum[r + 1]) {   // wave-uniform
                        const unsigned long long hit = __ballot(sel_rank[r] == ai - sel_cum[r]);
                        const int l = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)hit) - 1);
                        t.a = __builtin_amdgcn_readlane(sel_a[r], l);
                        t.neg = __builtin_amdgcn_readlane(sel_neg[r], l) != 0;
                        t.k = 64 * r + l;
                    }
            } else {
                t.k = t.a = clist[1 + ai];
                t.neg = (cls[t.a] >> 1) != 0;
            }
            t.a = __builtin_amdgcn_readfirstlane(t.a);
            t.k = __builtin_amdgcn_readfirstlane(t.k);
            t.ng = __builtin_amdgcn_readfirstlane(ngt[t.a * L.nJB + jb]);
            t.j = lane_to_bin(g.PW, jb, lane);
            t.adr = kLutBytes + (int)start[(size_t)t.a * L.PWpad + jb * 64 + lane] * (4 * NS);
            const uint4 *p = codes + (size_t)t.a * L.NQ * L.PWpad + jb * 64 + lane;
            t.c0 = p[0];
            if (L.NQ > 1) t.c1 = p[st];
            t.p = p + 2 * st;
            if constexpr (EPI != 0) {   // (loaded in the epilogue these cost the wave a round trip behind its walk)
                const int jc = min(max(t.j, 0), g.PW - 1);   // dead lanes (bins off the detector) read a live one's sample
#pragma unroll
                for (int n = 0; n < NS; ++n) {
                    const int sl = min(s + n, g_S - 1);
                    // measured samples and masks: compact like the outputs, or the dense arrays read at the plan angle
                    const size_t sa = SEL && epi.dense ? (size_t)sl * g.A + t.a : (size_t)sl * A_out + t.k;
                    t.em[n] = epi.mask[sa];
                    t.ex[n] = epi.meas[sa * g.PW + jc];
                }
            }
        }
        return t;
    };
    Task cur = prepare(gi + G * wave);
    int *next_task = reinterpret_cast<int *>(image + (size_t)L.cells * NS);
    if (threadIdx.x == 0) *next_task = nwaves;

    cplan_init_lut<NS>(lds, L.pitch);
    cplan_zero_border<NS>(image, g.H, g.W, L.pitch);
    float *core = image + (size_t)(1 + L.pitch) * NS;   // cell of (iy = 0, column 0)
    {   // (both slices of a pair in one load round trip, written as float2; 64 / 128 / 256-column slices in the lean form)
        const float *srcs[NS];
        srcs[0] = im;
        if constexpr (NS == 2) srcs[1] = im + (has2 ? (size_t)g.H * g.W : 0);
        stage_unit<NS>(core, srcs, g.H, g.W, g.W, L.pitch, c == 0, lane, wave, nwaves);
    }
    __syncthreads();

    while (cur.valid) {
        int m = 0;
        if (lane == 0) m = atomicAdd(next_task, 1);
        const Task nxt = prepare(__builtin_amdgcn_readfirstlane(m) * G + gi);

        vec_t acc = 0.0f;
        const int ng = __builtin_amdgcn_readfirstlane(cur.ng);
        const bool neg = __builtin_amdgcn_readfirstlane((int)cur.neg) != 0;
        if (ng > 0)
            acc = neg ? cwalk<NS, true>(cur.adr, ng, (lane & 31) << 3, cur.c0, cur.c1, cur.p, st, L.NQ)
                      : cwalk<NS, false>(cur.adr, ng, (lane & 31) << 3, cur.c0, cur.c1, cur.p, st, L.NQ);
        if constexpr (EPI == 2) {
            const bool live = (unsigned)cur.j < (unsigned)g.PW;
            auto reduce = [&](int n, float v) {
                const int sl = s + n;
                float lpv = 0.0f;
                if (live) {
                    const size_t o = ((size_t)sl * A_out + cur.k) * g.PW + cur.j;
                    if (sino) sino[o] = v;
                    lpv = epi.eval_loaded(o, cur.em[n], cur.ex[n], epnm, einv, v);
                }
                const float tot = wave_sum(lpv);
                if (lane == 0) epi.store_part(((size_t)sl * A_out + cur.k) * L.nJB + cur.jb, tot);
            };
            if constexpr (NS == 1) {
                reduce(0, acc);
            } else {
                reduce(0, acc.x);
                if (has2) reduce(1, acc.y);
            }
        } else if ((unsigned)cur.j < (unsigned)g.PW) {
            auto store = [&](int n, float v) {
                const size_t o = ((size_t)(s + n) * A_out + cur.k) * g.PW + cur.j;
                sino[o] = v;
                if constexpr (EPI == 1) epi.write_loaded(o, cur.em[n], cur.ex[n], epnm, einv, v);
            };
            if constexpr (NS == 1) {
                store(0, acc);
            } else {
                store(0, acc.x);
                if (has2) store(1, acc.y);
            }
        }
        cur = nxt;
    }
    if constexpr (EPI == 2) {
        // round 4: the workgroup that finishes this unit last adds its slices' partials in the fixed order -- no second launch
        if (epi.sum != nullptr && arrived_last(epi.arrive + u, (unsigned)wgs_per_slice, next_task + 1)) {
            if (wave < NS && s + wave < g_S) {
                const float total = object_sum_of_parts<true>(epi.part + (size_t)(s + wave) * A_out * L.nJB, A_out, L.nJB, lane);
                if (lane == 0) epi.sum[s + wave] = total;
            }
        }
    }
}

}  // namespace ctpvae

using namespace ctpvae;

extern "C" {

int ctpvae_rotate_cplan_supported(int H, int W, int PH, int PW, int A, int interp)
{
    if (interp != CTPVAE_NEAREST || H <= 0 || W <= 0 || PH < H || PW < W || A <= 0) return 0;
    if (knob(kKnobNoPlan) >= 0 || knob(kKnobNoCompact) >= 0 || knob(kKnobTiledForce) == 1) return 0;
    return cplan_fits(PlanGeom{H, W, PH, PW, 0, 0, A}) ? 1 : 0;
}

long long ctpvae_rotate_cplan_bytes(int H, int W, int PH, int PW, int A)
{
    if (H <= 0 || W <= 0 || PH < H || PW < W || A <= 0) return fail(CTPVAE_EINVAL, "rotate_cplan_bytes: bad sizes");
    return c_layout(PlanGeom{H, W, PH, PW, 0, 0, A}).bytes;
}

int ctpvae_rotate_cplan_build_f32(const float *T8_dev, int A, int H, int W, int PH, int PW, int py, int px, void *cplan_dev,
                                  ctpvae_stream_t stream)
{
    if (int rc = check_plan_geom("rotate_cplan_build", H, W, PH, PW, py, px, A)) return rc;
    CTPVAE_REQUIRE(T8_dev && cplan_dev, "rotate_cplan_build: null pointer");
    const PlanGeom g{H, W, PH, PW, py, px, A};
    CTPVAE_REQUIRE(cplan_fits(g), "rotate_cplan_build: a %dx%d slice does not fit the compact plan's LDS image", H, W);
    CTPVAE_REQUIRE(A <= 65535, "rotate_cplan_build: at most 65535 angles");
    const CLayout L = c_layout(g);
    CTPVAE_HIP(hipMemsetAsync((char *)cplan_dev + L.off_flag, 0, 256, (hipStream_t)stream));
    hipLaunchKernelGGL(rotate_cplan_kernel, dim3(L.nJB, A), dim3(64), 0, (hipStream_t)stream, g, T8_dev, L, (char *)cplan_dev);
    CTPVAE_LAUNCH_CHECK("rotate_cplan_kernel");
    hipLaunchKernelGGL(rotate_cplan_class_list_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, L, (char *)cplan_dev);
    CTPVAE_LAUNCH_CHECK("rotate_cplan_class_list_kernel");
    return CTPVAE_OK;
}

// 1 if some ray's steps do not fit the code: keep the u16 plan (ctpvae_rotate_plan_build_f32).  SYNCHRONISES the stream.
int ctpvae_rotate_cplan_overflowed(const void *cplan_dev, int H, int W, int PH, int PW, int A, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(cplan_dev && H > 0 && W > 0 && A > 0 && PH >= H && PW >= W, "rotate_cplan_overflowed: bad arguments");
    const CLayout L = c_layout(PlanGeom{H, W, PH, PW, 0, 0, A});
    int flag = 0;
    CTPVAE_HIP(hipMemcpyAsync(&flag, (const char *)cplan_dev + L.off_flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    CTPVAE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return flag ? 1 : 0;
}

int ctpvae_rotate_fwd_compact_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int A, const void *cplan_dev,
                                  const int *angle_idx, int n_idx, int idx_on_host, const float *mask_dev,
                                  const float *meas_dev, int dense_inputs, const float *pnm_dev, float eps, float *sino_dev,
                                  float *lp_dev, float *dlp_dev, float *lp_part_dev, float *lp_sum_dev, ctpvae_stream_t stream)
{
    const int *angle_idx_dev = angle_idx;
    const bool red = lp_sum_dev != nullptr;
    // round 4, built and measured NEGATIVE (tools/time_fold.py, profiles/r04_time_fold.txt): with the knob FOLD_SUMS = 1 the
    // ordered sum of a slice's partials happens inside this launch -- the partials' workspace ends with one arrival counter per
    // slice (ctpvae_loglik_part_floats), zero before its first use and left zero by every launch; the workgroup that finishes a
    // unit last re-reads the partials at agent scope and adds them.  Same bits, one launch less -- and 0.3 us MORE per training
    // call (10.7 vs 10.4 us), 2 us more on config 5: the arrival add and the re-read are two fabric round trips on the last
    // workgroup's critical path, a dependent launch in a stream costs less.  Default: round 3's loglik_sum_partials_kernel launch.
    const bool fold = red && knob(kKnobFoldSums) == 1;
    CTPVAE_REQUIRE(img_dev && cplan_dev && (sino_dev || red), "rotate_fwd_compact: null pointer");
    CTPVAE_REQUIRE(!red || lp_part_dev, "rotate_fwd_compact: per-object sums need the partial-sum workspace");
    CTPVAE_REQUIRE(S > 0, "rotate_fwd_compact: need at least one slice");
    if (int rc = check_plan_geom("rotate_fwd_compact", H, W, PH, PW, 0, 0, A)) return rc;
    const int *sel_dev = angle_idx_dev;   // (host memory when idx_on_host)
    CTPVAE_REQUIRE(sel_dev == nullptr || (n_idx >= 1 && n_idx <= 64 * kCSelRounds),
                   "rotate_fwd_compact: an angle subset holds 1..%d angles (got %d); build a plan for larger ones",
                   64 * kCSelRounds, n_idx);
    const bool lik = lp_dev != nullptr || red;
    CTPVAE_REQUIRE(!lik || (mask_dev && meas_dev && pnm_dev), "rotate_fwd_compact: the likelihood epilogue needs mask, meas and pnm");
    CTPVAE_REQUIRE(lik || dlp_dev == nullptr, "rotate_fwd_compact: dlp without lp");
    const PlanGeom g{H, W, PH, PW, 0, 0, A};
    CTPVAE_REQUIRE(cplan_fits(g), "rotate_fwd_compact: a %dx%d slice does not fit the compact plan's LDS image", H, W);
    const CLayout L = c_layout(g);
    SelHost selh = {};
    const int selm = sel_dev == nullptr ? 0 : (idx_on_host ? 2 : 1);
    if (selm == 2) {   // host indices: clamped here, carried in the kernel arguments
        for (int k = 0; k < n_idx; ++k) selh.set(k, std::min(std::max(sel_dev[k], 0), A - 1));
        sel_dev = nullptr;
    }
    const int A_run = selm != 0 ? n_idx : A;    // angles this launch projects
    LogLikEpilogue epi = lik ? LogLikEpilogue{mask_dev, meas_dev, pnm_dev, eps, lp_dev, dlp_dev, dense_inputs ? 1 : 0,
                                              red ? lp_part_dev : nullptr}
                             : LogLikEpilogue{};
    if (fold) {
        epi.sum = lp_sum_dev;
        epi.arrive = reinterpret_cast<unsigned *>(lp_part_dev + (size_t)S * A_run * L.nJB);
    }
    const int T = A_run * L.nJB;                // (angle, bin block) tasks per slice
    // Launch shape.  With 0.25 B of plan per sample the index stream no longer counts; what a workgroup costs is the fill of
    // its unit (a slice, or a pair of slices interleaved as float2) and its tasks, which are bound by the CU's LDS pipe when
    // all 16 waves gather (10 LDS instructions of ~3.7 cycles per group of eight rows, the same for one slice or a pair)
    // and by the latency of a wave's chain of dependent row groups when few waves do (measured on the MI355X, round 3:
    // B=50 A=20 pairs G=5 7.4 us, G=4 7.8, G=3 8.1, singles 9.1; A=180 pairs G=5 22.5, G=4 26.2, G=10 26.3).  In microseconds:
    const bool pairs_fit = cplan_fits(g, 2) && S >= 2;
    int ns = 1, G = 1;
    {
        const double ngroups = 0.6 * ceil_div(PH, 8);             // row groups (of eight) of an average task
        const double task_lds_us = ngroups * 0.0176, chain_us = ngroups * 0.075;
        double best = 0.0;
        for (int cand_ns = 1; cand_ns <= (pairs_fit ? 2 : 1); ++cand_ns) {
            const double fill_us = 0.6 + 0.011 * ((double)g.H * g.W * 4.0 * cand_ns / 1024.0);
            const long long cand_units = (S + cand_ns - 1) / cand_ns;
            for (int cand = 1; cand <= std::min(12, std::max(1, T / 2)); ++cand) {
                const long long wgs = 2ll * cand_units * cand;
                const double tasks_wg = T / (2.0 * cand);
                // (several rounds: each costs ~3 us of ramp and drain on top, and r.x rounds take nearer to ceil(r.x) than to r.x
                // of them -- round 4, fitted to tools/sweep_fwd.py at 100 to 400 slices: profiles/r04_sweep_G.txt)
                const double real = (double)wgs / 256.0, whole = std::ceil(real);
                const double rounds = whole > 1.0 ? real + 0.75 * (whole - real) : 1.0;
                const double cost = rounds * (fill_us + 3.0 + std::max(tasks_wg * task_lds_us, std::ceil(tasks_wg / 16.0) * chain_us));
                if (best == 0.0 || cost < best || (cost == best && cand_ns == ns)) {   // ties (chain-bound launches) go to more
                    best = cost;                                                        // workgroups: S=10, 20 of 180 angles
                    ns = cand_ns;                                                       // G=8 7.78 us, G=10 7.58, G=12 7.61
                    G = cand;
                }
            }
        }
    }
    if (knob(kKnobNs) >= 0) ns = (knob(kKnobNs) == 2 && pairs_fit) ? 2 : 1;
    if (knob(kKnobG) > 0) G = knob(kKnobG);
    const size_t shmem = c_lds_bytes(L, ns);
    const int units = (S + ns - 1) / ns;
    // never fewer waves than stage the unit in ONE batch of eight 16-byte loads per lane (see launch_fwd_planned)
    const int stage_waves = (int)std::min<size_t>(16, ceil_div((int)((size_t)g.H * g.W * 4 * ns / 1024), 8));
    int waves = std::min(16, std::max(stage_waves, (T + 2 * G - 1) / (2 * G)));
    if (knob(kKnobWaves) > 0) waves = std::min(16, std::max(1, knob(kKnobWaves)));
    const int wgs_per_slice = 2 * G;
    CTPVAE_REQUIRE((long long)units * wgs_per_slice < (1ll << 31), "rotate_fwd_compact: too many slices");
    auto launch = [&](auto kernel) -> int {
        static std::atomic<unsigned long long> attr_set{0}, abs_ok{0};   // per kernel instantiation: devices done
        CTPVAE_REQUIRE_NO_STATIC_LDS(kernel, "rotate_fwd_compact_kernel", abs_ok);   // the step table sits at LDS address 0 (lut_issue)
        CTPVAE_SET_MAX_LDS_ONCE(kernel, attr_set);
        hipLaunchKernelGGL(kernel, dim3((unsigned)(units * wgs_per_slice)), dim3(64 * waves), shmem, (hipStream_t)stream,
                           img_dev, g, L, (const char *)cplan_dev, wgs_per_slice, S, sino_dev, epi, sel_dev, n_idx, selh);
        CTPVAE_LAUNCH_CHECK("rotate_fwd_compact_kernel");
        return CTPVAE_OK;
    };
    // (slices per workgroup, epilogue, subset mode) -> instantiation
    auto by_sel = [&](auto ns_tag, auto epi_tag) -> int {
        constexpr int NS_ = decltype(ns_tag)::value, EPI_ = decltype(epi_tag)::value;
        if (selm == 0) return launch(rotate_fwd_compact_kernel<NS_, EPI_, 0>);
        if (selm == 1) return launch(rotate_fwd_compact_kernel<NS_, EPI_, 1>);
        return launch(rotate_fwd_compact_kernel<NS_, EPI_, 2>);
    };
    auto by_ns = [&](auto epi_tag) -> int {
        return ns == 2 ? by_sel(std::integral_constant<int, 2>{}, epi_tag) : by_sel(std::integral_constant<int, 1>{}, epi_tag);
    };
    if (red) {
        if (int rc = by_ns(std::integral_constant<int, 2>{})) return rc;
        if (fold) return CTPVAE_OK;
        hipLaunchKernelGGL(loglik_sum_partials_kernel, dim3(S), dim3(64), 0, (hipStream_t)stream, lp_part_dev, S,
                           A_run, L.nJB, lp_sum_dev);
        CTPVAE_LAUNCH_CHECK("loglik_sum_partials_kernel");
        return CTPVAE_OK;
    }
    return lik ? by_ns(std::integral_constant<int, 1>{}) : by_ns(std::integral_constant<int, 0>{});
}

}  // extern "C"
