// rotate_plan.hip -- gather plans for the NEAREST rotate-and-sum projector and its TensorFlow-compatible
// backward, and the kernels that execute them (gfx950).
//
// Why a plan.  The tap a sample reads depends on (angle, canvas row, detector bin) only -- not on the object.
// CT_PVAE projects batches (50 objects per step in the README recipe), so evaluating TensorFlow's index
// arithmetic per object repeats the same ~20 VALU ops per sample 50 times, and on gfx950 everything except a
// plain fp32 add/mul (conversions, VOP3, packed ops) issues at a quarter of the lane rate: the direct kernels
// in rotate.hip are VALU-issue bound with HBM and LDS nearly idle (tools/probe_valu.hip, profiles/).  A plan is
// the table of LDS tap indices (u16), written ONCE per geometry by a kernel that evaluates the reference
// arithmetic exactly (unfused fp32, round half away from zero, zero fill); the per-object kernels then only
// stream indices (16 B per lane per load, coalesced, L2-resident), gather from LDS and add in row order.  The
// result is bit-identical to the direct kernels and to the oracle, because the same indices are used and every
// ray still adds its rows in ascending order (skipped rows and dead taps contribute exactly +0.0f).
//
// Forward plan   first[a][j]  : the first canvas row at which ray (a, j) is inside the H x W core (its own entry row).
//                idx[a][g][j] : uint4 = the 8 taps of rows first + 8g .. first + 8g + 7 of ray (a, j), each a dword
//                               index into the staged slice (row pitch == 1 mod 32), or `zero` (a cell holding 0.0f)
//                               if the tap is outside the core or the row is past the canvas.
//                rng[a][jb]   : first and last row group that any of the 64 rays of bin block jb needs -- the
//                               kernel's wave-uniform loop bounds.
//                clist[c]     : (count, angles of class c ascending) -- the planned kernel's task lists.
//                cls[a]       : 1 if lanes walk the slice with column and row moving the same way, else 0; for
//                               class 0 the slice is staged (and indexed) column-mirrored, so that consecutive
//                               detector bins always advance by |dx| + |dy| in [1, 1.42] LDS banks: at most two
//                               lanes of a 32-lane group share a bank at any angle.
// Backward plan  idx[a16][y][x]: uint4 = for angles 16*a16..16*a16+15, one BYTE each: the detector bin that
//                               TensorFlow's gradient op reads for pixel (y, x), or 255 (a cell holding 0.0f).
#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "lds_stage.h"
#include "loglik_math.h"
#include "rotate_plan.h"
#include "tune_stamps.h"

namespace ctpvae {

struct FwdLayout {
    int nJB, PWpad, NG, Galloc, pitch, zero, skew0;
    long long off_cls, off_clist, off_first, off_rng, off_idx, bytes;
};
constexpr int kBwdPitch = 257;   // dwords per staged cotangent row (== 1 mod 32; bins, then zeros up to cell 256)
constexpr int kBwdChunk = 64;    // angles staged per pass: row offsets (<= 63 * 1028 B) fit ds_read's 16-bit immediate
struct BwdLayout {
    int nXB, Wpad, NA16, pitchg, chunkA;   // chunkA: angles staged per pass (multiple of 16)
    long long bytes;
};

static FwdLayout fwd_layout(const PlanGeom &g)
{
    FwdLayout L;
    L.nJB = num_bin_blocks(g.PW);
    L.PWpad = L.nJB * 64;
    L.NG = ceil_div(g.PH, 8);
    L.Galloc = L.NG + 8;  // dead groups behind the canvas: the kernel prefetches up to 7 groups past a range
    // EXPERIMENT (developer knob SKEW0, DESIGN.md section 9 "LDS bank conflicts"): row pitch == 0 (mod 32), no mirroring, and
    // lane l of a half-wave delayed by round(alpha * l) rows so that consecutive lanes' taps step by one 8-byte slot --
    // tools/sim_lds_conflicts.py's "skew k=0", which only helps angles within 45 degrees of the row direction.  The plan is
    // then built for measurements on such angle sets; results stay exact (delayed rows read the zero cell).
    L.skew0 = knob(kKnobSkew0) > 0 ? 1 : 0;
    L.pitch = L.skew0 ? (g.W + 31) / 32 * 32 : pitch_mod32_is_1(g.W);
    if (L.skew0) L.Galloc += 6;   // delayed lanes walk up to ~40 rows more
    L.zero = g.H * L.pitch;
    L.off_cls = 0;
    L.off_clist = (long long)g.A * 4;                                     // two lists of (count, angles...)
    L.off_first = (L.off_clist + 2ll * (g.A + 1) * 4 + 8 + 255) / 256 * 256;   // (+ the classes' division words: fwd_magic) first live canvas row of every ray
    L.off_rng = (L.off_first + (long long)g.A * L.PWpad * 4 + 255) / 256 * 256;
    L.off_idx = (L.off_rng + (long long)g.A * L.nJB * 8 + 255) / 256 * 256;
    L.bytes = L.off_idx + (long long)g.A * L.Galloc * L.PWpad * 16;
    return L;
}
// Backward taps are detector bins: one BYTE per (angle, pixel), 255 = dead.  A cotangent row is staged as 257 dwords
// (== 1 mod 32): bins 0..PW-1, then zeros up to cell 256, so a dead tap reads row cell 255 = 0.0f and needs no select.
// dup = 2: the EXACT-transpose plan -- two taps per (angle, pixel), stored as "virtual angles" 2a and 2a + 1 that both read
// the staged cotangent row of angle a (see rotate_exact_plan_kernel)
static BwdLayout bwd_layout(const PlanGeom &g, int dup = 1)
{
    BwdLayout L;
    L.nXB = ceil_div(g.W, 64);
    L.Wpad = L.nXB * 64;
    L.NA16 = ceil_div(g.A * dup, 16);
    L.pitchg = kBwdPitch;
    L.chunkA = std::min(L.NA16 * 16, kBwdChunk);
    L.bytes = (long long)L.NA16 * g.H * L.Wpad * 16;
    return L;
}
static bool fwd_plan_fits(const PlanGeom &g)
{
    const FwdLayout L = fwd_layout(g);
    return L.zero < 65535 && (size_t)(L.zero + 1) * 4 + 16 <= (size_t)kMaxLdsBytes;   // image, zero cell, task counter
}
static bool bwd_plan_fits(const PlanGeom &g) { return g.PW <= 255; }

// ---- plan builders: the reference arithmetic, evaluated exactly, once per geometry ---------------------------
// The taps of a ray are stored from ITS OWN first live row on: group g of ray (a, j) holds canvas rows
// first[a][j] + 8g .. + 8g + 7.  The 64 rays of a bin block enter the slice at different rows (by up to 64 |tan| rows),
// and walking them from a common first row visited 134 rows per task for 85 useful ones; aligned at their own entry
// they need max-chord / 8 groups.  The order of every ray's sum is unchanged (rows ascending; skipped rows were +0.0f).
// ImageProjectiveTransformV3, NEAREST: (t0*x + t1*y) + t2, std::round, zero fill -- LDS index of the tap or -1
__device__ __forceinline__ int fwd_tap(const PlanGeom &g, const FwdLayout &L, bool plus, float xj, float yj, float t1,
                                       float t2, float t4, float t5, int i)
{
    const float fi = (float)i;
    const float x = (xj + t1 * fi) + t2;
    const float y = (yj + t4 * fi) + t5;
    const int ix = (int)__builtin_roundf(x) - g.px;
    const int iy = (int)__builtin_roundf(y) - g.py;
    if ((unsigned)ix < (unsigned)g.W && (unsigned)iy < (unsigned)g.H) return iy * L.pitch + (plus ? ix : g.W - 1 - ix);
    return -1;
}
// pass 1, one wave per (bin block, angle): first live row of every ray (PH if the ray misses the slice)
__global__ __launch_bounds__(64) void rotate_fwd_first_kernel(PlanGeom g, const float *__restrict__ T8, FwdLayout L,
                                                              char *__restrict__ plan)
{
    const int a = blockIdx.y, jb = blockIdx.x, lane = threadIdx.x;
    const int j = lane_to_bin(g.PW, jb, lane);
    const float *t = T8 + 8 * a;
    const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
    const bool plus = L.skew0 ? true : (t0 >= 0.0f) == (t3 >= 0.0f);
    int *cls = reinterpret_cast<int *>(plan + L.off_cls);
    int *first = reinterpret_cast<int *>(plan + L.off_first);
    if (jb == 0 && lane == 0) cls[a] = L.skew0 ? (a & 1) : (plus ? 1 : 0);   // skew0: one layout, angles dealt to both "classes"
    const float xj = t0 * (float)j, yj = t3 * (float)j;
    int f = g.PH;
    if ((unsigned)j < (unsigned)g.PW)
        for (int i = 0; i < g.PH; ++i)
            if (fwd_tap(g, L, plus, xj, yj, t1, t2, t4, t5, i) >= 0) {
                f = i;
                break;
            }
    if (L.skew0) {
        // start_l = d_l + c, d_l = round(alpha * (l % 32)), c = min over the half-wave's live lanes of (first_l - d_l): every
        // lane starts at or before its first live row; alpha makes the column step between consecutive lanes +-1
        int best_steps = INT_MAX, best_start = f;
        for (int sgn = 0; sgn < 2; ++sgn) {
            const float target = sgn ? -1.0f : 1.0f;
            const float alpha = fabsf(t1) < 1e-3f ? 0.0f : (target - t0) / t1;
            const int d = (int)rintf(alpha * (float)(lane & 31));
            int cmin = f < g.PH ? f - d : INT_MAX;
            for (int off = 16; off > 0; off >>= 1) cmin = min(cmin, __shfl_xor(cmin, off, 64));   // within the 32-lane half
            const int start = cmin == INT_MAX ? f : d + cmin;
            // rows this half would walk: the largest (last live row - start + 1); last live row = first + live count - 1
            int last = -1;
            if (f < g.PH)
                for (int i = g.PH - 1; i >= f; --i)
                    if (fwd_tap(g, L, plus, xj, yj, t1, t2, t4, t5, i) >= 0) {
                        last = i;
                        break;
                    }
            int steps = f < g.PH ? last - start + 1 : 0;
            for (int off = 32; off > 0; off >>= 1) steps = max(steps, __shfl_xor(steps, off, 64));
            if (steps < best_steps) {
                best_steps = steps;
                best_start = start;
            }
        }
        f = f < g.PH ? max(best_start, -(1 << 20)) : f;
    }
    first[(size_t)a * L.PWpad + jb * 64 + lane] = f;
}
// pass 2, one wave per (bin block, angle, row group): rng[a][jb] = {first live group, kRngBias - last live group}, both
// reduced with atomicMin over the groups (the buffer is preset to 0x7f7f7f7f by the host-side memset).
constexpr int kRngBias = 0x7f7f7f7f;
__global__ __launch_bounds__(64) void rotate_fwd_plan_kernel(PlanGeom g, const float *__restrict__ T8, FwdLayout L,
                                                             char *__restrict__ plan)
{
    const int a = blockIdx.y, jb = blockIdx.x, gq = blockIdx.z, lane = threadIdx.x;
    const int j = lane_to_bin(g.PW, jb, lane);
    const float *t = T8 + 8 * a;
    const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
    const bool plus = L.skew0 ? true : (t0 >= 0.0f) == (t3 >= 0.0f);
    const int *first = reinterpret_cast<const int *>(plan + L.off_first);
    int *rng = reinterpret_cast<int *>(plan + L.off_rng);
    uint4 *idx = reinterpret_cast<uint4 *>(plan + L.off_idx);
    const float xj = t0 * (float)j, yj = t3 * (float)j;
    const int i0 = first[(size_t)a * L.PWpad + jb * 64 + lane] + 8 * gq;
    unsigned e16[8];
    bool any = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int i = i0 + e;
        unsigned v = (unsigned)L.zero;
        if (i >= 0 && i < g.PH && (unsigned)j < (unsigned)g.PW) {
            const int tap = fwd_tap(g, L, plus, xj, yj, t1, t2, t4, t5, i);
            if (tap >= 0) {
                v = (unsigned)tap;
                any = true;
            }
        }
        e16[e] = v;
    }
    uint4 q;
    q.x = e16[0] | (e16[1] << 16);
    q.y = e16[2] | (e16[3] << 16);
    q.z = e16[4] | (e16[5] << 16);
    q.w = e16[6] | (e16[7] << 16);
    idx[((size_t)a * L.Galloc + gq) * L.PWpad + jb * 64 + lane] = q;
    if (__any(any) && lane == 0) {
        atomicMin(&rng[(a * L.nJB + jb) * 2 + 0], gq);
        atomicMin(&rng[(a * L.nJB + jb) * 2 + 1], kRngBias - gq);
    }
}

__global__ __launch_bounds__(64) void rotate_bwd_plan_kernel(PlanGeom g, const float *__restrict__ Tinv8, BwdLayout L,
                                                             uint4 *__restrict__ idx)
{
    const int xb = blockIdx.x, yrow = blockIdx.y, a16 = blockIdx.z, lane = threadIdx.x;
    const int xcol = xb * 64 + lane;
    const float fx = (float)(xcol + g.px), fy = (float)(yrow + g.py);
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int a = 16 * a16 + e;
        unsigned v = 255u;
        if (a < g.A && xcol < g.W) {
            // the gradient op re-samples the row-broadcast cotangent with the inverted transform (same arithmetic)
            const float *t = Tinv8 + 8 * a;
            const float x = (t[0] * fx + t[1] * fy) + t[2];
            const float y = (t[3] * fx + t[4] * fy) + t[5];
            const int ix = (int)__builtin_roundf(x), iy = (int)__builtin_roundf(y);
            if ((unsigned)ix < (unsigned)g.PW && (unsigned)iy < (unsigned)g.PH) v = (unsigned)ix;
        }
        w[e >> 2] |= v << (8 * (e & 3));
    }
    idx[((size_t)a16 * g.H + yrow) * L.Wpad + xcol] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Exact-transpose plan.  The forward adds img[tap(a, i, j)] into sino[a][j]; its transpose adds g[a][j] into every pixel
// that is the tap of a canvas sample (i, j).  A rotation followed by rounding sends at most two samples of an angle to one
// pixel (the rotated unit lattice has at most two points in a pixel's rounding cell), so the transpose is a GATHER of at
// most two bins per (angle, pixel): the plan stores them as bytes (255 = none) in the order the scatter would have added
// them (canvas row i ascending, then bin j), as "virtual angles" 2a (first hit) and 2a + 1 (second).  The hits are found
// among the 3 x 3 canvas samples around the pixel's inverse-rotated position by evaluating the FORWARD tap exactly.  A pixel
// with more than two hits (not a rotation) raises *overflow: the caller then keeps the scatter kernel.
__global__ __launch_bounds__(64) void rotate_exact_plan_kernel(PlanGeom g, const float *__restrict__ T8,
                                                               const float *__restrict__ Tinv8, BwdLayout L,
                                                               unsigned char *__restrict__ idx, int *__restrict__ overflow)
{
    const int xb = blockIdx.x, yrow = blockIdx.y, a = blockIdx.z, lane = threadIdx.x;
    const int xcol = xb * 64 + lane;
    unsigned hit0 = 255u, hit1 = 255u;
    if (xcol < g.W) {
        const float *ti = Tinv8 + 8 * a, *t = T8 + 8 * a;
        const float fx = (float)(xcol + g.px), fy = (float)(yrow + g.py);
        const int jc = (int)__builtin_roundf((ti[0] * fx + ti[1] * fy) + ti[2]);
        const int ic = (int)__builtin_roundf((ti[3] * fx + ti[4] * fy) + ti[5]);
        int n = 0;
        for (int i = ic - 1; i <= ic + 1; ++i)
            for (int j = jc - 1; j <= jc + 1; ++j) {
                if ((unsigned)i >= (unsigned)g.PH || (unsigned)j >= (unsigned)g.PW) continue;
                const float x = (t[0] * (float)j + t[1] * (float)i) + t[2];
                const float y = (t[3] * (float)j + t[4] * (float)i) + t[5];
                if ((int)__builtin_roundf(x) - g.px == xcol && (int)__builtin_roundf(y) - g.py == yrow) {
                    if (n == 0) hit0 = (unsigned)j;
                    if (n == 1) hit1 = (unsigned)j;
                    ++n;
                }
            }
        if (n > 2) atomicOr(overflow, 1);
    }
    // byte (2a) & 15 and (2a + 1) & 15 of the uint4 of virtual-angle group (2a) >> 4
    unsigned char *cell = idx + (((size_t)((2 * a) >> 4) * g.H + yrow) * L.Wpad + xcol) * 16 + ((2 * a) & 15);
    cell[0] = (unsigned char)hit0;
    cell[1] = (unsigned char)hit1;
}

// ---- executing a plan -------------------------------------------------------------------------------------------
// LDS byte addresses of the two u16 dword indices packed in `pk`: one SDWA op each (select a 16-bit half, shift by 2)
__device__ __forceinline__ void unpack2(unsigned pk, int &lo4, int &hi4)
{
    asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(lo4)
        : "v"(pk));
    asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=v"(hi4)
        : "v"(pk));
}
// The gathers below address LDS by ABSOLUTE byte addresses: these kernels have no static LDS, so their dynamic LDS starts at
// address 0 and an unpacked tap is the ds_read's address as it stands.  Through the `lds` pointer every tap paid a v_add_u32 of
// the array's link-time base -- zero -- that the compiler cannot fold: one of the three vector operations per tap of a slice
// pair.  That the kernel has no static LDS is checked on the HOST, once per kernel and device, before its first launch
// (CTPVAE_REQUIRE_NO_STATIC_LDS, common.h; round 3 trapped on the device).
__device__ __forceinline__ float lds_at(const float *, int byte_off) { return *(lds_cptr)(size_t)(unsigned)byte_off; }
__device__ __forceinline__ void unpack2x8(unsigned pk, int &lo8, int &hi8)
{
    asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(lo8)
        : "v"(pk));
    asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=v"(hi8)
        : "v"(pk));
}
__device__ __forceinline__ void gather8(const float *lds, const uint4 q, f32x2 (&v)[8])
{
    int a[8];
    unpack2x8(q.x, a[0], a[1]);
    unpack2x8(q.y, a[2], a[3]);
    unpack2x8(q.z, a[4], a[5]);
    unpack2x8(q.w, a[6], a[7]);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = *(const __attribute__((address_space(3))) f32x2 *)(size_t)(unsigned)a[e];
}
__device__ __forceinline__ void gather8(const float *lds, const uint4 q, float (&v)[8])
{
    int a0, a1, a2, a3, a4, a5, a6, a7;
    unpack2(q.x, a0, a1);
    unpack2(q.y, a2, a3);
    unpack2(q.z, a4, a5);
    unpack2(q.w, a6, a7);
    v[0] = lds_at(lds, a0);
    v[1] = lds_at(lds, a1);
    v[2] = lds_at(lds, a2);
    v[3] = lds_at(lds, a3);
    v[4] = lds_at(lds, a4);
    v[5] = lds_at(lds, a5);
    v[6] = lds_at(lds, a6);
    v[7] = lds_at(lds, a7);
}

// x / d for the small operands of the launch's index arithmetic as ONE s_mul_hi_u32: magic = 2^32 / d + 1 is exact while
// x * d < 2^32 (the callers check).  A 32-bit division costs a wave ~30 instructions, a workgroup's prologue had three of them in
// front of its first loads, and that prologue is bound by the instructions its 16 waves issue.
// (d = 1 has no 32-bit word -- and needs none: word 0 = "x itself"; d = 0 is never divided by)
__host__ __device__ inline unsigned div_magic(unsigned d) { return d > 1 ? (unsigned)((1ull << 32) / d) + 1u : 0u; }
__host__ __device__ inline unsigned div_by_magic(unsigned x, unsigned magic) { return magic ? (unsigned)(((unsigned long long)x * magic) >> 32) : x; }

// clist[c] = (count, the angles of class c in ascending order): the planned kernels' task lists; behind the two lists the
// division words of the two counts (div_magic)
__global__ __launch_bounds__(64) void rotate_class_list_kernel(int A, FwdLayout L, char *__restrict__ plan)
{
    const int *cls = reinterpret_cast<const int *>(plan + L.off_cls);
    int *clist = reinterpret_cast<int *>(plan + L.off_clist);
    const int lane = threadIdx.x;
    for (int c = 0; c < 2; ++c) {
        int *list = clist + c * (A + 1);
        int n = 0;
        for (int a0 = 0; a0 < A; a0 += 64) {
            const bool in = a0 + lane < A && cls[a0 + lane] == c;
            const unsigned long long m = __ballot(in);
            if (in) list[1 + n + __popcll(m & ((1ull << lane) - 1ull))] = a0 + lane;
            n += __popcll(m);
        }
        if (lane == 0) {
            list[0] = n;
            reinterpret_cast<unsigned *>(clist + 2 * (A + 1))[c] = div_magic((unsigned)n);
        }
    }
}

// Forward.  Workgroup = (slice s, class c, group gi): stages the slice once (LDS-DMA, mirrored for class 0), then each
// wave takes the (angle, bin block) tasks of its class round-robin.  A task streams its index groups four loads deep,
// gathers the previous group's eight taps while the next indices are in flight, and adds in row order.
// NS = 2: two slices per workgroup, interleaved as float2 in LDS -- ONE index stream and ONE ds_read_b64 per tap serve
// both slices (the index stream is what binds the kernel, section 6 of DESIGN.md).  Used when a launch has enough
// tasks to keep every CU busy with half as many workgroups.
// EPI: the log-likelihood epilogue -- besides the ray-sum, every store also writes the log-probability
// of the measured sample under it (mask [S][A], measured [S][A][PW]), the expression of loglik.hip, so that
// calculate_log_prob_M_given_R costs one launch instead of two and the sinogram is not read back.
// SEL: the launch projects a SUBSET of the plan's angles -- sel[k], k < n_sel <= 64 * kSelRounds, are plan angle numbers
// (the training loop's per-step `api` random angles, ctvae/helper_functions.py:350-357, on ONE dense plan built from
// the host's theta); output row k of every sinogram is plan angle sel[k].  Each wave derives its class's task list
// from sel and the plan's cls[] with ballots (no list in memory, no extra launch): lane l of round r holds entry
// 64 r + l, its plan angle and its rank among the entries of class c; the ai-th angle of the class is then one ballot
// + readlane away.
constexpr int kSelRounds = 4;
template <int NS, bool EPI, bool SEL>
__global__ __launch_bounds__(1024) void rotate_fwd_planned_kernel(const float *__restrict__ img, PlanGeom g, FwdLayout L,
                                                                  const char *__restrict__ plan, int wgs_per_slice,
                                                                  int g_S, float *__restrict__ sino, LogLikEpilogue epi,
                                                                  const int *__restrict__ sel, int n_sel, int affine,
                                                                  int units1, int wgs2, unsigned inv_wgs1, unsigned inv_wgs2, int small_div,
                                                                  int units2, int wgs3, unsigned inv_wgs3)
{
    typedef typename SliceVec<NS>::type vec_t;
    extern __shared__ float lds[];
    // Workgroups b and b + 8 share an XCD (round-robin dispatch; speed only).  Slices (slice pairs) are dealt to the 8
    // XCDs so that all workgroups of one slice read it through the same L2: block = (u / 8) * 8 * wgs + wg * 8 + u % 8.
    // `affine` (round 4; wgs_per_slice a multiple of 8): the ANGLES are dealt to the XCDs instead -- block = u * wgs + wg runs
    // on XCD wg % 8 and takes the angles ai = gi (mod G) of its class, so an XCD's L2 sees one eighth of the plan, launch after
    // launch (17 MB of u16 taps at 180 angles: streamed into all eight L2s otherwise), and every unit is staged through all of
    // them (3.3 MB x 8 at B = 50) -- the smaller stream by far at many angles.
    const int units = (g_S + NS - 1) / NS;
    // A launch is a list of PIECES (unit, class, task group), one workgroup each, dispatched in order: units1 * wgs_per_slice
    // pieces of the first units1 units, then wgs2 pieces of every later unit (round 5: a coarse cut for whole rounds of
    // workgroups, a finer one behind it so that the last, partial round spreads over all CUs; units1 == units otherwise).
    // Inside either part pieces are numbered in octets of units, as above.
    // (... and, from unit units2 on, wgs3 pieces per unit: a third, finer part where it fills the last round better)
    auto piece_of = [&](int q, int &u_, int &wg_, int &wgs_) {
        int u0 = 0, nu = units1;
        unsigned inv = inv_wgs1;
        wgs_ = wgs_per_slice;
        const int Q1 = units1 * wgs_per_slice, Q2 = (units2 - units1) * wgs2;
        if (q >= Q1 + Q2) {
            q -= Q1 + Q2;
            u0 = units2;
            nu = units - units2;
            wgs_ = wgs3;
            inv = inv_wgs3;
        } else if (q >= Q1) {
            q -= Q1;
            u0 = units1;
            nu = units2 - units1;
            wgs_ = wgs2;
            inv = inv_wgs2;
        }
        // q / (8 wgs) = (q / 8) / wgs, by multiplication (the host checked the range: small_div)
        const int octet = small_div ? (int)div_by_magic((unsigned)q >> 3, inv) : q / (8 * wgs_), rem = q - octet * 8 * wgs_;
        if ((octet + 1) * 8 <= nu) {
            wg_ = rem >> 3;
            u_ = u0 + octet * 8 + (rem & 7);
        } else {   // the last, partial octet is laid out unit-major
            const int ru = small_div ? (int)div_by_magic((unsigned)rem, inv) : rem / wgs_;
            u_ = u0 + octet * 8 + ru;
            wg_ = rem - ru * wgs_;
        }
    };
    int u, wg, wgs = wgs_per_slice;
    [[maybe_unused]] const int q = blockIdx.x;
    if (affine) {
        u = blockIdx.x / wgs_per_slice;
        wg = blockIdx.x - u * wgs_per_slice;
    } else {
        piece_of(q, u, wg, wgs);
    }
    const int s = u * NS;
    const bool has2 = NS == 2 && s + 1 < g_S;     // an odd batch ends with a half-empty pair (slice s staged twice)
    const int c = wg & 1, gi = wg >> 1, G = wgs >> 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const float *im = img + (size_t)s * g.H * g.W;
    CTPVAE_PSTAMP(0);

    // Tasks of this workgroup: (angle of its class, bin block), m = jb * ncls + ai counted from the central bin blocks
    // (the longest rays) outwards; group gi of G takes m = gi, gi + G, ...  A wave's FIRST task is fixed (its wave
    // number), so its range and first four index vectors are requested here, before the slice is even staged; later
    // tasks come from an LDS counter, each one prepared (scalar loads of the range, index loads) while the previous
    // one is being gathered.
    const int *clist = reinterpret_cast<const int *>(plan + L.off_clist) + c * (g.A + 1);
    const int *rng = reinterpret_cast<const int *>(plan + L.off_rng);
    const uint4 *idx = reinterpret_cast<const uint4 *>(plan + L.off_idx);
    const int A_out = SEL ? n_sel : g.A;   // rows of an output sinogram
    int sel_a[kSelRounds], sel_rank[kSelRounds], sel_cum[kSelRounds + 1];
    int ncls;
    if constexpr (SEL) {
        const int *cls = reinterpret_cast<const int *>(plan + L.off_cls);
        sel_cum[0] = 0;
#pragma unroll
        for (int r = 0; r < kSelRounds; ++r) {
            const int k = 64 * r + lane;
            int a = 0, cl = -1;
            if (k < n_sel) {
                a = min(max(sel[k], 0), g.A - 1);   // a bad index cannot leave the plan
                cl = cls[a];
            }
            const unsigned long long m = __ballot(cl == c);
            sel_a[r] = a;
            sel_rank[r] = cl == c ? (int)__popcll(m & ((1ull << lane) - 1ull)) : -1;
            sel_cum[r + 1] = sel_cum[r] + (int)__popcll(m);
        }
        ncls = sel_cum[kSelRounds];
    } else {
        ncls = clist[0];
    }
    [[maybe_unused]] const unsigned inv_ncls = reinterpret_cast<const unsigned *>(plan + L.off_clist + 2ll * (g.A + 1) * 4)[c];
    // a workgroup's w-th task: round-robin over the class's (bin block, angle) list, or -- affine -- its own angles' blocks
    const int n_gi = affine ? max(0, (ncls - gi + G - 1) / G) : 0;
    const int ntask = affine ? n_gi * L.nJB : ncls * L.nJB;
#ifdef CTPVAE_TUNE_NOIDX
    const size_t st = 0;   // timing only: every group re-reads the first index vector (no index streaming)
#else
    const size_t st = (size_t)L.PWpad;
#endif
    struct Task {
        bool valid;
        int a, k, j, ng;   // plan angle, output row
        const uint4 *p;
        uint4 q0, q1, q2, q3;
        float em[NS], ex[NS];   // EPI: mask entry and measured sample of the task's outputs, requested with the task
    };
    [[maybe_unused]] float epnm = 0.0f, einv = 0.0f;
    if constexpr (EPI) epnm = *epi.pnm, einv = 1.0f / epnm;   // the derivative multiplies by the reciprocal (loglik_math.h)
    auto prepare = [&](int w) -> Task {   // w: the workgroup's own task number
        Task t;
        const int m = affine ? w : w * G + gi;
        t.valid = m < ntask;
        t.a = t.k = t.j = t.ng = 0;
        t.p = idx;
        t.q0 = t.q1 = t.q2 = t.q3 = uint4{0, 0, 0, 0};
#pragma unroll
        for (int n = 0; n < NS; ++n) t.em[n] = t.ex[n] = 0.0f;
        if (t.valid) {   // wave-uniform
            int jb, ai;
            if (affine) {
                jb = m / n_gi;
                ai = gi + G * (m - jb * n_gi);
            } else {
                jb = !SEL && small_div ? (int)div_by_magic((unsigned)m, inv_ncls) : m / ncls;
                ai = m - jb * ncls;
            }
            if constexpr (SEL) {
#pragma unroll
                for (int r = 0; r < kSelRounds; ++r)
                    if (ai >= sel_cum[r] && ai < sel_cum[r + 1]) {   // wave-uniform
                        const unsigned long long hit = __ballot(sel_rank[r] == ai - sel_cum[r]);
                        const int l = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)hit) - 1);
                        t.a = __builtin_amdgcn_readlane(sel_a[r], l);
                        t.k = 64 * r + l;
                    }
            } else {
                t.k = t.a = clist[1 + ai];
            }
            const int first = rng[(t.a * L.nJB + jb) * 2], last = kRngBias - rng[(t.a * L.nJB + jb) * 2 + 1];
            const int g0 = last >= first ? first : 0;
            t.ng = last >= first ? last - first + 1 : 0;   // row groups any of this block's rays needs
            t.j = lane_to_bin(g.PW, jb, lane);
            t.p = idx + ((size_t)t.a * L.Galloc + g0) * L.PWpad + jb * 64 + lane;
            // index vectors of groups 0..3 (loads run up to four groups past ng: the table keeps 8 dead groups)
            t.q0 = t.p[0];
            t.q1 = t.p[st];
            t.q2 = t.p[2 * st];
            t.q3 = t.p[3 * st];
            t.p += 4 * st;
            if constexpr (EPI) {   // (loaded in the epilogue these cost the wave a round trip behind its gathers)
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
    Task cur = prepare(wave);
    int *next_task = reinterpret_cast<int *>(lds + (L.zero + 1) * NS);
    if (threadIdx.x == 0) *next_task = nwaves;

    // Stage the slice(s): 16-byte loads, conflict-free ds_write_b32 (see stage_rows_v4).
    {   // (a pair in one load round trip, written as float2; units of 64 / 128 / 256 columns in the lean form: stage_unit)
        const float *srcs[NS];
        srcs[0] = im;
        if constexpr (NS == 2) srcs[1] = im + (has2 ? (size_t)g.H * g.W : 0);
        stage_unit<NS>(lds, srcs, g.H, g.W, g.W, L.pitch, c == 0 && !L.skew0, lane, wave, nwaves);
    }
    if (threadIdx.x < NS) lds[L.zero * NS + threadIdx.x] = 0.0f;
    CTPVAE_PSTAMP(1);
    __syncthreads();
    CTPVAE_PSTAMP(2);

    while (cur.valid) {
        int m = 0;
        if (lane == 0) m = atomicAdd(next_task, 1);
        const Task nxt = prepare(__builtin_amdgcn_readfirstlane(m));

        const int ng = __builtin_amdgcn_readfirstlane(cur.ng);   // wave-uniform: the loop's exits are scalar branches
        const uint4 *p = cur.p;
        vec_t acc = 0.0f;
        if (ng > 0) {
            // q0..q3: index vectors of groups n..n+3 (loads in flight); va/vb: gathers of group n / n+1 in flight.
            uint4 q0 = cur.q0, q1 = cur.q1, q2 = cur.q2, q3 = cur.q3;
            vec_t va[8], vb[8];
            gather8(lds, q0, va);
            q0 = p[0];
            if constexpr (EPI) {   // (the epilogue variants have no registers for the form below: they keep round 3's loop)
                for (int n = 0;; n += 4) {
                    gather8(lds, q1, vb);
                    q1 = p[st];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc += va[e];          // group n
                    if (n + 1 >= ng) break;
                    gather8(lds, q2, va);
                    q2 = p[2 * st];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc += vb[e];          // group n + 1
                    if (n + 2 >= ng) break;
                    gather8(lds, q3, vb);
                    q3 = p[3 * st];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc += va[e];          // group n + 2
                    if (n + 3 >= ng) break;
                    gather8(lds, q0, va);
                    p += 4 * st;
                    q0 = p[0];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc += vb[e];          // group n + 3
                    if (n + 4 >= ng) break;
                }
            } else {
            // (round 4: the task's last group issues nothing behind it -- the gathers of a group past the last were ~7 % of the
            // launch's LDS instructions)
#define CTPVAE_PSTEP(LAST, QN, VN, VC, LOADNEXT)                                                   \
                if (LAST >= ng) {   /* VC is the last group: added behind the loop (see cplan_walk.h cwalk) */ \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) vt[e] = VC[e];                    \
                    break;                                                                         \
                }                                                                                  \
                gather8(lds, QN, VN);                                                              \
                LOADNEXT;                                                                          \
                __builtin_amdgcn_sched_barrier(0);                                                 \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) acc += VC[e];
            vec_t vt[8];
            for (int n = 0;; n += 4) {
                CTPVAE_PSTEP(n + 1, q1, vb, va, q1 = p[st])                 // adds group n
                CTPVAE_PSTEP(n + 2, q2, va, vb, q2 = p[2 * st])             // group n + 1
                CTPVAE_PSTEP(n + 3, q3, vb, va, q3 = p[3 * st])             // group n + 2
                CTPVAE_PSTEP(n + 4, q0, va, vb, (p += 4 * st, q0 = p[0]))   // group n + 3
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += vt[e];
#undef CTPVAE_PSTEP
            }
        }
        if ((unsigned)cur.j < (unsigned)g.PW) {
            auto store = [&](int n, float v) {
                const size_t o = ((size_t)(s + n) * A_out + cur.k) * g.PW + cur.j;
                sino[o] = v;
                if constexpr (EPI) epi.write_loaded(o, cur.em[n], cur.ex[n], epnm, einv, v);
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
    CTPVAE_PSTAMP(3);
}

// four byte taps of one dword -> four LDS byte offsets inside a cotangent row (SDWA: select a byte, shift by
// log2(bytes per cell): 2, or 3 when two slices are interleaved as float2)
template <int SHIFT>
__device__ __forceinline__ void unpack4(unsigned pk, int &b0, int &b1, int &b2, int &b3)
{
    asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(b0) : "v"(pk), "i"(SHIFT));
    asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(b1) : "v"(pk), "i"(SHIFT));
    asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(b2) : "v"(pk), "i"(SHIFT));
    asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(b3) : "v"(pk), "i"(SHIFT));
}
template <int NS> __device__ __forceinline__ typename SliceVec<NS>::type lds_at_vec(const float *, int byte_off)
{
    return *(const __attribute__((address_space(3))) typename SliceVec<NS>::type *)(size_t)(unsigned)byte_off;   // see lds_at
}
// the 16 taps of `q` are staged rows AL0 .. AL0+15: the row offset is a compile-time ds_read immediate, the address
// VGPR is just the SDWA-extracted bin * cell size -- no address arithmetic per tap
template <int AL0, int NS, int DUP = 1>
__device__ __forceinline__ void gather16(const float *lds, const uint4 q, int n_live, typename SliceVec<NS>::type (&v)[16])
{
    // DUP = 2 (exact-transpose plan): taps AL0 + e are "virtual angles"; virtual angle v reads staged row v / 2
    // n_live (wave-uniform): staged rows AL0 .. AL0+n_live-1 exist; a partial last group skips whole dwords of taps
    constexpr int ROW = kBwdPitch * 4 * NS;   // bytes per staged row
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (4 * d < n_live) {
            int b0, b1, b2, b3;
            unpack4<NS == 1 ? 2 : 3>(w[d], b0, b1, b2, b3);
            v[4 * d + 0] = lds_at_vec<NS>(lds, b0 + ((AL0 + 4 * d + 0) / DUP) * ROW);
            v[4 * d + 1] = lds_at_vec<NS>(lds, b1 + ((AL0 + 4 * d + 1) / DUP) * ROW);
            v[4 * d + 2] = lds_at_vec<NS>(lds, b2 + ((AL0 + 4 * d + 2) / DUP) * ROW);
            v[4 * d + 3] = lds_at_vec<NS>(lds, b3 + ((AL0 + 4 * d + 3) / DUP) * ROW);
        } else {
            v[4 * d + 0] = v[4 * d + 1] = v[4 * d + 2] = v[4 * d + 3] = 0.0f;
        }
    }
}

// eight taps (two index dwords w0, w1) of staged rows AL0 .. AL0+7; n_live counted from AL0 (a partial group skips whole dwords)
template <int AL0, int NS, int DUP = 1>
__device__ __forceinline__ void gather8(const float *lds, unsigned w0, unsigned w1, int n_live, typename SliceVec<NS>::type (&v)[8])
{
    constexpr int ROW = kBwdPitch * 4 * NS;
    const unsigned w[2] = {w0, w1};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        if (4 * d < n_live) {
            int b0, b1, b2, b3;
            unpack4<NS == 1 ? 2 : 3>(w[d], b0, b1, b2, b3);
            v[4 * d + 0] = lds_at_vec<NS>(lds, b0 + ((AL0 + 4 * d + 0) / DUP) * ROW);
            v[4 * d + 1] = lds_at_vec<NS>(lds, b1 + ((AL0 + 4 * d + 1) / DUP) * ROW);
            v[4 * d + 2] = lds_at_vec<NS>(lds, b2 + ((AL0 + 4 * d + 2) / DUP) * ROW);
            v[4 * d + 3] = lds_at_vec<NS>(lds, b3 + ((AL0 + 4 * d + 3) / DUP) * ROW);
        } else {
            v[4 * d + 0] = v[4 * d + 1] = v[4 * d + 2] = v[4 * d + 3] = 0.0f;
        }
    }
}

// Backward (TensorFlow-compatible).  Workgroup = (slice s, 64-column x (waves x PPT)-row tile): stages a chunk of the
// slice's cotangent rows (257-cell rows, zeros behind the bins), then every lane owns one column and PPT rows; for
// each group of sixteen angles it loads the PPT index vectors (16 B = 16 taps), gathers and adds in angle order.
// NS = 2: two slices per workgroup, their cotangent rows fetched together and interleaved as float2 -- one index
// stream, one SDWA unpack and one ds_read_b64 per tap serve both (rows are then 2056 B, so a chunk is 32 angles to keep
// the row offsets immediates).
// DUP = 2: the exact-transpose plan (two taps per angle and pixel, as virtual angles 2a, 2a + 1 reading staged row a): the same
// kernel with twice the index vectors per staged row; sums are added in virtual-angle order = the scatter's order.
// SHORT (at most 32 angles, DUP = 1: one staged chunk, two groups of sixteen): the second group's index vectors are requested at
// the start with the first's -- loaded after the first group's gathers they put an L2 round trip behind the barrier of a launch
// that is a few microseconds long -- and the chunk pipeline's registers make room for them.
template <int PPT, int MAXT, int NS, int DUP = 1, bool SHORT = false>
__global__ __launch_bounds__(MAXT) void rotate_bwd_planned_kernel(const float *__restrict__ gsino, PlanGeom g, BwdLayout L,
                                                                 const uint4 *__restrict__ idx, int tiles_y, int g_S,
                                                                 SliceScale scale, float *__restrict__ gimg, unsigned inv_tiles,
                                                                 unsigned inv_nxb)
{
    typedef typename SliceVec<NS>::type vec_t;
    constexpr int kChunk = kBwdChunk / NS;
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int tiles = L.nXB * tiles_y;
    const int units = (g_S + NS - 1) / NS;
    // Workgroups b and b + 8 share an XCD (round-robin dispatch; speed only): the tiles of one slice (pair) are placed
    // on one XCD so that its cotangent rows are fetched into one L2 -- block = (u / 8) * 8 * tiles + tile * 8 + u % 8.
    int u, tile;
    {
        // (divisions by multiplication, div_magic: inv_tiles == 0 -- the host's "operands too large" -- divides)
        const int per8 = 8 * tiles;
        const int octet = inv_tiles ? (int)div_by_magic(blockIdx.x >> 3, inv_tiles) : (int)blockIdx.x / per8, rem = blockIdx.x - octet * per8;
        if ((octet + 1) * 8 <= units) {
            tile = rem >> 3;
            u = octet * 8 + (rem & 7);
        } else {   // the last, partial octet is laid out unit-major
            const int ru = inv_tiles ? (int)div_by_magic((unsigned)rem, inv_tiles) : rem / tiles;
            u = octet * 8 + ru;
            tile = rem - ru * tiles;
        }
    }
    const int s = u * NS;
    const bool has2 = NS == 2 && s + 1 < g_S;   // an odd batch ends with a half-empty pair (slice s staged twice)
    CTPVAE_PSTAMP(0);
    const float k0 = scale.at(s), k1 = has2 ? scale.at(s + 1) : 1.0f;
    const int ty = inv_tiles ? (int)div_by_magic((unsigned)tile, inv_nxb) : tile / L.nXB, xb = tile - ty * L.nXB;
    const float *gs = gsino + (size_t)s * g.A * g.PW;
    const int xcol = xb * 64 + lane;
    const int y0 = ty * (nwaves * PPT) + wave;   // this wave's rows: y0, y0 + nwaves, ...
    vec_t acc[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) acc[k] = 0.0f;
    const uint4 *p = idx + (size_t)xcol;
    // Index vectors of TWO groups of sixteen angles in flight (round 4; SHORT launches hold both of theirs anyway): with one,
    // every wave of a many-angle launch waited out an L2 round trip per group -- SQ_WAIT_ANY was 46 % of the wave cycles at
    // B = 50 x 180 angles with the LDS array 22 % busy -- and all sixteen waves of the CU's one workgroup did so together.
    uint4 q[2][PPT];
    auto load_group = [&](int a16, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int y = min(y0 + k * nwaves, g.H - 1);   // rows past the slice re-read the last row, never stored
#ifdef CTPVAE_TUNE_BWD_NOIDX
            if (a16 > 0) continue;   // timing only: every group reuses the first index vectors (no index streaming)
#endif
            q[SLOT][k] = p[((size_t)a16 * g.H + y) * L.Wpad];
        }
    };
    load_group(0, std::integral_constant<int, 0>{});       // index loads fly while the cotangent rows land
    if constexpr (!SHORT) load_group(min(1, L.NA16 - 1), std::integral_constant<int, 1>{});
    [[maybe_unused]] uint4 q2nd[PPT];
    if constexpr (SHORT) {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            q2nd[k] = p[((size_t)min(1, L.NA16 - 1) * g.H + min(y0 + k * nwaves, g.H - 1)) * L.Wpad];   // (its second group)
    }

    const int VA = g.A * DUP;                            // virtual angles (= angles unless DUP = 2)
    const int chunk = min(L.NA16 * 16, kChunk * DUP);    // virtual angles per staged chunk (a multiple of 16)
    // Pairs in 16-wave workgroups (many angles, several chunks): the cotangent rows of chunk c + 1 are requested into
    // registers (two units per lane) before the gathers of chunk c and written to LDS after them -- each chunk's load
    // round trip hides behind the previous chunk's gather phase instead of standing between two barriers.
    constexpr bool kPipe = NS == 2 && !SHORT;       // (the single-slice form has no registers to spare: 143 VGPRs with it)
    constexpr int kAheadUnits = 2;                  // 32 rows of a pair over 16 waves
    StagedRows<NS, kPipe ? kAheadUnits : 1> ahead;
    bool ahead_valid = false;
    auto chunk_srcs = [&](int ac, const float *(&srcs)[NS]) {
        srcs[0] = gs + (size_t)ac * g.PW;
        if constexpr (NS == 2) srcs[1] = gs + (has2 ? (size_t)g.A * g.PW : 0) + (size_t)ac * g.PW;
    };
    for (int acv = 0; acv < VA; acv += chunk) {
        const int nav = min(chunk, VA - acv);
        const int na4 = (nav + 3) & ~3;             // taps are consumed a dword (4 virtual angles) at a time
        const int ac = acv / DUP;                   // first staged (real) angle of the chunk
        const int na = (nav + DUP - 1) / DUP;       // staged rows
        const int na_z = (na4 + DUP - 1) / DUP;     // rows a tap of this chunk may name
        if (acv > 0) __syncthreads();
        // a dead tap is byte 255: only cell 255 of every row (never a bin: PW <= 255) must hold 0.0f
        for (int t = threadIdx.x; t < na_z * NS; t += blockDim.x) lds[((t / NS) * kBwdPitch + 255) * NS + (t % NS)] = 0.0f;
        if (ahead_valid) {
            ahead.commit(lds, kBwdPitch);           // requested during the previous chunk's gathers
        } else if constexpr (NS == 1) {
            stage_rows(lds, gs + (size_t)ac * g.PW, na, g.PW, g.PW, kBwdPitch, false, lane, wave, nwaves);
        } else {
            const float *srcs[NS];
            chunk_srcs(ac, srcs);
            stage_rows_interleaved<2>(lds, srcs, na, g.PW, g.PW, kBwdPitch, false, lane, wave, nwaves);
        }
        ahead_valid = false;
        if (kPipe && acv + chunk < VA) {            // wave-uniform
            const float *srcs[NS];
            chunk_srcs(ac + chunk / DUP, srcs);
            const int nna = (min(chunk, VA - (acv + chunk)) + DUP - 1) / DUP;
            if (StagedRows<NS, kPipe ? kAheadUnits : 1>::fits(srcs, nna, g.PW, g.PW, nwaves)) {
                ahead.issue(srcs, nna, g.PW, g.PW, lane, wave, nwaves);
                ahead_valid = true;
            }
        }
        if (acv == 0) CTPVAE_PSTAMP(1);
        __syncthreads();
        if (acv == 0) CTPVAE_PSTAMP(2);
        // up to four (eight with DUP = 2) groups of sixteen virtual angles, unrolled so that every row offset is an immediate
        auto group = [&](auto al_tag) {
            constexpr int AL = decltype(al_tag)::value;
            if constexpr (AL < kChunk * DUP) {
                if (AL >= na4) return;                       // wave-uniform
                const int n_live = min(16, na4 - AL);
                if constexpr (SHORT) {
                    vec_t v[PPT][16];
#pragma unroll
#ifdef CTPVAE_TUNE_BWD_NOLDS
                    for (int k = 0; k < PPT; ++k)   // timing only: no gathers, the index words stand in for the taps
                        for (int e = 0; e < 16; ++e) v[k][e] = __uint_as_float((&q[0][k].x)[e & 3] & 0x3fffffu);
#else
                    for (int k = 0; k < PPT; ++k) gather16<AL, NS, DUP>(lds, q[0][k], n_live, v[k]);
#endif
#pragma unroll
                    for (int k = 0; k < PPT; ++k) q[0][k] = q2nd[k];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < PPT; ++k)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[k] += v[k][e];   // skipped taps hold +0.0f
                } else {
                    // the chunks hold an even number of groups, so a group's slot in the two-deep index queue is a constant;
                    // taps in two halves of eight (a wave holds at most 15 LDS operations in flight anyway): the registers of
                    // the other half are what pays for the second index group
                    constexpr int SLOT = (AL / 16) & 1;
                    constexpr int ROW = kBwdPitch * 4 * NS;   // bytes per staged row
                    // one index dword = four angles at a time for all PPT rows: PPT x 4 gathers in flight, added in angle order
                    auto quarter = [&](auto d_tag) {
                        constexpr int D = decltype(d_tag)::value;
                        if (4 * D >= n_live) return;                 // wave-uniform: a partial last group skips whole dwords
                        vec_t v[PPT][4];
#pragma unroll
                        for (int k = 0; k < PPT; ++k) {
                            const unsigned w = D == 0 ? q[SLOT][k].x : D == 1 ? q[SLOT][k].y : D == 2 ? q[SLOT][k].z : q[SLOT][k].w;
#ifdef CTPVAE_TUNE_BWD_NOLDS
                            for (int e = 0; e < 4; ++e) v[k][e] = __uint_as_float(w & 0x3fffffu);
#else
                            int b0, b1, b2, b3;
                            unpack4<NS == 1 ? 2 : 3>(w, b0, b1, b2, b3);
                            v[k][0] = lds_at_vec<NS>(lds, b0 + ((AL + 4 * D + 0) / DUP) * ROW);
                            v[k][1] = lds_at_vec<NS>(lds, b1 + ((AL + 4 * D + 1) / DUP) * ROW);
                            v[k][2] = lds_at_vec<NS>(lds, b2 + ((AL + 4 * D + 2) / DUP) * ROW);
                            v[k][3] = lds_at_vec<NS>(lds, b3 + ((AL + 4 * D + 3) / DUP) * ROW);
#endif
                        }
                        if constexpr (D == 3) {   // this group's index vectors are consumed: request the group after next
                            const int next = (acv + AL) / 16 + 2;
                            if (next < L.NA16) load_group(next, std::integral_constant<int, SLOT>{});
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < PPT; ++k)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[k] += v[k][e];
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    quarter(std::integral_constant<int, 0>{});
                    quarter(std::integral_constant<int, 1>{});
                    quarter(std::integral_constant<int, 2>{});
                    quarter(std::integral_constant<int, 3>{});
                    // (a partial group is the launch's last: nothing is requested behind it)
                }
            }
        };
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 16>{});
        group(std::integral_constant<int, 32>{});
        group(std::integral_constant<int, 48>{});
        group(std::integral_constant<int, 64>{});
        group(std::integral_constant<int, 80>{});
        group(std::integral_constant<int, 96>{});
        group(std::integral_constant<int, 112>{});
    }
    if (xcol < g.W) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int y = y0 + k * nwaves;
            if (y < g.H) {
                if constexpr (NS == 1) {
                    gimg[((size_t)s * g.H + y) * g.W + xcol] = k0 * acc[k];
                } else {
                    gimg[((size_t)s * g.H + y) * g.W + xcol] = k0 * acc[k].x;
                    if (has2) gimg[((size_t)(s + 1) * g.H + y) * g.W + xcol] = k1 * acc[k].y;
                }
            }
        }
    }
    CTPVAE_PSTAMP(3);
}

// ---- angle-selecting planned backward ("bwd4" plan) ---------------------------------------------------------------------
// The 16-angles-per-vector plan above cannot project a SUBSET of its angles (the training loop's per-step `api` random
// angles, ctvae/helper_functions.py:350-357): a selected angle would drag its fifteen neighbours' taps along.  The bwd4 plan
// stores the same byte taps as idx4[a][y / 4][x] = one DWORD = the bins angle a reads for pixels (4q .. 4q + 3, x): a lane
// that owns four consecutive rows of one column loads exactly one dword per selected angle.
__global__ __launch_bounds__(64) void rotate_bwd4_plan_kernel(PlanGeom g, const float *__restrict__ Tinv8, int Wpad, int HQ,
                                                              unsigned *__restrict__ idx4)
{
    const int xb = blockIdx.x, yq = blockIdx.y, a = blockIdx.z, lane = threadIdx.x;
    const int xcol = xb * 64 + lane;
    const float *t = Tinv8 + 8 * a;
    const float fx = (float)(xcol + g.px);
    unsigned w = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int yrow = 4 * yq + e;
        unsigned v = 255u;
        if (yrow < g.H && xcol < g.W) {   // the expressions of rotate_bwd_plan_kernel
            const float fy = (float)(yrow + g.py);
            const float x = (t[0] * fx + t[1] * fy) + t[2];
            const float y = (t[3] * fx + t[4] * fy) + t[5];
            const int ix = (int)__builtin_roundf(x), iy = (int)__builtin_roundf(y);
            if ((unsigned)ix < (unsigned)g.PW && (unsigned)iy < (unsigned)g.PH) v = (unsigned)ix;
        }
        w |= v << (8 * e);
    }
    idx4[((size_t)a * HQ + yq) * Wpad + xcol] = w;
}

// Workgroup = (slice or slice pair, 64 columns x 4 * waves rows); lane = column, a wave owns four consecutive rows.  The
// subset's n cotangent rows are staged in chunks (257-cell rows as in rotate_bwd_planned_kernel: a dead tap, byte 255,
// reads 0.0f); row k of the cotangent belongs to plan angle sel[k] and the sum runs over k ascending -- the order of
// ctpvae_rotate_bwd_sel_scaled_f32 and of the oracle on the gathered table.
// HOSTSEL: the subset arrived in host memory and travels in the kernel arguments (selh) instead of device memory (sel).
template <int NS, bool HOSTSEL, int MAXT = 256>
__global__ __launch_bounds__(MAXT) void rotate_bwd_planned_sel_kernel(const float *__restrict__ gsino, PlanGeom g, int Wpad,
                                                                     int HQ, const unsigned *__restrict__ idx4,
                                                                     const int *__restrict__ sel, int n_sel, int tiles_y,
                                                                     int g_S, SliceScale scale, float *__restrict__ gimg,
                                                                     SelHost selh)
{
    typedef typename SliceVec<NS>::type vec_t;
    constexpr int kChunk = kBwdChunk / NS;          // staged rows per pass: row offsets stay ds_read immediates
    constexpr int ROW = kBwdPitch * 4 * NS;         // bytes per staged row
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int nXB = Wpad >> 6, tiles = nXB * tiles_y;
    const int units = (g_S + NS - 1) / NS;
    int u, tile;
    {   // tiles of one unit on one XCD (see rotate_bwd_planned_kernel)
        const int per8 = 8 * tiles, octet = blockIdx.x / per8, rem = blockIdx.x - octet * per8;
        if ((octet + 1) * 8 <= units) {
            tile = rem >> 3;
            u = octet * 8 + (rem & 7);
        } else {
            u = octet * 8 + rem / tiles;
            tile = rem - (rem / tiles) * tiles;
        }
    }
    const int s = u * NS;
    const bool has2 = NS == 2 && s + 1 < g_S;
    const float k0 = scale.at(s), k1 = has2 ? scale.at(s + 1) : 1.0f;
    const int xb = tile % nXB, ty = tile / nXB;
    const int xcol = xb * 64 + lane;
    const int yq = min(ty * nwaves + wave, HQ - 1);   // a wave past the slice re-reads the last row quad, never stores
    const bool wave_live = ty * nwaves + wave < HQ;
    const float *gs = gsino + (size_t)s * n_sel * g.PW;
    const unsigned *p = idx4 + (size_t)yq * Wpad + xcol;
    const size_t astride = (size_t)HQ * Wpad;
    vec_t acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = 0.0f;

    auto load16 = [&](int kbase, unsigned (&q)[16]) {   // the taps of 16 selected angles: one dword per lane each
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = min(kbase + e, n_sel - 1);
            const int a = min(max(HOSTSEL ? selh.get(k) : sel[k], 0), g.A - 1);   // a bad index cannot leave the plan
            q[e] = p[(size_t)a * astride];
        }
    };
    for (int kc = 0; kc < n_sel; kc += kChunk) {
        const int nk = min(kChunk, n_sel - kc);
        unsigned q0[16];
        load16(kc, q0);   // in flight while the cotangent rows are staged
        if (kc > 0) __syncthreads();
        for (int t = threadIdx.x; t < nk * NS; t += blockDim.x) lds[((t / NS) * kBwdPitch + 255) * NS + (t % NS)] = 0.0f;
        if constexpr (NS == 1) {
            stage_rows(lds, gs + (size_t)kc * g.PW, nk, g.PW, g.PW, kBwdPitch, false, lane, wave, nwaves);
        } else {
            const float *srcs[2] = {gs + (size_t)kc * g.PW, gs + (has2 ? (size_t)n_sel * g.PW : 0) + (size_t)kc * g.PW};
            stage_rows_interleaved<2>(lds, srcs, nk, g.PW, g.PW, kBwdPitch, false, lane, wave, nwaves);
        }
        // the taps of up to 16 selected angles at a time: scalar loads of their plan angles, then one dword per lane each
        auto block16 = [&](auto k0_tag) {
            constexpr int K0 = decltype(k0_tag)::value;
            if constexpr (K0 < kChunk) {
                if (K0 >= nk) return;   // wave-uniform
                unsigned q[16];
                if constexpr (K0 == 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) q[e] = q0[e];
                    __syncthreads();   // the staged rows
                } else {
                    load16(kc + K0, q);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (K0 + e < nk) {   // wave-uniform
                        int b0, b1, b2, b3;
                        unpack4<NS == 1 ? 2 : 3>(q[e], b0, b1, b2, b3);
                        acc[0] += lds_at_vec<NS>(lds, b0 + (K0 + e) * ROW);
                        acc[1] += lds_at_vec<NS>(lds, b1 + (K0 + e) * ROW);
                        acc[2] += lds_at_vec<NS>(lds, b2 + (K0 + e) * ROW);
                        acc[3] += lds_at_vec<NS>(lds, b3 + (K0 + e) * ROW);
                    }
                }
            }
        };
        block16(std::integral_constant<int, 0>{});
        block16(std::integral_constant<int, 16>{});
        block16(std::integral_constant<int, 32>{});
        block16(std::integral_constant<int, 48>{});
    }
    if (xcol < g.W && wave_live) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int y = 4 * yq + e;
            if (y < g.H) {
                if constexpr (NS == 1) {
                    gimg[((size_t)s * g.H + y) * g.W + xcol] = k0 * acc[e];
                } else {
                    gimg[((size_t)s * g.H + y) * g.W + xcol] = k0 * acc[e].x;
                    if (has2) gimg[((size_t)(s + 1) * g.H + y) * g.W + xcol] = k1 * acc[e].y;
                }
            }
        }
    }
}

}  // namespace ctpvae

using namespace ctpvae;

extern "C" {

int ctpvae_rotate_plan_supported(int H, int W, int PH, int PW, int A, int interp, int which)
{
    if (interp != CTPVAE_NEAREST || H <= 0 || W <= 0 || PH < H || PW < W || A <= 0) return 0;
    if (knob(kKnobNoPlan) >= 0) return 0;
    if (which == 0 && knob(kKnobTiledForce) == 1) return 0;
    const PlanGeom g{H, W, PH, PW, 0, 0, A};
    return which == 0 ? (fwd_plan_fits(g) ? 1 : 0) : (bwd_plan_fits(g) ? 1 : 0);
}

long long ctpvae_rotate_plan_bytes(int H, int W, int PH, int PW, int A, int which)
{
    if (H <= 0 || W <= 0 || PH < H || PW < W || A <= 0) return fail(CTPVAE_EINVAL, "rotate_plan_bytes: bad sizes");
    const PlanGeom g{H, W, PH, PW, 0, 0, A};
    return which == 0 ? fwd_layout(g).bytes : bwd_layout(g).bytes;
}

int ctpvae_rotate_plan_build_f32(const float *T8_dev, const float *Tinv8_dev, int A, int H, int W, int PH, int PW, int py,
                                 int px, void *fwd_plan_dev, void *bwd_plan_dev, ctpvae_stream_t stream)
{
    if (int rc = check_plan_geom("rotate_plan_build", H, W, PH, PW, py, px, A)) return rc;
    CTPVAE_REQUIRE(fwd_plan_dev || bwd_plan_dev, "rotate_plan_build: no plan buffer given");
    const PlanGeom g{H, W, PH, PW, py, px, A};
    if (fwd_plan_dev) {
        CTPVAE_REQUIRE(T8_dev, "rotate_plan_build: forward plan needs the forward transforms");
        CTPVAE_REQUIRE(fwd_plan_fits(g), "rotate_plan_build: a %dx%d slice does not fit the forward plan's LDS image", H, W);
        CTPVAE_REQUIRE(A <= 65535, "rotate_plan_build: at most 65535 angles");
        const FwdLayout L = fwd_layout(g);
        CTPVAE_REQUIRE(L.Galloc <= 65535, "rotate_plan_build: canvas too tall");
        CTPVAE_HIP(hipMemsetAsync((char *)fwd_plan_dev + L.off_rng, 0x7f, (size_t)A * L.nJB * 8, (hipStream_t)stream));
        hipLaunchKernelGGL(rotate_fwd_first_kernel, dim3(L.nJB, A), dim3(64), 0, (hipStream_t)stream, g, T8_dev, L,
                           (char *)fwd_plan_dev);
        CTPVAE_LAUNCH_CHECK("rotate_fwd_first_kernel");
        hipLaunchKernelGGL(rotate_fwd_plan_kernel, dim3(L.nJB, A, L.Galloc), dim3(64), 0, (hipStream_t)stream, g, T8_dev, L,
                           (char *)fwd_plan_dev);
        CTPVAE_LAUNCH_CHECK("rotate_fwd_plan_kernel");
        hipLaunchKernelGGL(rotate_class_list_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, L, (char *)fwd_plan_dev);
        CTPVAE_LAUNCH_CHECK("rotate_class_list_kernel");
    }
    if (bwd_plan_dev) {
        CTPVAE_REQUIRE(Tinv8_dev, "rotate_plan_build: backward plan needs the inverted transforms");
        CTPVAE_REQUIRE(bwd_plan_fits(g), "rotate_plan_build: the backward plan stores bins as bytes (PW=%d > 255)", PW);
        CTPVAE_REQUIRE(H <= 65535, "rotate_plan_build: at most 65535 rows");
        const BwdLayout L = bwd_layout(g);
        hipLaunchKernelGGL(rotate_bwd_plan_kernel, dim3(L.nXB, H, L.NA16), dim3(64), 0, (hipStream_t)stream, g, Tinv8_dev, L,
                           (uint4 *)bwd_plan_dev);
        CTPVAE_LAUNCH_CHECK("rotate_bwd_plan_kernel");
    }
    return CTPVAE_OK;
}

// sel_dev == nullptr: all A angles of the plan; otherwise the n_sel plan angles sel_dev[0..n_sel) (device int32), in
// that order, are projected -- the launch shape is then sized for n_sel angles.
// Launches of several rounds (round 4, tools/sweep_fwd.py at 100 to 400 slices x 20 / 90 / 180 angles, profiles/r04_sweep_G.txt):
// a round of workgroups costs its fill and tasks plus ~2.5 us of ramp and drain (kRoundKb in the model's KB-per-CU currency), and
// a launch of r.x rounds takes nearer to ceil(r.x) than to r.x of them.  Fitted so that the model picks the measured best task
// groups on all eight shapes, both plan formats (the previous model: up to 10 % off, e.g. G = 5 where G = 3 was 9 % faster).
constexpr double kRoundKb = 150.0;
static inline double launch_rounds(long long wgs)
{
    const double real = (double)wgs / 256.0, whole = std::ceil(real);
    return whole > 1.0 ? real + 0.75 * (whole - real) : 1.0;
}

static int launch_fwd_planned(const float *img_dev, int S, int H, int W, int PH, int PW, int A, const void *fwd_plan_dev,
                              float *sino_dev, const LogLikEpilogue &epi, const int *sel_dev, int n_sel,
                              ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(img_dev && fwd_plan_dev && sino_dev, "rotate_fwd_planned: null pointer");
    CTPVAE_REQUIRE(S > 0, "rotate_fwd_planned: need at least one slice");
    if (int rc = check_plan_geom("rotate_fwd_planned", H, W, PH, PW, 0, 0, A)) return rc;
    CTPVAE_REQUIRE(sel_dev == nullptr || (n_sel >= 1 && n_sel <= 64 * kSelRounds),
                   "rotate_fwd_planned: an angle subset holds 1..%d angles (got %d); build a plan for larger ones",
                   64 * kSelRounds, n_sel);
    const PlanGeom g{H, W, PH, PW, 0, 0, A};
    CTPVAE_REQUIRE(fwd_plan_fits(g), "rotate_fwd_planned: a %dx%d slice does not fit the plan's LDS image", H, W);
    const FwdLayout L = fwd_layout(g);
    const int A_run = sel_dev ? n_sel : A;      // angles this launch projects
    const int T = A_run * L.nJB;   // (angle, bin block) tasks per slice
    // Launch shape.  What a launch costs is the bytes its busiest CU pulls through its L2->CU path (DESIGN.md section
    // 6): per workgroup one staged unit -- a slice, or a PAIR of slices interleaved as float2, whose index stream, unpack
    // and ds_read_b64 serve both -- plus ~1 KB of indices per row group of each of its tasks; the busiest CU holds
    // ceil(workgroups / 256) of them.  The tasks of a unit are dealt to G groups per mirror class; take the (pairing, G)
    // that minimises the cost (measured at B=50, A=20: pairs G=5 7.9 us, singles G=5 8.9 us, G=2 10.3 us, G=3 12.3 us,
    // as the model orders them).  Near-ties go to more workgroups (more staging overlaps tasks); every group re-stages
    // the unit, so beyond ~12 groups more workgroups cost more than they buy (B=5: 8.1 us at 12, 13.9 us at 30).
    const bool pairs_fit = (size_t)(L.zero + 1) * 8 + 16 <= (size_t)kMaxLdsBytes && S >= 2;
    int ns = 1, G = 1, affine = 0;
    {
        const double task_kb = 0.6 * L.NG;   // ~1 KB per row group, ~0.6 NG groups per task (rays aligned at their own first row)
        const double plan_kb = (double)L.bytes * A_run / A / 1024.0;
        double best = 0.0, best_aff = 0.0;
        int ns_aff = 1, G_aff = 4;
        for (int cand_ns = 1; cand_ns <= (pairs_fit ? 2 : 1); ++cand_ns) {
            const double fill_kb = (double)g.H * g.W * 4.0 * cand_ns / 1024.0;
            const long long cand_units = (S + cand_ns - 1) / cand_ns;
            // the index table is streamed into the L2 of every XCD that holds a unit (units are dealt to XCDs by
            // octets); with few units and many angles that fabric traffic, not the CU's own path, is what pairing
            // halves (B=8, A=180: 19 us single, 12 us paired).  In the same KB-per-CU currency, fabric at ~2/3 of
            // the CUs' aggregate rate:
            const double fabric_kb = 1.5 * plan_kb * (double)std::min<long long>(8, cand_units) / 256.0;
            for (int cand = 1; cand <= std::min(12, std::max(1, T / 2)); ++cand) {
                const long long wgs = 2ll * cand_units * cand;
                const double cost = launch_rounds(wgs) * (fill_kb + kRoundKb + task_kb * T / (2.0 * cand)) + fabric_kb;
                // near-ties go to more workgroups -- inside ONE round only (more staging overlaps tasks); a launch of several
                // rounds pays for every extra workgroup
                if (best == 0.0 || cost <= best * (wgs <= 256 ? 1.03 : 1.0)) {
                    best = best == 0.0 ? cost : std::min(best, cost);
                    ns = cand_ns;
                    G = cand;
                }
            }
            // angles dealt to the XCDs (round 4; see the kernel): the plan crosses the fabric once in all -- an eighth per XCD,
            // resident from launch to launch --, every unit eight times.  Task groups in fours: workgroup wg runs on XCD wg % 8.
            // Measured (tools/time_affine.py, profiles/r04_time_affine.txt): it wins where the plan stream was the bound -- 8 to
            // 16 units at >= 90 angles (A = 180: B = 16 17.2 -> 10.5 us, B = 32 18.2 -> 14.2) -- and loses from 20 units on (B = 40
            // 19.5 -> 20.4 us, B = 50 20.3 -> 22.7): an XCD's 4 MB L2 then holds its eighth of the plan (2.1 MB at 180 angles)
            // plus a 128 KB fill per unit no longer, every workgroup's fill misses it (200 fills from the fabric at once), and
            // task groups come in fours, which fills 256 CUs worse.
            const double fabric_aff = 1.5 * (plan_kb / 8.0 + (double)cand_units * fill_kb * 8.0) / 256.0;
            for (int cand = 4; cand <= 12 && 2 * cand <= std::max(8, T) && cand_units <= 16; cand += 4) {
                const long long wgs = 2ll * cand_units * cand;
                const double cost = (double)((wgs + 255) / 256) * (fill_kb + task_kb * T / (2.0 * cand)) + fabric_aff;
                if (best_aff == 0.0 || cost <= best_aff * 1.03) {
                    best_aff = best_aff == 0.0 ? cost : std::min(best_aff, cost);
                    ns_aff = cand_ns;
                    G_aff = cand;
                }
            }
        }
        // dense launches of plans larger than an XCD's L2 only (knob AFFINE: 0 never, 1 whenever the shape allows)
        const bool can = !sel_dev && T >= 8 && best_aff > 0.0;
        if (can && knob(kKnobAffine) != 0 && (knob(kKnobAffine) == 1 || (plan_kb > 3072.0 && best_aff < best))) {
            affine = 1;
            ns = ns_aff;
            G = G_aff;
        }
    }
    if (knob(kKnobNs) >= 0) {
        const int want = knob(kKnobNs) == 2 ? 2 : 1;
        if (want != ns) {   // forced pairing: best G for it
            ns = want;
            affine = 0;
            const double task_kb = 0.6 * L.NG, fill_kb = (double)g.H * g.W * 4.0 * ns / 1024.0;
            const long long cand_units = (S + ns - 1) / ns;
            double best = 0.0;
            for (int cand = 1; cand <= std::min(12, std::max(1, T / 2)); ++cand) {
                const long long wgs = 2ll * cand_units * cand;
                const double cost = launch_rounds(wgs) * (fill_kb + kRoundKb + task_kb * T / (2.0 * cand));
                if (best == 0.0 || cost <= best * (wgs <= 256 ? 1.03 : 1.0)) {
                    best = best == 0.0 ? cost : std::min(best, cost);
                    G = cand;
                }
            }
        }
    }
    const size_t shmem = (size_t)(L.zero + 1) * sizeof(float) * ns + 16;   // + the task counter
    const int units = (S + ns - 1) / ns;
    if (knob(kKnobG) > 0) {
        G = knob(kKnobG);
        if (G % 4 != 0) affine = 0;
    }
    // one wave per task of the busiest group, but never fewer than stage the unit in ONE batch of eight 16-byte loads
    // per lane (64 KiB -> 8 waves): a workgroup of 6 waves spends two load round trips on its fill (B=50, G=5:
    // 12.1 us with 6 waves, 8.7 us with 8)
    const int stage_waves = (int)std::min<size_t>(16, ceil_div((int)(shmem / 1024), 8));
    int waves = std::min(16, std::max(stage_waves, (T + 2 * G - 1) / (2 * G)));
    if (knob(kKnobWaves) > 0) waves = std::min(16, knob(kKnobWaves));
    int wgs_per_slice = 2 * G;
    CTPVAE_REQUIRE((long long)units * wgs_per_slice < (1ll << 31), "rotate_fwd_planned: too many slices");
#ifdef CTPVAE_TUNE_STAMPS
    g_pshape[0] = units, g_pshape[1] = wgs_per_slice, g_pshape[2] = waves, g_pshape[3] = ns, g_pshape[4] = affine;
#endif
    // Launches of more than one round of workgroups (round 5; tools/stamp_rounds.hip + tools/analyse_rounds.py, profiles/r05_rounds.txt):
    // a CU's next workgroup starts 0.3-0.5 us after its previous one ended -- the dispatcher is not what a later round waits
    // for -- and then spends 3.7-4.0 us before its unit is staged, as every workgroup does; what a launch of r.x rounds loses is
    // QUANTISATION: 300 equal pieces on 256 CUs take two rounds for 1.17 rounds of work (B = 300 x 20 angles: 25 us against
    // 14.5 at B = 256).  So the piece list gets TWO PARTS: G1 task groups per class for the first units1 units -- whole rounds
    // of coarse pieces: one fill per unit and class, waves evenly loaded -- and G2 > G1 for the rest, so that the last round's
    // work spreads over all CUs.  (G1, G2, units1) by list scheduling of the pieces, in order, on 256 CUs: a piece costs ~4.4 us
    // from its CU falling free to its staged unit plus ceil(tasks / 16 waves) task rounds at 0.12-0.19 us per row group (2.7 us with 6 waves gathering, 4.35 us at
    // 128 x 128 with every CU busy).  A PERSISTENT form of the kernel -- 256 workgroups walking the list, the next unit's rows
    // requested by the waves that run out of tasks first -- was built and measured: bit-equal, its hand-over no cheaper than a
    // fresh workgroup (~3.4 us behind the slowest wave) and its task loop 5 % slower (127 registers, another schedule): removed.
    int units1 = units, wgs2 = wgs_per_slice, units2 = units, wgs3 = wgs_per_slice;
    if (!affine && !sel_dev && (long long)units * wgs_per_slice > 256 && knob(kKnobMixG) != 0 && knob(kKnobG) <= 0) {
        // (a task round is the faster the fewer waves of the CU gather at once: 2.7 us with 6 waves, 4.35 with all 16, at 23 row groups)
        const double t_fresh = 4.4;
        const int Tc = (T + 1) / 2;     // tasks per unit and class
        auto piece_us = [&](int Gx) {
            const int tp = (Tc + Gx - 1) / Gx;
            return t_fresh + std::ceil(tp / 16.0) * (0.072 + 0.0073 * std::min(16, tp)) * L.NG;
        };
        auto launch_us = [&](int G1, int G2, int u1) {   // pieces in order, each to the CU that falls free first
            const long long Q1 = 2ll * u1 * G1, Q2 = 2ll * (units - u1) * G2;
            const double t1 = piece_us(G1), t2 = piece_us(G2);
            // part 1: r1 whole rounds + rem1 pieces; the CUs fall free in two groups, part 2's equal pieces go to whichever is earlier
            const long long r1 = Q1 / 256, rem1 = Q1 % 256;
            double t_free[2] = {r1 * t1, (r1 + 1) * t1};
            long long n_cu[2] = {256 - rem1, rem1}, left = Q2;
            double end = Q1 > 0 ? (rem1 ? t_free[1] : t_free[0]) : 0.0;
            while (left > 0) {
                const int k = n_cu[1] == 0 || t_free[0] <= t_free[1] ? 0 : 1;
                left -= std::min(left, n_cu[k]);
                t_free[k] += t2;
                end = std::max(end, t_free[k]);
            }
            return end;
        };
        int G1 = G, G2 = G, G3 = G;
        // The search below costs 2-15 us of host time at a few hundred units (it was 0.9 ms at 2000 units before its first parts were
        // limited to the last whole rounds): the last few shapes' cuts are kept per thread.
        struct CutKey { int units, T, NG, G, mix3; };
        struct Cut { CutKey key; int G1, G2, G3, units1, units2; };
        static thread_local Cut cut_cache[8];
        static thread_local unsigned cut_next = 0;
        const CutKey key{units, T, L.NG, G, knob(kKnobMixG3) != 0 ? 1 : 0};
        const Cut *hit = nullptr;
        for (const Cut &c : cut_cache)
            if (c.key.units == key.units && c.key.T == key.T && c.key.NG == key.NG && c.key.G == key.G && c.key.mix3 == key.mix3) hit = &c;
        if (hit) {
            G1 = hit->G1, G2 = hit->G2, G3 = hit->G3, units1 = hit->units1, units2 = hit->units2;
        } else {
        double best = launch_us(G, G, units);
        const int gmax = std::min(12, std::max(1, Tc / 4));
        for (int c1 = 1; c1 <= std::min(8, gmax); ++c1) {
            // first parts of (about) k whole rounds -- the last four such boundaries: a finer rest of more rounds than that never
            // won -- and the whole launch
            const int kmax = (int)((2ll * c1 * units) / 256);
            for (int c2 = c1; c2 <= gmax; ++c2)
                for (int k = std::max(0, kmax - 3);; ++k) {
                    const int u1 = std::min(units, (int)((256ll * k) / (2 * c1)));
                    const double t = launch_us(c1, c2, u1);
                    if (t < best * 0.97) best = t, G1 = c1, G2 = c2, units1 = u1;
                    if (u1 == units) break;
                }
        }
        if (units1 == units) G2 = G1;
        // a THIRD part, finer again, for the end of the second: its pieces fill the CUs that the second part's last round leaves
        // idle (B = 400 x 180 angles: [128 units x 1][40 x 3][32 x 4] 113-115 us against [128 x 1][72 x 3] 117-118)
        G3 = G2;
        if (units1 < units && knob(kKnobMixG3) != 0) {
            auto launch3_us = [&](int c3, int u2) {
                struct Free { double t; long long n; } f[16] = {{0.0, 256}};
                int nf = 1;
                const long long cnt[3] = {2ll * units1 * G1, 2ll * (u2 - units1) * G2, 2ll * (units - u2) * c3};
                const double cost[3] = {piece_us(G1), piece_us(G2), piece_us(c3)};
                double end = 0.0;
                for (int part = 0; part < 3; ++part)
                    for (long long left = cnt[part]; left > 0;) {
                        int k = 0;
                        for (int i = 1; i < nf; ++i) if (f[i].t < f[k].t) k = i;
                        const long long n = std::min(left, f[k].n);
                        if (n < f[k].n && nf < 16) f[nf++] = {f[k].t, f[k].n - n}, f[k].n = n;
                        f[k].t += cost[part];
                        end = std::max(end, f[k].t);
                        left -= n;
                    }
                return end;
            };
            double best3 = launch3_us(G2, units);
            for (int c3 = G2 + 1; c3 <= std::min(gmax, G2 + 3); ++c3)
                for (int u2 = units1 + 4; u2 < units; u2 += 4) {
                    const double t = launch3_us(c3, u2);
                    if (t < best3 * 0.98) best3 = t, G3 = c3, units2 = u2;
                }
            if (G3 == G2) units2 = units;
        }
        cut_cache[cut_next++ % 8] = Cut{key, G1, G2, G3, units1, units2};
        }
        if (knob(kKnobMixG2) > 0 && knob(kKnobMixU1) >= 0)
            G1 = knob(kKnobMixG1) > 0 ? knob(kKnobMixG1) : G, G2 = knob(kKnobMixG2), units1 = std::min(units, knob(kKnobMixU1));
        wgs_per_slice = 2 * G1;
        wgs2 = 2 * G2, wgs3 = 2 * G3;
        if (knob(kKnobMixG2) > 0 && knob(kKnobMixU1) >= 0) wgs3 = wgs2, units2 = units;
        if (knob(kKnobMixG3) > 0 && knob(kKnobMixU2) >= 0)
            wgs3 = 2 * knob(kKnobMixG3), units2 = std::max(units1, std::min(units, knob(kKnobMixU2)));
        waves = std::min(16, std::max(stage_waves, (T + 2 * G1 - 1) / (2 * G1)));
        if (knob(kKnobWaves) > 0) waves = std::min(16, knob(kKnobWaves));
    }
    units2 = std::max(units1, std::min(units2, units));
    const long long grid = (long long)units1 * wgs_per_slice + (long long)(units2 - units1) * wgs2 + (long long)(units - units2) * wgs3;
    CTPVAE_REQUIRE(grid < (1ll << 31), "rotate_fwd_planned: too many slices");
#ifdef CTPVAE_TUNE_STAMPS
    g_pshape[1] = wgs_per_slice, g_pshape[2] = waves, g_pshape[5] = units1, g_pshape[6] = wgs2, g_pshape[7] = units2, g_pshape[8] = wgs3;   // (timing builds: tools/stamp_rounds.hip)
#endif
    auto launch = [&](auto kernel) -> int {
        static std::atomic<unsigned long long> attr_set{0}, abs_ok{0};   // per kernel instantiation: devices done
        CTPVAE_REQUIRE_NO_STATIC_LDS(kernel, "rotate_fwd_planned_kernel", abs_ok);
        CTPVAE_SET_MAX_LDS_ONCE(kernel, attr_set);
        // divisions by multiplication (div_magic): piece numbers / 8 by the pieces per unit, task numbers by a class's angle count
        const int small_div = (grid / 8 + 1) * (long long)std::max(wgs_per_slice, std::max(wgs2, wgs3)) < (1ll << 32) && (long long)A * L.nJB * A < (1ll << 32) &&
                                      knob(kKnobNoMagic) <= 0 ? 1 : 0;
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * waves), shmem, (hipStream_t)stream,
                           img_dev, g, L, (const char *)fwd_plan_dev, wgs_per_slice, S, sino_dev, epi, sel_dev, n_sel, affine, units1, wgs2,
                           div_magic((unsigned)wgs_per_slice), div_magic((unsigned)wgs2), small_div, units2, wgs3, div_magic((unsigned)wgs3));
        CTPVAE_LAUNCH_CHECK("rotate_fwd_planned_kernel");
        return CTPVAE_OK;
    };
    if (sel_dev) {
        if (epi.lp) return ns == 2 ? launch(rotate_fwd_planned_kernel<2, true, true>) : launch(rotate_fwd_planned_kernel<1, true, true>);
        return ns == 2 ? launch(rotate_fwd_planned_kernel<2, false, true>) : launch(rotate_fwd_planned_kernel<1, false, true>);
    }
    if (epi.lp) return ns == 2 ? launch(rotate_fwd_planned_kernel<2, true, false>) : launch(rotate_fwd_planned_kernel<1, true, false>);
    return ns == 2 ? launch(rotate_fwd_planned_kernel<2, false, false>) : launch(rotate_fwd_planned_kernel<1, false, false>);
}

int ctpvae_rotate_fwd_planned_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int A, const void *fwd_plan_dev,
                                  float *sino_dev, ctpvae_stream_t stream)
{
    return launch_fwd_planned(img_dev, S, H, W, PH, PW, A, fwd_plan_dev, sino_dev, LogLikEpilogue{}, nullptr, 0, stream);
}

int ctpvae_rotate_fwd_planned_sel_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int A,
                                      const void *fwd_plan_dev, const int *angle_idx_dev, int n_idx, float *sino_dev,
                                      ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(angle_idx_dev, "rotate_fwd_planned_sel: null angle index");
    return launch_fwd_planned(img_dev, S, H, W, PH, PW, A, fwd_plan_dev, sino_dev, LogLikEpilogue{}, angle_idx_dev, n_idx,
                              stream);
}

int ctpvae_rotate_fwd_planned_loglik_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int A,
                                         const void *fwd_plan_dev, const float *mask_dev, const float *meas_dev,
                                         const float *pnm_dev, float eps, float *sino_dev, float *lp_dev,
                                         float *dlp_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(mask_dev && meas_dev && pnm_dev && lp_dev, "rotate_fwd_planned_loglik: null pointer");
    return launch_fwd_planned(img_dev, S, H, W, PH, PW, A, fwd_plan_dev, sino_dev,
                              LogLikEpilogue{mask_dev, meas_dev, pnm_dev, eps, lp_dev, dlp_dev, 0}, nullptr, 0, stream);
}

int ctpvae_rotate_fwd_planned_loglik_sel_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int A,
                                             const void *fwd_plan_dev, const int *angle_idx_dev, int n_idx,
                                             const float *mask_dev, const float *meas_dev, int dense_inputs,
                                             const float *pnm_dev, float eps, float *sino_dev, float *lp_dev,
                                             float *dlp_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(angle_idx_dev && mask_dev && meas_dev && pnm_dev && lp_dev, "rotate_fwd_planned_loglik_sel: null pointer");
    return launch_fwd_planned(img_dev, S, H, W, PH, PW, A, fwd_plan_dev, sino_dev,
                              LogLikEpilogue{mask_dev, meas_dev, pnm_dev, eps, lp_dev, dlp_dev, dense_inputs ? 1 : 0},
                              angle_idx_dev, n_idx, stream);
}

int ctpvae_rotate_bwd_planned_f32(const float *gsino_dev, int S, int H, int W, int PH, int PW, int A, const void *bwd_plan_dev,
                                  float *gimg_dev, ctpvae_stream_t stream)
{
    return ctpvae_rotate_bwd_planned_scaled_f32(gsino_dev, S, H, W, PH, PW, A, bwd_plan_dev, nullptr, 0, gimg_dev, stream);
}

// dup = 1: the TF-compatible plan (one tap per angle and pixel); dup = 2: the exact-transpose plan (two)
static int launch_bwd_planned(const float *gsino_dev, int S, int H, int W, int PH, int PW, int A, const void *bwd_plan_dev,
                              const float *scale_dev, long long scale_stride, float *gimg_dev, int dup,
                              ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(gsino_dev && bwd_plan_dev && gimg_dev, "rotate_bwd_planned: null pointer");
    CTPVAE_REQUIRE(S > 0, "rotate_bwd_planned: need at least one slice");
    if (int rc = check_plan_geom("rotate_bwd_planned", H, W, PH, PW, 0, 0, A)) return rc;
    const PlanGeom g{H, W, PH, PW, 0, 0, A};
    CTPVAE_REQUIRE(bwd_plan_fits(g), "rotate_bwd_planned: the backward plan stores bins as bytes (PW=%d > 255)", PW);
    const BwdLayout L = bwd_layout(g, dup);
    // Two slices per workgroup (one index stream, one unpack and one ds_read_b64 per tap serve both; each lane then
    // owns 2 rows instead of 4) once the batch is large enough to still fill the chip: measured B=50: 7.6 -> 6.4 us at
    // A=20, 16.5 -> 13.8 us at A=90, 28.3 -> 25.4 us at A=180.  Small batches pair up too, in short 4-wave tiles
    // (tools/sweep_bwd.py: 4.7 -> 4.1 us at S=2..16, 5.3 -> 4.8 us at S=24, A=20; even at A=90).
    int ns = S >= 2 ? 2 : 1;
    // (slices of ONE column of tiles, 33 .. 64 pixels wide, make few workgroups, and every workgroup of a pair stages both slices' rows:
    // single slices in 16-row tiles until the launch has ~512 of them; tools/sweep_nearest_rules.py with N=64: 50 x 90 angles 11.7 ->
    // 9.4 us, 20 x 180: 21.0 -> 16.0, 50 x 20: 5.5 -> 4.8; N=48: 50 x 90 10.9 -> 8.8; at 32 x 32 the pairs stay ahead)
    const bool narrow = L.nXB == 1 && H > 32 && (long long)S * ceil_div(H, 16) < 512;
    if (narrow) ns = 1;
    if (knob(kKnobBns) >= 0) ns = (knob(kKnobBns) == 2 && S >= 2) ? 2 : 1;
    const int ppt = ns == 2 ? 2 : 4;
    const int units = ceil_div(S, ns);
    // staged chunk: up to 64 rows of one slice, or 32 rows of an interleaved pair
    const size_t shmem = (size_t)std::min(ceil_div(L.NA16 * 16, dup), kBwdChunk / ns) * L.pitchg * sizeof(float) * ns;
    // Tile = 64 columns x (waves x ppt) rows.  Every workgroup of a slice stages ALL the slice's cotangent rows: with many
    // angles taller tiles amortise that staging (measured, B=50 A=180: 44 -> 28 us from 4 to 16 waves); with few angles
    // the staging is small and short tiles win, because four small workgroups per CU overlap each other's staging and
    // barrier waits while one 16-wave workgroup (114 VGPRs: one per CU) cannot (B=400 A=20: 34 us vs 45 us).
    const int Aeff = A * dup;
    int waves = 4;
    if (ns == 2) {   // pairs: 64 x 16-row tiles for few angles, 64 x 32 for many; half as tall below 32 slices (sweeps)
        waves = Aeff >= 64 ? (S >= 32 ? 16 : 8) : (S > 16 ? 8 : 4);
        // (32 .. 63 angles, second look at other image sizes -- at 20 angles the short tiles win whatever the rounds: 16-row tiles that need a second round of workgroups where 32-row tiles fit in
        // one lose it all -- 64 x 96^2 x 45 angles: 384 workgroups 13.2 us, 192 of twice the height 9.9)
        if (Aeff >= 32 && Aeff < 64 && waves == 8 && (long long)units * L.nXB * ceil_div(H, 8 * ppt) > 256 && (long long)units * L.nXB * ceil_div(H, 16 * ppt) <= 256)
            waves = 16;
        // Many angles (round 4, tools/sweep_bwd_waves.py, profiles/r04_sweep_bwd_waves.txt): a workgroup's cost is mostly the
        // staging of ALL its pair's cotangent rows (~14 of 18 us at 180 angles), whatever its height -- so (a) while 16-row tiles
        // are at most one workgroup per CU they win (B=32: 15.2 vs 17.3 us), and (b) one more row of tiles (26-row tiles of 13
        // waves for H = 128) is free while it adds no round: 250 instead of 200 workgroups at B=50, 17.3 vs 18.1 us.
        if (Aeff >= 64 && S >= 32) {
            auto wgs = [&](int w) { return (long long)units * L.nXB * ceil_div(H, w * ppt); };
            const int w_more = ceil_div(H, ppt * (ceil_div(H, 16 * ppt) + 1));   // waves of a tile one row of tiles shorter
            if (wgs(8) <= 256)
                waves = 8;
            else if (w_more >= 9 && w_more < 16 && ceil_div((int)wgs(w_more), 256) == ceil_div((int)wgs(16), 256))
                waves = w_more;
        }
    }
    if (ns != 2 && Aeff >= 32 && !narrow)
        while (waves < 16 && (long long)units * L.nXB * ceil_div(H, 2 * waves * ppt) >= 200 && waves * ppt < H) waves *= 2;
    while (waves > 1 && (waves / 2) * ppt >= H) waves /= 2;   // tiny slices: no more rows per tile than the slice has
    if (knob(kKnobBw) > 0) waves = std::min(16, knob(kKnobBw));
    const int rows_per_wg = waves * ppt;
#ifdef CTPVAE_TUNE_STAMPS
    g_pshape[0] = units, g_pshape[1] = L.nXB * ceil_div(H, waves * ppt), g_pshape[2] = waves, g_pshape[3] = ns, g_pshape[4] = 0, g_pshape[5] = units, g_pshape[6] = g_pshape[1];
#endif
    const int tiles_y = ceil_div(H, rows_per_wg);
    const long long nblk = (long long)units * L.nXB * tiles_y;
    CTPVAE_REQUIRE(nblk < (1ll << 31), "rotate_bwd_planned: too many slices");
    // (block numbers / 8 by the tiles per unit, by multiplication: exact while (nblk / 8) * tiles < 2^32; a unit of ONE tile
    // divides as before: its word would be 0, which the kernel reads as "divide")
    const bool small = (nblk / 8) * (long long)(L.nXB * tiles_y) < (1ll << 32) && L.nXB * tiles_y > 1 && knob(kKnobNoMagic) <= 0;
    auto launch = [&](auto kernel) -> int {
        static std::atomic<unsigned long long> attr_set{0}, abs_ok{0};   // per kernel instantiation: devices done
        CTPVAE_REQUIRE_NO_STATIC_LDS(kernel, "rotate_bwd_planned_kernel", abs_ok);
        CTPVAE_SET_MAX_LDS_ONCE(kernel, attr_set);
        hipLaunchKernelGGL(kernel, dim3((unsigned)nblk), dim3(64 * waves), shmem, (hipStream_t)stream, gsino_dev, g, L,
                           (const uint4 *)bwd_plan_dev, tiles_y, S, SliceScale{scale_dev, scale_stride}, gimg_dev,
                           small ? div_magic((unsigned)(L.nXB * tiles_y)) : 0u, div_magic((unsigned)L.nXB));
        return CTPVAE_OK;
    };
    int rc;
    if (dup == 2) {
        if (ns == 2)
            rc = waves <= 4 ? launch(rotate_bwd_planned_kernel<2, 256, 2, 2>) : launch(rotate_bwd_planned_kernel<2, 1024, 2, 2>);
        else
            rc = waves <= 4 ? launch(rotate_bwd_planned_kernel<4, 256, 1, 2>) : launch(rotate_bwd_planned_kernel<4, 1024, 1, 2>);
    } else if (A <= 32) {   // two index groups at most: both requested up front
        if (ns == 2)
            rc = waves <= 4 ? launch(rotate_bwd_planned_kernel<2, 256, 2, 1, true>) : launch(rotate_bwd_planned_kernel<2, 1024, 2, 1, true>);
        else
            rc = waves <= 4 ? launch(rotate_bwd_planned_kernel<4, 256, 1, 1, true>) : launch(rotate_bwd_planned_kernel<4, 1024, 1, 1, true>);
    } else if (ns == 2) {
        rc = waves <= 4 ? launch(rotate_bwd_planned_kernel<2, 256, 2>) : launch(rotate_bwd_planned_kernel<2, 1024, 2>);
    } else {
        rc = waves <= 4 ? launch(rotate_bwd_planned_kernel<4, 256, 1>) : launch(rotate_bwd_planned_kernel<4, 1024, 1>);
    }
    if (rc) return rc;
    CTPVAE_LAUNCH_CHECK("rotate_bwd_planned_kernel");
    return CTPVAE_OK;
}

int ctpvae_rotate_bwd_planned_scaled_f32(const float *gsino_dev, int S, int H, int W, int PH, int PW, int A,
                                         const void *bwd_plan_dev, const float *scale_dev, long long scale_stride,
                                         float *gimg_dev, ctpvae_stream_t stream)
{
    return launch_bwd_planned(gsino_dev, S, H, W, PH, PW, A, bwd_plan_dev, scale_dev, scale_stride, gimg_dev, 1, stream);
}

// ---- angle subsets of a dense backward plan ---------------------------------------------------------------------------------
long long ctpvae_rotate_bwd4_plan_bytes(int H, int W, int PH, int PW, int A)
{
    if (H <= 0 || W <= 0 || PH < H || PW < W || A <= 0) return fail(CTPVAE_EINVAL, "rotate_bwd4_plan_bytes: bad sizes");
    const PlanGeom g{H, W, PH, PW, 0, 0, A};
    if (!bwd_plan_fits(g) || knob(kKnobNoPlan) >= 0) return 0;   // bins do not fit a byte: keep ctpvae_rotate_bwd_sel_scaled_f32
    return (long long)A * ceil_div(H, 4) * (ceil_div(W, 64) * 64) * 4;
}

int ctpvae_rotate_bwd4_plan_build_f32(const float *Tinv8_dev, int A, int H, int W, int PH, int PW, int py, int px,
                                      void *plan_dev, ctpvae_stream_t stream)
{
    if (int rc = check_plan_geom("rotate_bwd4_plan_build", H, W, PH, PW, py, px, A)) return rc;
    CTPVAE_REQUIRE(Tinv8_dev && plan_dev, "rotate_bwd4_plan_build: null pointer");
    const PlanGeom g{H, W, PH, PW, py, px, A};
    CTPVAE_REQUIRE(bwd_plan_fits(g), "rotate_bwd4_plan_build: the plan stores bins as bytes (PW=%d > 255)", PW);
    const int HQ = ceil_div(H, 4), nXB = ceil_div(W, 64);
    CTPVAE_REQUIRE(HQ <= 65535 && A <= 65535, "rotate_bwd4_plan_build: at most 262140 rows and 65535 angles");
    hipLaunchKernelGGL(rotate_bwd4_plan_kernel, dim3(nXB, HQ, A), dim3(64), 0, (hipStream_t)stream, g, Tinv8_dev, nXB * 64, HQ,
                       (unsigned *)plan_dev);
    CTPVAE_LAUNCH_CHECK("rotate_bwd4_plan_kernel");
    return CTPVAE_OK;
}

int ctpvae_rotate_bwd_planned_sel_scaled_f32(const float *gsino_dev, int S, int H, int W, int PH, int PW, int A,
                                             const void *bwd4_plan_dev, const int *angle_idx, int n_idx, int idx_on_host,
                                             const float *scale_dev, long long scale_stride, float *gimg_dev,
                                             ctpvae_stream_t stream)
{
    const int *angle_idx_dev = angle_idx;   // (host memory when idx_on_host)
    CTPVAE_REQUIRE(gsino_dev && bwd4_plan_dev && angle_idx_dev && gimg_dev, "rotate_bwd_planned_sel: null pointer");
    CTPVAE_REQUIRE(!idx_on_host || n_idx <= kMaxSelAngles, "rotate_bwd_planned_sel: at most %d host-resident angles (got %d)",
                   kMaxSelAngles, n_idx);
    CTPVAE_REQUIRE(S > 0 && n_idx > 0, "rotate_bwd_planned_sel: need slices and angles (S=%d, subset %d)", S, n_idx);
    if (int rc = check_plan_geom("rotate_bwd_planned_sel", H, W, PH, PW, 0, 0, A)) return rc;
    const PlanGeom g{H, W, PH, PW, 0, 0, A};
    CTPVAE_REQUIRE(bwd_plan_fits(g), "rotate_bwd_planned_sel: the plan stores bins as bytes (PW=%d > 255)", PW);
    const int HQ = ceil_div(H, 4), nXB = ceil_div(W, 64);
    int ns = S >= 2 ? 2 : 1;
    if (knob(kKnobBns) >= 0) ns = (knob(kKnobBns) == 2 && S >= 2) ? 2 : 1;
    const int units = ceil_div(S, ns);
    // tile = 64 columns x 4 * waves rows.  Four waves: every workgroup stages all n cotangent rows of its unit, and fewer waves
    // would need more than one batch of loads per lane for it (20 rows of a pair = 29 KB = 7 loads per lane of 256 threads).  MORE
    // waves for larger subsets in LONG launches (round 5): every workgroup stages all n rows again, and from ~768 four-wave workgroups
    // on that redundancy is what the launch costs -- 100 x 128^2, 90 of 180 angles: 40.3 us with four waves, 25.2 with sixteen (256 x:
    // 81 -> 53); in short launches the small workgroups win (5 x, 90 of 180: 18.9 us against 24.2)
    int waves = std::min(4, HQ);
    if ((long long)units * nXB * ceil_div(HQ, 4) >= 768) waves = std::min(n_idx >= 48 ? 16 : (n_idx >= 24 ? 8 : 4), HQ);
    if (knob(kKnobBw) > 0) waves = std::min(16, std::max(1, knob(kKnobBw)));
    const int tiles_y = ceil_div(HQ, waves);
    const long long nblk = (long long)units * nXB * tiles_y;
    CTPVAE_REQUIRE(nblk < (1ll << 31), "rotate_bwd_planned_sel: too many slices");
    const size_t shmem = (size_t)std::min(n_idx, kBwdChunk / ns) * kBwdPitch * sizeof(float) * ns;
    SelHost selh = {};
    if (idx_on_host) {
        for (int k = 0; k < n_idx; ++k) selh.set(k, std::min(std::max(angle_idx_dev[k], 0), A - 1));
        angle_idx_dev = nullptr;
    }
    auto launch = [&](auto kernel) -> int {
        static std::atomic<unsigned long long> abs_ok{0}, attr_set{0};   // per kernel instantiation: devices checked
        CTPVAE_REQUIRE_NO_STATIC_LDS(kernel, "rotate_bwd_planned_sel_kernel", abs_ok);
        CTPVAE_SET_MAX_LDS_ONCE(kernel, attr_set);   // (a full chunk is 256 bytes over 64 KB)
        hipLaunchKernelGGL(kernel, dim3((unsigned)nblk), dim3(64 * waves), shmem, (hipStream_t)stream, gsino_dev, g, nXB * 64, HQ,
                           (const unsigned *)bwd4_plan_dev, angle_idx_dev, n_idx, tiles_y, S, SliceScale{scale_dev, scale_stride},
                           gimg_dev, selh);
        return CTPVAE_OK;
    };
    int rc;
    if (waves > 4) {
        if (idx_on_host)
            rc = ns == 2 ? launch(rotate_bwd_planned_sel_kernel<2, true, 1024>) : launch(rotate_bwd_planned_sel_kernel<1, true, 1024>);
        else
            rc = ns == 2 ? launch(rotate_bwd_planned_sel_kernel<2, false, 1024>) : launch(rotate_bwd_planned_sel_kernel<1, false, 1024>);
    } else if (idx_on_host)
        rc = ns == 2 ? launch(rotate_bwd_planned_sel_kernel<2, true>) : launch(rotate_bwd_planned_sel_kernel<1, true>);
    else
        rc = ns == 2 ? launch(rotate_bwd_planned_sel_kernel<2, false>) : launch(rotate_bwd_planned_sel_kernel<1, false>);
    if (rc) return rc;
    CTPVAE_LAUNCH_CHECK("rotate_bwd_planned_sel_kernel");
    return CTPVAE_OK;
}

// ---- exact transpose through a plan (NEAREST): deterministic gather, see rotate_exact_plan_kernel ----------------------
long long ctpvae_rotate_exact_plan_bytes(int H, int W, int PH, int PW, int A)
{
    if (H <= 0 || W <= 0 || PH < H || PW < W || A <= 0) return fail(CTPVAE_EINVAL, "rotate_exact_plan_bytes: bad sizes");
    const PlanGeom g{H, W, PH, PW, 0, 0, A};
    if (!bwd_plan_fits(g) || knob(kKnobNoPlan) >= 0) return 0;      // bins do not fit a byte: keep the scatter kernel
    return bwd_layout(g, 2).bytes + 256;                              // + the overflow word, on its own line
}

int ctpvae_rotate_exact_plan_build_f32(const float *T8_dev, const float *Tinv8_dev, int A, int H, int W, int PH, int PW,
                                       int py, int px, void *plan_dev, ctpvae_stream_t stream)
{
    if (int rc = check_plan_geom("rotate_exact_plan_build", H, W, PH, PW, py, px, A)) return rc;
    CTPVAE_REQUIRE(T8_dev && Tinv8_dev && plan_dev, "rotate_exact_plan_build: null pointer");
    const PlanGeom g{H, W, PH, PW, py, px, A};
    CTPVAE_REQUIRE(bwd_plan_fits(g), "rotate_exact_plan_build: the plan stores bins as bytes (PW=%d > 255)", PW);
    CTPVAE_REQUIRE(H <= 65535 && A <= 65535, "rotate_exact_plan_build: at most 65535 rows and angles");
    const BwdLayout L = bwd_layout(g, 2);
    // 255 everywhere first: the padding columns and the virtual angles behind 2A must be dead taps
    CTPVAE_HIP(hipMemsetAsync(plan_dev, 0xff, (size_t)L.bytes, (hipStream_t)stream));
    CTPVAE_HIP(hipMemsetAsync((char *)plan_dev + L.bytes, 0, 256, (hipStream_t)stream));
    hipLaunchKernelGGL(rotate_exact_plan_kernel, dim3(L.nXB, H, A), dim3(64), 0, (hipStream_t)stream, g, T8_dev, Tinv8_dev, L,
                       (unsigned char *)plan_dev, (int *)((char *)plan_dev + L.bytes));
    CTPVAE_LAUNCH_CHECK("rotate_exact_plan_kernel");
    return CTPVAE_OK;
}

// 1 if some pixel had more than two hits (the rows are not a rotation): use the scatter kernel.  SYNCHRONISES the stream.
int ctpvae_rotate_exact_plan_overflowed(const void *plan_dev, int H, int W, int PH, int PW, int A, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(plan_dev && H > 0 && W > 0 && A > 0, "rotate_exact_plan_overflowed: bad arguments");
    const PlanGeom g{H, W, PH, PW, 0, 0, A};
    const BwdLayout L = bwd_layout(g, 2);
    int flag = 0;
    CTPVAE_HIP(hipMemcpyAsync(&flag, (const char *)plan_dev + L.bytes, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    CTPVAE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return flag ? 1 : 0;
}

int ctpvae_rotate_bwd_exact_planned_f32(const float *gsino_dev, int S, int H, int W, int PH, int PW, int A,
                                        const void *exact_plan_dev, float *gimg_dev, ctpvae_stream_t stream)
{
    return launch_bwd_planned(gsino_dev, S, H, W, PH, PW, A, exact_plan_dev, nullptr, 0, gimg_dev, 2, stream);
}

}  // extern "C"
