// cplan_walk.h -- the compact (step-coded) gather plan's shared pieces: how a ray is encoded (plan builders) and how a wave
// walks it (kernels).  Used by rotate_cplan.hip (slices that fit LDS) and by the tiled forward of rotate.hip (tiles of a
// slice that does not).  See rotate_cplan.hip for the scheme.
#pragma once
#include <hip/hip_runtime.h>

#include "rotate_plan.h"

namespace ctpvae {

constexpr int kLutBytes = 64 * 32 * 8;   // 64 entries x 32 lane copies x (3 x i16 + pad), at LDS offset 0
constexpr int kRowsPerChunk = 48;       // a uint4 of codes: 16 bytes x 3 rows
constexpr int kRowsPerGroup = 6;        // rows gathered per step of the walk: two code bytes
// (Round 3, built and measured: THREE bytes -- nine rows -- per step, on 12-byte chunks, to shorten a ray's chain of dependent
// steps from 19 to 13 (the u16 plan has 14 of eight taps).  It lost everywhere -- 7.99 vs 7.61 us at the headline shape, 155 vs
// 145 us at B = 400 x 180 angles, 152 vs 139 us for the 512 x 512 tiles, whose four-slice form then spilled: twelve LDS
// operations per step plus the next step's leave a wave at the 15 it may have in flight, and 18 more VGPRs.  Reverted.)
template <> struct SliceVec<4> { typedef float type __attribute__((ext_vector_type(4))); };

// ---- encoding ------------------------------------------------------------------------------------------------------------
struct RawTap {
    int ix, iy;   // rounded source column / row relative to the core's origin (may lie outside the core)
};
// ImageProjectiveTransformV3, NEAREST: (t0*x + t1*y) + t2, std::round -- the expressions of rotate_plan.hip's fwd_tap
__device__ __forceinline__ RawTap raw_tap(const PlanGeom &g, float xj, float yj, float t1, float t2, float t4, float t5, int i)
{
    const float fi = (float)i;
    const float x = (xj + t1 * fi) + t2;
    const float y = (yj + t4 * fi) + t5;
    return RawTap{(int)__builtin_roundf(x) - g.px, (int)__builtin_roundf(y) - g.py};
}
__device__ __forceinline__ bool in_core(const PlanGeom &g, RawTap t)
{
    return (unsigned)t.ix < (unsigned)g.W && (unsigned)t.iy < (unsigned)g.H;
}
// cell of (ix, iy), ix in [-1, W], iy in [-1, H], in the bordered (and, for class 0, column-mirrored) image
__device__ __forceinline__ int c_cell(const PlanGeom &g, int pitch, bool plus, RawTap t)
{
    return 1 + (t.iy + 1) * pitch + (plus ? t.ix : g.W - 1 - t.ix);
}


// per angle: bit 0 = mirror class (rotate_plan.hip: 1 = lanes walk the slice with column and row moving the same way; class 0
// is staged column-mirrored), bit 1 = sigma < 0 (a step moves the cell by sigma * (by * pitch - bx))
__device__ __forceinline__ int cplan_class_word(const float *t)
{
    const bool plus = (t[0] >= 0.0f) == (t[3] >= 0.0f);
    return (plus ? 1 : 0) | (t[4] < 0.0f ? 2 : 0);
}

// Pass 1 over ray (transform row t, detector bin j) through the H x W core of g: its live rows (one interval), and the cell
// it starts on -- its first tap, or the guard cell (0.0f) when it misses the core
struct RayScan {
    int n, first, start;
    bool bad;
};
__device__ __forceinline__ RayScan cplan_scan_ray(const PlanGeom &g, const float *t, int j, bool valid, int pitch)
{
    const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
    const bool plus = (t0 >= 0.0f) == (t3 >= 0.0f);
    const float xj = t0 * (float)j, yj = t3 * (float)j;
    int first = g.PH, last = -1, cnt = 0;
    if (valid)
        for (int i = 0; i < g.PH; ++i)
            if (in_core(g, raw_tap(g, xj, yj, t1, t2, t4, t5, i))) {
                first = min(first, i);
                last = i;
                ++cnt;
            }
    RayScan rs;
    rs.bad = cnt > 0 && cnt != last - first + 1;   // live rows must be one interval
    rs.n = cnt;
    rs.first = first;
    rs.start = cnt > 0 ? c_cell(g, pitch, plus, raw_tap(g, xj, yj, t1, t2, t4, t5, first)) : 0;
    return rs;
}
// Pass 2: the ray's step codes, three rows per byte, NQ chunks of 16 bytes written at codes[q * stride].  All rays of a task
// walk R rows: steps inside the core, then (if the ray is shorter than R) the step onto the zero border, then "stay".
// Returns true if some step does not fit the code (see rotate_cplan.hip).
__device__ __forceinline__ bool cplan_encode_ray(const PlanGeom &g, const float *t, int j, const RayScan &rs, int R, int pitch,
                                                 int NQ, uint4 *codes, size_t stride)
{
    const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
    const bool plus = (t0 >= 0.0f) == (t3 >= 0.0f);
    const int sigma = t4 < 0.0f ? -1 : 1;
    const float xj = t0 * (float)j, yj = t3 * (float)j;
    const int n = rs.n, first = rs.first;
    bool bad = rs.bad;
    RawTap cur{0, 0};
    int curcell = rs.start;
    if (n > 0) cur = raw_tap(g, xj, yj, t1, t2, t4, t5, first);
    for (int q = 0; q < NQ; ++q) {
        unsigned w[4] = {0u, 0u, 0u, 0u};
        for (int e = 0; e < kRowsPerChunk; ++e) {
            const int r = kRowsPerChunk * q + e;   // the step from row first + r to row first + r + 1
            const bool inner = r + 1 < n, leave = r + 1 == n && n < R;
            if (inner || leave) {
                const RawTap nxt = raw_tap(g, xj, yj, t1, t2, t4, t5, first + r + 1);   // (row PH: the same arithmetic)
                const int bx = nxt.ix != cur.ix, by = nxt.iy != cur.iy;
                if (leave && (in_core(g, nxt) || nxt.ix < -1 || nxt.ix > g.W || nxt.iy < -1 || nxt.iy > g.H)) {
                    bad = true;   // the ray cannot step onto the border
                } else {
                    const int nc = c_cell(g, pitch, plus, nxt);
                    if (nc - curcell != sigma * (by * pitch - bx)) bad = true;
                    w[e / 12] |= (unsigned)(bx | (by << 1)) << (8 * ((e % 12) / 3) + 2 * (e % 3));   // byte e / 3, step e % 3
                    cur = nxt;
                    curcell = nc;
                }
            }
        }
        codes[(size_t)q * stride] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return bad;
}

// The same codes as BYTES, for plans whose lanes walk more than one ray (round 4, paired tile tasks): code byte b of the ray --
// its steps 3 b .. 3 b + 2 -- goes to put(b, value).  All n codes of the ray are written: n - 1 steps inside the core and the
// step onto the zero border (the ray may ride in any task, before or behind another ray).  Returns true if a step does not fit.
template <class Put>
__device__ __forceinline__ bool cplan_encode_ray_bytes(const PlanGeom &g, const float *t, int j, const RayScan &rs, int pitch, Put put)
{
    const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
    const bool plus = (t0 >= 0.0f) == (t3 >= 0.0f);
    const int sigma = t4 < 0.0f ? -1 : 1;
    const float xj = t0 * (float)j, yj = t3 * (float)j;
    const int n = rs.n, first = rs.first;
    bool bad = rs.bad;
    if (n <= 0) return bad;
    RawTap cur = raw_tap(g, xj, yj, t1, t2, t4, t5, first);
    int curcell = rs.start;
    unsigned byte = 0;
    for (int r = 0; r < n; ++r) {   // the step from row first + r to row first + r + 1
        const RawTap nxt = raw_tap(g, xj, yj, t1, t2, t4, t5, first + r + 1);
        const int bx = nxt.ix != cur.ix, by = nxt.iy != cur.iy;
        const bool leave = r + 1 == n;
        if (leave && (in_core(g, nxt) || nxt.ix < -1 || nxt.ix > g.W || nxt.iy < -1 || nxt.iy > g.H)) {
            bad = true;   // the ray cannot step onto the border
        } else {
            const int nc = c_cell(g, pitch, plus, nxt);
            if (nc - curcell != sigma * (by * pitch - bx)) bad = true;
            byte |= (unsigned)(bx | (by << 1)) << (2 * (r % 3));
            cur = nxt;
            curcell = nc;
        }
        if (r % 3 == 2 || leave) {
            put(r / 3, byte);
            byte = 0;
        }
    }
    return bad;
}

// In-kernel set-up shared by the kernels that walk compact plans: the replicated step table at LDS offset 0 (entry e = the
// cumulative byte offsets after 1..3 of the steps coded in e -- bit 2m: column, 2m + 1: row --, one copy per lane of a
// half-wave at byte e * 256 + l * 8) and the zero border of an H x W image of NS interleaved slices (guard + row -1, row H +
// guard, the gutter columns W .. pitch - 1 of every row).
template <int NS> __device__ __forceinline__ void cplan_init_lut(float *lds, int pitch)
{
    for (int t = threadIdx.x; t < 64 * 32; t += blockDim.x) {
        const int e = t >> 5;
        int acc = 0;
        unsigned short cum[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            acc += (((e >> (2 * m + 1)) & 1) * pitch - ((e >> (2 * m)) & 1)) * (4 * NS);
            cum[m] = (unsigned short)(short)acc;
        }
        reinterpret_cast<uint2 *>(lds)[t] = make_uint2(cum[0] | ((unsigned)cum[1] << 16), cum[2]);
    }
}
template <int NS> __device__ __forceinline__ void cplan_zero_border(float *image, int H, int W, int pitch)
{
    const int nthreads = blockDim.x, gut = pitch - W;
    for (int t = threadIdx.x; t < (pitch + 1) * NS; t += nthreads) image[t] = 0.0f;
    for (int t = threadIdx.x; t < (pitch + 1) * NS; t += nthreads) image[(size_t)(1 + (H + 1) * pitch) * NS + t] = 0.0f;
    for (int t = threadIdx.x; t < H * gut * NS; t += nthreads) {
        const int row = t / (gut * NS), k = t - row * (gut * NS);
        image[(size_t)(1 + (row + 1) * pitch + W) * NS + k] = 0.0f;
    }
}

// ---- walking -------------------------------------------------------------------------------------------------------------
// table address of code byte BYTE of `w` for this lane: byte * 256 + (lane & 31) * 8.  `la` holds (lane & 31) * 8 in its
// byte 0; one SDWA move writes the code into byte 1 and preserves the rest.
template <int BYTE> __device__ __forceinline__ void lut_addr(int &la, unsigned w)
{
    if constexpr (BYTE == 0)
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(la) : "v"(w));
    else if constexpr (BYTE == 1)
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1" : "+v"(la) : "v"(w));
    else if constexpr (BYTE == 2)
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(la) : "v"(w));
    else
        asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(la) : "v"(w));
}
// base +- the sign-extended 16-bit half HALF of `pk`: one SDWA op
template <bool NEG, int HALF> __device__ __forceinline__ int step16(int base, unsigned pk)
{
    int r;
    if constexpr (!NEG && HALF == 0)
        asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(base), "v"(pk));
    else if constexpr (!NEG && HALF == 1)
        asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(base), "v"(pk));
    else if constexpr (NEG && HALF == 0)
        asm("v_sub_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(base), "v"(pk));
    else
        asm("v_sub_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(base), "v"(pk));
    return r;
}
// group GG (six rows = two code bytes) of the chunk pair (c0: groups 0..7, c1: groups 8..15): its code dword and byte pair
template <int GG> __device__ __forceinline__ unsigned group_dword(const uint4 &c0, const uint4 &c1)
{
    constexpr int d = GG >> 1;
    if constexpr (d == 0) return c0.x;
    else if constexpr (d == 1) return c0.y;
    else if constexpr (d == 2) return c0.z;
    else if constexpr (d == 3) return c0.w;
    else if constexpr (d == 4) return c1.x;
    else if constexpr (d == 5) return c1.y;
    else if constexpr (d == 6) return c1.z;
    else return c1.w;
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct LutPair {
    u32x2 e0, e1;   // the table entries of a group's two code bytes: (c1 | c2 << 16, c3) each
};
// LDS is addressed by absolute 32-bit byte addresses here: the kernel has no static LDS, so its dynamic LDS -- the table
// first -- starts at address 0 (checked once per workgroup), and a table address is just code byte * 8.
typedef const __attribute__((address_space(3))) u32x2 *lds_u2_cptr;
template <int GG> __device__ __forceinline__ LutPair lut_issue(int &la0, int &la1, const uint4 &c0, const uint4 &c1)
{
    const unsigned w = group_dword<GG>(c0, c1);
    constexpr int b = (GG & 1) * 2;
    LutPair p;
    lut_addr<b>(la0, w);
    p.e0 = *(lds_u2_cptr)(size_t)(unsigned)la0;
    lut_addr<b + 1>(la1, w);
    p.e1 = *(lds_u2_cptr)(size_t)(unsigned)la1;
    return p;
}
// the six tap addresses of a group from the ray's running address and the group's two table entries
template <bool NEG> __device__ __forceinline__ void group_addr(int &adr, const LutPair &p, int (&a)[6])
{
    a[0] = adr;
    a[1] = step16<NEG, 0>(adr, p.e0.x);
    a[2] = step16<NEG, 1>(adr, p.e0.x);
    const int mid = step16<NEG, 0>(adr, p.e0.y);
    a[3] = mid;
    a[4] = step16<NEG, 0>(mid, p.e1.x);
    a[5] = step16<NEG, 1>(mid, p.e1.x);
    adr = step16<NEG, 0>(mid, p.e1.y);
}
template <int NS> __device__ __forceinline__ void group_gather(const int (&a)[6], typename SliceVec<NS>::type (&v)[6])
{
    typedef const __attribute__((address_space(3))) typename SliceVec<NS>::type *lds_vec_cptr;
#pragma unroll
    for (int e = 0; e < 6; ++e) v[e] = *(lds_vec_cptr)(size_t)(unsigned)a[e];   // a[e]: absolute LDS byte address
}

// One ray-sum per lane: walks ng groups of six rows from byte address adr0.  c0 / c1: the ray's code chunks 0 and 1 (48 rows
// each; chunks behind the plan's NQ-th are zeros: stay); chunk q + 2 is loaded from pc (chunk 2 on) while chunk q is walked.
// The table entries of group n + 3 and the gathers of group n + 1 are in flight while group n is added.
//
// PAIRED (round 4): a lane walks TWO rays back to back -- ray A for its own gsw groups, then ray B, whose first tap sits at
// byte address adrB and whose codes follow A's in the lane's stream (from code byte 2 gsw on).  At group gsw the running
// sum becomes *accA and restarts from zero; lanes switch at their own group, so the per-lane tests sit behind a wave-uniform
// "does any lane switch here" (one compare and a scalar branch per group when none does).  gsw = 0: the lane starts with B
// (the caller passes adr = adrB); gsw >= ng: the lane never switches and the returned sum is ray A's.
template <int NS, bool NEG, bool PAIRED = false>
__device__ __forceinline__ typename SliceVec<NS>::type cwalk(int adr, int ng, int lane8, uint4 c0, uint4 c1, const uint4 *pc,
                                                              size_t st, int NQ, int gsw = 0, int adrB = 0,
                                                              typename SliceVec<NS>::type *accA = nullptr)
{
    typedef typename SliceVec<NS>::type vec_t;
    vec_t acc = 0.0f;
    vec_t va[6], vb[6];
    int an[6], nlast = 0;
    int la0 = lane8, la1 = lane8;   // table address registers: byte 0 = (lane & 31) * 8, byte 1 = the code
    LutPair l0 = lut_issue<0>(la0, la1, c0, c1), l1 = lut_issue<1>(la0, la1, c0, c1);
    const LutPair l2 = lut_issue<2>(la0, la1, c0, c1);
    __builtin_amdgcn_sched_barrier(0);   // all six table reads of the first three groups in flight together
    group_addr<NEG>(adr, l0, an);
    group_gather<NS>(an, va);
    if constexpr (PAIRED)
        if (gsw == 1) adr = adrB;
    group_addr<NEG>(adr, l1, an);
    l0 = l2;
    int n = 0;   // first group of the current chunk
#define CTPVAE_CSWITCH_ACC(M)   /* ray A ends before group M: its sum leaves the accumulator */                 \
        if constexpr (PAIRED)                                                                                  \
            if (__builtin_amdgcn_ballot_w64(gsw == (M)) != 0ull)                                               \
                if (gsw == (M)) {                                                                              \
                    *accA = acc;                                                                               \
                    acc = 0.0f;                                                                                \
                }
#define CTPVAE_CSWITCH_ADR(M)   /* group M is ray B's first: its addresses start from adrB */                    \
        if constexpr (PAIRED)                                                                                  \
            if (__builtin_amdgcn_ballot_w64(gsw == (M)) != 0ull)                                               \
                if (gsw == (M)) adr = adrB;
    for (int q = 0;; ++q) {
        const bool more = q + 2 < NQ && n + 8 < ng;   // wave-uniform: chunk q + 2 exists and may be walked
        uint4 c2;
        if (more) c2 = pc[(size_t)q * st];
#define CTPVAE_CSTEP(G, VCUR, VNXT, LNEW, LUSE)                                                    \
        if (n + G + 1 >= ng) {   /* the task's last group: nothing more is issued (round 4: the gathers of a group behind */ \
            nlast = n + G;   /* the last were a tenth of the tile kernel's LDS cycles); it is added behind the loop, */ \
            break;           /* out of va (even G) or vb (odd G)                                                      */ \
        }                                                                                          \
        group_gather<NS>(an, VNXT);                        /* group n + G + 1 */             \
        LNEW = lut_issue<G + 3>(la0, la1, c0, c1);        /* table entries of group n + G + 3 */ \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        CTPVAE_CSWITCH_ACC(n + G)                                                                  \
        _Pragma("unroll") for (int e = 0; e < 6; ++e) acc += VCUR[e];   /* group n + G */          \
        CTPVAE_CSWITCH_ADR(n + G + 2)                                                              \
        group_addr<NEG>(adr, LUSE, an);                          /* addresses of group n + G + 2 */
        CTPVAE_CSTEP(0, va, vb, l1, l0)
        CTPVAE_CSTEP(1, vb, va, l0, l1)
        CTPVAE_CSTEP(2, va, vb, l1, l0)
        CTPVAE_CSTEP(3, vb, va, l0, l1)
        CTPVAE_CSTEP(4, va, vb, l1, l0)
        CTPVAE_CSTEP(5, vb, va, l0, l1)
        CTPVAE_CSTEP(6, va, vb, l1, l0)
        CTPVAE_CSTEP(7, vb, va, l0, l1)
#undef CTPVAE_CSTEP
        n += 8;
        c0 = c1;
        c1 = more ? c2 : uint4{0u, 0u, 0u, 0u};
    }
    // The last group, gathered one step ago.  (Added inside the loop's exits instead, these adds were hoisted by hipcc above the
    // branches -- "both paths begin with them" -- and with that above the steady path's gathers: a wave then waited for a group
    // before it issued the next, and small launches lost what large ones gained.)
    CTPVAE_CSWITCH_ACC(nlast)
    if (nlast & 1) {   // (wave-uniform; no copy of the group into a third buffer: the four-slice kernels have no registers for one)
#pragma unroll
        for (int e = 0; e < 6; ++e) acc += vb[e];
    } else {
#pragma unroll
        for (int e = 0; e < 6; ++e) acc += va[e];
    }
#undef CTPVAE_CSWITCH_ACC
#undef CTPVAE_CSWITCH_ADR
    return acc;
}


}  // namespace ctpvae
