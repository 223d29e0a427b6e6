// lds_stage.h -- staging rows of fp32 from global memory into LDS, shared by the rotate kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace ctpvae {

// rows x cols floats from global (row stride src_stride) into LDS (row stride pitch == +-1 mod 32), optionally
// column-mirrored.  16-byte global loads, and a lane arrangement that makes the four ds_write_b32 of a float4
// conflict-free: inside a 32-lane group, lane l takes row (l >> 3) and float4 number (l & 7) of a 4-row x 32-column
// block (8 consecutive lanes read one 128-B line), so for component e the group writes dword (k*pitch + 4m + e), k = 0..3, m = 0..7 -- and with pitch == 1
// (mod 32) the bank k + 4m + e runs over all 32 banks exactly once.  The two halves of a wave take adjacent blocks.
// (LDS-DMA was measured here and lost: its dword form costs ~47 cycles per 256-B instruction per CU, and the odd
// pitch rules out its 16-byte form.  Plain dword loads also lost: 4x the requests of float4 loads.)
// Requires cols % 4 == 0 and 16-byte aligned rows; stage_rows_scalar covers everything else.
__device__ __forceinline__ void stage_rows_v4(float *lds, const float *__restrict__ src, int rows, int cols, int src_stride,
                                              int pitch, bool mirror, int lane, int wave, int nwaves, int es = 1)
{   // es: element stride in dwords (2 when two slices are interleaved as float2; `lds` then points at the slice's lane)
    const int ncb = (cols + 31) >> 5, npc = (ncb + 1) >> 1, nrq = (rows + 3) >> 2;
    const int npairs = nrq * npc;
    const int h = lane >> 5, k = (lane & 31) >> 3, m = lane & 7;   // 8 consecutive lanes = one 128-B line of one row
    constexpr int NB = 8;   // float4 loads in flight per lane: 64 KiB lands in one batch with >= 8 waves
    // pair index pp = wave, wave + nwaves, ... -> (rq, pc) = (pp / npc, pp % npc), advanced without dividing
    const int d_rq = nwaves / npc, d_pc = nwaves - d_rq * npc;
    int rq = wave / npc, pc = wave - rq * npc;
    for (int p0 = wave; p0 < npairs; p0 += NB * nwaves) {
        float4 v[NB];
        int r_[NB], c_[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            r_[u] = 4 * rq + k;
            c_[u] = 32 * (2 * pc + h) + 4 * m;
            const bool ok = rq < nrq && r_[u] < rows && c_[u] < cols;
            // The load itself is UNCONDITIONAL (address clamped into the slice): a select between a load and a
            // constant makes hipcc branch around every load and wait vmcnt(0) after each.
            const int rl = min(r_[u], rows - 1), cl = min(c_[u], cols - 4);
            v[u] = *reinterpret_cast<const float4 *>(src + (size_t)rl * src_stride + (mirror ? cols - 4 - cl : cl));
            if (!ok) r_[u] = -1;
            rq += d_rq;
            pc += d_pc;
            if (pc >= npc) {
                pc -= npc;
                ++rq;
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (r_[u] < 0) continue;
            float *d = lds + (r_[u] * pitch + c_[u]) * es;
            d[0] = mirror ? v[u].w : v[u].x;
            d[es] = mirror ? v[u].z : v[u].y;
            d[2 * es] = mirror ? v[u].y : v[u].z;
            d[3 * es] = mirror ? v[u].x : v[u].w;
        }
    }
}
__device__ __forceinline__ void stage_rows_scalar(float *lds, const float *__restrict__ src, int rows, int cols,
                                                  int src_stride, int pitch, bool mirror, int tid, int nthreads, int es = 1)
{
    for (int p = tid; p < rows * cols; p += nthreads) {
        const int r = p / cols, c = p - r * cols;
        lds[(r * pitch + c) * es] = src[(size_t)r * src_stride + (mirror ? cols - 1 - c : c)];
    }
}
// NS slices interleaved per LDS pixel (pixel (r, c) = NS consecutive floats, one per slice): the NS float4 loads of a
// 4-pixel unit are issued together -- one load round trip for all slices, not one per slice -- and written as four
// NS-wide vectors (ds_write_b64 / _b128), a quarter of the LDS instructions of per-slice dword writes.
// src[n] are the slices' first rows (same stride), optionally column-mirrored; needs cols % 4 == 0 and 16-byte aligned rows like stage_rows_v4.
template <int NS>
__device__ __forceinline__ void stage_rows_interleaved_v4(float *lds, const float *const (&src)[NS], int rows, int cols,
                                                          int src_stride, int pitch, bool mirror, int lane, int wave, int nwaves)
{
    typedef float vec_t __attribute__((ext_vector_type(NS)));
    const int ncb = (cols + 31) >> 5, npc = (ncb + 1) >> 1, nrq = (rows + 3) >> 2;
    const int npairs = nrq * npc;
    const int h = lane >> 5, k = (lane & 31) >> 3, m = lane & 7;   // 8 consecutive lanes = one 128-B line of one row
    constexpr int NB = 8 / NS;   // units in flight per lane: 8 float4 loads, as in stage_rows_v4
    const int d_rq = nwaves / npc, d_pc = nwaves - d_rq * npc;
    int rq = wave / npc, pc = wave - rq * npc;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // (the units a wave has left to stage: a scalar count)
    for (int p0 = wave; p0 < npairs; p0 += NB * nwaves) {
        float4 v[NB][NS];
        int r_[NB], c_[NB];
        const int left = npairs - (wave_u + (p0 - wave));        // units of this wave from p0 on: 1 + (left - 1) / nwaves of them
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            r_[u] = 4 * rq + k;
            c_[u] = 32 * (2 * pc + h) + 4 * m;
            const bool ok = rq < nrq && r_[u] < rows && c_[u] < cols;
            const int rl = min(r_[u], rows - 1), cl = min(c_[u], cols - 4);   // per lane: unconditional, clamped loads
            // ... but whole units past the block are not requested at all (wave-uniform: a small block -- 20 detector rows of a
            // short launch -- is one unit per wave, and the other NB - 1 were loads of the last row over and over)
            if (u * nwaves < left) {
#pragma unroll
                for (int n = 0; n < NS; ++n)
                    v[u][n] = *reinterpret_cast<const float4 *>(src[n] + (size_t)rl * src_stride + (mirror ? cols - 4 - cl : cl));
            } else {
#pragma unroll
                for (int n = 0; n < NS; ++n) v[u][n] = float4{0.0f, 0.0f, 0.0f, 0.0f};
            }
            if (!ok) r_[u] = -1;
            rq += d_rq;
            pc += d_pc;
            if (pc >= npc) {
                pc -= npc;
                ++rq;
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (r_[u] < 0) continue;
            vec_t *d = reinterpret_cast<vec_t *>(lds) + (r_[u] * pitch + c_[u]);
            vec_t w0, w1, w2, w3;
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                w0[n] = mirror ? v[u][n].w : v[u][n].x;
                w1[n] = mirror ? v[u][n].z : v[u][n].y;
                w2[n] = mirror ? v[u][n].y : v[u][n].z;
                w3[n] = mirror ? v[u][n].x : v[u][n].w;
            }
            d[0] = w0;
            d[1] = w1;
            d[2] = w2;
            d[3] = w3;
        }
    }
}
// The same staging in two phases, for software pipelining: issue() requests up to NB units per lane into registers,
// commit() writes them to LDS later (after the consumers of the buffer's previous contents have passed a barrier).
// fits(): the whole block must be covered by NB units per lane, and meet stage_rows_interleaved_v4's alignment rules.
template <int NS, int NB>
struct StagedRows {
    float4 v[NB][NS];
    int r_[NB], c_[NB];
    __device__ __forceinline__ static bool fits(const float *const (&src)[NS], int rows, int cols, int src_stride, int nwaves)
    {
        const int ncb = (cols + 31) >> 5, npc = (ncb + 1) >> 1, nrq = (rows + 3) >> 2;
        bool ok = nrq * npc <= NB * nwaves && (cols & 3) == 0 && (src_stride & 3) == 0;
#pragma unroll
        for (int n = 0; n < NS; ++n) ok = ok && (reinterpret_cast<uintptr_t>(src[n]) & 15) == 0;
        return ok;
    }
    __device__ __forceinline__ void issue(const float *const (&src)[NS], int rows, int cols, int src_stride, int lane, int wave,
                                          int nwaves)
    {
        const int ncb = (cols + 31) >> 5, npc = (ncb + 1) >> 1, nrq = (rows + 3) >> 2;
        const int h = lane >> 5, k = (lane & 31) >> 3, m = lane & 7;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int pp = wave + u * nwaves, rq = pp / npc, pc = pp - rq * npc;
            r_[u] = 4 * rq + k;
            c_[u] = 32 * (2 * pc + h) + 4 * m;
            const bool ok = rq < nrq && r_[u] < rows && c_[u] < cols;
            const int rl = min(r_[u], rows - 1), cl = min(c_[u], cols - 4);   // unconditional, clamped loads
#pragma unroll
            for (int n = 0; n < NS; ++n) v[u][n] = *reinterpret_cast<const float4 *>(src[n] + (size_t)rl * src_stride + cl);
            if (!ok) r_[u] = -1;
        }
    }
    __device__ __forceinline__ void commit(float *lds, int pitch) const
    {
        typedef float vec_t __attribute__((ext_vector_type(NS)));
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (r_[u] < 0) continue;
            vec_t *d = reinterpret_cast<vec_t *>(lds) + (r_[u] * pitch + c_[u]);
            vec_t w0, w1, w2, w3;
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                w0[n] = v[u][n].x;
                w1[n] = v[u][n].y;
                w2[n] = v[u][n].z;
                w3[n] = v[u][n].w;
            }
            d[0] = w0;
            d[1] = w1;
            d[2] = w2;
            d[3] = w3;
        }
    }
};

// A whole unit whose shape makes every index a shift: rows % 4 == 0, cols == 64 << cs, covered by NB blocks of 4 rows x 64
// columns per wave (the host checks, and that the rows are 16-byte aligned).  `wave` must be WAVE-UNIFORM: block i of the wave is
// pp = wave + i * nwaves -- scalar arithmetic --, a lane adds ONE offset of its own, the mirrored form is its own copy of the
// code (no per-value selects).  ~150 instructions per wave where the general form above has ~270: the 16 waves of a workgroup
// share 4 SIMDs, and the launch's prologue is bound by instructions issued per SIMD (tools/stamp_rounds.hip).
// between(): called once, between the requests and the writes (work whose own loads should fly beside the rows).
template <int NS, int NB, bool MIRROR, class F>
__device__ __forceinline__ void stage_unit_pow2_m(float *lds, const float *const (&src)[NS], int rows, int cs, int src_stride, int pitch,
                                                  int lane, int wave, int nwaves, F between)
{
    const int h = lane >> 5, k = (lane & 31) >> 3, m = lane & 7, cols = 64 << cs, nrq = rows >> 2;
    const int lo_src = k * src_stride + (MIRROR ? -(32 * h + 4 * m) : 32 * h + 4 * m);
    const int lo_dst = k * pitch + 32 * h + 4 * m;
    float4 v[NB][NS];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pp = wave + i * nwaves, rq = pp >> cs, pc = pp & ((1 << cs) - 1);
        const int so = rq < nrq ? 4 * rq * src_stride + (MIRROR ? cols - 4 - 64 * pc : 64 * pc) : (MIRROR ? cols - 4 : 0);   // (past the unit: block 0, dropped)
#pragma unroll
        for (int n = 0; n < NS; ++n) v[i][n] = *reinterpret_cast<const float4 *>(src[n] + so + lo_src);
    }
    between();
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pp = wave + i * nwaves, rq = pp >> cs, pc = pp & ((1 << cs) - 1);
        if (rq >= nrq) continue;   // wave-uniform
        const int cell = 4 * rq * pitch + 64 * pc + lo_dst;
        if constexpr (NS == 1) {
            float *d = lds + cell;
            d[0] = MIRROR ? v[i][0].w : v[i][0].x;
            d[1] = MIRROR ? v[i][0].z : v[i][0].y;
            d[2] = MIRROR ? v[i][0].y : v[i][0].z;
            d[3] = MIRROR ? v[i][0].x : v[i][0].w;
        } else {
            typedef float vec_t __attribute__((ext_vector_type(NS)));
            vec_t *d = reinterpret_cast<vec_t *>(lds) + cell;
            vec_t w0, w1, w2, w3;
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                w0[n] = MIRROR ? v[i][n].w : v[i][n].x;
                w1[n] = MIRROR ? v[i][n].z : v[i][n].y;
                w2[n] = MIRROR ? v[i][n].y : v[i][n].z;
                w3[n] = MIRROR ? v[i][n].x : v[i][n].w;
            }
            d[0] = w0;
            d[1] = w1;
            d[2] = w2;
            d[3] = w3;
        }
    }
}
// (mirror is wave-uniform: two copies of the code, no per-value selects)
template <int NS, int NB, class F>
__device__ __forceinline__ void stage_unit_pow2(float *lds, const float *const (&src)[NS], int rows, int cs, int src_stride, int pitch,
                                                bool mirror, int lane, int wave, int nwaves, F between)
{
    if (mirror) stage_unit_pow2_m<NS, NB, true>(lds, src, rows, cs, src_stride, pitch, lane, wave, nwaves, between);
    else stage_unit_pow2_m<NS, NB, false>(lds, src, rows, cs, src_stride, pitch, lane, wave, nwaves, between);
}

// log2 of a unit's 64-column blocks per row if the lean form serves it (wave-uniform), else -1
__device__ __forceinline__ int unit_pow2_cs(int rows, int cols, int src_stride, int nwaves, int nb)
{
#ifdef CTPVAE_TUNE_NO_LEAN_STAGE
    return -1;
#else
    if ((rows & 3) != 0 || (src_stride & 3) != 0 || (cols != 64 && cols != 128 && cols != 256)) return -1;
    const int cs = cols == 64 ? 0 : (cols == 128 ? 1 : 2);
    return ((rows >> 2) << cs) <= nb * nwaves ? cs : -1;
#endif
}
__device__ __forceinline__ void stage_rows(float *lds, const float *__restrict__ src, int rows, int cols, int src_stride,
                                           int pitch, bool mirror, int lane, int wave, int nwaves, int es = 1)
{
    if ((cols & 3) == 0 && (src_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0)
        stage_rows_v4(lds, src, rows, cols, src_stride, pitch, mirror, lane, wave, nwaves, es);
    else
        stage_rows_scalar(lds, src, rows, cols, src_stride, pitch, mirror, wave * 64 + lane, nwaves * 64, es);
}

template <int NS>
__device__ __forceinline__ void stage_rows_interleaved(float *lds, const float *const (&src)[NS], int rows, int cols,
                                                       int src_stride, int pitch, bool mirror, int lane, int wave, int nwaves)
{
    bool aligned = (cols & 3) == 0 && (src_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(lds) & (4 * NS - 1)) == 0;
#pragma unroll
    for (int n = 0; n < NS; ++n) aligned = aligned && (reinterpret_cast<uintptr_t>(src[n]) & 15) == 0;
    if (aligned) {
        stage_rows_interleaved_v4<NS>(lds, src, rows, cols, src_stride, pitch, mirror, lane, wave, nwaves);
    } else {
#pragma unroll
        for (int n = 0; n < NS; ++n)
            stage_rows_scalar(lds + n, src[n], rows, cols, src_stride, pitch, mirror, wave * 64 + lane, nwaves * 64, NS);
    }
}

// A UNIT (a slice, a slice pair / quad, a tile) into LDS: the lean form where its shape allows (64 / 128 / 256 columns, rows in
// fours, one batch of loads per lane, aligned rows), the general forms above otherwise.  For the forward kernels, whose units are
// square slices and 64-column tiles; kernels that stage detector rows (184 or 728 bins) call the general forms directly -- with
// the lean form compiled in beside them the planned backward ran 4 % slower (a longer prologue, another register allocation).
template <int NS>
__device__ __forceinline__ void stage_unit(float *lds, const float *const (&src)[NS], int rows, int cols, int src_stride, int pitch,
                                           bool mirror, int lane, int wave, int nwaves)
{
    bool aligned = (reinterpret_cast<uintptr_t>(lds) & (4 * NS - 1)) == 0;
#pragma unroll
    for (int n = 0; n < NS; ++n) aligned = aligned && (reinterpret_cast<uintptr_t>(src[n]) & 15) == 0;
    const int cs = aligned ? unit_pow2_cs(rows, cols, src_stride, nwaves, 8 / NS) : -1;
    if (cs >= 0) {
        stage_unit_pow2<NS, 8 / NS>(lds, src, rows, cs, src_stride, pitch, mirror, lane, __builtin_amdgcn_readfirstlane(wave), nwaves, [] {});
    } else if constexpr (NS == 1) {
        stage_rows(lds, src[0], rows, cols, src_stride, pitch, mirror, lane, wave, nwaves);
    } else {
        stage_rows_interleaved<NS>(lds, src, rows, cols, src_stride, pitch, mirror, lane, wave, nwaves);
    }
}

}  // namespace ctpvae
