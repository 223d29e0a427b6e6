// rotate_dev.h -- device helpers and geometry structs shared by the direct rotate kernels (rotate.hip: nearest direct / tiled
// kernels, segment backward; rotate_bilin.hip: the bilinear forward, backward and exact adjoint).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>

#include "common.h"
#include "loglik_math.h"
#include "rotate_plan.h"

namespace ctpvae {

struct RotGeom {
    int S, H, W, PH, PW, py, px, A;
};

__device__ __forceinline__ float round_half_away(float v) { return __builtin_roundf(v); }

__device__ __forceinline__ int cvt_rpi(float v)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ int cvt_flr(float v)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ int med3i(int v, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) float *lds_cptr;
typedef __attribute__((address_space(3))) float *lds_ptr;
typedef const __attribute__((address_space(1))) float *glb_cptr;

// smallest pitch >= wb with pitch == +1 (mod 32) if want_plus else == -1 (mod 32)
__host__ __device__ __forceinline__ int pitch_for(int wb, bool want_plus)
{
    const int r = want_plus ? 1 : 31;
    return wb + ((r - (wb & 31)) & 31);
}

// an opaque copy in a VGPR: keeps loop-invariant operands of the asm helpers out of the loop body
__device__ __forceinline__ int pin_vgpr(int v)
{
    int r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(v));
    return r;
}

__device__ __forceinline__ float lds_abs(int byte_addr) { return *(lds_cptr)(uintptr_t)(unsigned)byte_addr; }
// NS interleaved slices per LDS pixel: one ds_read_b32 / _b64 / _b128 per tap
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NS> struct PixVec { typedef float type; };
template <> struct PixVec<2> { typedef f32x2 type; };
template <> struct PixVec<4> { typedef f32x4 type; };
template <int NS> __device__ __forceinline__ typename PixVec<NS>::type lds_abs_vec(int byte_addr)
{
    typedef const __attribute__((address_space(3))) typename PixVec<NS>::type *vptr;
    return *(vptr)(uintptr_t)(unsigned)byte_addr;
}

// max over the 64 lanes of a wave of a non-negative int, as an SGPR value (DPP row shifts + row broadcasts)
__device__ __forceinline__ int wave_max_nonneg(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}

// rows [lo, hi) of the canvas on which base + slope*i may fall inside [L, U]
__device__ __forceinline__ void clip_rows(float base, float slope, float L, float U, float &lo, float &hi)
{
    if (fabsf(slope) < 1e-6f) {
        // |slope * i| < 1e-6 * 2^24: treat as constant, with a margin far above that drift
        if (base < L - 1.0f || base > U + 1.0f) hi = -1.0f;
    } else {
        const float inv = __builtin_amdgcn_rcpf(slope);   // 1 ulp is ample: the range is widened by whole rows
        const float i1 = (L - base) * inv, i2 = (U - base) * inv;
        lo = fmaxf(lo, fminf(i1, i2));
        hi = fminf(hi, fmaxf(i1, i2));
    }
}

struct TileSpec {
    int ntx, nty;   // tiles per slice
    int tw, th;     // nominal tile size (edge tiles are smaller)
    int nb;         // ray slots per (tile, angle), a multiple of 64
    int span;       // ... of which the first `span` can touch the tile (the rest pad nb to whole waves): only those are stored
                    // by the tile kernels and read back by the reduce pass
    float radius;   // half the tile diagonal + 3 px
};
__host__ __device__ __forceinline__ void tile_rect(const RotGeom &g, const TileSpec &ts, int t, int &y0, int &x0, int &h, int &w)
{
    const int ty = t / ts.ntx, tx = t - ty * ts.ntx;
    y0 = ty * ts.th;
    x0 = tx * ts.tw;
    h = min(ts.th, g.H - y0);
    w = min(ts.tw, g.W - x0);
}
// first ray slot's bin: the orthonormal transform maps canvas (x, y) to bin t0*(x - t2) + t3*(y - t5)
__device__ __forceinline__ int tile_first_bin(const float *t, float cx, float cy, float radius)
{
    const float jc = t[0] * (cx - t[2]) + t[3] * (cy - t[5]);
    return (int)floorf(jc - radius);
}
// The workspace of partial sums: [slice / 4][tile][angle][slot][slice % 4] -- the four slices a tile workgroup walks together
// leave it as ONE 16-byte store per ray and reach the reduce pass as one 16-byte load (per-slice planes cost four 4-byte
// accesses each way).  Sized for a whole number of slice quads (ctpvae_rotate_fwd_tiled_workspace_bytes).
constexpr int kPartialQuad = 4;
__device__ __forceinline__ size_t partial_index(int s, int nt, int t, size_t nrays, size_t ray)
{
    return ((((size_t)(s >> 2) * nt + t) * nrays + ray) << 2) + (size_t)(s & 3);
}

template <class F>
static inline int for_slice_chunks(int S, int chunk, F launch_chunk)
{
    for (int s0 = 0; s0 < S; s0 += chunk)
        if (int rc = launch_chunk(s0, std::min(chunk, S - s0))) return rc;
    return CTPVAE_OK;
}

// ---- host functions shared between rotate.hip and rotate_bilin.hip ---------------------------------------------------------
// tiling of a slice that does not fit LDS whole (ntx == 0: not tiled): a function of (H, W, interp) alone -- the shape fixes the
// association of the fp32 row sum (ctpvae_rotate_tile_shape)
TileSpec pick_tiles(int H, int W, int interp);
// the reduce pass over the tiles' partial sums (+ the log-likelihood epilogue): rotate.hip
int launch_tile_reduce(const float *workspace_dev, const RotGeom &g, const TileSpec &ts, const float *T8_dev, float *sino_dev,
                       const LogLikEpilogue &epi, ctpvae_stream_t stream);
// bilinear forward (rotate_bilin.hip): whole slices in LDS / tiles into the partial-sum workspace
constexpr size_t kBilinLdsReserve = 16 * 1024;   // LDS kept free of the image for the transform copy and class list (A <= ~450)
bool bilin_fwd_whole_geometry(int H, int W);          // (H, W) is projected whole (else: tiles)
bool bilin_fwd_whole_ok(int H, int W, int A);         // ... and this many angles fit beside the image
bool bilin_fwd_tiles_ok(const TileSpec &ts, int A);   // a tile and this many angles' tables fit LDS
int bilin_fwd_whole(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px, const float *T8_dev, int A,
                    float *sino_dev, ctpvae_stream_t stream);
int bilin_fwd_tiles(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px, const float *T8_dev, int A,
                    const TileSpec &ts, float *workspace_dev, ctpvae_stream_t stream);
// bilinear TensorFlow-compatible backward (rotate_bilin.hip): cotangent segments, slices interleaved per cell
int bilin_bwd_tfcompat(const float *gsino_dev, int S, int A, int PH, int PW, const float *Tinv8_dev, int H, int W, int py, int px,
                       float *gimg_dev, ctpvae_stream_t stream);

}  // namespace ctpvae
