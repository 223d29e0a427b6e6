// rotate_plan.h -- geometry and lane helpers shared by the gather-plan kernels (rotate_plan.hip: u16 tap plans;
// rotate_cplan.hip: compact step-coded plans).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace ctpvae {

struct PlanGeom {
    int H, W, PH, PW, py, px, A;
};

__host__ __device__ inline int pitch_mod32_is_1(int w) { return w + ((1 - (w & 31)) & 31); }

// Detector bins are dealt to 64-lane blocks in two 32-bin bands mirrored about the detector centre: block k holds
// bins [c - 32(k+1), c - 32k) in lanes 0..31 and [c + 32k, c + 32(k+1)) in lanes 32..63, c = PW / 2.  Rays at equal
// distance from the centre have equal chords through the slice, so a wave's 64 rays need nearly the same rows
// (a contiguous 64-bin block mixes long central chords with short outer ones: 147 vs 134 visited rows per task at
// N = 128).  Each 32-lane half is still a contiguous run of bins, which is what the LDS bank argument needs.
__host__ __device__ inline int lane_to_bin(int PW, int jb, int lane)
{
    const int c = PW >> 1;
    return lane < 32 ? c - 32 * (jb + 1) + lane : c + 32 * jb + (lane - 32);   // may fall outside [0, PW): dead lane
}
__host__ __device__ inline int num_bin_blocks(int PW) { return (PW - (PW >> 1) + 31) / 32; }

typedef const __attribute__((address_space(3))) float *lds_cptr;
typedef __attribute__((address_space(3))) float *lds_ptr;
typedef const __attribute__((address_space(1))) float *glb_cptr;

__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

// NS slices interleaved in LDS (float or float2 per pixel): byte offset = index * 4 * NS, one ds_read_b32 / _b64 per tap
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NS> struct SliceVec { typedef float type; };
template <> struct SliceVec<2> { typedef f32x2 type; };
// An angle subset handed over in HOST memory travels in the kernel arguments (<= 256 plan angles, 512 B): no upload, no
// device-side copy to wait for -- the kernels read it like any other argument.
constexpr int kMaxSelAngles = 256;
struct SelHost {
    unsigned w[kMaxSelAngles / 2];   // two 16-bit plan angles per dword: a wave-uniform index reads them with SCALAR loads
    __host__ __device__ void set(int k, int a) { w[k >> 1] = (w[k >> 1] & ~(0xffffu << (16 * (k & 1)))) | ((unsigned)a << (16 * (k & 1))); }
    __device__ __forceinline__ int get(int k) const { return (int)((w[k >> 1] >> (16 * (k & 1))) & 0xffffu); }
};
inline int check_plan_geom(const char *who, int H, int W, int PH, int PW, int py, int px, int A)
{
    CTPVAE_REQUIRE(H > 0 && W > 0 && A > 0, "%s: sizes must be positive (H=%d W=%d A=%d)", who, H, W, A);
    CTPVAE_REQUIRE(py >= 0 && px >= 0 && PH >= H + py && PW >= W + px,
                   "%s: the %dx%d slice at (%d,%d) does not fit the %dx%d canvas", who, H, W, py, px, PH, PW);
    CTPVAE_REQUIRE((long long)PH * PW < (1ll << 24), "%s: canvas too large for fp32 index arithmetic", who);
    return CTPVAE_OK;
}

}  // namespace ctpvae
