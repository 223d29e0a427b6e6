// tune_stamps.h -- timing builds only (-DCTPVAE_TUNE_STAMPS; tools/stamp_rounds.hip): per-wave time stamps of the planned kernels
// and of the bilinear forward.  Without the define CTPVAE_PSTAMP is empty.
#pragma once
#include <hip/hip_runtime.h>
namespace ctpvae {
#ifdef CTPVAE_TUNE_STAMPS
// timing builds (tools/stamp_rounds.hip): per wave {s_memtime at start / fill issued / barrier passed / end, s_memrealtime (100 MHz)
// at start / end, HW_ID | XCC_ID << 32}
__device__ long long g_pstamps[8 * 65536];
static int g_pshape[10];   // host: the last planned forward's {units, workgroups per unit, waves, slices per unit, affine}
#define CTPVAE_PSTAMP(slot)                                                                                  \
    do {                                                                                                     \
        long long t_;                                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
        const size_t w_ = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);                       \
        if ((threadIdx.x & 63) == 0 && w_ < 65536) {                                                         \
            g_pstamps[8 * w_ + (slot)] = t_;                                                                 \
            if ((slot) == 0 || (slot) == 3) g_pstamps[8 * w_ + 4 + (slot) / 3] = __builtin_amdgcn_s_memrealtime(); \
            if ((slot) == 0) {                                                                               \
                unsigned hw_, xcc_;                                                                          \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_), "=s"(xcc_)); \
                g_pstamps[8 * w_ + 6] = (long long)hw_ | ((long long)xcc_ << 32);                            \
            }                                                                                                \
        }                                                                                                    \
    } while (0)
#else
#define CTPVAE_PSTAMP(slot)
#endif
}  // namespace ctpvae
