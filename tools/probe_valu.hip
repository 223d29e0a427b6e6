// probe_valu.hip -- issue rate of the VALU ops the projector's inner loop uses, per SIMD, at 1/2/4 waves per SIMD.
// Prints cycles per wave-instruction per SIMD (s_memtime ticks at the shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(s) s s s s s s s s
#define BODY(name, ASM)                                                                             \
    __global__ void name(float *out, long long *cyc, int iters)                                     \
    {                                                                                               \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = 1.5f; int lo = 1, hi = 100;                                                        \
        long long r0 = __builtin_amdgcn_s_memrealtime(); long long t0 = __builtin_amdgcn_s_memtime();  \
        for (int i = 0; i < iters; ++i) {                                                           \
            asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(lo), "v"(hi), "s"(524)); \
        }                                                                                           \
        long long t1 = __builtin_amdgcn_s_memtime(); long long r1 = __builtin_amdgcn_s_memrealtime();  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;        \
        if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long *)&cyc[blockIdx.x], (unsigned long long)(t1 - t0)); if (threadIdx.x == 0) cyc[512 + blockIdx.x] = r1 - r0; }         \
    }

// 8 independent instructions per REP8 element? no: each ASM string holds 8 independent ops (one per chain)
BODY(k_add,  "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k_pkadd(float *out, long long *cyc, int iters)
{
    f32x2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f32x2 b = {1.5f, 2.5f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
BODY(k_rpi,  "v_cvt_rpi_i32_f32 %0, %0\n v_cvt_rpi_i32_f32 %1, %1\n v_cvt_rpi_i32_f32 %2, %2\n v_cvt_rpi_i32_f32 %3, %3\n v_cvt_rpi_i32_f32 %4, %4\n v_cvt_rpi_i32_f32 %5, %5\n v_cvt_rpi_i32_f32 %6, %6\n v_cvt_rpi_i32_f32 %7, %7\n")
BODY(k_med3, "v_med3_i32 %0, %0, %9, %10\n v_med3_i32 %1, %1, %9, %10\n v_med3_i32 %2, %2, %9, %10\n v_med3_i32 %3, %3, %9, %10\n v_med3_i32 %4, %4, %9, %10\n v_med3_i32 %5, %5, %9, %10\n v_med3_i32 %6, %6, %9, %10\n v_med3_i32 %7, %7, %9, %10\n")
BODY(k_mad24, "v_mad_i32_i24 %0, %0, %11, %9\n v_mad_i32_i24 %1, %1, %11, %9\n v_mad_i32_i24 %2, %2, %11, %9\n v_mad_i32_i24 %3, %3, %11, %9\n v_mad_i32_i24 %4, %4, %11, %9\n v_mad_i32_i24 %5, %5, %11, %9\n v_mad_i32_i24 %6, %6, %11, %9\n v_mad_i32_i24 %7, %7, %11, %9\n")
BODY(k_lshladd, "v_lshl_add_u32 %0, %0, 2, %9\n v_lshl_add_u32 %1, %1, 2, %9\n v_lshl_add_u32 %2, %2, 2, %9\n v_lshl_add_u32 %3, %3, 2, %9\n v_lshl_add_u32 %4, %4, 2, %9\n v_lshl_add_u32 %5, %5, 2, %9\n v_lshl_add_u32 %6, %6, 2, %9\n v_lshl_add_u32 %7, %7, 2, %9\n")
BODY(k_cmp, "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n")
BODY(k_mul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
BODY(k_fma, "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
BODY(k_addu, "v_add_u32 %0, %0, %9\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %9\n v_add_u32 %3, %3, %9\n v_add_u32 %4, %4, %9\n v_add_u32 %5, %5, %9\n v_add_u32 %6, %6, %9\n v_add_u32 %7, %7, %9\n")

BODY(k_sdwaadd, "v_add_u32_sdwa %0, %0, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %1, %1, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %3, %3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %4, %4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %5, %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %6, %6, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %7, %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n")
BODY(k_sdwalshl, "v_lshlrev_b32_sdwa %0, 3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %1, 3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %2, 3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %3, 3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %4, 3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %5, 3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %6, 3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %7, 3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n")
BODY(k_bfe, "v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9\n v_bfe_u32 %4, %4, 3, 9\n v_bfe_u32 %5, %5, 3, 9\n v_bfe_u32 %6, %6, 3, 9\n v_bfe_u32 %7, %7, 3, 9\n")
BODY(k_and, "v_and_b32 %0, %0, %9\n v_and_b32 %1, %1, %9\n v_and_b32 %2, %2, %9\n v_and_b32 %3, %3, %9\n v_and_b32 %4, %4, %9\n v_and_b32 %5, %5, %9\n v_and_b32 %6, %6, %9\n v_and_b32 %7, %7, %9\n")
BODY(k_dep1, "v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n")
BODY(k_dep2, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n")
BODY(k_dep4, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n")
// the projector's tap-address sequence as it stands (two rpi, two med3, mad, lshl_add), chained through %0/%1
BODY(k_tap, "v_cvt_rpi_i32_f32 %2, %0\n v_cvt_rpi_i32_f32 %3, %1\n v_med3_i32 %2, %2, %9, %10\n v_med3_i32 %3, %3, %9, %10\n v_mad_i32_i24 %3, %3, %11, %9\n v_lshl_add_u32 %2, %2, 2, %3\n v_cvt_f32_i32 %0, %2\n v_cvt_f32_i32 %1, %3\n")
#define PKBODY(name, ASM)                                                                           \
__global__ void name(float *out, long long *cyc, int iters)                                          \
{                                                                                                    \
    f32x2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;               \
    f32x2 b = {1.5f, 2.5f}; float c0 = 1.f, c1 = 2.f;                                                 \
    long long r0 = __builtin_amdgcn_s_memrealtime(); long long t0 = __builtin_amdgcn_s_memtime();    \
    for (int i = 0; i < iters; ++i) {                                                                \
        asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1) : "v"(b)); \
    }                                                                                                \
    long long t1 = __builtin_amdgcn_s_memtime(); long long r1 = __builtin_amdgcn_s_memrealtime();    \
    f32x2 s = a0 + a1 + a2 + a3;                                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + c0 + c1;                                \
    if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long *)&cyc[blockIdx.x], (unsigned long long)(t1 - t0)); if (threadIdx.x == 0) cyc[512 + blockIdx.x] = r1 - r0; }           \
}
PKBODY(k_pkdep1, "v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %0, %0, %6\n")
PKBODY(k_pkdep2, "v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %6\n v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %6\n v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %6\n v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %6\n")
PKBODY(k_pkmuldep, "v_pk_mul_f32 %0, %0, %6\n v_pk_mul_f32 %0, %0, %6\n v_pk_mul_f32 %0, %0, %6\n v_pk_mul_f32 %0, %0, %6\n v_pk_mul_f32 %0, %0, %6\n v_pk_mul_f32 %0, %0, %6\n v_pk_mul_f32 %0, %0, %6\n v_pk_mul_f32 %0, %0, %6\n")
// the projector's 19-instruction pair block as it stands (tools copy of nearest_pair_addr), 4 blocks per iteration
#define PAIRBLK(A0, A1) \
        "v_pk_mul_f32 v[60:61], %[sx], %[fi]\n v_pk_mul_f32 v[62:63], %[sy], %[fi]\n" \
        "v_pk_add_f32 v[60:61], %[bx], v[60:61]\n v_pk_add_f32 v[62:63], %[by], v[62:63]\n" \
        "v_pk_add_f32 v[60:61], v[60:61], %[hx]\n v_pk_add_f32 v[62:63], v[62:63], %[hy]\n" \
        "v_pk_add_f32 %[fi], %[fi], 2.0 op_sel_hi:[1,0]\n" \
        "v_cvt_rpi_i32_f32 " A0 ", v60\n v_cvt_rpi_i32_f32 " A1 ", v61\n v_cvt_rpi_i32_f32 v62, v62\n v_cvt_rpi_i32_f32 v63, v63\n" \
        "v_med3_i32 " A0 ", " A0 ", %[xlo], %[xhi]\n v_med3_i32 " A1 ", " A1 ", %[xlo], %[xhi]\n" \
        "v_med3_i32 v62, v62, %[ylo], %[yhi]\n v_med3_i32 v63, v63, %[ylo], %[yhi]\n" \
        "v_mad_i32_i24 v62, v62, %[p4], %[o4]\n v_mad_i32_i24 v63, v63, %[p4], %[o4]\n" \
        "v_lshl_add_u32 " A0 ", " A0 ", 2, v62\n v_lshl_add_u32 " A1 ", " A1 ", 2, v63\n"
__global__ void k_pair(float *out, long long *cyc, int iters)
{
    f32x2 fi = {0.f, 1.f}, sx = {0.7f, 0.7f}, sy = {-0.7f, -0.7f}, bx = {(float)threadIdx.x, (float)threadIdx.x}, by = {3.f, 3.f}, hx = {30.f, 30.f}, hy = {90.f, 90.f};
    int xlo = 27, xhi = 156, ylo = 27, yhi = 156, o4 = 64, a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    long long r0 = __builtin_amdgcn_s_memrealtime(); long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(PAIRBLK("%[a0]", "%[a1]") PAIRBLK("%[a2]", "%[a3]") PAIRBLK("%[a4]", "%[a5]") PAIRBLK("%[a6]", "%[a7]")
                     : [fi] "+v"(fi), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6), [a7] "=&v"(a7)
                     : [sx] "v"(sx), [sy] "v"(sy), [bx] "v"(bx), [by] "v"(by), [hx] "v"(hx), [hy] "v"(hy), [xlo] "v"(xlo), [xhi] "v"(xhi),
                       [ylo] "v"(ylo), [yhi] "v"(yhi), [p4] "s"(524), [o4] "v"(o4)
                     : "v60", "v61", "v62", "v63");
    }
    long long t1 = __builtin_amdgcn_s_memtime(); long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = fi.x + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long *)&cyc[blockIdx.x], (unsigned long long)(t1 - t0)); if (threadIdx.x == 0) cyc[512 + blockIdx.x] = r1 - r0; }
}
__global__ void k_pair_big(float *out, long long *cyc, int iters)
{
    f32x2 fi = {0.f, 1.f}, sx = {0.7f, 0.7f}, sy = {-0.7f, -0.7f}, bx = {(float)threadIdx.x, (float)threadIdx.x}, by = {3.f, 3.f}, hx = {30.f, 30.f}, hy = {90.f, 90.f};
    int xlo = 27, xhi = 156, ylo = 27, yhi = 156, o4 = 64, a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    for (int d = 0; d < (int)(threadIdx.x >> 6) * 37; ++d) __builtin_amdgcn_s_sleep(1);
    long long r0 = __builtin_amdgcn_s_memrealtime(); long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters / 4; ++i) {
        asm volatile(PAIRBLK("%[a0]", "%[a1]") PAIRBLK("%[a2]", "%[a3]") PAIRBLK("%[a4]", "%[a5]") PAIRBLK("%[a6]", "%[a7]") PAIRBLK("%[a0]", "%[a1]") PAIRBLK("%[a2]", "%[a3]") PAIRBLK("%[a4]", "%[a5]") PAIRBLK("%[a6]", "%[a7]") PAIRBLK("%[a0]", "%[a1]") PAIRBLK("%[a2]", "%[a3]") PAIRBLK("%[a4]", "%[a5]") PAIRBLK("%[a6]", "%[a7]") PAIRBLK("%[a0]", "%[a1]") PAIRBLK("%[a2]", "%[a3]") PAIRBLK("%[a4]", "%[a5]") PAIRBLK("%[a6]", "%[a7]")
                     : [fi] "+v"(fi), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6), [a7] "=&v"(a7)
                     : [sx] "v"(sx), [sy] "v"(sy), [bx] "v"(bx), [by] "v"(by), [hx] "v"(hx), [hy] "v"(hy), [xlo] "v"(xlo), [xhi] "v"(xhi),
                       [ylo] "v"(ylo), [yhi] "v"(yhi), [p4] "s"(524), [o4] "v"(o4)
                     : "v60", "v61", "v62", "v63");
    }
    long long t1 = __builtin_amdgcn_s_memtime(); long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = fi.x + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long *)&cyc[blockIdx.x], (unsigned long long)(t1 - t0)); if (threadIdx.x == 0) cyc[512 + blockIdx.x] = r1 - r0; }
}
template <class K> void run(const char *name, K k, double ops_per_asm)
{
    float *out; long long *cyc;
    CK(hipMalloc(&out, 1024 * 1024 * 4)); CK(hipMalloc(&cyc, 1024 * 8));
    const int iters = 2000;
    printf("%-12s", name);
    for (int wps : {1, 2, 4, 8}) {   // waves per SIMD: one block per CU of 4*wps waves, 256 CUs
        int block = 256 * wps; int grid = 256;
        if (block > 1024) { grid = 256 * (block / 1024); block = 1024; }
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, cyc, 10);
        CK(hipMemset(cyc, 0, 1024 * 8));
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, cyc, iters);
        CK(hipDeviceSynchronize());
        std::vector<long long> h(1024);
        CK(hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost));
        double avg = 0, rt = 0; for (int i = 0; i < grid; ++i) { avg += h[i]; rt += h[512 + i]; } avg /= grid; rt /= grid;
        if (wps == 1) printf(" [clk %.0f MHz]", avg / rt * 100.0);
        // wave-instructions per SIMD = wps * iters * 8 * ops_per_asm
        printf("  wps=%d: %.2f cyc/inst/SIMD", wps, avg / ((double)wps * iters * 8 * ops_per_asm));
    }
    // per-WAVE issue interval at the projector's launch shapes (one block per CU)
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int block : {768, 769, 770}) {
        const int lds = block == 768 ? 0 : (block == 769 ? 83720 : 140000);
        if (block != 768) block = 768;
        CK(hipMemset(cyc, 0, 1024 * 8));
        hipLaunchKernelGGL(k, dim3(250), dim3(block), lds, 0, out, cyc, iters);
        CK(hipDeviceSynchronize());
        std::vector<long long> h(1024);
        CK(hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (int i = 0; i < 250; ++i) avg += h[i]; avg /= 250;
        printf("  | blk%d: %.2f cyc/inst/wave", block, avg / ((double)iters * 8 * ops_per_asm));
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main()
{
    run("v_add_f32", k_add, 8); run("v_mul_f32", k_mul, 8); run("v_fma_f32", k_fma, 8); run("v_pk_add_f32", k_pkadd, 8);
    run("v_cvt_rpi", k_rpi, 8); run("v_med3_i32", k_med3, 8); run("v_mad_i32_i24", k_mad24, 8);
    run("v_lshl_add", k_lshladd, 8); run("v_cmp_lt_f32", k_cmp, 8); run("v_add_u32", k_addu, 8);
    run("v_add_u32_sdwa", k_sdwaadd, 8); run("v_lshlrev_sdwa", k_sdwalshl, 8); run("v_bfe_u32", k_bfe, 8); run("v_and_b32", k_and, 8);
    run("pair block", k_pair, 76.0 / 8);
    run("pair big/skew", k_pair_big, 76.0 / 8);
    run("pk dep x1", k_pkdep1, 8); run("pk dep x2", k_pkdep2, 8); run("pk_mul dep", k_pkmuldep, 8);
    run("dep chain x1", k_dep1, 8); run("dep chain x2", k_dep2, 8); run("dep chain x4", k_dep4, 8); run("tap sequence", k_tap, 8);
    return 0;
}
