// Developer probe for the bilinear kernels (round 5):
//  (1) what a sample's pair of horizontally adjacent LDS cells costs by read form -- two ds_read_b32/_b64/_b128 with immediate
//      offsets, one ds_read2_b32/_b64, one ds_read_b64/_b128 at an address aligned to HALF its width (unaligned access mode) --
//      on linear (conflict-free) addresses and on ray-like ones (lane l reads cell round(l * 1.31) + row * pitch);
//  (2) whether v_fract_f32(x) == x - floorf(x) and 1 - fract(x) == (floorf(x) + 1) - x bit for bit on x >= 0 (the weights of
//      TensorFlow's bilinear sample), and what happens below zero.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o probe_bilin.bin probe_bilin.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// FORM: 0 two b32 | 1 read2_b32 | 2 b64 at 4-byte alignment | 3 two b64 | 4 read2_b64 | 5 b128 at 8-byte alignment | 6 two b128
template <int FORM> __global__ __launch_bounds__(1024) void k(float *out, int iters, int raylike, int check)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int CELL = FORM <= 2 ? 4 : (FORM <= 5 ? 8 : 16);   // bytes per cell
    const int cell = raylike ? (int)rintf(lane * 1.31f) + ((lane * 3) >> 3) * 161 : lane * 2;
    unsigned addr = (unsigned)(cell + 1 + wave * 7) * CELL;       // odd cell offsets too: half-width alignment for forms 2 and 5
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f4 v = {0, 0, 0, 0};
            if constexpr (FORM == 0) {
                asm volatile("ds_read_b32 %0, %2 offset:%3\n\tds_read_b32 %1, %2 offset:%4" : "=&v"(v.x), "=&v"(v.y) : "v"(addr), "n"(u * 2048), "n"(u * 2048 + 4));
            } else if constexpr (FORM == 1) {
                f2 w;
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(w) : "v"(addr), "n"(u * 16), "n"(u * 16 + 1));
                v.x = w.x; v.y = w.y;
            } else if constexpr (FORM == 2) {
                f2 w;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w) : "v"(addr), "n"(u * 2048));
                v.x = w.x; v.y = w.y;
            } else if constexpr (FORM == 3) {
                f2 a, b;
                asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=&v"(a), "=&v"(b) : "v"(addr), "n"(u * 4096), "n"(u * 4096 + 8));
                v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
            } else if constexpr (FORM == 4) {
                asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(u * 16), "n"(u * 16 + 1));
            } else if constexpr (FORM == 5) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(u * 4096));
            } else {
                f4 b;
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(v), "=&v"(b) : "v"(addr), "n"(u * 8192), "n"(u * 8192 + 16));
                v.y += b.x;
            }
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FORM == 0 || FORM == 3 || FORM == 6 ? 6 : 3) : "memory");
            acc += v.x + v.y + v.z + v.w;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (check) {   // one read of each form, returned for the host to compare with the dwords it expects
        f4 v = {0, 0, 0, 0};
        if constexpr (FORM == 2) {
            f2 w;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr));
            v.x = w.x; v.y = w.y;
        } else if constexpr (FORM == 5) {
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
        }
        acc = (v.x == (float)(addr / 4) && v.y == (float)(addr / 4 + 1) && (FORM == 2 || (v.z == (float)(addr / 4 + 2) && v.w == (float)(addr / 4 + 3)))) ? 1.0f : 0.0f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void fract_kernel(const float *x, int n, int *bad, float *ex)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float fr;
    asm("v_fract_f32 %0, %1" : "=v"(fr) : "v"(v));
    int fl;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(fl) : "v"(v));
    const float xf = floorf(v), xc = xf + 1.0f;
    const float w1 = v - xf, w0 = xc - v;
    const float w0b = 1.0f - fr;
    if (fr != w1 || w0 != w0b || fl != (int)xf) {
        const int k = atomicAdd(bad, 1);
        if (k < 8) { ex[4 * k] = v; ex[4 * k + 1] = fr; ex[4 * k + 2] = w1; ex[4 * k + 3] = w0 - w0b; }
    }
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    std::vector<float> host(256 * 1024);
    auto run = [&](auto kern, int raylike, const char *name, bool check) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        kern<<<256, 1024, 160 * 1024>>>(out, 10, raylike, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<256, 1024, 160 * 1024>>>(out, iters, raylike, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        int okc = -1;
        if (check) {
            kern<<<256, 1024, 160 * 1024>>>(out, 1, raylike, 1);
            hipMemcpy(host.data(), out, host.size() * 4, hipMemcpyDeviceToHost);
            okc = 0;
            for (float f : host) okc += f == 1.0f;
        }
        printf("%-44s %s  %8.3f ms  %6.2f ns per cell pair per wave per CU%s\n", name, raylike ? "ray-like" : "linear  ", ms,
               ms * 1e6 / (16.0 * iters * 4), check ? (okc == (int)host.size() ? "  [data ok]" : "  [DATA WRONG]") : "");
    };
    for (int rl = 0; rl < 2; ++rl) {
        run(k<0>, rl, "4-B cells: two ds_read_b32", false);
        run(k<1>, rl, "4-B cells: ds_read2_b32", false);
        run(k<2>, rl, "4-B cells: ds_read_b64, 4-byte aligned", true);
        run(k<3>, rl, "8-B cells: two ds_read_b64", false);
        run(k<4>, rl, "8-B cells: ds_read2_b64", false);
        run(k<5>, rl, "8-B cells: ds_read_b128, 8-byte aligned", true);
        run(k<6>, rl, "16-B cells: two ds_read_b128", false);
    }
    // (2) fract
    const int n = 1 << 22;
    std::vector<float> xs(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const double u = (double)(st >> 11) / 9007199254740992.0;
        xs[i] = (float)(i < n / 2 ? u * 760.0 : (i < 3 * n / 4 ? u * 2.0 : -u * 3.0));   // [0, 760), [0, 2), (-3, 0]
    }
    float *dx, *dex;
    int *dbad;
    hipMalloc(&dx, n * 4);
    hipMalloc(&dex, 128);
    hipMalloc(&dbad, 4);
    hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
    for (int part = 0; part < 3; ++part) {
        const int off = part == 0 ? 0 : (part == 1 ? n / 2 : 3 * n / 4), cnt = part == 0 ? n / 2 : n / 4;
        hipMemset(dbad, 0, 4);
        fract_kernel<<<(cnt + 255) / 256, 256>>>(dx + off, cnt, dbad, dex);
        int bad;
        float ex[32];
        hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
        hipMemcpy(ex, dex, 128, hipMemcpyDeviceToHost);
        printf("v_fract / 1 - fract / v_cvt_flr vs x - floor / (floor + 1) - x / (int)floor on %s: %d of %d differ", part == 0 ? "[0, 760)" : (part == 1 ? "[0, 2)" : "(-3, 0]"), bad, cnt);
        if (bad) printf("  e.g. x=%.9g fract=%.9g x-floor=%.9g d(w0)=%.3g", ex[0], ex[1], ex[2], ex[3]);
        printf("\n");
    }
    return 0;
}
