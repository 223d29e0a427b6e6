// Developer probe: LDS cycles of a ds_read_b64 whose 32 lanes of a half-wave read DIFFERENT rows (row = a per-lane code) at a
// per-lane column (lane & 31) * 8 -- the compact plan's replicated step table -- against all lanes reading one row, and
// against an un-replicated 8-byte table (address = code * 8).   hipcc -O3 --offload-arch=gfx950 -o probe_lut.bin probe_lut.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x2 *lds_u2;
template <int MODE> __global__ __launch_bounds__(1024) void k(const unsigned *codes, unsigned *out, int iters)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned acc = 0;
    unsigned c = codes[threadIdx.x];
    unsigned addr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // addresses fixed per lane: the loop below is reads + one accumulate each
        const unsigned code = (c >> (u * 3)) & 63u;
        if (MODE == 0) addr[u] = (code << 8) | ((lane & 31) << 3);          // replicated per lane of a half-wave
        else if (MODE == 1) addr[u] = (u << 8) | ((lane & 31) << 3);        // same row for every lane
        else if (MODE == 2) addr[u] = code << 3;                           // one 64-entry table, no replication
        else if (MODE == 3) addr[u] = (code << 9) | (lane << 3);           // replicated per lane of the WAVE (64 copies)
        else addr[u] = ((code << 5) + ((lane * 17 + u * 5) & 31)) << 3;    // MODE 4: 32 distinct cells, scattered rows+cols
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            u32x2 v;
            asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr[u]));
            asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
            acc ^= v.x;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    unsigned *codes, *out;
    hipMalloc(&codes, 1024 * 4);
    hipMalloc(&out, 256 * 1024 * 4);
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 2654435761u * (i + 1);
    hipMemcpy(codes, h.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](auto kern, const char *name) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        kern<<<256, 1024, 64 * 1024>>>(codes, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<256, 1024, 64 * 1024>>>(codes, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_cu = 16.0 * iters * 8;
        printf("%-40s %.3f ms  -> %.2f ns per wave-instruction per CU (%.2f cycles at 2.1 GHz)\n", name, ms, ms * 1e6 / inst_per_cu,
               ms * 1e6 / inst_per_cu * 2.1);
    };
    run(k<1>, "same row, column = lane%32");
    run(k<0>, "row = code, column = lane%32 (32 copies)");
    run(k<3>, "row = code, column = lane (64 copies)");
    run(k<2>, "one table, address = code * 8");
    run(k<4>, "32 distinct cells, scattered");
    return 0;
}
