// Developer probe: does the LDS skip the passes of INACTIVE lanes?  A ds_read_b128 is served in four groups of 16 lanes
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32), a ds_read_b64 in two of 32.  Rays of one task end at different
// rows; if a group whose lanes are all masked off (EXEC) costs nothing, a walk that drops finished lanes saves LDS time.
//   hipcc -O3 --offload-arch=gfx950 -o probe_lds_exec.bin probe_lds_exec.hip
// Each variant: 256 workgroups x 16 waves, conflict-free addresses (lane * width), 8 reads in flight, EXEC = mask in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int WIDTH> __global__ __launch_bounds__(1024) void k(unsigned long long mask, float *out, int iters)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned addr = lane * WIDTH;
    float acc = 0.0f;
    if ((mask >> lane) & 1ull) {   // the loop runs with EXEC = mask
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (WIDTH == 16) {
                    f4 v;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(u * 1024));
                    asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                    acc += v.x;
                } else {
                    f2 v;
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(u * 512));
                    asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                    acc += v.x;
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
static unsigned long long lanes(std::initializer_list<int> ranges)
{
    unsigned long long m = 0;
    const int *p = ranges.begin();
    for (size_t i = 0; i + 1 < ranges.size(); i += 2)
        for (int l = p[i]; l <= p[i + 1]; ++l) m |= 1ull << l;
    return m;
}
int main()
{
    float *out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](auto kern, unsigned long long mask, const char *name) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        kern<<<256, 1024, 64 * 1024>>>(mask, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<256, 1024, 64 * 1024>>>(mask, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_cu = 16.0 * iters * 8;
        printf("%-64s %.3f ms  %.2f ns per wave-instruction per CU\n", name, ms, ms * 1e6 / inst_per_cu);
    };
    const unsigned long long g0 = lanes({0, 3, 12, 15, 20, 27}), g1 = lanes({4, 11, 16, 19, 28, 31});
    run(k<16>, ~0ull, "b128 all 64 lanes");
    run(k<16>, 0xffffffffull, "b128 lanes 0-31 (hardware groups 0 and 1)");
    run(k<16>, g0, "b128 one hardware group (lanes 0-3, 12-15, 20-27)");
    run(k<16>, g0 | (g0 << 32), "b128 groups 0 and 2");
    run(k<16>, g0 | g1 | (g0 << 32), "b128 three groups");
    run(k<16>, 0xffffull, "b128 lanes 0-15 (parts of groups 0 and 1)");
    run(k<16>, 0x000f000f000f000full, "b128 four lanes of every group... (0-3, 16-19, 32-35, 48-51)");
    run(k<16>, 1ull, "b128 one lane");
    run(k<8>, ~0ull, "b64 all 64 lanes");
    run(k<8>, 0xffffffffull, "b64 lanes 0-31");
    run(k<8>, 0xffff0000ffffull, "b64 lanes 0-15 and 32-47");
    run(k<8>, 0xffffull, "b64 lanes 0-15");
    run(k<8>, 1ull, "b64 one lane");
    return 0;
}
