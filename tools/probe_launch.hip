// probe_launch.hip -- developer probe: cost of launching workgroups with large LDS allocations / many waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(1024) void k_empty(float *out) { extern __shared__ float lds[]; if (out && threadIdx.x == 9999) out[0] = lds[0]; }
__global__ __launch_bounds__(1024) void k_touch(float *out, int n) { extern __shared__ float lds[]; for (int p = threadIdx.x; p < n; p += blockDim.x) lds[p] = 0.f; __syncthreads(); if (out && threadIdx.x == 9999) out[0] = lds[0]; }
template <class F> static float time_us(F f, int n = 200)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) f();
    CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1000.f / n;
}
int main()
{
    CK(hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int wgs : {192, 256, 512, 2048})
        for (int threads : {256, 1024})
            for (int lds_kb : {0, 64, 152}) {
                float t = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(threads), lds_kb * 1024, 0, nullptr); });
                float t2 = time_us([&] { hipLaunchKernelGGL(k_touch, dim3(wgs), dim3(threads), lds_kb * 1024, 0, nullptr, lds_kb * 256); });
                printf("wgs=%4d threads=%4d lds=%3d KB: empty %.2f us, zero-fill LDS %.2f us\n", wgs, threads, lds_kb, t, t2);
            }
    return 0;
}
