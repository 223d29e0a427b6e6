// probe_floor.hip -- the launch floor of kernels shaped like the projector's: a stream of back-to-back launches (HIP graph
// replay) of an (almost) empty kernel, by workgroups, threads per workgroup and dynamic LDS.  The body only touches one
// LDS word and stores one float per workgroup, so what is timed is dispatch + ramp + drain + the kernel boundary.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void empty_kernel(float *out)
{
    extern __shared__ float lds[];
    if (threadIdx.x == 0) lds[0] = (float)blockIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0];
}

int main()
{
    float *out;
    CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipFuncSetAttribute((const void *)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const int wgs[] = {64, 125, 250, 400, 1000, 4000};
    const int thr[] = {64, 256, 512, 1024};
    const int lds[] = {1024, 30 * 1024, 64 * 1024, 128 * 1024};
    printf("%6s %6s %8s   us per launch\n", "wgs", "thr", "lds");
    for (int w : wgs)
        for (int t : thr)
            for (int l : lds) {
                hipGraph_t g;
                hipGraphExec_t x;
                CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (int s = 0; s < 20; ++s) hipLaunchKernelGGL(empty_kernel, dim3(w), dim3(t), l, st, out);
                CHECK(hipStreamEndCapture(st, &g));
                CHECK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0));
                CHECK(hipEventCreate(&e1));
                CHECK(hipGraphLaunch(x, st));
                CHECK(hipStreamSynchronize(st));
                CHECK(hipEventRecord(e0, st));
                for (int r = 0; r < 25; ++r) CHECK(hipGraphLaunch(x, st));
                CHECK(hipEventRecord(e1, st));
                CHECK(hipStreamSynchronize(st));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("%6d %6d %8d   %.2f\n", w, t, l, ms * 1e3 / 500);
                CHECK(hipGraphExecDestroy(x));
                CHECK(hipGraphDestroy(g));
            }
    return 0;
}
