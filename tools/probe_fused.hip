// probe_fused.hip -- what ONE launch for forward + adjoint could save at the headline shape.
//
// The training call is forward (250 workgroups x 16 waves x 128 KB LDS at B=50, A=20) -> adjoint (400 workgroups x 8 waves
// x 29 KB), the adjoint of slice pair u reading what the 10 forward workgroups of pair u wrote.  This probe times the
// SKELETON of both structures with stand-in work (every workgroup stores its share of the real output bytes, then idles
// for about the time the real body takes), so that what is compared is exactly what fusion changes: dispatch, ramp,
// drain and the hand-off.
//   A  two launches back to back (replayed from a HIP graph), the kernel boundary doing the synchronisation;
//   B  ONE launch of 250 + 400 workgroups: producers publish with the guide's recipe (every wave s_waitcnt vmcnt(0) ->
//      barrier -> lane 0 agent-scope release -> s_waitcnt -> relaxed agent fetch_add on the pair's counter), consumers poll
//      their pair's counter (relaxed sc1 load + s_sleep, bounded), then one agent-scope acquire -> barrier -> read;
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_fused.hip -o tools/probe_fused.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int kPairs = 25, kFwdPerPair = 10, kBwdPerPair = 16;        // 250 forward, 400 adjoint workgroups
constexpr int kFwdWgs = kPairs * kFwdPerPair, kBwdWgs = kPairs * kBwdPerPair;
constexpr int kRowFloats = 2 * 20 * 184;                               // a pair's dlp: 2 slices x 20 angles x 184 bins

__device__ __forceinline__ void idle_cycles(long long cycles)
{
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}

// stand-in forward body: touch LDS, idle ~busy cycles, store this workgroup's 1/10 of the pair's rows
__device__ __forceinline__ void fwd_body(float *dlp, int pair, int part, long long busy, float *lds)
{
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    idle_cycles(busy);
    const int n = kRowFloats / kFwdPerPair;
    float *dst = dlp + (size_t)pair * kRowFloats + part * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        dst[i] = lds[i & 32767] + (float)pair;
    }
}

// stand-in adjoint body: read the pair's rows, idle, store 1/16 of the pair's two gradient images
__device__ __forceinline__ void bwd_body(const float *dlp, float *gimg, int pair, int tile, long long busy, float *lds)
{
    for (int i = threadIdx.x; i < kRowFloats; i += blockDim.x) lds[i] = dlp[(size_t)pair * kRowFloats + i];
    __syncthreads();
    idle_cycles(busy);
    const int n = 2 * 128 * 128 / kBwdPerPair;
    float *dst = gimg + (size_t)pair * 2 * 128 * 128 + tile * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = lds[i % kRowFloats];
}

__global__ __launch_bounds__(1024) void fwd_kernel(float *dlp, long long busy)
{
    extern __shared__ float lds[];
    fwd_body(dlp, blockIdx.x / kFwdPerPair, blockIdx.x % kFwdPerPair, busy, lds);
}
__global__ __launch_bounds__(512) void bwd_kernel(const float *dlp, float *gimg, long long busy)
{
    extern __shared__ float lds[];
    bwd_body(dlp, gimg, blockIdx.x / kBwdPerPair, blockIdx.x % kBwdPerPair, busy, lds);
}

// ONE launch: blocks [0, 250) are forward workgroups, the rest adjoint workgroups (a launch has ONE block size and ONE LDS
// request: the adjoint workgroups carry the forward's 16 waves and 128 KB too).  counters[pair] counts published forward workgroups; `epoch` makes the
// counters monotone across replays (no memset between launches).
__global__ __launch_bounds__(1024) void fused_kernel(float *dlp, float *gimg, unsigned *counters, unsigned epoch,
                                                     long long busy_f, long long busy_b, unsigned *timeouts)
{
    extern __shared__ float lds[];
    if (blockIdx.x < kFwdWgs) {
        const int pair = blockIdx.x / kFwdPerPair;
        fwd_body(dlp, pair, blockIdx.x % kFwdPerPair, busy_f, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&counters[pair], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int b = blockIdx.x - kFwdWgs, pair = b / kBwdPerPair;
    if (threadIdx.x == 0) {
        const unsigned want = epoch * kFwdPerPair;
        unsigned spins = 0;
        while (__hip_atomic_load(&counters[pair], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > 4000000u) {                    // bounded: a stuck grid ends with the timeout word set
                atomicAdd(timeouts, 1u);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    bwd_body(dlp, gimg, pair, b % kBwdPerPair, busy_b, lds);
}

static float time_graph(hipStream_t st, hipGraphExec_t g, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipGraphLaunch(g, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(g, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char **argv)
{
    const double busy_f_us = argc > 1 ? atof(argv[1]) : 2.5, busy_b_us = argc > 2 ? atof(argv[2]) : 1.2;
    const long long busy_f = (long long)(busy_f_us * 100), busy_b = (long long)(busy_b_us * 100);   // s_memtime: 100 MHz
    float *dlp, *gimg;
    unsigned *counters, *timeouts;
    CHECK(hipMalloc(&dlp, sizeof(float) * kPairs * kRowFloats));
    CHECK(hipMalloc(&gimg, sizeof(float) * kPairs * 2 * 128 * 128));
    CHECK(hipMalloc(&counters, sizeof(unsigned) * kPairs * 64));
    CHECK(hipMalloc(&timeouts, sizeof(unsigned)));
    CHECK(hipMemset(counters, 0, sizeof(unsigned) * kPairs * 64));
    CHECK(hipMemset(timeouts, 0, sizeof(unsigned)));
    CHECK(hipFuncSetAttribute((const void *)fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CHECK(hipFuncSetAttribute((const void *)fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const int steps = 10;

    // A: two launches per step
    hipGraph_t ga;
    hipGraphExec_t xa;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < steps; ++s) {
        hipLaunchKernelGGL(fwd_kernel, dim3(kFwdWgs), dim3(1024), 131072, st, dlp, busy_f);
        hipLaunchKernelGGL(bwd_kernel, dim3(kBwdWgs), dim3(512), 29440, st, dlp, gimg, busy_b);
    }
    CHECK(hipStreamEndCapture(st, &ga));
    CHECK(hipGraphInstantiate(&xa, ga, nullptr, nullptr, 0));
    const float ta = time_graph(st, xa, 50);
    printf("A  two launches per step            : %.2f us per step\n", ta * 1e3 / (50 * steps));

    // B: one fused launch per step.  NOTE the LDS request is the forward's (128 KB) for every workgroup of the launch: the
    // adjoint workgroups of a real fused kernel would carry it too, one workgroup per CU instead of five.
    for (int variant = 0; variant < 1; ++variant) {
        unsigned epoch = 0;
        // epochs are baked into the captured launches: capture `steps` launches with epochs 1..steps, and offset the counters
        // back to 0 with a memset node at the head of every replay
        hipGraph_t gb;
        hipGraphExec_t xb;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        CHECK(hipMemsetAsync(counters, 0, sizeof(unsigned) * kPairs * 64, st));
        for (int s = 0; s < steps; ++s) {
            ++epoch;
            hipLaunchKernelGGL(fused_kernel, dim3(kFwdWgs + kBwdWgs), dim3(1024), 131072, st, dlp, gimg, counters, epoch, busy_f,
                               busy_b, timeouts);
        }
        CHECK(hipStreamEndCapture(st, &gb));
        CHECK(hipGraphInstantiate(&xb, gb, nullptr, nullptr, 0));
        const float tb = time_graph(st, xb, 50);
        unsigned to = 0;
        CHECK(hipMemcpy(&to, timeouts, sizeof(unsigned), hipMemcpyDeviceToHost));
        printf("B  one launch, counter hand-off     : %.2f us per step (timeouts: %u)\n", tb * 1e3 / (50 * steps), to);
    }
    // C: forward alone and adjoint alone, for reference
    hipGraph_t gc;
    hipGraphExec_t xc;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(fwd_kernel, dim3(kFwdWgs), dim3(1024), 131072, st, dlp, busy_f);
    CHECK(hipStreamEndCapture(st, &gc));
    CHECK(hipGraphInstantiate(&xc, gc, nullptr, nullptr, 0));
    printf("   forward stand-in alone           : %.2f us per launch\n", time_graph(st, xc, 50) * 1e3 / (50 * steps));
    hipGraph_t gd;
    hipGraphExec_t xd;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(bwd_kernel, dim3(kBwdWgs), dim3(512), 29440, st, dlp, gimg, busy_b);
    CHECK(hipStreamEndCapture(st, &gd));
    CHECK(hipGraphInstantiate(&xd, gd, nullptr, nullptr, 0));
    printf("   adjoint stand-in alone           : %.2f us per launch\n", time_graph(st, xd, 50) * 1e3 / (50 * steps));
    return 0;
}
