// probe_fill.hip -- how fast can one workgroup per CU pull a 64 KiB slice from L2/HBM into LDS?  (cycles per fill)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 0: row-contiguous float4 loads, all issued first, summed (no LDS)
// MODE 1: same + ds_write_b128 to a linear LDS image
// MODE 2: 4-row x 32-col blocks per 32-lane group (the projector's conflict-free arrangement), ds_write_b32 x4, pitch 129
template <int MODE, int NB>
__global__ __launch_bounds__(1024) void fill(const float *img, int slices, float *out, long long *cyc)
{
    extern __shared__ float lds[];
    const int s = (blockIdx.x * 7) % slices;
    const float4 *src = reinterpret_cast<const float4 *>(img + (size_t)s * 16384);
    const int tid = threadIdx.x, nt = blockDim.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    if (MODE < 2) {
        for (int p0 = tid; p0 < 4096; p0 += NB * nt) {
            float4 v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) v[u] = src[min(p0 + u * nt, 4095)];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (MODE == 0) acc += v[u].x + v[u].y + v[u].z + v[u].w;
                else if (p0 + u * nt < 4096) reinterpret_cast<float4 *>(lds)[p0 + u * nt] = v[u];
            }
        }
    } else {
        const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
        const int h = lane >> 5, k = (lane & 31) >> 3, m = lane & 7;
        for (int p0 = wave; p0 < 64; p0 += NB * nw) {
            float4 v[NB]; int r_[NB], c_[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int pp = p0 + u * nw, rq = pp >> 1, pc = pp & 1;
                r_[u] = 4 * rq + k; c_[u] = 32 * (2 * pc + h) + 4 * m;
                v[u] = *reinterpret_cast<const float4 *>(img + (size_t)s * 16384 + min(r_[u], 127) * 128 + c_[u]);
                if (pp >= 64) r_[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (r_[u] < 0) continue;
                float *d = lds + r_[u] * 129 + c_[u];
                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
            }
        }
    }
    __syncthreads();
    long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE > 0) acc = lds[tid];
    out[blockIdx.x * nt + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K> void run(const char *name, K k, const float *img, int slices, float *out, long long *cyc)
{
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int grid : {64, 128, 250, 500}) for (int block : {256, 512, 960}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 70000, 0, img, slices, out, cyc);
        CK(hipDeviceSynchronize());
        std::vector<long long> h(grid);
        CK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
        double avg = 0, mx = 0; for (auto v : h) { avg += v; mx = std::max<double>(mx, v); } avg /= grid;
        printf("%-28s grid=%3d block=%4d : avg %6.0f max %6.0f cycles per 64 KiB fill (%.1f B/cyc/CU)\n", name, grid, block, avg, mx, 65536.0 / avg);
    }
}

int main()
{
    const int slices = 50;
    float *img, *out; long long *cyc;
    CK(hipMalloc(&img, slices * 65536)); CK(hipMalloc(&out, 1024 * 1024 * 4)); CK(hipMalloc(&cyc, 4096 * 8));
    CK(hipMemset(img, 0, slices * 65536));
    run("rowmajor f4, no LDS, NB=4", fill<0, 4>, img, slices, out, cyc);
    run("rowmajor f4, no LDS, NB=16", fill<0, 16>, img, slices, out, cyc);
    run("rowmajor f4 + ds_write_b128", fill<1, 8>, img, slices, out, cyc);
    run("4x32 blocks + 4 ds_write_b32", fill<2, 8>, img, slices, out, cyc);
    return 0;
}
