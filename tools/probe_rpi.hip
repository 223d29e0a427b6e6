// probe_rpi.hip -- does v_cvt_rpi_i32_f32 on gfx950 equal floor(x + 0.5) evaluated EXACTLY (ties toward +inf)?
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_rpi probe_rpi.hip ; run on the MI355X.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k(const float *in, int n, int *rpi, int *flr, int *away)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = in[i];
    int r, f;
    asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(f) : "v"(v));
    rpi[i] = r;
    flr[i] = f;
    away[i] = (int)__builtin_roundf(v);
}

int main()
{
    std::vector<float> v;
    auto add3 = [&](float x) { v.push_back(std::nextafterf(x, -1e30f)); v.push_back(x); v.push_back(std::nextafterf(x, 1e30f)); };
    for (int i = -300; i <= 300; ++i) { add3((float)i); add3((float)i + 0.5f); add3((float)i + 0.25f); }
    for (float x : {1e6f, 8388607.5f, 8388608.0f, 1e9f, -1e9f, 3e9f, -3e9f, 0.0f, -0.0f}) add3(x);
    unsigned s = 12345;
    for (int i = 0; i < 200000; ++i) { s = s * 1664525u + 1013904223u; v.push_back(((int)(s >> 8) - (1 << 23)) * (1.0f / 32768.0f)); }
    int n = (int)v.size();
    float *d; int *a, *b, *c;
    hipMalloc(&d, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4);
    hipMemcpy(d, v.data(), n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(d, n, a, b, c);
    std::vector<int> ra(n), rb(n), rc(n);
    hipMemcpy(ra.data(), a, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rb.data(), b, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rc.data(), c, n * 4, hipMemcpyDeviceToHost);
    int bad_rpi = 0, bad_flr = 0, bad_away = 0, rpi_vs_away = 0;
    for (int i = 0; i < n; ++i) {
        double x = v[i];
        if (std::fabs(x) > 2e9) continue;
        long long want_rpi = (long long)std::floor(x + 0.5), want_flr = (long long)std::floor(x), want_away = (long long)std::round(x);
        if (ra[i] != want_rpi) { if (bad_rpi++ < 10) printf("rpi(%.9g) = %d want %lld\n", v[i], ra[i], want_rpi); }
        if (rb[i] != want_flr) { if (bad_flr++ < 10) printf("flr(%.9g) = %d want %lld\n", v[i], rb[i], want_flr); }
        if (rc[i] != want_away) { if (bad_away++ < 10) printf("roundf(%.9g) = %d want %lld\n", v[i], rc[i], want_away); }
        if (ra[i] != rc[i]) { if (rpi_vs_away++ < 6) printf("rpi != away at %.9g: %d vs %d\n", v[i], ra[i], rc[i]); }
    }
    printf("n=%d bad_rpi=%d bad_flr=%d bad_roundf=%d rpi!=away=%d (expected: negative ties only)\n", n, bad_rpi, bad_flr, bad_away, rpi_vs_away);
    return bad_rpi || bad_flr || bad_away;
}
