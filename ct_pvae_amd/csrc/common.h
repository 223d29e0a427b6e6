// common.h -- shared host-side plumbing of the C ABI (include/ctpvae_radon.h).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/ctpvae_radon.h"

namespace ctpvae {

// thread-local last-error string (the only state of the library)
char *err_buf();
int fail(int code, const char *fmt, ...);

#define CTPVAE_REQUIRE(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) return ::ctpvae::fail(CTPVAE_EINVAL, __VA_ARGS__); \
    } while (0)

#define CTPVAE_HIP(call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ::ctpvae::fail(CTPVAE_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// a launch leaves its error in hipGetLastError(); never synchronises
#define CTPVAE_LAUNCH_CHECK(name)                                                             \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess)                                                                 \
            return ::ctpvae::fail(CTPVAE_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

constexpr int kMaxLdsBytes = 160 * 1024;  // LDS per CU on gfx950; one workgroup may take all of it

// Optional per-slice factor applied by a backward kernel in its final store: gimg[s] = scale[s * stride] * (sum over
// angles).  ptr == nullptr: no factor (x 1.0f, exact).  The backward is linear, so this is the upstream gradient of a
// per-slice sum (an expanded tensor: stride 0 over angles and bins) applied without materialising it.
struct SliceScale {
    const float *ptr;
    long long stride;
    __device__ __forceinline__ float at(int s) const { return ptr ? ptr[(long long)s * stride] : 1.0f; }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a kernel ON A DEVICE: it is set once per (kernel
// instantiation, device) -- `seen` is that instantiation's bit mask of devices already done.
inline bool first_use_on_this_device(std::atomic<unsigned long long> &seen)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (seen.load(std::memory_order_relaxed) & bit) return false;
    seen.fetch_or(bit);
    return true;
}

// Kernels that index slices with a grid y / z dimension take at most this many per launch; their entry points split
// longer batches (CTPVAE_TUNE_MAX_SLICES: a smaller limit, for the tests of that splitting).
inline int max_slices_per_launch()
{
    if (const char *e = getenv("CTPVAE_TUNE_MAX_SLICES")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 65535) return v;
    }
    return 65535;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace ctpvae
