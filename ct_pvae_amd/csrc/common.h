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
// instantiation, device) -- `seen` is that instantiation's bit mask of devices already done.  The bit is published only
// AFTER the attribute call succeeded: a second thread that races the first simply sets the attribute again (harmless),
// it can never launch ahead of it; a failed call leaves the bit clear, so the next launch retries.
#define CTPVAE_SET_MAX_LDS_ONCE(kernel, seen)                                                                     \
    do {                                                                                                          \
        int dev_ = 0;                                                                                             \
        CTPVAE_HIP(hipGetDevice(&dev_));                                                                          \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                      \
        if (dev_ >= 64 || !((seen).load(std::memory_order_acquire) & bit_)) {                                     \
            CTPVAE_HIP(hipFuncSetAttribute((const void *)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                           ::ctpvae::kMaxLdsBytes));                                              \
            if (dev_ < 64) (seen).fetch_or(bit_, std::memory_order_release);                                      \
        }                                                                                                         \
    } while (0)

// Kernels that address LDS by ABSOLUTE byte addresses (rotate_plan.hip, cplan_walk.h) rely on their dynamic LDS starting at
// address 0, i.e. on having NO static LDS.  That is a property of the code object, knowable on the host: checked once per
// (kernel instantiation, device) before the first launch -- a kernel that acquired static LDS (another hipcc, a __shared__ in
// an inlined helper, a profiler's instrumentation) makes the entry point fail with CTPVAE_EHIP and a message, instead of the
// device-side trap of round 3, which aborted the caller's process.  `seen` as in CTPVAE_SET_MAX_LDS_ONCE; the bit is
// published only after the check passed.  (Knob FAKE_STATIC_LDS = n adds n bytes to what the runtime reports and forces the
// check to run on every launch: the tests' stub for the failure path.)
#define CTPVAE_REQUIRE_NO_STATIC_LDS(kernel, name, seen)                                                           \
    do {                                                                                                          \
        int dev_ = 0;                                                                                             \
        CTPVAE_HIP(hipGetDevice(&dev_));                                                                          \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                      \
        const int fake_ = ::ctpvae::knob(::ctpvae::kKnobFakeStaticLds);                                            \
        if (dev_ >= 64 || fake_ > 0 || !((seen).load(std::memory_order_acquire) & bit_)) {                        \
            hipFuncAttributes fa_;                                                                                \
            CTPVAE_HIP(hipFuncGetAttributes(&fa_, (const void *)(kernel)));                                       \
            const long long static_ = (long long)fa_.sharedSizeBytes + (fake_ > 0 ? fake_ : 0);                    \
            if (static_ != 0)                                                                                     \
                return ::ctpvae::fail(CTPVAE_EHIP, "%s was built with %lld bytes of static LDS: its gathers address LDS " \
                                      "absolutely and need the dynamic array at address 0 -- rebuild the library (no "    \
                                      "__shared__ variables in these kernels)", name, static_);                   \
            if (dev_ < 64 && fake_ <= 0) (seen).fetch_or(bit_, std::memory_order_release);                        \
        }                                                                                                         \
    } while (0)

// Developer knobs (tools/ sweeps and the tests that force one code path against another).  None changes results.
// They are NOT read from the environment in the launch path: the registry is filled once, when the library is loaded,
// from CTPVAE_TUNE_<NAME> / CTPVAE_NO_PLAN / CTPVAE_FORCE_GENERIC, and changed afterwards only through
// ctpvae_tune_set() (include/ctpvae_radon.h).  A knob reads -1 when unset.
enum Knob {
    kKnobNoPlan, kKnobForceGeneric, kKnobNs, kKnobG, kKnobWaves, kKnobBns, kKnobBw, kKnobSegNs, kKnobSegChunk,
    kKnobSegPpt, kKnobTiledNs, kKnobTiledG, kKnobSiddonNs, kKnobSiddonThreads, kKnobSiddonPpb, kKnobMaxSlices,
    kKnobSiddonBwdNs, kKnobSiddonBwdChunks, kKnobNoCompact, kKnobSkew0, kKnobTiledSort, kKnobTiledPair, kKnobAffine, kKnobFakeStaticLds, kKnobFoldSums, kKnobTiledXcd, kKnobTiledWaves, kKnobTiledTh, kKnobReduceWaves, kKnobStepNs, kKnobStepLdsKb, kKnobTiledForce, kKnobBsort, kKnobMixG, kKnobMixG2, kKnobMixU1, kKnobNoMagic, kKnobMixG1, kKnobMixG3, kKnobMixU2, kKnobBrsplit, kKnobCount
};
int knob(Knob k);

// Kernels that index slices with a grid y / z dimension take at most this many per launch; their entry points split
// longer batches (knob MAX_SLICES: a smaller limit, for the tests of that splitting).
inline int max_slices_per_launch()
{
    const int v = knob(kKnobMaxSlices);
    return (v >= 1 && v <= 65535) ? v : 65535;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace ctpvae
