// core.hip -- error state, size rules (a1) and the transform-table kernel (a3/a4).
#include <cmath>
#include <cstring>

#include "common.h"

namespace ctpvae {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ---- developer knobs ---------------------------------------------------------------------------------------------
static const char *const kKnobNames[kKnobCount] = {
    "NO_PLAN", "FORCE_GENERIC", "NS", "G", "WAVES", "BNS", "BW", "SEG_NS", "SEG_CHUNK", "SEG_PPT", "TILED_NS", "TILED_G",
    "SIDDON_NS", "SIDDON_THREADS", "SIDDON_PPB", "MAX_SLICES", "SIDDON_BWD_NS", "SIDDON_BWD_CHUNKS", "NO_COMPACT", "SKEW0", "TILED_SORT", "TILED_PAIR", "AFFINE", "FAKE_STATIC_LDS", "FOLD_SUMS", "TILED_XCD", "TILED_WAVES", "TILED_TH", "REDUCE_WAVES", "STEP_NS", "STEP_LDS_KB", "TILED_FORCE", "BSORT", "MIXG", "MIXG_G2", "MIXG_U1", "NO_MAGIC", "MIXG_G1", "MIXG_G3", "MIXG_U2", "BRSPLIT"};
static std::atomic<int> g_knobs[kKnobCount];
static int find_knob(const char *name)
{
    for (int k = 0; k < kKnobCount; ++k)
        if (strcmp(name, kKnobNames[k]) == 0) return k;
    return -1;
}
// Filled once at load time from the environment (CTPVAE_NO_PLAN / CTPVAE_FORCE_GENERIC: set = 1; CTPVAE_TUNE_<NAME>=n).
static const bool g_knobs_loaded = [] {
    for (int k = 0; k < kKnobCount; ++k) {
        int v = -1;
        char env[64];
        if (k == kKnobNoPlan || k == kKnobForceGeneric) {
            snprintf(env, sizeof env, "CTPVAE_%s", kKnobNames[k]);
            if (getenv(env) != nullptr) v = 1;
        } else {
            snprintf(env, sizeof env, "CTPVAE_TUNE_%s", kKnobNames[k]);
            if (const char *e = getenv(env)) v = atoi(e);
        }
        g_knobs[k].store(v, std::memory_order_relaxed);
    }
    return true;
}();
int knob(Knob k) { return g_knobs[k].load(std::memory_order_relaxed); }

// 3x3 fp32 inverse by LU with partial pivoting -- the arithmetic TensorFlow's matrix_inverse
// performs on the flat transform inside the gradient of ImageProjectiveTransformV3.
__host__ __device__ static void inv3x3(const float m[9], float out[9])
{
    float a[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[r][c] = m[3 * r + c];
            a[r][3 + c] = (r == c) ? 1.0f : 0.0f;
        }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int p = k;
        for (int r = k + 1; r < 3; ++r)
            if (fabsf(a[r][k]) > fabsf(a[p][k])) p = r;
        if (p != k)
            for (int c = 0; c < 6; ++c) {
                float t = a[k][c];
                a[k][c] = a[p][c];
                a[p][c] = t;
            }
        for (int r = k + 1; r < 3; ++r) {
            const float f = a[r][k] / a[k][k];
            for (int c = k; c < 6; ++c) a[r][c] = a[r][c] - f * a[k][c];
        }
    }
    for (int c = 3; c < 6; ++c)
        for (int r = 2; r >= 0; --r) {
            float v = a[r][c];
            for (int q = r + 1; q < 3; ++q) v = v - a[r][q] * a[q][c];
            a[r][c] = v / a[r][r];
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[3 * r + c] = a[r][3 + c];
}

// One table row pair.  cos/sin are the correctly rounded fp32 values (fp64 evaluation, rounded once): on the host by
// the C library, in the kernel by the device library -- the two agree except for rare double roundings, which is why
// host-resident angle sets take the HOST path (ctpvae_rotate_transforms_host_f32): the rows are then the very bits any
// other host code (the CPU oracle, a NumPy restatement) computes from the same expressions.
__host__ __device__ static void transform_rows(float theta, float hm1, float wm1, float *t8, float *tinv8)
{
    const float ang = -theta;
    const float c = (float)cos((double)ang);
    const float s = (float)sin((double)ang);
    const float cw = c * wm1, sh = s * hm1, sw = s * wm1, ch = c * hm1;
    const float xo = (wm1 - (cw - sh)) / 2.0f;
    const float yo = (hm1 - (sw + ch)) / 2.0f;
    const float t[8] = {c, -s, xo, s, c, yo, 0.0f, 0.0f};
    for (int k = 0; k < 8; ++k) t8[k] = t[k];
    if (tinv8) {
        const float m[9] = {t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], 1.0f};
        float inv[9];
        inv3x3(m, inv);
        for (int k = 0; k < 8; ++k) tinv8[k] = inv[k] / inv[8];
    }
}

// One thread per angle.
__global__ void rotate_transforms_kernel(const float *__restrict__ theta, int A, float hm1, float wm1,
                                         float *__restrict__ T8, float *__restrict__ Tinv8)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    transform_rows(theta[a], hm1, wm1, T8 + 8 * a, Tinv8 ? Tinv8 + 8 * a : nullptr);
}

}  // namespace ctpvae

using namespace ctpvae;

extern "C" {

int ctpvae_abi_version(void) { return CTPVAE_ABI_VERSION; }

int ctpvae_tune_set(const char *name, int value)
{
    CTPVAE_REQUIRE(name != nullptr, "tune_set: null name");
    if (strcmp(name, "*") == 0) {   // reset every knob
        for (int k = 0; k < kKnobCount; ++k) g_knobs[k].store(-1, std::memory_order_relaxed);
        return CTPVAE_OK;
    }
    const int k = find_knob(name);
    CTPVAE_REQUIRE(k >= 0, "tune_set: unknown knob '%s'", name);
    g_knobs[k].store(value < 0 ? -1 : value, std::memory_order_relaxed);
    return CTPVAE_OK;
}

int ctpvae_tune_active(void)
{
    int n = 0;
    for (int k = 0; k < kKnobCount; ++k) n += g_knobs[k].load(std::memory_order_relaxed) >= 0;
    return n;
}

const char *ctpvae_last_error(void) { return err_buf(); }

int ctpvae_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(CTPVAE_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int ctpvae_num_proj_pix(int nx, int ny)
{
    CTPVAE_REQUIRE(nx > 0 && ny > 0, "num_proj_pix: sizes must be positive (got %d, %d)", nx, ny);
    const double v = std::sqrt((double)((long long)nx * nx + (long long)ny * ny)) + 2.0;
    return (int)(std::ceil(v / 2.0) * 2.0);
}

int ctpvae_pad_amounts(int n, int P, int *lo, int *hi)
{
    CTPVAE_REQUIRE(lo && hi, "pad_amounts: null output");
    CTPVAE_REQUIRE(n > 0 && P >= n, "pad_amounts: need 0 < n <= P (got n=%d, P=%d)", n, P);
    *lo = (P - n) / 2;
    *hi = *lo + ((P - n) % 2);
    return CTPVAE_OK;
}

int ctpvae_rotate_transforms_host_f32(const float *theta, int A, int H, int W, float *T8, float *Tinv8)
{
    CTPVAE_REQUIRE(theta && T8, "rotate_transforms_host: null pointer");
    CTPVAE_REQUIRE(A > 0 && H > 0 && W > 0, "rotate_transforms_host: bad sizes A=%d H=%d W=%d", A, H, W);
    for (int a = 0; a < A; ++a)
        transform_rows(theta[a], (float)H - 1.0f, (float)W - 1.0f, T8 + 8 * a, Tinv8 ? Tinv8 + 8 * a : nullptr);
    return CTPVAE_OK;
}

int ctpvae_rotate_transforms_f32(const float *theta_dev, int A, int H, int W, float *T8_dev,
                                 float *Tinv8_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(theta_dev && T8_dev, "rotate_transforms: null pointer");
    CTPVAE_REQUIRE(A > 0 && H > 0 && W > 0, "rotate_transforms: bad sizes A=%d H=%d W=%d", A, H, W);
    const int block = 64;
    hipLaunchKernelGGL(rotate_transforms_kernel, dim3(ceil_div(A, block)), dim3(block), 0,
                       (hipStream_t)stream, theta_dev, A, (float)H - 1.0f, (float)W - 1.0f, T8_dev,
                       Tinv8_dev);
    CTPVAE_LAUNCH_CHECK("rotate_transforms_kernel");
    return CTPVAE_OK;
}

}  // extern "C"
