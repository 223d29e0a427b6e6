// poisson.hip -- sparse noisy measurements on the device: the Poisson-noise forward model of
// ctvae/create_masks.py:80-103 as ONE kernel over the dense sinograms the TomoPy-style projector wrote,
//     out[s][a][j] = Poisson( max(sino[s][a][j], 0) * mask[s][a] * pnm ) / pnm            (:32, :82, :94-95)
// -- clamp, dose mask, noise multiplier, the draw and the division fused; nothing but the result touches HBM.
//
// The reference draws with tensorflow-probability's tfd.Poisson(rate).sample() (TensorFlow's stateful generator: the
// DISTRIBUTION is what the reference defines, its bits are unpinnable).  This sampler is counter-based and fully
// specified, so that the CPU twin (oracle/radon_oracle.c, oracle_poisson_measure) reproduces every count exactly:
//
//   element   e = (s * A + a) * P + j                                  (64-bit)
//   rate      lam = (double)( (max(sino[e], 0) * mask[s][a]) * pnm )   (the two products in fp32, as the reference's)
//   uniforms  block t of element e = Philox4x32-10( counter = (e_lo, e_hi, t, 0), key = (seed_lo, seed_hi) ),
//             u_i = (word_i + 0.5) * 2^-32 in double: never 0, never 1
//   lam < 10  multiplication method (Knuth): count the uniforms, taken word by word from blocks t = 0, 1, ..., whose
//             running product stays above exp(-lam)
//   lam >= 10 transformed rejection with squeeze (PTRS, Hormann 1993), iteration t using U = u_0 - 1/2, V = u_1 of block t
//   result    out = (float)count / pnm                                 (fp32 division, as the reference's)
//   exp, log  evaluated by the fixed double-precision series below (+, -, *, / only, compiled with -ffp-contract=off):
//             the same bits on the host and on the device, which library exp / log do not promise.
#include "common.h"

namespace ctpvae {

struct Philox4 {
    unsigned w[4];
};

__host__ __device__ inline void mulhilo32(unsigned a, unsigned b, unsigned &hi, unsigned &lo)
{
    const unsigned long long p = (unsigned long long)a * b;
    hi = (unsigned)(p >> 32);
    lo = (unsigned)p;
}

// Philox4x32 with 10 rounds (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11)
__host__ __device__ inline Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0, lo0, hi1, lo1;
        mulhilo32(0xD2511F53u, c0, hi0, lo0);
        mulhilo32(0xCD9E8D57u, c2, hi1, lo1);
        const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

__host__ __device__ inline double bits_to_double(unsigned long long b)
{
    union {
        unsigned long long u;
        double d;
    } v;
    v.u = b;
    return v.d;
}
__host__ __device__ inline unsigned long long double_to_bits(double d)
{
    union {
        unsigned long long u;
        double d;
    } v;
    v.d = d;
    return v.u;
}

// log(x), x > 0 and normal: x = m * 2^e with m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <=
// 0.1716, by its odd series up to s^23 (truncation < 1e-19), Horner in s^2; then e * ln 2 + log m.
__host__ __device__ inline double det_log(double x)
{
    unsigned long long b = double_to_bits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    b = (b & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = bits_to_double(b);   // [1, 2)
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e += 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

// exp(x) for -746 < x <= 0: k = nearest integer to x / ln 2, r = x - k ln 2 (two-part ln 2), Taylor to r^14, scaled by 2^k
__host__ __device__ inline double det_exp_neg(double x)
{
    if (x < -700.0) return 0.0;
    const double kf = (double)(long long)(x * 1.4426950408889634 - 0.5);   // x <= 0: truncation towards zero of (y - 0.5)
    const double r = (x - kf * 0.693147180369123816490) - kf * 1.90821492927058770002e-10;
    double p = 1.0 / 87178291200.0;      // 1/14!
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    const long long k = (long long)kf;                                     // >= -1011
    return p * bits_to_double((unsigned long long)(k + 1023) << 52);
}

// log(k!) for integer k >= 0: table below 10, Stirling's series from there (error < 1e-13)
__host__ __device__ inline double det_logfact(double k)
{
    if (k < 10.0) {
        const double t[10] = {0.0, 0.0, 0.6931471805599453, 1.791759469228055, 3.1780538303479458, 4.787491742782046,
                              6.579251212010101, 8.525161361065415, 10.60460290274525, 12.801827480081469};
        return t[(int)k];
    }
    const double n = k + 1.0, i = 1.0 / n, i2 = i * i;
    return (n - 0.5) * det_log(n) - n + 0.9189385332046727 +
           i * (1.0 / 12.0 - i2 * (1.0 / 360.0 - i2 * (1.0 / 1260.0 - i2 * (1.0 / 1680.0))));
}

__host__ __device__ inline double u01(unsigned w) { return ((double)w + 0.5) * 2.3283064365386963e-10; }

// one Poisson(lam) count for element `e` under `seed`
__host__ __device__ inline double poisson_count(double lam, unsigned long long e, unsigned long long seed)
{
    const unsigned e0 = (unsigned)e, e1 = (unsigned)(e >> 32), k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    if (!(lam > 0.0)) return 0.0;
    if (lam < 10.0) {
        const double enlam = det_exp_neg(-lam);
        double prod = 1.0, count = 0.0;
        for (unsigned t = 0;; ++t) {
            const Philox4 b = philox4x32_10(e0, e1, t, 0u, k0, k1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                prod = prod * u01(b.w[i]);
                if (!(prod > enlam)) return count;
                count = count + 1.0;
            }
        }
    }
    // Hormann's PTRS
    const double slam = sqrt(lam), loglam = det_log(lam);
    const double bb = 0.931 + 2.53 * slam;
    const double a = -0.059 + 0.02483 * bb;
    const double invalpha = 1.1239 + 1.1328 / (bb - 3.4);
    const double vr = 0.9277 - 3.6224 / (bb - 2.0);
    for (unsigned t = 0;; ++t) {
        const Philox4 b = philox4x32_10(e0, e1, t, 0u, k0, k1);
        const double U = u01(b.w[0]) - 0.5, V = u01(b.w[1]);
        const double us = 0.5 - fabs(U);
        const double k = floor((2.0 * a / us + bb) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        if (det_log(V) + det_log(invalpha) - det_log(a / (us * us) + bb) <= -lam + k * loglam - det_logfact(k)) return k;
        if (t == 0xffffffffu) return k;   // unreachable in practice; bounds the loop
    }
}

__global__ __launch_bounds__(256) void poisson_measure_kernel(const float *__restrict__ sino, const float *__restrict__ mask,
                                                             long long n, int P, float pnm, unsigned long long seed,
                                                             float *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const float loc = fmaxf(sino[e], 0.0f) * mask[e / P];
        const float rate = loc * pnm;
        float v;
        if (rate < 1.0e15f)
            v = (float)poisson_count((double)rate, (unsigned long long)e, seed) / pnm;
        else
            v = loc;   // beyond any count a float can tell from its neighbours (also inf / NaN rates): no noise to add
        out[e] = v;
    }
}

}  // namespace ctpvae

using namespace ctpvae;

extern "C" {

int ctpvae_poisson_measure_f32(const float *sino_dev, const float *mask_dev, int S, int A, int P, float pnm,
                               unsigned long long seed, float *out_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(sino_dev && mask_dev && out_dev, "poisson_measure: null pointer");
    CTPVAE_REQUIRE(S > 0 && A > 0 && P > 0, "poisson_measure: sizes must be positive (S=%d A=%d P=%d)", S, A, P);
    CTPVAE_REQUIRE(pnm > 0.0f, "poisson_measure: the noise multiplier must be positive (got %g)", (double)pnm);
    const long long n = (long long)S * A * P;
    const int block = 256;
    const long long want = (n + block - 1) / block;
    const unsigned grid = (unsigned)std::min<long long>(want, 256ll * 64);   // grid-stride: every wave exits
    hipLaunchKernelGGL(poisson_measure_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, sino_dev, mask_dev, n, P, pnm,
                       seed, out_dev);
    CTPVAE_LAUNCH_CHECK("poisson_measure_kernel");
    return CTPVAE_OK;
}

// Host twin of the generator's building block, for known-answer tests of the Philox rounds (tests/test_abi.py): writes the
// four words of Philox4x32-10(counter, key).
int ctpvae_philox4x32_10(const unsigned *counter4, const unsigned *key2, unsigned *out4)
{
    CTPVAE_REQUIRE(counter4 && key2 && out4, "philox4x32_10: null pointer");
    const Philox4 b = philox4x32_10(counter4[0], counter4[1], counter4[2], counter4[3], key2[0], key2[1]);
    for (int i = 0; i < 4; ++i) out4[i] = b.w[i];
    return CTPVAE_OK;
}

}  // extern "C"
