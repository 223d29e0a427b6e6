// gridrec.hip -- TomoPy's `gridrec` reconstruction (the encoder's default input channel) on gfx950.
//
// Reference call sites: tomopy.recon(..., algorithm='gridrec') at ctvae/helper_functions.py:503 (iradon_all; default
// algorithms ['gridrec'], ctvae/main_ct_vae.py:111-112,122; README.md:80,221), ctvae/helper_functions.py:445-457
// (evaluate_sinogram) and bin/final_merit.py:58,81.  The algorithm itself is TomoPy 1.11.0's libtomo/gridrec/gridrec.c
// [3P-recalled, see oracle/gridrec_oracle.c]: per projection a zero-padded 1-D FFT (two slices ride one complex transform),
// filter x centre phase, convolution of the polar samples onto a pdim x pdim Cartesian frequency grid with a separable
// prolate-spheroidal window (Legendre series, 4 x 4 cells), a 2-D FFT, and the window's correction on the way out.
//
// MI355X form: four launches per call, all deterministic --
//   gridrec_fft_rows_kernel   one workgroup per transform, the whole row in LDS (radix-2, twiddles from a host table); the
//                             first use also packs the slice pair, pads and multiplies by the filter-phase table
//   gridrec_grid_kernel       the convolution as a GATHER: a thread per frequency cell walks the angles in order and, per
//                             angle, the <= 7 samples whose 4 x 4 box can reach it -- the additions of gridrec.c's scatter
//                             loop in gridrec.c's order, without atomics
//   gridrec_fft_rows_kernel   rows, then (strided) columns of H
//   gridrec_copy_kernel       crop, correction table, the two slices out of the real / imaginary parts
// The FFTs are hand-written because a 256..1024-point row fits LDS and the library's own bits are then fixed: same
// butterflies, same twiddles as the oracle's radix-2 transform.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace ctpvae {

constexpr int kGrLtbl = 512;
constexpr float kGrC = 7.0f, kGrLambda = 0.99998546f;
constexpr int kGrNt = 20;

struct cpx {
    float re, im;
};

// gridrec.c legendre(): SUM(coefs[k] * P(2k, x)), three-term recurrence (host; the tables are built once per call)
static float gr_legendre(int n, const float *coefs, float x)
{
    float penult = 1.0f, last = x, newer, y = coefs[0];
    int even = 1, k = 1;
    for (int j = 2; j <= n; ++j) {
        newer = (x * (2 * j - 1) * last - (j - 1) * penult) / j;
        if (even) {
            y += newer * coefs[k];
            even = 0;
            ++k;
        } else {
            even = 1;
        }
        penult = last;
        last = newer;
    }
    return y;
}

static float gr_filter(int name, float x, int j, const float *pars)
{
    const float pi = (float)M_PI;
    switch (name) {
    case 0: return 1.0f;                                                                     // none
    case 1: return j == 0 ? 0.0f : fabsf(2 * x) * (sinf(pi * x) / (pi * x));                 // shepp
    case 2: return fabsf(2 * x) * cosf(pi * x);                                              // cosine
    case 3: return fabsf(2 * x) * 0.5f * (1.0f + cosf(2 * pi * x));                          // hann
    case 4: return fabsf(2 * x) * (0.54f + 0.46f * cosf(2 * pi * x));                        // hamming
    case 5: return fabsf(2 * x);                                                             // ramlak
    case 6: return fabsf(2 * x) * (x <= 0.25f ? (1 - 24 * x * x * (1 - 2 * x)) : (2 * powf(1 - 2 * x, 3)));   // parzen
    default: return fabsf(2 * x) * (1.0f / (1.0f + powf(x / pars[0], 2 * pars[1])));          // butterworth
    }
}

static int gr_pdim(int dx)
{
    int pdim = 16;
    while (pdim < dx) pdim *= 2;
    return pdim;
}

// Tables (built on the host, ctpvae_gridrec_tables_host_f32): twiddles, window, correction, trig, filter x phase.
// Workspace: [C: pairs x dt x pdim2 x 2 cpx][H: pairs x pdim x pdim cpx]
struct GrLayout {
    int pdim, pdim2, pairs;
    long long off_tw, off_wtbl, off_winv, off_trig, off_filphase, tables_bytes, off_c, off_h, bytes;
};
static GrLayout gr_layout(int dy, int dt, int dx)
{
    GrLayout L;
    L.pdim = gr_pdim(dx);
    L.pdim2 = L.pdim / 2;
    L.pairs = (dy + 1) / 2;
    auto up = [](long long v) { return (v + 255) / 256 * 256; };
    L.off_tw = 0;                                                       // pdim / 2 twiddles (cos, sin)
    L.off_wtbl = up(L.off_tw + (long long)L.pdim2 * 8);                 // kGrLtbl + 1 floats
    L.off_winv = up(L.off_wtbl + (kGrLtbl + 1) * 4);                    // pdim - 1 floats
    L.off_trig = up(L.off_winv + (long long)(L.pdim - 1) * 4);          // dt x (cos, sin)
    L.off_filphase = up(L.off_trig + (long long)dt * 8);                // pdim2 cpx
    L.tables_bytes = up(L.off_filphase + (long long)L.pdim2 * 8);
    L.off_c = 0;
    L.off_h = up(L.off_c + (long long)L.pairs * dt * L.pdim2 * 16);
    L.bytes = L.off_h + (long long)L.pairs * L.pdim * L.pdim * 8;
    return L;
}

// ---- FFT of rows held in LDS --------------------------------------------------------------------------------------------
// In-place radix-2 decimation in time on n = 2^log2n points in LDS, blockDim.x = n / 2 threads; tw[m] = (cos, sin)(2 pi m / n);
// sign = -1: e^{-i} kernel, +1: e^{+i}.  The butterflies and their order are those of oracle/gridrec_oracle.c fft1d().
__device__ __forceinline__ void lds_fft(cpx *a, int n, int log2n, int sign, const cpx *__restrict__ tw)
{
    const int t = threadIdx.x;
    for (int s = 1; s <= log2n; ++s) {
        const int half = 1 << (s - 1), k = t & (half - 1), i = ((t >> (s - 1)) << s) + k, j = i + half;
        const cpx w = tw[k << (log2n - s)];
        const float wr = w.re, wi = sign < 0 ? -w.im : w.im;
        const cpx u = a[i], v = a[j];
        const float tr = v.re * wr - v.im * wi, ti = v.re * wi + v.im * wr;
        a[j] = cpx{u.re - tr, u.im - ti};
        a[i] = cpx{u.re + tr, u.im + ti};
        __syncthreads();
    }
}
__device__ __forceinline__ int bitrev(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

// MODE 0: projection rows.  block = (angle p, pair q): packs data[2q][p][:] + i data[2q+1][p][:], zero-pads to pdim, transforms
//         with the e^{+i} kernel (gridrec.c's 1-D transform) and writes, for j = 1 .. pdim2 - 1,
//         C[q][p][j][0] = filphase[j] * F[j],  C[q][p][j][1] = conj(filphase[j]) * F[pdim - j].
// MODE 1: rows of H (stride 1) and MODE 2: columns of H (stride pdim), in place, e^{-i} kernel.
template <int MODE>
__global__ __launch_bounds__(1024) void gridrec_fft_rows_kernel(const float *__restrict__ data, int dy, int dt, int dx,
                                                                GrLayout L, const char *__restrict__ tab,
                                                                char *__restrict__ ws, int log2n)
{
    extern __shared__ float lds_raw[];
    cpx *a = reinterpret_cast<cpx *>(lds_raw);
    const cpx *tw = reinterpret_cast<const cpx *>(tab + L.off_tw);
    const int n = L.pdim, t = threadIdx.x;
    if constexpr (MODE == 0) {
        const int p = blockIdx.x, q = blockIdx.y, s0 = 2 * q;
        const float *r0 = data + ((size_t)s0 * dt + p) * dx;
        const float *r1 = s0 + 1 < dy ? data + ((size_t)(s0 + 1) * dt + p) * dx : nullptr;
        for (int i = t; i < n; i += blockDim.x) {
            cpx v{0.0f, 0.0f};
            if (i < dx) {
                v.re = r0[i];
                v.im = r1 ? r1[i] : 0.0f;
            }
            a[bitrev(i, log2n)] = v;
        }
        __syncthreads();
        lds_fft(a, n, log2n, +1, tw);
        const cpx *filphase = reinterpret_cast<const cpx *>(tab + L.off_filphase);
        cpx *C = reinterpret_cast<cpx *>(ws + L.off_c) + ((size_t)q * dt + p) * L.pdim2 * 2;
        for (int j = t; j < L.pdim2; j += blockDim.x) {
            cpx c1{0.0f, 0.0f}, c2{0.0f, 0.0f};
            if (j >= 1) {
                const cpx f = filphase[j], x = a[j], y = a[n - j];
                c1 = cpx{f.re * x.re - f.im * x.im, f.re * x.im + f.im * x.re};
                c2 = cpx{f.re * y.re + f.im * y.im, f.re * y.im - f.im * y.re};
            }
            C[2 * j] = c1;
            C[2 * j + 1] = c2;
        }
    } else {
        const int r = blockIdx.x, q = blockIdx.y;
        cpx *H = reinterpret_cast<cpx *>(ws + L.off_h) + (size_t)q * n * n;
        const size_t base = MODE == 1 ? (size_t)r * n : (size_t)r, stride = MODE == 1 ? 1 : (size_t)n;
        for (int i = t; i < n; i += blockDim.x) a[bitrev(i, log2n)] = H[base + (size_t)i * stride];
        __syncthreads();
        lds_fft(a, n, log2n, -1, tw);
        for (int i = t; i < n; i += blockDim.x) H[base + (size_t)i * stride] = a[i];
    }
}

// ---- the convolution onto the frequency grid, as a gather ----------------------------------------------------------------
// gridrec.c adds, for p ascending and j = 1 .. pdim2 - 1 ascending, convolv * Cdata1 into H[iu][iv] and convolv * Cdata2 into
// H[pdim - iu][pdim - iv] for every (iu, iv) of the sample's box [ceil(U - 2), floor(U + 2)] x [ceil(V - 2), floor(V + 2)]
// (clipped to 1 .. pdim - 1), U = j cos + M2, V = j sin + M2.  A thread owns one cell X = (iu, iv) and performs exactly the
// additions that reach it, in that order: per angle only the samples near the projection of X (direct) or of pdim - X
// (mirrored) onto the angle's direction can; where one sample reaches X both ways the loop order of gridrec.c decides.
// NQ slice pairs per thread: which samples reach a cell, and with what weights, is geometry -- the same for every pair.
template <int NQ>
__global__ __launch_bounds__(256) void gridrec_grid_kernel(int dt, int nq, GrLayout L, const char *__restrict__ tab, char *__restrict__ ws)
{
    extern __shared__ float lds_raw[];
    float *wtbl_s = lds_raw;                                   // kGrLtbl + 1
    float2 *trig_s = reinterpret_cast<float2 *>(lds_raw + kGrLtbl + 4);   // dt x (cos, sin)
    const float *wtbl = reinterpret_cast<const float *>(tab + L.off_wtbl);
    const float2 *trig = reinterpret_cast<const float2 *>(tab + L.off_trig);
    for (int i = threadIdx.x; i <= kGrLtbl; i += blockDim.x) wtbl_s[i] = wtbl[i];
    for (int i = threadIdx.x; i < dt; i += blockDim.x) trig_s[i] = trig[i];
    __syncthreads();
    const int n = L.pdim, M2 = n / 2, q = blockIdx.y * NQ;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= n * n) return;
    const int iu = cell / n, iv = cell - iu * n;
    cpx *H = reinterpret_cast<cpx *>(ws + L.off_h) + (size_t)q * n * n;
    const size_t hstride = (size_t)n * n, cstride = (size_t)dt * L.pdim2 * 2;
    if (iu < 1 || iv < 1) {                                    // the aliasing row / column stays zero
#pragma unroll
        for (int k = 0; k < NQ; ++k)
            if (q + k < nq) H[k * hstride + cell] = cpx{0.0f, 0.0f};
        return;
    }
    const cpx *C = reinterpret_cast<const cpx *>(ws + L.off_c) + (size_t)q * cstride;
    const cpx *Ck[NQ];                                         // a ragged last group re-reads its first pair (never stored)
#pragma unroll
    for (int k = 0; k < NQ; ++k) Ck[k] = C + (q + k < nq ? k : 0) * cstride;
    const float L2 = 2.0f, tblspcg = 2 * kGrLtbl / 4.0f;
    const float fu = (float)iu, fv = (float)iv, fum = (float)(n - iu), fvm = (float)(n - iv);
    const bool mirror_first = (n - iu < iu) || (n - iu == iu && n - iv < iv);   // pdim - X before X in gridrec.c's loop order
    const float u = (float)(iu - M2), v = (float)(iv - M2);
    float hre[NQ], him[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) hre[k] = him[k] = 0.0f;
    for (int p = 0; p < dt; ++p) {
        const float cs = trig_s[p].x, sn = trig_s[p].y;
        const float tproj = u * cs + v * sn;                   // where X projects onto the angle's direction (samples: j)
        int jlo, jhi;
        if (tproj > 3.5f) {
            jlo = (int)floorf(tproj - 3.5f), jhi = (int)ceilf(tproj + 3.5f);
        } else if (tproj < -3.5f) {
            jlo = (int)floorf(-tproj - 3.5f), jhi = (int)ceilf(-tproj + 3.5f);
        } else {
            jlo = 1, jhi = 8;
        }
        jlo = max(jlo, 1);
        jhi = min(jhi, L.pdim2 - 1);
        const size_t prow = (size_t)p * L.pdim2 * 2;
        for (int j = jlo; j <= jhi; ++j) {
            const float U = j * cs + M2, V = j * sn + M2;
            const float ulo = U - L2, uhi = U + L2, vlo = V - L2, vhi = V + L2;
            const bool direct = fu >= ulo && fu <= uhi && fv >= vlo && fv <= vhi;
            const bool mirror = fum >= ulo && fum <= uhi && fvm >= vlo && fvm <= vhi;
            if (!(direct || mirror)) continue;
            float wd = 0.0f, wm = 0.0f;
            if (direct) wd = wtbl_s[(int)roundf(fabsf(U - fu) * tblspcg)] * wtbl_s[(int)roundf(fabsf(V - fv) * tblspcg)];
            if (mirror) wm = wtbl_s[(int)roundf(fabsf(U - fum) * tblspcg)] * wtbl_s[(int)roundf(fabsf(V - fvm) * tblspcg)];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const cpx c1 = Ck[k][prow + 2 * j], c2 = Ck[k][prow + 2 * j + 1];
                if (mirror && mirror_first) {
                    hre[k] += wm * c2.re;
                    him[k] += wm * c2.im;
                }
                if (direct) {
                    hre[k] += wd * c1.re;
                    him[k] += wd * c1.im;
                }
                if (mirror && !mirror_first) {
                    hre[k] += wm * c2.re;
                    him[k] += wm * c2.im;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NQ; ++k)
        if (q + k < nq) H[k * hstride + cell] = cpx{hre[k], him[k]};
}

// ---- copy-out: the central ngridx x ngridy region (wrap-around order: the image centre sits at H[0][0]) times the window's
// correction; pixel (row k: x, column j: y), rows mirrored as gridrec.c writes them
__global__ __launch_bounds__(256) void gridrec_copy_kernel(int dy, int ngridx, int ngridy, GrLayout L, const char *__restrict__ tab,
                                                           const char *__restrict__ ws, float *__restrict__ recon)
{
    const int q = blockIdx.y, n = L.pdim, M02 = n / 2 - 1;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ngridx * ngridy) return;
    const int k = e / ngridy, j = e - k * ngridy;
    const float *winv = reinterpret_cast<const float *>(tab + L.off_winv);
    const cpx *H = reinterpret_cast<const cpx *>(ws + L.off_h) + (size_t)q * n * n;
    const int iu = (j - ngridy / 2 + n) % n, iv = (k - ngridx / 2 + n) % n;
    // a grid as wide as the padded row (any power-of-two detector width) has a pixel at -n / 2, one step outside the table's
    // 2 M02 + 1 entries: it takes the outermost entry (oracle/gridrec_oracle.c GR_WINV)
    const float corrn_u = winv[min(max(M02 + j - ngridy / 2, 0), 2 * M02)];
    const float corrn = corrn_u * winv[min(max(M02 + k - ngridx / 2, 0), 2 * M02)];
    const cpx h = H[(size_t)iu * n + iv];
    const int s = 2 * q;
    recon[((size_t)s * ngridx + (ngridx - 1 - k)) * ngridy + j] = corrn * h.re;
    if (s + 1 < dy) recon[((size_t)(s + 1) * ngridx + (ngridx - 1 - k)) * ngridy + j] = corrn * h.im;
}

}  // namespace ctpvae

using namespace ctpvae;

extern "C" {

long long ctpvae_gridrec_tables_bytes(int dt, int dx)
{
    if (dt <= 0 || dx <= 0) return fail(CTPVAE_EINVAL, "gridrec_tables_bytes: bad sizes");
    if (gr_pdim(dx) > 2048) return fail(CTPVAE_EINVAL, "gridrec: detector rows of at most 2048 bins (got %d)", dx);
    return gr_layout(1, dt, dx).tables_bytes;
}

long long ctpvae_gridrec_workspace_bytes(int dy, int dt, int dx)
{
    if (dy <= 0 || dt <= 0 || dx <= 0) return fail(CTPVAE_EINVAL, "gridrec_workspace_bytes: bad sizes");
    if (gr_pdim(dx) > 2048) return fail(CTPVAE_EINVAL, "gridrec: detector rows of at most 2048 bins (got %d)", dx);
    return gr_layout(dy, dt, dx).bytes;
}

// The tables of one (angle set, detector width, centre, filter), on the HOST (all pointers host memory): gridrec.c
// set_pswf_tables / set_trig_tables / set_filter_tables in fp32, plus the FFT twiddles -- the bits any host code computes
// from the same expressions (oracle/gridrec_oracle.c).  filter_name: 0 none, 1 shepp, 2 cosine, 3 hann, 4 hamming, 5 ramlak,
// 6 parzen (tomopy's default for gridrec), 7 butterworth (filter_par = {cutoff, order}).
int ctpvae_gridrec_tables_host_f32(int dt, int dx, float center, const float *theta, int filter_name, const float *filter_par,
                                   void *tables)
{
    CTPVAE_REQUIRE(theta && tables && dt > 0 && dx > 0, "gridrec_tables: null pointer or empty sizes");
    CTPVAE_REQUIRE(filter_name >= 0 && filter_name <= 7, "gridrec_tables: unknown filter %d", filter_name);
    CTPVAE_REQUIRE(filter_name != 7 || filter_par, "gridrec_tables: the butterworth filter needs its two parameters");
    const int pdim = gr_pdim(dx);
    CTPVAE_REQUIRE(pdim <= 2048, "gridrec: detector rows of at most 2048 bins (got %d)", dx);
    const GrLayout L = gr_layout(1, dt, dx);
    const int pdim2 = L.pdim2, M02 = pdim / 2 - 1;
    static const float coefs[11] = {0.5767616E+02f, -0.8931343E+02f, 0.4167596E+02f, -0.1053599E+02f, 0.1662374E+01f, -0.1780527E-00f,
                                    0.1372983E-01f, -0.7963169E-03f, 0.3593372E-04f, -0.1295941E-05f, 0.3817796E-07f};
    char *host = (char *)tables;
    memset(host, 0, (size_t)L.tables_bytes);
    cpx *tw = reinterpret_cast<cpx *>(host + L.off_tw);
    for (int m = 0; m < pdim2; ++m) {
        const double ang = 2.0 * M_PI * m / pdim;
        tw[m] = cpx{(float)cos(ang), (float)sin(ang)};
    }
    float *wtbl = reinterpret_cast<float *>(host + L.off_wtbl), *winv = reinterpret_cast<float *>(host + L.off_winv);
    const float fac = (float)kGrLtbl / (M02 + 0.5f);
    const float polyz = gr_legendre(kGrNt, coefs, 0.0f);
    wtbl[0] = 1.0f;
    for (int i = 1; i <= kGrLtbl; ++i) wtbl[i] = gr_legendre(kGrNt, coefs, (float)i / kGrLtbl) / polyz;
    float norm = sqrtf((float)M_PI / 2 / kGrC / kGrLambda) / 1.2f;
    winv[M02] = norm / wtbl[0];
    for (int i = 1; i <= M02; ++i) {
        norm = -norm;
        winv[M02 + i] = winv[M02 - i] = norm / wtbl[(int)roundf(i * fac)];
    }
    float *trig = reinterpret_cast<float *>(host + L.off_trig);
    for (int p = 0; p < dt; ++p) {
        trig[2 * p] = cosf(theta[p]);
        trig[2 * p + 1] = sinf(theta[p]);
    }
    cpx *filphase = reinterpret_cast<cpx *>(host + L.off_filphase);
    const float fnorm = (float)M_PI / pdim / dt, rtmp1 = 2 * (float)M_PI * center / pdim;
    for (int j = 0; j < pdim2; ++j) {
        const float x = j * rtmp1, f = gr_filter(filter_name, (float)j / pdim, j, filter_par) * fnorm;
        filphase[j] = cpx{f * cosf(x), -f * sinf(x)};
    }
    return CTPVAE_OK;
}

// data_dev [dy][dt][dx] (sinogram order) -> recon_dev [dy][ngridx][ngridy]; tables_dev: the device copy of
// ctpvae_gridrec_tables_host_f32's buffer; workspace_dev: ctpvae_gridrec_workspace_bytes() bytes (contents undefined).
int ctpvae_gridrec_f32(const float *data_dev, int dy, int dt, int dx, const void *tables_dev, int ngridx, int ngridy,
                       void *workspace_dev, float *recon_dev, ctpvae_stream_t stream)
{
    CTPVAE_REQUIRE(data_dev && tables_dev && workspace_dev && recon_dev, "gridrec: null pointer");
    CTPVAE_REQUIRE(dy > 0 && dt > 0 && dx > 0 && ngridx > 0 && ngridy > 0, "gridrec: sizes must be positive");
    const int pdim = gr_pdim(dx);
    CTPVAE_REQUIRE(pdim <= 2048, "gridrec: detector rows of at most 2048 bins (got %d)", dx);
    CTPVAE_REQUIRE(ngridx <= pdim && ngridy <= pdim, "gridrec: the grid (%d x %d) must not exceed the padded row (%d)", ngridx,
                   ngridy, pdim);
    CTPVAE_REQUIRE(dt <= 4096, "gridrec: at most 4096 angles");
    const GrLayout L = gr_layout(dy, dt, dx);
    const int pdim2 = L.pdim2;
    int log2n = 0;
    while ((1 << log2n) < pdim) ++log2n;
    const char *tab = (const char *)tables_dev;
    char *ws = (char *)workspace_dev;
    const int fft_threads = pdim / 2;
    const size_t fft_lds = (size_t)pdim * sizeof(cpx);
    for (int q0 = 0; q0 < L.pairs; q0 += 65535) {
        const int nq = std::min(65535, L.pairs - q0);
        GrLayout Lq = L;                                        // this chunk's pairs start at the workspace's q0-th block
        Lq.off_c += (long long)q0 * dt * pdim2 * 16;
        Lq.off_h += (long long)q0 * pdim * pdim * 8;
        const float *dq = data_dev + (size_t)2 * q0 * dt * dx;
        const int dyq = std::min(dy - 2 * q0, 2 * nq);
        hipLaunchKernelGGL(gridrec_fft_rows_kernel<0>, dim3(dt, nq), dim3(fft_threads), fft_lds, (hipStream_t)stream, dq, dyq, dt, dx,
                           Lq, tab, ws, log2n);
        CTPVAE_LAUNCH_CHECK("gridrec_fft_rows_kernel<0>");
        {   // pairs per thread: 5 divides the training set's 25 pairs; enough workgroups must remain to fill the chip
            const size_t shm = (size_t)(kGrLtbl + 4) * 4 + (size_t)dt * 8;
            const int cells = ceil_div(pdim * pdim, 256);
            const int npt = nq % 5 == 0 || nq > 20 ? 5 : nq >= 4 ? 4 : nq >= 2 ? 2 : 1;
            const dim3 grid(cells, ceil_div(nq, npt));
            if (npt == 5)
                hipLaunchKernelGGL(gridrec_grid_kernel<5>, grid, dim3(256), shm, (hipStream_t)stream, dt, nq, Lq, tab, ws);
            else if (npt == 4)
                hipLaunchKernelGGL(gridrec_grid_kernel<4>, grid, dim3(256), shm, (hipStream_t)stream, dt, nq, Lq, tab, ws);
            else if (npt == 2)
                hipLaunchKernelGGL(gridrec_grid_kernel<2>, grid, dim3(256), shm, (hipStream_t)stream, dt, nq, Lq, tab, ws);
            else
                hipLaunchKernelGGL(gridrec_grid_kernel<1>, grid, dim3(256), shm, (hipStream_t)stream, dt, nq, Lq, tab, ws);
        }
        CTPVAE_LAUNCH_CHECK("gridrec_grid_kernel");
        hipLaunchKernelGGL(gridrec_fft_rows_kernel<1>, dim3(pdim, nq), dim3(fft_threads), fft_lds, (hipStream_t)stream, nullptr, dyq, dt,
                           dx, Lq, tab, ws, log2n);
        hipLaunchKernelGGL(gridrec_fft_rows_kernel<2>, dim3(pdim, nq), dim3(fft_threads), fft_lds, (hipStream_t)stream, nullptr, dyq, dt,
                           dx, Lq, tab, ws, log2n);
        CTPVAE_LAUNCH_CHECK("gridrec_fft_rows_kernel<1,2>");
        hipLaunchKernelGGL(gridrec_copy_kernel, dim3(ceil_div(ngridx * ngridy, 256), nq), dim3(256), 0, (hipStream_t)stream, dyq, ngridx,
                           ngridy, Lq, tab, ws, recon_dev + (size_t)2 * q0 * ngridx * ngridy);
        CTPVAE_LAUNCH_CHECK("gridrec_copy_kernel");
    }
    return CTPVAE_OK;
}

}  // extern "C"
