/*
 * gridrec_oracle.c -- CPU restatement of TomoPy's `gridrec` reconstruction.  TEST INFRASTRUCTURE ONLY (see radon_oracle.c).
 *
 * Reference call sites (paths relative to the vganapati/CT_PVAE root):
 *   ctvae/helper_functions.py:503     tomopy.recon(proj_sample_expand, theta, center=None, sinogram_order=True,
 *                                                  algorithm=algorithm)      -- algorithms default ['gridrec'],
 *   ctvae/main_ct_vae.py:111-112,122  README.md:80 (--algorithms gridrec), README.md:221 (sirt tv fbp gridrec)
 *   ctvae/helper_functions.py:445-457 evaluate_sinogram: tomopy.recon(..., algorithm='gridrec', sinogram_order=False)
 *   bin/final_merit.py:58,81
 *
 * PARITY UNPINNED, [3P-recalled].  The algorithm lives in TomoPy 1.11.0 (environment.yml:6), which is not in /root/reference
 * and cannot be installed here: libtomo/gridrec/gridrec.c (Dowd et al. 1999 "gridrec": 1-D FFT of every projection, filter
 * x centre phase, convolution onto a Cartesian frequency grid with a prolate-spheroidal-wave-function (PSWF) window built
 * from a Legendre expansion, 2-D inverse FFT, division by the window's transform) and tomopy/recon/algorithm.py (per-
 * algorithm defaults: gridrec's filter_name is 'parzen', fbp's is 'none'; grid = detector width; center None = width / 2).
 * Everything below is written from memory of that source and must be read as a specification by recollection.  What pins it
 * here: (1) reconstruction of a ray-driven projection (oracle_siddon_project) of an asymmetric phantom returns the phantom
 * in place -- orientation, centre and scale (tests/test_oracle.py); (2) agreement to a few per cent with the independently
 * written ramp-filtered back-projection on the same grid; (3) the PSWF window is 1 at 0, decays monotonically, and its
 * correction table is symmetric with alternating sign.
 *
 * FFTs: gridrec.c calls FFTW (or MKL); the bits of those libraries' butterflies are not restated -- a plain radix-2 float
 * FFT stands in (results agree to fp32 rounding).  Compile with -ffp-contract=off like radon_oracle.c.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    float re, im;
} cpx;

/* in-place radix-2 decimation-in-time FFT, n a power of two; sign = -1: forward (e^{-i...}), +1: backward, unnormalised */
static void fft1d(cpx *a, int n, int sign)
{
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            cpx t = a[i];
            a[i] = a[j];
            a[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        for (int k = 0; k < len / 2; ++k) {
            const double ang = 2.0 * M_PI * k / len;
            const float wr = (float)cos(ang), wi = (float)(sign * sin(ang));
            for (int i = k; i < n; i += len) {
                cpx *u = a + i, *v = a + i + len / 2;
                const float tr = v->re * wr - v->im * wi, ti = v->re * wi + v->im * wr;
                v->re = u->re - tr;
                v->im = u->im - ti;
                u->re = u->re + tr;
                u->im = u->im + ti;
            }
        }
    }
}

/* gridrec.c legendre(): SUM(coefs[k] * P(2k, x), k = 0 .. n/2), P the Legendre polynomials, by the three-term recurrence */
static float legendre(int n, const float *coefs, float x)
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

#define GR_LTBL 512
static const float kC = 7.0f, kLambda = 0.99998546f;
static const int kNt = 20;
static const float kCoefs[11] = {0.5767616E+02f, -0.8931343E+02f, 0.4167596E+02f, -0.1053599E+02f, 0.1662374E+01f, -0.1780527E-00f,
                                 0.1372983E-01f, -0.7963169E-03f, 0.3593372E-04f, -0.1295941E-05f, 0.3817796E-07f};

/* gridrec.c set_pswf_tables(): wtbl[0..ltbl] the convolvent on [0, 1]; winv[0..2 linv] the final correction (reciprocal of
 * the convolvent's transform, the 2-D FFT's normalisation "norm^2 ... hacked" by 1.2, alternating sign for the centred H) */
void oracle_gridrec_pswf_tables(int linv, float *wtbl, float *winv)
{
    const float fac = (float)GR_LTBL / (linv + 0.5f);
    const float polyz = legendre(kNt, kCoefs, 0.0f);
    wtbl[0] = 1.0f;
    for (int i = 1; i <= GR_LTBL; ++i) wtbl[i] = legendre(kNt, kCoefs, (float)i / GR_LTBL) / polyz;
    float norm = sqrtf((float)M_PI / 2 / kC / kLambda) / 1.2f;
    winv[linv] = norm / wtbl[0];
    for (int i = 1; i <= linv; ++i) {
        norm = -norm;
        winv[linv + i] = winv[linv - i] = norm / wtbl[(int)roundf(i * fac)];
    }
}

/* gridrec.c filter functions: x = j / pdim in [0, 0.5); every one but 'none' carries the ramp |2 x| */
#define GR_NONE 0
#define GR_SHEPP 1
#define GR_COSINE 2
#define GR_HANN 3
#define GR_HAMMING 4
#define GR_RAMLAK 5
#define GR_PARZEN 6
#define GR_BUTTERWORTH 7
float oracle_gridrec_filter(int name, float x, int j, const float *pars)
{
    switch (name) {
    case GR_NONE: return 1.0f;
    case GR_SHEPP: return j == 0 ? 0.0f : fabsf(2 * x) * (sinf((float)M_PI * x) / ((float)M_PI * x));
    case GR_COSINE: return fabsf(2 * x) * cosf((float)M_PI * x);
    case GR_HANN: return fabsf(2 * x) * 0.5f * (1.0f + cosf(2 * (float)M_PI * x));
    case GR_HAMMING: return fabsf(2 * x) * (0.54f + 0.46f * cosf(2 * (float)M_PI * x));
    case GR_RAMLAK: return fabsf(2 * x);
    case GR_PARZEN: return fabsf(2 * x) * (x <= 0.25f ? (1 - 24 * x * x * (1 - 2 * x)) : (2 * powf(1 - 2 * x, 3)));
    default: return fabsf(2 * x) * (1.0f / (1.0f + powf(x / pars[0], 2 * pars[1])));
    }
}

/* pdim = the power of two >= dx (at least 16), gridrec.c: for (pdim = 16; pdim < dx; pdim *= 2); */
int oracle_gridrec_pdim(int dx)
{
    int pdim = 16;
    while (pdim < dx) pdim *= 2;
    return pdim;
}

/* gridrec(): data [dy][dt][dx] (sinogram order), theta [dt], one centre for all slices -> recon [dy][ngridx][ngridy].
 * Two slices ride one complex transform (real part: slice s, imaginary part: slice s + 1). */
int oracle_gridrec(const float *data, int dy, int dt, int dx, float center, const float *theta, int ngridx, int ngridy,
                   int filter_name, const float *filter_par, float *recon)
{
    const int pdim = oracle_gridrec_pdim(dx), pdim2 = pdim / 2, M2 = pdim / 2, M02 = pdim / 2 - 1;
    const float L2 = (float)((int)(2 * kC / (float)M_PI)) / 2;     /* L = 4 grid cells, L2 = 2 */
    const float tblspcg = 2 * GR_LTBL / (2 * L2);
    if (ngridx > pdim || ngridy > pdim) return -1;
    float *wtbl = malloc(sizeof(float) * (GR_LTBL + 1)), *winv = malloc(sizeof(float) * (pdim - 1));
    float *sine = malloc(sizeof(float) * dt), *cose = malloc(sizeof(float) * dt);
    cpx *sino = malloc(sizeof(cpx) * pdim), *filphase = malloc(sizeof(cpx) * pdim2), *H = malloc(sizeof(cpx) * pdim * pdim);
    cpx *col = malloc(sizeof(cpx) * pdim);
    float work[8];
    oracle_gridrec_pswf_tables(M02, wtbl, winv);
    for (int p = 0; p < dt; ++p) {   /* set_trig_tables */
        sine[p] = sinf(theta[p]);
        cose[p] = cosf(theta[p]);
    }
    {   /* set_filter_tables(): filter x phase shifting the origin to the rotation axis, x pi / pdim / dt */
        const float norm = (float)M_PI / pdim / dt, rtmp1 = 2 * (float)M_PI * center / pdim;
        for (int j = 0; j < pdim2; ++j) {
            const float x = j * rtmp1, f = oracle_gridrec_filter(filter_name, (float)j / pdim, j, filter_par) * norm;
            filphase[j].re = f * cosf(x);
            filphase[j].im = -f * sinf(x);
        }
    }
    for (int s = 0; s < dy; s += 2) {
        memset(H, 0, sizeof(cpx) * pdim * pdim);
        for (int p = 0; p < dt; ++p) {
            const float *r0 = data + ((size_t)s * dt + p) * dx, *r1 = s + 1 < dy ? data + ((size_t)(s + 1) * dt + p) * dx : NULL;
            for (int j = 0; j < dx; ++j) {
                sino[j].re = r0[j];
                sino[j].im = r1 ? r1[j] : 0.0f;
            }
            memset(sino + dx, 0, sizeof(cpx) * (pdim - dx));
            fft1d(sino, pdim, +1);   /* gridrec.c's 1-D transform has the e^{+i} kernel (four1(..., 1) in its first form) */
            for (int j = 1; j < pdim2; ++j) {
                const cpx f = filphase[j], a = sino[j], b = sino[pdim - j];
                const cpx c1 = {f.re * a.re - f.im * a.im, f.re * a.im + f.im * a.re};          /* filphase[j] * sino[j] */
                const cpx c2 = {f.re * b.re + f.im * b.im, f.re * b.im - f.im * b.re};          /* conj(filphase[j]) * sino[pdim - j] */
                const float U = j * cose[p] + M2, V = j * sine[p] + M2;
                int iul = (int)ceilf(U - L2), iuh = (int)floorf(U + L2), ivl = (int)ceilf(V - L2), ivh = (int)floorf(V + L2);
                if (iul < 1) iul = 1;
                if (iuh >= pdim) iuh = pdim - 1;
                if (ivl < 1) ivl = 1;
                if (ivh >= pdim) ivh = pdim - 1;
                for (int iv = ivl, k = 0; iv <= ivh; ++iv, ++k) work[k] = wtbl[(int)roundf(fabsf(V - iv) * tblspcg)];
                for (int iu = iul; iu <= iuh; ++iu) {
                    const float rtmp = wtbl[(int)roundf(fabsf(U - iu) * tblspcg)];
                    for (int iv = ivl, k = 0; iv <= ivh; ++iv, ++k) {
                        const float convolv = rtmp * work[k];
                        cpx *h1 = H + (size_t)iu * pdim + iv, *h2 = H + (size_t)(pdim - iu) * pdim + (pdim - iv);
                        h1->re += convolv * c1.re;
                        h1->im += convolv * c1.im;
                        h2->re += convolv * c2.re;
                        h2->im += convolv * c2.im;
                    }
                }
            }
        }
        /* 2-D FFT with the e^{-i} kernel (fourn(..., -1)), unnormalised: rows, then columns */
        for (int r = 0; r < pdim; ++r) fft1d(H + (size_t)r * pdim, pdim, -1);
        for (int c = 0; c < pdim; ++c) {
            for (int r = 0; r < pdim; ++r) col[r] = H[(size_t)r * pdim + c];
            fft1d(col, pdim, -1);
            for (int r = 0; r < pdim; ++r) H[(size_t)r * pdim + c] = col[r];
        }
        /* copy the central ngridx x ngridy region (wrap-around order: the image centre sits at H[0][0]) with the window
         * correction; output pixel (k, j): row index k counts x, column index j counts y */
        /* (a grid as wide as the padded row -- any power-of-two detector -- has a pixel at -pdim / 2, one step outside the
         * correction table's 2 M02 + 1 entries: it takes the table's outermost entry, GR_WINV()) */
#define GR_WINV(i) winv[(i) < 0 ? 0 : ((i) > 2 * M02 ? 2 * M02 : (i))]
        for (int j = 0; j < ngridy; ++j) {
            const int iu = (j - ngridy / 2 + pdim) % pdim;
            const float corrn_u = GR_WINV(M02 + j - ngridy / 2);
            for (int k = 0; k < ngridx; ++k) {
                const int iv = (k - ngridx / 2 + pdim) % pdim;
                const float corrn = corrn_u * GR_WINV(M02 + k - ngridx / 2);
                const cpx h = H[(size_t)iu * pdim + iv];
                recon[((size_t)s * ngridx + (ngridx - 1 - k)) * ngridy + j] = corrn * h.re;
                if (s + 1 < dy) recon[((size_t)(s + 1) * ngridx + (ngridx - 1 - k)) * ngridy + j] = corrn * h.im;
            }
        }
    }
#undef GR_WINV
    free(wtbl), free(winv), free(sine), free(cose), free(sino), free(filphase), free(H), free(col);
    return 0;
}
