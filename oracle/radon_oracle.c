/*
 * radon_oracle.c -- CPU restatement of CT_PVAE's Radon hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product path (ct_pvae_amd/) never does.
 *
 * PARITY UNPINNED.  The reference (vganapati/CT_PVAE) is pure Python on top of TensorFlow 2.8.1,
 * tensorflow-addons 0.17.1, tensorflow-probability 0.14.0 and TomoPy 1.11.0.  None of those is
 * installed (or installable) here and the reference holds no tests, fixtures or golden files
 * for this path.  The arithmetic below therefore restates the published algorithms of those
 * pinned third-party versions from their call sites in the reference; what pins it is
 *   - the 2x2 toy known answers in scripts/images_to_sinograms.py:54-59 with the images of
 *     scripts/create_toy_images.py:36-40,
 *   - the size identities of ctvae/forward_functions.py:29-36 and ctvae/main_ct_vae.py:160-161,
 *   - axis-aligned analytic cases and structural properties (tests/test_oracle.py).
 *
 * Every function cites the reference file:line whose behaviour it follows.  All citations are
 * relative to the reference repo root.  Compile with -ffp-contract=off: the index arithmetic of
 * the nearest-neighbour projector is only meaningful with separate multiplies and adds.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_NEAREST 0
#define ORACLE_BILINEAR 1

/* ---------------------------------------------------------------------------------------------
 * a1: pad_phantom size rule.  ctvae/forward_functions.py:29-36
 *   num_proj_pix = sqrt(float64(Nx^2 + Ny^2)) + 2 ; P = int(ceil(num_proj_pix / 2) * 2)
 *   pad_lo = (P - N) // 2 ; pad_hi = pad_lo + (P - N) % 2
 * ------------------------------------------------------------------------------------------- */
int oracle_num_proj_pix(int nx, int ny)
{
    double v = sqrt((double)((long long)nx * nx + (long long)ny * ny)) + 2.0;
    return (int)(ceil(v / 2.0) * 2.0);
}

void oracle_pad_amounts(int n, int P, int *lo, int *hi)
{
    *lo = (P - n) / 2;
    *hi = *lo + ((P - n) % 2);
}

/* a1: materialised zero padding, [S][H][W] -> [S][PH][PW].  ctvae/forward_functions.py:38-45 */
void oracle_pad_phantom(const float *img, int S, int H, int W, int py, int px, int PH, int PW,
                        float *out)
{
    memset(out, 0, (size_t)S * PH * PW * sizeof(float));
    for (int s = 0; s < S; ++s)
        for (int r = 0; r < H; ++r)
            memcpy(out + ((size_t)s * PH + (r + py)) * PW + px, img + ((size_t)s * H + r) * W,
                   (size_t)W * sizeof(float));
}

/* ---------------------------------------------------------------------------------------------
 * a3: tfa.image.rotate -> angles_to_projective_transforms (tensorflow-addons 0.17.1), called at
 * ctvae/forward_functions.py:70-74,113 with angles = -theta.
 *   row = [cos, -sin, x_off, sin, cos, y_off, 0, 0]
 *   x_off = ((W-1) - (cos*(W-1) - sin*(H-1))) / 2 ; y_off = ((H-1) - (sin*(W-1) + cos*(H-1))) / 2
 * all in fp32.  `angles` holds the angle handed to rotate (i.e. already -theta, fp32).  cos/sin are
 * the correctly rounded fp32 values (double libm, then rounded) so that every host agrees.
 * ------------------------------------------------------------------------------------------- */
void oracle_rotate_transforms(const float *angles, int A, int H, int W, float *T8)
{
    const float wm1 = (float)W - 1.0f, hm1 = (float)H - 1.0f;
    for (int a = 0; a < A; ++a) {
        const float c = (float)cos((double)angles[a]);
        const float s = (float)sin((double)angles[a]);
        float *t = T8 + 8 * a;
        const float cw = c * wm1, sh = s * hm1, sw = s * wm1, ch = c * hm1;
        const float xo = (wm1 - (cw - sh)) / 2.0f;
        const float yo = (hm1 - (sw + ch)) / 2.0f;
        t[0] = c;  t[1] = -s; t[2] = xo;
        t[3] = s;  t[4] = c;  t[5] = yo;
        t[6] = 0.0f; t[7] = 0.0f;
    }
}

/* ---------------------------------------------------------------------------------------------
 * a4: the transform TensorFlow's registered gradient of ImageProjectiveTransformV3 applies to the
 * incoming gradient: flat -> 3x3, fp32 matrix_inverse (LU with partial pivoting), 3x3 -> flat with
 * division by the [2][2] element.  Reached through tf.GradientTape at ctvae/main_ct_vae.py:471-481.
 * ------------------------------------------------------------------------------------------- */
static void inv3x3_f32(const float m[9], float out[9])
{
    float a[3][6];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            a[r][c] = m[3 * r + c];
            a[r][3 + c] = (r == c) ? 1.0f : 0.0f;
        }
    for (int k = 0; k < 3; ++k) {
        int p = k;
        for (int r = k + 1; r < 3; ++r)
            if (fabsf(a[r][k]) > fabsf(a[p][k])) p = r;
        if (p != k)
            for (int c = 0; c < 6; ++c) { float t = a[k][c]; a[k][c] = a[p][c]; a[p][c] = t; }
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

void oracle_invert_transforms(const float *T8, int A, float *Tinv8)
{
    for (int a = 0; a < A; ++a) {
        const float *t = T8 + 8 * a;
        float m[9] = { t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], 1.0f }, inv[9];
        inv3x3_f32(m, inv);
        for (int k = 0; k < 8; ++k) Tinv8[8 * a + k] = inv[k] / inv[8];
    }
}

/* ---------------------------------------------------------------------------------------------
 * a3: one sample of TensorFlow 2.8.1's ImageProjectiveTransformV3 (ProjectiveGenerator, CPU
 * functor), fill_mode CONSTANT, fill_value 0.  `img` is the UNPADDED H x W core that sits at
 * (py, px) inside a PH x PW zero canvas (reading the canvas is the same as reading the zero fill).
 * ------------------------------------------------------------------------------------------- */
static inline float canvas_read(const float *img, int H, int W, int py, int px, long iy, long ix)
{
    const long r = iy - py, c = ix - px;
    return (r >= 0 && r < H && c >= 0 && c < W) ? img[r * W + c] : 0.0f;
}

static inline void map_coord(const float *t, int ox, int oy, float *x, float *y)
{
    /* projection = t6*x + t7*y + 1 == 1 for rotations; the division is exact */
    *x = (t[0] * (float)ox + t[1] * (float)oy) + t[2];
    *y = (t[3] * (float)ox + t[4] * (float)oy) + t[5];
}

static inline float sample_canvas(const float *img, int H, int W, int py, int px, float y, float x,
                                  int interp)
{
    if (interp == ORACLE_NEAREST)
        return canvas_read(img, H, W, py, px, (long)roundf(y), (long)roundf(x));
    const float yf = floorf(y), xf = floorf(x);
    const float yc = yf + 1.0f, xc = xf + 1.0f;
    const float v_yf = (xc - x) * canvas_read(img, H, W, py, px, (long)yf, (long)xf) +
                       (x - xf) * canvas_read(img, H, W, py, px, (long)yf, (long)xc);
    const float v_yc = (xc - x) * canvas_read(img, H, W, py, px, (long)yc, (long)xf) +
                       (x - xf) * canvas_read(img, H, W, py, px, (long)yc, (long)xc);
    return (yc - y) * v_yf + (y - yf) * v_yc;
}

/* ---------------------------------------------------------------------------------------------
 * a2 / a5: rotate-and-sum forward.  ctvae/forward_functions.py:106-114 (fast, NEAREST by default)
 * and :69-77 (low_mem, BILINEAR).  sino[s][a][j] = sum_{i=0}^{PH-1} Rot_a(canvas_s)[i][j], rows
 * summed in order (reduce_sum over axis 1 = image rows).  img [S][H][W], sino [S][A][PW].
 * ------------------------------------------------------------------------------------------- */
void oracle_rotate_fwd(const float *img, int S, int H, int W, int PH, int PW, int py, int px,
                       const float *T8, int A, int interp, float *sino)
{
    for (int s = 0; s < S; ++s) {
        const float *im = img + (size_t)s * H * W;
        for (int a = 0; a < A; ++a) {
            const float *t = T8 + 8 * a;
            float *out = sino + ((size_t)s * A + a) * PW;
            for (int j = 0; j < PW; ++j) {
                float acc = 0.0f;
                for (int i = 0; i < PH; ++i) {
                    float x, y;
                    map_coord(t, j, i, &x, &y);
                    acc += sample_canvas(im, H, W, py, px, y, x, interp);
                }
                out[j] = acc;
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * a2 / a5 on FLOAT64 pixels (ctvae/tomopy_forward_compare.py:52,56 feeds xdesign's float64 phantoms to
 * project_tf_fast and project_tf_low_mem).  TensorFlow 2.8.1's ImageProjectiveTransformV3<T = double>
 * keeps the coordinates and the interpolation weights in fp32 (ProjectiveGenerator: `const float
 * input_x = ...`; bilinear_interpolation: `static_cast<T>(x_ceil - x) * value`) and multiplies, adds and
 * (reduce_sum) row-sums in T.
 * ------------------------------------------------------------------------------------------- */
static inline double canvas_read_f64(const double *img, int H, int W, int py, int px, long iy, long ix)
{
    const long r = iy - py, c = ix - px;
    return (r >= 0 && r < H && c >= 0 && c < W) ? img[r * W + c] : 0.0;
}

void oracle_rotate_fwd_f64(const double *img, int S, int H, int W, int PH, int PW, int py, int px,
                           const float *T8, int A, int interp, double *sino)
{
    for (int s = 0; s < S; ++s) {
        const double *im = img + (size_t)s * H * W;
        for (int a = 0; a < A; ++a) {
            const float *t = T8 + 8 * a;
            double *out = sino + ((size_t)s * A + a) * PW;
            for (int j = 0; j < PW; ++j) {
                double acc = 0.0;
                for (int i = 0; i < PH; ++i) {
                    float x, y;
                    map_coord(t, j, i, &x, &y);
                    if (interp == ORACLE_NEAREST) {
                        acc += canvas_read_f64(im, H, W, py, px, (long)roundf(y), (long)roundf(x));
                    } else {
                        const float yf = floorf(y), xf = floorf(x);
                        const float yc = yf + 1.0f, xc = xf + 1.0f;
                        const double v_yf = (double)(xc - x) * canvas_read_f64(im, H, W, py, px, (long)yf, (long)xf) +
                                            (double)(x - xf) * canvas_read_f64(im, H, W, py, px, (long)yf, (long)xc);
                        const double v_yc = (double)(xc - x) * canvas_read_f64(im, H, W, py, px, (long)yc, (long)xf) +
                                            (double)(x - xf) * canvas_read_f64(im, H, W, py, px, (long)yc, (long)xc);
                        acc += (double)(yc - y) * v_yf + (double)(y - yf) * v_yc;
                    }
                }
                out[j] = acc;
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * a2 / a5 with the tile-blocked association of the row sum.  reduce_sum(axis=1)
 * (ctvae/forward_functions.py:76,108,114) fixes the terms, not the order in which fp32 adds them; the
 * build's kernels for slices larger than LDS add the samples of each th x tw tile of the slice first
 * (rows ascending) and then the tile sums in ascending row-major tile order:
 *   sino[s][a][j] = ((0 + p_0) + p_1) + ...,   p_t = sum_{i ascending, sample in tile t} sample(i, j).
 * NEAREST: a sample belongs to the tile that holds its tap.  BILINEAR (round 5): to the tile that holds
 * its FLOOR tap (yf, xf), taps at -1 counting to the first row / column of tiles; the sample itself is
 * TensorFlow's four-tap expression, unchanged (sample_canvas).  Samples whose taps all lie outside the
 * slice are exact zeros and belong to no tile.  Same terms as oracle_rotate_fwd; the two differ by fp32
 * rounding of the sum only.
 * ------------------------------------------------------------------------------------------- */
int oracle_rotate_fwd_tiled_interp(const float *img, int S, int H, int W, int PH, int PW, int py, int px,
                                   const float *T8, int A, int interp, int th, int tw, float *sino)
{
    if (th <= 0 || tw <= 0) return -1;
    const int ntx = (W + tw - 1) / tw, nty = (H + th - 1) / th, nt = ntx * nty;
    float *part = (float *)malloc((size_t)nt * sizeof(float));
    if (!part) return -1;
    for (int s = 0; s < S; ++s) {
        const float *im = img + (size_t)s * H * W;
        for (int a = 0; a < A; ++a) {
            const float *t = T8 + 8 * a;
            float *out = sino + ((size_t)s * A + a) * PW;
            for (int j = 0; j < PW; ++j) {
                for (int k = 0; k < nt; ++k) part[k] = 0.0f;
                for (int i = 0; i < PH; ++i) {
                    float x, y;
                    map_coord(t, j, i, &x, &y);
                    if (interp == ORACLE_NEAREST) {
                        const long r = (long)roundf(y) - py, c = (long)roundf(x) - px;
                        if (r >= 0 && r < H && c >= 0 && c < W) part[(r / th) * ntx + c / tw] += im[r * W + c];
                    } else {
                        const long r = (long)floorf(y) - py, c = (long)floorf(x) - px;
                        if (r >= -1 && r < H && c >= -1 && c < W)
                            part[((r < 0 ? 0 : r) / th) * ntx + (c < 0 ? 0 : c) / tw] += sample_canvas(im, H, W, py, px, y, x, interp);
                    }
                }
                float acc = 0.0f;
                for (int k = 0; k < nt; ++k) acc += part[k];
                out[j] = acc;
            }
        }
    }
    free(part);
    return 0;
}

int oracle_rotate_fwd_tiled(const float *img, int S, int H, int W, int PH, int PW, int py, int px,
                            const float *T8, int A, int th, int tw, float *sino)
{
    return oracle_rotate_fwd_tiled_interp(img, S, H, W, PH, PW, py, px, T8, A, ORACLE_NEAREST, th, tw, sino);
}

/* ---------------------------------------------------------------------------------------------
 * a4: the backward TensorFlow actually runs for a2 (tf.GradientTape, ctvae/main_ct_vae.py:471-481):
 *   reduce_sum(axis=1)  <-> broadcast g[a][j] over all rows of a PH x PW image,
 *   ImageProjectiveTransformV3 <-> the same op on that image with the inverted transform, same
 *     interpolation, zero fill,
 *   repeat <-> sum over angles (in order), pad <-> crop of the H x W core.
 * gsino [S][A][PW], Tinv8 from oracle_invert_transforms, gimg [S][H][W].
 * ------------------------------------------------------------------------------------------- */
static inline float bcast_read(const float *grow, int PH, int PW, long iy, long ix)
{
    return (iy >= 0 && iy < PH && ix >= 0 && ix < PW) ? grow[ix] : 0.0f;
}

void oracle_rotate_bwd_tfcompat(const float *gsino, int S, int A, int PH, int PW, const float *Tinv8,
                                int interp, int H, int W, int py, int px, float *gimg)
{
    for (int s = 0; s < S; ++s)
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < W; ++c) {
                float acc = 0.0f;
                for (int a = 0; a < A; ++a) {
                    const float *t = Tinv8 + 8 * a;
                    const float *grow = gsino + ((size_t)s * A + a) * PW;
                    float x, y, v;
                    map_coord(t, c + px, r + py, &x, &y);
                    if (interp == ORACLE_NEAREST) {
                        v = bcast_read(grow, PH, PW, (long)roundf(y), (long)roundf(x));
                    } else {
                        const float yf = floorf(y), xf = floorf(x);
                        const float yc = yf + 1.0f, xc = xf + 1.0f;
                        const float v_yf = (xc - x) * bcast_read(grow, PH, PW, (long)yf, (long)xf) +
                                           (x - xf) * bcast_read(grow, PH, PW, (long)yf, (long)xc);
                        const float v_yc = (xc - x) * bcast_read(grow, PH, PW, (long)yc, (long)xf) +
                                           (x - xf) * bcast_read(grow, PH, PW, (long)yc, (long)xc);
                        v = (yc - y) * v_yf + (y - yf) * v_yc;
                    }
                    acc += v;
                }
                gimg[((size_t)s * H + r) * W + c] = acc;
            }
}

/* ---------------------------------------------------------------------------------------------
 * K2x (not in the reference): the exact transpose of oracle_rotate_fwd, as a scatter in (a, i, j)
 * order.  <fwd(x), g> == <x, bwd_exact(g)> up to fp32 summation order.
 * ------------------------------------------------------------------------------------------- */
static inline void canvas_add(float *gimg, int H, int W, int py, int px, long iy, long ix, float v)
{
    const long r = iy - py, c = ix - px;
    if (r >= 0 && r < H && c >= 0 && c < W) gimg[r * W + c] += v;
}

void oracle_rotate_bwd_exact(const float *gsino, int S, int A, int PH, int PW, const float *T8,
                             int interp, int H, int W, int py, int px, float *gimg)
{
    memset(gimg, 0, (size_t)S * H * W * sizeof(float));
    for (int s = 0; s < S; ++s) {
        float *gi = gimg + (size_t)s * H * W;
        for (int a = 0; a < A; ++a) {
            const float *t = T8 + 8 * a;
            const float *grow = gsino + ((size_t)s * A + a) * PW;
            for (int i = 0; i < PH; ++i)
                for (int j = 0; j < PW; ++j) {
                    float x, y;
                    map_coord(t, j, i, &x, &y);
                    const float g = grow[j];
                    if (interp == ORACLE_NEAREST) {
                        canvas_add(gi, H, W, py, px, (long)roundf(y), (long)roundf(x), g);
                    } else {
                        const float yf = floorf(y), xf = floorf(x);
                        const float yc = yf + 1.0f, xc = xf + 1.0f;
                        canvas_add(gi, H, W, py, px, (long)yf, (long)xf, (yc - y) * ((xc - x) * g));
                        canvas_add(gi, H, W, py, px, (long)yf, (long)xc, (yc - y) * ((x - xf) * g));
                        canvas_add(gi, H, W, py, px, (long)yc, (long)xf, (y - yf) * ((xc - x) * g));
                        canvas_add(gi, H, W, py, px, (long)yc, (long)xc, (y - yf) * ((x - xf) * g));
                    }
                }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * a7: tomopy.project (TomoPy 1.11.0: sim/project.py + libtomo/recon/project.c + utils.c) as called
 * by create_sinogram, ctvae/helper_functions.py:33-38: center=None, emission=True,
 * sinogram_order=False.  obj [oy][ox][oz] fp32, data [oy][dt][dx] (libtomo's own order; the Python
 * wrapper swaps axes 0 and 1 afterwards).  `center` is dx/2 when the caller passes None.
 * ------------------------------------------------------------------------------------------- */
int oracle_siddon_dx(int ox, int oz, int pad)
{
    /* sim/project.py: dx = _round_to_even(sqrt(ox^2 + oz^2) + 2) if pad else ox */
    if (!pad) return ox;
    return (int)(ceil((sqrt((double)ox * ox + (double)oz * oz) + 2.0) / 2.0) * 2.0);
}

static int siddon_quadrant(float theta_p)
{
    /* utils.c calc_quadrant: M_PI is a double there, so the offset and the bounds are doubles */
    const int32_t ipi_c = 340870420;
    int32_t theta_i = (int32_t)(theta_p * ipi_c);
    theta_i += (theta_i < 0) ? (2.0f * M_PI * ipi_c) : 0;
    return ((theta_i >= 0 && theta_i < 0.5f * M_PI * ipi_c) ||
            (theta_i >= 1.0f * M_PI * ipi_c && theta_i < 1.5f * M_PI * ipi_c))
               ? 1 : 0;
}

/* Work arrays of one ray walk (libtomo allocates the same set per call). */
typedef struct {
    int ox, oz, dx;
    float mov;
    float *gridx, *gridy, *coordx, *coordy, *ax, *ay, *bx, *by, *coorx, *coory, *dist;
    int *indi;
} siddon_work;

static void siddon_work_init(siddon_work *w, int ox, int oz, int dx, float center)
{
    w->ox = ox; w->oz = oz; w->dx = dx;
    w->gridx = (float *)malloc((ox + 1) * sizeof(float));
    w->gridy = (float *)malloc((oz + 1) * sizeof(float));
    w->coordx = (float *)malloc((oz + 1) * sizeof(float));
    w->coordy = (float *)malloc((ox + 1) * sizeof(float));
    w->ax = (float *)malloc((ox + oz + 2) * sizeof(float));
    w->ay = (float *)malloc((ox + oz + 2) * sizeof(float));
    w->bx = (float *)malloc((ox + oz + 2) * sizeof(float));
    w->by = (float *)malloc((ox + oz + 2) * sizeof(float));
    w->coorx = (float *)malloc((ox + oz + 2) * sizeof(float));
    w->coory = (float *)malloc((ox + oz + 2) * sizeof(float));
    w->dist = (float *)malloc((ox + oz + 1) * sizeof(float));
    w->indi = (int *)malloc((ox + oz + 1) * sizeof(int));
    /* utils.c preprocessing */
    for (int i = 0; i <= ox; ++i) w->gridx[i] = -ox * 0.5f + i;
    for (int i = 0; i <= oz; ++i) w->gridy[i] = -oz * 0.5f + i;
    float mov = ((float)dx - 1) * 0.5f - center;
    if (mov - floorf(mov) < 0.01f) mov += 0.01f;
    w->mov = mov + 0.5f;
}

static void siddon_work_free(siddon_work *w)
{
    free(w->gridx); free(w->gridy); free(w->coordx); free(w->coordy); free(w->ax); free(w->ay); free(w->bx);
    free(w->by); free(w->coorx); free(w->coory); free(w->dist); free(w->indi);
}

/* One ray (angle sin_p / cos_p / quadrant, detector bin d): utils.c calc_coords -> trim_coords -> sort_intersections ->
 * calc_dist.  Fills w->dist[n], w->indi[n] for n < (return value) = csize - 1 segments. */
static int siddon_ray(siddon_work *w, float sin_p, float cos_p, int quadrant, int d)
{
    const int ox = w->ox, oz = w->oz, dx = w->dx;
    const float *gridx = w->gridx, *gridy = w->gridy;
    float *coordx = w->coordx, *coordy = w->coordy, *ax = w->ax, *ay = w->ay, *bx = w->bx, *by = w->by;
    float *coorx = w->coorx, *coory = w->coory;
    const float xi = (float)(-ox - oz);
    const float yi = (1 - dx) / 2.0f + d + w->mov;
    /* calc_coords */
    const float srcx = xi * cos_p - yi * sin_p, srcy = xi * sin_p + yi * cos_p;
    const float detx = -xi * cos_p - yi * sin_p, dety = -xi * sin_p + yi * cos_p;
    const float slope = (srcy - dety) / (srcx - detx);
    const float islope = (srcx - detx) / (srcy - dety);
    for (int n = 0; n <= oz; ++n) coordx[n] = islope * (gridy[n] - srcy) + srcx;
    for (int n = 0; n <= ox; ++n) coordy[n] = slope * (gridx[n] - srcx) + srcy;
    /* trim_coords */
    int asize = 0, bsize = 0;
    const float gx_gt = gridx[0] + 0.01f, gx_le = gridx[ox] - 0.01f;
    for (int n = 0; n <= oz; ++n)
        if (coordx[n] >= gx_gt && coordx[n] <= gx_le) {
            ax[asize] = coordx[n]; ay[asize] = gridy[n]; ++asize;
        }
    const float gy_gt = gridy[0] + 0.01f, gy_le = gridy[oz] - 0.01f;
    for (int n = 0; n <= ox; ++n)
        if (coordy[n] >= gy_gt && coordy[n] <= gy_le) {
            bx[bsize] = gridx[n]; by[bsize] = coordy[n]; ++bsize;
        }
    /* sort_intersections */
    int i = 0, j = 0, k = 0;
    while (i < asize && j < bsize) {
        const int a_ind = quadrant ? i : (asize - 1 - i);
        if (ax[a_ind] < bx[j]) { coorx[k] = ax[a_ind]; coory[k] = ay[a_ind]; ++i; }
        else { coorx[k] = bx[j]; coory[k] = by[j]; ++j; }
        ++k;
    }
    while (i < asize) {
        const int a_ind = quadrant ? i : (asize - 1 - i);
        coorx[k] = ax[a_ind]; coory[k] = ay[a_ind]; ++i; ++k;
    }
    while (j < bsize) { coorx[k] = bx[j]; coory[k] = by[j]; ++j; ++k; }
    const int csize = asize + bsize;
    /* calc_dist */
    for (int n = 0; n < csize - 1; ++n) {
        const float diffx = coorx[n + 1] - coorx[n], diffy = coory[n + 1] - coory[n];
        w->dist[n] = sqrtf(diffx * diffx + diffy * diffy);
        const float midx = (coorx[n + 1] + coorx[n]) * 0.5f;
        const float midy = (coory[n + 1] + coory[n]) * 0.5f;
        const float x1 = midx + ox * 0.5f, x2 = midy + oz * 0.5f;
        const int i1 = (int)x1, i2 = (int)x2;
        const int indx = i1 - (i1 > x1), indy = i2 - (i2 > x2);
        w->indi[n] = indy + indx * oz;
    }
    return csize > 0 ? csize - 1 : 0;
}

void oracle_siddon_project(const float *obj, int oy, int ox, int oz, const float *theta, int dt,
                           int dx, float center, float *data)
{
    siddon_work w;
    siddon_work_init(&w, ox, oz, dx, center);
    memset(data, 0, (size_t)oy * dt * dx * sizeof(float));
    for (int p = 0; p < dt; ++p) {
        const float theta_p = fmodf(theta[p], 2.0f * (float)M_PI);
        const int quadrant = siddon_quadrant(theta_p);
        const float sin_p = sinf(theta_p), cos_p = cosf(theta_p);
        for (int d = 0; d < dx; ++d) {
            const int nseg = siddon_ray(&w, sin_p, cos_p, quadrant, d);
            /* calc_simdata, every slice */
            for (int s = 0; s < oy; ++s) {
                const float *model = obj + (size_t)s * ox * oz;
                float *out = data + ((size_t)s * dt + p) * dx + d;
                for (int n = 0; n < nseg; ++n) *out += model[w.indi[n]] * w.dist[n];
            }
        }
    }
    siddon_work_free(&w);
}

/* ---------------------------------------------------------------------------------------------
 * tomopy.recon(..., algorithm='fbp', filter_name='none')  [3P-recalled: TomoPy 1.11.0 libtomo/recon/fbp.c], the call
 * ctvae/helper_functions.py:514 makes for the encoder's mask channel: for every slice s, angle p, detector bin d, in
 * that order, the ray's segments add data[s][p][d] * dist[n] into recon[s][indi[n]] -- the transpose of project.c's
 * calc_simdata.  data [oy][dt][dx] (libtomo's order), recon [oy][ngridx][ngridy], ADDED to its initial contents
 * (tomopy initialises it; the caller passes zeros or 1e-6 as it wishes).
 * ------------------------------------------------------------------------------------------- */
void oracle_siddon_backproject(const float *data, int oy, int dt, int dx, const float *theta, float center, int ngridx,
                               int ngridy, float *recon)
{
    siddon_work w;
    siddon_work_init(&w, ngridx, ngridy, dx, center);
    for (int s = 0; s < oy; ++s)
        for (int p = 0; p < dt; ++p) {
            const float theta_p = fmodf(theta[p], 2.0f * (float)M_PI);
            const int quadrant = siddon_quadrant(theta_p);
            const float sin_p = sinf(theta_p), cos_p = cosf(theta_p);
            for (int d = 0; d < dx; ++d) {
                const int nseg = siddon_ray(&w, sin_p, cos_p, quadrant, d);
                const float v = data[((size_t)s * dt + p) * dx + d];
                float *img = recon + (size_t)s * ngridx * ngridy;
                for (int n = 0; n < nseg; ++n) img[w.indi[n]] += v * w.dist[n];
            }
        }
    siddon_work_free(&w);
}

/* ---------------------------------------------------------------------------------------------
 * tomopy.recon(..., algorithm='sirt')  [3P-recalled: TomoPy 1.11.0 libtomo/recon/sirt.c], reached from
 * ctvae/helper_functions.py:445-457,503 and the README recipe (README.md:221).  Per iteration and slice:
 *   simdata = A recon (project.c's calc_simdata);  for every ray:  sum_dist2 = sum dist[n]^2,
 *   sum_dist[indi[n]] += dist[n];  if sum_dist2 != 0:  upd = (data - simdata) / sum_dist2,  update[indi[n]] += upd * dist[n];
 *   then  recon[pix] += update[pix] / sum_dist[pix]  where sum_dist[pix] != 0.
 * tomopy's default is num_iter = 1 and an initial recon of 1e-6 everywhere (the caller passes both).
 * ------------------------------------------------------------------------------------------- */
void oracle_sirt(const float *data, int oy, int dt, int dx, const float *theta, float center, int ngridx, int ngridy,
                 int num_iter, float *recon)
{
    siddon_work w;
    siddon_work_init(&w, ngridx, ngridy, dx, center);
    const size_t npix = (size_t)ngridx * ngridy;
    float *sum_dist = (float *)malloc(npix * sizeof(float));
    float *update = (float *)malloc(npix * sizeof(float));
    for (int it = 0; it < num_iter; ++it)
        for (int s = 0; s < oy; ++s) {
            float *img = recon + (size_t)s * npix;
            memset(sum_dist, 0, npix * sizeof(float));
            memset(update, 0, npix * sizeof(float));
            for (int p = 0; p < dt; ++p) {
                const float theta_p = fmodf(theta[p], 2.0f * (float)M_PI);
                const int quadrant = siddon_quadrant(theta_p);
                const float sin_p = sinf(theta_p), cos_p = cosf(theta_p);
                for (int d = 0; d < dx; ++d) {
                    const int nseg = siddon_ray(&w, sin_p, cos_p, quadrant, d);
                    float sim = 0.0f, sum_dist2 = 0.0f;
                    for (int n = 0; n < nseg; ++n) sim += img[w.indi[n]] * w.dist[n];
                    for (int n = 0; n < nseg; ++n) {
                        sum_dist2 += w.dist[n] * w.dist[n];
                        sum_dist[w.indi[n]] += w.dist[n];
                    }
                    if (sum_dist2 != 0.0f) {
                        const float upd = (data[((size_t)s * dt + p) * dx + d] - sim) / sum_dist2;
                        for (int n = 0; n < nseg; ++n) update[w.indi[n]] += upd * w.dist[n];
                    }
                }
            }
            for (size_t q = 0; q < npix; ++q)
                if (sum_dist[q] != 0.0f) img[q] += update[q] / sum_dist[q];
        }
    free(sum_dist); free(update);
    siddon_work_free(&w);
}

/* ---------------------------------------------------------------------------------------------
 * a6: iradon, ctvae/fbp_tensorflow.py:39-74, float64 / complex128 as the reference runs it.
 *   :49-50  filt = Re(ifft(fft(sino) * filter_1d))        (plain DFT here, O(P^2))
 *   :52-59  t[x][y][a] = ypr*cos(theta_a) - xpr*sin(theta_a), xpr = i - X/2, ypr = j - Y/2
 *   :61-70  tfp.math.interp_regular_1d_grid(x=t, x_ref_min=-P/2, x_ref_max=P-1-P/2, y_ref=filt row,
 *           fill_value='constant_extension')
 *   :72-74  sum over angles, times pi / (2A)
 * sino [B][A][P], theta [A], filter [P] (re, im), recon [B][X][Y].
 * ------------------------------------------------------------------------------------------- */
void oracle_fbp_filter(const double *sino, int R, int P, const double *filt_re, const double *filt_im,
                       double *out)
{
    double *fr = (double *)malloc(P * sizeof(double)), *fi = (double *)malloc(P * sizeof(double));
    double *cs = (double *)malloc(P * sizeof(double)), *sn = (double *)malloc(P * sizeof(double));
    for (int k = 0; k < P; ++k) {
        cs[k] = cos(2.0 * M_PI * k / P);
        sn[k] = sin(2.0 * M_PI * k / P);
    }
    for (int r = 0; r < R; ++r) {
        const double *x = sino + (size_t)r * P;
        for (int k = 0; k < P; ++k) {
            double re = 0.0, im = 0.0;
            for (int n = 0; n < P; ++n) {
                const int m = (int)(((long long)k * n) % P);
                re += x[n] * cs[m];
                im -= x[n] * sn[m];
            }
            const double hr = filt_re[k], hi = filt_im ? filt_im[k] : 0.0;
            fr[k] = re * hr - im * hi;
            fi[k] = re * hi + im * hr;
        }
        for (int n = 0; n < P; ++n) {
            double re = 0.0;
            for (int k = 0; k < P; ++k) {
                const int m = (int)(((long long)k * n) % P);
                re += fr[k] * cs[m] - fi[k] * sn[m];
            }
            out[(size_t)r * P + n] = re / P;
        }
    }
    free(fr); free(fi); free(cs); free(sn);
}

static inline double interp_regular_1d(const double *y_ref, int ny, double x, double x_min, double x_max)
{
    /* tensorflow-probability 0.14.0 math/interpolation.py _interp_regular_1d_grid_impl */
    double idx_unclipped = (x - x_min) / (x_max - x_min) * (double)(ny - 1);
    double idx = idx_unclipped;
    if (idx < 0.0) idx = 0.0;
    if (idx > (double)(ny - 1)) idx = (double)(ny - 1);
    double below = floor(idx);
    double above = fmin(below + 1.0, (double)(ny - 1));
    below = fmax(above - 1.0, 0.0);
    const double t = idx - below;
    double y = t * y_ref[(int)above] + (1.0 - t) * y_ref[(int)below];
    if (idx_unclipped < 0.0) y = y_ref[0];
    if (idx_unclipped > (double)(ny - 1)) y = y_ref[ny - 1];
    return y;
}

void oracle_fbp_backproject(const double *filt, int B, int A, int P, const double *theta, int X, int Y,
                            double *recon)
{
    const double x_min = 0.0 - P / 2.0, x_max = (double)(P - 1) - P / 2.0;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < X; ++i)
            for (int j = 0; j < Y; ++j) {
                const double xpr = (double)i - X / 2.0, ypr = (double)j - Y / 2.0;
                double acc = 0.0;
                for (int a = 0; a < A; ++a) {
                    const double t = ypr * cos(theta[a]) - xpr * sin(theta[a]);
                    acc += interp_regular_1d(filt + ((size_t)b * A + a) * P, P, t, x_min, x_max);
                }
                recon[((size_t)b * X + i) * Y + j] = acc * M_PI / (2.0 * A);
            }
}

void oracle_iradon(const double *sino, int B, int A, int P, const double *theta, int X, int Y,
                   const double *filt_re, const double *filt_im, double *recon)
{
    double *filt = (double *)malloc((size_t)B * A * P * sizeof(double));
    oracle_fbp_filter(sino, B * A, P, filt_re, filt_im, filt);
    oracle_fbp_backproject(filt, B, A, P, theta, X, Y, recon);
    free(filt);
}

/* ---------------------------------------------------------------------------------------------
 * a8: calculate_log_prob_M_given_R, ctvae/helper_functions.py:360-368 (after the projector call):
 *   loc = proj * mask[b][a] ; scale = eps + sqrt(loc / pnm + eps)
 *   log_prob = -0.5*(x/scale - loc/scale)^2 - (0.5*log(2*pi) + log(scale))   (tfd.Normal._log_prob)
 * proj, x [B][A][P], mask [B][A], out [B][A][P], all fp32.
 * ------------------------------------------------------------------------------------------- */
void oracle_loglik(const float *proj, const float *mask, const float *x, int B, int A, int P, float pnm,
                   float eps, float *out)
{
    const float half_log_2pi = (float)(0.5 * log(2.0 * M_PI));
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < A; ++a) {
            const float m = mask[(size_t)b * A + a];
            for (int j = 0; j < P; ++j) {
                const size_t k = ((size_t)b * A + a) * P + j;
                const float loc = proj[k] * m;
                const float scale = eps + sqrtf(loc / pnm + eps);
                const float z = x[k] / scale - loc / scale;
                out[k] = -0.5f * (z * z) - (half_log_2pi + logf(scale));
            }
        }
}

/* ---------------------------------------------------------------------------------------------
 * f2: the Poisson-noise forward model of ctvae/create_masks.py:80-103,
 *   out[s][a][j] = Poisson( max(sino, 0) * mask[s][a] * pnm ) / pnm        (:32 clamp, :82 mask, :94-95 draw)
 * The reference draws with tfd.Poisson(rate).sample() -- TensorFlow's generator, whose bits nothing pins; what it defines
 * is the DISTRIBUTION.  The build's sampler is counter-based and specified in full (ct_pvae_amd/csrc/poisson.hip header);
 * this is its CPU twin, written from that specification: same Philox4x32-10 blocks per element, multiplication method
 * below rate 10, Hormann's transformed rejection (PTRS) from there, exp / log by the same fixed double series -- so every
 * count can be compared exactly, and the distribution itself is tested against scipy (tests/test_oracle.py).
 * ------------------------------------------------------------------------------------------- */
static void oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void oracle_philox4x32_10(const uint32_t *ctr4, const uint32_t *key2, uint32_t *out4) { oracle_philox(ctr4, key2, out4); }

static double oracle_series_log(double x)
{
    uint64_t b;
    memcpy(&b, &x, 8);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &b, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 23.0;
    for (int d = 21; d >= 3; d -= 2) p = p * z + 1.0 / (double)d;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

static double oracle_series_exp_neg(double x)
{
    static const double inv_fact[15] = { 1.0, 1.0, 0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0,
                                         1.0 / 40320.0, 1.0 / 362880.0, 1.0 / 3628800.0, 1.0 / 39916800.0,
                                         1.0 / 479001600.0, 1.0 / 6227020800.0, 1.0 / 87178291200.0 };
    if (x < -700.0) return 0.0;
    const double kf = (double)(long long)(x * 1.4426950408889634 - 0.5);
    const double r = (x - kf * 0.693147180369123816490) - kf * 1.90821492927058770002e-10;
    double p = inv_fact[14];
    for (int d = 13; d >= 0; --d) p = p * r + inv_fact[d];
    const uint64_t sb = (uint64_t)((long long)kf + 1023) << 52;
    double scale;
    memcpy(&scale, &sb, 8);
    return p * scale;
}

static double oracle_log_factorial(double k)
{
    static const double small[10] = { 0.0, 0.0, 0.6931471805599453, 1.791759469228055, 3.1780538303479458,
                                      4.787491742782046, 6.579251212010101, 8.525161361065415, 10.60460290274525,
                                      12.801827480081469 };
    if (k < 10.0) return small[(int)k];
    const double n = k + 1.0, i = 1.0 / n, i2 = i * i;
    return (n - 0.5) * oracle_series_log(n) - n + 0.9189385332046727 +
           i * (1.0 / 12.0 - i2 * (1.0 / 360.0 - i2 * (1.0 / 1260.0 - i2 * (1.0 / 1680.0))));
}

static double oracle_u01(uint32_t w) { return ((double)w + 0.5) * 2.3283064365386963e-10; }

double oracle_poisson_count(double lam, uint64_t e, uint64_t seed)
{
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t ctr[4] = { (uint32_t)e, (uint32_t)(e >> 32), 0u, 0u }, w[4];
    if (!(lam > 0.0)) return 0.0;
    if (lam < 10.0) {           /* multiplication method: uniforms word by word from blocks 0, 1, ... */
        const double enlam = oracle_series_exp_neg(-lam);
        double prod = 1.0, count = 0.0;
        for (;; ++ctr[2]) {
            oracle_philox(ctr, key, w);
            for (int i = 0; i < 4; ++i) {
                prod = prod * oracle_u01(w[i]);
                if (!(prod > enlam)) return count;
                count = count + 1.0;
            }
        }
    }
    /* PTRS (Hormann 1993): iteration t draws U, V from block t */
    const double slam = sqrt(lam), loglam = oracle_series_log(lam);
    const double b = 0.931 + 2.53 * slam;
    const double a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4);
    const double vr = 0.9277 - 3.6224 / (b - 2.0);
    for (;; ++ctr[2]) {
        oracle_philox(ctr, key, w);
        const double U = oracle_u01(w[0]) - 0.5, V = oracle_u01(w[1]);
        const double us = 0.5 - fabs(U);
        const double k = floor((2.0 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        if (oracle_series_log(V) + oracle_series_log(invalpha) - oracle_series_log(a / (us * us) + b) <=
            -lam + k * loglam - oracle_log_factorial(k))
            return k;
        if (ctr[2] == 0xffffffffu) return k;
    }
}

void oracle_poisson_measure(const float *sino, const float *mask, int S, int A, int P, float pnm, uint64_t seed,
                            float *out)
{
    const size_t n = (size_t)S * A * P;
    for (size_t e = 0; e < n; ++e) {
        const float loc = fmaxf(sino[e], 0.0f) * mask[e / (size_t)P];
        const float rate = loc * pnm;
        out[e] = rate < 1.0e15f ? (float)oracle_poisson_count((double)rate, (uint64_t)e, seed) / pnm : loc;
    }
}

/* the two series against libm, for the tests (they must agree to ~1e-15 relative; bits need not) */
double oracle_series_log_public(double x) { return oracle_series_log(x); }
double oracle_series_exp_neg_public(double x) { return oracle_series_exp_neg(x); }
