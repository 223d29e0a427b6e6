/*
 * ctpvae_radon.h -- C ABI of the MI355X-native Radon projector that replaces CT_PVAE's physics
 * decoder operators.  Plain pointers and sizes only; every pointer named *_dev is a DEVICE pointer
 * (HIP, gfx950), everything else is host memory.  No allocation of outputs, no host synchronisation,
 * no global state except a thread-local last-error string.  `stream` is a hipStream_t passed as
 * void* (NULL = the default stream).
 *
 * Batches may be of any length: entry points whose kernels index slices with a grid dimension (<= 65535) launch longer
 * batches in chunks, back to back on `stream`.
 *
 * All functions return 0 on success and a negative CTPVAE_E* code otherwise; the message is
 * available from ctpvae_last_error().  Nothing is thrown across this boundary.
 *
 * Reference interfaces each entry point replaces (paths relative to the vganapati/CT_PVAE root):
 *   ctpvae_num_proj_pix / ctpvae_pad_amounts   ctvae/forward_functions.py:29-36   (pad_phantom size rule)
 *   ctpvae_rotate_transforms_f32               tfa.image.rotate's transform table, call sites
 *                                              ctvae/forward_functions.py:70-74,113; inverse used by
 *                                              TF's gradient, reached from ctvae/main_ct_vae.py:471-481
 *   ctpvae_rotate_fwd_f32                      project_tf_fast  ctvae/forward_functions.py:80-123
 *                                              project_tf_low_mem ctvae/forward_functions.py:49-78
 *   ctpvae_rotate_fwd_tiled_f32                the same operator for slices larger than LDS (config 5, 512 x 512)
 *   ctpvae_rotate_bwd_f32                      autodiff of the above (tf.GradientTape,
 *                                              ctvae/main_ct_vae.py:471-481) and its exact transpose
 *   ctpvae_rotate_plan_* / _planned_f32        the same two operators, batched: index arithmetic hoisted
 *                                              out of the per-object work (no counterpart in the reference)
 *   ctpvae_rotate_cplan_* / _fwd_compact_f32   the planned forward with step-coded (2 bits per sample) plans
 *   ctpvae_rotate_transforms_host_f32          the same table for a host-resident theta (the dataset's angle list,
 *                                              ctvae/main_ct_vae.py:152), evaluated on the host
 *   ctpvae_rotate_*_sel_*                      the per-step angle subset of calculate_log_prob_M_given_R,
 *                                              ctvae/helper_functions.py:350-357 (tf.gather of theta / mask / proj_sample)
 *   ctpvae_rotate_fwd_planned_loglik_f32       project_tf_fast + the Normal log_prob of calculate_log_prob_M_given_R
 *                                              ctvae/helper_functions.py:359-368, one launch
 *   ctpvae_rotate_bwd{,_planned}_scaled_f32    the gradient of the same caller w.r.t. the reconstruction (autodiff of
 *                                              ctvae/helper_functions.py:359-368 under the per-object sum of
 *                                              :305-312), one launch
 *   ctpvae_siddon_tables_f32 / _fwd_f32        create_sinogram -> tomopy.project
 *                                              ctvae/helper_functions.py:33-38
 *   ctpvae_siddon_bwd_f32 / _rownorm_f32       tomopy.recon(algorithm='fbp' | 'sirt') behind iradon_all / evaluate_sinogram
 *                                              ctvae/helper_functions.py:445-457,503,514
 *   ctpvae_gridrec_*                           tomopy.recon(algorithm='gridrec'), the default of iradon_all / evaluate_sinogram
 *                                              ctvae/helper_functions.py:445-457,503; ctvae/main_ct_vae.py:111-112
 *   ctpvae_fbp_filter_f64 / _backproject{,_bwd}_f64   iradon  ctvae/fbp_tensorflow.py:14-75
 *   ctpvae_loglik_fwd_f32 / _bwd_f32           calculate_log_prob_M_given_R
 *                                              ctvae/helper_functions.py:360-368
 *   ctpvae_poisson_measure_f32                 create_all_masks' noisy sparse sinograms  ctvae/create_masks.py:80-103
 */
#ifndef CTPVAE_RADON_H
#define CTPVAE_RADON_H

#ifdef __cplusplus
extern "C" {
#endif

#define CTPVAE_OK 0
#define CTPVAE_EINVAL (-1)   /* bad argument (shape, enum, null pointer) */
#define CTPVAE_EHIP (-2)     /* a HIP runtime call failed */
#define CTPVAE_ENODEV (-3)   /* no gfx950 device visible */

#define CTPVAE_NEAREST 0     /* tfa.image.rotate default, project_tf_fast */
#define CTPVAE_BILINEAR 1    /* project_tf_low_mem */

#define CTPVAE_BWD_TF_COMPAT 0 /* what TensorFlow's registered gradient computes (gather) */
#define CTPVAE_BWD_EXACT 1     /* true transpose of the forward (scatter) */

typedef void *ctpvae_stream_t;

/* Version of this ABI: major * 1000 + minor.  CTPVAE_ABI_VERSION is what THIS header describes: host code compiled against
 * it (csrc/torch_node.cpp, a maintainer's own binding) compares the macro with ctpvae_abi_version() of the library it loaded
 * and refuses a mismatch -- an entry point called with another version's argument list is a silent wrong-argument call. */
#define CTPVAE_ABI_VERSION 3400
int ctpvae_abi_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char *ctpvae_last_error(void);
/* Number of visible HIP devices, or a negative error code. */
int ctpvae_device_count(void);
/* Developer knobs (tools/ sweeps, tests that force one code path against another); none changes results.  The launch
 * path never reads the environment: the registry is filled once, when the library is loaded, from CTPVAE_TUNE_<NAME>
 * (and CTPVAE_NO_PLAN / CTPVAE_FORCE_GENERIC), and changed afterwards only here.  name: "NS", "G", "WAVES", "BNS",
 * "BW", "SEG_NS", "SEG_CHUNK", "SEG_PPT", "TILED_NS", "TILED_G", "SIDDON_NS", "SIDDON_THREADS", "SIDDON_PPB",
 * "SIDDON_BWD_NS", "SIDDON_BWD_CHUNKS", "MAX_SLICES", "NO_PLAN", "NO_COMPACT", "FORCE_GENERIC", ... (every name: DESIGN.md section 8); value < 0 unsets; name "*" unsets every knob.
 * _active: how many knobs are set (bench.py prints it next to its numbers). */
int ctpvae_tune_set(const char *name, int value);
int ctpvae_tune_active(void);

/* ---- a1: pad_phantom size rule (host arithmetic only) ------------------------------------- */
int ctpvae_num_proj_pix(int nx, int ny);
int ctpvae_pad_amounts(int n, int P, int *lo, int *hi);

/* ---- a3/a4: transform tables ---------------------------------------------------------------
 * theta_dev [A] fp32 radians as the caller of project_tf_fast passes them (the kernel negates).
 * H, W: height (rows) and width (columns) of the image that is rotated (the padded canvas).
 * T8_dev [A][8]: tfa's flat projective rows; Tinv8_dev [A][8] (may be NULL): the rows TensorFlow's
 * gradient uses (fp32 3x3 inverse, divided by its [2][2] element). */
int ctpvae_rotate_transforms_f32(const float *theta_dev, int A, int H, int W, float *T8_dev,
                                 float *Tinv8_dev, ctpvae_stream_t stream);
/* The same tables for a HOST-resident angle set, computed on the host (all pointers are host memory): cos/sin through
 * the C library's fp64 functions rounded once to fp32, then the same unfused fp32 expressions -- the very bits any other
 * host code gets from those expressions ("trig tables are computed by the caller on host in fp32 so CPU and
 * GPU see identical bits").  The device kernel above differs from it only where the device library's cos/sin round a
 * double differently (rare 1-ulp cases); it is kept for angle sets that exist only on the device. */
int ctpvae_rotate_transforms_host_f32(const float *theta, int A, int H, int W, float *T8, float *Tinv8);

/* ---- a2/a5: rotate-and-sum forward ---------------------------------------------------------
 * img_dev  [S][H][W] fp32, contiguous: the UNPADDED slices.  The PH x PW zero canvas with the
 *          slice at (py, px) is never materialised.
 * T8_dev   [A][8] rows applied to the canvas (elements 6, 7 must be 0).
 * sino_dev [S][A][PW]: sino[s][a][j] = sum_{i<PH} sample(canvas_s, T_a(j, i)), rows added in order. */
int ctpvae_rotate_fwd_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                          const float *T8_dev, int A, int interp, float *sino_dev,
                          ctpvae_stream_t stream);

/* The reference's float64 callers (ctvae/tomopy_forward_compare.py:52,56: xdesign's float64 phantoms through both projectors):
 * TensorFlow keeps the coordinates and the interpolation weights in fp32 whatever the image type, casts each weight to the
 * image type and multiplies, adds and row-sums in it.  Same geometry arguments as ctpvae_rotate_fwd_f32; img / sino are
 * double; a correctness-first kernel (one ray per lane, the slice in LDS when 8 H (W + 1) bytes fit).  Round 5, ABI 3400. */
int ctpvae_rotate_fwd_f64(const double *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                          const float *T8_dev, int A, int interp, double *sino_dev, ctpvae_stream_t stream);

/* ---- a2 for slices larger than LDS (512 x 512): tiled forward, NEAREST -----------------------
 * The slice is cut into tiles 64 wide x tile_h tall, tile_h = ceil(H / ceil(H / 128)) (EQUAL rows of tiles, ABI 3310: 128 for
 * H = 512; up to ABI 3300 it was 96 with a 32-row remainder -- the same taps, another association of the sum);
 * every tile is staged once and serves all angles, the tiles' partial sums (workspace) are then added in ascending tile order
 * (row-major over the slice):
 *     sino[s][a][j] = ((0 + p_0) + p_1) + ...,  p_t = sum over canvas rows i, ascending, of the taps inside tile t.
 * Tap indices are exactly those of ctpvae_rotate_fwd_f32; only the association of the fp32 sum differs.
 * ctpvae_rotate_tile_shape: 1 and the tile shape when (H, W, interp) is a tiled geometry, 0 (and zeros) when it is not --
 * what a checker needs to restate the sum (oracle_rotate_fwd_tiled); a function of its arguments alone, not of the device.
 * _workspace_bytes returns 0 when the slice fits LDS whole (use ctpvae_rotate_fwd_f32),
 * otherwise the size of the caller-owned device workspace (contents undefined on return). */
int ctpvae_rotate_tile_shape(int H, int W, int interp, int *tile_h, int *tile_w);
long long ctpvae_rotate_fwd_tiled_workspace_bytes(int S, int H, int W, int PH, int PW, int A, int interp);
int ctpvae_rotate_fwd_tiled_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                                const float *T8_dev, int A, void *workspace_dev, float *sino_dev,
                                ctpvae_stream_t stream);

/* BILINEAR slices larger than LDS (round 5, ABI 3400; ctvae/forward_functions.py:69-77 at 512 x 512): tiles 64 wide x
 * ceil(H / ceil(H / 96)) tall (86 for H = 512), each staged with a one-pixel halo below and to its right.  A sample's 2 x 2
 * footprint belongs to the tile of its FLOOR tap and is evaluated there exactly as ctpvae_rotate_fwd_f32 evaluates it; a tile's
 * partial sum adds its samples in ascending canvas-row order and the tiles are added in ascending order as above
 * (oracle_rotate_fwd_tiled with interp = 1).  interp = CTPVAE_NEAREST: ctpvae_rotate_fwd_tiled_f32.  Workspace and tile shape:
 * the two functions above with the same interp. */
int ctpvae_rotate_fwd_tiled_interp_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                                       const float *T8_dev, int A, int interp, void *workspace_dev, float *sino_dev,
                                       ctpvae_stream_t stream);

/* ... and with the log-likelihood epilogue of ctpvae_rotate_fwd_planned_loglik_f32 (below) in its reduce pass. */
int ctpvae_rotate_fwd_tiled_loglik_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                                       const float *T8_dev, int A, void *workspace_dev, const float *mask_dev,
                                       const float *meas_dev, const float *pnm_dev, float eps, float *sino_dev,
                                       float *lp_dev, float *dlp_dev, ctpvae_stream_t stream);

/* ... through COMPACT tile plans (round 3): the per-sample index arithmetic the tiled kernel is bound by, hoisted into a plan as
 * for slices that fit LDS (ctpvae_rotate_cplan_*): a section per (tile, angle, ray slot) = the ray's first tap inside the tile +
 * 2 bits per row.  Same taps, same order, same partial sums and reduce pass: the bits of ctpvae_rotate_fwd_tiled{,_loglik}_f32.
 * _bytes: 0 if the slice is not tiled (or compact plans are switched off); _overflowed (SYNCHRONISES): 1 = do not use the plan.
 * tplan_dev NULL: the direct tiled kernel (this is then ctpvae_rotate_fwd_tiled{,_loglik}_f32 with the options below).
 * lp_dev NULL: ray-sums only; else the log-likelihood epilogue in the reduce pass (mask, meas, pnm required; dlp_dev optional).
 * lp_sum_dev [S] (with lp_part_dev: ctpvae_loglik_part_floats(S, A, PW, 1) floats, its counters zero): the per-object log-likelihood sums
 * reduced in the reduce pass, as ctpvae_rotate_fwd_compact_f32 does for slices that fit LDS -- the bits of
 * ctpvae_loglik_object_sums_f32(lp, partition 1); sino_dev and lp_dev may then be NULL. */
long long ctpvae_rotate_tplan_bytes(int H, int W, int PH, int PW, int A);
int ctpvae_rotate_tplan_build_f32(const float *T8_dev, int A, int H, int W, int PH, int PW, int py, int px, void *tplan_dev,
                                  ctpvae_stream_t stream);
int ctpvae_rotate_tplan_overflowed(const void *tplan_dev, int H, int W, int PH, int PW, int A, ctpvae_stream_t stream);
int ctpvae_rotate_fwd_tiled_compact_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int py, int px,
                                        const float *T8_dev, int A, const void *tplan_dev, void *workspace_dev,
                                        const float *mask_dev, const float *meas_dev, const float *pnm_dev, float eps,
                                        float *sino_dev, float *lp_dev, float *dlp_dev, float *lp_part_dev, float *lp_sum_dev,
                                        ctpvae_stream_t stream);

/* ---- a4: backward of the above -------------------------------------------------------------
 * gsino_dev [S][A][PW] cotangent.  gimg_dev [S][H][W] (overwritten).
 * mode CTPVAE_BWD_TF_COMPAT: T8_dev must hold the INVERTED rows (Tinv8 above).
 * mode CTPVAE_BWD_EXACT:     T8_dev must hold the FORWARD rows. */
int ctpvae_rotate_bwd_f32(const float *gsino_dev, int S, int A, int PH, int PW, const float *T8_dev,
                          int interp, int mode, int H, int W, int py, int px, float *gimg_dev,
                          ctpvae_stream_t stream);
/* ... with a per-slice factor applied in the kernel's own store: gimg[s] = scale_dev[s * scale_stride] * (the sum).
 * This is the backward half of SURVEY 8 f1: with gsino_dev = the dlp the fused forward wrote and scale = the upstream
 * gradient of sum(lp[s]) (TF's gradient of reduce_sum is a broadcast: stride 0 over angles and bins), the whole
 * backward of calculate_log_prob_M_given_R (ctvae/helper_functions.py:336-368) is this one launch.
 * scale_dev NULL = ctpvae_rotate_bwd_f32; otherwise interp NEAREST and mode TF_COMPAT only. */
int ctpvae_rotate_bwd_scaled_f32(const float *gsino_dev, int S, int A, int PH, int PW, const float *T8_dev,
                                 int interp, int mode, int H, int W, int py, int px, const float *scale_dev,
                                 long long scale_stride, float *gimg_dev, ctpvae_stream_t stream);

/* ---- gather plans (NEAREST): the tap indices of a geometry, computed once, reused for every slice ----------
 * The tap a sample reads depends on (angle, canvas row, detector bin) only, so for batched projection the index
 * arithmetic of a3/a4 is evaluated ONCE by ctpvae_rotate_plan_build_f32 (exactly as the direct kernels do) and
 * stored as u16 LDS indices; ctpvae_rotate_{fwd,bwd}_planned_f32 then compute the SAME sums, bit for bit, as
 * ctpvae_rotate_fwd_f32 / ctpvae_rotate_bwd_f32(mode TF_COMPAT) with interp NEAREST.
 * which: 0 = forward plan, 1 = backward (TF_COMPAT) plan.  _supported returns 1 if the geometry fits the planned
 * kernels (slice / cotangent block must fit LDS), else 0 -- use the direct entry points then.
 * Plan buffers are caller-owned device memory of ctpvae_rotate_plan_bytes() bytes, 256-byte aligned. */
int ctpvae_rotate_plan_supported(int H, int W, int PH, int PW, int A, int interp, int which);
long long ctpvae_rotate_plan_bytes(int H, int W, int PH, int PW, int A, int which);
/* T8_dev: forward rows (needed for the forward plan); Tinv8_dev: inverted rows (needed for the backward plan);
 * either plan pointer may be NULL to skip it. */
int ctpvae_rotate_plan_build_f32(const float *T8_dev, const float *Tinv8_dev, int A, int H, int W, int PH, int PW,
                                 int py, int px, void *fwd_plan_dev, void *bwd_plan_dev, ctpvae_stream_t stream);
int ctpvae_rotate_fwd_planned_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int A,
                                  const void *fwd_plan_dev, float *sino_dev, ctpvae_stream_t stream);
/* a8 fused into a2: the planned forward that also writes, for every ray-sum, the log-probability of the
 * measured sample under it -- lp[s][a][j] = ctpvae_loglik_fwd_f32's expression on (sino[s][a][j], mask[s][a],
 * meas[s][a][j]) -- in the same launch.  sino_dev is still written.  dlp_dev (may be NULL) receives
 * d lp / d sino [S][A][PW], i.e. what ctpvae_loglik_bwd_f32 multiplies the upstream gradient by, so that the backward
 * needs no elementwise pass: see ctpvae_rotate_bwd_planned_scaled_f32. */
int ctpvae_rotate_fwd_planned_loglik_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int A,
                                         const void *fwd_plan_dev, const float *mask_dev, const float *meas_dev,
                                         const float *pnm_dev, float eps, float *sino_dev, float *lp_dev,
                                         float *dlp_dev, ctpvae_stream_t stream);
/* Angle subsets of a DENSE plan (the training loop projects `api` random angles of the dataset's 180 per step,
 * ctvae/helper_functions.py:350-357): the plan and tables are built ONCE for all A angles; angle_idx_dev [n_idx] int32 on
 * the device (1 <= n_idx <= 256; values are clamped into [0, A)) selects the plan angles a launch projects, in that
 * order: sino / lp / dlp are [S][n_idx][PW].  No plan or table kernel runs per step.  dense_inputs != 0: mask_dev and
 * meas_dev are the DENSE [S][A] / [S][A][PW] arrays and are read at the selected plan angles (the caller's gathers
 * mask[:, angles_i], proj_sample[:, angles_i] folded into the load); 0: they are [S][n_idx] / [S][n_idx][PW]. */
int ctpvae_rotate_fwd_planned_sel_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int A,
                                      const void *fwd_plan_dev, const int *angle_idx_dev, int n_idx, float *sino_dev,
                                      ctpvae_stream_t stream);
int ctpvae_rotate_fwd_planned_loglik_sel_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int A,
                                             const void *fwd_plan_dev, const int *angle_idx_dev, int n_idx,
                                             const float *mask_dev, const float *meas_dev, int dense_inputs,
                                             const float *pnm_dev, float eps, float *sino_dev, float *lp_dev,
                                             float *dlp_dev, ctpvae_stream_t stream);
/* COMPACT forward plans (round 3): the same taps as the u16 plan above, stored as the ray's first tap + 2 bits per canvas
 * row (does the source column step? does the source row step? -- x_in and y_in of ImageProjectiveTransformV3 are monotone in
 * the row number with slope <= 1, so their rounded values stay or step by one): 1/3 B instead of 2 B per sample (three rows
 * per code byte), 2.4 MB instead of 17 MB at the dataset's 180 angles, L2-resident on every XCD.  The plan kernel evaluates the reference
 * arithmetic exactly as the u16 plan's does; ctpvae_rotate_fwd_compact_f32 computes the SAME sums, bit for bit, as
 * ctpvae_rotate_fwd_planned{,_sel,_loglik,_loglik_sel}_f32 -- one entry point, optional operands:
 *   angle_idx      NULL = all A plan angles; else the n_idx (1..256) plan angles to project, outputs [S][n_idx][PW];
 *                  idx_on_host = 0: a DEVICE int32 vector; != 0: HOST memory (the training loop draws its subset on the
 *                  host, ctvae/helper_functions.py:104-107) -- the indices then travel in the kernel arguments: no upload,
 *                  no device copy for the kernel to wait on; the array may be reused as soon as the call returns
 *   lp_dev         NULL = ray-sums only; else the log-likelihood epilogue of ctpvae_rotate_fwd_planned_loglik_f32
 *                  (mask_dev, meas_dev, pnm_dev required; dense_inputs as in the _sel entry point; dlp_dev may be NULL)
 *   lp_sum_dev     NULL, or [S]: the PER-OBJECT log-likelihood sums the loss takes (ctvae/helper_functions.py:305-312),
 *                  reduced inside the launch: every (angle, 64-bin) task adds its log-probabilities by a
 *                  fixed xor butterfly and writes one partial into lp_part_dev (workspace of
 *                  ctpvae_loglik_part_floats(S, angles, PW, 0) floats, its arrival counters zero), a second tiny launch adds a
 *                  slice's partials in the fixed order stated at ctpvae_loglik_object_sums_f32 -- its bits (partition 0).  sino_dev and lp_dev
 *                  may then be NULL (nothing but dlp [S][n][PW] and the sums leaves the kernel).
 * _supported: 1 if the slice with its one-cell zero border fits LDS (interp NEAREST).  _overflowed (SYNCHRONISES): 1 if some
 * ray's steps do not fit the code (rows that are not a rotation, a ray still inside the slice at the canvas' last row, a
 * rounding tie that makes a coordinate jump by two) -- the plan must then not be used: keep the u16 plan. */
int ctpvae_rotate_cplan_supported(int H, int W, int PH, int PW, int A, int interp);
long long ctpvae_rotate_cplan_bytes(int H, int W, int PH, int PW, int A);
int ctpvae_rotate_cplan_build_f32(const float *T8_dev, int A, int H, int W, int PH, int PW, int py, int px, void *cplan_dev,
                                  ctpvae_stream_t stream);
int ctpvae_rotate_cplan_overflowed(const void *cplan_dev, int H, int W, int PH, int PW, int A, ctpvae_stream_t stream);
int ctpvae_rotate_fwd_compact_f32(const float *img_dev, int S, int H, int W, int PH, int PW, int A, const void *cplan_dev,
                                  const int *angle_idx, int n_idx, int idx_on_host, const float *mask_dev,
                                  const float *meas_dev, int dense_inputs, const float *pnm_dev, float eps, float *sino_dev,
                                  float *lp_dev, float *dlp_dev, float *lp_part_dev, float *lp_sum_dev, ctpvae_stream_t stream);

/* ... and the TF_COMPAT / NEAREST backward of such a subset: gsino_dev [S][n_idx][PW], Tinv8_dev the DENSE inverted table
 * [A_plan][8]; row k of a cotangent uses table row angle_idx_dev[k].  Same bits as ctpvae_rotate_bwd_scaled_f32 on the
 * gathered table (scale_dev may be NULL). */
int ctpvae_rotate_bwd_sel_scaled_f32(const float *gsino_dev, int S, int A_plan, int PH, int PW, const float *Tinv8_dev,
                                     const int *angle_idx_dev, int n_idx, int H, int W, int py, int px,
                                     const float *scale_dev, long long scale_stride, float *gimg_dev,
                                     ctpvae_stream_t stream);

/* ... through a plan: the "bwd4" layout stores the byte taps of the backward plan as one dword per (angle, four consecutive rows,
 * column), so a launch can take any subset of the plan's angles (the 16-angles-per-vector layout of the plan above
 * cannot).  Same bits as ctpvae_rotate_bwd_sel_scaled_f32 (sum over subset rows k ascending), no per-sample index
 * arithmetic.  angle_idx / idx_on_host as in ctpvae_rotate_fwd_compact_f32 (host indices: n_idx <= 256).  _bytes: 0 if the geometry does not fit byte taps (PW > 255: keep ctpvae_rotate_bwd_sel_scaled_f32). */
long long ctpvae_rotate_bwd4_plan_bytes(int H, int W, int PH, int PW, int A);
int ctpvae_rotate_bwd4_plan_build_f32(const float *Tinv8_dev, int A, int H, int W, int PH, int PW, int py, int px,
                                      void *plan_dev, ctpvae_stream_t stream);
int ctpvae_rotate_bwd_planned_sel_scaled_f32(const float *gsino_dev, int S, int H, int W, int PH, int PW, int A,
                                             const void *bwd4_plan_dev, const int *angle_idx, int n_idx, int idx_on_host,
                                             const float *scale_dev, long long scale_stride, float *gimg_dev,
                                             ctpvae_stream_t stream);

/* STEP PLAN of the direct (segment) backward, for slices too large for the planned backward above (512 x 512; round 3):
 * down a column of pixels the tap of an angle stays or steps by one per row (|t1| <= 1), always the same way, so a lane that
 * owns eight consecutive rows needs its first row's tap (7 bits, relative to the 80-bin segment its 64 x 32 tile stages for
 * the angle) and seven bits: one u16 per (angle, row octet, column), written by a kernel that evaluates the reference
 * arithmetic exactly as ctpvae_rotate_bwd_f32's kernel does.  Same taps, same order: the same bits as ctpvae_rotate_bwd_scaled_f32
 * (mode TF_COMPAT, NEAREST), with two index operations per tap instead of five.  _overflowed (SYNCHRONISES): 1 if the geometry
 * does not fit the code (a tile that leaves the canvas at some angle -- unpadded canvases --, rows that are not a rotation):
 * the plan must then not be used.  Small launches (fewer than 512 tiles of 64 x 32) run ctpvae_rotate_bwd_scaled_f32's kernel. */
long long ctpvae_rotate_bwd_step_plan_bytes(int H, int W, int A);
int ctpvae_rotate_bwd_step_plan_build_f32(const float *Tinv8_dev, int A, int H, int W, int PH, int PW, int py, int px, void *plan_dev,
                                          ctpvae_stream_t stream);
int ctpvae_rotate_bwd_step_plan_overflowed(const void *plan_dev, int H, int W, int A, ctpvae_stream_t stream);
int ctpvae_rotate_bwd_stepped_scaled_f32(const float *gsino_dev, int S, int A, int PH, int PW, const float *Tinv8_dev, int H, int W,
                                         int py, int px, const void *step_plan_dev, const float *scale_dev, long long scale_stride,
                                         float *gimg_dev, ctpvae_stream_t stream);

/* (Which backward: both give the same bits.  The planned one wins except for large batches at few angles -- S >= 80
 * and A <= 64 -- where ctpvae_rotate_bwd_f32's segment kernel, which streams no indices, is up to 25 % faster.) */
int ctpvae_rotate_bwd_planned_f32(const float *gsino_dev, int S, int H, int W, int PH, int PW, int A,
                                  const void *bwd_plan_dev, float *gimg_dev, ctpvae_stream_t stream);
/* ... with the per-slice factor of ctpvae_rotate_bwd_scaled_f32 (scale_dev NULL = no factor). */
int ctpvae_rotate_bwd_planned_scaled_f32(const float *gsino_dev, int S, int H, int W, int PH, int PW, int A,
                                         const void *bwd_plan_dev, const float *scale_dev, long long scale_stride,
                                         float *gimg_dev, ctpvae_stream_t stream);

/* Exact transpose of the NEAREST forward as a deterministic gather (no atomics, bit-reproducible): a rotation followed by
 * rounding sends at most two canvas samples of an angle to one pixel, so the plan stores, per (angle, pixel), the <= 2
 * detector bins whose cotangent the scatter would have added, in the scatter's order (canvas row, then bin), and the
 * planned backward kernel gathers them -- the same sums, bit for bit, as ctpvae_rotate_bwd_f32(mode EXACT) would produce
 * if its atomic adds arrived in order.  _bytes: size of the caller-owned plan buffer (256-byte aligned), 0 if the geometry
 * does not fit (PW > 255: keep ctpvae_rotate_bwd_f32).  _build needs both tables.  _overflowed (synchronises): 1 if
 * some pixel had more than two hits -- the rows were not a rotation -- and the plan must not be used. */
long long ctpvae_rotate_exact_plan_bytes(int H, int W, int PH, int PW, int A);
int ctpvae_rotate_exact_plan_build_f32(const float *T8_dev, const float *Tinv8_dev, int A, int H, int W, int PH, int PW,
                                       int py, int px, void *plan_dev, ctpvae_stream_t stream);
int ctpvae_rotate_exact_plan_overflowed(const void *plan_dev, int H, int W, int PH, int PW, int A, ctpvae_stream_t stream);
int ctpvae_rotate_bwd_exact_planned_f32(const float *gsino_dev, int S, int H, int W, int PH, int PW, int A,
                                        const void *exact_plan_dev, float *gimg_dev, ctpvae_stream_t stream);

/* Exact transpose as a deterministic gather through a plan of SUMMED WEIGHTS (round 5, ABI 3400; north_star: "bilinear
 * sample/scatter").  For a rotation the canvas samples that touch a pixel lie in at most three consecutive detector bins; the
 * plan holds, per (angle, pixel), the first of them and the pixel's summed fp32 weight in each (the forward's own weights,
 * summed over canvas rows in ascending order; interp = NEAREST: the number of samples whose tap the pixel is): 16 bytes per
 * (angle, pixel).  The backward adds ((W0 g[first] + W1 g[first+1]) + W2 g[first+2]) over the angles in ascending order: no
 * atomics at any size, equal bits run to run, <= 1e-5 (of the largest value) from ctpvae_rotate_bwd_f32(mode EXACT) / the
 * oracle's in-order scatter.  BILINEAR: the exact adjoint at every size; NEAREST: for geometries the byte plan above does not
 * hold (PW > 255: 512 x 512).  _bytes: the caller-owned plan buffer (16-byte aligned); _build needs both tables; _overflowed
 * (SYNCHRONISES): 1 if some pixel's samples span more than three bins (the rows are not a rotation) -- keep
 * ctpvae_rotate_bwd_f32 then.  Tinv8_dev places the cotangent segments a pixel tile stages. */
long long ctpvae_rotate_exact_wplan_bytes(int H, int W, int A);
int ctpvae_rotate_exact_wplan_build_f32(const float *T8_dev, const float *Tinv8_dev, int A, int H, int W, int PH, int PW,
                                        int py, int px, int interp, void *plan_dev, ctpvae_stream_t stream);
int ctpvae_rotate_exact_wplan_overflowed(const void *plan_dev, int H, int W, int A, ctpvae_stream_t stream);
int ctpvae_rotate_bwd_exact_wplan_f32(const float *gsino_dev, int S, int A, int PH, int PW, const float *Tinv8_dev,
                                      int H, int W, int py, int px, const void *plan_dev, float *gimg_dev,
                                      ctpvae_stream_t stream);

/* ---- a7: TomoPy-style ray-driven projector --------------------------------------------------
 * Tables (host side, fp32): theta [dt] -> sin, cos of fmodf(theta, 2*pi) and libtomo's quadrant flag. */
int ctpvae_siddon_dx(int ox, int oz, int pad);
int ctpvae_siddon_tables_f32(const float *theta, int dt, float *sin_out, float *cos_out, int *quadrant_out);
/* obj_dev [oy][ox][oz]; sin_dev/cos_dev [dt] fp32, quad_dev [dt] int32; data_dev [oy][dt][dx]
 * (libtomo's order; tomopy.project(sinogram_order=False) returns it with axes 0,1 swapped). */
int ctpvae_siddon_fwd_f32(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev,
                          const float *cos_dev, const int *quad_dev, int dt, int dx, float center,
                          float *data_dev, ctpvae_stream_t stream);

/* The transpose of ctpvae_siddon_fwd_f32 -- what libtomo's fbp.c accumulates (recon[indi[n]] += data * dist[n]; with
 * filter_name 'none', as ctvae/helper_functions.py:514 asks for the mask channel, that IS tomopy.recon(algorithm='fbp')) and
 * the A^T of its sirt.c (helper_functions.py:503 with algorithm='sirt').  data_dev [oy][dt][dx] -> recon_dev [oy][ox][oz].
 * Pixel-driven (round 3): a lane owns a pixel and asks the two rays per angle that can cross it for their segment in it,
 * found with libtomo's own fp32 expressions (the crossings around the pixel, trim_coords' 0.01 rule, the midpoint's pixel);
 * a pixel's terms arrive in libtomo's order (angles, then rays, ascending), so the result equals the ray-driven accumulation
 * bit for bit except for the corner-cutting slivers of rays that pass within fp32 rounding of a grid corner (~1e-6 of the
 * image's range, about one pixel per angle).  No atomics: bit-reproducible.  <A x, y> = <x, A^T y> to rounding.
 *   _workspace_bytes  device memory the calls below need (ray table, flags, one scratch image per slice); 256-byte aligned
 *   _prepare          geometry only: fills the workspace's ray table; once per (grid, angles, dx, center)
 *   _prepared         colsum_dev NULL:  recon = A^T data  (overwritten)
 *                     colsum_dev [ox][oz]:  recon += (A^T data) / colsum where colsum != 0 -- sirt.c's update, in place
 *   _bwd_f32          _prepare + _prepared(colsum NULL)
 *   ctpvae_siddon_fwd_resid_f32   the forward with sirt.c's per-ray factor as its store: upd = (meas - A obj) / rn2 where
 *                     rn2 != 0, else 0 (meas_dev [oy][dt][dx], rn2_dev [dt][dx] from _rownorm) -- a SIRT iteration is this
 *                     launch and one _prepared(colsum) launch.
 * _rownorm: rn2_dev [dt][dx] = sum of squared segment lengths of every ray (sirt.c's sum_dist2; geometry only). */
long long ctpvae_siddon_bwd_workspace_bytes(int oy, int ox, int oz, int dt, int dx);
int ctpvae_siddon_bwd_prepare_f32(int ox, int oz, const float *sin_dev, const float *cos_dev, const int *quad_dev, int dt, int dx,
                                  float center, void *workspace_dev, ctpvae_stream_t stream);
int ctpvae_siddon_bwd_prepared_f32(const float *data_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                                   const int *quad_dev, int dt, int dx, float center, const void *workspace_dev,
                                   const float *colsum_dev, float *recon_dev, ctpvae_stream_t stream);
int ctpvae_siddon_bwd_f32(const float *data_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                          const int *quad_dev, int dt, int dx, float center, void *workspace_dev, float *recon_dev,
                          ctpvae_stream_t stream);
int ctpvae_siddon_fwd_resid_f32(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                                const int *quad_dev, int dt, int dx, float center, const float *meas_dev,
                                const float *rn2_dev, float *upd_dev, ctpvae_stream_t stream);
/* The forward with a workspace (round 3): with >= 3 slices the slices are interleaved per pixel in workspace_dev
 * (_fwd_workspace_bytes() bytes, 16-byte aligned) and one walk of a ray serves 4 or 8 of them from L2 -- the same bits as ctpvae_siddon_fwd_f32,
 * 2-5x faster on grids whose slice pairs do not fit LDS.  meas_dev / rn2_dev both NULL: ray-sums; both given: the store of
 * ctpvae_siddon_fwd_resid_f32.  _fwd_workspace_bytes() == 0: no workspace needed (the LDS kernels are taken). */
long long ctpvae_siddon_fwd_workspace_bytes(int oy, int ox, int oz);
int ctpvae_siddon_fwd_ws_f32(const float *obj_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                             const int *quad_dev, int dt, int dx, float center, const float *meas_dev, const float *rn2_dev,
                             void *workspace_dev, float *data_dev, ctpvae_stream_t stream);
int ctpvae_siddon_rownorm_f32(int ox, int oz, const float *sin_dev, const float *cos_dev, const int *quad_dev, int dt,
                              int dx, float center, float *rn2_dev, ctpvae_stream_t stream);
/* Round 4: an iteration of the TV STAND-IN of tomopy.recon(algorithm='tv') (README.md:221 of the reference asks for 'tv';
 * libtomo's tv.c is NOT restated -- ct_pvae_amd/recon.py says so on every call) as TWO projector launches instead of ~15 torch
 * ops around them: the diagonally preconditioned Chambolle-Pock iteration for min 1/2 |A x - b|^2 + lam TV(x) on K = (A; grad),
 *   _fwd_ws_tv_dual:  p <- (p + sigma (A xbar - b)) / (1 + sigma)         sigma_dev [dt][dx] = 1 / (row sums of A), p in place
 *   _bwd_tv_primal:   q' = q + 0.5 grad xbar; q <- q' / (max(|q'|, lam) / lam); x <- x - tau (A^T p - div q); xbar <- 2 x - x_old
 *                     tau_dev [ox][oz] = 1 / (column sums of A + 4); x in place; xbar / qx / qy read at neighbouring pixels:
 *                     *_in and *_out must be distinct buffers (the caller swaps them every iteration)
 * with forward differences (zero at the far edges) and their negative transpose, every operation in the order
 * oracle/radon_oracle.py tv_standin() states: the kernels give its bits. */
int ctpvae_siddon_fwd_ws_tv_dual_f32(const float *xbar_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                                     const int *quad_dev, int dt, int dx, float center, const float *meas_dev,
                                     const float *sigma_dev, void *workspace_dev, float *p_dev, ctpvae_stream_t stream);
int ctpvae_siddon_bwd_tv_primal_f32(const float *p_dev, int oy, int ox, int oz, const float *sin_dev, const float *cos_dev,
                                    const int *quad_dev, int dt, int dx, float center, const void *workspace_dev,
                                    const float *tau_dev, float lam, float *x_dev, const float *xbar_in_dev, float *xbar_out_dev,
                                    const float *qx_in_dev, const float *qy_in_dev, float *qx_out_dev, float *qy_out_dev,
                                    ctpvae_stream_t stream);

/* ---- a6: filtered back-projection (float64, as the reference runs it) -----------------------
 * filter: circular convolution of every sinogram row with hker_dev [P] = Re(ifft(filter_1d)), which
 * equals Re(ifft(fft(row) * filter_1d)).  sino_dev, out_dev [R][P].
 * backproject: filt_dev [B][A][P], cos_dev/sin_dev [A] fp64 of theta, recon_dev [B][X][Y]. */
int ctpvae_fbp_filter_f64(const double *sino_dev, int R, int P, const double *hker_dev, double *out_dev,
                          ctpvae_stream_t stream);
int ctpvae_fbp_backproject_f64(const double *filt_dev, int B, int A, int P, const double *cos_dev,
                               const double *sin_dev, int X, int Y, double *recon_dev,
                               ctpvae_stream_t stream);
/* ... with the sampling geometry spelled out: pixel (i, j) at (i - x0, j - y0), detector sample k at k - t0.  The reference's
 * iradon is x0 = X / 2, y0 = Y / 2, t0 = P / 2 (the entry point above); tomopy's ray-driven grid -- pixel centres at half-
 * integers, bin d at d - (P - 1) / 2 (libtomo utils.c preprocessing / calc_coords) -- is x0 = (X - 1) / 2, y0 = (Y - 1) / 2,
 * t0 = (P - 1) / 2: what the gridrec stand-in of ct_pvae_amd/recon.py uses on sinograms made by create_sinogram. */
int ctpvae_fbp_backproject_geom_f64(const double *filt_dev, int B, int A, int P, const double *cos_dev,
                                    const double *sin_dev, int X, int Y, double x0, double y0, double t0,
                                    double *recon_dev, ctpvae_stream_t stream);

/* The transpose of ctpvae_fbp_backproject_geom_f64 (iradon's gradient; the reference's iradon, ctvae/fbp_tensorflow.py:14-75, is
 * TF ops and so differentiable): grecon_dev [B][X][Y] -> gfilt_dev [B][A][P], every output one ordered fp64 sum over pixels.
 * The transpose of ctpvae_fbp_filter_f64 is the same entry point with the kernel reversed, hker[(P - n) % P]. */
int ctpvae_fbp_backproject_bwd_f64(const double *grecon_dev, int B, int A, int P, const double *cos_dev,
                                   const double *sin_dev, int X, int Y, double x0, double y0, double t0,
                                   double *gfilt_dev, ctpvae_stream_t stream);

/* ---- f3: TomoPy's gridrec (the encoder's default input channel: tomopy.recon(..., algorithm='gridrec') at
 * ctvae/helper_functions.py:503, defaults ctvae/main_ct_vae.py:111-112; evaluate_sinogram ctvae/helper_functions.py:445-457;
 * bin/final_merit.py:58,81) -- TomoPy 1.11.0 libtomo/gridrec/gridrec.c [recalled, see oracle/gridrec_oracle.c]: zero-padded
 * 1-D FFT of every projection (two slices per complex transform), filter x centre phase, convolution onto a pdim x pdim
 * frequency grid with the separable prolate-spheroidal window, 2-D FFT, window correction.  pdim = the power of two >= dx.
 * _tables_host_f32 fills a HOST buffer of _tables_bytes() bytes (twiddles, window, correction, trig, filter x phase; every
 * pointer host memory; filter_name 0 none, 1 shepp, 2 cosine, 3 hann, 4 hamming, 5 ramlak, 6 parzen = tomopy's default for
 * gridrec, 7 butterworth with filter_par = {cutoff, order}); the caller uploads it.  _f32: data_dev [dy][dt][dx] (sinogram
 * order) -> recon_dev [dy][ngridx][ngridy]; deterministic (the convolution is a gather in gridrec.c's order, no atomics);
 * workspace_dev: _workspace_bytes() bytes of device memory, contents undefined. */
long long ctpvae_gridrec_tables_bytes(int dt, int dx);
int ctpvae_gridrec_tables_host_f32(int dt, int dx, float center, const float *theta, int filter_name, const float *filter_par,
                                   void *tables);
long long ctpvae_gridrec_workspace_bytes(int dy, int dt, int dx);
int ctpvae_gridrec_f32(const float *data_dev, int dy, int dt, int dx, const void *tables_dev, int ngridx, int ngridy,
                       void *workspace_dev, float *recon_dev, ctpvae_stream_t stream);

/* Per-object sums of a log-probability array lp_dev [S][A][PW] -> out_dev [S] (reduce_sum over angles and bins,
 * ctvae/helper_functions.py:305-306), in a FIXED order so that every path gives the same bits: the [A][PW] values of a slice
 * are cut into 64-lane tasks (partition 0: the planned kernels' tasks -- angle a, bin block jb = the two 32-bin bands
 * [c - 32 (jb + 1), c - 32 jb) and [c + 32 jb, c + 32 (jb + 1)), c = PW / 2, lanes 0..31 and 32..63; partition 1: the tiled
 * reduce pass's contiguous 64-bin blocks), a task's values are added by the xor butterfly 32, 16, 8, 4, 2, 1 (lanes
 * without a bin add +0.0f); an angle's task sums are added in ascending order (S_a), and the object's sum is
 * ((0 + B_0) + B_1) + ... with B_g = the same butterfly over S_(64 g) .. S_(64 g + 63) (angles past A add +0.0f).
 * _tasks_per_row: tasks per angle.
 * _part_floats (round 4, ABI 3300): the floats of the `lp_part_dev` workspace the fused per-object sums of
 * ctpvae_rotate_fwd_compact_f32 (partition 0) / ctpvae_rotate_fwd_tiled_compact_f32 (partition 1) take for S slices and
 * n_angles projected angles: one partial sum per (slice, angle, task) and, behind them, ONE ARRIVAL COUNTER PER SLICE, which
 * must be ZERO before the workspace's first use (hipMemset once, when it is allocated); every launch leaves them zero.  The
 * counters serve the developer knob FOLD_SUMS = 1 (the ordered sum of a slice's partials inside the projector launch, by the
 * workgroup that finishes the slice last: same bits, one launch less, measured 0.3-2 us SLOWER than the default's second
 * launch); the default does not touch them.  One workspace serves one launch at a time (launches on ONE
 * stream, as every other caller-owned workspace of this header). */
int ctpvae_loglik_tasks_per_row(int PW, int partition);
long long ctpvae_loglik_part_floats(int S, int n_angles, int PW, int partition);
int ctpvae_loglik_object_sums_f32(const float *lp_dev, int S, int A, int PW, int partition, float *out_dev,
                                  ctpvae_stream_t stream);

/* ---- a8: Gaussian-approximated Poisson log-likelihood epilogue ------------------------------
 * proj_dev, x_dev, out_dev [B][A][P]; mask_dev [B][A]; pnm_dev points at ONE fp32 on the device (the
 * reference keeps poisson_noise_multiplier in a Variable).
 *   loc = proj*mask ; scale = eps + sqrt(loc/pnm + eps) ; out = Normal(loc, scale).log_prob(x)
 * bwd: gproj = gout * d out / d proj  (gpnm_dev, if not NULL, receives sum gout * d out / d pnm). */
int ctpvae_loglik_fwd_f32(const float *proj_dev, const float *mask_dev, const float *x_dev, int B, int A,
                          int P, const float *pnm_dev, float eps, float *out_dev, ctpvae_stream_t stream);
int ctpvae_loglik_bwd_f32(const float *proj_dev, const float *mask_dev, const float *x_dev,
                          const float *gout_dev, int B, int A, int P, const float *pnm_dev, float eps,
                          float *gproj_dev, float *gpnm_dev, ctpvae_stream_t stream);

/* ---- f2: sparse noisy measurements (the step that feeds the training loop) -------------------------------------
 * ctvae/create_masks.py:80-103 in one launch: out[s][a][j] = Poisson(max(sino[s][a][j], 0) * mask[s][a] * pnm) / pnm.
 * sino_dev, out_dev [S][A][P]; mask_dev [S][A].  Counter-based and fully specified (csrc/poisson.hip): element e draws
 * from Philox4x32-10(counter = (e, block), key = seed), multiplication method below rate 10, transformed rejection
 * (PTRS) from there, exp / log by fixed double-precision series -- the same counts on any device or host for a seed,
 * whatever the launch shape.  The reference draws with tfd.Poisson(...).sample(): same distribution, other bits. */
int ctpvae_poisson_measure_f32(const float *sino_dev, const float *mask_dev, int S, int A, int P, float pnm,
                               unsigned long long seed, float *out_dev, ctpvae_stream_t stream);
/* Host-side known-answer hook for the generator: the four words of Philox4x32-10(counter4, key2) (host pointers). */
int ctpvae_philox4x32_10(const unsigned *counter4, const unsigned *key2, unsigned *out4);

#ifdef __cplusplus
}
#endif
#endif /* CTPVAE_RADON_H */
