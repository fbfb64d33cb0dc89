/* nrhints_hip.h - C ABI of the MI355X (gfx950) NRHints volumetric-rendering hot path.
 *
 * libnrhints_hip.so replaces, for the default `nr-hints` model configuration, the arithmetic inside the
 * reference's  NeuSHintRenderer.forward  (models/neus_hint_model.py:653-751) and the field networks it calls
 * (fields/sdf_field.py:106-148, fields/reflectance_network.py:68-96, fields/encodings.py:155-176).
 * The reference has no FFI layer of its own (it is pure PyTorch); the seam a maintainer binds is the Python
 * class  nrhints_amd.NeuSHintRenderer  which loads this library through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to caller-owned, contiguous float32 memory unless marked (host)
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises the host
 *   - return value: 0 = NRH_OK, negative = NRH_E_*; nrh_last_error_string() describes the last failure of the
 *     calling thread; no C++ exception, no ownership, crosses this boundary
 *   - weights arrive PACKED (nrhints_amd/packing.py documents the layout): weight-norm already folded
 *     (W = g * v / ||v||_row, fields/sdf_field.py:81-82), transposed copies for the reverse chain included
 *   - `precision` selects the matrix arithmetic: 0 = v_mfma_f32_16x16x4_f32 on float32 weights; 1 = "f16x3",
 *     three v_mfma_f32_16x16x32_f16 per product on weights pre-split into fp16 (hi, lo*2^11) pairs - the packed
 *     weight buffer then holds fp16 pairs in the same number of bytes; biases, heads and all I/O stay float32
 */
#ifndef NRHINTS_HIP_H
#define NRHINTS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define NRH_OK 0
#define NRH_E_INVALID (-1)     /* null pointer / bad size / bad mode */
#define NRH_E_LAUNCH (-2)      /* HIP reported an error at launch */
#define NRH_E_WORKSPACE (-3)   /* workspace too small */
#define NRH_E_UNSUPPORTED (-4) /* configuration outside the compiled network shape */

/* ABI version (major * 100 + minor) and a human-readable build string (names the source hash and the -D variant flags).
 * nrh_source_hash: the 16-hex-digit hash of the sources this binary was built from (nrhints_amd/build_id.py, embedded by
 * csrc/Makefile); the Python binding refuses a library whose hash differs from the tree's - a stale binary cannot pass for
 * the sources beside it.  "unknown" for a build that bypassed the Makefile. */
int nrh_version(void);
const char* nrh_build_info(void);
const char* nrh_source_hash(void);
const char* nrh_last_error_string(void);

/* Sizes (in floats) of the packed parameter buffers and of the per-wave scratch the gradient kernels need.
 * out[0] sdf packed weights, out[1] sdf biases, out[2] sdf head, out[3] colour packed weights,
 * out[4] colour biases, out[5] per-ray colour-input table stride, out[6] scratch floats per resident wave,
 * out[7] waves per MLP workgroup (scratch = nrh_mlp_grid() * out[7] * out[6] floats).
 * `out` is a HOST pointer to 8 ints. */
int nrh_param_sizes(int* out);

/* Number of persistent workgroups the MLP kernels will launch on the current device (scratch sizing). */
int nrh_mlp_grid(void);

/* ---- live kernel timing (measurement only; used by bench.py's roofline leg) --------------------------------
 * nrh_kernel_timing_select(kind): from now on bracket every launch of one kernel family with HIP events recorded
 * on the launching stream; kind 0/1/2 = SDF kernel in that mode, 3 = reflectance kernel, -1 = off (default).
 * nrh_kernel_timing_read: synchronises those events, returns their summed duration (ms) and count, and clears.
 * total_ms / launches are HOST pointers.  Not thread-safe; leave it off outside benchmarks. */
int nrh_kernel_timing_select(int kind);
/* Tests / measurements only: the sampler of a small training batch (passes of at most 16 384 points, 16 new samples per step) runs
 * each per-ray step in the tail of the SDF pass that feeds it (one launch instead of two; csrc/nrh_sdf_split.hip).  on = 0: never;
 * 1 (default): while a pass is one tile per workgroup (up to one ray per CU: where it is faster); 2: wherever the kernel supports
 * it (passes of at most 16 384 points); any other value only queries; returns the previous setting.  All forms give the same bits
 * (tests/test_gpu_split.py). */
int nrh_sampler_fusion(int on);
int nrh_kernel_timing_read(double* total_ms, long long* launches);

/* ---- SDF network -------------------------------------------------------------------------------------------
 * Evaluates the SDF MLP at points  p = ro[ray] + rd[ray] * t[ray * t_stride + j],  j < n_per_ray.
 *   mode 0: sdf only              (SDFNetwork.sdf,      fields/sdf_field.py:125-126; sampler call sites
 *                                  models/neus_hint_model.py:699, :325, :399)
 *   mode 1: sdf + d(sdf)/dp       (SDFNetwork.gradient, fields/sdf_field.py:136-148; shadow-ray get_alpha :335-336)
 *   mode 2: sdf + feature + grad  (SDFNetwork.forward,  fields/sdf_field.py:106-123 + .gradient; render_core :504-508)
 * sdf  is written at sdf[ray * sdf_stride + j];  grad [nrays*n_per_ray,3];  feat in 16-point D-layout tiles
 * [ceil(npts/16)][16][64][4] (nrhints_amd/packing.py: feat_tiles_to_rows converts to [npts,256]).
 * scratch: nrh_mlp_grid() * out[7] * out[6] floats (modes 1, 2), may be null for mode 0. */
int nrh_sdf_eval(int precision, int mode, const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro,
                 const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays, float* sdf,
                 int sdf_stride, float* grad, float* feat, float* scratch, void* stream);

/* The same evaluation by the "wide" f16x3 kernels (csrc/nrh_mlp32.h, nrh_sdf32.hip: 32-point tiles, one wavefront per SIMD,
 * activations resident in AGPRs, v_mfma_f32_32x32x16_f16).  Same arithmetic class as precision 1, same outputs and layouts;
 * different packed parameters:  sdf_w32 = the three per-mode chunk streams back to back (nrh_sdf_wide_stream_bytes() bytes of
 * fp16 hi/lo pairs, nrhints_amd/packing32.py: pack_sdf32), sdf_tab32 = [11][256] float32 bias / head tables.
 * scratch as for nrh_sdf_eval (the same buffer serves both).  NrhNet.sdf_w32 / sdf_tab32 select these kernels inside
 * nrh_render_forward for every SDF evaluation of the evaluation path when precision is 1.
 * Additional mode 3 (wide kernels only): sdf + the derivative along the ray in forward mode (16 points and their 16 tangents per
 * tile, no scratch); `grad` [npts,3] receives rd * (d sdf / dt) / |rd|^2, i.e. <rd, grad> equals <rd, true gradient>. */
int nrh_sdf_eval_wide(int mode, const void* sdf_w32, const float* sdf_tab32, const float* ro, const float* rd, const float* t,
                      int t_stride, int n_per_ray, long long nrays, float* sdf, int sdf_stride, float* grad, float* feat,
                      float* scratch, void* stream);
/* the same on the one-term builds (NrhNet.precision 2, "f16"): one v_mfma_f32_32x32x16_f16 per K step instead of three */
int nrh_sdf_eval_wide_f16(int mode, const void* sdf_w32, const float* sdf_tab32, const float* ro, const float* rd, const float* t,
                          int t_stride, int n_per_ray, long long nrays, float* sdf, int sdf_stride, float* grad, float* feat,
                          float* scratch, void* stream);
long long nrh_sdf_wide_stream_bytes(void);

/* SDF value (mode 0) of a SMALL point set, precision f16x3: one 16-point tile's 256 output channels are split over the four
 * waves of a workgroup (csrc/nrh_sdf_split.hip), so a pass of a few thousand points costs a quarter of a tile's matrix time
 * instead of all of it.  Replaces SDFNetwork.sdf (fields/sdf_field.py:125-126) as the hierarchical sampler calls it
 * (models/neus_hint_model.py:175-246) when the batch is the reference's per-rank share (trainer/trainer.py:116-123).
 * sdf_w / sdf_b / sdf_head are nrh_sdf_eval's PRECISION-1 parameters; every value is bit-identical to
 * nrh_sdf_eval(precision 1, mode 0).  tiles = 16-point tiles per workgroup: 1, 2, or 0 (one while that fills the CUs at most once, else two).
 * nrh_render_forward_train takes this kernel for sampler passes of at most 16 384 points. */
int nrh_sdf_eval_split(const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro, const float* rd,
                       const float* t, int t_stride, int n_per_ray, long long nrays, float* sdf, int sdf_stride, int tiles,
                       void* stream);
/* The same for sdf + d sdf / dx (mode 1: SDFNetwork.gradient, fields/sdf_field.py:136-148, as the shadow rays' get_alpha needs it,
 * models/neus_hint_model.py:335-336): one 16-point tile per workgroup, sigma' in the producing wave's registers.  grad [npts,3].
 * Bit-identical to nrh_sdf_eval(precision 1, mode 1).  nrh_render_forward_train takes it for the shadow pass of batches of at most
 * 12 288 points (96 rays). */
int nrh_sdf_grad_split(const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro, const float* rd,
                       const float* t, int t_stride, int n_per_ray, long long nrays, float* sdf, int sdf_stride, float* grad,
                       void* stream);

/* ---- SDF network, training --------------------------------------------------------------------------------
 * The reference differentiates d(sdf)/dp a second time with autograd (create_graph=True, fields/sdf_field.py:145;
 * loss.backward(), pipelines/base_pipeline.py:59-62).  Here that second-order backward is two more register-chain
 * sweeps over arrays the training forward saves (maths: nrhints_amd/sdf_function.py), and the weight gradients are
 * products over the saved arrays:
 *     dW_l = zbar[l]^T x_l + save_t[l]^T abar_in_l ,  db_l = colsum zbar[l]
 *     x_0 = embedding, x_l = save_h[l-1];   abar_in_0 = gebar, abar_in_l = abar[l-1];   d w_s += colsum abar[7] / 3
 * LAYOUT OF THE [8][npts][256] ARRAYS (save_h, save_s1, save_t, abar, zbar; float32): NOT row-major.  When nrh_train_arrays_tiled()
 * returns 1 (every build since ABI 146) they are TILED - per layer, per tile of 16 consecutive points one contiguous 16 KiB block
 * [channel block 16][point 16][channel 16]: element (point p, channel c) of a layer lives at float offset
 *     (p / 16) * 4096 + (c / 16) * 256 + (p % 16) * 16 + (c % 16)
 * (one contiguous KiB per wave instruction in the kernels that write and read them).  They are opaque hand-offs between
 * nrh_sdf_train_forward, nrh_sdf_train_backward and nrh_dw_gemm: nrh_dw_gemm (NrhDwJob.tiled_a / tiled_b) is the ONLY supported
 * consumer of the weight-gradient operands - a caller that ran its own GEMMs over them as if they were row-major would get wrong
 * gradients silently.  (nrhints_amd/dw.py: from_tiled is the de-tiling used by the tests.)  ROW-MAJOR are only: feat_rows
 * [npts,256], save_ge [npts][128], gebar [npts][64], pbar [npts,3], and every array of the reflectance network's training entries.
 * npts = nrays * n_per_ray must be a multiple of 16 and < 2^22 = 4 194 304 (32-bit byte offsets inside a layer; checked, NRH_E_INVALID).
 *
 * nrh_sdf_train_forward: as nrh_sdf_eval mode 2 (sdf [npts], grad [npts,3]) with the feature ROW-MAJOR feat_rows
 *   [npts,256], plus  save_h [8][npts][256] (softplus outputs; layer 3 already holds the skip concatenation),
 *   save_s1 [8][npts][256] (sigmoid(100 z)), save_t [8][npts][256] (reverse-chain stage inputs),
 *   save_ge [npts][128] (cols 0..38: d sdf/d embedding via layer 0; cols 73..111: via the skip connection).
 * nrh_sdf_train_backward: given the adjoints  sbar [npts], fbar [npts,256], gbar [npts,3]  of the three outputs
 *   writes  abar, zbar [8][npts][256] (tiled, see above), coup (8 * npts * 256 floats of hand-off between the two sweeps,
 *   tile-native: opaque to the caller), gebar [npts][64] and pbar [npts,3] (row-major) (adjoint of the points through the
 *   value path; the caller adds the term through the encoding's second derivative, see sdf_function.py).
 *   wt_feat: the feature head transposed, packed as one 256x256 stage (packing.pack_feat_transposed).
 * adj_scale (this entry, nrh_color_train_backward, nrh_outside_backward): a power of two S in [2^-60, 2^60], used by precision f16x3
 *   only: the kernel carries S x (its input adjoints) through the chain and writes 1 / S x (the result) - same values, but the
 *   fp16 halves of the 3-term split see adjoints of a magnitude that does not depend on the batch size.  The loss is normalised
 *   by the ray count (pipelines/base_pipeline.py:57-62): at 1 024 rays per step unscaled adjoints are ~ 1e-3 of a single ray's
 *   and the split's absolute floor (3e-11 below 6e-5) cost up to 6e-3 of a gradient tensor's scale against the reference's
 *   float64 step (tests/test_gpu_train1024.py).  Pass 2^round(log2(rays in the batch / 8)) (at least 1) for a loss normalised by
 *   the ray count - nrhints_amd._lib.adjoint_scale: the adjoints then have the magnitude of an 8-ray batch whatever the batch
 *   size, which leaves three more octaves of head-room below fp16's 65 504 than 2^round(log2(rays)) would (the largest seed,
 *   d alpha / d sdf <= inv_s / 4 per unit of colour adjoint, grows with the trained sharpness); 1 reproduces the unscaled chain.
 *   For losses that are NOT ~1 / rays (sum-reduced, custom weights) derive S from the seeds' range instead, as the _half entries
 *   do on the device (`dyn`) and the Python binding does on request (nrhints_amd._lib.ADJOINT_SCALE_FROM_SEEDS). */
int nrh_sdf_train_forward(int precision, const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro,
                          const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays, float* sdf,
                          float* grad, float* feat_rows, float* save_h, float* save_s1, float* save_t, float* save_ge,
                          void* stream);
int nrh_sdf_train_backward(int precision, const float* sdf_w, const float* wt_feat, const float* sdf_head, const float* ro,
                           const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays,
                           const float* save_s1, const float* save_t, const float* gbar, const float* fbar,
                           const float* sbar, float* abar, float* coup, float* gebar, float* zbar, float* pbar,
                           float adj_scale, void* stream);
/* 16-BIT HAND-OFFS of the weight-gradient operands (precision f16x3, batches above the channel-split kernels' range - more than 32
 * points per CU: nrh_train_half_supported).
 * h, abar and zbar are read by nothing but nrh_dw_gemm, which spends most of its time fetching them; these variants write them as
 * fp16 in the HALF-TILED layout (per tile of 16 points: [block pair 8][point 16][quarter 4][block of the pair 2][4 channels], 8 KiB;
 * arrays [8][npts][256] fp16, 16-byte aligned), which NrhDwJob.half_ops consumes with one fp16 MFMA pass and no conversion:
 *   save_h16  layers 0..6 of h (they are NOT written to save_h; layer 7 is, in float32: the heads' jobs read it)
 *   save_t16  a copy of layers 1..7 of t (save_t keeps all 8 in float32: the tangent sweep reads them)
 *   abar16    layers 0..6 of abar, zbar16 layers 1..7 of zbar, both as S x the value (NOT written to abar / zbar; abar layer 7 and
 *             zbar layer 0 are, in float32 and unscaled)
 *   dyn       float32 [4], zero before the first call: {S, 1 / S, 2 work words}.  S is the step's adjoint scale, a power of two
 *             taken from the range of the seeds (max |sbar|, |gbar|, |fbar| -> [8, 16] after scaling) by a small kernel ahead of the
 *             sweeps: it replaces adj_scale (the adjoints' range moves by 2^16 between batches - one sample next to the surface at
 *             inv_s ~ 1000 - which a constant cannot follow inside fp16's range); pass it on as NrhDwJob.dyn_scale.
 * Accuracy: operands of the products carry 11 bits instead of 16; measured against the reference's float64 gradients every tensor of
 * the 1 024-ray step stays inside the bound of tests/test_gpu_train1024.py (profiles/r05/dw16_emulation.log; bf16 would not). */
int nrh_train_half_supported(int precision, long long npts);
int nrh_sdf_train_forward_half(int precision, const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro,
                               const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays, float* sdf,
                               float* grad, float* feat_rows, float* save_h, float* save_s1, float* save_t, float* save_ge,
                               void* save_h16, void* save_t16, void* stream);
int nrh_sdf_train_backward_half(int precision, const float* sdf_w, const float* wt_feat, const float* sdf_head, const float* ro,
                                const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays,
                                const float* save_s1, const float* save_t, const float* gbar, const float* fbar,
                                const float* sbar, float* abar, float* coup, float* gebar, float* zbar, float* pbar,
                                void* abar16, void* zbar16, float* dyn, const void* save_t16, void* stream);

/* ---- the outside-NeRF background network (renderer.use_outside_nerf) -----------------------------------------------------
 * fields/nerf_density_field.py:30-89 as called from models/neus_hint_model.py:434-473: per point of the inverted-sphere
 * parameterisation pts4 = (p / |p|, 1 / |p|) [npts,4] with the ray's view direction and light position (views, pls [nrays,3];
 * point P belongs to ray P / pts_per_ray):  density [npts] (alpha_linear, before the softplus) and rgb [npts,3] (rgb_linear,
 * before the sigmoid).  on_w / on_b / on_wt: nrhints_amd.outside.pack_outside (sizes: nrh_outside_sizes -> packed floats,
 * bias floats, transposed packed floats, row widths 96 / 64 of the two encodings; f16x3: the same numbers of fp16 pairs).
 * Training: with the five save_* arrays (all or none; npts % 16 == 0) the forward also writes the operands of the weight
 * gradients row-major - save_x [npts][96] enc10(pts4), save_v [npts][64] enc4(cat[view, light]), save_h [8][npts][256] (ReLU
 * outputs), save_f [npts][256] (feature_linear), save_hv [npts][128] - and nrh_outside_backward turns the adjoints of the two
 * outputs into zbar [8][npts][256], fbar [npts][256], zvbar [npts][128] (operands of nrh_dw_gemm jobs) and the adjoints of the
 * two encodings xbar [npts][96], vbar [npts][64] (for the ray gradients; the caller applies the encodings' derivative). */
int nrh_outside_sizes(int* out5);
int nrh_outside_forward(int precision, const float* on_w, const float* on_b, const float* pts4, const float* views, const float* pls,
                        int pts_per_ray, long long npts, float* density, float* rgb, float* save_x, float* save_v, float* save_h,
                        float* save_f, float* save_hv, void* stream);
int nrh_outside_backward(int precision, const float* on_wt, const float* alpha_w, const float* density_bar, const float* rgb_bar,
                         const float* save_h, const float* save_hv, long long npts, float* zbar, float* fbar, float* zvbar, float* xbar,
                         float* vbar, float adj_scale, void* stream);

/* ---- weight-norm fold ---------------------------------------------------------------------------------------
 * W = v * g / ||v||_row for up to 16 linears in one launch (old-style nn.utils.weight_norm, dim = 0:
 * fields/sdf_field.py:81-82, fields/reflectance_network.py:61-62) and its adjoint in one launch.  rows / cols and the
 * pointer arrays are HOST arrays of length nlayers; v [rows,cols], g [rows] (or [rows,1]), w / wbar / vbar as v,
 * gbar as g; cols <= 384.  wbar[l] may be null (that layer's gradients are written as zero). */
int nrh_weight_norm_fold(int nlayers, const int* rows, const int* cols, const float* const* v, const float* const* g,
                         float* const* w, void* stream);
int nrh_weight_norm_fold_backward(int nlayers, const int* rows, const int* cols, const float* const* v, const float* const* g,
                                  const float* const* wbar, float* const* vbar, float* const* gbar, void* stream);

/* ---- reflectance network, training ------------------------------------------------------------------------
 * ReflectanceNetwork.forward (fields/reflectance_network.py:68-96) on row-major inputs with the arrays its backward
 * needs, and the adjoint sweep (what autograd does through the 5 linears, ReLUs and the sigmoid in the reference).
 *   forward : feat_rows [P,256], pts [P,3], normal [P,3], raymisc [nrays, out[5]] (as nrh_color_eval)
 *             -> color [P,3];  save_h [4][P][256] (ReLU outputs),  save_misc [P][128 | 64] (non-feature input, kernel order:
 *             pts 3, normal 3, enc4(view) 27, enc4(pl) 27 (, enc4(vis) 9, enc4(cue) 36), zero padded)
 *   backward: zbar4 [P,3] = adjoint of the pre-sigmoid output, col_wt = transposed stages
 *             (packing.pack_color_transposed, nrh_color_transposed_floats(hints) floats or fp16 pairs)
 *             -> zbar [4][P][256], fbar [P,256] (adjoint of feat), mbar [P][128 | 64] (adjoint of the non-feature input)
 *   weight gradients are products over these (row-major) arrays:  dW_l = zbar[l]^T save_h[l-1], ... - nrh_dw_gemm jobs in this
 *   package; being row-major, any GEMM would do */
long long nrh_color_transposed_floats(int hints);
/* nrh_color_train_forward with the per-ray table indexed per GROUP of `samples_per_row` consecutive samples (a power of two
 * <= 128; 128 = per ray): raymisc [nrays * 128 / samples_per_row, 100].  The partial visibility hint's training forward. */
int nrh_color_train_forward_grouped(int precision, int hints, const float* col_w, const float* col_b, const float* feat_rows,
                                    const float* pts, const float* normal, const float* raymisc, int samples_per_row, long long nrays,
                                    float* color, float* save_h, float* save_misc, void* stream);
int nrh_color_train_forward(int precision, int hints, const float* col_w, const float* col_b, const float* feat_rows,
                            const float* pts, const float* normal, const float* raymisc, long long nrays, float* color,
                            float* save_h, float* save_misc, void* stream);
int nrh_color_train_backward(int precision, int hints, const float* col_wt, const float* zbar4, const float* save_h,
                             long long nrays, float* zbar, float* fbar, float* mbar, float adj_scale, void* stream);
/* The same with 16-BIT HAND-OFFS to nrh_dw_gemm (precision f16x3; see nrh_sdf_train_forward_half for the layout): save_h16 = fp16
 * [4][nrays*128][256] half-tiled, layers 0..2 of the ReLU outputs (NOT written to save_h; layer 3 is) - the adjoint sweep reads them
 * as masks; zbar16 = layers 1..3 of zbar as S x half_gain x the value (NOT written to zbar; layer 0 is, unscaled).  The seeds of
 * this chain are bounded (|zbar4| <= 1 / (12 rays): a sigmoid's derivative times a weight <= 1 times the loss normalisation), so
 * half_gain is a constant power of two: with adj_scale = rays / 8, 1 024 puts the arrays' maxima at 2^-6 .. 2^2. */
int nrh_color_train_forward_half(int precision, int hints, const float* col_w, const float* col_b, const float* feat_rows,
                                 const float* pts, const float* normal, const float* raymisc, int samples_per_row, long long nrays,
                                 float* color, float* save_h, float* save_misc, void* save_h16, void* stream);
int nrh_color_train_backward_half(int precision, int hints, const float* col_wt, const float* zbar4, const float* save_h,
                                  long long nrays, float* zbar, float* fbar, float* mbar, float adj_scale, const void* save_h16,
                                  void* zbar16, float half_gain, void* stream);

/* ---- alpha stage, training ---------------------------------------------------------------------------------
 * NeuSHintRenderer.get_alpha + compositing weights + unit normals (models/neus_hint_model.py:339-356, :521-525, :584)
 * for 128 samples per ray and the adjoint of exactly that (what autograd does in the reference's backward).
 *   forward : sdf [nrays,128], grad [nrays*128,3], rd [nrays,3], dists [nrays,128] -> weights [nrays,128],
 *             nhat [nrays*128,3]
 *   backward: + weights_bar [nrays,128], nhat_bar [nrays*128,3] (may be null) -> sdf_bar, grad_bar, rd_bar [nrays,3],
 *             invs_bar [nrays] (per-ray partial of the adjoint of inv_s; the caller sums and chains to `variance`) */
int nrh_alpha_train_forward(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                            float cos_anneal, const float* dyn_scalars /* as NrhNet.dyn_scalars, may be null */,
                            long long nrays, float* weights, float* nhat, void* stream);
int nrh_alpha_train_backward(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                             float cos_anneal, const float* dyn_scalars, long long nrays, const float* weights_bar, const float* nhat_bar,
                             float* sdf_bar, float* grad_bar, float* rd_bar, float* invs_bar, void* stream);

/* ---- hierarchical sampler (one launch = merge the previous 16 samples and/or draw 16 new ones) ------------
 * NeuSHintRenderer.up_sample (models/neus_hint_model.py:270-315) + sample_pdf (:21-65, det=True) +
 * cat_z_vals (:317-331) + section mid-points (:491-496, :416-418).
 *   z, s          [nrays,128] sorted ray parameters / sdf values, `n` valid on entry
 *   do_merge      merge znew_in (and snew_in if merge_sdf) into z (and s); n grows by 16
 *   do_upsample   write 16 new samples per ray to znew_out using fixed sharpness inv_s (= 64 * 2^step)
 *   do_finalize   (n == 128 after the merge) write section lengths `dists` and mid-points `tmid`; the last
 *                 section length is last_dist_ray[ray] if non-null, else last_dist
 *   lin16         torch.linspace(0,1,16) as float32 */
int nrh_sampler_step(const float* ro, const float* rd, float* z, float* s, const float* znew_in,
                     const float* snew_in, float* znew_out, const float* lin16, const float* last_dist_ray,
                     float* tmid, float* dists, float inv_s, float last_dist, int nrays, int n, int do_merge,
                     int merge_sdf, int do_upsample, int do_finalize, void* stream);

/* ---- reflectance network ------------------------------------------------------------------------------------
 * ReflectanceNetwork.forward (fields/reflectance_network.py:68-96) for 128 samples per ray.
 *   feat     D-layout tiles from nrh_sdf_eval(mode 2);  nhat [nrays*128,3] unit normals;
 *   raymisc  [nrays, out[5]] per-ray part of the input: enc4(view) | enc4(pl) | enc4(vis) | enc4(cue)
 *   color    [nrays*128,3] */
int nrh_color_eval(int precision, int hints, const float* col_w, const float* col_b, const float* feat, const float* ro, const float* rd,
                   const float* tmid, const float* nhat, const float* raymisc, long long nrays, float* color,
                   void* stream);

/* ---- the renderer -------------------------------------------------------------------------------------------
 * NeuSHintRenderer.forward, default nr-hints configuration, no autograd (models/neus_hint_model.py:653-751):
 * 64 coarse + 4x16 importance samples, shadow hint through the alpha-blended hit point, 4-roughness specular cue.
 * Inputs  (RayBundle fields, camera/ray_utils.py:214-235): origins, directions, pl_positions [n,3]; nears, fars [n].
 * Outputs (RenderOutput fields, models/neus_hint_model.py:216-233), any of the optional ones may be null:
 *   rgb [n,3], depth [n], weights [n,128], inside_sphere [n,128], analytic_normals [n,128,3],
 *   normalized_normals [n,128,3], visibilities [n], specular_cue [n,128,4]; and, for callers that continue the
 *   computation themselves (the autograd training path), the section mid-points mid_z [n,128] and lengths dists [n,128]
 *   (models/neus_hint_model.py:491-493); and the per-pixel normal maps of the evaluation loop,
 *   normal_map / normalized_normal_map [n,3] = sum_j normal_j * weight_j * inside_j in world space
 *   (pipelines/base_pipeline.py:125-131 computes them on the CPU from the per-sample arrays).
 * Scalars: inv_s = clip(exp(10 * variance), 1e-6, 1e6) (:110, :337); cos_anneal (:669-671).
 * background (device, [3]) may be null.  t_rand_primary [n] / t_rand_shadow [n,64]: training jitter (:682, :394),
 * null at evaluation.  lin64 / lin16: torch.linspace(0,1,64|16) as float32.  zero_hints: geometry warm-up (:577, :617).
 * workspace: nrh_render_workspace_floats(n) floats. */
typedef struct NrhNet {
  const float* sdf_w;
  const float* sdf_b;
  const float* sdf_head;
  const float* col_w;
  const float* col_b;
  float inv_s;
  int precision; /* 0 = f32 MFMA (exact fp32), 1 = f16x3 split MFMA (fp32-equivalent accuracy, 16/3 the rate), 2 = f16: precision 1's
                    buffers with the wide SDF kernels in their ONE-TERM builds (a single fp16 MFMA pass per K step: weights and
                    activations of the SDF network at 11 bits; evaluation only, needs sdf_w32 / sdf_tab32) - a reduced-precision
                    mode, narrower than the reference's float32 */
  int hints;       /* 1 = shadow + specular hints (nr-hints presets); 0 = none (pl-naive preset, configs/main_config.py:67-76):
                      no shadow march, reflectance input 316 wide, col_w packed accordingly */
  int normal_type; /* 0 = NormalizedAnalytic, 1 = Analytic normal fed to the reflectance net (models/neus_hint_model.py:621-625) */
  int depth_type;  /* 0 = AlphaBlend, 1 = MaximalWeightPoint, 2 = SphereTracing (:526-538) */
  const float* dyn_scalars; /* optional DEVICE [inv_s, cos_anneal]: when non-null it overrides `inv_s` above and the `cos_anneal`
                               argument of the render calls, read by the kernels at run time - a captured hipGraph of a
                               training step then follows the changing variance parameter and anneal schedule */
  const void* sdf_w32;      /* optional: packed streams of the wide f16x3 SDF kernels (see nrh_sdf_eval_wide); when both are   */
  const float* sdf_tab32;   /* non-null and precision is 1, the evaluation path uses them for every SDF evaluation            */
  int feat_fused;           /* 1: the FEAT block of sdf_w32 / row 8 of sdf_tab32 hold W0feat * W_feat and W0feat * b_feat, the
                               feature head multiplied into the feature block of the reflectance net's first layer (both are
                               linear, fields/sdf_field.py:119-123 -> fields/reflectance_network.py:77-84): nrh_render_forward
                               then skips that block in the reflectance kernel.  Evaluation only; 0 = plain feature head */
  const void* col_w32;      /* optional, with feat_fused: the reflectance net as a block stream for the wide kernel          */
  const float* col_tab32;   /* (packing32.pack_color32: nrh_color_wide_stream_bytes() bytes + [5][256] float32 tables); when   */
                            /* both are non-null the evaluation render runs fields/reflectance_network.py:68-96 on it         */
  int shadow_jvp;           /* 1 (with sdf_w32): the shadow march's last evaluation uses nrh_sdf_eval mode 3 - value + derivative
                               ALONG the ray in forward mode, returned as rd * (d sdf / dt) / |rd|^2 in place of the gradient:
                               get_alpha only uses <dirs, gradients> (models/neus_hint_model.py:343) */
  int shadow_clip;          /* renderer.n_shadow_importance_clip (models/neus_hint_model.py:553-575): -1 / 0 = one shadow ray per
                               primary ray, aimed at the hit point; 1, 2, 4, 8 or 16 = the partial visibility hint - one shadow ray
                               per group of 128 / clip consecutive samples, aimed at the group's first sample position; the
                               `visibilities` output is then the group value at the maximal-weight sample, t_rand_shadow is
                               [nrays * clip, 64] (row ray * clip + group) and NrhTrainSaves.raymisc [nrays * clip, 100] */
  int samples;              /* 0 or 128: 64 stratified + 64 importance samples per ray; 64: renderer.n_importance_samples = 0
                               (models/neus_hint_model.py:696 - no hierarchical sampling, the 64 coarse samples are final).  The
                               per-sample arrays keep 128 entries per ray; entries 64..127 are padding with weight exactly 0 */
  /* renderer.use_outside_nerf (models/neus_hint_model.py:516-519, :630-633): the background network is the caller's; these carry its
     results into the call and what the caller needs back out of it.  All NULL otherwise. */
  const float* bg_alpha;    /* [nrays,160] alpha of render_outside at the merged sample positions (first 128 = this ray's samples):
                               a sample outside the unit sphere takes it instead of the NeuS alpha - weights, depth, hit point,
                               shadow march and cue follow */
  float* tail_t;            /* out [nrays]: transmittance behind sample 127 (the 32 samples beyond the sphere start from it) */
  float* sampled_color;     /* out [nrays,128,3]: the reflectance net's colour per sample (the caller blends and composites; the
                               call's own `rgb` is then NOT the final colour) */
  /* the renderer's two free scalars (models/neus_hint_model.py:161, :163) - kernel constants, not compiled shapes.  custom_consts
     0: the reference's defaults (roughness 0.02, 0.05, 0.13, 0.34; offset 1e-2); 1: the values below.  Doubles, because the
     reference forms k = (rho + 1)^2 / 8, rho^2 and 1 - offset as Python scalars before they meet a float32 tensor (:387, :600-611) */
  int custom_consts;
  double specular_roughness[4];
  double shadow_ray_offset;
  /* Sample counts off the reference's defaults (models/neus_hint_model.py:139-171; n_coarse = 0: the defaults - 64 + 64 importance
     samples in 4 steps on the primary ray, 64 + 64 on the shadow ray - and `samples` as described above).  Otherwise:
       n_coarse  renderer.n_samples (2 .. 128)        n_steps  renderer.up_sample_steps (0 with n_importance_samples = 0)
       n_new     n_importance_samples / up_sample_steps, at most 16;   samples = n_coarse + n_steps * n_new <= 128
       s_coarse  n_shadow_samples (2 .. 64)           s_new    n_shadow_importance_samples / 4 (the shadow march always takes 4
                 steps, :373; 0 = none), at most 16;  s_coarse + 4 s_new <= 128
     lin_tables: DEVICE [4][128] float32, rows = torch.linspace(0, 1, k) for k = n_coarse, n_new, s_coarse, s_new (bit-exact
     tables from the host, like lin64 / lin16; t_rand_shadow is then [nrays, s_coarse]).  The per-sample arrays keep 128 entries
     per ray; entries past `samples` are padding with weight exactly 0.  Not with bg_alpha or shadow_clip > 0. */
  int n_coarse;
  int n_steps;
  int n_new;
  int s_coarse;
  int s_new;
  const float* lin_tables;
} NrhNet;

long long nrh_render_workspace_floats(long long nrays);

int nrh_render_forward(const NrhNet* net /* host */, const float* origins, const float* directions,
                       const float* pl_positions, const float* nears, const float* fars, long long nrays,
                       const float* background, float cos_anneal, const float* t_rand_primary,
                       const float* t_rand_shadow, int zero_hints, const float* lin64, const float* lin16,
                       float* rgb, float* depth, float* weights, float* inside_sphere, float* analytic_normals,
                       float* normalized_normals, float* visibilities, float* specular_cue, float* mid_z, float* dists,
                       float* normal_map, float* normalized_normal_map, float* workspace, long long workspace_floats,
                       void* stream);

/* Training forward of the same path (NeuSHintRenderer.forward with is_training=True under autograd): identical to
 * nrh_render_forward up to and including the shadow march - samplers, depth / hit point, visibility, specular cue -
 * with the SDF network at the 128 section mid-points evaluated by the training kernel (nrh_sdf_train_forward), whose
 * outputs land in `saves`.  The reflectance network and the compositing are NOT run: they, and everything else the
 * loss differentiates (models/neus_hint_model.py:504-510, :521-525, :626-637), belong to the caller's backward pass,
 * which feeds nrh_sdf_train_backward.  All per-sample outputs as in nrh_render_forward (optional ones may be null). */
typedef struct NrhTrainSaves {
  float* sdf;        /* [nrays,128]        sdf at the section mid-points */
  float* feat_rows;  /* [nrays*128,256]    row-major feature */
  float* save_h;     /* [8][nrays*128][256] */
  float* save_s1;    /* [8][nrays*128][256] */
  float* save_t;     /* [8][nrays*128][256] */
  float* save_ge;    /* [nrays*128][128] */
  float* raymisc;    /* optional [nrays,100]: the reflectance net's per-ray encodings enc4(view) | enc4(pl) | enc4(vis) | enc4(cue)
                        (what nrh_visibility writes); NULL = kept in the workspace */
  float* shadow_mid_z;  /* optional [nrays,128]: section mid-points along the shadow ray light -> hit point ... */
  float* shadow_dists;  /* optional [nrays,128]: ... and section lengths (get_visibility :411-415), for callers that differentiate
                           the visibility hint (renderer.shadow_hint_gradient); NULL = kept in the workspace */
  float* vis_groups;    /* optional [nrays, clip]: the partial visibility hint per sample group (NrhNet.shadow_clip > 0; the
                           `visibilities` output is only the value at the maximal-weight sample); NULL = kept in the workspace */
  void* save_h16;       /* optional, both or neither (nrh_sdf_train_forward_half below): 16-bit hand-offs of h ...             */
  void* save_t16;       /* ... and t to nrh_dw_gemm, fp16 [8][nrays*128][256] in the half-tiled layout                          */
  float* pts;           /* optional [nrays*128,3]: p = o + d * mid_z, the reflectance net's point input (ABI 148: written by the alpha
                           stage, which forms the point anyway - it replaced two elementwise launches of the caller); NULL = not wanted */
} NrhTrainSaves;
int nrh_render_forward_train(const NrhNet* net, const float* origins, const float* directions, const float* pl_positions,
                             const float* nears, const float* fars, long long nrays, float cos_anneal,
                             const float* t_rand_primary, const float* t_rand_shadow, int zero_hints, const float* lin64,
                             const float* lin16, float* depth, float* weights, float* inside_sphere, float* analytic_normals,
                             float* normalized_normals, float* visibilities, float* specular_cue, float* mid_z, float* dists,
                             const NrhTrainSaves* saves, float* workspace, long long workspace_floats, void* stream);

/* ---- pixel -> ray for one pinhole view ------------------------------------------------------------------------
 * RayGenerator.forward without pose / light deltas (camera/ray_generator.py:79-139): rows [row0, row0+nrows) of a
 * `width`-wide image, pose = row-major [3,4] camera-to-world (HOST pointer, 12 floats), pl = light position (HOST, 3).
 * Writes origins/directions/pl_positions [nrows*width,3] and nears/fars [nrows*width] (unit-sphere near/far, :135-139). */
int nrh_generate_rays(const float* pose, const float* pl, float cx, float cy, float fx, float fy, int width, int row0,
                      int nrows, float* origins, float* directions, float* pl_positions, float* nears, float* fars,
                      void* stream);

/* ---- pixel bundle -> rays with per-view refinement ---------------------------------------------------------------
 * RayGenerator.forward (camera/ray_generator.py:75-150) for a training / evaluation pixel bundle: ray i has pixel
 * (h_indices[i], w_indices[i]) (floats, as RawPixelBundle stores them), camera-to-world poses[i*pose_stride .. +12)
 * (row-major [3,4]; pose_stride 12 or 16) and light pls[i*3..].  img_indices [n] (int64) selects the view's entry of
 * `delta` [ncam,3,4] - the left delta exp(adjustment) o noise, composed per VIEW by the caller (:108-121) - and of
 * `pl_delta` [ncam,3] (:123-127); either may be NULL, and img_indices NULL means "no refinement" (:103-105).
 * near_far_from_sphere != 0: unit-sphere chord mid-point -+ 1 (:135-139), else the constants zn / zf (:141-142).
 * All pointers are DEVICE pointers.  The backward entry ACCUMULATES d(loss)/d(delta) [ncam,3,4] and
 * d(loss)/d(pl_delta) [ncam,3] (either may be NULL; caller zeroes) from the output gradients (any may be NULL). */
int nrh_generate_rays_indexed(const long long* img_indices, const float* h_indices, const float* w_indices, const float* poses,
                              int pose_stride, const float* pls, long long nrays, const float* delta, const float* pl_delta,
                              int ncam, float cx, float cy, float fx, float fy, int near_far_from_sphere, float zn, float zf,
                              float* origins, float* directions, float* pl_positions, float* nears, float* fars, void* stream);
int nrh_generate_rays_indexed_backward(const long long* img_indices, const float* h_indices, const float* w_indices,
                                       const float* poses, int pose_stride, const float* pls, long long nrays, const float* delta,
                                       const float* pl_delta, int ncam, float cx, float cy, float fx, float fy,
                                       int near_far_from_sphere, const float* g_origins, const float* g_directions,
                                       const float* g_pl_positions, const float* g_nears, const float* g_fars, float* g_delta,
                                       float* g_pl_delta, void* stream);

/* Bytes of the block stream NrhNet.col_w32 points to (33 blocks of 32 KiB: layer 0 without its feature block, layers 1-3,
 * the 3-row output layer; layout in nrhints_amd/packing32.py, pack_color32). */
long long nrh_color_wide_stream_bytes(void);
/* The reflectance net of the hinted model on the wide kernel (what nrh_render_forward runs when NrhNet.col_w32 is set):
 * ReflectanceNetwork.forward (fields/reflectance_network.py:68-96) at the 128 samples of `nrays` rays, with the feature block of
 * its first layer given as `part_tiles` = W0[:, 60:316] * feature (+ b_feat folded), D-layout tiles of 16 points as nrh_sdf_eval
 * mode 2 writes them with the fused streams (NrhNet.feat_fused); the other arguments as nrh_color_eval (raymisc: rows of 100
 * floats, 99 used). */
int nrh_color_eval_wide(const void* col_w32, const float* col_tab32, const float* part_tiles, const float* ro, const float* rd,
                        const float* tmid, const float* nhat, const float* raymisc, long long nrays, float* color, void* stream);

/* ---- the per-ray stages of the evaluation render, one entry per kernel (what nrh_render_forward launches between its
 * network evaluations; unit-testable against the reference's recorded intermediates, tests/golden/core_*.npz) -------------
 * nrh_alpha_composite: NeuSHintRenderer.get_alpha + the compositing weights, depth / hit point, hit normal and Cook-Torrance
 *   cue of render_core (models/neus_hint_model.py:339-356, :512-533, :583-616) and the shadow ray's set-up (:380-395).
 *   In: origins / directions / pl_positions [n,3]; sdf, dists, mid_z [n,128]; grad [n*128,3] (d sdf/dx at the mid-points);
 *   inv_s, cos_anneal; depth_type 0 AlphaBlend | 1 MaximalWeightPoint; zero_hints != 0: cue = 0 (geometry warm-up);
 *   lin64 = torch.linspace(0,1,64); t_rand_shadow [n,64] or NULL (training jitter of the coarse shadow samples).
 *   Out: weights, inside_sphere [n,128]; normalized_normals [n*128,3]; depth, weight_sum [n]; specular_cue [n,4] (per ray);
 *   hit_points, hit_normals [n,3] (either may be NULL); shadow_dirs [n,3] = unit(hit - pl), shadow_last_dist [n] = |hit - pl| / 64,
 *   shadow_z [n,128] (first 64: coarse shadow samples lin64 * |hit - pl| * (1 - 1e-2)).
 * nrh_visibility: the tail of get_visibility (:417-432): alpha along the shadow ray from sdf / grad / dists [n,128 | n*128,3]
 *   at its 128 section mid-points, visibilities [n] = transmittance in front of the last sample; plus the reflectance net's
 *   per-ray inputs raymisc [n, 100] = enc4(view dir) 27 | enc4(light position) 27 | enc4(visibility) 9 | enc4(cue) 36
 *   (fields/reflectance_network.py:70-84; fields/encodings.py:168-174).  zero_hints != 0: visibility and cue are zero.
 * nrh_color_composite: rgb = sum_j c_j w_j + background (1 - sum_j w_j) (:635-637; background [3] or NULL) and, when asked
 *   for, the per-pixel normal maps sum_j n_j w_j inside_j of the evaluation loop (pipelines/base_pipeline.py:125-131). */
long long nrh_sphere_trace_workspace_floats(long long nrays);
/* NeuSHintRenderer.sphere_trace (models/neus_hint_model.py:359-372; what DepthComputationType.SphereTracing calls with
 * iterations 2000, threshold 1e-4, far 100, :527-528): from the ray origins, every ray advances by the SDF at its point until
 * |sdf| < threshold or its travelled depth exceeds far_depth.  Out: points [n,3], depths [n].  One SDF launch + one step kernel
 * per iteration; the "a ray moved" flag is read back every 4th iteration (a stream synchronisation - not graph-capturable),
 * and the loop ends at the first read-back that saw no movement (further iterations would change nothing).
 * nrh_render_forward runs it by itself when NrhNet.depth_type == 2. */
int nrh_sphere_trace(const NrhNet* net, const float* origins, const float* directions, long long nrays, int iterations,
                     float threshold, float far_depth, float* points, float* depths, float* workspace, long long workspace_floats,
                     void* stream);
int nrh_alpha_composite(const float* origins, const float* directions, const float* pl_positions, const float* sdf, const float* grad,
                        const float* dists, const float* mid_z, float inv_s, float cos_anneal, int depth_type, int zero_hints,
                        const float* lin64, const float* t_rand_shadow, long long nrays, float* weights, float* inside_sphere,
                        float* normalized_normals, float* depth, float* weight_sum, float* specular_cue, float* hit_points,
                        float* hit_normals, float* shadow_dirs, float* shadow_last_dist, float* shadow_z, void* stream);
int nrh_visibility(const float* directions, const float* pl_positions, const float* shadow_dirs, const float* sdf, const float* grad,
                   const float* dists, const float* specular_cue, float inv_s, float cos_anneal, int zero_hints, long long nrays,
                   float* visibilities, float* raymisc, void* stream);
int nrh_color_composite(const float* sampled_color, const float* weights, const float* weight_sum, const float* background,
                        const float* inside_sphere, const float* analytic_normals, const float* normalized_normals, long long nrays,
                        float* rgb, float* normal_map, float* normalized_normal_map, void* stream);

/* ---- the fused training step: weight gradients, loss and adjoint seeds without library GEMMs or framework ops ---------------
 * nrh_dw_gemm: every dW = X^T Y of loss.backward() through the nn.Linear layers (fields/sdf_field.py:81-101,
 *   fields/reflectance_network.py:52-66) as split-K MFMA GEMMs (v_mfma_f32_32x32x16_bf16, hi/lo 3-term products, fp32
 *   accumulation) in ONE launch + one deterministic reduction launch.  A job is  out[i][j] = scale * sum_pairs sum_p A_k[p][i] B_k[p][j]
 *   over the row-major arrays A_k [npts, lda_k], B_k [npts, ldb_k] (channels i < m, j < n exist, m, n <= 256; up to two pairs
 *   accumulate into one product: dW_l = zbar_l^T x_l + t_l^T abar_l), written for i < rows, j < cols to out[i * ldo + col_map[j]]
 *   (col_map NULL = identity) or, transposed, to out[j * ldo + i].  colsum_a / colsum_b (optional, [m] / [n]): scale_a * sum_p A_0[p][i]
 *   and scale_b * sum_p B_0[p][j] - the bias gradients come for free.  `slabs` = work items the points are split into (the
 *   launch has sum(slabs) workgroups: about one per CU in total).  `jobs` is a HOST array; all data pointers are DEVICE pointers.
 *   workspace: nrh_dw_workspace_floats(jobs, njobs) floats, 16-byte aligned.  npts must be a multiple of 32. */
typedef struct NrhDwJob {
  const float* a[2];
  const float* b[2];
  int lda[2], ldb[2];
  int npairs, m, n, slabs;
  float* out;
  const int* col_map;
  int ldo, transpose, rows, cols;
  float scale;
  float* colsum_a;
  float scale_a;
  float* colsum_b;
  float scale_b;
  int tiled_a[2];   /* pair k: the operand is in the TILED layout of the training arrays instead of row-major (256 channels only): */
  int tiled_b[2];   /* [tile of 16 points][block of 16 channels][point 16][16 channels] - what nrh_sdf_train_forward / _backward    */
                    /* write save_h, save_t, abar, zbar in when nrh_train_arrays_tiled() says so                                  */
  int half_ops;     /* 1: all operands of the job are fp16 arrays in the HALF-TILED layout ([tile of 16 points][block pair 8][point 16] */
                    /* [quarter 4][block of the pair 2][4 channels], 8 KiB per tile; NrhTrainHalf) - full 256 x 256 products only;     */
                    /* one fp16 MFMA per K step instead of three bf16 ones (the a / b pointers are cast, lda = ldb = 256)              */
  const float* dyn_scale;   /* optional DEVICE pointer {S, 1 / S} (nrh_adjoint_range): out and colsum_a are multiplied by dyn_scale[1] */
} NrhDwJob;
/* 1 if save_h, save_t (nrh_sdf_train_forward), abar and zbar (nrh_sdf_train_backward) are tiled - opaque hand-offs between those
 * kernels and nrh_dw_gemm, which a caller passes on with tiled_a / tiled_b set - 0 if they are row-major [layer][npts][256].
 * (save_s1 and coup are private to the kernels either way.) */
int nrh_train_arrays_tiled(void);
long long nrh_dw_workspace_floats(const NrhDwJob* jobs, int njobs);
int nrh_dw_gemm(const NrhDwJob* jobs, int njobs, long long npts, float* workspace, long long workspace_floats, void* stream);
/* enc_6(3 p) of the points p = ro[ray] + rd[ray] * t[ray * t_stride + j] as rows [npts][64] (39 used, fields/encodings.py:168-174):
 * the B operand of layer 0's weight gradient. */
int nrh_embedding_rows(const float* ro, const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays, float* rows,
                       void* stream);
/* rgb = sum_j c_j w_j + background (1 - sum w) (models/neus_hint_model.py:635-637), the per-ray partial sums of the loss terms of
 * pipelines/base_pipeline.py:57-62 (partials [n,4]: sum_c |rgb - gt|, sum_j inside (|g| - 1)^2, sum_j inside, sum_c (rgb - gt)^2) and
 * the adjoint seeds of d loss: zbar_out [n*128,3] = w dl/drgb c (1 - c) (through the output sigmoid) and weights_bar [n,128].
 * nrh_loss_finish reduces the partials (deterministic order) to out8 = {loss, rgb_loss, eikonal_loss, s_val, psnr,
 * igr_weight / (sum inside + 1e-5), 0, 0}; entry 5 is the eikonal seed coefficient nrh_alpha_train_backward_fused takes. */
int nrh_composite_loss(const float* sampled_color, const float* weights, const float* rgb_gt, const float* background,
                       const float* analytic_normals, const float* inside_sphere, long long nrays, float* rgb, float* zbar_out,
                       float* weights_bar, float* partials, void* stream);
/* Adjoints of the rays for pose / light refinement (ray_bundle.origins / directions / pl_positions of the reference's autograd,
 * camera/ray_generator.py:75-150 downstream of pipelines/base_pipeline.py:41-69 and :80-85), from what the sweeps left behind:
 * pbar [nrays*128,3] (nrh_sdf_train_backward), gbar [nrays*128,3] (the spatial gradient's adjoint: nrh_alpha_train_backward*),
 * save_ge [nrays*128,128] (NrhTrainSaves), mbar [nrays*128,mbar_width] (nrh_color_train_backward) and rd_bar [nrays,3] (the
 * alpha stage).  Feeds nrh_generate_rays_indexed_backward. */
int nrh_ray_adjoint(const float* origins, const float* directions, const float* pl_positions, const float* mid_z, const float* pbar,
                    const float* gbar, const float* save_ge, const float* mbar, int mbar_width, const float* rd_bar, long long nrays,
                    float* origins_bar, float* directions_bar, float* pl_bar, void* stream);
int nrh_loss_finish(const float* partials, long long nrays, float inv_s, const float* dyn_scalars, float igr_weight, float* out8,
                    void* stream);
/* nrh_alpha_train_backward with (a) strided rows of nhat_bar (the normal's three columns inside the reflectance adjoint's output)
 * and (b) the eikonal term's seed added to grad_bar: eikonal_coef[0] * inside * 2 (|g| - 1) g / |g| (both NULL = plain adjoint).
 * n_real: samples per ray that exist (2 .. 128, NrhNet.samples; the padded ones behind them carry and receive nothing). */
int nrh_alpha_train_backward_fused(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                                   float cos_anneal, const float* dyn_scalars, long long nrays, const float* weights_bar,
                                   const float* nhat_bar, int nhat_bar_stride, const float* inside_sphere, const float* eikonal_coef,
                                   float* sdf_bar, float* grad_bar, float* rd_bar, float* invs_bar, int n_real, void* stream);
/* The same stage for a SHADOW ray (renderer.shadow_hint_gradient, models/neus_hint_model.py:379, :417-432): alpha from sdf /
 * grad / dists at its 128 sections, visibilities [n] = transmittance in front of the last sample; the adjoint takes
 * d loss / d visibility [n] and returns the adjoints of sdf [n,128], grad [n*128,3], the shadow ray's direction [n,3] and the
 * per-ray partial of d loss / d inv_s [n] (feed nrh_variance_grad). */
/* ---- renderer.use_outside_nerf: what surrounds the caller's background network ------------------------------------------------
 * nrh_sample_primary: coarse z + the four hierarchical sampling steps of NeuSHintRenderer.forward (:673-713) on their own:
 *   z_vals [n,128] sorted sample positions, mid_z / dists [n,128] as nrh_render_forward computes them (the same call sequence,
 *   so a later nrh_render_forward on the same inputs places the same samples).  workspace: >= 128 n + 32 n + 192 floats.
 * nrh_alpha_blend_forward / _backward: nrh_alpha_train_* with alpha <- alpha inside_sphere + bg_alpha[:, :128] (1 - inside_sphere)
 *   (bg_alpha row stride 160) and the transmittance behind the last sample as an extra output (tail_t [n]) / adjoint input; the
 *   adjoint also returns d loss / d bg_alpha[:, :128] (bg_alpha_bar [n,128]). */
int nrh_sample_primary(const NrhNet* net, const float* origins, const float* directions, const float* nears, const float* fars,
                       long long nrays, const float* t_rand_primary, const float* lin64, const float* lin16, float* z_vals, float* mid_z,
                       float* dists, float* workspace, long long workspace_floats, void* stream);
int nrh_alpha_blend_forward(const float* sdf, const float* grad, const float* rd, const float* dists, const float* inside_sphere,
                            const float* bg_alpha, float inv_s, float cos_anneal, const float* dyn_scalars, long long nrays,
                            float* weights, float* nhat, float* tail_t, void* stream);
int nrh_alpha_blend_backward(const float* sdf, const float* grad, const float* rd, const float* dists, const float* inside_sphere,
                             const float* bg_alpha, float inv_s, float cos_anneal, const float* dyn_scalars, long long nrays,
                             const float* weights_bar, const float* nhat_bar, const float* tail_t_bar, float* sdf_bar, float* grad_bar,
                             float* rd_bar, float* invs_bar, float* bg_alpha_bar, void* stream);
/* nrh_alpha_train_forward / _backward for rays with n_real = 64 or 128 existing samples of the 128 slots (NrhNet.samples):
 * padded samples have alpha = 0 and receive zero adjoints. */
int nrh_alpha_train_forward_n(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                              float cos_anneal, const float* dyn_scalars, long long nrays, int n_real, float* weights, float* nhat,
                              void* stream);
int nrh_alpha_train_backward_n(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                               float cos_anneal, const float* dyn_scalars, long long nrays, int n_real, const float* weights_bar,
                               const float* nhat_bar, float* sdf_bar, float* grad_bar, float* rd_bar, float* invs_bar, void* stream);
/* The shadow ray's alpha stage for renderer.shadow_hint_gradient (models/neus_hint_model.py:379, :411-432): visibility =
 * transmittance in front of the LAST sample that exists, taus[..., -1].  n_real = n_shadow_samples + 4 * (n_shadow_importance_samples
 * // 4) of the 128 slots (128 with the reference's defaults); padded slots behind it have alpha = 0, do not enter the product
 * and receive zero adjoints (ABI 147: the argument is new; before it the product always ran to slot 127). */
int nrh_shadow_alpha_forward(const float* sdf, const float* grad, const float* shadow_dirs, const float* dists, float inv_s,
                             float cos_anneal, const float* dyn_scalars, long long nrays, int n_real, float* visibilities, void* stream);
int nrh_shadow_alpha_backward(const float* sdf, const float* grad, const float* shadow_dirs, const float* dists, float inv_s,
                              float cos_anneal, const float* dyn_scalars, long long nrays, int n_real, const float* visibilities_bar,
                              float* sdf_bar, float* grad_bar, float* dirs_bar, float* invs_bar, void* stream);
/* d loss / d variance from the per-ray partials of nrh_alpha_train_backward (inv_s = clip(exp(10 variance), 1e-6, 1e6),
 * models/neus_hint_model.py:104-110): variance_bar[0] = 10 inv_s sum(invs_bar) inside the clip range, else 0. */
int nrh_variance_grad(const float* invs_bar, long long nrays, float inv_s, const float* dyn_scalars, float* variance_bar, void* stream);
/* The scalars of a training step in ONE launch (ABI 148): values[i] -> *dst[i] for i < n <= 4 (dst, values: HOST arrays; the
 * addresses are device pointers - the cos-anneal ratio models/neus_hint_model.py:669-671, the two learning rates), and, with
 * `variance` (device [1]) non-null, inv_s_out[0] = clip(exp(10 variance), 1e-6, 1e6) (SingleVarianceNetwork, :104-110, :337-338). */
int nrh_step_scalars(float* const* dst, const float* values, int n, const float* variance, float* inv_s_out, void* stream);

/* ---- the optimiser step (trainer/trainer.py:99-102, 281: torch.optim.Adam over two parameter groups) in one launch ------------
 * Arithmetic of torch.optim.Adam's default implementation (what the reference runs; bias corrections in double), state layout
 * of its capturable one (float32 `step` scalar per tensor on the device, exp_avg `m`, exp_avg_sq `v`; no amsgrad / weight
 * decay / maximize):
 *   step += 1;  m += (g - m)(1 - b1);  v = v b2 + (1 - b2) g g;
 *   p += -(lr / (1 - b1^step)) * (m / (sqrt(v) / sqrt(1 - b2^step) + eps))
 * tensors_dev: DEVICE array of `ntensors` descriptors; chunks_dev: DEVICE int pairs (tensor index, chunk index), one per block of
 * 2048 elements, `nchunks` of them (sum over tensors of ceil(n / 2048)); per-group (`ngroups` <= 4) HOST arrays lr / beta1 /
 * beta2 / eps, and optionally lr_dev: HOST array of device pointers to the group's learning rate (read at run time: hipGraph
 * replays follow the schedule without re-capture), NULL entries fall back to lr[g]. */
typedef struct NrhAdamTensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  float* step;
  long long n;
  int group;
  int reserved;
} NrhAdamTensor;
int nrh_adam_step(const NrhAdamTensor* tensors_dev, int ntensors, const int* chunks_dev, int nchunks, int ngroups, const double* lr,
                  const float* const* lr_dev, const double* beta1, const double* beta2, const double* eps, void* stream);

/* ---- re-packing of the kernel buffers after an optimiser step (one launch per buffer instead of ~70 framework ops) ------------
 * nrh_pack_gather: out[e] = f(flat[index[e]]) for e < n over a fixed index plan (nrhints_amd/packing.py PackPlan, packing32.py
 *   PackPlan32; index 0 addresses the constant 0.0 in front of the flattened weights): mode 0 float32 copy; mode 1 x = v / factor[e];
 *   mode 2 x = v * factor[e]; modes 1, 2 write fp16 (hi, lo = (x - hi) * 2^11) as 512-element blocks [hi | lo] (n % 512 == 0).
 *   Same roundings as the torch expressions of the packers: the result is bit-identical to them.
 * nrh_sdf32_tables: packing32.sdf32_tables - the [11][256] bias / head tables of the wide SDF kernels from the eight bias vectors
 *   (HOST array of 8 device pointers + their lengths), b_feat [256], b_s [1], w_s [256]. */
/* packing32.fuse_feature_head: out_w [256,256] = W0[:, 60:316] W_feat, out_b [256] = W0[:, 60:316] b_feat (float64 accumulation,
 * one rounding): the feature head multiplied into the feature block of the reflectance net's first layer (col_w0 [256, ld0]
 * row-major, ld0 = 316 or 361) - what NrhNet.feat_fused expects in the FEAT block of the wide streams. */
int nrh_fuse_feature_head(const float* col_w0, int ld0, const float* feat_w, const float* feat_b, float* out_w, float* out_b, void* stream);
int nrh_pack_gather(const float* flat, const int* index, const float* factor, long long n, int mode, void* out, void* stream);
int nrh_sdf32_tables(const float* const* sdf_bias, const int* rows, const float* feat_b, const float* head_b, const float* head_w,
                     float* tables, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NRHINTS_HIP_H */
