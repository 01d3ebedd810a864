/*
 * lnrf.h — C ABI of the MI355X-native NeRF volume-rendering hot path (liblnrf.so).
 *
 * The reference (unixpickle/learn-nerf) has no FFI: its hot path is Python on JAX
 * (SURVEY.md §8b).  This header is the boundary a maintainer binds instead of
 * jax.jit: every entry point names the reference function (file:line under
 * /root/reference/learn_nerf) whose arithmetic it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / HIP types in signatures
 *     (lnrf_stream_t is a hipStream_t passed as void*; NULL = default stream).
 *   - every pointer is DEVICE memory unless marked (host); the caller owns every
 *     buffer, the library allocates nothing persistent.
 *   - all work is enqueued asynchronously on the given stream; no hidden sync.
 *   - return 0 = ok, <0 = lnrf argument/shape error, >0 = hipError_t passthrough;
 *     message via lnrf_last_error() (thread-local).  Never throws or aborts.
 *   - arrays are row-major contiguous fp32 unless noted; rays are
 *     (origin[3], direction[3]) with `ray_stride` floats between consecutive rays
 *     (6 for [N,2,3] batches, 9 for [N,3,3] training batches with colours).
 *   - sampling noise: `u` (explicit uniforms in [0,1)) or, when u == NULL, the
 *     Philox4x32-10 stream (seed, stream_id, element (ray_offset+n)*count+i)
 *     documented in oracle/philox.py.
 */
#ifndef LNRF_H
#define LNRF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lnrf_stream_t;

#define LNRF_VERSION 100 /* 0.1.0 */

#define LNRF_OK 0
#define LNRF_ERR_ARG (-1)
#define LNRF_ERR_SHAPE (-2)
#define LNRF_ERR_UNSUPPORTED (-3)

/* activation codes for the dense kernels */
#define LNRF_ACT_NONE 0
#define LNRF_ACT_RELU 1
#define LNRF_ACT_SOFTPLUS 2
#define LNRF_ACT_TANH 3
#define LNRF_ACT_EXP 4
#define LNRF_ACT_SIGMOID 5

int lnrf_version(void);
const char* lnrf_last_error(void);

/* ------------------------------------------------------------------ rays ---- */

/* ray_t_range (render.py:346-389) + NeRFRenderer.t_range (render.py:93-111) fused with
 * RaySamples.stratified_sampling (render.py:121-143).  bbox_min/max: (host) 3 floats.
 * count may be 0 (then ts/u are ignored).  mask: 1 byte per ray (0/1). */
int lnrf_ray_aabb_stratified(const float* rays, int64_t ray_stride, int64_t n_rays,
                             const float* bbox_min, const float* bbox_max, float min_t_range,
                             float epsilon, int32_t count, const float* u, uint64_t seed,
                             uint32_t stream_id, int64_t ray_offset, float* t_min, float* t_max,
                             uint8_t* mask, float* ts, lnrf_stream_t stream);

/* CameraView.bare_rays (dataset.py:52-78): all rays of a pinhole view in raster order, rays[H*W, 2, 3]
 * (device).  origin / axes: (host) 3 floats each; fields of view in radians. */
int lnrf_camera_rays(const float* origin, const float* x_axis, const float* y_axis, const float* z_axis,
                     float x_fov, float y_fov, int32_t width, int32_t height, float* rays,
                     lnrf_stream_t stream);

/* The shuffled batch iterator's row movement (ShuffledDataset.iterate_batches, dataset.py:222-240: the reference
 * permutes a shard and concatenates it to the pending rows on the host for every batch): with the shards resident in
 * HBM, out[i, :] = src[idx[i], :] for i < n, rows of row_floats fp32 (9 = origin, direction, colour).  idx: int32
 * row numbers in [0, n_src); a row number outside that range writes zeros (never reads out of bounds). */
int lnrf_gather_rows(const float* src, int64_t n_src, int32_t row_floats, const int32_t* idx, int64_t n,
                     float* out, lnrf_stream_t stream);

/* RaySamples.stratified_sampling (render.py:121-143) from given t_min/t_max. */
int lnrf_stratified(const float* t_min, const float* t_max, int64_t n_rays, int32_t count,
                    const float* u, uint64_t seed, uint32_t stream_id, int64_t ray_offset,
                    float* ts, lnrf_stream_t stream);

/* RaySamples.points (render.py:145-153) + direction tile (render.py:319):
 * x[n,t,:] = o + d*ts[n,t];  dirs[n,t,:] = d.  Either output may be NULL. */
int lnrf_ray_points(const float* rays, int64_t ray_stride, const float* ts, int64_t n_rays,
                    int32_t t, float* points, float* dirs, lnrf_stream_t stream);

/* RaySamples.fine_sampling (render.py:211-257): termination probs of the coarse pass ->
 * inverse-CDF (jnp.interp) at stratified u' -> optional sort with the coarse ts.
 * ts_out is [N, tc+tf] when combine != 0 else [N, tf]. */
int lnrf_fine_sample(const float* ts_c, const float* t_min, const float* t_max,
                     const float* density_c, int64_t n_rays, int32_t tc, int32_t tf, float eps,
                     int32_t combine, const float* u, uint64_t seed, uint32_t stream_id,
                     int64_t ray_offset, float* ts_out, lnrf_stream_t stream);

/* RaySamples.starts / ends (render.py:259-265): bin edges [N,T] each (either may be NULL). */
int lnrf_bin_edges(const float* ts, const float* t_min, const float* t_max, int64_t n_rays,
                   int32_t t, float* starts, float* ends, lnrf_stream_t stream);

/* RaySamples.termination_probs (render.py:270-287): probs [N, T+1]. */
int lnrf_termination_probs(const float* ts, const float* t_min, const float* t_max,
                           const float* density, int64_t n_rays, int32_t t, float* probs,
                           lnrf_stream_t stream);

/* RaySamples.render_rays / render_alpha / average_aux_losses and the free render_rays
 * (render.py:155-209, 293-343) in one pass per ray:
 *   outputs[N,3] = mask ? sum_t probs*rgb + probs_T*background : background
 *   alphas[N]    = mask ? 1 - probs_T : 0
 *   coords[N,3]  = mask ? sum_t probs*(o + d*ts) : 0
 *   aux_sum[N,n_aux] = mask ? sum_t probs*aux : 0           (aux [N,T,n_aux], may be NULL)
 * If targets != NULL (stride target_stride floats per ray), sum over rays/channels of
 * (outputs-targets)^2 is atomically added to *sq_err (train.py:141-142 numerator).
 * Any output pointer may be NULL. */
int lnrf_composite_fwd(const float* rays, int64_t ray_stride, const float* ts, const float* t_min,
                       const float* t_max, const uint8_t* mask, const float* density,
                       const float* rgb, const float* aux, int32_t n_aux, const float* background,
                       int64_t n_rays, int32_t t, float* outputs, float* alphas, float* coords,
                       float* aux_sum, const float* targets, int64_t target_stride, float* sq_err,
                       lnrf_stream_t stream);

/* Backward of the compositing integral for the loss of TrainLoop.losses (train.py:140-151).
 * Upstream gradient per ray: g_out[N,3] if non-NULL, else out_scale*(outputs - targets)
 * (i.e. d/d outputs of mean squared error when out_scale = 2/(3*N_global)).
 * g_aux_w: (host) n_aux weights = d total / d aux_sum[n,k] (same for every ray).
 * Writes g_density[N,T], g_rgb[N,T,3], g_aux[N,T,n_aux]; atomically accumulates
 * g_background[3].  No gradient flows to ts (fine sampling uses stop_gradient,
 * render.py:76; stratified ts do not depend on parameters). */
int lnrf_composite_bwd(const float* ts, const float* t_min, const float* t_max,
                       const uint8_t* mask, const float* density, const float* rgb,
                       const float* aux, int32_t n_aux, const float* background, int64_t n_rays,
                       int32_t t, const float* g_out, const float* outputs, const float* targets,
                       int64_t target_stride, float out_scale, const float* g_aux_w,
                       float* g_density, float* g_rgb, float* g_aux, float* g_background,
                       lnrf_stream_t stream);

/* The same, with the background gradient (the one sum over rays this call forms, render.py:170-176) added in a fixed
 * order: per-workgroup partial sums in `scratch` (lnrf_composite_bwd_scratch_bytes(n_rays) bytes, 16-byte aligned), folded
 * by a second launch — bit-reproducible, where lnrf_composite_bwd lets the workgroups meet in fp32 atomics. */
int64_t lnrf_composite_bwd_scratch_bytes(int64_t n_rays);
int lnrf_composite_bwd_det(const float* ts, const float* t_min, const float* t_max, const uint8_t* mask,
                           const float* density, const float* rgb, const float* aux, int32_t n_aux,
                           const float* background, int64_t n_rays, int32_t t, const float* g_out, const float* outputs,
                           const float* targets, int64_t target_stride, float out_scale, const float* g_aux_w,
                           float* g_density, float* g_rgb, float* g_aux, float* g_background, void* scratch,
                           int64_t scratch_bytes, lnrf_stream_t stream);

/* ----------------------------------------------------------- generic dense ---- */

/* sinusoidal_emb (model.py:65-77): out[m, col_off + c*2F + {f | F+f}] = sin|cos(2^f x[m,c]).
 * x has `dims` columns (row stride ldx), out row stride ldo. */
int lnrf_sinusoidal_emb(const float* x, int64_t ldx, int64_t m, int32_t dims, int32_t freqs,
                        float* out, int64_t ldo, int64_t col_off, lnrf_stream_t stream);

/* Operand precision of the generic dense path (lnrf_dense_fwd / _bwd_input / _bwd_weight / lnrf_gemm_f32), a
 * thread-local mode like a rounding mode.  LNRF_DENSE_FP32 (default): exact fp32 products on the f32 MFMA, the
 * arithmetic of the reference's jnp.float32 Dense layers.  LNRF_DENSE_BF16: both operands of every product are
 * rounded to bf16 (round-to-nearest-even) when staged, fp32 accumulate / bias / activation — the arithmetic of the
 * fused kernels, for the models that have no fused kernel (RefNERFModel, InstantNGPRefNERFModel, non-default
 * shapes). */
#define LNRF_DENSE_FP32 0
#define LNRF_DENSE_BF16 1
int lnrf_set_dense_precision(int32_t precision);
int32_t lnrf_get_dense_precision(void);

/* flax.linen.Dense + activation (model.py:51-60): y = act(x[M,K] @ w[K,N] + b[N]).
 * fp32 in / fp32 accumulate on the f32 MFMA.  ldx/ldy row strides (concat without copies).
 * b may be NULL. */
int lnrf_dense_fwd(const float* x, int64_t ldx, const float* w, const float* b, int32_t act,
                   float* y, int64_t ldy, int64_t m, int32_t k, int32_t n, lnrf_stream_t stream);

/* g_pre = g_y * act'(y) elementwise in place on g (y = saved activation output). */
int lnrf_act_bwd(float* g, int64_t ldg, const float* y, int64_t ldy, int32_t act, int64_t m,
                 int32_t n, lnrf_stream_t stream);

/* g_x[M,K] (+)= g_y[M,N] @ w[K,N]^T. accumulate != 0 adds into g_x. */
int lnrf_dense_bwd_input(const float* gy, int64_t ldgy, const float* w, float* gx, int64_t ldgx,
                         int32_t accumulate, int64_t m, int32_t k, int32_t n, lnrf_stream_t stream);

/* lnrf_dense_bwd_input fused with the activation backward of the layer below:
 * gx[i][c] (+)= (sum_r gy[i][r] w[c][r]) * act'(y_below[i][c]) for c < n_gated (act' expressed through the activation's
 * OUTPUT, as lnrf_act_bwd), plain input gradient for the other columns (e.g. the x_emb part of a concatenated input).
 * Saves one read-modify-write pass over gx per layer. */
int lnrf_dense_bwd_input_gated(const float* gy, int64_t ldgy, const float* w, const float* y_below, int64_t ldy,
                               int32_t act_below, int32_t n_gated, float* gx, int64_t ldgx, int32_t accumulate,
                               int64_t m, int32_t k, int32_t n, lnrf_stream_t stream);
/* lnrf_dense_fwd whose result is multiplied by act_gate'(y_gate[i][j]): the masked forward product of the
 * second-order (normal) chain of ref_nerf.py:38-43, tbar_l = relu'(h_l) * (tbar_{l-1} W_l). */
int lnrf_dense_fwd_gated(const float* x, int64_t ldx, const float* w, const float* b, int32_t act,
                         const float* y_gate, int64_t ldg, int32_t act_gate, float* y, int64_t ldy, int64_t m,
                         int32_t k, int32_t n, lnrf_stream_t stream);

/* g_w[K,N] += x[M,K]^T @ g_y[M,N];  g_b[N] += sum_m g_y (g_b may be NULL).
 * x == NULL and gw == NULL: bias gradient only. */
int lnrf_dense_bwd_weight(const float* x, int64_t ldx, const float* gy, int64_t ldgy, float* gw,
                          float* gb, int64_t m, int32_t k, int32_t n, lnrf_stream_t stream);

/* The same with a FIXED summation order (bit-reproducible): the splits of the reduction over m leave their partial
 * sums in `scratch` (lnrf_dense_bwd_weight_scratch_bytes(m, k, n) bytes; k = 0 when x / gw are null) and a second launch
 * adds them in order, instead of fp32 atomics whose arrival order changes from run to run.  What the package's exact-fp32
 * path uses (reference: jax.grad through nn.Dense, model.py:51-60, is deterministic on one device). */
int64_t lnrf_dense_bwd_weight_scratch_bytes(int64_t m, int32_t k, int32_t n);
int lnrf_dense_bwd_weight_det(const float* x, int64_t ldx, const float* gy, int64_t ldgy, float* gw, float* gb,
                              int64_t m, int32_t k, int32_t n, void* scratch, int64_t scratch_bytes,
                              lnrf_stream_t stream);

/* General strided fp32 GEMM on the f32 MFMA behind the dense entry points above:
 *   C[i*ldc + j] (op)= sum_r A[i*sa_i + r*sa_r] * B[r*sb_r + j*sb_j],  i < I, j < J, r < R
 * mode 0: C = act(sum + bias[j]);  1: C += sum;  2: atomic C += sum with the reduction split over
 * `splits` workgroups (0 = choose).  Lets a layer consume a transposed operand (e.g. the feature-major
 * hash-grid encoding) without a copy. */
int lnrf_gemm_f32(const float* a, int64_t sa_i, int64_t sa_r, const float* b, int64_t sb_r, int64_t sb_j,
                  float* c, int64_t ldc, const float* bias, int32_t act, int32_t mode, int64_t i_rows,
                  int32_t j_cols, int64_t r_depth, int32_t splits, lnrf_stream_t stream);

/* lnrf_gemm_f32 mode 2 (C += A B, reduction split over workgroups) with a FIXED summation order: C has contiguous
 * rows (ldc == J); the splits leave their partial tiles in `scratch` (lnrf_gemm_f32_det_scratch_bytes bytes) and a
 * second launch adds them in order — bit-reproducible, unlike the atomic form. */
int64_t lnrf_gemm_f32_det_scratch_bytes(int64_t i_rows, int32_t j_cols, int64_t r_depth);
int lnrf_gemm_f32_det(const float* a, int64_t sa_i, int64_t sa_r, const float* b, int64_t sb_r, int64_t sb_j, float* c,
                      int64_t i_rows, int32_t j_cols, int64_t r_depth, void* scratch, int64_t scratch_bytes,
                      lnrf_stream_t stream);

/* --------------------------------------------------- hash-grid encoding ---- */

/* MultiresHashTableEncoding / HashTableEncoding (instant_ngp.py:92-208). Tables of all levels live in one
 * flat fp32 buffer; level l is [table_size[l], feature_dim] row-major at table_offset[l] (floats). */
typedef struct {
  int32_t n_levels, feature_dim, smooth, pad_;
  float bbox_min[3], bbox_max[3];
  int32_t grid_size[32];
  int32_t table_size[32];
  int64_t table_offset[32];
  int32_t hashed[32]; /* 1 when grid_size^3 > requested table size (instant_ngp.py:178) */
} lnrf_hashgrid_desc;

/* enc_t[(level*F + f) * m_total + m] = sum over the 8 cell corners of weight * table[index][f]
 * (feature-major output, [L*F][M]). x: [M,3] points. */
int lnrf_hashgrid_fwd(const lnrf_hashgrid_desc* desc, const float* tables, const float* x, int64_t m,
                      float* enc_t, lnrf_stream_t stream);

/* g_tables += scatter of g_enc_t (same layout as enc_t) through the same indices/weights. */
int lnrf_hashgrid_bwd(const lnrf_hashgrid_desc* desc, const float* x, int64_t m, const float* g_enc_t,
                      float* g_tables, lnrf_stream_t stream);

/* Same scatter with caller-provided scratch (lnrf_hashgrid_bwd_scratch_bytes): hashed levels are reduced
 * without global float atomics (bin by 8K-entry table slice, then one workgroup per bucket accumulates in
 * LDS).  u == NULL: value weights (first-order gradient); u [M,3]: derivative weights (see bwd_dir).
 * level_absmax (optional, u == NULL only): n_levels floats, level_absmax[l] >= max |g_enc_t rows 2l, 2l+1|;
 * lnrf_ngp_mlp_bwd produces it while writing g_enc_t.  WITH it the contributions are binned as 26-bit fixed point of
 * that bound (8-byte tuples: resolution 2^-25 of the level's bound per contribution — a QUANTISED scatter, the fused
 * bf16 path's; values beyond a too-small bound saturate).  NULL: 12-byte fp32 tuples, the level maximum is found while
 * binning and the reduce pass sums in 64-bit fixed point at 2^-46 of it per contribution — the exact-fp32 path's
 * scatter (instant_ngp.py:211-224 under jax.grad).  Either way NaN / Inf contributions reach the table through float
 * atomics, as a floating-point scatter-add would propagate them.
 * scratch == NULL falls back to the LDS-sliced / atomic kernels. */
int64_t lnrf_hashgrid_bwd_scratch_bytes(const lnrf_hashgrid_desc* desc, int64_t m);
int lnrf_hashgrid_bwd_bucketed(const lnrf_hashgrid_desc* desc, const float* x, const float* u, int64_t m,
                               const float* g_enc_t, const float* level_absmax, float* g_tables, void* scratch,
                               int64_t scratch_bytes, lnrf_stream_t stream);

/* Input-derivative maps of the encoding, needed when a Ref-NeRF head sits on the hash grid
 * (InstantNGPRefNERFModel, instant_ngp.py:57-89: normals = -d out[:,0]/dx, ref_nerf.py:38-43):
 *   jvp:        enc_t-shaped (d enc / d x) u,            u [M,3]
 *   input_grad: g_x[M,3] = (d enc / d x)^T g_enc
 *   bwd_dir:    g_tables += d/d tables of < (d enc / d x) u , g_enc >   (second-order term) */
int lnrf_hashgrid_jvp(const lnrf_hashgrid_desc* desc, const float* tables, const float* x, const float* u,
                      int64_t m, float* enc_t, lnrf_stream_t stream);
int lnrf_hashgrid_input_grad(const lnrf_hashgrid_desc* desc, const float* tables, const float* x, int64_t m,
                             const float* g_enc_t, float* g_x, lnrf_stream_t stream);
int lnrf_hashgrid_bwd_dir(const lnrf_hashgrid_desc* desc, const float* x, const float* u, int64_t m,
                          const float* g_enc_t, float* g_tables, lnrf_stream_t stream);

/* ------------------------------------------------------------- Ref-NeRF ---- */

/* Transpose / tangent of d sinusoidal_emb / d x (model.py:65-77), needed by the analytic normals
 * -d out[:,0]/dx of RefNERFBase (ref_nerf.py:38-43) and by their second-order term:
 *   bwd: g_x[m,c]   = sum_f 2^f (cos(2^f x) g_emb[sin f] - sin(2^f x) g_emb[cos f])
 *   jvp: v_emb[...] = (d emb / d x) u,  u [M,dims]. */
int lnrf_sinusoidal_emb_bwd(const float* x, int64_t ldx, int64_t m, int32_t dims, int32_t freqs,
                            const float* g_emb, int64_t ldg, int64_t col_off, float* g_x,
                            lnrf_stream_t stream);
int lnrf_sinusoidal_emb_jvp(const float* x, int64_t ldx, int64_t m, int32_t dims, int32_t freqs,
                            const float* u, float* v_emb, int64_t ldv, int64_t col_off,
                            lnrf_stream_t stream);

/* integrated_directional_encoding (ref_nerf.py:121-143) on spherical_harmonic (:146-311):
 * out[M, sh_degree^2]; roughness [M] or NULL (plain spherical harmonics). */
int lnrf_integrated_directional_encoding(int32_t sh_degree, const float* coords, const float* roughness,
                                         int64_t m, float* out, lnrf_stream_t stream);

/* RefNERFBase.__call__ between spatial_block and directional_block (ref_nerf.py:41-63, 72-75).
 * spatial: [M, >=9] (row stride lds): density logit, diffuse(3), spectral, roughness, normal(3).
 * nraw [M,3] = -d sum(out[:,0]) / dx (unnormalised analytic normal), d [M,3] view directions.
 * Writes density[M], diffuse[M,3], spectral[M], tail[m, 0..sh^2] = IDE of the reflection direction,
 * tail[m, sh^2] = -d.n  (the columns appended to spatial_out for the directional block), and
 * aux[M,2] = (normal_mse, neg_normal) per sample. */
int lnrf_refnerf_head_fwd(const float* spatial, int64_t lds, const float* nraw, const float* d, int64_t m,
                          int32_t sh_degree, float* density, float* diffuse, float* spectral, float* tail,
                          int64_t ld_tail, float* aux, lnrf_stream_t stream);
/* VJP of the above: g_spatial[m, 0..8] += ..., g_nraw[M,3] = ... */
int lnrf_refnerf_head_bwd(const float* spatial, int64_t lds, const float* nraw, const float* d, int64_t m,
                          int32_t sh_degree, const float* g_density, const float* g_diffuse,
                          const float* g_spectral, const float* g_tail, int64_t ld_tail, const float* g_aux,
                          float* g_spatial, int64_t ldgs, float* g_nraw, lnrf_stream_t stream);

/* full_color = linear_rgb_to_srgb(_leaky_clip(sigmoid(dir_out) * spectral + diffuse)) * 2 - 1
 * (ref_nerf.py:64-71, 110-118, 320-326) and its VJP. */
int lnrf_refnerf_color_fwd(const float* dir_out, const float* spectral, const float* diffuse, int64_t m,
                           float* rgb, lnrf_stream_t stream);
int lnrf_refnerf_color_bwd(const float* dir_out, const float* spectral, const float* diffuse, int64_t m,
                           const float* g_rgb, float* g_dir_out, float* g_spectral, float* g_diffuse,
                           lnrf_stream_t stream);

/* ------------------------------------------------ fused NeRF MLP (bf16 MFMA) ---- */

/* NeRFModel hyper-parameters (model.py:35-40). The fused kernels support the reference
 * default {5,4,256,128,10,4}; anything else returns LNRF_ERR_UNSUPPORTED (use the dense path). */
typedef struct {
  int32_t input_layers, mid_layers, hidden_dim, color_layer_dim, x_freqs, d_freqs;
} lnrf_nerf_shape;

/* number of fp32 parameters in Flax creation order (Dense_i.kernel[in,out], Dense_i.bias). */
int64_t lnrf_nerf_param_count(const lnrf_nerf_shape* shape);
/* bytes of the opaque MFMA-fragment-ordered bf16 copy produced by lnrf_nerf_pack_weights. */
int64_t lnrf_nerf_packed_bytes(const lnrf_nerf_shape* shape);
/* bytes of the saved-activation buffer for m evaluations (forward -> backward). */
int64_t lnrf_nerf_save_bytes(const lnrf_nerf_shape* shape, int64_t m);
/* bytes of backward scratch for m evaluations: the pre-activation gradients in fragment order (written by
 * lnrf_nerf_mlp_bwd_chain, read by lnrf_nerf_mlp_bwd_weights) followed by the partial-sum slabs that
 * lnrf_nerf_mlp_bwd_weights writes and folds (deterministic reduction instead of fp32 atomics). */
int64_t lnrf_nerf_bwd_scratch_bytes(const lnrf_nerf_shape* shape, int64_t m);

/* Repack fp32 Flax-layout parameters into the bf16 fragment streams (forward + transposed). */
int lnrf_nerf_pack_weights(const lnrf_nerf_shape* shape, const float* params, void* packed,
                           lnrf_stream_t stream);

/* NeRFModel.__call__ (model.py:43-62) for M evaluations, positional encoding included.
 * Points come either from explicit x[M,3], d[M,3] (rays == NULL) or, without materialising
 * them (render.py:318-319), from rays + ts: evaluation m = n*t + i is x = o_n + d_n*ts[n,i].
 * density[M] >= 0, rgb[M,3] in (-1,1).  save (nullable): activations for the backward. */
int lnrf_nerf_mlp_fwd(const lnrf_nerf_shape* shape, const void* packed, const float* x,
                      const float* d, const float* rays, int64_t ray_stride, const float* ts,
                      int32_t t, int64_t m, float* density, float* rgb, void* save,
                      lnrf_stream_t stream);

/* The same forward for a backward by lnrf_nerf_mlp_bwd_ls / _ls2 (save must not be NULL): that backward takes the ReLU
 * masks of Dense_0..7 from the saved activations, so their mask slots in `save` are left unwritten (8 KiB per 32
 * evaluations less to store, ~1/4 fewer epilogue instructions).  A save written by this entry must NOT be handed to
 * lnrf_nerf_mlp_bwd / _bwd_chain. */
int lnrf_nerf_mlp_fwd_ls(const lnrf_nerf_shape* shape, const void* packed, const float* x,
                         const float* d, const float* rays, int64_t ray_stride, const float* ts,
                         int32_t t, int64_t m, float* density, float* rgb, void* save,
                         lnrf_stream_t stream);

/* Split-precision ("bf16x3") evaluation of NeRFModel.__call__ (model.py:43-62), the render / evaluation path:
 * the reference computes this in fp32 (model.py:72, render.py:140); here every fp32 operand is carried as
 * a bf16 pair hi + lo and every product is hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation
 * (16 significant bits per operand; rendered RGB within 1e-3 of the fp32 reference, tests/test_gpu_nerf_mlp.py).
 * lnrf_nerf_packed_split_bytes: size of the opaque [hi,lo] fragment stream made by lnrf_nerf_pack_weights_split.
 * lnrf_nerf_mlp_fwd_split: same arguments as lnrf_nerf_mlp_fwd without the save buffer (inference only). */
int64_t lnrf_nerf_packed_split_bytes(const lnrf_nerf_shape* shape);
int lnrf_nerf_pack_weights_split(const lnrf_nerf_shape* shape, const float* params, void* packed_split,
                                 lnrf_stream_t stream);
int lnrf_nerf_mlp_fwd_split(const lnrf_nerf_shape* shape, const void* packed_split, const float* x,
                            const float* d, const float* rays, int64_t ray_stride, const float* ts,
                            int32_t t, int64_t m, float* density, float* rgb, lnrf_stream_t stream);

/* Backward of the above wrt the parameters: grads[param_count] += d L / d params given
 * g_density[M], g_rgb[M,3] (= d L / d outputs), the forward outputs and the save buffer.
 * Equals lnrf_nerf_mlp_bwd_chain followed by lnrf_nerf_mlp_bwd_weights. */
int lnrf_nerf_mlp_bwd(const lnrf_nerf_shape* shape, const void* packed, const void* save,
                      const float* density, const float* rgb, const float* g_density,
                      const float* g_rgb, int64_t m, void* scratch, float* grads,
                      lnrf_stream_t stream);

/* Part 1: input-gradient chain (what jax.grad does through model.py:49-60 back to front);
 * writes the pre-activation gradients of every Dense layer into scratch (fragment order). */
int lnrf_nerf_mlp_bwd_chain(const lnrf_nerf_shape* shape, const void* packed, const void* save,
                            const float* density, const float* rgb, const float* g_density,
                            const float* g_rgb, int64_t m, void* scratch, lnrf_stream_t stream);

/* Part 2: grads += X_l^T dy_l for every Dense kernel and sum_m dy_l for every bias.  Reads the gradient dump at
 * the start of scratch and uses the slab region behind it (lnrf_nerf_bwd_scratch_bytes covers both). */
int lnrf_nerf_mlp_bwd_weights(const lnrf_nerf_shape* shape, const void* save, void* scratch,
                              int64_t m, float* grads, lnrf_stream_t stream);

/* The same backward (jax.grad through model.py:43-62, as lnrf_nerf_mlp_bwd) as a LAYER-STATIONARY pipeline
 * (csrc/nerf_bwd_ls.hip): a head launch (Dense_11, Dense_10, Dense_9 -> dz), then ONE persistent launch in which every
 * CU owns one of Dense_8 ... Dense_1 — W_l^T and the dW_l accumulators stay in its registers for the whole launch — and
 * the 32-evaluation tiles pass from CU to CU (dy handed over through write-through stores and flag words), then the
 * five small weight-gradient problems and a fixed-order fold of the per-pipeline partial sums (bit-reproducible).
 * grads += d L / d params.  scratch: lnrf_nerf_bwd_ls_scratch_bytes(shape, m) bytes; the 32-bit word at byte offset
 * lnrf_nerf_bwd_ls_status_offset(shape, m) is non-zero after the launch if a bounded hand-off wait gave up (gradients of
 * that call are then invalid).  lnrf_nerf_mlp_bwd_ls2 runs two models (train.py:141-142: coarse and fine) in the same
 * persistent launch, pipelines shared in proportion to their evaluations.
 * phases: 7 = the whole backward; a caller that wants to time the parts passes 1 (head launches), then 2 (the persistent
 * pipeline launch), then 4 (small weight-gradient problems + folds) with the same arguments, in that order. */
int64_t lnrf_nerf_bwd_ls_scratch_bytes(const lnrf_nerf_shape* shape, int64_t m);
int64_t lnrf_nerf_bwd_ls_status_offset(const lnrf_nerf_shape* shape, int64_t m);
int lnrf_nerf_mlp_bwd_ls(const lnrf_nerf_shape* shape, const void* packed, const void* save,
                         const float* density, const float* rgb, const float* g_density,
                         const float* g_rgb, int64_t m, void* scratch, float* grads, int32_t phases,
                         lnrf_stream_t stream);
int lnrf_nerf_mlp_bwd_ls2(const lnrf_nerf_shape* shape, const void* packed_a, const void* save_a,
                          const float* density_a, const float* rgb_a, const float* g_density_a,
                          const float* g_rgb_a, int64_t m_a, void* scratch_a, float* grads_a,
                          const void* packed_b, const void* save_b, const float* density_b,
                          const float* rgb_b, const float* g_density_b, const float* g_rgb_b, int64_t m_b,
                          void* scratch_b, float* grads_b, int32_t phases, lnrf_stream_t stream);

/* ---- fused spatial block of RefNERFModel (reference learn_nerf/ref_nerf.py:80-107; csrc/refnerf_fused.hip) ----
 * The spatial block of RefNERFModel is the NeRFModel trunk (Dense_0..8: 60 -> 256 x5, 316 -> 256, 256 x3, default
 * widths and x_freqs = 10 only; parameters in Flax creation order, so Dense_0..8 sit where they sit in NeRFModel's
 * vector).  These entries run it — and everything that differentiates through it — on the fused bf16-MFMA chain:
 *   lnrf_refnerf_trunk_pack     fp32 parameters -> opaque fragment streams (lnrf_refnerf_trunk_packed_bytes bytes)
 *   lnrf_refnerf_trunk_fwd      spatial_out[m, 0:256] (fp32, `ld` floats per row, rows 16-byte aligned) =
 *                               spatial_block(x) (ref_nerf.py:37); `save` (lnrf_nerf_save_bytes of the default
 *                               lnrf_nerf_shape) receives the bf16 activations and ReLU masks of the pass
 *   lnrf_refnerf_normal_pass    n_raw[m, 3] = -d spatial_out[:, 0] / dx (ref_nerf.py:38-43: the jax.grad inside
 *                               RefNERFBase.__call__); cdump (lnrf_nerf_bwd_scratch_bytes) keeps the chain states
 *   lnrf_refnerf_trunk_bwd      grads += d L / d Dense_0..8 given g_spatial = d L / d spatial_out [m, ld]
 *                               (first-order path, on the layer-stationary pipeline of lnrf_nerf_mlp_bwd_ls;
 *                               scratch: lnrf_refnerf_trunk_bwd_scratch_bytes(m))
 *   lnrf_refnerf_normal_bwd     grads += the second-order term: d L / d Dense_0..8 through n_raw, given
 *                               u = d L / d n_raw [m, 3] (jax.grad of a function that calls jax.grad, train.py:89-90
 *                               over ref_nerf.py:42; scratch: lnrf_nerf_save_bytes; the slab region behind the chain
 *                               states in cdump is used as workspace for the fixed-order weight-gradient fold)
 * Head, integrated directional encoding and the 273 -> 128 -> 3 directional block: lnrf_refnerf_head_*,
 * lnrf_refnerf_color_*, lnrf_dense_*. */
int64_t lnrf_refnerf_trunk_packed_bytes(void);
int lnrf_refnerf_trunk_pack(const float* params, void* packed, lnrf_stream_t stream);
int lnrf_refnerf_trunk_fwd(const void* packed, const float* x, int64_t m, void* save, float* spatial_out,
                           int64_t ld, lnrf_stream_t stream);
int lnrf_refnerf_normal_pass(const void* packed, const void* save, const float* x, int64_t m, void* cdump,
                             float* nraw, lnrf_stream_t stream);
int64_t lnrf_refnerf_trunk_bwd_scratch_bytes(int64_t m);
int lnrf_refnerf_trunk_bwd(const void* packed, const void* save, const float* g_spatial, int64_t ld, int64_t m,
                           void* scratch, float* grads, lnrf_stream_t stream);
int lnrf_refnerf_normal_bwd(const void* packed, const void* save, const void* cdump, const float* x,
                            const float* u, int64_t m, void* scratch, float* grads, lnrf_stream_t stream);
/* Directional block of RefNERFModel (ref_nerf.py:100-107: Dense_9 273 -> 128 relu, Dense_10 128 -> 3; sh_degree 4,
 * color_layer_dim 128), fused like the trunk (the same `packed` blob carries its streams).
 *   lnrf_refnerf_dir_fwd   dir_out[m, 3] (pre-sigmoid) from dir_in[m, 0:273] = [spatial_out, IDE, -d.n] (fp32, `ld` >=
 *                          276 floats per row, rows 16-byte aligned); dsave: lnrf_refnerf_dir_save_bytes(m)
 *   lnrf_refnerf_dir_bwd   g_dir_in[m, 0:273] = d L / d dir_in (overwritten) and grads += d L / d Dense_9, Dense_10 given
 *                          g_dir_out[m, 3]; scratch: lnrf_refnerf_dir_scratch_bytes(m) */
int64_t lnrf_refnerf_dir_save_bytes(int64_t m);
int64_t lnrf_refnerf_dir_scratch_bytes(int64_t m);
int lnrf_refnerf_dir_fwd(const void* packed, const float* dir_in, int64_t ld, int64_t m, void* dsave,
                         float* dir_out, lnrf_stream_t stream);
int lnrf_refnerf_dir_bwd(const void* packed, const void* dsave, const float* g_dir_out, int64_t m, void* scratch,
                         float* g_dir_in, int64_t ld, float* grads, lnrf_stream_t stream);

/* Split-precision render path of RefNERFModel ("bf16x3": every forward WITHOUT a backward — rendering, evaluation,
 * model.apply).  The reference evaluates ref_nerf.py:35-77 in fp32; here every fp32 operand of the trunk forward, the
 * normal pass and the directional block is a bf16 pair hi + lo and every product is lo*hi + hi*lo + hi*hi on the bf16 MFMA
 * with fp32 accumulation (~1e-5 of fp32), so that rendered RGB meets the 1e-3 gate on the fused path.
 *   lnrf_refnerf_render_pack         fp32 parameters -> the split streams (lnrf_refnerf_render_packed_bytes bytes)
 *   lnrf_refnerf_trunk_normal_split  spatial_out[m, 0:256] (fp32, `ld` floats per row) and n_raw[m, 3] = -d spatial_out[:, 0]
 *                                    / dx in one launch; scratch: lnrf_refnerf_trunk_normal_split_scratch_bytes(m)
 *   lnrf_refnerf_dir_fwd_split       dir_out[m, 3] from dir_in[m, 0:273] as lnrf_refnerf_dir_fwd, nothing saved */
int64_t lnrf_refnerf_render_packed_bytes(void);
int lnrf_refnerf_render_pack(const float* params, void* packed_split, lnrf_stream_t stream);
int64_t lnrf_refnerf_trunk_normal_split_scratch_bytes(int64_t m);
int lnrf_refnerf_trunk_normal_split(const void* packed_split, const float* x, int64_t m, float* spatial_out, int64_t ld,
                                    float* n_raw, void* scratch, lnrf_stream_t stream);
int lnrf_refnerf_dir_fwd_split(const void* packed_split, const float* dir_in, int64_t ld, int64_t m, float* dir_out,
                               lnrf_stream_t stream);

/* ------------------------------------------------------- data-parallel exchange ---- */

/* One process per GPU; rays shard across ranks and the ONLY exchange of a training step is one all-reduce (sum) of
 * the flat fp32 gradient over RCCL/xGMI, between the backward pass and Adam — the data-parallel form of
 * TrainLoop.step_fn (train.py:85-106; the reference itself is single-device).  The mean over ranks is
 * lnrf_adam_step's grad_scale = 1/world.  The communicator is an opaque handle (the only persistent object the
 * library creates).  Bootstrap: rank 0 calls lnrf_comm_get_unique_id and hands the LNRF_COMM_UNIQUE_ID_BYTES bytes to
 * the other ranks by any channel (file, TCP store); then EVERY rank calls lnrf_comm_init (collective; binds to the
 * calling thread's current HIP device).  Errors of RCCL are returned as positive ncclResult_t values. */
#define LNRF_COMM_UNIQUE_ID_BYTES 128
typedef struct lnrf_comm* lnrf_comm_t;
int lnrf_comm_get_unique_id(void* unique_id /* (host) LNRF_COMM_UNIQUE_ID_BYTES bytes out */);
int lnrf_comm_init(const void* unique_id /* (host) */, int32_t rank, int32_t world, lnrf_comm_t* comm_out);
/* in-place sum over all ranks of buf[n] (device), enqueued on `stream` */
int lnrf_comm_allreduce(lnrf_comm_t comm, float* buf, int64_t n, lnrf_stream_t stream);
int lnrf_comm_info(lnrf_comm_t comm, int32_t* rank, int32_t* world);
int lnrf_comm_destroy(lnrf_comm_t comm);

/* ------------------------------------------------------------- optimiser ---- */

/* ---- fused InstantNGPModel MLP (reference learn_nerf/instant_ngp.py:38-54: the Dense stack after the
 * MultiresHashTableEncoding: Dense(hidden) relu x density_layers, Dense(density_dim), density = exp(out[:, :1]),
 * concat [sinusoidal_emb(d), out], Dense(hidden) relu x color_layers, tanh(Dense(3))).  bf16 MFMA operands,
 * fp32 accumulate.  Fused configuration: hidden_dim 64, density_dim 16, density_layers 1, color_layers 2,
 * d_freqs 4 (the reference defaults) and enc_dim = L*F <= 32; anything else returns LNRF_ERR_UNSUPPORTED and the
 * caller uses lnrf_dense_* / lnrf_gemm_f32.  dense_offset = float offset of Dense_0/kernel in the flat parameter
 * vector (parameters are laid out Dense_i kernel [fan_in][fan_out] then bias, i = 0..4). */
typedef struct lnrf_ngp_mlp_desc {
  int32_t enc_dim;        /* L * table_feature_dim */
  int32_t hidden_dim;
  int32_t density_dim;
  int32_t density_layers;
  int32_t color_layers;
  int32_t d_freqs;
  int64_t dense_offset;
} lnrf_ngp_mlp_desc;

int64_t lnrf_ngp_mlp_packed_bytes(const lnrf_ngp_mlp_desc* desc);
int64_t lnrf_ngp_mlp_scratch_bytes(const lnrf_ngp_mlp_desc* desc, int64_t m);
/* params: the flat fp32 parameter vector (tables included); packed: lnrf_ngp_mlp_packed_bytes bytes. */
int lnrf_ngp_mlp_pack(const lnrf_ngp_mlp_desc* desc, const float* params, void* packed, lnrf_stream_t stream);
/* enc_t: [enc_dim][m] feature-major output of lnrf_hashgrid_fwd; d: [m][3]; density [m], rgb [m][3]. */
int lnrf_ngp_mlp_fwd(const lnrf_ngp_mlp_desc* desc, const void* packed, const float* enc_t, const float* d,
                     int64_t m, float* density, float* rgb, lnrf_stream_t stream);
/* The same forward in split precision ("bf16x3", the render / evaluation path): every fp32 operand is carried as a
 * bf16 pair hi + lo and every product is lo*hi + hi*lo + hi*hi on the bf16 MFMA with fp32 accumulation, which
 * reproduces the reference's fp32 arithmetic (instant_ngp.py:38-54) to ~1e-5, so that rendered RGB meets the 1e-3
 * gate.  packed_split: lnrf_ngp_mlp_packed_split_bytes bytes, written by lnrf_ngp_mlp_pack_split. */
int64_t lnrf_ngp_mlp_packed_split_bytes(const lnrf_ngp_mlp_desc* desc);
int lnrf_ngp_mlp_pack_split(const lnrf_ngp_mlp_desc* desc, const float* params, void* packed_split,
                            lnrf_stream_t stream);
int lnrf_ngp_mlp_fwd_split(const lnrf_ngp_mlp_desc* desc, const void* packed_split, const float* enc_t,
                           const float* d, int64_t m, float* density, float* rgb, lnrf_stream_t stream);
/* VJP of lnrf_ngp_mlp_fwd (recomputes the forward): g_enc_t [enc_dim][m] = d loss / d enc (written),
 * grads (the flat gradient vector, same layout as params) += Dense kernel / bias gradients.
 * level_absmax (optional): enc_dim / 2 floats ZEROED by the caller; on return level_absmax[l] = max |g_enc_t| over
 * the two feature rows of level l (max-combined with what was there), for lnrf_hashgrid_bwd_bucketed.
 * scratch: lnrf_ngp_mlp_scratch_bytes(desc, m) bytes. */
int lnrf_ngp_mlp_bwd(const lnrf_ngp_mlp_desc* desc, const void* packed, const float* enc_t, const float* d,
                     const float* g_density, const float* g_rgb, int64_t m, void* scratch, float* g_enc_t,
                     float* level_absmax, float* grads, lnrf_stream_t stream);

/* optax.adam (train.py:59), fused over a flat buffer:
 *   g' = g*grad_scale; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2;
 *   p -= lr * (m/(1-b1^step)) / (sqrt(v/(1-b2^step)) + eps).   step counts from 1. */
int lnrf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                   float b2, float eps, int32_t step, float grad_scale, lnrf_stream_t stream);

/* lnrf_adam_step that also accumulates the two tree_norm numerators of the step's log (train.py:92-104) while it
 * streams the buffers: sq_norms[0] += sum g^2 (g as passed in, i.e. the all-reduced SUM; multiply the root by
 * grad_scale), sq_norms[1] += sum p^2 of the parameters BEFORE the update.  sq_norms: 2 floats, zeroed by the caller. */
int lnrf_adam_step_norms(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                         float b2, float eps, int32_t step, float grad_scale, float* sq_norms,
                         lnrf_stream_t stream);

/* The scalar log of one step (train.py:141-144: mean squared errors; train.py:99-104: norms) from its accumulators:
 * sums = [sum sq err coarse, sum sq err fine, sum g^2, sum p^2] -> out = [sums[0] * inv_count, sums[1] * inv_count,
 * sqrt(sums[2]) * grad_scale, sqrt(sums[3])]; clear != 0 zeroes sums afterwards (ready for the next step). */
int lnrf_step_log(float* sums, float inv_count, float grad_scale, int32_t clear, float* out,
                  lnrf_stream_t stream);

/* *out += sum x^2 (tree_norm numerator, train.py:92-97). out must be zeroed by the caller. */
int lnrf_sq_norm(const float* x, int64_t n, float* out, lnrf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LNRF_H */
