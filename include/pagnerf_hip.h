/*
 * pagnerf_hip.h - C ABI of libpagnerf_hip.so: the MI355X (gfx950) kernels behind the
 * kaolin-wisp grid / nef / tracer plugin API that PAg-NeRF's trainer drives.
 *
 * The reference is 100 % Python and has no FFI of its own; each entry point below replaces
 * a call the reference makes into a third-party CUDA package (or an in-tree torch op
 * sequence) at the cited file:line of the upstream repository.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is allocated by the caller (device memory
 *     unless the name ends in _host); the library owns nothing and keeps no global state
 *     except the thread-local last-error string.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and is safe to capture into a hipGraph (no allocation, no synchronisation).
 *   - return value: 0 on success, negative PAG_ERR_* otherwise (pag_last_error_string()).
 *   - "dtype" arguments take PAG_F32 / PAG_F16 / PAG_BF16.
 *   - strides are in ELEMENTS; a [M, C] row-major tensor has stride_m = C, stride_c = 1,
 *     a feature-major one stride_m = 1, stride_c = M.
 */
#ifndef PAGNERF_HIP_H
#define PAGNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAG_ABI_VERSION 14

enum { PAG_F32 = 0, PAG_F16 = 1, PAG_BF16 = 2 };
enum { PAG_OK = 0, PAG_ERR_ARG = -1, PAG_ERR_LAUNCH = -2, PAG_ERR_UNSUPPORTED = -3 };
enum { PAG_ACT_NONE = 0, PAG_ACT_SIGMOID = 1, PAG_ACT_SOFTMAX = 2 };
enum { PAG_MLP_MFMA_BF16 = 0, PAG_MLP_FP32 = 1 };
enum { PAG_BG_BLACK = 0, PAG_BG_WHITE = 1 };
/* feature-tensor layouts of the encoders / decoder inputs:
 *   PAG_LAYOUT_STRIDED  [M, L*F] addressed through (stride_m, stride_c); column = level*F + f
 *   PAG_LAYOUT_XCD8     bf16 [8][M][8]: element e = j*F + f of group g holds level 8j + g (j even) or 8j + 7 - g (j odd) - every group,
 *                       i.e. every XCD of the encoders' launches, gets a mix of coarse and fine levels - zero padded (ABI 6; before: 8j + g)
 *                       (requires ceil(L/8)*F <= 8).  Strides are ignored. */
enum { PAG_LAYOUT_STRIDED = 0, PAG_LAYOUT_XCD8 = 1 };
/* `flags` of the encode entry points:
 *   PAG_ENC_HALF_COORDS  xyz is rounded to fp16 (round-to-nearest-even) and widened back to f32 before anything else uses it -
 *                        what `@torch.cuda.amp.custom_fwd(cast_inputs=torch.half)` + `.type(torch.float)` do to the coordinates under
 *                        the reference trainer's autocast (grids/permuto_grid.py:65,71; grids/hash_grid_tinycudann.py:36-41).
 *                        Forward, table gradient and position gradient must be called with the same flags. */
enum { PAG_ENC_HALF_COORDS = 1 };
#define PAG_MAX_LEVELS 32
#define PAG_MAX_FEATS 64

int pag_abi_version(void);
const char *pag_last_error_string(void);

/* ------------------------------------------------------------------------------------------
 * Grid feature interpolation ("encode")
 * ------------------------------------------------------------------------------------------ */

/* Multiresolution hash grid.  Replaces HashEmbedder.forward (grids/hash_grid_torch.py:95-108:
 * get_voxel_vertices :26-46, hash :13-24, trilinear_interp :69-93) as reached through
 * HashGridTorch.interpolate (:130-140); also the tinycudann-backed variant
 * (grids/hash_grid_tinycudann.py:36-47) with its own resolution list.
 *   xyz          f32 [M,3]
 *   tables       [L, 2^log2_T, F] (F = 2 or 4), PAG_F32 or PAG_F16
 *   resolutions_host  f32 [L]  (grids/hash_grid_torch.py:99)
 *   feat_scale_host   f32 [L*F] or NULL: per-feature multiplier (nef.lod_weights,
 *                     pc_nerf/panoptic_delta_nef.py:171)
 *   out          [M, L*F] via strides, PAG_F32 or PAG_BF16; column = level*F + f
 * fp32 tables + fp32 out reproduce the reference's fp32 op order (no FMA contraction). */
int pag_hash_encode_fwd(const float *xyz, int64_t M, const void *tables, int table_dtype,
                        int n_levels, int n_feat, int log2_T, const float *resolutions_host,
                        const float *feat_scale_host, void *out, int out_dtype,
                        int64_t out_stride_m, int64_t out_stride_c, int layout, int flags, void *stream);

/* d loss / d tables (what autograd through grids/hash_grid_torch.py:95-108 yields).
 *   grad_out  [M, L*F] via strides (PAG_F32 or PAG_BF16);  grad_tables f32 [L,T,F], ACCUMULATED
 *   into (caller zeroes).  workspace: see pag_encode_bwd_workspace_bytes(). */
int pag_hash_encode_bwd(const float *xyz, int64_t M, const void *grad_out, int grad_dtype,
                        int64_t g_stride_m, int64_t g_stride_c, int layout, int n_levels, int n_feat,
                        int log2_T, const float *resolutions_host, const float *feat_scale_host,
                        float *grad_tables, void *workspace, int64_t workspace_bytes, int flags, void *stream);

/* Permutohedral-lattice hash encoding.  Replaces permutohedral_encoding.PermutoEncoding's
 * forward as called at grids/permuto_grid.py:57-62,71.
 *   scale_factor_host f32 [L,3] = 1/(sqrt((i+1)(i+2)) * scale[l]), scale = np.geomspace(...)
 *                     (grids/permuto_grid.py:53)
 *   shift_host        f32 [L,3] per-level random shift
 *   capacity          table rows per level (need not be a power of two) */
int pag_permuto_encode_fwd(const float *xyz, int64_t M, const void *tables, int table_dtype,
                           int n_levels, int n_feat, uint32_t capacity,
                           const float *scale_factor_host, const float *shift_host,
                           const float *feat_scale_host, void *out, int out_dtype,
                           int64_t out_stride_m, int64_t out_stride_c, int layout, int flags, void *stream);

/* Same encoders writing  out = bf16(addend + bf16(features))  in the XCD8 layout (addend, out: bf16 [8][M][8]).
 * pc_nerf/panoptic_delta_nef.py:226 forms the panoptic features as `feats.detach() + delta`; with the main grid's
 * features as addend the delta grid's encoder emits that sum directly (bit-identical to the separate bf16 add). */
int pag_hash_encode_fwd_add(const float *xyz, int64_t M, const void *tables, int table_dtype,
                            int n_levels, int n_feat, int log2_T, const float *resolutions_host,
                            const float *feat_scale_host, const void *addend, void *out, int flags, void *stream);
int pag_permuto_encode_fwd_add(const float *xyz, int64_t M, const void *tables, int table_dtype,
                               int n_levels, int n_feat, uint32_t capacity,
                               const float *scale_factor_host, const float *shift_host,
                               const float *feat_scale_host, const void *addend, void *out, int flags, void *stream);

int pag_permuto_encode_bwd(const float *xyz, int64_t M, const void *grad_out, int grad_dtype,
                           int64_t g_stride_m, int64_t g_stride_c, int layout, int n_levels, int n_feat,
                           uint32_t capacity, const float *scale_factor_host,
                           const float *shift_host, const float *feat_scale_host,
                           float *grad_tables, void *workspace, int64_t workspace_bytes, int flags, void *stream);

/* As pag_hash_encode_bwd / pag_permuto_encode_bwd, but grad_tables is OVERWRITTEN: every row of every level is written
 * (zeros where no gradient arrived), so the caller need not clear the table first and the reduce pass does not read it.
 * Binned algorithm only (workspace required).  M == 0 leaves grad_tables untouched. */
int pag_hash_encode_bwd_set(const float *xyz, int64_t M, const void *grad_out, int grad_dtype,
                            int64_t g_stride_m, int64_t g_stride_c, int layout, int n_levels, int n_feat,
                            int log2_T, const float *resolutions_host, const float *feat_scale_host,
                            float *grad_tables, void *workspace, int64_t workspace_bytes, int flags, void *stream);
int pag_permuto_encode_bwd_set(const float *xyz, int64_t M, const void *grad_out, int grad_dtype,
                               int64_t g_stride_m, int64_t g_stride_c, int layout, int n_levels, int n_feat,
                               uint32_t capacity, const float *scale_factor_host, const float *shift_host,
                               const float *feat_scale_host, float *grad_tables, void *workspace,
                               int64_t workspace_bytes, int flags, void *stream);

/* d loss / d xyz of the two encoders (camera pose optimisation, pc_nerf/ba_pipeline.py:85-92: the
 * samples o + t*d depend on the learnable extrinsics).  The reference gets this from autograd through
 * grids/hash_grid_torch.py:69-108 (hash) and from permutohedral_encoding's position gradient
 * (permuto); within a cell / simplex the features are (tri)linear in xyz.
 *   tables     as in *_encode_fwd (F32 or F16);  grad_out / strides / layout as in *_encode_bwd
 *   d_xyz      f32 [M,3] (out, overwritten)
 *   workspace  >= 8*M*3 floats: per-XCD-group partial sums, added deterministically */
int pag_hash_encode_bwd_xyz(const float *xyz, int64_t M, const void *tables, int table_dtype,
                            const void *grad_out, int grad_dtype, int64_t g_stride_m,
                            int64_t g_stride_c, int layout, int n_levels, int n_feat, int log2_T,
                            const float *resolutions_host, const float *feat_scale_host,
                            float *d_xyz, void *workspace, int64_t workspace_bytes, int flags, void *stream);
int pag_permuto_encode_bwd_xyz(const float *xyz, int64_t M, const void *tables, int table_dtype,
                               const void *grad_out, int grad_dtype, int64_t g_stride_m,
                               int64_t g_stride_c, int layout, int n_levels, int n_feat,
                               uint32_t capacity, const float *scale_factor_host,
                               const float *shift_host, const float *feat_scale_host, float *d_xyz,
                               void *workspace, int64_t workspace_bytes, int flags, void *stream);

/* Scratch size for the atomic-free ("binned") backward of either encoder: n_vertices = 8 (hash) or
 * 4 (permuto), rows_per_level = 2^log2_T or capacity.  The caller allocates it (device memory) and
 * passes it as `workspace`; with workspace == NULL the backward falls back to per-vertex fp32
 * global atomics (slow on MI355X, kept as the reference path for tests). */
int64_t pag_encode_bwd_workspace_bytes(int64_t M, int n_levels, int n_feat, int n_vertices,
                                       int64_t rows_per_level);

/* ------------------------------------------------------------------------------------------
 * Tiny-MLP decoders
 * ------------------------------------------------------------------------------------------ */

/* One wisp BasicDecoder (n_layers Linear layers, ReLU between, none after the last) fused in
 * one launch, with an optional output activation.  Replaces decoder_density / decoder_color /
 * decoder_semantics / decoder_inst (pc_nerf/panoptic_nef.py:114-164) as applied at
 * pc_nerf/panoptic_delta_nef.py:184,203,240-243,250-255, including the input concat of
 * :198-199 (x2 rows are gathered per sample through x2_index, so the view embedding is never
 * repeated in memory).
 *   x1        [M, k1] row-major (x1_dtype F32 or BF16), k1 % 8 == 0
 *   x2        f32 [R, k2p] row-major or NULL; k2p % 8 == 0 (caller zero-pads), x2_index i32 [M]
 *   W[i]      f32 [out_i, in_i] row-major (nn.Linear.weight), b[i] f32 [out_i];
 *             in_0 = in_dim <= k1 + k2p (columns beyond in_dim are treated as absent)
 *   n_layers  2 or 3; hidden width = 64; out_dim <= 224
 *   out       [M, out_dim] row-major (out_dtype F32 or BF16), after out_act
 *   hidden_save[i]  bf16 (MFMA mode) or f32 (FP32 mode) [M, 64] row-major post-ReLU
 *             activations of hidden layer i, or NULL when no backward will follow
 *   mode      PAG_MLP_MFMA_BF16: bf16 operands, fp32 accumulate on the matrix cores;
 *             PAG_MLP_FP32: fp32 FMA chain in k order (parity path) */
struct pag_head_composite_args;
typedef struct pag_mlp_fwd_args {
    const void *x1; int x1_dtype; int k1;
    int x1_layout; int x1_levels; int x1_feats;   /* PAG_LAYOUT_XCD8: x1 is the encoders' bf16 [8][M][8]
                                                     output for (levels, feats); k1 = 64, in_dim = levels*feats */
    const float *x2; int k2p; const int32_t *x2_index;
    int in_dim; int n_layers; int out_dim;
    const float *W[3]; const float *b[3];
    int out_act;
    void *out; int out_dtype;
    void *hidden_save[2];
    int mode;
    float *softmax_stats;   /* optional (MFMA mode, out_act SOFTMAX, out_dim > 64): f32 [M,2] = (max logit * log2(e),
                               1 / sum exp) per sample, from which the backward and pag_head_composite_fwd rebuild the
                               probabilities; `out` may then be NULL (nothing but the statistics is written) */
    float *x1_col0_relu;    /* optional (MFMA mode, strided bf16 x1, out_dim <= 64): f32 [M] = relu(x1[m][0]) written by the
                               same launch - the density pc_nerf/panoptic_delta_nef.py:188 reads off column 0 of the density
                               decoder's output, which is this (colour) decoder's x1 */
    /* Optional, statistics-only wide softmax head only (out NULL): a second decoder on the SAME XCD8 input - two layers, softmax,
     * out_dim <= 8, bf16 out, no hidden_save (the semantic head next to the instance head) - evaluated in this call's launch while the
     * tile is in registers.  The caller does not call pag_mlp_fwd for it.  pag_mlp_fwd_pair_supported() tells whether the pair qualifies. */
    const struct pag_mlp_fwd_args *pair;
    /* Optional (ABI 12), statistics-only wide softmax head only: the per-ray weighted sum of pag_head_composite_fwd formed in the SAME launch, in one pass
     * over the logits (each logit and each exponential once instead of twice) - out[ray] = alpha[ray] * sum_i weights[i] * softmax(...)[i] over the ray's
     * pack.  softmax_stats and hidden_save[1] are written as without it (for the backward); samples past pack_start[P] (fillers of a padded batch) get
     * statistics that rebuild to probability 0 and a zero hidden row.  The caller does not call pag_head_composite_fwd.
     * pag_mlp_fwd_composite_supported() tells whether the arguments qualify (pair included). */
    const struct pag_head_composite_args *composite;
    /* Optional (ABI 12), colour-like decoder only (strided bf16 x1 [M,16] + per-ray x2, three layers, sigmoid, x1_col0_relu, no hidden_save): the
     * density-like decoder whose `out` IS this decoder's x1 (XCD8 bf16 input, two layers, no activation, out_dim 16, bf16 out, no hidden_save) is
     * evaluated in the same launch - its output is written as without this field and handed on in registers (pc_nerf/panoptic_delta_nef.py:184 ->
     * :198-203 as one launch instead of two; bit-identical outputs).  The caller does not call pag_mlp_fwd for the producer.
     * pag_mlp_fwd_producer_supported() tells whether the two argument blocks qualify. */
    const struct pag_mlp_fwd_args *x1_producer;
} pag_mlp_fwd_args;
typedef struct pag_head_composite_args {
    const int64_t *pack_start; const int32_t *ray_of_pack; int64_t P;      /* as pag_head_composite_fwd */
    const float *weights;       /* f32 [M] compositing weights w_i */
    const float *alpha;         /* f32 [N] */
    float *out;                 /* f32 [N, out_dim]: rows of rays that have a pack are overwritten */
    int64_t n_samples;          /* pack_start[P] as the host knows it (0 = unknown): a speed hint only */
} pag_head_composite_args;
int pag_mlp_fwd(const pag_mlp_fwd_args *args, int64_t M, void *stream);
int pag_mlp_fwd_pair_supported(const pag_mlp_fwd_args *args, const pag_mlp_fwd_args *pair);
int pag_mlp_fwd_composite_supported(const pag_mlp_fwd_args *args, int64_t M);
int pag_mlp_fwd_producer_supported(const pag_mlp_fwd_args *args, const pag_mlp_fwd_args *producer, int64_t M);

/* Data gradients of pag_mlp_fwd.  grad_out is d loss / d (activated output); `out` is the
 * activated output saved from the forward (needed for sigmoid / softmax).  Writes
 *   dz[i]   [M, out_i (padded to 64 for hidden layers)] pre-activation gradients of layer i
 *           (bf16 in MFMA mode, f32 in FP32 mode); dz of the last layer is [M, out_dim]
 *   dx1     [M, k1] (dx1_dtype) or NULL
 * Weight gradients are dz[i]^T @ input_i - a plain GEMM left to the caller's BLAS. */
typedef struct pag_mlp_bwd_args {
    const void *grad_out; const void *out; int out_dtype; int out_act;   /* grad_out has out's dtype */
    int k1; int in_dim; int n_layers; int out_dim;
    int x1_layout; int x1_levels; int x1_feats;   /* as in pag_mlp_fwd_args: dx1 is then written as bf16 [8][M][8] */
    const float *W[3];
    const void *hidden_save[2];
    void *dz[3];          /* with wgrad_workspace (fused weight gradients) no dz tensor is written - except dz[0] (bf16 [M,64], optional) by the
                           * colour-like kernel: the caller sums it per ray for the gradient of the per-ray input x2 (pose optimisation) */
    void *dx1; int dx1_dtype;
    int mode;
    /* Optional rank-1 upstream gradient (MFMA mode): grad_out[m][c] = g_scale[m] * g_ray[g_index[m]][c] with
     * g_ray f32 [N, out_dim].  This is exactly d(feats) of tracers/panoptic_packed_rf_tracer.py:197-205
     * (alpha * w_m * d out[ray]) - passing it in this form means the [M, out_dim] gradient of the composited
     * semantic / instance probabilities is never materialised.  grad_out may then be NULL. */
    const float *g_ray; const float *g_scale; const int32_t *g_index;
    /* Optional (MFMA mode, out_act SOFTMAX, out_dim > 64, bf16 out): with the forward's softmax_stats and the last layer's
     * bias the wide-head backward recomputes the probabilities from hidden_save (4 MFMAs per 32-channel block on idle
     * matrix cores) instead of streaming the [M, out_dim] output through twice; `out` may then be NULL. */
    const float *softmax_stats; const float *b_last;
    int dx1_accumulate;   /* XCD8 dx1 only: add this decoder's input gradient to what dx1 already holds (two heads on the
                             same panoptic features - saves the separate gradient-sum pass) */
    const float *dx1_col0_add;   /* optional f32 [M] (strided dx1, MFMA mode, out_dim <= 64): added to column 0 of dx1 - the
                                    gradient of the density that pc_nerf/panoptic_delta_nef.py:188 reads off column 0 of the
                                    density decoder's output, which is also this (colour) decoder's x1 */
    const float *dx1_col0_gate;  /* optional f32 [M] with dx1_col0_add: the addend is applied only where gate[m] > 0 (the relu of
                                    that column: pass the forward's x1_col0_relu) */
    const float *g_ray_scale;    /* optional f32 [N] with the rank-1 gradient: grad_out[m][c] = g_scale[m] * g_ray_scale[g_index[m]] *
                                    g_ray[g_index[m]][c] (g_scale = the compositing weights w_m, g_ray_scale = the rays' alpha) */
    /* Fused weight gradients (optional; pag_mlp_bwd_fused_supported() == 1).  With wgrad_workspace != NULL the launch also forms
     * dW[i] = dz_i^T . input_i (f32 [out_i, in_i], nn.Linear layout) and db[i] = column sums of dz_i for every layer, from the
     * tiles it holds anyway: no dz tensor is written (dz[] may be NULL) and no second pass re-reads [M,64] activations
     * (536 MB per 64 x 64 layer at M = 2.1 M - the separate pag_mlp_wgrad_batch ran at the HBM rate on exactly those bytes).
     * Needs the forward's layer-0 input: x1 (bf16; layout / levels / feats as above) and, if the forward had it, x2 / x2_index.
     * wgrad_workspace: device scratch of pag_mlp_bwd_fused_workspace_bytes(args, M) bytes (per-wave partial sums, reduced in a
     * fixed order by a second tiny launch: deterministic). */
    const void *x1; int x1_dtype; const float *x2; int k2p; const int32_t *x2_index;
    float *wgrad_workspace; int64_t wgrad_workspace_bytes;
    float *dW[3]; float *db[3];
    /* Fused mode RECOMPUTES the hidden activations from x1 (same MFMA sequence as pag_mlp_fwd: bit-identical) instead of reading
     * hidden_save - the forward then need not write them (pag_mlp_fwd_args.hidden_save = NULL): 268 MB less each way per hidden
     * layer at M = 2.1 M.  Needs the hidden layers' biases b[0 .. n_layers-2]; hidden_save[] may be NULL except, for the wide
     * softmax head, the LAST hidden layer's (the probabilities are rebuilt from it). */
    const float *b[3];
    /* Optional, wide softmax head with wgrad_workspace only: a second decoder on the SAME XCD8 input - a two-layer softmax head with
     * out_dim <= 8 (the semantic head next to the instance head), filled in as for its own pag_mlp_bwd call (fused workspace, dW, db,
     * b[0], rank-1 gradient, dx1 = this decoder's dx1 with dx1_accumulate = 1).  Its backward then runs inside this call, in the launch of
     * the layers below the wide output layer: one read of the input and one write of the summed input gradient for both decoders.
     * The caller does not call pag_mlp_bwd for it. */
    const struct pag_mlp_bwd_args *pair;
    /* Optional (ABI 11), colour-like decoder with wgrad_workspace and a per-ray x2 whose x2_index is NON-DECREASING (samples packed ray by ray): instead
     * of the [M,64] dz[0] tensor the launch writes one f32 [64] row per (32-sample tile, ray) - the column sums of dz_0 over that ray's samples of the
     * tile, formed on the matrix cores - at row tile + ray of dz0_slots (pag_mlp_dz0_slots_bytes(M, R) bytes, R = rows of x2); pag_mlp_dz0_slots_sum adds
     * a ray's rows: the per-ray sum of dz_0 (x W_0[:, k1:] = the gradient of x2) for 8 B per sample instead of 128 written and 128 read again. */
    float *dz0_slots;
} pag_mlp_bwd_args;
int pag_mlp_bwd(const pag_mlp_bwd_args *args, int64_t M, void *stream);
int64_t pag_mlp_dz0_slots_bytes(int64_t M, int64_t R);
/* out f32 [R,64] = per-ray sums of pag_mlp_bwd_args.dz0_slots; pack_start i64 [R+1] = the rays' sample ranges (one pack per ray).  (ABI 11) */
int pag_mlp_dz0_slots_sum(const int64_t *pack_start, int64_t R, const float *slots, float *out, void *stream);
/* 1 when pag_mlp_bwd has a fused weight-gradient kernel for these (fully filled in, wgrad_workspace aside) arguments.  Three
 * decoder shapes are covered - the narrow decoders of pc_nerf/panoptic_nef.py:114-164 on the bf16 path:
 *   density-like   XCD8 bf16 x1, dense bf16 grad_out, no output activation, out_dim % 4 == 0, XCD8 bf16 dx1
 *   colour-like    bf16 x1 [M,16] + f32 x2 [R,32] through x2_index, dense f32 grad_out, sigmoid, out_dim <= 4, bf16 dx1 [M,16] with
 *                  dx1_col0_add + dx1_col0_gate
 *   semantic-like  XCD8 bf16 x1, rank-1 gradient (g_ray, g_scale, g_index, g_ray_scale), softmax with the saved bf16 `out`,
 *                  out_dim <= 8, XCD8 bf16 dx1 (dx1_accumulate allowed)
 *   wide softmax   XCD8 bf16 x1, rank-1 gradient, softmax_stats + b_last (the instance head: 3 layers, 192 < out_dim <= 224), XCD8
 *                  bf16 dx1: two launches - the output layer with its weight gradient (the [M, out_dim] softmax gradient is never
 *                  written), then the two layers below it on the hidden gradient
 * the first three with 2 or 3 layers, out_dim <= 32, and input column 63 free (XCD8: staged position 63 is padding; it carries the
 * constant 1 whose weight gradient is the layer-0 bias gradient).  Anything else: dz[] + pag_mlp_wgrad_batch. */
int pag_mlp_bwd_fused_supported(const pag_mlp_bwd_args *args);
/* The wide softmax kernels address the [M,64] bf16 tensors with 32-bit byte offsets: pag_mlp_bwd refuses M above this on that path
 * (PAG_ERR_ARG; callers split the batch or leave wgrad_workspace NULL). */
#define PAG_MLP_FUSED_WIDE_MAX_M ((int64_t)1 << 24)
int64_t pag_mlp_bwd_fused_workspace_bytes(const pag_mlp_bwd_args *args, int64_t M);
/* 1 when `pair` (fully filled in) can ride in args' call as pag_mlp_bwd_args.pair: args is a wide softmax head, pair a two-layer narrow
 * softmax head on the same XCD8 input that accumulates into args->dx1. */
int pag_mlp_bwd_pair_supported(const pag_mlp_bwd_args *args, const pag_mlp_bwd_args *pair);

/* One affine map of the XCD8 features: the activation-free `decoder_delta_density` of pc_nerf/panoptic_dd_nef.py:49-56, :238
 * (its layers composed to one [n_out, in_dim] matrix by the caller; n_out <= 8).
 *   fwd     out f32 [M, n_out] = x . W^T + b        x: bf16 [8][M][8] (PAG_LAYOUT_XCD8 of x_levels x x_feats features), W f32 [n_out, in_dim]
 *   bwd_dx  dx bf16 [8][M][8] = grad_out [M, n_out] . W   (padding positions 0)
 * Weight gradients: pag_mlp_wgrad_batch with grad_out (bf16) as `dz` and x as `a1`. */
int pag_affine_xcd8_fwd(const void *x, int64_t M, int x_levels, int x_feats, const float *W, const float *b, int n_out, int in_dim,
                        float *out, void *stream);
int pag_affine_xcd8_bwd_dx(const float *grad_out, int64_t M, int x_levels, int x_feats, const float *W, int n_out, int in_dim, void *dx,
                           void *stream);

/* Wide softmax head fused with the per-ray weighted sum of tracers/panoptic_packed_rf_tracer.py:197-205:
 *   out[ray][c] = alpha[ray] * sum_{i in pack} weights[i] * softmax(W_last . hidden[i] + b_last)[c]
 * from the forward's saved last hidden layer (bf16 [M,64]) and softmax_stats (pag_mlp_fwd with out = NULL): the
 * [M, out_dim] probabilities are rebuilt tile by tile and never stored.  64 < out_dim <= 224; out f32 [N,out_dim]
 * (rows of rays that have a pack are overwritten).  n_samples = pack_start[P] as the host knows it (0 = unknown): short packs
 * (fewer than ~5 tiles of 32 samples on average: the voxel march after the first prune) are processed one per wave instead of
 * one per workgroup - same sums in the same order per pack, a speed hint only. */
int pag_head_composite_fwd(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P,
                           const void *hidden, const float *W_last, const float *b_last, int out_dim,
                           const float *softmax_stats, const float *weights, const float *alpha,
                           float *out, int64_t n_samples, void *stream);

/* Weight and bias gradients of one Linear layer of pag_mlp_fwd (MFMA mode):
 *   dW[o][i] = sum_m dz[m][o] * a[m][i],  db[o] = sum_m dz[m][o]
 * with a = the layer's input: [M,k1] (F32 or BF16) optionally followed by gathered per-ray columns
 * a2[a2_index[m]] (layer 0 of the colour decoder), n_in <= 64 columns used.
 *   dz      bf16 [M, dz_cols] (dz_cols >= n_out; the hidden layers' dz are [M,64])
 *   a1_layout PAG_LAYOUT_XCD8: a1 is the encoders' bf16 [8][M][8] tensor (k1 = n_in = 64); slab column p
 *           then holds the gradient of feature column level*F+f with p = 8*(level%8) + (level/8)*F + f.
 *   slabs   f32 [n_blocks][ceil(n_out/32)*32][96] per-workgroup partial sums written (not
 *           accumulated) by the kernel: columns 0..n_in-1 = dW rows, column 64 = db.  The caller sums
 *           over n_blocks (n_blocks = pag_mlp_wgrad_blocks(M)) - deterministic, no atomics. */
int pag_mlp_wgrad_blocks(int64_t M);
int pag_mlp_wgrad(const void *dz, int dz_cols, int n_out, const void *a1, int a1_dtype,
                  int a1_layout, int k1, const float *a2, int k2p, const int32_t *a2_index, int n_in,
                  float *slabs, int n_blocks, int64_t M, void *stream);

/* Deterministic sum of those slabs into the final gradients: dW f32 [n_out, n_in] (for XCD8 inputs the staged
 * positions are mapped back to feature columns level*F + f) and db f32 [n_out]. */
int pag_mlp_wgrad_finish(const float *slabs, int n_blocks, int n_out, int n_in, int a1_layout,
                         int a1_levels, int a1_feats, float *dW, float *db, void *stream);

/* Every weight gradient of one decoder in 2-3 launches: layers with the same kernel variant share one slab launch
 * (grid.y = layer), one finish launch sums all of them.  Same arithmetic (and bits) as pag_mlp_wgrad +
 * pag_mlp_wgrad_finish per layer.  Field meanings as in those two; n_layers <= 6; M >= 1. */
typedef struct pag_wgrad_layer {
    const void *dz;          /* bf16 [M, dz_cols] */
    int dz_cols, n_out;
    const void *a1;          /* layer input (a1_dtype, a1_layout, k1 columns) */
    int a1_dtype, a1_layout, k1;
    const float *a2;         /* gathered second input or NULL */
    int k2p;
    const int32_t *a2_index;
    int n_in;
    float *slabs;            /* f32 [n_blocks, round32(n_out), 96] */
    int n_blocks;
    int a1_levels, a1_feats; /* XCD8 inputs only */
    float *dW, *db;          /* f32 [n_out, n_in] (XCD8 inputs: [n_out, a1_levels * a1_feats]), f32 [n_out] */
} pag_wgrad_layer;
int pag_mlp_wgrad_batch(const pag_wgrad_layer *layers, int n_layers, int64_t M, void *stream);

/* ------------------------------------------------------------------------------------------
 * Ray march (wisp OctreeAS.raymarch, 'ray' mode) - tracers/panoptic_packed_rf_tracer.py:85-86
 * ------------------------------------------------------------------------------------------ */

/* Pass 1: per-ray number of surviving samples.
 *   origins, dirs f32 [N,3]; tvals f32 [S] = linspace(0,1,S); jitter f32 [N,S] in [0,1)
 *   occupancy_bits u32 [R^3/32] (bit (x*R+y)*R+z), R = 2^blas_level, or NULL for a dense grid
 *   counts i32 [N] */
int pag_raymarch_count(const float *origins, const float *dirs, int64_t N, int S,
                       const float *tvals, const float *jitter, float dist_min, float dist_max,
                       const uint32_t *occupancy_bits, int blas_level, int32_t *counts,
                       void *stream);
/* pack_start i64 [N+1]: exclusive prefix sums of counts i32 [N] (pack_start[N] = total number of samples) - the write
 * offsets pag_raymarch_pack takes and the per-ray pack table of the compositing kernels (kaolin's
 * mark_pack_boundaries / cumsum bookkeeping, tracers/panoptic_packed_rf_tracer.py:114).  One workgroup.
 * total_host (optional): device-accessible PINNED HOST memory that also receives pack_start[N] (system-scope release store):
 * a host that preset it to a negative value can poll it instead of issuing a stream-synchronising read-back. */
int pag_pack_offsets(const int32_t *counts, int64_t N, int64_t *pack_start, int64_t *total_host, void *stream);
/* Pad a packed batch to `capacity` samples (a multiple of samples_per_entry) WITHOUT the host knowing the sample count: the
 * M = pack_start[N] real samples are followed by filler samples that belong to no pack (coordinates 0, depth 0, delta 0, ray index
 * N - 1; pack_start is not modified).  Per-ray kernels never touch them; per-sample kernels compute on them and the results are
 * ignored; with the per-sample gradient tensors of the compositing backward zero-filled past M they carry zero gradient.  The step
 * after the march then has shapes that do not depend on device data and can be replayed as a HIP graph (pagnerf_amd/graphs.py - the
 * reference's boolean-mask indexing, wisp OctreeAS.raymarch, forces a host read-back instead).
 * M > capacity: no filler is written (the caller learns M from pag_pack_offsets' total_host, discards the step and falls back to
 * exact shapes).  pack_start_clamped i64 [N + 1] (optional, ABI 8) receives min(pack_start[r], capacity): the pack table to hand to
 * every launch that is queued on capacity-sized views BEFORE the host knows M - equal to pack_start when the batch fits, a
 * truncated batch otherwise, so that no per-ray kernel ever indexes a per-sample tensor past `capacity`.
 * ridx_sample i32 [capacity] / ridx_entry i32 [capacity / k] / ridx64 i64 [capacity / k] may be NULL; pidx i32 [capacity / k]. */
int pag_pad_packed(const int64_t *pack_start, int64_t N, int64_t capacity, int samples_per_entry, float *samples, float *depths,
                   float *deltas, int32_t *ridx_sample, int32_t *ridx_entry, int64_t *ridx64, int32_t *pidx,
                   uint8_t *boundary, int64_t *pack_start_clamped, void *stream);

/* pag_pack_offsets + pag_pad_packed + the copy of the ray directions (dirs_src f32 [N,3] -> dirs_dst, both may be NULL) as ONE
 * launch (16 workgroups: each repeats the scan for the total, the first writes the tables, all share the filler stores and the
 * copy): the head of a graph-replayed step (pagnerf_amd/graphs.py).  Arguments as the two entry points';
 * pack_start_clamped is required.  The fillers lie behind the samples the pack pass writes, so it may be queued before that pass. (ABI 9) */
int pag_pack_offsets_pad(const int32_t *counts, int64_t N, int64_t *pack_start, int64_t *total_host, int64_t capacity,
                         int samples_per_entry, float *samples, float *depths, float *deltas, int32_t *ridx_sample,
                         int32_t *ridx_entry, int64_t *ridx64, int32_t *pidx, uint8_t *boundary, int64_t *pack_start_clamped,
                         const float *dirs_src, float *dirs_dst, void *stream);

/* Up to 16 device-to-device copies in ONE launch (pagnerf_amd/graphs.py: the caller-owned copies of a graph replay's static
 * outputs, the upstream gradients into the backward graph's static inputs).  dst / src: host arrays of device pointers, nbytes[i]
 * bytes each (0 = skipped); 16-byte pieces where both pointers are 16-byte aligned.  (ABI 9) */
int pag_copy_batch(int n, void *const *dst, const void *const *src, const int64_t *nbytes, void *stream);

/* View-direction embedding of the colour decoder (wisp PositionalEmbedder on -ray_d, pc_nerf/panoptic_delta_nef.py:196-200):
 * out f32 [R, width] = (-d, sin(-d 2^k) for k < n_freq, cos(-d 2^k) for k < n_freq), frequency-major, zero padded;
 * dirs f32 [R,3], width >= 3 + 6 n_freq. */
int pag_view_embed(const float *dirs, int64_t R, int n_freq, int width, float *out, void *stream);

/* Pass 2: pack.  offsets i64 [N] = exclusive prefix sum of counts.
 *   ridx i32 [M], pidx i32 [M] (linear cell id), samples f32 [M,3], depths f32 [M],
 *   deltas f32 [M], boundary u8 [M] (1 at the first sample of each ray);
 *   ridx64 i64 [M] or NULL: the ray ids once more in the int64 form wisp's raymarch returns */
int pag_raymarch_pack(const float *origins, const float *dirs, int64_t N, int S,
                      const float *tvals, const float *jitter, float dist_min, float dist_max,
                      const uint32_t *occupancy_bits, int blas_level, const int64_t *offsets,
                      int32_t *ridx, int32_t *pidx, float *samples, float *depths, float *deltas,
                      uint8_t *boundary, int64_t *ridx64, void *stream);

/* 'voxel' mode (wisp OctreeAS.raymarch after pc_nerf/trainer.py:362-366 switches the tracer): every ray is walked
 * through the 2^blas_level occupancy grid (3-D DDA); each occupied cell it crosses (a "nugget") receives
 * samples_per_voxel samples.  Pass 1 counts nuggets per ray; pass 2 (offsets = exclusive prefix sum) writes
 *   ridx i32 [M'], pidx i32 [M'] per nugget; samples f32 [M',k,3], depths f32 [M',k], deltas f32 [M'*k],
 *   boundary u8 [M'*k] (1 at the first sample of a ray's first nugget) - the shapes
 *   tracers/panoptic_packed_rf_tracer.py:88-108 indexes. */
/* counts (out, i32 [N]) are in SAMPLES (nuggets * samples_per_voxel): pag_pack_offsets() of them is at once the pack
 * pass's offset table and the compositing kernels' pack_start (one pack per ray, empty packs allowed).
 *   max_travel        the travel filter of tracers/panoptic_packed_rf_tracer.py:88-108 applied inside the walk: a nugget is kept
 *                     iff (depth of its first sample) - (depth of the ray's first sample) < max_travel (strict; fp32 subtraction
 *                     as in the tensor expression :91-92).  INFINITY disables it (plain OctreeAS.raymarch 'voxel' output).
 *   occupancy_coarse  optional u32 bitfield of the (2^blas_level / 4)^3 grid from pag_occupancy_coarse(): the walk consults it
 *                     from LDS and reads the fine word only inside non-empty coarse cells.  Same result with or without.
 *   ridx_sample       optional i32 [M'*k]: the ray of every SAMPLE (the decoders' per-sample ray index); ridx64 optional
 *                     i64 [M'] copy of ridx (wisp hands out int64 ray ids). */
int pag_raymarch_voxel_count(const float *origins, const float *dirs, int64_t N, int samples_per_voxel,
                             float dist_min, float dist_max, const uint32_t *occupancy_bits,
                             const uint32_t *occupancy_coarse, int blas_level, float max_travel,
                             int32_t *counts, void *stream);
int pag_raymarch_voxel_pack(const float *origins, const float *dirs, int64_t N, int samples_per_voxel,
                            float dist_min, float dist_max, const uint32_t *occupancy_bits,
                            const uint32_t *occupancy_coarse, int blas_level, float max_travel,
                            const int64_t *offsets, int32_t *ridx, int32_t *pidx, float *samples,
                            float *depths, float *deltas, uint8_t *boundary, int32_t *ridx_sample,
                            int64_t *ridx64, void *stream);

/* The same two passes with ONE walk: pass 1 also records every kept nugget (t_in, t_out) and its cell, pass 2 turns them into the
 * packed arrays in parallel (one wave per ray) instead of walking every ray a second time.  Outputs are bit-identical to
 * pag_raymarch_voxel_count / _pack.
 *   nugget_t     f32 [2][cap][N][2], nugget_cell i32 [2][cap][N]  caller-allocated scratch, cap = pag_raymarch_voxel_nugget_capacity(blas_level)
 *                (3 * 2^blas_level + 3: the most cells a ray can cross).  First half: the walk's candidates, [step][ray]; second half (ABI 11): the
 *                kept nuggets ray-major, [ray][slot] - what _pack_nuggets (which takes blas_level for `cap`) reads */
int64_t pag_raymarch_voxel_nugget_capacity(int blas_level);
int pag_raymarch_voxel_count_nuggets(const float *origins, const float *dirs, int64_t N, int samples_per_voxel,
                                     float dist_min, float dist_max, const uint32_t *occupancy_bits,
                                     const uint32_t *occupancy_coarse, int blas_level, float max_travel,
                                     int32_t *counts, float *nugget_t, int32_t *nugget_cell, void *stream);
int pag_raymarch_voxel_pack_nuggets(const float *origins, const float *dirs, int64_t N, int samples_per_voxel,
                                    const int64_t *offsets, const float *nugget_t, const int32_t *nugget_cell, int blas_level,
                                    int32_t *ridx, int32_t *pidx, float *samples, float *depths, float *deltas,
                                    uint8_t *boundary, int32_t *ridx_sample, int64_t *ridx64, void *stream);

/* Coarse occupancy for the voxel march: bit ((x/4)*RC + y/4)*RC + z/4 (RC = 2^blas_level / 4) is set iff any of the 64 fine
 * cells under it is.  pag_occupancy_coarse_bytes() = size of `coarse` in bytes, 0 when blas_level is outside [5,8]
 * (the march then runs without it).  Rebuild after every prune (pc_nerf/panoptic_delta_nef.py:98-104). */
int64_t pag_occupancy_coarse_bytes(int blas_level);
int pag_occupancy_coarse(const uint32_t *occupancy_bits, int blas_level, uint32_t *coarse, void *stream);

/* Occupancy update of pc_nerf/panoptic_delta_nef.py:63-104 (prune), one launch:
 *   occupancy[i] <- max(density[i * density_stride], occupancy[i] * decay)     (:74, :90)
 *   bit i of occupancy_bits <- occupancy[i] > min_density                      (:75, :98-104)
 * density f32 (one value per dense cell, x slowest), occupancy f32 [num_cells] (in/out),
 * occupancy_bits u32 [ceil(num_cells/32)] (out) - the bitfield pag_raymarch_* consume. */
int pag_occupancy_update(const float *density, int64_t density_stride, float *occupancy,
                         uint32_t *occupancy_bits, int64_t num_cells, float decay,
                         float min_density, void *stream);

/* ------------------------------------------------------------------------------------------
 * Alpha compositing - kaolin spc_render.exponential_integration / sum_reduce as used at
 * tracers/panoptic_packed_rf_tracer.py:134-182,197-205
 * ------------------------------------------------------------------------------------------ */

/* Segmented exclusive scan + per-ray sums for the base channels.
 *   pack_start i64 [P+1] first packed sample of each non-empty ray (pack), ray_of_pack i32 [P]
 *   sigma f32 [M] (density, already ReLU'd), deltas f32 [M], depths f32 [M] or NULL,
 *   rgb f32 [M,3] or NULL
 *   weights f32 [M] (out): w_i = exp(-sum_{j<i} tau_j) (1 - exp(-tau_i)), tau = sigma*delta
 *   out_alpha f32 [N], out_rgb f32 [N,3], out_depth f32 [N], out_hit u8 [N]: rows of rays that
 *   have a pack are overwritten; the caller pre-fills the rest with the background.
 *   n_samples (ABI 9): 0, or the length of `weights` when the batch carries filler samples past pack_start[P]
 *   (pag_pad_packed): weights[pack_start[P] .. n_samples) is zeroed by the same launch. */
int pag_composite_fwd(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P,
                      const float *sigma, const float *deltas, const float *depths,
                      const float *rgb, int bg_color, float *weights, float *out_alpha,
                      float *out_rgb, float *out_depth, uint8_t *out_hit, int64_t n_samples, void *stream);

/* Gradients of pag_composite_fwd w.r.t. sigma and rgb.
 *   g_rgb f32 [N,3] / g_depth f32 [N] / g_alpha f32 [N]: upstream gradients (any may be NULL)
 *   d_sigma f32 [M], d_rgb f32 [M,3] (NULL allowed when rgb is NULL)
 *   n_samples (ABI 9): 0, or the length of d_sigma / d_rgb of a padded batch: the rows of the filler samples
 *   [pack_start[P], n_samples) are zeroed by the same launch (they carry no gradient). */
int pag_composite_bwd(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P,
                      const float *sigma, const float *deltas, const float *depths,
                      const float *rgb, int bg_color, const float *weights,
                      const float *out_alpha, const float *g_rgb, const float *g_depth,
                      const float *g_alpha, float *d_sigma, float *d_rgb, int64_t n_samples, void *stream);

/* Pose optimisation (pc_nerf/ba_pipeline.py:85-92): gradients of samples[m] = origins[ray] + dirs[ray] * depths[m] with respect to the rays -
 * out f32 [N,6], row r = (sum of grad_samples over ray r's pack | sum of grad_samples * depth); rows of rays without a pack are NOT written
 * (the caller zero-fills when packs do not cover every ray).  One pass over grad_samples f32 [M,3] and depths f32 [M].  (ABI 7) */
int pag_ray_sample_grad(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P, const float *grad_samples, const float *depths,
                        float *out, void *stream);

/* Per-ray weighted feature sums (tracer :197-205): out[ray, c] = alpha[ray] * sum_i w_i f[i, c].
 *   feats [M,C] row-major (feat_dtype F32 or BF16); out f32 [N,C] (rows of rays with a pack) */
int pag_composite_feats_fwd(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P,
                            const float *weights, const float *alpha, const void *feats,
                            int feat_dtype, int C, float *out, void *stream);
/* d feats[i, c] = alpha[ray] * w_i * g_out[ray, c]   (weights and alpha are detached, :148-155);
 * d_feats [M,C] in feat_dtype (F32 or BF16). */
int pag_composite_feats_bwd(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P,
                            const float *weights, const float *alpha, const float *g_out, int C,
                            void *d_feats, int feat_dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * Linear-assignment instance loss, device half (loss/lin_assignment_things.py:23-54,
 * loss/lin_assignment.py:16-26, utils/outlier_rejection.py:56-71)
 * ------------------------------------------------------------------------------------------ */

/* sums[k, c] = sum over rays p with labels_gt[p] == label_list[k] (and row_mask[p] != 0 when given)
 *              of values[p, col0 + c];   counts[k] = number of such rays.
 *   values [P, row_stride] (PAG_F32 or PAG_BF16), labels_gt i64 [P], row_mask u8 [P] or NULL,
 *   label_list i64 [K] (device), sums f32 [K, C], counts i32 [K].
 * fp32 accumulation in ray order (the order of a sequential sum over dim 0); the caller forms
 * cost = -(sums / (counts + 1e-4)) (:33) and runs the Hungarian solver on the host. */
int pag_label_sums(const void *values, int value_dtype, int64_t P, int64_t row_stride, int col0, int C,
                   const int64_t *labels_gt, const uint8_t *row_mask, const int64_t *label_list, int K,
                   float *sums, int32_t *counts, void *stream);

/* The same loss with ONE host synchronisation per step (ABI 10; loss/lin_assignment_things.py:23-82).
 *
 * All three take a BATCH of B images in one set of launches (the reference loops over the images of a step, :58): prob is
 * [B][P] rows of n_cols floats with element strides image_stride / row_stride, every other array is contiguous with the image as its
 * leading dimension (labels_gt [B,P], stuff_mask [B,P], info [B,2], labels / targets [B,max_rows], cost [B,max_rows,n_cols-col0],
 * sums_ws [B,max_rows,n_cols-col0], counts_ws [B,max_rows], virt / loss / valid / grad [B,P], wrong [B], d_prob [B,P,n_cols]).
 * Per image:
 * pag_assign_cost (three launches):
 *   labels i64 [max_rows]          the distinct positive gt ids of the image, ascending: `sorted(torch.unique(things_gt))[:max_rows]`
 *                                  (:29); entries past info[0] hold a sentinel no ray carries.  max_rows <= 1024.
 *   cost   f32 [max_rows, n_cols - col0]   row r = -(sum of prob[p, col0:] over the rays with gt == labels[r]) / (count + 1e-4)
 *                                  (:31-33: fp32 sums in ray order, int64 count + python float -> fp32, fp32 division); rows past info[0] untouched
 *   info   i32 [2]                 info[0] = number of labels, info[1] = 1 if the image carries more than 1024 distinct ids (take the
 *                                  general path: pag_label_sums with an explicit list)
 * so that ONE fixed-size device-to-host copy carries everything SciPy needs.  sums_ws f32 [max_rows, n_cols - col0] and counts_ws i32
 * [max_rows] are scratch.
 *
 *   points f32 [B,P,3] or NULL (outlier rejection, :38-43 + utils/outlier_rejection.py:8-51,56-71): the per-id centres of the same labels and
 *                                  id_lo_hi i32 [B,max_rows,2] = the id range [lo, hi] each may take (id_slope = (max_num_inst_at_x + id_margin) /
 *                                  frame_min_length and id_x_limit = (n_ids - id_margin) / id_slope as fp32, n_ids = n_cols - col0); every id outside
 *                                  costs 10000 - applied by the caller on the host.  psums_ws f32 [B,max_rows,3], pcounts_ws i32 [B,max_rows]: scratch.
 *
 * pag_assign_nll_fwd: per ray  valid = stuff_mask | gt > 0 (:60; stuff_mask u8 [P] or NULL),
 *   virt = gt > 0 ? (gt == labels[r] for some r < info[0] ? targets[r] : default_label) : 0   - targets i64 [max_rows] holds the assignment
 *   (:47-53: assigned column + 1; default_label = 1 for ids that got none), arg-max of the ray's n_cols probabilities,
 *   loss[ray] = -log(prob[ray, virt] + 1e-27) (:80) for valid rays when ANY valid ray of the image has virt != arg-max (:79), else 0;
 *   wrong i32 [1] (zeroed by the caller) receives that flag, virt / valid are kept for the backward.
 * pag_assign_nll_bwd: d_prob f32 [P, n_cols] (contiguous, every element written) from grad f32 [P]. */
int pag_assign_cost(const float *prob, int B, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols, int col0,
                    const int64_t *labels_gt, int max_rows, float *sums_ws, int32_t *counts_ws, int32_t *info, int64_t *labels,
                    float *cost, const float *points, float id_slope, float id_x_limit, int id_margin, float *psums_ws,
                    int32_t *pcounts_ws, int32_t *id_lo_hi, void *stream);
/* The Hungarian step itself on the device (ABI 13): replaces `rows, cols = scipy.optimize.linear_sum_assignment(np.nan_to_num(cost))` of
 * loss/lin_assignment_things.py:45 (loss/lin_assignment.py:22) for the batch pag_assign_cost prepared - no device-to-host copy, no host wait, the train step
 * stays one uninterrupted stream of launches.  One wave per image runs SciPy's algorithm (rectangular_lsap.cpp: shortest augmenting paths, Crouse 2016) in
 * float64 in SciPy's operation order and tie rule, so the assigned columns are the same integers (oracle/lin_assign.py::lsap_jv is the sequential
 * restatement, pinned against the installed SciPy; tests/test_gpu_loss.py compares the kernel with SciPy itself on random, tie-heavy and masked matrices).
 *   cost f32 [B,max_rows,n_ids] (pag_assign_cost's output; rows >= info[b][0] are not read), widened to float64; id_lo_hi i32 [B,max_rows,2] or NULL: columns
 *   outside [lo, hi] of a row cost 10000 (utils/outlier_rejection.py:8-51); NaN -> 0, +-inf -> +-DBL_MAX (np.nan_to_num).
 *   targets i64 [B,max_rows]: assigned column + 1 for the rows < info[b][0], 1 elsewhere (:47-53) - what pag_assign_nll_fwd reads.
 *   status i32 [B]: 0 solved; 1 not attempted (info[b][1] set: more distinct ids than pag_assign_cost holds; targets all 1); 2 infeasible matrix (SciPy raises).
 * max_rows, n_ids <= 256. */
int pag_assign_solve(const float *cost, int B, int max_rows, int n_ids, const int32_t *info, const int32_t *id_lo_hi, int64_t *targets, int32_t *status,
                     void *stream);
int pag_assign_nll_fwd(const float *prob, int B, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols,
                       const int64_t *labels_gt, const uint8_t *stuff_mask, const int64_t *labels, const int64_t *targets,
                       const int32_t *info, int max_rows, int64_t default_label, int64_t *virt, float *loss, uint8_t *valid,
                       int32_t *wrong, void *stream);
int pag_assign_nll_bwd(const float *prob, int B, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols,
                       const int64_t *virt, const uint8_t *valid, const int32_t *wrong, const float *grad, float *d_prob,
                       void *stream);

/* segment_consistency_regularizer (ABI 12; loss/regularizers.py:5-35, called at pc_nerf/trainer.py:525-527 on `inst_embedding + 1e-27`) over a batch of B images
 * without a host synchronisation.  prob f32: [B][P] rows of n_cols probabilities (element strides image_stride / row_stride), eps is added to every
 * probability read (the caller's `+ 1e-27`), labels i64 [B,P] contiguous.  Per image every distinct value of labels is a segment (at most 2048 per image and
 * never INT64_MIN - otherwise out is NaN and the gradient 0); per segment: histogram of its rays' first arg-max column (:22), skipped when every ray predicts
 * column 0 (:24-25), label = first most frequent column among 1.. (:27) or 0 when bins[0] * 0.5 > bins[label] (:29-30), term = mean over its rays of
 * -log(prob[ray, label] + eps) (:32); the running total is divided by each image's segment count in turn (:33) and finally by B (:35) -> out f32 [1].
 * workspace (pag_segment_reg_workspace_bytes(B, P) bytes) keeps what the backward needs; pag_segment_reg_bwd writes EVERY element of d_prob f32 [B,P,n_cols]
 * (contiguous) from grad f32 [1].  Sums are lane-strided with a fixed butterfly: bitwise reproducible. */
int64_t pag_segment_reg_workspace_bytes(int B, int64_t P);
int pag_segment_reg_fwd(const float *prob, int B, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols, float eps,
                        const int64_t *labels, void *workspace, int64_t workspace_bytes, float *out, void *stream);
int pag_segment_reg_bwd(const float *prob, int B, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols, float eps,
                        const void *workspace, int64_t workspace_bytes, const float *grad, float *d_prob, void *stream);

/* ------------------------------------------------------------------------------------------
 * Per-ray training loss of the rendered buffers (pc_nerf/trainer.py:443-446 rgb, :459-465 semantics,
 * loss/lin_assignment_things.py:80 instance term after the assignment) - one launch forward, one backward
 * ------------------------------------------------------------------------------------------ */

/* loss = rgb_weight * mean|rgb - rgb_gt|
 *      + sum over the (up to two) NLL terms t of  weight_t * sum_n l_tn / denom_t,
 *        l_tn = -log(prob_t[n, target_t[n]] + eps) * inv_temp_t * (conf_t[n] if conf_t else 1)
 *        for rows with 0 <= target < C_t (others - F.nll_loss's ignore_index included - contribute nothing);
 *        denom_t = N if all_t (reduction 'none' then .mean(), trainer.py:459-465) else the number of valid rows
 *        (F.nll_loss reduction 'mean').
 *   rgb, rgb_gt f32 [N,3] (both NULL: no rgb term); prob_t f32 [N, C_t] contiguous (NULL: term absent);
 *   target_t i64 [N]; conf_t f32 [N] or NULL.
 *   workspace: pag_render_loss_workspace_bytes() bytes, ZEROED ONCE by the caller and then reused as is.
 *   out f32 [6]: total, rgb term, term A, term B, denom A, denom B.
 * Deterministic (fixed-order block partials, the last block adds them in block order). */
int64_t pag_render_loss_workspace_bytes(void);
int pag_render_loss_fwd(const float *rgb, const float *rgb_gt, int64_t N, float rgb_weight,
                        const float *prob_a, int C_a, const int64_t *target_a, const float *conf_a, float weight_a,
                        float inv_temp_a, int all_a,
                        const float *prob_b, int C_b, const int64_t *target_b, const float *conf_b, float weight_b,
                        float inv_temp_b, int all_b,
                        float eps, void *workspace, float *out, void *stream);

/* Gradients of the above times the upstream scalar *g (device pointer; NULL = 1): d_rgb [N,3] = g rgb_weight sgn(rgb-gt)/(3N),
 * d_t [N, C_t] dense: -g weight_t inv_temp_t conf / (denom_t (p + eps)) in the target column of valid rows, 0 elsewhere
 * (every element is written).  fwd_out = the forward's `out`.  Any of d_rgb / d_a / d_b may be NULL. */
int pag_render_loss_bwd(const float *g, const float *fwd_out, const float *rgb, const float *rgb_gt, int64_t N,
                        float rgb_weight,
                        const float *prob_a, int C_a, const int64_t *target_a, const float *conf_a, float weight_a,
                        float inv_temp_a, int all_a,
                        const float *prob_b, int C_b, const int64_t *target_b, const float *conf_b, float weight_b,
                        float inv_temp_b, int all_b,
                        float eps, float *d_rgb, float *d_a, float *d_b, void *stream);

/* ---- optimiser step of the train step's caller (round 4, ABI 9) ------------------------------------------------------------------
 * torch.optim.Adam(params, lr, betas, eps, weight_decay) as the reference builds it (config_parser.py:667-673: eps = 1e-15;
 * pc_nerf/trainer.py:268-286 parameter groups; :583 scaler.step(optimizer)) for fp32 tensors, op for op the single-tensor formula
 * of torch/optim/adam.py (maximize / amsgrad off):  g += weight_decay p ; m += (g - m)(1 - beta1) ; v = v beta2 + (1 - beta2) g g ;
 * p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps).   `step` = the step count AFTER this update (>= 1).
 * n_tensors host arrays of device pointers (f32, contiguous, numel[i] elements each); ONE launch per 48 tensors (704 MB per step
 * for the two 50.3 MB tables: the update is a pure stream; a tensor gets one block per 4096 elements, at most 4096). */
int pag_adam_step(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                  const int64_t *numel, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                  void *stream);

/* ---- pose optimisation: the per-ray camera transform (round 5, ABI 11) -------------------------------------------------------------
 * pc_nerf/ba_pipeline.py:85-92 `transform_rays` (kaolin `inv_transform_rays` on the 'matrix_6dof_rotation' camera parameters registered at
 * :49-51, then re-normalised directions) - what configs/bup20/best.yaml runs at the head of EVERY train step (optimize_extrinsics with
 * extrinsics_epoch_end 900 > epochs 800, pc_nerf/trainer.py:308) - and its gradient.
 *   params f32 [C,9] = (a1, a2, t) per camera:  b1 = a1/|a1|, b2 = normalise(a2 - (b1.a2) b1), b3 = b1 x b2,  R rows = (b1, b2, b3)
 *   cam i32 [ceil(N / rays_per_entry)]: ray i belongs to camera cam[i / rays_per_entry] (rays_per_entry = 1: one index per ray; = rays per
 *   image: one per image, the layout of `transform_rays`).  Indices must lie in [0, C); one outside is clamped to 0 / C - 1 in the forward AND the backward
 *   (the ray renders through that camera and sends its gradient there) - the tensor-op form would raise or wrap instead.
 *   origins_w[i] = sum_k (origins_c[i] - t)[k] R[k]        dirs_w[i] = normalise(sum_k dirs_c[i][k] R[k])          all f32 [N,3]
 * _bwd: d_params f32 [C,9] = d loss / d params from g_origins / g_dirs f32 [N,3] (either may be NULL = zero); EVERY row is written (zeros
 * for cameras without a ray in the batch); two launches (per-(camera, ray slice) partial sums into `workspace`, >= pag_pose_rays_bwd_workspace_bytes(C)
 * bytes; then one thread per camera), fixed summation order (bitwise reproducible). */
int pag_pose_rays_fwd(const float *params, int64_t C, const int32_t *cam, int64_t rays_per_entry, const float *origins_c, const float *dirs_c,
                      int64_t N, float *origins_w, float *dirs_w, void *stream);
int64_t pag_pose_rays_bwd_workspace_bytes(int64_t C);
int pag_pose_rays_bwd(const float *params, int64_t C, const int32_t *cam, int64_t rays_per_entry, const float *origins_c, const float *dirs_c,
                      int64_t N, const float *g_origins, const float *g_dirs, float *d_params, void *workspace, int64_t workspace_bytes, void *stream);

/* The position gradient of pag_*_encode_bwd_xyz reduced PER RAY inside the gather pass (pose optimisation, pc_nerf/ba_pipeline.py:85-92: the packed samples
 * are origin[ray] + dir[ray] * depth, so what is wanted of d loss / d xyz is its per-ray sum and its per-ray sum weighted by depth):
 *   out f32 [N,6], row r = (sum of d xyz over ray r's samples | sum of d xyz * depth); rays without samples get zeros.
 *   ridx i32 [M] = ray of each packed sample (non-decreasing), depths f32 [M], pack_start i64 [N+1] (one pack per ray, empty packs allowed);
 *   workspace >= pag_encode_bwd_rays_workspace_bytes(M, N) (6 floats per (XCD group, wave of 64 samples, ray) instead of 8 x 3 floats per sample).
 * Replaces pag_*_encode_bwd_xyz + pag_ray_sample_grad where the samples come straight from the ray march (fixed summation order: bitwise reproducible;
 * the sums are formed in a different order than by those two calls, so the values agree to fp32 rounding, not bit for bit).  Other arguments as
 * pag_*_encode_bwd_xyz.  (ABI 11) */
int64_t pag_encode_bwd_rays_workspace_bytes(int64_t M, int64_t N);
int pag_hash_encode_bwd_rays(const float *xyz, int64_t M, const void *tables, int table_dtype, const void *grad_out, int grad_dtype,
                             int64_t g_stride_m, int64_t g_stride_c, int layout, int n_levels, int n_feat, int log2_T,
                             const float *resolutions_host, const float *feat_scale_host, const int32_t *ridx, const float *depths,
                             const int64_t *pack_start, int64_t N, float *out, void *workspace, int64_t workspace_bytes, int flags, void *stream);
int pag_permuto_encode_bwd_rays(const float *xyz, int64_t M, const void *tables, int table_dtype, const void *grad_out, int grad_dtype,
                                int64_t g_stride_m, int64_t g_stride_c, int layout, int n_levels, int n_feat, uint32_t capacity,
                                const float *scale_factor_host, const float *shift_host, const float *feat_scale_host, const int32_t *ridx,
                                const float *depths, const int64_t *pack_start, int64_t N, float *out, void *workspace, int64_t workspace_bytes,
                                int flags, void *stream);

/* utils/outlier_rejection.py:74-97 `rays_to_3d_points` as pc_nerf/trainer.py:508-518 calls it (ABI 12): points f32 [N,3] = the camera-frame base rays
 * unprojected by depth f32 [N] and mapped to the world by the cameras' current extrinsics - sum_k (o_c - t + d_c depth)[k] R[k] with params / cam /
 * rays_per_entry as in pag_pose_rays_fwd.  No gradient (the trainer wraps the call in torch.no_grad()). */
int pag_pose_points(const float *params, int64_t C, const int32_t *cam, int64_t rays_per_entry, const float *origins_c, const float *dirs_c,
                    const float *depth, int64_t N, float *points, void *stream);

/* Gradient of pag_view_embed with respect to the directions (the view direction depends on the camera rotation: pc_nerf/ba_pipeline.py:89-90
 * -> pc_nerf/panoptic_delta_nef.py:196-200): d_dirs f32 [R,3] from g_out f32 [R, width].  (ABI 11) */
int pag_view_embed_bwd(const float *dirs, int64_t R, int n_freq, int width, const float *g_out, float *d_dirs, void *stream);

/* Touched-rows exchange of a table gradient (ABI 14; this build's multi-GPU addition, pagnerf_amd/shard.py::SparseRows - the reference is single-GPU): the four passes
 * around the collective.  After the first prune (configs/bup20/best.yaml:187, pc_nerf/trainer.py:362-366) the coarse and middle lattice levels touch few of their rows and
 * pag_*_encode_bwd_set leaves exact zeros elsewhere, so only the UNION over the ranks of the non-zero rows has to travel.
 *   pag_sparse_rows_mask    grad f32 [L][T][F] -> bits u32 [L][W], W = ceil(T / 32): bit r of word w = any(grad[l][32 w + r][:] != 0)
 *                           (the caller all_gathers the ranks' bits and ORs them)
 *   pag_sparse_rows_plan    union bits, caps i32 [L] (slots per level; caps[l] >= T: the level travels whole) -> word_prefix i32 [L][W] (union rows of the level before
 *                           word w; 32 w for a whole level), counts i64 [L + 1] (union rows per level; [L] = rows that do not fit their level's slots)
 *   pag_sparse_rows_pack    member rows in row order -> buf f32 [sum of slots][F] at offs[l] + rank while rank < caps[l] (offs i64 [L]); the caller zero-fills buf
 *   pag_sparse_rows_unpack  EVERY row of grad rewritten: a member with rank < caps[l] takes buf[offs[l] + rank], any other row 0
 * 1 <= F <= 64, T <= 2^31. */
int pag_sparse_rows_mask(const float *grad, int L, int64_t T, int F, uint32_t *bits, void *stream);
int pag_sparse_rows_plan(const uint32_t *bits, int L, int64_t T, const int32_t *caps, int32_t *word_prefix, int64_t *counts, void *stream);
int pag_sparse_rows_pack(const float *grad, int L, int64_t T, int F, const uint32_t *bits, const int32_t *word_prefix, const int32_t *caps, const int64_t *offs,
                         float *buf, void *stream);
int pag_sparse_rows_unpack(const float *buf, int L, int64_t T, int F, const uint32_t *bits, const int32_t *word_prefix, const int32_t *caps, const int64_t *offs,
                           float *grad, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PAGNERF_HIP_H */
