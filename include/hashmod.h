/*
 * hashmod.h - C ABI of libhashmod.so, the MI355X (gfx950) implementation of the
 * hash-grid encode / SDF-MLP / sphere-tracing hot path of HashModNFFBanks-IDR.
 *
 * This is the drop-in boundary: plain C, raw device pointers and sizes, no torch types.
 * The reference has no native ABI on its live path (its PyTorch modules call ATen ops);
 * each entry point below names the reference Python interface it replaces
 * (paths relative to the reference's code/ directory).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; hm_last_error() returns a
 *     thread-local message for the last failing call on this thread (no exceptions cross
 *     the ABI; the reference reports the same conditions as Python exceptions);
 *   - the caller owns all memory; pointers are DEVICE pointers unless marked [host];
 *     nothing is retained after the call returns; workspace is passed in explicitly;
 *   - `stream` is a hipStream_t (as void*); kernels are launched on it asynchronously and
 *     the functions never synchronise;  NULL means the HIP default stream;
 *   - all tensors are fp32, contiguous, row-major unless a stride argument says otherwise;
 *   - functions are re-entrant; the only mutable global is the error string.
 */
#ifndef HASHMOD_H
#define HASHMOD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define HM_API __attribute__((visibility("default")))
#else
#define HM_API
#endif

#define HM_MAX_LEVELS 32
#define HM_MAX_LAYERS 16

/* frac_mode */
#define HM_FRAC_REFERENCE 0 /* reference behaviour: xf = x - x.float() == 0 (hashGridEmbedding.py:86) */
#define HM_FRAC_TRILINEAR 1 /* build-defined opt-in: floor + fractional weights (NOT a parity mode)    */

#define HM_OK 0
#define HM_ERR_INVALID (-1) /* bad argument (reference: ValueError / assert)      */
#define HM_ERR_HIP (-2)     /* HIP runtime / launch failure (reference: RuntimeError) */

HM_API int hm_version(void);
HM_API const char *hm_last_error(void);
/* number of HIP devices visible, or <0; never initialises a context */
HM_API int hm_device_count(void);

/* ---- level table ------------------------------------------------------------------------
 * Replaces the per-level bookkeeping of MultiResHashGridMLP.__init__
 * (model/embeddings/hashGridEmbedding.py:106-147): res[l], rows[l] = min(res^3, 2^T) and the
 * row offset of level l inside the fused [sum(rows), F] table.  The arrays are [host] and are
 * computed by the caller in double precision exactly as the reference does; the library never
 * recomputes them.  n_features is F (max_points_per_level).                                   */
typedef struct hm_grid_desc hm_grid_desc;
HM_API int hm_grid_desc_create(int n_levels, int n_features, const int32_t *res, const uint32_t *rows,
                        const uint64_t *row_off /* [n_levels+1] */, hm_grid_desc **out);
HM_API void hm_grid_desc_destroy(hm_grid_desc *desc);
HM_API int hm_grid_embed_dim(const hm_grid_desc *desc); /* 3 + 2L + L*F (hashGridEmbedding.py:143-145) */

/* ---- hash indices -----------------------------------------------------------------------
 * Replaces hash_func + the index half of _HashGridMLP.forward
 * (hashGridEmbedding.py:32-40, 84-98) for one level: xi = trunc(x*res) and the 8 corner
 * row ids in bin_mask order.  xi_out [n,3] int32 (may be NULL), ids_out [n,8] uint32.        */
HM_API int hm_corner_ids(const hm_grid_desc *desc, int level, const float *x, int64_t n, int32_t *xi_out,
                  uint32_t *ids_out, void *stream);

/* ---- diagnostic: PMC calibration of 8-byte row gathers ---------------------------------------------
 * Not part of the reference's interface.  Groups of `group` consecutive lanes read 8-byte rows of one pseudo-random
 * 128-B block of `table` (table_bytes >> Infinity Cache), lane g at byte offset g*stride_bytes: known bytes and
 * known 32/64/128-B units per block, against which FETCH_SIZE / TCC_EA0_RDREQ are read (profiles/r02_gather_calib.json).
 * out [n] receives a checksum per lane.                                                                  */
HM_API int hm_diag_gather_calib(const float *table, int64_t table_bytes, int64_t n, int group, int stride_bytes,
                                float *out, void *stream);

/* ---- diagnostic: what a stream of exact-fp32 MFMAs sustains --------------------------------------------
 * Not part of the reference's interface.  `workgroups` workgroups of 8 waves (two per SIMD, as the fused SDF kernel)
 * issue `iters` x 16 v_mfma_f32_32x32x2_f32 each on four independent accumulators, operands in registers, no memory
 * traffic: iters * 16 * 4096 flop per wave.  The rate it reaches (bench.py times the call) is the ceiling of the
 * matrix pipe under this instruction - below the nominal 256 flop / clk / CU x 2.4 GHz because the clock drops under
 * the load and an MFMA occupies the pipe for more than its 64 cycles.  out [workgroups * 512] receives checksums.   */
HM_API int hm_diag_mfma_f32_stream(int workgroups, int iters, float *out, void *stream);

/* ---- encoder forward --------------------------------------------------------------------
 * Replaces MultiResHashGridMLP.forward (hashGridEmbedding.py:150-155) including
 * FourierFeature.forward (frequency_enc.py:63-67):
 *   out[i] = [ x(3) | sin(2*pi*x@B)(L) | cos(2*pi*x@B)(L) | level features (L*F) ].
 * x [n,3]; table [sum(rows),F]; B_fourier [3,L]; out rows are out_stride floats apart
 * (out_stride >= E).  All 8 corners of every level are gathered in both frac modes.
 * B_fourier == NULL: only the L*F hash-feature columns are produced, out[i*out_stride + l*F + f]
 * (the torch.cat([level(x) ...]) half of hashGridEmbedding.py:153).                            */
HM_API int hm_encode_fwd(const hm_grid_desc *desc, const float *x, int64_t n, const float *table,
                  const float *B_fourier, float *out, int64_t out_stride, int frac_mode, void *stream);

/* Same operator with a caller-owned scratch buffer of hm_encode_workspace_bytes(desc, n) bytes.  Big launches
 * (n >= 131072 over tables larger than 8 MiB) then bucket the points by z first and gather in that order: the
 * reference hash puts every (x, y) corner of one z-plane into one 16-KB window of a level's table, so z-ordered
 * points share their table lines on every level (out[i] is still the embedding of x[i]; only the order of the
 * work changes).  Without a workspace (or for small launches) the call is identical to hm_encode_fwd.
 * Padded rows: with out_stride a multiple of 4 floats >= ceil(E/4)*4 (272 bytes at E = 67) and a 16-byte aligned `out`
 * every row leaves the z-ordered kernel as ONE dwordx4 store instruction (the pad floats of a row are written as zeros);
 * any other stride takes the dword path.                                                                          */
HM_API int64_t hm_encode_workspace_bytes(const hm_grid_desc *desc, int64_t n);
HM_API int hm_encode_fwd_ws(const hm_grid_desc *desc, const float *x, int64_t n, const float *table,
                     const float *B_fourier, float *out, int64_t out_stride, int frac_mode, void *workspace,
                     int64_t workspace_bytes, void *stream);

/* ---- encoder backward (table) -----------------------------------------------------------
 * Replaces the embedding_dense_backward that autograd runs for nn.Embedding in
 * _HashGridMLP.forward (hashGridEmbedding.py:99-102): d_table[row] += w * d_out.
 * d_feat points at the hash-feature columns of the upstream gradient: d_feat[i*d_feat_stride + l*F + f]
 * (for a full [n,E] gradient pass d_out + 3 + 2L and stride E);
 * d_table [sum(rows),F] is ACCUMULATED into (caller zeroes it, like the reference's
 * optimizer.zero_grad()).  fp32 atomics: the sum is order-dependent in the last bits.         */
HM_API int hm_encode_bwd_table(const hm_grid_desc *desc, const float *x, int64_t n, const float *d_feat,
                        int64_t d_feat_stride, float *d_table, int frac_mode, void *stream);

/* Same gradient with a caller-owned scratch buffer of hm_encode_bwd_workspace_bytes(desc, n) bytes.  Big launches
 * (F = 2, n >= 131072, tables > 8 MiB) bucket the points by z, lay x / d_feat out in that order and let one workgroup
 * per (z-slab, level) accumulate its contributions in LDS copies of the few 2^11-row blocks it touches before adding
 * them to d_table with dense, coalesced atomics; otherwise identical to hm_encode_bwd_table.                     */
HM_API int64_t hm_encode_bwd_workspace_bytes(const hm_grid_desc *desc, int64_t n);
HM_API int hm_encode_bwd_table_ws(const hm_grid_desc *desc, const float *x, int64_t n, const float *d_feat,
                                  int64_t d_feat_stride, float *d_table, int frac_mode, void *workspace,
                                  int64_t workspace_bytes, void *stream);

/* Deterministic form of the same gradient (no atomics).  hm_encode_rows lists, for every (point, level[, corner]),
 * the destination row in the fused table (keys_out [n*L*C] int32, C = 1 in reference frac mode - only corner 0 carries
 * weight - and 8 in trilinear mode, where weights_out [n*L*C] receives the interpolation weights).  The caller sorts
 * the keys with ANY stable sort (perm = the sorting permutation, int64; hm_sort_pairs_i32 below is the library's own);
 * hm_encode_bwd_table_sorted then sums each run
 * of equal keys in sorted order in one thread and adds it to d_table with a plain read-modify-write.  Bitwise
 * reproducible for a given contribution order (torch's embedding_dense_backward is deterministic on the CPU path the
 * reference oracle runs; the atomic kernel above is the fast default).                                                */
HM_API int hm_encode_rows(const hm_grid_desc *desc, const float *x, int64_t n, int frac_mode, int32_t *keys_out,
                          float *weights_out, void *stream);
HM_API int hm_encode_bwd_table_sorted(const hm_grid_desc *desc, const int32_t *keys_sorted, const int64_t *perm,
                                      int64_t n_keys, int corners, const float *d_feat, int64_t d_feat_stride,
                                      const float *weights, float *d_table, void *stream);

/* Stable sort of non-negative int32 keys (< 2^key_bits) with the sorting permutation - the library's own LSD radix sort
 * (8-bit digits, ceil(key_bits/8) passes of three kernel launches; csrc/hm_sort.hip) for the pair
 * hm_encode_rows -> hm_encode_bwd_table_sorted: keys_sorted[i] = keys[perm[i]], equal keys keep their input order.
 * keys and keys_sorted must not alias; workspace: hm_sort_workspace_bytes(n) bytes.  Sync-free, graph-capturable.   */
HM_API int64_t hm_sort_workspace_bytes(int64_t n);
HM_API int hm_sort_pairs_i32(const int32_t *keys, int64_t n, int key_bits, int32_t *keys_sorted, int64_t *perm,
                             void *workspace, int64_t workspace_bytes, void *stream);


/* ---- encoder input gradient (frac_mode = HM_FRAC_TRILINEAR only) ---------------------------------------------
 * In the reference d(hash features)/dx is identically zero (xf = x - x.float() == 0, hashGridEmbedding.py:86), so
 * autograd never runs anything here; these three entry points make the opt-in trilinear mode trainable under IDR's
 * eikonal / normal terms, where ImplicitNetwork.gradient differentiates the embedding with create_graph=True
 * (implicit_differentiable_renderer.py:116-127).  J_i = d feat[i,:] / d x[i,:]  [L*F, 3], exact inside a voxel.
 *   hm_encode_bwd_input      a == NULL: gx[i,:] = J_i^T d_feat[i,:]            (the embedding's backward w.r.t. x)
 *                            a != NULL: gx[i,:] = d/dx ( a[i,:] . J_i^T d_feat[i,:] )   (its second-order term)
 *   hm_encode_jvp            out[i,:] = J_i a[i,:]                              (backward of gx w.r.t. d_feat)
 *   hm_encode_bwd_table_jvp  d_table += d/dT ( sum_i a[i,:] . J_i^T d_feat[i,:] )  (fp32 atomics, accumulates)
 * x, a, gx [n,3]; d_feat / out rows of L*F floats with the given row stride.                                        */
HM_API int hm_encode_bwd_input(const hm_grid_desc *desc, const float *x, int64_t n, const float *table,
                               const float *d_feat, int64_t d_feat_stride, const float *a, float *gx, void *stream);
HM_API int hm_encode_jvp(const hm_grid_desc *desc, const float *x, int64_t n, const float *table, const float *a,
                         float *out, int64_t out_stride, void *stream);
HM_API int hm_encode_bwd_table_jvp(const hm_grid_desc *desc, const float *x, int64_t n, const float *a,
                                   const float *d_feat, int64_t d_feat_stride, float *d_table, void *stream);

/* ---- embedding row: input gradient of the Fourier-feature columns -------------------------------------------------
 * MultiResHashGridMLP.forward returns [x | sin(a) | cos(a) | level features] with a = 2 pi x B (hashGridEmbedding.py:
 * 150-155, frequency_enc.py:63-67).  In the reference autograd differentiates that expression w.r.t. x with
 * create_graph=True (ImplicitNetwork.gradient, implicit_differentiable_renderer.py:116-127: ~10 elementwise kernels
 * and a K=3 matmul per pass) and once more in loss.backward(); the hash features contribute nothing (:86).
 *   hm_fourier_bwd_input      gx[i,:] = d_row[i,0:3] + 2 pi sum_c B[:,c] (cos_c d_sin_c - sin_c d_cos_c)
 *   hm_fourier_bwd_input_bwd  given gg [n,3]:  dd_row = d(gg . gx)/d(d_row)  ([n,width], hash columns zero; may be NULL)
 *                                              d_x    = d(gg . gx)/dx        ([n,3]; may be NULL)
 * B_fourier [3, n_channels]; d_row rows [x(3) | sin(n_channels) | cos(n_channels) | ...] with the given stride.     */
HM_API int hm_fourier_bwd_input(const float *x, int64_t n, const float *B_fourier, int n_channels, const float *d_row,
                                int64_t d_row_stride, float *gx, void *stream);
HM_API int hm_fourier_bwd_input_bwd(const float *x, int64_t n, const float *B_fourier, int n_channels,
                                    const float *d_row, int64_t d_row_stride, const float *gg, float *d_x,
                                    float *dd_row, int64_t dd_row_stride, int width, void *stream);

/* ---- fused SDF network forward (no grad) ---------------------------------------------------
 * Replaces ImplicitNetwork.forward evaluated under torch.no_grad()
 * (model/implicit_differentiable_renderer.py:89-113 + density_net.py:20-30), i.e. the `sdf`
 * callable of RayTracing.forward (model/ray_tracing.py:26-95): embed -> L linear layers with
 * Softplus(beta=100) -> column 0 through tanh(s / (2 + LaplaceDensity(s))).
 *
 * The weight-norm fold W = g*v/||v|| (nn.utils.weight_norm, dim=0) is done by the caller; the
 * library takes each layer's folded matrix in a packed MFMA operand image:
 *   K space of a layer = its input segments back to back, each zero-padded to a multiple of 8:
 *     seg_src 1 = the embedding (E -> ceil(E/8)*8 slots), seg_src 0 = the previous layer's output;
 *     the skip layer (cat[x, emb]/sqrt(2), :99-100) lists {previous output, embedding}, and the
 *     layer BEFORE it sets post_div_sqrt2 (the embedding half is rescaled by the kernel);
 *   rows are zero-padded to n_tiles*32;  n_oct = seg_octets[0] + seg_octets[1];
 *   w_packed[((u*n_oct + g)*64 + l)*4 + s] = W[32u + (l&31)][8g + 4(l>>5) + s]
 *     for tile u < n_tiles, octet g < n_oct, lane l < 64, s < 4   (device pointer, 16-B aligned);
 *   bias: device pointer, n_tiles*32 floats, zero padded, 16-B aligned.
 * hm_mlp_desc itself is a [host] struct.                                                        */
typedef struct hm_mlp_layer {
    const float *w_packed;
    const float *bias;
    int32_t out_dim;        /* true number of output features            */
    int32_t n_tiles;        /* ceil(out_dim/32), at most 16              */
    int32_t seg_octets[2];  /* K octets per input segment (second may be 0) */
    int32_t seg_src[2];     /* 0 previous layer output, 1 embedding      */
    int32_t activation;     /* 1: Softplus(beta=100, threshold=20), 0: none */
    int32_t post_div_sqrt2; /* 1: outputs are divided by sqrt(2) (feeds the skip concat) */
    /* optional second image for the 16-point-tile kernel (small batches); NULL = not provided.
     * Same K space but every segment zero-padded to a multiple of 16; nb = seg_blocks16[0]+[1];
     * w_packed_m16[((u*nb + t)*64 + l)*4 + e] = W[16u + (l&15)][16t + 4(l>>4) + e],  u < 2*n_tiles. */
    const float *w_packed_m16;
    int32_t seg_blocks16[2];
    /* optional bf16 image for hm_sdf_fwd_bf16 (NULL = not provided): same K space as the 16-k image,
     * w_packed_bf16[((u*nb + t)*64 + l)*8 + j] = bf16(W[32u + (l&31)][16t + 8(l>>5) + j]),  u < n_tiles (2-byte elements). */
    const void *w_packed_bf16;
    /* optional SPLIT image for hm_sdf_fwd_split (NULL = not provided): every weight as a pair (hi, lo) of 2-byte floats of
     * the kind hm_mlp_desc.split_kind names, c_w W ~= hi + lo (c_w = 1 for bf16, 2^8 for fp16), same K space as the 16-k
     * image:  w_packed_split[(((u*nb + t)*2 + part)*64 + l)*8 + j] = part(c_w c * W[32u + (l&31)][16t + 8(l>>5) + j]),
     * part 0 = hi, 1 = lo, c = the segment's scale given to hm_pack_mlp_layer_split.                                      */
    const void *w_packed_split;
} hm_mlp_layer;

#define HM_SPLIT_NONE (-1)
#define HM_SPLIT_BF16X2 0 /* hi, lo bf16: 16 significant bits per operand */
#define HM_SPLIT_F16X2 1  /* hi, lo fp16 (activations scaled by 2^4, weights by 2^8): 22 significant bits, |x| <= 4094 */

typedef struct hm_mlp_desc {
    int32_t n_layers;
    float beta; /* LaplaceDensity: |beta_param| + beta_min (density_net.py:28-30) */
    hm_mlp_layer layer[HM_MAX_LAYERS];
    int32_t split_kind; /* kind of the layers' w_packed_split images (HM_SPLIT_*), HM_SPLIT_NONE when absent */
} hm_mlp_desc;

/* Builds the operand images of one layer from its folded matrix W [out_dim, seg_width0 + seg_width1]
 * (row stride ldw) and bias: w_packed (n_tiles*n_oct*256 floats), w_packed_m16 (2*n_tiles*nb*256 floats),
 * bias_padded (n_tiles*32 floats); seg_width1 = 0 when the layer has one input segment.             */
HM_API int hm_pack_mlp_layer(const float *W, int64_t ldw, const float *bias, int out_dim, int seg_width0,
                             int seg_width1, float *w_packed, float *w_packed_m16, float *bias_padded,
                             void *stream);
/* the same for every layer of a network in ONE launch (the per-step re-pack); items is a [host] array of at most
 * HM_MAX_LAYERS entries with the arguments of hm_pack_mlp_layer                                                  */
typedef struct hm_pack_item {
    const float *W;
    const float *bias;
    float *w_packed, *w_packed_m16, *bias_padded;
    int64_t ldw;
    int32_t out_dim, seg_width0, seg_width1, pad_;
} hm_pack_item;
HM_API int hm_pack_mlp_layers(const hm_pack_item *items, int n_items, void *stream);

HM_API int hm_pack_mlp_layer_bf16(const float *W, int64_t ldw, int out_dim, int seg_width0, int seg_width1,
                                  void *w_packed_bf16, void *stream);

/* x [n,3] -> out.  out_cols == 1: only the clamped sdf, out[i*out_stride];  out_cols == last
 * layer's out_dim: the whole [sdf | feature vector] row.
 * tile_points: 32 (two 4-wave workgroups per CU: throughput), 64 (one 8-wave workgroup per CU), 16 (small batches),
 * 8 or 4 (<= 2048 / 1024 points: bound by the weight stream alone), 0 = choose by n (on the device when n_dev is given),
 * -1 = like 0 but ONLY for n <= 8192 (larger batches are left to another launch, e.g. hm_sdf_fwd_bf16 with run_min 8193).
 * n_dev: optional DEVICE int32; when non-NULL the kernel evaluates min(n, *n_dev) points, so a
 *        caller that compacts work on the device needs no host synchronisation (n is the capacity).
 * max_workgroups <= 0: fill the chip once (persistent grid-stride over tiles).                    */
HM_API int hm_sdf_fwd(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *x, int64_t n,
                      const float *table, const float *B_fourier, float *out, int64_t out_stride, int out_cols,
                      int frac_mode, int tile_points, const int32_t *n_dev, int max_workgroups, void *stream);

/* bf16 variant (BASELINE configs[4]), sdf-only output out[i*out_stride]: hidden-layer weights and activations in bf16
 * on v_mfma_f32_32x32x16_bf16 (fp32 accumulate); the embedding, every product that consumes it (layer 0, the skip
 * segment), bias / Softplus, the last layer and the clamp stay fp32.  Meant for the tracer's coarse scans (sign-change
 * sampler, closest approach); runs only when the live point count is >= run_min (pair it with hm_sdf_fwd(tile_points
 * = -1) for the small counts).  Needs w_packed, bias and w_packed_bf16 in every layer.  hm_sdf_fwd_emb_bf16: the same
 * on precomputed embedding rows.  No reference behaviour exists for reduced precision: the error against
 * hm_sdf_fwd and the loss-curve criterion are measured in tests/test_bf16_gpu.py.                                      */
HM_API int hm_sdf_fwd_bf16(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *x, int64_t n,
                           const float *table, const float *B_fourier, float *out, int64_t out_stride, int frac_mode,
                           const int32_t *n_dev, int64_t run_min, void *stream);
HM_API int hm_sdf_fwd_emb_bf16(const hm_mlp_desc *mlp, const float *emb, int64_t emb_stride, int emb_width, int64_t n,
                               float *out, int64_t out_stride, const int32_t *n_dev, int64_t run_min, void *stream);

/* Split-operand variant, sdf-only output out[i*out_stride] (csrc/hm_sdf_split.hip): EVERY operand of every matrix product
 * (weights, hidden activations, embedding) is a (hi, lo) pair of 16-bit floats and W x is evaluated as
 * Wh xh + Wh xl + Wl xh on v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation - three MFMAs at 16x the fp32
 * MFMA rate.  HM_SPLIT_BF16X2: relative product error 2^-16 (the "bf16" configuration, BASELINE configs[4], with 250x the
 * accuracy of plain bf16 operands); HM_SPLIT_F16X2: <= 3 * 2^-22, below the rounding noise of an fp32 accumulation over
 * K = 512.  Bias, Softplus, the last layer and the clamp are fp32.  Runs only when the live point count is >= run_min
 * (pair it with hm_sdf_fwd(tile_points = -1) for the small counts).  Needs w_packed, bias and w_packed_split in every
 * layer and mlp->split_kind.  hm_pack_mlp_layer_split builds one layer's split image (n_tiles*nb*1024 2-byte elements)
 * from the folded matrix; seg_scale0/1 multiply the two K segments (the skip layer's embedding segment carries the
 * 1/sqrt(2) of cat[x, emb]/sqrt(2), implicit_differentiable_renderer.py:99-100).  No reference behaviour exists for
 * these modes: tests/test_split_gpu.py measures them against hm_sdf_fwd and an fp64 evaluation.                        */
HM_API int hm_pack_mlp_layer_split(const float *W, int64_t ldw, int out_dim, int seg_width0, int seg_width1,
                                   float seg_scale0, float seg_scale1, int split_kind, void *w_packed_split,
                                   void *stream);
HM_API int hm_sdf_fwd_split(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *x, int64_t n,
                            const float *table, const float *B_fourier, float *out, int64_t out_stride, int frac_mode,
                            const int32_t *n_dev, int64_t run_min, void *stream);
HM_API int hm_sdf_fwd_emb_split(const hm_mlp_desc *mlp, const float *emb, int64_t emb_stride, int emb_width, int64_t n,
                                float *out, int64_t out_stride, const int32_t *n_dev, int64_t run_min, void *stream);

/* The same network evaluated on PRECOMPUTED embedding rows emb[i*emb_stride .. + emb_width) instead of encoding x in
 * the kernel: the SDF network on top of an embedder other than the plain hash grid (FourierFilterBanks via
 * hm_nffb_fwd; custom_embedder_decoder.py:147-164 + implicit_differentiable_renderer.py:96-113).  mlp's embedding
 * segments must span ceil(emb_width/8) octets (ceil(emb_width/16) 16-blocks).                                    */
HM_API int hm_sdf_fwd_emb(const hm_mlp_desc *mlp, const float *emb, int64_t emb_stride, int emb_width, int64_t n,
                          float *out, int64_t out_stride, int out_cols, int tile_points, const int32_t *n_dev,
                          int max_workgroups, void *stream);

/* ---- Fourier-filter-bank embedders ('FFB', 'StyleModNFFB') forward, no grad ------------------------------------
 * Replaces FourierFilterBanks.forward (model/embeddings/nffb3d.py:122-194, registry settings PositionalEncodingNET /
 * SIREN / has_out=False) with PositionalEncoding (frequency_enc.py:6-51), Sine (Sine.py:5-25) and StyleAttention
 * (style_Attention/styleMod.py:16-43) in ONE kernel:
 *   out[i] = [ u(3) | mean over the L grid levels of out_layer(e_l) (W = 8 + 8L) ],  u = (x + bound) / (2 bound).
 * desc: the embedder's own hash grid (n_levels == L in {6, 8}, F = 2); table / B_fourier as for hm_encode_fwd.
 * trunk_w[0] [W,3], trunk_w[1..L-2] [W,W], trunk_b[l] [W]: ff_lin{l};  out_w [W,W], out_b [W]: out_layer;
 * style_w / style_b: StyleAttention.linear_transform ([W,W], [W]) or both NULL for the plain 'FFB' embedder;
 * w0 = L^F - L (Sine); style_eps = 1e-5.  All DEVICE pointers, row-major fp32.  hm_nffb_desc itself is [host].
 * n_dev: optional device-side point count, as for hm_sdf_fwd.                                                   */
typedef struct hm_nffb_desc {
    int32_t n_levels;
    float bound, w0, style_eps;
    const float *trunk_w[HM_MAX_LEVELS];
    const float *trunk_b[HM_MAX_LEVELS];
    const float *out_w, *out_b;
    const float *style_w, *style_b;
} hm_nffb_desc;
HM_API int hm_nffb_fwd(const hm_grid_desc *desc, const hm_nffb_desc *nf, const float *x, int64_t n, const float *table,
                       const float *B_fourier, float *out, int64_t out_stride, int frac_mode, const int32_t *n_dev,
                       void *stream);

/* ---- ray / surface intersection search ------------------------------------------------------
 * Replaces RayTracing.forward and its helpers sphere_tracing / ray_sampler / secant /
 * minimal_sdf_points (model/ray_tracing.py:26-298) when the `sdf` callable is the network above:
 * bidirectional sphere tracing with line-search back-off, the n_steps sign-change sampler with
 * n_secant_steps secant iterations for rays that did not converge, and (training) the
 * closest-approach search for the mask-loss rays.  Same constructor values as the reference
 * class (ray_tracing.py:6-24); `training` = self.training.  One call enqueues the whole search
 * on `stream` without any host synchronisation (per-ray state machines + device-side compaction).
 *
 * cam_loc [B,3]; ray_dirs [N,3] (N = B*rays_per_image); object_mask [N] bytes;
 * t_sphere [N,2] / hit_mask [N]: near/far parameters and hit flags of the bounding sphere
 *   (utils/rend_util.py:141-162, computed by the caller);
 * sampler_fracs [n_steps] = linspace(0,1,n_steps) (ray_tracing.py:198);
 * steps_u [n_steps]: the U(0,1) fractions shared by all rays (ray_tracing.py:277; training only);
 * outputs: points [N,3], network_object_mask [N] bytes, dists [N];
 * stats_out (optional, 16 device int32): sampler rays, sampler points, secant rays, mask-loss rays,
 *   their points, any-iteration flag, total SDF evaluations, unfinished rays (must be 0), non-finite SDF
 *   values met by the search (must be 0), points of the big launch, lazy sampler: rays of the second pass, points of
 *   the first pass, points of the second pass, 3 reserved.                                          */
typedef struct hm_trace_cfg {
    float object_bounding_sphere;
    float sdf_threshold;
    double line_search_step;       /* back-off factors (1-step)/2^k are formed in double, like the reference */
    int32_t line_step_iters;       /* <= 3  */
    int32_t sphere_tracing_iters;  /* <= 15 */
    int32_t n_steps;
    int32_t n_secant_steps;
    int32_t training;
    int32_t coarse_bf16;           /* precision of the sampler / closest-approach scans: 0 exact fp32, 1 hm_sdf_fwd_bf16 (needs
                                      w_packed_bf16), 2 hm_sdf_fwd_split (needs w_packed_split; kind = mlp->split_kind);
                                      sphere tracing and the secant refinement stay on the exact-fp32 kernels */
    int32_t sampler_head;          /* lazy sampler.  ray_sampler (ray_tracing.py:189-249) evaluates n_steps samples per
                                      unconverged ray but reads only those up to the FIRST negative one (:212-218, and
                                      its predecessor for the secant bracket, :238-243) unless the ray falls back to the
                                      minimal sample (:221-226).  h = sampler_head in [1, n_steps-2]: a first pass
                                      evaluates samples 0..h-1 and n_steps-1; rays with a negative head sample inside
                                      the object mask are finished, the others get samples h..n_steps-2 in a second
                                      pass.  Outputs are bit-identical to the single pass (0 / out of range).       */
} hm_trace_cfg;

HM_API int64_t hm_trace_workspace_bytes(int64_t n_rays, const hm_trace_cfg *cfg);
HM_API int hm_trace_forward(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *table,
                            const float *B_fourier, int frac_mode, int tile_points, const hm_trace_cfg *cfg,
                            const float *cam_loc, const float *ray_dirs, const uint8_t *object_mask,
                            const float *t_sphere, const uint8_t *hit_mask, int64_t n_rays, int64_t rays_per_image,
                            const float *sampler_fracs, const float *steps_u, float *out_points,
                            uint8_t *out_net_mask, float *out_dists, void *workspace, int64_t workspace_bytes,
                            int32_t *stats_out, void *stream);

/* The same search with the SDF network on a Fourier-filter-bank embedder (hm_nffb_fwd + hm_sdf_fwd_emb per round);
 * desc / table / B_fourier are the embedder's own hash grid; workspace of hm_trace_workspace_bytes_nffb() bytes.     */
HM_API int64_t hm_trace_workspace_bytes_nffb(int64_t n_rays, const hm_trace_cfg *cfg, int n_levels);
HM_API int hm_trace_forward_nffb(const hm_grid_desc *desc, const hm_nffb_desc *nffb, const hm_mlp_desc *mlp,
                                 const float *table, const float *B_fourier, int frac_mode, int tile_points,
                                 const hm_trace_cfg *cfg, const float *cam_loc, const float *ray_dirs,
                                 const uint8_t *object_mask, const float *t_sphere, const uint8_t *hit_mask,
                                 int64_t n_rays, int64_t rays_per_image, const float *sampler_fracs,
                                 const float *steps_u, float *out_points, uint8_t *out_net_mask, float *out_dists,
                                 void *workspace, int64_t workspace_bytes, int32_t *stats_out, void *stream);

/* ---- exact-fp32 GEMM (grad-enabled MLP path) ------------------------------------------------
 * Replaces the nn.Linear matmuls autograd runs for the SDF and rendering MLPs when gradients are
 * needed (implicit_differentiable_renderer.py:102,116-128,211-221): forward X*W^T + b, backward
 * dY*W and dY^T*X, and the same shapes again under create_graph=True.
 *   C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N]);  row-major, leading dimensions in floats;
 *   transA: A is stored [K,M];  transB: B is stored [N,K];  bias may be NULL;
 *   accumulate != 0 adds into C (fp32 atomics), otherwise C is overwritten.                      */
HM_API int hm_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda,
                       const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int accumulate,
                       void *stream);

/* GEMM with a fused elementwise epilogue on v = (acc + bias) * scale - the Softplus passes that follow (forward)
 * or consume (backward / double backward) every nn.Linear of the SDF MLP, done on the accumulator registers
 * instead of as separate passes over [points, 512] tensors.  s1 / s2 = first / second derivative of
 * nn.Softplus(beta, threshold) at z.  C may be NULL (raw product not stored).  No split-K, no accumulate.
 *   HM_EPI_NONE      C = v
 *   HM_EPI_SOFTPLUS  C = v;  out1 = softplus(v)                                   (forward: z and h in one pass)
 *   HM_EPI_S1MUL     C = v;  out1[:, :nz] = v[:, :nz] * s1(z) (+ g)               (gradient sweeps: u = v * s1(z))
 *   HM_EPI_ADJOINT   out1 = v * s1(z);  out2 = v * g * s2(z);  out3 = g * s1(z)    (adjoint of u = g * s1(z); out3 optional)
 *   HM_EPI_RELU      C = v;  out1 = max(v, 0)                                     (rendering MLP forward, :214-219)
 *   HM_EPI_RELUMASK  C = v;  out1[:, :nz] = z > 0 ? v : 0                        (its backward; z = the layer's ReLU output) */
enum { HM_EPI_NONE = 0, HM_EPI_SOFTPLUS = 1, HM_EPI_S1MUL = 2, HM_EPI_ADJOINT = 3, HM_EPI_RELU = 4, HM_EPI_RELUMASK = 5 };
typedef struct hm_gemm_epilogue {
    int32_t mode;
    int32_t nz;               /* S1MUL: columns [0, nz) get the s1 product (z has nz columns)          */
    float scale, beta, threshold;
    const float *z;  int64_t ldz;
    const float *g;  int64_t ldg;  /* ADJOINT: second factor;  S1MUL: optional addend (may be NULL)    */
    float *out1;     int64_t ld1;
    float *out2;     int64_t ld2;
    float *out3;     int64_t ld3;
} hm_gemm_epilogue;
HM_API int hm_gemm_f32_ep(int transA, int transB, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda,
                          const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc,
                          const hm_gemm_epilogue *ep, void *stream);

/* ---- fused activation passes of the grad-enabled MLP path ------------------------------------
 * nn.Softplus(beta, threshold) (implicit_differentiable_renderer.py:84) over n contiguous floats:
 *   order 0: out0 = softplus(z)
 *   order 1: out0 = gy * s1(z)                         (its backward; s1 = d softplus / dz)
 *   order 2: out0 = gg * s1(z),  out1 = gg * gy * s2(z)  (backward of order 1 w.r.t. gy and z; the
 *            double backward autograd needs for ImplicitNetwork.gradient(create_graph=True), :116-128)
 * All pointers 16-byte aligned.                                                                   */
HM_API int hm_softplus(int order, const float *z, const float *gy, const float *gg, float *out0, float *out1,
                       int64_t n, float beta, float threshold, void *stream);

/* The same three passes for the two elementwise pieces of the Fourier-filter-bank embedders (BASELINE configs 3 / 5):
 *   hm_sine    SIREN activation y = sin(w0 x) (embeddings/Sine.py:10-12) over n floats:
 *                order 0: out0 = sin(w0 x);  order 1: out0 = gy w0 cos(w0 x);
 *                order 2: out0 = gg w0 cos(w0 x) (d/d gy),  out1 = -gg gy w0^2 sin(w0 x) (d/d x)
 *   hm_posenc  NeRF positional encoding of rows c [n, dim] (frequency_enc.py:6-51 with include_input, sin / cos,
 *              `freqs` = HOST array of n_freq bands): row = [c | c | sin(f0 c) | cos(f0 c) | sin(f1 c) | ...],
 *              width W = 2 dim + 2 n_freq dim:
 *                order 0: out0 [n, W] = row;   order 1: out0 [n, dim] = (d row / d c)^T g  (g [n, W]);
 *                order 2 (given gg [n, dim] contiguous): out0 [n, W] = d(gg . order1)/d g,  out1 [n, dim] = d(gg . order1)/d c */
/*   hm_weight_norm_multi  W = g * v / ||v||_row (nn.utils.weight_norm with dim = 0, as every Linear of the SDF and
 *              rendering networks is wrapped, implicit_differentiable_renderer.py:95-97,205-206) for SEVERAL layers in
 *              one launch (backward == 0: w and norm are written) and its backward (backward != 0: grad_v, grad_g from
 *              grad_w and the stored norm); arithmetic of ATen's weight_norm first-dim kernels.  `layers` is a HOST array. */
typedef struct hm_wn_layer {
    const float *v;        /* [rows, cols] */
    const float *g;        /* [rows]       */
    float *w;              /* [rows, cols] forward output                  */
    float *norm;           /* [rows]       forward output / backward input */
    const float *grad_w;   /* [rows, cols] backward input                  */
    float *grad_v;         /* [rows, cols] backward output                 */
    float *grad_g;         /* [rows]       backward output                 */
    int32_t rows, cols;
} hm_wn_layer;
HM_API int hm_weight_norm_multi(int backward, int n_layers, const hm_wn_layer *layers, void *stream);

/*   hm_rownorm  (y - mean) / sqrt(var + eps) per row of a contiguous [rows, width] tensor, biased variance - the
 *              nn.InstanceNorm1d call of StyleAttention on a 2-D tensor (style_Attention/styleMod.py:41-43):
 *                order 0: out0 = normalised rows;  order 1: out0 = its backward applied to g;
 *                order 2 (given gg): out0 = d(gg . order1)/d g,  out1 = d(gg . order1)/d y                              */
HM_API int hm_rownorm(int order, const float *y, const float *g, const float *gg, float *out0, float *out1, int64_t rows,
                      int width, float eps, void *stream);
HM_API int hm_sine(int order, const float *x, const float *gy, const float *gg, float *out0, float *out1, int64_t n,
                   float w0, void *stream);
HM_API int hm_posenc(int order, const float *freqs, int n_freq, int dim, const float *c, int64_t c_stride,
                     const float *g, int64_t g_stride, const float *gg, float *out0, int64_t out0_stride, float *out1,
                     int64_t n, void *stream);

/* Soft clamp of the SDF column of the last layer's output zL [n, cols] (contiguous) and its backward
 * (implicit_differentiable_renderer.py:112, density_net.py:20-30; the density is evaluated without gradient):
 *   backward == 0:  out = zL with column 0 -> sdf = tanh(s / (2 + rho(s)));  sdf, denom = 2 + rho, c = d sdf/d s [n]
 *   backward != 0:  out = d_out with column 0 -> d_out[:,0] * c (+ cb * (-2 sdf c / denom) when cb != NULL);
 *                   sdf, c, denom are inputs here.                                                       */
HM_API int hm_sdf_head(int backward, const float *in, int64_t n, int64_t cols, float beta_rho, float *out, float *sdf,
                       float *c, float *denom, const float *cb, void *stream);

/* out[n] = sum over rows of x[M,N] (row stride ld) - the bias gradient of an nn.Linear.          */
HM_API int hm_colsum(const float *x, int64_t M, int64_t N, int64_t ld, float *out, void *stream);
/* same, added into out (no zeroing: the caller zeroes all its gradient buffers with one launch)  */
HM_API int hm_colsum_acc(const float *x, int64_t M, int64_t N, int64_t ld, float *out, void *stream);
/* out_i[n] += sum_m x_i[m, n] for a list of matrices in ONE launch (all bias gradients of one backward pass; the caller
 * zeroes the outputs).  items is a [host] array.                                                                 */
#define HM_COLSUM_MAX_ITEMS 16
typedef struct hm_colsum_item {
    const float *x;
    float *out;
    int64_t M, N, ld;
} hm_colsum_item;
HM_API int hm_colsum_acc_multi(const hm_colsum_item *items, int n_items, void *stream);

/* dst[r*ld_dst + c] = src[r*ld_src + c], r < rows, c < cols, as an ordinary kernel: the copies torch would issue as
 * hipMemcpyAsync (Tensor.copy_ / clone of contiguous tensors; the saved-activation stacking of the fused MLP-gradient
 * node, torch.cat in implicit_differentiable_renderer.py:99-100,280-284) become MEMCPY nodes of a captured HIP graph,
 * which this library keeps out of its captured training iteration (graph replay fault).            */
HM_API int hm_copy2d_f32(float *dst, int64_t ld_dst, const float *src, int64_t ld_src, int64_t rows, int64_t cols,
                         void *stream);

/* ---- IDRLoss: value and gradients in one launch (code/model/loss.py:4-70) --------------------------
 * terms[4] = {loss, rgb_loss, eikonal_loss, mask_loss};  d_rgb[n,3], d_sdf[n], d_grad[m,3] = d loss / d input.
 * rgb, rgb_gt [n,3]; sdf [n]; hit (network_object_mask), inside (object_mask) [n] as bytes; grad_theta [m,3].  */
HM_API int hm_idr_loss(const float *rgb, const float *rgb_gt, const float *sdf, const uint8_t *hit,
                       const uint8_t *inside, int64_t n_rays, const float *grad_theta, int64_t n_grad,
                       float eikonal_weight, float mask_weight, float alpha, float *terms, float *d_rgb, float *d_sdf,
                       float *d_grad, void *stream);

/* ---- camera rays + bounding-sphere intersections --------------------------------------------------
 * Replaces rend_util.get_camera_params for 4x4 poses (utils/rend_util.py:48-75: lift, pose x pixel, normalise) and
 * rend_util.get_sphere_intersection (:141-162) for the rays it produces, as one launch: uv [B,N,2], pose / intrinsics
 * [B,4,4] -> ray_dirs [B,N,3], cam_loc [B,3], t_sphere [B,N,2] (clamped at 0), hit [B,N] (bytes).  Fixed cameras only:
 * nothing here is differentiated.                                                                        */
HM_API int hm_camera_rays(const float *uv, const float *pose, const float *intrinsics, int64_t n_images, int64_t n_pixels,
                          float radius, float *ray_dirs, float *cam_loc, float *t_sphere, uint8_t *hit, void *stream);

/* ---- optimizer tail of the training iteration ---------------------------------------------------
 * torch.nn.utils.clip_grad_norm_(parameters, max_norm) followed by torch.optim.Adam.step()
 * (training/idr_train.py:128,306-309; amsgrad / weight decay off as in the reference) over all tensors in
 * three launches.  In place: grad *= clip coefficient, exp_avg, exp_avg_sq, param.  max_norm <= 0 skips the
 * clipping.  Every tensor's step counter is incremented by the call (the bias corrections use the incremented
 * value); tensors left out of a call (no gradient) keep theirs, as in torch.  scratch_dev: hm_adam_scratch_floats()
 * device floats (2 + one partial sum per 8192 gradient elements); scratch_dev[1] receives the total gradient norm
 * (before clipping) when max_norm > 0.  The norm is summed in a fixed order (no float atomics): the whole update is
 * bitwise reproducible, so data-parallel replicas fed the same all-reduced gradients stay bitwise identical.
 * Sync-free and graph-capturable.                                                                     */
#define HM_ADAM_MAX_TENSORS 64
typedef struct hm_adam_tensor {
    float *param, *grad, *exp_avg, *exp_avg_sq;  /* device, fp32, contiguous, same numel */
    int64_t *step;                               /* device iteration counter of THIS tensor (torch: state['step']) */
    int64_t numel;
} hm_adam_tensor;
HM_API int64_t hm_adam_scratch_floats(const hm_adam_tensor *tensors, int n_tensors);
HM_API int hm_adam_step(const hm_adam_tensor *tensors, int n_tensors, float lr, float beta1, float beta2, float eps,
                        float max_norm, float *scratch_dev, void *stream);

/* Weight gradients of one backward pass in ONE launch: C_p += A_p^T B_p for every item (A_p [K, M] and B_p [K, N]
 * row-major with leading dimensions, C_p [M, N] accumulated with fp32 atomics - the caller zeroes it, like the
 * reference's optimizer.zero_grad()).  Replaces the per-layer grad_weight GEMMs autograd runs for nn.Linear in
 * ImplicitNetwork / RenderingNetwork (implicit_differentiable_renderer.py:102,211-221).  K a multiple of 128 takes the
 * grouped kernel, any other K falls back to hm_gemm_f32 for that item.  items is a [host] array.                     */
#define HM_GEMM_GROUP_MAX 16
typedef struct hm_gemm_group_item {
    const float *A, *B;
    float *C;
    int64_t M, N, K, lda, ldb, ldc;
} hm_gemm_group_item;
HM_API int hm_gemm_f32_group_tn(const hm_gemm_group_item *items, int n_items, void *stream);

/* ---- data-parallel gradient exchange (device side) ---------------------------------------------------
 * The reference runner is single-GPU (training/idr_train.py:92-93,278-321): there is no interface to
 * replace, the exchange is the build's own addition in the place the north star names - between loss.backward()
 * (idr_train.py:302) and clip_grad_norm_ / optimizer.step() (:306-308).  Everything except the collectives themselves
 * (RCCL, issued by the host side) is a kernel below, so that it can be captured into the training step's graphs.
 *
 * Hash-table gradient.  hm_encode_bwd_table_tracked is hm_encode_bwd_table (same atomics into d_table) that also lists
 * each row it touches ONCE: touched_bits [ceil(total_rows/32)] one claim bit per row (all zero between steps),
 * touched_count [1] the number of listed rows, touched_rows [cap] their ids in the fused table.  Several calls of one
 * step share the three buffers.  hm_rows_pack turns the list into this rank's payload [cap, 1+F] int32 (row id, then
 * the F gradient values as fp32 bits; entries beyond the count carry row -1), zeroing the listed rows of d_table and
 * their claim bits; status[0] (optional) receives the overflow if more than `cap` rows were claimed.  After ONE
 * all-gather of the payloads hm_rows_apply adds list r = lists + r*list_stride (r = 0..n_lists-1, one launch each, in
 * that order) into d_table scaled by `scale` (1/world): rows are unique inside a list, so plain read-modify-writes
 * suffice and every replica sums every row in the same (rank) order - bitwise identical replicas, no sort.
 * hm_rows_clear (after the optimizer step) zeroes the rows named by all lists again and resets *touched_count.      */
HM_API int hm_encode_bwd_table_tracked(const hm_grid_desc *desc, const float *x, int64_t n, const float *d_feat,
                                       int64_t d_feat_stride, float *d_table, int frac_mode, uint32_t *touched_bits,
                                       int32_t *touched_count, int32_t *touched_rows, int64_t cap, void *stream);
HM_API int hm_rows_pack(float *d_table, int n_features, const int32_t *touched_rows, const int32_t *touched_count,
                        int64_t cap, uint32_t *touched_bits, int32_t *payload, int32_t *status, void *stream);
HM_API int hm_rows_apply(float *d_table, int64_t total_rows, int n_features, const int32_t *lists, int64_t cap,
                         int64_t list_stride, int n_lists, float scale, void *stream);
HM_API int hm_rows_clear(float *d_table, int64_t total_rows, int n_features, const int32_t *lists, int64_t cap,
                         int64_t list_stride, int n_lists, int32_t *touched_count, void *stream);

/* MLP gradients -> one flat bucket in ONE launch (the bucket is all-reduced in place; hm_adam_step then reads the
 * averaged gradients from it).  items is a [host] array, copied by value into the kernel arguments.              */
#define HM_COPY_MAX_ITEMS 96
typedef struct hm_copy_item {
    const float *src;
    float *dst;
    int64_t numel;
} hm_copy_item;
HM_API int hm_multi_copy_f32(const hm_copy_item *items, int n_items, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HASHMOD_H */
