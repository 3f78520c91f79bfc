/*
 * gkg_hip.h — C ABI of libgkg_hip.so: the MI355X (gfx950) implementation of GKGNet's
 * Group-KNN graph-convolution hot path.
 *
 * The reference (jin-s13/GKGNet) has no native boundary: the path is a chain of ATen ops inside
 * DyGraphConv2d*.forward.  The operator boundary this library introduces sits exactly at the two
 * calls made there (paths relative to the reference tree):
 *
 *   self.dilated_knn_graph(x, y, relative_pos)        mmcls/models/backbones/vig_model/torch_vertex.py:203,226,247,272
 *        -> DenseDilatedKnnGraph.forward              .../vig_model/torch_edge.py:164-176   ==> gkg_knn_fwd
 *   GraphConv2d.forward(x, edge_index, y) / MRConv2d  .../vig_model/torch_vertex.py:47-54   ==> gkg_mr_fwd
 *        (batched_index_select x2 + max(x_j - x_i))   .../vig_model/torch_nn.py:84-105
 *   autograd of the above (max_backward, index_put_)  (ATen; SURVEY.md §8a "Backward contract")  ==> gkg_mr_bwd
 *
 * Conventions
 *   - All pointers are DEVICE pointers owned by the caller; tensors are contiguous, channel-major
 *     exactly like the reference's (B*G, c, N, 1) layout:   x[bg][ch][n].
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*); the library never
 *     synchronises, allocates or frees, and keeps no mutable global state besides the last-error
 *     string (thread-local).  Workspace, when needed, is passed in by the caller.
 *   - Return value: 0 on success; >0 a hipError_t from a launch; <0 an argument error
 *     (GKG_ERR_*).  Nothing throws or exits.
 *   - dtype: element type of the feature tensors x / y / src / g / outputs (GKG_F32, GKG_BF16, GKG_F16).  Distances are
 *     always accumulated in fp32 (SURVEY.md §5 AMP row).
 *
 * Arithmetic contract of gkg_knn_fwd (restated bit-for-bit by oracle/gkg_oracle.c):
 *     s      = fma-chain_{ch} t[ch]^2            den = max(sqrt(s), 1e-12)       (GKG_KNN_NORMALIZE)
 *     th[ch] = t[ch] / den                       sq  = fma-chain_{ch} th[ch]^2
 *     dot    = fma-chain_{ch=0..c-1} yh[ch][m] * xh[ch][n]      (fp32 MFMA == ordered fmaf chain)
 *     dist   = ((sqx[n] + (-2*dot)) + sqy[m]) + relpos[n][m]    (reference op order, torch_edge.py:17-20,82)
 *     neighbours = the k*dilation smallest (dist, m) pairs in ascending lexicographic order
 *                  (tie rule: smaller key index first); ranks 0, d, 2d, ... are emitted.
 */
#ifndef GKG_HIP_H_
#define GKG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GKG_ABI_VERSION 9

/* dtype codes */
#define GKG_F32 0
#define GKG_BF16 1
#define GKG_F16 2 /* the reference trains under fp16 AMP (configs/gkgnet/gkgnet_coco_576.py:146): inputs widened exactly */

/* gkg_knn_fwd flags */
#define GKG_KNN_NORMALIZE 1u /* L2-normalise tokens over the group's channels first (torch_edge.py:167-173) */
#define GKG_KNN_BF16_CONTRACT 2u /* x.y^T on the bf16 matrix cores: tokens rounded to bf16 after normalisation, exact
                                  * products, fp32 accumulation and fp32 norms.  For callers under bf16 autocast, where the
                                  * reference computes this product in bf16 and rounds it to bf16.  Outside the bit-exact
                                  * index contract; ignored for c < 9. */
#define GKG_KNN_SELECT_DIRECT 4u   /* force the direct sorted insert / the buffered selection of the tile kernel instead of */
#define GKG_KNN_SELECT_BUFFERED 8u /* the library's per-shape rule (measurement, tests): identical results either way */
#define GKG_KNN_NO_PREFILTER 16u   /* evaluate every distance with the contract's fp32 chain (knn_tile_kernel) instead of the
                                    * bf16 matrix-core prefilter + exact re-rank of the survivors (knn_pf_kernel): identical
                                    * results either way (measurement, tests) */
#define GKG_KNN_FORCE_PREFILTER 32u /* take the prefilter kernel wherever it is applicable (normalised tokens, un-split keys,
                                    * k*dilation <= 36, c >= 16), not only where the library's rule says it pays (tests) */
#define GKG_KNN_RELPOS_UNIT 64u     /* the caller guarantees |relative_pos| <= 1.125 everywhere (GKGNet's bias -2 PE PE^T / D lies
                                    * in [-1, 0], pos_embed.py:21-29; its bicubic resize for pooled keys, torch_vertex.py:311-315,
                                    * can overshoot by a fraction of a percent): precondition of the prefilter kernel's error
                                    * bound when a bias is given.  Without it a call with a bias always takes knn_tile_kernel (same results,
                                    * no range assumption) */

#define GKG_KNN_X_PREPARED 128u     /* token-major fp32 callers: the queries' normalised copies / norms (/ prefilter planes) are already
                                    * in `workspace`, left there by gkg_bn_apply_knn_prep called with the same problem — the call
                                    * launches no preparation for them (none at all for a self graph) */
#define GKG_KNN_Y_PREPARED 256u     /* likewise for the KEYS (gkg_bn_apply_knn_prep with as_keys: a label graph's keys, produced by the
                                    * Grapher in front of it) */

/* argument errors */
#define GKG_ERR_NULL -1        /* required pointer is NULL */
#define GKG_ERR_SHAPE -2       /* non-positive / inconsistent sizes, k*dilation > M, ... */
#define GKG_ERR_UNSUPPORTED -3 /* dtype / size outside what this build supports */
#define GKG_ERR_WORKSPACE -4   /* workspace too small */

/* ABI version of the loaded library (== GKG_ABI_VERSION it was built with). */
int gkg_version(void);
/* Identifier of the hipGraph capture `stream` is recording into; 0 when it is not capturing. */
unsigned long long gkg_stream_capture_id(void* stream);

/* Human-readable description of the last non-zero return on this thread ("" if none). */
const char* gkg_last_error_string(void);

/* Bytes of scratch gkg_knn_fwd needs for these sizes (normalised token copies, squared norms,
 * split-key partial lists).  Pure function of its arguments. */
size_t gkg_knn_workspace_bytes(int BG, int c, int N, int M, int k, int dilation, int dtype, unsigned flags);

/*
 * Dilated k-NN graph.  Replaces DenseDilatedKnnGraph.forward (torch_edge.py:164-176) including
 * dense_knn_matrix / xy_dense_knn_matrix (:54-106), pairwise distance (:9-51) and DenseDilated (:139-149).
 *   x        (BG, c, N)      query tokens
 *   y        (BG, c, M)      key tokens, or NULL for the self graph (then M must equal N)
 *   relpos   (N, M) fp32     positional bias shared by all BG problems, or NULL
 *   nn_idx   (BG, N, k) i64  out: neighbour index per query, ascending distance  (== edge_index[0])
 *   center   (BG, N, k) i64  out, optional (NULL to skip): centre index n         (== edge_index[1])
 */
int gkg_knn_fwd(const void* x, const void* y, const float* relpos, int64_t* nn_idx, int64_t* center,
                int BG, int c, int N, int M, int k, int dilation, int dtype, unsigned flags,
                void* workspace, size_t workspace_bytes, void* stream);

/*
 * Max-relative aggregation: m[bg][ch][n] = max_j ( src[bg][ch][nn_idx[bg][n][j]] - x[bg][ch][n] ).
 * Replaces batched_index_select x2 + torch.max(x_j - x_i, -1) (torch_vertex.py:49-54, torch_nn.py:84-105).
 *   src      (BG, c, M) or NULL (-> src = x, M = N)
 *   m_out    (BG, c, N)
 *   argmax   (BG, c, N) u8   out, optional: first j attaining the max (needed by gkg_mr_bwd)
 */
int gkg_mr_fwd(const void* x, const void* src, const int64_t* nn_idx, void* m_out, uint8_t* argmax,
               int BG, int c, int N, int M, int k, int dtype, void* stream);

/*
 * Backward of gkg_mr_fwd for upstream gradient g (BG,c,N) on m:
 *     gx[bg][ch][n]                        = -g[bg][ch][n]
 *     gsrc[bg][ch][nn_idx[bg][n][argmax]] +=  g[bg][ch][n]
 *   gsrc == NULL  -> self graph: both terms are accumulated into gx (M must equal N).
 *   gx, gsrc are fully overwritten (no pre-zeroing needed).
 */
int gkg_mr_bwd(const void* g, const int64_t* nn_idx, const uint8_t* argmax, void* gx, void* gsrc,
               int BG, int c, int N, int M, int k, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * EdgeConv aggregation (Grapher(conv='edge'); reference torch_vertex.py:82-101 EdgeConv2d.forward + torch_nn.py:57-69
 * BasicConv).  The reference evaluates  max_k act(norm(Conv2d_1x1,groups=4(cat[x_i, x_j - x_i])))  on a (B, 2C, N, k)
 * tensor.  With the hard-coded groups = 4 the output channels of groups 2 and 3 depend on (x_j - x_i) only, and the
 * convolution is linear, so for them z[b][o][n][k] = Q[b][o][nn_idx[b][n][k]] - Qc[b][o][n] + bias[o] with the per-node
 * projections Q = W src, Qc = W x (ordinary GEMMs in the caller: k times less work, no (B, 2C, N, k) tensor).  These four
 * entry points are the gather side; channel-major fp32 (B, O, N) / (B, O, M), nn_idx (B, N, k), k <= 255.
 *   gkg_edge_stats      sums[o] += sum (Q[j] - Qc), sums[O + o] += sum (Q[j] - Qc)^2 over all (b, n, k): the batch statistics
 *                       of train-mode BN (the bias shifts the mean only).  sums: 2*O doubles, zero on entry.
 *   gkg_edge_fwd        out = max_k act(a[o] (Q[j] - Qc) + c[o]); argmax (optional) = first maximising k.  The caller folds
 *                       norm and bias into a, c.  act: 0 none, 1 GELU (erf), 2 ReLU.
 *   gkg_edge_bwd_stats  sums[o] += sum_n g', sums[O + o] += sum_n g' zhat over the argmax elements, g' = g act'(.), zhat =
 *                       (z - mean0) invstd: dbeta, dgamma and (divided by B N k) the two means of the BN backward.
 *   gkg_edge_bwd        dz[n][k] = a (g' [k == argmax] - mg - zhat[n][k] mgz) for every edge; dqs[nn_idx] += dz (atomics; dqs
 *                       zero on entry), dqc[n] = -sum_k dz.  mg == NULL: statistics are constants, only the winning edge
 *                       carries gradient. */
int gkg_edge_stats(const float* qs, const float* qc, const int64_t* nn_idx, double* sums, int B, int O, int N, int M, int k,
                   void* stream);
int gkg_edge_fwd(const float* qs, const float* qc, const int64_t* nn_idx, const float* a, const float* c, float* out,
                 uint8_t* argmax, int B, int O, int N, int M, int k, int act, void* stream);
int gkg_edge_bwd_stats(const float* g, const float* qs, const float* qc, const int64_t* nn_idx, const uint8_t* argmax,
                       const float* a, const float* c, const float* mean0, const float* invstd, double* sums, int B, int O,
                       int N, int M, int k, int act, void* stream);
int gkg_edge_bwd(const float* g, const float* qs, const float* qc, const int64_t* nn_idx, const uint8_t* argmax,
                 const float* a, const float* c, const float* mean0, const float* invstd, const float* mg, const float* mgz,
                 float* dqs, float* dqc, int B, int O, int N, int M, int k, int act, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Token-major variants used inside the fused Grapher block (activations (B, N, C), C = G*c, fp32).
 * Same arithmetic contracts as above; only the addressing differs.
 *   gkg_knn_fwd_tm : x (B,N,C), y (B,M,C) or NULL; nn_idx/center (B*G, N, k)
 *   gkg_mr_fwd_tm  : mode 0 -> out = m (B,N,C);
 *                    mode 1 -> out = XM (B*N, 2C): the grouped projection's operand buffer (see "XM layout" below);
 *                              needs C % 16 == 0
 *                    argmax (B,N,C) u8
 *   gkg_mr_bwd_tm  : mode 0: gin = g (B,N,C); mode 1: gin = dXM (B*N, 2C) in the XM layout (x chunks = gradient reaching x
 *                    directly, m chunks = gradient of m).  gx (B,N,C) and gsrc (B,M,C)|NULL fully overwritten.
 *                    Default for destination images of up to 512 rows (where it measured faster; with
 *                    GKG_MR_DETERMINISTIC wherever 8 channels of 64-bit accumulators per row fit the LDS): EXACT fixed-point
 *                    accumulation — every gradient becomes sign * (24-bit mantissa << shift) relative to the chunk's
 *                    largest magnitude, the fan-in is summed with 64-bit integer LDS atomics and rounded ONCE to fp32:
 *                    independent of the arrival order (bit-identical from run to run) and at least as accurate as an
 *                    fp32 sum; non-finite gradients fall back to fp32 atomics per chunk (inf / NaN propagate).
 *                    From 160 query rows per image the same accumulation runs in ONE sweep (round 5): the scale comes from
 *                    a strided sample of the chunk's rows plus 6 binary orders of headroom (the exact maximum when a thread's
 *                    first rows are the whole sweep); a gradient beyond the headroom makes that workgroup redo its chunk with
 *                    the exact two-sweep form.  Still order-independent and bit-identical from run to run.
 *                    flags & GKG_MR_FP32_ATOMICS: the round-1 fp32 LDS-atomic kernels (fan-in summed in arrival order);
 *                    flags & GKG_MR_DETERMINISTIC (shapes beyond the LDS budget): private accumulators, fixed order.
 *
 * XM layout (round 6) — the reference's MRConv2d interleaves [x_0, m_0, x_1, m_1, ...] (torch_vertex.py:57-61) and feeds
 * Conv2d(2C, 2C, 1, groups=4) (torch_nn.py:61): conv group q reads interleaved channels [q C/2, (q+1) C/2) = the x and m values
 * of original channels [q h, (q+1) h), h = C/4.  Nothing is interleaved here.  The grouped projection's operand is ONE buffer
 *     XM (T, 2C) fp32,   row t = [x_0 | m_0 | x_1 | m_1 | x_2 | m_2 | x_3 | m_3],   x_q = x[t][q h .. (q+1) h), m_q likewise,
 * i.e. group q's 2h inputs are contiguous with the x half first; the weight's input columns are reordered to match when its
 * bf16 planes are built (gkg_x6_prep_desc_fill kperm) and the weight gradient is permuted back where it is added to dW
 * (GkgWgradProblem.kperm) — the weight tensor keeps the reference's layout.  The PRODUCER of x (the Grapher's fc1 BN-apply:
 * gkg_bn_apply_train / gkg_affine_act with ochunk = h and ldo = 2C) writes x straight into the x chunks, the aggregation
 * (gkg_knn_mr_fwd_tm / gkg_mr_fwd_tm mode 1) fills the m chunks and, in the backward, the grouped input-gradient GEMM writes
 * dXM in the same layout (gkg_linear_dgrad_x6_sk ldx = 2C, x_bstride = 2h) for gkg_mr_bwd_tm mode 1: x is never copied.
 * Token-major fp32 inputs are therefore passed as VIEWS (pointer, row pitch ld, chunk): channel ch of token t sits at
 * p[t * ld + ch + (ch / chunk) * chunk] for chunk > 0 (chunk % 4 == 0, chunk | C, ld >= 2C: the x half of an XM buffer is
 * (XM, 2C, h)), at p[t * ld + ch] for chunk == 0; ld == 0 means ld = C.  A self graph gathers its neighbours through the same view.
 * When x passed to a mode-1 aggregation IS the output buffer's x half (x == out, ldx == 2C, xchunk == h) only m is written;
 * otherwise both halves are (a caller whose x lives elsewhere).
 */
int gkg_knn_fwd_tm(const void* x, int ldx, int xchunk, const void* y, const float* relpos, int64_t* nn_idx, int64_t* center,
                   int B, int G, int c, int N, int M, int k, int dilation, int dtype, unsigned flags,
                   void* workspace, size_t workspace_bytes, void* stream);
/* The Grapher's fc1 BN-apply and the k-NN's token preparation in ONE pass (round 6; reference torch_vertex.py:326 fc1's BN ->
 * torch_edge.py:167-173 F.normalize + |x|^2).  y (B N, C = G c) is fc1's pre-BN projection output whose train-mode batch
 * statistics lie in `sums` (gkg_linear_bn_fwd_x6 with train == 2; everything about sums / a / c / mean / invstd / running
 * statistics / zero_buf is gkg_bn_apply_train's contract with nb == 1, act == 0, no residual).  x = a y + c goes to `out` (row
 * pitch ldo floats, 0 = C; ochunk > 0: the x half of the grouped projection's operand buffer, "XM layout") and, from the same
 * registers, normalised into `knn_workspace`: the workspace of the k-NN call (gkg_knn_fwd_tm / gkg_knn_fwd_tm16 /
 * gkg_knn_mr_fwd_tm — fused_mr != 0 for the latter) with the SAME B, G, c, N, M, k, dilation, presence of y / relative_pos and
 * flags, which follows with GKG_KNN_X_PREPARED set and x = out.  Same operations in the same order per value as gkg_bn_apply_train
 * followed by that call's own preparation: bit-identical x, graphs and aggregation.
 * as_keys != 0 — the KEYS of the problem instead (a GrapherLabel's keys are the feature map a Grapher in front of it returns,
 * torch_vertex.py:331 -> :392-403): y (B M, C) is that Grapher's fc2 pre-BN output, res_tm (B M, C) or NULL its token-major
 * residual; a y + c + res_tm goes to `out` (B M, C) plain (the token-major companion the label block gathers its values from), to
 * out_nchw (B, C, M) or NULL (the block's channel-major result) and, normalised, into the keys' part of the label call's workspace;
 * that call then sets GKG_KNN_Y_PREPARED (and GKG_KNN_X_PREPARED when its own fc1 prepared the queries into the same workspace). */
int gkg_bn_apply_knn_prep(const float* y, const double* sums, const float* gamma, const float* beta, const float* bias,
                          float* running_mean, float* running_var, long long* num_batches_tracked, float* a, float* c_out,
                          float* mean, float* invstd, float* out, int ldo, int ochunk, int B, int G, int c, int N, int M, int k,
                          int dilation, int has_y, int has_relpos, unsigned knn_flags, int fused_mr, int as_keys, const float* res_tm,
                          float* out_nchw, void* knn_workspace, size_t knn_workspace_bytes, float momentum, float eps, double* zero_buf,
                          size_t zero_doubles, void* stream);
/* Pooled key set of a Grapher with r > 1 (reference torch_vertex.py:194-196, F.avg_pool2d(x, r, r)) from a token-major map
 * x (B, H, W, C) given as a view -> out (B, H/r, W/r, C) plain fp32 (floor mode; window sum in (h, w) order, one division). */
int gkg_avgpool_tm(const float* x, int ldx, int xchunk, float* out, int B, int H, int W, int C, int r, void* stream);

/* Compact graph (round 5): gkg_knn_fwd_tm's neighbour lists as u16 rows, nn16 (B*G, N, k), INSTEAD of the int64 planes — for
 * callers that consume the graph on the device and do not hand it out (Grapher.forward discards it, reference
 * torch_vertex.py:330): no int64 index plane, no centre plane (GKGNet-576 stage 1: 24 MB written per launch instead of 191 MB).
 * Same contract, same neighbours in the same order as gkg_knn_fwd_tm; M <= 65536.  gkg_mr_fwd_tm16 / gkg_mr_linear_bf16_nn16
 * are gkg_mr_fwd_tm / gkg_mr_linear_bf16 reading these lists (same outputs, same bits). */
int gkg_knn_fwd_tm16(const void* x, int ldx, int xchunk, const void* y, const float* relpos, uint16_t* nn16, int B, int G, int c,
                     int N, int M, int k, int dilation, int dtype, unsigned flags, void* workspace, size_t workspace_bytes,
                     void* stream);
int gkg_mr_fwd_tm16(const float* x, int ldx, int xchunk, const float* src, const uint16_t* nn16, void* out /* out_dtype elements */,
                    uint8_t* argmax, int B, int G, int c, int N, int M, int k, int mode, int out_dtype, int arg_kind, void* stream);

/* Row g2 (round 5): the k-NN graph AND the max-relative aggregation over it in ONE kernel, for token-major fp32 callers — the
 * reference chain DenseDilatedKnnGraph.forward (torch_edge.py:164-176) -> MRConv2d.forward's two batched_index_select + max
 * (torch_vertex.py:49-61) without the (2, B*G, N, k) int64 edge_index in between.  x (B, N, C = G*c), y (B, M, C) or NULL (self
 * graph), relative_pos (N, M) or NULL; flags as gkg_knn_fwd (GKG_KNN_NORMALIZE ...).  The graph is the one gkg_knn_fwd_tm
 * builds (same contract, same bits); it is consumed by the workgroup that built it:
 *   xm_out  (B*N, 2C) fp32      the grouped projection's operand buffer (XM layout), exactly gkg_mr_fwd_tm(mode 1)'s output:
 *                                m into the m chunks, x into the x chunks unless x is that buffer's x half already
 *   arg_out (B, N, C)     u16   the winning neighbour ROW per channel, exactly gkg_mr_fwd_tm(arg_kind 1)'s argmax (the
 *                                backward gkg_mr_bwd_tm(arg_kind 1) scatters from it and needs no index tensor)
 *   nn16_out (B*G, N, k)  u16   the neighbour lists, or NULL (M <= 65536)
 *   nn_idx_out, center_out (B*G, N, k) int64   gkg_knn_fwd_tm's outputs, or NULL: only for callers that RETURN the graph
 *                                (GrapherLabel.forward, torch_vertex.py:403); a Grapher never materialises them
 * gkg_knn_mr_fused_supported: 1 when this shape takes the fused form — the fp32 tile kernel with merged per-wave lists (no
 * key splits, no prefilter, lists of <= 36 entries, k <= 18, c % 4 == 0, C % 16 == 0); otherwise call gkg_knn_fwd_tm and
 * gkg_mr_fwd_tm.  Workspace: gkg_knn_workspace_bytes. */
int gkg_knn_mr_fused_supported(int B, int G, int c, int N, int M, int k, int dilation, int has_y, int has_relpos, unsigned flags);
int gkg_knn_mr_fwd_tm(const float* x, int ldx, int xchunk, const float* y, const float* relative_pos, float* xm_out,
                      uint16_t* arg_out, uint16_t* nn16_out, int64_t* nn_idx_out, int64_t* center_out, int B, int G, int c, int N,
                      int M, int k, int dilation, unsigned flags, void* workspace, size_t workspace_bytes, void* stream);
/* arg_kind: what `argmax` holds — 0: (B,N,C) u8, the winning slot j (as gkg_mr_fwd); 1: (B,N,C) u16, the winning
 * neighbour's row index itself (M <= 65536), which lets the backward scatter without looking the index row up again. */
int gkg_mr_fwd_tm(const float* x, int ldx, int xchunk, const float* src, const int64_t* nn_idx, void* out /* out_dtype elements */,
                  uint8_t* argmax, int B, int G, int c, int N, int M, int k, int mode, int out_dtype, int arg_kind,
                  void* stream);
#define GKG_MR_DETERMINISTIC 1u /* fixed summation order in the scatter: bit-identical results from run to run */
#define GKG_MR_FP32_ATOMICS 2u  /* keep the fp32 LDS-atomic scatter instead of the exact integer accumulation (measurement, tests) */
int gkg_mr_bwd_tm(const float* gin, const int64_t* nn_idx, const uint8_t* argmax, float* gx, float* gsrc,
                  int B, int G, int c, int N, int M, int k, int mode, int arg_kind, unsigned flags, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SURVEY §8 row g1, inference: the aggregation as the OPERAND PRODUCER of the grouped 1x1 projection.
 * One launch replaces MRConv2d.forward's gather -> max(x_j - x_i) -> interleave -> BasicConv (Conv2d(2C, 2C, 1,
 * groups=4) + BN(eval) + GELU) chain (reference torch_vertex.py:47-62, torch_nn.py:57-69): the [x, m] tensor is built
 * tile by tile in LDS (bf16, round-to-nearest-even — the operand rounding of the reference's autocast convolution) and
 * consumed by v_mfma_f32_32x32x16_bf16 in place; it never exists in memory.
 *   x (B, N, C) fp32 token-major; src (B, M, C) fp32 or NULL (self graph); nn_idx (B*G, N, k) int64; C = G*c, C % 16 == 0
 *   wplanes: bf16 weight fragments [4][ci_pad/8][co_pad][8], ci = co = C/2, ci_pad = ci rounded up to 16, co_pad = co
 *            rounded up to 32: fragment (q, f, n) = W[q*co + n][8f .. 8f+7] (zero outside); gkg_mr_linear_planes_bytes(C)
 *   a, cshift (2C) fp32: out = act(a * conv + cshift) per output channel (eval-mode BN folded with the conv bias)
 *   out (B*N, ldo) bf16, columns [0, 2C) written (column q*co + n), ldo >= 2C, ldo % 8 == 0; act: 0 none, 1 GELU (erf)
 * The max-relative arithmetic is gkg_mr_fwd_tm's (bit-identical m before rounding). */
size_t gkg_mr_linear_planes_bytes(int C);
int gkg_mr_linear_bf16(const float* x, const float* src, const int64_t* nn_idx, const void* wplanes, const float* a,
                       const float* cshift, void* out, int ldo, int B, int G, int c, int N, int M, int k, int act,
                       void* stream);
int gkg_mr_linear_bf16_nn16(const float* x, const float* src, const uint16_t* nn16, const void* wplanes, const float* a,
                            const float* cshift, void* out, int ldo, int B, int G, int c, int N, int M, int k, int act,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * Bandwidth kernels between the dense 1x1 projections (Conv2d 1x1 + SyncBN [+ GELU], reference
 * torch_vertex.py:290-306,334-360; torch_nn.py:57-69).  The projections themselves are plain GEMMs run by the
 * caller in the vendor library on token-major (rows = tokens) fp32 matrices.  `nb` stacks nb independent
 * (R, C) matrices (the 4 groups of the grouped projection) with parameters laid out [nb][C].
 */
/* (B,C,N) fp32 -> (B*N,C) in out_dtype (GKG_F32, or GKG_BF16 when the result only feeds a bf16 GEMM); img_scale (B) or
 * NULL multiplies image b (the backward of a stochastic-depth branch) */
int gkg_nchw_to_tm(const float* x, void* out, int B, int C, int N, int out_dtype, const float* img_scale, void* stream);
/* out(B,C,N) = (a[ch]*y[t][ch] + c[ch]) * img_scale[b] + res(B,C,N); a/c, img_scale and res optional (NULL).
 * img_scale is the reference's DropPath (torch_vertex.py:332, timm): per-image Bernoulli keep mask / keep probability. */
int gkg_tm_affine_to_nchw(const float* y, const float* a, const float* c, const float* res, float* out,
                          int B, int C, int N, const float* img_scale, void* stream);
/* Round 5: a block output handed out in BOTH layouts.  A Grapher that takes and returns NCHW (reference torch_vertex.py:325-333)
 * in front of a GrapherLabel (torch_vertex.py:392-403, which reads the feature map token-major as keys / values) used to pay a
 * layout pass in the label branch's forward and three in its backward (token-major gradient -> NCHW, + the other gradient,
 * -> token-major again).  gkg_tm_affine_to_nchw_dual: out(B,C,N) and out_tm(B*N,C) = a*y + c + res_tm, the residual given
 * token-major (the block's own token-major copy of its input).  gkg_nchw_to_tm_add: out(B*N,C) = x(B,C,N)^T + add_tm — the two
 * upstream gradients of such an output summed while the NCHW one is re-laid out. */
int gkg_tm_affine_to_nchw_dual(const float* y, const float* a, const float* c, const float* res_tm, float* out, float* out_tm,
                               int B, int C, int N, void* stream);
int gkg_nchw_to_tm_add(const float* x, const float* add_tm, float* out, int B, int C, int N, void* stream);
size_t gkg_bn_workspace_bytes(int R, int C, int nb);
/* Train-mode batch statistics of y (R,C) (conv bias NOT included in y; it is folded: it cancels in the output and
 * is added to running_mean).  Writes scale a, shift c (out = a*y + c), saved mean / invstd; updates running stats
 * (momentum, unbiased variance) when given.  Deterministic two-stage reduction. */
int gkg_bn_train_stats(const float* y, const float* gamma, const float* beta, const float* bias,
                       float* running_mean, float* running_var, float* a, float* c, float* mean, float* invstd,
                       int R, int C, int nb, float momentum, float eps, long long* num_batches_tracked /* += 1, or NULL */,
                       void* workspace, size_t workspace_bytes, void* stream);
/* Eval mode: a = gamma/sqrt(rv+eps), c = beta + a*(bias - rm) */
int gkg_bn_eval_affine(const float* gamma, const float* beta, const float* bias, const float* running_mean,
                       const float* running_var, float* a, float* c, int C, float eps, void* stream);
/* out = act(a*y + c) * row_scale[r / rows_per_scale] (+ res); act 0 = identity, 1 = GELU(erf); row_scale optional (NULL).
 * out[q] has row pitch ldo and batch stride out_bstride (both in elements of out_dtype: GKG_F32, or GKG_BF16 =
 * round-to-nearest-even of the fp32 result).  ochunk > 0: column ch is written at ch + (ch / ochunk) * ochunk — the x half of
 * an XM buffer (ldo >= 2C; see "XM layout") */
int gkg_affine_act(const float* y, const float* a, const float* c, const float* res, void* out, int R, int C,
                   int nb, int ldo, size_t out_bstride, int ochunk, int act, int out_dtype, const float* row_scale,
                   int rows_per_scale, void* stream);
/* gkg_affine_act for one batch of contiguous rows writing the result twice: fp32 (the residual stream) and its bf16
 * rounding (the next projection's operand in bf16 inference) — no stand-alone cast pass between blocks. */
int gkg_affine_act_dual(const float* y, const float* a, const float* c, const float* res, float* out_f32, void* out_bf16,
                        int R, int C, int act, const float* row_scale, int rows_per_scale, void* stream);

/* out = act(a * y + c) for a BF16 matrix y (R, C), C % 8 == 0: the output of a library convolution under bf16 autocast viewed
 * token-major (channels-last), eval-mode BN (+ conv bias) folded into a / c (reference gkgnet.py:79-118 in eval mode).  Writes
 * out_f32 (fp32: the residual stream) and / or out_bf16 (the next operand); at least one must be non-null.  act: 0 / 1 (GELU). */
int gkg_affine_act_bf16in(const void* y_bf16, const float* a, const float* c, float* out_f32, void* out_bf16, int R, int C,
                          int act, void* stream);
/* Backward of out = act(BN_train(y)): dy, dgamma, dbeta from dout (row pitch ldg, batch stride dout_bstride). */
int gkg_bn_bwd(const float* dout, const float* y, const float* a, const float* c, const float* mean,
               const float* invstd, float* dy, float* dgamma, float* dbeta, int R, int C, int nb, int ldg,
               size_t dout_bstride, int act, void* workspace, size_t workspace_bytes, void* stream);

/* Train-mode BN-apply straight from the projection kernel's fp64 column sums (gkg_linear_bn_fwd / gkg_linear_bn_fwd_x6 with
 * train == 2: statistics only): the consumer derives scale / shift itself — no finalize launch in between.  Every workgroup
 * computes its channels' coefficients once, the first one per group also writes the saved a / c / mean / invstd [nb][C],
 * updates running_mean / running_var (conv `bias` folded) and num_batches_tracked, and clears `zero_buf` (`zero_doubles`
 * doubles): the OTHER of the caller's two alternating scratch buffers, i.e. what the previous projection accumulated into.
 * nchw_B == 0: token-major output like gkg_affine_act (fp32); nchw_B > 0: (nchw_B, C, R / nchw_B) channel-major output and
 * residual like gkg_tm_affine_to_nchw (nb == 1, act == 0, row_scale = one factor per image).  Same arithmetic as the
 * finalize + apply launches it replaces. */
int gkg_bn_apply_train(const float* y, const double* sums, const float* gamma, const float* beta, const float* bias,
                       float* running_mean, float* running_var, long long* num_batches_tracked, float* a, float* c,
                       float* mean, float* invstd, const float* res, float* out, int R, int C, int nb, int ldo,
                       size_t out_bstride, int ochunk /* as gkg_affine_act */, int act, int nchw_B, const float* row_scale,
                       int rows_per_scale, float momentum, float eps, double* zero_buf, size_t zero_doubles, void* stream);
/* gkg_bn_apply_train's channel-major form as gkg_tm_affine_to_nchw_dual: residual token-major, result in both layouts. */
int gkg_bn_apply_train_dual(const float* y, const double* sums, const float* gamma, const float* beta, const float* bias,
                            float* running_mean, float* running_var, long long* num_batches_tracked, float* a, float* c,
                            float* mean, float* invstd, const float* res_tm, float* out, float* out_tm, int B, int C, int N,
                            float momentum, float eps, double* zero_buf, size_t zero_doubles, void* stream);
/* gkg_bn_bwd in two launches instead of three: the statistics pass accumulates its column sums into `sums` (fp64,
 * 2 * nb * C doubles, ZERO on entry) with atomics, the apply pass reads them, writes dgamma / dbeta and clears `zero_buf`
 * (`zero_doubles` doubles; NULL / 0: nothing).  The caller alternates between two scratch buffers and passes the region the
 * previous call used as `zero_buf`, so every buffer is clean before its next use without a memset launch.  The sums are
 * run-dependent in their last fp64 bits (atomics); callers that need bit-reproducibility use gkg_bn_bwd. */
int gkg_bn_bwd_atomic(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                      const float* invstd, float* dy, float* dgamma, float* dbeta, int R, int C, int nb, int ldg,
                      size_t dout_bstride, int act, double* sums, double* zero_buf, size_t zero_doubles, void* stream);


/* gkg_bn_bwd_atomic for a branch whose output was scaled per image (DropPath: torch_vertex.py:332,355,402): the incoming
 * gradient is multiplied by row_scale[row / rows_per_scale] inside both passes. */
int gkg_bn_bwd_atomic_scaled(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                             const float* invstd, float* dy, float* dgamma, float* dbeta, int R, int C, int nb, int ldg,
                             size_t dout_bstride, int act, double* sums, double* zero_buf, size_t zero_doubles,
                             const float* row_scale, int rows_per_scale, void* stream);
/* Cross-rank batch statistics (the reference's SyncBatchNorm under DDP, torch_nn.py:37 / mmcv build_norm_layer):
 * gkg_bn_train_stats and gkg_bn_bwd split where the ranks exchange statistics.  Forward: gkg_bn_stats_sums ->
 * caller all-reduces `sums` [nb][2][C] (column sum, sum of squares) and the row count over the ranks ->
 * gkg_bn_finalize with the device scalar `count` (total rows).  Backward: gkg_bn_bwd_sums (also writes the LOCAL
 * dgamma/dbeta, like torch's batch_norm_backward_reduce, and parks dz in dy when act == 1) -> all-reduce `sums`
 * [nb][2][C] (sum dz, sum dz*yhat) -> gkg_bn_bwd_apply with the same `count`. */
int gkg_bn_stats_sums(const float* y, float* sums, int R, int C, int nb, void* workspace, size_t workspace_bytes,
                      void* stream);
int gkg_bn_finalize(const float* sums, const float* count, const float* gamma, const float* beta, const float* bias,
                    float* running_mean, float* running_var, float* a, float* c, float* mean, float* invstd, int C,
                    int nb, float momentum, float eps, long long* num_batches_tracked, void* stream);
int gkg_bn_bwd_sums(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                    const float* invstd, float* dy, float* sums, float* dgamma, float* dbeta, int R, int C, int nb,
                    int ldg, size_t dout_bstride, int act, void* workspace, size_t workspace_bytes, void* stream);
int gkg_bn_bwd_apply(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                     const float* invstd, const float* sums, const float* count, float* dy, int R, int C, int nb,
                     int ldg, size_t dout_bstride, int act, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The dense 1x1 projections (reference torch_vertex.py:290-306 fc1 / fc2 = Conv2d(1x1) + BN, torch_nn.py:57-69 BasicConv =
 * Conv2d(1x1, groups=4) + BN + GELU behind the aggregation, torch_vertex.py:334-360 FFNLabel) with their input and weight
 * gradients, on the bf16 matrix cores at fp32 accuracy (csrc/gkg_gemm_x6.hip): every fp32 operand is split exactly into three
 * bf16 terms and six of the nine cross products are accumulated in fp32 (error vs fp64 measured 3-4x BELOW an fp32 fma chain
 * of the same length).  The WEIGHTS are split ahead of time into bf16 "planes", once per optimiser step and for all layers in
 * one launch; the activations are split inside the GEMM.  Token-major fp32: x (nb, R, cin), w (nb, cout, cin), y (nb, R, cout);
 * nb = 4 stacks the groups of the grouped projection.  (An fp32-MFMA forward kernel — rounds 1-5, csrc/gkg_gemm.hip — lost to
 * these at every shape and was removed in round 6.)
 *   gkg_linear_stats_doubles doubles of the fp64 column-sum scratch the statistics epilogue accumulates into
 *   gkg_x6_planes_bytes     bytes of one orientation's planes of a weight w (nb, cout, cin): dgrad = 0 forward, 1 dgrad
 *   gkg_x6_prep_desc_bytes  size of one descriptor of the batched split
 *   gkg_x6_prep_desc_fill   writes descriptor `index` into a HOST array (device pointers inside; one of the two plane
 *                           pointers may be NULL: that orientation is skipped); returns the running unit count to pass
 *                           as `unit_begin` of the next descriptor (-1: bad arguments)
 *   gkg_x6_prep_weights     ONE launch: every described weight -> its forward and dgrad planes.  `descs_dev`: the array
 *                           copied to device memory; total_units: the last gkg_x6_prep_desc_fill return value
 *   gkg_linear_bn_fwd_x6    y = x w^T with w given as `planes_fwd`; x has row pitch ldx and batch stride x_bstride (floats; a
 *                           column slice of a wider matrix is allowed).  train != 0: the train-mode BN statistics of y are
 *                           taken in the kernel's epilogue (per tile centred mean / M2 from the accumulator registers,
 *                           accumulated into fp64 column sums with atomics) and a small finalize kernel writes scale a, shift c
 *                           (out = a*y + c; the conv bias is folded: it cancels in the output and is added to running_mean),
 *                           saved mean / invstd and updates the running statistics.  `stats`: gkg_linear_stats_doubles()
 *                           doubles, zero on entry, zeroed again before the call's work completes (one buffer can serve every
 *                           layer on a stream).  train == 2: statistics only — the fp64 sums stay in `stats` for
 *                           gkg_bn_apply_train.  train == 0: plain projection.
 *   gkg_linear_dgrad_x6     dx (nb, R, cin) = dy (nb, R, cout; pitch ldg, batch stride g_bstride) w
 *   gkg_linear_wgrad_x6     dw (nb, cout, cin) += dy^T x (see below)
 * Rows must be 16-byte aligned (base pointer % 16 == 0, pitches % 4 == 0); each batch of x / dy below 4 GiB. */
int gkg_linear_stats_doubles(void);
size_t gkg_x6_planes_bytes(int cin, int cout, int nb, int dgrad);
int gkg_x6_prep_desc_bytes(void);
/* kperm != 0 (the grouped projection behind the aggregation, "XM layout"): the planes hold w's input columns as [even columns |
 * odd columns] per group — plane position p < cin/2 is column 2p (an x channel), p >= cin/2 column 2 (p - cin/2) + 1 (its m). */
long long gkg_x6_prep_desc_fill(void* host_descs, int index, const float* w, void* planes_fwd, void* planes_dgrad, int cin,
                                int cout, int nb, long long unit_begin, int kperm);
int gkg_x6_prep_weights(const void* descs_dev, int ndesc, long long total_units, void* stream);
/* The same launch also clearing up to two caller buffers (16-byte aligned, sizes multiples of 16; NULL / 0: none) — round 5: a
 * training step clears its flat gradient buffer and the fp64 BN scratch in front of the first projection anyway; riding in the
 * weight-split launch they cost no launch of their own. */
int gkg_x6_prep_weights_zero(const void* descs_dev, int ndesc, long long total_units, void* zero0, size_t zero0_bytes,
                             void* zero1, size_t zero1_bytes, void* stream);
int gkg_linear_bn_fwd_x6(const float* x, int ldx, size_t x_bstride, const void* planes_fwd, float* y, int R, int cin,
                         int cout, int nb, int train, const float* gamma, const float* beta, const float* bias,
                         float* running_mean, float* running_var, long long* num_batches_tracked, float* bn_a, float* bn_c,
                         float* bn_mean, float* bn_invstd, float momentum, float eps, double* stats, void* stream);
int gkg_linear_dgrad_x6(const float* dy, int ldg, size_t g_bstride, const void* planes_dgrad, float* dx, int R, int cin,
                        int cout, int nb, void* stream);
/* Split-K forms (round 5) for few rows under a long contraction (the label branch's 2 560-row matrices: 100 workgroups of 40
 * K-steps on 256 CUs): the same calls with a caller-owned workspace of gkg_x6_splitk_workspace_bytes() bytes (NULL: never
 * split) whose first 4 KiB (tile counters) are ZERO before the first use — every launch leaves them zero again.  The library cuts the contraction
 * into up to 8 ranges when that fills the chip (shapes it does not split run exactly as without the workspace); the last
 * workgroup to arrive at a tile adds the partial tiles in range order (run-to-run identical bits) and runs the normal
 * epilogue, BN statistics included.  gkg_linear_dgrad_x6_sk: `residual` (nb, R, cin) contiguous or NULL is added to dx in
 * the epilogue (the skip connection's gradient, reference torch_vertex.py:331,354,402). */
size_t gkg_x6_splitk_workspace_bytes(void);
/* With a workspace the _sk entry points run matrices of at most 4 096 rows on a body whose four waves split K inside the
 * workgroup (32-row x 64-column tiles, private LDS rings, B fragments straight from the weight planes, no barrier in the loop,
 * no cross-workgroup hand-off: csrc/gkg_gemm_x6.hip gemm_x6_ks_kernel) — the cross-workgroup split-K form above then only
 * serves what that body does not.  Taken for un-grouped projections of at most 640 output columns (where it measured faster:
 * a 32-row workgroup re-reads all of B).  `flags` (per call; measurement, tests): GKG_X6_NO_KS never takes it,
 * GKG_X6_FORCE_KS takes it for every short matrix. */
#define GKG_X6_NO_KS 1u
#define GKG_X6_FORCE_KS 2u
int gkg_linear_bn_fwd_x6_sk(const float* x, int ldx, size_t x_bstride, const void* planes_fwd, float* y, int R, int cin,
                            int cout, int nb, int train, const float* gamma, const float* beta, const float* bias,
                            float* running_mean, float* running_var, long long* num_batches_tracked, float* bn_a, float* bn_c,
                            float* bn_mean, float* bn_invstd, float momentum, float eps, double* stats, void* splitk_ws,
                            size_t splitk_bytes, unsigned flags, void* stream);
/* ldx / x_bstride: row pitch and batch stride of dx in floats (ldx == 0: contiguous (nb, R, cin)); `residual` has dx's layout.
 * The grouped projection behind the aggregation writes dXM (R, 2C) directly: nb = 4, cin = C/2, ldx = 2C, x_bstride = C/2. */
int gkg_linear_dgrad_x6_sk(const float* dy, int ldg, size_t g_bstride, const void* planes_dgrad, float* dx, int R, int cin,
                           int cout, int nb, const float* residual, void* splitk_ws, size_t splitk_bytes, int ldx,
                           size_t x_bstride, unsigned flags, void* stream);
/* dw (nb, cout, cin) += dy^T x over the R rows (both operands split in registers; no LDS staging, each wave streams its own
 * rows).  dw must be ZERO on entry: slabs of rows are added with fp32 atomics (run-dependent summation order, like a
 * split-K GEMM).  x (nb, R, cin) with row pitch ldx / batch stride x_bstride (floats).  Any cin, cout >= 1. */
/* kperm != 0: x arrives with its columns as [x chunk | m chunk] per group (an XM buffer read with ldx = 2C, x_bstride = C/2):
 * operand column j is added to dw column 2j (j < cin/2) or 2 (j - cin/2) + 1 — dw keeps the reference's interleaved layout. */
int gkg_linear_wgrad_x6(const float* dy, int ldg, size_t g_bstride, const float* x, int ldx, size_t x_bstride, float* dw,
                        int R, int cin, int cout, int nb, int kperm, void* stream);
/* The weight gradients of SEVERAL layers in one launch (round 5).  Nothing downstream of a backward pass reads a dW, so the
 * host side may queue the weight gradients of every projection (reference torch_vertex.py:290-306, :334-360, torch_nn.py:57-69
 * backward) while the input gradients run and issue them together when the backward ends: at this path's sizes each of them
 * alone is a launch of 50-400 workgroups on 256 CUs.  Problem i is exactly gkg_linear_wgrad_x6(dy, ldg, g_bstride, x, ldx,
 * x_bstride, dw, R, cin, cout, nb): every dw ZERO on entry, accumulated with fp32 atomics.  At most 256 problems per call
 * (16 per launch); operands must stay valid until the launch has run.  units_per_slab: rows per workgroup in units of 128
 * (0: the library's default, 20). */
typedef struct GkgWgradProblem {
  const float* dy;
  const float* x;
  float* dw;
  size_t g_bstride, x_bstride;
  int ldg, ldx, R, cin, cout, nb;
  int kperm;
} GkgWgradProblem;
int gkg_linear_wgrad_x6_batch(const GkgWgradProblem* problems, int n, int units_per_slab, void* stream);

/*
 * Input gradient of a projection with the BACKWARD statistics of the layer in front of it in its epilogue (round 4).
 * dx (R, cin) = dy (R, cout; row pitch ldg) * w, weights as dgrad planes — like gkg_linear_dgrad_x6 with nb == 1 — where dx is
 * at the same time the upstream gradient g of the layer  h = act(BN_train(py))  that produced this projection's input
 * (reference: the chains fc1 -> BN -> GELU -> fc2 of FFN gkgnet.py:66-72 / FFNLabel torch_vertex.py:352-358 and
 * BasicConv -> fc2 of Grapher torch_vertex.py:329-330).  psums [pnb][2][pco] fp64 receives, with atomics, per channel
 *     sum_r dz   and   sum_r dz * yhat,     dz = g * act'(pa * py + pc),  yhat = (py - pmean) * pinvstd,
 * i.e. exactly what the statistics pass of gkg_bn_bwd_atomic would compute from g and py — which then runs as
 * gkg_bn_bwd_apply_from_sums (apply pass only).  py (pnb, R, pco) with cin == pnb * pco; column n of dx is group n / pco,
 * channel n % pco; pact 0 none / 1 GELU.
 */
int gkg_linear_dgrad_x6_bnbwd(const float* dy, int ldg, const void* planes_dgrad, float* dx, int R, int cin, int cout,
                              const float* py, const float* pa, const float* pc, const float* pmean, const float* pinvstd,
                              double* psums, int pnb, int pco, int pact, void* stream);
int gkg_bn_bwd_apply_from_sums(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                               const float* invstd, float* dy, float* dgamma, float* dbeta, int R, int C, int nb, int ldg,
                               size_t dout_bstride, int act, const double* sums, double* zero_buf, size_t zero_doubles,
                               void* stream);

/* ------------------------------------------------------------------------------------------------
 * Block-level entry points (round 6; csrc/gkg_block.hip): ONE call runs the whole launch sequence of a Grapher / GrapherLabel
 * block's forward or backward (reference torch_vertex.py:325-333, :392-403 + FFNLabel :334-360) from a descriptor — the host side
 * allocates, fills the descriptor and calls.  Every launch inside is one of the entry points above, in the order and with the
 * arguments of the per-layer composition: bit-identical results.  Scope: fp32, train-mode BatchNorm with rank-local statistics,
 * no DropPath scaling, un-pooled keys (r == 1), C % 16 == 0, every projection on the split-bf16 kernels.  The library allocates
 * nothing and keeps no state: all buffers — including, per BN pass, the fp64 column-sum buffer to accumulate into and the region
 * of the OTHER buffer to clear (the caller's alternating pair, see gkg_bn_bwd_atomic) — come in the descriptor.
 *   GkgProjBN   one 1x1 projection + BatchNorm: weight as x6 planes (gkg_x6_prep_weights; the grouped projection behind the
 *               aggregation with kperm), BN parameters, what the forward saves for the backward (Y: pre-BN output (nb, R, cout);
 *               bn: [4][nb cout] = a, c, mean, invstd), the backward's outputs (dgamma, dbeta; dw: where the weight-gradient
 *               problem it emits will add — zero on entry).
 *   GkgGraphOp  the k-NN + aggregation of the block: flags as gkg_knn_fwd (GKG_KNN_X_PREPARED: fc1's BN-apply prepares the
 *               queries — gkg_bn_apply_knn_prep; GKG_KNN_Y_PREPARED: the keys are already in knn_ws); fused_mr: one kernel
 *               (gkg_knn_mr_fused_supported), else k-NN + aggregation with u16 lists (nn16) or — a caller that returns the
 *               graph — int64 lists (nn_idx, center).  arg (B N, C) u16: the winning rows (saved for the backward).
 * Weight gradients: the backward calls FILL wq[] (3 / 5 problems, in backward order) for gkg_linear_wgrad_x6_batch; the caller
 * launches them at once or queues them with the rest of its backward pass (operands: dY buffers and saved activations must stay
 * valid until then). */
typedef struct GkgProjBN {
  const void* planes_fwd; const void* planes_dgrad;
  const float* gamma; const float* beta; const float* bias;
  float* running_mean; float* running_var; long long* nbt;
  float momentum, eps;
  int cin, cout, nb;
  double* fsum; double* fzero; size_t fzero_n;      /* forward BN pass: accumulate into / clear */
  double* bsum; double* bzero; size_t bzero_n;      /* backward BN pass */
  float* Y; float* bn;
  float* dw; float* dgamma; float* dbeta;
} GkgProjBN;
typedef struct GkgGraphOp {
  int G, k, d, fused_mr;
  const float* relpos; unsigned knn_flags, mr_flags;
  void* knn_ws; size_t knn_ws_bytes;
  uint16_t* arg; uint16_t* nn16; int64_t* nn_idx; int64_t* center;
} GkgGraphOp;
typedef struct GkgGrapherBlock {
  int B, C, H, W;
  const float* x; float* out; float* out_tm;                 /* out_tm (B N, C) or NULL: the token-major companion */
  float* xt; float* XM; float* A2;                            /* saved: (T, C), (T, 2C), (T, 2C) */
  GkgProjBN fc1, conv, fc2;
  GkgGraphOp graph;
  void* sk_ws; size_t sk_bytes;                               /* gkg_x6_splitk_workspace_bytes() */
  /* optional (with out_tm): the label graph behind this block — its keys are prepared by the last pass (gkg_bn_apply_knn_prep as_keys) */
  int keys_G, keys_L, keys_k, keys_d, keys_fused_mr; unsigned keys_flags; void* keys_ws; size_t keys_ws_bytes;
  /* backward */
  const float* dout; const float* dout_tm; float* dx;
  float* g3; float* dY3; float* dA2; float* dY2; float* dXM; float* gx1; float* dY1; float* dxt;
} GkgGrapherBlock;
typedef struct GkgLabelBlock {
  int B, C, L, M;
  const float* e; const float* ft; float* out;               /* e (B L, C), keys / values ft (B, M, C), out (B L, C) */
  float* XM; float* A2; float* h2; float* f1;                 /* saved: (T, 2C), (T, 2C), (T, C), (T, Cf) */
  GkgProjBN fc1, conv, fc2, ffn1, ffn2;
  GkgGraphOp graph;
  void* sk_ws; size_t sk_bytes;
  /* backward */
  const float* dout; float* de; float* dft;
  float* dY5; float* df1; float* dY4; float* dh2; float* dY3; float* dA2; float* dY2; float* dXM; float* gx1; float* dY1;
} GkgLabelBlock;
int gkg_grapher_fwd(const GkgGrapherBlock* b, void* stream);
int gkg_grapher_bwd(const GkgGrapherBlock* b, GkgWgradProblem* wq /* [3] */, void* stream);
int gkg_grapher_label_fwd(const GkgLabelBlock* b, void* stream);
int gkg_grapher_label_bwd(const GkgLabelBlock* b, GkgWgradProblem* wq /* [5] */, void* stream);

/*
 * The backbone's first stem convolution (reference gkgnet.py:79-81: Conv2d(3 -> C1/2, 3x3, stride 2, padding 1) on the image),
 * as a direct kernel — with 3 input channels the library's implicit-GEMM forms run at a few TFLOP/s — writing the
 * channels-last tensor the next convolution and the blocks' token-major kernels want; optionally the eval-mode BN (a, c:
 * gkg_bn_eval_affine, conv bias folded) and GELU behind it (gkgnet.py:81-83) in the epilogue.
 *   out (B, Ho, Wo, cout) = act(a * (conv(x) + bias) + c);  x (B, cin, H, W) fp32, w (cout, cin, 3, 3) fp32;
 *   a, c both NULL: the plain convolution (+ bias);  Ho = (H + 1) / 2, Wo = (W + 1) / 2;  out_dtype GKG_F32 / GKG_BF16.
 */
int gkg_stem_conv3x3s2_supported(int cin, int cout);
int gkg_stem_conv3x3s2_fwd(const float* x, const float* w, const float* bias, const float* a, const float* c, void* out, int B,
                           int cin, int H, int W, int cout, int act, int out_dtype, void* stream);

/*
 * Opt-in kernel timing (measurement only; off by default, nothing is recorded on the hot path when off).
 * When enabled, every kernel launch made by this library is bracketed by hipEventRecord on the SAME
 * stream it is launched on.  gkg_prof_read synchronises on the recorded events (so call it outside any
 * timed / captured region) and accumulates per kernel id.
 */
#define GKG_PROF_TOKEN_PREP 0
#define GKG_PROF_KNN_TILE 1
#define GKG_PROF_KNN_MERGE 2
#define GKG_PROF_MR_FWD 3
#define GKG_PROF_MR_BWD 4
#define GKG_PROF_GEMM_X6 5  /* gemm_x6_kernel (forward and dgrad launches) */
#define GKG_PROF_NUM 6
void gkg_prof_enable(int on);
void gkg_prof_reset(void);
/* total milliseconds and number of launches recorded for `kernel_id` since the last reset; 0 on success. */
int gkg_prof_read(int kernel_id, double* total_ms, long* launches);
/* Algorithmic work of the launches counted by gkg_prof_read since the last reset, for the kernels that report it:
 * GKG_PROF_GEMM_X6: flop, 2 R cin cout nb per launch; GKG_PROF_KNN_TILE: flop of the distance contraction, 2 BG c N M per
 * gkg_knn_fwd[_tm] call (SURVEY §8d flops_knn); GKG_PROF_MR_FWD / _BWD (token-major entry points): BYTES per SURVEY §8d —
 * fwd: x + keys (bipartite) + int64 indices + m + 1 B/element of argmax when saved; bwd: g + indices + argmax + gx + gsrc;
 * 0 otherwise. */
double gkg_prof_work(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* GKG_HIP_H_ */
