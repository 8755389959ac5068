/* gmlm_hip.h — C ABI of libgmlm_hip.so: the MI355X (gfx950) kernels behind the GraphTextLM hot path.
 *
 * The reference (chungimungi/GMLM, main.py) has no FFI layer: its hot path is the Python
 * nn.Module surface GraphTextLM.forward / get_graph_embeddings (main.py:250, 322) calling into
 * PyTorch-Geometric and HuggingFace operators.  This header is the boundary a drop-in for that
 * path binds to (ctypes stub: INTEGRATION.md).  Each entry cites the reference interface it
 * replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors); the library never
 *    allocates or frees device memory, never synchronises the device, and enqueues only on `stream`
 *    (a hipStream_t passed as void*; NULL = the null stream);
 *  - every call returns 0 on success or a negative GMLM_E* code; gmlm_last_error() then returns a
 *    thread-local message; arguments are validated on the host BEFORE any launch;
 *  - tensors are row-major and dense unless a stride argument says otherwise (strides in elements);
 *  - dtype: GMLM_F32 or GMLM_BF16 selects the storage type of activations; accumulation and all
 *    statistics are fp32; index arrays are int32 unless the reference hands over int64 (edge_index);
 *  - re-entrant and thread-safe: the only state kept is a write-once "kernel attributes set on device d" bit per
 *    kernel family, taken under a lock (one process may drive several GPUs / call from several host threads);
 *    safe to call twice with the same arguments (torch.utils.checkpoint recompute, main.py:278-314).
 */
#ifndef GMLM_HIP_H
#define GMLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMLM_ABI_VERSION 1

enum gmlm_dtype { GMLM_F32 = 0, GMLM_BF16 = 1 };
enum gmlm_status {
  GMLM_OK = 0,
  GMLM_EINVAL = -1,    /* bad argument (null pointer, size, alignment, unsupported shape) */
  GMLM_EWORKSPACE = -2,/* workspace too small */
  GMLM_ELAUNCH = -3,   /* hip launch / runtime error */
  GMLM_EDEVICE = -4    /* not a gfx950 device */
};

typedef void* gmlm_stream_t;

int gmlm_version(void);
const char* gmlm_last_error(void);
/* Fills CU count / wavefront size / arch name of the current device; GMLM_EDEVICE unless gfx950. */
int gmlm_device_check(int* cu_count, int* wave_size, char* arch, int arch_len);

/* ---------------------------------------------------------------------------------------------
 * K1  graph preprocessing (integer, bit-exact)
 * replaces: torch_geometric.utils.degree (main.py:65, 256), the per-edge Python bucketing loop
 *           (main.py:257-267) and RGCNConv's per-relation masked_edge_index compaction
 *           (PyG RGCNConv.forward, call sites main.py:272-308).
 * ------------------------------------------------------------------------------------------- */
/* deg[v] = #{e : index[e] == v}; deg is zeroed by the call.  index values must be in [0, n). */
int gmlm_degree_i32(const int64_t* index, int64_t e, int64_t n, int32_t* deg, gmlm_stream_t stream);
/* PyG `degree` drop-in: float32 counts. */
int gmlm_degree_f32(const int64_t* index, int64_t e, int64_t n, float* deg, gmlm_stream_t stream);
/* edge_type[e] = 0 if deg[src[e]] <= 2, 1 if <= 5, 2 if <= 10, else 3   (main.py:260-267).  deg has n entries; a
 * source id outside [0, n) is typed 0 here (no out-of-bounds read) and rejected by gmlm_segment_sort's checks. */
int gmlm_edge_bucket(const int64_t* src, const int32_t* deg, int64_t e, int64_t n, int64_t* edge_type,
                     gmlm_stream_t stream);
/* rel_count[r] = #{e : edge_type[e] == r}, r < num_relations (zeroed by the call). */
int gmlm_relation_histogram(const int64_t* edge_type, int64_t e, int num_relations, int32_t* rel_count,
                            gmlm_stream_t stream);

/* Stable segment sort.  key[e] = node[e] * r_active + (rel ? rel_remap[rel[e]] : 0).
 * Outputs: keys[e] (unsorted int32 keys, optional), perm[t] = original id of the t-th edge in
 * (key, original id) order, rowptr[s] = first t with sorted key >= s, s in [0, num_segments].
 * Edges whose relation maps to a negative remap entry are an error (GMLM_EINVAL is reported by
 * the caller-side check; the kernel clamps them to segment 0 and sets *bad_flag != 0). */
size_t gmlm_segment_sort_workspace_bytes(int64_t e);
int gmlm_segment_sort(const int64_t* node, const int64_t* rel, const int32_t* rel_remap, int r_active, int64_t e,
                      int64_t num_segments, int32_t* keys, int32_t* perm, int32_t* rowptr, int32_t* bad_flag,
                      void* workspace, size_t workspace_bytes, gmlm_stream_t stream);
/* out[t] = (int32) src[perm[t]] (int64 source) / out[t] = src[perm[t]] (int32 source) */
int gmlm_gather_i64_to_i32(const int64_t* src, const int32_t* perm, int64_t e, int32_t* out, gmlm_stream_t stream);
int gmlm_gather_i32(const int32_t* src, const int32_t* perm, int64_t e, int32_t* out, gmlm_stream_t stream);
/* inv_cnt[s] = 1 / max(rowptr[s+1] - rowptr[s], 1) */
int gmlm_segment_inv_count(const int32_t* rowptr, int64_t num_segments, float* inv_cnt, gmlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2/K3  relation-segmented CSR mean aggregation (the "SpMM") and its transpose
 * replaces: RGCNConv.propagate(aggr='mean') = gather x[src] + scatter-mean to dst per relation
 *           (PyG; call sites main.py:272, 285, 298, 308) and its autograd backward.
 *
 * out[s, :] = scale_s * sum_{t in [rowptr[s], rowptr[s+1])} w_t * src[idx[t], :]
 *   forward : src = x [n, f],           idx = col (source node of each sorted edge),
 *             scale_s = 1/len(s) (mean != 0), w_t = 1 (edge_w == NULL); out = H viewed [n*r_active, f]
 *   backward: src = dH viewed [n*r_active, f], idx = tseg (segment of each source-sorted edge),
 *             scale_s = 1, w_t = edge_w[idx[t]] = inv_cnt[segment];      out = dX [n, f]
 * Rows of src/out are `*_stride` elements apart.  Deterministic: fixed summation order, no atomics.
 * f % (16 / sizeof(dtype)) == 0 with 16-byte aligned rows takes the vector path; anything else
 * the scalar path.
 *
 * Skewed (power-law) graphs: segments longer than `long_threshold` edges are cut into chunks of that
 * many edges, each chunk is reduced by its own lane group into `partial` (fp32 [n_chunks, f]) and a third
 * kernel adds the partials of a segment in chunk order (still deterministic).  The plan arrays
 * (`long_seg` [n_long], `chunk_ptr` [n_long+1], `chunk_owner` [n_chunks]) are index data built once per
 * graph by the caller (or on the device: gmlm_split_plan_build, entries of -1 = unused slot); pass long_threshold = 0 /
 * n_long = 0 for no splitting.
 * ------------------------------------------------------------------------------------------- */
/* Split plan built on the device, for segment sets that change every step (the embedding-gradient segment sums: token /
 * position ids of the step's batch; replaces the scatter-add of hf BertEmbeddings' backward).  Arrays have CAPACITY size -
 * long_seg [cap_long], chunk_ptr [cap_long + 1], chunk_owner [cap_chunks], partial [cap_chunks, f] - from
 * gmlm_split_plan_capacity(num_items = rowptr[num_segments], long_threshold); unused slots hold -1 and are skipped by
 * gmlm_rgcn_mean_spmm, which is then called with n_long = cap_long, n_chunks = cap_chunks.  One launch, no host round trip,
 * deterministic. */
int gmlm_split_plan_capacity(int64_t num_items, int64_t long_threshold, int64_t* cap_long, int64_t* cap_chunks);
int gmlm_split_plan_build(const int32_t* rowptr, int64_t num_segments, int64_t num_items, int64_t long_threshold,
                          int32_t* long_seg, int32_t* chunk_ptr, int32_t* chunk_owner, gmlm_stream_t stream);

int gmlm_rgcn_mean_spmm(const void* src, int64_t src_rows, int64_t src_stride, const int32_t* rowptr,
                        const int32_t* idx, const float* edge_w, int mean, int64_t num_segments, int64_t f,
                        void* out, int64_t out_stride, int dtype, int64_t long_threshold, const int32_t* long_seg,
                        const int32_t* chunk_ptr, const int32_t* chunk_owner, int64_t n_long, int64_t n_chunks,
                        float* partial, gmlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K10  RGCN basis composition and its backward (PyG RGCNConv with num_bases; ctor sites main.py:189-203)
 *   W[r, :] = sum_b comp[r, b] * weight[b, :]     comp [r_active, num_bases] (rows of the relations that occur),
 *   weight [num_bases, cols = in*out], W [r_active, cols]; all fp32 (master parameters).
 *   backward (one pass): dweight[b, :] = sum_r comp[r, b] * dW[r, :],  dcomp[r, b] = <dW[r, :], weight[b, :]>.
 * r_active <= 5, num_bases <= 32, cols % 4 == 0.  Deterministic (fixed-order reductions).
 * ------------------------------------------------------------------------------------------- */
int gmlm_basis_compose_fwd(const float* comp, const float* weight, int r_active, int num_bases, int64_t cols,
                           float* w, gmlm_stream_t stream);
size_t gmlm_basis_compose_bwd_workspace_bytes(int r_active, int num_bases, int64_t cols);
int gmlm_basis_compose_bwd(const float* comp, const float* weight, const float* dw, int r_active, int num_bases,
                           int64_t cols, float* dweight, float* dcomp, void* workspace, size_t workspace_bytes,
                           gmlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4  GraphNorm (+ exact-erf GELU + dropout) forward / backward, single graph (batch = all rows)
 * replaces: torch_geometric.nn.GraphNorm.forward -> F.gelu -> nn.Dropout (main.py:273-275 ...)
 * ------------------------------------------------------------------------------------------- */
/* Column statistics of x [n, f] (`dtype`): s1[c] = sum_i (x[i,c] - shift[c]), s2[c] = sum_i (x[i,c]-shift[c])^2
 * (fp32).  shift may be NULL (= 0).  partial: workspace of gmlm_colstats_workspace_bytes(n, f).
 * In all K4 entries x, y, g and dx share ONE storage dtype (the GEMM output feeds the norm directly). */
size_t gmlm_colstats_workspace_bytes(int64_t n, int64_t f);
int gmlm_colstats(const void* x, int dtype, const float* shift, int64_t n, int64_t f, float* s1, float* s2,
                  void* workspace, size_t workspace_bytes, gmlm_stream_t stream);
/* mean[c], rstd[c] from (s1, s2, shift) over n_total rows:  mu = shift + s1/n; o = x - mu*ms;
 * var = E[o^2]; rstd = 1/sqrt(var + eps). */
int gmlm_graphnorm_finalize(const float* s1, const float* s2, const float* shift, const float* mean_scale,
                            int64_t n_total, int64_t f, float eps, float* mean, float* rstd, gmlm_stream_t stream);
/* y = dropout(gelu(weight * (x - mean*ms) * rstd + bias)); y stored as `dtype`; act != 0 applies GELU.
 * dropout: keep-probability scaling 1/(1-p), mask = hash(seed, element index) (replayable).
 * Every entry with a `seed` also takes `seed_dev`: NULL, or a device pointer to one uint64 that the kernel ADDS to `seed`
 * when it runs.  A captured hipGraph replays its arguments unchanged; pointing seed_dev at a counter that the graph itself
 * increments gives every replay fresh masks while forward / recompute / backward of one replay still agree. */
int gmlm_graphnorm_apply(const void* x, const float* mean, const float* rstd, const float* weight,
                         const float* bias, const float* mean_scale, int64_t n, int64_t f, int act,
                         float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* y, int dtype, gmlm_stream_t stream);
/* backward, pass 1: column sums needed by the closed form.  g = dL/dy (dtype), x = saved pre-norm input.
 * gs[0,c] = sum_i gz[i,c], gs[1,c] = sum_i gz[i,c] * ohat[i,c]   (gz = g through dropout and GELU) */
int gmlm_graphnorm_bwd_stats(const void* g, int dtype, const void* x, const float* mean, const float* rstd,
                             const float* weight, const float* bias, const float* mean_scale, int64_t n, int64_t f,
                             int act, float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* gs /* [2, f] */,
                             void* workspace, size_t workspace_bytes, gmlm_stream_t stream);
/* backward, pass 2: dx (`dtype`) and parameter grads from the (all-reduced) column sums over n_total rows. */
int gmlm_graphnorm_bwd_apply(const void* g, int dtype, const void* x, const float* mean, const float* rstd,
                             const float* weight, const float* bias, const float* mean_scale, const float* gs,
                             int64_t n, int64_t n_total, int64_t f, int act, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                             void* dx, float* dweight, float* dbias, float* dmean_scale, gmlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K6  fused bias + dropout + residual + LayerNorm (+ GELU) forward / backward (row-wise)
 * replaces: BertSelfOutput / BertOutput dense-epilogue (hf:modeling_bert.py:289-293, 347-351),
 *           BertEmbeddings LayerNorm (hf:...:105), MultiScaleFusion.layer_norm (main.py:180),
 *           fusion_network LayerNorm+GELU (main.py:238-239).
 * z = dropout(x + bias) + residual;  y = act(LN(z) * gamma + beta)
 * ------------------------------------------------------------------------------------------- */
int gmlm_bias_res_layernorm_fwd(const void* x, const float* bias, const void* residual, const float* gamma,
                                const float* beta, int64_t rows, int64_t f, float eps, int act, float dropout_p,
                                uint64_t seed, const uint64_t* seed_dev, void* y, float* mean, float* rstd, int dtype, gmlm_stream_t stream);
/* dz (written to dx; the residual branch receives the same dz; dbias = column sum of dx through dropout)
 * partial parameter grads are reduced inside: dgamma/dbeta/dbias [f] fp32 (zero-initialised by the call). */
size_t gmlm_layernorm_bwd_workspace_bytes(int64_t rows, int64_t f);
int gmlm_bias_res_layernorm_bwd(const void* dy, const void* x, const float* bias, const void* residual,
                                const float* gamma, const float* beta, const float* mean, const float* rstd,
                                int64_t rows, int64_t f, int act, float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* dx,
                                void* dresidual, float* dgamma, float* dbeta, float* dbias, int dtype,
                                void* workspace, size_t workspace_bytes, gmlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K5/K7  multi-head attention core, streaming softmax (flash style), forward / backward
 * replaces: K5 BertSelfAttention's attention_interface call softmax(QK^T*scale + padmask)V
 *           (hf:modeling_bert.py:188-201, eager form 111-136) with a per-sequence key length;
 *           K7 CrossAttention.forward's dense [1,8,N,N] softmax (main.py:159-163), no mask.
 * Layout: q [b, lq, h, d], k/v [b, lk, h, d] with row strides (elements) *_stride between
 * consecutive positions and h*d contiguous inside a position (so a fused [.., 3*h*d] QKV buffer
 * works without a copy); out [b, lq, h, d] dense; lse [b, h, lq] fp32 (log-sum-exp of scaled scores).
 * kv_len: int32 [b] valid key count per sequence (NULL = all lk keys valid); keys >= kv_len[b]
 * are masked out exactly like the additive -inf padding mask of create_bidirectional_mask.
 * d in {64, 96}.  MFMA: bf16 32x32x16 for GMLM_BF16, f32 32x32x2 for GMLM_F32.
 * Packed (variable-length) mode: cu_seqlens int32 [b+1] != NULL.  q, k, v, out are then [total_rows, h*d]
 * row blocks with sequence i owning rows [cu[i], cu[i+1]) of ALL of them (self-attention), lq = lk =
 * total_rows, max_len = longest sequence, kv_len = NULL, lse is [h, total_rows].  No padded token is ever
 * computed: this is how the text encoder runs (one packed batch of all active nodes).
 * dropout_p > 0 drops attention probabilities (after normalisation, scaled by 1/(1-p)) exactly like
 * nn.Dropout on `attn` in main.py:161 / hf eager attention; the mask is hash(seed, (b,h,q,key)) and
 * is regenerated, not stored, by the backward pass when given the same seed.
 * Short sequences: bf16, d = 64, every sequence <= 128 rows (lq, lk <= 128, or max_len <= 128 in packed mode) and
 * b*h >= 512 runs with Q / K / V (/ dO) of a work item resident in LDS; the backward is ONE launch (delta + dQ + dK/dV)
 * that forms delta = sum_k P dP from its own fp32 P, dP (it does not read `out`); everything else runs the streaming
 * forward and three backward launches (delta = rowsum(dO * O), dQ, dK/dV).
 * seq_groups (packed mode, optional; int32 [num_groups + 1], device): work item g of the short-sequence kernels is the run
 * of sequences [seq_groups[g], seq_groups[g+1]) - at most 13 sequences and 128 rows in total (the caller guarantees both;
 * rows past 128 would be ignored) - attended block-diagonally.  Ignored by the streaming kernels.
 * out_lo (bf16 only, optional): the forward also stores lo = bf16(O - bf16(O)); handed to the backward, delta is formed
 * from out + out_lo, i.e. with fp32-like accuracy.  With `out` alone delta carries the 2^-9 rounding of the stored
 * output, which does NOT cancel in dS = P (dP - delta) and dominates the query / key gradients whenever those are small
 * against |dO||O| (deep encoder layers).  The short-sequence kernels neither write nor need it.
 * ------------------------------------------------------------------------------------------- */
int gmlm_attention_fwd(const void* q, const void* k, const void* v, const int32_t* kv_len, int64_t b, int64_t h,
                       int64_t lq, int64_t lk, int64_t d, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                       float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* out, void* out_lo, float* lse,
                       int dtype, const int32_t* cu_seqlens, int64_t max_len, const int32_t* seq_groups, int64_t num_groups,
                       gmlm_stream_t stream);
/* packed mode: pass b = 1 (delta is [h, total_rows]) */
size_t gmlm_attention_bwd_workspace_bytes(int64_t b, int64_t h, int64_t lq, int64_t lk, int64_t d);
int gmlm_attention_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                       const float* lse, const int32_t* kv_len, int64_t b, int64_t h, int64_t lq, int64_t lk,
                       int64_t d, int64_t q_stride, int64_t k_stride, int64_t v_stride, float scale,
                       float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* dq, void* dk, void* dv, int64_t dq_stride,
                       int64_t dk_stride, int64_t dv_stride, int dtype, const int32_t* cu_seqlens, int64_t max_len,
                       void* workspace, size_t workspace_bytes,
                       float* dbias_partial /* [b or num_groups, 3*h*d] scratch */, float* dbias /* [3*h*d]: column sums of dq | dk | dv over all rows */,
                       const void* out_lo, const int32_t* seq_groups, int64_t num_groups, gmlm_stream_t stream);
/* dbias (optional, both pointers or neither; short-sequence path only, EINVAL otherwise): the bias gradient of a fused
 * QKV projection, sum over rows of [dq | dk | dv], formed inside the backward kernel from tiles it already holds
 * (three matrix-vector products on the MFMA pipe) instead of a separate pass over dqkv. */

/* BERT embedding sum (hf:modeling_bert.py:53-108 ahead of the LayerNorm): out[t] = word[tok[t]] + type0 + pos[pos_ids[t]],
 * fp32 tables, stored as `dtype` [rows, p]; replaces two gathers, two adds and a cast.  Its backward is a segment sum of the
 * output gradient by token id / position id (gmlm_segment_sort + gmlm_rgcn_mean_spmm with mean = 0).
 * F.embedding raises on an id outside its table; here an id outside [0, vocab) / [0, npos) is never read out of bounds (it is
 * clamped) and sets *bad_flag = 1 (device int32, nullable, NOT cleared by the call): the caller checks it when it next
 * synchronises, or validates its (static) id tensors once up front, as the Python host does. */
int gmlm_embed_sum_fwd(const float* word, const float* pos, const float* type0, const int64_t* tok, const int64_t* pos_ids,
                       int64_t rows, int64_t p, int64_t vocab, int64_t npos, void* out, int dtype, int32_t* bad_flag,
                       gmlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K8  attention-mask-weighted mean pooling + row scatter          (main.py:351-358)
 * out[node_idx[b], :] = sum_t hs[b,t,:] * [t < len[b]] / max(len[b], 1e-9)     (out fp32 [n, p])
 * -------------------------------------------------------------------------------------------
 * cu_seqlens != NULL: hs is packed [total_rows, p], sequence b = rows [cu[b], cu[b+1]) (len / l unused).
 */
int gmlm_meanpool_scatter_fwd(const void* hs, const int32_t* len, const int64_t* node_idx, int64_t b, int64_t l,
                              int64_t p, float* out, int dtype, const int32_t* cu_seqlens, gmlm_stream_t stream);
int gmlm_meanpool_scatter_bwd(const float* dout, const int32_t* len, const int64_t* node_idx, int64_t b, int64_t l,
                              int64_t p, void* dhs, int dtype, const int32_t* cu_seqlens, gmlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K9  soft-mask row blend                                           (main.py:92-99)
 * out[i,:] = mask[i] ? (1-beta)*x[i,:] + beta*token : x[i,:]     (out row stride out_stride, dtype out)
 * bwd: dtoken[c] = beta * sum_{i: mask[i]} dout[i,c]
 * ------------------------------------------------------------------------------------------- */
int gmlm_softmask_blend_fwd(const float* x, const uint8_t* mask, const float* token, float beta, int64_t n,
                            int64_t f, void* out, int64_t out_stride, int dtype, gmlm_stream_t stream);
int gmlm_softmask_blend_bwd(const float* dout, int64_t dout_stride, const uint8_t* mask, float beta, int64_t n,
                            int64_t f, float* dtoken, void* workspace, size_t workspace_bytes, gmlm_stream_t stream);

/* GELU (exact erf) + dropout elementwise with optional bias, used for BertIntermediate
 * (hf:modeling_bert.py:333-336) and classifier.1 (main.py:245): y = dropout(gelu(x + bias)). */
int gmlm_bias_gelu_fwd(const void* x, const float* bias, int64_t rows, int64_t f, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                       void* y, int dtype, gmlm_stream_t stream);
/* dx = dy * dropout_mask * gelu'(x + bias); dbias (nullable) [f] = column sums of dx, accumulated in fp32
 * inside the same pass (workspace: gmlm_bias_gelu_bwd_workspace_bytes). */
size_t gmlm_bias_gelu_bwd_workspace_bytes(int64_t rows, int64_t f, int dtype);
int gmlm_bias_gelu_bwd(const void* dy, const void* x, const float* bias, int64_t rows, int64_t f, float dropout_p,
                       uint64_t seed, const uint64_t* seed_dev, void* dx, float* dbias, int dtype, void* workspace, size_t workspace_bytes,
                       gmlm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GMLM_HIP_H */
