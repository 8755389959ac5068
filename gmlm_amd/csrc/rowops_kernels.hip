// K8 masked mean pooling + row scatter (main.py:351-358), K9 soft-mask row blend (main.py:92-99),
// bias+GELU(+dropout) elementwise (hf:modeling_bert.py:333-336, main.py:245).  HBM-bound streaming
// kernels: thread <-> column, rows stream through, coalesced; reductions have a fixed order.
#include "colreduce.hpp"

namespace gmlm {

// block (b, column tile of 256): out[node_idx[b], c] = sum_{t < len[b]} hs[b, t, c] / max(len, 1e-9)
template <typename T>
__global__ __launch_bounds__(256) void meanpool_fwd_kernel(const T* __restrict__ hs, const int32_t* __restrict__ len,
                                                            const int64_t* __restrict__ node_idx, int64_t l, int64_t p,
                                                            float* __restrict__ out, const int32_t* __restrict__ cu) {
  const int64_t b = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= p) return;
  int n = cu ? cu[b + 1] - cu[b] : len[b];
  if (!cu && n > l) n = (int)l;
  const T* base = hs + (cu ? (int64_t)cu[b] : b * l) * p + c;
  float acc = 0.f;
  for (int t = 0; t < n; ++t) acc += Store<T>::ld(base + (int64_t)t * p);
  out[node_idx[b] * p + c] = acc / fmaxf((float)n, 1e-9f);
}

template <typename T>
__global__ __launch_bounds__(256) void meanpool_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ len,
                                                            const int64_t* __restrict__ node_idx, int64_t l, int64_t p,
                                                            T* __restrict__ dhs, const int32_t* __restrict__ cu) {
  const int64_t b = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= p) return;
  int n = cu ? cu[b + 1] - cu[b] : len[b];
  if (!cu && n > l) n = (int)l;
  const float g = dout[node_idx[b] * p + c] / fmaxf((float)n, 1e-9f);
  T* base = dhs + (cu ? (int64_t)cu[b] : b * l) * p + c;
  const int64_t rows = cu ? n : l;
  for (int64_t t = 0; t < rows; ++t) Store<T>::st(base + t * p, t < n ? g : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void softmask_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                            const float* __restrict__ tok, float beta, int64_t n, int64_t f,
                                                            T* __restrict__ out, int64_t out_stride) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= out_stride) return;
  const float tk = c < f ? beta * tok[c] : 0.f;
  const float om = 1.f - beta;
  for (int64_t r = blockIdx.y; r < n; r += gridDim.y) {
    float v = 0.f;                                   // columns [f, out_stride) are zero padding
    if (c < f) {
      v = x[r * f + c];
      if (mask[r]) v = om * v + tk;
    }
    Store<T>::st(out + r * out_stride + c, v);
  }
}

struct SoftmaskBwdFn {
  const float* dout;
  int64_t stride;
  const uint8_t* mask;
  __device__ void operator()(int64_t r, int64_t c, float (&v)[1]) const { v[0] = mask[r] ? dout[r * stride + c] : 0.f; }
};

// Threads own V fixed columns (blockDim.x chunk-columns per block, picked on the host so no lane idles) and
// stream rows blockIdx.y, +gridDim.y, ...; two rows are in flight per iteration.
template <typename T, bool DROP>
__global__ __launch_bounds__(256) void bias_gelu_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                             int64_t rows, int64_t f, uint32_t thresh, float keep_scale,
                                                             SeedArg seed_a, T* __restrict__ y) {
  const uint64_t seed = seed_a.get();
  constexpr int V = Store<T>::kVec;
  const int64_t nch = f / V;
  const int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= nch) return;
  float bv[V];
#pragma unroll
  for (int v = 0; v < V; ++v) bv[v] = bias ? bias[ch * V + v] : 0.f;
  const int64_t gy = gridDim.y;
  for (int64_t r = blockIdx.y; r < rows; r += 2 * gy) {
    const bool two = r + gy < rows;
    const int64_t off0 = r * f + ch * V, off1 = two ? off0 + gy * f : off0;
    float a[2][V];
    Store<T>::ldv(x + off0, a[0]);
    Store<T>::ldv(x + off1, a[1]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t off = u ? off1 : off0;
      const uint32_t kb = DROP ? dropout_keep_bits<V>(seed, (uint64_t)off, thresh) : 0u;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        a[u][v] = gelu_fwd_t<T>(a[u][v] + bv[v]);
        if (DROP) a[u][v] = ((kb >> v) & 1u) ? a[u][v] * keep_scale : 0.f;
      }
    }
    Store<T>::stv(y + off0, a[0]);
    if (two) Store<T>::stv(y + off1, a[1]);
  }
}

// dbias partials ride along: a thread owns V fixed columns, so it keeps their column sums in registers
// over the rows it streams and writes one partial row per blockIdx.y (summed by rows_sum_kernel).
template <typename T, bool DROP>
__global__ __launch_bounds__(256) void bias_gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ bias, int64_t rows, int64_t f,
                                                             uint32_t thresh, float keep_scale, SeedArg seed_a,
                                                             T* __restrict__ dx, float* __restrict__ dbias_partial) {
  const uint64_t seed = seed_a.get();
  constexpr int V = Store<T>::kVec;
  const int64_t nch = f / V;
  const int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= nch) return;
  float bv[V], acc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) { bv[v] = bias ? bias[ch * V + v] : 0.f; acc[v] = 0.f; }
  const int64_t gy = gridDim.y;
  for (int64_t r = blockIdx.y; r < rows; r += 2 * gy) {
    const bool two = r + gy < rows;
    const int64_t off0 = r * f + ch * V, off1 = two ? off0 + gy * f : off0;
    float a[2][V], g[2][V];
    Store<T>::ldv(x + off0, a[0]);
    Store<T>::ldv(dy + off0, g[0]);
    Store<T>::ldv(x + off1, a[1]);
    Store<T>::ldv(dy + off1, g[1]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t off = u ? off1 : off0;
      const uint32_t kb = DROP ? dropout_keep_bits<V>(seed, (uint64_t)off, thresh) : 0u;
      const bool count = u == 0 || two;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        float gv = g[u][v];
        if (DROP) gv = ((kb >> v) & 1u) ? gv * keep_scale : 0.f;
        gv *= gelu_grad_t<T>(a[u][v] + bv[v]);
        g[u][v] = gv;
        acc[v] += count ? gv : 0.f;
      }
    }
    Store<T>::stv(dx + off0, g[0]);
    if (two) Store<T>::stv(dx + off1, g[1]);
  }
  if (dbias_partial) {
#pragma unroll
    for (int v = 0; v < V; ++v) dbias_partial[(int64_t)blockIdx.y * f + ch * V + v] = acc[v];
  }
}

template <typename T>
struct ColSumFn {
  const T* a;
  int64_t f;
  __device__ void operator()(int64_t r, int64_t c, float (&v)[1]) const { v[0] = Store<T>::ld(a + r * f + c); }
};

static inline dim3 stream_grid(int64_t rows, int64_t cols_units) {
  const int ct = (int)cdiv(cols_units, 256);
  int64_t ry = cdiv(4096, ct);
  if (ry > rows) ry = rows;
  if (ry < 1) ry = 1;
  return dim3(ct, (unsigned)ry);
}

// bias+GELU geometry: chunk-columns per block = the largest of 256/192/128/64 that divides the chunk count (no
// idle lanes, e.g. 384 chunks -> 2 x 192), row lanes in y so that the grid holds ~`target` blocks.
struct BgGeom { dim3 grid; unsigned block; };
static inline BgGeom bg_geom(int64_t rows, int64_t chunks, int64_t target) {
  unsigned bx = 256;
  for (unsigned c : {256u, 192u, 128u, 64u}) if (chunks % c == 0) { bx = c; break; }
  const int ct = (int)cdiv(chunks, bx);
  int64_t ry = cdiv(target, ct);
  if (ry > rows) ry = rows;
  if (ry < 1) ry = 1;
  return BgGeom{dim3(ct, (unsigned)ry), bx};
}

}  // namespace gmlm

using namespace gmlm;

extern "C" int gmlm_meanpool_scatter_fwd(const void* hs, const int32_t* len, const int64_t* node_idx, int64_t b, int64_t l,
                                         int64_t p, float* out, int dtype, const int32_t* cu_seqlens, gmlm_stream_t stream) {
  GMLM_REQUIRE(b >= 0 && l >= 0 && p > 0, "meanpool_scatter_fwd: bad sizes");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "meanpool_scatter_fwd: unsupported dtype");
  if (b == 0) return GMLM_OK;
  GMLM_REQUIRE((len || cu_seqlens) && node_idx && out && (hs || l == 0), "meanpool_scatter_fwd: null pointer");
  GMLM_REQUIRE(b <= 65535, "meanpool_scatter_fwd: micro-batch %ld > 65535", (long)b);
  dim3 grid((unsigned)cdiv(p, 256), (unsigned)b);
  if (dtype == GMLM_F32)
    meanpool_fwd_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)hs, len, node_idx, l, p, out, cu_seqlens);
  else
    meanpool_fwd_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((const bf16_t*)hs, len, node_idx, l, p, out, cu_seqlens);
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

extern "C" int gmlm_meanpool_scatter_bwd(const float* dout, const int32_t* len, const int64_t* node_idx, int64_t b,
                                         int64_t l, int64_t p, void* dhs, int dtype, const int32_t* cu_seqlens,
                                         gmlm_stream_t stream) {
  GMLM_REQUIRE(b >= 0 && l >= 0 && p > 0, "meanpool_scatter_bwd: bad sizes");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "meanpool_scatter_bwd: unsupported dtype");
  if (b == 0 || (l == 0 && !cu_seqlens)) return GMLM_OK;
  GMLM_REQUIRE((len || cu_seqlens) && node_idx && dout && dhs, "meanpool_scatter_bwd: null pointer");
  GMLM_REQUIRE(b <= 65535, "meanpool_scatter_bwd: micro-batch %ld > 65535", (long)b);
  dim3 grid((unsigned)cdiv(p, 256), (unsigned)b);
  if (dtype == GMLM_F32)
    meanpool_bwd_kernel<float><<<grid, 256, 0, as_stream(stream)>>>(dout, len, node_idx, l, p, (float*)dhs, cu_seqlens);
  else
    meanpool_bwd_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>(dout, len, node_idx, l, p, (bf16_t*)dhs, cu_seqlens);
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

extern "C" int gmlm_softmask_blend_fwd(const float* x, const uint8_t* mask, const float* token, float beta, int64_t n,
                                       int64_t f, void* out, int64_t out_stride, int dtype, gmlm_stream_t stream) {
  GMLM_REQUIRE(n >= 0 && f > 0 && out_stride >= f, "softmask_blend_fwd: bad sizes");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "softmask_blend_fwd: unsupported dtype");
  if (n == 0) return GMLM_OK;
  GMLM_REQUIRE(x && mask && token && out, "softmask_blend_fwd: null pointer");
  dim3 grid = stream_grid(n, out_stride);
  if (dtype == GMLM_F32)
    softmask_fwd_kernel<float><<<grid, 256, 0, as_stream(stream)>>>(x, mask, token, beta, n, f, (float*)out, out_stride);
  else
    softmask_fwd_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>(x, mask, token, beta, n, f, (bf16_t*)out, out_stride);
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

extern "C" int gmlm_softmask_blend_bwd(const float* dout, int64_t dout_stride, const uint8_t* mask, float beta, int64_t n,
                                       int64_t f, float* dtoken, void* workspace, size_t workspace_bytes,
                                       gmlm_stream_t stream) {
  GMLM_REQUIRE(n >= 0 && f > 0 && dout_stride >= f && dtoken, "softmask_blend_bwd: bad arguments");
  GMLM_REQUIRE(n == 0 || (dout && mask), "softmask_blend_bwd: null pointer");
  return col_reduce<1>(n, f, SoftmaskBwdFn{dout, dout_stride, mask}, dtoken, workspace, workspace_bytes, as_stream(stream),
                       beta);
}

static int bg_check(const char* who, int64_t rows, int64_t f, int dtype, float p) {
  GMLM_REQUIRE(rows >= 0 && f > 0, "%s: bad sizes", who);
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "%s: unsupported dtype", who);
  GMLM_REQUIRE(f % (dtype == GMLM_F32 ? 4 : 8) == 0, "%s: width %ld must be a multiple of %d", who, (long)f, dtype == GMLM_F32 ? 4 : 8);
  GMLM_REQUIRE(p >= 0.f && p < 1.f, "%s: dropout_p must be in [0,1)", who);
  return GMLM_OK;
}

extern "C" int gmlm_bias_gelu_fwd(const void* x, const float* bias, int64_t rows, int64_t f, float dropout_p, uint64_t seed_host, const uint64_t* seed_dev,
                                  void* y, int dtype, gmlm_stream_t stream) {
  const SeedArg seed{seed_host, seed_dev};
  int rc = bg_check("bias_gelu_fwd", rows, f, dtype, dropout_p);
  if (rc != GMLM_OK) return rc;
  if (rows == 0) return GMLM_OK;
  GMLM_REQUIRE(x && y && aligned16(x) && aligned16(y), "bias_gelu_fwd: null or misaligned pointer");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = dropout_keep_scale(th);
  const BgGeom ge = bg_geom(rows, f / (dtype == GMLM_F32 ? 4 : 8), 4096);
#define FW(T, D) bias_gelu_fwd_kernel<T, D><<<ge.grid, ge.block, 0, as_stream(stream)>>>((const T*)x, bias, rows, f, th, ks, seed, (T*)y)
  if (dtype == GMLM_F32) { if (th) FW(float, true); else FW(float, false); }
  else { if (th) FW(bf16_t, true); else FW(bf16_t, false); }
#undef FW
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

static inline BgGeom bias_gelu_bwd_geom(int64_t rows, int64_t chunks) { return bg_geom(rows, chunks, 2048); }

extern "C" size_t gmlm_bias_gelu_bwd_workspace_bytes(int64_t rows, int64_t f, int dtype) {
  const BgGeom g = bias_gelu_bwd_geom(rows, f / (dtype == GMLM_F32 ? 4 : 8));
  return (size_t)g.grid.y * f * sizeof(float);
}

extern "C" int gmlm_bias_gelu_bwd(const void* dy, const void* x, const float* bias, int64_t rows, int64_t f, float dropout_p,
                                  uint64_t seed_host, const uint64_t* seed_dev, void* dx, float* dbias, int dtype, void* workspace, size_t workspace_bytes,
                                  gmlm_stream_t stream) {
  const SeedArg seed{seed_host, seed_dev};
  int rc = bg_check("bias_gelu_bwd", rows, f, dtype, dropout_p);
  if (rc != GMLM_OK) return rc;
  hipStream_t st = as_stream(stream);
  if (rows == 0) {
    if (dbias) GMLM_HIP(zero_async(dbias, sizeof(float) * f, st));
    return GMLM_OK;
  }
  GMLM_REQUIRE(dy && x && dx && aligned16(dy) && aligned16(x) && aligned16(dx), "bias_gelu_bwd: null or misaligned pointer");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = dropout_keep_scale(th);
  const BgGeom ge = bias_gelu_bwd_geom(rows, f / (dtype == GMLM_F32 ? 4 : 8));
  float* partial = nullptr;
  if (dbias) {
    GMLM_REQUIRE(workspace && workspace_bytes >= gmlm_bias_gelu_bwd_workspace_bytes(rows, f, dtype), "bias_gelu_bwd: workspace too small");
    partial = static_cast<float*>(workspace);
  }
#define BW(T, D) bias_gelu_bwd_kernel<T, D><<<ge.grid, ge.block, 0, st>>>((const T*)dy, (const T*)x, bias, rows, f, th, ks, seed, (T*)dx, partial)
  if (dtype == GMLM_F32) { if (th) BW(float, true); else BW(float, false); }
  else { if (th) BW(bf16_t, true); else BW(bf16_t, false); }
#undef BW
  GMLM_LAUNCH_CHECK();
  if (dbias) {
    rows_sum_kernel<<<(unsigned)cdiv(f, 32), 256, 0, st>>>(partial, (int)ge.grid.y, f, dbias);
    GMLM_LAUNCH_CHECK();
  }
  return GMLM_OK;
}

// ------------------------------------------------------------------------------------------------
// BERT embedding sum: out[t, :] = word[tok[t], :] + type0[:] + pos[pos_ids[t], :]   (hf:modeling_bert.py:53-108 before the
// LayerNorm; one pass instead of two gathers, two adds and a cast).  fp32 tables, fp32 adds in that order (the same
// values the three torch ops produce), stored as `dtype`.  One wave handles a row: 64 lanes x float4 = 256 columns per step.
// ------------------------------------------------------------------------------------------------
namespace gmlm {
template <typename T>
__global__ __launch_bounds__(256) void embed_sum_fwd_kernel(const float* __restrict__ word, const float* __restrict__ pos,
                                                             const float* __restrict__ type0, const int64_t* __restrict__ tok,
                                                             const int64_t* __restrict__ pos_ids, int64_t rows, int p, int64_t vocab,
                                                             int64_t npos, T* __restrict__ out, int32_t* __restrict__ bad_flag) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + w; r < rows; r += (int64_t)gridDim.x * 4) {
    int64_t tv = tok[r], pv = pos_ids[r];
    // F.embedding raises on an id outside its table; a kernel cannot: it never reads out of bounds (the id is clamped) and
    // REPORTS the violation through *bad_flag, which the caller inspects (wave-uniform condition: one lane writes)
    if ((tv < 0 || tv >= vocab || pv < 0 || pv >= npos) && bad_flag && lane == 0) *bad_flag = 1;
    tv = tv < 0 ? 0 : (tv >= vocab ? vocab - 1 : tv);
    pv = pv < 0 ? 0 : (pv >= npos ? npos - 1 : pv);
    const float* wr = word + tv * p;
    const float* pr = pos + pv * p;
    for (int c = lane * 4; c < p; c += 256) {
      const float4 a = *reinterpret_cast<const float4*>(wr + c);
      const float4 t4 = *reinterpret_cast<const float4*>(type0 + c);
      const float4 b = *reinterpret_cast<const float4*>(pr + c);
      const float o[4] = {(a.x + t4.x) + b.x, (a.y + t4.y) + b.y, (a.z + t4.z) + b.z, (a.w + t4.w) + b.w};
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(out + r * p + c) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        uint2 v;
        v.x = pack_bf16x2(o[0], o[1]);
        v.y = pack_bf16x2(o[2], o[3]);
        *reinterpret_cast<uint2*>(out + r * p + c) = v;
      }
    }
  }
}
}  // namespace gmlm

extern "C" int gmlm_embed_sum_fwd(const float* word, const float* pos, const float* type0, const int64_t* tok,
                                  const int64_t* pos_ids, int64_t rows, int64_t p, int64_t vocab, int64_t npos, void* out,
                                  int dtype, int32_t* bad_flag, gmlm_stream_t stream) {
  using namespace gmlm;
  GMLM_REQUIRE(rows >= 0 && p > 0 && p % 4 == 0 && vocab > 0 && npos > 0, "embed_sum_fwd: bad sizes (p must be a multiple of 4)");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "embed_sum_fwd: dtype must be GMLM_F32 or GMLM_BF16");
  if (rows == 0) return GMLM_OK;
  GMLM_REQUIRE(word && pos && type0 && tok && pos_ids && out, "embed_sum_fwd: null pointer");
  GMLM_REQUIRE(aligned16(word) && aligned16(pos) && aligned16(type0) && aligned16(out), "embed_sum_fwd: tables / output must be 16-byte aligned");
  const unsigned grid = (unsigned)(cdiv(rows, 4) < 65536 ? cdiv(rows, 4) : 65536);
  if (dtype == GMLM_F32)
    embed_sum_fwd_kernel<float><<<grid, 256, 0, as_stream(stream)>>>(word, pos, type0, tok, pos_ids, rows, (int)p, vocab, npos, (float*)out, bad_flag);
  else
    embed_sum_fwd_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>(word, pos, type0, tok, pos_ids, rows, (int)p, vocab, npos, (bf16_t*)out, bad_flag);
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}
