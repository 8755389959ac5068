// K2/K3 — relation-segmented CSR mean aggregation ("SpMM") and its transpose, gfx950.
// Reference: PyG RGCNConv.propagate(aggr='mean') per relation (call sites main.py:272,285,298,308)
// and its autograd backward.
//
//   out[s, :] = scale_s * sum_{t in [rowptr[s], rowptr[s+1])} w_t * src[idx[t], :]
//
// Design (HBM/L2-bound gather-reduce, no atomics, fixed summation order => deterministic):
//  * a group of L lanes (power of two, <= 64) owns one (segment, column-tile); each lane keeps CH
//    16-byte chunks of the row in registers and streams the segment's source rows through them,
//    4 edges in flight per lane (4*CH independent 16-B loads) to cover L2/MALL/HBM latency;
//  * consecutive lanes read consecutive 16-B chunks of one source row: every wave-instruction
//    touches L*16 contiguous bytes per row (coalesced row reads);
//  * column tiles: the row is cut into `ntile` tiles and tile = blockIdx.x % ntile.  Blocks are
//    dealt round-robin over the 8 XCDs, so with ntile == 8 each XCD's private 4 MiB L2 only ever
//    sees 1/8 of the columns of `src` (a speed heuristic only; any placement is correct).  The
//    host picks ntile = 8 when src is small enough to be L2-resident that way, else 1 so that each
//    gather is one long contiguous row read (HBM-friendly);
//  * empty segments write zeros (PyG mean of an empty neighbourhood = 0).
#include "common.hpp"

namespace gmlm {

// acc[c][v] += sum_{t in [beg, end)} w_t * src[idx[t], chunk c of this lane]; 4 edges in flight per lane
template <typename T, int CH, bool EDGE_W>
__device__ __forceinline__ void accumulate_rows(float (&acc)[CH][Store<T>::kVec], const T* __restrict__ src, int64_t src_stride,
                                                const int32_t* __restrict__ idx, const float* __restrict__ edge_w, int beg,
                                                int end, int chunk0, int nch, int L, int lane_g) {
  constexpr int V = Store<T>::kVec;
  int t = beg;
  for (; t + 4 <= end; t += 4) {
    int j[4];
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) j[u] = idx[t + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = EDGE_W ? edge_w[j[u]] : 1.f;
    uint4 r[4][CH];   // raw 16-byte chunks; converted to fp32 only when accumulated
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int ch = c * L + lane_g;
        if (ch < nch)
          r[u][c] = *reinterpret_cast<const uint4*>(src + (int64_t)j[u] * src_stride + (int64_t)(chunk0 + ch) * V);
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (c * L + lane_g < nch) {
          float x[V];
          Store<T>::unpack(r[u][c], x);
#pragma unroll
          for (int v = 0; v < V; ++v) acc[c][v] = EDGE_W ? fmaf(w[u], x[v], acc[c][v]) : acc[c][v] + x[v];
        }
  }
  for (; t < end; ++t) {
    const int j = idx[t];
    const float w = EDGE_W ? edge_w[j] : 1.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = c * L + lane_g;
      if (ch < nch) {
        float r[V];
        Store<T>::ldv(src + (int64_t)j * src_stride + (int64_t)(chunk0 + ch) * V, r);
#pragma unroll
        for (int v = 0; v < V; ++v) acc[c][v] = EDGE_W ? fmaf(w, r[v], acc[c][v]) : acc[c][v] + r[v];
      }
    }
  }
}

struct SplitPlan {            // long segments are cut into chunks of `thresh` edges (power-law graphs)
  int thresh;                 // 0 = no splitting
  const int32_t* long_seg;    // [n_long] segment ids with more than `thresh` edges
  const int32_t* chunk_ptr;   // [n_long + 1] first chunk of each long segment
  const int32_t* chunk_owner; // [n_chunks] index into long_seg
  int64_t n_long, n_chunks;
  float* partial;             // [n_chunks, f] fp32, unscaled partial sums
};

// MODE 0: one group per segment (long segments skipped when a split plan is given)
// MODE 1: one group per chunk of a long segment -> fp32 partial row
template <typename T, int CH, bool EDGE_W, int MODE>
__global__ __launch_bounds__(256) void seg_reduce_vec_kernel(
    const T* __restrict__ src, int64_t src_stride, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx,
    const float* __restrict__ edge_w, int mean, int64_t num_items, int chunks_per_tile, int ntile, int log2_l,
    T* __restrict__ out, int64_t out_stride, int f_chunks, SplitPlan sp) {
  constexpr int V = Store<T>::kVec;
  const int L = 1 << log2_l;
  const int groups_per_block = 256 >> log2_l;
  const int lane_g = threadIdx.x & (L - 1);
  const int group = threadIdx.x >> log2_l;
  const int tile = blockIdx.x % ntile;
  const int64_t seg_blocks = gridDim.x / ntile;
  const int chunk0 = tile * chunks_per_tile;
  int nch = f_chunks - chunk0;            // chunks in this tile
  if (nch > chunks_per_tile) nch = chunks_per_tile;

  for (int64_t s = (int64_t)(blockIdx.x / ntile) * groups_per_block + group; s < num_items;
       s += seg_blocks * groups_per_block) {
    int beg, end;
    if (MODE == 0) {
      beg = rowptr[s];
      end = rowptr[s + 1];
      if (sp.thresh > 0 && end - beg > sp.thresh) continue;      // handled by the chunk + combine kernels
    } else {
      const int j = sp.chunk_owner[s];
      if (j < 0) continue;                                         // unused slot of a capacity-sized plan (gmlm_split_plan_build)
      const int seg = sp.long_seg[j];
      beg = rowptr[seg] + (int)(s - sp.chunk_ptr[j]) * sp.thresh;
      end = rowptr[seg + 1];
      if (end > beg + sp.thresh) end = beg + sp.thresh;
    }
    float acc[CH][V];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[c][v] = 0.f;
    accumulate_rows<T, CH, EDGE_W>(acc, src, src_stride, idx, edge_w, beg, end, chunk0, nch, L, lane_g);
    if (MODE == 0) {
      const float scale = mean ? 1.f / (float)(end - beg > 1 ? end - beg : 1) : 1.f;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int ch = c * L + lane_g;
        if (ch < nch) {
#pragma unroll
          for (int v = 0; v < V; ++v) acc[c][v] *= scale;
          Store<T>::stv(out + s * out_stride + (int64_t)(chunk0 + ch) * V, acc[c]);
        }
      }
    } else {
      float* prow = sp.partial + s * (int64_t)f_chunks * V;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int ch = c * L + lane_g;
        if (ch < nch)
#pragma unroll
          for (int v = 0; v < V; v += 4)
            *reinterpret_cast<float4*>(prow + (int64_t)(chunk0 + ch) * V + v) = make_float4(acc[c][v], acc[c][v + 1], acc[c][v + 2], acc[c][v + 3]);
      }
    }
  }
}

// sums the partial rows of each long segment in chunk order (deterministic), scales, stores
template <typename T>
__global__ __launch_bounds__(256) void seg_reduce_combine_kernel(const int32_t* __restrict__ rowptr, int mean, int64_t f,
                                                                  T* __restrict__ out, int64_t out_stride, SplitPlan sp) {
  const int64_t j = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= f) return;
  const int seg = sp.long_seg[j];
  if (seg < 0) return;                                             // unused slot of a capacity-sized plan
  float acc = 0.f;
  for (int k = sp.chunk_ptr[j]; k < sp.chunk_ptr[j + 1]; ++k) acc += sp.partial[(int64_t)k * f + c];
  const int len = rowptr[seg + 1] - rowptr[seg];
  Store<T>::st(out + (int64_t)seg * out_stride + c, mean ? acc / (float)len : acc);
}

// General path: any f / alignment.  One wave per (segment, 512-column tile), lane <-> column.
template <typename T, bool EDGE_W>
__global__ __launch_bounds__(256) void seg_reduce_scalar_kernel(
    const T* __restrict__ src, int64_t src_stride, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx,
    const float* __restrict__ edge_w, int mean, int64_t num_segments, int64_t f, int ntile, T* __restrict__ out,
    int64_t out_stride) {
  constexpr int CS = 8;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tile = blockIdx.x % ntile;
  const int64_t seg_blocks = gridDim.x / ntile;
  const int64_t col0 = (int64_t)tile * 64 * CS;
  for (int64_t s = (int64_t)(blockIdx.x / ntile) * 4 + wave; s < num_segments; s += seg_blocks * 4) {
    const int beg = rowptr[s], end = rowptr[s + 1];
    float acc[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) acc[c] = 0.f;
    for (int t = beg; t < end; ++t) {
      const int j = idx[t];
      const float w = EDGE_W ? edge_w[j] : 1.f;
      const T* row = src + (int64_t)j * src_stride;
#pragma unroll
      for (int c = 0; c < CS; ++c) {
        const int64_t col = col0 + c * 64 + lane;
        if (col < f) acc[c] = EDGE_W ? fmaf(w, Store<T>::ld(row + col), acc[c]) : acc[c] + Store<T>::ld(row + col);
      }
    }
    const float scale = mean ? 1.f / (float)(end - beg > 1 ? end - beg : 1) : 1.f;
#pragma unroll
    for (int c = 0; c < CS; ++c) {
      const int64_t col = col0 + c * 64 + lane;
      if (col < f) Store<T>::st(out + s * out_stride + col, acc[c] * scale);
    }
  }
}

template <typename T>
static int launch_spmm(const void* src_, int64_t src_rows, int64_t src_stride, const int32_t* rowptr, const int32_t* idx,
                       const float* edge_w, int mean, int64_t num_segments, int64_t f, void* out_, int64_t out_stride,
                       SplitPlan sp, hipStream_t st) {
  const T* src = static_cast<const T*>(src_);
  T* out = static_cast<T*>(out_);
  constexpr int V = Store<T>::kVec;
  const bool vec_ok = (f % V == 0) && (src_stride % V == 0) && (out_stride % V == 0) && aligned16(src) && aligned16(out);
  const int64_t src_bytes = src_rows * f * (int64_t)sizeof(T);
  if (vec_ok) {
    const int f_chunks = (int)(f / V);
    // XCD column tiling only while one tile of src fits an XCD's 4 MiB L2 with room to spare
    int ntile = (src_bytes <= (24ll << 20) && f_chunks >= 64) ? 8 : 1;
    int cpt = (int)cdiv(f_chunks, ntile);
    while ((int64_t)cpt > 256) { ntile *= 2; cpt = (int)cdiv(f_chunks, ntile); }  // <= 64 lanes x 4 chunks
    ntile = (int)cdiv(f_chunks, cpt);
    int log2_l = 0;
    while ((1 << log2_l) < cpt && log2_l < 6) ++log2_l;
    const int L = 1 << log2_l;
    const int ch = (int)cdiv(cpt, L);
    const int groups_per_block = 256 / L;
    const int64_t seg_blocks = cdiv(num_segments, groups_per_block);
    const int64_t cap = 256 * 16 / ntile > 0 ? 256 * 16 / ntile : 1;
    const int grid = (int)((seg_blocks < cap ? seg_blocks : cap) * ntile);
    const int64_t chunk_blocks = cdiv(sp.n_chunks, groups_per_block);
    const int cgrid = (int)((chunk_blocks < cap ? chunk_blocks : cap) * ntile);
#define GMLM_SPMM_LAUNCH(CHV)                                                                                          \
  do {                                                                                                                 \
    if (edge_w)                                                                                                        \
      seg_reduce_vec_kernel<T, CHV, true, 0><<<grid, 256, 0, st>>>(src, src_stride, rowptr, idx, edge_w, mean,         \
                                                                   num_segments, cpt, ntile, log2_l, out, out_stride,  \
                                                                   f_chunks, sp);                                      \
    else                                                                                                               \
      seg_reduce_vec_kernel<T, CHV, false, 0><<<grid, 256, 0, st>>>(src, src_stride, rowptr, idx, edge_w, mean,        \
                                                                    num_segments, cpt, ntile, log2_l, out, out_stride, \
                                                                    f_chunks, sp);                                     \
    if (sp.thresh > 0 && sp.n_chunks > 0) {                                                                            \
      if (edge_w)                                                                                                      \
        seg_reduce_vec_kernel<T, CHV, true, 1><<<cgrid, 256, 0, st>>>(src, src_stride, rowptr, idx, edge_w, mean,      \
                                                                      sp.n_chunks, cpt, ntile, log2_l, out,            \
                                                                      out_stride, f_chunks, sp);                       \
      else                                                                                                             \
        seg_reduce_vec_kernel<T, CHV, false, 1><<<cgrid, 256, 0, st>>>(src, src_stride, rowptr, idx, edge_w, mean,     \
                                                                       sp.n_chunks, cpt, ntile, log2_l, out,           \
                                                                       out_stride, f_chunks, sp);                      \
    }                                                                                                                  \
  } while (0)
    switch (ch) {
      case 1: GMLM_SPMM_LAUNCH(1); break;
      case 2: GMLM_SPMM_LAUNCH(2); break;
      case 3: GMLM_SPMM_LAUNCH(3); break;
      default: GMLM_SPMM_LAUNCH(4); break;
    }
#undef GMLM_SPMM_LAUNCH
    if (sp.thresh > 0 && sp.n_long > 0) {
      GMLM_LAUNCH_CHECK();
      seg_reduce_combine_kernel<T><<<dim3((unsigned)cdiv(f, 256), (unsigned)sp.n_long), 256, 0, st>>>(rowptr, mean, f, out,
                                                                                                       out_stride, sp);
    }
  } else {
    sp.thresh = 0;   // the scalar path (unaligned rows) does not split
    const int ntile = (int)cdiv(f, 512);
    const int64_t seg_blocks = cdiv(num_segments, 4);
    const int64_t cap = 4096 / ntile > 0 ? 4096 / ntile : 1;
    const int grid = (int)((seg_blocks < cap ? seg_blocks : cap) * ntile);
    if (edge_w)
      seg_reduce_scalar_kernel<T, true><<<grid, 256, 0, st>>>(src, src_stride, rowptr, idx, edge_w, mean, num_segments, f,
                                                              ntile, out, out_stride);
    else
      seg_reduce_scalar_kernel<T, false><<<grid, 256, 0, st>>>(src, src_stride, rowptr, idx, edge_w, mean, num_segments,
                                                               f, ntile, out, out_stride);
  }
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

}  // namespace gmlm

using namespace gmlm;

extern "C" int gmlm_rgcn_mean_spmm(const void* src, int64_t src_rows, int64_t src_stride, const int32_t* rowptr,
                                   const int32_t* idx, const float* edge_w, int mean, int64_t num_segments, int64_t f,
                                   void* out, int64_t out_stride, int dtype, int64_t long_threshold, const int32_t* long_seg,
                                   const int32_t* chunk_ptr, const int32_t* chunk_owner, int64_t n_long, int64_t n_chunks,
                                   float* partial, gmlm_stream_t stream) {
  GMLM_REQUIRE(num_segments >= 0 && f > 0 && src_rows >= 0, "rgcn_mean_spmm: bad sizes (segments=%ld f=%ld)",
               (long)num_segments, (long)f);
  GMLM_REQUIRE(src_stride >= f && out_stride >= f, "rgcn_mean_spmm: row stride smaller than f");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "rgcn_mean_spmm: unsupported dtype %d", dtype);
  if (num_segments == 0) return GMLM_OK;
  GMLM_REQUIRE(rowptr && out && (src || src_rows == 0), "rgcn_mean_spmm: null pointer");
  SplitPlan sp{};
  if (long_threshold > 0 && n_long > 0) {
    GMLM_REQUIRE(long_seg && chunk_ptr && chunk_owner && partial && n_chunks > 0 && n_long <= 65535 * 1024ll,
                 "rgcn_mean_spmm: incomplete split plan");
    GMLM_REQUIRE(n_long <= 65535, "rgcn_mean_spmm: more than 65535 long segments; raise long_threshold");
    sp.thresh = (int)long_threshold; sp.long_seg = long_seg; sp.chunk_ptr = chunk_ptr; sp.chunk_owner = chunk_owner;
    sp.n_long = n_long; sp.n_chunks = n_chunks; sp.partial = partial;
  }
  hipStream_t st = as_stream(stream);
  if (dtype == GMLM_F32)
    return launch_spmm<float>(src, src_rows, src_stride, rowptr, idx, edge_w, mean, num_segments, f, out, out_stride, sp, st);
  return launch_spmm<bf16_t>(src, src_rows, src_stride, rowptr, idx, edge_w, mean, num_segments, f, out, out_stride, sp, st);
}
