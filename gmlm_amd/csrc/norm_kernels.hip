// K4 GraphNorm(+GELU+dropout) and K6 bias+dropout+residual+LayerNorm(+GELU), forward and backward.
// Reference: PyG GraphNorm -> F.gelu -> Dropout (main.py:273-275 and the three sibling blocks);
// BertSelfOutput/BertOutput (hf:modeling_bert.py:289-293, 347-351), MultiScaleFusion.layer_norm
// (main.py:180), fusion_network LayerNorm+GELU (main.py:238-239).
//
// All of these are HBM/L2-bound streaming kernels: per-column (GraphNorm) or per-row (LayerNorm)
// parameters live in registers, rows stream through with coalesced 16-byte accesses where the width
// allows, statistics are fp32, and every reduction has a fixed order (no atomics).
#include "colreduce.hpp"

namespace gmlm {

// ------------------------------------------------------------------------------------------------
// GraphNorm
// ------------------------------------------------------------------------------------------------
template <typename T>
struct ColStatsFn {
  const T* x;
  const float* shift;
  int64_t f;
  __device__ void operator()(int64_t r, int64_t c, float (&v)[2]) const {
    const float d = Store<T>::ld(x + r * f + c) - (shift ? shift[c] : 0.f);
    v[0] = d;
    v[1] = d * d;
  }
};

__global__ void graphnorm_finalize_kernel(const float* __restrict__ s1, const float* __restrict__ s2,
                                          const float* __restrict__ shift, const float* __restrict__ ms, int64_t n,
                                          int64_t f, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= f) return;
  const float inv_n = 1.f / (float)n;
  const float sh = shift ? shift[c] : 0.f;
  const float mu = sh + s1[c] * inv_n;
  const float d = mu * ms[c] - sh;                       // o = (x - sh) - d
  float var = (s2[c] - 2.f * d * s1[c]) * inv_n + d * d;  // E[o^2]
  var = var > 0.f ? var : 0.f;
  mean[c] = mu;
  rstd[c] = 1.f / sqrtf(var + eps);
}

// rows stream through a block; thread <-> column (params in registers).
template <typename TY, bool ACT>
__global__ __launch_bounds__(256) void graphnorm_apply_kernel(const TY* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ w,
                                                               const float* __restrict__ b, const float* __restrict__ ms,
                                                               int64_t n, int64_t f, uint32_t thresh, float keep_scale,
                                                               SeedArg seed_a, TY* __restrict__ y) {
  const uint64_t seed = seed_a.get();
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= f) return;
  const float a = w[c] * rstd[c];
  const float sub = mean[c] * ms[c];
  const float bb = b[c];
  for (int64_t r = blockIdx.y; r < n; r += gridDim.y) {
    const int64_t i = r * f + c;
    float v = (Store<TY>::ld(x + i) - sub) * a + bb;
    if (ACT) v = gelu_fwd_t<TY>(v);
    if (thresh) v *= dropout_scale(seed, (uint64_t)i, thresh, keep_scale);
    Store<TY>::st(y + i, v);
  }
}

template <typename TG, bool ACT>
struct GraphNormBwdStatsFn {
  const TG* g;
  const TG* x;
  const float *mean, *rstd, *w, *b, *ms;
  int64_t f;
  uint32_t thresh;
  float keep_scale;
  SeedArg seed;
  __device__ void operator()(int64_t r, int64_t c, float (&v)[2]) const {
    const int64_t i = r * f + c;
    const float oh = (Store<TG>::ld(x + i) - mean[c] * ms[c]) * rstd[c];
    float gz = Store<TG>::ld(g + i);
    if (thresh) gz *= dropout_scale(seed.get(), (uint64_t)i, thresh, keep_scale);
    if (ACT) gz *= gelu_grad_t<TG>(w[c] * oh + b[c]);
    v[0] = gz;
    v[1] = gz * oh;
  }
};

template <typename TG, bool ACT>
__global__ __launch_bounds__(256) void graphnorm_bwd_apply_kernel(
    const TG* __restrict__ g, const TG* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ ms, const float* __restrict__ gs,
    int64_t n, int64_t n_total, int64_t f, uint32_t thresh, float keep_scale, SeedArg seed_a, TG* __restrict__ dx,
    float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dms) {
  const uint64_t seed = seed_a.get();
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= f) return;
  const float inv_n = 1.f / (float)n_total;
  const float mu = mean[c], rs = rstd[c], wc = w[c], bc = b[c], msc = ms[c];
  const float sg = gs[c], sgo = gs[f + c];
  const float m2 = sgo * inv_n;                               // mean_j gz_j * ohat_j
  const float mean_oh = mu * (1.f - msc) * rs;                // mean_j ohat_j
  const float mean_do = wc * rs * (sg * inv_n - m2 * mean_oh);
  if (blockIdx.y == 0) {
    if (dw) dw[c] = sgo;
    if (db) db[c] = sg;
    if (dms) dms[c] = -mu * mean_do * (float)n_total;
  }
  const float sub = mu * msc;
  for (int64_t r = blockIdx.y; r < n; r += gridDim.y) {
    const int64_t i = r * f + c;
    const float oh = (Store<TG>::ld(x + i) - sub) * rs;
    float gz = Store<TG>::ld(g + i);
    if (thresh) gz *= dropout_scale(seed, (uint64_t)i, thresh, keep_scale);
    if (ACT) gz *= gelu_grad_t<TG>(wc * oh + bc);
    Store<TG>::st(dx + i, wc * rs * (gz - oh * m2) - msc * mean_do);
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: LPR lanes (32 or 64) per row, row cached in registers (CH chunks of 16 B per lane, CH <= 4);
// with LPR = 32 a wave normalises two rows at once (f = 768 bf16 = 96 chunks = 32 lanes x 3: no idle lane).
// gamma / beta / bias are the same for every row a lane touches (its columns are fixed), so they are
// loaded once per wave with 16-byte loads and kept in registers.
// ------------------------------------------------------------------------------------------------
constexpr int kLnMaxCh = 4;

template <int V>
__device__ __forceinline__ void load_param(const float* __restrict__ p, int col, float (&o)[V], float fill) {
  if (p) {
#pragma unroll
    for (int v = 0; v < V; v += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + col + v);
      o[v] = t.x; o[v + 1] = t.y; o[v + 2] = t.z; o[v + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int v = 0; v < V; ++v) o[v] = fill;
  }
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int CH, int LPR, bool ACT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                      const T* __restrict__ res, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int64_t rows, int f, float eps,
                                                      uint32_t thresh, float keep_scale, SeedArg seed_a, T* __restrict__ y,
                                                      float* __restrict__ mean_o, float* __restrict__ rstd_o) {
  const uint64_t seed = seed_a.get();
  constexpr int V = Store<T>::kVec, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, lr = lane % LPR, sub = lane / LPR;
  const int nch = f / V;
  float gm[CH][V], bt[CH][V], bi[CH][V];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = c * LPR + lr;
    if (ch < nch) {
      load_param<V>(gamma, ch * V, gm[c], 1.f);
      load_param<V>(beta, ch * V, bt[c], 0.f);
      load_param<V>(bias, ch * V, bi[c], 0.f);
    }
  }
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const float inv_f = 1.f / (float)f;
  for (int64_t r0 = wave0 * RPW; r0 < rows; r0 += nwaves * RPW) {
    const int64_t r = r0 + sub;
    const bool live = r < rows;
    float z[CH][V];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = c * LPR + lr;
      if (ch < nch && live) {
        const int64_t off = r * f + (int64_t)ch * V;
        Store<T>::ldv(x + off, z[c]);
        float rr[V];
        if (res) Store<T>::ldv(res + off, rr);
#pragma unroll
        for (int v = 0; v < V; ++v) z[c][v] += bi[c][v];
        if (thresh) {
          const uint32_t kb = dropout_keep_bits<V>(seed, (uint64_t)off, thresh);
#pragma unroll
          for (int v = 0; v < V; ++v) z[c][v] = ((kb >> v) & 1u) ? z[c][v] * keep_scale : 0.f;
        }
        if (res) {
#pragma unroll
          for (int v = 0; v < V; ++v) z[c][v] += rr[v];
        }
#pragma unroll
        for (int v = 0; v < V; ++v) sum += z[c][v];
      }
    }
    const float mu = group_sum<LPR>(sum) * inv_f;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c * LPR + lr < nch && live)
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const float d = z[c][v] - mu;
          sq += d * d;
        }
    const float rs = 1.f / sqrtf(group_sum<LPR>(sq) * inv_f + eps);
    if (lr == 0 && live) {
      if (mean_o) mean_o[r] = mu;
      if (rstd_o) rstd_o[r] = rs;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = c * LPR + lr;
      if (ch < nch && live) {
        float o[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const float t = (z[c][v] - mu) * rs * gm[c][v] + bt[c][v];
          o[v] = ACT ? gelu_fwd_t<T>(t) : t;
        }
        Store<T>::stv(y + r * f + (int64_t)ch * V, o);
      }
    }
  }
}

// backward: each lane keeps partial dgamma/dbeta/dbias for its fixed columns; the 4 waves (x 64/LPR row
// groups) of a block are summed through LDS in a fixed order and the block writes ONE partial row
// [3][f]; a second kernel sums the blocks in a fixed tree order (deterministic).
template <typename T, int CH, int LPR, bool ACT, bool DROP, bool RES>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                      const float* __restrict__ bias, const T* __restrict__ res,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      int64_t rows, int f, uint32_t thresh, float keep_scale, SeedArg seed_a,
                                                      T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ partial) {
  const uint64_t seed = seed_a.get();
  constexpr int V = Store<T>::kVec, RPW = 64 / LPR;
  static_assert(CH * V <= 32, "keep bits of one row slice must fit one word");
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [4 waves][3 * f]
  const int lane = threadIdx.x & 63, lr = lane % LPR, sub = lane / LPR, wv = threadIdx.x >> 6;
  const int nch = f / V;
  float gm[CH][V], bt[CH][V], bi[CH][V];
  float dg[CH][V], db[CH][V], dbi[CH][V];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = c * LPR + lr;
    if (ch < nch) {
      load_param<V>(gamma, ch * V, gm[c], 1.f);
      if (ACT) load_param<V>(beta, ch * V, bt[c], 0.f);
      load_param<V>(bias, ch * V, bi[c], 0.f);
    }
#pragma unroll
    for (int v = 0; v < V; ++v) dg[c][v] = db[c][v] = dbi[c][v] = 0.f;
  }
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const float inv_f = 1.f / (float)f;
  // the loads of row group t+1 are issued before the arithmetic of row group t (two row groups in flight per wave)
  struct RowIn { uint4 x[CH], g[CH], rr[CH]; float mu, rs; };
  auto fetch = [&](int64_t r0, RowIn& in) {
    const int64_t r = r0 + sub;
    const bool live = r < rows;
    in.mu = live ? mean[r] : 0.f;
    in.rs = live ? rstd[r] : 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = c * LPR + lr;
      if (ch < nch && live) {
        const int64_t off = r * f + (int64_t)ch * V;
        in.x[c] = *reinterpret_cast<const uint4*>(x + off);
        in.g[c] = *reinterpret_cast<const uint4*>(dy + off);
        if (RES) in.rr[c] = *reinterpret_cast<const uint4*>(res + off);
      }
    }
  };
  RowIn cur;
  if (wave0 * RPW < rows) fetch(wave0 * RPW, cur);
  for (int64_t r0 = wave0 * RPW; r0 < rows; r0 += nwaves * RPW) {
    const int64_t r = r0 + sub;
    const bool live = r < rows;
    RowIn nxt;
    if (r0 + nwaves * RPW < rows) fetch(r0 + nwaves * RPW, nxt);
    const float mu = cur.mu, rs = cur.rs;
    float zh[CH][V], dzh[CH][V];
    float s1 = 0.f, s2 = 0.f;
    uint32_t keep = 0;                       // dropout decisions of this lane's elements, reused by the dx pass
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = c * LPR + lr;
      if (ch < nch && live) {
        const int64_t off = r * f + (int64_t)ch * V;
        float gv[V], rr[V];
        Store<T>::unpack(cur.x[c], zh[c]);
        Store<T>::unpack(cur.g[c], gv);
        if (RES) Store<T>::unpack(cur.rr[c], rr);
        uint32_t kb = 0;
        if (DROP) {
          kb = dropout_keep_bits<V>(seed, (uint64_t)off, thresh);
          keep |= kb << (c * V);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float z = zh[c][v] + bi[c][v];
          if (DROP) z = ((kb >> v) & 1u) ? z * keep_scale : 0.f;
          if (RES) z += rr[v];
          z = (z - mu) * rs;
          zh[c][v] = z;
          float dyn = gv[v];
          if (ACT) dyn *= gelu_grad_t<T>(z * gm[c][v] + bt[c][v]);
          dg[c][v] += dyn * z;
          db[c][v] += dyn;
          dzh[c][v] = dyn * gm[c][v];
          s1 += dzh[c][v];
          s2 += dzh[c][v] * z;
        }
      }
    }
    s1 = group_sum<LPR>(s1) * inv_f;
    s2 = group_sum<LPR>(s2) * inv_f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = c * LPR + lr;
      if (ch < nch && live) {
        const int64_t off = r * f + (int64_t)ch * V;
        float dz[V], dxv[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          dz[v] = rs * (dzh[c][v] - s1 - zh[c][v] * s2);
          dxv[v] = DROP ? (((keep >> (c * V + v)) & 1u) ? dz[v] * keep_scale : 0.f) : dz[v];
          dbi[c][v] += dxv[v];
        }
        Store<T>::stv(dx + off, dxv);
        if (dres) Store<T>::stv(dres + off, dz);
      }
    }
    cur = nxt;
  }
  // block reduction: the two row groups of a wave (LPR = 32) first add up by shuffle, then the 4 waves
  // through LDS (slot = wave), column-wise in a fixed order
  float* mine = lds + (size_t)wv * 3 * f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = c * LPR + lr;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      float a = dg[c][v], b2 = db[c][v], c2 = dbi[c][v];
      if (RPW == 2) {
        a += __shfl_xor(a, 32, 64);
        b2 += __shfl_xor(b2, 32, 64);
        c2 += __shfl_xor(c2, 32, 64);
      }
      if (ch < nch && sub == 0) {
        mine[ch * V + v] = a;
        mine[f + ch * V + v] = b2;
        mine[2 * f + ch * V + v] = c2;
      }
    }
  }
  __syncthreads();
  float* prow = partial + (int64_t)blockIdx.x * 3 * f;
  for (int i = threadIdx.x; i < 3 * f; i += 256)
    prow[i] = (lds[i] + lds[(size_t)3 * f + i]) + (lds[(size_t)6 * f + i] + lds[(size_t)9 * f + i]);
}

// out[c] = sum over `nrows` partial rows of partial[row][c], c in [0, 3f): block = 32 columns x 8 row lanes,
// each lane sums a strided subset (independent loads in flight), then a fixed-order LDS tree.
__global__ __launch_bounds__(256) void ln_bwd_final_kernel(const float* __restrict__ partial, int nrows, int f,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ dbias) {
  __shared__ float red[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  const int f3 = 3 * f;
  float acc = 0.f;
  if (c < f3) {
    int r = ty;
    for (; r + 24 < nrows; r += 32) {
      const float a0 = partial[(int64_t)r * f3 + c], a1 = partial[(int64_t)(r + 8) * f3 + c];
      const float a2 = partial[(int64_t)(r + 16) * f3 + c], a3 = partial[(int64_t)(r + 24) * f3 + c];
      acc += (a0 + a1) + (a2 + a3);
    }
    for (; r < nrows; r += 8) acc += partial[(int64_t)r * f3 + c];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c < f3) {
    const float s = ((red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx])) + ((red[4][tx] + red[5][tx]) + (red[6][tx] + red[7][tx]));
    float* out = c < f ? dgamma : (c < 2 * f ? dbeta : dbias);
    if (out) out[c % f] = s;
  }
}

struct LnGeom { int lpr, ch; };
// lanes per row: 32 when that leaves fewer idle lanes (e.g. 96 chunks = 32 x 3), else 64
static inline LnGeom ln_geom(int64_t f, int dtype, bool backward = false) {
  const int nch = (int)(f / (dtype == GMLM_F32 ? 4 : 8));
  const int idle32 = (int)cdiv(nch, 32) * 32 - nch, idle64 = (int)cdiv(nch, 64) * 64 - nch;
  LnGeom g;
  g.lpr = (nch <= 96 && idle32 < idle64) ? 32 : 64;
  // the backward kernel carries 3 accumulator sets per chunk: 3 chunks of 8 bf16 per lane need all 256 VGPRs
  // (1 wave/SIMD); 64 lanes x 2 chunks with some idle lanes keeps 2 waves/SIMD and is faster
  if (backward && dtype != GMLM_F32 && cdiv(nch, g.lpr) >= 3) g.lpr = 64;
  g.ch = (int)cdiv(nch, g.lpr);
  return g;
}

static inline int ln_bwd_blocks(int64_t rows) {
  int64_t b = cdiv(rows, 8);  // >= 2 rows per wave
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));   // 4 resident blocks per CU (LDS 36 KB each at f = 768)
}

}  // namespace gmlm

using namespace gmlm;

extern "C" size_t gmlm_colstats_workspace_bytes(int64_t n, int64_t f) { return col_reduce_workspace_bytes(n, f, 2); }

extern "C" int gmlm_colstats(const void* x, int dtype, const float* shift, int64_t n, int64_t f, float* s1, float* s2,
                             void* workspace, size_t workspace_bytes, gmlm_stream_t stream) {
  GMLM_REQUIRE(n >= 0 && f > 0 && s1 && s2 && (n == 0 || x), "colstats: bad arguments");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "colstats: unsupported dtype");
  GMLM_REQUIRE(s2 == s1 + f, "colstats: s1 and s2 must be the two rows of one [2, f] buffer");
  if (dtype == GMLM_F32)
    return col_reduce<2>(n, f, ColStatsFn<float>{(const float*)x, shift, f}, s1, workspace, workspace_bytes, as_stream(stream));
  return col_reduce<2>(n, f, ColStatsFn<bf16_t>{(const bf16_t*)x, shift, f}, s1, workspace, workspace_bytes, as_stream(stream));
}

extern "C" int gmlm_graphnorm_finalize(const float* s1, const float* s2, const float* shift, const float* mean_scale,
                                       int64_t n_total, int64_t f, float eps, float* mean, float* rstd,
                                       gmlm_stream_t stream) {
  GMLM_REQUIRE(s1 && s2 && mean_scale && mean && rstd && n_total > 0 && f > 0, "graphnorm_finalize: bad arguments");
  graphnorm_finalize_kernel<<<(int)cdiv(f, 256), 256, 0, as_stream(stream)>>>(s1, s2, shift, mean_scale, n_total, f, eps,
                                                                               mean, rstd);
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

static inline dim3 col_stream_grid(int64_t n, int64_t f) {
  const int ct = (int)cdiv(f, 256);
  int64_t ry = cdiv(4096, ct);
  if (ry > n) ry = n;
  if (ry < 1) ry = 1;
  return dim3(ct, (unsigned)ry);
}

extern "C" int gmlm_graphnorm_apply(const void* x, const float* mean, const float* rstd, const float* weight,
                                    const float* bias, const float* mean_scale, int64_t n, int64_t f, int act,
                                    float dropout_p, uint64_t seed_host, const uint64_t* seed_dev, void* y, int dtype, gmlm_stream_t stream) {
  const SeedArg seed{seed_host, seed_dev};
  GMLM_REQUIRE(n >= 0 && f > 0, "graphnorm_apply: bad sizes");
  GMLM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "graphnorm_apply: dropout_p must be in [0,1)");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "graphnorm_apply: unsupported dtype");
  if (n == 0) return GMLM_OK;
  GMLM_REQUIRE(x && mean && rstd && weight && bias && mean_scale && y, "graphnorm_apply: null pointer");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = dropout_keep_scale(th);
  const dim3 grid = col_stream_grid(n, f);
  hipStream_t st = as_stream(stream);
#define L(TY, A) graphnorm_apply_kernel<TY, A><<<grid, 256, 0, st>>>((const TY*)x, mean, rstd, weight, bias, mean_scale, n, f, th, ks, seed, (TY*)y)
  if (dtype == GMLM_F32) { if (act) L(float, true); else L(float, false); }
  else { if (act) L(bf16_t, true); else L(bf16_t, false); }
#undef L
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

extern "C" int gmlm_graphnorm_bwd_stats(const void* g, int dtype, const void* x, const float* mean, const float* rstd,
                                        const float* weight, const float* bias, const float* mean_scale, int64_t n,
                                        int64_t f, int act, float dropout_p, uint64_t seed_host, const uint64_t* seed_dev, float* gs, void* workspace,
                                        size_t workspace_bytes, gmlm_stream_t stream) {
  const SeedArg seed{seed_host, seed_dev};
  GMLM_REQUIRE(n >= 0 && f > 0 && gs, "graphnorm_bwd_stats: bad arguments");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "graphnorm_bwd_stats: unsupported dtype");
  GMLM_REQUIRE(n == 0 || (g && x && mean && rstd && weight && bias && mean_scale), "graphnorm_bwd_stats: null pointer");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = dropout_keep_scale(th);
  hipStream_t st = as_stream(stream);
#define L(TG, A) return col_reduce<2>(n, f, GraphNormBwdStatsFn<TG, A>{(const TG*)g, (const TG*)x, mean, rstd, weight, bias, mean_scale, f, th, ks, seed}, gs, workspace, workspace_bytes, st)
  if (dtype == GMLM_F32) { if (act) L(float, true); else L(float, false); }
  else { if (act) L(bf16_t, true); else L(bf16_t, false); }
#undef L
}

extern "C" int gmlm_graphnorm_bwd_apply(const void* g, int dtype, const void* x, const float* mean, const float* rstd,
                                        const float* weight, const float* bias, const float* mean_scale, const float* gs,
                                        int64_t n, int64_t n_total, int64_t f, int act, float dropout_p, uint64_t seed_host, const uint64_t* seed_dev,
                                        void* dx, float* dweight, float* dbias, float* dmean_scale, gmlm_stream_t stream) {
  const SeedArg seed{seed_host, seed_dev};
  GMLM_REQUIRE(n >= 0 && n_total >= n && n_total > 0 && f > 0 && gs, "graphnorm_bwd_apply: bad arguments");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "graphnorm_bwd_apply: unsupported dtype");
  GMLM_REQUIRE(mean && rstd && weight && bias && mean_scale && (n == 0 || (g && x && dx)), "graphnorm_bwd_apply: null pointer");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = dropout_keep_scale(th);
  dim3 grid = col_stream_grid(n > 0 ? n : 1, f);
  hipStream_t st = as_stream(stream);
#define L(TG, A) graphnorm_bwd_apply_kernel<TG, A><<<grid, 256, 0, st>>>((const TG*)g, (const TG*)x, mean, rstd, weight, bias, mean_scale, gs, n, n_total, f, th, ks, seed, (TG*)dx, dweight, dbias, dmean_scale)
  if (dtype == GMLM_F32) { if (act) L(float, true); else L(float, false); }
  else { if (act) L(bf16_t, true); else L(bf16_t, false); }
#undef L
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

static int ln_check(const char* who, int64_t rows, int64_t f, int dtype, float dropout_p) {
  GMLM_REQUIRE(rows >= 0 && f > 0, "%s: bad sizes", who);
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "%s: unsupported dtype", who);
  const int v = dtype == GMLM_F32 ? 4 : 8;
  GMLM_REQUIRE(f % v == 0 && f <= 64 * v * kLnMaxCh && f <= 1024, "%s: row width %ld must be a multiple of %d and <= %d", who,
               (long)f, v, 64 * v * kLnMaxCh < 1024 ? 64 * v * kLnMaxCh : 1024);
  GMLM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "%s: dropout_p must be in [0,1)", who);
  return GMLM_OK;
}

extern "C" int gmlm_bias_res_layernorm_fwd(const void* x, const float* bias, const void* residual, const float* gamma,
                                           const float* beta, int64_t rows, int64_t f, float eps, int act, float dropout_p,
                                           uint64_t seed_host, const uint64_t* seed_dev, void* y, float* mean, float* rstd, int dtype,
                                           gmlm_stream_t stream) {
  const SeedArg seed{seed_host, seed_dev};
  int rc = ln_check("bias_res_layernorm_fwd", rows, f, dtype, dropout_p);
  if (rc != GMLM_OK) return rc;
  if (rows == 0) return GMLM_OK;
  GMLM_REQUIRE(x && gamma && beta && y, "bias_res_layernorm_fwd: null pointer");
  GMLM_REQUIRE(aligned16(x) && aligned16(y) && (!residual || aligned16(residual)), "bias_res_layernorm_fwd: 16-byte alignment required");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = dropout_keep_scale(th);
  hipStream_t st = as_stream(stream);
  const LnGeom ge = ln_geom(f, dtype);
  const int grid = grid_cap(cdiv(rows, 4 * (64 / ge.lpr)));
#define L2(T, C, P, A) ln_fwd_kernel<T, C, P, A><<<grid, 256, 0, st>>>((const T*)x, bias, (const T*)residual, gamma, beta, rows, (int)f, eps, th, ks, seed, (T*)y, mean, rstd)
#define L(T, A) do { if (ge.lpr == 32) { if (ge.ch <= 1) L2(T, 1, 32, A); else if (ge.ch == 2) L2(T, 2, 32, A); else L2(T, 3, 32, A); } \
                     else { if (ge.ch <= 1) L2(T, 1, 64, A); else if (ge.ch == 2) L2(T, 2, 64, A); else if (ge.ch == 3) L2(T, 3, 64, A); else L2(T, 4, 64, A); } } while (0)
  if (dtype == GMLM_F32) { if (act) L(float, true); else L(float, false); }
  else { if (act) L(bf16_t, true); else L(bf16_t, false); }
#undef L
#undef L2
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

extern "C" size_t gmlm_layernorm_bwd_workspace_bytes(int64_t rows, int64_t f) {
  return (size_t)ln_bwd_blocks(rows) * 3 * f * sizeof(float);
}

extern "C" int gmlm_bias_res_layernorm_bwd(const void* dy, const void* x, const float* bias, const void* residual,
                                           const float* gamma, const float* beta, const float* mean, const float* rstd,
                                           int64_t rows, int64_t f, int act, float dropout_p, uint64_t seed_host, const uint64_t* seed_dev, void* dx,
                                           void* dresidual, float* dgamma, float* dbeta, float* dbias, int dtype,
                                           void* workspace, size_t workspace_bytes, gmlm_stream_t stream) {
  const SeedArg seed{seed_host, seed_dev};
  int rc = ln_check("bias_res_layernorm_bwd", rows, f, dtype, dropout_p);
  if (rc != GMLM_OK) return rc;
  hipStream_t st = as_stream(stream);
  if (rows == 0) {
    if (dgamma) GMLM_HIP(zero_async(dgamma, sizeof(float) * f, st));
    if (dbeta) GMLM_HIP(zero_async(dbeta, sizeof(float) * f, st));
    if (dbias) GMLM_HIP(zero_async(dbias, sizeof(float) * f, st));
    return GMLM_OK;
  }
  GMLM_REQUIRE(dy && x && gamma && beta && mean && rstd && dx, "bias_res_layernorm_bwd: null pointer");
  GMLM_REQUIRE(workspace && workspace_bytes >= gmlm_layernorm_bwd_workspace_bytes(rows, f), "bias_res_layernorm_bwd: workspace too small");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = dropout_keep_scale(th);
  const int blocks = ln_bwd_blocks(rows);
  float* partial = static_cast<float*>(workspace);
  const LnGeom ge = ln_geom(f, dtype, true);
  const size_t lds = (size_t)4 * 3 * f * sizeof(float);
#define L3(T, C, P, A, D, R) ln_bwd_kernel<T, C, P, A, D, R><<<blocks, 256, lds, st>>>((const T*)dy, (const T*)x, bias, (const T*)residual, gamma, beta, mean, rstd, rows, (int)f, th, ks, seed, (T*)dx, (T*)dresidual, partial)
#define L2(T, C, P, A) do { if (th) { if (residual) L3(T, C, P, A, true, true); else L3(T, C, P, A, true, false); } \
                            else { if (residual) L3(T, C, P, A, false, true); else L3(T, C, P, A, false, false); } } while (0)
#define L(T, A) do { if (ge.lpr == 32) { if (ge.ch <= 1) L2(T, 1, 32, A); else if (ge.ch == 2) L2(T, 2, 32, A); else L2(T, 3, 32, A); } \
                     else { if (ge.ch <= 1) L2(T, 1, 64, A); else if (ge.ch == 2) L2(T, 2, 64, A); else if (ge.ch == 3) L2(T, 3, 64, A); else L2(T, 4, 64, A); } } while (0)
  if (dtype == GMLM_F32) { if (act) L(float, true); else L(float, false); }
  else { if (act) L(bf16_t, true); else L(bf16_t, false); }
#undef L
#undef L2
#undef L3
  GMLM_LAUNCH_CHECK();
  ln_bwd_final_kernel<<<(int)cdiv(3 * f, 32), 256, 0, st>>>(partial, blocks, (int)f, dgamma, dbeta, dbias);
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}
