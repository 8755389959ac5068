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
struct ColStatsFn {
  const float* x;
  const float* shift;
  int64_t f;
  __device__ void operator()(int64_t r, int64_t c, float (&v)[2]) const {
    const float d = x[r * f + c] - (shift ? shift[c] : 0.f);
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
__global__ __launch_bounds__(256) void graphnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ w,
                                                               const float* __restrict__ b, const float* __restrict__ ms,
                                                               int64_t n, int64_t f, uint32_t thresh, float keep_scale,
                                                               uint64_t seed, TY* __restrict__ y) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= f) return;
  const float a = w[c] * rstd[c];
  const float sub = mean[c] * ms[c];
  const float bb = b[c];
  for (int64_t r = blockIdx.y; r < n; r += gridDim.y) {
    const int64_t i = r * f + c;
    float v = (x[i] - sub) * a + bb;
    if (ACT) v = gelu_erf(v);
    if (thresh) v *= dropout_scale(seed, (uint64_t)i, thresh, keep_scale);
    Store<TY>::st(y + i, v);
  }
}

template <typename TG, bool ACT>
struct GraphNormBwdStatsFn {
  const TG* g;
  const float *x, *mean, *rstd, *w, *b, *ms;
  int64_t f;
  uint32_t thresh;
  float keep_scale;
  uint64_t seed;
  __device__ void operator()(int64_t r, int64_t c, float (&v)[2]) const {
    const int64_t i = r * f + c;
    const float oh = (x[i] - mean[c] * ms[c]) * rstd[c];
    float gz = Store<TG>::ld(g + i);
    if (thresh) gz *= dropout_scale(seed, (uint64_t)i, thresh, keep_scale);
    if (ACT) gz *= gelu_erf_grad(w[c] * oh + b[c]);
    v[0] = gz;
    v[1] = gz * oh;
  }
};

template <typename TG, bool ACT>
__global__ __launch_bounds__(256) void graphnorm_bwd_apply_kernel(
    const TG* __restrict__ g, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ ms, const float* __restrict__ gs,
    int64_t n, int64_t n_total, int64_t f, uint32_t thresh, float keep_scale, uint64_t seed, float* __restrict__ dx,
    float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dms) {
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
    const float oh = (x[i] - sub) * rs;
    float gz = Store<TG>::ld(g + i);
    if (thresh) gz *= dropout_scale(seed, (uint64_t)i, thresh, keep_scale);
    if (ACT) gz *= gelu_erf_grad(wc * oh + bc);
    dx[i] = wc * rs * (gz - oh * m2) - msc * mean_do;
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row cached in registers (<= 4 chunks of 16 B per lane)
// ------------------------------------------------------------------------------------------------
constexpr int kLnMaxCh = 4;

template <typename T, bool ACT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                      const T* __restrict__ res, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int64_t rows, int f, float eps,
                                                      uint32_t thresh, float keep_scale, uint64_t seed, T* __restrict__ y,
                                                      float* __restrict__ mean_o, float* __restrict__ rstd_o) {
  constexpr int V = Store<T>::kVec;
  const int lane = threadIdx.x & 63;
  const int nch = f / V;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r = wave0; r < rows; r += nwaves) {
    float z[kLnMaxCh][V];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < kLnMaxCh; ++c) {
      const int ch = c * 64 + lane;
      if (ch < nch) {
        const int64_t off = r * f + (int64_t)ch * V;
        Store<T>::ldv(x + off, z[c]);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          if (bias) z[c][v] += bias[ch * V + v];
          if (thresh) z[c][v] *= dropout_scale(seed, (uint64_t)(off + v), thresh, keep_scale);
        }
        if (res) {
          float rr[V];
          Store<T>::ldv(res + off, rr);
#pragma unroll
          for (int v = 0; v < V; ++v) z[c][v] += rr[v];
        }
#pragma unroll
        for (int v = 0; v < V; ++v) sum += z[c][v];
      }
    }
    const float mu = wave_sum(sum) / (float)f;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < kLnMaxCh; ++c)
      if (c * 64 + lane < nch)
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const float d = z[c][v] - mu;
          sq += d * d;
        }
    const float rs = 1.f / sqrtf(wave_sum(sq) / (float)f + eps);
    if (lane == 0) {
      if (mean_o) mean_o[r] = mu;
      if (rstd_o) rstd_o[r] = rs;
    }
#pragma unroll
    for (int c = 0; c < kLnMaxCh; ++c) {
      const int ch = c * 64 + lane;
      if (ch < nch) {
        float o[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float t = (z[c][v] - mu) * rs * gamma[ch * V + v] + beta[ch * V + v];
          o[v] = ACT ? gelu_erf(t) : t;
        }
        Store<T>::stv(y + r * f + (int64_t)ch * V, o);
      }
    }
  }
}

// backward: each wave keeps per-lane partial dgamma/dbeta/dbias for its fixed columns
// and writes partial[wave][3][f]; a second kernel sums the waves in order (deterministic).
template <typename T, bool ACT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                      const float* __restrict__ bias, const T* __restrict__ res,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      int64_t rows, int f, uint32_t thresh, float keep_scale, uint64_t seed,
                                                      T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ partial) {
  constexpr int V = Store<T>::kVec;
  const int lane = threadIdx.x & 63;
  const int nch = f / V;
  float dg[kLnMaxCh][V], db[kLnMaxCh][V], dbi[kLnMaxCh][V];
#pragma unroll
  for (int c = 0; c < kLnMaxCh; ++c)
#pragma unroll
    for (int v = 0; v < V; ++v) dg[c][v] = db[c][v] = dbi[c][v] = 0.f;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r = wave0; r < rows; r += nwaves) {
    const float mu = mean[r], rs = rstd[r];
    float zh[kLnMaxCh][V], dzh[kLnMaxCh][V], dm[kLnMaxCh][V];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < kLnMaxCh; ++c) {
      const int ch = c * 64 + lane;
      if (ch < nch) {
        const int64_t off = r * f + (int64_t)ch * V;
        float xv[V], gv[V];
        Store<T>::ldv(x + off, xv);
        Store<T>::ldv(dy + off, gv);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float z = xv[v];
          if (bias) z += bias[ch * V + v];
          dm[c][v] = thresh ? dropout_scale(seed, (uint64_t)(off + v), thresh, keep_scale) : 1.f;
          z *= dm[c][v];
          zh[c][v] = z;
        }
        if (res) {
          float rr[V];
          Store<T>::ldv(res + off, rr);
#pragma unroll
          for (int v = 0; v < V; ++v) zh[c][v] += rr[v];
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
          zh[c][v] = (zh[c][v] - mu) * rs;
          const float gm = gamma[ch * V + v];
          float dyn = gv[v];
          if (ACT) dyn *= gelu_erf_grad(zh[c][v] * gm + beta[ch * V + v]);
          dg[c][v] += dyn * zh[c][v];
          db[c][v] += dyn;
          dzh[c][v] = dyn * gm;
          s1 += dzh[c][v];
          s2 += dzh[c][v] * zh[c][v];
        }
      }
    }
    s1 = wave_sum(s1) / (float)f;
    s2 = wave_sum(s2) / (float)f;
#pragma unroll
    for (int c = 0; c < kLnMaxCh; ++c) {
      const int ch = c * 64 + lane;
      if (ch < nch) {
        const int64_t off = r * f + (int64_t)ch * V;
        float dz[V], dxv[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          dz[v] = rs * (dzh[c][v] - s1 - zh[c][v] * s2);
          dxv[v] = dz[v] * dm[c][v];
          dbi[c][v] += dxv[v];
        }
        Store<T>::stv(dx + off, dxv);
        if (dres) Store<T>::stv(dres + off, dz);
      }
    }
  }
  // per-wave partials (fixed columns per lane) -> workspace; summed in wave order by the final kernel
  float* prow = partial + wave0 * 3 * (int64_t)f;
#pragma unroll
  for (int c = 0; c < kLnMaxCh; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nch)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        prow[ch * V + v] = dg[c][v];
        prow[f + ch * V + v] = db[c][v];
        prow[2 * f + ch * V + v] = dbi[c][v];
      }
  }
}

__global__ void ln_bwd_final_kernel(const float* __restrict__ partial, int blocks, int f, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, float* __restrict__ dbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= f) return;
  float a = 0.f, b = 0.f, d = 0.f;
  for (int j = 0; j < blocks; ++j) {
    a += partial[(int64_t)j * 3 * f + c];
    b += partial[(int64_t)j * 3 * f + f + c];
    d += partial[(int64_t)j * 3 * f + 2 * f + c];
  }
  if (dgamma) dgamma[c] = a;
  if (dbeta) dbeta[c] = b;
  if (dbias) dbias[c] = d;
}

static inline int ln_bwd_blocks(int64_t rows) {
  int64_t b = cdiv(rows, 8);  // >= 2 rows per wave
  return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

}  // namespace gmlm

using namespace gmlm;

extern "C" size_t gmlm_colstats_workspace_bytes(int64_t n, int64_t f) { return col_reduce_workspace_bytes(n, f, 2); }

extern "C" int gmlm_colstats(const float* x, const float* shift, int64_t n, int64_t f, float* s1, float* s2,
                             void* workspace, size_t workspace_bytes, gmlm_stream_t stream) {
  GMLM_REQUIRE(n >= 0 && f > 0 && s1 && s2 && (n == 0 || x), "colstats: bad arguments");
  GMLM_REQUIRE(s2 == s1 + f, "colstats: s1 and s2 must be the two rows of one [2, f] buffer");
  return col_reduce<2>(n, f, ColStatsFn{x, shift, f}, s1, workspace, workspace_bytes, as_stream(stream));
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

extern "C" int gmlm_graphnorm_apply(const float* x, const float* mean, const float* rstd, const float* weight,
                                    const float* bias, const float* mean_scale, int64_t n, int64_t f, int act,
                                    float dropout_p, uint64_t seed, void* y, int dtype, gmlm_stream_t stream) {
  GMLM_REQUIRE(n >= 0 && f > 0, "graphnorm_apply: bad sizes");
  GMLM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "graphnorm_apply: dropout_p must be in [0,1)");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "graphnorm_apply: unsupported dtype");
  if (n == 0) return GMLM_OK;
  GMLM_REQUIRE(x && mean && rstd && weight && bias && mean_scale && y, "graphnorm_apply: null pointer");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = 1.f / (1.f - dropout_p);
  const dim3 grid = col_stream_grid(n, f);
  hipStream_t st = as_stream(stream);
#define L(TY, A) graphnorm_apply_kernel<TY, A><<<grid, 256, 0, st>>>(x, mean, rstd, weight, bias, mean_scale, n, f, th, ks, seed, (TY*)y)
  if (dtype == GMLM_F32) { if (act) L(float, true); else L(float, false); }
  else { if (act) L(bf16_t, true); else L(bf16_t, false); }
#undef L
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

extern "C" int gmlm_graphnorm_bwd_stats(const void* g, int dtype, const float* x, const float* mean, const float* rstd,
                                        const float* weight, const float* bias, const float* mean_scale, int64_t n,
                                        int64_t f, int act, float dropout_p, uint64_t seed, float* gs, void* workspace,
                                        size_t workspace_bytes, gmlm_stream_t stream) {
  GMLM_REQUIRE(n >= 0 && f > 0 && gs, "graphnorm_bwd_stats: bad arguments");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "graphnorm_bwd_stats: unsupported dtype");
  GMLM_REQUIRE(n == 0 || (g && x && mean && rstd && weight && bias && mean_scale), "graphnorm_bwd_stats: null pointer");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = 1.f / (1.f - dropout_p);
  hipStream_t st = as_stream(stream);
#define L(TG, A) return col_reduce<2>(n, f, GraphNormBwdStatsFn<TG, A>{(const TG*)g, x, mean, rstd, weight, bias, mean_scale, f, th, ks, seed}, gs, workspace, workspace_bytes, st)
  if (dtype == GMLM_F32) { if (act) L(float, true); else L(float, false); }
  else { if (act) L(bf16_t, true); else L(bf16_t, false); }
#undef L
}

extern "C" int gmlm_graphnorm_bwd_apply(const void* g, int dtype, const float* x, const float* mean, const float* rstd,
                                        const float* weight, const float* bias, const float* mean_scale, const float* gs,
                                        int64_t n, int64_t n_total, int64_t f, int act, float dropout_p, uint64_t seed,
                                        float* dx, float* dweight, float* dbias, float* dmean_scale, gmlm_stream_t stream) {
  GMLM_REQUIRE(n >= 0 && n_total >= n && n_total > 0 && f > 0 && gs, "graphnorm_bwd_apply: bad arguments");
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "graphnorm_bwd_apply: unsupported dtype");
  GMLM_REQUIRE(mean && rstd && weight && bias && mean_scale && (n == 0 || (g && x && dx)), "graphnorm_bwd_apply: null pointer");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = 1.f / (1.f - dropout_p);
  dim3 grid = col_stream_grid(n > 0 ? n : 1, f);
  hipStream_t st = as_stream(stream);
#define L(TG, A) graphnorm_bwd_apply_kernel<TG, A><<<grid, 256, 0, st>>>((const TG*)g, x, mean, rstd, weight, bias, mean_scale, gs, n, n_total, f, th, ks, seed, dx, dweight, dbias, dmean_scale)
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
  GMLM_REQUIRE(f % v == 0 && f <= 64 * v * kLnMaxCh, "%s: row width %ld must be a multiple of %d and <= %d", who, (long)f, v,
               64 * v * kLnMaxCh);
  GMLM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "%s: dropout_p must be in [0,1)", who);
  return GMLM_OK;
}

extern "C" int gmlm_bias_res_layernorm_fwd(const void* x, const float* bias, const void* residual, const float* gamma,
                                           const float* beta, int64_t rows, int64_t f, float eps, int act, float dropout_p,
                                           uint64_t seed, void* y, float* mean, float* rstd, int dtype,
                                           gmlm_stream_t stream) {
  int rc = ln_check("bias_res_layernorm_fwd", rows, f, dtype, dropout_p);
  if (rc != GMLM_OK) return rc;
  if (rows == 0) return GMLM_OK;
  GMLM_REQUIRE(x && gamma && beta && y, "bias_res_layernorm_fwd: null pointer");
  GMLM_REQUIRE(aligned16(x) && aligned16(y) && (!residual || aligned16(residual)), "bias_res_layernorm_fwd: 16-byte alignment required");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = 1.f / (1.f - dropout_p);
  const int grid = grid_cap(cdiv(rows, 4));
  hipStream_t st = as_stream(stream);
#define L(T, A) ln_fwd_kernel<T, A><<<grid, 256, 0, st>>>((const T*)x, bias, (const T*)residual, gamma, beta, rows, (int)f, eps, th, ks, seed, (T*)y, mean, rstd)
  if (dtype == GMLM_F32) { if (act) L(float, true); else L(float, false); }
  else { if (act) L(bf16_t, true); else L(bf16_t, false); }
#undef L
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

extern "C" size_t gmlm_layernorm_bwd_workspace_bytes(int64_t rows, int64_t f) {
  return (size_t)ln_bwd_blocks(rows) * 4 * 3 * f * sizeof(float);
}

extern "C" int gmlm_bias_res_layernorm_bwd(const void* dy, const void* x, const float* bias, const void* residual,
                                           const float* gamma, const float* beta, const float* mean, const float* rstd,
                                           int64_t rows, int64_t f, int act, float dropout_p, uint64_t seed, void* dx,
                                           void* dresidual, float* dgamma, float* dbeta, float* dbias, int dtype,
                                           void* workspace, size_t workspace_bytes, gmlm_stream_t stream) {
  int rc = ln_check("bias_res_layernorm_bwd", rows, f, dtype, dropout_p);
  if (rc != GMLM_OK) return rc;
  hipStream_t st = as_stream(stream);
  if (rows == 0) {
    if (dgamma) GMLM_HIP(hipMemsetAsync(dgamma, 0, sizeof(float) * f, st));
    if (dbeta) GMLM_HIP(hipMemsetAsync(dbeta, 0, sizeof(float) * f, st));
    if (dbias) GMLM_HIP(hipMemsetAsync(dbias, 0, sizeof(float) * f, st));
    return GMLM_OK;
  }
  GMLM_REQUIRE(dy && x && gamma && beta && mean && rstd && dx, "bias_res_layernorm_bwd: null pointer");
  GMLM_REQUIRE(workspace && workspace_bytes >= gmlm_layernorm_bwd_workspace_bytes(rows, f), "bias_res_layernorm_bwd: workspace too small");
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = 1.f / (1.f - dropout_p);
  const int blocks = ln_bwd_blocks(rows);
  float* partial = static_cast<float*>(workspace);
#define L(T, A) ln_bwd_kernel<T, A><<<blocks, 256, 0, st>>>((const T*)dy, (const T*)x, bias, (const T*)residual, gamma, beta, mean, rstd, rows, (int)f, th, ks, seed, (T*)dx, (T*)dresidual, partial)
  if (dtype == GMLM_F32) { if (act) L(float, true); else L(float, false); }
  else { if (act) L(bf16_t, true); else L(bf16_t, false); }
#undef L
  GMLM_LAUNCH_CHECK();
  ln_bwd_final_kernel<<<(int)cdiv(f, 256), 256, 0, st>>>(partial, blocks * 4, (int)f, dgamma, dbeta, dbias);
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}
