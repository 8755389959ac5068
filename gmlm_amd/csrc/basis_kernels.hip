// K10 — RGCN basis composition W_r = sum_b comp[r, b] * weight[b] and its backward (PyG RGCNConv,
// num_bases = 30; call sites main.py:189-203).  Pure HBM streaming over the [B, in*out] basis tensor
// (2.3 GB fp32 for the 3072 -> 6144 layer): each thread owns 4 consecutive columns, reads the B basis rows
// once with 16-byte loads, keeps the R_a x B coefficients in LDS.
//   forward : W[r, c]       = sum_b comp[r, b] * weight[b, c]            reads B rows, writes R_a rows
//   backward: dweight[b, c] = sum_r comp[r, b] * dW[r, c]                writes B rows (the optimiser's gradient)
//             dcomp[r, b]   = sum_c dW[r, c] * weight[b, c]              block partials -> fixed-order sum
// One fused backward pass reads weight and dW once and produces both gradients.
#include "colreduce.hpp"

namespace gmlm {

constexpr int kMaxRB = 5 * 32;   // R_a * B coefficients kept in LDS

// The basis rows of one column chunk lie in*out*4 bytes apart, so a serial loop over b is a chain of dependent
// long-latency loads.  Both kernels therefore fetch the rows in groups of kGroup with unconditional loads (the
// row index is clamped, the use is predicated) so that kGroup 16-byte loads are in flight per thread.
constexpr int kGroup = 6;

__global__ __launch_bounds__(256) void basis_compose_fwd_kernel(const float* __restrict__ comp, const float* __restrict__ weight,
                                                                 int ra, int nb, int64_t cols, float* __restrict__ w) {
  __shared__ float cs[kMaxRB];
  for (int i = threadIdx.x; i < ra * nb; i += 256) cs[i] = comp[i];
  __syncthreads();
  const int64_t nch = cols / 4;
  for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < nch; ch += (int64_t)gridDim.x * 256) {
    float4 acc[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b0 = 0; b0 < nb; b0 += kGroup) {
      float4 x[kGroup];
#pragma unroll
      for (int j = 0; j < kGroup; ++j) {
        const int b = b0 + j < nb ? b0 + j : nb - 1;
        x[j] = *reinterpret_cast<const float4*>(weight + (int64_t)b * cols + ch * 4);
      }
#pragma unroll
      for (int j = 0; j < kGroup; ++j)
        if (b0 + j < nb) {
#pragma unroll
          for (int r = 0; r < 5; ++r)
            if (r < ra) {
              const float c = cs[r * nb + b0 + j];
              acc[r].x = fmaf(c, x[j].x, acc[r].x); acc[r].y = fmaf(c, x[j].y, acc[r].y);
              acc[r].z = fmaf(c, x[j].z, acc[r].z); acc[r].w = fmaf(c, x[j].w, acc[r].w);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 5; ++r)
      if (r < ra) *reinterpret_cast<float4*>(w + (int64_t)r * cols + ch * 4) = acc[r];
  }
}

// RA is a template parameter so the per-thread dcomp partials (RA x 32 registers) stay small
template <int RA>
__global__ __launch_bounds__(256) void basis_compose_bwd_kernel(const float* __restrict__ comp, const float* __restrict__ weight,
                                                                 const float* __restrict__ dw, int nb, int64_t cols,
                                                                 float* __restrict__ dweight, float* __restrict__ dcomp_partial) {
  __shared__ float cs[kMaxRB];
  __shared__ float red[4][kMaxRB];
  for (int i = threadIdx.x; i < RA * nb; i += 256) cs[i] = comp[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t nch = cols / 4;
  float dc[RA][36];                         // 32 rounded up to a multiple of kGroup; entries >= nb stay 0
#pragma unroll
  for (int r = 0; r < RA; ++r)
#pragma unroll
    for (int b = 0; b < 36; ++b) dc[r][b] = 0.f;
  for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < nch; ch += (int64_t)gridDim.x * 256) {
    float4 g[RA];
#pragma unroll
    for (int r = 0; r < RA; ++r) g[r] = *reinterpret_cast<const float4*>(dw + (int64_t)r * cols + ch * 4);
#pragma unroll
    for (int b0 = 0; b0 < 36; b0 += kGroup) {
      if (b0 < nb) {                        // block-uniform
        float4 x[kGroup];
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
          const int b = b0 + j < nb ? b0 + j : nb - 1;
          x[j] = *reinterpret_cast<const float4*>(weight + (int64_t)b * cols + ch * 4);
        }
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
          const int b = b0 + j;
          if (b < nb) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < RA; ++r) {
              const float c = cs[r * nb + b];
              o.x = fmaf(c, g[r].x, o.x); o.y = fmaf(c, g[r].y, o.y); o.z = fmaf(c, g[r].z, o.z); o.w = fmaf(c, g[r].w, o.w);
              dc[r][b] += g[r].x * x[j].x + g[r].y * x[j].y + g[r].z * x[j].z + g[r].w * x[j].w;
            }
            *reinterpret_cast<float4*>(dweight + (int64_t)b * cols + ch * 4) = o;
          }
        }
      }
    }
  }
  // block reduction of dcomp in a fixed order: wave shuffle tree, then the 4 waves through LDS
#pragma unroll
  for (int r = 0; r < RA; ++r)
#pragma unroll
    for (int b = 0; b < 32; ++b)
      if (b < nb) {
        const float s = wave_sum(dc[r][b]);
        if (lane == 0) red[wv][r * nb + b] = s;
      }
  __syncthreads();
  for (int i = threadIdx.x; i < RA * nb; i += 256)
    dcomp_partial[(int64_t)blockIdx.x * RA * nb + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

static inline int basis_blocks(int64_t cols) {
  const int64_t b = cdiv(cols / 4, 256);
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace gmlm

using namespace gmlm;

static int basis_check(const char* who, int ra, int nb, int64_t cols) {
  GMLM_REQUIRE(ra >= 1 && ra <= 5, "%s: 1..5 active relations supported (got %d)", who, ra);
  GMLM_REQUIRE(nb >= 1 && nb <= 32, "%s: 1..32 bases supported (got %d)", who, nb);
  GMLM_REQUIRE(cols > 0 && cols % 4 == 0, "%s: in*out (%ld) must be a positive multiple of 4", who, (long)cols);
  return GMLM_OK;
}

extern "C" int gmlm_basis_compose_fwd(const float* comp, const float* weight, int r_active, int num_bases, int64_t cols,
                                      float* w, gmlm_stream_t stream) {
  int rc = basis_check("basis_compose_fwd", r_active, num_bases, cols);
  if (rc != GMLM_OK) return rc;
  GMLM_REQUIRE(comp && weight && w && aligned16(weight) && aligned16(w), "basis_compose_fwd: null or misaligned pointer");
  basis_compose_fwd_kernel<<<basis_blocks(cols), 256, 0, as_stream(stream)>>>(comp, weight, r_active, num_bases, cols, w);
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

extern "C" size_t gmlm_basis_compose_bwd_workspace_bytes(int r_active, int num_bases, int64_t cols) {
  return (size_t)basis_blocks(cols) * r_active * num_bases * sizeof(float);
}

extern "C" int gmlm_basis_compose_bwd(const float* comp, const float* weight, const float* dw, int r_active, int num_bases,
                                      int64_t cols, float* dweight, float* dcomp, void* workspace, size_t workspace_bytes,
                                      gmlm_stream_t stream) {
  int rc = basis_check("basis_compose_bwd", r_active, num_bases, cols);
  if (rc != GMLM_OK) return rc;
  GMLM_REQUIRE(comp && weight && dw && dweight && dcomp && aligned16(weight) && aligned16(dw) && aligned16(dweight),
               "basis_compose_bwd: null or misaligned pointer");
  GMLM_REQUIRE(workspace && workspace_bytes >= gmlm_basis_compose_bwd_workspace_bytes(r_active, num_bases, cols),
               "basis_compose_bwd: workspace too small");
  const int blocks = basis_blocks(cols);
  float* partial = static_cast<float*>(workspace);
  hipStream_t st = as_stream(stream);
#define L(RA) basis_compose_bwd_kernel<RA><<<blocks, 256, 0, st>>>(comp, weight, dw, num_bases, cols, dweight, partial)
  switch (r_active) {
    case 1: L(1); break;
    case 2: L(2); break;
    case 3: L(3); break;
    case 4: L(4); break;
    default: L(5); break;
  }
#undef L
  GMLM_LAUNCH_CHECK();
  rows_sum_kernel<<<(unsigned)cdiv((int64_t)r_active * num_bases, 32), 256, 0, st>>>(partial, blocks, (int64_t)r_active * num_bases, dcomp);
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}
