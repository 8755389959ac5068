// Internal helpers shared by the gfx950 kernels of libgmlm_hip.so (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "../../include/gmlm_hip.h"

namespace gmlm {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

#define GMLM_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::gmlm::set_error(__VA_ARGS__);      \
      return GMLM_EINVAL;                  \
    }                                      \
  } while (0)

#define GMLM_HIP(expr)                                                                  \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      ::gmlm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return GMLM_ELAUNCH;                                                              \
    }                                                                                   \
  } while (0)

#define GMLM_LAUNCH_CHECK()                                                             \
  do {                                                                                  \
    hipError_t _e = hipGetLastError();                                                  \
    if (_e != hipSuccess) {                                                             \
      ::gmlm::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return GMLM_ELAUNCH;                                                              \
    }                                                                                   \
  } while (0)

// Kernel attributes (dynamic LDS beyond 64 KiB) are a property of (function, device): set them once per DEVICE, under a lock,
// so that the library is safe to call from several host threads and from one process that drives several GPUs.  The only
// state kept is "device d has been initialised" (write-once bits).
struct PerDeviceOnce {
  std::mutex mu;
  uint64_t done = 0;
  template <typename F> int operator()(F&& init) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> guard(mu);
    if (dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return GMLM_OK;
    const int rc = init();
    if (rc == GMLM_OK && dev >= 0 && dev < 64) done |= 1ull << dev;
    return rc;
  }
};

inline hipStream_t as_stream(gmlm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
// memory-bound grid cap: 256 CUs x 8 blocks (guide G11), grid-stride the rest
inline int grid_cap(int64_t blocks, int64_t cap = 2048) { return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks)); }

// ---- storage types -------------------------------------------------------------------------
using bf16_t = uint16_t;  // raw bits; converted with the helpers below

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  // plain cast path keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
  __hip_bfloat16 b = __float2bfloat16(f);
  return *reinterpret_cast<bf16_t*>(&b);
}

// two f32 -> one dword of two bf16 (lo in bits 0-15): ONE v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN) instead of two
// single conversions + shift + or
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct Store;
template <> struct Store<float> {
  static constexpr int kVec = 4;  // elements per 16 bytes
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  __device__ static __forceinline__ void ldv(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  __device__ static __forceinline__ void unpack(const uint4& v, float (&o)[4]) {
    o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
  }
  __device__ static __forceinline__ void stv(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct Store<bf16_t> {
  static constexpr int kVec = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  __device__ static __forceinline__ void ldv(const bf16_t* p, float (&o)[8]) {
    unpack(*reinterpret_cast<const uint4*>(p), o);
  }
  __device__ static __forceinline__ void unpack(const uint4& v, float (&o)[8]) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(w[i] << 16);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void stv(bf16_t* p, const float (&o)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// ---- zero fill as a KERNEL ---------------------------------------------------------------------
// Not hipMemsetAsync: a memset node recorded into a hipGraph writes a wrong value from the second replay on with this ROCm /
// PyTorch stack (tools/dev/memset_graph_probe.py), and every entry of this library may be called under stream capture.
static __global__ void zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {      // bytes: a multiple of 4
  const size_t n = bytes / 4;
  if (n == 0) return hipSuccess;
  const size_t blocks = (n + 255) / 256;
  zero_words_kernel<<<(unsigned)(blocks > 4096 ? 4096 : blocks), 256, 0, st>>>(static_cast<uint32_t*>(p), n);
  return hipGetLastError();
}

// ---- wave64 reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- replayable dropout mask: counter hash of (seed, element index) ---------------------------
// 32-bit "lowbias" finaliser over (seed, idx >> 1): one hash word decides TWO adjacent elements with
// 16-bit thresholds (P(drop) = th / 65536), so vector kernels pay ~1.5 integer multiplies per element
// instead of three 64-bit ones.  The same (seed, idx) gives the same decision in forward, in the
// checkpoint recompute and in backward, so no mask tensor is stored.
// Dropout seed as the kernels receive it: the host's per-call-site value plus an optional DEVICE-resident counter.
// A captured hipGraph replays its kernel arguments unchanged; with `dev` set the effective seed moves with the counter
// (bumped once per forward replay by a kernel inside the graph), so every replay draws fresh masks while forward,
// checkpoint recompute and backward of the same replay still agree.  dev == nullptr: the plain host seed.
struct SeedArg {
  uint64_t v;
  const uint64_t* dev;
  __device__ __forceinline__ uint64_t get() const { return dev ? v + *dev : v; }
};

__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
  uint32_t x = (uint32_t)idx * 0x9E3779B1u + (uint32_t)(idx >> 32) * 0x85EBCA77u + (uint32_t)seed;
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x ^ (uint32_t)(seed >> 32);
}
// returns the multiplier applied to a kept / dropped element
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, uint32_t thresh16, float keep_scale) {
  const uint32_t w = hash_u32(seed, idx >> 1);
  return ((w >> (16 * (uint32_t)(idx & 1))) & 0xFFFFu) >= thresh16 ? keep_scale : 0.f;
}
// keep bits of V consecutive elements starting at `off` (a multiple of V, V even): bit v set = element off+v is
// kept.  Same decisions as dropout_scale(); the per-chunk part of the hash input is computed once (the low word of
// (off >> 1) + p cannot carry inside an aligned chunk, so the multiply distributes over the pair index).
template <int V>
__device__ __forceinline__ uint32_t dropout_keep_bits(uint64_t seed, uint64_t off, uint32_t thresh16) {
  const uint64_t i0 = off >> 1;
  const uint32_t base = (uint32_t)i0 * 0x9E3779B1u + (uint32_t)(i0 >> 32) * 0x85EBCA77u + (uint32_t)seed;
  const uint32_t hi = (uint32_t)(seed >> 32);
  uint32_t bits = 0;
#pragma unroll
  for (int p = 0; p < V / 2; ++p) {
    uint32_t x = base + (uint32_t)p * 0x9E3779B1u;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    x ^= hi;
    bits |= ((x & 0xFFFFu) >= thresh16 ? 1u : 0u) << (2 * p);
    bits |= ((x >> 16) >= thresh16 ? 1u : 0u) << (2 * p + 1);
  }
  return bits;
}
inline uint32_t dropout_threshold(float p) {
  if (p <= 0.f) return 0u;
  long t = lroundf(p * 65536.f);
  return (uint32_t)(t < 0 ? 0 : (t > 65535 ? 65535 : t));
}
inline float dropout_keep_scale(uint32_t thresh16) { return 65536.f / (65536.f - (float)thresh16); }

// Branch-free erf for the bf16 kernels (Abramowitz & Stegun 7.1.26, |abs err| <= 1.5e-7, i.e. ~2^-15 of a bf16
// ulp at 1): ~20 VALU slots against ~40 for the branchy libm erff, which made bias+GELU issue-bound.  Also
// returns exp(-z^2), which the GELU derivative needs anyway.  The f32 kernels keep erff.
__device__ __forceinline__ float erf_fast(float z, float& exp_mz2) {
  const float a = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  exp_mz2 = __builtin_amdgcn_exp2f(-1.4426950408889634f * a * a);
  return copysignf(fmaf(-p, exp_mz2, 1.f), z);
}
template <typename T> __device__ __forceinline__ float gelu_fwd_t(float x);
template <> __device__ __forceinline__ float gelu_fwd_t<float>(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
template <> __device__ __forceinline__ float gelu_fwd_t<bf16_t>(float x) {
  float e;
  return 0.5f * x * (1.f + erf_fast(x * 0.70710678118654752440f, e));
}
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x);
template <> __device__ __forceinline__ float gelu_grad_t<float>(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  return cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}
template <> __device__ __forceinline__ float gelu_grad_t<bf16_t>(float x) {
  float e;                                                       // = exp(-x^2 / 2)
  const float cdf = 0.5f * (1.f + erf_fast(x * 0.70710678118654752440f, e));
  return cdf + x * 0.39894228040143267794f * e;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

}  // namespace gmlm
