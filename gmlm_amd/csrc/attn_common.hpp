// Shared device helpers of the attention kernels (K5/K7): MFMA operand fragments, LDS staging, dropout hash.
// Included by attn_kernels.hip (f32 + backward kernels) and attn_fwd_pipe.hip (software-pipelined bf16 forward).
#pragma once
#include "common.hpp"

namespace gmlm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define GMLM_LDS3(T, p) ((__attribute__((address_space(3))) T*)(p))

struct AttnParams {
  const void *q, *k, *v, *out, *dout;
  const void* out_lo;         // backward, bf16: optional residual of the forward output, O = out + out_lo to 2^-17 (delta = rowsum(dO * O) then
                              // has fp32-like accuracy instead of carrying the 2^-9 rounding of the stored output); NULL = out alone
  void* o_lo_w;               // forward, bf16: where to store that residual (NULL = not wanted)
  const int32_t* groups;      // short-sequence kernels, packed mode: sequences [groups[g], groups[g+1]) form work item g (NULL: one sequence each)
  const float* lse;
  const int32_t* kv_len;
  const int32_t* cu;      // varlen (packed) mode: sequence b owns rows [cu[b], cu[b+1]) of q AND k/v; lq = total rows
  float* delta;
  float* dbias_part;          // short-sequence backward: [sequences, 3, h*d] column sums of this workgroup's dQ / dK / dV rows (NULL = off)
  void *o_w, *dq, *dk, *dv;
  float* lse_w;
  int64_t b, h, lq, lk;
  int64_t q_stride, k_stride, v_stride, dq_stride, dk_stride, dv_stride;
  float scale;
  uint32_t drop_thresh;   // attention-probability dropout: 8-bit threshold, P(drop) = drop_thresh / 256 (0 = off)
  float keep_scale;
  uint64_t seed;
  const uint64_t* seed_dev;   // optional device-resident counter added to seed (hipGraph replays; see SeedArg in common.hpp)
  uint32_t grid_q, grid_pairs;   // pipelined forward: 1-D grid of grid_q query blocks x grid_pairs (batch, head) pairs
};

// per-(batch, head) view: padded layout [b, l, ...] or packed layout (cu_seqlens)
__device__ __forceinline__ void seq_view(const AttnParams& p, int64_t b, int64_t hd, int64_t& lq_, int64_t& lk_, int64_t& qbase,
                                         int64_t& kbase, int64_t& lse_base) {
  if (p.cu) {
    qbase = kbase = p.cu[b];
    lq_ = lk_ = p.cu[b + 1] - p.cu[b];
    lse_base = hd * p.lq + qbase;                 // lse / delta laid out [h, total_rows]
  } else {
    qbase = b * p.lq; kbase = b * p.lk; lq_ = p.lq; lk_ = p.lk;
    lse_base = (b * p.h + hd) * p.lq;             // [b, h, lq]
  }
}

template <typename T> struct Pad;
template <> struct Pad<bf16_t> { static constexpr int v = 8; };
template <> struct Pad<float> { static constexpr int v = 4; };

// ---- row fragment: D elements of one row, laid out as the "B from a row" MFMA operand -----------
template <typename T, int D> struct RowFrag;
template <int D> struct RowFrag<bf16_t, D> {
  bf16x8 v[D / 16];
  __device__ __forceinline__ void load(const bf16_t* row, bool valid, int h) {
#pragma unroll
    for (int s = 0; s < D / 16; ++s) {
      if (valid) v[s] = *reinterpret_cast<const bf16x8*>(row + 16 * s + 8 * h);
      else
#pragma unroll
        for (int j = 0; j < 8; ++j) v[s][j] = (__bf16)0.f;
    }
  }
};
template <int D> struct RowFrag<float, D> {
  float v[D / 2];
  __device__ __forceinline__ void load(const float* row, bool valid, int h) {
#pragma unroll
    for (int s = 0; s < D / 2; ++s) v[s] = valid ? row[2 * s + h] : 0.f;
  }
};

// acc += A(tile rows row0..row0+31, k = D) * B(frag):  result[row][col = lane's row entity]
template <int D>
__device__ __forceinline__ void mma_rows(const bf16_t* ts, int pitch, int row0, const RowFrag<bf16_t, D>& f, f32x16& acc,
                                         int r, int h) {
#pragma unroll
  for (int s = 0; s < D / 16; ++s) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(ts + (row0 + r) * pitch + 16 * s + 8 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, f.v[s], acc, 0, 0, 0);
  }
}
template <int D>
__device__ __forceinline__ void mma_rows(const float* ts, int pitch, int row0, const RowFrag<float, D>& f, f32x16& acc,
                                         int r, int h) {
#pragma unroll
  for (int s = 0; s < D / 2; ++s) {
    const float a = ts[(row0 + r) * pitch + 2 * s + h];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, f.v[s], acc, 0, 0, 0);
  }
}

// out[db] += A(tile^T: rows = D-block db, k = tile rows row0..row0+31) * B(x)
// x is a 32x32 accumulator tile whose ROW index is the summed index (guide: accumulator as operand).
template <int D>
__device__ __forceinline__ void mma_acc(const bf16_t* ts, int pitch, int row0, const f32x16& x, f32x16 (&out)[D / 32],
                                        int lane) {
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3, hh = g >> 1;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)x[8 * s + j];
    const bf16_t* base = ts + (row0 + 16 * s + 4 * hh + qq) * pitch + 16 * (g & 1) + 4 * pp;
#pragma unroll
    for (int db = 0; db < D / 32; ++db) {
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(GMLM_LDS3(bf16x4, base + db * 32));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(GMLM_LDS3(bf16x4, base + db * 32 + 8 * pitch));
      bf16x8 a;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
      out[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, out[db], 0, 0, 0);
    }
  }
}
// ---- dense (unpadded) d = 64 tile images with an XOR swizzle of the 16-byte slot index: 128-byte rows, no padding bytes.
// Row-operand image (mma_rows_sw): slot = chunk ^ ((row >> 1) & 7): the 16 rows of a ds_read_b128 phase hit 16 distinct
// (row parity, slot) positions = all 64 banks once.  Transposed-read image (mma_acc_sw): slot = chunk ^ (4 * ((row >> 1) & 1)):
// the four rows of a ds_read_b64_tr_b16 quarter land in four different bank quarters.  (Same functions as the pipelined
// forward's K / V images, attn_fwd_pipe.hip.)
__device__ __forceinline__ int sw_row(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int sw_tr(int row) { return 4 * ((row >> 1) & 1); }
__device__ __forceinline__ void mma_rows_sw(const bf16_t* ts, int row0, const RowFrag<bf16_t, 64>& f, f32x16& acc, int r, int h) {
  const bf16_t* rowp = ts + (row0 + r) * 64;
  const int x = sw_row(row0 + r);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(rowp + (((2 * s + h) ^ x) << 3));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, f.v[s], acc, 0, 0, 0);
  }
}
__device__ __forceinline__ void mma_acc_sw(const bf16_t* ts, int row0, const f32x16& x, f32x16 (&out)[2], int lane) {
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3, hh = g >> 1;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)x[8 * s + j];
    const int row = row0 + 16 * s + 4 * hh + qq;                       // the +8 row of the hi half has the same swizzle phase
    const bf16_t* rowp = ts + row * 64 + 4 * (pp & 1);
    const int xs = sw_tr(row);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int slot = (4 * db + 2 * (g & 1) + (pp >> 1)) ^ xs;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(GMLM_LDS3(bf16x4, rowp + (slot << 3)));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(GMLM_LDS3(bf16x4, rowp + (slot << 3) + 8 * 64));
      bf16x8 a;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
      out[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, out[db], 0, 0, 0);
    }
  }
}

// out[db] += A(tile^T) * B with B[k][every column] = coef[tile row k] (bf16, in LDS): the matrix-vector product
// tile^T coef on the matrix pipe; all 32 result columns are identical.  Same row order as mma_acc.
template <int D>
__device__ __forceinline__ void mma_acc_coef(const bf16_t* ts, int pitch, int row0, const bf16_t* coef, f32x16 (&out)[D / 32], int lane) {
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3, hh = g >> 1;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const bf16x4 c0 = *reinterpret_cast<const bf16x4*>(coef + row0 + 16 * s + 4 * hh);        // rows +0..3
    const bf16x4 c1 = *reinterpret_cast<const bf16x4*>(coef + row0 + 16 * s + 4 * hh + 8);    // rows +8..11
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { b[j] = c0[j]; b[4 + j] = c1[j]; }
    const bf16_t* base = ts + (row0 + 16 * s + 4 * hh + qq) * pitch + 16 * (g & 1) + 4 * pp;
#pragma unroll
    for (int db = 0; db < D / 32; ++db) {
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(GMLM_LDS3(bf16x4, base + db * 32));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(GMLM_LDS3(bf16x4, base + db * 32 + 8 * pitch));
      bf16x8 a;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
      out[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, out[db], 0, 0, 0);
    }
  }
}
template <int D>
__device__ __forceinline__ void mma_acc(const float* ts, int pitch, int row0, const f32x16& x, f32x16 (&out)[D / 32],
                                        int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int krow = row0 + (t & 3) + 8 * (t >> 2) + 4 * h;
#pragma unroll
    for (int db = 0; db < D / 32; ++db) {
      const float a = ts[krow * pitch + db * 32 + r];
      out[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[t], out[db], 0, 0, 0);
    }
  }
}

// cooperative staging of ROWS rows of D elements (row-major, pitch D+pad); rows >= limit are zero
template <typename T, int D, int ROWS, int NT>
__device__ __forceinline__ void stage_rows(T* ts, const T* g, int64_t g_stride, int64_t row0, int64_t limit, int tid) {
  constexpr int V = Store<T>::kVec;
  constexpr int CPR = D / V;
  constexpr int PITCH = D + Pad<T>::v;
  for (int i = tid; i < ROWS * CPR; i += NT) {
    const int row = i / CPR, c = i % CPR;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (row0 + row < limit) val = *reinterpret_cast<const uint4*>(g + (row0 + row) * g_stride + c * V);
    *reinterpret_cast<uint4*>(ts + row * PITCH + c * V) = val;
  }
}

// transposed epilogue store: acc holds [d-block rows (regs), entity column (lane)]
template <typename T>
__device__ __forceinline__ void store_t(T* rowptr /* + db*32 applied */, const f32x16& acc, float mul, int h) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    T* p = rowptr + 8 * g4 + 4 * h;
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(p) = make_float4(acc[4 * g4] * mul, acc[4 * g4 + 1] * mul, acc[4 * g4 + 2] * mul, acc[4 * g4 + 3] * mul);
    } else {
      uint2 w;
      w.x = pack_bf16x2(acc[4 * g4] * mul, acc[4 * g4 + 1] * mul);
      w.y = pack_bf16x2(acc[4 * g4 + 2] * mul, acc[4 * g4 + 3] * mul);
      *reinterpret_cast<uint2*>(p) = w;
    }
  }
}

// bf16 residual of the same store: lo = bf16(x - bf16(x)), x = acc * mul, so that hi + lo carries x to 2^-17 (the backward's
// delta = rowsum(dO * O) is then as accurate as with an fp32 output)
__device__ __forceinline__ void store_t_lo(bf16_t* rowptr, const f32x16& acc, float mul, int h) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    float x[4], d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = acc[4 * g4 + j] * mul;
    const uint32_t h0 = pack_bf16x2(x[0], x[1]), h1 = pack_bf16x2(x[2], x[3]);
    d[0] = x[0] - __uint_as_float(h0 << 16); d[1] = x[1] - __uint_as_float(h0 & 0xffff0000u);
    d[2] = x[2] - __uint_as_float(h1 << 16); d[3] = x[3] - __uint_as_float(h1 & 0xffff0000u);
    uint2 w;
    w.x = pack_bf16x2(d[0], d[1]);
    w.y = pack_bf16x2(d[2], d[3]);
    *reinterpret_cast<uint2*>(rowptr + 8 * g4 + 4 * h) = w;
  }
}

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// register-staged tile: global -> registers (issued early, latency hidden under the MFMAs of the
// current tile) -> LDS (written after the compute; the buffer being written was last read one
// iteration ago, behind a barrier).  One barrier per tile with two LDS buffers (bf16); the exact-f32
// parity path keeps one buffer (LDS budget) and two barriers.
template <typename T, int D, int ROWS, int NT>
struct TileRegs {
  static constexpr int V = Store<T>::kVec, CPR = D / V, TOTAL = ROWS * CPR, PER = (TOTAL + NT - 1) / NT,
                       PITCH = D + Pad<T>::v;
  uint4 v[PER];
  int goff[PER];   // element offset of this thread's k-th chunk inside a tile (row * g_stride + col), computed once
  int loff[PER];   // LDS element offset
  int rowk[PER];   // tile row of the chunk (bounds check against the rows left)
  __device__ __forceinline__ void init(int64_t g_stride, int tid) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * NT;
      const int row = i / CPR, c = i % CPR;
      rowk[k] = i < TOTAL ? row : (1 << 30);
      goff[k] = (int)(row * g_stride) + c * V;
      loff[k] = row * PITCH + c * V;
    }
  }
  // g: start of the (batch, head) slab; row0: first row of the tile (block-uniform); limit: rows in the slab
  __device__ __forceinline__ void load(const T* g, int64_t g_stride, int64_t row0, int64_t limit, int /*tid*/) {
    const T* base = g + row0 * g_stride;          // wave-uniform
    const int left = (int)(limit - row0 < (1 << 29) ? limit - row0 : (1 << 29));
    if (left >= ROWS && TOTAL % NT == 0) {        // interior tile (block-uniform): no bounds checks, no zero fill
#pragma unroll
      for (int k = 0; k < PER; ++k) v[k] = *reinterpret_cast<const uint4*>(base + goff[k]);
    } else {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        if (rowk[k] < left) v[k] = *reinterpret_cast<const uint4*>(base + goff[k]);
        else v[k] = make_uint4(0, 0, 0, 0);
      }
    }
  }
  __device__ __forceinline__ void store(T* ts, int /*tid*/) const {
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (rowk[k] < ROWS) *reinterpret_cast<uint4*>(ts + loff[k]) = v[k];
  }
};

// Attention-probability dropout, replayable: ONE 32-bit hash word decides a 2 x 2 tile of (query, key) pairs through
// four 8-bit fields (an element is dropped when its field < th8: P(drop) = th8 / 256, the keep scale uses that
// quantised rate).  word(q, k) = lowbias32(base + (q >> 1) * C1 + (k >> 1) * C2), base = f(seed, slab) with slab =
// the (batch, head) identity; field index = 2 * (q & 1) + (k & 1), i.e. bits [16 (q&1) + 8 (k&1), +8).
// The same function is evaluated by the forward, dQ and dK/dV kernels (nothing is stored); the 2 x 2 tile makes it
// cost ONE hash per TWO elements in both register layouts: query-on-lane kernels hold keys (k, k+1) of one query in
// adjacent accumulator registers (halfword 16 (q&1), bytes 0 / 1), key-on-lane kernels hold queries (q, q+1) of one
// key (shift 8 (k&1), bytes 0 / 2).
constexpr uint32_t kDropC1 = 0x9E3779B1u, kDropC2 = 0x85EBCA77u, kDropC3 = 0xC2B2AE3Du;
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {   // one multiply round: enough for a dropout mask
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
  return x;
}
__device__ __forceinline__ uint64_t attn_seed(const AttnParams& p) { return p.seed_dev ? p.seed + *p.seed_dev : p.seed; }
__device__ __forceinline__ uint32_t drop_base(uint64_t seed, int64_t slab) {
  return ((uint32_t)seed ^ (uint32_t)(seed >> 32)) + (uint32_t)slab * kDropC3;
}
// multiplier of a kept / dropped element from an 8-bit field
__device__ __forceinline__ float drop_keep8(uint32_t field, uint32_t th8, float ks) { return (field & 0xFFu) >= th8 ? ks : 0.f; }
// query-on-lane layout: multipliers of keys (k, k+1), k even, for the lane's query q.  u = base + (q>>1) C1 + (k>>1) C2
__device__ __forceinline__ void drop_pair_q(uint32_t u, int q_odd, uint32_t th8, float ks, float& m0, float& m1) {
  const uint32_t w = lowbias32(u) >> (16 * q_odd);
  m0 = drop_keep8(w, th8, ks);
  m1 = drop_keep8(w >> 8, th8, ks);
}
// key-on-lane layout: multipliers of queries (q, q+1), q even, for the lane's key k
__device__ __forceinline__ void drop_pair_k(uint32_t u, int k_odd, uint32_t th8, float ks, float& m0, float& m1) {
  const uint32_t w = lowbias32(u) >> (8 * k_odd);
  m0 = drop_keep8(w, th8, ks);
  m1 = drop_keep8(w >> 16, th8, ks);
}

// exchange between the two 32-lane halves of the wave on the VALU (v_permlane32_swap) instead of an LDS
// bpermute: r[0] / r[1] hold {own, other-half} values in some order for every lane, so max / sum of the
// pair need no select
__device__ __forceinline__ float xhalf_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// max of the 16 registers of a 32x32 accumulator tile (v_max3_f32 chain)
__device__ __forceinline__ float max16(const f32x16& s) {
  float a = fmaxf(fmaxf(s[0], s[1]), s[2]);
  float b = fmaxf(fmaxf(s[3], s[4]), s[5]);
  a = fmaxf(fmaxf(a, s[6]), s[7]);
  b = fmaxf(fmaxf(b, s[8]), s[9]);
  a = fmaxf(fmaxf(a, s[10]), s[11]);
  b = fmaxf(fmaxf(b, s[12]), s[13]);
  a = fmaxf(fmaxf(a, s[14]), s[15]);
  return fmaxf(a, b);
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kDefer = 6.f;   // log2 units: skip the online-softmax rescale while the max grows by < 2^6
constexpr float kLn2 = 0.6931471805599453f;
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

}  // namespace gmlm
