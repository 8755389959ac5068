// K7 backward (CrossAttention geometry, bf16, d = 96): dQ and dK/dV kernels with LDS-DMA staging.
// Reference: autograd of CrossAttention.forward's softmax(q k^T d^-1/2) v (main.py:159-163), flash style: P is recomputed from
// Q, K and the forward's log-sum-exp; delta = rowsum(dO * O) comes from attn_delta_kernel (attn_kernels.hip).
//
// Same arithmetic as attn_bwd_dq_kernel / attn_bwd_dkv_kernel (attn_kernels.hip: folded score chains, see there) - what changes
// is how the streamed tiles reach the MFMAs.  The register-staged kernels hold a whole tile in flight in VGPRs (48 registers
// per thread at d = 96 with 64-row tiles) next to 96-144 accumulator registers: 256 VGPRs, 44-88 bytes of scratch spills in the
// 2- and 4-wave variants, and a ds_write pass per tile.  Here a tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4,
// no VGPR round trip, attn_dma.hpp) into a DUAL-USE image (256-byte rows, slot = chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))):
// the same image is read by rows (ds_read_b128, the S / dP products) and transposed (ds_read_b64_tr_b16, the dQ / dK / dV
// products), both without bank conflicts.  Two buffers per operand, the tile after the current one is requested at the top
// of an iteration, ONE barrier per tile.  No staging registers: no spills, and two waves per SIMD stay resident.
#include "attn_dma.hpp"

#include <type_traits>

namespace gmlm {

template <int D> struct BwdLane {
  static constexpr int NQ = D / 16, DB = D / 32, RP = Img<D>::RP;
  uint32_t rowc[NQ];        // row-operand read (A rows of a 32-row block): k-step s -> byte offset of (row r, chunk 2s + h)
  uint32_t trc[DB][2];      // transposed read: d-block db, rows +0 / +8 -> byte offset of the lane's 8 bytes
  __device__ __forceinline__ void init(uint32_t lds0, int lane) {
    using G = Img<D>;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < NQ; ++s) rowc[s] = lds0 + r * RP + (((2 * s + h) ^ G::fd(r)) << 4);
    const int g1 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int row = 4 * h + qq + 8 * e;
        trc[db][e] = lds0 + row * RP + (((4 * db + 2 * g1 + (pp >> 1)) ^ G::fd(row)) << 4) + 8 * (pp & 1);
      }
  }
};

typedef __attribute__((address_space(3))) const bf16x8* lds_b128_t;
typedef __attribute__((address_space(3))) bf16x4* lds_b64_t;

// acc += A(image rows off .. off + 31 rows, k = D) * B(frag)
template <int D>
__device__ __forceinline__ void mma_rows_img(const BwdLane<D>& c, uint32_t off, const RowFrag<bf16_t, D>& f, f32x16& acc) {
#pragma unroll
  for (int s = 0; s < D / 16; ++s) {
    const bf16x8 a = *(lds_b128_t)(size_t)(c.rowc[s] + off);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, f.v[s], acc, 0, 0, 0);
  }
}
// out[db] += A(image^T: rows = d-block db, k = the 32 image rows at off) * B(x), x = 32 x 32 accumulator tile (row index summed)
template <int D>
__device__ __forceinline__ void mma_acc_img(const BwdLane<D>& c, uint32_t off, const f32x16& x, f32x16 (&out)[D / 32]) {
  constexpr int RP = Img<D>::RP;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)x[8 * s + j];
#pragma unroll
    for (int db = 0; db < D / 32; ++db) {
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(size_t)(c.trc[db][0] + off + (uint32_t)(16 * s * RP)));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(size_t)(c.trc[db][1] + off + (uint32_t)(16 * s * RP)));
      bf16x8 a;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
      out[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, out[db], 0, 0, 0);
    }
  }
}

// request tile t (64 rows from row t * 64 of the slab g) into the LDS buffer at byte offset dst: the register-free form when
// the tile lies wholly inside the slab, the clamping form when the end of the slab cuts it
template <typename Plan>
__device__ __forceinline__ void request_tile(const Plan& plan, const bf16_t* g, int64_t stride, int64_t t, int64_t limit, uint32_t dst, int w, int lane) {
  if ((t + 1) * 64 <= limit) {
#pragma unroll
    for (int k = 0; k < Plan::PER; ++k) plan.piece_fast(k, g + t * 64 * stride, dst, w);
  } else {
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));                    // keeps the per-piece clamped addresses out of the registers held across the loop
    plan.issue(g, stride, t * 64, limit, dst, w, lane_t);
  }
}

// ------------------------------------------------------------------------------------------------
// dQ: workgroup = NW * 32 queries (query on the lane), streams the key tiles
// ------------------------------------------------------------------------------------------------
template <int D, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dq_pipe_kernel(AttnParams p) {
  using T = bf16_t;
  using G = Img<D>;
  constexpr int KT = 64, DB = D / 32, RP = G::RP;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_dyn[];      // K ring [2][TILE], V ring [2][TILE]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem_dyn;
  const int64_t b = blockIdx.y / p.h, hd = blockIdx.y % p.h;
  int64_t lq_, lk_, qbase, kbase, lse_base;
  seq_view(p, b, hd, lq_, lk_, qbase, kbase, lse_base);
  if ((int64_t)blockIdx.x * (NW * 32) >= lq_) return;                 // varlen: tile past this sequence (block-uniform)
  const int64_t q_row = (int64_t)blockIdx.x * (NW * 32) + w * 32 + r;
  const bool q_ok = q_row < lq_;
  const bool wave_live = (int64_t)blockIdx.x * (NW * 32) + w * 32 < lq_;   // else: staging helper only
  int64_t kvlen = lk_;
  if (p.kv_len) { kvlen = p.kv_len[b]; if (kvlen > lk_) kvlen = lk_; if (kvlen < 0) kvlen = 0; }
  const int kvl = (int)(kvlen < (1 << 30) ? kvlen : (1 << 30));
  const int ntiles = (kvl + KT - 1) / KT;
  const T* kg = static_cast<const T*>(p.k) + kbase * p.k_stride + hd * D;
  const T* vg = static_cast<const T*>(p.v) + kbase * p.v_stride + hd * D;
  DmaPlan<D, NW, 2> kd, vd;
  kd.init(p.k_stride, w, lane);
  vd.init(p.v_stride, w, lane);
  if (ntiles > 0) {                                                    // tile 0 first: its latency runs under the operand set-up
    request_tile(kd, kg, p.k_stride, 0, lk_, lds0, w, lane);
    request_tile(vd, vg, p.v_stride, 0, lk_, lds0 + (uint32_t)(2 * G::TILE), w, lane);
  }
  const int64_t qr = q_ok ? q_row : 0;
  RowFrag<T, D> qf, dof;
  qf.load(static_cast<const T*>(p.q) + (qbase + qr) * p.q_stride + hd * D, q_ok, h);
  dof.load(static_cast<const T*>(p.dout) + ((qbase + qr) * p.h + hd) * D, q_ok, h);
  const float sl2 = p.scale * kLog2e;
  const float lse2 = q_ok ? p.lse[lse_base + q_row] * kLog2e : 0x1p100f;       // a row past the sequence: probability 0 everywhere
  const float dl = q_ok ? p.delta[lse_base + q_row] : 0.f;
  const uint32_t dq_u = drop_base(attn_seed(p), lse_base) + (uint32_t)(q_row >> 1) * kDropC1 + (uint32_t)(2 * h) * kDropC2;
  const int q_odd = (int)(q_row & 1);
  // folded chains (attn_bwd_dq_kernel): Q pre-multiplied by scale * log2 e, score chain starts from [1 1 m] x [-lse_hi -lse_lo -2^100],
  // dP chain (no dropout) from [1 1 0] x [-delta_hi -delta_lo 0]
#pragma unroll
  for (int s_ = 0; s_ < D / 16; ++s_)
#pragma unroll
    for (int j = 0; j < 8; ++j) qf.v[s_][j] = (__bf16)((float)qf.v[s_][j] * sl2);
  bf16x8 ones_a, ext_s, ext_dp;
  {
    const float lh = (float)(__bf16)lse2, ll = (float)(__bf16)(lse2 - lh);
    const float dh = (float)(__bf16)dl, dlo = (float)(__bf16)(dl - dh);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ones_a[j] = (__bf16)0.f; ext_s[j] = (__bf16)0.f; ext_dp[j] = (__bf16)0.f; }
    if (h == 0) {
      ones_a[0] = (__bf16)1.f; ones_a[1] = (__bf16)1.f;
      ext_s[0] = (__bf16)(-lh); ext_s[1] = (__bf16)(-ll); ext_s[2] = (__bf16)(-0x1p100f);
      ext_dp[0] = (__bf16)(-dh); ext_dp[1] = (__bf16)(-dlo);
    }
  }
  BwdLane<D> lc;
  lc.init(lds0, lane);
  f32x16 dq[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;
  if (ntiles > 0) { dma_wait(); __syncthreads(); }
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) {                                 // buffer cur ^ 1 was last read in iteration t - 1, behind its closing barrier
      request_tile(kd, kg, p.k_stride, t + 1, lk_, lds0 + (uint32_t)((cur ^ 1) * G::TILE), w, lane);
      request_tile(vd, vg, p.v_stride, t + 1, lk_, lds0 + (uint32_t)((2 + (cur ^ 1)) * G::TILE), w, lane);
    }
    if (wave_live) {
      const int kv0 = t * KT;
#pragma unroll
      for (int kb = 0; kb < KT / 32; ++kb) {
        if (kv0 + kb * 32 >= kvl) break;                  // block-uniform: a block with no valid key
        const uint32_t koffs = (uint32_t)(cur * G::TILE + kb * 32 * RP), voffs = (uint32_t)((2 + cur) * G::TILE + kb * 32 * RP);
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
        const bool full = kv0 + (kb + 1) * 32 <= kvl;     // block-uniform
        bf16x8 oa = ones_a;
        if (!full) oa[2] = (h == 0 && kv0 + kb * 32 + r >= kvl) ? (__bf16)1.f : (__bf16)0.f;   // A row r = key r of the block
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, ext_s, s, 0, 0, 0);
        if (!DROP) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones_a, ext_dp, dp, 0, 0, 0);
        mma_rows_img<D>(lc, koffs, qf, s);
        mma_rows_img<D>(lc, voffs, dof, dp);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const float p0 = fast_exp2(s[i]), p1 = fast_exp2(s[i + 1]);
          if (!DROP) {
            s[i] = p0 * dp[i];
            s[i + 1] = p1 * dp[i + 1];
          } else {
            float m0, m1;
            drop_pair_q(dq_u + ((uint32_t)(t * (KT / 2)) + (uint32_t)((kb * 32 + acc_row(i, 0)) >> 1)) * kDropC2, q_odd, p.drop_thresh, p.keep_scale, m0, m1);
            s[i] = p0 * (dp[i] * m0 - dl);
            s[i + 1] = p1 * (dp[i + 1] * m1 - dl);
          }
        }
        mma_acc_img<D>(lc, koffs, s, dq);
      }
    }
    dma_wait();
    __syncthreads();
  }
  if (q_ok) {
    T* og = static_cast<T*>(p.dq) + (qbase + q_row) * p.dq_stride + hd * D;
#pragma unroll
    for (int d = 0; d < DB; ++d) store_t<T>(og + d * 32, dq[d], p.scale, h);
  }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: workgroup = NW * 32 keys (key on the lane), streams the query tiles (Q and dO images + per-query chain fragments)
// ------------------------------------------------------------------------------------------------
template <int D, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dkv_pipe_kernel(AttnParams p) {
  using T = bf16_t;
  using G = Img<D>;
  constexpr int QT = 64, DB = D / 32, RP = G::RP;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_dyn[];      // Q ring [2][TILE], dO ring [2][TILE], then the small arrays
  T* ext_s = reinterpret_cast<T*>(smem_dyn + 4 * G::TILE);        // [2][QT * 16]: per query row [-lse_hi -lse_lo -2^100 0 x 13]
  T* ext_d = ext_s + 2 * QT * 16;                                  // [2][QT * 16]: [-delta_hi -delta_lo 0 x 14]
  float* dl_s = reinterpret_cast<float*>(ext_d + 2 * QT * 16);     // [2][QT] delta (dropout path: subtracted per score)
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem_dyn;
  const int64_t b = blockIdx.y / p.h, hd = blockIdx.y % p.h;
  int64_t lq_, lk_, qbase, kbase, lse_base;
  seq_view(p, b, hd, lq_, lk_, qbase, kbase, lse_base);
  if ((int64_t)blockIdx.x * (NW * 32) >= lk_) return;                 // varlen: tile past this sequence (block-uniform)
  int64_t kvlen = lk_;
  if (p.kv_len) { kvlen = p.kv_len[b]; if (kvlen > lk_) kvlen = lk_; if (kvlen < 0) kvlen = 0; }
  const int64_t key = (int64_t)blockIdx.x * (NW * 32) + w * 32 + r;
  const bool key_in = key < lk_;            // row exists in memory
  const bool key_ok = key < kvlen;          // takes part in the softmax
  const bool wave_live = (int64_t)blockIdx.x * (NW * 32) + w * 32 < kvlen;   // else: staging helper only (dK = dV = 0)
  const bool block_live = (int64_t)blockIdx.x * (NW * 32) < kvlen;           // a whole workgroup past kv_len has nothing to accumulate
  const int ntiles = block_live ? (int)((lq_ + QT - 1) / QT) : 0;
  const T* qg = static_cast<const T*>(p.q) + qbase * p.q_stride + hd * D;
  const T* dog = static_cast<const T*>(p.dout) + (qbase * p.h + hd) * D;
  const int64_t do_stride = p.h * D;
  DmaPlan<D, NW, 2> qd, dd;
  qd.init(p.q_stride, w, lane);
  dd.init(do_stride, w, lane);
  if (ntiles > 0) {
    request_tile(qd, qg, p.q_stride, 0, lq_, lds0, w, lane);
    request_tile(dd, dog, do_stride, 0, lq_, lds0 + (uint32_t)(2 * G::TILE), w, lane);
  }
  const int64_t kr_ = key_in ? key : 0;
  RowFrag<T, D> kf, vf;
  kf.load(static_cast<const T*>(p.k) + (kbase + kr_) * p.k_stride + hd * D, key_in, h);
  vf.load(static_cast<const T*>(p.v) + (kbase + kr_) * p.v_stride + hd * D, key_in, h);
  const float* lse_g = p.lse + lse_base;
  const float* dl_g = p.delta + lse_base;
  const float sl2 = p.scale * kLog2e;
  const uint32_t dk_u = drop_base(attn_seed(p), lse_base) + (uint32_t)(key >> 1) * kDropC2 + (uint32_t)(2 * h) * kDropC1;
  const int k_odd = (int)(key & 1);
  // K' = bf16(K * scale * log2 e); the key side of the chain steps is [1 1 masked 0 ..]
#pragma unroll
  for (int s_ = 0; s_ < D / 16; ++s_)
#pragma unroll
    for (int j = 0; j < 8; ++j) kf.v[s_][j] = (__bf16)((float)kf.v[s_][j] * sl2);
  bf16x8 ones_k, ones_2;
#pragma unroll
  for (int j = 0; j < 8; ++j) { ones_k[j] = (__bf16)0.f; ones_2[j] = (__bf16)0.f; }
  if (h == 0) {
    ones_k[0] = ones_2[0] = (__bf16)1.f; ones_k[1] = ones_2[1] = (__bf16)1.f;
    ones_k[2] = key_ok ? (__bf16)0.f : (__bf16)1.f;        // this lane's key is masked: its scores get -2^100
  }
  BwdLane<D> lc;
  lc.init(lds0, lane);
  f32x16 dk[DB], dv[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[d][i] = 0.f; dv[d][i] = 0.f; }
  // per-query chain fragments of a tile: loaded by the first QT threads one tile ahead, written to LDS ahead of the closing barrier
  float lr = 0.f, dr = 0.f;
  auto load_small = [&](int64_t q0) {
    if (tid < QT) {
      const bool ok = q0 + tid < lq_;
      lr = ok ? lse_g[q0 + tid] * kLog2e : 0x1p100f;       // a row past the sequence: probability 0
      dr = ok ? dl_g[q0 + tid] : 0.f;
    }
  };
  auto store_small = [&](int buf) {
    if (tid < QT) {
      const float lh = (float)(__bf16)lr, ll = (float)(__bf16)(lr - lh);
      const float dh = (float)(__bf16)dr, dlo = (float)(__bf16)(dr - dh);
      bf16x8 es, ed, z;
#pragma unroll
      for (int j = 0; j < 8; ++j) { es[j] = (__bf16)0.f; ed[j] = (__bf16)0.f; z[j] = (__bf16)0.f; }
      es[0] = (__bf16)(-lh); es[1] = (__bf16)(-ll); es[2] = (__bf16)(-0x1p100f);
      ed[0] = (__bf16)(-dh); ed[1] = (__bf16)(-dlo);
      *reinterpret_cast<bf16x8*>(ext_s + (buf * QT + tid) * 16) = es;
      *reinterpret_cast<bf16x8*>(ext_s + (buf * QT + tid) * 16 + 8) = z;
      *reinterpret_cast<bf16x8*>(ext_d + (buf * QT + tid) * 16) = ed;
      *reinterpret_cast<bf16x8*>(ext_d + (buf * QT + tid) * 16 + 8) = z;
      dl_s[buf * QT + tid] = dr;
    }
  };
  if (ntiles > 0) {
    load_small(0);
    store_small(0);
    dma_wait();
    __syncthreads();
  }
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const int64_t q0 = (int64_t)t * QT;
    const bool more = t + 1 < ntiles;
    if (more) {
      request_tile(qd, qg, p.q_stride, t + 1, lq_, lds0 + (uint32_t)((cur ^ 1) * G::TILE), w, lane);
      request_tile(dd, dog, do_stride, t + 1, lq_, lds0 + (uint32_t)((2 + (cur ^ 1)) * G::TILE), w, lane);
      load_small(q0 + QT);
    }
    if (wave_live) {
#pragma unroll
      for (int qb = 0; qb < QT / 32; ++qb) {
        if (q0 + qb * 32 >= lq_) break;                    // block-uniform
        const uint32_t qoffs = (uint32_t)(cur * G::TILE + qb * 32 * RP), dooffs = (uint32_t)((2 + cur) * G::TILE + qb * 32 * RP);
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
        const bf16x8 ea = *reinterpret_cast<const bf16x8*>(ext_s + (cur * QT + qb * 32 + r) * 16 + 8 * h);   // A row r = query r of the block
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ea, ones_k, s, 0, 0, 0);
        if (!DROP) {
          const bf16x8 da = *reinterpret_cast<const bf16x8*>(ext_d + (cur * QT + qb * 32 + r) * 16 + 8 * h);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, ones_2, dp, 0, 0, 0);
        }
        mma_rows_img<D>(lc, qoffs, kf, s);
        mma_rows_img<D>(lc, dooffs, vf, dp);
        if constexpr (!DROP) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float pr = fast_exp2(s[i]);
            s[i] = pr;
            dp[i] = pr * dp[i];
          }
        } else {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int q4 = qb * 32 + 8 * g4 + 4 * h;       // accumulator registers 4*g4 .. 4*g4+3 = queries q4 .. q4+3
            const float4 d4 = *reinterpret_cast<const float4*>(dl_s + cur * QT + q4);
            const float dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
              const int i = 4 * g4 + j;
              const float p0 = fast_exp2(s[i]), p1 = fast_exp2(s[i + 1]);
              float m0, m1;
              drop_pair_k(dk_u + ((uint32_t)(q0 >> 1) + (uint32_t)((qb * 32 + 8 * g4 + j) >> 1)) * kDropC1, k_odd, p.drop_thresh, p.keep_scale, m0, m1);
              s[i] = p0 * m0;
              s[i + 1] = p1 * m1;
              dp[i] = p0 * (dp[i] * m0 - dv4[j]);
              dp[i + 1] = p1 * (dp[i + 1] * m1 - dv4[j + 1]);
            }
          }
        }
        mma_acc_img<D>(lc, dooffs, s, dv);
        mma_acc_img<D>(lc, qoffs, dp, dk);
      }
    }
    if (more) store_small(cur ^ 1);
    dma_wait();
    __syncthreads();
  }
  if (key_in) {
    T* dkg = static_cast<T*>(p.dk) + (kbase + key) * p.dk_stride + hd * D;
    T* dvg = static_cast<T*>(p.dv) + (kbase + key) * p.dv_stride + hd * D;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      store_t<T>(dkg + d * 32, dk[d], p.scale, h);
      store_t<T>(dvg + d * 32, dv[d], 1.f, h);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int D, int NW>
static int launch_bwd_pipe(const AttnParams& p, int64_t rows_q, int64_t rows_k, int64_t bh, hipStream_t st) {
  constexpr int kLdsQ = 4 * Img<D>::TILE;
  constexpr int kLdsKV = 4 * Img<D>::TILE + 2 * 2 * 64 * 16 * (int)sizeof(bf16_t) + 2 * 64 * (int)sizeof(float);
  static PerDeviceOnce once;
  const int rc = once([&]() -> int {
    GMLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_pipe_kernel<D, NW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsQ));
    GMLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_pipe_kernel<D, NW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsQ));
    GMLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_pipe_kernel<D, NW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsKV));
    GMLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_pipe_kernel<D, NW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsKV));
    return GMLM_OK;
  });
  if (rc != GMLM_OK) return rc;
  const dim3 gq((unsigned)cdiv(rows_q, NW * 32), (unsigned)bh), gk((unsigned)cdiv(rows_k, NW * 32), (unsigned)bh);
  if (p.drop_thresh) attn_bwd_dq_pipe_kernel<D, NW, true><<<gq, NW * 64, kLdsQ, st>>>(p);
  else attn_bwd_dq_pipe_kernel<D, NW, false><<<gq, NW * 64, kLdsQ, st>>>(p);
  GMLM_LAUNCH_CHECK();
  if (p.drop_thresh) attn_bwd_dkv_pipe_kernel<D, NW, true><<<gk, NW * 64, kLdsKV, st>>>(p);
  else attn_bwd_dkv_pipe_kernel<D, NW, false><<<gk, NW * 64, kLdsKV, st>>>(p);
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

// bf16, d = 96 backward after the delta kernel; nw = 4 or 8 waves per workgroup
int attn_bwd_pipe_launch(const AttnParams& p, int d, int nw, int64_t rows_q, int64_t rows_k, int64_t bh, hipStream_t st) {
  if (d == 96 && nw == 8) return launch_bwd_pipe<96, 8>(p, rows_q, rows_k, bh, st);
  if (d == 96 && nw == 4) return launch_bwd_pipe<96, 4>(p, rows_q, rows_k, bh, st);
  set_error("attention_bwd: no pipelined kernel for d = %d with %d waves", d, nw);
  return GMLM_EINVAL;
}

}  // namespace gmlm
