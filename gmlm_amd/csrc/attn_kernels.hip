// K5/K7 — multi-head attention core with streaming softmax, forward + backward, gfx950 MFMA.
// Reference: K5 = BertSelfAttention's attention call softmax(QK^T*d^-1/2 + padding mask)V
// (hf:modeling_bert.py:188-201, eager form 111-136); K7 = CrossAttention.forward's dense
// [1,8,N,N] softmax over all nodes (main.py:159-163).
//
// Design (wave64, one process per GPU, no score matrix in HBM):
//  * forward: a workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 rows.
//    K/V tiles of 64 keys are staged row-major in LDS.  Scores are computed TRANSPOSED,
//    S^T = K Q^T, with v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact f32):
//    the query column then sits on the lane and the 32 keys in the accumulator registers, so the
//    row max / row sum are register reductions plus one half-wave exchange, the running rescale is
//    lane-local, and the probability tile feeds the next MFMA (O^T += V^T P^T) straight from the
//    accumulator registers as its B operand (no LDS round trip).  V^T fragments come from the same
//    row-major LDS image through ds_read_b64_tr_b16 (hardware transpose read).
//  * backward = three launches, no atomics (deterministic): delta = rowsum(dO*O); a dQ kernel
//    (workgroup = 128 queries, loops over key tiles) and a dK/dV kernel (workgroup = 128 keys,
//    loops over query tiles).  P is recomputed from Q, K and the forward's log-sum-exp.
//  * padding: keys >= kv_len[b] get probability exactly 0 (the reference adds finfo.min and
//    softmaxes: the same zeros in fp32); whole tiles past kv_len are skipped.
#include "attn_common.hpp"
#include "colreduce.hpp"

#include <type_traits>

namespace gmlm {

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <typename T, int D, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(AttnParams p) {
  constexpr bool DBUF = sizeof(T) == 2;
  constexpr int KT = 64, NT = NW * 64, PITCH = D + Pad<T>::v, DB = D / 32, NBUF = DBUF ? 2 : 1;
  __shared__ __attribute__((aligned(16))) T ks[NBUF][KT * PITCH];
  __shared__ __attribute__((aligned(16))) T vs[NBUF][KT * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int64_t b = blockIdx.y / p.h, hd = blockIdx.y % p.h;
  int64_t lq_, lk_, qbase, kbase, lse_base;
  seq_view(p, b, hd, lq_, lk_, qbase, kbase, lse_base);
  if ((int64_t)blockIdx.x * (NW * 32) >= (false ? lk_ : lq_)) return;   // varlen: tile past this sequence (block-uniform)
  const int64_t q_row = (int64_t)blockIdx.x * (NW * 32) + w * 32 + r;
  const bool q_ok = q_row < lq_;
  // a wave whose 32 queries all lie past the sequence (short packed sequences) only helps staging K/V
  const bool wave_live = (int64_t)blockIdx.x * (NW * 32) + w * 32 < lq_;
  int64_t kvlen = lk_;
  if (p.kv_len) { kvlen = p.kv_len[b]; if (kvlen > lk_) kvlen = lk_; if (kvlen < 0) kvlen = 0; }
  const T* qg = static_cast<const T*>(p.q) + (qbase + (q_ok ? q_row : 0)) * p.q_stride + hd * D;
  const T* kg = static_cast<const T*>(p.k) + kbase * p.k_stride + hd * D;
  const T* vg = static_cast<const T*>(p.v) + kbase * p.v_stride + hd * D;
  RowFrag<T, D> qf;
  qf.load(qg, q_ok, h);
  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  float m = -INFINITY, l = 0.f;              // running max / sum in the log2 domain (scores * scale * log2 e)
  const float sl2 = p.scale * kLog2e;
  // dropout hash input of this lane's query (+ 2h: the lane half's keys sit 4 further on, i.e. 2 key pairs)
  const uint32_t dq_u = drop_base(attn_seed(p), lse_base) + (uint32_t)(q_row >> 1) * kDropC1 + (uint32_t)(2 * h) * kDropC2;
  const int q_odd = (int)(q_row & 1);
  const int ntiles = (int)((kvlen + KT - 1) / KT);
  TileRegs<T, D, KT, NT> kr, vr;
  kr.init(p.k_stride, tid);
  vr.init(p.v_stride, tid);
  if (ntiles > 0) {
    kr.load(kg, p.k_stride, 0, lk_, tid);
    vr.load(vg, p.v_stride, 0, lk_, tid);
    kr.store(ks[0], tid);
    vr.store(vs[0], tid);
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int64_t kv0 = (int64_t)t * KT;
    const int cur = DBUF ? (t & 1) : 0;
    const bool more = t + 1 < ntiles;
    if (DBUF && more) {
      kr.load(kg, p.k_stride, kv0 + KT, lk_, tid);
      vr.load(vg, p.v_stride, kv0 + KT, lk_, tid);
    }
    if (wave_live) {
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
      mma_rows<D>(ks[cur], PITCH, kb * 32, qf, s[kb], r, h);
    }
    float rmax = -INFINITY;
    if (kv0 + KT <= kvlen) {                        // whole tile valid (block-uniform): no per-key compare
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) rmax = fmaxf(rmax, s[kb][i]);
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t key = kv0 + kb * 32 + acc_row(i, h);
          const float v = key < kvlen ? s[kb][i] : -INFINITY;
          s[kb][i] = v;
          rmax = fmaxf(rmax, v);
        }
    }
    rmax = xhalf_max(rmax);
    // deferred rescale: keep the old reference max while the new one exceeds it by < 2^kDefer (probabilities
    // then stay <= 2^kDefer, harmless in fp32 / bf16); the O and l rescale is skipped for the whole wave then
    float m_new = fmaxf(m, rmax * sl2);             // finite: key kv0 is always valid
    const bool keep = __all(m_new - m <= kDefer);   // m = -inf (first tile) -> false
    float alpha = 1.f;
    if (keep) m_new = m;
    else alpha = fast_exp2(m - m_new);              // m = -inf -> 0
    float rsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const float e0 = fast_exp2(fmaf(s[kb][i], sl2, -m_new));       // masked: exp2(-inf) = 0
        const float e1 = fast_exp2(fmaf(s[kb][i + 1], sl2, -m_new));
        rsum += e0 + e1;                                               // the normaliser uses the un-dropped probabilities
        if (DROP) {
          float m0, m1;                                // keys (k, k+1): (kv0 + kb*32 + acc_row(i, 0)) >> 1 pairs on from the tile start
          drop_pair_q(dq_u + ((uint32_t)(t * (KT / 2)) + (uint32_t)((kb * 32 + acc_row(i, 0)) >> 1)) * kDropC2, q_odd, p.drop_thresh, p.keep_scale, m0, m1);
          s[kb][i] = e0 * m0;
          s[kb][i + 1] = e1 * m1;
        } else {
          s[kb][i] = e0;
          s[kb][i + 1] = e1;
        }
      }
    rsum = xhalf_sum(rsum);
    l = l * alpha + rsum;
    m = m_new;
    if (!keep) {
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) mma_acc<D>(vs[cur], PITCH, kb * 32, s[kb], o, lane);
    }
    if (DBUF) {
      if (more) {
        kr.store(ks[cur ^ 1], tid);
        vr.store(vs[cur ^ 1], tid);
      }
      __syncthreads();
    } else if (more) {
      __syncthreads();
      kr.load(kg, p.k_stride, kv0 + KT, lk_, tid);
      vr.load(vg, p.v_stride, kv0 + KT, lk_, tid);
      kr.store(ks[0], tid);
      vr.store(vs[0], tid);
      __syncthreads();
    }
  }
  if (q_ok) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    T* og = static_cast<T*>(p.o_w) + ((qbase + q_row) * p.h + hd) * D;
#pragma unroll
    for (int d = 0; d < DB; ++d) store_t<T>(og + d * 32, o[d], inv, h);
    if constexpr (sizeof(T) == 2) {
      if (p.o_lo_w) {
        T* olg = static_cast<T*>(p.o_lo_w) + ((qbase + q_row) * p.h + hd) * D;
#pragma unroll
        for (int d = 0; d < DB; ++d) store_t_lo(olg + d * 32, o[d], inv, h);
      }
    }
    if (h == 0 && p.lse_w) p.lse_w[lse_base + q_row] = l > 0.f ? (m + __log2f(l)) * kLn2 : -INFINITY;
  }
}

// ------------------------------------------------------------------------------------------------
// backward: delta[b,h,q] = sum_d dO * O   (O = o + o_lo when the forward stored the bf16 residual of its output)
// ------------------------------------------------------------------------------------------------
// a group of d/V lanes owns one (b, q, h) row: the wave reads 64 x 16 contiguous bytes per instruction
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ o_lo, const T* __restrict__ dout,
                                                          int64_t rows /* b*lq*h */, int d, int64_t lq, int64_t hn,
                                                          float* __restrict__ delta) {
  constexpr int V = Store<T>::kVec;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lanes = d / V <= 8 ? 8 : (d / V <= 16 ? 16 : 32);
  const int64_t row = gid / lanes;
  const int c = (int)(gid % lanes);
  float acc = 0.f;
  if (row < rows && c < d / V) {
    float a[V], g[V];
    Store<T>::ldv(o + row * d + c * V, a);
    Store<T>::ldv(dout + row * d + c * V, g);
    if (o_lo) {
      float al[V];
      Store<T>::ldv(o_lo + row * d + c * V, al);
#pragma unroll
      for (int v = 0; v < V; ++v) a[v] += al[v];
    }
#pragma unroll
    for (int v = 0; v < V; ++v) acc += a[v] * g[v];
  }
  for (int off = lanes >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (row < rows && c == 0) {
    const int64_t hd = row % hn, bq = row / hn, q = bq % lq, b = bq / lq;
    delta[(b * hn + hd) * lq + q] = acc;         // packed mode passes b = 1, lq = total rows: [h, total_rows]
  }
}

// ------------------------------------------------------------------------------------------------
// backward: dQ.  workgroup = NW*32 queries, loop over key tiles.
//   S^T = K Q^T, P^T = exp(scale*S^T - lse[q]), dP^T = V dO^T, dS^T = P^T*(dP^T - delta[q]),
//   dQ^T += K^T dS^T  (all with the query on the lane)
// ------------------------------------------------------------------------------------------------
template <typename T, int D, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(AttnParams p) {
  constexpr bool DBUF = sizeof(T) == 2;
  constexpr int KT = DBUF ? (NW == 8 ? 128 : 64) : 32;   // 8-wave blocks own the CU (202 VGPRs): 128-key tiles halve their barriers (-4 %)
  constexpr int KB = KT / 32, NT = NW * 64, PITCH = D + Pad<T>::v, DB = D / 32, NBUF = DBUF ? 2 : 1;
  __shared__ __attribute__((aligned(16))) T ks[NBUF][KT * PITCH];
  __shared__ __attribute__((aligned(16))) T vs[NBUF][KT * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int64_t b = blockIdx.y / p.h, hd = blockIdx.y % p.h;
  int64_t lq_, lk_, qbase, kbase, lse_base;
  seq_view(p, b, hd, lq_, lk_, qbase, kbase, lse_base);
  if ((int64_t)blockIdx.x * (NW * 32) >= (false ? lk_ : lq_)) return;   // varlen: tile past this sequence (block-uniform)
  const int64_t q_row = (int64_t)blockIdx.x * (NW * 32) + w * 32 + r;
  const bool q_ok = q_row < lq_;
  const bool wave_live = (int64_t)blockIdx.x * (NW * 32) + w * 32 < lq_;   // else: staging helper only
  int64_t kvlen = lk_;
  if (p.kv_len) { kvlen = p.kv_len[b]; if (kvlen > lk_) kvlen = lk_; if (kvlen < 0) kvlen = 0; }
  const int64_t qr = q_ok ? q_row : 0;
  const T* kg = static_cast<const T*>(p.k) + kbase * p.k_stride + hd * D;
  const T* vg = static_cast<const T*>(p.v) + kbase * p.v_stride + hd * D;
  RowFrag<T, D> qf, dof;
  qf.load(static_cast<const T*>(p.q) + (qbase + qr) * p.q_stride + hd * D, q_ok, h);
  dof.load(static_cast<const T*>(p.dout) + ((qbase + qr) * p.h + hd) * D, q_ok, h);
  const float sl2 = p.scale * kLog2e;
  const float lse2 = q_ok ? p.lse[lse_base + q_row] * kLog2e : INFINITY;
  const float dl = q_ok ? p.delta[lse_base + q_row] : 0.f;
  const int kvl = (int)(kvlen < (1 << 30) ? kvlen : (1 << 30));
  const uint32_t dq_u = drop_base(attn_seed(p), lse_base) + (uint32_t)(q_row >> 1) * kDropC1 + (uint32_t)(2 * h) * kDropC2;
  const int q_odd = (int)(q_row & 1);
  // bf16: the per-score fma / compare / select work rides in the MFMA chains (as in the pipelined forward).  Q is
  // pre-multiplied by scale * log2 e (the SAME rounding the forward applies), and each chain starts with one extra
  // contraction step  [1 1 m 0 ..] x [-lse_hi -lse_lo -2^100 0 ..]^T  (m = 1 for a masked key): scores leave the
  // chain as s*scale*log2e - lse, masked keys at about -1.3e30, so P = exp2(.) with no further arithmetic; without
  // dropout the dP chain starts from  [1 1 0 ..] x [-delta_hi -delta_lo 0 ..]^T  and dS = P * (dP - delta) is one multiply.
  // lse / delta are split into two bf16 terms (hi + lo: relative error 2^-17).
  constexpr bool FOLD = sizeof(T) == 2;
  bf16x8 ones_a, ext_s, ext_dp;
  if constexpr (FOLD) {
#pragma unroll
    for (int s_ = 0; s_ < D / 16; ++s_)
#pragma unroll
      for (int j = 0; j < 8; ++j) qf.v[s_][j] = (__bf16)((float)qf.v[s_][j] * sl2);
    const float l2 = q_ok ? lse2 : 0x1p100f;                 // a row past the sequence: probability 0 everywhere
    const float lh = (float)(__bf16)l2, ll = (float)(__bf16)(l2 - lh);
    const float dh = (float)(__bf16)dl, dlo = (float)(__bf16)(dl - dh);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ones_a[j] = (__bf16)0.f; ext_s[j] = (__bf16)0.f; ext_dp[j] = (__bf16)0.f; }
    if (h == 0) {
      ones_a[0] = (__bf16)1.f; ones_a[1] = (__bf16)1.f;
      ext_s[0] = (__bf16)(-lh); ext_s[1] = (__bf16)(-ll); ext_s[2] = (__bf16)(-0x1p100f);
      ext_dp[0] = (__bf16)(-dh); ext_dp[1] = (__bf16)(-dlo);
    }
  }
  f32x16 dq[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;
  const int ntiles = (int)((kvlen + KT - 1) / KT);
  TileRegs<T, D, KT, NT> kr, vr;
  kr.init(p.k_stride, tid);
  vr.init(p.v_stride, tid);
  if (ntiles > 0) {
    kr.load(kg, p.k_stride, 0, lk_, tid);
    vr.load(vg, p.v_stride, 0, lk_, tid);
    kr.store(ks[0], tid);
    vr.store(vs[0], tid);
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int64_t kv0 = (int64_t)t * KT;
    const int cur = DBUF ? (t & 1) : 0;
    const bool more = t + 1 < ntiles;
    if (DBUF && more) {
      kr.load(kg, p.k_stride, kv0 + KT, lk_, tid);
      vr.load(vg, p.v_stride, kv0 + KT, lk_, tid);
    }
    if (wave_live) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
      const bool full = kv0 + (kb + 1) * 32 <= kvlen;          // block-uniform
      if constexpr (FOLD) {
        bf16x8 oa = ones_a;
        if (!full) oa[2] = (h == 0 && (int)kv0 + kb * 32 + r >= kvl) ? (__bf16)1.f : (__bf16)0.f;   // A row r = key r of the block
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, ext_s, s, 0, 0, 0);
        if (!DROP) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones_a, ext_dp, dp, 0, 0, 0);
      }
      mma_rows<D>(ks[cur], PITCH, kb * 32, qf, s, r, h);
      mma_rows<D>(vs[cur], PITCH, kb * 32, dof, dp, r, h);
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        float p0, p1;
        if constexpr (FOLD) {
          p0 = fast_exp2(s[i]);
          p1 = fast_exp2(s[i + 1]);
        } else {
          const int key = (int)kv0 + kb * 32 + acc_row(i, h);      // even; registers i, i+1 = keys key, key+1
          // a masked key gets exponent -inf through a select (no branch around the exp); whole blocks skip the test;
          // a query row past the sequence has lse = +inf, i.e. probability 0 everywhere
          float a0 = fmaf(s[i], sl2, -lse2), a1 = fmaf(s[i + 1], sl2, -lse2);
          if (!full) { a0 = key < kvl ? a0 : -INFINITY; a1 = key + 1 < kvl ? a1 : -INFINITY; }
          p0 = fast_exp2(a0); p1 = fast_exp2(a1);
        }
        if (FOLD && !DROP) {
          s[i] = p0 * dp[i];
          s[i + 1] = p1 * dp[i + 1];
        } else {
          float m0 = 1.f, m1 = 1.f;
          if (DROP) drop_pair_q(dq_u + ((uint32_t)(t * (KT / 2)) + (uint32_t)((kb * 32 + acc_row(i, 0)) >> 1)) * kDropC2, q_odd, p.drop_thresh, p.keep_scale, m0, m1);
          s[i] = p0 * (dp[i] * m0 - dl);
          s[i + 1] = p1 * (dp[i + 1] * m1 - dl);
        }
      }
      mma_acc<D>(ks[cur], PITCH, kb * 32, s, dq, lane);
    }
    }
    if (DBUF) {
      if (more) {
        kr.store(ks[cur ^ 1], tid);
        vr.store(vs[cur ^ 1], tid);
      }
      __syncthreads();
    } else if (more) {
      __syncthreads();
      kr.load(kg, p.k_stride, kv0 + KT, lk_, tid);
      vr.load(vg, p.v_stride, kv0 + KT, lk_, tid);
      kr.store(ks[0], tid);
      vr.store(vs[0], tid);
      __syncthreads();
    }
  }
  if (q_ok) {
    T* og = static_cast<T*>(p.dq) + (qbase + q_row) * p.dq_stride + hd * D;
#pragma unroll
    for (int d = 0; d < DB; ++d) store_t<T>(og + d * 32, dq[d], p.scale, h);
  }
}

// ------------------------------------------------------------------------------------------------
// backward: dK, dV.  workgroup = 128 keys (32 per wave), loop over query tiles of 32.
//   S = Q K^T, P = exp(scale*S - lse[q]), dV^T += dO^T P, dP = dO V^T, dS = P*(dP - delta[q]),
//   dK^T += Q^T dS   (all with the key on the lane, the query in the accumulator registers)
// ------------------------------------------------------------------------------------------------
// QT = query rows staged per barrier: 32 with two LDS buffers (long sequences: loads of tile t+1 under the MFMAs
// of tile t), or 128 in one buffer for sequences of <= 128 tokens (the whole sequence behind ONE barrier: at
// 16-128 tokens the per-tile barrier / LDS round trip, not the MFMA work, sets the block's lifetime).
template <typename T, int D, int NW, bool DROP, int QT = 32>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(AttnParams p) {
  constexpr bool DBUF = sizeof(T) == 2 && QT <= 64;
  constexpr int NT = NW * 64, PITCH = D + Pad<T>::v, DB = D / 32, NBUF = DBUF ? 2 : 1;
  __shared__ __attribute__((aligned(16))) T qs[NBUF][QT * PITCH];
  __shared__ __attribute__((aligned(16))) T dos[NBUF][QT * PITCH];
  __shared__ __attribute__((aligned(16))) float lse_s[NBUF][QT];
  __shared__ __attribute__((aligned(16))) float dl_s[NBUF][QT];
  // bf16: per query row, the A fragments of the extra contraction steps (see attn_bwd_dq_kernel): 16 elements each,
  // [-lse_hi -lse_lo -2^100 0 x 13] and [-delta_hi -delta_lo 0 x 14]; the key side holds [1 1 masked 0 ..] in registers
  constexpr bool FOLD = sizeof(T) == 2;
  __shared__ __attribute__((aligned(16))) bf16_t ext_s[FOLD ? NBUF : 1][FOLD ? QT * 16 : 8];
  __shared__ __attribute__((aligned(16))) bf16_t ext_d[FOLD ? NBUF : 1][FOLD ? QT * 16 : 8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int64_t b = blockIdx.y / p.h, hd = blockIdx.y % p.h;
  int64_t lq_, lk_, qbase, kbase, lse_base;
  seq_view(p, b, hd, lq_, lk_, qbase, kbase, lse_base);
  if ((int64_t)blockIdx.x * (NW * 32) >= (true ? lk_ : lq_)) return;   // varlen: tile past this sequence (block-uniform)
  int64_t kvlen = lk_;
  if (p.kv_len) { kvlen = p.kv_len[b]; if (kvlen > lk_) kvlen = lk_; if (kvlen < 0) kvlen = 0; }
  const int64_t key = (int64_t)blockIdx.x * (NW * 32) + w * 32 + r;
  const bool key_in = key < lk_;           // row exists in memory
  const bool key_ok = key < kvlen;          // takes part in the softmax
  const bool wave_live = (int64_t)blockIdx.x * (NW * 32) + w * 32 < kvlen;   // else: staging helper only (dK = dV = 0)
  const int64_t kr_ = key_in ? key : 0;
  RowFrag<T, D> kf, vf;
  kf.load(static_cast<const T*>(p.k) + (kbase + kr_) * p.k_stride + hd * D, key_in, h);
  vf.load(static_cast<const T*>(p.v) + (kbase + kr_) * p.v_stride + hd * D, key_in, h);
  const T* qg = static_cast<const T*>(p.q) + qbase * p.q_stride + hd * D;
  const T* dog = static_cast<const T*>(p.dout) + (qbase * p.h + hd) * D;
  const float* lse_g = p.lse + lse_base;
  const float* dl_g = p.delta + lse_base;
  const float sl2 = p.scale * kLog2e;
  const uint32_t dk_u = drop_base(attn_seed(p), lse_base) + (uint32_t)(key >> 1) * kDropC2 + (uint32_t)(2 * h) * kDropC1;   // + 2h: the lane half's queries sit 4 further on
  const int k_odd = (int)(key & 1);
  bf16x8 ones_k, ones_2;
  if constexpr (FOLD) {
    // K' = bf16(K * scale * log2 e) (one more bf16 rounding, the size of the one the forward applies to Q)
#pragma unroll
    for (int s_ = 0; s_ < D / 16; ++s_)
#pragma unroll
      for (int j = 0; j < 8; ++j) kf.v[s_][j] = (__bf16)((float)kf.v[s_][j] * sl2);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ones_k[j] = (__bf16)0.f; ones_2[j] = (__bf16)0.f; }
    if (h == 0) {
      ones_k[0] = ones_2[0] = (__bf16)1.f; ones_k[1] = ones_2[1] = (__bf16)1.f;
      ones_k[2] = key_ok ? (__bf16)0.f : (__bf16)1.f;        // this lane's key is masked: its scores get -2^100
    }
  }
  f32x16 dk[DB], dv[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[d][i] = 0.f; dv[d][i] = 0.f; }
  // a whole workgroup past kv_len has nothing to accumulate (block-uniform condition)
  const bool block_live = (int64_t)blockIdx.x * (NW * 32) < kvlen;
  if (block_live) {
    const int ntiles = (int)((lq_ + QT - 1) / QT);
    TileRegs<T, D, QT, NT> qr, dor;
    qr.init(p.q_stride, tid);
    dor.init(p.h * D, tid);
    float lr = 0.f, dr = 0.f;
    auto load_small = [&](int64_t q0) {
      if (tid < QT) {
        const bool ok = q0 + tid < lq_;
        lr = ok ? lse_g[q0 + tid] * kLog2e : INFINITY;     // +inf: probability 0 for a row past the sequence
        dr = ok ? dl_g[q0 + tid] : 0.f;
      }
    };
    auto store_small = [&](int buf) {
      if (tid < QT) {
        lse_s[buf][tid] = lr; dl_s[buf][tid] = dr;
        if constexpr (FOLD) {
          const float l2 = lr < 0x1p100f ? lr : 0x1p100f;            // +inf (row past the sequence): probability 0
          const float lh = (float)(__bf16)l2, ll = (float)(__bf16)(l2 - lh);
          const float dh = (float)(__bf16)dr, dlo = (float)(__bf16)(dr - dh);
          bf16x8 es, ed, z;
#pragma unroll
          for (int j = 0; j < 8; ++j) { es[j] = (__bf16)0.f; ed[j] = (__bf16)0.f; z[j] = (__bf16)0.f; }
          es[0] = (__bf16)(-lh); es[1] = (__bf16)(-ll); es[2] = (__bf16)(-0x1p100f);
          ed[0] = (__bf16)(-dh); ed[1] = (__bf16)(-dlo);
          *reinterpret_cast<bf16x8*>(&ext_s[buf][tid * 16]) = es;
          *reinterpret_cast<bf16x8*>(&ext_s[buf][tid * 16 + 8]) = z;
          *reinterpret_cast<bf16x8*>(&ext_d[buf][tid * 16]) = ed;
          *reinterpret_cast<bf16x8*>(&ext_d[buf][tid * 16 + 8]) = z;
        }
      }
    };
    if (ntiles > 0) {
      qr.load(qg, p.q_stride, 0, lq_, tid);
      dor.load(dog, p.h * D, 0, lq_, tid);
      load_small(0);
      qr.store(qs[0], tid);
      dor.store(dos[0], tid);
      store_small(0);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
      const int64_t q0 = (int64_t)t * QT;
      const int cur = DBUF ? (t & 1) : 0;
      const bool more = t + 1 < ntiles;
      if (DBUF && more) {
        qr.load(qg, p.q_stride, q0 + QT, lq_, tid);
        dor.load(dog, p.h * D, q0 + QT, lq_, tid);
        load_small(q0 + QT);
      }
      if (wave_live) {
#pragma unroll
      for (int qb = 0; qb < QT / 32; ++qb) {
        if (QT > 32 && q0 + qb * 32 >= lq_) break;            // block-uniform
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
        if constexpr (FOLD) {
          const bf16x8 ea = *reinterpret_cast<const bf16x8*>(&ext_s[cur][(qb * 32 + r) * 16 + 8 * h]);   // A row r = query r of the block
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ea, ones_k, s, 0, 0, 0);
          if (!DROP) {
            const bf16x8 da = *reinterpret_cast<const bf16x8*>(&ext_d[cur][(qb * 32 + r) * 16 + 8 * h]);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, ones_2, dp, 0, 0, 0);
          }
        }
        mma_rows<D>(qs[cur], PITCH, qb * 32, kf, s, r, h);
        mma_rows<D>(dos[cur], PITCH, qb * 32, vf, dp, r, h);
        if constexpr (FOLD && !DROP) {
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
          const float4 l4 = *reinterpret_cast<const float4*>(&lse_s[cur][q4]);
          const float4 d4 = *reinterpret_cast<const float4*>(&dl_s[cur][q4]);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const int i = 4 * g4 + j;                        // registers i, i+1 = queries q0 + q4 + j, + 1 (even, odd)
            // query rows past the sequence carry lse = +inf in LDS, masked keys select the exponent -inf: both give
            // probability exactly 0 without a branch around the exp
            const float p0 = FOLD ? fast_exp2(s[i]) : fast_exp2(key_ok ? fmaf(s[i], sl2, -lv[j]) : -INFINITY);
            const float p1 = FOLD ? fast_exp2(s[i + 1]) : fast_exp2(key_ok ? fmaf(s[i + 1], sl2, -lv[j + 1]) : -INFINITY);
            float m0 = 1.f, m1 = 1.f;
            if (DROP) drop_pair_k(dk_u + ((uint32_t)(q0 >> 1) + (uint32_t)((qb * 32 + 8 * g4 + j) >> 1)) * kDropC1, k_odd, p.drop_thresh, p.keep_scale, m0, m1);
            s[i] = p0 * m0;
            s[i + 1] = p1 * m1;
            dp[i] = p0 * (dp[i] * m0 - dv4[j]);
            dp[i + 1] = p1 * (dp[i + 1] * m1 - dv4[j + 1]);
          }
        }
        }
        mma_acc<D>(dos[cur], PITCH, qb * 32, s, dv, lane);
        mma_acc<D>(qs[cur], PITCH, qb * 32, dp, dk, lane);
      }
      }
      if (DBUF) {
        if (more) {
          qr.store(qs[cur ^ 1], tid);
          dor.store(dos[cur ^ 1], tid);
          store_small(cur ^ 1);
        }
        __syncthreads();
      } else if (more) {
        __syncthreads();
        qr.load(qg, p.q_stride, q0 + QT, lq_, tid);
        dor.load(dog, p.h * D, q0 + QT, lq_, tid);
        load_small(q0 + QT);
        qr.store(qs[0], tid);
        dor.store(dos[0], tid);
        store_small(0);
        __syncthreads();
      }
    }
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

// 8-bit dropout threshold: P(drop) = th / 256 (|error| <= 2e-3: 0.1 -> 0.1016, 0.3 -> 0.3008), keep scale uses the quantised rate
static inline uint32_t drop8(float p) { long t = lroundf(p * 256.f); return (uint32_t)(t < 0 ? 0 : (t > 255 ? 255 : t)); }

static int attn_check(const char* who, int64_t b, int64_t h, int64_t lq, int64_t lk, int64_t d, int dtype) {
  GMLM_REQUIRE(b >= 0 && h > 0 && lq >= 0 && lk >= 0, "%s: bad sizes", who);
  GMLM_REQUIRE(d == 64 || d == 96, "%s: head dim %ld not supported (64 or 96)", who, (long)d);
  GMLM_REQUIRE(dtype == GMLM_F32 || dtype == GMLM_BF16, "%s: unsupported dtype", who);
  GMLM_REQUIRE(b * h <= 65535, "%s: batch*heads %ld > 65535", who, (long)(b * h));   // grid.y of the two-dimensional launches
  return GMLM_OK;
}
static int stride_check(const char* who, const void* ptr, int64_t stride, int64_t min_stride, int dtype) {
  const int v = dtype == GMLM_F32 ? 4 : 8;
  GMLM_REQUIRE(ptr && aligned16(ptr), "%s: null or not 16-byte aligned pointer", who);
  GMLM_REQUIRE(stride >= min_stride && stride % v == 0 && stride < (1 << 24),
               "%s: row stride %ld must be >= %ld, < 2^24 and a multiple of %d", who, (long)stride, (long)min_stride, v);
  return GMLM_OK;
}

}  // namespace gmlm

using namespace gmlm;

#define GMLM_ATTN_DISPATCH2(KERNEL, NWV, DR, grid, st, prm)                                  \
  do {                                                                                       \
    if (dtype == GMLM_BF16) {                                                                \
      if (d == 64) KERNEL<bf16_t, 64, NWV, DR><<<grid, NWV * 64, 0, st>>>(prm);              \
      else KERNEL<bf16_t, 96, NWV, DR><<<grid, NWV * 64, 0, st>>>(prm);                      \
    } else {                                                                                 \
      if (d == 64) KERNEL<float, 64, NWV, DR><<<grid, NWV * 64, 0, st>>>(prm);               \
      else KERNEL<float, 96, NWV, DR><<<grid, NWV * 64, 0, st>>>(prm);                       \
    }                                                                                        \
  } while (0)
#define GMLM_ATTN_DISPATCH(KERNEL, NWV, grid, st, prm)                                       \
  do {                                                                                       \
    if ((prm).drop_thresh) GMLM_ATTN_DISPATCH2(KERNEL, NWV, true, grid, st, prm);            \
    else GMLM_ATTN_DISPATCH2(KERNEL, NWV, false, grid, st, prm);                             \
  } while (0)

// Rows per workgroup.  Every workgroup streams ALL keys of its (batch, head) through LDS, so the K/V bytes read
// per flop fall with the number of query rows that share a tile: 8 waves = 256 rows for long sequences when there
// are enough workgroups to fill the chip twice (measured: forward neutral, backward +4 % at N = 20k); 4 waves
// (128 rows) by default; 2 waves (64 rows) when the grid would under-fill the 256 CUs (+80 % at N = 5k).
namespace gmlm {
int attn_fwd_pipe_launch(const AttnParams& p, int d, int nw, int64_t rows_q, int64_t bh, hipStream_t st);
int attn_short_fwd_launch(const AttnParams& p, int64_t rows, int64_t items, hipStream_t st);
int attn_short_bwd_launch(const AttnParams& p, int64_t rows, int64_t items, float* dbias, hipStream_t st);
int attn_bwd_pipe_launch(const AttnParams& p, int d, int nw, int64_t rows_q, int64_t rows_k, int64_t bh, hipStream_t st);
}

static inline int pick_waves(int64_t rows, int64_t bh) {
  if (rows >= 2048 && cdiv(rows, 256) * bh >= 512) return 8;
  return cdiv(rows, 128) * bh >= 1024 ? 4 : 2;
}

#ifdef GMLM_ATTN_STAMP
static void* g_stamp_buffer = nullptr;     // diagnostic build only: per-wave cycle stamps of the pipelined forward
extern "C" void gmlm_debug_set_stamp_buffer(void* p) { g_stamp_buffer = p; }
#endif

extern "C" int gmlm_attention_fwd(const void* q, const void* k, const void* v, const int32_t* kv_len, int64_t b, int64_t h,
                                  int64_t lq, int64_t lk, int64_t d, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                  float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* out, void* out_lo, float* lse,
                                  int dtype, const int32_t* cu_seqlens, int64_t max_len, const int32_t* seq_groups, int64_t num_groups,
                                  gmlm_stream_t stream) {
  int rc = attn_check("attention_fwd", b, h, lq, lk, d, dtype);
  GMLM_REQUIRE(!seq_groups || (cu_seqlens && num_groups > 0 && num_groups <= b), "attention_fwd: seq_groups needs packed mode and 0 < num_groups <= b");
  GMLM_REQUIRE(!out_lo || (dtype == GMLM_BF16 && aligned16(out_lo)), "attention_fwd: out_lo is the bf16 residual of a bf16 output (16-byte aligned)");
  GMLM_REQUIRE(!cu_seqlens || (lq == lk && max_len > 0 && !kv_len), "attention_fwd: packed mode needs lq == lk = total rows, max_len > 0, kv_len NULL");
  GMLM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "attention_fwd: dropout_p must be in [0,1)");
  GMLM_REQUIRE(scale > 0.f, "attention_fwd: scale must be positive");
  if (rc != GMLM_OK) return rc;
  if (b == 0 || lq == 0) return GMLM_OK;
  if ((rc = stride_check("attention_fwd(q)", q, q_stride, h * d, dtype)) != GMLM_OK) return rc;
  if (lk > 0) {
    if ((rc = stride_check("attention_fwd(k)", k, k_stride, h * d, dtype)) != GMLM_OK) return rc;
    if ((rc = stride_check("attention_fwd(v)", v, v_stride, h * d, dtype)) != GMLM_OK) return rc;
  }
  GMLM_REQUIRE(out && aligned16(out), "attention_fwd: null or misaligned out");
  AttnParams p{};
  p.q = q; p.k = k; p.v = v; p.kv_len = kv_len; p.o_w = out; p.o_lo_w = out_lo; p.lse_w = lse; p.cu = cu_seqlens; p.groups = seq_groups;
  const int64_t rows_q = cu_seqlens ? max_len : lq;
  p.b = b; p.h = h; p.lq = lq; p.lk = lk; p.q_stride = q_stride; p.k_stride = k_stride; p.v_stride = v_stride;
  p.scale = scale;
  p.drop_thresh = drop8(dropout_p); p.keep_scale = 256.f / (256.f - (float)p.drop_thresh); p.seed = seed; p.seed_dev = seed_dev;
  hipStream_t st = as_stream(stream);
#ifdef GMLM_ATTN_STAMP
  p.delta = static_cast<float*>(g_stamp_buffer);
#endif
  if (dtype == GMLM_BF16 && d == 64 && rows_q <= 128 && (cu_seqlens ? max_len : lk) <= 128 && b * h >= 512) {
    // short sequences, enough of them to fill the chip: Q / K / V resident in LDS, one barrier (attn_short.hip); with
    // `seq_groups` a workgroup takes a run of sequences (<= 128 rows) instead of one
    const int64_t rk_ = cu_seqlens ? max_len : lk, rmax = rows_q > rk_ ? rows_q : rk_;
    return attn_short_fwd_launch(p, seq_groups ? 128 : rmax, seq_groups ? num_groups : b, st);
  }
  if (dtype == GMLM_BF16 && (d == 96 || rows_q > 128)) {
    // software-pipelined LDS-DMA kernel (attn_fwd_pipe.hip): CrossAttention geometry (d = 96: +40-45 % at N = 5k-20k
    // against attn_fwd_kernel) and BERT geometry beyond 128 tokens (d = 64: +8 % at L = 512, +15 % at L = 2048).
    // Sequences of <= 128 tokens (what the reference's tokeniser produces, main.py:340) go to attn_fwd_short_kernel above
    // when there are enough of them; this kernel measures the same there (171 vs 175 us on the bench's packed mix).
    const int nw = d == 96 ? (pick_waves(rows_q, b * h) == 8 ? 8 : 4) : 4;
    rc = attn_fwd_pipe_launch(p, (int)d, nw, rows_q, b * h, st);
    if (rc != GMLM_OK) return rc;
    GMLM_LAUNCH_CHECK();
    return GMLM_OK;
  }
  if (pick_waves(rows_q, b * h) == 8) {
    dim3 grid((unsigned)cdiv(rows_q, 256), (unsigned)(b * h));
    GMLM_ATTN_DISPATCH(attn_fwd_kernel, 8, grid, st, p);
  } else if (pick_waves(rows_q, b * h) == 4) {
    dim3 grid((unsigned)cdiv(rows_q, 128), (unsigned)(b * h));
    GMLM_ATTN_DISPATCH(attn_fwd_kernel, 4, grid, st, p);
  } else {
    dim3 grid((unsigned)cdiv(rows_q, 64), (unsigned)(b * h));
    GMLM_ATTN_DISPATCH(attn_fwd_kernel, 2, grid, st, p);
  }
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

extern "C" size_t gmlm_attention_bwd_workspace_bytes(int64_t b, int64_t h, int64_t lq, int64_t lk, int64_t d) {
  (void)lk; (void)d;
  return (size_t)(b * h * lq > 0 ? b * h * lq : 1) * sizeof(float);  // delta
}

extern "C" int gmlm_attention_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                                  const float* lse, const int32_t* kv_len, int64_t b, int64_t h, int64_t lq, int64_t lk,
                                  int64_t d, int64_t q_stride, int64_t k_stride, int64_t v_stride, float scale,
                                  float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* dq, void* dk, void* dv, int64_t dq_stride,
                                  int64_t dk_stride, int64_t dv_stride, int dtype, const int32_t* cu_seqlens, int64_t max_len,
                                  void* workspace, size_t workspace_bytes, float* dbias_partial, float* dbias,
                                  const void* out_lo, const int32_t* seq_groups, int64_t num_groups, gmlm_stream_t stream) {
  int rc = attn_check("attention_bwd", b, h, lq, lk, d, dtype);
  GMLM_REQUIRE(!seq_groups || (cu_seqlens && num_groups > 0 && num_groups <= b), "attention_bwd: seq_groups needs packed mode and 0 < num_groups <= b");
  GMLM_REQUIRE(!out_lo || (dtype == GMLM_BF16 && aligned16(out_lo)), "attention_bwd: out_lo is the bf16 residual of a bf16 output (16-byte aligned)");
  GMLM_REQUIRE((dbias_partial == nullptr) == (dbias == nullptr), "attention_bwd: dbias_partial and dbias come together");
  GMLM_REQUIRE(!cu_seqlens || (lq == lk && max_len > 0 && !kv_len), "attention_bwd: packed mode needs lq == lk = total rows, max_len > 0, kv_len NULL");
  if (rc != GMLM_OK) return rc;
  if (b == 0 || (lq == 0 && lk == 0)) return GMLM_OK;
  GMLM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "attention_bwd: dropout_p must be in [0,1)");
  GMLM_REQUIRE(lq > 0 && lk > 0, "attention_bwd: empty query or key set with a non-empty counterpart is not supported");
  if ((rc = stride_check("attention_bwd(q)", q, q_stride, h * d, dtype)) != GMLM_OK) return rc;
  if ((rc = stride_check("attention_bwd(k)", k, k_stride, h * d, dtype)) != GMLM_OK) return rc;
  if ((rc = stride_check("attention_bwd(v)", v, v_stride, h * d, dtype)) != GMLM_OK) return rc;
  if ((rc = stride_check("attention_bwd(dq)", dq, dq_stride, h * d, dtype)) != GMLM_OK) return rc;
  if ((rc = stride_check("attention_bwd(dk)", dk, dk_stride, h * d, dtype)) != GMLM_OK) return rc;
  if ((rc = stride_check("attention_bwd(dv)", dv, dv_stride, h * d, dtype)) != GMLM_OK) return rc;
  GMLM_REQUIRE(out && dout && lse && aligned16(out) && aligned16(dout), "attention_bwd: null or misaligned out/dout/lse");
  GMLM_REQUIRE(workspace && workspace_bytes >= gmlm_attention_bwd_workspace_bytes(cu_seqlens ? 1 : b, h, lq, lk, d),
               "attention_bwd: workspace too small");
  hipStream_t st = as_stream(stream);
  AttnParams p{};
  p.q = q; p.k = k; p.v = v; p.out = out; p.out_lo = out_lo; p.dout = dout; p.lse = lse; p.kv_len = kv_len; p.cu = cu_seqlens; p.groups = seq_groups;
  const int64_t rows_q = cu_seqlens ? max_len : lq, rows_k = cu_seqlens ? max_len : lk;
  const int64_t nb = cu_seqlens ? 1 : b;        // delta kernel: packed tensors are one [total_rows, h, d] block
  p.delta = static_cast<float*>(workspace);
  p.dq = dq; p.dk = dk; p.dv = dv; p.dbias_part = dbias_partial;
  p.b = b; p.h = h; p.lq = lq; p.lk = lk; p.q_stride = q_stride; p.k_stride = k_stride; p.v_stride = v_stride;
  p.dq_stride = dq_stride; p.dk_stride = dk_stride; p.dv_stride = dv_stride; p.scale = scale;
  p.drop_thresh = drop8(dropout_p); p.keep_scale = 256.f / (256.f - (float)p.drop_thresh); p.seed = seed; p.seed_dev = seed_dev;
  if (dtype == GMLM_BF16 && d == 64 && rows_q <= 128 && rows_k <= 128 && b * h >= 512) {
    // short sequences, enough of them to fill the chip: one fused launch (delta + dQ + dK/dV), no workspace, O is not read
    const int64_t rmax = rows_q > rows_k ? rows_q : rows_k;
    return attn_short_bwd_launch(p, seq_groups ? 128 : rmax, seq_groups ? num_groups : b, dbias, st);
  }
  GMLM_REQUIRE(!dbias, "attention_bwd: the fused bias-gradient sums exist only on the short-sequence path (bf16, d = 64, <= 128 rows, b*h >= 512)");
  const int64_t rows = nb * lq * h;
  {
    const int cpr = (int)d / (dtype == GMLM_BF16 ? 8 : 4);
    const int lanes = cpr <= 8 ? 8 : (cpr <= 16 ? 16 : 32);
    const int64_t threads = rows * lanes;
    if (dtype == GMLM_BF16)
      attn_delta_kernel<bf16_t><<<(unsigned)cdiv(threads, 256), 256, 0, st>>>((const bf16_t*)out, (const bf16_t*)out_lo, (const bf16_t*)dout, rows, (int)d, lq, h, p.delta);
    else
      attn_delta_kernel<float><<<(unsigned)cdiv(threads, 256), 256, 0, st>>>((const float*)out, nullptr, (const float*)dout, rows, (int)d, lq, h, p.delta);
  }
  GMLM_LAUNCH_CHECK();
  if (dtype == GMLM_BF16 && d == 96) {
    // CrossAttention geometry: LDS-DMA staged kernels (attn_bwd_pipe.hip): no staging registers (the register-staged 2- / 4-wave
    // variants spill 44-88 bytes per lane at d = 96), 4-wave workgroups unless the grid fills the chip twice with 8.
    // bf16 below this point is therefore d = 64
    return attn_bwd_pipe_launch(p, (int)d, pick_waves(rows_q, b * h) == 8 ? 8 : 4, rows_q, rows_k, b * h, st);
  }
  if (pick_waves(rows_q, b * h) == 8) {
    dim3 gq((unsigned)cdiv(rows_q, 256), (unsigned)(b * h));
    GMLM_ATTN_DISPATCH(attn_bwd_dq_kernel, 8, gq, st, p);
  } else if (pick_waves(rows_q, b * h) == 4) {
    dim3 gq((unsigned)cdiv(rows_q, 128), (unsigned)(b * h));
    GMLM_ATTN_DISPATCH(attn_bwd_dq_kernel, 4, gq, st, p);
  } else {
    dim3 gq((unsigned)cdiv(rows_q, 64), (unsigned)(b * h));
    GMLM_ATTN_DISPATCH(attn_bwd_dq_kernel, 2, gq, st, p);
  }
  GMLM_LAUNCH_CHECK();
  if (pick_waves(rows_k, b * h) == 8 && !(dtype == GMLM_F32 && d == 96)) {   // f32 d=96 at 512 threads would spill
    dim3 gk((unsigned)cdiv(rows_k, 256), (unsigned)(b * h));
    if (dtype == GMLM_BF16) {   // long sequences: 64 query rows per barrier (two 32-row blocks), double-buffered
      { if (p.drop_thresh) attn_bwd_dkv_kernel<bf16_t, 64, 8, true, 64><<<gk, 512, 0, st>>>(p); else attn_bwd_dkv_kernel<bf16_t, 64, 8, false, 64><<<gk, 512, 0, st>>>(p); }
    } else {
      GMLM_ATTN_DISPATCH(attn_bwd_dkv_kernel, 8, gk, st, p);
    }
  } else if (pick_waves(rows_k, b * h) >= 4) {
    dim3 gk((unsigned)cdiv(rows_k, 128), (unsigned)(b * h));
    if (dtype == GMLM_BF16 && rows_q <= 128) {
      // short sequences (the reference tokenises to <= 128 tokens): all query rows staged behind one barrier
      { if (p.drop_thresh) attn_bwd_dkv_kernel<bf16_t, 64, 4, true, 128><<<gk, 256, 0, st>>>(p); else attn_bwd_dkv_kernel<bf16_t, 64, 4, false, 128><<<gk, 256, 0, st>>>(p); }
    } else if (dtype == GMLM_BF16) {
      // 64 query rows per barrier (two 32-row blocks), double-buffered: -8 % at L = 512 against 32-row tiles
      { if (p.drop_thresh) attn_bwd_dkv_kernel<bf16_t, 64, 4, true, 64><<<gk, 256, 0, st>>>(p); else attn_bwd_dkv_kernel<bf16_t, 64, 4, false, 64><<<gk, 256, 0, st>>>(p); }
    } else {
      GMLM_ATTN_DISPATCH(attn_bwd_dkv_kernel, 4, gk, st, p);
    }
  } else {
    dim3 gk((unsigned)cdiv(rows_k, 64), (unsigned)(b * h));
    if (dtype == GMLM_BF16 && rows_q > 128) {      // 64-row tiles here too: -11 % at N = 5,201
      { if (p.drop_thresh) attn_bwd_dkv_kernel<bf16_t, 64, 2, true, 64><<<gk, 128, 0, st>>>(p); else attn_bwd_dkv_kernel<bf16_t, 64, 2, false, 64><<<gk, 128, 0, st>>>(p); }
    } else {
      GMLM_ATTN_DISPATCH(attn_bwd_dkv_kernel, 2, gk, st, p);
    }
  }
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}
