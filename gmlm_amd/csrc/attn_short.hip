// K5 for sequences of <= 128 tokens (bf16, d = 64): the text encoder's attention, forward and backward, gfx950.
// Reference: BertSelfAttention's softmax(QK^T d^-1/2 + padding mask)V (hf:modeling_bert.py:188-201) on what the reference's
// tokeniser produces (max_length = 128, main.py:340), and its autograd backward.
//
// One workgroup = one (GROUP, head) item.  A group is a run of consecutive sequences of the packed batch whose rows add up
// to <= R (= 128) - at most 13 of them - so a workgroup always stages full tiles: its Q / K / V (/ dO) rows are ONE contiguous
// row range of the packed tensors, fetched with the coalesced pattern (8 lanes x 16 B per 128-byte row piece), all loads in
// flight at once.  (One workgroup per sequence left a third of every 128-row tile empty at the 16..128-token mix and kept too
// few bytes in flight per CU: 3.0-3.9 TB/s.)  Attention stays inside a sequence through a block-diagonal mask that costs no
// VALU work: every score chain starts with ONE extra contraction step
//        onehot(sequence id of the tile row) x [ -2^100 where column != sequence id of the lane's row ]
// so a score between rows of different sequences (or against a padding row: id 15, which nobody matches) leaves the MFMA chain
// at about -1.3e30 and its probability is exactly 0.  32-row blocks that cannot hold a row of the wave's sequences are skipped
// (wave-uniform block range).  Padded batches (kv_len) run as one sequence per group through the same code.
//
// Forward: all keys of a row are resident, so the softmax is two plain passes over the score blocks held in registers.
// Backward: ONE launch.  Phase 1 (query on the lane) computes P and dP = dO V^T of all its key blocks, keeps them in
// registers, forms  delta = sum_k P dP  from them IN FP32 and only then dS = P (dP - delta) and dQ.  That delta is exact for
// the P the kernel uses: the rows of dS sum to zero to fp32 precision.  The usual flash-attention form delta = rowsum(dO * O)
// reads the bf16-ROUNDED forward output: its error (2^-9 |dO||O|) does not cancel and was the whole reason why the query / key
// projection gradients of the deep encoder layers had a cosine of 0.4 against fp32 (measured: oracle/bf16_emulation.py,
// DESIGN.md section 4); it also needed a fifth tensor read (O).  Phase 2 (key on the lane) takes delta from LDS and forms
// dK, dV.  Optionally the column sums of dQ | dK | dV (the fused QKV projection's bias gradient) come out as three
// matrix-vector products on the MFMA pipe; their per-key coefficients c_k = sum_q dS[q,k] sum to zero over a sequence, so
// they are carried as a bf16 hi + lo pair (2^-17) instead of one rounded bf16 value.
#include "attn_common.hpp"
#include "colreduce.hpp"

namespace gmlm {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#ifndef GMLM_SHORT_FWD_OCC
#define GMLM_SHORT_FWD_OCC 4
#endif
constexpr int kShortMaxSeq = 13;   // sequences per group: ids 0..12; 14 = padding query row, 15 = padding key row

// A fragment of the mask step for a tile row with sequence id `id`: 1 in column id (elements j = 0..7 <-> columns 8h + j)
__device__ __forceinline__ bf16x8 onehot_frag(int id, int h) {
  const int e = id - 8 * h;
  u32x4 w;
#pragma unroll
  for (int t = 0; t < 4; ++t) w[t] = (e >> 1) == t ? (0x3F80u << (16 * (e & 1))) : 0u;          // bf16 1.0 = 0x3F80
  return __builtin_bit_cast(bf16x8, w);
}
// B fragment of the mask step for the lane's own row: -2^100 (bf16 0xF180) in every column except its sequence id
__device__ __forceinline__ bf16x8 antihot_frag(int id, int h) {
  const int e = id - 8 * h;
  u32x4 w;
#pragma unroll
  for (int t = 0; t < 4; ++t) w[t] = (e >> 1) == t ? (0xF180F180u & ~(0xFFFFu << (16 * (e & 1)))) : 0xF180F180u;
  return __builtin_bit_cast(bf16x8, w);
}

// row bookkeeping of one item, in LDS: sequence id of every Q-side / K-side tile row and the row range a row may attend to
struct ShortMeta {
  uint8_t* qid;      // [R]
  uint8_t* kid;      // [R]
  int16_t* q2k_lo;   // [R] first key row (tile-relative) of the query row's sequence
  int16_t* q2k_hi;   // [R] one past its last key row
  int16_t* k2q_lo;   // [R] same for a key row -> query rows
  int16_t* k2q_hi;   // [R]
};

// item -> row ranges; fills the meta arrays (threads 0..R-1).  nq / nk = rows of the Q-side / K-side tiles that exist.
template <int R>
__device__ __forceinline__ void short_item_setup(const AttnParams& p, int64_t item, int tid, const ShortMeta& mt, int64_t& qbase,
                                                 int64_t& kbase, int& nq, int& nk, int64_t& lse_base, int64_t hd) {
  if (p.cu) {
    const int s0 = p.groups ? p.groups[item] : (int)item, s1 = p.groups ? p.groups[item + 1] : (int)item + 1;
    const int row0 = p.cu[s0];
    int n = p.cu[s1] - row0;
    n = n < 0 ? 0 : (n > R ? R : n);                          // the host builds groups of <= R rows; never index past the tiles
    qbase = kbase = row0;
    nq = nk = n;
    lse_base = hd * p.lq + row0;                              // lse laid out [h, total_rows]
    if (tid < R) {
      int id = 14, kidv = 15, lo = 0, hi = 0;
      if (tid < n) {
        const int grow = row0 + tid;
        id = 0;
        for (int j = s0 + 1; j < s1; ++j) id += grow >= p.cu[j] ? 1 : 0;
        id = id < kShortMaxSeq ? id : kShortMaxSeq - 1;
        kidv = id;
        lo = p.cu[s0 + id] - row0;
        hi = p.cu[s0 + id + 1] - row0;
        hi = hi > n ? n : hi;
      }
      mt.qid[tid] = (uint8_t)id; mt.kid[tid] = (uint8_t)kidv;
      mt.q2k_lo[tid] = mt.k2q_lo[tid] = (int16_t)lo;
      mt.q2k_hi[tid] = mt.k2q_hi[tid] = (int16_t)hi;
    }
  } else {
    const int64_t b = item;
    qbase = b * p.lq; kbase = b * p.lk;
    nq = (int)p.lq; nk = (int)p.lk;
    int kvl = nk;
    if (p.kv_len) { kvl = p.kv_len[b]; kvl = kvl > nk ? nk : (kvl < 0 ? 0 : kvl); }
    lse_base = (b * p.h + hd) * p.lq;
    if (tid < R) {
      mt.qid[tid] = tid < nq ? 0 : 14;
      mt.kid[tid] = tid < kvl ? 0 : 15;
      mt.q2k_lo[tid] = 0; mt.q2k_hi[tid] = (int16_t)kvl;
      mt.k2q_lo[tid] = 0; mt.k2q_hi[tid] = (int16_t)((tid & ~31) < kvl ? nq : 0);      // per 32-key block: any valid key -> all query rows
    }
  }
}

// wave-uniform range of 32-row blocks on the OTHER side that rows [32w, 32w + 32) of this side (n rows exist) can touch
__device__ __forceinline__ void block_range(const int16_t* lo, const int16_t* hi, int w, int n, int& b0, int& b1) {
  const int first = 32 * w, last = (32 * w + 31 < n ? 32 * w + 31 : n - 1);
  int l = lo[first], hgh = hi[last < first ? first : last];
  l = __builtin_amdgcn_readfirstlane(l);
  hgh = __builtin_amdgcn_readfirstlane(hgh);
  b0 = l >> 5;
  b1 = (hgh + 31) >> 5;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int D, bool DROP, int R>
__global__ __launch_bounds__(2 * R, GMLM_SHORT_FWD_OCC) void attn_fwd_short_kernel(AttnParams p) {
  using T = bf16_t;
  static_assert(D == 64, "dense swizzled images are laid out for d = 64");
  constexpr int NT = 2 * R, PITCH = D, DB = D / 32, CPR = D / 8, PER = R * CPR / NT, NB = R / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tk = reinterpret_cast<T*>(smem_raw);
  T* tv = tk + R * PITCH;
  ShortMeta mt;
  mt.q2k_lo = reinterpret_cast<int16_t*>(tv + R * PITCH);
  mt.q2k_hi = mt.q2k_lo + R; mt.k2q_lo = mt.q2k_hi + R; mt.k2q_hi = mt.k2q_lo + R;
  mt.qid = reinterpret_cast<uint8_t*>(mt.k2q_hi + R);
  mt.kid = mt.qid + R;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int64_t item = blockIdx.x / p.h, hd = blockIdx.x % p.h;
  int64_t qbase, kbase, lse_base;
  int nq, nk;
  short_item_setup<R>(p, item, tid, mt, qbase, kbase, nq, nk, lse_base, hd);
  const T* qg = static_cast<const T*>(p.q) + qbase * p.q_stride + hd * D;
  const T* kg = static_cast<const T*>(p.k) + kbase * p.k_stride + hd * D;
  const T* vg = static_cast<const T*>(p.v) + kbase * p.v_stride + hd * D;
  // Only K and V are staged (they are MFMA A operands of every wave): 32 KB at R = 128 = five workgroups per CU, and with them
  // five items' worth of loads in flight per CU.  A query row is used by its own lane pair alone: it goes from global memory
  // straight into the B-operand registers, issued together with the K / V loads.
  const int q_row = w * 32 + r;
  RowFrag<T, D> qf;
  qf.load(qg + (uint32_t)((q_row < nq ? q_row : 0) * (int)p.q_stride), q_row < nq, h);
  uint4 rk[PER], rv[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + k * NT, row = i / CPR, col = (i % CPR) * 8;
    rk[k] = rv[k] = make_uint4(0, 0, 0, 0);
    if (row < nk) {
      rk[k] = *reinterpret_cast<const uint4*>(kg + (uint32_t)(row * (int)p.k_stride + col));
      rv[k] = *reinterpret_cast<const uint4*>(vg + (uint32_t)(row * (int)p.v_stride + col));
    }
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + k * NT, row = i / CPR;
    const int c = i % CPR;                                      // 16-byte chunk of the row -> swizzled slot (attn_common.hpp)
    *reinterpret_cast<uint4*>(tk + row * PITCH + ((c ^ sw_row(row)) << 3)) = rk[k];
    *reinterpret_cast<uint4*>(tv + row * PITCH + ((c ^ sw_tr(row)) << 3)) = rv[k];
  }
  __syncthreads();
  if (w * 32 >= nq) return;                                   // (no barrier below)
  const bf16x8 qm = antihot_frag(mt.qid[q_row], h);
  int b0, b1;
  block_range(mt.q2k_lo, mt.q2k_hi, w, nq, b0, b1);           // key blocks that hold a key of this wave's sequences
  const float sl2 = p.scale * kLog2e;
  f32x16 s[NB];
  float rmax = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    if (kb < b0 || kb >= b1) continue;
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(onehot_frag(mt.kid[kb * 32 + r], h), qm, z, 0, 0, 0);   // block-diagonal / padding mask
    mma_rows_sw(tk, kb * 32, qf, s[kb], r, h);
    rmax = fmaxf(rmax, max16(s[kb]));
  }
  const float m = b1 > b0 ? xhalf_max(rmax) * sl2 : 0.f;      // finite for a real query row (its own sequence has >= 1 key)
  const uint32_t dq_u = drop_base(attn_seed(p), lse_base) + (uint32_t)(q_row >> 1) * kDropC1 + (uint32_t)(2 * h) * kDropC2;
  const int q_odd = q_row & 1;
  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  float l = 0.f;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    if (kb < b0 || kb >= b1) continue;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const float e0 = fast_exp2(fmaf(s[kb][i], sl2, -m)), e1 = fast_exp2(fmaf(s[kb][i + 1], sl2, -m));   // masked: exp2(-2e29) = 0
      l += e0 + e1;                                           // the normaliser uses the un-dropped probabilities
      float m0 = 1.f, m1 = 1.f;
      if (DROP) drop_pair_q(dq_u + (uint32_t)((kb * 32 + acc_row(i, 0)) >> 1) * kDropC2, q_odd, p.drop_thresh, p.keep_scale, m0, m1);
      s[kb][i] = e0 * m0;
      s[kb][i + 1] = e1 * m1;
    }
    mma_acc_sw(tv, kb * 32, s[kb], o, lane);
  }
  l = xhalf_sum(l);
  if (q_row < nq) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    T* og = static_cast<T*>(p.o_w) + ((qbase + q_row) * p.h + hd) * D;
#pragma unroll
    for (int d = 0; d < DB; ++d) store_t<T>(og + d * 32, o[d], inv, h);
    if (h == 0 && p.lse_w) p.lse_w[lse_base + q_row] = l > 0.f ? (m + __log2f(l)) * kLn2 : -INFINITY;
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
template <int D, bool DROP, int R>
__global__ __launch_bounds__(2 * R, R >= 96 ? 2 : 3) void attn_bwd_short_kernel(AttnParams p) {
  using T = bf16_t;
  constexpr int NT = 2 * R, PITCH = D + Pad<T>::v, DB = D / 32, CPR = D / 8, PER = R * CPR / NT, NB = R / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tk = reinterpret_cast<T*>(smem_raw);
  T* tv = tk + R * PITCH;
  T* tq = tv + R * PITCH;
  T* tdo = tq + R * PITCH;
  float* lse_s = reinterpret_cast<float*>(tdo + R * PITCH);
  float* dl_s = lse_s + R;
  // bias-gradient coefficients (p.dbias_part), bf16: per key c_k = sum_q dS[q,k] as hi + lo, per query pm_q = sum_k P'[q,k]: [3][R]
  T* coef = reinterpret_cast<T*>(dl_s + R);
  ShortMeta mt;
  mt.q2k_lo = reinterpret_cast<int16_t*>(coef + 3 * R);
  mt.q2k_hi = mt.q2k_lo + R; mt.k2q_lo = mt.q2k_hi + R; mt.k2q_hi = mt.k2q_lo + R;
  mt.qid = reinterpret_cast<uint8_t*>(mt.k2q_hi + R);
  mt.kid = mt.qid + R;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int64_t item = blockIdx.x / p.h, hd = blockIdx.x % p.h;
  const bool bgrad = p.dbias_part != nullptr;                // block-uniform
  if (bgrad)
    for (int i = tid; i < 3 * R; i += NT) coef[i] = f32_to_bf16(0.f);     // waves that skip a phase leave zeros
  int64_t qbase, kbase, lse_base;
  int nq, nk;
  short_item_setup<R>(p, item, tid, mt, qbase, kbase, nq, nk, lse_base, hd);
  const T* qg = static_cast<const T*>(p.q) + qbase * p.q_stride + hd * D;
  const T* kg = static_cast<const T*>(p.k) + kbase * p.k_stride + hd * D;
  const T* vg = static_cast<const T*>(p.v) + kbase * p.v_stride + hd * D;
  const T* dog = static_cast<const T*>(p.dout) + (qbase * p.h + hd) * D;
  const int do_stride = (int)(p.h * D);
  // ---- staging: all loads first (32-bit offsets: rows <= 128, strides < 2^24), then the LDS image ----
  uint4 rk[PER], rv[PER], rq[PER], rdo[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + k * NT, row = i / CPR, col = (i % CPR) * 8;
    rk[k] = rv[k] = rq[k] = rdo[k] = make_uint4(0, 0, 0, 0);
    if (row < nk) {
      rk[k] = *reinterpret_cast<const uint4*>(kg + (uint32_t)(row * (int)p.k_stride + col));
      rv[k] = *reinterpret_cast<const uint4*>(vg + (uint32_t)(row * (int)p.v_stride + col));
    }
    if (row < nq) {
      rq[k] = *reinterpret_cast<const uint4*>(qg + (uint32_t)(row * (int)p.q_stride + col));
      rdo[k] = *reinterpret_cast<const uint4*>(dog + (uint32_t)(row * do_stride + col));
    }
  }
  if (tid < R) lse_s[tid] = tid < nq ? p.lse[lse_base + tid] * kLog2e : INFINITY;   // +inf: a row past the item gets probability 0
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + k * NT, row = i / CPR, col = (i % CPR) * 8;
    *reinterpret_cast<uint4*>(tk + row * PITCH + col) = rk[k];
    *reinterpret_cast<uint4*>(tv + row * PITCH + col) = rv[k];
    *reinterpret_cast<uint4*>(tq + row * PITCH + col) = rq[k];
    *reinterpret_cast<uint4*>(tdo + row * PITCH + col) = rdo[k];
  }
  __syncthreads();
  const float sl2 = p.scale * kLog2e;
  const uint32_t d_base = drop_base(attn_seed(p), lse_base);
  auto frag = [&](const T* tile, int row, RowFrag<T, D>& f) {          // MFMA B-operand fragment of a staged row
#pragma unroll
    for (int s = 0; s < D / 16; ++s) f.v[s] = *reinterpret_cast<const bf16x8*>(tile + row * PITCH + 16 * s + 8 * h);
  };
  // ---- phase 1: delta and dQ (query on the lane; K, V as MFMA A operands) -------------------------------
  if (w * 32 < nq) {
    const int q_row = w * 32 + r;
    RowFrag<T, D> qf, dof;
    frag(tq, q_row, qf);
    frag(tdo, q_row, dof);
    const float lse2 = lse_s[q_row];
    const bf16x8 qm = antihot_frag(mt.qid[q_row], h);
    int b0, b1;
    block_range(mt.q2k_lo, mt.q2k_hi, w, nq, b0, b1);
    const uint32_t dq_u = d_base + (uint32_t)(q_row >> 1) * kDropC1 + (uint32_t)(2 * h) * kDropC2;
    const int q_odd = q_row & 1;
    // pass A: P and dP' (= dP through the dropout mask) of every key block, kept in registers; delta = sum_k P dP' in fp32
    f32x16 pr[NB], dpm[NB];
    float dsum = 0.f, pm = 0.f;                             // this half-wave's shares of delta and of sum_k P'[q,k]
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      if (kb < b0 || kb >= b1) continue;                    // wave-uniform
      f32x16 z;
#pragma unroll
      for (int i = 0; i < 16; ++i) { z[i] = 0.f; dpm[kb][i] = 0.f; }
      pr[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(onehot_frag(mt.kid[kb * 32 + r], h), qm, z, 0, 0, 0);
      mma_rows<D>(tk, PITCH, kb * 32, qf, pr[kb], r, h);
      mma_rows<D>(tv, PITCH, kb * 32, dof, dpm[kb], r, h);
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const float p0 = fast_exp2(fmaf(pr[kb][i], sl2, -lse2)), p1 = fast_exp2(fmaf(pr[kb][i + 1], sl2, -lse2));   // masked: exactly 0
        float m0 = 1.f, m1 = 1.f;
        if (DROP) drop_pair_q(dq_u + (uint32_t)((kb * 32 + acc_row(i, 0)) >> 1) * kDropC2, q_odd, p.drop_thresh, p.keep_scale, m0, m1);
        const float d0 = dpm[kb][i] * m0, d1 = dpm[kb][i + 1] * m1;
        dsum = fmaf(p1, d1, fmaf(p0, d0, dsum));
        if (bgrad) pm = fmaf(p1, m1, fmaf(p0, m0, pm));
        pr[kb][i] = p0; pr[kb][i + 1] = p1;
        dpm[kb][i] = d0; dpm[kb][i + 1] = d1;
      }
    }
    const float dl = xhalf_sum(dsum);
    if (h == 0) dl_s[q_row] = dl;
    // pass B: dS = P (dP' - delta); dQ^T += K^T dS^T
    f32x16 dq[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      if (kb < b0 || kb >= b1) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) pr[kb][i] *= dpm[kb][i] - dl;
      mma_acc<D>(tk, PITCH, kb * 32, pr[kb], dq, lane);
    }
    if (q_row < nq) {
      T* dqg = static_cast<T*>(p.dq) + (qbase + q_row) * p.dq_stride + hd * D;
#pragma unroll
      for (int d = 0; d < DB; ++d) store_t<T>(dqg + d * 32, dq[d], p.scale, h);
    }
    if (bgrad) {
      pm = xhalf_sum(pm);
      if (h == 0) coef[2 * R + q_row] = f32_to_bf16(pm);
    }
  }
  // rows past the item (their waves skipped phase 1): delta 0, never read with a non-zero probability
  if (w * 32 >= nq && lane < 32) dl_s[w * 32 + lane] = 0.f;
  __syncthreads();                                          // delta of every query row is in LDS
  // ---- phase 2: dK, dV (key on the lane; Q, dO as MFMA A operands) -------------------------------------
  if (w * 32 < nk) {
    const int key = w * 32 + r;
    int b0, b1;
    block_range(mt.k2q_lo, mt.k2q_hi, w, nk, b0, b1);       // empty when the wave holds padding keys only
    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dk[d][i] = 0.f; dv[d][i] = 0.f; }
    if (b1 > b0) {
      RowFrag<T, D> kf, vf;
      frag(tk, key, kf);
      frag(tv, key, vf);
      const bf16x8 km = antihot_frag(mt.kid[key], h);
      const uint32_t dk_u = d_base + (uint32_t)(key >> 1) * kDropC2 + (uint32_t)(2 * h) * kDropC1;
      const int k_odd = key & 1;
      float ck = 0.f;                                       // this half-wave's share of sum_q dS[q,key]
#pragma unroll
      for (int qb = 0; qb < NB; ++qb) {
        if (qb < b0 || qb >= b1) continue;                  // wave-uniform
        f32x16 s, dp, z;
#pragma unroll
        for (int i = 0; i < 16; ++i) { z[i] = 0.f; dp[i] = 0.f; }
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(onehot_frag(mt.qid[qb * 32 + r], h), km, z, 0, 0, 0);
        mma_rows<D>(tq, PITCH, qb * 32, kf, s, r, h);
        mma_rows<D>(tdo, PITCH, qb * 32, vf, dp, r, h);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int q4 = qb * 32 + 8 * g4 + 4 * h;           // accumulator registers 4*g4 .. 4*g4+3 = queries q4 .. q4+3
          const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q4);
          const float4 d4 = *reinterpret_cast<const float4*>(dl_s + q4);
          const float lq4[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const int i = 4 * g4 + j;                        // registers i, i+1 = queries q4 + j, + 1 (even, odd)
            // rows past the item carry lse = +inf in LDS, other sequences' rows / padding keys sit at -1.3e30: probability 0
            const float p0 = fast_exp2(fmaf(s[i], sl2, -lq4[j])), p1 = fast_exp2(fmaf(s[i + 1], sl2, -lq4[j + 1]));
            float m0 = 1.f, m1 = 1.f;
            if (DROP) drop_pair_k(dk_u + (uint32_t)((qb * 32 + 8 * g4 + j) >> 1) * kDropC1, k_odd, p.drop_thresh, p.keep_scale, m0, m1);
            s[i] = p0 * m0;
            s[i + 1] = p1 * m1;
            dp[i] = p0 * (dp[i] * m0 - dq4[j]);
            dp[i + 1] = p1 * (dp[i + 1] * m1 - dq4[j + 1]);
            if (bgrad) ck += dp[i] + dp[i + 1];
          }
        }
        mma_acc<D>(tdo, PITCH, qb * 32, s, dv, lane);
        mma_acc<D>(tq, PITCH, qb * 32, dp, dk, lane);
      }
      if (bgrad) {
        ck = xhalf_sum(ck);
        if (h == 0) {
          const bf16_t hi = f32_to_bf16(ck);
          coef[key] = hi;
          coef[R + key] = f32_to_bf16(ck - bf16_to_f32(hi));
        }
      }
    }
    if (key < nk) {
      T* dkg = static_cast<T*>(p.dk) + (kbase + key) * p.dk_stride + hd * D;
      T* dvg = static_cast<T*>(p.dv) + (kbase + key) * p.dv_stride + hd * D;
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        store_t<T>(dkg + d * 32, dk[d], p.scale, h);
        store_t<T>(dvg + d * 32, dv[d], 1.f, h);
      }
    }
  }
  // ---- bias gradients: column sums of this workgroup's dQ, dK, dV rows, WITHOUT a cross-lane reduction ----------
  //   sum_q dQ[q,:] = scale * sum_k c_k K[k,:],  c_k = sum_q dS[q,k]          sum_k dV[k,:] = sum_q pm_q dO[q,:],  pm_q = sum_k P'[q,k]
  //   sum_k dK[k,:] = scale * sum_q (sum_k dS[q,k]) Q[q,:] = 0: the rows of dS sum to zero (a key bias shifts every score of
  //   a query alike and the softmax does not see it; the reference's autograd returns rounding noise of order 1e-8 there).
  // Matrix-vector products with tiles that are already in LDS: tile^T on the MFMA A side (ds_read_b64_tr_b16), the coefficient
  // vector broadcast on the B side; one (tensor, 32-column block) unit per wave.  The c_k of a sequence sum to zero, so the
  // product is what is left after a cancellation: c goes in as hi + lo (two passes), not as one bf16 value.
  if (bgrad) {
    __syncthreads();                                       // coefficients of every wave are in LDS
    float* outp = p.dbias_part + (item * 3 * p.h + hd) * D;
    for (int u = w; u < 2 * DB; u += R / 32) {             // wave-uniform
      const int t = u / DB, db = u % DB;                   // t = 0: dQ sums from (K tile, c), t = 1: dV sums from (dO tile, pm)
      const T* tile = (t == 0 ? tk : tdo) + db * 32;
      const int rows = t == 0 ? nk : nq;
      f32x16 acc[1];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] = 0.f;
#pragma unroll
      for (int rb = 0; rb < R / 32; ++rb) {
        if (rb * 32 >= rows) break;
        if (t == 0) {
          mma_acc_coef<32>(tile, PITCH, rb * 32, coef + R, acc, lane);        // lo first: the small terms accumulate before the large
          mma_acc_coef<32>(tile, PITCH, rb * 32, coef, acc, lane);
        } else {
          mma_acc_coef<32>(tile, PITCH, rb * 32, coef + 2 * R, acc, lane);
        }
      }
      if (r == 0) {                                        // every column holds the same vector: column 0 of each half writes its rows
        const float mul = t == 0 ? p.scale : 1.f;
        float* o2 = outp + (t == 0 ? 0 : 2) * p.h * D + db * 32;
#pragma unroll
        for (int i = 0; i < 16; ++i) o2[acc_row(i, h)] = acc[0][i] * mul;
      }
    }
    if (tid < D) outp[p.h * D + tid] = 0.f;               // dK column sums
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static size_t fwd_lds(int r) { return (size_t)2 * r * 64 * sizeof(bf16_t) + (size_t)r * (4 * sizeof(int16_t) + 2); }
static size_t bwd_lds(int r) {
  return (size_t)4 * r * (64 + 8) * sizeof(bf16_t) + 2 * r * sizeof(float) + 3 * r * sizeof(bf16_t) + (size_t)r * (4 * sizeof(int16_t) + 2);
}

// rows: capacity needed (longest sequence, or 128 with groups); items: workgroups per head
int attn_short_fwd_launch(const AttnParams& p, int64_t rows, int64_t items, hipStream_t st) {
  static PerDeviceOnce once;
  int rc = once([&]() -> int {
#define GMLM_SHORT_ATTR(RR)                                                                                                        \
    GMLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_short_kernel<64, true, RR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds(RR))); \
    GMLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_short_kernel<64, false, RR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds(RR)));
    GMLM_SHORT_ATTR(32) GMLM_SHORT_ATTR(64) GMLM_SHORT_ATTR(96) GMLM_SHORT_ATTR(128)
#undef GMLM_SHORT_ATTR
    return GMLM_OK;
  });
  if (rc != GMLM_OK) return rc;
  const int top = (int)((rows + 31) / 32) * 32;
  const unsigned grid = (unsigned)(items * p.h);
#define GMLM_SHORT_LAUNCH(RR)                                                                                                      \
  if (RR == top) {                                                                                                                 \
    if (p.drop_thresh) attn_fwd_short_kernel<64, true, RR><<<grid, 2 * RR, fwd_lds(RR), st>>>(p);                                  \
    else attn_fwd_short_kernel<64, false, RR><<<grid, 2 * RR, fwd_lds(RR), st>>>(p);                                               \
  }
  GMLM_SHORT_LAUNCH(128) GMLM_SHORT_LAUNCH(96) GMLM_SHORT_LAUNCH(64) GMLM_SHORT_LAUNCH(32)
#undef GMLM_SHORT_LAUNCH
  GMLM_LAUNCH_CHECK();
  return GMLM_OK;
}

int attn_short_bwd_launch(const AttnParams& p, int64_t rows, int64_t items, float* dbias, hipStream_t st) {
  static PerDeviceOnce once;
  int rc = once([&]() -> int {
#define GMLM_SHORT_ATTR(RR)                                                                                                        \
    GMLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_short_kernel<64, true, RR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds(RR))); \
    GMLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_short_kernel<64, false, RR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds(RR)));
    GMLM_SHORT_ATTR(32) GMLM_SHORT_ATTR(64) GMLM_SHORT_ATTR(96) GMLM_SHORT_ATTR(128)
#undef GMLM_SHORT_ATTR
    return GMLM_OK;
  });
  if (rc != GMLM_OK) return rc;
  const int top = (int)((rows + 31) / 32) * 32;
  const unsigned grid = (unsigned)(items * p.h);
#define GMLM_SHORT_LAUNCH(RR)                                                                                                      \
  if (RR == top) {                                                                                                                 \
    if (p.drop_thresh) attn_bwd_short_kernel<64, true, RR><<<grid, 2 * RR, bwd_lds(RR), st>>>(p);                                  \
    else attn_bwd_short_kernel<64, false, RR><<<grid, 2 * RR, bwd_lds(RR), st>>>(p);                                               \
  }
  GMLM_SHORT_LAUNCH(128) GMLM_SHORT_LAUNCH(96) GMLM_SHORT_LAUNCH(64) GMLM_SHORT_LAUNCH(32)
#undef GMLM_SHORT_LAUNCH
  GMLM_LAUNCH_CHECK();
  if (dbias) {                                             // [items, 3 h d] partial column sums -> [3 h d], fixed order
    const int64_t width = 3 * p.h * 64;
    rows_sum_kernel<<<(unsigned)cdiv(width, 32), 256, 0, st>>>(p.dbias_part, (int)items, width, dbias, 1.f);
    GMLM_LAUNCH_CHECK();
  }
  return GMLM_OK;
}

}  // namespace gmlm
