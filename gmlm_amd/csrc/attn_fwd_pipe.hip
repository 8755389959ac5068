// K5/K7 forward, bf16: software-pipelined streaming-softmax attention for gfx950.
// Reference: BertSelfAttention's softmax(QK^T*d^-1/2 + padding mask)V (hf:modeling_bert.py:188-201) and
// CrossAttention.forward's dense [1,8,N,N] softmax (main.py:159-163).
//
// Same data orientation as attn_fwd_kernel (attn_kernels.hip): S^T = K Q^T so that the query sits on the lane and the
// keys in the accumulator registers; P feeds O^T += V^T P^T from registers; V^T through ds_read_b64_tr_b16.
// What is different:
//  * schedule.  The unit of work is 32 keys (one 32x32 score block).  At step u a wave issues, in ONE basic block,
//        QK(u+1) and PV(u-1): 1 + D/16 + 2*D/32 MFMAs
//        SM(u):               exp2 / row sum / bf16 packing of block u -- one probability pair per MFMA gap
//        check(u):            wave vote on the block's ROW SUM (already there: no extra arithmetic): "did some probability of
//                             block u leave the deferral window", see `settle` below
//    hand-interleaved (one MFMA per gap, fenced by sched_barrier(0)), so the matrix pipe works on the neighbouring
//    blocks while the VALU does the softmax of this one: the two pipes overlap INSIDE a wave.
//  * softmax arithmetic per score = v_exp + v_add (+ half a v_cvt_pk, half a v_max3).  Q is pre-multiplied by
//    scale * log2(e), and the reference max rides in the contraction: one extra MFMA k-step with a constant
//    [1 0 .. 0] fragment against [-m 0 .. 0]^T, so scores leave the MFMA chain already max-subtracted (m is kept
//    bf16-representable, the product is exact).  The rescale is deferred (guide T13) and, when it fires, O is scaled
//    AFTER the pending PV(u-1) has been accumulated: everything summed so far is at the old scale exactly once.
//  * staging.  K/V tiles (64 keys) arrive by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) into XOR-swizzled
//    images whose operand reads are bank-conflict free; the DMA pieces ride in the MFMA gaps.  K is consumed one
//    tile ahead of V, so two buffers each need ONE workgroup barrier per tile.
// Measured (MI355X, h = 8, d = 96, N = 20,804): 1.95 ms -> 1.59 ms against attn_fwd_kernel; what each step bought and
// what did not pay (3-deep LDS ring, wave stagger, compiler-tracked DMA builtin) is in DESIGN.md section 5.
#include "attn_dma.hpp"

#include <type_traits>

namespace gmlm {

template <int D, int NW, bool DROP, int NB>
__global__ __launch_bounds__(NW * 64) void attn_fwd_pipe_kernel(AttnParams p) {
  using T = bf16_t;
  using G = Img<D>;
  constexpr int KT = 64, DB = D / 32, NQ = D / 16, NP = 2 * DB, NM = NQ + 1 + NP, RP = G::RP;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_dyn[];      // K ring [NB][TILE], then V ring [NB][TILE]
#ifdef GMLM_ATTN_STAMP
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  // 1-D grid of (query block, (batch, head) pair) items.  Workgroups are dealt round-robin over the 8 XCDs (id % 8 says
  // which ids share an XCD, hence an L2): the query blocks of one pair are mapped to ids with equal id % 8 and
  // consecutive id / 8, so that a pair's K / V is fetched into ONE L2, once, instead of into up to nq of them
  // (speed only, never correctness: guide T1).
  const uint32_t nq = p.grid_q, npairs = p.grid_pairs, id = blockIdx.x;
  uint32_t qb, pair;
  if (nq > 1 && npairs % 8 == 0) { const uint32_t slot = id >> 3; pair = (slot / nq) * 8 + (id & 7); qb = slot % nq; }
  else { qb = id % nq; pair = id / nq; }
  const uint32_t nh = (uint32_t)p.h;
  const int64_t b = pair / nh, hd = pair % nh;        // 32-bit division (the 64-bit form is ~80 scalar instructions)
  int64_t lq_, lk_, qbase, kbase, lse_base;
  seq_view(p, b, hd, lq_, lk_, qbase, kbase, lse_base);
  if ((int64_t)qb * (NW * 32) >= lq_) return;                    // varlen: tile past this sequence (block-uniform)
  const int64_t q_row = (int64_t)qb * (NW * 32) + w * 32 + r;
  const bool q_ok = q_row < lq_;
  // a wave whose 32 queries all lie past the sequence (short packed sequences) only helps staging K/V
  const bool wave_live = (int64_t)qb * (NW * 32) + w * 32 < lq_;
  int64_t kvlen64 = lk_;
  if (p.kv_len) { kvlen64 = p.kv_len[b]; if (kvlen64 > lk_) kvlen64 = lk_; if (kvlen64 < 0) kvlen64 = 0; }
  const int kvlen = (int)(kvlen64 < (1 << 30) ? kvlen64 : (1 << 30));
  const T* qg = static_cast<const T*>(p.q) + (qbase + (q_ok ? q_row : 0)) * p.q_stride + hd * D;
  const T* kg = static_cast<const T*>(p.k) + kbase * p.k_stride + hd * D;
  const T* vg = static_cast<const T*>(p.v) + kbase * p.v_stride + hd * D;
  // Q' = bf16(Q * scale * log2 e): scores leave the MFMA in the log2 domain (one more bf16 rounding of q, the size of
  // the rounding q already carries).  The extra contraction step [1 0 .. 0] x [-m 0 .. 0]^T subtracts the reference
  // max m (kept bf16-representable, so the product is exact) inside the MFMA chain: p = exp2(s') needs no fma.
  const int nunits = (kvlen + 31) >> 5;       // 32-key score blocks that hold at least one valid key
  const int ntiles = (nunits + 1) >> 1;
  const int last_valid = kvlen - 32 * (nunits - 1);   // valid keys of the last block (1..32)
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem_dyn;
#ifdef GMLM_ATTN_STAMP
  uint64_t t_steps = 0, t_stage = 0, t_bar = 0, t_load = 0, t0_, t1_, t2_, t3_, t4_, t_main = 0, t_tail = 0, t_loop_end = 0, t_landed = 0;
#define STAMP(x) x = __builtin_amdgcn_s_memtime()
#else
#define STAMP(x)
#endif
  // The first tiles are requested before anything else (K tile 0, V tile 0, K tile 1): their latency runs under the
  // Q fetch and the lane-constant set-up below instead of behind it.
  DmaPlan<D, NW, false> kd;
  DmaPlan<D, NW, true> vd;
  kd.init(p.k_stride, w, lane);
  vd.init(p.v_stride, w, lane);
  if (ntiles > 0) {
    kd.issue(kg, p.k_stride, 0, lk_, lds0, w, lane);
    vd.issue(vg, p.v_stride, 0, lk_, lds0 + (uint32_t)(NB * G::TILE), w, lane);
    if (ntiles > 1) kd.issue(kg, p.k_stride, KT, lk_, lds0 + (uint32_t)G::TILE, w, lane);
  }
  RowFrag<T, D> qf;
  qf.load(qg, q_ok, h);
  const float sl2 = p.scale * kLog2e;
#pragma unroll
  for (int s = 0; s < NQ; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) qf.v[s][j] = (__bf16)((float)qf.v[s][j] * sl2);
  bf16x8 ones, qx;
#pragma unroll
  for (int j = 0; j < 8; ++j) { ones[j] = (__bf16)0.f; qx[j] = (__bf16)0.f; }
  // Key masking rides in the same extra contraction step: its second column is [key masked ? 1 : 0] x [-2^100], so a masked
  // key's score leaves the MFMA chain at about -1.3e30 and its probability is exp2(.) = 0 exactly, with no compare / select
  // per score.  Only the last block of a sequence can hold masked keys: `ones` (no key masked) serves every other block.
  if (h == 0) { ones[0] = (__bf16)1.f; qx[1] = (__bf16)(-0x1p100f); }
  auto ones_for = [&](int blk) {                       // A fragment of the extra step for block blk (prologue / tail only)
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));                   // opaque: keeps the mask fragment out of the registers held across the hot loop
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)0.f;
    if (lane_t < 32) { a[0] = (__bf16)1.f; a[1] = (blk == nunits - 1 && lane_t >= last_valid) ? (__bf16)1.f : (__bf16)0.f; }
    return a;
  };
  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  float m = 0.f, l = 0.f;      // reference max (log2 domain, bf16-representable, identical in both half-waves) / this half-wave's partial row sum
  const uint32_t dq_u = drop_base(attn_seed(p), lse_base) + (uint32_t)(q_row >> 1) * kDropC1 + (uint32_t)(2 * h) * kDropC2;   // dropout hash input of the lane's query
  const int q_odd = (int)(q_row & 1);
  // lane constants of the LDS operand reads: ABSOLUTE LDS byte addresses of the lane's spot in tile buffer 0 of the K
  // ring; everything else (buffer, V ring, row block, contraction step) is an immediate offset in the hot loop
  uint32_t koff[NQ];                                    // K A-operand, contraction step s: row r, chunk 2s + h
#pragma unroll
  for (int s = 0; s < NQ; ++s) koff[s] = lds0 + r * RP + (((2 * s + h) ^ G::fk(r)) << 4);
  uint32_t voff[DB];                                    // V^T A-operand, d-block db: row 4*(lane>>5) + qq, chunk 4 db + 2 g1 + (pp >> 1)
  {
    const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1, row = 4 * (lane >> 5) + qq;
#pragma unroll
    for (int db = 0; db < DB; ++db) voff[db] = lds0 + row * RP + (((4 * db + 2 * g1 + (pp >> 1)) ^ G::fv(row)) << 4) + 8 * (pp & 1);
  }
  typedef __attribute__((address_space(3))) const bf16x8* lds_b128_t;
  typedef __attribute__((address_space(3))) bf16x4* lds_b64_t;
  // off: byte offset of the tile buffer + row block from the start of the K ring (runtime in the generic step,
  // a compile-time constant in the hot loop, where it folds into the ds_read offset field)
  auto kfrag = [&](uint32_t off, int s) { return *(lds_b128_t)(size_t)(koff[s] + off); };
  auto vfrag = [&](uint32_t off, int j, bf16x4& lo, bf16x4& hi) {   // PV MFMA j: k-step j / DB, d-block j % DB
    const uint32_t a = voff[j % DB] + off + (uint32_t)(16 * (j / DB) * RP);
    lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(size_t)a);
    hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(size_t)(a + 8 * RP));
  };
  auto k_at = [&](int buf, int row0) { return (uint32_t)(buf * G::TILE + row0 * RP); };            // K ring
  auto v_at = [&](int buf, int row0) { return (uint32_t)((NB + buf) * G::TILE + row0 * RP); };     // V ring (behind the K ring)

  // probabilities of scores 2j, 2j+1 of block u: exp2, (dropout), bf16 pair.  The row-sum adds run ONE PAIR BEHIND
  // (pe0 / pe1 hold the previous pair's exponentials): a v_add that consumes a v_exp issued two instructions earlier
  // stalls on the transcendental pipe's latency; consuming the previous gap's pair does not.
  auto sm_pair = [&](const f32x16& sc, bf16x8 (&pc)[2], int u, int j, float& rs0, float& rs1, float& pe0, float& pe1) {
    const int i = 2 * j;
    const float e0 = fast_exp2(sc[i]);                               // masked: exp2(-inf) = 0
    const float e1 = fast_exp2(sc[i + 1]);
#ifndef GMLM_NO_PIPE_ADD
    if (j == 1) { rs0 = pe0; rs1 = pe1; }                            // the normaliser uses the un-dropped probabilities
    else if (j > 1) { rs0 += pe0; rs1 += pe1; }
    pe0 = e0; pe1 = e1;
#else
    rs0 += e0; rs1 += e1;
#endif
    float d0 = e0, d1 = e1;
    if (DROP) {
      float m0, m1;
      drop_pair_q(dq_u + (uint32_t)(16 * u + (acc_row(i, 0) >> 1)) * kDropC2, q_odd, p.drop_thresh, p.keep_scale, m0, m1);
      d0 = e0 * m0;
      d1 = e1 * m1;
    }
    pc[i >> 3][i & 7] = (__bf16)d0;
    pc[i >> 3][(i & 7) + 1] = (__bf16)d1;
  };
  // S'^T(block) = K(rows row0..row0+31) Q'^T - m: the max column first, then the D/16 real contraction steps
  auto qk_unit = [&](uint32_t koffs, f32x16& s, const bf16x8& ones_a) {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones_a, qx, z, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NQ; ++t) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(koffs, t), qf.v[t], s, 0, 0, 0);
  };
  // O^T[db] += V^T(tile rows row0..row0+31) * P^T, P packed to bf16 in accumulator order (k-step s = regs 8s..8s+7)
  auto pv_unit = [&](uint32_t voffs, const bf16x8 (&pk)[2]) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      bf16x4 lo, hi;
      vfrag(voffs, j, lo, hi);
      bf16x8 a;
#pragma unroll
      for (int t = 0; t < 4; ++t) { a[t] = lo[t]; a[4 + t] = hi[t]; }
      o[j % DB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pk[j / DB], o[j % DB], 0, 0, 0);
    }
  };
  // Block u against the reference max m.  The exponentials of block u are formed against the CURRENT reference without looking
  // at the scores first; the lane's partial row sum of the block then tells whether that was fine: 16 probabilities that sum to
  // <= 2^8 are each <= 2^8 (bf16 P keeps its 8 mantissa bits at any scale, the f32 sums have room), and a probability beyond
  // that - or an overflow to +inf for a score that jumped by more than 2^7 - makes the sum exceed the bound (NaN-safe compare).
  // That replaces the per-block max scan of the scores (8 v_max3 + canonicalisation: 13 of ~70 VALU per block).  When the vote
  // fails (first block, or a real jump) the block is REDONE out of line: reference <- bf16(row max), exponentials, packing and
  // row sum recomputed, the scores of block u+1 (already produced against the old reference by this step's QK) shifted, l and -
  // after the pending PV(u-1), which this step has finished - O rescaled: everything summed so far is at the old scale
  // exactly once (guide T13 hazard).  The probabilities of block u are consumed by the NEXT step's PV, so nothing stale is used.
  constexpr float kSumDefer = 256.f;
  auto scale_o = [&](float alpha) {
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
  };
  auto settle = [&](f32x16& sc, bf16x8 (&pc)[2], f32x16& sn, int u) {
    const float gr = xhalf_max(max16(sc));               // finite: the first key of every block is valid
    float m_new = bf16_round(m + gr);
    if (u != 0) m_new = fmaxf(m, m_new);                 // only ever raised after the first block
    const float delta = m_new - m;
    if (u != 0) {                                        // first block: O = l = 0, nothing to scale (and -delta may be huge)
      const float alpha = fast_exp2(-delta);
      scale_o(alpha);
      l *= alpha;
    }
    m = m_new;
    qx[0] = h == 0 ? (__bf16)(-m_new) : (__bf16)0.f;
    __builtin_amdgcn_sched_barrier(0);                   // keep the register pressure of this rare path sequential: O first, then the block
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const float e0 = fast_exp2(sc[i] - delta), e1 = fast_exp2(sc[i + 1] - delta);
      sn[i] -= delta; sn[i + 1] -= delta;
      r0 += e0; r1 += e1;
      float d0 = e0, d1 = e1;
      if (DROP) {
        float m0, m1;
        drop_pair_q(dq_u + (uint32_t)(16 * u + (acc_row(i, 0) >> 1)) * kDropC2, q_odd, p.drop_thresh, p.keep_scale, m0, m1);
        d0 = e0 * m0;
        d1 = e1 * m1;
      }
      pc[i >> 3][i & 7] = (__bf16)d0;
      pc[i >> 3][(i & 7) + 1] = (__bf16)d1;
      __builtin_amdgcn_sched_barrier(0);
    }
    l += r0 + r1;
  };
  // ---- steady-state step: PV(u-1) and QK(u+1) on the matrix pipe under the softmax of block u ---------------
  //  sc: scores of block u (log2 domain, reference max subtracted)   pc: packed probabilities of block u (output)
  //  sn: receives the scores of block u+1                              pp: packed probabilities of block u-1
  //  ones_a: extra-step A fragment of block u+1.  Every step does all of it: at the ends of the sequence PV(-1) runs with
  //  P = 0 against a loaded V tile and QK(nunits) produces scores nobody reads.
  auto step_full = [&](f32x16& sc, bf16x8 (&pc)[2], f32x16& sn, const bf16x8 (&pp)[2], int u,
                       auto voffs_c, auto koffs_c, const bf16x8& ones_a, auto&& gap_hook) {
    const uint32_t voffs = voffs_c, koffs = koffs_c;   // integral_constant in the hot loop (folds into the ds_read offset field), runtime in the tail
    // operand reads of the first MFMAs go out before the check: its ~12 VALU instructions cover their latency
    bf16x8 ka[NQ];
#pragma unroll
    for (int s = 0; s < NQ; ++s) ka[s] = kfrag(koffs, s);
    bf16x4 vlo[NP], vhi[NP];
    constexpr int LA = 3;                               // PV operand reads run LA gaps ahead of their MFMA
    // hand-placed: one MFMA per gap, the operand reads of a later MFMA, one probability pair;
    // sched_barrier(0) keeps every gap's instructions inside the gap
    float rs0 = 0.f, rs1 = 0.f, pe0 = 0.f, pe1 = 0.f;
    int pair = 0;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NM; ++g) {
      if (g == 0) {
        f32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
        sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones_a, qx, z, 0, 0, 0);
      } else if (g <= NQ) {
        sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[g - 1], qf.v[g - 1], sn, 0, 0, 0);
      } else {
        const int j = g - NQ - 1;
        bf16x8 a;
#pragma unroll
        for (int t = 0; t < 4; ++t) { a[t] = vlo[j][t]; a[4 + t] = vhi[j][t]; }
        o[j % DB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pp[j / DB], o[j % DB], 0, 0, 0);
      }
      {
        const int j = g + LA - NQ - 1;                  // PV MFMA whose operands are fetched in this gap
        if (j >= 0 && j < NP) vfrag(voffs, j, vlo[j], vhi[j]);
      }
      gap_hook(g);
      if (pair < 8 && (NM <= 9 || (g % 3) != 2)) { sm_pair(sc, pc, u, pair, rs0, rs1, pe0, pe1); ++pair; }
      __builtin_amdgcn_sched_barrier(0);
    }
    // pin the probabilities inside this basic block (their consumers are in later blocks: without a use here the
    // compiler sinks the whole softmax below the MFMAs, past the branch that follows)
    asm volatile("" : "+v"(pc[0]), "+v"(pc[1]), "+v"(rs0), "+v"(rs1), "+v"(pe0), "+v"(pe1));
    const float rs = (rs0 + pe0) + (rs1 + pe1);
    if (u == 0 || !__all(rs <= kSumDefer)) settle(sc, pc, sn, u);      // rare, out of line
    else l += rs;
  };

  f32x16 sa, sb;
  bf16x8 pa[2], pb[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) { pb[0][j] = (__bf16)0.f; pb[1][j] = (__bf16)0.f; }       // "P(-1)" = 0 for step 0's PV
  static_assert(NB == 2, "the parity-unrolled loop assumes two buffers per operand");
  constexpr int PK = DmaPlan<D, NW, false>::PER;       // DMA pieces per tile and wave
  if (ntiles > 0) {
    // Invariant at the top of iteration i (steps 2i-1 and 2i; reads K tile i and V tile i-1): both are in LDS and
    // published; K tile i+1 and V tile i have not been requested.  The iteration requests them (K tile i+1 into the
    // buffer K tile i-1 left, V tile i into the buffer V tile i-2 left, both last read before the previous barrier),
    // waits for its own DMA and closes with the ONE barrier per 64 keys.
    // prologue (tiles requested at the top of the kernel): QK(0); step 0
    dma_wait();
    __syncthreads();
    STAMP(t_landed);
#pragma unroll
    for (int s = 0; s < NQ; ++s) asm volatile("" :: "v"(qf.v[s]));   // Q has landed too: no vmcnt wait for it inside the loop
    if (wave_live) {
      qk_unit(k_at(0, 0), sa, ones_for(0));
      step_full(sa, pa, sb, pb, 0, std::integral_constant<uint32_t, (uint32_t)(NB * G::TILE)>{},
                std::integral_constant<uint32_t, (uint32_t)(32 * RP)>{}, ones_for(1), [](int) {});
    }
    __syncthreads();                                    // every wave is done with K tile 0 before iteration 1 refills its buffer
    // ---- main loop.  HOT iterations: both steps have a block behind and a NON-LAST block ahead (so the extra-step
    // fragment is the constant `ones`), and the requested tiles are INTERIOR (all 64 rows inside the slab: the DMA
    // pieces need no per-lane clamping and ride in the MFMA gaps).  TAIL iterations (the last one or two): clamping
    // DMA issued up front, the masked fragment where the next block is the last.  Unrolled by the buffer parity:
    // every LDS operand address is lane constant + immediate.
    int i = 1;
    STAMP(t_main);
    auto hot = [&](int ii) { return 2 * ii + 2 < nunits && (int64_t)(ii + 2) * KT <= lk_; };
    auto iteration = [&](auto par_c, int ii) {
      constexpr int PAR = decltype(par_c)::value;       // = ii & 1: K tile ii lives in K buffer PAR, V tile ii-1 in V buffer PAR ^ 1
      STAMP(t0_);
      const T* kbase = kg + (int64_t)(ii + 1) * KT * p.k_stride;      // wave-uniform: SGPR pair
      const T* vbase = vg + (int64_t)ii * KT * p.v_stride;
      auto dma_hook = [&](int g) {                      // this iteration's pieces ride in gaps 1 .. NM-1 of its first step, K first
        constexpr int C = (2 * PK + NM - 2) / (NM - 1);
        if (g < 1) return;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const int q = (g - 1) * C + c;
          if (q < PK) kd.piece_fast(q, kbase, lds0 + (uint32_t)((PAR ^ 1) * G::TILE), w);
          else if (q < 2 * PK) vd.piece_fast(q - PK, vbase, lds0 + (uint32_t)((NB + PAR) * G::TILE), w);
        }
      };
      using VO0 = std::integral_constant<uint32_t, (uint32_t)((NB + (PAR ^ 1)) * G::TILE)>;
      using VO1 = std::integral_constant<uint32_t, (uint32_t)((NB + (PAR ^ 1)) * G::TILE + 32 * RP)>;
      using KO0 = std::integral_constant<uint32_t, (uint32_t)(PAR * G::TILE)>;
      using KO1 = std::integral_constant<uint32_t, (uint32_t)(PAR * G::TILE + 32 * RP)>;
      STAMP(t1_);
      if (wave_live) {
        step_full(sb, pb, sa, pa, 2 * ii - 1, VO0{}, KO0{}, ones, dma_hook);
        step_full(sa, pa, sb, pb, 2 * ii, VO1{}, KO1{}, ones, [](int) {});
      } else {
#pragma unroll
        for (int g = 0; g < NM; ++g) dma_hook(g);
      }
      STAMP(t2_);
      dma_wait();
      STAMP(t3_);
      __syncthreads();
      STAMP(t4_);
#ifdef GMLM_ATTN_STAMP
      t_load += t1_ - t0_; t_steps += t2_ - t1_; t_stage += t3_ - t2_; t_bar += t4_ - t3_;
#endif
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    for (; hot(i + 1); i += 2) {                        // i is odd at the top; hot(i + 1) implies hot(i)
      iteration(P1{}, i);
      iteration(P0{}, i + 1);
    }
    if (hot(i)) {                                       // one more hot iteration (odd i)
      iteration(P1{}, i);
      ++i;
    }
    STAMP(t_tail);
    // ---- tail (the last one or two iterations): same steps with run-time buffer offsets, clamping DMA issued up front
    for (; 2 * i - 1 < nunits; ++i) {
      const int par = i & 1;
      {
        // (lane made opaque: otherwise the per-lane row / column of every clamped piece is hoisted out of this loop and
        // held in registers across the hot loop)
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        if (i + 1 < ntiles) kd.issue(kg, p.k_stride, (int64_t)(i + 1) * KT, lk_, lds0 + (uint32_t)((par ^ 1) * G::TILE), w, lane_t);
        if (i < ntiles) vd.issue(vg, p.v_stride, (int64_t)i * KT, lk_, lds0 + (uint32_t)((NB + par) * G::TILE), w, lane_t);
      }
      if (wave_live) {
        const uint32_t vo = (uint32_t)((NB + (par ^ 1)) * G::TILE), ko = (uint32_t)(par * G::TILE);
        step_full(sb, pb, sa, pa, 2 * i - 1, vo, ko, ones_for(2 * i), [](int) {});
        if (2 * i < nunits) step_full(sa, pa, sb, pb, 2 * i, vo + (uint32_t)(32 * RP), ko + (uint32_t)(32 * RP), ones_for(2 * i + 1), [](int) {});
      }
      dma_wait();
      __syncthreads();
    }
    // the last block's PV is still pending: block nunits-1 = rows 32 * ((nunits - 1) & 1) of V tile ntiles-1, its
    // probabilities in pa (even block) / pb (odd block)
    if (wave_live) {
      if (nunits & 1) pv_unit(v_at((ntiles - 1) % NB, 0), pa);
      else pv_unit(v_at((ntiles - 1) % NB, 32), pb);
    }
  }
  STAMP(t_loop_end);
  if (q_ok) {
    l = xhalf_sum(l);
    const float inv = l > 0.f ? __builtin_amdgcn_rcpf(l) : 0.f;      // 1 ulp: the output is rounded to bf16 right after
    T* og = static_cast<T*>(p.o_w) + ((qbase + q_row) * p.h + hd) * D;
#pragma unroll
    for (int d = 0; d < DB; ++d) store_t<T>(og + d * 32, o[d], inv, h);
    if (p.o_lo_w) {
      T* olg = static_cast<T*>(p.o_lo_w) + ((qbase + q_row) * p.h + hd) * D;
#pragma unroll
      for (int d = 0; d < DB; ++d) store_t_lo(olg + d * 32, o[d], inv, h);
    }
    if (h == 0 && p.lse_w) p.lse_w[lse_base + q_row] = l > 0.f ? (m + __log2f(l)) * kLn2 : -INFINITY;
  }
#ifdef GMLM_ATTN_STAMP
  if (lane == 0 && p.delta) {     // diagnostic build: p.delta = uint64 [waves][12]
    uint64_t* dbg = reinterpret_cast<uint64_t*>(p.delta) + ((size_t)blockIdx.x * NW + w) * 12;
    dbg[0] = t_load; dbg[1] = t_steps; dbg[2] = t_stage; dbg[3] = t_bar; dbg[4] = __builtin_amdgcn_s_memtime() - t_begin; dbg[5] = (uint64_t)nunits;
    dbg[6] = t_begin; dbg[3] = t_landed - t_begin; dbg[7] = t_main; dbg[8] = t_tail; dbg[9] = t_loop_end; dbg[10] = __builtin_amdgcn_s_memtime();
    dbg[11] = ((uint64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (uint32_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // XCC_ID, HW_ID
  }
#endif
}

template <int D, int NW, int NB>
static int launch_pipe(dim3 grid2, hipStream_t st, const AttnParams& p0) {
  AttnParams p = p0;
  p.grid_q = grid2.x; p.grid_pairs = grid2.y;
  const dim3 grid(grid2.x * grid2.y);
  constexpr int kLds = 2 * NB * Img<D>::TILE;
  static PerDeviceOnce once;                             // > 64 KiB of dynamic LDS needs the attribute once per instantiation and device
  const int rc = once([&]() -> int {
    GMLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_pipe_kernel<D, NW, true, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    GMLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_pipe_kernel<D, NW, false, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    return GMLM_OK;
  });
  if (rc != GMLM_OK) return rc;
  if (p.drop_thresh) attn_fwd_pipe_kernel<D, NW, true, NB><<<grid, NW * 64, kLds, st>>>(p);
  else attn_fwd_pipe_kernel<D, NW, false, NB><<<grid, NW * 64, kLds, st>>>(p);
  return GMLM_OK;
}

// bf16 forward entry used by gmlm_attention_fwd (attn_kernels.hip); nw = waves per workgroup.  Only the
// configurations that are dispatched are instantiated.
int attn_fwd_pipe_launch(const AttnParams& p, int d, int nw, int64_t rows_q, int64_t bh, hipStream_t st) {
  dim3 grid((unsigned)cdiv(rows_q, nw * 32), (unsigned)bh);
  if (d == 96 && nw == 8) return launch_pipe<96, 8, 2>(grid, st, p);
  if (d == 96 && nw == 4) return launch_pipe<96, 4, 2>(grid, st, p);
  if (d == 64 && nw == 4) return launch_pipe<64, 4, 2>(grid, st, p);
  set_error("attention_fwd: no pipelined kernel for d = %d with %d waves", d, nw);
  return GMLM_EINVAL;
}

}  // namespace gmlm
