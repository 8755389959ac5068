// LDS-DMA staging of 64-row K / V / Q / dO tiles for the pipelined attention kernels (attn_fwd_pipe.hip, attn_bwd_pipe.hip).
#pragma once
#include "attn_common.hpp"

namespace gmlm {

// round-to-nearest bf16 value of x, as f32
__device__ __forceinline__ float bf16_round(float x) { return (float)(__bf16)x; }

// ---- LDS images of a 64-row K / V tile, filled by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB of
// CONTIGUOUS LDS per wave-instruction, per-lane source address) --------------------------------------------------
// Rows are RP = 128 B (d = 64) or 256 B (d = 96: 12 real 16-byte chunks + 4 filler chunks, so that a row spans all
// 64 banks) with no padding; bank conflicts are avoided by an XOR on the chunk index applied through the SOURCE
// address (chunk c of row r is stored in slot c ^ f(r)) and again on the read (guide rule 21):
//   K image, read as MFMA A-operand rows (ds_read_b128, 16 lanes = 16 different rows, same chunk):
//       d = 64: f = (r >> 1) & 7 (with the row parity that is 16 distinct 16-byte slots);  d = 96: f = r & 15
//   V image, read through ds_read_b64_tr_b16 (a half-wave = 4 rows x 64 contiguous bytes):
//       d = 64: f = 4 * ((r >> 1) & 1);  d = 96: f = 4 * (r & 3)   -> the four rows land in four different bank quarters
template <int D> struct Img {
  static constexpr int RP = D == 64 ? 128 : 256;        // row pitch, bytes
  static constexpr int CPR = RP / 16;                   // 16-byte slots per row
  static constexpr int TILE = 64 * RP;                  // bytes per 64-row tile
  static constexpr int NDMA = TILE / 1024;              // wave-instructions per tile
  __device__ static __forceinline__ int fk(int r) { return D == 64 ? ((r >> 1) & 7) : (r & 15); }
  __device__ static __forceinline__ int fv(int r) { return D == 64 ? 4 * ((r >> 1) & 1) : 4 * (r & 3); }
  // DUAL-USE image, 256-byte rows (d = 96; guide T10 image (b)): ONE tile serves the row reads of one MFMA (ds_read_b128) AND the
  // transposed reads of another (ds_read_b64_tr_b16), both bank-conflict free: slot = chunk ^ (((r & 3) << 2) | ((r >> 2) & 3))
  __device__ static __forceinline__ int fd(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
};

// this wave's share of the DMA instructions of one tile: instruction I = w + NW * k writes LDS bytes [1024 I, 1024 I + 1024)
template <int D, int NW, int SW>
struct DmaPlan {
  using G = Img<D>;
  static constexpr int PER = G::NDMA / NW;
  // Piece k covers rows STEP * k + (rows of piece 0), and STEP is a multiple of 16, so the swizzle (a function of the low
  // four row bits) and with it the lane's column are the same in every piece: ONE per-lane byte offset serves all pieces,
  // piece k adds the wave-uniform k * STEP * stride to the SGPR base.
  static constexpr int STEP = 64 * NW / G::CPR;
  static_assert(STEP % 16 == 0, "pieces must preserve the swizzle phase");
  uint32_t boff0;       // BYTE offset of this lane's source chunk inside piece 0: 2 * (row * stride + 8 * chunk)
  int64_t stride_;
  __device__ static __forceinline__ void where(int k, int w, int lane, int& row, int& col) {
    const int idx = 64 * (w + NW * k) + lane, slot = idx % G::CPR;
    row = idx / G::CPR;
    int chunk = slot ^ (SW == 2 ? G::fd(row) : (SW == 1 ? G::fv(row) : G::fk(row)));
    if (chunk >= D / 8) chunk = D / 8 - 1;              // filler slots of the d = 96 image: any valid address
    col = 8 * chunk;
  }
  __device__ __forceinline__ void init(int64_t g_stride, int w, int lane) {
    int row, col;
    where(0, w, lane, row, col);
    boff0 = 2u * (uint32_t)(row * g_stride + col);
    stride_ = g_stride;
  }
  // LDS-DMA through inline asm on purpose: hipcc orders a builtin LDS-DMA against every later ds_read it cannot prove
  // disjoint (s_waitcnt vmcnt(0) right behind the DMA), which exposes the whole load latency.  The asm form is
  // invisible to that bookkeeping; the kernel waits for its DMA itself (dma_wait() ahead of the barrier that
  // publishes the tile).  M0 = LDS base of the piece, saved / restored around the instruction (guide 5.7, glds16).
  //
  // Hot-loop form, INTERIOR tiles only (all 64 rows inside the slab): wave-uniform 64-bit base in SGPRs + the lane's
  // 32-bit byte offset: no VALU address arithmetic at all inside the MFMA gaps.
  __device__ __forceinline__ void piece_fast(int k, const bf16_t* base /* g + row0 * stride, wave-uniform */, uint32_t tile, int w) const {
    const uint32_t dst_u = __builtin_amdgcn_readfirstlane(tile + 1024u * (uint32_t)(w + NW * k));
    const bf16_t* src = base + (int64_t)(k * STEP) * stride_;          // wave-uniform: scalar add
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(boff0), "s"(src), "s"(dst_u) : "memory");
  }
  // General form (prologue / tail): rows past the slab re-read its last row (finite data; their scores are masked /
  // their probabilities are exactly 0).  g: start of the (batch, head) slab; limit: rows of the slab (>= 1, row0 < limit).
  __device__ __forceinline__ void issue(const bf16_t* g, int64_t g_stride, int64_t row0, int64_t limit, uint32_t tile, int w, int lane) const {
    const int left = (int)(limit - row0 < (1 << 29) ? limit - row0 : (1 << 29));
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      int row, col;
      where(k, w, lane, row, col);
      const bf16_t* src = g + (row0 + (row < left ? row : left - 1)) * g_stride + col;
      const uint32_t dst_u = __builtin_amdgcn_readfirstlane(tile + 1024u * (uint32_t)(w + NW * k));
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst_u) : "memory");
    }
  }
};

// all LDS-DMA pieces this wave has issued have landed (vmcnt also counts ordinary loads / stores: none are pending in the loop)
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace gmlm
