// Micro-benchmark: do v_mfma_f32_32x32x16_bf16 and VALU work (fma / exp2) overlap on one SIMD of gfx950?
//   mode 0: MFMA only          mode 1: VALU only (fma)        mode 2: VALU only (exp2)
//   mode 3: MFMA + fma, interleaved in ONE wave             mode 4: MFMA + exp2 interleaved in one wave
//   mode 5: slot-even waves MFMA only, slot-odd waves fma only (two waves per SIMD, different work)
//   mode 6: slot-even waves MFMA only, slot-odd waves exp2 only
//   mode 7: slot-even waves MFMA only, slot-odd waves idle      mode 8: slot-odd waves fma only, slot-even idle
//   mode 9: slot-even waves MFMA, slot-odd waves HALF the fma work (16 / iter)
// Observed on MI355X (round 1; one workgroup per CU, cycles per iteration from s_memtime on SIMD 0):
//   one wave / SIMD : 4 MFMA = 128 cycles; 32 v_fma = 104; 8 v_exp + 8 v_mul = 128; MFMA + fma in one wave = 204
//                     (sum 232); MFMA + exp2 in one wave = 160 (sum 256: the transcendental pipe overlaps).
//   two waves / SIMD: both MFMA = 256 each (the matrix pipe is shared exactly); MFMA wave next to a v_fma wave = 248
//                     for both (= 128 + 104 + 7 %: NO overlap); next to an idle partner the MFMA wave still takes 180.
//   The 16-fma variant came out slower than the 32-fma one, so treat absolute numbers with care: the loop bodies
//   are compiler-scheduled C++, not fixed instruction sequences.  The robust signal — matrix and plain vector work of
//   two co-resident waves add up — agrees with the attention kernels' counters (VALU-busy + MFMA-busy = 85 %).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip ; run: ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
  extern __shared__ float pad_lds[];      // 100 KB of dynamic LDS: at most ONE workgroup per CU
  if (iters < 0) pad_lds[threadIdx.x] = 1.f;
  // role by the hardware wave slot on its SIMD (HW_ID: [3:0] wave slot, [5:4] SIMD), so that with two resident waves
  // per SIMD one of each pair runs MFMAs and the other VALU work
  const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  const int w = (int)(hwid & 0xf);
    bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * j); }
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  const bool do_mfma = MODE == 0 || MODE == 3 || MODE == 4 || ((MODE == 5 || MODE == 6 || MODE == 7 || MODE == 9) && (w & 1) == 0);
  const bool do_fma = MODE == 1 || MODE == 3 || ((MODE == 5 || MODE == 8 || MODE == 9) && (w & 1) == 1);
  const bool do_exp = MODE == 2 || MODE == 4 || (MODE == 6 && (w & 1) == 1);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {   // 4 MFMAs (2 independent chains)
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
    }
    if (do_fma) {    // 32 independent-ish fma
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.0001f);
      if (MODE != 9) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 0.9999f, 0.0002f);
      }
    }
    if (do_exp) {    // 8 exp2
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i] * 0.5f);
    }
    if (do_mfma) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (hwid & 0xf) < 2 && ((hwid >> 4) & 3) == 0) cyc[hwid & 0xf] = t1 - t0;   // SIMD 0, slots 0 and 1
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  float* out; unsigned long long* cyc;
  const int threads = 64 * 4 * waves_per_simd;     // one block per CU, `waves_per_simd` waves on each SIMD
  hipMalloc(&out, sizeof(float) * 256 * threads);
  hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const int iters = 20000;
  k<MODE><<<256, threads, 100 * 1024>>>(out, 10, cyc);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<256, threads, 100 * 1024>>>(out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-44s waves/SIMD %d: %.3f ms; cycles per iteration on SIMD 0: slot0 %.1f  slot1 %.1f\n", name, waves_per_simd, ms, (double)c[0] / iters, (double)c[1] / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("MFMA only (4 x 32x32x16 bf16 / iter)", w);
    run<1>("fma only (32 v_fma / iter)", w);
    run<2>("exp2 only (8 v_exp + 8 v_mul / iter)", w);
    run<3>("MFMA + fma in the same wave", w);
    run<4>("MFMA + exp2 in the same wave", w);
  }
  run<5>("slot-even waves MFMA, slot-odd waves fma", 2);
  run<6>("slot-even waves MFMA, slot-odd waves exp2", 2);
  run<7>("slot-even waves MFMA, slot-odd waves idle", 2);
  run<8>("slot-odd waves fma, slot-even idle", 2);
  run<9>("slot-even waves MFMA, slot-odd waves 16 fma", 2);
  return 0;
}
