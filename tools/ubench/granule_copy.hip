// How fast can the chip move [rows x G bytes] pieces that sit at a 4,608-byte row stride (one head's / two heads' /
// all heads' slice of a fused QKV row)?  Ceiling for the short-sequence attention kernels, whose workgroups read
// exactly that shape.  build: hipcc --offload-arch=gfx950 -O3 tools/ubench/granule_copy.hip -o tools/ubench/granule_copy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// workgroup = 256 threads; item = (row block of RB rows, piece p of width G bytes); copies src -> dst (same layout)
template <int G>
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int rows, int stride16, int rb, int pieces) {
  const int item = blockIdx.x, p = item % pieces, blk = item / pieces;
  constexpr int CPR = G / 16;                       // 16-byte chunks per piece row
  for (int i = threadIdx.x; i < rb * CPR; i += 256) {
    const int row = blk * rb + i / CPR, c = i % CPR;
    if (row < rows) dst[(size_t)row * stride16 + p * CPR + c] = src[(size_t)row * stride16 + p * CPR + c];
  }
}

int main() {
  const int rows = 91698, stride = 4608, stride16 = stride / 16;   // tokens x fused QKV row (bf16, 2304 elements)
  uint4 *a, *b;
  CK(hipMalloc(&a, (size_t)rows * stride)); CK(hipMalloc(&b, (size_t)rows * stride));
  CK(hipMemset(a, 1, (size_t)rows * stride));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rb : {32, 73, 128}) {
    for (int g : {128, 256, 512, 1536, 4608}) {
      const int pieces = stride / g, blocks = (rows + rb - 1) / rb * pieces;
      auto run = [&]() {
        switch (g) {
          case 128: copy_kernel<128><<<blocks, 256>>>(a, b, rows, stride16, rb, pieces); break;
          case 256: copy_kernel<256><<<blocks, 256>>>(a, b, rows, stride16, rb, pieces); break;
          case 512: copy_kernel<512><<<blocks, 256>>>(a, b, rows, stride16, rb, pieces); break;
          case 1536: copy_kernel<1536><<<blocks, 256>>>(a, b, rows, stride16, rb, pieces); break;
          default: copy_kernel<4608><<<blocks, 256>>>(a, b, rows, stride16, rb, pieces); break;
        }
      };
      for (int i = 0; i < 3; ++i) run();
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < 10; ++i) run();
      CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
      printf("rows/block %3d  piece %4d B: %7.1f us  %.2f TB/s (read + write)\n", rb, g, ms * 1e3, 2.0 * rows * stride / ms / 1e9);
    }
  }
  return 0;
}
