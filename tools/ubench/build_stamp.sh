#!/bin/bash
# Diagnostic build: libgmlm_hip with in-kernel cycle stamps in the pipelined attention forward + the bench that reads them.
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/stamp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -mllvm -amdgpu-mfma-vgpr-form -DGMLM_ATTN_STAMP"
for f in core graph_kernels spmm_kernels norm_kernels rowops_kernels basis_kernels; do cp build/csrc/$f.o build/stamp/; done
/opt/rocm/bin/hipcc $F -c gmlm_amd/csrc/attn_kernels.hip -o build/stamp/attn_kernels.o &
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c gmlm_amd/csrc/attn_fwd_pipe.hip -o build/stamp/attn_fwd_pipe.o &
/opt/rocm/bin/hipcc $F -c gmlm_amd/csrc/attn_short.hip -o build/stamp/attn_short.o &
/opt/rocm/bin/hipcc $F -c gmlm_amd/csrc/attn_bwd_pipe.hip -o build/stamp/attn_bwd_pipe.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libgmlm_hip_stamp.so build/stamp/*.o
/opt/rocm/bin/hipcc -O2 -std=c++17 -DGMLM_ATTN_STAMP tools/ubench/attn_bench.cpp -Iinclude -Ltools/ubench -lgmlm_hip_stamp -Wl,-rpath,'$ORIGIN' -o tools/ubench/attn_bench_stamp
