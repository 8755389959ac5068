#!/bin/bash
# A/B builds of libgmlm_hip.so for in-one-run comparisons on the SAME device (MI355X boxes differ by several % on
# MFMA-heavy kernels: never rank builds across gpurun calls).  usage: ab_build.sh NAME "extra hipcc flags"
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
out=build/ab/$name
mkdir -p $out/gmlm_amd $out/tools/ubench
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -Iinclude -mllvm -amdgpu-mfma-vgpr-form $*"
/opt/rocm/bin/hipcc $F -c gmlm_amd/csrc/attn_kernels.hip -o $out/attn_kernels.o &
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c gmlm_amd/csrc/attn_fwd_pipe.hip -o $out/attn_fwd_pipe.o &
/opt/rocm/bin/hipcc $F -c gmlm_amd/csrc/attn_short.hip -o $out/attn_short.o &
/opt/rocm/bin/hipcc $F -c gmlm_amd/csrc/attn_bwd_pipe.hip -o $out/attn_bwd_pipe.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/gmlm_amd/libgmlm_hip.so build/csrc/core.o build/csrc/graph_kernels.o build/csrc/spmm_kernels.o build/csrc/norm_kernels.o build/csrc/rowops_kernels.o build/csrc/basis_kernels.o $out/attn_kernels.o $out/attn_fwd_pipe.o $out/attn_short.o $out/attn_bwd_pipe.o
cp tools/ubench/attn_bench $out/tools/ubench/
echo built $out
