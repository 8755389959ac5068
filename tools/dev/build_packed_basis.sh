#!/bin/bash
# The library with basis_kernels.hip compiled WITH the SLP vectoriser (what csrc/Makefile avoids): the A side of
# the soaks in DESIGN.md section 5 (b): BF16=1 GMLM_LIB=build/ab/packed/gmlm_amd/libgmlm_hip.so python tools/dev/replay_diff.py conc 02100210...  Needs the product build's objects (make -C gmlm_amd/csrc).
set -e
cd "$(dirname "$0")/../.."
out=build/ab/packed
mkdir -p $out/gmlm_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -mllvm -amdgpu-mfma-vgpr-form -c gmlm_amd/csrc/basis_kernels.hip -o $out/basis_kernels.o
objs=$(ls build/csrc/*.o | grep -v basis_kernels)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/gmlm_amd/libgmlm_hip.so $objs $out/basis_kernels.o
echo built $out/gmlm_amd/libgmlm_hip.so
