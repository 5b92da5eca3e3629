#!/bin/bash
# Build a variant of libm324.so for tools/gemm_lab --lib: tools/build_lab_lib.sh NAME [-DFLAG=V ...]
# (gemm.hip is recompiled with the extra flags, the other objects come from motion324_amd/csrc/build).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/lablibs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form "$@" \
    -c motion324_amd/csrc/gemm.hip -o tools/lablibs/gemm_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize "$@" \
    -c motion324_amd/csrc/gemm_ring4.hip -o tools/lablibs/gemm_ring4_$name.o
b=motion324_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lablibs/libm324_$name.so tools/lablibs/gemm_$name.o \
    $b/runtime.o tools/lablibs/gemm_ring4_$name.o $b/attention.o $b/attention_pwg.o $b/elementwise.o $b/backward.o $b/comm.o -ldl
rm -f tools/lablibs/gemm_$name.o tools/lablibs/gemm_ring4_$name.o
echo tools/lablibs/libm324_$name.so
