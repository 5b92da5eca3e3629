#!/bin/bash
# A/B build of attention.hip: tools/build_attn_lab.sh NAME [-DFLAG ...]  ->  tools/lablibs/libm324_NAME.so  (use with M324_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/lablibs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form "$@" \
    -c ${ATTN_SRC:-motion324_amd/csrc/attention.hip} -o tools/lablibs/attn_$name.o
b=motion324_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lablibs/libm324_$name.so tools/lablibs/attn_$name.o \
    $b/runtime.o $b/gemm.o $b/gemm_ring4.o $b/attention_pwg.o $b/elementwise.o $b/backward.o $b/comm.o -ldl
rm -f tools/lablibs/attn_$name.o
echo tools/lablibs/libm324_$name.so
