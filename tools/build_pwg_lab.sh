#!/bin/bash
# Lab build of the one-wave-per-SIMD attention's timing-only ablations (gen_attn_pwg.py --lab writes attn_pwg_lab1..9.inc next to
# the product stream): tools/build_pwg_lab.sh -> tools/lablibs/libm324_pwglab.so = the product objects + tools/lab_src/attention_pwg_lab.hip
# (entry point m324_lab_attn_pwg).   M324_LIB=tools/lablibs/libm324_pwglab.so tools/pwg_check.py --ablate --trace
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/lablibs
(cd motion324_amd/csrc && python3 gen_attn_pwg.py --lab)
python3 -m motion324_amd.build >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize "$@" \
    -c tools/lab_src/attention_pwg_lab.hip -o tools/lablibs/attention_pwg_lab.o
b=motion324_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lablibs/libm324_pwglab.so tools/lablibs/attention_pwg_lab.o \
    $b/runtime.o $b/gemm.o $b/gemm_ring4.o $b/attention.o $b/attention_pwg.o $b/elementwise.o $b/backward.o $b/comm.o -ldl
rm -f tools/lablibs/attention_pwg_lab.o
echo tools/lablibs/libm324_pwglab.so
