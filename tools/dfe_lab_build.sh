#!/bin/bash
# Fast compile of only two v14 (deferred-epilogue GEMM) instantiations with the ISA kept: tools/dfe_lab_build.sh [flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DM324_DFE_LAB "$@" \
    -c motion324_amd/csrc/gemm_ring4.hip -o /tmp/dfe_lab.o -save-temps=obj 2>&1 | grep -E "error|warning" | head
python tools/kernel_regs.py /tmp/gemm_ring4-hip-amdgcn-amd-amdhsa-gfx950.s dfe
