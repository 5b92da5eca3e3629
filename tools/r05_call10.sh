#!/bin/bash
out=gpurun_out/r5l
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
setsid timeout -s KILL 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $out/tests_gemm.txt 2>&1; tail -2 $out/tests_gemm.txt
for rep in 1 2 3; do
for lib in tools/lablibs/libm324_prev.so motion324_amd/libm324.so; do
  M324_LIB=$lib setsid timeout -s KILL 200 python3 tools/microbench.py gemm --iters 40 --only "fc1" 2>&1 | grep "^gemm" | sed "s|^|$lib |" >> $out/stage0_ab.txt
  M324_LIB=$lib setsid timeout -s KILL 200 python3 tools/microbench.py gemm --iters 40 --only "dec fc" 2>&1 | grep "^gemm" | sed "s|^|$lib |" >> $out/stage0_ab.txt
done
done
sort $out/stage0_ab.txt | awk '{print $1, $3, $4, $5, $(NF-3), $(NF-2)}'
