#!/bin/bash
out=gpurun_out/r5i
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
setsid timeout -s KILL 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $out/tests_gemm.txt 2>&1; tail -3 $out/tests_gemm.txt
for shape in "trunk fc2" "dino fc2" "dec fc2+res"; do
  setsid timeout -s KILL 200 python3 tools/microbench.py gemm --iters 40 --only "$shape" --ab M324_XCD=3,11 2>&1 | grep -v amdgpu >> $out/refetch_ab2.txt
done
cat $out/refetch_ab2.txt
setsid timeout -s KILL 600 python3 tools/clip_ab.py M324_XCD=3,11 --rounds 5 2>&1 | grep -v amdgpu > $out/clip_refetch_ab2.txt
tail -3 $out/clip_refetch_ab2.txt
