#!/bin/bash
out=gpurun_out/r5c
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "v14" > $out/tests_v14.txt 2>&1
tail -3 $out/tests_v14.txt
for shape in "trunk qkv" "trunk fc1" "dino qkv" "dino fc1" "dec fc1"; do
  timeout 300 python3 tools/microbench.py gemm --iters 40 --only "$shape" --ab M324_GEMM=v10,v13,v14 >> $out/microbench_ab.txt 2>&1
done
for shape in "trunk qkv" "trunk fc1" "dec fc1"; do
  M324_GEMM=v14 timeout 300 python3 tools/microbench.py gemm --iters 40 --only "$shape" --ab M324_PP_SKEW=-1,2,4,7,10,14 >> $out/microbench_skew.txt 2>&1
done
cat $out/microbench_ab.txt $out/microbench_skew.txt | grep -v amdgpu.ids
timeout 900 python3 tools/clip_ab.py M324_PP=0,1 --rounds 5 > $out/clip_ab.txt 2>&1
tail -8 $out/clip_ab.txt
