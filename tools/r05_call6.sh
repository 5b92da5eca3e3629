#!/bin/bash
out=gpurun_out/r5i
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for shape in "trunk qkv" "trunk fc+res" "trunk fc1" "dino fc1" "dec fc+res" "dec fc1" "dec fc bf16"; do
  setsid timeout -s KILL 200 python3 tools/microbench.py gemm --iters 40 --only "$shape" --ab M324_XCD=3,11 2>&1 | grep -v amdgpu >> $out/refetch_ab.txt
done
cat $out/refetch_ab.txt
setsid timeout -s KILL 600 python3 tools/clip_ab.py M324_XCD=3,11 --rounds 5 2>&1 | grep -v amdgpu > $out/clip_refetch_ab.txt
tail -4 $out/clip_refetch_ab.txt
