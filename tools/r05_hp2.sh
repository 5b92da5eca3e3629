#!/bin/bash
out=gpurun_out/r5hp
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
setsid timeout -s KILL 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "v15" --timeout 300 > $out/tests_v15b.txt 2>&1; tail -3 $out/tests_v15b.txt
for hp in 0 3 0 3; do
  M324_HP=$hp setsid timeout -s KILL 200 python3 tools/train_bench.py --batch 8 --steps 6 --warmup 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/M324_HP=$hp /"
done | tee $out/train_hp.txt
setsid timeout -s KILL 300 python3 tools/microbench.py gemm --only "trunk qkv" --ab M324_GEMM=v15,v14,v13,v10 2>&1 | tail -3 | tee $out/micro_qkv.txt
