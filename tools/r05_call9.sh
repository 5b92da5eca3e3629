#!/bin/bash
out=gpurun_out/r5k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
setsid timeout -s KILL 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_backward_gpu.py -x -q -m gpu -k "tn or wgrad or weight_grad or linear_bwd or backward" > $out/tests_tn.txt 2>&1; tail -3 $out/tests_tn.txt
setsid timeout -s KILL 300 python3 tools/train_bench.py --batch 8 --steps 6 --warmup 2 --profile > $out/train_after_tn.json 2> $out/train_after_tn_shapes.txt
cat $out/train_after_tn.json; grep "TN " $out/train_after_tn_shapes.txt | head -12
