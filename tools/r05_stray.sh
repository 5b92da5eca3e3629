#!/bin/bash
out=gpurun_out/r5hp; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
setsid timeout -s KILL 400 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" --timeout 300 > $out/tests_attn.txt 2>&1; tail -3 $out/tests_attn.txt
setsid timeout -s KILL 200 python3 tools/microbench.py attn --only dino --ab M324_ATTN_EXP=0,32 2>&1 | tail -6 | tee $out/micro_dino.txt
setsid timeout -s KILL 400 python3 tools/clip_ab.py M324_ATTN_EXP=0,32 --rounds 6 > $out/clip_stray.txt 2>&1; tail -3 $out/clip_stray.txt
