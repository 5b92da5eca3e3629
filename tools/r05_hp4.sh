#!/bin/bash
out=gpurun_out/r5hp; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
setsid timeout -s KILL 500 python3 tools/clip_ab.py M324_HP=6,2,7 --rounds 6 > $out/clip_hp6.txt 2>&1; tail -4 $out/clip_hp6.txt
setsid timeout -s KILL 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "v15" --timeout 300 2>&1 | tail -2
setsid timeout -s KILL 400 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "golden or fold or graph" --timeout 300 2>&1 | tail -2
