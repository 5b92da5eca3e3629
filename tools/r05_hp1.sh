#!/bin/bash
# round 5: schedule v15 -- values against v10, the new kernel tests, the clip with and without it
out=gpurun_out/r5hp
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
setsid timeout -s KILL 300 python3 tools/hp_check.py > $out/check4.txt 2>&1; grep -E "v15:|ALL|FAILED|False" $out/check4.txt | tail -40
setsid timeout -s KILL 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "v15 or rowstat or fold" --timeout 300 > $out/tests_v15.txt 2>&1; tail -5 $out/tests_v15.txt
setsid timeout -s KILL 400 python3 tools/clip_ab.py M324_HP=1,0 --rounds 5 > $out/clip_ab.txt 2>&1; tail -8 $out/clip_ab.txt
