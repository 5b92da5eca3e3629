#!/bin/bash
out=gpurun_out/r5b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 tools/ta_lab > $out/ta_lab_fc1.txt 2>&1
timeout 300 tools/ta_lab 10368 2304 > $out/ta_lab_qkv.txt 2>&1
timeout 1500 python3 -m pytest tests/test_trained_like_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "trained_like or scores_bounded or out_of_memory" -s > $out/new_tests.txt 2>&1
tail -5 $out/new_tests.txt
