#!/bin/bash
out=gpurun_out/r5hp
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
setsid timeout -s KILL 1500 python3 -m pytest tests -x -q -m gpu --timeout 600 > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt
for hp in 0 2 0 2; do
  M324_HP=$hp setsid timeout -s KILL 200 python3 tools/train_bench.py --batch 8 --steps 6 --warmup 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/M324_HP=$hp /"
done | tee $out/train_hp2.txt
