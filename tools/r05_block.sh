#!/bin/bash
out=gpurun_out/r5hp; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for hp in 2 3 2 3; do
  echo "M324_HP=$hp"; M324_HP=$hp setsid timeout -s KILL 200 python3 tools/block_lab.py 40 2>/dev/null | grep -o "'ms_per_step': [0-9.]*, \|'rounds_in_order_ms': \[[0-9., ]*\]" | paste - - 
done | tee $out/block_hp.txt
