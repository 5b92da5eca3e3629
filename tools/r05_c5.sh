#!/bin/bash
out=gpurun_out/r5hp; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for hp in 6 2 6 2; do
  echo "M324_HP=$hp"; M324_HP=$hp setsid timeout -s KILL 300 python3 tools/long_clip.py --iters 4 2>/dev/null | tail -3
done | tee $out/c5_hp.txt
