#!/bin/bash
out=gpurun_out/r5h
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pp in 0 1 3; do
  M324_PP=$pp setsid timeout -s KILL 300 python3 tools/train_bench.py --batch 8 --steps 6 --warmup 2 2>/dev/null | tail -1 | sed "s/^/PP=$pp /" >> $out/train_pp_ab.txt
done
cat $out/train_pp_ab.txt
for w in 0 8 2; do
  M324_BENCH_COLLECT=1 M324_KV_REHEARSE=$w setsid timeout -s KILL 400 python3 bench.py --mode frame-parallel --frames 256 --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print('REHEARSE=$w', j['ms_per_step'], 'ms per 256-frame clip;', j['launch'], '; finite', j['finite'])" >> $out/fp_rehearsal.txt
done
cat $out/fp_rehearsal.txt
