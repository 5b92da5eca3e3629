#!/bin/bash
out=gpurun_out/r5h
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f $out/fp_rehearsal.txt
for rep in 1 2; do
for w in 0 8 2; do
  M324_BENCH_COLLECT=1 M324_KV_REHEARSE=$w setsid timeout -s KILL 400 python3 bench.py --mode frame-parallel --frames 256 --steps 5 --warmup 2 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
att=[r for r in j['roofline']['by_symbol'] if 'attn_pwg' in r['symbol']]
print('M324_KV_REHEARSE=$w', j['ms_per_step'], 'ms per 256-frame clip (graph chain);', 'global attention', att[0]['launches_per_step'], 'launches,', att[0]['ms_per_step'], 'ms, shapes', att[0]['shapes'], '; finite', j['finite'])" >> $out/fp_rehearsal.txt
done
done
cat $out/fp_rehearsal.txt
