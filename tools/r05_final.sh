#!/bin/bash
# round 5, final GPU pass: full GPU test suite, default bench line, per-kernel profile (stats + traffic + utilisation), training profile
out=gpurun_out/r5z
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
setsid timeout -s KILL 1500 python3 -m pytest tests -x -q -m gpu --timeout 600 --durations=25 > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt
setsid timeout -s KILL 900 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; head -c 400 $out/bench_default.json; echo
bash tools/profile_round.sh r05_p3 > $out/profile_round.log 2>&1; tail -2 $out/profile_round.log
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o kt -- python3 tools/train_bench.py --batch 8 --steps 3 --warmup 2 > $out/train_profiled.json 2> $out/train_trace.err
db=$(find $out/trace -name "*.db" | head -1)
python3 tools/rocpd_stats.py "$db" > $out/train_kernel_stats.md
rm -rf $out/trace
setsid timeout -s KILL 300 python3 tools/train_bench.py --batch 8 --steps 6 --warmup 2 --profile > $out/train_unprofiled.json 2> $out/train_shapes.txt
cat $out/train_unprofiled.json
