#!/bin/bash
# One call on the GPU box at the end of a round: the whole GPU suite, the smoke entry, the default bench line, a rocprofv3 kernel
# table of the training step and its un-profiled time (c3 and c4).
# usage (inside gpurun): bash tools/round_check.sh [tag]     -> gpurun_out/<tag>/{pytest_gpu.txt,smoke.txt,bench.json,train_kernel_stats.md,train_c3.json,train_c4.json}
out=gpurun_out/${1:-round_check}; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o kt -- python3 tools/train_bench.py --batch 8 --steps 3 --warmup 2 > $out/train_profiled.json 2> $out/trace.err
db=$(find $out/trace -name "*.db" | head -1)
python3 tools/rocpd_stats.py "$db" > $out/train_kernel_stats.md
rm -rf $out/trace
python tools/train_bench.py --steps 10 --warmup 3 2>&1 | tail -1 > $out/train_c3.json
python tools/train_bench.py --batch 32 --steps 5 --warmup 2 2>&1 | tail -1 > $out/train_c4.json
cut -c1-220 $out/train_c3.json $out/train_c4.json
