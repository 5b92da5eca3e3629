#!/bin/bash
# round 5, GPU call 1: texture-path lab + GEMM microbench baseline + training-step profile
out=gpurun_out/r5a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 tools/ta_lab > $out/ta_lab_fc1.txt 2>&1
timeout 200 tools/ta_lab 10368 2304 > $out/ta_lab_qkv.txt 2>&1
timeout 600 python3 tools/microbench.py gemm --iters 30 > $out/microbench_gemm.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $out/trace -o kt -- python3 tools/train_bench.py --batch 8 --steps 3 --warmup 2 > $out/train_profiled.json 2> $out/train_trace.err
db=$(find $out/trace -name "*.db" | head -1)
python3 tools/rocpd_stats.py "$db" > $out/train_kernel_stats.md
python3 tools/rocpd_stats.py "$db" --by-grid > $out/train_kernel_stats_by_grid.md
rm -rf $out/trace
timeout 600 python3 tools/train_bench.py --batch 8 --steps 5 --warmup 2 --profile > $out/train_unprofiled.json 2> $out/train_shapes.txt
ls -la $out
