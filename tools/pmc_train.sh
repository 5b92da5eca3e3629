#!/bin/bash
# HBM traffic per kernel of the training step (two rocprofv3 PMC passes over tools/train_bench.py, FETCH_SIZE / WRITE_SIZE; the
# corrections of tools/pmc_traffic.py).  usage (inside gpurun): bash tools/pmc_train.sh [tag] -> gpurun_out/<tag>/train_traffic.md
out=gpurun_out/${1:-pmc_train}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc --output-format csv -- python3 tools/train_bench.py --batch 8 --steps 1 --warmup 1 > /dev/null 2> $out/pmc_fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o pmc --output-format csv -- python3 tools/train_bench.py --batch 8 --steps 1 --warmup 1 > /dev/null 2> $out/pmc_write.err
f=$(find $out/pmc_fetch -name "*counter_collection.csv" | head -1)
w=$(find $out/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$f" "$w" --by-grid --top 60 > $out/train_traffic.md 2> $out/train_traffic.err
head -1 "$f" > $out/csv_header.txt
rm -rf $out/pmc_fetch $out/pmc_write
cat $out/train_traffic.md
