#!/bin/bash
# One call on the GPU box: rocprofv3 kernel statistics of the bench command + the two PMC traffic passes.
# usage (inside gpurun): tools/profile_round.sh r02_p1     -> gpurun_out/<tag>/{kernel_stats.md,traffic.md,traffic.json,bench.json}
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --stats -d $out/trace -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --sustain 0 > $out/bench_profiled.json 2> $out/trace.err
db=$(find $out/trace -name "*.db" | head -1)
python3 tools/rocpd_stats.py "$db" > $out/kernel_stats.md
python3 tools/rocpd_stats.py "$db" --by-grid > $out/kernel_stats_by_grid.md
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc --output-format csv -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --sustain 0 > /dev/null 2> $out/pmc_fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o pmc --output-format csv -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --sustain 0 > /dev/null 2> $out/pmc_write.err
f=$(find $out/pmc_fetch -name "*counter_collection.csv" | head -1)
w=$(find $out/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$f" "$w" --json $out/traffic.json > $out/traffic.md
# matrix-pipe / VALU utilisation of every kernel (one more PMC pass: SQ + GRBM counters only)
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/pmc_sq -o pmc --output-format csv -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --sustain 0 > /dev/null 2> $out/pmc_sq.err
q=$(find $out/pmc_sq -name "*counter_collection.csv" | head -1)
python3 tools/pmc_util.py "$q" > $out/utilisation.md 2> $out/utilisation.err
rm -rf $out/pmc_sq
rm -rf $out/trace $out/pmc_fetch $out/pmc_write
ls -la $out
