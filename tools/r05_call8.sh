#!/bin/bash
out=gpurun_out/r5k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $out/pmc -o pmc --output-format csv -- python3 tools/train_bench.py --batch 8 --steps 1 --warmup 1 > $out/train_pmc.log 2>&1
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
python3 tools/pmc_digest.py "$f" > $out/train_pmc_digest.txt 2>&1
rm -rf $out/pmc
grep -A9 "gemm_tn_pipe_kernel\|gemm_ring_kernel<unsigned short, 0, 0>\|gemm_ring_kernel<unsigned short, 3, 0>" $out/train_pmc_digest.txt | head -120
