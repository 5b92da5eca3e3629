#!/bin/bash
# round 5: fabric-side bytes of single GEMM shapes (FETCH_SIZE / WRITE_SIZE in separate PMC passes), per XCD tile order
out=gpurun_out/r5hp/pmc
rm -rf $out; mkdir -p $out
for xcd in 3 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    M324_XCD=$xcd bash tools/pmc.sh $out/x${xcd}_$c $c -- tools/microbench.py gemm --only "fc1 gelu" --iters 4 > $out/x${xcd}_$c.log 2>&1
  done
  f=$(find $out/x${xcd}_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/x${xcd}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  echo "== M324_XCD=$xcd"; python3 tools/pmc_traffic.py "$f" "$w" 2>&1 | tail -12
done | tee $out/summary.txt
find $out -name "*.csv" -size +1M -delete
