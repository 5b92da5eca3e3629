#!/bin/bash
out=gpurun_out/r5hp/pmc2
rm -rf $out; mkdir -p $out
for nt in 128 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    M324_NT_MB=$nt bash tools/pmc.sh $out/nt${nt}_$c $c -- tools/microbench.py gemm --only "trunk fc1 gelu" --iters 4 > $out/nt${nt}_$c.log 2>&1
  done
  f=$(find $out/nt${nt}_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/nt${nt}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  echo "== M324_NT_MB=$nt (trunk fc1 only)"; python3 tools/pmc_traffic.py "$f" "$w" 2>&1 | grep gemm
  M324_NT_MB=$nt python3 tools/microbench.py gemm --only "trunk fc1 gelu" --iters 20 2>/dev/null | tail -1
done | tee $out/summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc.sh $out/dino_$c $c -- tools/microbench.py gemm --only "dino fc1 gelu" --iters 4 > $out/dino_$c.log 2>&1
  bash tools/pmc.sh $out/dec_$c $c -- tools/microbench.py gemm --only "dec fc1 gelu" --iters 4 > $out/dec_$c.log 2>&1
  bash tools/pmc.sh $out/qkv_$c $c -- tools/microbench.py gemm --only "trunk qkv" --iters 4 > $out/qkv_$c.log 2>&1
done
for k in dino dec qkv; do
  f=$(find $out/${k}_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/${k}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  echo "== $k"; python3 tools/pmc_traffic.py "$f" "$w" 2>&1 | grep "gemm"
done | tee -a $out/summary.txt
find $out -name "*.csv" -size +1M -delete
