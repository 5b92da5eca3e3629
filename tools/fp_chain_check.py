#!/usr/bin/env python3
"""Frame-parallel forward, eager vs the chain of hipGraphs, N gloo ranks on one device (tests/test_configs_gpu.py's worker at any case).
usage: tools/fp_chain_check.py CASE WORLD PRECISION OVERLAP   (e.g. tiny_resize 2 bf16 1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
case, world, precision, overlap = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
os.environ["M324_KV_OVERLAP"] = overlap
import faulthandler
faulthandler.dump_traceback_later(100, exit=True)
import torch
import torch.multiprocessing as mp
import test_configs_gpu as T

if __name__ == "__main__":
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(T._fp_worker, args=(world, 24567, case, precision, ret), nprocs=world, join=True)
    for r in range(world):
        print(case, world, precision, "overlap", overlap, "rank", r, "chain == eager:", ret[f"chain{r}"], ret[f"chain_diff{r}"], flush=True)
