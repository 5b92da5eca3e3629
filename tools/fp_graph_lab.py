#!/usr/bin/env python3
"""Can forward_frame_parallel -- with its RCCL collectives -- be captured into a hipGraph?  1-rank "nccl" group on one GPU
(parallel.ALWAYS_COLLECT runs the collectives anyway).  Prints eager vs replay timing and equality.
usage: tools/fp_graph_lab.py [frames]"""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import motion324_amd as m
from motion324_amd import parallel, synth
import bench
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model, _ = bench.build_model(torch.device("cuda", 0), T)
s = synth.synth_inputs(1, T, 2048, 4096, 512, seed=1)
sample = {k: torch.from_numpy(v).cuda() for k, v in s.items()}
m.set_precision("bf16")
parallel.ALWAYS_COLLECT = True
with torch.no_grad():
    for _ in range(2):
        ref = model.forward_frame_parallel(sample).pcd_moved.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        model.forward_frame_parallel(sample)
    torch.cuda.synchronize()
    print(f"eager with collectives: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
    try:
        fast = m.GraphedForward(model, forward=model.forward_frame_parallel)
        out = fast(sample).pcd_moved
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fast(sample)
        torch.cuda.synchronize()
        print(f"graph replay with collectives: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms  equal={bool(torch.equal(out, ref))}")
    except Exception as e:
        print("capture failed:", type(e).__name__, str(e)[:500])
dist.destroy_process_group()
