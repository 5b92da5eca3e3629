#!/usr/bin/env python3
"""The decoder cross-attention block of the c2 clip on its own: hipGraph replay timing (as bench.py reports it) and, under
`rocprofv3 --kernel-trace --stats`, the per-kernel durations inside the replays.
usage: tools/block_lab.py [replays] [burn] [pair|branch|serial]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import motion324_amd as m
from motion324_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
w = bench.WORKLOAD
model, _ = bench.build_model(dev, w["frames"])
s = synth.synth_inputs(w["B"], w["T"], w["N"], w["S"], w["HW"], seed=1)
sample = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
m.set_precision("bf16")
q_mode = next((a_ for a_ in sys.argv[2:] if a_ in ("pair", "branch", "serial")), "pair")
for _ in range(3):
    r = bench.decoder_block_replay(model, sample, n, q_mode=q_mode)
    print(f"{q_mode}:", {k: v for k, v in r.items() if k != "timing"}, flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "burn":
    # does the block slow down behind a few seconds of clip replays (clock / power state) or only inside bench.py's process state?
    fast = m.GraphedForward(model)
    with torch.no_grad():
        clip = fast.static_inputs(sample)
        import time
        t0 = time.time()
        while time.time() - t0 < 4.0:
            for _ in range(32):
                fast(clip)
            torch.cuda.synchronize()
    for _ in range(3):
        r = bench.decoder_block_replay(model, sample, n)
        print("after 4 s of clip replays:", {k: v for k, v in r.items() if k != "timing"}, flush=True)
