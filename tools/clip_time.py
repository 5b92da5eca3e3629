#!/usr/bin/env python3
"""The c2 clip as a hipGraph replay, five rounds of 20 replays: one line per process.  For A/Bs between LIBRARIES (M324_LIB=path of another
libm324.so build, e.g. the previous commit's) or load-time switches, run alternately in separate processes on one box.
usage: [M324_LIB=...] [M324_<SWITCH>=...] tools/clip_time.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
import motion324_amd as m
from motion324_amd import synth
dev = torch.device("cuda")
model, _ = bench.build_model(dev, 32)
m.set_precision("bf16")
s = synth.synth_inputs(1, 32, 2048, 4096, 512, seed=1)
sample = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
fast = m.GraphedForward(model)
with torch.no_grad():
    clip = fast.static_inputs(sample)
    for _ in range(5): fast(clip)
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fast(clip)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
print(os.environ.get("M324_LIB", "product"), "clip ms:", " ".join(f"{t:.3f}" for t in ts), flush=True)
