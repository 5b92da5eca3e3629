#!/usr/bin/env python3
"""Interleaved A/B of library switches on the WHOLE c2 clip: one process, one model, one hipGraph per setting (the switches are
read at capture time), replays alternated round by round on the same box.
  usage: tools/clip_ab.py M324_ATTN_PWG=1,0 [M324_QKV_RING=0,1 ...] [--rounds 5] [--steps 20] [--frames 32]
Every combination of the listed values is one arm."""
import argparse, itertools, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import motion324_amd as m
from motion324_amd import lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("switches", nargs="+")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--frames", type=int, default=32)
args = ap.parse_args()
names, values = [], []
for sw in args.switches:
    k, v = sw.split("=")
    names.append(k)
    values.append([int(x.lstrip("vV")) for x in v.split(",")])
dev = torch.device("cuda")
model, _ = bench.build_model(dev, args.frames)
m.set_precision("bf16")
s = synth.synth_inputs(1, args.frames, 2048, 4096, 512, seed=1)
sample = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
HOST = {"M324_ATTN_BOUNDED": ("motion324_amd.transformer", "ATTN_BOUNDED"), "M324_FOLD_LN": ("motion324_amd.transformer", "FOLD_LN"),
        "M324_FOLD_MERGE": ("motion324_amd.transformer", "FOLD_MERGE"),
        "M324_BF16_DECODER": ("motion324_amd.Pcd_motion", "BF16_DECODER_STREAM"), "M324_FUSE_HEAD": ("motion324_amd.Pcd_motion", "FUSE_HEAD_N3"),
        "M324_OVERLAP": ("motion324_amd.Pcd_motion", "OVERLAP_SHAPE_ENCODER"), "M324_HOIST_Q": ("motion324_amd.Pcd_motion", "HOIST_DECODER_Q")}


BOTH = {"M324_HP": ("motion324_amd.transformer", "HP")}          # library switches the host mirrors


def set_switch(k, v):
    if k in BOTH and v is not None:
        import importlib
        mod, attr = BOTH[k]
        setattr(importlib.import_module(mod), attr, v)
    if k in HOST:                      # host switches are module constants read at capture time
        import importlib
        mod, attr = HOST[k]
        setattr(importlib.import_module(mod), attr, v)
    elif v is None:
        lib.set_tunable(k)
    else:
        lib.set_tunable(k, v)


arms = {}
for combo in itertools.product(*values):
    for k, v in zip(names, combo):
        set_switch(k, v)
    fast = m.GraphedForward(model)
    with torch.no_grad():
        clip = fast.static_inputs(sample)
        for _ in range(3):
            out = fast(clip).pcd_moved.clone()
    arms[combo] = (fast, clip, out)
for k in names:
    if k not in HOST:
        set_switch(k, None)
base = arms[next(iter(arms))][2]
res = {c: [] for c in arms}
for rnd in range(args.rounds):
    for combo, (fast, clip, _) in arms.items():
        with torch.no_grad():
            fast(clip)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                fast(clip)
            e1.record()
            torch.cuda.synchronize()
        res[combo].append(e0.elapsed_time(e1) / args.steps)
for combo, ts in res.items():
    ts = sorted(ts)
    d = float((arms[combo][2].double() - base.double()).norm() / base.double().norm())
    print("  ".join(f"{k}={v}" for k, v in zip(names, combo)) + f": median {ts[len(ts) // 2]:.3f} ms  (min {ts[0]:.3f}, max {ts[-1]:.3f})  "
          f"output vs first arm {d:.2e}", flush=True)
