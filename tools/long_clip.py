#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU: a 256-frame clip (82 944 trunk tokens, 81 % global attention) in a single forward.
Prints ms per clip and checks that the transposed-V projection epilogue and the m324_qkv_split path agree.
usage: python tools/long_clip.py [--frames 256] [--iters 3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import motion324_amd as m
from motion324_amd import synth, transformer
from motion324_amd.Pcd_motion import Motion_Latent_Model

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--iters", type=int, default=3)
args = ap.parse_args()
T = args.frames
cfg = synth.make_config(frames=T)
model = Motion_Latent_Model(cfg).cuda().eval()
s = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_inputs(1, T, 2048, 4096, 512, seed=1).items()}
m.set_precision("bf16")
outs = {}
with torch.no_grad():
    for fused in (True, False):
        transformer.FUSE_QKV_VT = fused
        model(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            out = model(s)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.iters * 1e3
        outs[fused] = out["pcd_moved"].float().clone()
        print(f"T={T}: {ms:8.2f} ms per clip = {T / ms * 1e3:7.1f} frames/s  ({'transposed-V epilogue' if fused else 'm324_qkv_split'}), "
              f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
d = (outs[True] - outs[False]).norm() / outs[False].norm()
print(f"relative difference between the two paths: {float(d):.3e}")
assert torch.isfinite(outs[True]).all() and float(d) < 2e-2
