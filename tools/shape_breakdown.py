#!/usr/bin/env python3
"""Per-shape time table of one eager c2 forward (HIP events around every m324_gemm / m324_attention call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import motion324_amd as m
from motion324_amd import synth, timing
from motion324_amd.Pcd_motion import Motion_Latent_Model

cfg = synth.make_config(frames=32)
model = Motion_Latent_Model(cfg).cuda().eval()
s = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_inputs(1, 32, 2048, 4096, 512, seed=1).items()}
m.set_precision(os.environ.get("PREC", "bf16"))
with torch.no_grad():
    for _ in range(2):
        model(s)
    torch.cuda.synchronize()
    reps = 3
    with timing.Recorder() as rec:
        for _ in range(reps):
            model(s)
    torch.cuda.synchronize()
rows = sorted(rec.by_tag().items(), key=lambda kv: -kv[1]["total_ms"])
tot = sum(v["total_ms"] for _, v in rows) / reps
print(f"instrumented kernels: {tot:.3f} ms per clip")
for (name, tag), v in rows[:40]:
    ms = v["total_ms"] / reps
    print(f"{name:15s} {tag:60s} x{v['launches'] // reps:3d}  {ms:7.3f} ms  {ms / (v['launches'] / reps) * 1e3:7.1f} us each  "
          f"{v['flops'] / v['total_ms'] / 1e9:7.0f} TF/s")
