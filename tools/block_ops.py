#!/usr/bin/env python3
"""Which kernels does one decoder block launch (torch.profiler, eager)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import motion324_amd as m
from motion324_amd import synth
from motion324_amd.prepared import Prepared, compute_dtype
dev = torch.device("cuda", 0)
w = bench.WORKLOAD
model, _ = bench.build_model(dev, w["frames"])
s = synth.synth_inputs(w["B"], w["T"], w["N"], w["S"], w["HW"], seed=1)
sample = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
m.set_precision("bf16")
P = Prepared.for_module(model, dev, compute_dtype())
C, K = model.embed_dim, model.num_learnable_tokens
B, T = 1, w["T"]
Lt = 4 + K + 256
tok = torch.randn(B * T * Lt, C, device=dev)
dec = model.decoder_cross_attn
with torch.no_grad():
    pf = model._point_features(P, sample["ref_pcd"][0].float().contiguous(), sample["ref_normal"][0].float().contiguous(), sample["ref_rgb"][0].float().contiguous())
    def block():
        Kd, Vd = dec.project_kv(P, tok, B * T, K, row_map=(K, Lt, 4))
        return model.decoder_block(P, Kd[:T], Vd[:T], pf)
    for _ in range(2):
        block()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        block()
        torch.cuda.synchronize()
    for e in prof.key_averages():
        if e.device_time_total > 0 or "copy" in e.key.lower() or "to" == e.key:
            print(f"{e.key[:90]:90s} n={e.count} dev_us={e.device_time_total:.1f}")
