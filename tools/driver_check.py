#!/usr/bin/env python3
"""Full-size check of the long-video driver (motion324_amd/inference.py): are the pipelined / reuse / byte-frame forms equal to the
plain loop bit for bit at the BASELINE clip's shapes, and if not, where do they part?  (tests/test_configs_gpu.py checks a small model.)
usage: tools/driver_check.py [T]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import motion324_amd as m
from motion324_amd import synth
from motion324_amd.inference import run_model_inference
from motion324_amd.prepared import Prepared

T = int(sys.argv[1]) if len(sys.argv) > 1 else 94
dev = torch.device("cuda", 0)
w = bench.WORKLOAD
model, _ = bench.build_model(dev, w["frames"])
s = synth.synth_inputs(1, 4, w["N"], w["S"], 64, seed=1)
inp = {k: torch.from_numpy(v).to(dev) for k, v in s.items() if k != "rgb_video"}
g = torch.Generator().manual_seed(11)
vid8 = torch.randint(0, 256, (T, w["HW"], w["HW"], 3), generator=g, dtype=torch.uint8).pin_memory()
vidf = (vid8.float() / 255.0).pin_memory()
cfg = {"training": {"frames": w["T"], "use_amp": True}}
m.set_precision("bf16")


def cmp(name, a, b):
    d = (a.double() - b.double()).abs()
    print(f"{name}: equal={torch.equal(a, b)} max|d|={float(d.max()):.3e} differing={int((d > 0).sum())}/{d.numel()}", flush=True)


with torch.no_grad():
    # 1. the image encoder alone: frames 1..31 as a batch of 31 against the same frames inside a batch of 32
    P = Prepared.for_module(model, dev, torch.bfloat16)
    fr = vidf[:32].to(dev)
    x32 = model.image_encoder.run(P, fr).clone()
    x31 = model.image_encoder.run(P, fr[1:].contiguous()).clone()
    cmp("DINO tokens, 31-frame batch vs rows of the 32-frame batch", x31, x32[257:])
    x16 = model.image_encoder.run(P, fr[:16].contiguous()).clone()
    cmp("DINO tokens, 16-frame batch vs rows of the 32-frame batch", x16, x32[:16 * 257])
    x15 = model.image_encoder.run(P, fr[17:].contiguous()).clone()
    cmp("DINO tokens, 15-frame batch vs rows of the 32-frame batch", x15, x32[17 * 257:])
    model.auto_graph = False
    plain = run_model_inference(model, inp, vidf, cfg, dev, pipelined=False)
    cmp("eager: pipelined, no reuse vs plain", run_model_inference(model, inp, vidf, cfg, dev, reuse=False), plain)
    cmp("eager: pipelined + reuse vs plain", run_model_inference(model, inp, vidf, cfg, dev), plain)
    cmp("eager: bytes, plain loop vs plain", run_model_inference(model, inp, vid8, cfg, dev, pipelined=False), plain)
    model.auto_graph = True
    for k in range(3):
        cmp(f"graph pass {k}: plain loop vs eager plain", run_model_inference(model, inp, vidf, cfg, dev, pipelined=False), plain)
    model._drop_auto_graph()
    for k in range(3):
        cmp(f"graph pass {k}: pipelined + reuse vs eager plain", run_model_inference(model, inp, vidf, cfg, dev), plain)
    model._drop_auto_graph()
    for k in range(3):
        cmp(f"graph pass {k}: pipelined bytes vs eager plain", run_model_inference(model, inp, vid8, cfg, dev), plain)
