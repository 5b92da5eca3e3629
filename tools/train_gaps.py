#!/usr/bin/env python3
"""Where is the GPU idle inside ONE training step, and what was the host doing meanwhile?  torch.profiler around a step (every
motion324_amd.ops / backward / training function wrapped in a record_function); prints the idle gaps above a threshold with the
innermost host range that covers the gap's start.  usage: tools/train_gaps.py [--batch 8] [--min-us 40]"""
import argparse, functools, os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--min-us", type=float, default=40.0)
args = ap.parse_args()
import motion324_amd as m
from motion324_amd import synth, training, ops, backward, optim, prepared
from motion324_amd.optim import FusedAdamW, backward_completion_order

dev = torch.device("cuda", 0)
cfg = synth.make_config(frames=12)
model = m.Motion_Latent_Model(cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.Dims(frames=12), seed=0).items()}, strict=False)
model = model.train().to(dev)
s = synth.synth_inputs(args.batch, 12, 4096, 4096, 224, seed=1, with_target=True)
sample = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
opt = FusedAdamW(model.named_parameters(), lr=4e-4, betas=(0.9, 0.95), weight_decay=0.05, grad_clip_norm=1.0, allowed_gradnorm_factor=1e9,
                 order=backward_completion_order(model))
m.set_precision("bf16")


def wrap(mod, prefix):
    for name, fn in list(vars(mod).items()):
        if isinstance(fn, types.FunctionType) and not name.startswith("__"):
            def make(f, label):
                @functools.wraps(f)
                def inner(*a, **k):
                    with torch.profiler.record_function(label):
                        return f(*a, **k)
                return inner
            setattr(mod, name, make(fn, f"{prefix}.{name}"))


def step():
    loss, _, G = training.forward_backward(model, sample, sink=opt)
    opt.finish_reduce()
    info = opt.step(lr=4e-4)
    return float(loss)


for _ in range(2):
    step()
# coarse ranges only (a block's backward, a phase of the step): wrapping every ops.* call makes the step host-bound
for mod, p, names in ((backward, "bw", ("self_attn_block_bwd", "cross_attn_block_bwd", "self_attn_block_internals", "cross_attn_block_internals")),
                      (training, "tr", ("_point_features_train", "_point_features_bwd", "_store_budget"))):
    for name in names:
        fn = getattr(mod, name)
        def make(f, label):
            @functools.wraps(f)
            def inner(*a, **k):
                with torch.profiler.record_function(label):
                    return f(*a, **k)
            return inner
        setattr(mod, name, make(fn, f"{p}.{name}"))
for cls, p, names in ((FusedAdamW, "opt", ("step", "finish_reduce", "begin_step")), (backward.GradStore, "G", ("done",)),
                      (type(model.image_encoder), "dino", ("run",))):
    for name in names:
        fn = getattr(cls, name)
        def make(f, label):
            @functools.wraps(f)
            def inner(*a, **k):
                with torch.profiler.record_function(label):
                    return f(*a, **k)
            return inner
        setattr(cls, name, make(fn, f"{p}.{name}"))
step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
gpu = sorted([(e.time_range.start, e.time_range.end, e.name) for e in ev if str(e.device_type).endswith("CUDA") and e.time_range.end > e.time_range.start])
cpu = sorted([(e.time_range.start, e.time_range.end, e.name) for e in ev if str(e.device_type).endswith("CPU") and "." in e.name and e.name.split(".")[0] in ("ops", "bw", "tr", "P", "G", "opt", "dino")])
t0, t1 = gpu[0][0], gpu[-1][1]
print(f"step: {len(gpu)} GPU kernels over {(t1 - t0) / 1e3:.2f} ms; {len(cpu)} host ranges")
cur = gpu[0][1]
gaps = []
for s_, e_, n_ in gpu[1:]:
    if s_ - cur >= args.min_us:
        gaps.append((cur, s_, n_))
    cur = max(cur, e_)
print(f"idle gaps >= {args.min_us} us: {len(gaps)}, {sum(b - a for a, b, _ in gaps) / 1e3:.2f} ms in total")
for a, b, nxt in gaps:
    cover = [c for c in cpu if c[0] <= a <= c[1]]
    inner = " > ".join(c[2] for c in cover[-3:]) if cover else "(no wrapped host range: python between calls)"
    during = [c[2] for c in cpu if a <= c[0] <= b][:4]
    print(f"  at {(a - t0) / 1e3:7.2f} ms  idle {b - a:7.1f} us  host: {inner}   started during the gap: {during}   next kernel: {nxt[:50]}")

import collections
pos = collections.Counter()
cur = gpu[0][1]
for s_, e_, n_ in gpu[1:]:
    if s_ > cur:
        pos[int((cur - t0) / (t1 - t0) * 40)] += (s_ - cur)
    cur = max(cur, e_)
print("idle us per 1/40 of the step:", " ".join(f"{pos.get(i, 0):.0f}" for i in range(40)))
# host ranges on the timeline
for c in cpu:
    if c[2].startswith(("tr.", "opt.", "dino.")) or c[2] == "G.done":
        print(f"   host {c[2]:32s} {(c[0] - t0) / 1e3:8.2f} .. {(c[1] - t0) / 1e3:8.2f} ms")
