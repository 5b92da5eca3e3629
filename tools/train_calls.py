#!/usr/bin/env python3
"""Which call sites issue the small launches of a training step: counts every motion324_amd.ops function by (function, caller file:line)
over ONE c3 step after warm-up, plus torch's own copy / cat / elementwise operators (torch.profiler, CPU side only).
usage: tools/train_calls.py [--batch 8] [--frames 12] [--min 4]"""
import argparse, collections, inspect, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--points", type=int, default=4096)
ap.add_argument("--min", type=int, default=4)
args = ap.parse_args()
import motion324_amd as m
from motion324_amd import ops, synth, training
from motion324_amd.optim import FusedAdamW, backward_completion_order

dev = torch.device("cuda")
cfg = synth.make_config(frames=args.frames)
model = m.Motion_Latent_Model(cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.Dims(frames=args.frames), seed=0).items()}, strict=False)
model = model.train().to(dev)
sample = {k: torch.from_numpy(v).to(dev) for k, v in synth.synth_inputs(args.batch, args.frames, args.points, args.points, 224, seed=1, with_target=True).items()}
opt = FusedAdamW(model.named_parameters(), lr=4e-4, betas=(0.9, 0.95), weight_decay=0.05, allowed_gradnorm_factor=1e9, order=backward_completion_order(model))
m.set_precision("bf16")


def step():
    training.forward_backward(model, sample, sink=opt)
    opt.finish_reduce()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
counts = collections.Counter()
OPS = os.path.abspath(ops.__file__)


def wrap(name, fn):
    def inner(*a, **k):
        f = inspect.currentframe().f_back
        while f is not None and os.path.abspath(f.f_code.co_filename) == OPS:
            f = f.f_back
        counts[(name, f"{os.path.basename(f.f_code.co_filename)}:{f.f_lineno}" if f else "?")] += 1
        return fn(*a, **k)
    return inner


for name in ("colsum", "transpose", "cast", "gemm", "gemm_tn", "layernorm", "layernorm_bwd", "qkv_split", "qkv_split_bwd", "attention", "gemm_splitk"):
    if hasattr(ops, name):
        setattr(ops, name, wrap(name, getattr(ops, name)))
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
    step()
torch.cuda.synchronize()
by_fn = collections.Counter()
for (name, site), n in counts.items():
    by_fn[name] += n
print("ops calls per step:", dict(by_fn))
for (name, site), n in sorted(counts.items(), key=lambda e: -e[1]):
    if n >= args.min and name in ("colsum", "transpose", "cast"):
        print(f"  {name:10s} {site:28s} {n}")
print("torch operators per step (>= --min):")
for e in sorted(prof.key_averages(), key=lambda e: -e.count):
    if e.count >= args.min and e.key.startswith("aten::") and any(t in e.key for t in ("copy", "cat", "add", "mul", "zero", "fill", "clone", "to", "contiguous", "empty")):
        print(f"  {e.key:32s} {e.count}")
