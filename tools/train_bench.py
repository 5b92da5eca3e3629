#!/usr/bin/env python3
"""Training-step benchmark (BASELINE config 3: dyscene.yaml shapes, synthetic data): forward + hand-written backward +
fused AdamW per step, one process per GPU (torchrun for N > 1: flat-gradient all-reduce over RCCL).
usage: tools/train_bench.py [--batch 8] [--steps 5] [--warmup 2] [--frames 12] [--points 4096]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--points", type=int, default=4096)
ap.add_argument("--hw", type=int, default=224)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()

world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
if world > 1:
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
import motion324_amd as m
from motion324_amd import synth, training
from motion324_amd.optim import FusedAdamW, backward_completion_order, cosine_with_warmup

cfg = synth.make_config(frames=args.frames)
model = m.Motion_Latent_Model(cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.Dims(frames=args.frames), seed=0).items()}, strict=False)
model = model.train().to(dev)
s = synth.synth_inputs(args.batch, args.frames, args.points, args.points, args.hw, seed=1 + rank, with_target=True)
sample = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
opt = FusedAdamW(model.named_parameters(), lr=4e-4, betas=(0.9, 0.95), weight_decay=0.05, grad_clip_norm=1.0, allowed_gradnorm_factor=1e9,
                 order=backward_completion_order(model))
m.set_precision(args.precision)


def step(i):
    loss, _, G = training.forward_backward(model, sample, sink=opt)          # gradients land in the optimizer's flat buffer (as bench.py runs it)
    opt.finish_reduce()
    info = opt.step(lr=cosine_with_warmup(i, 1000, 30000, 4e-4) or 4e-7)
    return float(loss), info


losses = []
for i in range(args.warmup):
    losses.append(step(i)[0])
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
t0 = time.perf_counter()
for i in range(args.steps):
    l, info = step(args.warmup + i)
    losses.append(l)
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
dt = (time.perf_counter() - t0) / args.steps
if args.profile and rank == 0:          # per-shape HIP-event table of one more step (GEMM / attention launches)
    from motion324_amd import timing
    with timing.Recorder() as rec:
        step(args.warmup + args.steps)
    torch.cuda.synchronize()
    rows = sorted(rec.by_tag().items(), key=lambda kv: -kv[1]["total_ms"])
    print(f"instrumented launches: {sum(v['total_ms'] for _, v in rows):.1f} ms of the step", file=sys.stderr)
    for (name, tag), v in rows[:45]:
        print(f"{name:15s} {tag:62s} x{v['launches']:4d} {v['total_ms']:8.3f} ms {v['total_ms'] / v['launches'] * 1e3:8.1f} us each "
              f"{v['flops'] / max(v['total_ms'], 1e-9) / 1e9:7.0f} TF/s", file=sys.stderr)
if rank == 0:
    print(json.dumps({"metric": "training samples/sec (dyscene.yaml shapes, synthetic)", "value": round(world * args.batch / dt, 2),
                      "unit": "samples/s", "n_gpus": world, "ms_per_step": round(dt * 1e3, 1), "batch_per_gpu": args.batch,
                      "frames": args.frames, "points": args.points, "dtype": args.precision, "losses": [round(x, 5) for x in losses],
                      "last_grad_norm": round(info["grad_norm"], 4), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
