#!/usr/bin/env python3
"""The training step's GEMM shapes under every schedule the chooser could pick (M324_GEMM forced per launch), interleaved rounds on one box:
is the chooser's pick the fastest at the decoder's 49152-row shapes (where 256 x 256 tiles fill 2.25 rounds) and at the trunk's 31104 rows?
usage: tools/train_gemm_ab.py [--rows 49152] [--iters 10]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import lib, ops
from motion324_amd.lib import ACT_GELU

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=49152)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--variants", default="0,2,10,11,12,13,15")
args = ap.parse_args()
dev, bf = "cuda", torch.bfloat16
M = args.rows
g = torch.Generator(device="cpu").manual_seed(0)


def rnd(*shape, dtype=bf, s=1.0):
    return (torch.randn(shape, generator=g) * s).to(dtype).to(dev)


a768, a3072 = rnd(M, 768), rnd(M, 3072)
w768, w3072x768, w768x3072 = rnd(768, 768, s=0.02), rnd(3072, 768, s=0.02), rnd(768, 3072, s=0.02)
x32 = rnd(M, 768, dtype=torch.float32)
o32, o16, o16b, o3072, o3072b = torch.empty((M, 768), dtype=torch.float32, device=dev), torch.empty((M, 768), dtype=bf, device=dev), torch.empty((M, 768), dtype=bf, device=dev), torch.empty((M, 3072), dtype=bf, device=dev), torch.empty((M, 3072), dtype=bf, device=dev)
z3072 = rnd(M, 3072)
bias768, bias3072 = rnd(768, dtype=torch.float32), rnd(3072, dtype=torch.float32)
CASES = {
    "K=768 N=768 fp32 out + residual (out-projection)": lambda: ops.gemm(a768, w768, o32, residual=x32),
    "K=768 N=768 bias + GELU + pre-activation (head fc1)": lambda: ops.gemm(a768, w768, o16, bias=bias768, act=ACT_GELU, preact_out=o16b),
    "K=768 N=768 plain bf16 (dgrad of the two above)": lambda: ops.gemm(a768, w768, o16),
    "K=3072 N=768 plain bf16 (dgrad of fc1)": lambda: ops.gemm(a3072, w768x3072, o16),
    "K=3072 N=768 fp32 out + residual (fc2)": lambda: ops.gemm(a3072, w768x3072, o32, residual=x32),
    "K=768 N=3072 GELU + pre-activation (fc1)": lambda: ops.gemm(a768, w3072x768, o3072, act=ACT_GELU, preact_out=o3072b),
    "K=768 N=3072 x gelu'(z) (dgrad of fc2)": lambda: ops.gemm(a768, w3072x768, o3072, gelu_grad_of=z3072),
}
variants = [int(v) for v in args.variants.split(",")]
for name, fn in CASES.items():
    res = {v: [] for v in variants}
    for rnd_ in range(5):
        for v in variants:
            lib.set_tunable("M324_GEMM", v)
            try:
                fn()
            except Exception as e:
                res[v].append(float("nan")); continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / args.iters * 1e3)
    lib.set_tunable("M324_GEMM")
    med = {v: sorted(t)[len(t) // 2] for v, t in res.items()}
    print(f"M={M} {name}: " + "  ".join(f"v{v}: {m:.1f}" for v, m in med.items()) + "  us", flush=True)
