#!/usr/bin/env python3
"""Interleaved A/B of GEMM schedules on the IN-CLIP forms of the wide projections (bf16 inference): fused head-major q|k|v epilogue
and fc1 + GELU, both as LayerNorm-fold consumers that merge the producer's unmerged statistics table themselves.
usage: tools/fold_ab.py [v13,v15,...] [--rounds 6] [--iters 20]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import lib, ops
from motion324_amd.lib import ACT_GELU

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="?", default="v10,v13,v15")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
variants = args.variants.split(",")
dev, dt = "cuda", torch.bfloat16


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(name, M, N, K, L, kind):
    a = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    bias = torch.randn(N, device=dev) * 0.1
    part = torch.rand(K // 64, M, 2, device=dev) + 0.5          # (sum, M2) per 64-column block
    colsum = w.float().sum(1).contiguous()
    ln = (part, colsum, 1e-5)
    if kind == "qkv":
        H, B = 12, M // L
        Q, Kk = (torch.empty((B, H, L, 64), dtype=dt, device=dev) for _ in range(2))
        long_seq = L >= 2048
        V = torch.empty((B, H, 64, L) if long_seq else (B, H, L, 64), dtype=dt, device=dev)
        qw = torch.ones(64, device=dev)
        fn = lambda: ops.gemm(a, w, None, bias=bias, qkv_heads=(Q, Kk, V, qw, qw, 1e-5, ops.Q_PRESCALE, L, H), ln=ln)
    else:
        out = torch.empty(M, N, dtype=dt, device=dev)
        fn = lambda: ops.gemm(a, w, out, bias=bias, act=ACT_GELU, ln=ln)
    res = {v: [] for v in variants}
    for _ in range(args.rounds):
        for v in variants:
            lib.set_tunable("M324_GEMM", int(v.lstrip("v")))
            res[v].append(timeit(fn, args.iters))
    lib.set_tunable("M324_GEMM")
    med = {v: sorted(t)[len(t) // 2] for v, t in res.items()}
    print(f"{name:34s} M={M:6d} N={N:5d} K={K:5d}  " + "  ".join(f"{v}: {m:6.1f} us" for v, m in med.items()), flush=True)


case("trunk global q|k|v (Vt, fold)", 10368, 2304, 768, 10368, "qkv")
case("trunk per-frame q|k|v (fold)", 10368, 2304, 768, 324, "qkv")
case("dino q|k|v (fold)", 8224, 2304, 768, 257, "qkv")
case("dino half-batch q|k|v (fold)", 4112, 2304, 768, 257, "qkv")
case("trunk fc1 + GELU (fold)", 10368, 3072, 768, 0, "fc1")
case("dino fc1 + GELU (fold)", 8224, 3072, 768, 0, "fc1")
case("dino half-batch fc1 + GELU (fold)", 4112, 3072, 768, 0, "fc1")
case("decoder fc1 + GELU (fold)", 65536, 3072, 768, 0, "fc1")
