#!/usr/bin/env python3
"""Do the trunk GEMMs run slower in the clip than in a loop because their operands are COLD there?  Times each shape with
(a) the same W / A every launch (what tools/gemm_lab and microbench.py do: W stays in the 4-MiB L2s), (b) a different W per
launch out of a set larger than the 256-MiB MALL, (c) different W and A, (d) like (b) but a 64-MiB scrub kernel between
launches is NOT used -- launches stay back to back, only the operands rotate."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import ops
from motion324_amd.lib import ACT_GELU
dev, dt = "cuda", torch.bfloat16
NW = 96          # 96 x 3.5 MB = 340 MB of q|k|v weights: more than the MALL keeps


def bench(fn, n, iters=3):
    best = 1e9
    for _ in range(iters):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def run(name, M, N, K, mode):
    Ws = [(torch.randn(N, K, device=dev) * 0.02).to(dt) for _ in range(NW)]
    As = [torch.randn(M, K, device=dev).to(dt) for _ in range(16)]
    qw = torch.ones(64, device=dev)
    if mode == "qkv":
        L, H = M, N // 192
        Q, Kk = (torch.empty(1, H, L, 64, device=dev, dtype=dt) for _ in range(2))
        V = torch.empty(1, H, 64, L, device=dev, dtype=dt)
        f = lambda a, w: ops.gemm(a, w, None, qkv_heads=(Q, Kk, V, qw, qw, 1e-5, ops.Q_PRESCALE, L, H))
    elif mode == "res":
        x = torch.randn(M, N, device=dev)
        f = lambda a, w: ops.gemm(a, w, x, residual=x)
    else:
        out = torch.empty(M, N, device=dev, dtype=dt)
        f = lambda a, w: ops.gemm(a, w, out, act=ACT_GELU)
    same = bench(lambda i: f(As[0], Ws[0]), 48)
    rot_w = bench(lambda i: f(As[0], Ws[i % NW]), NW)
    rot_aw = bench(lambda i: f(As[i % 16], Ws[i % NW]), NW)
    print(f"{name:14s} M={M} N={N} K={K}: same operands {same:6.1f} us | W rotating {rot_w:6.1f} us | W and A rotating {rot_aw:6.1f} us", flush=True)


run("trunk qkv", 10368, 2304, 768, "qkv")
run("trunk fc+res", 10368, 768, 768, "res")
run("trunk fc1", 10368, 3072, 768, "gelu")
run("trunk fc2+res", 10368, 768, 3072, "res")
run("dino fc1", 8224, 3072, 768, "gelu")
