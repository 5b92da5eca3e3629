#!/usr/bin/env python3
"""The decoder block's two residual-stream GEMMs in their product forms (bf16 stream + LayerNorm-fold statistics: out-projection with the
broadcast fp32 point features as residual, fc2 updating the stream in place), every schedule that builds them, interleaved.
usage: tools/dec_gemm_ab.py [--variants 0,10,11,12,13]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import lib, ops
ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,10,11,12,13")
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
dev, dt = "cuda", torch.bfloat16
M = 65536


def t(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


for name, N, K in (("dec out-proj (broadcast residual + statistics)", 768, 768), ("dec fc2 (bf16 stream in place + statistics)", 768, 3072)):
    a = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    bias = torch.randn(N, device=dev)
    part = torch.empty((N // 64, M, 2), device=dev)
    if K == 3072:
        out = torch.randn(M, N, device=dev).to(dt)
        fn = lambda: ops.gemm(a, w, out, bias=bias, residual=out, stats_out=part)
    else:
        out = torch.empty(M, N, device=dev, dtype=dt)
        res = torch.randn(2048, N, device=dev)
        fn = lambda: ops.gemm(a, w, out, bias=bias, residual=res, res_rows=2048, stats_out=part)
    res_t = {v: [] for v in args.variants.split(",")}
    plans = {}
    for rnd in range(args.rounds):
        for v in res_t:
            lib.set_tunable("M324_GEMM", int(v))
            try:
                res_t[v].append(t(fn))
            except Exception as e:                       # a schedule that does not build this epilogue
                res_t[v].append(float("nan"))
            lib.set_tunable("M324_GEMM")
    print(name + ": " + "  ".join(f"v{v}: {sorted(ts)[len(ts) // 2]:.1f} us" for v, ts in res_t.items()), flush=True)
