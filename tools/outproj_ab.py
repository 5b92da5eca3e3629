#!/usr/bin/env python3
"""The decoder out-projection (65536 x 768 x 768, bf16 out) with its epilogue parts added one at a time.  usage: tools/outproj_ab.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from motion324_amd import ops
dev, dt = "cuda", torch.bfloat16
M, N, K = 65536, 768, 768
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    return sorted(ts)[2]
a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.02).to(dt); bias = torch.randn(N, device=dev)
part = torch.empty((N // 64, M, 2), device=dev)
out = torch.empty(M, N, device=dev, dtype=dt); res = torch.randn(2048, N, device=dev); resf = torch.randn(M, N, device=dev)
for rnd in range(2):
    print("plain bf16 out:", t(lambda: ops.gemm(a, w, out)))
    print("bias:", t(lambda: ops.gemm(a, w, out, bias=bias)))
    print("bias + broadcast residual:", t(lambda: ops.gemm(a, w, out, bias=bias, residual=res, res_rows=2048)))
    print("bias + broadcast residual + stats:", t(lambda: ops.gemm(a, w, out, bias=bias, residual=res, res_rows=2048, stats_out=part)))
