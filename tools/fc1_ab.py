#!/usr/bin/env python3
"""fc1 + GELU at the decoder's (65536 rows) and the trunk's (10368 rows) shape in its three product forms (bias, LayerNorm-fold consumer with a
merged table, fold consumer merging the block table itself), through the chooser.  For A/Bs between library builds: M324_LIB=...
usage: [M324_LIB=...] tools/fc1_ab.py"""
import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/motion324_amd") else os.getcwd())
from motion324_amd import lib as L
from motion324_amd import ops
from motion324_amd.lib import ACT_GELU
dev, dt = "cuda", torch.bfloat16
for M in (65536, 10368):
    N, K = 3072, 768
    a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    bias = torch.randn(N, device=dev); cs = w.float().sum(1).contiguous()
    stat = torch.empty((M, 2), device=dev); ops.rowstats(a.float(), 1e-5, stat)
    out = torch.empty(M, N, device=dev, dtype=dt)
    part = torch.randn(K // 64, M, 2, device=dev).abs().contiguous()
    modes = [("plain gelu", {}), ("fold consumer (merged table)", dict(ln=(stat, cs)))]
    modes.append(("fold consumer (block table, merged in the kernel)", dict(ln=(part, cs, 1e-5))))
    for name, kw in (modes[::-1] + modes):
        fn = lambda: ops.gemm(a, w, out, bias=bias, act=ACT_GELU, **kw)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        print(f"M={M} {name}: {sorted(ts)[2]:.1f} us", flush=True)
