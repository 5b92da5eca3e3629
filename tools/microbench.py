#!/usr/bin/env python3
"""Per-kernel microbenchmarks at the shapes of the c2 clip (used for A/B work and PMC collection).
usage: tools/microbench.py [gemm|attn|all] [--iters N] [--only SUBSTR]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import lib, ops
from motion324_amd.lib import ACT_GELU

ap = argparse.ArgumentParser()
ap.add_argument("what", nargs="?", default="all")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--ab", default="", help="ENV=a,b : interleaved A/B of an environment switch read per launch")
args = ap.parse_args()
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
dev = "cuda"


def timeit(fn, iters):
    if args.ab:
        var, vals = args.ab.split("=")
        vals = vals.split(",")
        res = {v: [] for v in vals}
        for rnd in range(6):
            for v in vals:
                lib.set_tunable(var, int(v.lstrip("vV")))           # the library reads its environment once, at load
                res[v].append(_timeit(fn, max(3, iters // 4)))
        lib.set_tunable(var)
        med = {v: sorted(t)[len(t) // 2] for v, t in res.items()}
        print("   A/B " + "  ".join(f"{var}={v}: {m * 1e3:.1f} us" for v, m in med.items()))
        return min(med.values())
    return _timeit(fn, iters)


def _timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


GEMMS = [  # name, M, N, K, mode
    ("trunk qkv", 10368, 2304, 768, "plain"), ("trunk fc+res", 10368, 768, 768, "res"),
    ("trunk fc1 gelu", 10368, 3072, 768, "gelu"), ("trunk fc2+res", 10368, 768, 3072, "res"),
    ("dino qkv", 8224, 2304, 768, "bias"), ("dino fc1 gelu", 8224, 3072, 768, "gelu"), ("dino fc2+res", 8224, 768, 3072, "res"),
    ("dec fc+res", 65536, 768, 768, "res"), ("dec fc1 gelu", 65536, 3072, 768, "gelu"), ("dec fc2+res", 65536, 768, 3072, "res"),
    ("dec fc2 bf16 stream", 65536, 768, 3072, "resb"), ("dec fc bf16 stream", 65536, 768, 768, "resb"),
    ("pcd fc2 M=64", 64, 768, 3072, "res"), ("pcd qkv M=64", 64, 2304, 768, "plain"),
    ("square 4096", 4096, 4096, 4096, "plain"),
]
if args.what in ("gemm", "all"):
    for name, M, N, K, mode in GEMMS:
        if args.only and args.only not in name:
            continue
        a = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
        bias = torch.randn(N, device=dev)
        if mode == "res":
            out = torch.randn(M, N, device=dev)
            fn = lambda: ops.gemm(a, w, out, residual=out)
        elif mode == "resb":                  # the decoder's bf16 residual stream, updated in place
            out = torch.randn(M, N, device=dev).to(dt)
            fn = lambda: ops.gemm(a, w, out, residual=out)
        elif mode == "gelu":
            out = torch.empty(M, N, device=dev, dtype=dt)
            fn = lambda: ops.gemm(a, w, out, bias=bias, act=ACT_GELU)
        elif mode == "bias":
            out = torch.empty(M, N, device=dev, dtype=dt)
            fn = lambda: ops.gemm(a, w, out, bias=bias)
        else:
            out = torch.empty(M, N, device=dev, dtype=dt)
            fn = lambda: ops.gemm(a, w, out)
        ms = timeit(fn, args.iters)
        print(f"gemm {name:16s} M={M:6d} N={N:5d} K={K:5d} {mode:5s}  {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)

ATTNS = [("global", 1, 12, 10368, 10368), ("local", 32, 12, 324, 324), ("dino", 32, 12, 257, 257), ("decoder", 32, 12, 2048, 64),
         ("long 82944", 1, 12, 82944, 82944)]
if args.what in ("attn", "all"):
    for name, B, H, Lq, Lk in ATTNS:
        if args.only and args.only not in name:
            continue
        if Lk > 20000 and not args.only:
            continue
        shared = name == "decoder"
        q = (torch.randn(1 if shared else B, H, Lq, 64, device=dev) * ops.Q_PRESCALE).to(dt)
        k = torch.randn(B, H, Lk, 64, device=dev).to(dt)
        vt = torch.randn(B, H, 64, (Lk + 63) // 64 * 64, device=dev).to(dt)
        out = torch.empty(B * Lq, H * 64, device=dev, dtype=dt)
        fn = lambda: ops.attention(q, k, vt, out, shared_q=shared, prescaled=True)
        ms = timeit(fn, max(3, args.iters // (1 + Lk // 20000 * 10)))
        print(f"attn {name:12s} B={B:3d} H={H} Lq={Lq:6d} Lk={Lk:6d}  {ms * 1e3:9.1f} us  {4.0 * B * H * Lq * Lk * 64 / ms / 1e9:7.1f} TF/s", flush=True)
        if dt == torch.bfloat16:                    # row-major V operand (transposing LDS reads)
            vr = torch.randn(B, H, Lk, 64, device=dev).to(dt)
            fr = lambda: ops.attention(q, k, vr, out, shared_q=shared, prescaled=True, v_rowmajor=True)
            ms2 = timeit(fr, max(3, args.iters // (1 + Lk // 20000 * 10)))
            print(f"     {'row-major V':12s} {'':38s}{ms2 * 1e3:9.1f} us  {4.0 * B * H * Lq * Lk * 64 / ms2 / 1e9:7.1f} TF/s", flush=True)
