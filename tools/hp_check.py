#!/usr/bin/env python3
"""Schedule v15 (csrc/gemm_hp.hip, the hand-placed K = 768 stream) against v10 / v13 on the model's shapes: values (the arithmetic is the
same, so the outputs are expected to be bit-identical to v10's) and interleaved timing.
usage (GPU box): python3 tools/hp_check.py [--quick] [--iters N]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import lib, ops
from motion324_amd.lib import ACT_GELU, ACT_NONE

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--no-time", action="store_true")
args = ap.parse_args()
dev = "cuda"
torch.manual_seed(0)


def run(variant, fn):
    lib.set_tunable("M324_GEMM", variant)
    try:
        return fn()
    finally:
        lib.set_tunable("M324_GEMM")


def timeit(variant, fn, iters):
    lib.set_tunable("M324_GEMM", variant)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    lib.set_tunable("M324_GEMM")
    return e0.elapsed_time(e1) / iters * 1e3


SHAPES = [("small 256x128", 256, 128), ("one tile ragged 200x128", 200, 128), ("2 x 3 tiles 512x384", 512, 384), ("ragged 1000x256", 1000, 256),
          ("many tiles per wg 2304x3072", 2304 * 4, 3072),
          ("trunk fc1", 10368, 3072), ("dino fc1", 8224, 3072), ("trunk N=2304", 10368, 2304), ("dec fc1", 65536, 3072)]
if args.quick:
    SHAPES = SHAPES[:4] + SHAPES[5:6]
bad = 0
for name, M, N in SHAPES:
    K = 768
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    rowstat = torch.stack([torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev) * 0.1], dim=1).contiguous()
    colsum = torch.randn(N, device=dev)
    for mode in ("gelu", "plain", "fold_gelu", "fold"):
        act = ACT_GELU if "gelu" in mode else ACT_NONE
        ln = (rowstat, colsum) if "fold" in mode else None
        outs = {}
        for var in (15, 10):
            out = torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
            run(var, lambda: ops.gemm(a, w, out, bias=bias, act=act, ln=ln))
            torch.cuda.synchronize()
            outs[var] = out.float()
        ref = a.float() @ w.float().t()
        if ln is not None:
            ref = rowstat[:, :1] * ref + rowstat[:, 1:2] * colsum[None, :]
        ref = ref + bias[None, :]
        if act == ACT_GELU:
            ref = torch.nn.functional.gelu(ref)
        d15 = (outs[15] - ref).abs().max().item()
        d10 = (outs[10] - ref).abs().max().item()
        same = torch.equal(outs[15], outs[10])
        nbad = (outs[15] != outs[10]).sum().item()
        line = f"{name:32s} {mode:10s} v15 vs fp32 ref {d15:.3e} (v10: {d10:.3e})  bit-identical to v10: {same}"
        if not same:
            bad += 1
            idx = (outs[15] != outs[10]).nonzero()
            rows = idx[:, 0].unique()
            cols = idx[:, 1].unique()
            line += f"  [{nbad} differ; rows {rows[:6].tolist()}..{rows[-1].item()} ({len(rows)}), cols {cols[:6].tolist()}..{cols[-1].item()} ({len(cols)}); nan {torch.isnan(outs[15]).sum().item()}]"
        print(line, flush=True)
        if not args.no_time and M >= 2000 and mode in ("gelu", "fold_gelu", "plain"):
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            fn = lambda: ops.gemm(a, w, out, bias=bias, act=act, ln=ln)
            t = {v: [] for v in (15, 10, 13)}
            for _ in range(3):
                for v in t:
                    t[v].append(timeit(v, fn, args.iters))
            print("    " + "   ".join(f"v{v}: {sorted(x)[1]:.1f} us" for v, x in t.items()) + f"   ({2.0 * M * N * K / sorted(t[15])[1] / 1e6:.0f} TF/s v15)", flush=True)
print("FAILED" if bad else "ALL BIT-IDENTICAL")
