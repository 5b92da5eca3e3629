#!/usr/bin/env python3
"""Weight-gradient GEMM (m324_gemm_tn) at the training step's shapes against the number of split-K slices: kernel time and the time of
the column sums over the slices' partials that follow (ops.colsum of [slices, N * Kc]).  backward.weight_grad's rule: 252 // tiles.
usage: tools/gemm_tn_slices.py [--rows 31104]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=31104)
ap.add_argument("--fine", action="store_true")
args = ap.parse_args()
M = args.rows
dev, bf = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, Kc in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    dy = torch.randn((M, N), generator=g).to(bf).to(dev)
    a = torch.randn((M, Kc), generator=g).to(bf).to(dev)
    tiles = (N // 256) * (Kc // 256)
    rule = max(1, min(64, 252 // tiles, M // 512))
    line = []
    cand = {max(1, rule // 2), max(1, rule * 3 // 4), rule, min(64, rule * 3 // 2), min(64, rule * 2)} if not args.fine else \
        set(range(max(1, rule // 3), rule + 2, 1 if rule <= 12 else 2))
    for s in sorted(cand):
        res = []
        for _ in range(3):
            tk = t(lambda: ops.gemm_tn(dy, a, s))
            part = torch.empty((s, N * Kc), dtype=torch.float32, device=dev)
            tc = t(lambda: ops.colsum(part)) if s > 1 else 0.0
            res.append((tk, tc))
        tk, tc = sorted(res)[1]
        line.append(f"{s}{'*' if s == rule else ''}: {tk:.0f} + {tc:.0f}")
    print(f"M={M} dW[{N}, {Kc}] ({tiles} tiles) slices: kernel + column sums (us)   " + "   ".join(line), flush=True)
