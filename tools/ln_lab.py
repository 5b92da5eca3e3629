#!/usr/bin/env python3
"""LayerNorm pass alone at the clip's shapes: interleaved A/B of M324_LN_ROWS (rows per wave)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import lib, ops
dev = "cuda"
for rows, xdt, odt in ((10368, torch.float32, torch.bfloat16), (8224, torch.float32, torch.bfloat16), (2048, torch.float32, torch.bfloat16),
                       (65536, torch.bfloat16, torch.bfloat16), (31104, torch.float32, torch.bfloat16)):
    x = torch.randn(rows, 768, device=dev).to(xdt)
    w, b = torch.ones(768, device=dev), torch.zeros(768, device=dev)
    out = torch.empty(rows, 768, device=dev, dtype=odt)
    res = {1: [], 2: []}
    for rnd in range(5):
        for r in (1, 2):
            lib.set_tunable("M324_LN_ROWS", r)
            for _ in range(5):
                ops.layernorm(x, w, b, 1e-5, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.layernorm(x, w, b, 1e-5, out)
            e1.record()
            torch.cuda.synchronize()
            res[r].append(e0.elapsed_time(e1) / 50 * 1e3)
    lib.set_tunable("M324_LN_ROWS")
    med = {r: sorted(v)[len(v) // 2] for r, v in res.items()}
    nbytes = rows * 768 * (x.element_size() + out.element_size())
    print(f"rows={rows:6d} {str(xdt)[6:]:9s}: 1 row/wave {med[1]:6.2f} us ({nbytes / med[1] / 1e6:5.2f} TB/s)   2 rows/wave {med[2]:6.2f} us ({nbytes / med[2] / 1e6:5.2f} TB/s)")
