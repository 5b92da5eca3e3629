#!/usr/bin/env python3
"""Does a kernel compute the same thing when another kernel shares its CUs?  200 LayerNorm launches on one stream while a
second stream runs GEMMs of one schedule; every result is compared with the solo result, bit for bit.
Round 3: with the compiler's SLP packing (v_mov_b32_dpp + v_pk_add_f32 in the two-rows-per-wave reduction) 39 of 200 differed
beside the chunk-ring GEMM v13 and none beside v2 / alone; built with -fno-slp-vectorize (one v_add_f32_dpp per step): 0 / 0 / 0.
usage: tools/ln_stress.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import lib, ops
from motion324_amd.lib import ACT_GELU
dev = "cuda"
torch.manual_seed(0)
M, C = 1285, 768
x = torch.randn(M, C, device=dev) * 1.3 + 0.2
w, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
a2 = torch.randn(1028, 768, device=dev).to(torch.bfloat16)
w2 = (torch.randn(3072, 768, device=dev) * 0.02).to(torch.bfloat16)
o2 = torch.empty(1028, 3072, device=dev, dtype=torch.bfloat16)
bias2 = torch.randn(3072, device=dev)
for rows in (2, 1):
    lib.set_tunable("M324_LN_ROWS", rows)
    for gemm in ("v13", "v2", "none"):
        if gemm != "none": lib.set_tunable("M324_GEMM", int(gemm[1:]))
        ref = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
        ops.layernorm(x, w, b, 1e-6, ref)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        outs = [torch.empty(M, C, device=dev, dtype=torch.bfloat16) for _ in range(200)]
        with torch.cuda.stream(side):
            if gemm != "none":
                for _ in range(400): ops.gemm(a2, w2, o2, bias=bias2, act=ACT_GELU)
        for o in outs: ops.layernorm(x, w, b, 1e-6, o)
        torch.cuda.synchronize()
        bad = sum(int(not torch.equal(o, ref)) for o in outs)
        print(f"LN rows/wave={rows} beside {gemm}: {bad} of {len(outs)} LayerNorm results differ from the solo result", flush=True)
        lib.set_tunable("M324_GEMM", 0)
