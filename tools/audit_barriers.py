#!/usr/bin/env python3
"""Static check of the compiled kernels: every s_barrier of a kernel that stages tiles by LDS-DMA (buffer_load ... lds) must
have a vmcnt wait in the 14 instructions before it -- __syncthreads() alone does NOT make the compiler wait for an
in-flight LDS-DMA (round 1: the attention dQ kernel read a stage that had not landed).  Barriers that carry the asm comment
`m324-audit:` are exempt: they order epilogue scratch, not ring stages.
usage: tools/audit_barriers.py   (reads the assembly motion324_amd.build.assembly() keeps under csrc/build/asm)"""
import os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from motion324_amd import build as B

bad_total = 0
n_dma_kernels = 0
SRCS = ("gemm.hip", "gemm_ring4.hip", "attention.hip", "attention_pwg.hip")
asm = B.assembly(SRCS)
for src in SRCS:
    if True:
        out = asm[src]
        kernels, name = {}, None
        for line in open(out):
            m = re.match(r"^(_ZN\S+):", line)
            if m:
                name = m.group(1)
                kernels[name] = []
            if name:
                kernels[name].append(line)
    for k, ls in kernels.items():
        if not any("global_load_lds" in l or ("buffer_load_dword" in l and " lds" in l) for l in ls):
            continue
        n_dma_kernels += 1
        # (a barrier written with the marker `m324-audit:` orders epilogue scratch only -- gemm_tile.h store_tile_lds)
        bad = sum(1 for i, l in enumerate(ls) if "s_barrier" in l and "m324-audit:" not in l and "vmcnt" not in "".join(ls[max(0, i - 14):i]))
        if bad:
            bad_total += bad
            print(f"{src}: {re.sub(r'_ZN12_GLOBAL__N_1[0-9]+', '', k)[:90]}: {bad} barrier(s) without a vmcnt wait")
print(f"{n_dma_kernels} kernels stage tiles by LDS-DMA (global_load_lds / buffer_load ... lds)")
if n_dma_kernels < 20:
    print("the audit found (almost) no LDS-DMA kernel: the instruction pattern it looks for has changed")
    sys.exit(1)
print("OK: every LDS-DMA kernel waits (vmcnt) before its barriers" if not bad_total else f"{bad_total} suspicious barriers")
sys.exit(1 if bad_total else 0)
