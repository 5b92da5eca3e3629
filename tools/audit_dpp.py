#!/usr/bin/env python3
"""Static check of the compiled kernels for the instruction pair that returned wrong values on MI355X beside a chunk-ring
GEMM on the same CUs (round 3): a packed-fp32 instruction (v_pk_*_f32) that reads a register a DPP
move (v_mov_b32_dpp) wrote within the last few instructions.  `v += dpp(v)` must compile to ONE v_add_f32_dpp; the build's
-fno-slp-vectorize keeps the compiler from splitting it.
usage: tools/audit_dpp.py   (reads the assembly motion324_amd.build.assembly() keeps under csrc/build/asm)"""
import os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from motion324_amd import build as B

WINDOW = 6
bad_total = 0
asm = B.assembly()
for src in B.SOURCES:
    if True:
        out = asm[src]
        name, recent, n_pk, n_dppmov = None, [], 0, 0
        for line in open(out):
            m = re.match(r"^(_Z\S+):", line)
            if m:
                name, recent = m.group(1), []
                continue
            t = line.strip()
            if not t or t.startswith((";", ".")):
                continue
            op = t.split()[0]
            if op == "v_mov_b32_dpp":
                n_dppmov += 1
                dst = t.split()[1].rstrip(",")
                recent.append((dst, 0))
            elif op.startswith("v_pk_") and "f32" in op:
                n_pk += 1
                regs = set()
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
                    regs.update(f"v{i}" for i in range(int(a), int(b) + 1))
                regs.update(re.findall(r"\bv\d+\b", t))
                hit = [d for d, age in recent if d in regs]
                if hit:
                    bad_total += 1
                    print(f"{src}: {name[:80]}: {op} reads {hit} written by v_mov_b32_dpp <= {WINDOW} instructions earlier")
            recent = [(d, age + 1) for d, age in recent if age + 1 <= WINDOW]
        print(f"{src}: {n_dppmov} v_mov_b32_dpp, {n_pk} packed-fp32 instructions")
print("OK: no packed-fp32 instruction consumes a fresh DPP move" if not bad_total else f"{bad_total} suspicious pairs")
sys.exit(1 if bad_total else 0)
