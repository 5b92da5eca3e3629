#!/usr/bin/env python3
"""Timeline of the 8-wave attention kernel from inside the kernel (lab build with -DM324_ATTN_TRACE):
    tools/build_attn_lab.sh trace -DM324_ATTN_TRACE ; M324_LIB=tools/lablibs/libm324_trace.so python tools/attn_trace.py
The first wave of each half of one workgroup (alone on its CU: the last 26 of the grid's 512 slots stay empty) stamps s_memtime
(shader cycles; tools/coissue_lab: a bare MFMA stream reads 32.0 per instruction) at points of key tiles 60-75 into the LSE
buffer; printed: cycles between consecutive stamps, median over the tiles.
    0 tile body done | 1 own LDS-DMA pieces landed | 2 barrier passed | 3 S MFMAs issued | 4 maximum / vote done |
    5 (exp / pack: the optimiser sinks most of it past this stamp, between the P.V MFMAs) | (next 0) P.V issued"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import lib, ops

dev, dt = "cuda", torch.bfloat16
B, H, L = 1, 12, 10368
q = (torch.randn(B, H, L, 64, device=dev) * ops.Q_PRESCALE).to(dt)
k = torch.randn(B, H, L, 64, device=dev).to(dt)
vt = torch.randn(B, H, 64, (L + 63) // 64 * 64, device=dev).to(dt)
out = torch.empty(B * L, H * 64, device=dev, dtype=dt)
lse = torch.zeros(B, H, L, dtype=torch.float32, device=dev)
for _ in range(3):
    ops.attention(q, k, vt, out, prescaled=True, lse=lse)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.attention(q, k, vt, out, prescaled=True, lse=lse)
e1.record()
torch.cuda.synchronize()
tb = lse.view(-1).view(torch.int64).cpu()
print(f"{e0.elapsed_time(e1) * 1e3:.1f} us (traced build)")
NSLOT = 6
for half in (0, 1):
    st = tb[half * 1024: half * 1024 + 16 * 8].reshape(16, 8).tolist()
    seq = [st[t][sl] for t in range(16) for sl in range(NSLOT)]
    deltas = {sl: [] for sl in range(NSLOT)}
    for i in range(len(seq) - 1):
        deltas[i % NSLOT].append(seq[i + 1] - seq[i])
    med = {sl: sorted(v)[len(v) // 2] for sl, v in deltas.items() if v}
    print(f"  wave {4 * half}: " + "  ".join(f"{sl}->{(sl + 1) % NSLOT}: {med[sl]}" for sl in range(NSLOT)) + f"   period {sum(med.values())}")
