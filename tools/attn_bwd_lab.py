#!/usr/bin/env python3
"""Attention backward (bf16 MFMA kernels) alone, at the training step's shapes: time of the dQ + dK/dV pair per call, A/B of
M324_ATTN_BWD_NW.  (Round 3's in-kernel s_memtime stamps of the dK/dV kernel -- the measurement behind lse / D riding with the
tile, profiles/r03_coissue_lab.md and DESIGN 7b -- needed trace macros inside the kernels; round 4 took all lab hooks out of
the product sources, the stamped build is commit 427cabf.)
usage: tools/attn_bwd_lab.py [--B 8] [--L 3888]"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--L", type=int, default=3888)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev, dt = "cuda", torch.bfloat16
B, H, L = a.B, 12, a.L
Lp = (L + 63) // 64 * 64
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *sh: (torch.randn(*sh, device=dev, generator=g)).to(dt)
Qs, K, V, dO = r(B, H, L, 64) * 0.18, r(B, H, L, 64), r(B, H, L, 64), r(B, H, L, 64)
tr = lambda x: torch.nn.functional.pad(x.transpose(2, 3), (0, Lp - L)).contiguous()      # [B,H,64,Lp] (key order irrelevant for timing)
Qst, Kt, dOt = tr(Qs), tr(K), tr(dO)
lse = torch.full((B, H, L), 9.0, dtype=torch.float32, device=dev)
D = torch.zeros((B, H, L), dtype=torch.float32, device=dev)
dQ = torch.empty((B, H, L, 64), dtype=dt, device=dev)
dK = torch.empty((B, H, L, 64), dtype=dt, device=dev)
dV = torch.empty((B * H * L * 64 + 8192,), dtype=dt, device=dev)                              # + room for the stamps
P = lambda t: C.c_void_p(t.data_ptr())
Lb = lib.load()


def call():
    lib.check(Lb.m324_attention_bwd_mfma(P(Qs), P(Qst), H * L * 64, H * 64 * Lp, P(K), P(Kt), P(V), P(dO), P(dOt), P(lse), P(D), P(dQ), P(dK),
                                         P(dV), B, H, L, L, 0.125, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "bwd")


def timeit():
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters


flops = 7 * 2.0 * B * H * L * L * 64        # dQ kernel 3 GEMMs, dK/dV kernel 4
res = {4: [], 8: [], 2: [], 84: [], 48: []}
for rnd in range(5):                         # interleaved: the clock drifts by several percent within a process
    for nw in (4, 8, 2, 84, 48):             # 84: dQ eight waves + dK/dV four; 48: the reverse; 2 (round 6): the dQ kernel with 64 queries per wave, four waves; dK/dV as with 8
        lib.set_tunable("M324_ATTN_BWD_NW", nw)
        res[nw].append(timeit())
for nw in (4, 8, 2, 84, 48):
    ms = sorted(res[nw])[len(res[nw]) // 2]
    print(f"B={B} H={H} L={L} M324_ATTN_BWD_NW={nw}: dQ + dK/dV {ms * 1e3:.1f} us per call = {flops / ms / 1e9:.0f} TF/s (median of 5 interleaved rounds)", flush=True)
lib.set_tunable("M324_ATTN_BWD_NW", 0)
