#!/usr/bin/env python3
"""Timeline of the 256 x 256 chunk-ring GEMM (v10) from inside the kernel (lab build):
    tools/build_lab_lib.sh gtrace -DM324_GEMM_TRACE ; M324_LIB=tools/lablibs/libm324_gtrace.so python tools/gemm_trace.py
Waves 0 and 4 of one workgroup stamp s_memtime (shader cycles): entry, prologue issued, stage 0 landed, first barrier, then
per K-stage {32 MFMAs issued, own LDS-DMA landed, barrier passed}, then ring drained and epilogue done.  Ideal stage: two
waves per SIMD x 32 MFMAs x 32 cycles = 2048 cycles."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import lib, ops
from motion324_amd.lib import ACT_GELU

dev, dt = "cuda", torch.bfloat16
Lb = lib.load()
buf = torch.zeros(256, dtype=torch.int64, device=dev)      # 2 waves x 128 slots (the kernel stops stamping at slot 120)
assert Lb.m324_lab_trace_buffer(C.c_void_p(buf.data_ptr())) == 0
lib.set_tunable("M324_GEMM", 10)
for name, M, N, K, gelu in (("dec fc1 gelu", 65536, 3072, 768, True), ("trunk fc1 gelu", 10368, 3072, 768, True),
                            ("dec fc1 plain", 65536, 3072, 768, False), ("K = 3072 plain", 65536, 768, 3072, False)):
    a = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    fn = (lambda: ops.gemm(a, w, out, bias=bias, act=ACT_GELU)) if gelu else (lambda: ops.gemm(a, w, out))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    NS = min(K // 64, 38)                     # stamps: 6 + 3 per K-stage, 120 slots
    print(f"== {name}: M={M} N={N} K={K}: {e0.elapsed_time(e1) * 1e3:.1f} us (traced build), {NS} K-stages")
    tb = buf.cpu().tolist()
    for half in (0, 1):
        t = tb[half * 128: half * 128 + 6 + 3 * NS]
        if not t[0]:
            print("   (no stamps)"); continue
        pro = f"set-up->issued {t[1] - t[0]}  landed {t[2] - t[1]}  barrier {t[3] - t[2]}"
        st = []
        prev = t[3]
        for s in range(NS):
            m, l, b = t[4 + 3 * s], t[5 + 3 * s], t[6 + 3 * s]
            st.append((m - prev, l - m, b - l))
            prev = b
        med = lambda i: sorted(x[i] for x in st[1:-1])[len(st[1:-1]) // 2] if len(st) > 2 else st[0][i]
        full = NS == K // 64
        tail = t[4 + 3 * NS] - prev if full else -1
        epi = t[5 + 3 * NS] - t[4 + 3 * NS] if full else -1
        total = t[5 + 3 * NS] - t[0] if full else -1
        print(f"   wave {4 * half}: prologue [{pro}]  stage0 {st[0]}  median stage: mfma-issue {med(0)} dma-wait {med(1)} barrier {med(2)} "
              f"(sum {med(0) + med(1) + med(2)})  last k-step + drain {tail}  epilogue {epi}  tile {total}")
