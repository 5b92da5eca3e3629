"""Host side of libm324 under AddressSanitizer + UndefinedBehaviorSanitizer (race / memory checking).

GPU ASan is not available on this pool; what CAN be checked on a CPU is the part of the C ABI that handles a caller's raw
pointers and sizes before any HIP call: argument validation, the kernel chooser and its plan queries, the tunable table,
the error buffer.  motion324_amd.build.build_sanitized() compiles every translation unit --cuda-host-only with
-fsanitize=address,undefined (no device code: a few seconds); a child Python with the ASan runtime preloaded drives it."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

DRIVER = r'''
import ctypes as C, importlib.util, itertools, os, sys
spec = importlib.util.spec_from_file_location("m324lib", os.path.join(sys.argv[1], "motion324_amd", "lib.py"))
L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
h = L.load()
assert h.m324_abi_version() == L.ABI_VERSION
buf = C.create_string_buffer(8)
# the error text is longer than the caller's buffer: it must be cut, not overrun
assert h.m324_gemm(None, None) == -1
h.m324_last_error(buf, 8); assert len(buf.value) <= 7
assert h.m324_last_error(None, 0) > 0                  # no buffer: the length of the text
big = C.create_string_buffer(512)
# tunables: every known name, an unknown one, the restore value
for name in ("M324_GEMM", "M324_XCD", "M324_HP", "M324_ATTN_NW", "M324_QKV_RING", "M324_NT_MB", "M324_GEMM_PERSIST"):
    for v in (0, 1, 15, -1, 2 ** 31 - 1, L.TUNABLE_DEFAULT):
        assert h.m324_set_tunable(name.encode(), v) == 0, name
assert h.m324_set_tunable(b"M324_NO_SUCH_SWITCH", 1) < 0
assert h.m324_set_tunable(None, 1) < 0
# the chooser over the model's shapes and a sweep of ragged ones, every forced schedule, every epilogue flag it looks at
shapes = [(10368, 3072, 768), (8224, 2304, 768), (65536, 768, 3072), (64, 768, 768), (2048, 1536, 768), (1, 128, 64),
          (257, 768, 640), (4096, 768, 832), (31104, 2304, 768), (7, 96, 64), (10368, 768, 768), (300, 3, 768)]
n_plans = 0
for forced in (0, 1, 2, 5, 9, 10, 11, 12, 13, 14, 15, 7):
    assert h.m324_set_tunable(b"M324_GEMM", forced) == 0
    for (M, N, K), out_dt, act, res, fold, heads in itertools.product(shapes, (L.F32, L.BF16), (0, 1), (0, 1), (0, 1, 2), (0, 1)):
        a = L.GemmArgs()
        a.A = a.W = a.C = 4096
        a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, K, K, N
        a.in_dtype, a.out_dtype, a.act, a.batch = L.BF16, out_dt, act, 1
        if res:
            a.residual, a.ldr = 4096, N
        if fold:
            a.ln_rowstat, a.ln_colsum, a.ln_eps = 4096, 4096, 1e-5
            a.ln_ncb = 0 if fold == 1 else K // 64
        if heads and N % 192 == 0 and not res:
            a.aux_mode, a.qkv_q, a.qkv_k, a.qkv_v, a.qkv_L, a.qkv_H = 2, 4096, 4096, 4096, max(M // 4, 1), N // 192
        rc = h.m324_gemm_plan(C.byref(a), big, 512)          # the schedule's number, or a negative status
        assert rc != 0 and (rc < 0 or big.value), (M, N, K, rc)
        n_plans += 1
        h.m324_gemm_plan(C.byref(a), buf, 8)          # a short buffer
assert h.m324_set_tunable(b"M324_GEMM", L.TUNABLE_DEFAULT) == 0
assert h.m324_gemm_plan(None, big, 512) < 0
n_attn = 0
for B, H, Lq, Lk, flags, dt in itertools.product((1, 32), (1, 12), (1, 64, 257, 324, 2048, 10368), (1, 64, 257, 4096, 10368), (0, 1, 257, 3 + 256),
                                                 (L.F32, L.BF16)):
    h.m324_attention_plan(B, H, Lq, Lk, flags, dt, big, 512)
    n_attn += 1
assert h.m324_attention_plan(0, 0, 0, 0, 0, 7, big, 512) < 0
# validation failures of the launchers: they must return before any HIP call (this box has no GPU)
a = L.GemmArgs(); a.A = a.W = a.C = 16; a.M, a.N, a.K, a.lda, a.ldw, a.ldc = 4, 128, 40, 40, 40, 128
assert h.m324_gemm(C.byref(a), None) == -1
assert h.m324_gemm_pair(None, None, None) < 0
assert h.m324_attention(16, 0, 16, 16, 16, 64, 1, 1, 4, 4, 0.125, 0, None, 7, None) == -3
assert h.m324_layernorm(16, 768, 16, None, 1e-5, 16, 768, 0, 4, 770, 0, 0, 0, None) == -1
assert h.m324_rowstats_finish(None, 12, 4, 1e-5, None, None) < 0
assert h.m324_gemm_tn(None, 0, None, 0, None, 0, 4, 4, 4, 1, 0, None) < 0
assert h.m324_comm_init(None, None, 0, 1) < 0
items = (L.ColsumItem * 3)()
assert h.m324_colsum_multi(None, 0, None) < 0 and h.m324_colsum_multi(items, 3, None) < 0          # null pointers in the items
items[0].dst, items[0].src, items[0].ld, items[0].rows, items[0].cols, items[0].chain = 4096, 4096, 8, 2, 8, 1
assert h.m324_colsum_multi(items, 1, None) < 0                                                       # a chain without a head
assert h.m324_weight_mirror(None, None, None, None, 0, 0, None) < 0 and h.m324_weight_mirror(16, 16, None, 16, 1, 0, None) < 0
hexbuf = C.create_string_buffer(16)
assert h.m324_comm_unique_id(hexbuf, 16) < 0          # n < 257: refused before RCCL is looked up
print("SAN_OK", n_plans, n_attn)
'''


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_host_side_of_the_c_abi_is_clean_under_asan_and_ubsan():
    from motion324_amd import build
    lib = build.build_sanitized()
    rt = build.sanitizer_runtime()
    assert os.path.exists(lib) and os.path.exists(rt), (lib, rt)
    env = dict(os.environ, M324_LIB=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", DRIVER, REPO], capture_output=True, text=True, env=env, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "SAN_OK" in r.stdout, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
