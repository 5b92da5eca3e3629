"""Builds motion324_amd/libm324.so (gfx950 HIP kernels behind the C ABI of include/m324.h).

In-tree build with hipcc only (no torch extension machinery: the library has no torch symbols).
``python -m motion324_amd.build`` or ``build()``; rebuilds only stale objects.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libm324.so")
SOURCES = ["runtime.hip", "gemm.hip", "gemm_ring4.hip", "gemm_hp.hip", "attention.hip", "attention_pwg.hip", "elementwise.hip", "backward.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_tile.h"), os.path.join(HERE, "..", "include", "m324.h"),
           os.path.join(CSRC, "attn_pwg_asm.inc"), os.path.join(CSRC, "attn_pwg_bounded_asm.inc"), os.path.join(CSRC, "attn_pwg_clobbers.inc"),
           os.path.join(CSRC, "attn_pwg_kernel.inl"), os.path.join(CSRC, "gemm_hp_kernel.inl"), os.path.join(CSRC, "gemm_hp_clobbers.inc")] + [
           os.path.join(CSRC, f"gemm_hp_{k}.inc") for k in ("gelu", "fold_gelu", "plain", "fold")]
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950 has one unified 512-entry file), which removes the
# v_accvgpr_read/write shuffling around every softmax / epilogue access of an accumulator
# -fno-slp-vectorize: hipcc packs adjacent scalar fp32 adds / multiplies into v_pk_*_f32.  Next to a DPP reduction that turns
# `v += dpp(v)` (one v_add_f32_dpp) into v_mov_b32_dpp + v_pk_add_f32, and THAT pair returned wrong sums on MI355X whenever a
# chunk-ring GEMM (v10 / v13) ran on the same CUs from a second stream or graph branch: 39 of 200 LayerNorm launches off by an
# ulp-sized statistic, 23 of 40 graph replays of the two-branch image encoder (round 3; tools/ln_stress.py reproduces it,
# tools/audit_dpp.py scans the compiled code for the pattern).  Packed fp32 buys nothing here anyway: v_pk_add_f32 issues at
# 4.5 cycles, two v_add_f32 at 2.3 each (tools/valu_lab); the hand-written f32x2 code (GELU polynomial) is not affected.
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize"]
FLAGS = BASE_FLAGS + ["-mllvm", "-amdgpu-mfma-vgpr-form"]
# gemm_ring4.hip: 256 accumulators per wave -> they must stay in the AGPR half (see the file header)
FLAGS_OF = {"gemm_ring4.hip": BASE_FLAGS}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s, os.path.abspath(__file__)] + HEADERS):      # the flags live in this file
            jobs.append([hipcc] + FLAGS_OF.get(src, FLAGS) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


SAN_LIB = os.path.join(OBJ, "san", "libm324_san.so")


def build_sanitized() -> str:
    """Host side of every translation unit (``--cuda-host-only``: launchers, argument validation, the m324_*_plan queries, the
    kernel chooser, the tunable table; no device code) with AddressSanitizer + UndefinedBehaviorSanitizer, linked against the
    shared sanitizer runtime.  GPU ASan is not available on this pool, and the host side is where the C ABI takes raw pointers
    and sizes from a caller; tests/test_sanitizer.py loads the result in a child process (LD_PRELOAD of the runtime) and drives
    every entry point that returns before its first HIP call.  A few seconds: there is no device compilation."""
    sdir = os.path.dirname(SAN_LIB)
    os.makedirs(sdir, exist_ok=True)
    hipcc = _hipcc()
    flags = ["--offload-arch=gfx950", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fPIC", "-Wno-unused-function",
             "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared-libsan"]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(sdir, src.replace(".hip", ".o"))
        objs.append(o)
        if _stale(o, [s, os.path.abspath(__file__)] + HEADERS):
            jobs.append([hipcc] + flags + ["-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (sanitizer build) failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or _stale(SAN_LIB, objs):
        # a host-only object still refers to its translation unit's device-code bundle (__hip_fatbin_<hash>; the module
        # constructor hands it to the HIP runtime, which parses it lazily, at the first launch): an EMPTY bundle each
        r = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True)
        syms = sorted({ln.split()[-1] for ln in r.stdout.splitlines() if "__hip_fatbin_" in ln})
        stub = os.path.join(sdir, "fatbin_stub.c")
        with open(stub, "w") as f:
            f.write("/* GENERATED by motion324_amd.build.build_sanitized: empty offload bundles for a host-only build */\n")
            for sym in syms:
                f.write(f'__attribute__((aligned(4096))) const char {sym}[4096] = "__CLANG_OFFLOAD_BUNDLE__";\n')
        so = os.path.join(sdir, "fatbin_stub.o")
        run(["gcc", "-fPIC", "-c", stub, "-o", so])
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", SAN_LIB] + objs + [so, "-ldl"])
    return SAN_LIB


def sanitizer_runtime() -> str:
    """path of the shared ASan runtime the sanitized library needs preloaded into a Python child"""
    r = subprocess.run([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = r.stdout.strip()
    if not os.path.isabs(path):
        import glob
        hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
        path = hits[0] if hits else ""
    return path


def assembly(sources=None) -> dict:
    """Device assembly (.s) of the given translation units with the build's flags, for the static audits under tools/:
    csrc/build/asm/<name>.s, rebuilt only when stale, four compilations at a time."""
    adir = os.path.join(OBJ, "asm")
    os.makedirs(adir, exist_ok=True)
    hipcc = _hipcc()
    out, jobs = {}, []
    for src in (sources or SOURCES):
        s = os.path.join(CSRC, src)
        a = os.path.join(adir, src.replace(".hip", ".s"))
        out[src] = a
        if _stale(a, [s, os.path.abspath(__file__)] + HEADERS):
            flags = [f for f in FLAGS_OF.get(src, FLAGS) if f not in ("-fPIC", "-Wall")]
            jobs.append([hipcc] + flags + ["--cuda-device-only", "-S", "-o", a, s])
    with ThreadPoolExecutor(max_workers=4) as ex:
        for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs):
            if r.returncode != 0:
                raise RuntimeError("hipcc -S failed:\n" + r.stderr)
    return out


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
