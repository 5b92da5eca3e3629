"""Static checks on the compiled gfx950 code (no GPU needed: hipcc cross-compiles)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_lds_dma_kernels_wait_for_their_tiles_before_every_barrier():
    """__syncthreads() does not make hipcc wait for an in-flight global_load_lds: every barrier of a kernel that stages
    tiles by LDS-DMA needs an explicit vmcnt wait (tools/audit_barriers.py reads the generated assembly)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_barriers.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_compiler_is_the_one_the_lds_alias_views_were_validated_on():
    """The ring GEMMs hand ONE LDS buffer to their body through three __restrict__ views (csrc/gemm.hip gemm_ring_body, tn_pipe_body:
    the invariant is written there).  That is validated by reading the ISA, per compiler: a new hipcc must be re-read (no compiler
    `s_waitcnt vmcnt(0)` inside the K loops, every barrier behind its counted wait -- tools/audit_barriers.py) before this pin moves."""
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True)
    assert "roc-7.2.0" in r.stdout and "HIP version: 7.2." in r.stdout, \
        "hipcc changed: re-validate the __restrict__ LDS views of the ring GEMMs (DESIGN.md section 6), then update this pin\n" + r.stdout


def test_no_packed_fp32_instruction_consumes_a_fresh_dpp_move():
    """v_mov_b32_dpp + v_pk_add_f32 (what the SLP vectoriser makes of two interleaved `v += dpp(v)` butterflies) returned wrong
    sums on MI355X beside a chunk-ring GEMM on the same CUs (round 3); the build's -fno-slp-vectorize keeps
    every step one v_add_f32_dpp.  tools/audit_dpp.py compiles every translation unit with the build's flags and scans for the pair."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_dpp.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    from motion324_amd import build
    assert "-fno-slp-vectorize" in build.FLAGS and "-fno-slp-vectorize" in build.FLAGS_OF["gemm_ring4.hip"]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_hand_placed_attention_stream_owns_its_registers():
    """attention_pwg.hip names its registers (v32-255, a0-195) inside one asm statement; the compiler keeps out of them only because
    they are on the clobber list.  Audit the generated code: no spill, no scratch, one wave per SIMD worth of registers, no compiler
    v_accvgpr_* / v_mfma outside the statement, and the committed .inc files are what the generator writes today."""
    import re
    import subprocess as sp
    sys.path.insert(0, ROOT)
    from motion324_amd import build as B
    asm = B.assembly(["attention_pwg.hip"])["attention_pwg.hip"]
    text = open(asm).read()
    kernels = re.findall(r"\.amdhsa_kernel (\S*attn_pwg\S*)", text)
    assert len(kernels) == 2, kernels                                    # lazy-maximum and bounded streams
    meta = text[text.index("amdhsa.kernels"):]
    for field, want in ((".vgpr_spill_count", "0"), (".sgpr_spill_count", "0"), (".private_segment_fixed_size", "0")):
        vals = re.findall(re.escape(field) + r":\s*(\d+)", meta)
        assert vals and all(v == want for v in vals), (field, vals)
    assert all(int(v) > 256 for v in re.findall(r"\.vgpr_count:\s*(\d+)", meta))        # VGPRs + AGPRs: one wave per SIMD
    outside, inside = [], False
    for line in text.splitlines():
        if ";;#ASMSTART" in line:
            inside = True
        elif ";;#ASMEND" in line:
            inside = False
        elif not inside and ("v_accvgpr" in line or "v_mfma" in line):
            outside.append(line.strip())
    assert not outside, outside[:5]
    # the generator is deterministic: regenerating must reproduce the committed streams byte for byte
    csrc = os.path.join(ROOT, "motion324_amd", "csrc")
    before = {f: open(os.path.join(csrc, f)).read() for f in ("attn_pwg_asm.inc", "attn_pwg_bounded_asm.inc", "attn_pwg_clobbers.inc")}
    r = sp.run([sys.executable, os.path.join(csrc, "gen_attn_pwg.py")], capture_output=True, text=True, cwd=csrc)
    assert r.returncode == 0, r.stdout + r.stderr                        # includes the generator's own hazard checks
    for f, old in before.items():
        assert open(os.path.join(csrc, f)).read() == old, f"{f} is stale: run csrc/gen_attn_pwg.py"


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_hand_placed_gemm_stream_owns_its_registers():
    """gemm_hp.hip (schedule v15) names its registers (v16-255, a0-255, s40-87) inside one asm statement per kernel: no spill, no
    scratch, 512 registers, no compiler v_accvgpr_* / v_mfma outside the statement; every stream is built twice (plain and
    nontemporal stores through the HP_ST_FLAG macro); and the committed .inc files are what the generator writes today -- the
    generator's own checks run with it (hazard distances, wait counts resolved from the instruction order and equal on every
    entry path of a tile body, packed registers consumed before they are rewritten, epilogue operands reloaded behind their last use)."""
    import re
    import subprocess as sp
    sys.path.insert(0, ROOT)
    from motion324_amd import build as B
    asm = B.assembly(["gemm_hp.hip"])["gemm_hp.hip"]
    text = open(asm).read()
    kernels = re.findall(r"\.amdhsa_kernel (\S*gemm_hp\S*)", text)
    assert len(kernels) == 8, kernels
    meta = text[text.index("amdhsa.kernels"):]
    for field, want in ((".vgpr_spill_count", "0"), (".sgpr_spill_count", "0"), (".private_segment_fixed_size", "0")):
        vals = re.findall(re.escape(field) + r":\s*(\d+)", meta)
        assert vals and all(v == want for v in vals), (field, vals)
    assert all(int(v) == 512 for v in re.findall(r"\.vgpr_count:\s*(\d+)", meta))
    assert all(int(v) == 163840 for v in re.findall(r"\.group_segment_fixed_size:\s*(\d+)", meta))      # ring + table + store scratch
    outside, inside, nt = [], False, 0
    for line in text.splitlines():
        if ";;#ASMSTART" in line:
            inside = True
        elif ";;#ASMEND" in line:
            inside = False
        elif not inside and ("v_accvgpr" in line or "v_mfma" in line):
            outside.append(line.strip())
        elif inside and "buffer_store_dwordx4" in line and " nt" in line:
            nt += 1
    assert not outside, outside[:5]
    assert nt == 4 * 64, nt                                               # 16 stores x (2 bodies + 2 tails) in each of the four nt kernels
    csrc = os.path.join(ROOT, "motion324_amd", "csrc")
    names = ["gemm_hp_gelu.inc", "gemm_hp_fold_gelu.inc", "gemm_hp_plain.inc", "gemm_hp_fold.inc", "gemm_hp_clobbers.inc"]
    before = {f: open(os.path.join(csrc, f)).read() for f in names}
    r = sp.run([sys.executable, os.path.join(csrc, "gen_gemm_hp.py")], capture_output=True, text=True, cwd=csrc)
    assert r.returncode == 0, r.stdout + r.stderr
    for f, old in before.items():
        assert open(os.path.join(csrc, f)).read() == old, f"{f} is stale: run csrc/gen_gemm_hp.py"
