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


def test_no_packed_fp32_instruction_consumes_a_fresh_dpp_move():
    """v_mov_b32_dpp + v_pk_add_f32 (what the SLP vectoriser makes of two interleaved `v += dpp(v)` butterflies) returned wrong
    sums on MI355X beside a chunk-ring GEMM on the same CUs (round 3); the build's -fno-slp-vectorize keeps
    every step one v_add_f32_dpp.  tools/audit_dpp.py compiles every translation unit with the build's flags and scans for the pair."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_dpp.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    from motion324_amd import build
    assert "-fno-slp-vectorize" in build.FLAGS and "-fno-slp-vectorize" in build.FLAGS_OF["gemm_ring4.hip"]
