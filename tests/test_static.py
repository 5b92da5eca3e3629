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
