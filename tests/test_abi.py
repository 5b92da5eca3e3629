"""The C-ABI library loads (no GPU needed) and exports exactly what include/m324.h declares."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def declared_symbols():
    text = open(os.path.join(REPO, "include", "m324.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(m324_\w+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "m324_gemm" in syms and "m324_attention" in syms and len(syms) >= 12


def test_library_loads_and_exports_every_declared_symbol():
    from motion324_amd import lib
    h = lib.load()
    for s in declared_symbols():
        assert hasattr(h, s), f"{s} declared in include/m324.h but not exported by libm324.so"
    assert h.m324_abi_version() == lib.ABI_VERSION


def test_binding_table_matches_header():
    from motion324_amd import lib
    assert sorted(lib.SIGNATURES) == declared_symbols()
    # argument counts of the ctypes table agree with the prototypes
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "m324.h")).read(), flags=re.S)
    for name, argtypes in lib.SIGNATURES.items():
        proto = re.search(r"\bint\s+" + name + r"\s*\(([^;]*?)\)\s*;", text, flags=re.S).group(1).strip()
        n = 0 if proto in ("", "void") else len([a for a in proto.split(",") if a.strip()])
        assert n == len(argtypes), (name, n, len(argtypes))


def test_gemm_args_struct_layout_matches_header():
    """Field order/sizes of the ctypes mirror of m324_gemm_args (LP64)."""
    from motion324_amd.lib import GemmArgs
    names = [f[0] for f in GemmArgs._fields_]
    assert names == ["A", "lda", "W", "ldw", "C", "ldc", "M", "N", "K", "in_dtype", "out_dtype", "bias", "act", "gamma",
                     "residual", "ldr", "res_rows", "row_gin", "row_gout", "row_off", "batch", "strideA", "strideW", "strideC", "aux", "ldaux",
                     "aux_mode", "qkv_q", "qkv_k", "qkv_v", "qkv_qw", "qkv_kw", "qkv_eps", "qkv_qscale", "qkv_L", "qkv_H",
                     "ln_rowstat", "ln_colsum", "ln_stats_out", "ln_copy_out", "ln_ldcopy", "ln_ncb", "ln_eps"]
    assert GemmArgs.A.offset == 0 and GemmArgs.M.offset == 48 and GemmArgs.bias.offset == 72
    assert GemmArgs.residual.offset == 96 and GemmArgs.row_gin.offset == 116 and GemmArgs.batch.offset == 128 and GemmArgs.aux.offset == 160 \
        and GemmArgs.aux_mode.offset == 176 and GemmArgs.qkv_q.offset == 184 and GemmArgs.qkv_eps.offset == 224 \
        and GemmArgs.ln_rowstat.offset == 240 and GemmArgs.ln_ldcopy.offset == 272 and GemmArgs.ln_ncb.offset == 280 and GemmArgs.ln_eps.offset == 284 \
        and ctypes.sizeof(GemmArgs) == 288


def test_mirror_item_struct_layout_matches_header():
    """m324_mirror_item (include/m324.h, ABI 21): four longs, four ints."""
    from motion324_amd.lib import MirrorItem
    assert [f[0] for f in MirrorItem._fields_] == ["src_off", "dst_off", "dstT_off", "first_tile", "rows", "cols", "ldT", "pad_"]
    assert MirrorItem.first_tile.offset == 24 and MirrorItem.rows.offset == 32 and MirrorItem.ldT.offset == 40 and ctypes.sizeof(MirrorItem) == 48
    text = open(os.path.join(REPO, "include", "m324.h")).read()
    body = re.search(r"typedef struct m324_mirror_item \{(.*?)\} m324_mirror_item;", text, flags=re.S).group(1)
    assert re.sub(r"\s+", " ", body).strip() == "long src_off, dst_off, dstT_off; long first_tile; int rows, cols, ldT, pad_;"


def test_errors_are_reported_not_thrown_across_the_abi():
    """Argument validation happens before any HIP call, so it is observable without a GPU."""
    from motion324_amd import lib
    h = lib.load()
    rc = h.m324_gemm(None, None)
    assert rc == -1 and "null" in lib.last_error()
    args = lib.GemmArgs()
    args.A, args.W, args.C = 16, 16, 16
    args.M, args.N, args.K, args.lda, args.ldw, args.ldc = 4, 128, 40, 40, 40, 128
    assert h.m324_gemm(ctypes.byref(args), None) == -1 and "K=40" in lib.last_error()
    assert h.m324_attention(16, 0, 16, 16, 16, 64, 1, 1, 4, 4, 0.125, 0, None, 7, None) == -3
    assert h.m324_layernorm(16, 768, 16, None, 1e-5, 16, 768, 0, 4, 770, 0, 0, 0, None) == -1


def test_ops_refuse_cpu_tensors():
    import torch
    from motion324_amd import ops
    from motion324_amd.lib import M324Error
    with pytest.raises(M324Error, match="HIP device"):
        ops.gemm(torch.zeros(4, 64), torch.zeros(128, 64), torch.zeros(4, 128))
