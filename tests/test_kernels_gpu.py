"""GPU parity of every libm324 entry point (called through the C ABI via motion324_amd.ops) against
plain torch fp64/fp32 CPU math of the same op.  fp32 kernels: <= 1e-4 relative (f32 MFMA = fmaf chain);
bf16 kernels: inputs are pre-rounded to bf16 so the only error left is the bf16 rounding of the output
(<= 2^-8 relative per element) and of P in attention -- bands stated per test."""
import math

import pytest
import torch

from conftest import rel_err, vt_layout

pytestmark = pytest.mark.gpu

DEV = "cuda"
DT = [torch.float32, torch.bfloat16]


def _ops():
    from motion324_amd import ops
    return ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


def _q(t, dtype):
    """round to the kernel's operand dtype (and back to fp32 for the reference)"""
    return t.to(dtype).to(torch.float32)


TOL = {torch.float32: 2e-5, torch.bfloat16: 6e-3}


# ------------------------------------------------------------------------------------------- gemm
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 192, 192), (1, 768, 64), (257 * 3, 2304, 768), (2048, 768, 3072),
                                   (130, 576, 256)])
def test_gemm_plain(dtype, M, N, K):
    ops = _ops()
    a, w = _q(_rand((M, K), 1), dtype), _q(_rand((N, K), 2, 0.05), dtype)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), out)
    ref = a.double() @ w.double().T
    assert rel_err(out.float(), ref) < TOL[dtype]
    if dtype == torch.bfloat16:          # element-wise: bf16 rounding of an (almost) exact fp32 sum
        assert torch.allclose(out.float().cpu(), ref.float(), rtol=2 ** -7, atol=2 ** -9 * float(ref.abs().max()))


def test_gemm_transpose_detecting():
    """A = identity-like selector with an asymmetric W catches a swapped C layout (rows <-> cols)."""
    ops = _ops()
    M = N = 128
    K = 128
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / (N * K)
    out = torch.empty((M, N), dtype=torch.float32, device=DEV)
    ops.gemm(a.to(DEV), w.to(DEV), out)
    assert torch.allclose(out.cpu(), a @ w.T, atol=1e-6)


@pytest.mark.parametrize("dtype", DT)
def test_gemm_epilogue_full(dtype):
    """bias -> GELU -> gamma -> residual (broadcast over row groups) -> row remap, fp32 out, in place residual."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    M, N, K = 3 * 50, 192, 128
    a, w = _q(_rand((M, K), 3), dtype), _q(_rand((N, K), 4, 0.1), dtype)
    bias, gamma, res = _rand((N,), 5), 1 + 0.1 * _rand((N,), 6), _rand((50, N), 7)
    gin, gout, off = 50, 53, 2
    out = torch.zeros((3 * 53, N), dtype=torch.float32, device=DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), out, bias=bias.to(DEV), act=ACT_GELU, gamma=gamma.to(DEV),
             residual=res.to(DEV), res_rows=50, row_map=(gin, gout, off))
    v = a.double() @ w.double().T + bias.double()
    v = 0.5 * v * (1 + torch.erf(v / math.sqrt(2.0)))
    v = v * gamma.double() + res.double().repeat(3, 1)
    ref = torch.zeros(3 * 53, N, dtype=torch.float64)
    rows = torch.arange(M)
    ref[(rows // gin) * gout + rows % gin + off] = v
    assert rel_err(out, ref) < (1e-5 if dtype == torch.float32 else 2e-5)   # fp32 accumulate, fp32 out in both modes
    # untouched rows stay zero
    mask = torch.ones(3 * 53, dtype=torch.bool)
    mask[(rows // gin) * gout + rows % gin + off] = False
    assert float(out.cpu()[mask].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DT)
def test_gemm_inplace_residual(dtype):
    ops = _ops()
    M, N, K = 200, 256, 192
    a, w = _q(_rand((M, K), 8), dtype), _q(_rand((N, K), 9, 0.1), dtype)
    x0 = _rand((M, N), 10)
    x = x0.clone().to(DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), x, residual=x)
    assert rel_err(x, x0.double() + a.double() @ w.double().T) < 1e-5


@pytest.mark.parametrize("variant", ["v0", "v1", "v2", "v10", "v11", "v12", "v13"])
@pytest.mark.parametrize("M,N", [(3 * 256, 768), (2 * 230, 328), (40, 192)])
def test_gemm_bf16_residual_stream(tune, variant, M, N):
    """The decoder's bf16 residual stream (bf16 inference): (1) bf16 output + fp32 residual broadcast over row groups (the
    out-projection over the mesh points), (2) x += A W^T + b with x the bf16 OUTPUT ITSELF (the MLP's second Linear in place:
    m324_gemm reads `residual` in the output's dtype exactly when it aliases C).  Interior tiles take the 8-column epilogue,
    ragged ones the 4-column one, small M the skinny / direct kernels: all must agree with fp64 to bf16 rounding."""
    ops = _ops()
    dtype = torch.bfloat16
    K = 192
    if variant != "v0":
        tune("M324_GEMM", variant)
    a, w = _q(_rand((M, K), 81), dtype), _q(_rand((N, K), 82, 0.1), dtype)
    bias = _rand((N,), 83)
    ad, wd, bd = a.to(dtype).to(DEV), w.to(dtype).to(DEV), bias.to(DEV)
    prod = a.double() @ w.double().T + bias.double()
    groups = 2 if M % 2 == 0 else 1
    res = _rand((M // groups, N), 84)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(ad, wd, out, bias=bd, residual=res.to(DEV), res_rows=M // groups)
    assert rel_err(out.float(), prod + res.double().repeat(groups, 1)) < 4e-3
    x0 = _q(_rand((M, N), 85), dtype)
    x = x0.to(dtype).to(DEV)
    ops.gemm(ad, wd, x, bias=bd, residual=x)
    assert rel_err(x.float(), prod + x0.double()) < 4e-3
    with pytest.raises(Exception):                     # a bf16 residual that is NOT the output would be read as fp32: rejected
        ops.gemm(ad, wd, out, residual=x)


@pytest.mark.parametrize("M,N,K", [(2048, 768, 768), (3 * 256 + 37, 256, 128), (100, 512, 192)])
def test_gemm_head_contraction_in_the_epilogue(M, N, K):
    """M324_AUX_N3: gelu(A W^T + b) contracted with a [3, N] weight inside the GEMM epilogue (partial sums per 64-column block)
    + m324_n3_finish == the unfused Linear -> GELU -> Linear(N -> 3) in fp64 (the fused form skips the bf16 rounding of the
    intermediate, so it is compared against the exact value); ragged row counts leave the rows past M untouched."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    dtype = torch.bfloat16
    a, w = _q(_rand((M, K), 95), dtype), _q(_rand((N, K), 96, 0.1), dtype)
    bias, w3, b3 = _rand((N,), 97), _rand((3, N), 98, 0.2), _rand((3,), 99)
    part = torch.full((N // 64, M, 3), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), None, bias=bias.to(DEV), act=ACT_GELU, n3=(w3.to(DEV), part))
    assert bool(torch.isfinite(part).all())
    out = torch.empty((M, 3), dtype=torch.float32, device=DEV)
    ops.n3_finish(part, b3.to(DEV), out)
    v = a.double() @ w.double().T + bias.double()
    g = 0.5 * v * (1 + torch.erf(v / math.sqrt(2.0)))
    ref = g @ w3.double().T + b3.double()
    assert rel_err(out, ref) < 2e-3
    # the unfused path on the same operands (bf16 intermediate) agrees to bf16 rounding
    h2 = torch.empty((M, N), dtype=dtype, device=DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), h2, bias=bias.to(DEV), act=ACT_GELU)
    un = torch.empty((M, 3), dtype=torch.float32, device=DEV)
    ops.linear_n3(h2, w3.to(DEV), b3.to(DEV), un)
    assert rel_err(out, un.cpu().double()) < 4e-3


@pytest.mark.parametrize("rows,C,with_bias", [(77, 768, True), (4 * 19 + 1, 192, False)])
def test_layernorm_bf16_input(rows, C, with_bias):
    """m324_layernorm_in with a bf16 input row (the decoder's bf16 stream): statistics in fp32 on the rounded values."""
    ops = _ops()
    x = _q(_rand((rows, C), 91) * 2 + 0.3, torch.bfloat16)
    w, b = 1 + 0.1 * _rand((C,), 92), (_rand((C,), 93) if with_bias else None)
    out = torch.empty((rows, C), dtype=torch.bfloat16, device=DEV)
    ops.layernorm(x.to(torch.bfloat16).to(DEV), w.to(DEV), None if b is None else b.to(DEV), 1e-5, out)
    ref = torch.nn.functional.layer_norm(x.double(), (C,), w.double(), None if b is None else b.double(), 1e-5)
    assert rel_err(out.float(), ref) < 4e-3
    with pytest.raises(Exception):
        ops.layernorm(x.to(torch.bfloat16).to(DEV), w.to(DEV), None, 1e-5, torch.empty((rows, C), dtype=torch.float32, device=DEV))


@pytest.mark.parametrize("variant", ["v1", "v2", "v5", "v10", "v11", "v12", "v13"])
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("K", [64, 192, 832])
def test_gemm_every_schedule_forced(tune, variant, dtype, K):
    """M324_GEMM=vN (set through m324_set_tunable here) forces one kernel schedule; each must handle ragged M / N tiles, a K shorter than
    its prefetch depth, and the whole epilogue chain, with bf16 and fp32 outputs.  (fp32 operands map v10 / v11 / v12 to v5, v13 to v2;
    the chunk-ring kernels v10 - v13 need two K-stages of 64 and hand K = 64 to v5 / v2.)"""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    tune("M324_GEMM", variant)
    M, N = 3 * 230, 328
    a, w = _q(_rand((M, K), 11), dtype), _q(_rand((N, K), 12, 0.1), dtype)
    bias, gamma, res = _rand((N,), 13), 1 + 0.1 * _rand((N,), 14), _rand((230, N), 15)
    v = a.double() @ w.double().T + bias.double()
    g = 0.5 * v * (1 + torch.erf(v / math.sqrt(2.0)))
    # (1) plain bias + GELU, output in the operand dtype
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), out, bias=bias.to(DEV), act=ACT_GELU)
    assert rel_err(out.float(), g) < TOL[dtype]
    # (2) gamma + broadcast residual + row remap, fp32 out
    gin, gout, off = 230, 233, 1
    out2 = torch.zeros((3 * 233, N), dtype=torch.float32, device=DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), out2, bias=bias.to(DEV), gamma=gamma.to(DEV), residual=res.to(DEV),
             res_rows=230, row_map=(gin, gout, off))
    ref = torch.zeros(3 * 233, N, dtype=torch.float64)
    rows = torch.arange(M)
    ref[(rows // gin) * gout + rows % gin + off] = v * gamma.double() + res.double().repeat(3, 1)
    assert rel_err(out2, ref) < 2e-5
    mask = torch.ones(3 * 233, dtype=torch.bool)
    mask[(rows // gin) * gout + rows % gin + off] = False
    assert float(out2.cpu()[mask].abs().max()) == 0.0
    # (3) in-place residual stream
    x0 = _rand((M, N), 16)
    x = x0.clone().to(DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), x, residual=x)
    assert rel_err(x, x0.double() + a.double() @ w.double().T) < 1e-5


@pytest.mark.parametrize("variant", ["v5", "v10", "v11", "v12", "v13"])
def test_gemm_chunk_ring_many_tiles(tune, variant):
    """The 256-wide kernels on a grid with more tiles than CUs (20 x 16 = 320 tiles of 256 x 256, ragged last row of
    tiles): the persistent v10 / v11 walk 1-2 tiles per workgroup with the next tile's first chunks prefetched under the
    epilogue; bias + GELU into the operand dtype, then the in-place fp32 residual stream.  The default choice must agree bit
    for bit with the forced schedule's summation order (same k order in every tile kernel).  v5 (fp32 operands: the parity
    mode's kernel) is NOT persistent and must get one workgroup per tile: round 3 once launched it on the persistent grid
    and the fp32 parity mode lost every tile past the 256th -- only the full-size golden test noticed."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    dtype = torch.float32 if variant == "v5" else torch.bfloat16
    M, N, K = 20 * 256 - 37, 4096, 192
    a, w = _q(_rand((M, K), 31), dtype), _q(_rand((N, K), 32, 0.1), dtype)
    bias = _rand((N,), 33)
    v = a.double() @ w.double().T + bias.double()
    g = 0.5 * v * (1 + torch.erf(v / math.sqrt(2.0)))
    ad, wd, bd = a.to(dtype).to(DEV), w.to(dtype).to(DEV), bias.to(DEV)
    tune("M324_GEMM", variant)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(ad, wd, out, bias=bd, act=ACT_GELU)
    assert rel_err(out.float(), g) < TOL[dtype]
    x0 = _rand((M, N), 34)
    x = x0.clone().to(DEV)
    ops.gemm(ad, wd, x, bias=bd, residual=x)
    assert rel_err(x, x0.double() + v) < 1e-5
    tune("M324_GEMM", "v2")
    x2 = x0.clone().to(DEV)
    ops.gemm(ad, wd, x2, bias=bd, residual=x2)
    assert torch.equal(x, x2)


@pytest.mark.parametrize("variant", ["v2", "v5", "v10", "v11", "v12", "v13"])
@pytest.mark.parametrize("M,N", [(20 * 256 - 37, 1300), (129, 128), (3 * 256, 7 * 256), (1100, 260)])
def test_gemm_grouped_tile_order_covers_every_tile_once(tune, variant, M, N):
    """M324_XCD=7 forces the 4 x 2 group tile order (gemm_tile.h tile_of; chosen by default only for weights larger than
    an XCD's L2): every output tile must be produced exactly once whatever the tile counts (fewer row tiles than row
    groups, odd column counts, ragged edges), and bit-identical to the row-major order."""
    ops = _ops()
    dtype = torch.bfloat16
    K = 192
    a, w = _q(_rand((M, K), 61), dtype).to(dtype).to(DEV), _q(_rand((N, K), 62, 0.1), dtype).to(dtype).to(DEV)
    tune("M324_GEMM", variant)
    tune("M324_XCD", 1)
    ref = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(a, w, ref)
    assert not bool(torch.isnan(ref.float()).any())
    x0 = _rand((M, N), 63).to(DEV)
    xr = x0.clone()
    ops.gemm(a, w, xr, residual=xr)
    tune("M324_XCD", 7)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(a, w, out)
    assert torch.equal(out, ref)
    x = x0.clone()
    ops.gemm(a, w, x, residual=x)                      # in place: a tile produced twice would add its product twice
    assert torch.equal(x, xr)


@pytest.mark.parametrize("variant", ["v10", "v11", "v12", "v13"])
@pytest.mark.parametrize("M,N,K", [(65, 8, 128), (37, 200, 128), (300, 136, 192), (513, 260, 128), (256, 256, 128), (257, 132, 320)])
def test_gemm_ring_edge_shapes(tune, variant, M, N, K):
    """The chunk-ring kernels at their smallest legal depth (K = 128: two K-stages, the peeled stage 0 plus one loop
    iteration whose look-ahead is clamped back onto the last stage), with fewer rows / columns than one tile, one row or
    column past a tile edge, and N as small as 8: plain bf16 output and the in-place fp32 residual stream."""
    ops = _ops()
    dtype = torch.bfloat16
    tune("M324_GEMM", variant)
    a, w = _q(_rand((M, K), 51), dtype), _q(_rand((N, K), 52, 0.1), dtype)
    ref = a.double() @ w.double().T
    ad, wd = a.to(dtype).to(DEV), w.to(dtype).to(DEV)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(ad, wd, out)
    assert rel_err(out.float(), ref) < TOL[dtype]
    x0 = _rand((M, N), 53)
    x = x0.clone().to(DEV)
    ops.gemm(ad, wd, x, residual=x)
    assert rel_err(x, x0.double() + ref) < 1e-5


@pytest.mark.parametrize("variant,M,N,K", [("v10", 8192, 3072, 768), ("v11", 8192, 3072, 768), ("v12", 10368, 768, 3072),
                                            ("v13", 10368, 768, 768)])
def test_gemm_ring_kernels_are_race_free(tune, variant, M, N, K):
    """The LDS-DMA rings state their own vmcnt waits (tests/test_static.py audits them); a missing one shows up as a tile
    read before it landed -- rarely, and only when the chip is full.  Forty launches at the model's shapes must give
    forty bit-identical results, and the first must match the 128 x 128 kernel (same k order)."""
    ops = _ops()
    dtype = torch.bfloat16
    a = _q(_rand((M, K), 41), dtype).to(dtype).to(DEV)
    w = _q(_rand((N, K), 42, 0.05), dtype).to(dtype).to(DEV)
    tune("M324_GEMM", "v2")
    ref = torch.empty((M, N), dtype=dtype, device=DEV)
    ops.gemm(a, w, ref)
    tune("M324_GEMM", variant)
    outs = [torch.empty((M, N), dtype=dtype, device=DEV) for _ in range(4)]
    for it in range(40):
        ops.gemm(a, w, outs[it % 4])
        if it % 4 == 3:
            for o in outs:
                assert torch.equal(o, ref), f"launch {it}: result differs"


def _plan(ops_mod, a, w, out, **kw):
    """the kernel m324_gemm would launch for this call (m324_gemm_plan through the timing label of ops.gemm)"""
    from motion324_amd import timing
    with timing.Recorder() as rec:
        ops_mod.gemm(a, w, out, **kw)
    torch.cuda.synchronize()
    return " ".join(item[5] for item in rec.items)


@pytest.mark.parametrize("M,N", [(256, 128), (200, 128), (512, 384), (1000, 256), (2304, 1536), (8224, 3072)])
@pytest.mark.parametrize("mode", ["gelu", "bias", "fold_gelu", "fold", "nobias", "nobias_gelu"])
def test_gemm_v15_hand_placed_stream_equals_the_tile_kernels(tune, M, N, mode):
    """Schedule v15 (csrc/gemm_hp.hip: one wave per SIMD, the previous tile's epilogue between the MFMAs of the current tile, stores
    through the wave's LDS scratch) at K = 768: one tile, a ragged single tile, more workgroups than tiles, a ragged last row of
    tiles, several tiles per workgroup (first tile without an epilogue, A / B accumulator sets, exposed tail) -- bias, bias + GELU and
    the LayerNorm-fold consumer forms of both with a merged statistics table.  Same k order, same epilogue arithmetic as the
    128 x 128 kernel: bit-identical; and within bf16 rounding of the fp64 result."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU, ACT_NONE
    K, dtype = 768, torch.bfloat16
    a, w = _q(_rand((M, K), 71), dtype).to(dtype).to(DEV), _q(_rand((N, K), 72, 0.03), dtype).to(dtype).to(DEV)
    bias = _rand((N,), 73).to(DEV)
    rowstat = torch.stack([torch.rand(M) + 0.5, 0.1 * torch.randn(M)], dim=1).contiguous().to(DEV)
    colsum = _rand((N,), 74).to(DEV)
    if "nobias" in mode:
        bias = None
    kw = dict(bias=bias, act=ACT_GELU if "gelu" in mode else ACT_NONE, ln=(rowstat, colsum) if "fold" in mode else None)
    tune("M324_GEMM", "v2")
    ref = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(a, w, ref, **kw)
    tune("M324_GEMM", "v15")
    out = torch.full((M + 3, N), 7.0, dtype=dtype, device=DEV)          # three guard rows behind the ragged tile
    ops.gemm(a, w, out[:M], **kw)
    assert torch.equal(out[:M], ref)
    assert float((out[M:].float() - 7.0).abs().max()) == 0.0              # rows past M are clipped by the store resource
    v = a.double().cpu() @ w.double().cpu().T
    if "fold" in mode:
        v = rowstat[:, :1].double().cpu() * v + rowstat[:, 1:2].double().cpu() * colsum.double().cpu()
    if bias is not None:
        v = v + bias.double().cpu()
    if "gelu" in mode:
        v = 0.5 * v * (1 + torch.erf(v / math.sqrt(2.0)))
    assert rel_err(out[:M].float(), v) < TOL[dtype]


def test_gemm_v15_is_chosen_where_it_was_measured_faster_and_only_there(tune):
    """The chooser sends K = 768 GEMMs with a plain / bias-only bf16 epilogue and at least two 256 x 128 tiles per CU to v15 (M324_HP bit 1,
    default; profiles/r05_gemm_hp.md), the GELU epilogues only with bit 0; what the stream does not build (residual, fp32 output, an
    unmerged statistics table, one tile per workgroup) stays on the round-4 schedules, and a forced v15 falls back to the chooser
    for those instead of failing."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    dtype = torch.bfloat16
    M, N, K = 10368, 3072, 768
    a, w = _q(_rand((M, K), 75), dtype).to(dtype).to(DEV), _q(_rand((N, K), 76, 0.03), dtype).to(dtype).to(DEV)
    bias = _rand((N,), 77).to(DEV)
    out = torch.empty((M, N), dtype=dtype, device=DEV)
    assert "gemm_hp_kernel" in _plan(ops, a, w, out, bias=bias)                        # default M324_HP=6: bias-only / plain epilogues ...
    assert "gemm_hp_kernel" in _plan(ops, a, w, out)                                   # no bias: a bias resource without records
    assert "gemm_hp_kernel" not in _plan(ops, a, w, out, bias=bias, act=ACT_GELU)      # ... GELU from 4096 tiles on (984 here); bit 0: always
    assert "gemm_hp_kernel" not in _plan(ops, a[:2048], w, out[:2048], bias=bias)      # 192 tiles: one per workgroup
    outf = torch.empty((M, N), dtype=torch.float32, device=DEV)
    assert "gemm_hp_kernel" not in _plan(ops, a, w, outf, bias=bias)
    tune("M324_HP", 3)
    assert "gemm_hp_kernel" in _plan(ops, a, w, out, bias=bias, act=ACT_GELU)
    part = torch.rand((K // 64, M, 2), device=DEV)
    colsum = _rand((N,), 78).to(DEV)
    assert "gemm_hp_kernel" not in _plan(ops, a, w, out, bias=bias, act=ACT_GELU, ln=(part, colsum, 1e-5))      # unmerged table
    rowstat = torch.rand((M, 2), device=DEV)
    assert "gemm_hp_kernel" in _plan(ops, a, w, out, bias=bias, act=ACT_GELU, ln=(rowstat, colsum))
    tune("M324_HP", 0)
    assert "gemm_hp_kernel" not in _plan(ops, a, w, out, bias=bias)
    tune("M324_HP", 6)
    big = torch.empty((45056, K), dtype=dtype, device=DEV)                             # 176 x 24 = 4224 tiles
    outb = torch.empty((45056, N), dtype=dtype, device=DEV)
    assert "gemm_hp_kernel" in _plan(ops, big, w, outb, bias=bias, act=ACT_GELU)
    del big, outb
    tune("M324_GEMM", "v15")
    x = torch.zeros((M, N), dtype=torch.float32, device=DEV)
    ops.gemm(a, w, x, residual=x)                                                        # forced, not built: the chooser's kernel runs
    assert rel_err(x, a.double().cpu() @ w.double().cpu().T) < 1e-5


@pytest.mark.parametrize("M", [10368, 65536])
def test_gemm_v15_is_race_free_and_streams_large_outputs(tune, M):
    """Forty launches at the model's fc1 shapes must give forty bit-identical results equal to the 128 x 128 kernel's: the stream states
    its own vmcnt / lgkmcnt waits (gen_gemm_hp.py resolves them from the instruction order) and a missing one shows up as a tile read
    before it landed -- rarely, and only when the chip is full.  M = 65536: the 402-MB output leaves through nontemporal stores."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    dtype = torch.bfloat16
    N, K = 3072, 768
    a = _q(_rand((M, K), 81), dtype).to(dtype).to(DEV)
    w = _q(_rand((N, K), 82, 0.05), dtype).to(dtype).to(DEV)
    bias = _rand((N,), 83).to(DEV)
    tune("M324_GEMM", "v2")
    ref = torch.empty((M, N), dtype=dtype, device=DEV)
    ops.gemm(a, w, ref, bias=bias, act=ACT_GELU)
    tune("M324_GEMM", "v15")
    outs = [torch.empty((M, N), dtype=dtype, device=DEV) for _ in range(2)]
    for it in range(40 if M < 20000 else 12):
        ops.gemm(a, w, outs[it % 2], bias=bias, act=ACT_GELU)
        if it % 2 == 1:
            for o in outs:
                assert torch.equal(o, ref), f"launch {it}: result differs"


@pytest.mark.parametrize("M,N,K", [(64, 768, 3072), (64, 2304, 768), (37, 200, 64), (5, 36, 448)])
def test_gemm_skinny_rows(M, N, K):
    """M <= 64 in bf16 takes the split-K-over-waves kernel (the shape encoder's 64 latent tokens): ragged M and N,
    1..48 chunks per wave, every epilogue stage, both output dtypes."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    dtype = torch.bfloat16
    a, w = _q(_rand((M, K), 21), dtype), _q(_rand((N, K), 22, 0.05), dtype)
    bias, gamma, res = _rand((N,), 23), 1 + 0.1 * _rand((N,), 24), _rand((M, N), 25)
    v = a.double() @ w.double().T + bias.double()
    g = 0.5 * v * (1 + torch.erf(v / math.sqrt(2.0)))
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), out, bias=bias.to(DEV), act=ACT_GELU)
    assert rel_err(out.float(), g) < TOL[dtype]
    out2 = torch.zeros((M + 3, N), dtype=torch.float32, device=DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), out2, bias=bias.to(DEV), gamma=gamma.to(DEV), residual=res.to(DEV),
             row_map=(M, M, 3))
    assert rel_err(out2[3:], v * gamma.double() + res.double()) < 2e-5
    assert float(out2[:3].abs().max()) == 0.0
    x = res.clone().to(DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), x, residual=x)
    assert rel_err(x, res.double() + a.double() @ w.double().T) < 1e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(300, 328, 192), (512, 768, 256), (40, 64, 64)])
def test_gemm_training_aux_operand(dtype, M, N, K):
    """M324_AUX_STORE_PREACT: one launch yields gelu(z) and z; M324_AUX_MUL_GELU_GRAD: result * gelu'(z)
    (ragged tiles, interior 16-byte-store path at N % 64 == 0, and the small-M route around the skinny kernel)."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU, M324Error
    a, w = _q(_rand((M, K), 31), dtype), _q(_rand((N, K), 32, 0.1), dtype)
    bias = _rand((N,), 33)
    zref = a.double() @ w.double().T + bias.double()
    gref = 0.5 * zref * (1 + torch.erf(zref / math.sqrt(2.0)))
    g = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    z = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), g, bias=bias.to(DEV), act=ACT_GELU, preact_out=z)
    assert rel_err(z.float(), zref) < TOL[dtype] and rel_err(g.float(), gref) < TOL[dtype]
    # backward form: dz = (dy @ W2) * gelu'(z) with z as stored above
    dy = _q(_rand((M, K), 34), dtype)
    zs = z.float().cpu().double()
    cdf = 0.5 * (1 + torch.erf(zs / math.sqrt(2.0)))
    pdf = torch.exp(-0.5 * zs * zs) / math.sqrt(2 * math.pi)
    dref = (dy.double() @ w.double().T) * (cdf + zs * pdf)
    dz = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(dy.to(dtype).to(DEV), w.to(dtype).to(DEV), dz, gelu_grad_of=z)
    assert rel_err(dz.float(), dref) < TOL[dtype]
    with pytest.raises(M324Error):          # the pre-activation output needs an activation
        ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), g, preact_out=z)
    # the same pair with erf evaluated once (ABI 22): M324_AUX_STORE_GELU_GRAD leaves gelu'(z) of the fp32 z next to gelu(z), M324_AUX_MUL multiplies
    g2 = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    d = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), g2, bias=bias.to(DEV), act=ACT_GELU, gelu_grad_out=d)
    assert torch.equal(g2, g)
    cdf0 = 0.5 * (1 + torch.erf(zref / math.sqrt(2.0)))
    pdf0 = torch.exp(-0.5 * zref * zref) / math.sqrt(2 * math.pi)
    assert rel_err(d.float(), cdf0 + zref * pdf0) < TOL[dtype]
    dz2 = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(dy.to(dtype).to(DEV), w.to(dtype).to(DEV), dz2, mul_by=d)
    assert rel_err(dz2.float(), (dy.double() @ w.double().T) * d.float().cpu().double()) < TOL[dtype]
    assert rel_err(dz2.float(), (dy.double() @ w.double().T) * (cdf0 + zref * pdf0)) < 2 * TOL[dtype]
    with pytest.raises(M324Error):
        ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), g, gelu_grad_out=d)
    with pytest.raises(M324Error):
        ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), g, act=ACT_GELU, mul_by=d)


@pytest.mark.parametrize("M,N,Kc,slices", [(64, 128, 128, 1), (150, 192, 328, 1), (1000, 768, 64, 3), (4096, 2304, 768, 4),
                                           (700, 8, 3072, 2), (65, 136, 72, 9), (1024, 256, 256, 2), (8192, 768, 3072, 5),
                                           (2080, 512, 256, 3)])
def test_gemm_tn_weight_gradient(M, N, Kc, slices):
    """m324_gemm_tn: dW = dY^T A straight from token-major bf16 operands (transposing LDS reads), ragged tiles in every
    dimension, a token count that is not a multiple of the 64-row stage, split-K slices with a short last slice."""
    ops = _ops()
    dtype = torch.bfloat16
    x, y = _q(_rand((M, N), 71), dtype), _q(_rand((M, Kc), 72, 0.3), dtype)
    out = ops.gemm_tn(x.to(dtype).to(DEV), y.to(dtype).to(DEV), slices)
    ref = x.double().T @ y.double()
    assert out.shape == (N, Kc) and torch.isfinite(out).all()
    assert rel_err(out, ref) < 2e-5                       # bf16 operands, fp32 accumulation: exact products
    # strided operands (column slices of a wider buffer, as dqkv / qkv are)
    wide = _q(_rand((M, N + 64), 73), dtype).to(dtype).to(DEV)
    out2 = ops.gemm_tn(wide[:, 64:], y.to(dtype).to(DEV), slices)
    assert rel_err(out2, wide[:, 64:].float().cpu().double().T @ y.double()) < 2e-5


@pytest.mark.parametrize("M,N,Kc,slices", [(31104, 768, 768, 28), (8192, 2304, 768, 9), (4096, 768, 3072, 7), (1000, 136, 72, 5)])
def test_gemm_tn_does_not_depend_on_where_a_tile_runs(tune, M, N, Kc, slices):
    """Round 6: the weight-gradient kernels take a flat grid of (slice, tile) items, a contiguous slice-major range per XCD (M324_XCD
    bit 0).  Which XCD computes a tile must not change a single bit of it: same partial sums with the old dispatch order."""
    ops = _ops()
    dtype = torch.bfloat16
    x, y = _q(_rand((M, N), 76), dtype).to(dtype).to(DEV), _q(_rand((M, Kc), 77, 0.3), dtype).to(dtype).to(DEV)
    flat = ops.gemm_tn(x, y, slices)
    tune("M324_XCD", "2")
    plain = ops.gemm_tn(x, y, slices)
    assert torch.equal(flat, plain)
    assert rel_err(flat, x.float().cpu().double().T @ y.float().cpu().double()) < 2e-5


def test_gemm_tn_tile_kernels_agree(tune):
    """The 256 x 256 pipelined TN kernel (default where it applies) and the 128 x 128 kernel (M324_GEMM_TN=128) compute
    the same sums over the same slices: only the order inside a slice differs."""
    ops = _ops()
    dtype = torch.bfloat16
    M, N, Kc = 12288, 768, 3072
    x, y = _q(_rand((M, N), 74), dtype).to(dtype).to(DEV), _q(_rand((M, Kc), 75, 0.3), dtype).to(dtype).to(DEV)
    big = ops.gemm_tn(x, y, 4)
    tune("M324_GEMM_TN", "128")
    small = ops.gemm_tn(x, y, 4)
    assert rel_err(big, small) < 1e-6
    assert rel_err(big, x.float().cpu().double().T @ y.float().cpu().double()) < 2e-5


@pytest.mark.parametrize("B,L,H,norm,bias", [(2, 100, 3, True, False), (32, 324, 12, True, False), (3, 257, 12, False, True),
                                            (1, 2100, 4, True, False)])
def test_gemm_qkv_heads_epilogue(B, L, H, norm, bias):
    """M324_AUX_QKV_HEADS: the fused q|k|v projection written head-major with per-head RMSNorm and the q pre-scale ==
    m324_gemm followed by m324_qkv_split (both round q, k, v to bf16 once, from fp32); row-major V feeds m324_attention
    with M324_ATTN_V_ROWMAJOR.  Ragged batches (L not a multiple of the 32-row blocks), DINO's bias-without-norm form."""
    ops = _ops()
    dtype = torch.bfloat16
    C, K = H * 64, 192
    x = _q(_rand((B * L, K), 91), dtype).to(dtype).to(DEV)
    w = _q(_rand((3 * C, K), 92, 0.1), dtype).to(dtype).to(DEV)
    b = _rand((3 * C,), 93).to(DEV) if bias else None
    qw, kw = ((1 + 0.1 * _rand((64,), 94)).to(DEV), (1 + 0.1 * _rand((64,), 95)).to(DEV)) if norm else (None, None)
    Q, Kk, V = (torch.full((B, H, L, 64), float("nan"), dtype=dtype, device=DEV) for _ in range(3))
    ops.gemm(x, w, None, bias=b, qkv_heads=(Q, Kk, V, qw, kw, 1e-5, ops.Q_PRESCALE, L, H))
    # reference in fp64 from the same bf16 operands
    y = x.float().cpu().double() @ w.float().cpu().double().T + (b.cpu().double() if bias else 0.0)
    q, k, v = (t.reshape(B, L, H, 64).permute(0, 2, 1, 3) for t in y.chunk(3, dim=-1))
    if norm:
        q = q * torch.rsqrt((q * q).mean(-1, keepdim=True) + 1e-5) * qw.cpu().double()
        k = k * torch.rsqrt((k * k).mean(-1, keepdim=True) + 1e-5) * kw.cpu().double()
    q = q * ops.Q_PRESCALE
    for got, ref in ((Q, q), (Kk, k), (V, v)):
        assert torch.isfinite(got.float()).all()
        assert rel_err(got.float(), ref) < 4e-3
    # attention on the fused outputs == attention on the two-pass outputs (up to one bf16 rounding of qkv in between)
    out_f = torch.empty((B * L, C), dtype=dtype, device=DEV)
    ops.attention(Q, Kk, V, out_f, prescaled=True, v_rowmajor=True)
    sc = torch.einsum("bhqd,bhkd->bhqk", q, k) * math.log(2.0)
    ref_o = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc, dim=-1), v).reshape(B * L, C)
    assert rel_err(out_f.float(), ref_o) < 1e-2


@pytest.mark.parametrize("variant", [None, "v2", "v10", "v11", "v13"])
def test_gemm_qkv_heads_transposed_v_epilogue(tune, variant):
    """M324_AUX_QKV_HEADS_VT: as the head-major epilogue, but V leaves as the transposed, key-permuted Vt the default
    attention kernel reads.  Vt must equal m324_gemm + m324_qkv_split bit for bit (both round acc + bias to bf16 once);
    Q / K differ by the one bf16 rounding the two-pass form puts before the RMSNorm.  Every tile kernel's ACT = 4 path."""
    ops = _ops()
    dtype = torch.bfloat16
    B, L, H, K = 2, 384, 4, 192
    C = H * 64
    x = _q(_rand((B * L, K), 191), dtype).to(dtype).to(DEV)
    w = _q(_rand((3 * C, K), 192, 0.1), dtype).to(dtype).to(DEV)
    b = _rand((3 * C,), 193).to(DEV)
    qw, kw = (1 + 0.1 * _rand((64,), 194)).to(DEV), (1 + 0.1 * _rand((64,), 195)).to(DEV)
    qkv = torch.empty((B * L, 3 * C), dtype=dtype, device=DEV)
    ops.gemm(x, w, qkv, bias=b)
    Q2, K2, Vt2 = ops.qkv_split(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], qw, kw, 1e-5, B, L, H, dtype, q_scale=ops.Q_PRESCALE)
    if variant:
        tune("M324_GEMM", variant)
    Q, Kk = (torch.full((B, H, L, 64), float("nan"), dtype=dtype, device=DEV) for _ in range(2))
    Vt = torch.full((B, H, 64, L), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(x, w, None, bias=b, qkv_heads=(Q, Kk, Vt, qw, kw, 1e-5, ops.Q_PRESCALE, L, H))
    assert torch.equal(Vt, Vt2)
    assert rel_err(Q.float(), Q2.float().double()) < 6e-3 and rel_err(Kk.float(), K2.float().double()) < 6e-3
    out_f, out_2 = (torch.empty((B * L, C), dtype=dtype, device=DEV) for _ in range(2))
    ops.attention(Q, Kk, Vt, out_f, prescaled=True)
    ops.attention(Q2, K2, Vt2, out_2, prescaled=True)
    assert rel_err(out_f.float(), out_2.float().double()) < 1e-2
    from motion324_amd.lib import M324Error
    with pytest.raises(M324Error, match="64"):         # L = 200: a 32-token block would straddle the batches
        ops.gemm(x[:400], w, None, bias=b, qkv_heads=(Q[:, :, :200].contiguous(), Kk[:, :, :200].contiguous(),
                                                       torch.empty((2, H, 64, 200), dtype=dtype, device=DEV), qw, kw, 1e-5, 1.0, 200, H))


def test_gemm_rejects_bad_k():
    ops = _ops()
    from motion324_amd.lib import M324Error
    a = torch.zeros((8, 40), dtype=torch.float32, device=DEV)
    w = torch.zeros((128, 40), dtype=torch.float32, device=DEV)
    with pytest.raises(M324Error, match="K=40"):
        ops.gemm(a, w, torch.empty((8, 128), dtype=torch.float32, device=DEV))


# ------------------------------------------------------------------------------------------- layernorm
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C,with_bias,eps", [(768, False, 1e-5), (768, True, 1e-6), (192, True, 1e-5)])
def test_layernorm(dtype, C, with_bias, eps):
    ops = _ops()
    rows = 77
    x = _rand((rows, C), 11) * 3 + 0.5
    w, b = 1 + 0.1 * _rand((C,), 12), (_rand((C,), 13) if with_bias else None)
    out = torch.empty((rows, C), dtype=dtype, device=DEV)
    ops.layernorm(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), eps, out)
    ref = torch.nn.functional.layer_norm(x.double(), (C,), w.double(), None if b is None else b.double(), eps)
    assert rel_err(out.float(), ref) < (1e-6 if dtype == torch.float32 else 4e-3)


def test_layernorm_row_gather():
    ops = _ops()
    C, gin, gout, off = 192, 8, 20, 4
    x = _rand((3 * gout, C), 14)
    w = 1 + 0.1 * _rand((C,), 15)
    out = torch.empty((3 * gin, C), dtype=torch.float32, device=DEV)
    ops.layernorm(x.to(DEV), w.to(DEV), None, 1e-5, out, row_map=(gin, gout, off))
    sel = x.reshape(3, gout, C)[:, off:off + gin].reshape(-1, C)
    ref = torch.nn.functional.layer_norm(sel.double(), (C,), w.double(), None, 1e-5)
    assert rel_err(out, ref) < 1e-6


@pytest.mark.parametrize("rows_per_wave", [2, 1])
def test_reductions_give_the_solo_result_beside_a_chunk_ring_gemm(tune, rows_per_wave):
    """Dynamic guard for the round-3 wrong-result hazard (v_mov_b32_dpp + v_pk_add_f32 returned ulp-shifted row sums whenever a
    chunk-ring GEMM shared the CU; the build's -fno-slp-vectorize removed the pair, tools/audit_dpp.py scans for it statically):
    200 LayerNorm launches and 100 row-statistics launches on one stream while a second stream runs chunk-ring GEMMs (v10, then
    v13) -- every result must equal the solo result bit for bit, whatever the compiler version makes of the DPP reductions."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    tune("M324_LN_ROWS", rows_per_wave)
    M, C = 1285, 768
    x = _rand((M, C), 301, 1.3).to(DEV) + 0.2
    w, b = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    a2 = _rand((1028, 768), 302).to(torch.bfloat16).to(DEV)
    w2 = _rand((3072, 768), 303, 0.02).to(torch.bfloat16).to(DEV)
    o2 = torch.empty(1028, 3072, device=DEV, dtype=torch.bfloat16)
    bias2 = torch.randn(3072, device=DEV)
    ref = torch.empty(M, C, device=DEV, dtype=torch.bfloat16)
    ops.layernorm(x, w, b, 1e-6, ref)
    st_ref = torch.empty((M, 2), dtype=torch.float32, device=DEV)
    ops.rowstats(x, 1e-6, st_ref, None)
    torch.cuda.synchronize()
    for sched in ("v10", "v13"):
        tune("M324_GEMM", sched)
        side = torch.cuda.Stream()
        outs = [torch.empty(M, C, device=DEV, dtype=torch.bfloat16) for _ in range(200)]
        stats = [torch.empty((M, 2), dtype=torch.float32, device=DEV) for _ in range(100)]
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(400):
                ops.gemm(a2, w2, o2, bias=bias2, act=ACT_GELU)
        for i, o in enumerate(outs):
            ops.layernorm(x, w, b, 1e-6, o)
            if i < len(stats):
                ops.rowstats(x, 1e-6, stats[i], None)
        torch.cuda.synchronize()
        bad = sum(int(not torch.equal(o, ref)) for o in outs) + sum(int(not torch.equal(t, st_ref)) for t in stats)
        assert bad == 0, f"{bad} of 300 reduction results beside {sched} differ from the solo result"


# ------------------------------------------------------------------------------------------- qkv split
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,L,H,norm", [(2, 100, 3, True), (1, 324, 12, True), (3, 257, 12, False), (2, 64, 2, True)])
def test_qkv_split(dtype, B, L, H, norm):
    ops = _ops()
    C = H * 64
    qkv = _q(_rand((B * L, 3 * C), 16), dtype)
    qw, kw = 1 + 0.1 * _rand((64,), 17), 1 + 0.1 * _rand((64,), 18)
    d = qkv.to(dtype).to(DEV)
    Q, K, Vt = ops.qkv_split(d[:, :C], d[:, C:2 * C], d[:, 2 * C:], qw.to(DEV) if norm else None,
                             kw.to(DEV) if norm else None, 1e-5, B, L, H, dtype)
    q, k, v = (t.reshape(B, L, H, 64).double() for t in qkv.chunk(3, dim=-1))
    if norm:
        q = q * torch.rsqrt((q * q).mean(-1, keepdim=True) + 1e-5) * qw.double()
        k = k * torch.rsqrt((k * k).mean(-1, keepdim=True) + 1e-5) * kw.double()
    tol = 1e-6 if dtype == torch.float32 else 4e-3
    assert rel_err(Q.float(), q.permute(0, 2, 1, 3)) < tol
    assert rel_err(K.float(), k.permute(0, 2, 1, 3)) < tol
    Q2, _, _ = ops.qkv_split(d[:, :C], None, None, qw.to(DEV) if norm else None, None, 1e-5, B, L, H, dtype, q_scale=0.25)
    assert rel_err(Q2.float(), 0.25 * q.permute(0, 2, 1, 3)) < tol        # power-of-two scale: same rounding
    Lp = (L + 63) // 64 * 64
    assert Vt.shape == (B, H, 64, Lp)
    assert torch.equal(Vt.float().cpu(), vt_layout(v.permute(0, 2, 1, 3).float()))   # pure data movement: exact


# ------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, scale):
    s = torch.einsum("bhqd,bhkd->bhqk", q.double(), k.double()) * scale
    return torch.einsum("bhqk,bhkd->bqhd", torch.softmax(s, dim=-1), v.double())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 2, 64, 64), (2, 3, 100, 100), (1, 12, 324, 324), (2, 12, 257, 257), (1, 2, 64, 4096),
                                       (1, 3, 1000, 70), (1, 1, 1, 1), (1, 2, 129, 65)])
def test_attention(dtype, B, H, Lq, Lk):
    ops = _ops()
    q, k, v = (_q(_rand((B, H, L, 64), s, sc), dtype) for L, s, sc in ((Lq, 19, 1.5), (Lk, 20, 1.5), (Lk, 21, 1.0)))
    vt = vt_layout(v)
    out = torch.full((B * Lq, H * 64), float("nan"), dtype=dtype, device=DEV)
    ops.attention(q.to(dtype).to(DEV), k.to(dtype).to(DEV), vt.to(dtype).to(DEV), out)
    ref = _attn_ref(q, k, v, 64 ** -0.5).reshape(B * Lq, H * 64)
    assert torch.isfinite(out.float()).all()
    assert rel_err(out.float(), ref) < (1e-5 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 2, 200, 333), (2, 3, 324, 324), (1, 1, 40, 1000)])
def test_attention_prescaled_q(dtype, B, H, Lq, Lk):
    """Q carrying scale*log2(e) (m324_qkv_split q_scale) + q_prescaled=1 == plain softmax attention."""
    ops = _ops()
    q, k, v = (_rand((B, H, L, 64), s, 1.5) for L, s in ((Lq, 51), (Lk, 52), (Lk, 53)))
    k, v = _q(k, dtype), _q(v, dtype)
    qs = _q(q * ops.Q_PRESCALE, dtype)                       # what qkv_split would store
    vt = vt_layout(v)
    out = torch.empty((B * Lq, H * 64), dtype=dtype, device=DEV)
    ops.attention(qs.to(dtype).to(DEV), k.to(dtype).to(DEV), vt.to(dtype).to(DEV), out, prescaled=True)
    sc = torch.einsum("bhqd,bhkd->bhqk", qs.double(), k.double()) * math.log(2.0)     # exp2(x) = exp(x ln 2)
    ref = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc, dim=-1), v.double()).reshape(B * Lq, H * 64)
    assert rel_err(out.float(), ref) < (1e-5 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("dtype", DT)
def test_attention_lazy_max_growth(dtype):
    """Scores that keep growing along the key axis force the lazy reference maximum to move many times,
    by more and by less than its threshold, and a block of very negative scores must not disturb it."""
    ops = _ops()
    B, H, L = 1, 1, 640
    q = _rand((B, H, 96, 64), 54)
    k = _rand((B, H, L, 64), 55)
    ramp = torch.linspace(0.2, 6.0, L)                       # later keys align more and more with q
    k = k * 0.3 + q[0, 0, 5][None, None, None, :] * ramp[None, None, :, None] / 8
    k[0, 0, 200:264] = -k[0, 0, 200:264] * 3                # one tile of strongly negative scores for q5
    v = _rand((B, H, L, 64), 56)
    q, k, v = _q(q, dtype), _q(k, dtype), _q(v, dtype)
    out = torch.empty((96, 64), dtype=dtype, device=DEV)
    ops.attention(q.to(dtype).to(DEV), k.to(dtype).to(DEV), vt_layout(v).to(dtype).to(DEV), out)
    ref = _attn_ref(q, k, v, 64 ** -0.5).reshape(96, 64)
    assert torch.isfinite(out.float()).all()
    assert rel_err(out.float(), ref) < (1e-5 if dtype == torch.float32 else 8e-3)
    assert rel_err(out.float()[5], ref[5]) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", DT)
def test_attention_shared_q(dtype):
    """decoder pattern: one query set, per-frame K/V of 64 latent tokens (reference Pcd_motion.py:539-560)."""
    ops = _ops()
    T, H, Lq, Lk = 5, 3, 200, 64
    q, k, v = _q(_rand((1, H, Lq, 64), 22), dtype), _q(_rand((T, H, Lk, 64), 23), dtype), _q(_rand((T, H, Lk, 64), 24), dtype)
    out = torch.empty((T * Lq, H * 64), dtype=dtype, device=DEV)
    ops.attention(q.to(dtype).to(DEV), k.to(dtype).to(DEV), vt_layout(v).to(dtype).to(DEV), out, shared_q=True)
    ref = _attn_ref(q.expand(T, -1, -1, -1), k, v, 64 ** -0.5).reshape(T * Lq, H * 64)
    assert rel_err(out.float(), ref) < (1e-5 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("T,H,Lq,Lk", [(8, 3, 2048, 64), (4, 2, 700, 64), (6, 1, 513, 37), (32, 12, 2048, 64)])
def test_attention_shared_q_frame_loop(tune, T, H, Lq, Lk):
    """The decoder's cross-attention kernel (one query set in registers, two frames per workgroup, frame j + 1's K / Vt in flight
    under frame j): ragged query tiles, fewer than 64 keys, the LSE, and the SAME bits as the one-workgroup-per-frame form
    (M324_ATTN_EXP bit 3), whose arithmetic it repeats.  Reference Pcd_motion.py:539-560."""
    ops = _ops()
    dtype = torch.bfloat16
    q, k, v = _rand((1, H, Lq, 64), 122, 1.5), _rand((T, H, Lk, 64), 123, 1.5), _rand((T, H, Lk, 64), 124)
    k[3 % T, 0, Lk // 2] = q[0, 0, 11] * 3.0
    k, v = _q(k, dtype), _q(v, dtype)
    qs = _q(q * ops.Q_PRESCALE, dtype)
    qd, kd, vd = qs.to(dtype).to(DEV), k.to(dtype).to(DEV), vt_layout(v).to(dtype).to(DEV)
    outs, lses = [], []
    for bits in ("0", "8"):
        tune("M324_ATTN_EXP", bits)
        plan = ops._attn_plan(T, H, Lq, Lk, 1 | 256, ops.code_of(dtype))
        assert ("attn_frames_kernel" in plan) == (bits == "0"), plan
        out = torch.full((T * Lq, H * 64), float("nan"), dtype=dtype, device=DEV)
        lse = torch.full((T, H, Lq), float("nan"), dtype=torch.float32, device=DEV)
        ops.attention(qd, kd, vd, out, prescaled=True, shared_q=True, lse=lse)
        outs.append(out), lses.append(lse)
    sc = torch.einsum("bhqd,bhkd->bhqk", qs.expand(T, -1, -1, -1).double(), k.double())
    ref = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc * math.log(2.0), dim=-1), v.double()).reshape(T * Lq, H * 64)
    assert torch.isfinite(outs[0].float()).all()
    assert rel_err(outs[0].float(), ref) < 8e-3
    assert (lses[0].double().cpu() - torch.logsumexp(sc * math.log(2.0), dim=-1) / math.log(2.0)).abs().max() < 2e-3
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(lses[0], lses[1])


@pytest.mark.parametrize("dtype", DT)
def test_attention_online_softmax_rescale(dtype):
    """A late key with a much larger score forces the running-max rescale of every earlier tile."""
    ops = _ops()
    B, H, L = 1, 1, 320
    q, k, v = _q(_rand((B, H, L, 64), 25), dtype), _q(_rand((B, H, L, 64), 26), dtype), _q(_rand((B, H, L, 64), 27), dtype)
    k[0, 0, 300] = _q(q[0, 0, 7] * 4.0, dtype)           # q7 . k300 >> everything else, in the 5th tile
    out = torch.empty((L, 64), dtype=dtype, device=DEV)
    ops.attention(q.to(dtype).to(DEV), k.to(dtype).to(DEV), vt_layout(v).to(dtype).to(DEV), out)
    ref = _attn_ref(q, k, v, 64 ** -0.5).reshape(L, 64)
    assert rel_err(out.float(), ref) < (1e-5 if dtype == torch.float32 else 8e-3)
    assert rel_err(out.float()[7], ref[7]) < (1e-5 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("sched", ["nw4", "nw8"])
@pytest.mark.parametrize("Lq,Lk,odd", [(2100, 2100, False), (2304, 1088, True), (2048, 640, False)])
def test_attention_long_sequence_schedules(tune, sched, Lq, Lk, odd):
    """The long-sequence forward schedules (4- and 8-wave workgroups: M324_ATTN_NW) on ragged lengths, odd
    and even tile counts, prescaled Q, late / early dominant keys that force the lazy reference to move, and the LSE."""
    ops = _ops()
    dtype = torch.bfloat16
    tune("M324_ATTN_NW", "4" if sched == "nw4" else "8")
    B, H = 1, 2
    q, k, v = (_rand((B, H, L, 64), s_, 1.5) for L, s_ in ((Lq, 61), (Lk, 62), (Lk, 63)))
    k[0, 0, Lk - 70] = q[0, 0, 9] * 3.0                       # q9 . k >> everything, second-to-last tile
    k[0, 1, 130] = q[0, 1, 300] * 3.0                         # and an early one for the other head
    k, v = _q(k, dtype), _q(v, dtype)
    qs = _q(q * ops.Q_PRESCALE, dtype)
    out = torch.full((B * Lq, H * 64), float("nan"), dtype=dtype, device=DEV)
    lse = torch.full((B, H, Lq), float("nan"), dtype=torch.float32, device=DEV)
    ops.attention(qs.to(dtype).to(DEV), k.to(dtype).to(DEV), vt_layout(v).to(dtype).to(DEV), out, prescaled=True, lse=lse)
    sc = torch.einsum("bhqd,bhkd->bhqk", qs.double(), k.double())                      # log2-domain scores
    ref = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc * math.log(2.0), dim=-1), v.double()).reshape(B * Lq, H * 64)
    assert torch.isfinite(out.float()).all()
    assert rel_err(out.float(), ref) < 8e-3
    assert rel_err(out.float()[9], ref[9]) < 1e-2
    lse_ref = torch.logsumexp(sc * math.log(2.0), dim=-1) / math.log(2.0)
    assert float((lse.cpu().double() - lse_ref).abs().max()) < 2e-2


@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 2, 2048, 512), (1, 2, 2100, 2048), (1, 2, 2304, 1088), (2, 3, 2049, 576), (1, 1, 4096, 64 * 37),
                                       (1, 2, 2100, 2100), (1, 2, 2304, 1025), (2, 3, 2049, 639), (1, 1, 3888, 3888), (1, 2, 2048, 513)])
@pytest.mark.parametrize("spike", [True, False])
def test_attention_one_wave_per_simd_stream(tune, B, H, Lq, Lk, spike):
    """attention_pwg.hip (the chooser's pick for long sequences): hand-placed software pipeline over the key tiles.  Tile counts
    8, 32, 17, 9, 37, 33, 10, 61 (every loop body as the last one, both tails), ragged query counts, RAGGED KEY counts (1, 52,
    63, 48 valid keys in the last tile: masked before the last vote), batches, dominant keys late and early (the lazy reference
    moves inside the pipeline, with P.V of the previous tile in flight -- also in the masked last tile), the LSE -- against fp64
    softmax attention, and against the eight-wave kernel (M324_ATTN_PWG=0) on the same operands."""
    ops = _ops()
    dtype = torch.bfloat16
    assert "attn_pwg_kernel" in ops._attn_plan(B, H, Lq, Lk, 1, ops.code_of(dtype))
    q, k, v = (_rand((B, H, L, 64), s_, 1.5) for L, s_ in ((Lq, 161), (Lk, 162), (Lk, 163)))
    if spike:
        k[0, 0, Lk - 70] = q[0, 0, 9] * 3.0                       # q9 . k >> everything, second-to-last tile
        k[0, H - 1, 130] = q[0, H - 1, 300] * 3.0                 # an early one
        k[B - 1, 0, 64 * 3 + 5] = q[B - 1, 0, Lq - 1] * 2.5       # the last (possibly ragged) query row
    k, v = _q(k, dtype), _q(v, dtype)
    qs = _q(q * ops.Q_PRESCALE, dtype)
    dq, dk, dvt = qs.to(dtype).to(DEV), k.to(dtype).to(DEV), vt_layout(v).to(dtype).to(DEV)
    res = {}
    for pwg in ("1", "0"):
        tune("M324_ATTN_PWG", pwg)
        out = torch.full((B * Lq + 3, H * 64), float("nan"), dtype=dtype, device=DEV)     # three guard rows behind the output
        lse = torch.full((B, H, Lq), float("nan"), dtype=torch.float32, device=DEV)
        ops.attention(dq, dk, dvt, out, prescaled=True, lse=lse)
        assert torch.isnan(out[B * Lq:].float()).all()                                     # rows past Lq are never stored
        res[pwg] = (out[:B * Lq].float().cpu(), lse.cpu())
    sc = torch.einsum("bhqd,bhkd->bhqk", qs.double(), k.double())                          # log2-domain scores
    ref = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc * math.log(2.0), dim=-1), v.double()).reshape(B * Lq, H * 64)
    out, lse = res["1"]
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < 8e-3
    assert rel_err(out[9], ref[9]) < 1e-2 and rel_err(out[B * Lq - 1], ref[B * Lq - 1]) < 1e-2
    lse_ref = torch.logsumexp(sc * math.log(2.0), dim=-1) / math.log(2.0)
    assert float((lse.double() - lse_ref).abs().max()) < 2e-2
    assert rel_err(out, res["0"][0]) < 6e-3                       # same mathematics, different summation order


@pytest.mark.parametrize("B,H,Lq,Lk,amp", [(1, 2, 2048, 512, 1.5), (1, 2, 2100, 2048, 1.5), (2, 3, 2049, 576, 2.2), (1, 1, 4096, 64 * 37, 1.0),
                                           (1, 2, 2100, 2100, 1.5), (2, 3, 2049, 513, 1.5), (1, 1, 3888, 3888, 1.0)])
def test_attention_bounded_scores_stream(tune, B, H, Lq, Lk, amp):
    """M324_ATTN_SCORES_BOUNDED: the long-sequence kernel without a reference maximum (exp2, sum, pack only).  Same softmax as the
    lazy-maximum stream and as fp64 attention for scores inside the vouched range -- here up to |s| ~ 40 in the log2 domain
    (amp 2.2) -- and the same log2-domain LSE."""
    ops = _ops()
    dtype = torch.bfloat16
    assert "attn_pwg_bounded_kernel" in ops._attn_plan(B, H, Lq, Lk, 1 | 4, ops.code_of(dtype))
    q, k, v = (_rand((B, H, L, 64), s_, amp) for L, s_ in ((Lq, 181), (Lk, 182), (Lk, 183)))
    k, v = _q(k, dtype), _q(v, dtype)
    qs = _q(q * ops.Q_PRESCALE, dtype)
    sc = torch.einsum("bhqd,bhkd->bhqk", qs.double(), k.double())                          # log2-domain scores
    assert float(sc.abs().max()) < 60.0
    dq, dk, dvt = qs.to(dtype).to(DEV), k.to(dtype).to(DEV), vt_layout(v).to(dtype).to(DEV)
    res = {}
    for bounded in (True, False):
        out = torch.full((B * Lq, H * 64), float("nan"), dtype=dtype, device=DEV)
        lse = torch.full((B, H, Lq), float("nan"), dtype=torch.float32, device=DEV)
        ops.attention(dq, dk, dvt, out, prescaled=True, lse=lse, bounded=bounded)
        res[bounded] = (out.float().cpu(), lse.cpu())
    ref = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc * math.log(2.0), dim=-1), v.double()).reshape(B * Lq, H * 64)
    lse_ref = torch.logsumexp(sc * math.log(2.0), dim=-1) / math.log(2.0)
    out, lse = res[True]
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < 8e-3
    assert float((lse.double() - lse_ref).abs().max()) < 2e-2
    assert rel_err(out, res[False][0]) < 6e-3


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,Lq,cuts,bounded", [(1, 2, 2100, (0, 700, 1500, 2304), False), (1, 3, 2304, (0, 1152, 2304), True),
                                                  (2, 2, 300, (0, 64, 200, 333), False), (1, 1, 2048, (0, 520, 2048 + 520), True)])
def test_attention_merge_of_key_parts_equals_attention_over_all_keys(dtype, B, H, Lq, cuts, bounded):
    """m324_attention_merge: the keys are attended in two or three disjoint ranges (each attention leaves its normalised output and
    the log2-domain log-sum-exp of its rows); the merged result must be the attention over all keys -- against fp64 softmax
    attention and against ONE m324_attention call on the same operands.  Long (the one-wave-per-SIMD streams, lazy and bounded)
    and short parts, ragged part lengths, a dominant key in one part only (the other parts' weights underflow to ~0)."""
    ops = _ops()
    Lk = cuts[-1]
    q, k, v = (_rand((B, H, L, 64), s_, 1.2) for L, s_ in ((Lq, 261), (Lk, 262), (Lk, 263)))
    if not bounded:
        k[0, 0, cuts[1] + 3] = q[0, 0, 7] * 3.0                   # q7's softmax lives in the second part
    k, v = _q(k, dtype), _q(v, dtype)
    qs = _q(q * ops.Q_PRESCALE, dtype)
    dq = qs.to(dtype).to(DEV)
    flag = bounded and dtype == torch.bfloat16
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        o = torch.full((B * Lq, H * 64), float("nan"), dtype=dtype, device=DEV)
        lse = torch.full((B, H, Lq), float("nan"), dtype=torch.float32, device=DEV)
        ops.attention(dq, k[:, :, a:b].contiguous().to(dtype).to(DEV), vt_layout(v[:, :, a:b]).to(dtype).to(DEV), o, prescaled=True, lse=lse,
                      bounded=flag)
        parts.append((o, lse))
    out = torch.full((B * Lq + 2, H * 64), float("nan"), dtype=dtype, device=DEV)
    ops.attention_merge(parts, out, B, H, Lq)
    assert torch.isnan(out[B * Lq:].float()).all()
    whole = torch.empty((B * Lq, H * 64), dtype=dtype, device=DEV)
    ops.attention(dq, k.to(dtype).to(DEV), vt_layout(v).to(dtype).to(DEV), whole, prescaled=True, bounded=flag)
    sc = torch.einsum("bhqd,bhkd->bhqk", qs.double(), k.double())
    ref = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc * math.log(2.0), dim=-1), v.double()).reshape(B * Lq, H * 64)
    got = out[:B * Lq].float().cpu()
    tol = 8e-3 if dtype == torch.bfloat16 else 2e-5
    assert torch.isfinite(got).all()
    assert rel_err(got, ref) < tol and rel_err(got[7], ref[7]) < 1.5 * tol
    assert rel_err(got, whole.float().cpu()) < (6e-3 if dtype == torch.bfloat16 else 5e-6)
    with pytest.raises(Exception):
        ops.attention_merge(parts[:1], out, B, H, Lq)


def test_attention_one_wave_per_simd_nan_propagates():
    """a NaN key poisons every row of its (batch, head) in the long-sequence kernel too, and only those"""
    ops = _ops()
    dtype = torch.bfloat16
    B, H, L = 1, 2, 2048
    q, k, v = (_q(_rand((B, H, L, 64), s_), dtype) for s_ in (171, 172, 173))
    k[0, 1, 777, 5] = float("nan")
    out = torch.empty((B * L, H * 64), dtype=dtype, device=DEV)
    ops.attention((q * ops.Q_PRESCALE).to(dtype).to(DEV), k.to(dtype).to(DEV), vt_layout(v).to(dtype).to(DEV), out, prescaled=True)
    assert torch.isnan(out[:, 64:].float()).all() and torch.isfinite(out[:, :64].float()).all()


@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 2, 64, 64), (2, 3, 100, 100), (1, 12, 324, 324), (2, 12, 257, 257), (1, 2, 2100, 2100),
                                       (3, 2, 200, 64), (1, 1, 70, 1000)])
def test_attention_row_major_v(B, H, Lq, Lk):
    """M324_ATTN_V_ROWMAJOR: V[B,H,Lk,64] as the fused QKV projection writes it, fragments transposed in the LDS read --
    same result as the transposed / permuted Vt operand (only the bf16 rounding of P is shared; sums in the same order)."""
    ops = _ops()
    dtype = torch.bfloat16
    q, k, v = (_q(_rand((B, H, L, 64), s_, sc), dtype) for L, s_, sc in ((Lq, 81, 1.5), (Lk, 82, 1.5), (Lk, 83, 1.0)))
    dq, dk, dv = (t.to(dtype).to(DEV) for t in (q, k, v))
    out_t = torch.empty((B * Lq, H * 64), dtype=dtype, device=DEV)
    out_r = torch.full((B * Lq, H * 64), float("nan"), dtype=dtype, device=DEV)
    ops.attention(dq, dk, vt_layout(v).to(dtype).to(DEV), out_t)
    ops.attention(dq, dk, dv, out_r, v_rowmajor=True)
    ref = _attn_ref(q, k, v, 64 ** -0.5).reshape(B * Lq, H * 64)
    assert torch.isfinite(out_r.float()).all()
    assert rel_err(out_r.float(), ref) < 8e-3
    assert torch.equal(out_r, out_t)                       # identical arithmetic, different operand layout


def test_attention_nan_propagates():
    """No silent clamping: a NaN key poisons the rows that see it (failure detection)."""
    ops = _ops()
    q, k, v = _rand((1, 1, 64, 64), 28), _rand((1, 1, 64, 64), 29), _rand((1, 1, 64, 64), 30)
    k[0, 0, 3, 5] = float("nan")
    out = torch.empty((64, 64), dtype=torch.float32, device=DEV)
    ops.attention(q.to(DEV), k.to(DEV), vt_layout(v).to(DEV), out)
    assert torch.isnan(out).all()


# ------------------------------------------------------------------------------------------- patchify
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Hin", [64, 224, 512])
def test_patchify(dtype, Hin):
    ops = _ops()
    Fr, size, patch, Kp = 2, 224, 14, 640
    g = torch.Generator().manual_seed(31)
    video = torch.rand((Fr, Hin, Hin, 3), generator=g)
    out = ops.patchify(video.to(DEV), size, patch, Kp, dtype).float().cpu()
    img = torch.nn.functional.interpolate(video.permute(0, 3, 1, 2), (size, size), mode="bilinear", align_corners=False)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    img = (img - mean) / std
    ref = torch.nn.functional.unfold(img, kernel_size=patch, stride=patch).transpose(1, 2).reshape(Fr * 256, 588)
    assert rel_err(out[:, :588], ref) < (2e-6 if dtype == torch.float32 else 4e-3)
    assert float(out[:, 588:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------- points
@pytest.mark.parametrize("dtype", DT)
def test_point_encode_and_concat(dtype):
    ops = _ops()
    P, C, Kp = 333, 192, 256
    g = torch.Generator().manual_seed(32)
    xyz = torch.rand((P, 3), generator=g) - 0.5
    enc = ops.point_encode(xyz.to(DEV), dtype).float().cpu()
    e = (2.0 ** torch.arange(8, dtype=torch.float32)) * math.pi
    proj = torch.cat([xyz[:, i:i + 1] * e for i in range(3)], dim=1)           # fp32 products, like the reference einsum
    ref = torch.cat([proj.double().sin(), proj.double().cos(), xyz.double()], dim=1)
    assert rel_err(enc[:, :51], ref) < (1e-6 if dtype == torch.float32 else 4e-3)
    assert float(enc[:, 51:].abs().max()) == 0.0
    feat = torch.full((P, Kp), 7.0, dtype=dtype, device=DEV)
    nrm, rgb = _rand((P, 3), 33), torch.rand((P, 3), generator=g)
    ops.point_concat(nrm.to(DEV), rgb.to(DEV), feat, C)
    f = feat.float().cpu()
    assert float((f[:, :C] - 7.0).abs().max()) == 0.0
    assert rel_err(f[:, C:C + 6], torch.cat([nrm, rgb], dim=1)) < (1e-7 if dtype == torch.float32 else 4e-3)
    assert float(f[:, C + 6:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------- assemble
def test_assemble_tokens():
    ops = _ops()
    B, T, K, Pn, C = 2, 3, 8, 16, 192
    dino_x = _rand((B * T * (Pn + 1), C), 34) * 2
    dw, db = 1 + 0.1 * _rand((C,), 35), 0.1 * _rand((C,), 36)
    pos, sp0, spr, mesh = _rand((T * Pn, C), 37), _rand((4, C), 38), _rand((4, C), 39), _rand((B * K, C), 40)
    lw = 1 + 0.1 * _rand((C,), 41)
    out = ops.assemble_tokens(dino_x.to(DEV), dw.to(DEV), db.to(DEV), 1e-6, pos.to(DEV), sp0.to(DEV), spr.to(DEV),
                              mesh.to(DEV), lw.to(DEV), 1e-5, B, T, K, Pn)
    LN = torch.nn.functional.layer_norm
    dn = LN(dino_x.double().reshape(B, T, Pn + 1, C)[:, :, 1:], (C,), dw.double(), db.double(), 1e-6)
    vid = dn + pos.double().reshape(1, T, Pn, C)
    special = torch.stack([sp0] + [spr] * (T - 1), dim=0).double().unsqueeze(0).expand(B, -1, -1, -1)
    tok = torch.cat([special, mesh.double().reshape(B, 1, K, C).expand(-1, T, -1, -1), vid], dim=2)
    ref = LN(tok, (C,), lw.double(), None, 1e-5).reshape(-1, C)
    assert rel_err(out, ref) < 1e-6
    # training mode: pos_drop on the video rows with the documented counter-based mask; pre-LN rows and LN'ed rows
    from motion324_amd import synth
    p_drop, seed = 0.3, 0xDEADBEEFCAFE1234
    keep = torch.from_numpy(synth.dropout_keep(seed, B * T * Pn * C, p_drop)).reshape(B, T, Pn, C)
    assert 0.6 < keep.double().mean() < 0.8
    vid_d = vid * keep.double() / (1.0 - p_drop)
    tok_d = torch.cat([special, mesh.double().reshape(B, 1, K, C).expand(-1, T, -1, -1), vid_d], dim=2)
    args = (dino_x.to(DEV), dw.to(DEV), db.to(DEV), 1e-6, pos.to(DEV), sp0.to(DEV), spr.to(DEV), mesh.to(DEV))
    pre = ops.assemble_tokens(*args, None, 1e-5, B, T, K, Pn, p_drop, seed)
    assert rel_err(pre, tok_d.reshape(-1, C)) < 1e-6
    dropped = (pre.reshape(B, T, 4 + K + Pn, C)[:, :, 4 + K:] == 0).cpu()
    assert torch.equal(dropped, ~keep)                                      # the mask itself is bit-exact
    out_d = ops.assemble_tokens(*args, lw.to(DEV), 1e-5, B, T, K, Pn, p_drop, seed)
    assert rel_err(out_d, LN(tok_d, (C,), lw.double(), None, 1e-5).reshape(-1, C)) < 1e-6
    with pytest.raises(Exception):
        ops.assemble_tokens(*args, None, 1e-5, B, T, K, Pn, 1.0, seed)


def test_dino_cls_rows():
    ops = _ops()
    Fr, rpf, C = 3, 5, 192
    x = torch.zeros((Fr * rpf, C), device=DEV)
    cls, pos0 = _rand((C,), 42), _rand((C,), 43)
    ops.dino_cls_rows(cls.to(DEV), pos0.to(DEV), x, Fr, rpf)
    ref = torch.zeros(Fr, rpf, C)
    ref[:, 0] = cls + pos0
    assert torch.equal(x.cpu(), ref.reshape(-1, C))


# ------------------------------------------------------------------------------------------- head / loss
@pytest.mark.parametrize("dtype", DT)
def test_linear_n3(dtype):
    ops = _ops()
    M, K = 1001, 768
    a, w, b = _q(_rand((M, K), 44), dtype), _rand((3, K), 45, 0.05), _rand((3,), 46)
    out = torch.empty((M, 3), dtype=torch.float32, device=DEV)
    ops.linear_n3(a.to(dtype).to(DEV), w.to(DEV), b.to(DEV), out)
    assert rel_err(out, a.double() @ w.double().T + b.double()) < 1e-5


def test_mse():
    ops = _ops()
    a, b = _rand((2, 3, 1000, 3), 47), _rand((2, 3, 1000, 3), 48)
    out = ops.mse(a.to(DEV), b.to(DEV), 0.5)
    assert abs(float(out) - 0.5 * float(((a.double() - b.double()) ** 2).mean())) < 1e-6



# ------------------------------------------------------------------------------------------- LayerNorm fold
def _ln_ref(x, w, b, eps):
    x = x.double()
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    y = (x - mu) / torch.sqrt(var + eps) * w.double()
    return y + b.double() if b is not None else y


@pytest.mark.parametrize("variant", ["v0", "v2", "v10", "v11", "v12", "v13"])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,groups", [(3 * 256, 768, 192, 1), (2 * 230, 192, 256, 2), (513, 576, 128, 1)])
def test_gemm_ln_fold_producer(tune, variant, out_dtype, M, N, K, groups):
    """Producer side of the LayerNorm fold: the epilogue that writes the stream x = residual + A W^T + b also leaves the
    per-64-column-block (sum, M2) of every row (the fp32 values it is about to round) and, next to an fp32 stream, its bf16
    twin; m324_rowstats_finish merges the blocks into (rstd, -rstd mean).  Checked against fp64 LayerNorm statistics of the
    exact stream, with a row offset of +30 on some rows (mean >> std: the E[x^2] - mean^2 form would lose 3 digits there)."""
    ops = _ops()
    if variant != "v0":
        tune("M324_GEMM", variant)
    dtype = torch.bfloat16
    a, w = _q(_rand((M, K), 201), dtype), _q(_rand((N, K), 202, 0.1), dtype)
    bias = _rand((N,), 203)
    res = _rand((M // groups, N), 204)
    res[::7] += 30.0
    ad, wd = a.to(dtype).to(DEV), w.to(dtype).to(DEV)
    exact = a.double() @ w.double().T + bias.double() + res.double().repeat(groups, 1)
    part = torch.full((N // 64, M, 2), float("nan"), device=DEV)
    copy = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV) if out_dtype == torch.float32 else None
    if out_dtype == torch.float32 or groups > 1:
        out = torch.full((M, N), float("nan"), dtype=out_dtype, device=DEV)
        ops.gemm(ad, wd, out, bias=bias.to(DEV), residual=res.to(DEV), res_rows=M // groups if groups > 1 else 0,
                 stats_out=part, copy_out=copy)
    else:                                                    # the bf16 stream updated in place (the decoder MLP's fc2)
        out = res.to(dtype).to(DEV)
        exact = a.double() @ w.double().T + bias.double() + res.to(dtype).double()
        ops.gemm(ad, wd, out, bias=bias.to(DEV), residual=out, stats_out=part)
    assert rel_err(out.float(), exact) < (1e-5 if out_dtype == torch.float32 else 4e-3)
    if copy is not None:
        assert torch.equal(copy.cpu(), out.cpu().to(torch.bfloat16))         # the twin is the stored value, rounded once
    p = part.double().cpu()
    assert torch.isfinite(p).all()
    blocks = exact.reshape(M, N // 64, 64)
    assert torch.allclose(p[..., 0].T, blocks.sum(-1), rtol=1e-5, atol=1e-3)
    m2 = ((blocks - blocks.mean(-1, keepdim=True)) ** 2).sum(-1)
    assert torch.allclose(p[..., 1].T, m2, rtol=1e-4, atol=1e-4)
    stat = torch.empty((M, 2), device=DEV)
    ops.rowstats_finish(part, 1e-5, stat)
    mu = exact.mean(1)
    rstd = 1.0 / torch.sqrt(exact.var(1, unbiased=False) + 1e-5)
    s = stat.double().cpu()
    assert torch.allclose(s[:, 0], rstd, rtol=2e-5) and torch.allclose(s[:, 1], -rstd * mu, rtol=2e-5, atol=1e-5)


def test_rowstats_from_stream():
    ops = _ops()
    rows, C = 700, 768
    x = _rand((rows, C), 211)
    x[::5] += 12.0
    stat = torch.empty((rows, 2), device=DEV)
    copy = torch.empty((rows, C), dtype=torch.bfloat16, device=DEV)
    ops.rowstats(x.to(DEV), 1e-6, stat, copy)
    rstd = 1.0 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-6)
    s = stat.double().cpu()
    assert torch.allclose(s[:, 0], rstd, rtol=1e-5) and torch.allclose(s[:, 1], -rstd * x.double().mean(1), rtol=1e-5, atol=1e-6)
    assert torch.equal(copy.cpu(), x.to(torch.bfloat16))


def _folded(lnw, lnb, w, b):
    """What Prepared.folded() hands to the kernel, on the CPU: (W' bf16, colsum, bias')."""
    wf = (w * lnw[None, :]).to(torch.bfloat16)
    colsum = wf.double().sum(1).float()
    bias = None
    if lnb is not None or b is not None:
        bias = torch.zeros(w.shape[0], dtype=torch.float64)
        if b is not None:
            bias += b.double()
        if lnb is not None:
            bias += w.double() @ lnb.double()
        bias = bias.float()
    return wf, colsum, bias


def test_layernorm_pair_and_gemm_pair_equal_the_separate_launches():
    """m324_layernorm_pair / m324_gemm_pair (horizontal fusion of the decoder's norm_q + norm_kv and q + k|v projections): the
    same bits as four separate launches, including the gathered rows of the second LayerNorm; a pair the library does not build
    falls back to two launches.  Reference transformer.py:112-132,365-369."""
    ops = _ops()
    dt = torch.bfloat16
    C, H, N, T, K, Lt = 768, 12, 2048, 8, 64, 324
    pf, tok = _rand((N, C), 401).to(DEV), _rand((T * Lt, C), 402).to(DEV)
    lw = [(1 + 0.1 * _rand((C,), 403 + i)).to(DEV) for i in range(2)]
    lb = [(0.1 * _rand((C,), 405 + i)).to(DEV) for i in range(2)]
    wq, wkv = _rand((C, C), 407, 0.05).to(dt).to(DEV), _rand((2 * C, C), 408, 0.05).to(dt).to(DEV)
    bq, bkv = _rand((C,), 409, 0.1).to(DEV), _rand((2 * C,), 410, 0.1).to(DEV)
    qw, kw = (1 + 0.1 * _rand((64,), 411)).to(DEV), (1 + 0.1 * _rand((64,), 412)).to(DEV)
    rm = (K, Lt, 4)

    def outs():
        return (torch.full((1, H, N, 64), float("nan"), dtype=dt, device=DEV), torch.full((T, H, K, 64), float("nan"), dtype=dt, device=DEV),
                torch.full((T, H, 64, K), float("nan"), dtype=dt, device=DEV))
    # separate launches
    qn, kn = torch.empty((N, C), dtype=dt, device=DEV), torch.empty((T * K, C), dtype=dt, device=DEV)
    ops.layernorm(pf, lw[0], lb[0], 1e-5, qn)
    ops.layernorm(tok, lw[1], lb[1], 1e-5, kn, row_map=rm)
    Q0, K0, V0 = outs()
    ops.gemm(qn, wq, None, bias=bq, qkv_heads=(Q0, None, None, qw, None, 1e-5, ops.Q_PRESCALE, N, H))
    ops.gemm(kn, wkv, None, bias=bkv, qkv_heads=(None, K0, V0, None, kw, 1e-5, 1.0, K, H, True))
    # paired
    qn2, kn2 = torch.full_like(qn, float("nan")), torch.full_like(kn, float("nan"))
    ops.layernorm_pair(pf, lw[0], lb[0], 1e-5, qn2, tok, lw[1], lb[1], 1e-5, kn2, row_map1=rm)
    assert torch.equal(qn2, qn) and torch.equal(kn2, kn)
    Q1, K1, V1 = outs()
    pair = []
    ops.gemm(qn2, wq, None, bias=bq, qkv_heads=(Q1, None, None, qw, None, 1e-5, ops.Q_PRESCALE, N, H), defer=pair)
    ops.gemm(kn2, wkv, None, bias=bkv, qkv_heads=(None, K1, V1, None, kw, 1e-5, 1.0, K, H, True), defer=pair)
    assert torch.isnan(Q1.float()).all()                     # nothing launched yet
    ops.gemm_pair(pair)
    assert torch.equal(Q1, Q0) and torch.equal(K1, K0) and torch.equal(V1, V0)
    # a pair the library does not build (K = 64: the two-stage kernel, not the chunk ring): two launches, same results
    a64, w64 = _rand((256, 64), 413).to(dt).to(DEV), _rand((C, 64), 414, 0.1).to(dt).to(DEV)
    Qa, Qb, Qc = (torch.full((1, H, 256, 64), float("nan"), dtype=dt, device=DEV) for _ in range(3))
    ops.gemm(a64, w64, None, bias=bq, qkv_heads=(Qa, None, None, qw, None, 1e-5, 1.0, 256, H))
    pair = []
    ops.gemm(a64, w64, None, bias=bq, qkv_heads=(Qb, None, None, qw, None, 1e-5, 1.0, 256, H), defer=pair)
    ops.gemm(a64, w64, None, bias=bq, qkv_heads=(Qc, None, None, qw, None, 1e-5, 1.0, 256, H), defer=pair)
    ops.gemm_pair(pair)
    assert torch.equal(Qb, Qa) and torch.equal(Qc, Qa)


def _block_table(xb: torch.Tensor) -> torch.Tensor:
    """What a producer GEMM leaves for the rows of its output: [K / 64, M, 2] (sum, sum of squared deviations from the block mean)."""
    M, K = xb.shape
    blocks = xb.double().reshape(M, K // 64, 64)
    s = blocks.sum(2)
    m2 = ((blocks - s[..., None] / 64) ** 2).sum(2)
    return torch.stack([s, m2], dim=2).permute(1, 0, 2).contiguous().float()


@pytest.mark.parametrize("variant", ["v0", "v2", "v10", "v11", "v12", "v13"])
@pytest.mark.parametrize("M,N,K", [(470, 192, 256), (513, 3072, 768), (1100, 768, 1024)])
def test_gemm_ln_fold_consumer_merges_block_table(tune, variant, M, N, K):
    """ln = (part, colsum, eps): the consumer merges the producer's per-block statistics of its rows itself (no m324_rowstats_finish
    launch between the two GEMMs).  Same result as the merged table -- the two merges differ by fp32 rounding only -- on every
    schedule, with ragged last tiles, several tiles per persistent workgroup (v10) and rows whose mean dwarfs their deviation."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    if variant != "v0":
        tune("M324_GEMM", variant)
    x = _rand((M, K), 321)
    x[::7] += 5.0
    x[3::11] *= 0.01
    xb = x.to(torch.bfloat16)
    lnw, lnb = 1 + 0.2 * _rand((K,), 322), 0.1 * _rand((K,), 323)
    w, b = _rand((N, K), 324, 0.05), _rand((N,), 325)
    wf, colsum, bias = _folded(lnw, lnb, w, b)
    part = _block_table(xb).to(DEV)
    stat = torch.empty((M, 2), device=DEV)
    ops.rowstats_finish(part, 1e-5, stat)
    outs = []
    for ln in ((stat, colsum.to(DEV)), (part, colsum.to(DEV), 1e-5)):
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.gemm(xb.to(DEV), wf.to(DEV), out, bias=bias.to(DEV), act=ACT_GELU, ln=ln)
        outs.append(out.float().cpu())
    ref = _ln_ref(xb.float(), lnw, lnb, 1e-5) @ w.double().T + b.double()
    ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2.0)))
    assert torch.isfinite(outs[1]).all()
    assert rel_err(outs[1], ref) < 5e-3
    assert rel_err(outs[1], outs[0]) < 2e-4                  # bf16 outputs: a last-bit flip here and there


@pytest.mark.parametrize("M,N,K", [(513, 3072, 768), (1100, 768, 1024)])
def test_gemm_ln_fold_consumer_result_does_not_depend_on_the_schedule(tune, M, N, K):
    """Every consumer merges a row's block statistics with ONE arithmetic (gemm_tile.h ln_combine_halves: two halves by Chan's
    steps, then the pairwise form), whatever its tile shape keeps in registers: the product must be bit-identical across
    schedules.  (A hipGraph branch that holds half of the rows may get another schedule than the eager pass over all of them:
    round 5 found the frame-parallel chain differing from its eager form by a bf16 band because the 256-wide kernels merged in
    another order than the 128-wide ones.)"""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    x = _rand((M, K), 421)
    x[::5] += 4.0
    xb = x.to(torch.bfloat16)
    lnw, lnb = 1 + 0.2 * _rand((K,), 422), 0.1 * _rand((K,), 423)
    w, b = _rand((N, K), 424, 0.05), _rand((N,), 425)
    wf, colsum, bias = _folded(lnw, lnb, w, b)
    part = _block_table(xb).to(DEV)
    outs = {}
    for variant in ("v2", "v10", "v11", "v12", "v13"):
        tune("M324_GEMM", variant)
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.gemm(xb.to(DEV), wf.to(DEV), out, bias=bias.to(DEV), act=ACT_GELU, ln=(part, colsum.to(DEV), 1e-5))
        outs[variant] = out.cpu()
    for variant, out in outs.items():
        assert torch.equal(out, outs["v2"]), variant


def test_gemm_ln_fold_block_table_rejects_odd_block_counts():
    ops = _ops()
    from motion324_amd.lib import M324Error
    M, N, K = 256, 128, 192
    a = torch.zeros((M, K), dtype=torch.bfloat16, device=DEV)
    w = torch.zeros((N, K), dtype=torch.bfloat16, device=DEV)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(M324Error, match="even count"):
        ops.gemm(a, w, out, ln=(torch.zeros((3, M, 2), device=DEV), torch.zeros((N,), device=DEV), 1e-5))


@pytest.mark.parametrize("variant", ["v0", "v2", "v10", "v11", "v12", "v13"])
@pytest.mark.parametrize("mode", ["plain", "gelu", "f32out"])
@pytest.mark.parametrize("M,N,K", [(3 * 256, 768, 192), (470, 192, 256), (513, 3072, 768)])
def test_gemm_ln_fold_consumer(tune, variant, mode, M, N, K):
    """Consumer side: A = the raw bf16 stream, W' = w_ln W, epilogue rstd (acc - mean colsum) + b' == Linear(LayerNorm(x)).
    Against fp64 LayerNorm + Linear of the same bf16 stream (rows with mean >> std included: the subtraction in the epilogue
    cancels there, the band states what is left)."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    if variant != "v0":
        tune("M324_GEMM", variant)
    x = _rand((M, K), 221)
    x[::9] += 3.0                                            # mean = 3 sigma on some rows
    xb = x.to(torch.bfloat16)
    lnw, lnb = 1 + 0.2 * _rand((K,), 222), 0.1 * _rand((K,), 223)
    w, b = _rand((N, K), 224, 0.05), _rand((N,), 225)
    wf, colsum, bias = _folded(lnw, lnb, w, b)
    stat = torch.empty((M, 2), device=DEV)
    ops.rowstats(xb.float().to(DEV), 1e-5, stat)
    odt = torch.float32 if mode == "f32out" else torch.bfloat16
    out = torch.full((M, N), float("nan"), dtype=odt, device=DEV)
    ops.gemm(xb.to(DEV), wf.to(DEV), out, bias=bias.to(DEV), act=ACT_GELU if mode == "gelu" else 0, ln=(stat, colsum.to(DEV)))
    ref = _ln_ref(xb.float(), lnw, lnb, 1e-5) @ w.double().T + b.double()
    if mode == "gelu":
        ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2.0)))
    # W' is rounded to bf16 (2^-9 per weight, averaging out over K) and the output once more
    assert rel_err(out.float(), ref) < (3e-3 if odt == torch.float32 else 5e-3)


@pytest.mark.parametrize("L,vt", [(257, False), (324, False), (256, True)])
def test_gemm_ln_fold_qkv_heads(L, vt):
    """The fused q|k|v projection epilogue (head-major Q / K / V, RMSNorm, q pre-scale, optional transposed V) behind a folded
    LayerNorm == the same epilogue fed with the explicitly normalised rows."""
    ops = _ops()
    B, H, C = 3, 3, 192
    M = B * L
    x = _rand((M, C), 231)
    xb = x.to(torch.bfloat16)
    lnw, lnb = 1 + 0.2 * _rand((C,), 232), 0.1 * _rand((C,), 233)
    w, b = _rand((3 * C, C), 234, 0.08), _rand((3 * C,), 235, 0.1)
    qw, kw = 1 + 0.1 * _rand((64,), 236), 1 + 0.1 * _rand((64,), 237)
    wf, colsum, bias = _folded(lnw, lnb, w, b)
    stat = torch.empty((M, 2), device=DEV)
    ops.rowstats(xb.float().to(DEV), 1e-5, stat)

    def run(a, wmat, bvec, ln):
        Q, K = (torch.full((B, H, L, 64), float("nan"), dtype=torch.bfloat16, device=DEV) for _ in range(2))
        V = torch.full((B, H, 64, L) if vt else (B, H, L, 64), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.gemm(a.to(DEV), wmat.to(DEV), None, bias=bvec.to(DEV),
                 qkv_heads=(Q, K, V, qw.to(DEV), kw.to(DEV), 1e-5, ops.Q_PRESCALE, L, H), ln=ln)
        return Q.float().cpu(), K.float().cpu(), V.float().cpu()
    got = run(xb, wf, bias, (stat, colsum.to(DEV)))
    h = _ln_ref(xb.float(), lnw, lnb, 1e-5).to(torch.bfloat16)
    want = run(h, w.to(torch.bfloat16), b, None)
    for g, r in zip(got, want):
        assert torch.isfinite(g).all() and rel_err(g, r) < 8e-3     # two bf16 roundings apart (h vs W')
    # C = 192 is three blocks: m324_gemm takes even block counts only -- the per-block table goes through the launch of its own
    with pytest.raises(Exception, match="even count"):
        run(xb, wf, bias, (_block_table(xb).to(DEV), colsum.to(DEV), 1e-5))


def test_gemm_ln_fold_n3_head():
    """M324_AUX_N3 (Linear -> GELU -> Linear(C -> 3) in one epilogue) behind a folded LayerNorm."""
    ops = _ops()
    from motion324_amd.lib import ACT_GELU
    M, C = 1000, 768
    x = _rand((M, C), 241)
    xb = x.to(torch.bfloat16)
    lnw, lnb = 1 + 0.2 * _rand((C,), 242), 0.1 * _rand((C,), 243)
    w, b = _rand((C, C), 244, 0.05), _rand((C,), 245, 0.1)
    w3, b3 = _rand((3, C), 246, 0.1), _rand((3,), 247)
    wf, colsum, bias = _folded(lnw, lnb, w, b)
    stat = torch.empty((M, 2), device=DEV)
    ops.rowstats(xb.float().to(DEV), 1e-5, stat)
    part = torch.empty((C // 64, M, 3), device=DEV)
    ops.gemm(xb.to(DEV), wf.to(DEV), None, bias=bias.to(DEV), act=ACT_GELU, n3=(w3.to(DEV), part), ln=(stat, colsum.to(DEV)))
    out = torch.empty((M, 3), device=DEV)
    ops.n3_finish(part, b3.to(DEV), out)
    v = _ln_ref(xb.float(), lnw, lnb, 1e-5) @ w.double().T + b.double()
    v = 0.5 * v * (1 + torch.erf(v / math.sqrt(2.0)))
    assert rel_err(out, v @ w3.double().T + b3.double()) < 3e-3
    # the same with the consumer merging the per-block table itself
    part2 = torch.empty((C // 64, M, 3), device=DEV)
    ops.gemm(xb.to(DEV), wf.to(DEV), None, bias=bias.to(DEV), act=ACT_GELU, n3=(w3.to(DEV), part2),
             ln=(_block_table(xb).to(DEV), colsum.to(DEV), 1e-5))
    out2 = torch.empty((M, 3), device=DEV)
    ops.n3_finish(part2, b3.to(DEV), out2)
    assert rel_err(out2, out.double()) < 1e-4


def test_gemm_ln_fold_rejects_unsupported_shapes():
    ops = _ops()
    from motion324_amd.lib import M324Error
    a = torch.zeros((64, 128), dtype=torch.bfloat16, device=DEV)
    w = torch.zeros((128, 128), dtype=torch.bfloat16, device=DEV)
    out = torch.zeros((64, 128), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(M324Error):                            # M <= 64: the skinny kernel has no fold
        ops.gemm(a, w, out, ln=(torch.zeros((64, 2), device=DEV), torch.zeros(128, device=DEV)))
    a = torch.zeros((256, 128), dtype=torch.bfloat16, device=DEV)
    w = torch.zeros((96, 128), dtype=torch.bfloat16, device=DEV)
    out = torch.zeros((256, 96), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(M324Error):                            # N % 64 != 0
        ops.gemm(a, w, out, ln=(torch.zeros((256, 2), device=DEV), torch.zeros(96, device=DEV)))


@pytest.mark.parametrize("B,L", [(32, 64), (2, 4096), (3, 192)])
def test_gemm_cross_attention_projection_heads(B, L):
    """The q and the k|v projections of a cross-attention block (transformer.py:112-132) with the head split in the GEMM
    epilogue (qkv_heads with NULL parts: q alone; k|v with the transposed, key-permuted Vt -- L = 64 is the decoder's latent
    tokens, one 64-key tile per frame) == the plain projection followed by m324_qkv_split (Vt bit for bit: both round acc + bias to bf16 once; Q / K up to the one
    bf16 rounding the two-pass form puts in front of the RMSNorm)."""
    ops = _ops()
    H, C = 3, 192
    M = B * L
    x = _rand((M, C), 301).to(torch.bfloat16).to(DEV)
    wq, bq = (_rand((C, C), 302, 0.08)).to(torch.bfloat16).to(DEV), _rand((C,), 303, 0.1).to(DEV)
    wkv, bkv = (_rand((2 * C, C), 304, 0.08)).to(torch.bfloat16).to(DEV), _rand((2 * C,), 305, 0.1).to(DEV)
    qw, kw = (1 + 0.1 * _rand((64,), 306)).to(DEV), (1 + 0.1 * _rand((64,), 307)).to(DEV)
    # q alone
    q = torch.empty((M, C), dtype=torch.bfloat16, device=DEV)
    ops.gemm(x, wq, q, bias=bq)
    Qs, _, _ = ops.qkv_split(q, None, None, qw, None, 1e-5, B, L, H, torch.bfloat16, q_scale=ops.Q_PRESCALE)
    Qf = torch.full((B, H, L, 64), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm(x, wq, None, bias=bq, qkv_heads=(Qf, None, None, qw, None, 1e-5, ops.Q_PRESCALE, L, H))
    assert torch.isfinite(Qf.float()).all() and rel_err(Qf.float(), Qs.float().double()) < 6e-3
    # k|v
    kv = torch.empty((M, 2 * C), dtype=torch.bfloat16, device=DEV)
    ops.gemm(x, wkv, kv, bias=bkv)
    _, Ks, Vts = ops.qkv_split(None, kv[:, :C], kv[:, C:], None, kw, 1e-5, B, L, H, torch.bfloat16)
    Kf = torch.full((B, H, L, 64), float("nan"), dtype=torch.bfloat16, device=DEV)
    Vtf = torch.full((B, H, 64, L), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm(x, wkv, None, bias=bkv, qkv_heads=(None, Kf, Vtf, None, kw, 1e-5, 1.0, L, H, True))
    assert torch.isfinite(Kf.float()).all() and rel_err(Kf.float(), Ks.float().double()) < 6e-3
    assert torch.equal(Vtf, Vts)
