"""Tensor-level wrappers over the libm324 C ABI (motion324_amd/lib.py).

PyTorch is only the owner of device memory and of the HIP stream here; every op below is one
C-ABI call that enqueues hand-written gfx950 kernels on torch's current stream.  All tensors must
live on a HIP device -- there is deliberately no CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import lib as L
from . import switches
from .timing import active as _timing, span

F32, BF16 = L.F32, L.BF16
_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}


def code_of(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise L.M324Error(f"unsupported dtype {dtype}")


def torch_dtype(code: int) -> torch.dtype:
    return _TORCH_DT[code]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise L.M324Error("libm324 ops need HIP device tensors (no CPU fallback on this path)")
    return t.data_ptr()


def _wrote(*tensors) -> None:
    """libm324 writes through raw pointers: torch's in-place version counters do not see it.  Every wrapper marks the tensors
    its kernels write, so that `_version`-keyed caches (backward.Carry's bf16 twin of a gradient, the Prepared weight stamp) are
    invalidated by library writes exactly as by torch writes."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def _vec(t: Optional[torch.Tensor], n: int, name: str) -> Optional[int]:
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
        raise L.M324Error(f"{name}: expected contiguous fp32[{n}], got {t.dtype}{tuple(t.shape)}")
    return _p(t)


def _rows(t: torch.Tensor, name: str):
    """2-D view contract: unit inner stride; returns (ptr, leading dimension in elements)."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise L.M324Error(f"{name}: expected a 2-D tensor with unit inner stride, got {tuple(t.shape)} {t.stride()}")
    return _p(t), t.stride(0)


def _gemm_plan(args) -> str:
    buf = C.create_string_buffer(192)
    L.load().m324_gemm_plan(C.byref(args), buf, 192)
    return buf.value.decode()


def gemm_schedule(M: int, N: int, K: int, *, act: int = L.ACT_NONE, out_dtype: torch.dtype = torch.bfloat16, bias: bool = True,
                  fold_merged: bool = False) -> int:
    """The schedule m324_gemm WOULD pick for a bf16 GEMM of this shape and epilogue with the library's live tunables (host-only
    query, m324_gemm_plan: no launch, no device memory touched).  fold_merged: a LayerNorm-fold consumer handed the MERGED
    statistics table.  The host never mirrors the chooser's rules: it asks."""
    args = L.GemmArgs()
    args.A = args.W = args.C = 4096                       # aligned stand-ins: the plan query reads sizes and flags only
    args.M, args.N, args.K, args.lda, args.ldw, args.ldc = M, N, K, K, K, N
    args.in_dtype, args.out_dtype, args.act, args.batch = BF16, code_of(out_dtype), act, 1
    if bias:
        args.bias = 4096
    if fold_merged:
        args.ln_rowstat, args.ln_colsum, args.ln_ncb, args.ln_eps = 4096, 4096, 0, 1e-5
    buf = C.create_string_buffer(192)
    return int(L.load().m324_gemm_plan(C.byref(args), buf, 192))


def _attn_plan(B, H, Lq, Lk, flags, dtype_code) -> str:
    buf = C.create_string_buffer(192)
    L.load().m324_attention_plan(B, H, Lq, Lk, flags, dtype_code, buf, 192)
    return buf.value.decode()


def gemm_splitk(a: torch.Tensor, w: torch.Tensor, slices: int) -> torch.Tensor:
    """out[M,N] fp32 = a[M,K] @ w[N,K]^T with the contraction cut into `slices` independent GEMMs launched together
    (grid.y) and summed by m324_colsum: for weight gradients, where M x N is small and K (tokens) is huge."""
    M, K = a.shape
    N = w.shape[0]
    esz = a.element_size()
    tile = 64 if esz == 2 else 32
    ks = (K // tile // slices) * tile
    if slices <= 1 or ks == 0:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        return gemm(a, w, out)
    main = ks * slices
    part = torch.empty((slices + (1 if main < K else 0), M, N), dtype=torch.float32, device=a.device)
    args = L.GemmArgs()
    args.A, args.lda = _rows(a, "a")
    args.W, args.ldw = _rows(w, "w")
    args.C, args.ldc = _p(part), N
    args.M, args.N, args.K = M, N, ks
    args.in_dtype, args.out_dtype = code_of(a.dtype), F32
    args.batch, args.strideA, args.strideW, args.strideC = slices, ks, ks, M * N
    with span(f"gemm_{'bf16' if esz == 2 else 'f32'}", 2.0 * M * N * main, esz * (M + N) * main + 4.0 * slices * M * N,
              f"split-K M={M} N={N} K={main} slices={slices}" if _timing() else ""):
        L.check(L.load().m324_gemm(C.byref(args), _stream()), "m324_gemm")
    if main < K:                                        # remainder of the contraction
        gemm(a[:, main:], w[:, main:], part[slices])
    return colsum(part.reshape(part.shape[0], M * N)).reshape(M, N)


def gemm_tn(x: torch.Tensor, y: torch.Tensor, slices: int = 1, out: Optional[torch.Tensor] = None,
            accumulate: bool = False) -> torch.Tensor:
    """out[N, Kc] fp32 = x[M, N]^T @ y[M, Kc] (bf16, token-major operands): the weight gradient dW = dY^T A without
    transposed copies (m324_gemm_tn).  The M tokens are cut into `slices` ranges whose partials m324_colsum adds.
    out: contiguous fp32 [N, Kc] (or any view of N * Kc contiguous floats) that receives the result -- the optimizer's flat
    gradient buffer, so that no copy follows."""
    M, N = x.shape
    Kc = y.shape[1]
    if y.shape[0] != M or x.dtype != torch.bfloat16 or y.dtype != torch.bfloat16:
        raise L.M324Error(f"gemm_tn: x{tuple(x.shape)} {x.dtype} vs y{tuple(y.shape)} {y.dtype} (bf16 operands with equal rows)")
    slices = max(1, min(slices, (M + 63) // 64))
    while slices > 1 and ((M + slices - 1) // slices + 63) // 64 * 64 * (slices - 1) >= M:
        slices -= 1                                   # no empty slice
    if out is not None and (out.dtype != torch.float32 or out.numel() != N * Kc or not out.is_contiguous()):
        raise L.M324Error("gemm_tn: out must be N * Kc contiguous fp32 values")
    # accumulate (out += x^T y: a shared weight's gradient over the decoder's per-sample passes): the slice partials are summed
    # INTO out by the same m324_colsum launch that would have produced a temporary (a single slice goes through it too)
    if accumulate and out is None:
        raise L.M324Error("gemm_tn: accumulate needs out")
    direct = out is not None and slices == 1 and not accumulate
    if direct and COLSUMS.busy(out.view(-1)):
        COLSUMS.flush()
    part = out.view(1, N, Kc) if direct else torch.empty((slices, N, Kc), dtype=torch.float32, device=x.device)
    px, ldx = _rows(x, "x")
    py, ldy = _rows(y, "y")
    with span("gemm_bf16", 2.0 * M * N * Kc, 2.0 * M * (N + Kc) + 4.0 * slices * N * Kc,
              f"TN M={N} N={Kc} K={M} slices={slices}" if _timing() else ""):
        L.check(L.load().m324_gemm_tn(px, ldx, py, ldy, _p(part), Kc, M, N, Kc, slices, N * Kc, _stream()), "m324_gemm_tn")
    if slices == 1 and not accumulate:
        return out if direct else part[0]
    if out is not None:
        if DEFER_COLSUM:
            # nobody reads a weight gradient before its bucket is declared done (backward.GradStore.done flushes): the sum of the
            # slice partials joins the queue and leaves with the other sums of the block in one m324_colsum_multi launch
            COLSUMS.defer(part.reshape(slices, N * Kc), out.view(-1), accumulate)
            return out
        colsum(part.reshape(slices, N * Kc), out=out.view(-1), accumulate=accumulate)
        _wrote(out)
        return out
    return colsum(part.reshape(slices, N * Kc)).reshape(N, Kc)


def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, bias=None, act: int = L.ACT_NONE, gamma=None,
         residual: Optional[torch.Tensor] = None, res_rows: int = 0, row_map=(0, 0, 0),
         preact_out: Optional[torch.Tensor] = None, gelu_grad_of: Optional[torch.Tensor] = None,
         gelu_grad_out: Optional[torch.Tensor] = None, mul_by: Optional[torch.Tensor] = None,
         qkv_heads: Optional[tuple] = None, n3: Optional[tuple] = None, ln: Optional[tuple] = None,
         stats_out: Optional[torch.Tensor] = None, copy_out: Optional[torch.Tensor] = None, defer: Optional[list] = None) -> torch.Tensor:
    """out[row_map(m), :N] = epilogue(a[M,K] @ w[N,K]^T); see include/m324.h m324_gemm.
    preact_out [M, N] (out's dtype) also receives the value the activation is applied to (M324_AUX_STORE_PREACT);
    gelu_grad_of [M, N] = z: the result is multiplied by gelu'(z) (M324_AUX_MUL_GELU_GRAD).  Training only.
    gelu_grad_out [M, N] receives gelu'(pre-activation) (M324_AUX_STORE_GELU_GRAD) and mul_by [M, N] multiplies the result
    (M324_AUX_MUL): the same pair with erf evaluated once, in the forward.
    qkv_heads = (Q, K, V, q_w, k_w, eps, q_scale, L, H): the fused q|k|v projection is written head-major into Q / K / V
    [B, H, L, 64] with per-head RMSNorm and the q pre-scale (M324_AUX_QKV_HEADS); `out` is ignored (may be None).
    A V of shape [B, H, 64, L] selects M324_AUX_QKV_HEADS_VT: V leaves transposed and key-permuted, the operand
    attention() reads by default (L % 128 == 0).
    defer (qkv_heads only): a list that receives the prepared call instead of a launch; gemm_pair(list) then runs two of them
    as ONE launch (m324_gemm_pair).
    LayerNorm fold (include/m324.h): ln = (rowstat fp32 [M, 2], colsum fp32 [N]) -- `a` is the raw stream, `w` carries the
    LayerNorm scale; ln = (part fp32 [K / 64, M, 2], colsum, eps): the producer's stats_out itself, merged by this GEMM; stats_out fp32 [N / 64, M, 2] receives the per-block row statistics of the stored values and copy_out
    bf16 [M, N] their bf16 twin (fp32 `out` only)."""
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or a.dtype != w.dtype:
        raise L.M324Error(f"gemm: a{tuple(a.shape)} {a.dtype} vs w{tuple(w.shape)} {w.dtype}")
    args = L.GemmArgs()
    args.A, args.lda = _rows(a, "a")
    args.W, args.ldw = _rows(w, "w")
    args.M, args.N, args.K = M, N, K
    if ln is not None and len(ln) == 3:
        part, colsum, eps = ln                       # the producer's unmerged table: the consumer merges its rows' blocks itself
        if (part.dtype != torch.float32 or not part.is_contiguous() or K % 64 or tuple(part.shape) != (K // 64, M, 2) or a.dtype != torch.bfloat16):
            raise L.M324Error(f"gemm: ln block table {part.dtype}{tuple(part.shape)} (want contiguous fp32 [{K // 64}, {M}, 2], bf16 operands)")
        args.ln_rowstat, args.ln_colsum = _p(part), _vec(colsum, N, "ln colsum")
        args.ln_ncb, args.ln_eps = K // 64, float(eps)
    elif ln is not None:
        rowstat, colsum = ln
        if (rowstat.dtype != torch.float32 or not rowstat.is_contiguous() or rowstat.numel() != 2 * M or a.dtype != torch.bfloat16):
            raise L.M324Error(f"gemm: ln rowstat {rowstat.dtype}{tuple(rowstat.shape)} (want contiguous fp32 [{M}, 2], bf16 operands)")
        args.ln_rowstat, args.ln_colsum = _p(rowstat), _vec(colsum, N, "ln colsum")
    if stats_out is not None:
        if (stats_out.dtype != torch.float32 or not stats_out.is_contiguous() or N % 64 or tuple(stats_out.shape) != (N // 64, M, 2)):
            raise L.M324Error(f"gemm: stats_out {stats_out.dtype}{tuple(stats_out.shape)} (want contiguous fp32 [{N // 64}, {M}, 2])")
        args.ln_stats_out = _p(stats_out)
    if copy_out is not None:
        if copy_out.dtype != torch.bfloat16 or copy_out.shape[0] < M or copy_out.shape[1] < N:
            raise L.M324Error(f"gemm: copy_out {copy_out.dtype}{tuple(copy_out.shape)} (want bf16 [{M}, {N}])")
        args.ln_copy_out, args.ln_ldcopy = _rows(copy_out, "copy_out")
    if qkv_heads is not None:
        # (Q, K, V, q_w, k_w, eps, q_scale, L, H[, vt]): Q None = the k|v projection of a cross-attention, K and V None = its
        # q projection; vt (default: V has the transposed shape and L != 64) = V leaves transposed + key-permuted
        Qo, Ko, Vo, qw, kw, eps, q_scale, Lh, Hh = qkv_heads[:9]
        Bh = M // Lh
        vt = qkv_heads[9] if len(qkv_heads) > 9 else (Vo is not None and tuple(Vo.shape) == (Bh, Hh, 64, Lh) and Lh != 64)
        for t in (Qo, Ko, Vo):
            if t is None:
                continue
            want = (Bh, Hh, 64, Lh) if (vt and t is Vo) else (Bh, Hh, Lh, 64)
            if t.dtype != torch.bfloat16 or not t.is_contiguous() or tuple(t.shape) != want:
                raise L.M324Error(f"gemm: qkv_heads output {t.dtype}{tuple(t.shape)} (want bf16 {want})")
        if vt and Lh % 64:
            raise L.M324Error(f"gemm: a transposed V output needs L % 64 == 0 (L={Lh})")
        args.C, args.ldc = None, N
        args.in_dtype, args.out_dtype = code_of(a.dtype), BF16
        args.bias = _vec(bias, N, "bias")
        args.aux_mode = 4 if vt else 3
        args.qkv_q, args.qkv_k, args.qkv_v = _p(Qo), _p(Ko), _p(Vo)
        args.qkv_qw, args.qkv_kw = _vec(qw, 64, "q_w"), _vec(kw, 64, "k_w")
        first = next(t for t in (Qo, Ko, Vo) if t is not None)
        args.qkv_eps, args.qkv_qscale, args.qkv_L, args.qkv_H = eps, q_scale, Lh, Hh
        if defer is not None:                       # gemm_pair() launches it together with another one
            defer.append((args, 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N), f"M={M} N={N} K={K} qkv-heads", (a, w, bias, Qo, Ko, Vo, qw, kw)))
            return first
        with span("gemm_bf16", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N),
                  f"{_gemm_plan(args)} | M={M} N={N} K={K} qkv-heads" if _timing() else ""):
            L.check(L.load().m324_gemm(C.byref(args), _stream()), "m324_gemm")
        return first
    if n3 is not None:
        # n3 = (w3 fp32 [3, N], part fp32 [N / 64, M, 3]): gelu(a w^T + bias) is contracted with w3 in the epilogue
        # (M324_AUX_N3); `out` is ignored (may be None); finish with n3_finish(part, bias3, out3)
        w3, part = n3
        if (a.dtype != torch.bfloat16 or w3.dtype != torch.float32 or tuple(w3.shape) != (3, N) or not w3.is_contiguous()
                or part.dtype != torch.float32 or tuple(part.shape) != (N // 64, M, 3) or not part.is_contiguous()
                or residual is not None or gamma is not None or act != L.ACT_GELU or N % 256):
            raise L.M324Error(f"gemm: n3 mode needs bf16 operands, act=GELU, N % 256 == 0, w3 [3,{N}] and part [{N // 64},{M},3] fp32")
        args.C, args.ldc = None, N
        args.in_dtype, args.out_dtype = BF16, BF16
        args.bias = _vec(bias, N, "bias")
        args.act = act
        args.aux, args.ldaux, args.aux_mode = _p(part), 3, 5
        args.qkv_qw = _p(w3)
        with span("gemm_bf16", 2.0 * M * N * K + 2.0 * M * N * 3, 2.0 * (M * K + N * K) + 4.0 * part.numel(),
                  f"{_gemm_plan(args)} | M={M} N={N} K={K} gelu n3" if _timing() else ""):
            L.check(L.load().m324_gemm(C.byref(args), _stream()), "m324_gemm")
        return part
    args.C, args.ldc = _rows(out, "out")
    _wrote(out)
    args.in_dtype, args.out_dtype = code_of(a.dtype), code_of(out.dtype)
    args.bias = _vec(bias, N, "bias")
    args.act = act
    args.gamma = _vec(gamma, N, "gamma")
    if residual is not None:
        in_place_bf16 = residual.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and residual.data_ptr() == out.data_ptr()
        if residual.dtype != torch.float32 and not in_place_bf16:
            raise L.M324Error("gemm: residual must be fp32 (or the bf16 output itself, updated in place)")
        args.residual, args.ldr = _rows(residual, "residual")
        args.res_rows = res_rows
    gin, gout, off = row_map
    need = ((M - 1) // gin * gout + (M - 1) % gin + off + 1) if gin > 0 else M
    if out.shape[0] < need or out.shape[1] < N:
        raise L.M324Error(f"gemm: out{tuple(out.shape)} too small for {need} x {N}")
    args.row_gin, args.row_gout, args.row_off = gin, gout, off
    auxes = [(t, mode) for t, mode in ((preact_out, 1), (gelu_grad_of, 2), (gelu_grad_out, 6), (mul_by, 7)) if t is not None]
    if auxes:
        if len(auxes) > 1:
            raise L.M324Error("gemm: preact_out, gelu_grad_of, gelu_grad_out and mul_by are mutually exclusive")
        aux, mode = auxes[0]
        if aux.dtype != out.dtype or aux.shape[0] < M or aux.shape[1] < N:
            raise L.M324Error(f"gemm: aux operand {aux.dtype}{tuple(aux.shape)} does not match out {out.dtype} [{M}, {N}]")
        args.aux, args.ldaux = _rows(aux, "aux")
        args.aux_mode = mode
    esz = a.element_size()
    tag = "" if not _timing() else (
        f"{_gemm_plan(args)} | M={M} N={N} K={K}{' bias' if bias is not None else ''}{' gelu' if act else ''}"
        f"{' gamma' if gamma is not None else ''}{' res' if residual is not None else ''} out={'bf16' if out.element_size() == 2 else 'f32'}"
        f"{' ln-fold' if ln is not None else ''}{' ln-stats' if stats_out is not None else ''}")
    # algorithmic bytes: both operands once, the output once, the fp32 residual rows once
    nbytes = esz * (M * K + N * K) + out.element_size() * M * N
    if residual is not None:
        nbytes += residual.element_size() * N * (res_rows if 0 < res_rows < M else M)
    if copy_out is not None:
        nbytes += 2 * M * N
    with span(f"gemm_{'bf16' if esz == 2 else 'f32'}", 2.0 * M * N * K, nbytes, tag):
        L.check(L.load().m324_gemm(C.byref(args), _stream()), "m324_gemm")
    return out


def gemm_pair(deferred: list) -> None:
    """Launch the two projections prepared with gemm(..., qkv_heads=..., defer=deferred) as one kernel (m324_gemm_pair: horizontal
    fusion of two small latency-bound GEMMs); pairs the library does not build run as two m324_gemm launches."""
    if len(deferred) != 2:
        raise L.M324Error("gemm_pair: exactly two deferred projections")
    (a0, f0, b0, t0, _), (a1, f1, b1, t1, _) = deferred
    lib = L.load()
    with span("gemm_bf16", f0 + f1, b0 + b1, f"gemm_ring2_pair_kernel<unsigned short, 4, 0> | {t0} + {t1}" if _timing() else "") as sp:
        rc = lib.m324_gemm_pair(C.byref(a0), C.byref(a1), _stream())
        if rc == L.ERR_UNSUPPORTED:
            sp.cancel()                          # nothing ran: the two fallback launches below carry the work
    if rc == L.ERR_UNSUPPORTED:
        for args, fl, by, tag, _ in deferred:
            with span("gemm_bf16", fl, by, f"{_gemm_plan(args)} | {tag}" if _timing() else ""):
                L.check(lib.m324_gemm(C.byref(args), _stream()), "m324_gemm")
    else:
        L.check(rc, "m324_gemm_pair")
    deferred.clear()


def layernorm_pair(x0, w0, b0, eps0, out0, x1, w1, b1, eps1, out1, row_map1=(0, 0, 0)) -> None:
    """Two fp32 -> out-dtype LayerNorms of the same width in one launch (m324_layernorm_pair); the second may gather its rows
    (row_map1 as in layernorm)."""
    Cdim = x0.shape[1]
    if (x0.dtype != torch.float32 or x1.dtype != torch.float32 or x1.shape[1] != Cdim or out0.shape[1] != Cdim or out1.shape[1] != Cdim
            or out0.dtype != out1.dtype or out0.shape[0] > x0.shape[0]):
        raise L.M324Error("layernorm_pair: two fp32 inputs of one width, outputs of one dtype")
    gin, gout, off = row_map1
    rows1 = out1.shape[0]
    need = ((rows1 - 1) // gin * gout + (rows1 - 1) % gin + off + 1) if gin > 0 else rows1
    if x1.shape[0] < need:
        raise L.M324Error("layernorm_pair: shape mismatch")
    (p0, l0), (q0, m0), (p1, l1), (q1, m1) = _rows(x0, "x0"), _rows(out0, "out0"), _rows(x1, "x1"), _rows(out1, "out1")
    tag = f"layernorm_pair_kernel | rows={out0.shape[0]}+{rows1} C={Cdim}" if _timing() else ""
    with span("hbm_pass", 0.0, float(out0.shape[0] + rows1) * Cdim * (4 + out0.element_size()), tag):
        L.check(L.load().m324_layernorm_pair(p0, l0, _vec(w0, Cdim, "w0"), _vec(b0, Cdim, "b0"), eps0, q0, m0, out0.shape[0], 0, 0, 0,
                                             p1, l1, _vec(w1, Cdim, "w1"), _vec(b1, Cdim, "b1"), eps1, q1, m1, rows1, gin, gout, off,
                                             Cdim, code_of(out0.dtype), _stream()), "m324_layernorm_pair")
    _wrote(out0, out1)


def rowstats_finish(part: torch.Tensor, eps: float, rowstat: torch.Tensor) -> torch.Tensor:
    """rowstat[m] = (rstd, -rstd mean) from the per-block (sum, M2) pairs part[ncb, M, 2] a producer GEMM left."""
    ncb, M, two = part.shape
    if (two != 2 or part.dtype != torch.float32 or not part.is_contiguous() or rowstat.dtype != torch.float32
            or not rowstat.is_contiguous() or rowstat.numel() != 2 * M):
        raise L.M324Error("rowstats_finish: part [ncb, M, 2] and rowstat [M, 2] must be contiguous fp32")
    with span("hbm_pass", 0.0, 8.0 * (ncb + 1) * M, f"rowstats_finish_kernel | rows={M} blocks={ncb}" if _timing() else ""):
        L.check(L.load().m324_rowstats_finish(_p(part), ncb, M, eps, _p(rowstat), _stream()), "m324_rowstats_finish")
    return rowstat


def rowstats(x: torch.Tensor, eps: float, rowstat: torch.Tensor, copy: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The same table straight from an fp32 stream x [rows, C], plus its bf16 twin `copy` (head of a folded chain)."""
    px, ldx = _rows(x, "x")
    rows, Cdim = x.shape
    if x.dtype != torch.float32 or rowstat.dtype != torch.float32 or not rowstat.is_contiguous() or rowstat.numel() != 2 * rows:
        raise L.M324Error("rowstats: x fp32 [rows, C], rowstat contiguous fp32 [rows, 2]")
    pc, ldc = (None, 0)
    if copy is not None:
        if copy.dtype != torch.bfloat16 or copy.shape[0] < rows or copy.shape[1] != Cdim:
            raise L.M324Error("rowstats: copy must be bf16 [rows, C]")
        pc, ldc = _rows(copy, "copy")
    with span("hbm_pass", 0.0, rows * Cdim * (4.0 + (2.0 if copy is not None else 0.0)), f"rowstats_kernel | rows={rows} C={Cdim}" if _timing() else ""):
        L.check(L.load().m324_rowstats(px, ldx, rows, Cdim, eps, _p(rowstat), pc, ldc, _stream()), "m324_rowstats")
    return rowstat


def layernorm(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], eps: float, out: torch.Tensor,
              rows: Optional[int] = None, row_map=(0, 0, 0)) -> torch.Tensor:
    if x.dtype != torch.float32 and not (x.dtype == torch.bfloat16 and out.dtype == torch.bfloat16):
        raise L.M324Error("layernorm: x must be fp32 (or bf16 with a bf16 output)")
    px, ldx = _rows(x, "x")
    py, ldy = _rows(out, "out")
    Cdim = x.shape[1]
    rows = out.shape[0] if rows is None else rows
    gin, gout, off = row_map
    need = ((rows - 1) // gin * gout + (rows - 1) % gin + off + 1) if gin > 0 else rows
    if x.shape[0] < need or out.shape[0] < rows or out.shape[1] != Cdim:
        raise L.M324Error("layernorm: shape mismatch")
    tag = (f"layernorm_kernel<{'unsigned short' if out.element_size() == 2 else 'float'}, {'unsigned short' if x.element_size() == 2 else 'float'}>"
           f" | rows={rows} C={Cdim}") if _timing() else ""
    with span("hbm_pass", 0.0, float(rows) * Cdim * (x.element_size() + out.element_size()), tag):
        if x.dtype == torch.bfloat16:
            L.check(L.load().m324_layernorm_in(px, BF16, ldx, _vec(w, Cdim, "w"), _vec(b, Cdim, "b"), eps, py, ldy, code_of(out.dtype),
                                               rows, Cdim, gin, gout, off, _stream()), "m324_layernorm_in")
        else:
            L.check(L.load().m324_layernorm(px, ldx, _vec(w, Cdim, "w"), _vec(b, Cdim, "b"), eps, py, ldy, code_of(out.dtype),
                                            rows, Cdim, gin, gout, off, _stream()), "m324_layernorm")
    return out


LOG2E = 1.4426950408889634
Q_PRESCALE = (64 ** -0.5) * LOG2E      # softmax scale of head_dim 64, log2 domain


def qkv_split(q_src, k_src, v_src, q_w, k_w, eps: float, B: int, Lq: int, H: int, dtype: torch.dtype, *,
              q_scale: float = 1.0, train: bool = False):
    """Head-major attention operands from token-major sources (any source may be None).

    Inference (train=False): returns (Q[B,H,L,64], K[B,H,L,64], Vt[B,H,64,Lp]).
    Training  (train=True):  returns a dict with Q, K, V (row-major) and Qt, Kt, Vt (transposed, permuted) for every
    given source -- what m324_attention_bwd needs.  All given sources share B and L."""
    dev = next(t for t in (q_src, k_src, v_src) if t is not None).device
    Lp = (Lq + 63) // 64 * 64

    def row():
        return torch.empty((B, H, Lq, 64), dtype=dtype, device=dev)

    def tr():
        return torch.empty((B, H, 64, Lp), dtype=dtype, device=dev)
    Q = row() if q_src is not None else None
    K = row() if k_src is not None else None
    V = row() if (v_src is not None and train) else None
    Qt = tr() if (q_src is not None and train) else None
    Kt = tr() if (k_src is not None and train) else None
    Vt = tr() if v_src is not None else None

    def src(t, name):
        if t is None:
            return None, 0
        if t.dtype != dtype or t.shape[0] < B * Lq:
            raise L.M324Error(f"qkv_split: {name} {t.dtype}{tuple(t.shape)}")
        return _rows(t, name)
    pq, ldq = src(q_src, "q_src")
    pk, ldk = src(k_src, "k_src")
    pv, ldv = src(v_src, "v_src")
    L.check(L.load().m324_qkv_split(pq, ldq, pk, ldk, pv, ldv, _vec(q_w, 64, "q_w"), _vec(k_w, 64, "k_w"), eps, q_scale,
                                    _p(Q), _p(K), _p(V), _p(Qt), _p(Kt), _p(Vt), B, Lq, H, code_of(dtype), _stream()),
            "m324_qkv_split")
    if train:
        return {"Q": Q, "K": K, "V": V, "Qt": Qt, "Kt": Kt, "Vt": Vt}
    return Q, K, Vt


def attention(Q: torch.Tensor, K: torch.Tensor, Vt: torch.Tensor, out: torch.Tensor, *, shared_q: bool = False,
              scale: Optional[float] = None, prescaled: bool = False, lse: Optional[torch.Tensor] = None,
              v_rowmajor: bool = False, bounded: bool = False) -> torch.Tensor:
    """out[B*Lq, H*64] = softmax(Q K^T scale) V.  Q[Bq,H,Lq,64] (Bq == 1 with shared_q), K[B,H,Lk,64],
    Vt[B,H,64,Lkp] -- or, with v_rowmajor (bf16 only), V[B,H,Lk,64].  prescaled: Q carries Q_PRESCALE.
    bounded: the caller vouches for |log2-domain score| <= 64 (M324_ATTN_SCORES_BOUNDED; see qk_score_bound)."""
    B, H, Lk, D = K.shape
    Lq = Q.shape[2]
    vshape = (B, H, Lk, 64) if v_rowmajor else (B, H, 64, (Lk + 63) // 64 * 64)
    if D != 64 or Q.shape[3] != 64 or Q.shape[1] != H or tuple(Vt.shape) != vshape:
        raise L.M324Error(f"attention: Q{tuple(Q.shape)} K{tuple(K.shape)} V{tuple(Vt.shape)} (want {vshape})")
    if not (Q.is_contiguous() and K.is_contiguous() and Vt.is_contiguous()):
        raise L.M324Error("attention: operands must be contiguous")
    if Q.dtype != K.dtype or K.dtype != Vt.dtype or out.dtype != Q.dtype:
        raise L.M324Error("attention: dtype mismatch")
    if not shared_q and Q.shape[0] != B:
        raise L.M324Error("attention: batch mismatch")
    po, ldo = _rows(out, "out")
    if out.shape[0] < B * Lq or out.shape[1] < H * 64:
        raise L.M324Error("attention: out too small")
    qbs = 0 if shared_q else H * Lq * 64
    scale = 64 ** -0.5 if scale is None else scale
    esz = Q.element_size()
    with span(f"attention_{'bf16' if esz == 2 else 'f32'}", 4.0 * B * H * Lq * Lk * 64,
              esz * 64.0 * H * ((1 if shared_q else B) * Lq + 2 * B * Lk + B * Lq),
              f"{_attn_plan(B, H, Lq, Lk, int(prescaled) | (2 if v_rowmajor else 0) | (4 if bounded else 0) | (256 if shared_q else 0), code_of(Q.dtype))} | B={B} H={H} Lq={Lq} Lk={Lk}"
              if _timing() else ""):
        L.check(L.load().m324_attention(_p(Q), qbs, _p(K), _p(Vt), po, ldo, B, H, Lq, Lk, scale,
                                        int(prescaled) | (2 if v_rowmajor else 0) | (4 if bounded else 0), _p(lse), code_of(Q.dtype), _stream()),
                "m324_attention")
    return out


def attention_merge(parts, out: torch.Tensor, B: int, H: int, Lq: int) -> torch.Tensor:
    """out[B*Lq, H*64] = the attention over the union of two or three disjoint key sets from the (O_i, lse_i) pairs attention(...,
    lse=...) left for each of them (m324_attention_merge: log2-domain weights 2^(lse_i - max))."""
    if not 2 <= len(parts) <= 3:
        raise L.M324Error("attention_merge: two or three (output, lse) parts")
    O0 = parts[0][0]
    p0, ldp = _rows(O0, "O0")
    ptrs = []
    for O, lse in parts:
        po, ld = _rows(O, "part")
        if (O.dtype != out.dtype or ld != ldp or O.shape[0] < B * Lq or O.shape[1] < H * 64 or lse.dtype != torch.float32
                or not lse.is_contiguous() or lse.numel() != B * H * Lq):
            raise L.M324Error(f"attention_merge: part {O.dtype}{tuple(O.shape)} / lse {lse.dtype}{tuple(lse.shape)}")
        ptrs += [po, _p(lse)]
    if len(parts) == 2:
        ptrs += [None, None]
    po, ldo = _rows(out, "out")
    if out.shape[0] < B * Lq or out.shape[1] < H * 64:
        raise L.M324Error("attention_merge: out too small")
    esz = out.element_size()
    with span("hbm_pass", 0.0, float(B * Lq) * H * 64 * esz * (len(parts) + 1),
              f"attention_merge_kernel | rows={B * Lq} parts={len(parts)}" if _timing() else ""):
        L.check(L.load().m324_attention_merge(*ptrs, ldp, po, ldo, B, H, Lq, code_of(out.dtype), _stream()), "m324_attention_merge")
    _wrote(out)
    return out


def patchify(video: torch.Tensor, size: int, patch: int, Kp: int, dtype: torch.dtype) -> torch.Tensor:
    """video [F,Hin,Win,3] fp32 in [0,1] -- or uint8 in 0..255, converted per tap as v / 255 (m324_patchify_u8: the rows equal the
    fp32 path's on `video.float() / 255` bit for bit) -> [F*(size/patch)^2, Kp] normalised, bilinearly resized patch rows."""
    if video.dtype not in (torch.float32, torch.uint8) or not video.is_contiguous() or video.dim() != 4 or video.shape[3] != 3:
        raise L.M324Error(f"patchify: video {video.dtype}{tuple(video.shape)}")
    Fr, Hin, Win, _ = video.shape
    g = size // patch
    out = torch.empty((Fr * g * g, Kp), dtype=dtype, device=video.device)
    fn, name = (L.load().m324_patchify_u8, "m324_patchify_u8") if video.dtype == torch.uint8 else (L.load().m324_patchify, "m324_patchify")
    L.check(fn(_p(video), Fr, Hin, Win, size, patch, _p(out), Kp, code_of(dtype), _stream()), name)
    _wrote(out)
    return out


def point_encode(xyz: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    xyz = xyz.reshape(-1, 3)
    if xyz.dtype != torch.float32 or not xyz.is_contiguous():
        raise L.M324Error("point_encode: xyz must be contiguous fp32")
    out = torch.empty((xyz.shape[0], 64), dtype=dtype, device=xyz.device)
    L.check(L.load().m324_point_encode(_p(xyz), xyz.shape[0], _p(out), 64, code_of(dtype), _stream()), "m324_point_encode")
    return out


def point_concat(normal: torch.Tensor, rgb: torch.Tensor, feat: torch.Tensor, Cdim: int) -> torch.Tensor:
    normal, rgb = normal.reshape(-1, 3), rgb.reshape(-1, 3)
    for t in (normal, rgb):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise L.M324Error("point_concat: normal/rgb must be contiguous fp32")
    P, Kp = feat.shape
    if not feat.is_contiguous() or normal.shape[0] != P or rgb.shape[0] != P:
        raise L.M324Error("point_concat: shape mismatch")
    L.check(L.load().m324_point_concat(_p(normal), _p(rgb), P, _p(feat), Cdim, Kp, code_of(feat.dtype), _stream()),
            "m324_point_concat")
    return feat


def dino_cls_rows(cls: torch.Tensor, pos0: torch.Tensor, x: torch.Tensor, Fr: int, rows_per_frame: int) -> None:
    Cdim = x.shape[1]
    L.check(L.load().m324_dino_cls_rows(_vec(cls, Cdim, "cls"), _vec(pos0, Cdim, "pos0"), _p(x), Fr, rows_per_frame, Cdim,
                                        _stream()), "m324_dino_cls_rows")


def assemble_tokens(dino_x, dino_w, dino_b, eps_dino, pos, sp0, spr, mesh, ln_w, eps_in, B, T, K, P,
                    drop_p: float = 0.0, drop_seed: int = 0) -> torch.Tensor:
    Cdim = dino_x.shape[1]
    for name, t, n in (("dino_x", dino_x, B * T * (P + 1) * Cdim), ("pos", pos, T * P * Cdim), ("sp0", sp0, 4 * Cdim),
                       ("spr", spr, 4 * Cdim), ("mesh", mesh, B * K * Cdim)):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise L.M324Error(f"assemble_tokens: {name} {t.dtype}{tuple(t.shape)} (want {n} fp32)")
    out = torch.empty((B * T * (4 + K + P), Cdim), dtype=torch.float32, device=dino_x.device)
    L.check(L.load().m324_assemble_tokens(_p(dino_x), _vec(dino_w, Cdim, "dino_w"), _vec(dino_b, Cdim, "dino_b"), eps_dino,
                                          _p(pos), _p(sp0), _p(spr), _p(mesh), _vec(ln_w, Cdim, "ln_w") if ln_w is not None else None, eps_in, _p(out),
                                          B, T, K, P, Cdim, float(drop_p), int(drop_seed) & 0xFFFFFFFFFFFFFFFF, _stream()),
            "m324_assemble_tokens")
    return out


def linear_n3(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    M, K = a.shape
    pa, lda = _rows(a, "a")
    if w.dtype != torch.float32 or not w.is_contiguous() or tuple(w.shape) != (3, K):
        raise L.M324Error("linear_n3: w must be contiguous fp32 [3,K]")
    if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != M * 3:
        raise L.M324Error("linear_n3: out must be contiguous fp32 [M,3]")
    L.check(L.load().m324_linear_n3(pa, lda, _p(w), _vec(bias, 3, "bias"), _p(out), M, K, code_of(a.dtype), _stream()),
            "m324_linear_n3")
    return out


def n3_finish(part: torch.Tensor, bias: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[M, 3] (fp32, contiguous) = bias + sum over the column blocks of part[ncb, M, 3] (second half of gemm(..., n3=...))."""
    ncb, M, three = part.shape
    if three != 3 or part.dtype != torch.float32 or not part.is_contiguous() or out.dtype != torch.float32 \
            or not out.is_contiguous() or out.numel() != M * 3:
        raise L.M324Error("n3_finish: part [ncb, M, 3] and out [M, 3] must be contiguous fp32")
    L.check(L.load().m324_n3_finish(_p(part), ncb, M, _vec(bias, 3, "bias"), _p(out), _stream()), "m324_n3_finish")
    return out


def mse(pred: torch.Tensor, target: torch.Tensor, weight: float) -> torch.Tensor:
    if pred.shape != target.shape:
        raise L.M324Error(f"mse: shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")
    pred = pred.contiguous().float()
    target = target.contiguous().float()
    partial = torch.empty(1024, dtype=torch.float32, device=pred.device)
    out = torch.empty((), dtype=torch.float32, device=pred.device)
    L.check(L.load().m324_mse(_p(pred), _p(target), pred.numel(), weight, _p(partial), _p(out), _stream()), "m324_mse")
    return out


def smooth_trajectories(trajs: torch.Tensor, threshold: float, sigma: float) -> torch.Tensor:
    """[B,T,N,3] fp32 -> thresholded (threshold >= 0) and/or gaussian-filtered (sigma > 0) trajectories."""
    if trajs.dim() != 4 or trajs.shape[3] != 3:
        raise L.M324Error(f"smooth_trajectories: expected [B,T,N,3], got {tuple(trajs.shape)}")
    x = trajs.detach().to(torch.float32).contiguous()
    B, T, N, _ = x.shape
    tmp, out = torch.empty_like(x), torch.empty_like(x)
    L.check(L.load().m324_smooth_trajectories(_p(x), _p(tmp), _p(out), B, T, N, threshold, sigma, _stream()),
            "m324_smooth_trajectories")
    return out


def smooth_savgol(trajs: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
    """[B,T,N,3] fp32 filtered along T with the odd-length fp64 coefficient vector `coef` (clamped borders)."""
    x = trajs.detach().to(torch.float32).contiguous()
    B, T, N, _ = x.shape
    if coef.dtype != torch.float64 or not coef.is_cuda or not coef.is_contiguous():
        raise L.M324Error("smooth_savgol: coef must be a contiguous fp64 HIP tensor")
    out = torch.empty_like(x)
    L.check(L.load().m324_smooth_savgol(_p(x), _p(out), B, T, N, _p(coef), coef.numel(), _stream()), "m324_smooth_savgol")
    return out


def smooth_oneeuro(trajs: torch.Tensor, mincutoff: float, beta: float, dcutoff: float = 1.0) -> torch.Tensor:
    x = trajs.detach().to(torch.float32).contiguous()
    B, T, N, _ = x.shape
    out = torch.empty_like(x)
    L.check(L.load().m324_smooth_oneeuro(_p(x), _p(out), B, T, N, mincutoff, beta, dcutoff, _stream()), "m324_smooth_oneeuro")
    return out


def nearest_point(query: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    """query [Nq,3], ref [Nr,3] fp32 -> int64 [Nq] index of the nearest reference point."""
    q = query.detach().to(torch.float32).contiguous()
    r = ref.detach().to(torch.float32).contiguous()
    if q.dim() != 2 or q.shape[1] != 3 or r.dim() != 2 or r.shape[1] != 3:
        raise L.M324Error(f"nearest_point: expected [n,3] point sets, got {tuple(q.shape)} / {tuple(r.shape)}")
    idx = torch.empty((q.shape[0],), dtype=torch.int32, device=q.device)
    L.check(L.load().m324_nearest_point(_p(q), q.shape[0], _p(r), r.shape[0], _p(idx), _stream()), "m324_nearest_point")
    return idx.long()


# ------------------------------------------------------------------------------------------------ training side
def transpose(x: torch.Tensor, rows_pad: Optional[int] = None) -> torch.Tensor:
    """[R, C] -> [C, rows_pad] (rows_pad = round_up(R, 64) by default; the pad columns are zeros)."""
    R, Cc = x.shape
    px, ld = _rows(x, "x")
    rp = (R + 63) // 64 * 64 if rows_pad is None else rows_pad
    out = torch.empty((Cc, rp), dtype=x.dtype, device=x.device)
    L.check(L.load().m324_transpose(px, ld, _p(out), rp, R, Cc, rp, code_of(x.dtype), _stream()), "m324_transpose")
    return out


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    R, Cc = x.shape
    px, ld = _rows(x, "x")
    if out is None:
        out = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        accumulate = False
    elif COLSUMS.busy(out):                        # queued sums into this destination come first
        COLSUMS.flush()
    scratch, srows = None, 0
    if R >= 256:                                   # two-stage: row chunks in parallel, then a short fixed-order sum
        srows = 256 if R >= 16384 else (64 if R >= 2048 else 16)
        scratch = torch.empty((srows, Cc), dtype=torch.float32, device=x.device)
    L.check(L.load().m324_colsum(px, ld, _vec(out, Cc, "out"), R, Cc, code_of(x.dtype), int(accumulate), _p(scratch), srows,
                                 _stream()), "m324_colsum")
    return out


class _ColsumQueue:
    """Column sums that nobody reads yet, collected and launched together (m324_colsum_multi): the split-K partials of the weight
    gradients, the per-workgroup partials of LayerNorm / RMSNorm weight gradients.  ``defer(src, dst, accumulate)`` = what
    ``colsum(src, out=dst, accumulate=...)`` would do, later.  At the flush the entries are grouped by destination (order kept
    within a destination; destinations are independent of each other): one destination = one chain, summed in order by the same
    threads -- the arithmetic of consecutive m324_colsum calls.  ``flush()`` MUST run before anyone reads or torch-writes a
    destination (backward.GradStore does: done(), add() on a gradient it already holds, get(); colsum(out=...) checks too)."""

    LIMIT = 256

    def __init__(self):
        self.by_dst = {}         # data_ptr of a destination -> [(dst 1-D fp32 view, src 2-D fp32, accumulate)]
        self.count = 0

    def defer(self, src: torch.Tensor, dst: torch.Tensor, accumulate: bool) -> None:
        if src.dtype != torch.float32 or dst.dtype != torch.float32 or src.dim() != 2 or src.stride(1) != 1 or not dst.is_contiguous() \
                or dst.numel() != src.shape[1]:
            raise L.M324Error(f"colsum queue: fp32 [rows, cols] -> contiguous fp32 [cols], got {src.dtype}{tuple(src.shape)} -> {dst.dtype}{tuple(dst.shape)}")
        chain = self.by_dst.setdefault(dst.data_ptr(), [])
        if chain and not accumulate:
            raise L.M324Error("colsum queue: a second sum into a pending destination must accumulate")
        chain.append((dst.view(-1), src, bool(accumulate)))
        self.count += 1
        if self.count >= self.LIMIT:
            self.flush()

    def busy(self, t: torch.Tensor) -> bool:
        return self.count > 0 and t.data_ptr() in self.by_dst

    def clear(self) -> None:
        """Drops the queue (a step that failed half way: its partial buffers are gone)."""
        self.by_dst, self.count = {}, 0

    def flush(self) -> None:
        if not self.count:
            return
        arr = (L.ColsumItem * self.count)()
        k, dsts = 0, []
        # tall sources first: their workgroups (16 waves over up to 1024 rows) are the longest-running of the launch
        for chain in sorted(self.by_dst.values(), key=lambda ch: -max(src.shape[0] for _, src, _a in ch)):
            for j, (dst, src, acc) in enumerate(chain):
                it = arr[k]
                it.dst, it.src, it.ld, it.rows, it.cols = dst.data_ptr(), src.data_ptr(), src.stride(0), src.shape[0], src.shape[1]
                it.accumulate, it.chain = int(acc and j == 0), int(j > 0)
                k += 1
            dsts.append(chain[0][0])
        keep = self.by_dst                                # the sources stay referenced until the launch is enqueued
        self.by_dst, self.count = {}, 0
        L.check(L.load().m324_colsum_multi(arr, k, _stream()), "m324_colsum_multi")
        _wrote(*dsts)
        del keep


class Rows:
    """Marks an fp32 [rows, cols] tensor whose COLUMN SUMS are a gradient (the per-workgroup partials of a norm-weight gradient):
    backward.GradStore.add queues the sum into the gradient's own memory instead of taking a reduced temporary and copying it."""

    def __init__(self, t: torch.Tensor):
        self.t = t

    def reduce(self) -> torch.Tensor:
        return colsum(self.t)


COLSUMS = _ColsumQueue()
DEFER_COLSUM = switches.flag("M324_DEFER_COLSUM")


def gelu(z: torch.Tensor) -> torch.Tensor:
    assert z.is_contiguous()
    h = torch.empty_like(z)
    L.check(L.load().m324_gelu(_p(z), _p(h), z.numel(), code_of(z.dtype), _stream()), "m324_gelu")
    return h


def gelu_bwd(z: torch.Tensor, dh: torch.Tensor) -> torch.Tensor:
    assert z.is_contiguous() and dh.is_contiguous() and z.shape == dh.shape and z.dtype == dh.dtype
    dz = torch.empty_like(z)
    L.check(L.load().m324_gelu_bwd(_p(z), _p(dh), _p(dz), z.numel(), code_of(z.dtype), _stream()), "m324_gelu_bwd")
    return dz


def layernorm_bwd(x: torch.Tensor, w: torch.Tensor, eps: float, dy: torch.Tensor, dx: torch.Tensor, accumulate: bool,
                  row_map=(0, 0, 0), cast_out: Optional[torch.Tensor] = None, reduce: bool = True):
    """dx[in_row(r)] (+)= LN backward of row r; returns (dw [C], db [C]) fp32.  x, dx fp32; dy in the compute dtype.
    cast_out (bf16, shaped like dx): also receives the resulting dx rounded to bf16, and a third vector is returned: the
    column sums of that rounded copy (m324_layernorm_bwd_cast: one pass instead of LayerNorm backward + cast + column sum).
    reduce=False: the vectors come back as ops.Rows -- the per-workgroup partial rows, whose column sums the caller queues
    (backward.GradStore.add) instead of reducing them here with a launch each."""
    if x.dtype != torch.float32 or dx.dtype != torch.float32:
        raise L.M324Error("layernorm_bwd: x and dx must be fp32")
    rows, Cdim = dy.shape
    px, ldx = _rows(x, "x")
    pdy, ldy = _rows(dy, "dy")
    pdx, lddx = _rows(dx, "dx")
    _wrote(dx)
    # workgroups of 8 waves, one row per wave at a time; their partial rows are summed by m324_colsum (two stages from 256 rows
    # on) -- or (reduce=False) queued for the block's m324_colsum_multi launch
    n_partial = min(512, (rows + 7) // 8)
    nb = 2 if cast_out is None else 3
    partial = torch.empty((n_partial, nb * Cdim), dtype=torch.float32, device=x.device)
    gin, gout, off = row_map
    if cast_out is None:
        L.check(L.load().m324_layernorm_bwd(px, ldx, _vec(w, Cdim, "w"), eps, pdy, ldy, code_of(dy.dtype), pdx, lddx,
                                            int(accumulate), _p(partial), n_partial, rows, Cdim, gin, gout, off, _stream()),
                "m324_layernorm_bwd")
        if not reduce:
            return Rows(partial[:, :Cdim]), Rows(partial[:, Cdim:])
        both = colsum(partial)
        return both[:Cdim], both[Cdim:]
    if cast_out.dtype != torch.bfloat16 or cast_out.shape != dx.shape:
        raise L.M324Error("layernorm_bwd: cast_out must be a bf16 tensor shaped like dx")
    pc, ldc = _rows(cast_out, "cast_out")
    _wrote(cast_out)
    L.check(L.load().m324_layernorm_bwd_cast(px, ldx, _vec(w, Cdim, "w"), eps, pdy, ldy, code_of(dy.dtype), pdx, lddx,
                                             int(accumulate), _p(partial), n_partial, rows, Cdim, gin, gout, off, pc, ldc,
                                             _stream()), "m324_layernorm_bwd_cast")
    if not reduce:
        return Rows(partial[:, :Cdim]), Rows(partial[:, Cdim:2 * Cdim]), Rows(partial[:, 2 * Cdim:])
    three = colsum(partial)
    return three[:Cdim], three[Cdim:2 * Cdim], three[2 * Cdim:]


def cast(x: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    R, Cc = x.shape
    px, ld = _rows(x, "x")
    if out is None:
        out = torch.empty((R, Cc), dtype=dtype, device=x.device)
    po, ldo = _rows(out, "out")
    L.check(L.load().m324_cast(px, ld, code_of(x.dtype), po, ldo, code_of(out.dtype), R, Cc, _stream()), "m324_cast")
    return out


def attention_delta(O: torch.Tensor, dO: torch.Tensor, B: int, H: int, Lq: int) -> torch.Tensor:
    po, ld = _rows(O, "O")
    pd, ld2 = _rows(dO, "dO")
    if ld != ld2 or O.dtype != dO.dtype:
        raise L.M324Error("attention_delta: O and dO must share layout and dtype")
    D = torch.empty((B, H, Lq), dtype=torch.float32, device=O.device)
    L.check(L.load().m324_attention_delta(po, pd, ld, _p(D), B, H, Lq, code_of(O.dtype), _stream()), "m324_attention_delta")
    return D


def attention_bwd(Qs, K, V, dO, lse, D, *, shared_q: bool = False, scale: float = 64 ** -0.5):
    """Head-major operands (see include/m324.h) -> (dQ [B,H,Lq,64] w.r.t. the unscaled normalised q, dK, dV)."""
    B, H, Lk, _ = K.shape
    Lq = Qs.shape[2]
    for t in (Qs, K, V, dO):
        if not t.is_contiguous() or t.dtype != K.dtype:
            raise L.M324Error("attention_bwd: operands must be contiguous and share a dtype")
    dQ = torch.empty((B, H, Lq, 64), dtype=K.dtype, device=K.device)
    dK, dV = torch.empty_like(K), torch.empty_like(V)
    L.check(L.load().m324_attention_bwd(_p(Qs), 0 if shared_q else H * Lq * 64, _p(K), _p(V), _p(dO), _p(lse), _p(D), _p(dQ),
                                        _p(dK), _p(dV), B, H, Lq, Lk, scale, code_of(K.dtype), _stream()), "m324_attention_bwd")
    return dQ, dK, dV


def attention_bwd_mfma(spq: dict, spk: dict, spdo: dict, lse, D, *, shared_q: bool = False, scale: float = 64 ** -0.5):
    """bf16 MFMA backward.  spq / spk / spdo: the dicts qkv_split(train=True) returns for q, for k|v and for dO
    (passed as its q source).  Returns (dQ, dK, dV) like attention_bwd."""
    Qs, Qst, K, Kt, V = spq["Q"], spq["Qt"], spk["K"], spk["Kt"], spk["V"]
    dO, dOt = spdo["Q"], spdo["Qt"]
    B, H, Lk, _ = K.shape
    Lq = Qs.shape[2]
    Lqp = (Lq + 63) // 64 * 64
    if K.dtype != torch.bfloat16:
        raise L.M324Error("attention_bwd_mfma is the bf16 kernel; use attention_bwd for fp32")
    dQ = torch.empty((B, H, Lq, 64), dtype=K.dtype, device=K.device)
    dK, dV = torch.empty_like(K), torch.empty_like(V)
    L.check(L.load().m324_attention_bwd_mfma(_p(Qs), _p(Qst), 0 if shared_q else H * Lq * 64, 0 if shared_q else H * 64 * Lqp,
                                             _p(K), _p(Kt), _p(V), _p(dO), _p(dOt), _p(lse), _p(D), _p(dQ), _p(dK), _p(dV),
                                             B, H, Lq, Lk, scale, _stream()), "m324_attention_bwd_mfma")
    return dQ, dK, dV


def qkv_split_bwd(dQ, dK, dV, q_raw, k_raw, q_w, k_w, eps: float, B: int, Lq: int, H: int, dq_out, dk_out, dv_out, reduce: bool = True):
    """Writes token-major gradients into dq_out / dk_out / dv_out (2-D views, any may be None with its dX);
    returns (dq_norm_w [64] or None, dk_norm_w [64] or None)."""
    dtype = next(t for t in (dQ, dK, dV) if t is not None).dtype
    n_partial = min(1024, max(1, (B * Lq * H + 31) // 32))      # a workgroup takes 32 (token, head) rows per pass
    dev = next(t for t in (dQ, dK, dV) if t is not None).device
    partial = torch.empty((n_partial, 128), dtype=torch.float32, device=dev)

    def rw(t):
        return _rows(t, "view") if t is not None else (None, 0)
    pq, ldq = rw(q_raw)
    pk, ldk = rw(k_raw)
    poq, ldoq = rw(dq_out)
    pok, ldok = rw(dk_out)
    pov, ldov = rw(dv_out)
    L.check(L.load().m324_qkv_split_bwd(_p(dQ), _p(dK), _p(dV), pq, ldq, pk, ldk, _vec(q_w, 64, "q_w"), _vec(k_w, 64, "k_w"),
                                        eps, poq, ldoq, pok, ldok, pov, ldov, _p(partial), n_partial, B, Lq, H, code_of(dtype),
                                        _stream()), "m324_qkv_split_bwd")
    if not reduce:                                          # ops.Rows: see layernorm_bwd
        return (Rows(partial[:, :64]) if (dQ is not None and q_w is not None) else None,
                Rows(partial[:, 64:]) if (dK is not None and k_w is not None) else None)
    both = colsum(partial)
    return (both[:64] if (dQ is not None and q_w is not None) else None,
            both[64:] if (dK is not None and k_w is not None) else None)


def linear_n3_bwd(a: torch.Tensor, w: torch.Tensor, dout: torch.Tensor, mul_by: Optional[torch.Tensor] = None):
    """Backward of linear_n3: returns (dA [M,K] in a.dtype, dW [3,K] fp32, db [3] fp32).  mul_by [M, K] (a's dtype): dA is multiplied by it
    (the gelu'(z) the Linear in front left with gemm(gelu_grad_out=...): dA is then the gradient of ITS pre-activation)."""
    M, K = a.shape
    pa, lda = _rows(a, "a")
    dout = dout.reshape(M, 3)
    if dout.dtype != torch.float32 or not dout.is_contiguous():
        raise L.M324Error("linear_n3_bwd: dout must be contiguous fp32 [M,3]")
    pm, ldm = (None, 0)
    if mul_by is not None:
        if mul_by.dtype != a.dtype or mul_by.shape[0] < M or mul_by.shape[1] < K:
            raise L.M324Error(f"linear_n3_bwd: mul_by {mul_by.dtype}{tuple(mul_by.shape)} does not match a {a.dtype}[{M}, {K}]")
        pm, ldm = _rows(mul_by, "mul_by")
    dA = torch.empty((M, K), dtype=a.dtype, device=a.device)
    n_partial = min(1024, M)
    partial = torch.empty((n_partial, 3 * K), dtype=torch.float32, device=a.device)
    L.check(L.load().m324_linear_n3_bwd(pa, lda, _p(w), _p(dout), _p(dA), K, _p(partial), n_partial, M, K, code_of(a.dtype), pm, ldm,
                                        _stream()), "m324_linear_n3_bwd")
    return dA, colsum(partial).reshape(3, K), colsum(dout)


def mse_bwd(pred: torch.Tensor, target: torch.Tensor, weight: float, grad_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    pred, target = pred.contiguous().float(), target.contiguous().float()
    d = torch.empty_like(pred)
    L.check(L.load().m324_mse_bwd(_p(pred), _p(target), _p(grad_scale), 2.0 * weight / pred.numel(), _p(d), pred.numel(),
                                  _stream()), "m324_mse_bwd")
    return d


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float,
               eps: float, weight_decay: float, step: int, grad_scale: Optional[torch.Tensor] = None) -> None:
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
            raise L.M324Error("adamw_step: tensors must be contiguous fp32 of equal size")
    L.check(L.load().m324_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                                _p(grad_scale), _stream()), "m324_adamw")


def adamw_flat(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, n_decay: int, lr: float, beta1: float,
               beta2: float, eps: float, weight_decay: float, step: int, grad_scale: Optional[torch.Tensor] = None) -> None:
    """One launch over the optimizer's flat buffers: elements [0, n_decay) decay, the rest do not (m324_adamw_flat)."""
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
            raise L.M324Error("adamw_flat: tensors must be contiguous fp32 of equal size")
    L.check(L.load().m324_adamw_flat(_p(p), _p(g), _p(m), _p(v), p.numel(), int(n_decay), lr, beta1, beta2, eps, weight_decay,
                                     step, _p(grad_scale), _stream()), "m324_adamw_flat")


def weight_mirror(src: torch.Tensor, dst: torch.Tensor, dstT: Optional[torch.Tensor], table: torch.Tensor, n_items: int, n_tiles: int) -> None:
    """bf16 row-major and transposed copies of the Linear weights inside a flat fp32 parameter buffer, one launch
    (m324_weight_mirror; table: the items as a device byte tensor, see lib.MirrorItem)."""
    if src.dtype != torch.float32 or dst.dtype != torch.bfloat16 or (dstT is not None and dstT.dtype != torch.bfloat16) or table.dtype != torch.uint8 \
            or table.numel() != n_items * C.sizeof(L.MirrorItem) or not table.is_cuda:
        raise L.M324Error("weight_mirror: fp32 source, bf16 copies and a device table of lib.MirrorItem records")
    L.check(L.load().m324_weight_mirror(_p(src), _p(dst), _p(dstT), _p(table), n_items, n_tiles, _stream()), "m324_weight_mirror")


def grad_sumsq(g: torch.Tensor, out: torch.Tensor, partial: torch.Tensor, sanitize: bool, accumulate: bool) -> None:
    L.check(L.load().m324_grad_sumsq(_p(g), g.numel(), int(sanitize), _p(partial), _p(out), int(accumulate), _stream()),
            "m324_grad_sumsq")
