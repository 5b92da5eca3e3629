"""Backward passes of the blocks, built from libm324 kernels (no torch autograd inside).

Every function takes the block's saved INPUT (fp32 residual stream) and either the block's internal activations kept by
the forward or nothing -- then it recomputes them (the reference trains with activation checkpointing per block,
Pcd_motion.py:375-448: a memory policy for 40-80 GB devices; training.py keeps the internals when they fit) -- and
returns the gradient w.r.t. the input; parameter gradients are accumulated into a GradStore in fp32.

Backward GEMMs reuse m324_gemm on transposed operands (include/m324.h "Training-side entry points"):
    dA = dY . W      -> gemm(dY, Wt)          Wt = transposed weight, cached in Prepared
    dW = dY^T . A    -> gemm(dYt, At)         transposes of the two activations, token dim padded to 64
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from . import switches
from .lib import ACT_GELU
from .prepared import Prepared

RMS_EPS = 1e-5
DIRECT_GRADS = switches.get("M324_DIRECT_GRADS") != "0"
ACC_GRADS = switches.flag("M324_ACC_GRADS")
GELU_GRAD_FWD = switches.flag("M324_GELU_GRAD_FWD")


class GradStore:
    """fp32 gradient accumulator keyed by parameter (what autograd's .grad would hold).

    With a ``sink`` (motion324_amd.optim.FusedAdamW) the gradients of the parameters it owns are written straight into
    its flat gradient buffer, and ``done(params)`` tells it that the backward will not touch those parameters again --
    the moment the optimizer may launch the bucket's all-reduce on its side stream (the reference's DDP reducer hooks,
    train.py:88-89,159-166)."""

    def __init__(self, sink=None):
        self.grads: Dict[int, torch.Tensor] = {}
        self.params: Dict[int, torch.nn.Parameter] = {}
        self.sink = sink
        ops.COLSUMS.clear()             # a new step: whatever an aborted one left in the queue belongs to gradients nobody keeps
        if sink is not None:
            sink.begin_step()

    def add(self, param: Optional[torch.nn.Parameter], g) -> None:
        if param is None or not param.requires_grad:
            return
        if isinstance(g, ops.Rows):
            # the gradient is the column sum of g.t (per-workgroup partials): queued straight into the gradient's own memory --
            # the optimizer's flat buffer, the gradient this parameter already holds, or a new tensor; nobody reads it before
            # done() / get() flush the queue
            k = id(param)
            if k in self.grads and self.grads[k].dtype == torch.float32 and self.grads[k].is_contiguous():
                ops.COLSUMS.defer(g.t, self.grads[k].view(-1), True)
                return
            if k not in self.grads:
                view = self.sink.grad_of(param) if (self.sink is not None and self.sink.owns(param)) else \
                    torch.empty(param.shape, dtype=torch.float32, device=g.t.device)
                ops.COLSUMS.defer(g.t, view.view(-1), False)
                self.grads[k], self.params[k] = view, param
                return
            g = g.reduce()
        g = g.reshape(param.shape)
        k = id(param)
        if k in self.grads:
            ops.COLSUMS.flush()         # queued sums into this gradient come first
            self.grads[k] += g          # accumulation of a few small tensors (shared weights): torch add on fp32
            return
        if self.sink is not None and self.sink.owns(param):
            view = self.sink.grad_of(param)
            view.copy_(g)
            self.grads[k] = view
        else:
            self.grads[k] = g.float().clone() if g.dtype != torch.float32 or not g.is_contiguous() else g.clone()
        self.params[k] = param

    def existing(self, param: Optional[torch.nn.Parameter]) -> Optional[torch.Tensor]:
        """The fp32 gradient `param` already holds (contiguous), for kernels that can add into it; None: use add()."""
        if not ACC_GRADS or param is None or not param.requires_grad:
            return None
        g = self.grads.get(id(param))
        return g if g is not None and g.dtype == torch.float32 and g.is_contiguous() else None

    def target(self, param: Optional[torch.nn.Parameter]) -> Optional[torch.Tensor]:
        """Where the FIRST gradient of `param` may be written directly: its slice of the optimizer's flat gradient buffer
        (None: no sink, not owned, or the parameter already holds a gradient -- then add() accumulates).  A kernel that
        wrote there reports with wrote(); 284 device copies per training step went away this way (round 3)."""
        if not DIRECT_GRADS or param is None or not param.requires_grad or self.sink is None or id(param) in self.grads or not self.sink.owns(param):
            return None
        return self.sink.grad_of(param)

    def wrote(self, param: torch.nn.Parameter, view: torch.Tensor) -> None:
        self.grads[id(param)] = view
        self.params[id(param)] = param

    def get(self, param) -> Optional[torch.Tensor]:
        ops.COLSUMS.flush()             # a reader: every queued column sum lands first
        return self.grads.get(id(param))

    def done(self, params) -> None:
        """No further add() will touch these parameters in this step (parameters that received no gradient get zeros,
        what autograd would deliver)."""
        ops.COLSUMS.flush()             # the block's queued column sums (weight-gradient partials, norm weights): one launch
        if self.sink is None:
            return
        params = [p for p in params if p is not None and p.requires_grad]
        for p in params:
            if id(p) not in self.grads and self.sink.owns(p):
                view = self.sink.grad_of(p)
                view.zero_()
                self.grads[id(p)] = view
                self.params[id(p)] = p
        self.sink.ready(params)


def _attention_bwd(P: Prepared, spq: dict, spk: dict, do: torch.Tensor, lse, D, B: int, Lq: int, H: int, shared_q: bool):
    """bf16: MFMA kernels on the row-major + transposed operand copies; fp32 parity mode: the fp32-arithmetic kernels."""
    if P.dtype == torch.bfloat16:
        spdo = ops.qkv_split(do, None, None, None, None, 0.0, B, Lq, H, P.dtype, train=True)
        return ops.attention_bwd_mfma(spq, spk, spdo, lse, D, shared_q=shared_q)
    dO_hm = ops.qkv_split(do, None, None, None, None, 0.0, B, Lq, H, P.dtype)[0]
    return ops.attention_bwd(spq["Q"], spk["K"], spk["V"], dO_hm, lse, D, shared_q=shared_q)


def _wgrad(dYt: torch.Tensor, At: torch.Tensor) -> torch.Tensor:
    """dW [N, Ka] fp32 = dYt [N, Mp] . At [Ka, Mp]^T.  The output is small and the contraction (tokens) long: cut it into
    slices so that the launch has at least ~256 workgroups (split-K, partials summed deterministically)."""
    N, Mp = dYt.shape
    tiles = ((N + 127) // 128) * ((At.shape[0] + 127) // 128)
    slices = max(1, min(32, (320 + tiles - 1) // tiles, Mp // 512))
    return ops.gemm_splitk(dYt, At, slices)


def weight_grad(dy: torch.Tensor, a: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """dW [N, Ka] fp32 = dy[M, N]^T a[M, Ka].  bf16: m324_gemm_tn straight from the token-major operands (transposing LDS
    reads); fp32 parity mode: two transposed copies + the NN kernel with split-K.  out (bf16 path): see ops.gemm_tn; the
    returned tensor IS out when it was used.  accumulate (bf16 path, with out): out += dW."""
    M, N = dy.shape
    Ka = a.shape[1]
    if dy.dtype == torch.bfloat16 and N % 8 == 0 and Ka % 8 == 0:
        if N % 256 == 0 and Ka % 256 == 0 and M % 32 == 0:       # 256 x 256 tiles, one 8-wave workgroup per CU: fill 256 CUs
            slices = max(1, min(64, 252 // ((N // 256) * (Ka // 256)), M // 512))
        else:                                                    # 128 x 128 tiles, two workgroups per CU
            tiles = ((N + 127) // 128) * ((Ka + 127) // 128)
            slices = max(1, min(32, (640 + tiles - 1) // tiles, M // 512))
        # m324_gemm_tn addresses a slice's rows through 32-bit buffer offsets and refuses (ks + 64) * ld * 2 bytes >= 2 GiB with
        # ks = the tokens per slice rounded up to whole 64-row tiles (csrc/gemm.hip); very long token counts need more slices
        # than the occupancy rule asks for.  Same formula here, so the guard and the library agree at the boundary.
        ld = max(dy.stride(0), a.stride(0))

        def _ks(n: int) -> int:
            return ((M + n - 1) // n + 63) // 64 * 64

        while slices < 4096 and (_ks(slices) + 64) * ld * 2 >= 0x7FFFFFFF:
            slices *= 2
        while slices > 1 and _ks(slices) * (slices - 1) >= M:       # no slice may be left empty
            slices -= 1
        return ops.gemm_tn(dy, a, slices, out=out, accumulate=accumulate)
    if accumulate:
        out += _wgrad(ops.transpose(dy), ops.transpose(a)).reshape(out.shape)
        return out
    return _wgrad(ops.transpose(dy), ops.transpose(a))


def _wt(P: Prepared, weight: torch.Tensor) -> torch.Tensor:
    """[K', N_pad] transposed GEMM operand of a weight (for dgrad), cached like P.mat."""
    return P.mat_t(weight, ops.transpose)


def linear_bwd(P: Prepared, G: GradStore, weight, bias, a: torch.Tensor, dy: torch.Tensor, need_da: bool = True,
               gelu_grad_of: Optional[torch.Tensor] = None, dy_colsum: Optional[torch.Tensor] = None, mul_by: Optional[torch.Tensor] = None):
    """y = a W^T + b.  a [M, Ka] and dy [M, N] in the compute dtype.  Returns da [M, Ka] (compute dtype) or None.
    gelu_grad_of = z with a = gelu(z): the returned tensor is dz = da * gelu'(z) (fused into the dgrad GEMM's epilogue); mul_by = gelu'(z)
    itself, left by the forward (M324_GELU_GRAD_FWD): the epilogue multiplies.
    dy_colsum: the column sums of dy when the kernel that produced dy delivered them (Carry), else computed here."""
    M, N = dy.shape
    if bias is not None:
        tb = G.target(bias) if dy_colsum is None else None
        eb = G.existing(bias) if tb is None and dy_colsum is None else None
        if tb is not None:                                    # column sums straight into the flat gradient buffer
            ops.colsum(dy, out=tb.view(-1))
            G.wrote(bias, tb)
        elif eb is not None:
            ops.colsum(dy, out=eb.view(-1), accumulate=True)
        else:
            G.add(bias, dy_colsum if dy_colsum is not None else ops.colsum(dy))     # dy_colsum: a vector or ops.Rows (queued)
    k_true = weight[0].numel()
    tw = G.target(weight) if k_true == a.shape[1] else None
    ex = G.existing(weight) if tw is None and k_true == a.shape[1] else None
    if ex is not None:                                       # a later pass over a shared weight (the decoder's per-sample loop):
        weight_grad(dy, a, out=ex, accumulate=True)          # summed into the gradient it already holds, no temporary, no add
    else:
        dW = weight_grad(dy, a, out=tw)
        if tw is not None and dW is tw:
            G.wrote(weight, tw)
        else:
            G.add(weight, dW[:, :k_true] if k_true != a.shape[1] else dW)
    if not need_da:
        return None
    Wt = _wt(P, weight)                                       # [Kp, N_pad]
    if Wt.shape[1] != N:
        raise RuntimeError("linear_bwd: output width must be a multiple of 64")
    da = torch.empty((M, Wt.shape[0]), dtype=dy.dtype, device=dy.device)
    ops.gemm(dy, Wt, da, gelu_grad_of=gelu_grad_of, mul_by=mul_by)
    return da


class Carry:
    """The residual-stream gradient dx in the form the next backward GEMMs read it: its bf16 copy and that copy's column sums
    (= the bias gradient of the Linear in front).  Every LayerNorm backward that finishes a dx delivers both in the same pass
    (m324_layernorm_bwd_cast); a consumer that finds no carry (first block of a chain, fp32 parity mode) computes them itself."""

    def __init__(self):
        self.dx = None          # the fp32 tensor the copy belongs to (identity + version are checked)
        self.bf16 = None
        self.colsum = None
        self.version = -1

    def take(self, P: Prepared, dx: torch.Tensor):
        """(dx in the compute dtype, its column sums or None)."""
        if self.dx is dx and self.bf16 is not None and self.version == dx._version and P.dtype == torch.bfloat16:
            out = (self.bf16, self.colsum)
            self.dx = self.bf16 = self.colsum = None
            return out
        return ops.cast(dx, P.dtype), None

    def ln_bwd(self, P: Prepared, x, w, eps, dy, dx, accumulate=True, row_map=(0, 0, 0)):
        """ops.layernorm_bwd that also fills the carry for dx (bf16 mode); returns (dw, db)."""
        if P.dtype != torch.bfloat16:
            return ops.layernorm_bwd(x, w, eps, dy, dx, accumulate=accumulate, row_map=row_map, reduce=not ops.DEFER_COLSUM)
        c = torch.empty(dx.shape, dtype=torch.bfloat16, device=dx.device)
        dw, db, cs = ops.layernorm_bwd(x, w, eps, dy, dx, accumulate=accumulate, row_map=row_map, cast_out=c, reduce=not ops.DEFER_COLSUM)
        self.dx, self.bf16, self.colsum, self.version = dx, c, cs, dx._version
        return dw, db


def mlp_internals(P: Prepared, norm2, mlp, x_mid: torch.Tensor) -> dict:
    """h2 = LN2(x_mid), z = fc1(h2) + b, g = gelu(z): what the MLP half's backward reads (one LayerNorm, one GEMM launch)."""
    rows, C = x_mid.shape
    fc1 = mlp.mlp[0]
    h2 = torch.empty((rows, C), dtype=P.dtype, device=x_mid.device)
    ops.layernorm(x_mid, P.vec(norm2.weight), P.vec(norm2.bias), norm2.eps, h2)
    z = torch.empty((rows, fc1.out_features), dtype=P.dtype, device=x_mid.device)
    g = torch.empty_like(z)
    if GELU_GRAD_FWD:                          # g = gelu(z) and gelu'(z) in one launch: erf is evaluated once, where it is needed anyway
        ops.gemm(h2, P.mat(fc1.weight), g, bias=P.vec(fc1.bias), act=ACT_GELU, gelu_grad_out=z)
        return dict(h2=h2, dg=z, g=g)
    ops.gemm(h2, P.mat(fc1.weight), g, bias=P.vec(fc1.bias), act=ACT_GELU, preact_out=z)     # g = gelu(z) and z in one launch
    return dict(h2=h2, z=z, g=g)


def mlp_residual_bwd(P: Prepared, G: GradStore, norm2, mlp, x_mid: torch.Tensor, dx: torch.Tensor,
                     saved: Optional[dict] = None, carry: Optional[Carry] = None) -> None:
    """x_out = x_mid + fc2(gelu(fc1(LN2(x_mid)))).  dx (fp32): grad w.r.t. x_out on entry, w.r.t. x_mid on exit.
    saved: mlp_internals() of the forward (None: recomputed here).  carry: see Carry (in: dx's bf16 form; out: the new dx's)."""
    fc1, fc2 = mlp.mlp[0], mlp.mlp[2]
    carry = carry if carry is not None else Carry()
    m = saved if saved is not None else mlp_internals(P, norm2, mlp, x_mid)
    h2, g = m["h2"], m["g"]
    dxT, dsum = carry.take(P, dx)
    dz = linear_bwd(P, G, fc2.weight, fc2.bias, g, dxT, gelu_grad_of=m.get("z"), mul_by=m.get("dg"), dy_colsum=dsum)      # (dxT W2) * gelu'(z)
    dh2 = linear_bwd(P, G, fc1.weight, fc1.bias, h2, dz)
    dw, db = carry.ln_bwd(P, x_mid, P.vec(norm2.weight), norm2.eps, dh2, dx, accumulate=True)
    G.add(norm2.weight, dw)
    G.add(norm2.bias, db)


def self_attn_block_internals(blk, P: Prepared, x_in: torch.Tensor, B: int, L: int, x_out: Optional[torch.Tensor] = None) -> dict:
    """Everything the backward of QK_Norm_TransformerBlock reads, computed from the block's input (fp32 residual stream).
    Two uses: the backward's recompute (x_out None: the last GEMM, fc2, is not needed), and -- when memory allows, which on
    288 GB it does -- the training FORWARD itself (x_out: fp32 buffer for the block's output), so that the backward finds the
    internals in HBM instead of computing them a second time (training.py, M324_TRAIN_STORE)."""
    rows, C = x_in.shape
    a = blk.attn
    H = a.num_heads
    dev = x_in.device
    h1 = torch.empty((rows, C), dtype=P.dtype, device=dev)
    ops.layernorm(x_in, P.vec(blk.norm1.weight), P.vec(blk.norm1.bias), blk.norm1.eps, h1)
    qkv = torch.empty((rows, 3 * C), dtype=P.dtype, device=dev)
    ops.gemm(h1, P.mat(a.to_qkv.weight), qkv, bias=P.vec(a.to_qkv.bias))
    qw, kw = a._qk_w(P)
    sp = ops.qkv_split(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], qw, kw, RMS_EPS, B, L, H, P.dtype,
                       q_scale=ops.Q_PRESCALE, train=True)
    o = torch.empty((rows, C), dtype=P.dtype, device=dev)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=dev)
    ops.attention(sp["Q"], sp["K"], sp["Vt"], o, prescaled=True, lse=lse)
    x_mid = torch.empty((rows, C), dtype=torch.float32, device=dev)
    ops.gemm(o, P.mat(a.fc.weight), x_mid, bias=P.vec(a.fc.bias), residual=x_in)
    m = mlp_internals(P, blk.norm2, blk.mlp, x_mid)
    if x_out is not None:
        fc2 = blk.mlp.mlp[2]
        ops.gemm(m["g"], P.mat(fc2.weight), x_out, bias=P.vec(fc2.bias), residual=x_mid)
    return dict(h1=h1, qkv=qkv, sp=sp, o=o, lse=lse, x_mid=x_mid, mlp=m)


def self_attn_block_bwd(blk, P: Prepared, G: GradStore, x_in: torch.Tensor, dx: torch.Tensor, B: int, L: int,
                        saved: Optional[dict] = None, carry: Optional[Carry] = None) -> None:
    """Backward of QK_Norm_TransformerBlock.run.  dx: grad w.r.t. the block output on entry, w.r.t. x_in on exit.
    saved: self_attn_block_internals() of the forward (None: recomputed from x_in, the reference's checkpoint policy).
    carry: a Carry shared by the blocks of a chain (the bf16 form of dx travels from LayerNorm backward to the next GEMMs)."""
    carry = carry if carry is not None else Carry()
    rows, C = x_in.shape
    a = blk.attn
    H = a.num_heads
    dev = dx.device
    s = saved if saved is not None else self_attn_block_internals(blk, P, x_in, B, L)
    h1, qkv, sp, o, lse, x_mid = s["h1"], s["qkv"], s["sp"], s["o"], s["lse"], s["x_mid"]
    qw, kw = a._qk_w(P)
    # ---- MLP half
    mlp_residual_bwd(P, G, blk.norm2, blk.mlp, x_mid, dx, saved=s["mlp"], carry=carry)     # dx = d x_mid
    s["mlp"] = None
    # ---- attention half
    dxT, dsum = carry.take(P, dx)
    do = linear_bwd(P, G, a.fc.weight, a.fc.bias, o, dxT, dy_colsum=dsum)
    D = ops.attention_delta(o, do, B, H, L)
    dQ, dK, dV = _attention_bwd(P, sp, sp, do, lse, D, B, L, H, shared_q=False)
    dqkv = torch.empty((rows, 3 * C), dtype=P.dtype, device=dev)
    dqw, dkw = ops.qkv_split_bwd(dQ, dK, dV, qkv[:, :C], qkv[:, C:2 * C], qw, kw, RMS_EPS, B, L, H, dqkv[:, :C],
                                 dqkv[:, C:2 * C], dqkv[:, 2 * C:], reduce=not ops.DEFER_COLSUM)
    if a.use_qk_norm:
        G.add(a.q_norm.weight, dqw)
        G.add(a.k_norm.weight, dkw)
    dh1 = linear_bwd(P, G, a.to_qkv.weight, a.to_qkv.bias, h1, dqkv)
    dw, db = carry.ln_bwd(P, x_in, P.vec(blk.norm1.weight), blk.norm1.eps, dh1, dx, accumulate=True)
    G.add(blk.norm1.weight, dw)
    G.add(blk.norm1.bias, db)


def cross_attn_block_internals(blk, P: Prepared, query: torch.Tensor, kv: torch.Tensor, B: int, Lq: int, Lk: int,
                                kv_row_map=(0, 0, 0), shared_q: bool = False, x_out: Optional[torch.Tensor] = None) -> dict:
    """Everything the backward of QK_Norm_CrossAttentionBlock reads, from the block's inputs (see
    self_attn_block_internals: the backward's recompute, or -- with x_out -- the training forward that keeps it)."""
    a = blk.attn
    C, H = a.dim, a.num_heads
    dev = query.device
    Bq = 1 if shared_q else B
    qn = torch.empty((Bq * Lq, C), dtype=P.dtype, device=dev)
    ops.layernorm(query, P.vec(blk.norm_q.weight), P.vec(blk.norm_q.bias), blk.norm_q.eps, qn)
    qp = torch.empty((Bq * Lq, C), dtype=P.dtype, device=dev)
    ops.gemm(qn, P.mat(a.to_q.weight), qp, bias=P.vec(a.to_q.bias))
    qw, kw = a._qk_w(P)
    spq = ops.qkv_split(qp, None, None, qw, None, RMS_EPS, Bq, Lq, H, P.dtype, q_scale=ops.Q_PRESCALE, train=True)
    kn = torch.empty((B * Lk, C), dtype=P.dtype, device=dev)
    ops.layernorm(kv, P.vec(blk.norm_kv.weight), P.vec(blk.norm_kv.bias), blk.norm_kv.eps, kn, row_map=kv_row_map)
    w_kv, b_kv = P.cat_rows((a.to_k.weight, a.to_v.weight)), P.cat_vecs((a.to_k.bias, a.to_v.bias))
    kvp = torch.empty((B * Lk, 2 * C), dtype=P.dtype, device=dev)
    ops.gemm(kn, w_kv, kvp, bias=b_kv)
    spk = ops.qkv_split(None, kvp[:, :C], kvp[:, C:], None, kw, RMS_EPS, B, Lk, H, P.dtype, train=True)
    o = torch.empty((B * Lq, C), dtype=P.dtype, device=dev)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=dev)
    ops.attention(spq["Q"], spk["K"], spk["Vt"], o, shared_q=shared_q, prescaled=True, lse=lse)
    x_mid = torch.empty((B * Lq, C), dtype=torch.float32, device=dev)
    ops.gemm(o, P.mat(a.fc.weight), x_mid, bias=P.vec(a.fc.bias), residual=query, res_rows=Lq if shared_q else 0)
    m = mlp_internals(P, blk.norm2, blk.mlp, x_mid)
    if x_out is not None:
        fc2 = blk.mlp.mlp[2]
        ops.gemm(m["g"], P.mat(fc2.weight), x_out, bias=P.vec(fc2.bias), residual=x_mid)
    return dict(qn=qn, qp=qp, spq=spq, kn=kn, kvp=kvp, spk=spk, o=o, lse=lse, x_mid=x_mid, mlp=m)


def cross_attn_block_bwd(blk, P: Prepared, G: GradStore, query: torch.Tensor, kv: torch.Tensor, dx: torch.Tensor, B: int,
                         Lq: int, Lk: int, kv_row_map=(0, 0, 0), shared_q: bool = False, d_kv: Optional[torch.Tensor] = None,
                         need_dquery: bool = True, saved: Optional[dict] = None) -> Optional[torch.Tensor]:
    """Backward of QK_Norm_CrossAttentionBlock (project_q + project_kv + attend).

    query fp32 [Bq*Lq, C] (Bq = 1 when shared_q: one query set for all B key/value batches, the decoder case);
    kv fp32 rows read through kv_row_map; dx fp32 [B*Lq, C] = grad w.r.t. the block output (consumed).
    d_kv: fp32 buffer shaped like kv that receives (+=) the gradient of the key/value rows (None: not needed).
    saved: cross_attn_block_internals() of the forward (None: recomputed here).
    Returns the gradient w.r.t. query (fp32 [Bq*Lq, C]) or None."""
    a = blk.attn
    C, H = a.dim, a.num_heads
    dev = dx.device
    Bq = 1 if shared_q else B
    s = saved if saved is not None else cross_attn_block_internals(blk, P, query, kv, B, Lq, Lk, kv_row_map, shared_q)
    qn, qp, spq, kn, kvp, spk, o, lse, x_mid = (s[k] for k in ("qn", "qp", "spq", "kn", "kvp", "spk", "o", "lse", "x_mid"))
    qw, kw = a._qk_w(P)
    w_kv = P.cat_rows((a.to_k.weight, a.to_v.weight))
    # ---- MLP half
    carry = Carry()
    mlp_residual_bwd(P, G, blk.norm2, blk.mlp, x_mid, dx, saved=s["mlp"], carry=carry)     # dx = d x_mid  [B*Lq, C]
    s["mlp"] = None
    # ---- attention half
    dxT, dsum = carry.take(P, dx)
    do = linear_bwd(P, G, a.fc.weight, a.fc.bias, o, dxT, dy_colsum=dsum)
    D = ops.attention_delta(o, do, B, H, Lq)
    dQ, dK, dV = _attention_bwd(P, spq, spk, do, lse, D, B, Lq, H, shared_q=shared_q)
    # key / value path
    dkvp = torch.empty((B * Lk, 2 * C), dtype=P.dtype, device=dev)
    _, dkw = ops.qkv_split_bwd(None, dK, dV, None, kvp[:, :C], None, kw, RMS_EPS, B, Lk, H, None, dkvp[:, :C], dkvp[:, C:], reduce=not ops.DEFER_COLSUM)
    if a.use_qk_norm:
        G.add(a.k_norm.weight, dkw)
    if a.to_k.bias is not None:
        bsum = ops.colsum(dkvp)
        G.add(a.to_k.bias, bsum[:C])
        G.add(a.to_v.bias, bsum[C:])
    dWkv = weight_grad(dkvp, kn)
    G.add(a.to_k.weight, dWkv[:C])
    G.add(a.to_v.weight, dWkv[C:])
    if d_kv is not None:
        Wt = P.derived("catT", (a.to_k.weight, a.to_v.weight), lambda: ops.transpose(w_kv))      # [C, 2C]
        dkn = torch.empty((B * Lk, C), dtype=P.dtype, device=dev)
        ops.gemm(dkvp, Wt, dkn)
        dw, db = ops.layernorm_bwd(kv, P.vec(blk.norm_kv.weight), blk.norm_kv.eps, dkn, d_kv, accumulate=True,
                                   row_map=kv_row_map, reduce=not ops.DEFER_COLSUM)
    else:
        # the weight gradient of norm_kv is still needed even when the kv rows themselves are inputs
        Wt = P.derived("catT", (a.to_k.weight, a.to_v.weight), lambda: ops.transpose(w_kv))
        dkn = torch.empty((B * Lk, C), dtype=P.dtype, device=dev)
        ops.gemm(dkvp, Wt, dkn)
        scratch = torch.empty_like(kv)
        dw, db = ops.layernorm_bwd(kv, P.vec(blk.norm_kv.weight), blk.norm_kv.eps, dkn, scratch, accumulate=False,
                                   row_map=kv_row_map, reduce=not ops.DEFER_COLSUM)
    G.add(blk.norm_kv.weight, dw)
    G.add(blk.norm_kv.bias, db)
    # query path
    dQs = dQ
    if shared_q:                                              # one query set: sum the per-batch gradients
        dQs = ops.colsum(dQ.reshape(B, -1)).to(P.dtype).reshape(1, H, Lq, 64)
    dqp = torch.empty((Bq * Lq, C), dtype=P.dtype, device=dev)
    dqw, _ = ops.qkv_split_bwd(dQs, None, None, qp, None, qw, None, RMS_EPS, Bq, Lq, H, dqp, None, None, reduce=not ops.DEFER_COLSUM)
    if a.use_qk_norm:
        G.add(a.q_norm.weight, dqw)
    dqn = linear_bwd(P, G, a.to_q.weight, a.to_q.bias, qn, dqp)
    # residual: x_mid = query (broadcast over B when shared) + fc(o)
    if shared_q:
        dquery = ops.colsum(dx.reshape(B, -1)).reshape(Lq, C).contiguous()
    else:
        dquery = dx
    dw, db = ops.layernorm_bwd(query, P.vec(blk.norm_q.weight), blk.norm_q.eps, dqn, dquery, accumulate=True, reduce=not ops.DEFER_COLSUM)
    G.add(blk.norm_q.weight, dw)
    G.add(blk.norm_q.bias, db)
    return dquery if need_dquery else None
