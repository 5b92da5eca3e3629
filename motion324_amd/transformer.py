"""Parameter containers with the reference's names + the HIP execution of each block.

Mirrors the public classes of the reference's model/transformer.py (RMSNorm :30-42, MLP :46-81,
QK_Norm_CrossAttention :84-144, QK_Norm_SelfAttention :146-219, QK_Norm_CrossAttentionBlock
:324-377, QK_Norm_TransformerBlock :379-423, init_weights :15-25) so that state-dict keys and
constructor arguments are identical.  The arithmetic is NOT torch.nn: every block runs as a chain
of libm324 calls (motion324_amd.ops) on raw device buffers:

    LN -> GEMM(qkv) -> split/RMSNorm/transpose -> flash attention -> GEMM(fc)+residual
       -> LN -> GEMM(fc1)+GELU -> GEMM(fc2)+residual

Activations feeding a GEMM are held in the compute dtype (bf16 speed mode / fp32 parity mode);
the residual stream is always fp32, as it is in the reference under autocast.
"""
from __future__ import annotations

import os

from typing import Optional

import torch
import torch.nn as nn

from . import ops, switches
from .lib import ACT_GELU
from .prepared import Prepared

LN_EPS = 1e-5     # nn.LayerNorm default, reference transformer.py:345-346,357,400,411
RMS_EPS = 1e-5    # reference transformer.py:31


def init_weights(module, std=0.02):
    """Same initialisation rule as the reference (transformer.py:15-25)."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        torch.nn.init.normal_(module.weight, mean=0.0, std=std)
        if isinstance(module, nn.Linear) and module.bias is not None:
            torch.nn.init.zeros_(module.bias)


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = RMS_EPS):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))


class MLP(nn.Module):
    def __init__(self, dim, mlp_ratio=4, bias=False, dropout=0.0, activation=nn.GELU, mlp_dim=None):
        super().__init__()
        if bias or dropout != 0.0 or activation is not nn.GELU:
            raise NotImplementedError("the HIP path implements the configuration the reference instantiates "
                                      "(no bias, no dropout, exact GELU)")
        hidden = mlp_dim if mlp_dim is not None else int(dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(dim, hidden, bias=False), nn.GELU(), nn.Linear(hidden, dim, bias=False),
                                 nn.Dropout(0.0))


class _AttnBase(nn.Module):
    def __init__(self, dim, head_dim, use_qk_norm):
        super().__init__()
        if head_dim != 64:
            raise NotImplementedError("libm324 attention kernels are specialised for head_dim 64 (config d_head)")
        assert dim % head_dim == 0, f"Token dimension {dim} should be divisible by head dimension {head_dim}"
        self.dim, self.head_dim, self.num_heads, self.use_qk_norm = dim, head_dim, dim // head_dim, use_qk_norm

    def _register_qk_norm(self):
        """Called LAST by the subclasses: the reference registers q_norm / k_norm after its Linear layers
        (transformer.py:112-122,182-190), and named_parameters() order is what numbers the optimizer state
        (utils/training_utils.py:38-52) -- checkpoints are exchanged by that number."""
        if self.use_qk_norm:
            self.q_norm = RMSNorm(self.head_dim)
            self.k_norm = RMSNorm(self.head_dim)

    def _qk_w(self, P: Prepared):
        if not self.use_qk_norm:
            return None, None
        return P.vec(self.q_norm.weight), P.vec(self.k_norm.weight)

    def scores_bounded(self, P: Prepared) -> bool:
        """May m324_attention run without a reference maximum (M324_ATTN_SCORES_BOUNDED)?  The per-head RMSNorm of q and k
        (reference transformer.py:36-42,200-207) bounds every log2-domain score: a normalised 64-vector has norm <= 8, so
        |q^ . k^| scale log2(e) <= 64 max|w_q| max|w_k| * 0.18 = 11.5 at unit norm weights.  Vouched for only with a margin
        (<= 48 of the kernel's 64; fp32 sums of 2^48 x 10^5 keys are far inside range), from the CURRENT weights (cached per
        weight version: one device read at preparation, none per call); ATTN_BOUNDED = False (M324_ATTN_BOUNDED=0) never vouches."""
        if not (ATTN_BOUNDED and not _FUSE_OFF and self.use_qk_norm and P.dtype == torch.bfloat16):
            return False
        t = P.derived("qk_score_bound", (self.q_norm.weight, self.k_norm.weight), lambda: torch.tensor(
            [64.0 * ops.Q_PRESCALE * float(self.q_norm.weight.detach().abs().max()) * float(self.k_norm.weight.detach().abs().max())]))
        return float(t[0]) <= 48.0


class QK_Norm_SelfAttention(_AttnBase):
    def __init__(self, dim, head_dim, qkv_bias=False, fc_bias=True, attn_dropout=0.0, fc_dropout=0.0, use_qk_norm=True):
        super().__init__(dim, head_dim, use_qk_norm)
        if attn_dropout != 0.0 or fc_dropout != 0.0:
            raise NotImplementedError("attention / projection dropout is 0 everywhere in the reference model")
        self.to_qkv = nn.Linear(dim, 3 * dim, bias=qkv_bias)
        self.fc = nn.Linear(dim, dim, bias=fc_bias)
        self._register_qk_norm()


class QK_Norm_CrossAttention(_AttnBase):
    def __init__(self, dim, head_dim, kv_dim=None, qkv_bias=False, fc_bias=True, attn_dropout=0.0, fc_dropout=0.0,
                 use_qk_norm=True):
        super().__init__(dim, head_dim, use_qk_norm)
        if attn_dropout != 0.0 or fc_dropout != 0.0:
            raise NotImplementedError("attention / projection dropout is 0 everywhere in the reference model")
        kv_dim = dim if kv_dim is None else kv_dim
        self.kv_dim = kv_dim
        self.to_q = nn.Linear(dim, dim, bias=qkv_bias)
        self.to_k = nn.Linear(kv_dim, dim, bias=qkv_bias)
        self.to_v = nn.Linear(kv_dim, dim, bias=qkv_bias)
        self.fc = nn.Linear(dim, dim, bias=fc_bias)
        self._register_qk_norm()


# LayerNorm fold (bf16 inference, include/m324.h): the GEMM that writes a residual stream leaves the row statistics of
# what it stored (and, next to an fp32 stream, its bf16 twin); the GEMM behind the LayerNorm reads that twin with the
# LayerNorm's scale folded into its weight and applies mean / rstd in its epilogue; the LayerNorm pass itself disappears.
# Between producer and consumer the per-block statistics of a row are merged into (rstd, -rstd mean): by the consumer itself
# (M324_FOLD_MERGE=1, round 4: table entries requested in front of its first operand tiles, Chan's update behind them) or by
# m324_rowstats_finish in a launch of its own (round 3; =0).
# Measured on the c2 clip, interleaved A/B (profiles/r03_ln_fold_ab.md, profiles/r04_ln_fold_merge.md):
#   * bf16 streams (the decoder: 65 536 rows, 8-wave GEMMs, no twin to write): two 46-us passes go for +2.5 us (fc2's
#     statistics), +4.5 us (fc1's epilogue) and two 4-us merges: decoder block 0.903 -> 0.874 ms (round 3).
#   * fp32 streams (trunk, DINO: 10 368 / 8 224 rows, 56 LayerNorms per clip): a 9.6-us pass goes for +4 us in the producer
#     (statistics on a one-wave-per-SIMD kernel, 50 % more store bytes for the twin) and +2 us in the consumer -- and, round 3,
#     a 2.5-us merge launch + its boundary: break-even, left off.  With the merge inside the consumer: 8.82 -> 8.69 ms per clip
#     (the launch form: 8.98), so M324_FOLD_LN=2, every stream, is the default since round 4.
# M324_FOLD_LN=1 folds the bf16 streams only, =0 restores every separate pass.
FOLD_LN = int(switches.get("M324_FOLD_LN"))
PAIR_PROJ = switches.flag("M324_PAIR_PROJ")         # the decoder's q and k|v LayerNorms / projections as two launches instead of four
FOLD_MERGE = switches.flag("M324_FOLD_MERGE")      # folded consumers merge the producer's per-block statistics themselves
ATTN_BOUNDED = switches.flag("M324_ATTN_BOUNDED")


class LNFold:
    """What travels with a residual stream x [rows, C] under the LayerNorm fold: xb (bf16 twin = the folded GEMMs' A operand;
    the stream itself when that is bf16), part (per-64-column-block (sum, M2) left by the producer GEMM) and stat
    ((rstd, -rstd mean) per row, valid once ready(eps) has merged the blocks)."""

    def __init__(self, x: torch.Tensor):
        rows, C = x.shape
        self.rows, self.C = rows, C
        self.xb = x if x.dtype == torch.bfloat16 else torch.empty((rows, C), dtype=torch.bfloat16, device=x.device)
        self.part = torch.empty((C // 64, rows, 2), dtype=torch.float32, device=x.device)
        self.stat = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        self.pending = False
        self.own_copy = x.dtype != torch.bfloat16

    @staticmethod
    def usable(P: Prepared, rows: int, C: int, bf16_stream: bool = False) -> bool:
        """bf16_stream: the stream itself is bf16 (level 1 of M324_FOLD_LN folds only those; level 2 every stream)."""
        return (FOLD_LN >= (1 if bf16_stream else 2) and not _FUSE_OFF and P.dtype == torch.bfloat16 and rows > 64
                and C % 64 == 0 and C >= 128 and not torch.is_grad_enabled())       # m324_gemm's fold paths need K >= 128 (two K-stages)

    def from_stream(self, x: torch.Tensor, eps: float) -> "LNFold":
        """Head of a chain: statistics (and the bf16 twin) straight from the fp32 stream."""
        ops.rowstats(x, eps, self.stat, self.xb if self.own_copy else None)
        self.pending = False
        return self

    def producer(self) -> dict:
        """Keyword arguments for the ops.gemm call that writes the stream."""
        self.pending = True
        return dict(stats_out=self.part, copy_out=self.xb if self.own_copy else None)

    def ln(self, eps: float, colsum: torch.Tensor, merged: bool = False) -> tuple:
        """The `ln` argument of the ops.gemm call that consumes the stream: the merged table, or -- statistics still in the
        producer's per-block form and M324_FOLD_MERGE on -- that table itself, merged by the consumer (no launch in between).
        merged: the consumer is one that m324_gemm runs on schedule v15 (hp_consumer below), which reads the merged table: the
        m324_rowstats_finish launch (2.5-4 us, the same arithmetic as the consumers' own merge) buys a GEMM 7-45 us shorter."""
        if self.pending and FOLD_MERGE and not merged and self.C <= 1024 and self.C % 128 == 0:      # m324_gemm: an even block count <= 16
            return (self.part, colsum, eps)
        return (self.ready(eps), colsum)

    def ready(self, eps: float) -> torch.Tensor:
        if self.pending:
            ops.rowstats_finish(self.part, eps, self.stat)
            self.pending = False
        return self.stat


def hp_consumer(rows: int, n_out: int, k: int) -> bool:
    """Will m324_gemm run this fc1 + GELU LayerNorm-fold consumer on schedule v15 when it gets the MERGED statistics table?  (v15
    reads the merged table only; the other schedules merge the producer's block table themselves, which saves the
    m324_rowstats_finish launch.)  Asked of the library (ops.gemm_schedule -> m324_gemm_plan, live tunables included), so that
    `lib.set_tunable("M324_HP", ...)` / a forced M324_GEMM and this decision cannot drift apart."""
    if k != 768 or n_out % 128 != 0 or rows < 1:
        return False
    return ops.gemm_schedule(rows, n_out, k, act=ACT_GELU, fold_merged=True) == 15


def _mlp_residual(P: Prepared, norm2: nn.LayerNorm, mlp: MLP, x: torch.Tensor, fold: Optional[LNFold] = None,
                  feed_next: bool = True) -> torch.Tensor:
    """x += fc2(gelu(fc1(LN(x))))  (reference transformer.py:376,422), x [rows, C] (fp32, or the decoder's bf16 stream), in
    place.  fold: the stream's LNFold with statistics pending or ready; fc2 then leaves the next LayerNorm's (feed_next)."""
    rows, C = x.shape
    fc1, fc2 = mlp.mlp[0], mlp.mlp[2]
    h1 = torch.empty((rows, fc1.out_features), dtype=P.dtype, device=x.device)
    if fold is not None:
        w1, cs1, b1 = P.folded(norm2.weight, norm2.bias, fc1.weight, fc1.bias)
        ops.gemm(fold.xb, w1, h1, bias=b1, act=ACT_GELU, ln=fold.ln(norm2.eps, cs1, merged=hp_consumer(rows, fc1.out_features, C)))
        ops.gemm(h1, P.mat(fc2.weight), x, bias=P.vec(fc2.bias), residual=x, **(fold.producer() if feed_next else {}))
        return x
    h = torch.empty((rows, C), dtype=P.dtype, device=x.device)
    ops.layernorm(x, P.vec(norm2.weight), P.vec(norm2.bias), norm2.eps, h)
    ops.gemm(h, P.mat(fc1.weight), h1, bias=P.vec(fc1.bias), act=ACT_GELU)
    ops.gemm(h1, P.mat(fc2.weight), x, bias=P.vec(fc2.bias), residual=x)
    return x


# Fused q|k|v projection epilogue (m324_gemm M324_AUX_QKV_HEADS + row-major-V attention) for bf16 inference on sequences
# below 2048 tokens; longer ones keep m324_qkv_split's transposed V, whose 8-wave attention kernel is 9 % faster than
# its transposing-read variant.  M324_FUSE_QKV=0 disables (A/B measurements).
FUSE_QKV = switches.flag("M324_FUSE_QKV")
FUSE_QKV_VT = switches.flag("M324_FUSE_QKV_VT")     # long sequences: the epilogue writes the transposed V itself


_FUSE_OFF = 0


class fusion_disabled:
    """Training steps run their forward inside this context (the backward recomputes with m324_qkv_split's training
    outputs, and an autograd.Function's forward runs with grad mode off, so grad mode alone cannot tell)."""

    def __enter__(self):
        global _FUSE_OFF
        _FUSE_OFF += 1

    def __exit__(self, *exc):
        global _FUSE_OFF
        _FUSE_OFF -= 1


class fusion_allowed:
    """Inside a training step's fusion_disabled(): a frozen sub-network that no gradient flows through (the DINOv2 image encoder,
    reference dinov2.py:126-131) may run its inference form -- LayerNorm fold, head-major q|k|v epilogue -- when the caller also
    switches grad mode off for it."""

    def __enter__(self):
        global _FUSE_OFF
        self._saved, _FUSE_OFF = _FUSE_OFF, 0

    def __exit__(self, *exc):
        global _FUSE_OFF
        _FUSE_OFF = self._saved


def fuse_qkv(P: Prepared, rows: int, L: int) -> bool:
    """Per-frame blocks (L < 2048): head-major Q / K / V, attention with the row-major-V kernel.  Long sequences (the
    global blocks) keep the faster transposed-V attention; the projection epilogue writes Vt itself when L % 128 == 0."""
    return (FUSE_QKV and not _FUSE_OFF and P.dtype == torch.bfloat16 and rows > 64 and (L < 2048 or (FUSE_QKV_VT and L % 128 == 0))
            and not torch.is_grad_enabled())


def fuse_proj(P: Prepared, rows: int) -> bool:
    """The q / k|v projections of the cross-attention blocks write their head-major operands from the GEMM epilogue (bf16
    inference; the tile kernels need more than 64 rows)."""
    return FUSE_QKV and not _FUSE_OFF and P.dtype == torch.bfloat16 and rows > 64 and not torch.is_grad_enabled()


class QK_Norm_TransformerBlock(nn.Module):
    """Pre-norm self-attention block (reference transformer.py:379-423)."""

    def __init__(self, dim, head_dim, ln_bias=False, attn_qkv_bias=False, attn_dropout=0.0, attn_fc_bias=False,
                 attn_fc_dropout=0.0, mlp_ratio=4, mlp_bias=False, mlp_dropout=0.0, use_qk_norm=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, bias=ln_bias)
        self.attn = QK_Norm_SelfAttention(dim, head_dim, qkv_bias=attn_qkv_bias, fc_bias=attn_fc_bias,
                                          attn_dropout=attn_dropout, fc_dropout=attn_fc_dropout, use_qk_norm=use_qk_norm)
        self.norm2 = nn.LayerNorm(dim, bias=ln_bias)
        self.mlp = MLP(dim, mlp_ratio=mlp_ratio, bias=mlp_bias, dropout=mlp_dropout)

    def run(self, P: Prepared, x: torch.Tensor, B: int, L: int, kv_gather=None, fold: Optional[LNFold] = None,
            feed_next: bool = True) -> torch.Tensor:
        """x: fp32 [B*L, C] residual stream, updated in place (x + attn(LN x); x + mlp(LN x)).

        kv_gather (frame-parallel global attention): object with start(kv_local [B*L, 2C]) / finish() -> (all ranks'
        [B*L_full, 2C], L_full) -- Pcd_motion._KVGather; queries stay local, keys/values cover the whole clip.  The k|v
        projection runs first so that its all-gather travels while the q projection and the q split execute.
        fold: the stream's LNFold (statistics of x pending or ready): both LayerNorms of the block are folded into the
        projections behind them; the last GEMM leaves the statistics for the next block unless feed_next is False."""
        rows, C = x.shape
        assert rows == B * L
        a = self.attn
        h = torch.empty((rows, C), dtype=P.dtype, device=x.device)
        qw, kw = a._qk_w(P)
        if fold is not None:
            w, cs, bias = P.folded(self.norm1.weight, self.norm1.bias, a.to_qkv.weight, a.to_qkv.bias)
            src, lnk = fold.xb, (lambda lo, hi: dict(ln=fold.ln(self.norm1.eps, cs[lo:hi])))
        else:
            ops.layernorm(x, P.vec(self.norm1.weight), P.vec(self.norm1.bias), self.norm1.eps, h)
            w, bias = P.mat(a.to_qkv.weight), P.vec(a.to_qkv.bias)
            src, lnk = h, (lambda lo, hi: {})
        out_kw = fold.producer if fold is not None else dict
        if fuse_qkv(P, rows, L) and kv_gather is None:
            # short sequences (the per-frame blocks): the projection's epilogue writes head-major Q / K / V itself
            # (RMSNorm + q pre-scale on the fp32 accumulators) and the attention reads V row-major
            long_seq = L >= 2048
            Q, K = (torch.empty((B, a.num_heads, L, 64), dtype=P.dtype, device=x.device) for _ in range(2))
            V = torch.empty((B, a.num_heads, 64, L) if long_seq else (B, a.num_heads, L, 64), dtype=P.dtype, device=x.device)
            ops.gemm(src, w, None, bias=bias, qkv_heads=(Q, K, V, qw, kw, RMS_EPS, ops.Q_PRESCALE, L, a.num_heads),
                     **lnk(0, 3 * C))
            ops.attention(Q, K, V, h, prescaled=True, v_rowmajor=not long_seq, bounded=long_seq and a.scores_bounded(P))
            ops.gemm(h, P.mat(a.fc.weight), x, bias=P.vec(a.fc.bias), residual=x, **out_kw())
            return _mlp_residual(P, self.norm2, self.mlp, x, fold, feed_next)
        if kv_gather is None:
            qkv = torch.empty((rows, 3 * C), dtype=P.dtype, device=x.device)
            ops.gemm(src, w, qkv, bias=bias, **lnk(0, 3 * C))
            Q, K, Vt = ops.qkv_split(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], qw, kw, RMS_EPS, B, L, a.num_heads,
                                     P.dtype, q_scale=ops.Q_PRESCALE)
        else:
            # to_qkv.weight rows [C, 3C) are the k|v projection, rows [0, C) the q projection (contiguous row slices)
            kv = torch.empty((rows, 2 * C), dtype=P.dtype, device=x.device)
            ops.gemm(src, w[C:], kv, bias=None if bias is None else bias[C:], **lnk(C, 3 * C))
            kv_gather.start(kv)                                                       # collective on the side stream ...
            q = torch.empty((rows, C), dtype=P.dtype, device=x.device)
            ops.gemm(src, w[:C], q, bias=None if bias is None else bias[:C], **lnk(0, C))   # ... under the q projection + split
            Q, _, _ = ops.qkv_split(q, None, None, qw, None, RMS_EPS, B, L, a.num_heads, P.dtype, q_scale=ops.Q_PRESCALE)
            if getattr(kv_gather, "overlap", False):
                # ... and under the attention over the rank's OWN keys (1 / world of the block's attention work): every attention
                # kernel leaves the log2-domain log-sum-exp of its rows, so the softmax over all keys is the lse-weighted mean of
                # the partial outputs (m324_attention_merge).  Remote keys = the gathered rows in front of / behind the own range.
                H = a.num_heads
                bounded = a.scores_bounded(P)

                def part(kv_rows, Lk):
                    _, Kp, Vp = ops.qkv_split(None, kv_rows[:, :C], kv_rows[:, C:], None, kw, RMS_EPS, B, Lk, H, P.dtype)
                    o = torch.empty((rows, C), dtype=P.dtype, device=x.device)
                    lse = torch.empty((B, H, L), dtype=torch.float32, device=x.device)
                    ops.attention(Q, Kp, Vp, o, prescaled=True, bounded=bounded, lse=lse)
                    return o, lse
                lo, hi = kv_gather.local_rows()
                own = kv[lo:hi] if getattr(kv_gather, "rehearse", 0) else kv     # (rehearsal on one rank: a slice of the local rows)
                parts = [part(own, hi - lo)]
                kv_full, L_full = kv_gather.finish()
                for r0, r1 in ((0, lo), (hi, L_full)):
                    if r1 > r0:
                        parts.append(part(kv_full[r0:r1], r1 - r0))
                ops.attention_merge(parts, h, B, H, L)
                ops.gemm(h, P.mat(a.fc.weight), x, bias=P.vec(a.fc.bias), residual=x, **out_kw())
                return _mlp_residual(P, self.norm2, self.mlp, x, fold, feed_next)
            kv_full, L_full = kv_gather.finish()
            _, K, Vt = ops.qkv_split(None, kv_full[:, :C], kv_full[:, C:], None, kw, RMS_EPS, B, L_full, a.num_heads,
                                     P.dtype)
        ops.attention(Q, K, Vt, h, prescaled=True, bounded=a.scores_bounded(P))          # h reused as the attention output
        ops.gemm(h, P.mat(a.fc.weight), x, bias=P.vec(a.fc.bias), residual=x, **out_kw())
        return _mlp_residual(P, self.norm2, self.mlp, x, fold, feed_next)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Stand-alone use: x [B, L, C] on a HIP device; precision follows torch.autocast like the reference."""
        B, L, C = x.shape
        P = Prepared.for_module(self, x.device)
        y = x.detach().float().reshape(B * L, C).clone()
        return self.run(P, y, B, L).reshape(B, L, C)


class QK_Norm_CrossAttentionBlock(nn.Module):
    """Cross-attention block; key and value are the same tensor at both call sites of the reference
    (transformer.py:324-377, Pcd_motion.py:462,556-560)."""

    def __init__(self, dim, head_dim, kv_dim=None, ln_bias=False, attn_qkv_bias=False, attn_dropout=0.0,
                 attn_fc_bias=False, attn_fc_dropout=0.0, mlp_ratio=4, mlp_bias=False, mlp_dropout=0.0, use_qk_norm=True):
        super().__init__()
        self.norm_q = nn.LayerNorm(dim, bias=ln_bias)
        self.norm_kv = nn.LayerNorm(kv_dim if kv_dim is not None else dim, bias=ln_bias)
        self.attn = QK_Norm_CrossAttention(dim, head_dim, kv_dim=kv_dim, qkv_bias=attn_qkv_bias, fc_bias=attn_fc_bias,
                                           attn_dropout=attn_dropout, fc_dropout=attn_fc_dropout, use_qk_norm=use_qk_norm)
        self.norm2 = nn.LayerNorm(dim, bias=ln_bias)
        self.mlp = MLP(dim, mlp_ratio=mlp_ratio, bias=mlp_bias, dropout=mlp_dropout)

    # -- the two halves are exposed separately so the decoder can project the mesh points once and
    #    reuse them for every frame (the reference recomputes them T times, Pcd_motion.py:539-553)
    def project_q(self, P: Prepared, query: torch.Tensor, B: int, Lq: int) -> torch.Tensor:
        """query fp32 [B*Lq, C] -> Q[B,H,Lq,64] = RMSNorm(to_q(LN_q(query)))."""
        a = self.attn
        qn = torch.empty(query.shape, dtype=P.dtype, device=query.device)
        ops.layernorm(query, P.vec(self.norm_q.weight), P.vec(self.norm_q.bias), self.norm_q.eps, qn)
        qw, _ = a._qk_w(P)
        if fuse_proj(P, B * Lq):                   # head-major Q (RMSNorm, pre-scale) straight from the projection's epilogue
            Q = torch.empty((B, a.num_heads, Lq, 64), dtype=P.dtype, device=query.device)
            ops.gemm(qn, P.mat(a.to_q.weight), None, bias=P.vec(a.to_q.bias),
                     qkv_heads=(Q, None, None, qw, None, RMS_EPS, ops.Q_PRESCALE, Lq, a.num_heads))
            return Q
        q = torch.empty(query.shape, dtype=P.dtype, device=query.device)
        ops.gemm(qn, P.mat(a.to_q.weight), q, bias=P.vec(a.to_q.bias))
        Q, _, _ = ops.qkv_split(q, None, None, qw, None, RMS_EPS, B, Lq, a.num_heads, P.dtype, q_scale=ops.Q_PRESCALE)
        return Q

    def project_kv(self, P: Prepared, kv: torch.Tensor, B: int, Lk: int, row_map=(0, 0, 0)):
        """kv fp32 rows (optionally gathered through row_map) -> K[B,H,Lk,64], Vt[B,H,64,Lkp]."""
        a = self.attn
        C = kv.shape[1]
        kn = torch.empty((B * Lk, C), dtype=P.dtype, device=kv.device)
        ops.layernorm(kv, P.vec(self.norm_kv.weight), P.vec(self.norm_kv.bias), self.norm_kv.eps, kn, row_map=row_map)
        w_kv, b_kv = P.cat_rows((a.to_k.weight, a.to_v.weight)), P.cat_vecs((a.to_k.bias, a.to_v.bias))
        _, kw = a._qk_w(P)
        if fuse_proj(P, B * Lk) and Lk % 64 == 0:  # head-major K (RMSNorm) and the transposed, key-permuted Vt from the epilogue
            K = torch.empty((B, a.num_heads, Lk, 64), dtype=P.dtype, device=kv.device)
            Vt = torch.empty((B, a.num_heads, 64, Lk), dtype=P.dtype, device=kv.device)
            ops.gemm(kn, w_kv, None, bias=b_kv, qkv_heads=(None, K, Vt, None, kw, RMS_EPS, 1.0, Lk, a.num_heads, True))
            return K, Vt
        kvp = torch.empty((B * Lk, 2 * a.dim), dtype=P.dtype, device=kv.device)
        ops.gemm(kn, w_kv, kvp, bias=b_kv)
        _, K, Vt = ops.qkv_split(None, kvp[:, :a.dim], kvp[:, a.dim:], None, kw, RMS_EPS, B, Lk, a.num_heads, P.dtype)
        return K, Vt

    def project_q_kv(self, P: Prepared, query: torch.Tensor, Lq: int, kv: torch.Tensor, Bk: int, Lk: int, row_map=(0, 0, 0)):
        """project_q(query, 1, Lq) and project_kv(kv, Bk, Lk, row_map) as TWO launches instead of four: both LayerNorms in one
        (m324_layernorm_pair), both projections in one (m324_gemm_pair).  At 2048 rows each of the four is a latency chain of a
        few microseconds of work; the decoder runs them back to back when the q projection is not hoisted out of the block
        (eager forward, M324_HOIST_Q=0, bench.py's block).  Same kernels' arithmetic: results are bit-identical to the two calls."""
        a = self.attn
        if not (PAIR_PROJ and fuse_proj(P, Lq) and fuse_proj(P, Bk * Lk) and Lk % 64 == 0 and query.shape[1] == kv.shape[1]
                and query.dtype == torch.float32 and kv.dtype == torch.float32):
            Q = self.project_q(P, query, 1, Lq)
            K, Vt = self.project_kv(P, kv, Bk, Lk, row_map=row_map)
            return Q, K, Vt
        dev = query.device
        qn = torch.empty(query.shape, dtype=P.dtype, device=dev)
        kn = torch.empty((Bk * Lk, kv.shape[1]), dtype=P.dtype, device=dev)
        ops.layernorm_pair(query, P.vec(self.norm_q.weight), P.vec(self.norm_q.bias), self.norm_q.eps, qn,
                           kv, P.vec(self.norm_kv.weight), P.vec(self.norm_kv.bias), self.norm_kv.eps, kn, row_map1=row_map)
        qw, kw = a._qk_w(P)
        Q = torch.empty((1, a.num_heads, Lq, 64), dtype=P.dtype, device=dev)
        K = torch.empty((Bk, a.num_heads, Lk, 64), dtype=P.dtype, device=dev)
        Vt = torch.empty((Bk, a.num_heads, 64, Lk), dtype=P.dtype, device=dev)
        w_kv, b_kv = P.cat_rows((a.to_k.weight, a.to_v.weight)), P.cat_vecs((a.to_k.bias, a.to_v.bias))
        pair: list = []
        ops.gemm(qn, P.mat(a.to_q.weight), None, bias=P.vec(a.to_q.bias),
                 qkv_heads=(Q, None, None, qw, None, RMS_EPS, ops.Q_PRESCALE, Lq, a.num_heads), defer=pair)
        ops.gemm(kn, w_kv, None, bias=b_kv, qkv_heads=(None, K, Vt, None, kw, RMS_EPS, 1.0, Lk, a.num_heads, True), defer=pair)
        ops.gemm_pair(pair)
        return Q, K, Vt

    def attend(self, P: Prepared, Q, K, Vt, residual: torch.Tensor, res_rows: int, shared_q: bool,
               bf16_stream: bool = False, want_fold: bool = False):
        """x = residual[(row % res_rows)] + fc(attention); x += mlp(LN x).  Returns fp32 [B*Lq, C] -- or, with bf16_stream in
        bf16 inference, the same in bf16: the decoder's stream is two additions deep and only feeds a LayerNorm whose output
        is rounded to bf16 anyway (+1.6e-3 on pcd_moved against 4.7e-3 of the bf16 mode as a whole), while its fp32 form costs
        the out-projection, the MLP and both LayerNorms 400 MB of traffic per clip."""
        a = self.attn
        B, Lq = K.shape[0], Q.shape[2]
        o = torch.empty((B * Lq, a.dim), dtype=P.dtype, device=Q.device)
        ops.attention(Q, K, Vt, o, shared_q=shared_q, prescaled=True)
        xdt = torch.bfloat16 if (bf16_stream and P.dtype == torch.bfloat16 and not torch.is_grad_enabled()) else torch.float32
        x = torch.empty((B * Lq, a.dim), dtype=xdt, device=Q.device)
        fold = LNFold(x) if LNFold.usable(P, B * Lq, a.dim, bf16_stream=xdt == torch.bfloat16) else None
        ops.gemm(o, P.mat(a.fc.weight), x, bias=P.vec(a.fc.bias), residual=residual, res_rows=res_rows,
                 **(fold.producer() if fold is not None else {}))
        _mlp_residual(P, self.norm2, self.mlp, x, fold, feed_next=want_fold)
        return (x, fold) if want_fold else x

    def run(self, P: Prepared, query: torch.Tensor, kv: torch.Tensor, B: int, Lq: int, Lk: int) -> torch.Tensor:
        Q = self.project_q(P, query, B, Lq)
        K, Vt = self.project_kv(P, kv, B, Lk)
        return self.attend(P, Q, K, Vt, query, 0, shared_q=False)

    def forward(self, query, key, value=None):
        if value is not None and value is not key:
            raise NotImplementedError("the HIP path fuses the k/v projections of one tensor (as both reference call sites do)")
        B, Lq, C = query.shape
        Lk = key.shape[1]
        P = Prepared.for_module(self, query.device)
        out = self.run(P, query.detach().float().reshape(B * Lq, C).contiguous(),
                       key.detach().float().reshape(B * Lk, -1).contiguous(), B, Lq, Lk)
        return out.reshape(B, Lq, C)
