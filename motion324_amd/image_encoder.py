"""DINOv2 ViT-B/14 image encoder on libm324 kernels, with the hub model's parameter names.

Replaces the reference's model/image_encoder/dinov2.py::DinoEncoder, which wraps
``torch.hub.load('facebookresearch/dinov2', 'dinov2_vitb14')`` (:44) -- a network download of an
unpinned third-party module.  Here the architecture is built locally (random-init; weights come
from a checkpoint's ``image_encoder.model.*`` entries, which use the hub names kept below), and
the arithmetic follows the reference's in-tree restatement model/image_encoder/dino/model_dino.py:
patch conv k=s=14 (:160-170), [CLS] + interpolated position table (:83-134, bicubic, +0.1 offset),
12 x { x + ls1 * proj(attn(LN x)); x + ls2 * fc2(gelu(fc1(LN x))) } (:393-422), LayerNorm eps 1e-6.
The final LayerNorm + CLS drop (x_norm_patchtokens, dinov2.py:99-103) is fused into the trunk's token
assembly kernel (m324_assemble_tokens), so ``run`` returns the PRE-norm residual stream.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .lib import ACT_GELU
from .prepared import Prepared, pad_k
from .transformer import LNFold, fuse_qkv, hp_consumer

DINO_EPS = 1e-6
TWO_STREAMS = True       # run(): under graph capture, two half-batches of frames as two branches (see the comment there)
_STREAMS = {}


def _second_stream(dev: torch.device) -> "torch.cuda.Stream":
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _STREAMS:
        _STREAMS[idx] = torch.cuda.Stream(device=dev)
    return _STREAMS[idx]


class _Attention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden, bias=True)
        self.fc2 = nn.Linear(hidden, dim, bias=True)


class _LayerScale(nn.Module):
    def __init__(self, dim, init_values=1.0):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class _Block(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=DINO_EPS)
        self.attn = _Attention(dim)
        self.ls1 = _LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=DINO_EPS)
        self.mlp = _Mlp(dim, hidden)
        self.ls2 = _LayerScale(dim)


class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)


class DinoVisionTransformer(nn.Module):
    """Parameter tree of hub ``dinov2_vitb14`` (names = the released checkpoints' ``image_encoder.model.*`` keys)."""

    def __init__(self, embed_dim=768, depth=12, num_heads=12, patch_size=14, pos_grid=37, mlp_ratio=4):
        super().__init__()
        self.embed_dim, self.patch_size, self.num_heads, self.pos_grid = embed_dim, patch_size, num_heads, pos_grid
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + pos_grid * pos_grid, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        self.patch_embed = _PatchEmbed(embed_dim, patch_size)
        self.blocks = nn.ModuleList([_Block(embed_dim, embed_dim * mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=DINO_EPS)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def interpolated_pos(self, grid: int) -> torch.Tensor:
        """[1 + grid*grid, C] position table for a grid x grid image (model_dino.py:83-116)."""
        pe = self.pos_embed.detach().float()
        n = pe.shape[1] - 1
        m = int(math.sqrt(n))
        if m == grid:
            return pe[0].contiguous()
        dim = pe.shape[-1]
        patch = pe[:, 1:].reshape(1, m, m, dim).permute(0, 3, 1, 2)
        sf = float((grid + 0.1) / math.sqrt(n))
        patch = F.interpolate(patch, scale_factor=(sf, sf), mode="bicubic", align_corners=False)
        if patch.shape[-1] != grid or patch.shape[-2] != grid:
            raise ValueError("Width or height does not match with the interpolated position embeddings")
        patch = patch.permute(0, 2, 3, 1).reshape(-1, dim)
        return torch.cat([pe[0, :1], patch], dim=0).contiguous()


class DinoEncoder(nn.Module):
    """Frozen DINOv2 feature extractor (reference dinov2.py:39-131): parameters never require grad and the
    module stays in eval mode (train() override, :126-131)."""

    def __init__(self, patch_size=14, model_name="dinov2_vitb14", embed_dim=768, depth=12, num_heads=12, pos_grid=37):
        super().__init__()
        if model_name != "dinov2_vitb14":
            raise NotImplementedError(model_name)
        self.model = DinoVisionTransformer(embed_dim=embed_dim, depth=depth, num_heads=num_heads, patch_size=patch_size,
                                           pos_grid=pos_grid)
        self.patch_size = patch_size
        self.image_size = 224
        self.num_patches_per_dim = self.image_size // patch_size
        self.num_patches_total = self.num_patches_per_dim ** 2
        for p in self.model.parameters():
            p.requires_grad = False
        self.model.eval()

    def train(self, mode: bool = False):
        super().train(mode)
        self.model.eval()
        return self

    def run(self, P: Prepared, video: torch.Tensor, two_streams: bool = True, out: torch.Tensor = None) -> torch.Tensor:
        """video [F, Hin, Win, 3] fp32 in [0,1] or uint8 in 0..255 (channel-last, any size) -> pre-final-norm tokens
        [F * (1 + g*g), C] fp32 (written into `out` when given: a caller that already holds some frames' tokens -- the long-video
        driver's anchor frame -- passes the rows behind them).  Resize to 224^2 + ImageNet normalisation + im2col happen in one
        kernel (Pcd_motion.py:470-472, dinov2.py:78-80)."""
        m = self.model
        Fr = video.shape[0]
        g, C, H = self.num_patches_per_dim, m.embed_dim, m.num_heads
        Lt = 1 + g * g
        kp = pad_k(3 * self.patch_size ** 2)
        patches = ops.patchify(video, self.image_size, self.patch_size, kp, P.dtype)
        pos = P.derived(f"dino_pos{g}", (m.pos_embed,), lambda: m.interpolated_pos(g).to(P.device))
        if out is not None and (out.shape != (Fr * Lt, C) or out.dtype != torch.float32 or not out.is_contiguous()):
            raise ValueError(f"DinoEncoder.run: out must be contiguous fp32 [{Fr * Lt}, {C}], got {out.dtype}{tuple(out.shape)}")
        x = out if out is not None else torch.empty((Fr * Lt, C), dtype=torch.float32, device=video.device)
        ops.gemm(patches, P.mat(m.patch_embed.proj.weight), x, bias=P.vec(m.patch_embed.proj.bias),
                 residual=pos[1:], res_rows=g * g, row_map=(g * g, Lt, 1))
        ops.dino_cls_rows(P.f32(m.cls_token).reshape(-1), pos[0], x, Fr, Lt)
        h = torch.empty((Fr * Lt, C), dtype=P.dtype, device=video.device)
        fused = fuse_qkv(P, Fr * Lt, Lt)           # projection epilogue writes head-major Q / K / V (no m324_qkv_split pass)
        if fused:
            Qh, Kh, Vh = (torch.empty((Fr, H, Lt, 64), dtype=P.dtype, device=video.device) for _ in range(3))
        else:
            qkv = torch.empty((Fr * Lt, 3 * C), dtype=P.dtype, device=video.device)
        h1 = torch.empty((Fr * Lt, m.blocks[0].mlp.fc1.out_features), dtype=P.dtype, device=video.device)
        # every weight in its compute form BEFORE the fork below (a cold cache converts on the current stream)
        # LayerNorm fold (transformer.LNFold): norm1 / norm2 ride in the q|k|v and fc1 projections
        folding = LNFold.usable(P, Fr * Lt, C)
        W = [dict(n1=(P.vec(b.norm1.weight), P.vec(b.norm1.bias)), qkv=(P.mat(b.attn.qkv.weight), P.vec(b.attn.qkv.bias)),
                  proj=(P.mat(b.attn.proj.weight), P.vec(b.attn.proj.bias), P.vec(b.ls1.gamma)),
                  n2=(P.vec(b.norm2.weight), P.vec(b.norm2.bias)), fc1=(P.mat(b.mlp.fc1.weight), P.vec(b.mlp.fc1.bias)),
                  fc2=(P.mat(b.mlp.fc2.weight), P.vec(b.mlp.fc2.bias), P.vec(b.ls2.gamma)),
                  f_qkv=P.folded(b.norm1.weight, b.norm1.bias, b.attn.qkv.weight, b.attn.qkv.bias) if folding else None,
                  f_fc1=P.folded(b.norm2.weight, b.norm2.bias, b.mlp.fc1.weight, b.mlp.fc1.bias) if folding else None)
             for b in m.blocks]

        def block(w, f0, f1, fold=None, feed_next=True):
            """One ViT block on frames f0 .. f1 (rows f0*Lt .. f1*Lt of every buffer), on the current stream."""
            r = slice(f0 * Lt, f1 * Lt)
            if fold is not None:
                src, (wq, csq, bq) = fold.xb, w["f_qkv"]
                lnk = dict(ln=fold.ln(DINO_EPS, csq))
            else:
                ops.layernorm(x[r], *w["n1"], DINO_EPS, h[r])
                src, wq, bq, lnk = h[r], w["qkv"][0], w["qkv"][1], {}
            if fused:
                ops.gemm(src, wq, None, bias=bq,
                         qkv_heads=(Qh[f0:f1], Kh[f0:f1], Vh[f0:f1], None, None, 0.0, ops.Q_PRESCALE, Lt, H), **lnk)
                ops.attention(Qh[f0:f1], Kh[f0:f1], Vh[f0:f1], h[r], prescaled=True, v_rowmajor=True)
            else:
                ops.gemm(src, wq, qkv[r], bias=bq, **lnk)
                Q, K, Vt = ops.qkv_split(qkv[r, :C], qkv[r, C:2 * C], qkv[r, 2 * C:], None, None, 0.0, f1 - f0, Lt, H, P.dtype,
                                         q_scale=ops.Q_PRESCALE)
                ops.attention(Q, K, Vt, h[r], prescaled=True)
            ops.gemm(h[r], w["proj"][0], x[r], bias=w["proj"][1], gamma=w["proj"][2], residual=x[r],
                     **(fold.producer() if fold is not None else {}))
            if fold is not None:
                w1, cs1, b1 = w["f_fc1"]
                ops.gemm(fold.xb, w1, h1[r], bias=b1, act=ACT_GELU, ln=fold.ln(DINO_EPS, cs1, merged=hp_consumer(fold.rows, w1.shape[0], w1.shape[1])))
            else:
                ops.layernorm(x[r], *w["n2"], DINO_EPS, h[r])
                ops.gemm(h[r], w["fc1"][0], h1[r], bias=w["fc1"][1], act=ACT_GELU)
            ops.gemm(h1[r], w["fc2"][0], x[r], bias=w["fc2"][1], gamma=w["fc2"][2], residual=x[r],
                     **(fold.producer() if (fold is not None and feed_next) else {}))

        def chain(f0, f1):
            """All blocks on frames f0 .. f1; the statistics of the stream start from m324_rowstats and then travel from
            epilogue to epilogue (the final LayerNorm lives in the trunk's token assembly and reads the fp32 stream)."""
            fold = LNFold(x[f0 * Lt:f1 * Lt]).from_stream(x[f0 * Lt:f1 * Lt], DINO_EPS) if folding else None
            for i, w in enumerate(W):
                block(w, f0, f1, fold, feed_next=i + 1 < len(W))

        # Frames are independent through the whole ViT.  At the BASELINE clip (32 x 257 = 8224 rows) every GEMM of a
        # block fills only ~76 % of its last round of workgroups (33 x 12 tiles of 256 x 256 on 256 CUs = 1.55 rounds);
        # two half-batches as two branches of the hipGraph let one half's workgroups take the CUs the other half's last
        # round leaves idle (tools/overlap_lab: q|k|v and fc1 -13 % per pair of half launches, out-projection and fc2
        # unchanged; the whole encoder 2.46 -> 2.38 ms, the clip -1 %).  Only while a graph is being captured: an eager
        # pass would pay twice the host launches for it (Python enqueues ~14 us per launch, a half GEMM runs 9 - 20 us),
        # and results are bit-identical either way (rows are independent, every tile kernel sums k in the same order).
        side = None
        if TWO_STREAMS and two_streams and Fr >= 8 and torch.cuda.is_current_stream_capturing():
            side = _second_stream(video.device)
        if side is None:
            chain(0, Fr)
            return x
        main, Fh = torch.cuda.current_stream(video.device), (Fr + 1) // 2
        side.wait_stream(main)
        folds = [None, None]
        if folding:
            folds[0] = LNFold(x[:Fh * Lt]).from_stream(x[:Fh * Lt], DINO_EPS)
            with torch.cuda.stream(side):
                folds[1] = LNFold(x[Fh * Lt:]).from_stream(x[Fh * Lt:], DINO_EPS)
        for i, w in enumerate(W):
            block(w, 0, Fh, folds[0], feed_next=i + 1 < len(W))
            with torch.cuda.stream(side):
                block(w, Fh, Fr, folds[1], feed_next=i + 1 < len(W))
        main.wait_stream(side)
        return x

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        """Reference-compatible surface: image [B,3,224,224] in [0,1] -> x_norm_patchtokens [B,256,C]
        (dinov2.py:65-86).  The normalised output is produced by m324_layernorm here."""
        B, Cc, Hh, Ww = image.shape
        assert Hh == self.image_size and Ww == self.image_size, \
            f"Input image size must be {self.image_size}x{self.image_size}, but got {Hh}x{Ww}"
        P = Prepared.for_module(self, image.device)
        x = self.run(P, image.detach().float().permute(0, 2, 3, 1).contiguous())
        Lt = 1 + self.num_patches_total
        out = torch.empty((B * self.num_patches_total, x.shape[1]), dtype=torch.float32, device=image.device)
        ops.layernorm(x, P.vec(self.model.norm.weight), P.vec(self.model.norm.bias), DINO_EPS, out,
                      row_map=(self.num_patches_total, Lt, 1))
        return out.reshape(B, self.num_patches_total, -1)
