"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product (motion324_amd/).

A CPU restatement (plain PyTorch fp32, functional, no nn.Module) of the reference's per-frame
motion-prediction hot path, written from the arithmetic of

    /root/reference/model/Pcd_motion.py           (Motion_Latent_Model.forward, :450-598)
    /root/reference/model/transformer.py          (RMSNorm, MLP, QK_Norm_* blocks, :30-423)
    /root/reference/model/image_encoder/dinov2.py (DinoEncoder, :65-103)
    /root/reference/model/image_encoder/dino/model_dino.py (in-tree restatement of DINOv2 ViT)
    /root/reference/model/loss.py                 (MSELossComputer, :24-66)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline.

Pinning (see tests/golden/make_golden.py and DESIGN.md "Oracle"):
  * the Motion324 part is pinned against the *imported reference itself* run in the build
    container (goldens in tests/golden/*.npz, <= 2e-5 relative);
  * third-party arithmetic that is not under /root/reference:
      - facebookresearch/dinov2 (torch.hub, unpinned HEAD; call site dinov2.py:44,99).  The
        reference's own tests hold no vector for it -> "parity unpinned" upstream.  It is pinned
        here against the reference's in-tree restatement model_dino.py executed on the same
        weights (embeddings :67-137, attention :174-231, LayerScale :293-299, MLP :338-354,
        layer :374-422, final LN :645).
      - xformers==0.0.28 memory_efficient_attention (requirements.txt:4; call sites
        transformer.py:134-139,209-214): softmax(q k^T / sqrt(d_h)) v on [B, L, H, D].  No
        reference vector exists -> pinned by its published definition.

The state dict uses the reference's key names; DINO keys are hub names
``image_encoder.model.*``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

_RESNET_MEAN = (0.485, 0.456, 0.406)   # dinov2.py:7
_RESNET_STD = (0.229, 0.224, 0.225)    # dinov2.py:8


# ------------------------------------------------------------------ building blocks
def layer_norm(x, w, b=None, eps=1e-5):
    """nn.LayerNorm over the last dim (transformer.py:345-346,357,400,411: bias=False, eps 1e-5)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + eps) * w
    return y if b is None else y + b


def rms_norm(x, w, eps=1e-5):
    """transformer.py:36-42."""
    return x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps) * w


def gelu(x):
    """nn.GELU() default = exact erf form (transformer.py:58, Pcd_motion.py:339)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def attention(q, k, v):
    """xformers memory_efficient_attention semantics, layout [B, L, H, D] (transformer.py:134-139)."""
    d = q.shape[-1]
    if q.shape[1] * k.shape[1] > (1 << 22):
        # same definition through torch's fused CPU kernel: the explicit form would materialise a
        # [B,H,L,M] score tensor (5 GB at L = M = 10 368) and make the timed CPU baseline unfair
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
        return o.transpose(1, 2)
    s = torch.einsum("blhd,bmhd->bhlm", q, k) * (d ** -0.5)
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bhlm,bmhd->blhd", p, v)


def mlp(sd: SD, p: str, x):
    """transformer.py:73-81: Linear(no bias) -> GELU -> Linear(no bias)."""
    return gelu(x @ sd[f"{p}.mlp.0.weight"].T) @ sd[f"{p}.mlp.2.weight"].T


def self_attn_block(sd: SD, p: str, x, dh: int):
    """QK_Norm_TransformerBlock.forward (transformer.py:420-423) + QK_Norm_SelfAttention (:191-219)."""
    B, L, C = x.shape
    h = layer_norm(x, sd[f"{p}.norm1.weight"])
    qkv = h @ sd[f"{p}.attn.to_qkv.weight"].T
    q, k, v = qkv.chunk(3, dim=-1)
    q, k, v = (t.reshape(B, L, C // dh, dh) for t in (q, k, v))
    q = rms_norm(q, sd[f"{p}.attn.q_norm.weight"])
    k = rms_norm(k, sd[f"{p}.attn.k_norm.weight"])
    o = attention(q, k, v).reshape(B, L, C)
    x = x + o @ sd[f"{p}.attn.fc.weight"].T
    x = x + mlp(sd, f"{p}.mlp", layer_norm(x, sd[f"{p}.norm2.weight"]))
    return x


def cross_attn_block(sd: SD, p: str, query, kv, dh: int):
    """QK_Norm_CrossAttentionBlock.forward (transformer.py:365-377) + QK_Norm_CrossAttention (:123-144).
    key and value are the same tensor at both call sites (Pcd_motion.py:462,556-560)."""
    B, Lq, C = query.shape
    Lk = kv.shape[1]
    qn = layer_norm(query, sd[f"{p}.norm_q.weight"])
    kn = layer_norm(kv, sd[f"{p}.norm_kv.weight"])
    q = (qn @ sd[f"{p}.attn.to_q.weight"].T).reshape(B, Lq, C // dh, dh)
    k = (kn @ sd[f"{p}.attn.to_k.weight"].T).reshape(B, Lk, C // dh, dh)
    v = (kn @ sd[f"{p}.attn.to_v.weight"].T).reshape(B, Lk, C // dh, dh)
    q = rms_norm(q, sd[f"{p}.attn.q_norm.weight"])
    k = rms_norm(k, sd[f"{p}.attn.k_norm.weight"])
    o = attention(q, k, v).reshape(B, Lq, C)
    x = query + o @ sd[f"{p}.attn.fc.weight"].T
    x = x + mlp(sd, f"{p}.mlp", layer_norm(x, sd[f"{p}.norm2.weight"]))
    return x


# ------------------------------------------------------------------ point features
def point_basis():
    """PointEmbed.__init__ (Pcd_motion.py:163-173): e_j = 2^j * pi, j=0..7, block-diagonal [3, 24]."""
    e = (2.0 ** torch.arange(8, dtype=torch.float32)) * math.pi
    z = torch.zeros(8)
    return torch.stack([torch.cat([e, z, z]), torch.cat([z, e, z]), torch.cat([z, z, e])])


def point_features(sd: SD, xyz, normal, rgb):
    """PointEmbed.forward (Pcd_motion.py:178-187) then point_normal_rgb_proj (:459, :551-553)."""
    proj = xyz @ point_basis()                                       # [B,P,24]
    emb = torch.cat([proj.sin(), proj.cos(), xyz], dim=-1)           # [B,P,51]
    emb = emb @ sd["point_embed.mlp.weight"].T + sd["point_embed.mlp.bias"]
    feat = torch.cat([emb, normal, rgb], dim=-1)                     # [B,P,774]
    return feat @ sd["point_normal_rgb_proj.weight"].T + sd["point_normal_rgb_proj.bias"]


# ------------------------------------------------------------------ positional embedding
def generate_pos_embed(T: int, H: int, W: int, dim: int):
    """Pcd_motion.py:230-266."""
    def axis(n):
        a = torch.arange(n, dtype=torch.float32)
        return 2 * (a / (n - 1)) - 1 if n > 1 else torch.tensor([0.0])
    t, h, w = torch.meshgrid(axis(T), axis(H), axis(W), indexing="ij")
    pos = torch.stack([t, h, w], dim=-1)
    freq = 2.0 ** torch.linspace(0.0, 7.0, dim // 6)
    pos = pos.unsqueeze(-1) * freq.view(1, 1, 1, 1, -1)
    pos = torch.cat([torch.sin(pos), torch.cos(pos)], dim=-1)
    return pos.reshape(1, -1, dim)


def resize_pos_embed(pe, src, dst):
    """Pcd_motion.py:221-228 (trilinear, align_corners=False)."""
    pe = pe.reshape(1, src[0], src[1], src[2], -1).permute(0, 4, 1, 2, 3)
    pe = F.interpolate(pe, size=dst, mode="trilinear", align_corners=False)
    return pe.permute(0, 2, 3, 4, 1).reshape(1, dst[0] * dst[1] * dst[2], -1)


# ------------------------------------------------------------------ DINOv2 ViT
def dino_pos_embed(pos_embed, grid: int):
    """Interpolated position table for a grid x grid patch layout.

    model_dino.py:83-116 (the hub model's interpolate_pos_encoding with interpolate_offset=0.1):
    bicubic, align_corners=False, scale_factor=(grid+0.1)/sqrt(N); class token row kept as is."""
    n = pos_embed.shape[1] - 1
    m = int(math.sqrt(n))
    if m == grid:
        return pos_embed
    dim = pos_embed.shape[-1]
    cls_pe, patch_pe = pos_embed[:, :1], pos_embed[:, 1:]
    patch_pe = patch_pe.reshape(1, m, m, dim).permute(0, 3, 1, 2)
    sf = float((grid + 0.1) / math.sqrt(n))
    patch_pe = F.interpolate(patch_pe, scale_factor=(sf, sf), mode="bicubic", align_corners=False)
    assert patch_pe.shape[-1] == grid and patch_pe.shape[-2] == grid
    patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat([cls_pe, patch_pe], dim=1)


def dino_forward(sd: SD, images, patch: int, dh: int = 64, pre: str = "image_encoder.model"):
    """DinoEncoder.forward (dinov2.py:65-86) + forward_features -> x_norm_patchtokens (:99-103).

    ViT arithmetic per model_dino.py: patch conv k=s=patch (:160-170), cls + pos (:131-134), blocks
    x + ls1*attn(LN1(x)); x + ls2*fc2(gelu(fc1(LN2(x)))) (:393-422), LN eps 1e-6, final LN (:645)."""
    B = images.shape[0]
    mean = torch.tensor(_RESNET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(_RESNET_STD).view(1, 3, 1, 1)
    x = (images - mean) / std
    x = F.conv2d(x, sd[f"{pre}.patch_embed.proj.weight"], sd[f"{pre}.patch_embed.proj.bias"], stride=patch)
    g = x.shape[-1]
    x = x.flatten(2).transpose(1, 2)                                   # [B, g*g, C]
    C = x.shape[-1]
    x = torch.cat([sd[f"{pre}.cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + dino_pos_embed(sd[f"{pre}.pos_embed"], g)
    L = x.shape[1]
    i = 0
    while f"{pre}.blocks.{i}.norm1.weight" in sd:
        p = f"{pre}.blocks.{i}"
        h = layer_norm(x, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], 1e-6)
        qkv = h @ sd[f"{p}.attn.qkv.weight"].T + sd[f"{p}.attn.qkv.bias"]
        q, k, v = (t.reshape(B, L, C // dh, dh) for t in qkv.chunk(3, dim=-1))
        o = attention(q, k, v).reshape(B, L, C)
        o = o @ sd[f"{p}.attn.proj.weight"].T + sd[f"{p}.attn.proj.bias"]
        x = x + sd[f"{p}.ls1.gamma"] * o
        h = layer_norm(x, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], 1e-6)
        h = gelu(h @ sd[f"{p}.mlp.fc1.weight"].T + sd[f"{p}.mlp.fc1.bias"])
        h = h @ sd[f"{p}.mlp.fc2.weight"].T + sd[f"{p}.mlp.fc2.bias"]
        x = x + sd[f"{p}.ls2.gamma"] * h
        i += 1
    x = layer_norm(x, sd[f"{pre}.norm.weight"], sd[f"{pre}.norm.bias"], 1e-6)
    return x[:, 1:]


# ------------------------------------------------------------------ full forward
def forward(sd: SD, sample: Dict[str, torch.Tensor], *, frames: int, d_head: int = 64,
            image_size: int = 224, patch_size: int = 14, loss_weight: float = 1.0,
            stages: Optional[dict] = None, drop: Optional[tuple] = None) -> Dict[str, torch.Tensor]:
    """Motion_Latent_Model.forward in eval mode (Pcd_motion.py:450-598), stages A-G of SURVEY.md 3.2.
    ``drop`` = (keep bool[B*T*g*g*C], p) reproduces the training-mode pos_drop with a given mask.

    ``frames`` = config.training.frames (the length pos_embed was built for, Pcd_motion.py:352,364).
    If ``stages`` is a dict it receives intermediate activations for stage-wise parity tests."""
    dh = d_head
    rec = (lambda k, v: stages.__setitem__(k, v)) if stages is not None else (lambda k, v: None)
    B, N, _ = sample["ref_pcd"].shape
    C = sd["learnable_tokens"].shape[-1]
    K = sd["learnable_tokens"].shape[1]

    # A. shape encoder (:456-464)
    pts = point_features(sd, sample["ref_shape_pcd"], sample["ref_shape_normals"], sample["ref_shape_rgbs"])
    rec("shape_point_feat", pts)
    q_tok = sd["learnable_tokens"].expand(B, -1, -1)
    mesh = cross_attn_block(sd, "encoder_cross_attn", q_tok, pts, dh)
    rec("encoder_out", mesh)
    i = 0
    while f"points_transformer_blocks.{i}.norm1.weight" in sd:
        mesh = self_attn_block(sd, f"points_transformer_blocks.{i}", mesh, dh)
        i += 1
    rec("mesh_feat", mesh)

    # B. image encoder (:466-493)
    vid = sample["rgb_video"]
    _, T, H, W, _ = vid.shape
    img = vid.permute(0, 1, 4, 2, 3).reshape(B * T, 3, H, W)
    img = F.interpolate(img, (image_size, image_size), mode="bilinear", align_corners=False)
    rec("resized", img)
    feats = dino_forward(sd, img, patch_size, dh)                      # [B*T, g*g, C]
    rec("dino_tokens", feats)
    g = image_size // patch_size
    x = feats.reshape(B, T * g * g, C)
    pe = generate_pos_embed(frames, g, g, C)
    if T != frames:
        pe = resize_pos_embed(pe, (frames, g, g), (T, g, g))
    x = x + pe
    if drop is not None:        # training mode: x = self.pos_drop(x) (:369-370,490) with an explicit keep-mask
        keep, p = drop
        x = x * keep.reshape(x.shape).to(x.dtype) * (1.0 / (1.0 - p))
    x = x.reshape(B, T, g * g, C)

    # C. token assembly + input LN (:495-510)
    sp0 = sd["special_token_0"].expand(B, 4, C)
    spr = sd["special_token_rest"].expand(B, 4, C)
    special = torch.stack([sp0] + [spr] * (T - 1), dim=1)
    tok = torch.cat([special, mesh.unsqueeze(1).expand(B, T, K, C), x], dim=2)
    tok = layer_norm(tok, sd["transformer_input_layernorm.weight"])
    rec("trunk_in", tok)
    L = tok.shape[2]

    # D. alternating global / local trunk (:394-409)
    i = 0
    while f"global_transformer_blocks.{i}.norm1.weight" in sd:
        tok = self_attn_block(sd, f"global_transformer_blocks.{i}", tok.reshape(B, T * L, C), dh)
        tok = self_attn_block(sd, f"local_transformer_blocks.{i}", tok.reshape(B * T, L, C), dh)
        tok = tok.reshape(B, T, L, C)
        if i == 0:
            rec("trunk_block0", tok)
        i += 1
    rec("trunk_out", tok)

    # E. per-frame latent tokens (:520)
    lat = tok[:, :, 4:4 + K, :]                                         # [B,T,K,C]

    # F. decoder (:529-579); the reference recomputes the point features per t, results identical
    pf = point_features(sd, sample["ref_pcd"], sample["ref_normal"], sample["ref_rgb"])   # [B,N,C]
    outs = []
    for t in range(T):
        dec = cross_attn_block(sd, "decoder_cross_attn", pf, lat[:, t], dh)
        if t == 0:
            rec("decoder_out_t0", dec)
        h = layer_norm(dec, sd["shared_mlp_output.0.weight"], sd["shared_mlp_output.0.bias"])
        h = gelu(h @ sd["shared_mlp_output.1.weight"].T + sd["shared_mlp_output.1.bias"])
        outs.append(h @ sd["shared_mlp_output.3.weight"].T + sd["shared_mlp_output.3.bias"])
    out = torch.stack(outs, dim=1)                                      # [B,T,N,3]
    res = {"pcd_moved": out}

    # G. loss (:582-592, loss.py:59-64)
    if "point_clouds" in sample:
        mse = ((out - sample["point_clouds"]) ** 2).mean()
        res["xyz_loss"] = mse
        res["loss"] = loss_weight * mse
    return res


def to_torch(d):
    return {k: torch.from_numpy(v) if not isinstance(v, torch.Tensor) else v for k, v in d.items()}
