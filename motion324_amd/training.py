"""Training step of Motion_Latent_Model on libm324 kernels: forward with per-block checkpoints, hand-written
backward (motion324_amd.backward), loss, and the optimizer / gradient plumbing of the reference's train.py:135-219.

`forward_backward(model, sample)` runs the forward (training semantics: DINOv2 frozen and in eval mode, dropout on
video tokens must be 0), keeps only the residual stream at block boundaries (the reference checkpoints every
block / block pair, Pcd_motion.py:375-448) and immediately back-propagates d loss / d params, recomputing block
internals.  No torch autograd graph is built; `Motion_Latent_Model.forward` wraps this in an autograd.Function so that
`loss.backward()` of the reference's train.py delivers the same gradients to `.grad` (and to DDP's hooks).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import backward as bw
from . import ops
from . import switches
from .image_encoder import DINO_EPS
from .lib import ACT_GELU, M324Error
from .prepared import Prepared, compute_dtype, pad_k


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _point_features_train(model, P: Prepared, xyz, normal, rgb):
    """As Motion_Latent_Model._point_features, also returning the two GEMM inputs needed by the weight gradients."""
    C = model.embed_dim
    enc = ops.point_encode(xyz, P.dtype)
    feat = torch.empty((xyz.shape[0], pad_k(C + 6)), dtype=P.dtype, device=xyz.device)
    ops.gemm(enc, P.mat(model.point_embed.mlp.weight), feat, bias=P.vec(model.point_embed.mlp.bias))
    ops.point_concat(normal, rgb, feat, C)
    out = torch.empty((xyz.shape[0], C), dtype=torch.float32, device=xyz.device)
    ops.gemm(feat, P.mat(model.point_normal_rgb_proj.weight), out, bias=P.vec(model.point_normal_rgb_proj.bias))
    return out, enc, feat


def _point_features_bwd(model, P: Prepared, G: bw.GradStore, enc, feat, d_pf: torch.Tensor) -> None:
    """d_pf fp32 [n, C]: gradient of the point features -> weight / bias gradients of the two projections."""
    C = model.embed_dim
    dfeat = bw.linear_bwd(P, G, model.point_normal_rgb_proj.weight, model.point_normal_rgb_proj.bias, feat,
                          ops.cast(d_pf, P.dtype))
    demb = ops.cast(dfeat[:, :C], P.dtype)                     # contiguous copy of the embedding columns
    bw.linear_bwd(P, G, model.point_embed.mlp.weight, model.point_embed.mlp.bias, enc, demb, need_da=False)


DINO_FUSED = switches.flag("M324_TRAIN_DINO_FUSED")          # the frozen image encoder runs its inference form inside a training step
TRAIN_STORE = switches.get("M324_TRAIN_STORE")             # "1": keep block internals while they fit, "0": always recompute


LAST_STEP_BLOCKS = None      # {"trunk_blocks_kept": n, ...} of the most recent forward_backward (bench.py reports it)


def _store_budget(dev: torch.device) -> int:
    """Bytes the forward may spend on kept block internals: half of what is free now (the backward's own work buffers,
    the decoder and the gradient buckets need the rest); 0 with M324_TRAIN_STORE=0."""
    if TRAIN_STORE == "0":
        return 0
    free, _ = torch.cuda.mem_get_info(dev)
    reserved_free = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)      # cached by torch's allocator
    return int(0.5 * (free + reserved_free))


def forward_backward(model, sample: Dict[str, torch.Tensor], grad_scale: float = 1.0, drop_seed: Optional[int] = None,
                     sink=None) -> Tuple[torch.Tensor, torch.Tensor, bw.GradStore]:
    """One training step's forward + hand-written backward (see _forward_backward); inference-only fusions are off.
    sink: a motion324_amd.optim.FusedAdamW -- gradients are then written into its flat buffer and each bucket's
    all-reduce leaves on the optimizer's side stream as soon as the backward has finished the bucket's tensors."""
    from .transformer import fusion_disabled
    global TRAIN_STORE
    with fusion_disabled():
        oom = False
        try:
            return _forward_backward(model, sample, grad_scale, drop_seed, sink)
        except torch.cuda.OutOfMemoryError:
            ops.COLSUMS.clear()                    # queued column sums of the failed attempt: their partial buffers are gone
            # The keep-internals budget is an estimate taken before the forward (half of the free memory); on a shared or
            # smaller device the backward's own peak can still exceed what is left.  One retry with the reference's policy
            # (checkpoint every block, recompute in the backward) -- unless this step already sent gradient buckets to the
            # other ranks: they would reduce this rank's first attempt.
            if TRAIN_STORE == "0" or (sink is not None and getattr(sink, "step_launches", lambda: 0)() > 0):
                raise
            oom = True           # only recorded here: the live exception's traceback keeps the failed attempt's frames -- and
                                 # every activation they hold -- alive until this block is left (PyTorch FAQ on OOM recovery)
        assert oom
        import warnings
        warnings.warn("motion324_amd.training: out of memory with kept block internals; this step is redone with "
                      "checkpoint + recompute (M324_TRAIN_STORE=0 makes that the policy)", RuntimeWarning)
        if drop_seed is None and model.training and float(model.drop_rate) > 0.0:
            drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        torch.cuda.empty_cache()
        keep = TRAIN_STORE
        TRAIN_STORE = "0"
        try:
            return _forward_backward(model, sample, grad_scale, drop_seed, sink)
        finally:
            TRAIN_STORE = keep


def _forward_backward(model, sample: Dict[str, torch.Tensor], grad_scale: float = 1.0, drop_seed: Optional[int] = None,
                      sink=None) -> Tuple[torch.Tensor, torch.Tensor, bw.GradStore]:
    """Returns (loss [0-dim fp32], pcd_moved [B,T,N,3] fp32, GradStore with d(grad_scale * loss)/d param).
    pos_drop (reference :369-370,490; p = transformer.drop_rate, default 0.1) is applied to the video tokens when
    model.training; its mask is a function of ``drop_seed`` (default: drawn from torch's CPU generator, so
    torch.manual_seed(seed + rank) of train.py:61-64 makes runs repeatable and ranks independent)."""
    if "point_clouds" not in sample:
        raise M324Error("training step needs sample['point_clouds'] (the regression target)")
    drop_p = float(model.drop_rate) if model.training else 0.0
    if drop_p > 0.0 and drop_seed is None:
        drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    drop_seed = 0 if drop_seed is None else drop_seed
    ref_pcd = sample["ref_pcd"]
    dev = ref_pcd.device
    if dev.type != "cuda":
        raise M324Error("the training step runs only on a HIP device")
    P = Prepared.for_module(model, dev, compute_dtype())
    G = bw.GradStore(sink)
    B, N, _ = ref_pcd.shape
    C, K = model.embed_dim, model.num_learnable_tokens
    S = sample["ref_shape_pcd"].shape[1]
    T = sample["rgb_video"].shape[1]
    Pn = model.num_patches_h * model.num_patches_w
    Lt = 4 + K + Pn
    weight = float(model.loss_computer._weight)

    # ================================================================ forward (block inputs kept)
    # The image encoder first: it is frozen (no gradient flows through it: it runs its inference form, LayerNorm fold and fused
    # q|k|v epilogue included), and its chip-filling launches give the GPU a backlog while the host enqueues the shape encoder's
    # ~120 latency-bound ones -- a step starts right behind the optimizer's host read, with an empty queue.
    from .transformer import fusion_allowed
    video = _f32c(sample["rgb_video"])
    _, _, Hin, Win, _ = video.shape
    if DINO_FUSED:
        with fusion_allowed(), torch.no_grad():
            dino_x = model.image_encoder.run(P, video.reshape(B * T, Hin, Win, 3), two_streams=False)
    else:
        dino_x = model.image_encoder.run(P, video.reshape(B * T, Hin, Win, 3))

    pts, enc_s, feat_s = _point_features_train(model, P, _f32c(sample["ref_shape_pcd"]).reshape(-1, 3),
                                               _f32c(sample["ref_shape_normals"]), _f32c(sample["ref_shape_rgbs"]))
    query = P.f32(model.learnable_tokens).reshape(K, C).repeat(B, 1)
    mesh = model.encoder_cross_attn.run(P, query, pts, B, K, S)
    mesh_in = []
    for blk in model.points_transformer_blocks:
        mesh_in.append(mesh.clone())
        blk.run(P, mesh, B, K)
    enc_m = model.image_encoder.model
    pos = model._video_pos(P, T)
    sp0, spr = P.f32(model.special_token_0).reshape(4, C), P.f32(model.special_token_rest).reshape(4, C)
    ln_in = model.transformer_input_layernorm
    tok = ops.assemble_tokens(dino_x, P.vec(enc_m.norm.weight), P.vec(enc_m.norm.bias), DINO_EPS, pos, sp0, spr, mesh,
                              P.vec(ln_in.weight), ln_in.eps, B, T, K, Pn, drop_p, drop_seed)
    # Trunk: the reference checkpoints every block (its memory policy on 40-80 GB devices) and pays a second forward in the
    # backward.  With 288 GB the forward keeps each block's internals instead (bw.self_attn_block_internals: LN outputs,
    # q|k|v in both layouts, attention output + LSE, MLP pre-activation: ~37 KB per token and block = 1.15 GB per block at
    # B = 8, 28 GB for the 24 trunk blocks) while they fit a budget; blocks past the budget fall back to the checkpoint.
    trunk_in, trunk_saved = [], []
    rows_tok = tok.shape[0]
    per_block = rows_tok * C * 50                            # bytes, upper bound of one block's kept tensors
    budget = _store_budget(dev)
    for gblk, lblk in zip(model.global_transformer_blocks, model.local_transformer_blocks):
        for blk, (Bb, Lb) in ((gblk, (B, T * Lt)), (lblk, (B * T, Lt))):
            if budget >= per_block:
                budget -= per_block
                nxt = torch.empty_like(tok)
                trunk_saved.append(bw.self_attn_block_internals(blk, P, tok, Bb, Lb, x_out=nxt))
                trunk_in.append(tok)
                tok = nxt
            else:
                trunk_saved.append(None)
                trunk_in.append(tok.clone())
                blk.run(P, tok, Bb, Lb)

    dec = model.decoder_cross_attn
    head_ln, head_fc1, head_fc2 = model.shared_mlp_output[0], model.shared_mlp_output[1], model.shared_mlp_output[3]
    w3, b3 = P.f32(head_fc2.weight), P.vec(head_fc2.bias)
    out = torch.empty((B, T, N, 3), dtype=torch.float32, device=dev)
    pcd, nrm, rgb = (_f32c(sample[k]) for k in ("ref_pcd", "ref_normal", "ref_rgb"))
    dec_saved = []
    # bytes kept per sample: decoder block internals (o 2C, x_mid 4C, h2 2C, z 8C, g 8C, x 4C) + head (h, z2, h2: 6C) per row =
    # 34 C, plus the q / k|v side tensors; 38 C bounds it
    per_dec = T * N * C * 38
    for b in range(B):
        pf, enc_p, feat_p = _point_features_train(model, P, pcd[b], nrm[b].contiguous(), rgb[b].contiguous())
        tok_b = tok[b * T * Lt:(b + 1) * T * Lt]
        keep = budget >= per_dec
        if keep:                                              # same policy as the trunk: internals kept while they fit
            budget -= per_dec
            x = torch.empty((T * N, C), dtype=torch.float32, device=dev)
            sv = bw.cross_attn_block_internals(dec, P, pf, tok_b, T, N, K, kv_row_map=(K, Lt, 4), shared_q=True, x_out=x)
        else:
            Q = dec.project_q(P, pf, 1, N)
            Kd, Vd = dec.project_kv(P, tok_b, T, K, row_map=(K, Lt, 4))
            x = dec.attend(P, Q, Kd, Vd, pf, N, shared_q=True)                        # fp32 [T*N, C]
            sv = None
        h = torch.empty(x.shape, dtype=P.dtype, device=dev)
        ops.layernorm(x, P.vec(head_ln.weight), P.vec(head_ln.bias), head_ln.eps, h)
        h2 = torch.empty(x.shape, dtype=P.dtype, device=dev)
        if keep:
            z2 = torch.empty(x.shape, dtype=P.dtype, device=dev)              # the pre-activation, or gelu' of it (M324_GELU_GRAD_FWD)
            ops.gemm(h, P.mat(head_fc1.weight), h2, bias=P.vec(head_fc1.bias), act=ACT_GELU, **{"gelu_grad_out" if bw.GELU_GRAD_FWD else "preact_out": z2})
            head = (h, z2, h2)
        else:
            ops.gemm(h, P.mat(head_fc1.weight), h2, bias=P.vec(head_fc1.bias), act=ACT_GELU)
            head = None
        ops.linear_n3(h2, w3, b3, out[b])
        dec_saved.append((pf, enc_p, feat_p, x, sv, head))
    global LAST_STEP_BLOCKS
    LAST_STEP_BLOCKS = {"trunk_blocks_kept": sum(1 for t in trunk_saved if t is not None), "trunk_blocks_recomputed":
                        sum(1 for t in trunk_saved if t is None), "decoder_samples_kept": sum(1 for d in dec_saved if d[4] is not None),
                        "decoder_samples_recomputed": sum(1 for d in dec_saved if d[4] is None)}
    target = _f32c(sample["point_clouds"])
    mse = ops.mse(out, target, 1.0)
    loss = mse * weight

    # ================================================================ backward
    d_out = ops.mse_bwd(out, target, weight * grad_scale)                             # [B,T,N,3]
    d_tok = torch.zeros((B * T * Lt, C), dtype=torch.float32, device=dev)
    for b in range(B):
        pf, enc_p, feat_p, x, sv, head = dec_saved[b]
        tok_b = tok[b * T * Lt:(b + 1) * T * Lt]
        # head: LN -> Linear -> GELU -> Linear(3)
        if head is not None:
            h, z2, h2 = head
        else:
            h = torch.empty(x.shape, dtype=P.dtype, device=dev)
            ops.layernorm(x, P.vec(head_ln.weight), P.vec(head_ln.bias), head_ln.eps, h)
            z2 = torch.empty(x.shape, dtype=P.dtype, device=dev)
            if bw.GELU_GRAD_FWD:
                h2 = torch.empty(x.shape, dtype=P.dtype, device=dev)
                ops.gemm(h, P.mat(head_fc1.weight), h2, bias=P.vec(head_fc1.bias), act=ACT_GELU, gelu_grad_out=z2)
            else:
                ops.gemm(h, P.mat(head_fc1.weight), z2, bias=P.vec(head_fc1.bias))
                h2 = ops.gelu(z2)
        if bw.GELU_GRAD_FWD:                                  # z2 holds gelu'(pre-activation): the 3-wide Linear's backward multiplies by it
            dz2, dW3, db3 = ops.linear_n3_bwd(h2, w3, d_out[b], mul_by=z2)
        else:
            dh2, dW3, db3 = ops.linear_n3_bwd(h2, w3, d_out[b])
            dz2 = ops.gelu_bwd(z2, dh2)
        G.add(head_fc2.weight, dW3)
        G.add(head_fc2.bias, db3)
        dh = bw.linear_bwd(P, G, head_fc1.weight, head_fc1.bias, h, dz2)
        dx = torch.empty(x.shape, dtype=torch.float32, device=dev)
        dw, db = ops.layernorm_bwd(x, P.vec(head_ln.weight), head_ln.eps, dh, dx, accumulate=False, reduce=not ops.DEFER_COLSUM)
        G.add(head_ln.weight, dw)
        G.add(head_ln.bias, db)
        d_pf = bw.cross_attn_block_bwd(dec, P, G, pf, tok_b, dx, T, N, K, kv_row_map=(K, Lt, 4), shared_q=True,
                                       d_kv=d_tok[b * T * Lt:(b + 1) * T * Lt], saved=sv)
        sv = head = None
        _point_features_bwd(model, P, G, enc_p, feat_p, d_pf)
        dec_saved[b] = None
    G.done(list(model.shared_mlp_output.parameters()) + list(dec.parameters()))

    n_pairs = len(model.global_transformer_blocks)
    carry = bw.Carry()                                       # d_tok's bf16 form + column sums travel from block to block
    for i in reversed(range(n_pairs)):
        bw.self_attn_block_bwd(model.local_transformer_blocks[i], P, G, trunk_in[2 * i + 1], d_tok, B * T, Lt,
                               saved=trunk_saved[2 * i + 1], carry=carry)
        G.done(model.local_transformer_blocks[i].parameters())
        trunk_saved[2 * i + 1] = None
        bw.self_attn_block_bwd(model.global_transformer_blocks[i], P, G, trunk_in[2 * i], d_tok, B, T * Lt,
                               saved=trunk_saved[2 * i], carry=carry)
        G.done(model.global_transformer_blocks[i].parameters())
        trunk_in[2 * i + 1] = trunk_in[2 * i] = trunk_saved[2 * i] = None

    # token assembly + input LayerNorm: recompute the un-normalised concatenation, LN backward over every row (the LN
    # weight sees the video rows too), then fold the rows that carry parameters / the mesh latents
    pre = ops.assemble_tokens(dino_x, P.vec(enc_m.norm.weight), P.vec(enc_m.norm.bias), DINO_EPS, pos, sp0, spr, mesh,
                              None, ln_in.eps, B, T, K, Pn, drop_p, drop_seed)
    d_pre = torch.empty_like(pre)
    dw, _ = ops.layernorm_bwd(pre, P.vec(ln_in.weight), ln_in.eps, d_tok, d_pre, accumulate=False, reduce=not ops.DEFER_COLSUM)
    G.add(ln_in.weight, dw)
    d4 = d_pre.reshape(B, T, Lt, C)
    G.add(model.special_token_0, ops.colsum(d4[:, 0, :4].reshape(B, 4 * C).contiguous()).reshape(1, 4, C))
    if T > 1:
        G.add(model.special_token_rest,
              ops.colsum(d4[:, 1:, :4].reshape(B * (T - 1), 4 * C).contiguous()).reshape(1, 4, C))
    G.done([ln_in.weight, model.special_token_0, model.special_token_rest])
    d_mesh = torch.empty((B * K, C), dtype=torch.float32, device=dev)
    for b in range(B):
        d_mesh[b * K:(b + 1) * K] = ops.colsum(d4[b, :, 4:4 + K].reshape(T, K * C).contiguous()).reshape(K, C)

    carry = bw.Carry()
    for i in reversed(range(len(model.points_transformer_blocks))):
        bw.self_attn_block_bwd(model.points_transformer_blocks[i], P, G, mesh_in[i], d_mesh, B, K, carry=carry)
        G.done(model.points_transformer_blocks[i].parameters())
    d_pts = torch.zeros((B * S, C), dtype=torch.float32, device=dev)
    d_query = bw.cross_attn_block_bwd(model.encoder_cross_attn, P, G, query, pts, d_mesh, B, K, S, d_kv=d_pts)
    G.add(model.learnable_tokens, ops.colsum(d_query.reshape(B, K * C)).reshape(1, K, C))
    G.done(list(model.encoder_cross_attn.parameters()) + [model.learnable_tokens])
    _point_features_bwd(model, P, G, enc_s, feat_s, d_pts)
    G.done(p for p in model.parameters() if p.requires_grad)           # everything else (point embedding, stragglers)
    return loss, out, G
