"""Motion_Latent_Model on MI355X: the reference's model API, executed by libm324 HIP kernels.

Drop-in for the reference's model/Pcd_motion.py::Motion_Latent_Model (:268-598): same constructor
(``config`` with the keys of configs/dyscene.yaml), same parameter / buffer names and shapes
(state-dict compatible), same ``forward(sample) -> EasyDict{input_data, pcd_moved[, loss_metrics]}``.
Select it from the reference's own callers with ``model.class_name=motion324_amd.Pcd_motion.Motion_Latent_Model``
(train.py:84-86, scripts/inference_with_video_mesh.py:309-311).

Precision follows the caller exactly like the reference: under ``torch.autocast('cuda', bf16)`` the
bf16-operand / fp32-accumulate kernels run; without autocast the fp32 parity kernels run.  The point
Fourier embedding is always evaluated in fp32 (the reference's bf16 einsum destroys the phase).

There is no CPU path: inputs must live on a HIP device and libm324.so must be built.
"""
from __future__ import annotations

import contextlib
import os

from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, parallel, switches
from .easydict import EasyDict as edict
from .image_encoder import DINO_EPS, DinoEncoder
from .lib import ACT_GELU, M324Error
from .loss import MSELossComputer
from .prepared import Prepared, bump_generation, compute_dtype, pad_k
from .timing import span
from .transformer import LN_EPS, LNFold, QK_Norm_CrossAttentionBlock, QK_Norm_TransformerBlock, init_weights

# A/B switches: one table with defaults and meanings in motion324_amd/switches.py (bench.py echoes non-default values)
AUTO_GRAPH = switches.flag("M324_AUTO_GRAPH")               # forward(): graph replay for repeated inference shapes
FUSE_HEAD_N3 = switches.flag("M324_FUSE_HEAD")              # head fc1 + GELU + 768 -> 3 in one GEMM epilogue (bf16 inference)
KV_REHEARSE = int(switches.get("M324_KV_REHEARSE") or 0)       # one-rank rehearsal of the overlapped exchange as rank 0 of W
KV_OVERLAP = switches.flag("M324_KV_OVERLAP")              # frame-parallel: own keys attended while the K|V all-gather is in flight
BF16_DECODER_STREAM = switches.flag("M324_BF16_DECODER")    # the decoder's residual stream in bf16 (bf16 inference only)
HOIST_DECODER_Q = switches.flag("M324_HOIST_Q")             # hoisted decoder q projection (graph capture)
DECODE_ROWS = int(switches.get("M324_DECODE_ROWS"))         # max (frames x points) rows per decoder pass: bounds the [rows, 4C] MLP buffer


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class PointEmbed(nn.Module):
    """Fourier point embedding (reference Pcd_motion.py:157-187)."""

    def __init__(self, hidden_dim=48, dim=768):
        super().__init__()
        if hidden_dim != 48:
            raise NotImplementedError("m324_point_encode implements the reference's 48 Fourier features")
        self.embedding_dim = hidden_dim
        e = torch.pow(2, torch.arange(hidden_dim // 6)).float() * torch.pi
        z = torch.zeros(hidden_dim // 6)
        self.register_buffer("basis", torch.stack([torch.cat([e, z, z]), torch.cat([z, e, z]), torch.cat([z, z, e])]))
        self.mlp = nn.Linear(hidden_dim + 3, dim)


def generate_pos_embed(T: int, H: int, W: int, embed_dim: int) -> torch.Tensor:
    """3-D Fourier position table [1, T*H*W, embed_dim] (reference Pcd_motion.py:230-266); host-side constant."""
    def axis(n):
        a = torch.arange(n, dtype=torch.float32)
        return 2 * (a / (n - 1)) - 1 if n > 1 else torch.tensor([0.0], dtype=torch.float32)
    t, h, w = torch.meshgrid(axis(T), axis(H), axis(W), indexing="ij")
    pos = torch.stack([t, h, w], dim=-1).unsqueeze(-1)
    freq = (2.0 ** torch.linspace(0.0, 7.0, embed_dim // 6)).view(1, 1, 1, 1, -1)
    pos = pos * freq
    return torch.cat([torch.sin(pos), torch.cos(pos)], dim=-1).reshape(1, -1, embed_dim)


def resize_pos_embed(posemb, src_shape, target_shape):
    """Trilinear resize of the position table (reference Pcd_motion.py:221-228); cached per clip length."""
    posemb = posemb.reshape(1, src_shape[0], src_shape[1], src_shape[2], -1).permute(0, 4, 1, 2, 3)
    posemb = F.interpolate(posemb, size=target_shape, mode="trilinear", align_corners=False)
    return posemb.permute(0, 2, 3, 4, 1).reshape(1, target_shape[0] * target_shape[1] * target_shape[2], -1)


# inference only: run the shape encoder on a side stream (M324_OVERLAP=0 disables, for A/B measurements)
OVERLAP_SHAPE_ENCODER = switches.flag("M324_OVERLAP")
_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(dev: torch.device) -> "torch.cuda.Stream":
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _SIDE_STREAMS:
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[idx]


class _KVGather:
    """Frame-parallel global attention (BASELINE config 5): all ranks' token-major k|v projections of one global block.

    ``start(kv_local)`` is called right after the k|v projection GEMM and launches the collective on a side stream
    (RCCL over xGMI: 2 x 82 944 x 768 bf16 = 255 MB assembled per block at T = 256); the compute stream meanwhile runs
    the block's own q projection and q split; ``finish()`` joins.  Even shards with B = 1 (the 256 / 8 case) go through
    ONE all_gather_into_tensor straight into a buffer that every global block reuses -- rank order = frame order, so
    the gathered rows already are the clip's token order and nothing is copied afterwards.  Uneven shards (or B > 1,
    where the clip order is batch-major) pad to the largest shard and compact once."""

    def __init__(self, B, T_local, Lt, frames, group, dev, rank=0):
        self.B, self.T, self.Lt, self.frames, self.group, self.dev = B, T_local, Lt, list(frames), group, dev
        self.world = len(self.frames)
        self.rank = rank
        # M324_KV_OVERLAP: the block attends to the rank's own keys while the exchange is in flight and merges the remote keys'
        # partial softmax afterwards (transformer.QK_Norm_SelfAttentionBlock.run); B = 1 (c5): the clip order is frame-major, the
        # rank's own rows are ONE contiguous range of the gathered rows
        self.overlap = KV_OVERLAP and B == 1 and self.world > 1
        # M324_KV_REHEARSE=W on ONE rank (bench.py --mode frame-parallel with M324_BENCH_COLLECT=1): the block runs the overlapped
        # form as rank 0 of W would -- the first 1 / W of its frames are "own" keys, the rest arrive by the (identity) gather -- so
        # that the cost of the split attention + merge can be measured where no second GPU is available
        self.rehearse = 0
        if KV_OVERLAP and B == 1 and self.world == 1 and KV_REHEARSE > 1 and T_local >= KV_REHEARSE:
            self.rehearse, self.overlap = KV_REHEARSE, True
        self.T_full = sum(self.frames)
        self.even = B == 1 and all(f == self.frames[0] for f in self.frames)
        self.comm = torch.cuda.Stream(device=dev)
        self.buf = None
        self.full = None                         # uneven shards under segmented capture: the compacted rows at a fixed address
        self._pending = None
        self._deferred = None
        self.launch_log = []                     # ids of the streams the collectives were enqueued on (tests)

    def start(self, kv_local: torch.Tensor) -> None:
        """kv_local: contiguous [B * T_local * Lt, 2C]."""
        from . import graph
        seg = graph.active_segmenter()
        if seg is not None and self.overlap:
            # segmented capture, overlapped form: the collective is STARTED between two graphs (eagerly, on the side stream), the next
            # graph holds the work that runs beside it (q projection, the rank's own keys), finish() cuts again and joins
            self._seg_fixed(kv_local)
            seg.cut(lambda: self._start(kv_local))
            return
        if seg is not None:                            # segmented hipGraph capture: the exchange runs BETWEEN two graphs, in finish()
            self._deferred = kv_local
            return
        self._start(kv_local)

    def local_rows(self):
        """(lo, hi): the rank's own token rows inside the gathered [T_full * Lt] rows (B = 1)."""
        if self.rehearse:
            return 0, (self.T // self.rehearse) * self.Lt
        lo = sum(self.frames[:self.rank]) * self.Lt
        return lo, lo + self.frames[self.rank] * self.Lt

    def _seg_fixed(self, kv_local: torch.Tensor) -> torch.Tensor:
        """The buffer a captured graph reads the gathered rows from (fixed address across replays)."""
        rows, width = kv_local.shape
        if self.even:
            if self.buf is None or self.buf.dtype != kv_local.dtype or self.buf.shape[1] != width:
                self.buf = torch.empty((self.world * rows, width), dtype=kv_local.dtype, device=self.dev)
            return self.buf
        if self.full is None or self.full.dtype != kv_local.dtype or self.full.shape[1] != width:
            self.full = torch.empty((self.B * self.T_full * self.Lt, width), dtype=kv_local.dtype, device=self.dev)
        return self.full

    def _start(self, kv_local: torch.Tensor) -> None:
        rows, width = kv_local.shape
        main = torch.cuda.current_stream(self.dev)
        self.comm.wait_stream(main)              # the projection has been enqueued; earlier readers of `buf` too
        with torch.cuda.stream(self.comm):
            self.launch_log.append(self.comm.cuda_stream)
            if self.even:
                if self.buf is None or self.buf.dtype != kv_local.dtype or self.buf.shape[1] != width:
                    self.buf = torch.empty((self.world * rows, width), dtype=kv_local.dtype, device=self.dev)
                work = parallel.all_gather_into(self.buf, kv_local, group=self.group, async_op=True)
                self._pending = (work, None, kv_local)
            else:
                mx = max(self.frames) * self.Lt
                send = torch.zeros((self.B, mx, width), dtype=kv_local.dtype, device=self.dev)
                send[:, :self.T * self.Lt] = kv_local.reshape(self.B, self.T * self.Lt, width)
                parts = torch.empty((self.world,) + tuple(send.shape), dtype=kv_local.dtype, device=self.dev)
                work = parallel.all_gather_into(parts, send, group=self.group, async_op=True)
                self._pending = (work, parts, kv_local)

    def finish(self):
        """-> ([B * T_full * Lt, 2C] in clip order, T_full * Lt)."""
        from . import graph
        seg = graph.active_segmenter()
        if seg is not None and self.overlap:
            full = self.buf if self.even else self.full

            def join():
                got, _ = self._finish()
                if got.data_ptr() != full.data_ptr():
                    full.copy_(got)
            seg.cut(join)
            return full, self.T_full * self.Lt
        if seg is not None:
            # The captured graph that follows reads the gathered rows at a FIXED address: the even path's reused buffer, or
            # (uneven shards) a buffer this object keeps.  The exchange itself is replayed eagerly between the two graphs.
            kv_local = self._deferred
            self._deferred = None
            rows, width = kv_local.shape
            if self.even:
                if self.buf is None or self.buf.dtype != kv_local.dtype or self.buf.shape[1] != width:
                    self.buf = torch.empty((self.world * rows, width), dtype=kv_local.dtype, device=self.dev)
                full = self.buf
            else:
                if self.full is None or self.full.dtype != kv_local.dtype or self.full.shape[1] != width:
                    self.full = torch.empty((self.B * self.T_full * self.Lt, width), dtype=kv_local.dtype, device=self.dev)
                full = self.full

            def exchange():
                self._start(kv_local)
                got, _ = self._finish()
                if got.data_ptr() != full.data_ptr():
                    full.copy_(got)
            seg.cut(exchange)
            return full, self.T_full * self.Lt
        return self._finish()

    def _finish(self):
        work, parts, kv_local = self._pending
        self._pending = None
        with torch.cuda.stream(self.comm):
            work.wait()
            if parts is None:
                full = self.buf
            else:
                full = torch.cat([parts[r, :, :f * self.Lt] for r, f in enumerate(self.frames)], dim=1)
                full = full.reshape(self.B * self.T_full * self.Lt, -1)
        main = torch.cuda.current_stream(self.dev)
        main.wait_stream(self.comm)
        kv_local.record_stream(self.comm)
        full.record_stream(main)
        return full, self.T_full * self.Lt


class _TrainStepFunction(torch.autograd.Function):
    """loss, pcd_moved = f(params): forward AND backward are computed in forward() by libm324 kernels; backward()
    only scales the stored gradients by d(loss)."""

    @staticmethod
    def forward(ctx, model, sample, *params):
        from . import training
        loss, out, G = training.forward_backward(model, sample)
        # a trainable parameter the step never touched (special_token_rest at T == 1) gets zeros, as from autograd:
        # DDP's reducer (find_unused_parameters=False, train.py:89) must see every bucket marked ready
        ctx.grads = [G.get(p) if G.get(p) is not None else torch.zeros_like(p) for p in params]
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, dloss, dout):
        grads = tuple(None if g is None else g * dloss for g in ctx.grads)
        ctx.grads = None
        return (None, None) + grads


class Motion_Latent_Model(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        model_cfg = _get(config, "model")
        train_cfg = _get(config, "training")
        self.feat_dim = _get(model_cfg, "feat_dim")
        vcfg = _get(model_cfg, "video_encoder")
        tcfg = _get(vcfg, "transformer")
        icfg = _get(vcfg, "image_tokenizer")
        use_qk_norm = _get(tcfg, "use_qk_norm", True)
        d_model, d_head = _get(tcfg, "d"), _get(tcfg, "d_head")
        if self.feat_dim != d_model:
            raise NotImplementedError("feat_dim must equal transformer.d (as in configs/dyscene.yaml)")
        if d_model % 6 != 0 or d_model % 4 != 0 or d_model > 1024:
            raise NotImplementedError("transformer.d must be a multiple of 12 and <= 1024")

        # _init_video_patchify (reference :346-370)
        self.image_size = _get(icfg, "image_size", 224)
        self.video_length = _get(train_cfg, "frames")
        self.patch_size = _get(icfg, "patch_size", 14)
        self.patch_length = _get(icfg, "patch_length", 1)
        self.embed_dim = d_model
        self.num_patches_h = self.num_patches_w = self.image_size // self.patch_size
        self.num_patches_t = self.video_length // self.patch_length
        self.latent_length, self.latent_size = self.num_patches_t, self.num_patches_h
        self.register_buffer("pos_embed", generate_pos_embed(self.latent_length, self.latent_size, self.latent_size,
                                                             self.embed_dim))
        self.drop_rate = float(_get(tcfg, "drop_rate", 0.1))
        self.pos_drop = nn.Dropout(p=self.drop_rate)

        self.point_embed = PointEmbed(dim=d_model)
        self.point_normal_rgb_proj = nn.Linear(d_model + 3 + 3, d_model)
        self.point_normal_rgb_proj.apply(init_weights)

        self.num_learnable_tokens = _get(model_cfg, "tokens")
        self.learnable_tokens = nn.Parameter(torch.randn(1, self.num_learnable_tokens, d_model))
        self.special_token_0 = nn.Parameter(torch.randn(1, 4, d_model))
        self.special_token_rest = nn.Parameter(torch.randn(1, 4, d_model))

        self.encoder_cross_attn = QK_Norm_CrossAttentionBlock(dim=d_model, head_dim=d_head, kv_dim=d_model,
                                                              use_qk_norm=use_qk_norm)
        self.points_transformer_blocks = nn.ModuleList([
            QK_Norm_TransformerBlock(d_model, d_head, use_qk_norm=use_qk_norm) for _ in range(_get(model_cfg, "pcd_layers"))])
        self.points_transformer_blocks.apply(init_weights)

        dino_kw = _get(model_cfg, "dino", None) or {}          # optional override (tests use a shallow ViT)
        self.image_encoder = DinoEncoder(patch_size=14, embed_dim=d_model, num_heads=d_model // 64,
                                         depth=_get(dino_kw, "depth", 12), pos_grid=_get(dino_kw, "pos_grid", 37))
        self.alternating_layers = _get(tcfg, "n_layer", 12)
        assert self.alternating_layers % 2 == 0, "Alternating layers should be even."
        self.global_transformer_blocks = nn.ModuleList([
            QK_Norm_TransformerBlock(d_model, d_head, use_qk_norm=use_qk_norm) for _ in range(self.alternating_layers // 2)])
        self.global_transformer_blocks.apply(init_weights)
        self.local_transformer_blocks = nn.ModuleList([
            QK_Norm_TransformerBlock(d_model, d_head, use_qk_norm=use_qk_norm) for _ in range(self.alternating_layers // 2)])
        self.local_transformer_blocks.apply(init_weights)

        self.transformer_input_layernorm = nn.LayerNorm(d_model, bias=False)
        self.decoder_cross_attn = QK_Norm_CrossAttentionBlock(dim=d_model, head_dim=d_head, kv_dim=d_model,
                                                              use_qk_norm=use_qk_norm)
        self.shared_mlp_output = nn.Sequential(nn.LayerNorm(self.feat_dim), nn.Linear(self.feat_dim, self.feat_dim),
                                               nn.GELU(), nn.Linear(self.feat_dim, 3))
        self.shared_mlp_output.apply(init_weights)
        self.loss_computer = MSELossComputer(self.config)
        self.auto_graph = AUTO_GRAPH            # inference: repeated shapes are served by hipGraph replay (forward())
        self.auto_graph_after = 2

    def train(self, mode=True):
        # the reference's override returns None (Pcd_motion.py:372-373); returning self keeps
        # `model.eval()` as a statement working and also the usual chaining
        super().train(mode)
        bump_generation()            # eval after training steps of an external optimizer: re-derive weight copies
        return self

    # ------------------------------------------------------------------------------------ stages
    def _point_features(self, P: Prepared, xyz, normal, rgb) -> torch.Tensor:
        """[P,3] x3 -> fp32 [P, C] = point_normal_rgb_proj(cat[point_embed(xyz), normal, rgb])
        (reference :456-459, :550-553)."""
        C = self.embed_dim
        n = xyz.shape[0]
        enc = ops.point_encode(xyz, P.dtype)                                   # [n, 64]
        kp = pad_k(C + 6)
        feat = torch.empty((n, kp), dtype=P.dtype, device=xyz.device)
        ops.gemm(enc, P.mat(self.point_embed.mlp.weight), feat, bias=P.vec(self.point_embed.mlp.bias))
        ops.point_concat(normal, rgb, feat, C)
        out = torch.empty((n, C), dtype=torch.float32, device=xyz.device)
        ops.gemm(feat, P.mat(self.point_normal_rgb_proj.weight), out, bias=P.vec(self.point_normal_rgb_proj.bias))
        return out

    def decoder_block(self, P: Prepared, Kd, Vd, pf: torch.Tensor, Q=None):
        """The decoder cross-attention block on one sample's mesh points (reference transformer.py:365-377 through
        Pcd_motion.py:556-561): pf fp32 [n, C] point features = the queries, Kd / Vd the T frames' latent keys / values
        (project_kv).  Returns (x [T * n, C] stream after attention + MLP, its LNFold or None).  Q: an already projected
        query set (the graph hoists it onto the shape-encoder branch)."""
        dec = self.decoder_cross_attn
        n = pf.shape[0]
        if Q is None:
            Q = dec.project_q(P, pf, 1, n)
        return dec.attend(P, Q, Kd, Vd, pf, n, shared_q=True, bf16_stream=BF16_DECODER_STREAM, want_fold=True)

    def decoder_block_flops(self, B: int, T: int, N: int) -> float:
        """Reference FLOPs of the block (to_q counted once per frame, as the reference computes it)."""
        C, K = self.embed_dim, self.num_learnable_tokens
        lin = lambda mm, i, o: 2.0 * mm * i * o
        return B * T * (2 * lin(N, C, C) + 2 * lin(K, C, C) + 2 * lin(N, C, 4 * C) + 4.0 * (C // 64) * N * K * 64)

    def decoder_block_flops_executed(self, B: int, T: int, N: int) -> float:
        """FLOPs the block actually issues: the q projection of the mesh points runs once per sample, not once per frame
        (Pcd_motion.py:550-553 recomputes it inside the frame loop; same values).  SURVEY 8(d): hoisting is a speed-up, not
        utilisation -- bench.py reports the matrix-pipe rate on THIS count next to the north-star figure on the reference count."""
        C = self.embed_dim
        return self.decoder_block_flops(B, T, N) - B * (T - 1) * 2.0 * N * C * C

    def _video_pos(self, P: Prepared, T: int) -> torch.Tensor:
        g = self.latent_size
        if T == self.latent_length:
            return P.f32(self.pos_embed).reshape(-1, self.embed_dim)
        return P.derived(f"pos_T{T}", (self.pos_embed,), lambda: resize_pos_embed(
            self.pos_embed.detach().float(), (self.latent_length, g, g), (T, g, g)).reshape(-1, self.embed_dim)
            .contiguous().to(P.device))

    def _f32c(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(torch.float32).contiguous()

    def forward(self, sample: Dict[str, torch.Tensor]):
        if self.training and torch.is_grad_enabled() and "point_clouds" in sample \
                and any(p.requires_grad for p in self.parameters()):
            return self._forward_train(sample)
        out = self._forward_auto_graph(sample)
        return out if out is not None else self._forward(sample, None)

    def _forward_auto_graph(self, sample: Dict[str, torch.Tensor]):
        """Inference callers (scripts/inference_with_video_mesh.py:167,210; the sliding-window driver) call ``model(sample)``
        in a loop with one set of shapes.  An eager forward leaves the GPU idle between its ~350 launches (22 % of the
        clip, tools/rocpd_gaps.py) and cannot use the graph-only overlaps; so from the THIRD call with the same shapes on
        (``auto_graph_after`` eager ones first: single shots and tests stay eager) the call is served by a private
        GraphedForward, and the caller gets a copy of the replay's output -- same values bit for bit, same ownership as
        an eager result.  ``model.auto_graph = False`` (or M324_AUTO_GRAPH=0) turns it off."""
        if (not self.auto_graph or self.training or torch.is_grad_enabled() or getattr(self, "_capture", None) is not None
                or getattr(self, "_ag_busy", False)):
            return None
        from . import timing
        if timing.active() or any(isinstance(v, torch.Tensor) and v.device.type != "cuda" for v in sample.values()) \
                or torch.cuda.is_current_stream_capturing():
            return None
        from .graph import GraphedForward, shape_key
        key = (compute_dtype(),) + shape_key(sample)
        seen = self.__dict__.setdefault("_ag_seen", {})
        if key not in seen and len(seen) >= 8:               # a caller that keeps changing shapes: forget the oldest
            del seen[next(iter(seen))]
        seen[key] = seen.get(key, 0) + 1
        if seen[key] <= self.auto_graph_after:
            return None
        ag = self.__dict__.get("_ag")
        if ag is None:
            # at most three shape sets stay captured (each holds a clip's activations and its static inputs in a private
            # pool; the long-video driver alone uses two: the first window and the windows behind it); "thread_local": a DataLoader thread or another stream may keep calling into HIP during the capture
            ag = GraphedForward(self, warmup=1, weak=True, max_graphs=3, capture_error_mode="thread_local")
            self.__dict__["_ag"] = ag                       # not a submodule: plain attribute
        fresh = ag._key(sample) not in ag._graphs
        if fresh:
            n = self.__dict__.get("_ag_captures", 0) + 1
            self.__dict__["_ag_captures"] = n
            if n > 6:                                       # shapes cycle faster than graphs pay off: stay eager
                self.auto_graph = False
                self._drop_auto_graph()
                return None
        self.__dict__["_ag_busy"] = True                    # the capture's own warm-up / capture forwards stay eager
        try:
            res = ag(sample)
        except Exception as exc:                            # capture invalidated, out of memory in the private pool, ...:
            # the eager path still serves the call -- but say so once: a kernel fault during capture must not turn into a
            # silent 20 % slowdown (eager passes leave the GPU idle between launches)
            import warnings
            warnings.warn(f"motion324_amd: hipGraph capture of the inference forward failed ({type(exc).__name__}: {exc}); "
                          "this model continues with eager launches (model.auto_graph is now False)", RuntimeWarning, stacklevel=3)
            self.auto_graph = False
            self._drop_auto_graph()
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            return None
        finally:
            self.__dict__["_ag_busy"] = False
        out = edict(input_data=sample, pcd_moved=res.pcd_moved.clone())
        if "reuse" in res:
            out.reuse = edict({k: v.clone() for k, v in res.reuse.items()})
        if "loss_metrics" in res:
            lm = edict()
            for k, v in res.loss_metrics.items():
                lm[k] = v.clone() if isinstance(v, torch.Tensor) else v
            out.loss_metrics = lm
        return out

    def _drop_auto_graph(self) -> None:
        for k in ("_ag", "_ag_seen", "_ag_captures"):
            self.__dict__.pop(k, None)

    def __getstate__(self):
        # captured graphs are process-local device objects: copy.deepcopy / pickle / torch.save(model) go without them
        state = self.__dict__.copy()
        for k in ("_ag", "_ag_seen", "_ag_busy", "_ag_captures", "_capture"):
            state.pop(k, None)
        return state

    def _forward_train(self, sample: Dict[str, torch.Tensor]):
        """Training forward (the reference's train.py:150-166 calls model(batch) then loss.backward()).  The HIP
        forward+backward runs eagerly (motion324_amd.training.forward_backward); `loss` is returned through an
        autograd.Function whose backward hands the already-computed gradients (times the incoming scalar) to autograd,
        so `.grad`, GradScaler, clip_grad_norm_ and DDP's reducer hooks all behave as with the reference model."""
        # the caller's optimizer owns the parameters here, and torch.optim.AdamW(fused=True) (training_utils.py:52)
        # updates them without bumping tensor._version: kernel-ready copies cannot be trusted across steps
        bump_generation()
        params = [p for p in self.parameters() if p.requires_grad]
        loss, out = _TrainStepFunction.apply(self, sample, *params)
        lm = edict()
        lm.loss = loss
        lm.xyz_loss = loss.detach() / max(float(self.loss_computer._weight), 1e-30)
        return edict(input_data=sample, pcd_moved=out, loss_metrics=lm)

    def forward_frame_parallel(self, sample: Dict[str, torch.Tensor], group=None, *, local_frames: bool = False,
                               total_frames: Optional[int] = None):
        """One long clip, frames sharded over the ranks of `group` with EXACT single-GPU semantics
        (BASELINE config 5; SURVEY.md 8(e) third row).  Every rank passes the same sample (full `rgb_video`
        [B,T,H,W,3]); rank r encodes / decodes frames partition(T, world, r) only.  Per-frame stages (DINO, local
        blocks, decoder) need no communication; each GLOBAL block all-gathers its token-major k|v projection
        (RCCL over xGMI; 2 x T*324*768 bf16 = 255 MB assembled per block at T = 256) so local queries attend to the
        whole clip; `pcd_moved` is all-gathered at the end and returned complete on every rank.

        local_frames=True: `rgb_video` holds ONLY this rank's frames, [B, len(partition(total_frames, world, rank)), H, W, 3]
        (a loader that decodes its shard: at T = 256 the full fp32 clip is 805 MB of host-to-device traffic per rank, 7/8 of
        it for frames the rank never touches); total_frames = the clip's length."""
        from . import parallel
        rank, world = parallel.world_info(group)
        if local_frames and (total_frames is None or total_frames < 1):
            raise M324Error("forward_frame_parallel(local_frames=True) needs total_frames")
        return self._forward(sample, (rank, world, group, int(total_frames) if local_frames else None))

    def _forward(self, sample: Dict[str, torch.Tensor], shard):
        ref_pcd = sample["ref_pcd"]
        dev = ref_pcd.device
        if dev.type != "cuda":
            raise M324Error("motion324_amd.Motion_Latent_Model runs only on a HIP device (model.to('cuda'), inputs on "
                            "'cuda'); there is no CPU fallback on this path")
        P = Prepared.for_module(self, dev, compute_dtype())
        cap = getattr(self, "_capture", None)      # tests: dict that receives clones of stage activations
        B, N, _ = ref_pcd.shape
        C, K = self.embed_dim, self.num_learnable_tokens
        S = sample["ref_shape_pcd"].shape[1]

        # A. shape encoder (reference :456-464).  Its ~60 launches work on B*64 latent rows (latency-bound, a few
        # dozen workgroups each) and do not depend on the video, so they run on a second HIP stream underneath the
        # image encoder's chip-filling GEMMs and join before the token assembly (fork/join is captured into the graph).
        main_stream = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if OVERLAP_SHAPE_ENCODER and cap is None else None
        if side is not None:
            side.wait_stream(main_stream)
        # The long-video driver (inference.run_model_inference) runs many windows of ONE video over ONE mesh: stage A's result and
        # the anchor frame's image tokens are the same in every window (bit for bit: every kernel works row by row), so a window may
        # hand them in instead of recomputing them -- `m324_mesh_tokens` [B*K, C] and `m324_anchor_tokens` [1 + g*g, C] (B = 1;
        # `rgb_video` then holds the frames BEHIND the anchor) -- and a window asked to (`m324_keep_reuse`) returns them as `reuse`.
        mesh_in, anchor_in = sample.get("m324_mesh_tokens"), sample.get("m324_anchor_tokens")
        keep_reuse = bool(sample.get("m324_keep_reuse", False))
        if (mesh_in is not None or anchor_in is not None) and (shard is not None or cap is not None):
            raise M324Error("m324_mesh_tokens / m324_anchor_tokens: single-GPU inference windows only")
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            if mesh_in is not None:
                mesh = self._f32c(mesh_in).reshape(B * K, C)
            else:
                pts = self._point_features(P, self._f32c(sample["ref_shape_pcd"]).reshape(-1, 3),
                                           self._f32c(sample["ref_shape_normals"]), self._f32c(sample["ref_shape_rgbs"]))
                # fp32 [B*K, C], expanded once per weight version (the block reads it as the residual and writes a new stream)
                query = P.derived(f"latent_tokens_x{B}", (self.learnable_tokens,),
                                  lambda: self.learnable_tokens.detach().to(device=P.device, dtype=torch.float32).reshape(K, C).repeat(B, 1))
                mesh = self.encoder_cross_attn.run(P, query, pts, B, K, S)
                if cap is not None:
                    cap["shape_point_feat"], cap["encoder_out"] = pts.clone(), mesh.clone()
                for blk in self.points_transformer_blocks:
                    blk.run(P, mesh, B, K)
                if cap is not None:
                    cap["mesh_feat"] = mesh.clone()
            # the decoder's point features and q projection depend on the mesh points only: under graph capture they
            # ride on this branch too (seven small launches, ~40 us, off the critical path between trunk and decoder)
            hoisted = None
            if (HOIST_DECODER_Q and side is not None and torch.cuda.is_current_stream_capturing()
                    and DECODE_ROWS // (sample["rgb_video"].shape[1] + (anchor_in is not None)) >= N):
                pcd_h, nrm_h, rgb_h = (self._f32c(sample[k]) for k in ("ref_pcd", "ref_normal", "ref_rgb"))
                hoisted = []
                for b in range(B):
                    pf_b = self._point_features(P, pcd_h[b], nrm_h[b].contiguous(), rgb_h[b].contiguous())
                    hoisted.append((pf_b, self.decoder_cross_attn.project_q(P, pf_b, 1, N)))

        # B. image encoder (reference :466-475): resize + normalise + patchify + ViT, frozen
        video = sample["rgb_video"]
        T_full = video.shape[1]
        t0, kv_gather = 0, None
        if shard is not None:
            from . import parallel
            rank, world, group, t_local_of = shard
            if t_local_of is not None:
                T_full = t_local_of
            mine = parallel.partition(T_full, world, rank)
            if len(mine) == 0:
                raise M324Error(f"frame-parallel forward: {T_full} frames cannot feed {world} ranks")
            t0 = mine.start
            if t_local_of is None:
                video = video[:, mine.start:mine.stop]
            elif video.shape[1] != len(mine):
                raise M324Error(f"frame-parallel forward: rank {rank} of {world} owns {len(mine)} of {T_full} frames, rgb_video has {video.shape[1]}")
        # byte frames (what a decoder delivers) stay bytes: m324_patchify_u8 converts every tap as v / 255
        video = video.detach().contiguous() if video.dtype == torch.uint8 else self._f32c(video)
        _, T, Hin, Win, _ = video.shape
        Pn = self.num_patches_h * self.num_patches_w
        if anchor_in is not None:
            if B != 1 or tuple(anchor_in.shape) != (1 + Pn, C):
                raise M324Error(f"m324_anchor_tokens: expected [{1 + Pn}, {C}] with one sample per call, got {tuple(anchor_in.shape)} (B = {B})")
            dino_x = torch.empty(((T + 1) * (1 + Pn), C), dtype=torch.float32, device=dev)
            ops.cast(self._f32c(anchor_in), torch.float32, out=dino_x[:1 + Pn])      # m324_cast fp32 -> fp32: a row copy
            self.image_encoder.run(P, video.reshape(T, Hin, Win, 3), out=dino_x[1 + Pn:])
            T += 1
            T_full = T
        else:
            dino_x = self.image_encoder.run(P, video.reshape(B * T, Hin, Win, 3))

        if side is not None:                      # join: the assembly reads the latent tokens
            main_stream.wait_stream(side)
            mesh.record_stream(main_stream)

        # C. DINO final norm + pos-embed + token assembly + input LN in one pass (reference :477-510)
        enc = self.image_encoder.model
        # pos_drop acts whenever the module is in training mode (reference :490), also under no_grad
        drop_p = self.drop_rate if self.training else 0.0
        drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if drop_p > 0.0 else 0
        pos = self._video_pos(P, T_full)[t0 * Pn:(t0 + T) * Pn]
        sp_rest = P.f32(self.special_token_rest).reshape(4, C)
        sp_first = P.f32(self.special_token_0).reshape(4, C) if t0 == 0 else sp_rest     # only clip frame 0 is special
        tok = ops.assemble_tokens(dino_x, P.vec(enc.norm.weight), P.vec(enc.norm.bias), DINO_EPS, pos, sp_first, sp_rest,
                                  mesh, P.vec(self.transformer_input_layernorm.weight),
                                  self.transformer_input_layernorm.eps, B, T, K, Pn, drop_p, drop_seed)
        Lt = 4 + K + Pn
        if cap is not None:
            dn = torch.empty((B * T * Pn, C), dtype=torch.float32, device=dev)
            ops.layernorm(dino_x, P.vec(enc.norm.weight), P.vec(enc.norm.bias), DINO_EPS, dn, row_map=(Pn, Pn + 1, 1))
            cap["dino_tokens"], cap["trunk_in"] = dn, tok.clone()

        # D. alternating global / local trunk (reference :394-409)
        if shard is not None and parallel.collectives_on(shard[1]):
            rank, world, group = shard[:3]
            kv_gather = _KVGather(B, T, Lt, parallel.counts(T_full, world), group, dev, rank=rank)

        # LayerNorm fold (transformer.LNFold): the statistics of the stream travel from GEMM epilogue to GEMM epilogue
        fold = None
        if LNFold.usable(P, tok.shape[0], C):
            fold = LNFold(tok).from_stream(tok, self.global_transformer_blocks[0].norm1.eps)
        n_pairs = len(self.global_transformer_blocks)
        for i, (gblk, lblk) in enumerate(zip(self.global_transformer_blocks, self.local_transformer_blocks)):
            gblk.run(P, tok, B, T * Lt, kv_gather=kv_gather, fold=fold)
            lblk.run(P, tok, B * T, Lt, fold=fold, feed_next=i + 1 < n_pairs)     # the decoder gathers its own rows
            if cap is not None and "trunk_block0" not in cap:
                cap["trunk_block0"] = tok.clone()
        if cap is not None:
            cap["trunk_out"] = tok.clone()

        # E+F. decoder (reference :520-579): the mesh points are projected once per sample and attend to
        # each frame's K latent tokens; rows = (frame, point).
        dec = self.decoder_cross_attn
        out = torch.empty((B, T, N, 3), dtype=torch.float32, device=dev)
        head_ln, head_fc1, head_fc2 = self.shared_mlp_output[0], self.shared_mlp_output[1], self.shared_mlp_output[3]
        w3, b3 = P.f32(head_fc2.weight), P.vec(head_fc2.bias)
        # reference FLOPs of the decoder cross-attention block (to_q counted once per frame as the
        # reference computes it) -- attached to the stage span bench.py reports the 40 % MFMA target on
        nchunk = max(1, min(N, DECODE_ROWS // T))
        pcd, nrm, rgb = (self._f32c(sample[k]) for k in ("ref_pcd", "ref_normal", "ref_rgb"))
        paired = None
        if hoisted is None and B == 1 and nchunk == N:
            # one sample in one pass and the q projection inside the block: its LayerNorm / projection share their launches
            # with the k|v side's (transformer.project_q_kv)
            pf0 = self._point_features(P, pcd[0], nrm[0].contiguous(), rgb[0].contiguous())
            with span("stage:decoder_cross_attn_block", self.decoder_block_flops(B, T, N)):
                Q0, Kd, Vd = dec.project_q_kv(P, pf0, N, tok, B * T, K, row_map=(K, Lt, 4))
            paired = (pf0, Q0)
        else:
            with span("stage:decoder_cross_attn_block", self.decoder_block_flops(B, T, N)):
                Kd, Vd = dec.project_kv(P, tok, B * T, K, row_map=(K, Lt, 4))       # latent tokens 4..4+K of every frame
        for b in range(B):
            for n0 in range(0, N, nchunk):
                n1 = min(N, n0 + nchunk)
                if hoisted is not None:
                    pf, Q = hoisted[b]
                elif paired is not None:
                    pf, Q = paired
                else:
                    pf = self._point_features(P, pcd[b, n0:n1], nrm[b, n0:n1].contiguous(), rgb[b, n0:n1].contiguous())
                with span("stage:decoder_cross_attn_block", 0.0):
                    x, fold_d = self.decoder_block(P, Kd[b * T:(b + 1) * T], Vd[b * T:(b + 1) * T], pf,
                                                   None if (hoisted is None and paired is None) else Q)
                if cap is not None and n0 == 0 and n1 == N:
                    cap.setdefault("decoder_out_t0", []).append(x[:N].clone())
                if fold_d is not None:
                    # the head's LayerNorm rides in its first GEMM: statistics left by the MLP's last epilogue
                    hw, hcs, hb = P.folded(head_ln.weight, head_ln.bias, head_fc1.weight, head_fc1.bias)
                    h, lnk = fold_d.xb, dict(ln=fold_d.ln(head_ln.eps, hcs))
                else:
                    h = torch.empty(x.shape, dtype=P.dtype, device=dev)
                    ops.layernorm(x, P.vec(head_ln.weight), P.vec(head_ln.bias), head_ln.eps, h)
                    hw, hb, lnk = P.mat(head_fc1.weight), P.vec(head_fc1.bias), {}
                whole = n0 == 0 and n1 == N
                o = out[b] if whole else torch.empty((T, n1 - n0, 3), dtype=torch.float32, device=dev)
                if FUSE_HEAD_N3 and P.dtype == torch.bfloat16 and C % 256 == 0 and not torch.is_grad_enabled():
                    # Linear -> GELU -> Linear(C -> 3) without the [rows, C] intermediate: the first GEMM's epilogue contracts
                    # its GELU output with the 3 x C weight and leaves C / 64 partial sums per row (M324_AUX_N3)
                    part = torch.empty((C // 64, x.shape[0], 3), dtype=torch.float32, device=dev)
                    ops.gemm(h, hw, None, bias=hb, act=ACT_GELU, n3=(w3, part), **lnk)
                    ops.n3_finish(part, b3, o)
                else:
                    h2 = torch.empty(x.shape, dtype=P.dtype, device=dev)
                    ops.gemm(h, hw, h2, bias=hb, act=ACT_GELU, **lnk)
                    ops.linear_n3(h2, w3, b3, o)
                if not whole:
                    out[b, :, n0:n1] = o
        if cap is not None and "decoder_out_t0" in cap:
            cap["decoder_out_t0"] = torch.stack(cap["decoder_out_t0"], dim=0)

        if shard is not None and parallel.collectives_on(shard[1]):
            rank, world, group = shard[:3]
            frames = parallel.counts(T_full, world)
            buf = torch.zeros((max(frames), B, N, 3), dtype=torch.float32, device=dev)
            buf[:T] = out.transpose(0, 1)
            parts = torch.empty((world,) + tuple(buf.shape), dtype=torch.float32, device=dev)
            from . import graph
            seg = graph.active_segmenter()
            if seg is not None:                     # segmented capture: the exchange is replayed between two graphs
                seg.cut(lambda: parallel.all_gather_into(parts, buf, group=group))
            else:
                parallel.all_gather_into(parts, buf, group=group)
            if all(f == frames[0] for f in frames):
                out = parts.reshape(T_full, B, N, 3).transpose(0, 1).contiguous()
            else:
                out = torch.cat([parts[r, :f] for r, f in enumerate(frames)], dim=0).transpose(0, 1).contiguous()
        result = edict(input_data=sample, pcd_moved=out)
        if keep_reuse:
            Ld = 1 + Pn
            anchor = dino_x[:Ld] if B == 1 else dino_x.view(B, T, Ld, C)[:, 0].reshape(B * Ld, C).contiguous()
            result.reuse = edict(mesh_tokens=mesh, anchor_tokens=anchor)
        if "point_clouds" in sample:                                           # reference :582-592
            m = self.loss_computer(out, sample["point_clouds"].to(dev))
            lm = edict()
            lm.loss = m.loss
            lm.xyz_loss = m.coord_mse_loss
            result.loss_metrics = lm
        return result
