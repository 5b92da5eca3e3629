#!/usr/bin/env python3
"""Generate golden vectors by running THE REFERENCE ITSELF (imported from /root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box); the arrays it
writes (tests/golden/*.npz) are committed.  Weights and inputs are regenerated from seeds by
motion324_amd.synth on any machine, so the fixtures hold only expected outputs (+ small slices
of intermediate activations captured by forward hooks on the reference's own modules).

Shims installed before importing the reference (none is shipped; each replaces a package that is
absent offline):
  1. ``easydict``         -> 10-line dict with attribute access   (Pcd_motion.py:10, loss.py:4)
  2. ``xformers.ops``     -> memory_efficient_attention = softmax(q k^T/sqrt(d)) v on [B,L,H,D]
                             (transformer.py:8-11,134-139,209-214; xformers==0.0.28 is CUDA-only)
  3. ``torch.hub.load``   -> the reference's OWN in-tree DINOv2 restatement
                             model/image_encoder/dino/model_dino.py (embeddings/encoder/layernorm),
                             exposing patch_size / embed_dim / forward_features like the hub model
                             (dinov2.py:44,99-103).  It needs one more stub:
  4. ``transformers.pytorch_utils.find_pruneable_heads_and_indices`` (removed in transformers 5)
                             -> never called on this path.

Usage:  python tests/golden/make_golden.py [tiny c1 c2]
"""
from __future__ import annotations

import math
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)

from motion324_amd import synth  # noqa: E402
from oracle import ref_forward as oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ shims
class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _install_shims(dino_cfg):
    ed = types.ModuleType("easydict")
    ed.EasyDict = _EasyDict
    sys.modules["easydict"] = ed

    xf = types.ModuleType("xformers")
    xo = types.ModuleType("xformers.ops")

    def memory_efficient_attention(q, k, v, attn_bias=None, p=0.0, op=None):
        assert attn_bias is None and p == 0.0
        o = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
        return o.transpose(1, 2)

    xo.memory_efficient_attention = memory_efficient_attention
    xo.fmha = types.SimpleNamespace(flash=types.SimpleNamespace(FwOp=None, BwOp=None))
    xf.ops = xo
    sys.modules["xformers"] = xf
    sys.modules["xformers.ops"] = xo

    import transformers.pytorch_utils as pu
    if not hasattr(pu, "find_pruneable_heads_and_indices"):
        pu.find_pruneable_heads_and_indices = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("unused"))

    sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    from model.image_encoder.dino import model_dino as md
    from transformers import Dinov2Config

    class HubLikeDino(torch.nn.Module):
        """The in-tree HF-port DINOv2 (model_dino.py) behind the hub model's surface."""

        def __init__(self):
            super().__init__()
            cfg = Dinov2Config(**dino_cfg)
            self.embeddings = md.Dinov2Embeddings(cfg)
            self.encoder = md.Dinov2Encoder(cfg)
            self.layernorm = torch.nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)   # model_dino.py:587
            self.patch_size = cfg.patch_size
            self.embed_dim = cfg.hidden_size

        def forward_features(self, x):
            h = self.embeddings(x)
            h = self.encoder(h, head_mask=None, output_attentions=False, output_hidden_states=False,
                             return_dict=False)[0]
            h = self.layernorm(h)                                                          # model_dino.py:645
            return {"x_norm_patchtokens": h[:, 1:], "x_norm_clstoken": h[:, 0]}

    torch.hub.load = lambda *a, **k: HubLikeDino()


def hub_to_hfport(sd, pre="image_encoder.model"):
    """hub DINOv2 key names -> the in-tree HF-port module names (qkv split into query/key/value)."""
    out = {}
    d = sd[f"{pre}.cls_token"].shape[-1]
    out["embeddings.cls_token"] = sd[f"{pre}.cls_token"]
    out["embeddings.mask_token"] = sd[f"{pre}.mask_token"]
    out["embeddings.position_embeddings"] = sd[f"{pre}.pos_embed"]
    out["embeddings.patch_embeddings.projection.weight"] = sd[f"{pre}.patch_embed.proj.weight"]
    out["embeddings.patch_embeddings.projection.bias"] = sd[f"{pre}.patch_embed.proj.bias"]
    i = 0
    while f"{pre}.blocks.{i}.norm1.weight" in sd:
        s, t = f"{pre}.blocks.{i}", f"encoder.layer.{i}"
        for n in ("norm1", "norm2"):
            out[f"{t}.{n}.weight"] = sd[f"{s}.{n}.weight"]
            out[f"{t}.{n}.bias"] = sd[f"{s}.{n}.bias"]
        w, b = sd[f"{s}.attn.qkv.weight"], sd[f"{s}.attn.qkv.bias"]
        for j, n in enumerate(("query", "key", "value")):
            out[f"{t}.attention.attention.{n}.weight"] = w[j * d:(j + 1) * d]
            out[f"{t}.attention.attention.{n}.bias"] = b[j * d:(j + 1) * d]
        out[f"{t}.attention.output.dense.weight"] = sd[f"{s}.attn.proj.weight"]
        out[f"{t}.attention.output.dense.bias"] = sd[f"{s}.attn.proj.bias"]
        out[f"{t}.layer_scale1.lambda1"] = sd[f"{s}.ls1.gamma"]
        out[f"{t}.layer_scale2.lambda1"] = sd[f"{s}.ls2.gamma"]
        for n in ("fc1", "fc2"):
            out[f"{t}.mlp.{n}.weight"] = sd[f"{s}.mlp.{n}.weight"]
            out[f"{t}.mlp.{n}.bias"] = sd[f"{s}.mlp.{n}.bias"]
        i += 1
    out["layernorm.weight"] = sd[f"{pre}.norm.weight"]
    out["layernorm.bias"] = sd[f"{pre}.norm.bias"]
    return out


# ------------------------------------------------------------------ cases
CASES = {
    # name: dims kwargs, (B, T, N, S, HW), weight seed, input seed
    "tiny": dict(dims=dict(d=192, d_head=64, tokens=8, pcd_layers=1, n_layer=2, frames=3, dino_depth=2),
                 shape=(2, 3, 40, 100, 64)),
    "tiny_resize": dict(dims=dict(d=192, d_head=64, tokens=8, pcd_layers=1, n_layer=2, frames=5, dino_depth=2),
                        shape=(1, 2, 33, 70, 96)),
    "c1": dict(dims=dict(frames=12), shape=(1, 4, 512, 4096, 256)),
    "c2": dict(dims=dict(frames=32), shape=(1, 32, 2048, 4096, 512)),
    # BASELINE configs[4]: the 256-frame clip (82 944 trunk tokens; ~10 min per forward on 8 cores).  Only a sample of the
    # mesh points is stored (all frames), so that the fixture stays small.
    "c5": dict(dims=dict(frames=256), shape=(1, 256, 2048, 4096, 512), pcd_points=96),
}
STAGE_ROWS = 16  # rows kept per stage tensor (first dims flattened) for the full-size cases


def run_case(name):
    spec = CASES[name]
    dm = synth.Dims(**spec["dims"])
    B, T, N, S, HW = spec["shape"]
    dino_cfg = dict(hidden_size=dm.d, num_hidden_layers=dm.dino_depth, num_attention_heads=dm.d // dm.d_head,
                    image_size=dm.dino_pos_grid * dm.patch_size, patch_size=dm.patch_size,
                    layerscale_value=1.0, mlp_ratio=4, qkv_bias=True, layer_norm_eps=1e-6,
                    hidden_act="gelu", use_swiglu_ffn=False, hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0, drop_path_rate=0.0)
    _install_shims(dino_cfg)
    for m in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
        if "image_encoder.dino" not in m:
            del sys.modules[m]
    from model.Pcd_motion import Motion_Latent_Model

    cfg = _EasyDict(synth.make_config(frames=dm.frames, d=dm.d, d_head=dm.d_head, tokens=dm.tokens,
                                      pcd_layers=dm.pcd_layers, n_layer=dm.n_layer, drop_rate=0.0))
    torch.manual_seed(0)
    t0 = time.time()
    model = Motion_Latent_Model(cfg)
    sd_np = synth.synth_state_dict(dm, seed=0)
    sd = oracle.to_torch(sd_np)
    non_dino = {k: v for k, v in sd.items() if not k.startswith("image_encoder.")}
    missing, unexpected = model.load_state_dict(non_dino, strict=False)
    missing = [k for k in missing if not k.startswith("image_encoder.") and k not in ("pos_embed", "point_embed.basis")]
    assert not missing and not unexpected, (missing, unexpected)
    r = model.image_encoder.model.load_state_dict(hub_to_hfport(sd), strict=True)
    model.eval()
    print(f"[{name}] reference built in {time.time() - t0:.1f}s; params "
          f"{sum(p.numel() for p in model.parameters()) / 1e6:.1f} M")

    sample_np = synth.synth_inputs(B, T, N, S, HW, seed=1, with_target=True)
    sample = oracle.to_torch(sample_np)

    # hooks on the reference's own modules -> stage activations
    stages = {}
    hooks = []

    def grab(key, first_only=False):
        def fn(mod, inp, out):
            if first_only and key in stages:
                return
            stages[key] = out.detach().clone()
        return fn
    hooks.append(model.point_normal_rgb_proj.register_forward_hook(grab("shape_point_feat", True)))
    hooks.append(model.encoder_cross_attn.register_forward_hook(grab("encoder_out")))
    hooks.append(model.points_transformer_blocks[-1].register_forward_hook(grab("mesh_feat")))
    hooks.append(model.image_encoder.register_forward_hook(grab("dino_tokens")))
    hooks.append(model.transformer_input_layernorm.register_forward_hook(grab("trunk_in")))
    hooks.append(model.local_transformer_blocks[0].register_forward_hook(grab("trunk_block0")))
    hooks.append(model.local_transformer_blocks[-1].register_forward_hook(grab("trunk_out")))
    hooks.append(model.decoder_cross_attn.register_forward_hook(grab("decoder_out_t0", True)))

    t0 = time.time()
    with torch.no_grad():
        ret = model(dict(sample))
    t_ref = time.time() - t0
    for h in hooks:
        h.remove()
    ref_out = ret["pcd_moved"].float()
    ref_loss = float(ret["loss_metrics"]["loss"])
    assert isinstance(ret, dict) and ref_out.shape == (B, T, N, 3)

    # the oracle on the same inputs
    ostages = {}
    t0 = time.time()
    with torch.no_grad():
        ores = oracle.forward(sd, sample, frames=dm.frames, d_head=dm.d_head, stages=ostages)
    t_or = time.time() - t0

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-30))
    L = 4 + dm.tokens + dm.grid * dm.grid
    stages["trunk_block0"] = stages["trunk_block0"].reshape(B, T, L, dm.d)
    stages["trunk_out"] = stages["trunk_out"].reshape(B, T, L, dm.d)
    report = {"pcd_moved": rel(ores["pcd_moved"], ref_out)}
    for k in stages:
        report[k] = rel(ostages[k].reshape(stages[k].shape), stages[k])
    report["loss"] = abs(float(ores["loss"]) - ref_loss) / abs(ref_loss)
    print(f"[{name}] reference {t_ref:.2f}s  oracle {t_or:.2f}s  oracle-vs-reference rel err:")
    for k, v in report.items():
        print(f"    {k:18s} {v:.3e}")
    worst = max(report.values())
    assert worst < 2e-5, f"oracle disagrees with the reference: {report}"

    if spec.get("pcd_points"):
        pts = np.unique(np.linspace(0, N - 1, spec["pcd_points"]).astype(np.int64))
        ref_keep = ref_out[:, :, torch.from_numpy(pts)]
    else:
        pts, ref_keep = None, ref_out
    save = {"pcd_moved": ref_keep.numpy(), "loss": np.float32(ref_loss),
            "meta_shape": np.array([B, T, N, S, HW], dtype=np.int64),
            "oracle_vs_reference_max_rel": np.float64(worst)}
    if pts is not None:
        save["pcd_points"] = pts
    nrows = 64 if name.startswith("tiny") else STAGE_ROWS
    for k, v in stages.items():
        v2 = v.reshape(-1, v.shape[-1])
        idx = np.unique(np.linspace(0, v2.shape[0] - 1, min(nrows, v2.shape[0])).astype(np.int64))
        save["stage_" + k] = v2[idx].numpy()
        save["rows_" + k] = idx
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **save)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run_case(n)
