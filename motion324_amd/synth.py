"""Deterministic synthetic weights and inputs for the Motion324 hot path.

The reference ships no weights offline (torch.hub DINOv2 + a HuggingFace checkpoint, both
unreachable here), so every parity test and the benchmark run on random-init weights and
random frames / point clouds, as BASELINE.json's north_star prescribes.  The generator is a
counter-based SplitMix64 stream per tensor (seeded by the tensor's state-dict key), so any
machine regenerates bit-identical tensors without shipping 1 GB of weights.

Shapes and names follow the reference's state dict:
  model/Pcd_motion.py:269-344 (Motion_Latent_Model.__init__), model/transformer.py:84-423,
  and hub DINOv2 ViT-B/14 names under ``image_encoder.model.*``.
Initial scales mirror model/transformer.py:15-25 (Linear ~ N(0, 0.02^2)) and
model/Pcd_motion.py:288-292 (tokens ~ N(0,1)); norm weights / biases are *perturbed* away
from 1 / 0 (``perturb=True``) so that a kernel that drops a bias or a norm weight fails parity.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser on a uint64 array (wraps modulo 2^64)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _stream_seed(seed: int, key: str) -> np.uint64:
    mixed = _splitmix64(np.array([(_fnv1a64(key) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF],
                                 dtype=np.uint64))
    return mixed[0]


def uniform(seed: int, key: str, shape, lo: float = 0.0, hi: float = 1.0, offset: int = 0) -> np.ndarray:
    """U[lo, hi) float32 tensor, element i = f(seed, key, offset + i): `offset` cuts a window out of a longer stream (a rank's
    frames of a clip) without generating the rest."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    s = _stream_seed(seed, key)
    step = 1 << 22
    for a in range(0, n, step):
        idx = np.arange(offset + a, offset + min(n, a + step), dtype=np.uint64)
        with np.errstate(over="ignore"):
            bits = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + s)
        u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        out[a:a + step] = (lo + (hi - lo) * u).astype(np.float32)
    return out.reshape(shape)


def dropout_keep(drop_seed: int, n: int, p: float) -> np.ndarray:
    """bool[n]: the keep-mask m324_assemble_tokens applies for (drop_p=p, drop_seed) -- element i of the
    reference's x[B, T*P, C] (flattened) survives pos_drop iff its top 24 SplitMix64 bits >= p * 2^24
    (include/m324.h).  Host restatement for parity tests; the product path generates the mask on the GPU."""
    thr = np.uint64(int(np.float32(p) * np.float32(16777216.0)))
    out = np.empty(n, dtype=bool)
    s = np.uint64(drop_seed & 0xFFFFFFFFFFFFFFFF)
    step = 1 << 22
    for a in range(0, n, step):
        idx = np.arange(a, min(n, a + step), dtype=np.uint64)
        with np.errstate(over="ignore"):
            bits = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + s)
        out[a:a + step] = (bits >> np.uint64(40)) >= thr
    return out


def normal(seed: int, key: str, shape, mean: float = 0.0, std: float = 1.0) -> np.ndarray:
    """N(mean, std^2) float32 tensor via Box-Muller on two SplitMix64 streams."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    s = _stream_seed(seed, key)
    step = 1 << 22
    for a in range(0, n, step):
        idx = np.arange(a, min(n, a + step), dtype=np.uint64)
        with np.errstate(over="ignore"):
            b1 = _splitmix64((idx * np.uint64(2)) * np.uint64(0xD1342543DE82EF95) + s)
            b2 = _splitmix64((idx * np.uint64(2) + np.uint64(1)) * np.uint64(0xD1342543DE82EF95) + s)
        u1 = ((b1 >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)
        u2 = (b2 >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
        out[a:a + step] = (mean + std * z).astype(np.float32)
    return out.reshape(shape)


# ----------------------------------------------------------------------------------------------
# model dimensions
# ----------------------------------------------------------------------------------------------

class Dims:
    """Architecture sizes read from a reference-style config (configs/dyscene.yaml keys)."""

    def __init__(self, d=768, d_head=64, tokens=64, pcd_layers=4, n_layer=16, frames=12,
                 image_size=224, patch_size=14, dino_depth=12, dino_pos_grid=37, mlp_ratio=4):
        self.d, self.d_head, self.tokens = d, d_head, tokens
        self.pcd_layers, self.n_layer, self.frames = pcd_layers, n_layer, frames
        self.image_size, self.patch_size = image_size, patch_size
        self.dino_depth, self.dino_pos_grid, self.mlp_ratio = dino_depth, dino_pos_grid, mlp_ratio
        self.grid = image_size // patch_size

    @classmethod
    def from_config(cls, config, dino_depth=12, dino_pos_grid=37):
        tc = config["model"]["video_encoder"]["transformer"]
        ic = config["model"]["video_encoder"]["image_tokenizer"]
        return cls(d=tc["d"], d_head=tc["d_head"], tokens=config["model"]["tokens"],
                   pcd_layers=config["model"]["pcd_layers"], n_layer=tc.get("n_layer", 12),
                   frames=config["training"]["frames"], image_size=ic.get("image_size", 224),
                   patch_size=ic.get("patch_size", 14), dino_depth=dino_depth, dino_pos_grid=dino_pos_grid)


def state_dict_spec(dm: Dims) -> Dict[str, Tuple[tuple, str]]:
    """key -> (shape, kind).  kind in {linear, bias, norm_w, norm_b, token, gamma, buffer, dino_qkv}.

    Mirrors the reference state-dict manifest (model/Pcd_motion.py:283-342,
    model/transformer.py:112-121,182-189,345-363,400-417) and hub DINOv2 names.
    """
    d, dh, hid = dm.d, dm.d_head, dm.d * dm.mlp_ratio
    spec: Dict[str, Tuple[tuple, str]] = {}
    spec["learnable_tokens"] = ((1, dm.tokens, d), "token")
    spec["special_token_0"] = ((1, 4, d), "token")
    spec["special_token_rest"] = ((1, 4, d), "token")
    spec["point_embed.mlp.weight"] = ((d, 51), "linear51")
    spec["point_embed.mlp.bias"] = ((d,), "bias")
    spec["point_normal_rgb_proj.weight"] = ((d, d + 6), "linear")
    spec["point_normal_rgb_proj.bias"] = ((d,), "bias")
    for blk in ("encoder_cross_attn", "decoder_cross_attn"):
        spec[f"{blk}.norm_q.weight"] = ((d,), "norm_w")
        spec[f"{blk}.norm_kv.weight"] = ((d,), "norm_w")
        for p in ("to_q", "to_k", "to_v", "fc"):
            spec[f"{blk}.attn.{p}.weight"] = ((d, d), "linear")
        spec[f"{blk}.attn.q_norm.weight"] = ((dh,), "norm_w")
        spec[f"{blk}.attn.k_norm.weight"] = ((dh,), "norm_w")
        spec[f"{blk}.norm2.weight"] = ((d,), "norm_w")
        spec[f"{blk}.mlp.mlp.0.weight"] = ((hid, d), "linear")
        spec[f"{blk}.mlp.mlp.2.weight"] = ((d, hid), "linear")
    groups = (("points_transformer_blocks", dm.pcd_layers),
              ("global_transformer_blocks", dm.n_layer // 2),
              ("local_transformer_blocks", dm.n_layer // 2))
    for name, cnt in groups:
        for i in range(cnt):
            p = f"{name}.{i}"
            spec[f"{p}.norm1.weight"] = ((d,), "norm_w")
            spec[f"{p}.attn.to_qkv.weight"] = ((3 * d, d), "linear")
            spec[f"{p}.attn.fc.weight"] = ((d, d), "linear")
            spec[f"{p}.attn.q_norm.weight"] = ((dh,), "norm_w")
            spec[f"{p}.attn.k_norm.weight"] = ((dh,), "norm_w")
            spec[f"{p}.norm2.weight"] = ((d,), "norm_w")
            spec[f"{p}.mlp.mlp.0.weight"] = ((hid, d), "linear")
            spec[f"{p}.mlp.mlp.2.weight"] = ((d, hid), "linear")
    spec["transformer_input_layernorm.weight"] = ((d,), "norm_w")
    spec["shared_mlp_output.0.weight"] = ((d,), "norm_w")
    spec["shared_mlp_output.0.bias"] = ((d,), "norm_b")
    spec["shared_mlp_output.1.weight"] = ((d, d), "linear")
    spec["shared_mlp_output.1.bias"] = ((d,), "bias")
    spec["shared_mlp_output.3.weight"] = ((3, d), "linear")
    spec["shared_mlp_output.3.bias"] = ((3,), "bias")
    # hub DINOv2 ViT (dinov2_vitb14 when d == 768): names from facebookresearch/dinov2
    pre = "image_encoder.model"
    ps, g = dm.patch_size, dm.dino_pos_grid
    spec[f"{pre}.cls_token"] = ((1, 1, d), "small")
    spec[f"{pre}.pos_embed"] = ((1, 1 + g * g, d), "small")
    spec[f"{pre}.mask_token"] = ((1, d), "zero")
    spec[f"{pre}.patch_embed.proj.weight"] = ((d, 3, ps, ps), "linear")
    spec[f"{pre}.patch_embed.proj.bias"] = ((d,), "bias")
    for i in range(dm.dino_depth):
        p = f"{pre}.blocks.{i}"
        spec[f"{p}.norm1.weight"] = ((d,), "norm_w")
        spec[f"{p}.norm1.bias"] = ((d,), "norm_b")
        spec[f"{p}.attn.qkv.weight"] = ((3 * d, d), "dino_qkv")
        spec[f"{p}.attn.qkv.bias"] = ((3 * d,), "bias")
        spec[f"{p}.attn.proj.weight"] = ((d, d), "linear")
        spec[f"{p}.attn.proj.bias"] = ((d,), "bias")
        spec[f"{p}.ls1.gamma"] = ((d,), "norm_w")
        spec[f"{p}.norm2.weight"] = ((d,), "norm_w")
        spec[f"{p}.norm2.bias"] = ((d,), "norm_b")
        spec[f"{p}.mlp.fc1.weight"] = ((hid, d), "linear")
        spec[f"{p}.mlp.fc1.bias"] = ((hid,), "bias")
        spec[f"{p}.mlp.fc2.weight"] = ((d, hid), "linear")
        spec[f"{p}.mlp.fc2.bias"] = ((d,), "bias")
        spec[f"{p}.ls2.gamma"] = ((d,), "norm_w")
    spec[f"{pre}.norm.weight"] = ((d,), "norm_w")
    spec[f"{pre}.norm.bias"] = ((d,), "norm_b")
    return spec


def synth_tensor(seed: int, key: str, shape, kind: str, perturb: bool = True) -> np.ndarray:
    if kind == "linear":
        return normal(seed, key, shape, 0.0, 0.02)
    if kind == "linear51":          # nn.Linear default init scale for fan_in 51 (Pcd_motion.py:175)
        b = 1.0 / math.sqrt(51.0)
        return uniform(seed, key, shape, -b, b)
    if kind == "dino_qkv":          # a little larger so DINO's un-normalised softmax is not flat
        return normal(seed, key, shape, 0.0, 0.04)
    if kind == "token":
        return normal(seed, key, shape, 0.0, 1.0)
    if kind == "small":
        return normal(seed, key, shape, 0.0, 0.02)
    if kind == "zero":
        return np.zeros(shape, dtype=np.float32)
    if kind == "bias" or kind == "norm_b":
        return normal(seed, key, shape, 0.0, 0.02) if perturb else np.zeros(shape, np.float32)
    if kind == "norm_w":
        return normal(seed, key, shape, 1.0, 0.1) if perturb else np.ones(shape, np.float32)
    raise ValueError(kind)


def synth_state_dict(dm: Dims, seed: int = 0, perturb: bool = True) -> Dict[str, np.ndarray]:
    """All parameters (no buffers: pos_embed / point_embed.basis are derived, not random)."""
    return {k: synth_tensor(seed, k, shape, kind, perturb) for k, (shape, kind) in state_dict_spec(dm).items()}


def synth_inputs(B: int, T: int, N: int, S: int, HW: int, seed: int = 1, with_target: bool = False,
                 frames: range = None) -> Dict[str, np.ndarray]:
    """Random sample dict with the keys Motion_Latent_Model.forward reads (Pcd_motion.py:450-582).

    rgb_video ~ U[0,1) [B,T,HW,HW,3]; point sets ~ U[-0.5,0.5)^3 (the callers normalise meshes to a
    unit cube, scripts/inference_with_video_mesh.py:94-97); normals unit-length; colours U[0,1).
    frames: only that window of the clip's frames is generated (values identical to the full clip's).
    """
    def unit(key, shape):
        v = normal(seed, key, shape).astype(np.float64)
        v /= np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-12)
        return v.astype(np.float32)

    s = {
        "ref_shape_pcd": uniform(seed, "ref_shape_pcd", (B, S, 3), -0.5, 0.5),
        "ref_shape_normals": unit("ref_shape_normals", (B, S, 3)),
        "ref_shape_rgbs": uniform(seed, "ref_shape_rgbs", (B, S, 3)),
        "ref_pcd": uniform(seed, "ref_pcd", (B, N, 3), -0.5, 0.5),
        "ref_normal": unit("ref_normal", (B, N, 3)),
        "ref_rgb": uniform(seed, "ref_rgb", (B, N, 3)),
    }
    if frames is None:
        s["rgb_video"] = uniform(seed, "rgb_video", (B, T, HW, HW, 3))
    else:
        # only frames [start, stop) of the same T-frame clip (a frame-parallel rank's shard; B = 1: the frames are contiguous)
        if B != 1:
            raise ValueError("synth_inputs(frames=...) cuts a contiguous window: B must be 1")
        s["rgb_video"] = uniform(seed, "rgb_video", (B, len(frames), HW, HW, 3), offset=frames.start * HW * HW * 3)
    if with_target:
        s["point_clouds"] = (s["ref_pcd"][:, None] + 0.05 * normal(seed, "point_clouds", (B, T, N, 3))).astype(np.float32)
    return s


def make_config(frames=12, d=768, d_head=64, tokens=64, pcd_layers=4, n_layer=16, image_size=224,
                patch_size=14, drop_rate=0.0, class_name="motion324_amd.Motion_Latent_Model") -> dict:
    """A plain-dict config with exactly the keys configs/dyscene.yaml has on the hot path."""
    return {
        "model": {
            "class_name": class_name, "feat_dim": d, "tokens": tokens, "pcd_layers": pcd_layers,
            "video_encoder": {
                "image_tokenizer": {"image_size": image_size, "patch_size": patch_size, "patch_length": 1,
                                    "in_channels": 3},
                "transformer": {"d": d, "d_head": d_head, "n_layer": n_layer, "special_init": True,
                                "depth_init": True, "use_qk_norm": True, "drop_rate": drop_rate},
            },
        },
        "training": {"frames": frames, "use_checkpoint": False, "grad_checkpoint_every": 1,
                     "coord_mse_loss_weight": 1.0, "amp_dtype": "bf16", "use_amp": True},
    }
