#!/usr/bin/env python3
"""Training goldens produced by THE REFERENCE ITSELF (imported from /root/reference, CPU, fp32).

What train.py:135-219 does with one batch, executed with the reference's own pieces:
    model = model.Pcd_motion.Motion_Latent_Model(config).train()
    optimizer, ... = utils.training_utils.create_optimizer(model, weight_decay, lr, (beta1, beta2))      (:38-52)
    lr_scheduler = utils.training_utils.create_lr_scheduler(optimizer, train_steps, warmup, 'cosine')     (:73-82)
    per step: loss = model(batch).loss_metrics.loss; loss.backward(); grad.nan_to_num_(...);
              total_grad_norm = clip_grad_norm_(params, grad_clip_norm); [skip if > factor * clip];
              optimizer.step(); lr_scheduler.step(); optimizer.zero_grad()
(`fused=True` of create_optimizer needs CUDA tensors; on this CPU-only container the same call is made with torch's
default implementation -- same AdamW arithmetic -- by shadowing torch.optim.AdamW's `fused` keyword.)

Cases (weights / inputs regenerate from seeds through motion324_amd.synth, the fixtures hold expected values only):
  train_tiny : tiny dims, B=1 x 3 frames x 30 points x 80 surface x 64^2, 3 optimizer steps: per step loss, pre-clip
               gradient norm and lr; after step 0 slices of 10 gradients; after step 3 slices of the same 10 parameters;
               the schedule's lr for 40 steps of a (warmup 5, total 30) run.
  train_c3_b1: dyscene.yaml shapes (12 frames x 4096 points x 4096 surface x 224^2) at B=1, full-size model: loss, gradient
               norm and slices of 12 gradients of ONE forward/backward (so that full-size training is pinned to the
               reference, not only to properties of our own kernels).
pos_drop is set to p = 0 (drop_rate 0.0), as in make_golden.py: the reference's dropout draws from torch's RNG stream, which
no other implementation can reproduce; the dropout path itself is tested against the oracle with an injected mask.

Usage:  python tests/golden/make_train_golden.py [train_tiny train_c3_b1]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (shims + hub key mapping; also puts the repo and the reference on sys.path)
from motion324_amd import synth  # noqa: E402
from oracle import ref_forward as oracle  # noqa: E402

CASES = {
    "train_tiny": dict(dims=dict(d=192, d_head=64, tokens=8, pcd_layers=1, n_layer=2, frames=3, dino_depth=2),
                       shape=(1, 3, 30, 80, 64), seed=2, steps=3, lr=1e-3, warmup=1, total=10),
    "train_c3_b1": dict(dims=dict(frames=12), shape=(1, 12, 4096, 4096, 224), seed=3, steps=0, lr=4e-4, warmup=1000, total=30000),
}
HP = dict(weight_decay=0.05, betas=(0.9, 0.95), grad_clip_norm=1.0, allowed_gradnorm_factor=5.0)   # configs/dyscene.yaml:24-56

# gradient / parameter slices kept in the fixture: (state-dict name, number of leading flat elements)
SLICE_KEYS = ["shared_mlp_output.3.weight", "shared_mlp_output.1.weight", "decoder_cross_attn.attn.to_q.weight",
              "decoder_cross_attn.mlp.mlp.2.weight", "local_transformer_blocks.0.attn.to_qkv.weight",
              "global_transformer_blocks.0.mlp.mlp.0.weight", "global_transformer_blocks.0.norm1.weight",
              "transformer_input_layernorm.weight", "special_token_0", "learnable_tokens",
              "encoder_cross_attn.attn.to_k.weight", "point_embed.mlp.weight"]
SLICE_N = 512


def build_reference(dm):
    dino_cfg = dict(hidden_size=dm.d, num_hidden_layers=dm.dino_depth, num_attention_heads=dm.d // dm.d_head,
                    image_size=dm.dino_pos_grid * dm.patch_size, patch_size=dm.patch_size, layerscale_value=1.0, mlp_ratio=4,
                    qkv_bias=True, layer_norm_eps=1e-6, hidden_act="gelu", use_swiglu_ffn=False, hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0, drop_path_rate=0.0)
    mg._install_shims(dino_cfg)
    for m in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
        if "image_encoder.dino" not in m:
            del sys.modules[m]
    from model.Pcd_motion import Motion_Latent_Model
    cfg = mg._EasyDict(synth.make_config(frames=dm.frames, d=dm.d, d_head=dm.d_head, tokens=dm.tokens, pcd_layers=dm.pcd_layers,
                                         n_layer=dm.n_layer, drop_rate=0.0))
    torch.manual_seed(0)
    model = Motion_Latent_Model(cfg)
    sd = oracle.to_torch(synth.synth_state_dict(dm, seed=0))
    non_dino = {k: v for k, v in sd.items() if not k.startswith("image_encoder.")}
    missing, unexpected = model.load_state_dict(non_dino, strict=False)
    missing = [k for k in missing if not k.startswith("image_encoder.") and k not in ("pos_embed", "point_embed.basis")]
    assert not missing and not unexpected, (missing, unexpected)
    model.image_encoder.model.load_state_dict(mg.hub_to_hfport(sd), strict=True)
    return model


def run_case(name):
    spec = CASES[name]
    dm = synth.Dims(**spec["dims"])
    B, T, N, S, HW = spec["shape"]
    model = build_reference(dm)
    model.train()                                            # the reference's override keeps DINO in eval mode itself
    # the reference's own optimizer / scheduler factories; `fused=True` is CUDA-only -> default implementation on CPU
    import utils.training_utils as tu
    real_adamw = torch.optim.AdamW

    def cpu_adamw(groups, **kw):
        kw.pop("fused", None)
        return real_adamw(groups, **kw)
    torch.optim.AdamW = cpu_adamw
    try:
        optimizer, optimized, _ = tu.create_optimizer(model, HP["weight_decay"], spec["lr"], HP["betas"])
    finally:
        torch.optim.AdamW = real_adamw
    sched = tu.create_lr_scheduler(optimizer, spec["total"], spec["warmup"], scheduler_type="cosine")
    params = list(optimized.values())
    names = list(optimized.keys())
    assert all(k in optimized for k in SLICE_KEYS), [k for k in SLICE_KEYS if k not in optimized]
    sample = oracle.to_torch(synth.synth_inputs(B, T, N, S, HW, seed=spec["seed"], with_target=True))
    name_of = {id(p): n for n, p in optimized.items()}
    numbering = [name_of[id(p)] for g in optimizer.param_groups for p in g["params"]]     # state-dict index -> parameter
    save = {"meta_shape": np.array([B, T, N, S, HW], dtype=np.int64), "slice_n": np.int64(SLICE_N),
            "param_order": np.array(numbering),              # the optimizer's own numbering: decay group first
            "named_parameters_order": np.array(names)}
    losses, norms, lrs = [], [], []
    t0 = time.time()
    for step in range(max(1, spec["steps"])):
        lrs.append(optimizer.param_groups[0]["lr"])
        ret = model(dict(sample))
        loss = ret["loss_metrics"]["loss"]
        loss.backward()
        with torch.no_grad():
            for p in params:
                if p.grad is not None:
                    p.grad.nan_to_num_(nan=0.0, posinf=1e-6, neginf=-1e-6)
        if step == 0:
            for k in SLICE_KEYS:
                g = optimized[k].grad
                save["grad:" + k] = g.reshape(-1)[:SLICE_N].clone().numpy()
                save["gradnorm:" + k] = np.float64(g.double().norm())
            save["pcd_moved_step0"] = ret["pcd_moved"].detach().reshape(-1, 3)[:64].numpy()
        total = float(torch.nn.utils.clip_grad_norm_(params, max_norm=HP["grad_clip_norm"]))
        losses.append(float(loss))
        norms.append(total)
        print(f"[{name}] step {step}: loss {float(loss):.8f}  grad norm {total:.6f}  lr {lrs[-1]:.3e}  ({time.time() - t0:.0f}s)")
        if spec["steps"] == 0:
            break
        assert total <= HP["allowed_gradnorm_factor"] * HP["grad_clip_norm"], "the reference would skip this step"
        optimizer.step()
        sched.step()
        optimizer.zero_grad(set_to_none=True)
    save["loss"] = np.array(losses, dtype=np.float64)
    save["grad_norm"] = np.array(norms, dtype=np.float64)
    save["lr"] = np.array(lrs, dtype=np.float64)
    if spec["steps"] > 0:
        for k in SLICE_KEYS:
            save["param:" + k] = optimized[k].detach().reshape(-1)[:SLICE_N].clone().numpy()
        # the schedule alone, far enough to cover warm-up, decay and the flat end (transformers' LambdaLR, :73-82)
        probe = real_adamw([torch.nn.Parameter(torch.zeros(1))], lr=4e-4)
        s2 = tu.create_lr_scheduler(probe, 30, 5, scheduler_type="cosine")
        seq = []
        for _ in range(40):
            seq.append(probe.param_groups[0]["lr"])
            probe.step()
            s2.step()
        save["sched_lr_base4e-4_warmup5_total30"] = np.array(seq, dtype=np.float64)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **save)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    for n in (sys.argv[1:] or list(CASES)):
        run_case(n)
