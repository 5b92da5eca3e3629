#!/usr/bin/env python3
"""The REFERENCE's own bf16-vs-fp32 gap on the "trained-like" weight set of tests/test_trained_like_gpu.py (build container only:
imports /root/reference through make_golden.py's shims).  The reference runs its forward under torch.autocast(bf16)
(train.py:150-155, scripts/inference_with_video_mesh.py:207-211); here the same module runs on CPU once in fp32 and once under
torch.autocast("cpu", bfloat16) on the c1 inputs, with the synthetic weights and with the trained-like ones.  Output:
tests/golden/trained_like_band.json = {"synthetic": rel err, "trained_like": rel err, ...} -- the band the HIP path's bf16 mode is
held against (data, no reference source).  Round 6 adds the same pair at the c2 TRUNK LENGTH (10368 tokens through the global
blocks, shallow depth: "..._c2_trunk") and the gradient band of one training step of the tiny configuration ("train_tiny_*":
the reference's autocast(bf16) forward + backward against its own fp32 one)."""
import json, os, sys, time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as G                                                    # noqa: E402  (shims + key mapping)
from motion324_amd import synth                                            # noqa: E402
from oracle import ref_forward as oracle                                   # noqa: E402
from test_trained_like_gpu import trained_like                             # noqa: E402


def build(sd_np, dm):
    dino_cfg = dict(hidden_size=dm.d, num_hidden_layers=dm.dino_depth, num_attention_heads=dm.d // dm.d_head,
                    image_size=dm.dino_pos_grid * dm.patch_size, patch_size=dm.patch_size, layerscale_value=1.0, mlp_ratio=4,
                    qkv_bias=True, layer_norm_eps=1e-6, hidden_act="gelu", use_swiglu_ffn=False, hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0, drop_path_rate=0.0)
    G._install_shims(dino_cfg)
    for m in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
        if "image_encoder.dino" not in m:
            del sys.modules[m]
    from model.Pcd_motion import Motion_Latent_Model
    cfg = G._EasyDict(synth.make_config(frames=dm.frames, d=dm.d, d_head=dm.d_head, tokens=dm.tokens, pcd_layers=dm.pcd_layers,
                                        n_layer=dm.n_layer, drop_rate=0.0))
    torch.manual_seed(0)
    model = Motion_Latent_Model(cfg)
    sd = oracle.to_torch(sd_np)
    non_dino = {k: v for k, v in sd.items() if not k.startswith("image_encoder.")}
    missing, unexpected = model.load_state_dict(non_dino, strict=False)
    assert not unexpected
    model.image_encoder.model.load_state_dict(G.hub_to_hfport(sd), strict=True)
    model.eval()
    return model


def _forward_band(model, sample):
    with torch.no_grad():
        ref = model(dict(sample))["pcd_moved"].float()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            low = model(dict(sample))["pcd_moved"].float()
    return float((low - ref).norm() / ref.norm())


def _grad_band(model, sample):
    """One training-mode forward + backward of the reference in fp32 and under autocast(bf16) (train.py:150-166): loss and
    gradients of every trainable tensor; returns (relative loss difference, global gradient error, worst per-tensor error)."""
    model.train()
    grads = []
    losses = []
    for low in (False, True):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=low):
            ret = model(dict(sample))
        loss = ret.loss_metrics.loss
        loss.backward()
        losses.append(float(loss))
        grads.append({n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.requires_grad and p.grad is not None})
    g32, g16 = grads
    num = sum(float((g16[n] - g32[n]).pow(2).sum()) for n in g32)
    den = sum(float(g32[n].pow(2).sum()) for n in g32)
    per = {n: float((g16[n] - g32[n]).norm() / g32[n].norm().clamp_min(1e-30)) for n in g32}
    worst = max(per.items(), key=lambda kv: kv[1])
    model.eval()
    return abs(losses[1] - losses[0]) / abs(losses[0]), (num / den) ** 0.5, worst


def main():
    from test_trained_like_gpu import GUARD_DIMS, GUARD_SHAPE
    from conftest import CASES as TEST_CASES
    out = {"autocast": "torch.autocast('cpu', dtype=torch.bfloat16) around the reference's forward (and backward)"}
    # 1. inference at the c1 size (round 5) and at the c2 trunk length: 32 frames x 324 tokens = 10368 rows through the global
    #    blocks, shallow depth (the configuration of the scores-bounded guard test) -- round 6
    for tag, dims, shape, seed in (("", G.CASES["c1"]["dims"], G.CASES["c1"]["shape"], 1), ("_c2_trunk", GUARD_DIMS, GUARD_SHAPE, 7)):
        dm = synth.Dims(**dims)
        B, T, N, S, HW = shape
        sample = oracle.to_torch(synth.synth_inputs(B, T, N, S, HW, seed=seed))
        base = synth.synth_state_dict(dm, seed=0)
        for name, sd_np in (("synthetic", base), ("trained_like", trained_like(base))):
            model = build(sd_np, dm)
            t0 = time.time()
            err = _forward_band(model, sample)
            print(f"[{name}{tag}] reference autocast(bf16) vs its own fp32: rel err {err:.3e}  ({time.time() - t0:.1f} s)", flush=True)
            out[name + tag] = err
    out["case"], out["shape"] = "c1", list(G.CASES["c1"]["shape"])
    out["case_c2_trunk"], out["shape_c2_trunk"] = dict(GUARD_DIMS), list(GUARD_SHAPE)
    # 2. one training step (forward + backward) of the tiny configuration: the gradient band of the reference's own autocast
    dims, (B, T, N, S, HW) = TEST_CASES["tiny"]["dims"], TEST_CASES["tiny"]["shape"]
    dm = synth.Dims(**dims)
    sample = oracle.to_torch(synth.synth_inputs(B, T, N, S, HW, seed=1, with_target=True))
    base = synth.synth_state_dict(dm, seed=0)
    for name, sd_np in (("synthetic", base), ("trained_like", trained_like(base))):
        model = build(sd_np, dm)
        dl, gerr, worst = _grad_band(model, sample)
        print(f"[train tiny, {name}] reference autocast(bf16) vs its own fp32: loss {dl:.3e}, all gradients {gerr:.3e}, worst tensor {worst[0]} {worst[1]:.3e}",
              flush=True)
        out["train_tiny_" + name] = {"loss": dl, "grad": gerr, "worst_tensor": worst[0], "worst": worst[1]}
    json.dump(out, open(os.path.join(HERE, "trained_like_band.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
