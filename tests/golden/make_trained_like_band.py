#!/usr/bin/env python3
"""The REFERENCE's own bf16-vs-fp32 gap on the "trained-like" weight set of tests/test_trained_like_gpu.py (build container only:
imports /root/reference through make_golden.py's shims).  The reference runs its forward under torch.autocast(bf16)
(train.py:150-155, scripts/inference_with_video_mesh.py:207-211); here the same module runs on CPU once in fp32 and once under
torch.autocast("cpu", bfloat16) on the c1 inputs, with the synthetic weights and with the trained-like ones.  Output:
tests/golden/trained_like_band.json = {"synthetic": rel err, "trained_like": rel err, ...} -- the band the HIP path's bf16 mode is
held against (data, no reference source)."""
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


def main():
    spec = G.CASES["c1"]
    dm = synth.Dims(**spec["dims"])
    B, T, N, S, HW = spec["shape"]
    sample = oracle.to_torch(synth.synth_inputs(B, T, N, S, HW, seed=1))
    base = synth.synth_state_dict(dm, seed=0)
    out = {"case": "c1", "shape": list(spec["shape"]), "autocast": "torch.autocast('cpu', dtype=torch.bfloat16) around the reference's forward"}
    for name, sd_np in (("synthetic", base), ("trained_like", trained_like(base))):
        model = build(sd_np, dm)
        t0 = time.time()
        with torch.no_grad():
            ref = model(dict(sample))["pcd_moved"].float()
            with torch.autocast("cpu", dtype=torch.bfloat16):
                low = model(dict(sample))["pcd_moved"].float()
        err = float((low - ref).norm() / ref.norm())
        print(f"[{name}] reference autocast(bf16) vs its own fp32: rel err {err:.3e}  ({time.time() - t0:.1f} s)", flush=True)
        out[name] = err
    json.dump(out, open(os.path.join(HERE, "trained_like_band.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
