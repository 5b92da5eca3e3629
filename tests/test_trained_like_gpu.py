"""Parity where tests/golden's N(1, 0.1) norm weights are blind (reference: model/transformer.py:36-42,200-207 -- the per-head
RMSNorm of q and k; utils/inference_utils.py:23-55 -- released checkpoints replace every weight):

 * the host guard of the bounded-score softmax (transformer._AttnBase.scores_bounded, threshold 48 of the kernel's 64) at the c2
   trunk size, on both sides of the threshold: which kernel the plan picks, that a weight update re-evaluates it, and that either
   kernel matches the CPU oracle;
 * a "trained-like" weight set -- DINOv2-like LayerScale 1e-5 .. 1e-1, norm weights 0.2 .. 4, a few large-magnitude channels in
   the residual streams -- through the oracle and the HIP path at the c1 size: fp32 <= 1e-3, the bf16 band recorded and gated.
"""
import numpy as np
import pytest
import torch

from conftest import CASES, rel_err, synth_sd

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-3


def _model(dims, sd_np):
    import motion324_amd as m
    from motion324_amd import synth
    dm = synth.Dims(**dims)
    cfg = synth.make_config(frames=dm.frames, d=dm.d, d_head=dm.d_head, tokens=dm.tokens, pcd_layers=dm.pcd_layers, n_layer=dm.n_layer)
    cfg["model"]["dino"] = {"depth": dm.dino_depth}
    model = m.Motion_Latent_Model(cfg)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
    assert set(missing) <= {"pos_embed", "point_embed.basis"} and not unexpected
    return model.eval().to("cuda"), dm


def _run(model, sample, precision, record=False):
    import motion324_amd as m
    from motion324_amd import timing
    m.set_precision(precision)
    try:
        with torch.no_grad():
            if record:
                with timing.Recorder() as rec:
                    out = model(sample)
                torch.cuda.synchronize()
                return out, [t for *_x, t in rec.items]
            out = model(sample)
        torch.cuda.synchronize()
        return out, []
    finally:
        m.set_precision(None)


def _oracle(sd_np, s_np, frames):
    from oracle import ref_forward as oracle
    with torch.no_grad():
        return oracle.forward({k: torch.from_numpy(v) for k, v in sd_np.items()}, oracle.to_torch(s_np), frames=frames)["pcd_moved"]


def _global_attention_kernels(tags):
    """kernel symbols of the long-sequence attention launches (Lq >= 2048) in a recorded forward"""
    return [t.split("(")[0].split(" ")[0].split("<")[0] for t in tags if "Lq=" in t and int(t.split("Lq=")[1].split()[0]) >= 2048]


# c2 trunk size (32 frames x 324 tokens = 10368 rows through the global blocks), shallow everywhere else so that the CPU oracle
# takes seconds: one global + one local block, one DINO block, one point block
GUARD_DIMS = dict(d=768, d_head=64, tokens=64, pcd_layers=1, n_layer=2, frames=32, dino_depth=1)
GUARD_SHAPE = (1, 32, 256, 512, 64)


def _scaled_qk(sd_np, bound_target):
    """q_norm / k_norm weights of the global blocks scaled so that the guard's bound 64 Q_PRESCALE max|w_q| max|w_k| = bound_target"""
    from motion324_amd import ops
    sd = dict(sd_np)
    for k in [k for k in sd if k.startswith("global_transformer_blocks.") and k.endswith("attn.q_norm.weight")]:
        kk = k.replace("q_norm", "k_norm")
        cur = 64.0 * ops.Q_PRESCALE * float(np.abs(sd[k]).max()) * float(np.abs(sd[kk]).max())
        f = float(np.sqrt(bound_target / cur))
        sd[k], sd[kk] = (sd[k] * f).astype(np.float32), (sd[kk] * f).astype(np.float32)
    return sd


@pytest.mark.parametrize("bound,kernel", [(None, "attn_pwg_bounded_kernel"), (40.0, "attn_pwg_bounded_kernel"), (60.0, "attn_pwg_kernel")])
def test_scores_bounded_guard_picks_the_kernel_and_both_match_the_oracle(bound, kernel, monkeypatch):
    from motion324_amd import synth
    import motion324_amd.transformer as tr
    assert tr.ATTN_BOUNDED                                     # the default vouches when the weights allow it
    sd = synth.synth_state_dict(synth.Dims(**GUARD_DIMS), seed=0)
    if bound is not None:
        sd = _scaled_qk(sd, bound)
    model, dm = _model(GUARD_DIMS, sd)
    B, T, N, S, HW = GUARD_SHAPE
    s_np = synth.synth_inputs(B, T, N, S, HW, seed=7)
    sample = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
    out, tags = _run(model, sample, "bf16", record=True)
    kernels = _global_attention_kernels(tags)
    assert kernels and all(k == kernel for k in kernels), (bound, kernels)
    assert torch.isfinite(out.pcd_moved).all()
    ref = _oracle(sd, s_np, dm.frames)
    err = rel_err(out.pcd_moved, ref)
    # the same forward with the guard switched off (M324_ATTN_BOUNDED=0): the lazy-maximum kernel
    monkeypatch.setattr(tr, "ATTN_BOUNDED", False)
    plain, tags0 = _run(model, sample, "bf16", record=True)
    assert all(k == "attn_pwg_kernel" for k in _global_attention_kernels(tags0))
    err0 = rel_err(plain.pcd_moved, ref)
    print(f"[guard bound={bound}] {kernel}: vs oracle {err:.2e}; M324_ATTN_BOUNDED=0 vs oracle {err0:.2e}; "
          f"between them {rel_err(out.pcd_moved, plain.pcd_moved):.2e}")
    assert err < 6e-3 and err0 < 6e-3
    assert rel_err(out.pcd_moved, plain.pcd_moved) < 6e-3
    out32, _ = _run(model, sample, "fp32")
    assert rel_err(out32.pcd_moved, ref) < FP32_TOL


def test_scores_bounded_guard_follows_weight_updates():
    """An optimizer step / load_state_dict changes the RMSNorm weights: the plan must follow the CURRENT weights."""
    from motion324_amd import synth
    sd = synth.synth_state_dict(synth.Dims(**GUARD_DIMS), seed=0)
    model, dm = _model(GUARD_DIMS, sd)
    B, T, N, S, HW = GUARD_SHAPE
    sample = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_inputs(B, T, N, S, HW, seed=7).items()}
    _, tags = _run(model, sample, "bf16", record=True)
    assert set(_global_attention_kernels(tags)) == {"attn_pwg_bounded_kernel"}
    blk = model.global_transformer_blocks[0].attn
    with torch.no_grad():
        blk.q_norm.weight.mul_(2.0)                            # in place, like an optimizer step: bound 4 x ~19 > 48
        blk.k_norm.weight.mul_(2.0)
    out, tags = _run(model, sample, "bf16", record=True)
    assert set(_global_attention_kernels(tags)) == {"attn_pwg_kernel"}
    assert torch.isfinite(out.pcd_moved).all()
    # the automatic hipGraph replay (from the third call with one set of shapes on) is keyed by the weight versions: it must
    # capture the plan of the CURRENT weights, i.e. reproduce the eager result above bit for bit
    for _ in range(4):
        again, _ = _run(model, sample, "bf16")
        assert torch.equal(again.pcd_moved, out.pcd_moved)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)      # back under the threshold
    _, tags = _run(model, sample, "bf16", record=True)
    assert set(_global_attention_kernels(tags)) == {"attn_pwg_bounded_kernel"}


def trained_like(sd_np, seed=11):
    """A weight set with the statistics of a trained checkpoint that the synthetic one lacks."""
    rng = np.random.default_rng(seed)
    sd = {}
    for k, v in sd_np.items():
        v = v.copy()
        if k.endswith("ls1.gamma") or k.endswith("ls2.gamma"):                       # DINOv2 LayerScale: 1e-5 .. 1e-1
            v = (10.0 ** rng.uniform(-5.0, -1.0, v.shape)).astype(np.float32) * np.sign(rng.standard_normal(v.shape)).astype(np.float32)
        elif k.endswith("q_norm.weight") or k.endswith("k_norm.weight"):             # grown q / k norms: sharper softmax
            v = (2.0 ** rng.uniform(-1.0, 1.3, v.shape)).astype(np.float32)
        elif k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("norm.weight") or k.endswith("layernorm.weight") \
                or k.endswith("norm_q.weight") or k.endswith("norm_kv.weight") or k == "shared_mlp_output.0.weight":
            v = (2.0 ** rng.uniform(-2.3, 2.0, v.shape)).astype(np.float32)            # 0.2 .. 4
        elif (k.endswith("mlp.mlp.2.weight") or k.endswith("mlp.fc2.weight") or k.endswith("attn.fc.weight") or k.endswith("attn.proj.weight")) \
                and v.ndim == 2:
            rows = rng.choice(v.shape[0], size=3, replace=False)                     # a few massive-activation channels
            v[rows] *= 20.0
        sd[k] = v
    return sd


def test_trained_like_weights_at_c1_size(monkeypatch):
    from motion324_amd import synth
    dims = CASES["c1"]["dims"]
    sd = trained_like(synth_sd(dims))
    model, dm = _model(dims, sd)
    B, T, N, S, HW = CASES["c1"]["shape"]
    s_np = synth.synth_inputs(B, T, N, S, HW, seed=1)
    sample = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
    ref = _oracle(sd, s_np, dm.frames)
    assert torch.isfinite(ref).all()
    out32, _ = _run(model, sample, "fp32")
    e32 = rel_err(out32.pcd_moved, ref)
    out16, _ = _run(model, sample, "bf16")
    e16 = rel_err(out16.pcd_moved, ref)
    # offsets relative to the reference points: the quantity the head regresses (pcd_moved = ref_pcd + offset)
    refp = torch.from_numpy(s_np["ref_pcd"])[:, None]
    o16 = rel_err(out16.pcd_moved.cpu() - refp, ref - refp)
    print(f"[trained-like c1] fp32 {e32:.2e}  bf16 {e16:.2e}  (offsets alone: bf16 {o16:.2e})")
    # where the band comes from: the same forward with each bf16-mode shortcut switched off (diagnostics, not gated)
    import motion324_amd.transformer as tr
    import motion324_amd.Pcd_motion as pm
    for name, mod, attr, val in (("LayerNorm fold off", tr, "FOLD_LN", 0), ("fp32 decoder stream", pm, "BF16_DECODER_STREAM", False),
                                 ("bounded softmax off", tr, "ATTN_BOUNDED", False), ("fused q|k|v epilogue off", tr, "FUSE_QKV", False),
                                 ("fused head off", pm, "FUSE_HEAD_N3", False)):
        monkeypatch.setattr(mod, attr, val)
        o, _ = _run(model, sample, "bf16")
        print(f"    {name}: bf16 {rel_err(o.pcd_moved, ref):.2e}")
        monkeypatch.undo()
    assert e32 < FP32_TOL
    assert torch.isfinite(out16.pcd_moved).all()
    assert e16 < TRAINED_LIKE_BF16_TOL
    # the reference's OWN autocast(bf16) forward against its own fp32 one on these weights and inputs (measured by
    # tests/golden/make_trained_like_band.py on the imported reference): the HIP path's bf16 mode must not be wider
    import json, os
    band = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_like_band.json")))
    print(f"    reference autocast(bf16) vs its fp32: synthetic weights {band['synthetic']:.2e}, trained-like {band['trained_like']:.2e}")
    assert e16 < band["trained_like"]


# measured on MI355X (round 5): 1.84e-2 - 1.92e-2 (profiles/r05_parity_trained_like.md; none of the bf16-mode shortcuts contributes:
# each switched off leaves 1.7e-2 - 1.8e-2); gate = measured band + 25 %
TRAINED_LIKE_BF16_TOL = 2.4e-2


def _band():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_like_band.json")))


def test_trained_like_weights_at_c2_trunk_length():
    """Round 6: the trained-like weight set through the c2 TRUNK LENGTH -- 32 frames x 324 tokens = 10368 rows through a global and a
    local block (the hand-placed long-sequence attention stream, the 256 x 256 / 128 x 128 GEMM schedules of the real clip), shallow
    everywhere else so that the CPU oracle takes seconds.  The grown q / k RMSNorm weights put the score bound above the guard's
    threshold, so this is also the lazy-maximum stream under a SHARP softmax at full length.  fp32 <= 1e-3; the bf16 band is
    held against the reference's own autocast(bf16)-vs-fp32 gap on the same weights and inputs (make_trained_like_band.py)."""
    from motion324_amd import synth
    sd = trained_like(synth.synth_state_dict(synth.Dims(**GUARD_DIMS), seed=0))
    model, dm = _model(GUARD_DIMS, sd)
    B, T, N, S, HW = GUARD_SHAPE
    s_np = synth.synth_inputs(B, T, N, S, HW, seed=7)
    sample = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
    ref = _oracle(sd, s_np, dm.frames)
    assert torch.isfinite(ref).all()
    out16, tags = _run(model, sample, "bf16", record=True)
    assert set(_global_attention_kernels(tags)) == {"attn_pwg_kernel"}          # bound 64 * 0.18 * 2.46^2 > 48: no bounded stream
    out32, _ = _run(model, sample, "fp32")
    e32, e16 = rel_err(out32.pcd_moved, ref), rel_err(out16.pcd_moved, ref)
    band = _band()
    print(f"[trained-like, c2 trunk length] fp32 {e32:.2e}  bf16 {e16:.2e}  (reference autocast vs its fp32: synthetic "
          f"{band['synthetic_c2_trunk']:.2e}, trained-like {band['trained_like_c2_trunk']:.2e})")
    assert e32 < FP32_TOL
    assert torch.isfinite(out16.pcd_moved).all()
    assert e16 < band["trained_like_c2_trunk"]
    assert e16 < TRAINED_LIKE_C2_TRUNK_BF16_TOL


@pytest.mark.parametrize("weights", ["synthetic", "trained_like"])
def test_bf16_training_step_gradient_band_against_oracle_autograd(weights):
    """Round 6: ONE bf16 training step (training.forward_backward: the hand-written HIP forward + backward) against torch autograd
    through the fp32 CPU oracle, on the synthetic and on the trained-like weights: relative loss difference and the error of ALL
    gradients taken together (|| g - g_ref || / || g_ref || over the 62 trainable tensors of the tiny configuration).  The band is held
    against the reference's own: its autocast(bf16) forward + backward against its fp32 one on the same weights and inputs
    (tests/golden/make_trained_like_band.py, train.py:150-166).  fp32 mode on the same weights: <= 1e-3 overall."""
    import motion324_amd as m
    from motion324_amd import synth, training
    from oracle import ref_forward as oracle
    dims, (B, T, N, S, HW) = CASES["tiny"]["dims"], CASES["tiny"]["shape"]
    sd_np = synth_sd(dims) if weights == "synthetic" else trained_like(synth_sd(dims))
    model, dm = _model(dims, sd_np)
    model.train()
    model.drop_rate = 0.0
    s_np = synth.synth_inputs(B, T, N, S, HW, seed=1, with_target=True)
    sd = {k: torch.from_numpy(v).clone() for k, v in sd_np.items()}
    for k, v in sd.items():
        if not k.startswith("image_encoder."):
            v.requires_grad_(True)
    ref = oracle.forward(sd, oracle.to_torch(s_np), frames=dm.frames)
    ref["loss"].backward()
    sample = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
    res = {}
    for precision in ("fp32", "bf16"):
        m.set_precision(precision)
        try:
            loss, out, G = training.forward_backward(model, sample)
            torch.cuda.synchronize()
        finally:
            m.set_precision(None)
        num = den = 0.0
        per = {}
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            g, r = G.get(p).double().cpu(), sd[name].grad.double()
            num += float((g - r.reshape(g.shape)).pow(2).sum())
            den += float(r.pow(2).sum())
            per[name] = float((g - r.reshape(g.shape)).norm() / r.norm().clamp_min(1e-30))
        res[precision] = (abs(float(loss) - float(ref["loss"])) / abs(float(ref["loss"])), (num / den) ** 0.5, max(per.items(), key=lambda kv: kv[1]))
    band = _band()["train_tiny_" + weights]
    for precision, (dl, ge, worst) in res.items():
        print(f"[train step, {weights}, {precision}] loss {dl:.2e}  all gradients {ge:.2e}  worst tensor {worst[0]} {worst[1]:.2e}")
    print(f"    reference autocast(bf16) vs its fp32: loss {band['loss']:.2e}  all gradients {band['grad']:.2e}  worst tensor {band['worst_tensor']} {band['worst']:.2e}")
    assert res["fp32"][1] < 1e-3 and res["fp32"][0] < 1e-5
    assert res["bf16"][1] < band["grad"]                     # not wider than the reference's own autocast arithmetic
    assert res["bf16"][1] < TRAIN_STEP_BF16_TOL[weights]


# gates = band measured on MI355X (round 6, profiles/r06_parity_trained_like.md: c2 trunk length 8.56e-3; one training step, all
# gradients: synthetic 3.64e-3, trained-like 1.10e-2) + 25 %
TRAINED_LIKE_C2_TRUNK_BF16_TOL = 1.1e-2
TRAIN_STEP_BF16_TOL = {"synthetic": 4.6e-3, "trained_like": 1.4e-2}
