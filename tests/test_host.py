"""Host-side contract of the drop-in (no GPU): state-dict manifest, module semantics, return type, weight cache."""
import os
import numpy as np
import pytest
import torch

from motion324_amd import synth


def small_model(**kw):
    import motion324_amd as m
    cfg = synth.make_config(frames=3, d=192, tokens=8, pcd_layers=1, n_layer=2, **kw)
    cfg["model"]["dino"] = {"depth": 2}
    return m.Motion_Latent_Model(cfg)


@pytest.fixture(scope="module")
def full_model():
    import motion324_amd as m
    return m.Motion_Latent_Model(synth.make_config(frames=12))


def test_state_dict_manifest_matches_reference(full_model):
    """196 trainable tensors (157.04 M) + 2 buffers outside DINO, hub-named DINOv2 ViT-B/14 (86.58 M)."""
    sd = full_model.state_dict()
    spec = synth.state_dict_spec(synth.Dims(frames=12))
    assert set(sd) == set(spec) | {"pos_embed", "point_embed.basis"}
    for k, (shape, _) in spec.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    non_dino = [k for k in sd if not k.startswith("image_encoder.")]
    assert len(non_dino) == 198
    trainable = [(n, p) for n, p in full_model.named_parameters() if p.requires_grad]
    assert len(trainable) == 196 and all(not n.startswith("image_encoder.") for n, _ in trainable)
    assert round(sum(p.numel() for _, p in trainable) / 1e6, 2) == 157.04
    assert round(sum(p.numel() for n, p in full_model.named_parameters() if n.startswith("image_encoder.")) / 1e6, 2) == 86.58
    assert tuple(sd["pos_embed"].shape) == (1, 12 * 256, 768) and tuple(sd["point_embed.basis"].shape) == (3, 24)
    assert tuple(sd["image_encoder.model.pos_embed"].shape) == (1, 1370, 768)
    assert tuple(sd["image_encoder.model.patch_embed.proj.weight"].shape) == (768, 3, 14, 14)
    # optimizer grouping of the reference (utils/training_utils.py:39-47): 1-D params get no weight decay
    assert sum(1 for _, p in trainable if p.dim() == 1) > 0


def test_buffers_match_oracle_restatement(full_model):
    from oracle import ref_forward as oracle
    assert torch.equal(full_model.pos_embed, oracle.generate_pos_embed(12, 16, 16, 768))
    assert torch.equal(full_model.point_embed.basis, oracle.point_basis())
    from motion324_amd.Pcd_motion import resize_pos_embed
    a = resize_pos_embed(full_model.pos_embed, (12, 16, 16), (4, 16, 16))
    assert torch.equal(a, oracle.resize_pos_embed(full_model.pos_embed, (12, 16, 16), (4, 16, 16)))
    pe = full_model.image_encoder.model.pos_embed.detach()
    assert torch.allclose(full_model.image_encoder.model.interpolated_pos(16), oracle.dino_pos_embed(pe, 16)[0])


def test_train_eval_semantics():
    model = small_model()
    assert model.training
    model.eval()                      # statement use, as every reference caller does
    assert not model.training and not model.image_encoder.model.training
    model.train()
    assert model.training and not model.image_encoder.model.training      # DINO stays in eval (dinov2.py:126-131)
    assert all(not p.requires_grad for p in model.image_encoder.parameters())
    ddp_ready = [p for p in model.parameters() if p.requires_grad]
    assert len(ddp_ready) == len(list(model.parameters())) - len(list(model.image_encoder.parameters()))


def test_cpu_forward_fails_loudly():
    from motion324_amd.lib import M324Error
    model = small_model().eval()
    s = {k: torch.from_numpy(v) for k, v in synth.synth_inputs(1, 3, 8, 16, 32).items()}
    with pytest.raises(M324Error, match="no CPU fallback"):
        model(s)


def test_config_attribute_style_and_missing_loss_weight():
    import motion324_amd as m
    cfg = m.EasyDict(synth.make_config(frames=3, d=192, tokens=8, pcd_layers=1, n_layer=2))
    cfg.model.dino = {"depth": 1}
    model = m.Motion_Latent_Model(cfg)           # attribute-style config like the reference's EasyDict
    assert model.video_length == 3 and model.num_learnable_tokens == 8 and model.drop_rate == 0.0
    del cfg["training"]["coord_mse_loss_weight"]
    with pytest.raises(ValueError, match="coord_mse_loss_weight"):
        m.Motion_Latent_Model(cfg)


def test_easydict_contract():
    from motion324_amd import EasyDict
    lm = EasyDict()
    lm.loss = torch.tensor(1.5)
    out = EasyDict(input_data={"a": 1}, pcd_moved=torch.zeros(1), loss_metrics=lm)
    assert isinstance(out, dict) and "pcd_moved" in out and out.loss_metrics.loss.item() == 1.5
    assert [k for k, _ in out.loss_metrics.items()] == ["loss"]
    out.loss_metrics.loss.data = torch.tensor(0.0)          # train.py:174 assigns .data
    assert out["loss_metrics"]["loss"].item() == 0.0
    with pytest.raises(AttributeError):
        out.nope


def test_prepared_cache_pads_converts_and_invalidates():
    from motion324_amd.prepared import Prepared, pad_k
    assert pad_k(588) == 640 and pad_k(51) == 64 and pad_k(774) == 832 and pad_k(768) == 768
    lin = torch.nn.Linear(51, 8)
    P = Prepared(torch.device("cpu"), torch.bfloat16)
    w = P.mat(lin.weight)
    assert w.shape == (8, 64) and w.dtype == torch.bfloat16 and float(w[:, 51:].abs().max()) == 0.0
    assert P.mat(lin.weight) is w                                   # cached
    with torch.no_grad():
        lin.weight.add_(1.0)                                        # optimizer-style in-place update
    w2 = P.mat(lin.weight)
    assert w2 is not w and torch.equal(w2[:, :51], lin.weight.detach().to(torch.bfloat16))
    conv = torch.nn.Conv2d(3, 4, 14, 14)
    assert P.mat(conv.weight).shape == (4, 640)
    assert P.vec(None) is None and P.vec(lin.bias).dtype == torch.float32
    cat = P.cat_rows((lin.weight, lin.weight))
    assert cat.shape == (16, 64)


def test_prepared_hands_out_an_optimizers_mirror_only_while_it_is_current():
    """prepared.register_mirror / validate_mirrors (optim.FusedAdamW's bf16 weight copies): Prepared.mat / mat_t / cat_rows return the
    registered views while the generation counter, the tensor's in-place version and its storage are what they were when the optimizer
    vouched for them; anything else falls back to the per-weight conversion."""
    from motion324_amd import prepared
    from motion324_amd.prepared import Prepared
    a, b = torch.nn.Parameter(torch.randn(8, 64)), torch.nn.Parameter(torch.randn(8, 64))
    base = torch.zeros(2 * 8 * 64, dtype=torch.bfloat16)
    base_t = torch.zeros(2 * 64 * 64, dtype=torch.bfloat16)
    views = []
    for i, p in enumerate((a, b)):
        mat, mat_t = base[i * 512:(i + 1) * 512].view(8, 64), base_t[i * 4096:(i + 1) * 4096].view(64, 64)
        prepared.register_mirror(p, mat, mat_t, base, i * 512)
        views.append((mat, mat_t))
    P = Prepared(torch.device("cpu"), torch.bfloat16)
    fallback = lambda m: m.t().contiguous()
    try:
        assert P.mat(a).data_ptr() != views[0][0].data_ptr()                  # registered, not yet vouched for
        base.copy_(torch.cat([a.detach().reshape(-1), b.detach().reshape(-1)]).to(torch.bfloat16))
        prepared.validate_mirrors([a, b])
        assert P.mat(a).data_ptr() == views[0][0].data_ptr() and P.mat_t(b, fallback).data_ptr() == views[1][1].data_ptr()
        kv = P.cat_rows((a, b))                                               # neighbours in the flat buffer: one view
        assert kv.data_ptr() == base.data_ptr() and kv.shape == (16, 64) and torch.equal(kv[8:], b.detach().to(torch.bfloat16))
        assert P.cat_rows((b, a)).data_ptr() != base.data_ptr()               # not in buffer order: concatenated copies
        assert Prepared(torch.device("cpu"), torch.float32).mat(a).dtype == torch.float32      # the mirror is bf16 only
        with torch.no_grad():
            a.mul_(2.0)                                                       # tracked in-place edit: a's copy is stale, b's is not
        assert P.mat(a).data_ptr() != views[0][0].data_ptr() and torch.equal(P.mat(a), a.detach().to(torch.bfloat16))
        assert P.mat(b).data_ptr() == views[1][0].data_ptr()
        prepared.bump_generation()                                            # raw-pointer update somewhere: nothing is trusted
        assert P.mat(b).data_ptr() != views[1][0].data_ptr()
        prepared.validate_mirrors([a, b])
        assert P.mat(b).data_ptr() == views[1][0].data_ptr()
    finally:
        prepared.drop_mirrors([a, b])
    assert P.mat(b).data_ptr() != views[1][0].data_ptr()


def test_precision_follows_override_and_env(monkeypatch):
    import motion324_amd as m
    assert m.compute_dtype() == torch.float32
    m.set_precision("bf16")
    assert m.compute_dtype() == torch.bfloat16
    m.set_precision(None)
    monkeypatch.setenv("M324_PRECISION", "bf16")
    assert m.compute_dtype() == torch.bfloat16
    monkeypatch.delenv("M324_PRECISION")
    assert m.compute_dtype() == torch.float32


def test_unsupported_configurations_are_rejected_not_approximated():
    from motion324_amd.transformer import MLP, QK_Norm_SelfAttention
    with pytest.raises(NotImplementedError):
        QK_Norm_SelfAttention(192, 32)                  # head_dim != 64
    with pytest.raises(NotImplementedError):
        MLP(192, dropout=0.1)
    model = small_model(drop_rate=0.1)
    model.train()
    s = {k: torch.from_numpy(v) for k, v in synth.synth_inputs(1, 3, 8, 16, 32).items()}
    with pytest.raises(Exception):                      # CPU tensors: no silent CPU fallback
        model(s)


def test_dropout_keep_mask_restatement():
    """Host restatement of the pos_drop mask m324_assemble_tokens generates (include/m324.h): deterministic in the seed,
    keeps ~(1-p), independent across seeds, p=0 keeps everything."""
    n = 1 << 20
    k1 = synth.dropout_keep(1234, n, 0.1)
    assert np.array_equal(k1, synth.dropout_keep(1234, n, 0.1))
    assert abs(k1.mean() - 0.9) < 2e-3
    k2 = synth.dropout_keep(1235, n, 0.1)
    assert abs((k1 & k2).mean() - 0.81) < 3e-3
    assert synth.dropout_keep(7, 1000, 0.0).all()
    # nested in p: an element dropped at p is dropped at every p' > p (same draw, higher threshold)
    assert not (synth.dropout_keep(1234, n, 0.5) & ~k1).any()


def test_checkpoint_roundtrip_reference_format(tmp_path):
    """ckpt_{step:016}.pt with the reference's dict layout; pos_embed dropped on load so a model built for another
    clip length accepts it (utils/inference_utils.py:36-49)."""
    from motion324_amd import checkpoint as ck
    src = small_model()                                   # frames = 3
    with torch.no_grad():
        for p in src.parameters():
            p.add_(0.01)
    path = ck.save_checkpoint(str(tmp_path), src, param_update_step=60000, fwdbwd_pass_step=60000)
    assert path.endswith("ckpt_0000000000060000.pt")
    ck.save_checkpoint(str(tmp_path), src, param_update_step=10000, fwdbwd_pass_step=10000)
    assert ck.find_latest(str(tmp_path)) == path
    raw = torch.load(path, weights_only=False)
    assert set(raw) == {"model", "optimizer", "lr_scheduler", "fwdbwd_pass_step", "param_update_step"}
    assert "pos_embed" in raw["model"] and "image_encoder.model.blocks.0.attn.qkv.weight" in raw["model"]

    import motion324_amd as m
    cfg = synth.make_config(frames=7, d=192, tokens=8, pcd_layers=1, n_layer=2)      # different clip length
    cfg["model"]["dino"] = {"depth": 2}
    dst = m.Motion_Latent_Model(cfg)
    info = ck.load_checkpoint(path, dst, "cpu")
    assert info == {"fwdbwd_pass_step": 60000, "param_update_step": 60000}
    a, b = src.state_dict(), dst.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a if k != "pos_embed")
    assert b["pos_embed"].shape[1] == 7 * 256                                          # regenerated, not loaded

    bad = dict(raw)
    bad["model"] = {k: v for k, v in raw["model"].items() if "decoder_cross_attn" not in k}
    torch.save(bad, path)
    with pytest.raises(RuntimeError, match="does not match"):
        ck.load_checkpoint(path, dst, "cpu")


def test_cosine_schedule_matches_the_reference_scheduler():
    """optim.cosine_with_warmup against the lr sequence the reference's create_lr_scheduler
    (transformers.get_cosine_schedule_with_warmup, utils/training_utils.py:73-82) produced for 40 steps of a
    (base 4e-4, warm-up 5, total 30) run -- stored in tests/golden/train_tiny.npz by make_train_golden.py: warm-up from 0,
    half cosine down to 0 at `total`, and the rise beyond it that the reference's formula has."""
    import numpy as np
    from conftest import GOLDEN
    from motion324_amd.optim import cosine_with_warmup
    gold = np.load(os.path.join(GOLDEN, "train_tiny.npz"))
    seq = gold["sched_lr_base4e-4_warmup5_total30"]
    got = [cosine_with_warmup(i, 5, 30, 4e-4) for i in range(len(seq))]
    assert np.allclose(got, seq, rtol=1e-12, atol=1e-18)
    assert got[0] == 0.0 and got[5] == 4e-4 and got[30] < 1e-18 and got[33] > got[31] > 0.0
    # the run the goldens' three steps used
    assert np.allclose([cosine_with_warmup(i, 1, 10, 1e-3) for i in range(3)], gold["lr"], rtol=1e-12)


@pytest.mark.parametrize("case,golden", [("tiny", "train_tiny"), ("c1", "train_c3_b1")])
def test_parameter_order_is_the_references(case, golden):
    """named_parameters() order of the trainable tensors == the reference model's (recorded by make_train_golden.py from
    the imported reference): it numbers the optimizer state that checkpoints exchange (utils/training_utils.py:38-52), and
    FusedAdamW.ckpt_names (decay group first) == the reference optimizer's own numbering."""
    import numpy as np
    from conftest import CASES, GOLDEN
    import motion324_amd as m
    gold = np.load(os.path.join(GOLDEN, golden + ".npz"))
    dm = synth.Dims(**CASES[case]["dims"])
    cfg = synth.make_config(frames=dm.frames, d=dm.d, d_head=dm.d_head, tokens=dm.tokens, pcd_layers=dm.pcd_layers, n_layer=dm.n_layer)
    cfg["model"]["dino"] = {"depth": dm.dino_depth}
    model = m.Motion_Latent_Model(cfg)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    assert [n for n, _ in named] == [str(x) for x in gold["named_parameters_order"]]
    decay_first = [n for n, p in named if p.dim() > 1] + [n for n, p in named if p.dim() <= 1]
    assert decay_first == [str(x) for x in gold["param_order"]]


def test_the_switch_table_in_design_md_is_the_one_in_switches_py():
    """DESIGN.md 7c prints motion324_amd.switches.table(); a switch added to the code must show up there (and vice versa)."""
    from motion324_amd import switches
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "DESIGN.md")).read()
    assert switches.table() in text
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "motion324_amd", "csrc", "runtime.hip")).read()
    lib_names = set(re.findall(r'\{"(M324_[A-Z0-9_]+)",', src))
    assert lib_names == set(switches.LIBRARY), (lib_names ^ set(switches.LIBRARY))
