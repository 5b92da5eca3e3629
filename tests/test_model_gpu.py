"""End-to-end GPU parity of motion324_amd.Motion_Latent_Model (HIP path through the C ABI) against
 (a) the committed golden vectors produced by the imported reference (tests/golden/*.npz) and
 (b) the CPU oracle on the same seeded inputs.
Tolerances: fp32 parity mode <= 1e-3 relative (north_star); bf16 speed mode is reported against the same
goldens with its own band (the reference's own bf16-vs-fp32 gap is 0.9-11 %)."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import CASES, GOLDEN, load_golden, rel_err, synth_sd

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-3
# bf16 speed mode vs the fp32 reference goldens.  Measured on MI355X: pcd_moved 3.8e-3 (c2) - 4.8e-3 (tiny), every captured
# non-trunk stage <= 5.2e-3; the gate sits 25 % above the measured band (round 3's 8e-3 would have let a 2x regression pass).
BF16_TOL = 6e-3            # pcd_moved and the non-trunk stages
BF16_STAGE_TOL = {"trunk_block0": 1e-2, "trunk_out": 1e-2}


def build(case, device="cuda"):
    import motion324_amd as m
    from motion324_amd import synth
    dims = CASES[case]["dims"]
    dm = synth.Dims(**dims)
    cfg = synth.make_config(frames=dm.frames, d=dm.d, d_head=dm.d_head, tokens=dm.tokens, pcd_layers=dm.pcd_layers,
                            n_layer=dm.n_layer)
    cfg["model"]["dino"] = {"depth": dm.dino_depth}
    model = m.Motion_Latent_Model(cfg)
    sd = {k: torch.from_numpy(v) for k, v in synth_sd(dims).items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert set(missing) <= {"pos_embed", "point_embed.basis"} and not unexpected
    model.eval()
    return model.to(device), dm


def inputs(case, device="cuda", with_target=True):
    from motion324_amd import synth
    B, T, N, S, HW = CASES[case]["shape"]
    s = synth.synth_inputs(B, T, N, S, HW, seed=1, with_target=with_target)
    return {k: torch.from_numpy(v).to(device) for k, v in s.items()}


def run(model, sample, precision):
    import motion324_amd as m
    m.set_precision(precision)
    try:
        cap = {}
        model._capture = cap
        with torch.no_grad():
            out = model(sample)
        torch.cuda.synchronize()
        return out, cap
    finally:
        model._capture = None
        m.set_precision(None)


def stage_errs(cap, gold):
    errs = {}
    for k, v in cap.items():
        if "stage_" + k not in gold:
            continue
        rows = gold["rows_" + k]
        got = v.reshape(-1, v.shape[-1])[torch.from_numpy(rows).to(v.device)]
        errs[k] = rel_err(got, torch.from_numpy(gold["stage_" + k]))
    return errs


@pytest.mark.parametrize("case", ["tiny", "tiny_resize", "c1", "c2"])
def test_forward_fp32_matches_reference_golden(case):
    model, dm = build(case)
    gold = load_golden(case)
    out, cap = run(model, inputs(case), "fp32")
    assert isinstance(out, dict) and "pcd_moved" in out and out.pcd_moved.dtype == torch.float32
    errs = stage_errs(cap, gold)
    errs["pcd_moved"] = rel_err(out.pcd_moved, torch.from_numpy(gold["pcd_moved"]))
    print(f"[{case} fp32] " + "  ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert all(np.isfinite(v) for v in errs.values())
    assert max(errs.values()) < FP32_TOL, errs
    loss = float(out.loss_metrics.loss)
    assert abs(loss - float(gold["loss"])) <= 1e-3 * abs(float(gold["loss"]))
    assert float(out.loss_metrics.xyz_loss) == pytest.approx(loss)


@pytest.mark.parametrize("case", ["tiny", "c1", "c2"])
def test_forward_bf16_band(case):
    model, dm = build(case)
    gold = load_golden(case)
    out, cap = run(model, inputs(case, with_target=False), "bf16")
    errs = stage_errs(cap, gold)
    errs["pcd_moved"] = rel_err(out.pcd_moved, torch.from_numpy(gold["pcd_moved"]))
    print(f"[{case} bf16] " + "  ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert torch.isfinite(out.pcd_moved).all()
    for k, v in errs.items():
        assert v < BF16_STAGE_TOL.get(k, BF16_TOL), (k, errs)
    assert "loss_metrics" not in out


@pytest.mark.parametrize("case", ["tiny", "c1"])
def test_layernorm_fold_is_invisible_within_the_bf16_band(case, monkeypatch):
    """bf16 inference with the LayerNorms folded into the GEMMs around them (M324_FOLD_LN=2, the default: every stream; =1: the
    decoder's bf16 stream only) against the same forward with the separate LayerNorm passes (=0): all inside the band of
    the reference goldens, and closer to each other than either is to the fp32 reference."""
    import motion324_amd.transformer as tr
    model, dm = build(case)
    gold = load_golden(case)
    sample = inputs(case, with_target=False)
    assert tr.FOLD_LN == 2                            # default: every stream; 1 folds the bf16 streams (decoder) only
    folded, cap_f = run(model, sample, "bf16")
    monkeypatch.setattr(tr, "FOLD_LN", 1)
    default, _ = run(model, sample, "bf16")
    monkeypatch.setattr(tr, "FOLD_LN", 2)
    assert tr.FOLD_MERGE                              # default: the consumer GEMMs merge the producers' block statistics themselves
    monkeypatch.setattr(tr, "FOLD_MERGE", False)      # ... against the m324_rowstats_finish launch between producer and consumer
    launched, _ = run(model, sample, "bf16")
    dm_ = rel_err(folded.pcd_moved, launched.pcd_moved)          # fp32 rounding of the two merges, through 40 bf16 blocks
    print(f"[{case}] merged by the consumers vs by m324_rowstats_finish {dm_:.2e}")
    assert dm_ < BF16_TOL
    monkeypatch.setattr(tr, "FOLD_MERGE", True)
    monkeypatch.setattr(tr, "FOLD_LN", 0)
    plain, cap_p = run(model, sample, "bf16")
    ref = torch.from_numpy(gold["pcd_moved"])
    ef, ep, d = rel_err(folded.pcd_moved, ref), rel_err(plain.pcd_moved, ref), rel_err(folded.pcd_moved, plain.pcd_moved)
    ed = rel_err(default.pcd_moved, ref)
    print(f"[{case}] decoder folded only {ed:.2e}  everything folded (default) {ef:.2e}  separate LayerNorm passes {ep:.2e}  "
          f"folded vs separate {d:.2e}")
    assert ed < BF16_TOL and ef < BF16_TOL and ep < BF16_TOL and d < BF16_TOL
    assert ed < 1.25 * ep + 5e-4
    assert ef < 1.25 * ep + 5e-4                    # the fold does not widen the band
    for k in ("trunk_block0", "trunk_out", "decoder_out_t0"):
        assert rel_err(cap_f[k], cap_p[k]) < BF16_STAGE_TOL.get(k, BF16_TOL), k


def test_forward_matches_oracle_other_seed():
    """Same seeded inputs through the oracle (CPU) and the HIP path, a seed the goldens do not cover."""
    from motion324_amd import synth
    from oracle import ref_forward as oracle
    model, dm = build("tiny")
    B, T, N, S, HW = 1, 4, 77, 130, 80
    s_np = synth.synth_inputs(B, T, N, S, HW, seed=5)
    sd = {k: torch.from_numpy(v) for k, v in synth_sd(CASES["tiny"]["dims"]).items()}
    with torch.no_grad():
        ref = oracle.forward(sd, oracle.to_torch(s_np), frames=dm.frames)["pcd_moved"]
    out, _ = run(model, {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}, "fp32")
    assert rel_err(out.pcd_moved, ref) < FP32_TOL


def test_autocast_selects_bf16_and_precision_modes_differ():
    model, dm = build("tiny")
    sample = inputs("tiny", with_target=False)
    with torch.no_grad():
        a = model(sample).pcd_moved.clone()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            b = model(sample).pcd_moved.clone()
        c = model(sample).pcd_moved.clone()
    assert torch.equal(a, c)                       # deterministic
    assert not torch.equal(a, b)                   # autocast switched the kernels
    assert rel_err(b, a) < BF16_TOL


def test_decoder_chunking_is_invisible(monkeypatch):
    import motion324_amd.Pcd_motion as pm
    model, dm = build("tiny")
    sample = inputs("tiny", with_target=False)
    full, _ = run(model, sample, "fp32")
    monkeypatch.setattr(pm, "DECODE_ROWS", 3 * 16)     # 16 points per pass
    chunked, _ = run(model, sample, "fp32")
    assert torch.equal(full.pcd_moved, chunked.pcd_moved)


def test_state_dict_roundtrip_and_weight_update_invalidates_cache():
    model, dm = build("tiny")
    sample = inputs("tiny", with_target=False)
    a, _ = run(model, sample, "bf16")
    with torch.no_grad():
        model.shared_mlp_output[3].bias.add_(1.0)
        model.decoder_cross_attn.attn.fc.weight.mul_(0.5)
    b, _ = run(model, sample, "bf16")
    assert not torch.allclose(a.pcd_moved, b.pcd_moved)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model2, _ = build("tiny")
    model2.load_state_dict(sd)
    c, _ = run(model2, sample, "bf16")
    assert torch.equal(b.pcd_moved, c.pcd_moved)


def test_block_modules_standalone():
    """QK_Norm_TransformerBlock / CrossAttentionBlock forward() keep the reference's call signature."""
    from motion324_amd.transformer import QK_Norm_CrossAttentionBlock, QK_Norm_TransformerBlock
    from oracle import ref_forward as oracle
    torch.manual_seed(0)
    blk = QK_Norm_TransformerBlock(192, 64).cuda()
    x = torch.randn(2, 50, 192)
    sd = {"b." + k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref = oracle.self_attn_block(sd, "b", x, 64)
    assert rel_err(blk(x.cuda()), ref) < 1e-4
    xb = QK_Norm_CrossAttentionBlock(192, 64, kv_dim=192).cuda()
    q, kv = torch.randn(2, 30, 192), torch.randn(2, 70, 192)
    sd = {"b." + k: v.detach().cpu() for k, v in xb.state_dict().items()}
    ref = oracle.cross_attn_block(sd, "b", q, kv, 64)
    kvc = kv.cuda()
    assert rel_err(xb(q.cuda(), kvc, kvc), ref) < 1e-4


def test_graph_replay_matches_eager():
    import motion324_amd as m
    model, dm = build("tiny")
    sample = inputs("tiny", with_target=False)
    m.set_precision("bf16")
    try:
        with torch.no_grad():
            eager = model(sample).pcd_moved.clone()
        fast = m.GraphedForward(model)
        a = fast(sample).pcd_moved.clone()
        # new input values, same shapes: replay must pick them up through the static buffers
        sample2 = {k: (v * 0.5 if k == "rgb_video" else v) for k, v in sample.items()}
        with torch.no_grad():
            eager2 = model(sample2).pcd_moved.clone()
        b = fast(sample2).pcd_moved.clone()
        c = fast(sample).pcd_moved.clone()
    finally:
        m.set_precision(None)
    assert torch.equal(a, eager) and torch.equal(b, eager2) and torch.equal(c, eager)
    assert not torch.equal(a, b)


def test_forward_switches_to_graph_replay_for_repeated_shapes():
    """model(sample) in inference serves the third and later calls with the same shapes from a private hipGraph
    (Pcd_motion._forward_auto_graph): same values bit for bit, the caller owns its result (a later call must not change
    it), new shapes start eager again, and `model.auto_graph = False` keeps everything eager."""
    import motion324_amd as m
    model, dm = build("tiny")
    sample = inputs("tiny", with_target=False)
    m.set_precision("bf16")
    try:
        model.auto_graph = False
        with torch.no_grad():
            eager = model(sample).pcd_moved.clone()
        assert "_ag" not in model.__dict__
        model.auto_graph = True
        with torch.no_grad():
            outs = [model(sample).pcd_moved for _ in range(4)]
        assert "_ag" in model.__dict__ and len(model.__dict__["_ag"]._graphs) == 1          # calls 3 and 4 replayed
        assert all(torch.equal(o, eager) for o in outs)
        assert outs[2].data_ptr() != outs[3].data_ptr()
        sample2 = {k: (v * 0.5 if k == "rgb_video" else v) for k, v in sample.items()}
        with torch.no_grad():
            other = model(sample2).pcd_moved
        assert torch.equal(outs[3], eager) and not torch.equal(other, eager)               # earlier results untouched
        model.auto_graph = False
        with torch.no_grad():
            assert torch.equal(model(sample2).pcd_moved, other)
        # grad mode on: always the eager path (which also skips the inference-only fused epilogues: equal to rounding only)
        model.auto_graph = True
        n_graphs = len(model.__dict__["_ag"]._graphs)
        out = model(sample).pcd_moved
        assert rel_err(out, eager) < 1e-2 and len(model.__dict__["_ag"]._graphs) == n_graphs
    finally:
        m.set_precision(None)


def test_auto_graph_survives_inference_mode_deepcopy_and_a_failed_capture(monkeypatch):
    """Round-2 advisor findings on the automatic graph replay: (1) a capture made under torch.inference_mode() must serve a
    later call under plain no_grad (static buffers live outside inference mode); (2) copy.deepcopy(model) works once graphs
    exist (they are process-local and stay behind); (3) a capture that raises falls back to the eager path and switches the
    feature off for that model instead of propagating; (4) shape sets are kept LRU, at most three."""
    import copy
    import motion324_amd as m
    import motion324_amd.graph as mg
    model, dm = build("tiny")
    sample = inputs("tiny", with_target=False)
    m.set_precision("bf16")
    try:
        model.auto_graph = False
        with torch.no_grad():
            eager = model(sample).pcd_moved.clone()
        model.auto_graph = True
        with torch.inference_mode():
            outs = [model(sample).pcd_moved.clone() for _ in range(3)]          # third call captures, inside inference mode
        assert len(model.__dict__["_ag"]._graphs) == 1
        with torch.no_grad():
            again = model(sample).pcd_moved                                      # replay + input copy under plain no_grad
        assert torch.equal(again, eager) and all(torch.equal(o, eager) for o in outs)
        twin = copy.deepcopy(model)                                              # (2)
        assert "_ag" not in twin.__dict__
        with torch.no_grad():
            assert torch.equal(twin(sample).pcd_moved, eager)
        # (4) four shape sets: only the three most recent stay captured (the long-video driver alone uses two: its first window
        # and the windows behind it)
        def shaped(n):
            return {k: (v[:, :n].contiguous() if k in ("ref_pcd", "ref_normal", "ref_rgb") else v) for k, v in sample.items()}
        with torch.no_grad():
            for n in (30, 20, 10):
                for _ in range(3):
                    model(shaped(n))
        assert len(model.__dict__["_ag"]._graphs) == 3
        # (3) a capture that fails: the call is still served, eagerly, and the model stops trying
        def boom(self, sample):
            raise RuntimeError("capture invalidated")
        monkeypatch.setattr(mg.GraphedForward, "__call__", boom)
        with torch.no_grad():
            got = model(sample).pcd_moved
        assert torch.equal(got, eager) and model.auto_graph is False and "_ag" not in model.__dict__
    finally:
        m.set_precision(None)


def test_graph_static_inputs_are_a_zero_copy_handover():
    """GraphedForward.static_inputs returns the captured graph's own input tensors: a clip written into them in place is
    what the next replay reads (no copy made by the call), and the result equals the eager forward of the same values."""
    import motion324_amd as m
    model, dm = build("tiny")
    sample = inputs("tiny", with_target=False)
    m.set_precision("bf16")
    try:
        fast = m.GraphedForward(model)
        buf = fast.static_inputs(sample)
        assert all(buf[k].data_ptr() != sample[k].data_ptr() for k in buf)
        assert torch.equal(fast(buf).pcd_moved, fast(sample).pcd_moved)
        buf["rgb_video"].mul_(0.5)
        buf["ref_pcd"].add_(0.01)
        ptrs = {k: v.data_ptr() for k, v in buf.items()}
        got = fast(buf).pcd_moved.clone()
        assert {k: v.data_ptr() for k, v in fast.static_inputs(buf).items()} == ptrs
        with torch.no_grad():
            want = model({k: v.clone() for k, v in buf.items()}).pcd_moved
    finally:
        m.set_precision(None)
    assert torch.equal(got, want)


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_image_encoder_two_graph_branches_match_one_chain(precision):
    """Under graph capture the image encoder runs its frames as two half-batches on two streams (two branches of the
    hipGraph, image_encoder.py run()); rows are independent, so the replayed result must equal the eager single-chain
    pass bit for bit -- 9 frames: uneven halves of 5 and 4."""
    import motion324_amd as m
    from motion324_amd import image_encoder
    from motion324_amd.prepared import Prepared, compute_dtype
    torch.manual_seed(5)
    enc = image_encoder.DinoEncoder(depth=2).cuda()
    video = torch.rand(9, 112, 96, 3, device="cuda")
    m.set_precision(precision)
    try:
        P = Prepared.for_module(enc, video.device, compute_dtype())
        with torch.no_grad():
            eager = enc.run(P, video).clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                enc.run(P, video)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            outs = {}
            for two in (True, False):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = enc.run(P, video, two_streams=two)
                out.zero_()
                g.replay()
                torch.cuda.synchronize()
                outs[two] = out.clone()
    finally:
        m.set_precision(None)
    assert bool(torch.isfinite(eager).all())
    assert torch.equal(outs[True], eager) and torch.equal(outs[False], eager)


def _fp_worker(rank, world, port, case, precision, ret):
    """One process per rank, all on cuda:0 (single-GPU box): the collective goes through gloo, which carries
    CUDA tensors through host memory -- the code path of forward_frame_parallel is the one RCCL would run."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import motion324_amd as m
        model, dm = build(case)
        sample = inputs(case, with_target=True)
        m.set_precision(precision)
        with torch.no_grad():
            out = model.forward_frame_parallel(sample)
        torch.cuda.synchronize()
        ret[rank] = (out.pcd_moved.cpu(), float(out.loss_metrics.loss))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world,overlap", [("tiny", 2, "0"), ("tiny", 3, "1"), ("tiny_resize", 2, "0")])
def test_frame_parallel_equals_single_gpu(case, world, overlap, monkeypatch):
    """Frames sharded over ranks (uneven for world = 3: T = 3 -> 1+1+1; tiny_resize T = 2 is too short) with the
    K/V all-gather in every global block == the single-process forward, bit for bit in fp32 up to summation order.
    overlap "1" (M324_KV_OVERLAP=1, opt-in): the rank's own keys are attended while the gather is in flight, the remote
    ranges afterwards, and the partial softmaxes merged by their log-sum-exps -- case tiny_resize (B = 1, two frames over two
    ranks); B = 2 (tiny) takes the one-attention form: the own rows are not one range of the batch-major clip order; "0": one
    attention over the gathered keys."""
    import os
    import torch.multiprocessing as mp
    monkeypatch.setenv("M324_KV_OVERLAP", overlap)            # read at import by the spawned ranks
    model, dm = build(case)
    ref, _ = run(model, inputs(case), "fp32")
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 23000 + (os.getpid() * 7 + world + 3 * int(overlap) + 11 * len(case)) % 4000
    mp.spawn(_fp_worker, args=(world, port, case, "fp32", ret), nprocs=world, join=True)
    for r in range(world):
        got, loss = ret[r]
        assert got.shape == ref.pcd_moved.shape
        assert rel_err(got, ref.pcd_moved) < 2e-6, (r, rel_err(got, ref.pcd_moved))
        assert abs(loss - float(ref.loss_metrics.loss)) < 1e-6


def _train_golden_setup(case_dims, golden):
    """Model in train mode + the golden's sample (regenerated from its seed) + the native optimizer built like the reference's
    (configs/dyscene.yaml:24-56: betas (0.9, 0.95), weight decay 0.05, clip 1.0, skip above 5 x clip)."""
    from motion324_amd import synth
    from motion324_amd.optim import FusedAdamW, backward_completion_order
    model, dm = build(case_dims)
    model.train()
    model.drop_rate = 0.0                                   # the goldens were made with pos_drop p = 0 (make_train_golden.py)
    B, T, N, S, HW = (int(v) for v in golden["meta_shape"])
    return model, B, T, N, S, HW, FusedAdamW, backward_completion_order


def _slice_errs(golden, prefix, get):
    n = int(golden["slice_n"])
    errs = {}
    for key in golden.files:
        if key.startswith(prefix):
            name = key[len(prefix):]
            errs[name] = rel_err(get(name).detach().reshape(-1)[:n], torch.from_numpy(golden[key]))
    return errs


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_three_training_steps_match_the_reference_golden(precision):
    """tests/golden/train_tiny.npz was produced by THE REFERENCE (its model, its create_optimizer = torch AdamW, its
    create_lr_scheduler = transformers' cosine schedule, train.py:135-219's statement sequence) on CPU: loss, pre-clip gradient
    norm and lr of three optimizer steps, slices of 12 gradients after the first backward and of the same 12 parameters after
    the third update.  The native path (forward_backward + FusedAdamW + cosine_with_warmup) must reproduce them."""
    import motion324_amd as m
    from motion324_amd import synth, training
    from motion324_amd.optim import cosine_with_warmup
    gold = np.load(os.path.join(GOLDEN, "train_tiny.npz"))
    model, B, T, N, S, HW, FusedAdamW, order_of = _train_golden_setup("tiny", gold)
    sample = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_inputs(B, T, N, S, HW, seed=2, with_target=True).items()}
    opt = FusedAdamW(model.named_parameters(), lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, grad_clip_norm=1.0,
                     allowed_gradnorm_factor=5.0, order=order_of(model))
    assert opt.ckpt_names == [str(n) for n in gold["param_order"]]       # create_optimizer's numbering
    by_name = dict(model.named_parameters())
    fp32 = precision == "fp32"
    m.set_precision(precision)
    try:
        losses, norms, lrs, gerr = [], [], [], None
        for step in range(3):
            lr = cosine_with_warmup(step, 1, 10, 1e-3)
            loss, _, G = training.forward_backward(model, sample, sink=opt)
            opt.finish_reduce()
            if step == 0:
                gerr = _slice_errs(gold, "grad:", lambda name: opt.grad_of(by_name[name]))
                gn = {k[len("gradnorm:"):]: float(gold[k]) for k in gold.files if k.startswith("gradnorm:")}
                gn_err = {k: abs(float(opt.grad_of(by_name[k]).double().norm()) - v) / v for k, v in gn.items()}
            info = opt.step(lr=lr)
            assert not info["skipped"]
            losses.append(float(loss)); norms.append(info["grad_norm"]); lrs.append(lr)
        torch.cuda.synchronize()
    finally:
        m.set_precision(None)
    perr = _slice_errs(gold, "param:", lambda name: by_name[name])
    print(f"[train_tiny {precision}] loss {losses} (ref {gold['loss'].tolist()})  grad norm {norms} (ref {gold['grad_norm'].tolist()})")
    print(f"   worst gradient slice {max(gerr.items(), key=lambda kv: kv[1])}  worst per-tensor norm {max(gn_err.items(), key=lambda kv: kv[1])}"
          f"  worst parameter slice {max(perr.items(), key=lambda kv: kv[1])}")
    assert np.allclose(lrs, gold["lr"], rtol=1e-12, atol=0)
    assert np.allclose(losses, gold["loss"], rtol=2e-5 if fp32 else 2e-2)
    assert np.allclose(norms, gold["grad_norm"], rtol=1e-4 if fp32 else 2.5e-2)
    assert len(gerr) == 12 and max(gerr.values()) < (2e-3 if fp32 else 2.5e-2), gerr
    assert max(gn_err.values()) < (1e-4 if fp32 else 2.5e-2), gn_err
    # bf16: Adam's update is ~ +-lr per element whatever the gradient's size, so where the gradient is at the bf16 noise level
    # a flipped sign moves a weight by 2 lr = 10 % of its 0.02 scale: 1.0e-2 measured on the worst slice after two updates
    assert max(perr.values()) < (2e-5 if fp32 else 3e-2), perr


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_full_size_gradients_match_the_reference_golden(precision):
    """dyscene.yaml shapes (12 frames x 4096 points x 4096 surface samples x 224^2, the full 16 + 4 + 12-layer model) at B = 1:
    loss, global gradient norm and slices of 12 gradients from ONE forward / backward of the imported reference
    (tests/golden/train_c3_b1.npz) -- full-size training is pinned to the reference, not only to properties of our own
    kernels (BASELINE configs[2] runs 8 of these samples per step)."""
    import motion324_amd as m
    from motion324_amd import synth, training
    gold = np.load(os.path.join(GOLDEN, "train_c3_b1.npz"))
    model, B, T, N, S, HW, FusedAdamW, order_of = _train_golden_setup("c1", gold)
    sample = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_inputs(B, T, N, S, HW, seed=3, with_target=True).items()}
    by_name = dict(model.named_parameters())
    fp32 = precision == "fp32"
    m.set_precision(precision)
    try:
        loss, out, G = training.forward_backward(model, sample)
        torch.cuda.synchronize()
    finally:
        m.set_precision(None)
    total = math.sqrt(sum(float(G.get(p).double().pow(2).sum()) for p in model.parameters() if p.requires_grad))
    gerr = _slice_errs(gold, "grad:", lambda name: G.get(by_name[name]))
    gn = {k[len("gradnorm:"):]: float(gold[k]) for k in gold.files if k.startswith("gradnorm:")}
    gn_err = {k: abs(float(G.get(by_name[k]).double().norm()) - v) / v for k, v in gn.items()}
    perr = rel_err(out.reshape(-1, 3)[:64], torch.from_numpy(gold["pcd_moved_step0"]))
    print(f"[train_c3_b1 {precision}] loss {float(loss):.8f} (ref {float(gold['loss'][0]):.8f})  grad norm {total:.6f} "
          f"(ref {float(gold['grad_norm'][0]):.6f})  pcd_moved {perr:.2e}")
    print("   gradient slices: " + "  ".join(f"{k}={v:.2e}" for k, v in sorted(gerr.items(), key=lambda kv: -kv[1])))
    print("   per-tensor norms: " + "  ".join(f"{k}={v:.2e}" for k, v in sorted(gn_err.items(), key=lambda kv: -kv[1])[:6]))
    assert abs(float(loss) - float(gold["loss"][0])) <= (2e-5 if fp32 else 2e-2) * float(gold["loss"][0])
    assert abs(total - float(gold["grad_norm"][0])) <= (2e-4 if fp32 else 3e-2) * float(gold["grad_norm"][0])
    assert len(gerr) == 12 and max(gerr.values()) < (3e-3 if fp32 else 6e-2), gerr
    assert max(gn_err.values()) < (1e-3 if fp32 else 5e-2), gn_err


@pytest.mark.parametrize("precision,tol,drop", [("fp32", 2e-3, 0.0), ("bf16", 2.5e-2, 0.0), ("fp32", 2e-3, 0.25)])
def test_training_gradients_match_oracle_autograd(precision, tol, drop):
    """forward_backward (hand-written HIP backward) vs torch autograd through the CPU oracle: loss, pcd_moved and the
    gradient of all 196-equivalent trainable tensors of the tiny config.  drop > 0: training-mode pos_drop on the
    video tokens (reference Pcd_motion.py:490), the oracle being handed the same keep-mask."""
    import motion324_amd as m
    from motion324_amd import synth, training
    from oracle import ref_forward as oracle
    model, dm = build("tiny")
    model.train()
    B, T, N, S, HW = 2, 3, 40, 100, 64
    s_np = synth.synth_inputs(B, T, N, S, HW, seed=1, with_target=True)
    sd = {k: torch.from_numpy(v).clone() for k, v in synth_sd(CASES["tiny"]["dims"]).items()}
    for k, v in sd.items():
        if not k.startswith("image_encoder."):
            v.requires_grad_(True)
    model.drop_rate = drop
    g = model.num_patches_h
    keep = torch.from_numpy(synth.dropout_keep(99, B * T * g * g * dm.d, drop)) if drop else None
    ref = oracle.forward(sd, oracle.to_torch(s_np), frames=dm.frames, drop=(keep, drop) if drop else None)
    ref["loss"].backward()
    m.set_precision(precision)
    try:
        loss, out, G = training.forward_backward(model, {k: torch.from_numpy(v).cuda() for k, v in s_np.items()},
                                                 drop_seed=99)
        torch.cuda.synchronize()
    finally:
        m.set_precision(None)
    assert abs(float(loss) - float(ref["loss"])) <= (1e-5 if precision == "fp32" else 2e-3) * abs(float(ref["loss"]))
    errs = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g = G.get(p)
        assert g is not None, f"no gradient for {name}"
        errs[name] = rel_err(g, sd[name].grad)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print(f"[train grads {precision}] worst: " + "  ".join(f"{k}={v:.2e}" for k, v in worst))
    assert len(errs) == sum(1 for k, v in sd.items() if v.requires_grad)
    assert worst[0][1] < tol, worst


def test_two_clips_in_flight_reproduce_the_solo_result_bit_for_bit():
    """Whatever shares the CUs, a kernel must compute what it computes alone.  Round 3 found a violation (a DPP move feeding a
    packed-fp32 add beside a chunk-ring GEMM: DESIGN.md section 6, tools/ln_stress.py); this is the model-level guard: the
    full-size c1 clip as two hipGraphs replayed at the same time on two streams, 12 rounds, every output compared with the
    graph's own solo replay.  (Kernels of the two clips interleave on the chip in ever different ways.)"""
    import motion324_amd as m
    model, dm = build("c1")
    sample = inputs("c1", with_target=False)
    m.set_precision("bf16")
    try:
        with torch.no_grad():
            fa, fb = m.GraphedForward(model, warmup=1), m.GraphedForward(model, warmup=1)
            ca, cb = fa.static_inputs(sample), fb.static_inputs(sample)
            solo_a = fa(ca)["pcd_moved"].clone()
            solo_b = fb(cb)["pcd_moved"].clone()
            torch.cuda.synchronize()
            assert torch.equal(solo_a, solo_b)
            sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
            bad = 0
            for rnd in range(12):
                sa.wait_stream(torch.cuda.current_stream())
                sb.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(sa):
                    oa = fa(ca)["pcd_moved"]
                with torch.cuda.stream(sb):
                    ob = fb(cb)["pcd_moved"]
                torch.cuda.current_stream().wait_stream(sa)
                torch.cuda.current_stream().wait_stream(sb)
                torch.cuda.synchronize()
                bad += int(not torch.equal(oa, solo_a)) + int(not torch.equal(ob, solo_a))
    finally:
        m.set_precision(None)
    assert bad == 0, f"{bad} of 24 concurrent replays differ from the solo replay"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_kept_internals_and_recompute_give_the_same_step(precision, monkeypatch):
    """The training forward keeps every block's internals while they fit its memory budget (training.py, M324_TRAIN_STORE;
    97.8 ms per step at dyscene.yaml shapes against 115 ms with the reference's checkpoint-and-recompute policy); a block past
    the budget is checkpointed and recomputed in the backward.  All three forms -- everything kept, nothing kept, a budget that
    runs out in the middle of the trunk -- must deliver the same loss, outputs and gradients (same kernels on the same
    inputs; only the GELU of the kept forward comes from the fp32 accumulator instead of the rounded pre-activation)."""
    import motion324_amd as m
    from motion324_amd import synth, training
    model, dm = build("tiny")
    model.train()
    model.drop_rate = 0.0
    B, T, N, S, HW = 2, 3, 40, 100, 64
    s_np = synth.synth_inputs(B, T, N, S, HW, seed=1, with_target=True)
    sample = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
    runs = {}
    m.set_precision(precision)
    try:
        for mode in ("keep", "recompute", "half"):
            monkeypatch.setattr(training, "TRAIN_STORE", "0" if mode == "recompute" else "1")
            if mode == "half":                                # budget for three trunk blocks only, none for the decoder
                Lt = 4 + model.num_learnable_tokens + model.num_patches_h * model.num_patches_w
                monkeypatch.setattr(training, "_store_budget", lambda dev: 3 * B * T * Lt * model.embed_dim * 50 + 1)
            loss, out, G = training.forward_backward(model, sample)
            torch.cuda.synchronize()
            runs[mode] = (float(loss), out.clone(), {n: G.get(p).clone() for n, p in model.named_parameters() if p.requires_grad})
    finally:
        m.set_precision(None)
    tol = 1e-6 if precision == "fp32" else 6e-3
    for mode in ("recompute", "half"):
        assert abs(runs[mode][0] - runs["keep"][0]) <= tol * abs(runs["keep"][0])
        assert rel_err(runs[mode][1], runs["keep"][1]) <= tol
        worst = max(rel_err(g, runs["keep"][2][n]) for n, g in runs[mode][2].items())
        assert worst <= (1e-5 if precision == "fp32" else 2e-2), (mode, worst)
    assert len(runs["keep"][2]) == len(runs["recompute"][2]) >= 60


def test_out_of_memory_step_is_redone_with_recompute_after_the_failed_attempt_is_released(monkeypatch):
    """training.forward_backward redoes a step that ran out of memory once, with the reference's checkpoint + recompute policy.
    The retry must start AFTER the failed attempt's activations are gone: while an `except` block is active the exception's
    traceback keeps the failed frames -- and every tensor they hold -- alive, so the retry runs behind it."""
    import warnings
    from motion324_amd import synth, training
    model, dm = build("tiny")
    model.train()
    model.drop_rate = 0.0
    s_np = synth.synth_inputs(2, 3, 40, 100, 64, seed=1, with_target=True)
    sample = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
    monkeypatch.setattr(training, "TRAIN_STORE", "0")
    loss0, out0, G0 = training.forward_backward(model, sample)
    want = (float(loss0), out0.clone(), {n: G0.get(p).clone() for n, p in model.named_parameters() if p.requires_grad})
    del loss0, out0, G0
    monkeypatch.setattr(training, "TRAIN_STORE", "1")
    real = training._forward_backward
    state = {"calls": 0, "allocated_at_retry": None}
    big = 256 << 20

    def flaky(*a, **k):
        state["calls"] += 1
        if state["calls"] == 1:
            hog = torch.empty(big, dtype=torch.uint8, device="cuda")      # a local of the failing frame, like its activations
            hog.fill_(1)
            raise torch.cuda.OutOfMemoryError("synthetic: out of memory with kept block internals")
        state["allocated_at_retry"] = torch.cuda.memory_allocated()
        assert training.TRAIN_STORE == "0"                    # the retry runs under the recompute policy
        return real(*a, **k)

    monkeypatch.setattr(training, "_forward_backward", flaky)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        loss, out, G = training.forward_backward(model, sample)
    torch.cuda.synchronize()
    assert state["calls"] == 2 and any("redone with" in str(x.message) for x in w)
    assert state["allocated_at_retry"] < base + big // 2, (state, base)       # the failed attempt's 256 MiB were released first
    assert training.TRAIN_STORE == "1"                        # policy restored
    assert float(loss) == want[0] and torch.equal(out, want[1])
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert torch.equal(G.get(p), want[2][n]), n


def test_training_mode_dropout_is_seeded_by_torch_and_off_in_eval():
    """pos_drop: p = transformer.drop_rate in train(), identity in eval(); the mask follows torch.manual_seed."""
    from motion324_amd import synth
    model, dm = build("tiny")
    model.drop_rate = 0.1
    s = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_inputs(1, 3, 20, 50, 64, seed=4, with_target=True).items()}
    with torch.no_grad():
        model.eval()
        e1, e2 = model(s).pcd_moved.clone(), model(s).pcd_moved.clone()
        model.train()
        torch.manual_seed(5); t1 = model(s).pcd_moved.clone()
        torch.manual_seed(5); t2 = model(s).pcd_moved.clone()
        torch.manual_seed(6); t3 = model(s).pcd_moved.clone()
    assert torch.equal(e1, e2) and torch.equal(t1, t2)
    assert not torch.equal(t1, t3) and not torch.equal(t1, e1)
    assert rel_err(t1, e1) < 0.5             # a perturbation, not garbage
    torch.manual_seed(5)
    l1 = float(model(s).loss_metrics.loss)   # grad-enabled training path draws the same seed -> same forward
    with torch.no_grad():
        torch.manual_seed(5)
        l2 = float(model(s).loss_metrics.loss)
    assert l1 == pytest.approx(l2, rel=1e-6)


def test_autograd_wrapper_and_three_optimizer_steps_match_torch_on_oracle():
    """(1) model(sample).loss_metrics.loss.backward() fills .grad exactly like forward_backward;
    (2) three FusedAdamW steps (clip 1.0, decay on dim()>1 only) track torch.optim.AdamW driven by oracle autograd."""
    import motion324_amd as m
    from motion324_amd import synth
    from motion324_amd.optim import FusedAdamW
    from oracle import ref_forward as oracle
    model, dm = build("tiny")
    model.train()
    B, T, N, S, HW = 1, 3, 30, 80, 64
    s_np = synth.synth_inputs(B, T, N, S, HW, seed=2, with_target=True)
    sample = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
    # reference: oracle + torch AdamW on CPU (fp64 params would hide fp32 rounding: keep fp32 like the reference)
    sd = {k: torch.from_numpy(v).clone() for k, v in synth_sd(CASES["tiny"]["dims"]).items()}
    train_keys = [k for k in sd if not k.startswith("image_encoder.")]
    for k in train_keys:
        sd[k].requires_grad_(True)
    decay = [sd[k] for k in train_keys if sd[k].dim() > 1]
    no_decay = [sd[k] for k in train_keys if sd[k].dim() <= 1]
    ropt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.05}, {"params": no_decay, "weight_decay": 0.0}],
                             lr=1e-3, betas=(0.9, 0.95), eps=1e-8)
    m.set_precision("fp32")
    try:
        # (1) autograd wrapper
        ret = model(sample)
        assert isinstance(ret, dict) and ret.loss_metrics.loss.requires_grad
        ret.loss_metrics.loss.backward()
        from motion324_amd import training
        _, _, G = training.forward_backward(model, sample)
        for name, p in model.named_parameters():
            if p.requires_grad:
                assert p.grad is not None and torch.allclose(p.grad, G.get(p), rtol=1e-5, atol=1e-9), name
        model.zero_grad(set_to_none=True)
        # (2) optimizer trajectory
        opt = FusedAdamW(model.named_parameters(), lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05,
                         grad_clip_norm=1.0, allowed_gradnorm_factor=1e9)
        losses, ref_losses = [], []
        for step in range(3):
            ref = oracle.forward(sd, oracle.to_torch(s_np), frames=dm.frames)
            ropt.zero_grad()
            ref["loss"].backward()
            torch.nn.utils.clip_grad_norm_([sd[k] for k in train_keys], 1.0)
            ropt.step()
            ref_losses.append(float(ref["loss"]))
            loss, _, G = training.forward_backward(model, sample)
            opt.load_grads(G)
            info = opt.step()
            assert not info["skipped"]
            losses.append(float(loss))
        torch.cuda.synchronize()
    finally:
        m.set_precision(None)
    assert all(abs(a - b) <= 1e-4 * abs(b) for a, b in zip(losses, ref_losses)), (losses, ref_losses)
    got = dict(model.named_parameters())
    worst = max(rel_err(got[k].detach(), sd[k].detach()) for k in train_keys)
    assert worst < 1e-4, worst
    assert losses[2] < losses[0]


def test_gradient_accumulation_equals_the_whole_batch():
    """train.py:159-162 with grad_accum_steps = 2: two micro-batches, each back-propagating loss / 2, accumulate into .grad and
    ONE optimizer step follows.  Native form: forward_backward(..., grad_scale = 1/2) twice + FusedAdamW.load_grads(accumulate =
    True on the second) == one forward_backward over the concatenated batch (the loss is a mean over the batch), and the
    optimizer step that follows moves the parameters identically (fp32 parity kernels: summation order only)."""
    import motion324_amd as m
    from motion324_amd import synth, training
    from motion324_amd.optim import FusedAdamW
    s_np = synth.synth_inputs(2, 3, 30, 80, 64, seed=7, with_target=True)
    whole = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
    halves = [{k: v[i:i + 1].contiguous() for k, v in whole.items()} for i in range(2)]
    m.set_precision("fp32")
    try:
        results = []
        for mode in ("whole", "accumulated"):
            model, dm = build("tiny")
            model.train()
            model.drop_rate = 0.0
            opt = FusedAdamW(model.named_parameters(), lr=1e-3, allowed_gradnorm_factor=1e9)
            if mode == "whole":
                loss, _, G = training.forward_backward(model, whole)
                opt.load_grads(G)
                losses = [float(loss)]
            else:
                losses = []
                for i, part in enumerate(halves):
                    loss, _, G = training.forward_backward(model, part, grad_scale=0.5)
                    opt.load_grads(G, accumulate=i > 0)
                    losses.append(float(loss))
            grad = opt.flat_grad.clone()
            info = opt.step()
            torch.cuda.synchronize()
            results.append((losses, grad, info["grad_norm"], torch.cat([p.detach().reshape(-1) for p in opt.params]).clone()))
    finally:
        m.set_precision(None)
    (lw, gw, nw, pw), (la, ga, na, pa) = results
    assert lw[0] == pytest.approx(sum(la) / 2, rel=1e-6)
    assert rel_err(ga, gw) < 1e-5 and na == pytest.approx(nw, rel=1e-5)
    assert rel_err(pa, pw) < 1e-6


def _ddp_worker(rank, world, port, ret):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import motion324_amd as m
        from motion324_amd import synth, training
        from motion324_amd.optim import FusedAdamW
        model, dm = build("tiny")
        model.train()
        s_np = synth.synth_inputs(2, 3, 30, 80, 64, seed=4, with_target=True)
        mine = {k: torch.from_numpy(v[rank:rank + 1]).cuda() for k, v in s_np.items()}      # this rank's sample
        m.set_precision("fp32")
        opt = FusedAdamW(model.named_parameters(), lr=1e-3, allowed_gradnorm_factor=1e9)
        loss, _, G = training.forward_backward(model, mine)
        opt.load_grads(G)
        opt.all_reduce_mean()                                   # one flat all-reduce (gloo here, RCCL on a node)
        info = opt.step()
        torch.cuda.synchronize()
        ret[rank] = (opt.flat_grad.cpu(), torch.cat([p.detach().reshape(-1).cpu() for p in opt.params]), info["grad_norm"])
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_equals_single_process_on_the_concatenated_batch():
    """DDP semantics (train.py:88-89): 2 ranks x batch 1 with the flat gradient all-reduce(mean) == 1 process x batch 2."""
    import os
    import torch.multiprocessing as mp
    import motion324_amd as m
    from motion324_amd import synth, training
    from motion324_amd.optim import FusedAdamW
    model, dm = build("tiny")
    model.train()
    s_np = synth.synth_inputs(2, 3, 30, 80, 64, seed=4, with_target=True)
    m.set_precision("fp32")
    try:
        opt = FusedAdamW(model.named_parameters(), lr=1e-3, allowed_gradnorm_factor=1e9)
        loss, _, G = training.forward_backward(model, {k: torch.from_numpy(v).cuda() for k, v in s_np.items()})
        opt.load_grads(G)
        info = opt.step()
        torch.cuda.synchronize()
    finally:
        m.set_precision(None)
    ref_grad = opt.flat_grad.cpu()
    ref_params = torch.cat([p.detach().reshape(-1).cpu() for p in opt.params])
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 27000 + (os.getpid() * 3) % 4000
    mp.spawn(_ddp_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in range(2):
        g, p, norm = ret[r]
        assert rel_err(g, ref_grad) < 1e-5
        assert rel_err(p, ref_params) < 1e-6
        assert abs(norm - info["grad_norm"]) < 1e-5 * info["grad_norm"]
    assert torch.equal(ret[0][1], ret[1][1])                    # replicas stay bit-identical


def test_reference_train_loop_sequence_runs_unchanged_and_matches_native_path():
    """The statement sequence of the reference's train.py:150-213 (autocast bf16 -> model(batch) -> scaler.scale(loss)
    .backward() -> nan_to_num_ -> clip_grad_norm_ -> fused torch AdamW) on the drop-in model, vs the native
    forward_backward + FusedAdamW path: same losses, same weights."""
    import motion324_amd as m
    from motion324_amd import synth, training
    from motion324_amd.optim import FusedAdamW
    s_np = synth.synth_inputs(2, 3, 30, 80, 64, seed=6, with_target=True)

    def make():
        model, _ = build("tiny")
        model.train()
        return model
    # ---- the reference's loop body
    model = make()
    decay = [p for n, p in model.named_parameters() if p.requires_grad and p.dim() > 1]
    no_decay = [p for n, p in model.named_parameters() if p.requires_grad and p.dim() <= 1]
    optimizer = torch.optim.AdamW([{"params": decay, "weight_decay": 0.05}, {"params": no_decay, "weight_decay": 0.0}],
                                  lr=1e-3, betas=(0.9, 0.95), fused=True)
    scaler = torch.amp.GradScaler("cuda", enabled=False)
    ref_losses = []
    for step in range(2):
        batch = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
        batch["cur_train_step"] = step                                       # train.py:148 adds a non-tensor entry
        with torch.autocast(enabled=True, device_type="cuda", dtype=torch.bfloat16):
            ret_dict = model(batch)
        scaler.scale(ret_dict.loss_metrics.loss / 1).backward()
        assert not (torch.isnan(ret_dict.loss_metrics.loss) or torch.isinf(ret_dict.loss_metrics.loss))
        scaler.unscale_(optimizer)
        with torch.no_grad():
            for p in decay + no_decay:
                p.grad.nan_to_num_(nan=0.0, posinf=1e-6, neginf=-1e-6)
        total = torch.nn.utils.clip_grad_norm_(decay + no_decay, max_norm=1.0).item()
        assert math.isfinite(total)
        scaler.step(optimizer)
        scaler.update()
        optimizer.zero_grad(set_to_none=True)
        ref_losses.append({k: v.item() for k, v in ret_dict.loss_metrics.items()}["loss"])
    # ---- native path
    native = make()
    opt = FusedAdamW(native.named_parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, grad_clip_norm=1.0,
                     allowed_gradnorm_factor=1e9)
    m.set_precision("bf16")
    try:
        nat_losses = []
        for step in range(2):
            loss, _, G = training.forward_backward(native, {k: torch.from_numpy(v).cuda() for k, v in s_np.items()})
            opt.load_grads(G)
            opt.step()
            nat_losses.append(float(loss))
    finally:
        m.set_precision(None)
    # step 0 is the same code on the same weights; afterwards torch's fused AdamW and m324_adamw round differently
    # (1 ulp of an fp32 weight can flip its bf16 copy), so later losses agree to bf16 resolution only
    assert ref_losses[0] == pytest.approx(nat_losses[0], rel=1e-6)
    assert ref_losses == pytest.approx(nat_losses, rel=1e-3)
    assert ref_losses[1] < 0.6 * ref_losses[0]          # the caller's optimizer really moved the weights the kernels read
    a, b = dict(model.named_parameters()), dict(native.named_parameters())
    worst = max(rel_err(a[k].detach(), b[k].detach()) for k in a if a[k].requires_grad)
    assert worst < 2e-3, worst          # Adam normalises: bf16-level gradient differences of step 1 move small entries


def test_training_full_size_gradients_finite_and_precisions_agree():
    """dyscene.yaml shapes (T = 12, N = S = 4096, 224 x 224, full-depth model) at B = 2: every gradient of the bf16 path is
    finite and agrees with the fp32 parity path (the CPU oracle's autograd is out of reach at this size).  Catches
    shape-dependent kernel faults that the tiny-config parity tests cannot (the LDS-DMA race of the dQ kernel did)."""
    import motion324_amd as m
    from motion324_amd import synth, training
    cfg = synth.make_config(frames=12)
    model = m.Motion_Latent_Model(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.Dims(frames=12), seed=0).items()},
                          strict=False)
    model = model.train().cuda()
    s_np = synth.synth_inputs(2, 12, 4096, 4096, 224, seed=3, with_target=True)
    sample = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
    grads = {}
    try:
        for prec in ("bf16", "fp32"):
            m.set_precision(prec)
            loss, _, G = training.forward_backward(model, sample)
            torch.cuda.synchronize()
            grads[prec] = (float(loss), {n: G.get(p).float().clone() for n, p in model.named_parameters() if p.requires_grad})
    finally:
        m.set_precision(None)
    (lb, gb), (lf, gf) = grads["bf16"], grads["fp32"]
    assert lb == pytest.approx(lf, rel=5e-3)
    for n, g in gb.items():
        assert torch.isfinite(g).all(), f"non-finite bf16 gradient: {n}"
        assert torch.isfinite(gf[n]).all(), f"non-finite fp32 gradient: {n}"
    nb = math.sqrt(sum(float(g.double().pow(2).sum()) for g in gb.values()))
    nf = math.sqrt(sum(float(g.double().pow(2).sum()) for g in gf.values()))
    assert nb == pytest.approx(nf, rel=3e-2)
    worst = max((rel_err(gb[n], gf[n]), n) for n in gb)
    assert worst[0] < 0.2, worst          # bf16 vs fp32 through 37 blocks; the tiny-config test holds the tight band
