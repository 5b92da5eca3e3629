"""BASELINE.json configs beyond c1 / c2 on the MI355X, and the caller-side rows of SURVEY.md 8(f) driven through the real
HIP model:
  c3  single-GPU training step at dyscene.yaml shapes, batch_size_per_gpu = 8, fused AdamW (configs[2]);
  c4  the same step data-parallel (2 ranks here, gloo transport on the one device; RCCL on a node) incl. the
      bucketed, overlapped gradient all-reduce and torch's own DistributedDataParallel wrapper (configs[3]);
  c5  the 256-frame clip against a golden produced by the reference itself (configs[4]), frame-parallel at real sizes;
  (f)1 the sliding-window driver with the real model, (f)3 checkpoints on the device."""
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import CASES, GOLDEN, load_golden, rel_err, synth_sd
from test_model_gpu import BF16_TOL, FP32_TOL, build, inputs, run

pytestmark = pytest.mark.gpu


def _full_model(frames, train=False):
    import motion324_amd as m
    from motion324_amd import synth
    cfg = synth.make_config(frames=frames)
    model = m.Motion_Latent_Model(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_sd(dict(frames=frames)).items()}, strict=False)
    return (model.train() if train else model.eval()).cuda()


# ------------------------------------------------------------------------------------------------------ c3
def test_c3_training_step_batch8_gradients_and_three_optimizer_steps():
    """configs[2]: dyscene.yaml shapes (12 frames, 4096 + 4096 points, 224 x 224), batch_size_per_gpu = 8, bf16.
    The loss is a mean over the batch, so grad(B = 8) must equal the mean of the four B = 2 gradients (a
    size-independent property the CPU oracle cannot provide at this size); then three real optimizer steps."""
    import motion324_amd as m
    from motion324_amd import synth, training
    from motion324_amd.optim import FusedAdamW, backward_completion_order
    model = _full_model(12, train=True)
    s_np = synth.synth_inputs(8, 12, 4096, 4096, 224, seed=3, with_target=True)
    full = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    m.set_precision("bf16")
    try:
        loss8, out8, G8 = training.forward_backward(model, full)
        g8 = {n: G8.get(p).float().clone() for n, p in model.named_parameters() if p.requires_grad}
        acc = {n: torch.zeros_like(g) for n, g in g8.items()}
        losses2 = []
        for i in range(4):
            part = {k: v[2 * i:2 * i + 2].contiguous() for k, v in full.items()}
            l2, o2, G2 = training.forward_backward(model, part)
            losses2.append(float(l2))
            assert rel_err(o2, out8[2 * i:2 * i + 2]) < 2e-3          # samples are independent in the forward
            for n, p in model.named_parameters():
                if p.requires_grad:
                    acc[n] += G2.get(p).float() / 4
        torch.cuda.synchronize()
        assert float(loss8) == pytest.approx(sum(losses2) / 4, rel=1e-4)
        assert all(torch.isfinite(g).all() for g in g8.values())
        num = math.sqrt(sum(float((g8[n] - acc[n]).double().pow(2).sum()) for n in names))
        den = math.sqrt(sum(float(acc[n].double().pow(2).sum()) for n in names))
        assert num / den < 1e-2, num / den                             # bf16 GEMMs, different split-K / tile schedules
        worst = max((rel_err(g8[n], acc[n]), n) for n in names if acc[n].numel() >= 768 * 768)
        assert worst[0] < 0.1, worst                                   # every weight matrix on its own

        # three optimizer steps on the B = 8 batch: flat buffers, ONE m324_adamw_flat launch per step
        # (random-init weights give a first gradient norm of ~20: the reference's skip rule at 5 x clip would drop the step,
        # which the end of this test covers; here the rule is opened so that three real updates happen)
        # lr: the first steps of the reference's schedule (cosine with 1000 warm-up steps to 4e-4, training_utils.py:73-82);
        # the full 4e-4 on random-init weights overshoots
        opt = FusedAdamW(model.named_parameters(), lr=4e-6, betas=(0.9, 0.95), weight_decay=0.05, grad_clip_norm=1.0,
                         allowed_gradnorm_factor=1e9, order=backward_completion_order(model))
        assert opt.numel >= 157_000_000 and opt.n_decay > 0.99 * opt.numel and len(opt.buckets) >= 8
        losses = []
        for step in range(3):
            loss, _, G = training.forward_backward(model, full, sink=opt)
            opt.finish_reduce()
            info = opt.step()
            assert not info["skipped"] and math.isfinite(info["grad_norm"]), info
            losses.append(float(loss))
        # the reference's guard (train.py:198-201): a pre-clip norm above 5 x clip skips the update and leaves the weights alone
        strict = FusedAdamW(model.named_parameters(), lr=4e-4, grad_clip_norm=1e-3, allowed_gradnorm_factor=5.0,
                            order=backward_completion_order(model))
        before = strict.flat_param.clone()
        training.forward_backward(model, full, sink=strict)
        strict.finish_reduce()
        info = strict.step()
        assert info["skipped"] and torch.equal(before, strict.flat_param) and strict.step_count == 0
        torch.cuda.synchronize()
    finally:
        m.set_precision(None)
    assert losses[0] == pytest.approx(float(loss8), rel=1e-3)
    assert max(losses[1], losses[2]) < 0.8 * losses[0], losses           # Adam's sign-like first steps: down, not monotone
    assert all(torch.isfinite(p).all() for p in model.parameters())


# ------------------------------------------------------------------------------------------------------ c4 (per-GPU workload)
def test_c4_per_gpu_training_step_batch32():
    """configs[3]'s per-GPU workload: dyscene.yaml shapes at batch_size_per_gpu = 32 (reference configs/dyscene.yaml:21-56,
    README.md:115,124-125), bf16, ONE GPU (the 8-GPU gradient all-reduce is the driver's multi-GPU run; its bucketed launch is
    covered by the 1-rank RCCL and the 2-rank tests).  ~110 GB of kept block internals at this batch: the kept-vs-recompute
    budget of training.forward_backward is exercised at full scale.  Size-independent property: the loss is a batch mean, so
    grad(B = 32) equals the mean of the four B = 8 gradients; everything finite; one real optimizer step."""
    import motion324_amd as m
    from motion324_amd import synth, training
    from motion324_amd.optim import FusedAdamW, backward_completion_order
    model = _full_model(12, train=True)
    s_np = synth.synth_inputs(32, 12, 4096, 4096, 224, seed=5, with_target=True)
    full = {k: torch.from_numpy(v).cuda() for k, v in s_np.items()}
    del s_np
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    m.set_precision("bf16")
    try:
        loss32, out32, G32 = training.forward_backward(model, full)
        g32 = {n: G32.get(p).float().clone() for n, p in model.named_parameters() if p.requires_grad}
        assert out32.shape == (32, 12, 4096, 3) and torch.isfinite(out32).all() and math.isfinite(float(loss32))
        acc = {n: torch.zeros_like(g) for n, g in g32.items()}
        losses8 = []
        for i in range(4):
            part = {k: v[8 * i:8 * i + 8].contiguous() for k, v in full.items()}
            l8, o8, G8 = training.forward_backward(model, part)
            losses8.append(float(l8))
            assert rel_err(o8, out32[8 * i:8 * i + 8]) < 2e-3
            for n, p in model.named_parameters():
                if p.requires_grad:
                    acc[n] += G8.get(p).float() / 4
        torch.cuda.synchronize()
        assert float(loss32) == pytest.approx(sum(losses8) / 4, rel=1e-4)
        assert all(torch.isfinite(g).all() for g in g32.values())
        num = math.sqrt(sum(float((g32[n] - acc[n]).double().pow(2).sum()) for n in names))
        den = math.sqrt(sum(float(acc[n].double().pow(2).sum()) for n in names))
        assert num / den < 1e-2, num / den
        opt = FusedAdamW(model.named_parameters(), lr=4e-6, betas=(0.9, 0.95), weight_decay=0.05, grad_clip_norm=1.0,
                         allowed_gradnorm_factor=1e9, order=backward_completion_order(model))
        loss, _, _ = training.forward_backward(model, full, sink=opt)
        opt.finish_reduce()
        info = opt.step()
        assert not info["skipped"] and math.isfinite(info["grad_norm"]) and float(loss) == pytest.approx(float(loss32), rel=1e-3)
        torch.cuda.synchronize()
    finally:
        m.set_precision(None)
    assert all(torch.isfinite(p).all() for p in model.parameters())


# ------------------------------------------------------------------------------------------------------ c5
def _c5_available():
    return os.path.exists(os.path.join(GOLDEN, "c5.npz"))


@pytest.mark.skipif(not _c5_available(), reason="tests/golden/c5.npz not generated")
def test_c5_256_frame_clip_matches_reference_golden():
    """configs[4] on ONE GPU: 256 frames = 82 944 trunk tokens in one forward.  fp32 parity mode vs the golden the
    imported reference produced (tests/golden/make_golden.py c5; sampled mesh points, all frames), then the bf16 band."""
    gold = load_golden("c5")
    model, dm = build("c5")
    sample = inputs("c5", with_target=True)
    pts = torch.from_numpy(gold["pcd_points"]).cuda()
    ref = torch.from_numpy(gold["pcd_moved"])
    out, cap = run(model, sample, "fp32")
    errs = {}
    for k, v in cap.items():
        if "stage_" + k in gold:
            rows = torch.from_numpy(gold["rows_" + k]).to(v.device)
            errs[k] = rel_err(v.reshape(-1, v.shape[-1])[rows], torch.from_numpy(gold["stage_" + k]))
    errs["pcd_moved"] = rel_err(out.pcd_moved[:, :, pts], ref)
    print("[c5 fp32] " + "  ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert max(errs.values()) < FP32_TOL, errs
    assert abs(float(out.loss_metrics.loss) - float(gold["loss"])) <= 1e-3 * abs(float(gold["loss"]))
    del out, cap
    torch.cuda.empty_cache()
    sample.pop("point_clouds")
    outb, _ = run(model, sample, "bf16")
    eb = rel_err(outb.pcd_moved[:, :, pts], ref)
    print(f"[c5 bf16] pcd_moved={eb:.2e}")
    assert torch.isfinite(outb.pcd_moved).all() and eb < BF16_TOL, eb


def _fp_worker(rank, world, port, case, precision, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import motion324_amd as m
        model, dm = build(case)
        sample = inputs(case, with_target=False)
        m.set_precision(precision)
        with torch.no_grad():
            out = model.forward_frame_parallel(sample)
            torch.cuda.synchronize()
            # the same clip with this rank holding ONLY its frames, replayed as a chain of hipGraphs cut at the exchanges
            # (graph._Segmenter): must reproduce the eager run of the full-sample form bit for bit, twice
            import functools
            from motion324_amd import parallel
            from motion324_amd.graph import GraphedForward
            T_all = sample["rgb_video"].shape[1]
            mine = parallel.partition(T_all, world, rank)
            local = dict(sample)
            local["rgb_video"] = sample["rgb_video"][:, mine.start:mine.stop].contiguous()
            fp = functools.partial(model.forward_frame_parallel, local_frames=True, total_frames=T_all)
            chain = GraphedForward(model, forward=fp, segmented=True)
            c1 = chain(local).pcd_moved.clone()
            c2 = chain(local).pcd_moved.clone()
        torch.cuda.synchronize()
        ret[rank] = out.pcd_moved.cpu()
        ret[f"chain{rank}"] = bool(torch.equal(c1, out.pcd_moved)) and bool(torch.equal(c2, out.pcd_moved))
        ret[f"chain_diff{rank}"] = (float((c1 - out.pcd_moved).abs().max()), float((c2 - out.pcd_moved).abs().max()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,precision,tol,overlap", [(2, "fp32", 5e-6, "0"), (3, "bf16", 6e-3, "0"), (2, "bf16", 6e-3, "1")])
def test_frame_parallel_at_c2_size(world, precision, tol, overlap, monkeypatch):
    """Frame-parallel forward at the real trunk sizes (two global + two local blocks of the c2 clip's 10368 tokens): 32 frames over 2 ranks (16 + 16: the all_gather_into_tensor fast
    path, 5184 local / 10 368 global tokens) and over 3 ranks (11 + 11 + 10: uneven shards, padded gather) == the
    single-process forward (fp32: summation order only; bf16: the single-process run takes the fused transposed-V
    projection epilogue, the sharded one m324_qkv_split -- bf16 rounding apart).  overlap "1" (M324_KV_OVERLAP=1, opt-in): every
    global block attends to the rank's own 5184 / 3564 keys while the gather is in flight, then to the remote ranges (one or two),
    and merges by log-sum-exp -- eagerly and as the chain of hipGraphs with TWO cuts per global block; "0": one attention."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("M324_KV_OVERLAP", overlap)            # read at import by the spawned ranks
    # the trunk at its real length, everything else shallow (conftest CASES["c2_shallow"]): every rank builds its own model
    model, dm = build("c2_shallow")
    ref, _ = run(model, inputs("c2_shallow", with_target=False), precision)
    ref = ref.pcd_moved.cpu()
    del model
    torch.cuda.empty_cache()
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 21000 + (os.getpid() * 11 + world + 5 * int(overlap)) % 4000
    mp.spawn(_fp_worker, args=(world, port, "c2_shallow", precision, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r].shape == ref.shape
        assert rel_err(ret[r], ref) < tol, (r, rel_err(ret[r], ref))
    assert all(torch.equal(ret[0], ret[r]) for r in range(1, world))        # every rank holds the same complete result
    assert all(ret[f"chain{r}"] for r in range(world)), [ret[f"chain_diff{r}"] for r in range(world)]     # the graph chain with sharded frames == eager


# ------------------------------------------------------------------------------------------------------ c4 (2 ranks)
def _dp_worker(rank, world, port, mode, ret):
    """mode 'native': forward_backward with the optimizer as gradient sink (bucketed all-reduce on the side stream);
    mode 'ddp': the reference's own statement sequence around torch.nn.parallel.DistributedDataParallel(model)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import motion324_amd as m
        from motion324_amd import synth, training
        from motion324_amd.optim import FusedAdamW, backward_completion_order
        model, dm = build("tiny")
        model.train()
        s_np = synth.synth_inputs(2, 3, 30, 80, 64, seed=4, with_target=True)
        mine = {k: torch.from_numpy(v[rank:rank + 1]).cuda() for k, v in s_np.items()}      # this rank's sample
        if mode == "native":
            m.set_precision("fp32")
            opt = FusedAdamW(model.named_parameters(), lr=1e-3, allowed_gradnorm_factor=1e9,
                             order=backward_completion_order(model), bucket_mb=0.5)
            compute = torch.cuda.current_stream().cuda_stream
            loss, _, G = training.forward_backward(model, mine, sink=opt)
            early = list(opt.launch_log)                    # buckets that left DURING the backward
            opt.finish_reduce()
            info = opt.step()
            torch.cuda.synchronize()
            ret[rank] = dict(grad=opt.flat_grad.cpu(), names=list(opt.names), offsets=list(opt.offsets),
                             params={n: p.detach().cpu() for n, p in model.named_parameters() if p.requires_grad},
                             norm=info["grad_norm"], n_buckets=len(opt.buckets), early=len(early),
                             side_stream=all(sid != compute for _, sid in opt.launch_log))
        else:
            ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])           # train.py:88-89
            params = [p for p in ddp.parameters() if p.requires_grad]
            with torch.autocast(enabled=False, device_type="cuda"):
                ret_dict = ddp(mine)                                                          # train.py:150-155
            ret_dict.loss_metrics.loss.backward()                                             # train.py:166 (reducer hooks)
            torch.cuda.synchronize()
            ret[rank] = dict(grads={n: p.grad.detach().cpu() for n, p in ddp.module.named_parameters() if p.requires_grad},
                             loss=float(ret_dict.loss_metrics.loss), n=len(params))
    finally:
        dist.destroy_process_group()


def _single_process_reference():
    import motion324_amd as m
    from motion324_amd import synth, training
    model, dm = build("tiny")
    model.train()
    s_np = synth.synth_inputs(2, 3, 30, 80, 64, seed=4, with_target=True)
    m.set_precision("fp32")
    try:
        loss, _, G = training.forward_backward(model, {k: torch.from_numpy(v).cuda() for k, v in s_np.items()})
        torch.cuda.synchronize()
    finally:
        m.set_precision(None)
    return {n: G.get(p).float().cpu() for n, p in model.named_parameters() if p.requires_grad}, float(loss)


def test_bucketed_overlapped_allreduce_equals_single_process():
    """2 ranks x batch 1 with gradients written straight into the flat buffer and every bucket's all-reduce launched on
    the optimizer's side stream as soon as its tensors are final == 1 process x batch 2.  The launch log shows that
    buckets left before the backward ended and that none was enqueued on the compute stream."""
    import torch.multiprocessing as mp
    ref, _ = _single_process_reference()
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 25000 + (os.getpid() * 5) % 4000
    mp.spawn(_dp_worker, args=(2, port, "native", ret), nprocs=2, join=True)
    for r in range(2):
        d = ret[r]
        assert d["n_buckets"] >= 3 and d["early"] >= d["n_buckets"] - 1, (d["n_buckets"], d["early"])
        assert d["side_stream"]
        for n, o in zip(d["names"], d["offsets"]):
            g = d["grad"][o:o + ref[n].numel()].view(ref[n].shape)
            assert rel_err(g, ref[n]) < 1e-5 or float(ref[n].abs().max()) == 0.0, n
    a, b = ret[0]["params"], ret[1]["params"]
    assert all(torch.equal(a[n], b[n]) for n in a)                       # replicas stay bit-identical after the step


# ------------------------------------------------------------------------------------------------------ RCCL, one rank
def _rccl_worker(rank, port, ret):
    """torch.distributed backend "nccl" (= RCCL) on the one GPU of this box.  parallel.ALWAYS_COLLECT makes every
    collective of the multi-GPU paths run in the 1-rank group (identities), so the calls the driver's 8-GPU runs make --
    all_gather_into_tensor(async_op=True) + work.wait() on a side stream + record_stream, bucketed all_reduce(async_op)
    launched during the backward, all_gather of the window outputs -- go through RCCL streams here, which gloo never does."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import motion324_amd as m
        from motion324_amd import inference, parallel, synth, training
        from motion324_amd.optim import FusedAdamW, backward_completion_order
        res = {}
        one = torch.ones(4, device="cuda")
        dist.all_reduce(one)
        res["backend"] = dist.get_backend()
        res["allreduce_ok"] = bool((one == 1).all())
        # (1) frame-parallel forward: K/V all-gather per global block + final gather, both precisions
        model, dm = build("tiny")
        sample = inputs("tiny", with_target=True)
        for prec in ("fp32", "bf16"):
            m.set_precision(prec)
            with torch.no_grad():
                parallel.ALWAYS_COLLECT = False
                ref = model.forward_frame_parallel(sample)                 # no collective: plain forward
                parallel.ALWAYS_COLLECT = True
                got = model.forward_frame_parallel(sample)
            torch.cuda.synchronize()
            # the collective path splits q and k|v (m324_qkv_split instead of the fused epilogue): bf16 rounding apart
            res[f"fp_{prec}"] = (rel_err(got.pcd_moved, ref.pcd_moved), abs(float(got.loss_metrics.loss) - float(ref.loss_metrics.loss)))
        # (1b) the same forward as a CHAIN of hipGraphs cut at the exchanges (graph.py _Segmenter; collectives cannot be captured
        # on this stack), with the rank holding only its own frames (local_frames): replays equal the eager collective run
        import functools
        from motion324_amd.graph import GraphedForward
        m.set_precision("bf16")
        parallel.ALWAYS_COLLECT = True
        with torch.no_grad():
            eager = model.forward_frame_parallel(sample).pcd_moved.clone()
            T_all = sample["rgb_video"].shape[1]
            fp = functools.partial(model.forward_frame_parallel, local_frames=True, total_frames=T_all)
            chain = GraphedForward(model, forward=fp, segmented=True)
            r1 = chain(sample).pcd_moved.clone()
            r2 = chain(sample).pcd_moved.clone()
            seg = chain._graphs[chain._key(sample)][0]
            # B = 1: the even-shard path (one all_gather_into_tensor straight into the reused buffer)
            one_b = {k: v[:1].contiguous() for k, v in sample.items()}
            eager1 = model.forward_frame_parallel(one_b).pcd_moved.clone()
            c1 = chain(one_b).pcd_moved.clone()
        torch.cuda.synchronize()
        res["chain"] = (len(seg.graphs), len(seg.between), bool(torch.equal(r1, eager)), bool(torch.equal(r2, eager)))
        res["chain_b1"] = bool(torch.equal(c1, eager1))
        try:
            model.forward_frame_parallel(sample, local_frames=True)
            res["local_needs_total"] = False
        except Exception:
            res["local_needs_total"] = True
        m.set_precision(None)
        # (2) the sliding-window driver: windows gathered with all_gather over RCCL
        cfg = dict(training=dict(frames=dm.frames, use_amp=True))
        s = synth.synth_inputs(1, 7, 33, 70, 64, seed=6)
        video = torch.from_numpy(s["rgb_video"][0]).cuda()
        data = {k: torch.from_numpy(v).cuda() for k, v in s.items() if k != "rgb_video"}
        parallel.ALWAYS_COLLECT = False
        a = inference.run_model_inference(model, data, video, cfg, "cuda")
        parallel.ALWAYS_COLLECT = True
        b = inference.run_model_inference(model, data, video, cfg, "cuda")
        torch.cuda.synchronize()
        res["windows_equal"] = bool(torch.equal(a, b)) and tuple(b.shape) == (1, 7, 33, 3)
        # (3) training step: gradients written into the flat buffer, buckets all-reduced on the side stream during the backward
        def train(always):
            parallel.ALWAYS_COLLECT = always
            tm, _ = build("tiny")
            tm.train()
            sn = synth.synth_inputs(2, 3, 30, 80, 64, seed=4, with_target=True)
            smp = {k: torch.from_numpy(v).cuda() for k, v in sn.items()}
            opt = FusedAdamW(tm.named_parameters(), lr=1e-3, allowed_gradnorm_factor=1e9, order=backward_completion_order(tm),
                             bucket_mb=0.5)
            compute = torch.cuda.current_stream().cuda_stream
            m.set_precision("bf16")
            try:
                for _ in range(2):
                    loss, _, G = training.forward_backward(tm, smp, sink=opt)
                    early = len(opt.launch_log)
                    side = all(sid != compute for _, sid in opt.launch_log)
                    opt.finish_reduce()
                    info = opt.step()
                torch.cuda.synchronize()
            finally:
                m.set_precision(None)
            return ({n: p.detach().cpu() for n, p in tm.named_parameters() if p.requires_grad}, float(loss), info["grad_norm"],
                    early, len(opt.buckets), side)
        pa, la, na, _, _, _ = train(False)
        pb, lb, nb, early, nbuckets, side = train(True)
        res["train_equal"] = all(torch.equal(pa[n], pb[n]) for n in pa) and la == lb and na == nb
        res["buckets"] = (early, nbuckets, side)
        ret[0] = res
    finally:
        dist.destroy_process_group()


def test_multi_gpu_paths_over_rccl_with_one_rank():
    """The collectives of forward_frame_parallel, run_model_inference and the bucketed gradient all-reduce through
    torch.distributed "nccl" (RCCL) in a 1-rank group on this box's GPU (gloo never touches
    all_gather_into_tensor(async_op) + record_stream on RCCL streams).  Values must equal the collective-free runs."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 27000 + (os.getpid() * 3) % 4000
    mp.spawn(_rccl_worker, args=(port, ret), nprocs=1, join=True)
    r = ret[0]
    print(r)
    assert r["backend"] == "nccl" and r["allreduce_ok"]
    assert r["fp_fp32"][0] < 5e-6 and r["fp_fp32"][1] < 1e-6
    assert r["fp_bf16"][0] < BF16_TOL
    n_graphs, n_cuts, eq1, eq2 = r["chain"]
    assert n_cuts == 2 // 2 + 1 and n_graphs == n_cuts + 1, r["chain"]      # tiny: one global block (n_layer 2) + the output gather
    assert eq1 and eq2 and r["chain_b1"] and r["local_needs_total"]
    assert r["windows_equal"]
    assert r["train_equal"]
    early, nbuckets, side = r["buckets"]
    assert nbuckets >= 3 and early >= nbuckets - 1 and side


def test_torch_ddp_wrapper_runs_the_reference_sequence():
    """`DistributedDataParallel(model)` exactly as train.py:88-89 builds it (find_unused_parameters=False: every
    trainable parameter must receive a gradient in every step), 2 ranks x batch 1: the reducer-averaged .grad equals
    the single-process gradient of the concatenated batch."""
    import torch.multiprocessing as mp
    ref, ref_loss = _single_process_reference()
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 26000 + (os.getpid() * 5) % 4000
    mp.spawn(_dp_worker, args=(2, port, "ddp", ret), nprocs=2, join=True)
    assert (ret[0]["loss"] + ret[1]["loss"]) / 2 == pytest.approx(ref_loss, rel=1e-5)
    for r in range(2):
        assert ret[r]["n"] == len(ref)
        for n, g in ret[r]["grads"].items():
            assert rel_err(g, ref[n]) < 1e-5 or float(ref[n].abs().max()) == 0.0, n


def test_single_frame_clip_gives_every_parameter_a_gradient():
    """T = 1: special_token_rest is never read; autograd would deliver zeros, and so does the drop-in (a None would stall
    DDP's reducer, train.py:89)."""
    from motion324_amd import synth
    model, dm = build("tiny")
    model.train()
    s = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_inputs(1, 1, 20, 50, 64, seed=4, with_target=True).items()}
    model(s).loss_metrics.loss.backward()
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, n
    assert float(model.special_token_rest.grad.abs().max()) == 0.0
    assert float(model.special_token_0.grad.abs().max()) > 0.0


# ------------------------------------------------------------------------------------------------------ (f)1 driver
def _tiny4():
    """A tiny model built for 4-frame windows (the chunk goldens cover C = 4)."""
    import motion324_amd as m
    from motion324_amd import synth
    dims = dict(d=192, d_head=64, tokens=8, pcd_layers=1, n_layer=2, frames=4, dino_depth=2)
    dm = synth.Dims(**dims)
    cfg = synth.make_config(frames=4, d=192, tokens=8, pcd_layers=1, n_layer=2)
    cfg["model"]["dino"] = {"depth": 2}
    cfg["training"]["use_amp"] = False
    model = m.Motion_Latent_Model(cfg)
    sd = {k: torch.from_numpy(v) for k, v in synth_sd(dims).items()}
    model.load_state_dict(sd, strict=False)
    return model.eval().cuda(), sd, cfg, dm


def _driver_inputs(T):
    from motion324_amd import synth
    s = synth.synth_inputs(1, T, 24, 60, 64, seed=9)
    video = torch.from_numpy(s.pop("rgb_video"))[0]                    # [T, H, W, 3]
    return {k: torch.from_numpy(v) for k, v in s.items()}, video


def _expected_by_oracle(sd, dm, inp, video, T):
    """The oracle (CPU) run window by window, merged by the reference driver's golden index map."""
    from motion324_amd.inference import plan_windows
    from oracle import ref_forward as oracle
    gold = json.load(open(os.path.join(GOLDEN, "chunks.json")))[f"{T},4"]
    windows, out_map = plan_windows(T, 4)
    outs = []
    with torch.no_grad():
        for w in windows:
            sample = dict(inp)
            sample["rgb_video"] = video[w][None]
            outs.append(oracle.forward(sd, sample, frames=dm.frames)["pcd_moved"][0])
    frames = []
    for t, src in enumerate(gold):
        if src == -1:
            frames.append(inp["ref_pcd"][0])
        else:
            w, slot = out_map[t]
            assert windows[w][slot] == src                               # our plan == the reference driver's golden
            frames.append(outs[w][slot])
    return torch.stack(frames)[None]


@pytest.mark.parametrize("T", [5, 7, 30])
def test_sliding_window_driver_with_the_real_model(T):
    """run_model_inference (scripts/inference_with_video_mesh.py:132-256) on the HIP model for T = frames + 1,
    2 * frames - 1 and 30 with 4-frame windows vs the CPU oracle driven through the golden plan."""
    from motion324_amd.inference import run_model_inference
    model, sd, cfg, dm = _tiny4()
    inp, video = _driver_inputs(T)
    want = _expected_by_oracle(sd, dm, inp, video, T)
    got = run_model_inference(model, {k: v.cuda() for k, v in inp.items()}, video, cfg, "cuda")
    assert got.shape == (1, T, 24, 3) and got.dtype == torch.float32
    assert rel_err(got, want) < FP32_TOL


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_pipelined_driver_equals_the_plain_loop_bit_for_bit(precision):
    """The engineered driver -- window w + 1's frames on a copy stream under window w's forward, the shape encoder's tokens and
    the anchor frame's image tokens computed once per video, hipGraph replay from the third window of a shape on -- against the
    reference's literal loop (one synchronous upload per window, everything recomputed): the SAME trajectories, bit for bit, from
    pinned, pageable and device-resident videos; and without the reuse."""
    import motion324_amd as m
    from motion324_amd.inference import run_model_inference
    T = 30
    model, sd, cfg, dm = _tiny4()
    inp, video = _driver_inputs(T)
    inp = {k: v.cuda() for k, v in inp.items()}
    m.set_precision(precision)
    try:
        model.auto_graph = False
        plain = run_model_inference(model, inp, video, cfg, "cuda", pipelined=False)
        model.auto_graph = True
        for src in (video, video.pin_memory(), video.cuda()):
            got = run_model_inference(model, inp, src, cfg, "cuda")
            assert torch.equal(got, plain), float((got - plain).abs().max())
        assert model.__dict__.get("_ag") is not None and len(model._ag._graphs) >= 1      # the later windows were graph replays
        got = run_model_inference(model, inp, video, cfg, "cuda", reuse=False)
        assert torch.equal(got, plain)
        # a second video through the same model (graphs and static inputs are reused; the kept tokens are per video)
        inp2, video2 = _driver_inputs(T)
        video2 = video2.flip(0).contiguous()
        model.auto_graph = False
        plain2 = run_model_inference(model, inp, video2, cfg, "cuda", pipelined=False)
        model.auto_graph = True
        assert torch.equal(run_model_inference(model, inp, video2.pin_memory(), cfg, "cuda"), plain2)
        assert not torch.equal(plain2, plain)
        # videos back to back WITHOUT a synchronisation in between: the second video's first upload is issued while the first video's
        # last windows are still queued -- the staging buffers' events order them (bench.py's long_video_driver row is the full-size form)
        pa, pb = video.pin_memory(), video2.pin_memory()
        results = [run_model_inference(model, inp, v, cfg, "cuda") for v in (pa, pb, pa, pb)]
        assert all(torch.equal(r, want) for r, want in zip(results, (plain, plain2, plain, plain2)))
    finally:
        m.set_precision(None)


def test_byte_frames_equal_their_float_form_bit_for_bit():
    """uint8 frames (a quarter of the upload) are converted by m324_patchify_u8 as v / 255 per tap: the trajectories equal those
    of `video.float() / 255` exactly -- plain loop, pipelined driver, and the model called directly."""
    from motion324_amd import ops
    from motion324_amd.inference import run_model_inference
    T = 9
    model, sd, cfg, dm = _tiny4()
    inp, _ = _driver_inputs(T)
    inp = {k: v.cuda() for k, v in inp.items()}
    g = torch.Generator().manual_seed(5)
    vid8 = torch.randint(0, 256, (T, 50, 70, 3), generator=g, dtype=torch.uint8)
    vidf = vid8.float() / 255.0
    a = ops.patchify(vid8.cuda(), 224, 14, 640, torch.float32)
    b = ops.patchify(vidf.cuda(), 224, 14, 640, torch.float32)
    assert torch.equal(a, b)
    want = run_model_inference(model, inp, vidf, cfg, "cuda", pipelined=False)
    assert torch.equal(run_model_inference(model, inp, vid8, cfg, "cuda", pipelined=False), want)
    assert torch.equal(run_model_inference(model, inp, vid8.pin_memory(), cfg, "cuda"), want)
    assert torch.equal(run_model_inference(model, inp, vid8, cfg, "cuda"), want)
    with torch.no_grad():
        s8, sf = dict(inp, rgb_video=vid8[None, :4].cuda()), dict(inp, rgb_video=vidf[None, :4].cuda())
        assert torch.equal(model(s8).pcd_moved, model(sf).pcd_moved)


def test_window_reuse_keys_are_refused_where_they_do_not_apply():
    from motion324_amd.lib import M324Error
    model, sd, cfg, dm = _tiny4()
    inp, video = _driver_inputs(4)
    s = {k: v.cuda() for k, v in inp.items()}
    s["rgb_video"] = video[None].cuda()
    with torch.no_grad():
        out = model(dict(s, m324_keep_reuse=True))
        assert out.reuse.mesh_tokens.shape == (8, 192) and out.reuse.anchor_tokens.shape == (257, 192)
        later = dict(s, rgb_video=s["rgb_video"][:, 1:], m324_mesh_tokens=out.reuse.mesh_tokens, m324_anchor_tokens=out.reuse.anchor_tokens)
        assert torch.equal(model(later).pcd_moved, out.pcd_moved)               # the same window, anchor and mesh handed in
        with pytest.raises(M324Error, match="m324_anchor_tokens"):
            model(dict(later, m324_anchor_tokens=out.reuse.anchor_tokens[:100]))


def _driver_worker(rank, world, port, T, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from motion324_amd.inference import run_model_inference
        model, sd, cfg, dm = _tiny4()
        inp, video = _driver_inputs(T)
        ret[rank] = run_model_inference(model, {k: v.cuda() for k, v in inp.items()}, video, cfg, "cuda").cpu()
    finally:
        dist.destroy_process_group()


def test_sliding_window_driver_world2_shards_the_windows():
    import torch.multiprocessing as mp
    from motion324_amd.inference import run_model_inference
    T = 30
    model, sd, cfg, dm = _tiny4()
    inp, video = _driver_inputs(T)
    single = run_model_inference(model, {k: v.cuda() for k, v in inp.items()}, video, cfg, "cuda").cpu()
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 28000 + (os.getpid() * 3) % 1500
    mp.spawn(_driver_worker, args=(2, port, T, ret), nprocs=2, join=True)
    assert torch.equal(ret[0], single) and torch.equal(ret[1], single)


# ------------------------------------------------------------------------------------------------------ (f)3 checkpoints
def test_checkpoint_roundtrip_on_the_device(tmp_path):
    """save_checkpoint -> a fresh model built for ANOTHER clip length -> load_checkpoint -> identical forward to a model
    that received the same weights through load_state_dict; pos_embed is regenerated, not loaded
    (utils/inference_utils.py:36-49).  A GraphedForward of the live target model must notice the new weights."""
    import motion324_amd as m
    from motion324_amd import checkpoint as ck
    src, _ = build("tiny")                                               # frames = 3
    with torch.no_grad():
        for p in src.parameters():
            p.add_(0.01 * torch.randn_like(p))
    path = ck.save_checkpoint(str(tmp_path), src, param_update_step=60000, fwdbwd_pass_step=60000)
    dst, _ = build("tiny_resize")                                        # frames = 5
    sample = inputs("tiny_resize", with_target=False)
    m.set_precision("bf16")
    try:
        fast = m.GraphedForward(dst)
        before = fast(sample).pcd_moved.clone()
        info = ck.load_checkpoint(path, dst, "cuda")
        after = fast(sample).pcd_moved.clone()                           # same graph object, live model, new weights
    finally:
        m.set_precision(None)
    assert info == {"fwdbwd_pass_step": 60000, "param_update_step": 60000}
    assert not torch.equal(before, after)
    twin, _ = build("tiny_resize")
    state = {k: v for k, v in src.state_dict().items() if k != "pos_embed"}
    twin.load_state_dict(state, strict=False)
    want, _ = run(twin, sample, "bf16")
    assert torch.equal(after, want.pcd_moved)
    assert dst.pos_embed.shape[1] == 5 * 256 and torch.equal(dst.pos_embed, twin.pos_embed)


def test_optimizer_state_roundtrip_resumes_bit_identically(tmp_path):
    """FusedAdamW.state_dict() uses torch.optim.AdamW's layout (utils/training_utils.py:38-52 numbering): a run resumed
    from a checkpoint continues exactly like the uninterrupted one, and torch's own AdamW accepts the dict."""
    import motion324_amd as m
    from motion324_amd import checkpoint as ck, synth, training
    from motion324_amd.optim import FusedAdamW, backward_completion_order
    s = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_inputs(1, 3, 30, 80, 64, seed=2, with_target=True).items()}

    def make():
        model, _ = build("tiny")
        model.train()
        return model, FusedAdamW(model.named_parameters(), lr=1e-3, allowed_gradnorm_factor=1e9,
                                 order=backward_completion_order(model))

    def step(model, opt):
        loss, _, G = training.forward_backward(model, s, sink=opt)
        opt.finish_reduce()
        opt.step()
        return float(loss)
    m.set_precision("fp32")
    try:
        a, oa = make()
        for _ in range(2):
            step(a, oa)
        path = ck.save_checkpoint(str(tmp_path), a, 2, 2, optimizer=oa)
        l3 = step(a, oa)
        b, ob = make()
        ck.load_checkpoint(path, b, "cuda", optimizer=ob)
        assert ob.step_count == 2
        l3b = step(b, ob)
        torch.cuda.synchronize()
    finally:
        m.set_precision(None)
    assert l3 == l3b
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    assert all(torch.equal(pa[n], pb[n]) for n in pa)
    # The dict is torch.optim.AdamW's with the numbering of the reference's create_optimizer (utils/training_utils.py:38-52:
    # named_parameters() order, decay group first) -- although this optimizer's LAYOUT is sorted by gradient-completion
    # order.  Build torch's optimizer exactly as the reference does (no names involved) and check every tensor's moments
    # (the trunk blocks have identical shapes: a swapped numbering would pass any shape check).
    sd = torch.load(path, weights_only=True)["optimizer"]
    decay = [p for n, p in b.named_parameters() if p.requires_grad and not (p.dim() == 1 or getattr(p, "_no_weight_decay", False))]
    nodecay = [p for n, p in b.named_parameters() if p.requires_grad and (p.dim() == 1 or getattr(p, "_no_weight_decay", False))]
    topt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.05}, {"params": nodecay, "weight_decay": 0.0}],
                             lr=1e-3, betas=(0.9, 0.95))
    topt.load_state_dict({"state": sd["state"], "param_groups": sd["param_groups"]})
    assert ob.names != ob.ckpt_names                        # the layout really is re-sorted
    oc = FusedAdamW(b.named_parameters(), lr=1e-3, allowed_gradnorm_factor=1e9, order=backward_completion_order(b), flatten=False)
    oc.load_state_dict(sd)
    for p in decay + nodecay:
        st = topt.state[p]
        assert float(st["step"]) == 2.0
        i = oc._index[id(p)]
        o, n = oc.offsets[i], p.numel()
        assert torch.equal(st["exp_avg"].reshape(-1), oc.m[o:o + n]) and torch.equal(st["exp_avg_sq"].reshape(-1), oc.v[o:o + n])
    # and back: a dict written by torch's own AdamW (no names, the reference's numbering) lands on the right tensors
    od = FusedAdamW(b.named_parameters(), lr=1e-3, allowed_gradnorm_factor=1e9, order=backward_completion_order(b), flatten=False)
    od.load_state_dict(topt.state_dict())
    assert od.step_count == 2 and torch.equal(od.m, oc.m) and torch.equal(od.v, oc.v)


# ------------------------------------------------------------------------------------------------------ C-ABI collectives
def test_c_abi_collectives_over_rccl_single_rank():
    """m324_comm_* (include/m324.h) on the one GPU of this box: a 1-rank RCCL communicator -- init from the hex id,
    all-reduce (sum and mean are the identity), all-gather (a copy), destroy.  Multi-rank runs need one GPU per rank."""
    from motion324_amd.comm import Communicator
    uid = Communicator.unique_id()
    assert len(uid) == 256 and int(uid, 16) >= 0
    comm = Communicator(uid, 0, 1)
    try:
        g = torch.randn(1 << 20, device="cuda")
        want = g.clone()
        comm.all_reduce(g, average=True)
        comm.all_reduce(g, average=False)
        kv = torch.randn(324, 1536, device="cuda").bfloat16()
        out = torch.empty_like(kv)
        comm.all_gather(kv, out)
        torch.cuda.synchronize()
        assert torch.equal(g, want) and torch.equal(out, kv)
    finally:
        comm.close()
