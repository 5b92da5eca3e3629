"""GPU parity of the training-side kernels and block backward passes against torch autograd on the CPU (fp64)."""
import math

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = [torch.float32, torch.bfloat16]


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


def _q(t, dtype):
    return t.to(dtype).to(torch.float32)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("R,C", [(64, 64), (100, 192), (324, 768), (1, 70)])
def test_transpose_pads_with_zeros(dtype, R, C):
    from motion324_amd import ops
    x = _q(_rand((R, C), 1), dtype)
    out = ops.transpose(x.to(dtype).to(DEV)).float().cpu()
    Rp = (R + 63) // 64 * 64
    assert out.shape == (C, Rp)
    assert torch.equal(out[:, :R], x.T) and float(out[:, R:].abs().max() if Rp > R else 0.0) == 0.0


@pytest.mark.parametrize("dtype", DT)
def test_colsum_and_accumulate(dtype):
    from motion324_amd import ops
    x = _q(_rand((777, 200), 2), dtype)
    d = x.to(dtype).to(DEV)
    s = ops.colsum(d)
    assert rel_err(s, x.double().sum(0)) < 1e-5
    ops.colsum(d, out=s, accumulate=True)
    assert rel_err(s, 2 * x.double().sum(0)) < 1e-5


def test_colsum_multi_is_the_sequence_of_single_column_sums():
    """m324_colsum_multi (ABI 20): many (destination, chain of fp32 row blocks) pairs in one launch -- against the same sums done
    one m324_colsum call after the other: bit-identical for the wide form (<= 64 rows per source: the split-K partials of the weight
    gradients), equal to fp64 sums within fp32 rounding for the tall form (per-workgroup partials of a LayerNorm backward), with
    unaligned column counts, strided sources (column slices of one partial buffer), accumulation into a destination that already
    holds a value, and more items than one launch's table takes."""
    from motion324_amd import ops
    g = torch.Generator().manual_seed(5)
    q = ops._ColsumQueue()
    want, want64, dsts = [], [], []
    big = torch.randn((700, 3 * 192), generator=g).to(DEV)                  # a LayerNorm-backward partial: three column groups
    specs = [(7, 2304 * 768 // 64, False, 1), (28, 4096, True, 1), (3, 1000, False, 3), (62, 64, True, 2), (1, 20, False, 1), (5, 333, True, 2)]
    specs += [(2 + k % 5, 128 + 4 * k, bool(k % 2), 1 + k % 3) for k in range(40)]      # > 64 items in all
    for rows, cols, acc, nchain in specs:
        dst = torch.randn((cols,), generator=g).to(DEV)
        ref = dst.clone()
        r64 = dst.double().cpu() if acc else torch.zeros(cols, dtype=torch.float64)
        for c in range(nchain):
            src = torch.randn((rows + c, cols), generator=g).to(DEV)
            q.defer(src, dst, acc or c > 0)
            ops.colsum(src, out=ref, accumulate=acc or c > 0)
            r64 = r64 + src.double().cpu().sum(0)
        want.append(ref), want64.append(r64), dsts.append(dst)
    tall = []
    for k in range(3):
        dst = torch.zeros((192,), device=DEV)
        q.defer(big[:, k * 192:(k + 1) * 192], dst, False)
        tall.append((dst, big[:, k * 192:(k + 1) * 192].double().cpu().sum(0)))
    assert q.count == sum(n for *_x, n in specs) + 3 > 64
    q.flush()
    torch.cuda.synchronize()
    for dst, ref, r64 in zip(dsts, want, want64):
        assert torch.equal(dst, ref), float((dst - ref).abs().max())
        assert rel_err(dst, r64) < 1e-5
    for dst, r64 in tall:
        assert rel_err(dst, r64) < 1e-5
    with pytest.raises(Exception, match="must accumulate"):
        q.defer(torch.zeros((2, 8), device=DEV), dsts[0][:8].contiguous(), False)
        q.defer(torch.zeros((2, 8), device=DEV), dsts[0][:8].contiguous(), False)


def test_queued_column_sums_give_the_same_training_step(monkeypatch):
    """M324_DEFER_COLSUM: the sums of the weight gradients' split-K partials and of the norm-weight partials leave in one launch
    per block instead of one m324_colsum each.  Same step: loss and output bit for bit, every weight gradient whose partials
    are few rows bit for bit, the rest (tall partials: another summation order) to fp32 rounding."""
    import motion324_amd as m
    from motion324_amd import ops, synth, training
    from motion324_amd.optim import FusedAdamW, backward_completion_order
    from conftest import CASES, synth_sd
    dims, (B, T, N, S, HW) = CASES["tiny"]["dims"], CASES["tiny"]["shape"]
    dm = synth.Dims(**dims)
    cfg = synth.make_config(frames=dm.frames, d=dm.d, d_head=dm.d_head, tokens=dm.tokens, pcd_layers=dm.pcd_layers, n_layer=dm.n_layer)
    cfg["model"]["dino"] = {"depth": dm.dino_depth}
    model = m.Motion_Latent_Model(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_sd(dims).items()}, strict=False)
    model = model.train().to(DEV)
    model.drop_rate = 0.0
    sample = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synth_inputs(B, T, N, S, HW, seed=1, with_target=True).items()}
    res = {}
    m.set_precision("bf16")
    try:
        for defer in (False, True):
            monkeypatch.setattr(ops, "DEFER_COLSUM", defer)
            for with_sink in (False, True):
                opt = FusedAdamW(model.named_parameters(), lr=1e-3, order=backward_completion_order(model)) if with_sink else None
                loss, out, G = training.forward_backward(model, sample, sink=opt)
                assert ops.COLSUMS.count == 0                                   # the last done() flushed the queue
                res[(defer, with_sink)] = (float(loss), out.clone(), {n: G.get(p).clone() for n, p in model.named_parameters() if p.requires_grad})
    finally:
        m.set_precision(None)
    for with_sink in (False, True):
        l0, o0, g0 = res[(False, with_sink)]
        l1, o1, g1 = res[(True, with_sink)]
        assert l0 == l1 and torch.equal(o0, o1)
        worst = max(rel_err(g1[n], g0[n]) for n in g0)
        exact = sum(int(torch.equal(g1[n], g0[n])) for n in g0)
        print(f"[queued column sums, sink={with_sink}] {exact} of {len(g0)} gradients bit-identical, worst {worst:.2e}")
        assert worst < 1e-5 and exact >= len(g0) // 2


def test_weight_mirror_gives_the_same_training_steps(monkeypatch):
    """optim.FusedAdamW's bf16 weight mirror (one m324_weight_mirror launch per update instead of a cast per weight and a
    m324_transpose per dgrad operand): every copy equals torch's cast / its transpose bit for bit, Prepared hands the mirror's
    views out exactly while they are current, and three steps give the same losses and parameters with and without it."""
    import motion324_amd as m
    from motion324_amd import optim, prepared, synth, training
    from motion324_amd.optim import FusedAdamW, backward_completion_order
    from conftest import CASES, synth_sd
    dims, (B, T, N, S, HW) = CASES["tiny"]["dims"], CASES["tiny"]["shape"]
    dm = synth.Dims(**dims)
    cfg = synth.make_config(frames=dm.frames, d=dm.d, d_head=dm.d_head, tokens=dm.tokens, pcd_layers=dm.pcd_layers, n_layer=dm.n_layer)
    cfg["model"]["dino"] = {"depth": dm.dino_depth}
    sample = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synth_inputs(B, T, N, S, HW, seed=1, with_target=True).items()}
    res = {}
    m.set_precision("bf16")
    try:
        for mirror in (False, True):
            monkeypatch.setattr(optim, "WEIGHT_MIRROR", mirror)
            model = m.Motion_Latent_Model(cfg)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_sd(dims).items()}, strict=False)
            model = model.train().to(DEV)
            model.drop_rate = 0.0
            opt = FusedAdamW(model.named_parameters(), lr=1e-3, allowed_gradnorm_factor=1e9, order=backward_completion_order(model))
            P = prepared.Prepared.for_module(model, DEV, torch.bfloat16)
            losses = []
            for _ in range(3):
                loss, _, _ = training.forward_backward(model, sample, sink=opt)
                opt.finish_reduce()
                assert not opt.step()["skipped"]
                losses.append(float(loss))
            if mirror:
                w = model.global_transformer_blocks[0].attn.to_qkv.weight
                assert len(opt._mirror_params) > 10 and any(p is w for p in opt._mirror_params)
                for p in opt._mirror_params:
                    got, got_t = P.mat(p), P.mat_t(p, None)                     # (None: a call of the fallback would raise)
                    assert got.data_ptr() >= opt._mirror.data_ptr() and torch.equal(got, p.detach().to(torch.bfloat16))
                    n = p.shape[0]
                    assert got_t.shape == (p.shape[1], (n + 63) // 64 * 64) and torch.equal(got_t[:, :n], got.t()) and not got_t[:, n:].any()
                ca = model.decoder_cross_attn.attn
                kv = P.cat_rows((ca.to_k.weight, ca.to_v.weight))
                assert kv.data_ptr() == P.mat(ca.to_k.weight).data_ptr()      # neighbours in the flat buffer: one view, no copy
                assert torch.equal(kv, torch.cat([ca.to_k.weight, ca.to_v.weight]).to(torch.bfloat16))
                with torch.no_grad():
                    w.mul_(1.0)                                                 # a tracked in-place edit: the mirror is stale ...
                assert P.mat(w).data_ptr() != got.data_ptr() or w is not p
                assert not (opt._mirror.data_ptr() <= P.mat(w).data_ptr() < opt._mirror.data_ptr() + opt._mirror.numel() * 2)
                opt.sync_mirror()                                               # ... until the optimizer rewrites it
                assert opt._mirror.data_ptr() <= P.mat(w).data_ptr() < opt._mirror.data_ptr() + opt._mirror.numel() * 2
            res[mirror] = (losses, {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad})
    finally:
        m.set_precision(None)
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    for n, p in res[False][1].items():
        assert torch.equal(p, res[True][1][n]), n


@pytest.mark.parametrize("dtype", DT)
def test_gelu_forward_backward(dtype):
    from motion324_amd import ops
    z = _q(_rand((1000, 64), 3, 2.0), dtype)
    dh = _q(_rand((1000, 64), 4), dtype)
    zt = z.double().requires_grad_(True)
    h = torch.nn.functional.gelu(zt)
    h.backward(dh.double())
    tol = 1e-6 if dtype == torch.float32 else 5e-3
    assert rel_err(ops.gelu(z.to(dtype).to(DEV)).float(), h.detach()) < tol
    assert rel_err(ops.gelu_bwd(z.to(dtype).to(DEV), dh.to(dtype).to(DEV)).float(), zt.grad) < tol


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,C", [(77, 768), (5000, 192), (3, 768)])
def test_layernorm_backward(dtype, rows, C):
    from motion324_amd import ops
    x = _rand((rows, C), 5) * 2 + 0.3
    w = 1 + 0.1 * _rand((C,), 6)
    dy = _q(_rand((rows, C), 7), dtype)
    xt, wt = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bt = torch.zeros(C, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.layer_norm(xt, (C,), wt, bt, 1e-5).backward(dy.double())
    dx0 = _rand((rows, C), 8)
    dx = dx0.clone().to(DEV)
    dw, db = ops.layernorm_bwd(x.to(DEV), w.to(DEV), 1e-5, dy.to(dtype).to(DEV), dx, accumulate=True)
    assert rel_err(dx, dx0.double() + xt.grad) < 2e-6
    assert rel_err(dw, wt.grad) < 2e-5 and rel_err(db, bt.grad) < 2e-5
    dx2 = torch.full((rows, C), float("nan"), device=DEV)
    ops.layernorm_bwd(x.to(DEV), w.to(DEV), 1e-5, dy.to(dtype).to(DEV), dx2, accumulate=False)
    assert rel_err(dx2, xt.grad) < 2e-6


@pytest.mark.parametrize("rows,C,row_map", [(517, 768, (0, 0, 0)), (64, 192, (0, 0, 0)), (3 * 8, 768, (8, 40, 4)), (4100, 768, (0, 0, 0))])
def test_layernorm_backward_with_bf16_copy_and_column_sums(rows, C, row_map):
    """m324_layernorm_bwd_cast = m324_layernorm_bwd + m324_cast + m324_colsum in one pass: the same dx / dw / db bit for bit, the
    bf16 copy equal to the cast of the resulting dx, the third vector equal to the column sums of that copy (to the order of an
    fp32 sum).  Also with accumulate and with a row map (the decoder's k|v rows inside the trunk's token matrix)."""
    from motion324_amd import ops
    gin, gout, off = row_map
    xrows = rows if not gin else (rows // gin) * gout
    x = (_rand((xrows, C), 5) * 2 + 0.3).to(DEV)
    w = (1 + 0.1 * _rand((C,), 6)).to(DEV)
    dy = _q(_rand((rows, C), 7), torch.bfloat16).to(torch.bfloat16).to(DEV)
    dx0 = _rand((xrows, C), 8).to(DEV)
    a, b = dx0.clone(), dx0.clone()
    dw0, db0 = ops.layernorm_bwd(x, w, 1e-5, dy, a, accumulate=True, row_map=row_map)
    copy = torch.full((xrows, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    dw1, db1, cs = ops.layernorm_bwd(x, w, 1e-5, dy, b, accumulate=True, row_map=row_map, cast_out=copy)
    assert torch.equal(a, b) and torch.equal(dw0, dw1) and torch.equal(db0, db1)
    touched = torch.zeros(xrows, dtype=torch.bool, device=DEV)
    r = torch.arange(rows, device=DEV)
    touched[(r // gin) * gout + r % gin + off if gin else r] = True
    assert torch.equal(copy[touched], b[touched].to(torch.bfloat16))
    assert bool(torch.isnan(copy[~touched].float()).all()) or not bool((~touched).any())
    ref = copy[touched].double().sum(0)
    assert float((cs.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max() + 1)


def _head_major(t, B, L, H):
    return t.reshape(B, L, H, 64).permute(0, 2, 1, 3).contiguous()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,Lq,Lk,shared", [(2, 3, 70, 70, False), (1, 2, 33, 200, False), (3, 2, 50, 64, True)])
def test_attention_backward_against_autograd(dtype, B, H, Lq, Lk, shared):
    """forward (with saved LSE) + m324_attention_delta + m324_attention_bwd vs autograd of softmax(q k^T / 8) v."""
    from motion324_amd import ops
    from conftest import vt_layout
    Bq = 1 if shared else B
    qhat = _rand((Bq, H, Lq, 64), 11, 1.2)                                 # "normalised, unscaled" q
    k, v = _q(_rand((B, H, Lk, 64), 12, 1.2), dtype), _q(_rand((B, H, Lk, 64), 13), dtype)
    qs = _q(qhat * ops.Q_PRESCALE, dtype)                                  # what qkv_split stores
    dO_tok = _q(_rand((B * Lq, H * 64), 14), dtype)
    # reference in fp64 on the stored (rounded) operands; gradient w.r.t. qhat = qs / prescale
    qh = (qs.double() / ops.Q_PRESCALE).requires_grad_(True)
    kd, vd = k.double().requires_grad_(True), v.double().requires_grad_(True)
    s = torch.einsum("bhqd,bhkd->bhqk", qh.expand(B, -1, -1, -1), kd) * (64 ** -0.5)
    o = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(s, -1), vd).reshape(B * Lq, H * 64)
    o.backward(dO_tok.double())

    dev = lambda t: t.to(dtype).to(DEV)
    out = torch.empty((B * Lq, H * 64), dtype=dtype, device=DEV)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=DEV)
    ops.attention(dev(qs), dev(k), dev(vt_layout(v)), out, shared_q=shared, prescaled=True, lse=lse)
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    assert rel_err(out.float(), o.detach()) < tol
    ref_lse = torch.logsumexp(s.detach(), -1) / math.log(2.0)
    assert rel_err(lse, ref_lse) < 1e-5 if dtype == torch.float32 else rel_err(lse, ref_lse) < 1e-3
    D = ops.attention_delta(out, dev(dO_tok), B, H, Lq)
    dO_hm = _head_major(dO_tok, B, Lq, H)
    dQ, dK, dV = ops.attention_bwd(dev(qs), dev(k), dev(v), dev(dO_hm), lse, D, shared_q=shared)
    gtol = 2e-5 if dtype == torch.float32 else 1.5e-2
    dq_ref = qh.grad if not shared else qh.grad                              # autograd already summed over batches
    got_dq = dQ.float().cpu().double()
    if shared:
        got_dq = got_dq.sum(0, keepdim=True)
    assert rel_err(got_dq, dq_ref) < gtol
    assert rel_err(dK.float(), kd.grad) < gtol and rel_err(dV.float(), vd.grad) < gtol


@pytest.mark.parametrize("dtype", DT)
def test_qkv_split_train_outputs_and_backward(dtype):
    from motion324_amd import ops
    from conftest import vt_layout
    B, L, H = 2, 100, 3
    C = H * 64
    qkv = _q(_rand((B * L, 3 * C), 15), dtype)
    qw, kw = 1 + 0.1 * _rand((64,), 16), 1 + 0.1 * _rand((64,), 17)
    d = qkv.to(dtype).to(DEV)
    outs = ops.qkv_split(d[:, :C], d[:, C:2 * C], d[:, 2 * C:], qw.to(DEV), kw.to(DEV), 1e-5, B, L, H, dtype, q_scale=0.25,
                         train=True)
    # transposed copies are exactly vt_layout of the row-major ones
    for a, b in (("Q", "Qt"), ("K", "Kt"), ("V", "Vt")):
        assert torch.equal(outs[b].float().cpu(), vt_layout(outs[a].float().cpu()))
    # backward vs autograd of rmsnorm
    x = qkv.double().requires_grad_(True)
    q, k, v = (t.reshape(B, L, H, 64) for t in x.chunk(3, -1))
    qwd, kwd = qw.double().requires_grad_(True), kw.double().requires_grad_(True)
    qn = q * torch.rsqrt((q * q).mean(-1, keepdim=True) + 1e-5) * qwd
    kn = k * torch.rsqrt((k * k).mean(-1, keepdim=True) + 1e-5) * kwd
    gq, gk, gv = (_q(_rand((B, H, L, 64), s), dtype) for s in (18, 19, 20))
    (qn.permute(0, 2, 1, 3) * gq.double()).sum().backward(retain_graph=True)
    (kn.permute(0, 2, 1, 3) * gk.double()).sum().backward(retain_graph=True)
    (v.permute(0, 2, 1, 3) * gv.double()).sum().backward()
    dqkv = torch.zeros((B * L, 3 * C), dtype=dtype, device=DEV)
    dqw, dkw = ops.qkv_split_bwd(gq.to(dtype).to(DEV), gk.to(dtype).to(DEV), gv.to(dtype).to(DEV), d[:, :C], d[:, C:2 * C],
                                 qw.to(DEV), kw.to(DEV), 1e-5, B, L, H, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:])
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    assert rel_err(dqkv.float(), x.grad) < tol
    assert rel_err(dqw, qwd.grad) < 1e-4 and rel_err(dkw, kwd.grad) < 1e-4


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,K", [(999, 192), (5000, 768), (3, 64), (700, 2048), (130, 52)])
def test_linear_n3_backward(dtype, M, K):
    """(K = 52: the scalar kernel; the others the 16-byte one: 24 / 96 / 8 / 256 chunks per row, 10 / 2 / 32 / 1 rows side by side)"""
    from motion324_amd import ops
    a, w = _q(_rand((M, K), 21), dtype), _rand((3, K), 22, 0.1)
    dout = _rand((M, 3), 23)
    at, wt = a.double().requires_grad_(True), w.double().requires_grad_(True)
    bt = torch.zeros(3, dtype=torch.float64, requires_grad=True)
    (at @ wt.T + bt).backward(dout.double())
    dA, dW, db = ops.linear_n3_bwd(a.to(dtype).to(DEV), w.to(DEV), dout.to(DEV))
    assert rel_err(dA.float(), at.grad) < (1e-6 if dtype == torch.float32 else 5e-3)
    assert rel_err(dW, wt.grad) < 1e-5 and rel_err(db, bt.grad) < 1e-5
    f = _q(_rand((M, K), 24), dtype)                         # optional factor on dA (the gelu'(z) of the Linear in front)
    dA2, dW2, db2 = ops.linear_n3_bwd(a.to(dtype).to(DEV), w.to(DEV), dout.to(DEV), mul_by=f.to(dtype).to(DEV))
    assert rel_err(dA2.float(), at.grad * f.double()) < (1e-6 if dtype == torch.float32 else 5e-3)
    assert torch.equal(dW2, dW) and torch.equal(db2, db)


def test_mse_backward_and_cast():
    from motion324_amd import ops
    p, t = _rand((2, 3, 50, 3), 24), _rand((2, 3, 50, 3), 25)
    gs = torch.tensor(0.5, device=DEV)
    d = ops.mse_bwd(p.to(DEV), t.to(DEV), 2.0, gs)
    pt = p.double().requires_grad_(True)
    (0.5 * 2.0 * ((pt - t.double()) ** 2).mean()).backward()
    assert rel_err(d, pt.grad) < 1e-6
    x = _rand((37, 100), 26)
    b = ops.cast(x.to(DEV), torch.bfloat16)
    assert torch.equal(b.cpu(), x.to(torch.bfloat16))
    assert torch.equal(ops.cast(b, torch.float32).cpu(), x.to(torch.bfloat16).float())


def test_adamw_matches_torch_and_grad_norm():
    from motion324_amd import ops
    torch.manual_seed(0)
    p0, g = _rand((1000,), 27), _rand((1000,), 28)
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.AdamW([ref], lr=4e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    p, m, v = p0.clone().to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for step in range(1, 4):
        gi = g * step
        ref.grad = gi.double()
        opt.step()
        ops.adamw_step(p, gi.to(DEV), m, v, 4e-4, 0.9, 0.95, 1e-8, 0.05, step)
    assert rel_err(p, ref.detach()) < 1e-6
    # gradient norm with sanitising
    gg = g.clone()
    gg[3], gg[5], gg[7] = float("nan"), float("inf"), float("-inf")
    gd = gg.to(DEV)
    out, partial = torch.zeros((), device=DEV), torch.empty(1024, device=DEV)
    ops.grad_sumsq(gd, out, partial, sanitize=True, accumulate=False)
    clean = torch.nan_to_num(gg, nan=0.0, posinf=1e-6, neginf=-1e-6)
    assert torch.equal(gd.cpu(), clean)
    assert abs(float(out) - float((clean.double() ** 2).sum())) < 1e-3
    # the 16-byte form (aligned buffers of 4096 values and more; tail of n % 4 values) and the scalar form on an unaligned view
    big = _rand((1_000_003 + 1,), 29)
    for off in (0, 1):
        gg = big[off:off + 1_000_003 - off].clone()
        gg[3], gg[500_001], gg[-1], gg[-2] = float("nan"), float("inf"), float("-inf"), float("nan")
        gd = big.to(DEV)[off:off + gg.numel()]
        gd.copy_(gg)
        assert (gd.data_ptr() % 16 == 0) == (off == 0)
        ops.grad_sumsq(gd, out, partial, sanitize=True, accumulate=False)
        clean = torch.nan_to_num(gg, nan=0.0, posinf=1e-6, neginf=-1e-6)
        assert torch.equal(gd.cpu(), clean)
        want = float((clean.double() ** 2).sum())
        assert abs(float(out) - want) < 1e-5 * want
        ops.grad_sumsq(gd, out, partial, sanitize=False, accumulate=True)
        assert abs(float(out) - 2 * want) < 1e-5 * want


def _autograd_block(kind, sd, x, *args):
    from oracle import ref_forward as oracle
    sdd = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    xs = [a.double().requires_grad_(True) for a in (x,) + args]
    out = (oracle.self_attn_block(sdd, "b", xs[0], 64) if kind == "self"
           else oracle.cross_attn_block(sdd, "b", xs[0], xs[1], 64))
    return out, sdd, xs


@pytest.mark.parametrize("dtype", DT)
def test_self_attn_block_backward_vs_oracle_autograd(dtype):
    from motion324_amd import backward as bw
    from motion324_amd.prepared import Prepared
    from motion324_amd.transformer import QK_Norm_TransformerBlock
    torch.manual_seed(1)
    B, L, C = 2, 75, 192
    blk = QK_Norm_TransformerBlock(C, 64)
    with torch.no_grad():
        for n_, p in blk.named_parameters():
            p.copy_(1 + 0.1 * torch.randn_like(p) if p.dim() == 1 else 0.05 * torch.randn_like(p))
    x = _rand((B, L, C), 31)
    dout = _rand((B, L, C), 32)
    sd = {"b." + k: v.detach() for k, v in blk.state_dict().items()}
    out, sdd, xs = _autograd_block("self", sd, x)
    out.backward(dout.double())
    blk = blk.to(DEV)
    P = Prepared.for_module(blk, torch.device(DEV), dtype)
    G = bw.GradStore()
    dx = dout.reshape(B * L, C).clone().to(DEV)
    bw.self_attn_block_bwd(blk, P, G, x.reshape(B * L, C).to(DEV), dx, B, L)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel_err(dx, xs[0].grad.reshape(B * L, C)) < tol
    for name, p in blk.named_parameters():
        g = G.get(p)
        assert g is not None, name
        assert rel_err(g, sdd["b." + name].grad) < tol, (name, rel_err(g, sdd["b." + name].grad))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shared", [False, True])
def test_cross_attn_block_backward_vs_oracle_autograd(dtype, shared):
    from motion324_amd import backward as bw
    from motion324_amd.prepared import Prepared
    from motion324_amd.transformer import QK_Norm_CrossAttentionBlock
    torch.manual_seed(2)
    B, Lq, Lk, C = 3, 50, 64, 192
    blk = QK_Norm_CrossAttentionBlock(C, 64, kv_dim=C)
    with torch.no_grad():
        for n_, p in blk.named_parameters():
            p.copy_(1 + 0.1 * torch.randn_like(p) if p.dim() == 1 else 0.05 * torch.randn_like(p))
    q = _rand((1 if shared else B, Lq, C), 33)
    kv = _rand((B, Lk, C), 34)
    dout = _rand((B, Lq, C), 35)
    sd = {"b." + k: v.detach() for k, v in blk.state_dict().items()}
    from oracle import ref_forward as oracle
    sdd = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    qd, kvd = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    out = oracle.cross_attn_block(sdd, "b", qd.expand(B, -1, -1), kvd, 64)
    out.backward(dout.double())
    blk = blk.to(DEV)
    P = Prepared.for_module(blk, torch.device(DEV), dtype)
    G = bw.GradStore()
    dx = dout.reshape(B * Lq, C).clone().to(DEV)
    d_kv = torch.zeros((B * Lk, C), device=DEV)
    dquery = bw.cross_attn_block_bwd(blk, P, G, q.reshape(-1, C).to(DEV), kv.reshape(-1, C).to(DEV), dx, B, Lq, Lk,
                                     shared_q=shared, d_kv=d_kv)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel_err(dquery, qd.grad.reshape(-1, C)) < tol
    assert rel_err(d_kv, kvd.grad.reshape(-1, C)) < tol
    for name, p in blk.named_parameters():
        g = G.get(p)
        assert g is not None, name
        assert rel_err(g, sdd["b." + name].grad) < tol, (name, rel_err(g, sdd["b." + name].grad))


@pytest.mark.parametrize("nw", ["4", "8"])
@pytest.mark.parametrize("B,H,Lq,Lk,shared", [(2, 3, 70, 70, False), (1, 2, 33, 200, False), (3, 2, 50, 64, True),
                                               (1, 12, 324, 324, False), (1, 2, 700, 129, False)])
def test_attention_backward_mfma_against_autograd(tune, nw, B, H, Lq, Lk, shared):
    """bf16 MFMA backward kernels (through qkv_split's train outputs) vs fp64 autograd and vs the reference kernels;
    both workgroup sizes (M324_ATTN_BWD_NW; default: four waves)."""
    from motion324_amd import ops
    tune("M324_ATTN_BWD_NW", nw)
    dtype = torch.bfloat16
    Bq = 1 if shared else B
    C = H * 64
    q_tok, k_tok, v_tok = (_q(_rand((n * L_, C), sd_, 1.2), dtype) for n, L_, sd_ in ((Bq, Lq, 41), (B, Lk, 42), (B, Lk, 43)))
    dO_tok = _q(_rand((B * Lq, C), 44), dtype)
    dev = lambda t: t.to(dtype).to(DEV)
    spq = ops.qkv_split(dev(q_tok), None, None, None, None, 0.0, Bq, Lq, H, dtype, q_scale=ops.Q_PRESCALE, train=True)
    spk = ops.qkv_split(None, dev(k_tok), dev(v_tok), None, None, 0.0, B, Lk, H, dtype, train=True)
    out = torch.empty((B * Lq, C), dtype=dtype, device=DEV)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=DEV)
    ops.attention(spq["Q"], spk["K"], spk["Vt"], out, shared_q=shared, prescaled=True, lse=lse)
    D = ops.attention_delta(out, dev(dO_tok), B, H, Lq)
    spdo = ops.qkv_split(dev(dO_tok), None, None, None, None, 0.0, B, Lq, H, dtype, train=True)
    dQ, dK, dV = ops.attention_bwd_mfma(spq, spk, spdo, lse, D, shared_q=shared)
    rQ, rK, rV = ops.attention_bwd(spq["Q"], spk["K"], spk["V"], spdo["Q"], lse, D, shared_q=shared)
    for a, b_ in ((dQ, rQ), (dK, rK), (dV, rV)):
        assert torch.isfinite(a.float()).all()
        assert rel_err(a.float(), b_.float()) < 1.5e-2
    # fp64 autograd on the stored operands
    qs = spq["Q"].float().cpu().double()
    qh = (qs / ops.Q_PRESCALE).requires_grad_(True)
    kd = spk["K"].float().cpu().double().requires_grad_(True)
    vd = spk["V"].float().cpu().double().requires_grad_(True)
    sc = torch.einsum("bhqd,bhkd->bhqk", qh.expand(B, -1, -1, -1), kd) * (64 ** -0.5)
    o = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc, -1), vd).reshape(B * Lq, C)
    o.backward(dO_tok.double())
    got_dq = dQ.float().cpu().double()
    if shared:
        got_dq = got_dq.sum(0, keepdim=True)
    assert rel_err(got_dq, qh.grad) < 2e-2
    assert rel_err(dK.float(), kd.grad) < 2e-2 and rel_err(dV.float(), vd.grad) < 2e-2


def _poisoned_tail(t):
    """A copy of t whose storage is followed directly by NaNs (the tensor is a leading view of a larger buffer)."""
    big = torch.full((t.numel() + 65536,), float("nan"), dtype=t.dtype, device=t.device)
    big[:t.numel()] = t.reshape(-1)
    return big[:t.numel()].view(t.shape)


@pytest.mark.parametrize("Lq,Lk", [(70, 70), (33, 200), (324, 324), (130, 129)])
def test_attention_kernels_never_use_what_lies_behind_their_operands(Lq, Lk):
    """The kernels stage K / V / Q / dO tiles by buffer-form LDS-DMA whose resources end with the operand (`num_records`):
    rows of a ragged last tile that lie past the end must read as zeros -- they are multiplied by masked (zero) probabilities,
    and 0 x NaN would poison the result.  Every operand here is followed DIRECTLY by NaNs in memory (one batch, one head, so
    the tile overhang of the only head lands in the poison); forward with row-major and transposed V, then the backward."""
    from motion324_amd import ops
    dtype = torch.bfloat16
    B, H = 1, 1
    C = H * 64
    dev = lambda t: t.to(dtype).to(DEV)
    q_tok, k_tok, v_tok = (_q(_rand((L_, C), sd_, 1.2), dtype) for L_, sd_ in ((Lq, 41), (Lk, 42), (Lk, 43)))
    dO_tok = _q(_rand((Lq, C), 44), dtype)
    spq = ops.qkv_split(dev(q_tok), None, None, None, None, 0.0, B, Lq, H, dtype, q_scale=ops.Q_PRESCALE, train=True)
    spk = ops.qkv_split(None, dev(k_tok), dev(v_tok), None, None, 0.0, B, Lk, H, dtype, train=True)
    spdo = None
    spq = {k: _poisoned_tail(v) for k, v in spq.items() if torch.is_tensor(v)}
    spk = {k: _poisoned_tail(v) for k, v in spk.items() if torch.is_tensor(v)}
    out = torch.empty((Lq, C), dtype=dtype, device=DEV)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=DEV)
    ops.attention(spq["Q"], spk["K"], spk["Vt"], out, prescaled=True, lse=lse)
    out_r = torch.empty((Lq, C), dtype=dtype, device=DEV)
    ops.attention(spq["Q"], spk["K"], spk["V"], out_r, prescaled=True, v_rowmajor=True)
    ref = _attn = torch.softmax((q_tok.double() @ k_tok.double().T) * 64 ** -0.5, -1) @ v_tok.double()
    assert torch.isfinite(out.float()).all() and torch.isfinite(out_r.float()).all()
    assert rel_err(out.float(), ref) < 8e-3 and rel_err(out_r.float(), ref) < 8e-3
    D = ops.attention_delta(out, dev(dO_tok), B, H, Lq)
    spdo = {k: _poisoned_tail(v) for k, v in ops.qkv_split(dev(dO_tok), None, None, None, None, 0.0, B, Lq, H, dtype, train=True).items()
            if torch.is_tensor(v)}
    dQ, dK, dV = ops.attention_bwd_mfma(spq, spk, spdo, _poisoned_tail(lse), _poisoned_tail(D))
    for t in (dQ, dK, dV):
        assert torch.isfinite(t.float()).all()
    qh = (spq["Q"].float().cpu().double() / ops.Q_PRESCALE).requires_grad_(True)
    kd = spk["K"].float().cpu().double().requires_grad_(True)
    vd = spk["V"].float().cpu().double().requires_grad_(True)
    sc = torch.einsum("bhqd,bhkd->bhqk", qh, kd) * (64 ** -0.5)
    o = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc, -1), vd).reshape(Lq, C)
    o.backward(dO_tok.double())
    assert rel_err(dQ.float(), qh.grad) < 2e-2 and rel_err(dK.float(), kd.grad) < 2e-2 and rel_err(dV.float(), vd.grad) < 2e-2


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K,slices", [(192, 768, 4096, 8), (768, 64, 1000 // 64 * 64 + 64, 3), (100, 52, 640, 4)])
def test_gemm_splitk_matches_plain(dtype, M, N, K, slices):
    from motion324_amd import ops
    a, w = _q(_rand((M, K), 51, 0.3), dtype), _q(_rand((N, K), 52, 0.3), dtype)
    out = ops.gemm_splitk(a.to(dtype).to(DEV), w.to(dtype).to(DEV), slices)
    assert rel_err(out, a.double() @ w.double().T) < 2e-5


def test_colsum_wide_few_rows():
    from motion324_amd import ops
    x = _rand((12, 100003), 53)
    assert rel_err(ops.colsum(x.to(DEV)), x.double().sum(0)) < 1e-6
    xb = x.to(torch.bfloat16)
    assert rel_err(ops.colsum(xb.to(DEV)), xb.double().sum(0)) < 1e-6


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("R,Cc,ld", [(12, 100004, 100004), (8, 4096, 4100), (20000, 768, 768), (3000, 260, 264), (300, 64, 64), (2, 8, 8)])
def test_colsum_four_columns_per_lane(dtype, R, Cc, ld):
    """cols % 4 == 0 with aligned rows: 8- / 16-byte accesses (m324_colsum's wide and two-stage forms), strided sources, accumulation;
    the wide form sums a column's rows in order, exactly like the scalar kernel (bit-identical to it on an unaligned view)."""
    from motion324_amd import ops
    full = _q(_rand((R, ld), 61), dtype).to(dtype)
    x = full[:, :Cc]
    d = full.to(DEV)[:, :Cc]
    want = x.double().sum(0)
    s = ops.colsum(d)
    assert rel_err(s, want) < 1e-5
    ops.colsum(d, out=s, accumulate=True)
    assert rel_err(s, 2 * want) < 1e-5
    if R <= 64:
        pad = torch.zeros((R, ld + 1), dtype=dtype, device=DEV)          # every row starts at an odd element: the scalar kernel
        pad[:, 1:] = full.to(DEV)
        assert torch.equal(ops.colsum(pad[:, 1:1 + Cc]), ops.colsum(d))


@pytest.mark.parametrize("nw", ["4", "8"])
def test_attention_backward_mfma_many_workgroups_is_race_free(tune, nw):
    """Regression: the dQ kernel once relied on __syncthreads() to wait for its LDS-DMA tiles; the compiler emitted no
    vmcnt wait there, so under load a workgroup could read a stage still in flight (one NaN block in ~1 of 20 launches at
    the training shapes B*T = 96, L = 324).  Many co-resident workgroups, LDS poisoned with NaN patterns between launches,
    every launch compared with the first and with the fp32-arithmetic reference kernels."""
    from motion324_amd import ops
    tune("M324_ATTN_BWD_NW", nw)
    dtype = torch.bfloat16
    B, H, L = 24, 12, 324
    C = H * 64
    dev = lambda t: t.to(dtype).to(DEV)
    q_tok, k_tok, v_tok = (_rand((B * L, C), sd_, 1.0) for sd_ in (51, 52, 53))
    dO_tok = _rand((B * L, C), 54)
    spq = ops.qkv_split(dev(q_tok), None, None, None, None, 0.0, B, L, H, dtype, q_scale=ops.Q_PRESCALE, train=True)
    spk = ops.qkv_split(None, dev(k_tok), dev(v_tok), None, None, 0.0, B, L, H, dtype, train=True)
    out = torch.empty((B * L, C), dtype=dtype, device=DEV)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=DEV)
    ops.attention(spq["Q"], spk["K"], spk["Vt"], out, prescaled=True, lse=lse)
    D = ops.attention_delta(out, dev(dO_tok), B, H, L)
    spdo = ops.qkv_split(dev(dO_tok), None, None, None, None, 0.0, B, L, H, dtype, train=True)
    rQ, rK, rV = ops.attention_bwd(spq["Q"], spk["K"], spk["V"], spdo["Q"], lse, D)
    poison = torch.full((8192, 768), float("nan"), dtype=dtype, device=DEV)
    scratch = torch.empty((8192, 768), dtype=dtype, device=DEV)
    first = None
    for it in range(12):
        ops.gemm(poison, poison[:768], scratch)            # leaves NaN bit patterns in every CU's LDS
        dQ, dK, dV = ops.attention_bwd_mfma(spq, spk, spdo, lse, D)
        for a in (dQ, dK, dV):
            assert torch.isfinite(a.float()).all(), f"non-finite gradient in launch {it}"
        if first is None:
            first = (dQ.clone(), dK.clone(), dV.clone())
            for a, b_ in zip(first, (rQ, rK, rV)):
                assert rel_err(a.float(), b_.float()) < 1.5e-2
        else:
            for a, b_ in zip((dQ, dK, dV), first):
                assert torch.equal(a, b_), f"launch {it} differs from launch 0"
