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
