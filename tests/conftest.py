import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP device); run with -m gpu")


def has_gpu() -> bool:
    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# configs of the committed golden cases (must match tests/golden/make_golden.py::CASES)
CASES = {
    "tiny": dict(dims=dict(d=192, d_head=64, tokens=8, pcd_layers=1, n_layer=2, frames=3, dino_depth=2),
                 shape=(2, 3, 40, 100, 64)),
    "tiny_resize": dict(dims=dict(d=192, d_head=64, tokens=8, pcd_layers=1, n_layer=2, frames=5, dino_depth=2),
                        shape=(1, 2, 33, 70, 96)),
    "c1": dict(dims=dict(frames=12), shape=(1, 4, 512, 4096, 256)),
    "c2": dict(dims=dict(frames=32), shape=(1, 32, 2048, 4096, 512)),
    "c5": dict(dims=dict(frames=256), shape=(1, 256, 2048, 4096, 512)),      # golden keeps a sample of the mesh points
    # no golden: the c2 TRUNK LENGTH (32 frames x 324 tokens = 10368 rows, 512 x 512 frames) with two global + two local blocks, two
    # DINO blocks, one point block and fewer points -- what the multi-process tests need of c2 at a quarter of its start-up time
    "c2_shallow": dict(dims=dict(frames=32, n_layer=4, dino_depth=2, pcd_layers=1), shape=(1, 32, 512, 1024, 512), golden=False),
}


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


_SD_CACHE = {}


def synth_sd(dims_kwargs):
    """Cached synthetic state dict (numpy) for a Dims kwargs dict."""
    from motion324_amd import synth
    key = tuple(sorted(dims_kwargs.items()))
    if key not in _SD_CACHE:
        _SD_CACHE[key] = synth.synth_state_dict(synth.Dims(**dims_kwargs), seed=0)
    return _SD_CACHE[key]


def vt_layout(v: torch.Tensor) -> torch.Tensor:
    """V [B,H,Lk,64] -> the Vt operand m324_attention expects: [B,H,64,round_up(Lk,64)], zero padded, with the
    key quarters of every aligned 16-key group stored in the order 0,2,1,3 (include/m324.h, m324_qkv_split)."""
    B, H, Lk, D = v.shape
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros((B, H, D, Lp), dtype=v.dtype)
    vt[..., :Lk] = v.transpose(2, 3)
    vt = vt.reshape(B, H, D, Lp // 16, 4, 4)[..., [0, 2, 1, 3], :]
    return vt.reshape(B, H, D, Lp).contiguous()


@pytest.fixture
def tune():
    """tune("M324_GEMM", "v10"): overrides a kernel-chooser switch of libm324 through m324_set_tunable (the library reads
    the environment only once, at load) and restores every touched switch after the test."""
    from motion324_amd import lib
    touched = []

    def _set(name, value):
        touched.append(name)
        lib.set_tunable(name, int(str(value).lstrip("vV")))
    yield _set
    for name in touched:
        lib.set_tunable(name)
