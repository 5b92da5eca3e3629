"""Trajectory smoothing: oracle vs the reference function's goldens (CPU) and the HIP
kernels vs the same goldens (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

CASES = [("combined_0.002_1.0", "trajs", 0.002, 1.0), ("combined_0.005_2.0", "trajs", 0.005, 2.0),
         ("threshold_0.005", "trajs", 0.005, 0.0), ("gaussian_1.5", "trajs", -1.0, 1.5),
         ("short_combined_0.002_1.0", "trajs_short", 0.002, 1.0)]


@pytest.mark.parametrize("key,src,thr,sigma", CASES)
def test_oracle_matches_reference_function(key, src, thr, sigma):
    from oracle.ref_smooth import smooth_trajectories
    g = np.load(os.path.join(GOLDEN, "smooth.npz"))
    out = smooth_trajectories(g[src], thr, sigma)
    assert np.abs(out - g[key]).max() < 2e-6
    if sigma == 0.0:
        assert np.array_equal(out, g[key])          # threshold pass only moves data: exact


@pytest.mark.gpu
@pytest.mark.parametrize("key,src,thr,sigma", CASES)
def test_hip_smoothing_matches_reference_function(key, src, thr, sigma):
    from motion324_amd.postprocess import smooth_trajectories
    g = np.load(os.path.join(GOLDEN, "smooth.npz"))
    method = "combined" if (thr >= 0 and sigma > 0) else ("threshold" if sigma == 0.0 else "gaussian")
    out = smooth_trajectories(torch.from_numpy(g[src]).cuda(), method=method, motion_threshold=max(thr, 0.0), sigma=sigma)
    out = out.cpu().numpy()
    assert np.abs(out - g[key]).max() < 2e-6
    if sigma == 0.0:
        assert np.array_equal(out, g[key])


EXTRA = [("savgol_5_2", "trajs", "savgol", dict(window_size=5, savgol_polyorder=2)),
         ("savgol_8_3", "trajs", "savgol", dict(window_size=8, savgol_polyorder=3)),
         ("savgol_3_2", "trajs", "savgol", dict()),
         ("short_savgol_5_2", "trajs_short", "savgol", dict(window_size=5, savgol_polyorder=2)),
         ("oneeuro_1.0_0.007", "trajs", "oneeuro", dict()),
         ("oneeuro_0.3_0.5", "trajs", "oneeuro", dict(oneeuro_mincutoff=0.3, oneeuro_beta=0.5))]


@pytest.mark.parametrize("key,src,method,kw", EXTRA)
def test_oracle_savgol_oneeuro_match_reference_function(key, src, method, kw):
    from oracle import ref_smooth
    g = np.load(os.path.join(GOLDEN, "smooth.npz"))
    if method == "savgol":
        out = ref_smooth.savgol(g[src], kw.get("window_size", 3), kw.get("savgol_polyorder", 2))
    else:
        out = ref_smooth.oneeuro(g[src], kw.get("oneeuro_mincutoff", 1.0), kw.get("oneeuro_beta", 0.007))
    assert out.dtype == np.float32 and np.abs(out - g[key]).max() < 2e-6
    if key in ("savgol_3_2", "short_savgol_5_2"):
        assert np.abs(out - g[src]).max() < 1e-6          # an exact fit / an untouched short clip


def test_savgol_coefficients_known_answers():
    """The classic tables: 5-point quadratic (-3, 12, 17, 12, -3) / 35 and 7-point quadratic (-2, 3, 6, 7, 6, 3, -2) / 21."""
    from motion324_amd.postprocess import savgol_coeffs
    assert np.allclose(savgol_coeffs(5, 2), np.array([-3, 12, 17, 12, -3]) / 35.0, atol=1e-14)
    assert np.allclose(savgol_coeffs(7, 2), np.array([-2, 3, 6, 7, 6, 3, -2]) / 21.0, atol=1e-14)
    assert np.allclose(savgol_coeffs(3, 2), [0, 1, 0], atol=1e-14)
    with pytest.raises(ValueError):
        savgol_coeffs(4, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("key,src,method,kw", EXTRA)
def test_hip_savgol_oneeuro_match_reference_function(key, src, method, kw):
    from motion324_amd.postprocess import smooth_trajectories
    g = np.load(os.path.join(GOLDEN, "smooth.npz"))
    x = torch.from_numpy(g[src]).cuda()
    out = smooth_trajectories(x, method=method, **kw)
    assert out.shape == x.shape and out.dtype == x.dtype and out.data_ptr() != x.data_ptr()
    assert np.abs(out.cpu().numpy() - g[key]).max() < 2e-6


def test_unknown_method_returns_a_copy_like_the_reference():
    from motion324_amd.postprocess import smooth_trajectories
    x = torch.rand(1, 4, 2, 3)
    y = smooth_trajectories(x, method="none")
    assert torch.equal(x, y) and y.data_ptr() != x.data_ptr()
