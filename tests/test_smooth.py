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


def test_unsupported_method_is_rejected():
    from motion324_amd.postprocess import smooth_trajectories
    with pytest.raises(NotImplementedError):
        smooth_trajectories(torch.zeros(1, 4, 2, 3), method="oneeuro")
