"""GPU version of the caller-side trajectory smoothing (reference utils/inference_utils.py:99-195).

Once the forward takes milliseconds, the reference's CPU triple loop over B*N*3 (one scipy call per point
coordinate) dominates the wall clock of scripts/inference_with_video_mesh.py:395-400; here every method is one or
two HBM-bound kernels.  Same signature and defaults as the reference function.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def savgol_coeffs(window: int, polyorder: int) -> np.ndarray:
    """Savitzky-Golay smoothing coefficients (deriv 0, centre position), the published least-squares construction of
    scipy.signal.savgol_coeffs: row 0 of the pseudo-inverse of the Vandermonde matrix of x = -h..h, which is the
    weight of every sample in the fitted polynomial's value at the centre.  Symmetric, sums to 1."""
    if window % 2 != 1 or window < 1:
        raise ValueError("window must be odd")
    if polyorder >= window:
        raise ValueError("polyorder must be less than window")
    h = window // 2
    x = np.arange(-h, h + 1, dtype=np.float64)
    A = np.vander(x, polyorder + 1, increasing=True)             # [window, order + 1]
    return np.linalg.pinv(A)[0].copy()                            # [window]


def smooth_trajectories(trajs: torch.Tensor, method: str = "combined", motion_threshold: float = 0.005,
                        window_size: int = 3, sigma: float = 1.0, savgol_polyorder: int = 2,
                        oneeuro_mincutoff: float = 1.0, oneeuro_beta: float = 0.007, visualization_dir=None):
    """(B, T, N, 3) -> smoothed (B, T, N, 3), dtype/device preserved.  Methods as the reference: 'threshold', 'gaussian',
    'combined' (threshold then gaussian), 'savgol', 'oneeuro'; any other string returns a copy (as the reference does)."""
    if method in ("threshold", "gaussian", "combined"):
        thr = motion_threshold if method in ("threshold", "combined") else -1.0
        sg = sigma if method in ("gaussian", "combined") else 0.0
        return ops.smooth_trajectories(trajs, float(thr), float(sg)).to(trajs.dtype)
    if method == "savgol":
        if window_size % 2 == 0:
            window_size += 1                                      # reference :151-152
        if trajs.shape[1] < window_size:
            return trajs.clone()                                  # reference :153: shorter clips stay unfiltered
        coef = savgol_coeffs(window_size, min(savgol_polyorder, window_size - 1))
        return ops.smooth_savgol(trajs, torch.from_numpy(coef).to(trajs.device)).to(trajs.dtype)
    if method == "oneeuro":
        return ops.smooth_oneeuro(trajs, float(oneeuro_mincutoff), float(oneeuro_beta), 1.0).to(trajs.dtype)
    return trajs.clone()
