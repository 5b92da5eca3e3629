"""GPU version of the caller-side trajectory smoothing (reference utils/inference_utils.py:99-195).

Once the forward takes milliseconds, the reference's CPU triple loop over B*N*3 (one scipy call per point
coordinate) dominates the wall clock of scripts/inference_with_video_mesh.py:395-400; here the same filter is
two HBM-bound kernels.  Same signature and defaults as the reference function for the methods its callers use.
"""
from __future__ import annotations

import torch

from . import ops


def smooth_trajectories(trajs: torch.Tensor, method: str = "combined", motion_threshold: float = 0.005,
                        window_size: int = 3, sigma: float = 1.0, savgol_polyorder: int = 2,
                        oneeuro_mincutoff: float = 1.0, oneeuro_beta: float = 0.007, visualization_dir=None):
    """(B, T, N, 3) -> smoothed (B, T, N, 3), dtype/device preserved.  'threshold', 'gaussian', 'combined'."""
    if method not in ("threshold", "gaussian", "combined"):
        raise NotImplementedError(f"smooth_trajectories(method={method!r}) is not implemented on the HIP path "
                                  "(the reference's callers use 'combined')")
    thr = motion_threshold if method in ("threshold", "combined") else -1.0
    sg = sigma if method in ("gaussian", "combined") else 0.0
    return ops.smooth_trajectories(trajs, float(thr), float(sg)).to(trajs.dtype)
