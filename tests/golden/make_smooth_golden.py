#!/usr/bin/env python3
"""Golden vectors of the reference's trajectory smoothing, produced by executing the
reference function itself: `smooth_trajectories` is cut out of /root/reference/utils/inference_utils.py with `ast`
(the module's own imports -- matplotlib plotting helpers -- are not needed for the arithmetic) and run with scipy's
gaussian_filter1d / savgol_filter in scope.  Build container only.  Output: tests/golden/smooth.npz."""
import ast
import os

import numpy as np
import torch
from scipy.ndimage import gaussian_filter1d
from scipy.signal import savgol_filter

SRC = "/root/reference/utils/inference_utils.py"
HERE = os.path.dirname(os.path.abspath(__file__))
tree = ast.parse(open(SRC).read())
fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "smooth_trajectories")
euro = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "OneEuroFilter")
ns = {"torch": torch, "np": np, "gaussian_filter1d": gaussian_filter1d, "savgol_filter": savgol_filter,
      "print": lambda *a, **k: None}
exec(compile(ast.Module(body=[euro, fn], type_ignores=[]), SRC, "exec"), ns)
smooth = ns["smooth_trajectories"]

g = torch.Generator().manual_seed(7)
B, T, N = 2, 37, 50
base = torch.rand((B, 1, N, 3), generator=g) - 0.5
steps = torch.randn((B, T, N, 3), generator=g) * 0.004          # many displacements straddle the thresholds
steps[:, ::5] *= 4
trajs = base + torch.cumsum(steps, dim=1)
out = {"trajs": trajs.numpy()}
# the call the reference's callers make: method='combined', motion_threshold=0.002 (inference_with_video_mesh.py:395)
out["combined_0.002_1.0"] = smooth(trajs, method="combined", motion_threshold=0.002, sigma=1.0).numpy()
out["combined_0.005_2.0"] = smooth(trajs, method="combined", motion_threshold=0.005, sigma=2.0).numpy()
out["threshold_0.005"] = smooth(trajs, method="threshold", motion_threshold=0.005).numpy()
out["gaussian_1.5"] = smooth(trajs, method="gaussian", sigma=1.5).numpy()
short = trajs[:, :3].contiguous()                               # T shorter than the filter radius
out["trajs_short"] = short.numpy()
out["short_combined_0.002_1.0"] = smooth(short, method="combined", motion_threshold=0.002, sigma=1.0).numpy()
# the two remaining methods (utils/inference_utils.py:148-195).  NOTE: this container runs numpy 2.x, whose scalar
# promotion keeps the One Euro state in float32; the reference pins numpy 1.26 (float64 state).  The two differ by
# ~1e-7 relative, far inside the 2e-6 the tests allow.
out["savgol_5_2"] = smooth(trajs, method="savgol", window_size=5, savgol_polyorder=2).numpy()
out["savgol_8_3"] = smooth(trajs, method="savgol", window_size=8, savgol_polyorder=3).numpy()      # even window -> 9
out["savgol_3_2"] = smooth(trajs, method="savgol").numpy()                                          # defaults: identity fit
out["short_savgol_5_2"] = smooth(short, method="savgol", window_size=5, savgol_polyorder=2).numpy()   # T < window: untouched
out["oneeuro_1.0_0.007"] = smooth(trajs, method="oneeuro").numpy()
out["oneeuro_0.3_0.5"] = smooth(trajs, method="oneeuro", oneeuro_mincutoff=0.3, oneeuro_beta=0.5).numpy()
np.savez_compressed(os.path.join(HERE, "smooth.npz"), **out)
print({k: v.shape for k, v in out.items()})
