"""ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's smooth_trajectories
(utils/inference_utils.py:99-148: threshold pass :126-131, gaussian pass :134-144 via
scipy.ndimage.gaussian_filter1d(sigma, mode='nearest'), whose published algorithm is restated here: radius
int(4*sigma + 0.5), weights exp(-0.5 x^2 / sigma^2) normalised to 1, out-of-range samples clamped).
Pinned against the reference function itself: tests/golden/smooth.npz (tests/golden/make_smooth_golden.py)."""
import numpy as np


def smooth_trajectories(trajs: np.ndarray, threshold: float = -1.0, sigma: float = 0.0) -> np.ndarray:
    x = np.asarray(trajs, dtype=np.float32)
    B, T, N, _ = x.shape
    out = x.copy()
    if threshold >= 0:
        for t in range(1, T):
            still = np.linalg.norm(x[:, t] - x[:, t - 1], axis=-1) < threshold       # displacement of the ORIGINAL
            out[:, t][still] = out[:, t - 1][still]                                   # copies the SMOOTHED t-1
    if sigma > 0:
        r = int(4.0 * sigma + 0.5)
        w = np.exp(-0.5 * (np.arange(-r, r + 1, dtype=np.float64) ** 2) / (sigma * sigma))
        w /= w.sum()
        idx = np.clip(np.arange(T)[:, None] + np.arange(-r, r + 1)[None, :], 0, T - 1)     # [T, 2r+1]
        out = np.einsum("btknc,k->btnc", out[:, idx].astype(np.float64), w).astype(np.float32)
    return out
