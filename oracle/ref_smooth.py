"""ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's smooth_trajectories
(utils/inference_utils.py:99-148: threshold pass :126-131, gaussian pass :134-144 via
scipy.ndimage.gaussian_filter1d(sigma, mode='nearest'), whose published algorithm is restated here: radius
int(4*sigma + 0.5), weights exp(-0.5 x^2 / sigma^2) normalised to 1, out-of-range samples clamped).
savgol (:148-165) = scipy.signal.savgol_filter(mode='nearest'): least-squares polynomial-fit coefficients convolved with
clamped borders; oneeuro (:58-97,186-195) = the One Euro filter per scalar coordinate.
Pinned against the reference function itself: tests/golden/smooth.npz (tests/golden/make_smooth_golden.py)."""
import numpy as np


def savgol(trajs: np.ndarray, window: int, polyorder: int) -> np.ndarray:
    x = np.asarray(trajs, dtype=np.float32)
    T = x.shape[1]
    if window % 2 == 0:
        window += 1                                           # :151-152
    if T < window:
        return x.copy()                                       # :153
    order = min(polyorder, window - 1)
    h = window // 2
    A = np.vander(np.arange(-h, h + 1, dtype=np.float64), order + 1, increasing=True)
    c = np.linalg.pinv(A)[0]                                  # weights of the fitted value at the window centre
    idx = np.clip(np.arange(T)[:, None] + h - np.arange(window)[None, :], 0, T - 1)      # y[t] = sum_j c[j] x[t + h - j]
    return np.einsum("btknc,k->btnc", x[:, idx].astype(np.float64), c).astype(np.float32)


def oneeuro(trajs: np.ndarray, mincutoff: float = 1.0, beta: float = 0.007, dcutoff: float = 1.0) -> np.ndarray:
    x = np.asarray(trajs, dtype=np.float32)
    T = x.shape[1]
    out = x.copy()
    alpha_d = (2 * np.pi * dcutoff) / (2 * np.pi * dcutoff + 1)
    x_prev = x[:, 0].astype(np.float64)
    dx_prev = np.zeros_like(x_prev)
    for t in range(1, T):
        xt = x[:, t].astype(np.float64)
        dx = (x[:, t] - x[:, 0]).astype(np.float64) if t == 1 else xt - x_prev
        dx_hat = alpha_d * dx + (1 - alpha_d) * dx_prev
        r = 2 * np.pi * (mincutoff + beta * np.abs(dx_hat))
        alpha = r / (r + 1)
        x_hat = alpha * xt + (1 - alpha) * x_prev
        x_prev, dx_prev = x_hat, dx_hat
        out[:, t] = x_hat.astype(np.float32)
    return out


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
