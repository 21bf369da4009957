"""The scan helpers of baselines/common/math_util.py on the HIP kernels: NumPy arrays in and
out, as in the reference; the device-tensor forms are ``baselines_amd.hip.discount``."""
import numpy as np
import torch

from .. import hip


def _up(x):
    x = np.asarray(x)
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)          # lfilter and NumPy both promote integer input to float64
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def discount(x, gamma):
    """y[t] = x[t] + gamma*x[t+1] + gamma^2*x[t+2] + ... along axis 0, float64 out
    (baselines/common/math_util.py:5-23, scipy.signal.lfilter on the reversed array)."""
    x = np.asarray(x)
    assert x.ndim >= 1
    if x.size == 0:
        return np.zeros(x.shape, np.float64)
    return hip.discount(_up(x), gamma).cpu().numpy()


def discount_with_boundaries(X, New, gamma):
    """Discounted sums that restart where ``New[t]`` flags an episode start
    (baselines/common/math_util.py:63-73); the result has ``X``'s dtype."""
    X = np.asarray(X)
    if X.shape[0] == 0:
        return np.zeros_like(X)
    starts = torch.from_numpy(np.ascontiguousarray(np.asarray(New) != 0).astype(np.uint8)).cuda()
    x = _up(X)
    return hip.discount(x, gamma, starts=starts, out_dtype=x.dtype).cpu().numpy().astype(X.dtype, copy=False)
