"""RunningMeanStd of baselines/common/running_mean_std.py with its statistics in HBM."""
import numpy as np
import torch

from .. import hip


class RunningMeanStd(object):
    """Same constructor and attributes as the reference's class (:2-7); ``update`` takes a NumPy
    array or a device tensor of shape [n, *shape].  ``mean``, ``var`` and ``count`` read the
    device state back (a synchronising copy); ``state`` is the device tensor itself."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.shape = tuple(shape)
        self.d = int(np.prod(self.shape)) if self.shape else 1
        self.state = hip.rms_state(self.d, epsilon)
        self._ws = None

    @property
    def mean(self):
        return self.state[:self.d].cpu().numpy().reshape(self.shape)

    @property
    def var(self):
        return self.state[self.d:2 * self.d].cpu().numpy().reshape(self.shape)

    @property
    def count(self):
        return float(self.state[2 * self.d].item())

    def update(self, x):
        if not isinstance(x, torch.Tensor):
            x = np.asarray(x)
            if x.dtype not in (np.float32, np.float64):
                x = x.astype(np.float64)
            x = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        self._ws = hip.rms_update(self.state, x.contiguous(), self._ws)

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        """:15-29 on host values (kept for callers that bring their own moments)."""
        mean, var, count = self.mean.astype(np.float64), self.var.astype(np.float64), self.count
        delta = batch_mean - mean
        tot_count = count + batch_count
        new_mean = mean + delta * batch_count / tot_count
        M2 = var * count + batch_var * batch_count + np.square(delta) * count * batch_count / (count + batch_count)
        st = np.concatenate([np.ravel(new_mean), np.ravel(M2 / (count + batch_count)),
                             np.full(self.d, batch_count + count)]).astype(np.float64)
        self.state.copy_(torch.from_numpy(st))
