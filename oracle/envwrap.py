"""Env-wrapper filters: RunningMeanStd, the VecNormalize filters, the VecFrameStack roll.

Test infrastructure (see ``oracle/__init__.py``).  NumPy restatements that follow the reference line
by line; ``moments_dtype`` widens the batch moments to float64 where a test wants to separate the
device's float64 accumulation from NumPy's accumulation in the input's own dtype.
"""
import numpy as np


class RunningMeanStd:
    """``baselines/common/running_mean_std.py:2-29``."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, 'float64')
        self.var = np.ones(shape, 'float64')
        self.count = epsilon

    def update(self, x, moments_dtype=None):
        x = np.asarray(x)
        if moments_dtype is not None:
            x = x.astype(moments_dtype)
        self.update_from_moments(np.mean(x, axis=0), np.var(x, axis=0), x.shape[0])      # :10-13

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean                                                  # :16
        tot_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot_count                          # :19
        m_a = self.var * self.count
        m_b = batch_var * batch_count
        M2 = m_a + m_b + np.square(delta) * self.count * batch_count / (self.count + batch_count)   # :22
        self.mean, self.var, self.count = new_mean, M2 / (self.count + batch_count), batch_count + self.count


class VecNormalizeFilter:
    """The arithmetic of ``baselines/common/vec_env/vec_normalize.py:9-47`` without the venv."""

    def __init__(self, nenvs, ob_shape, ob=True, ret=True, clipob=10., cliprew=10., gamma=0.99, epsilon=1e-8,
                 moments_dtype=None):
        self.ob_rms = RunningMeanStd(shape=ob_shape) if ob else None
        self.ret_rms = RunningMeanStd(shape=()) if ret else None
        self.clipob, self.cliprew, self.gamma, self.epsilon = clipob, cliprew, gamma, epsilon
        self.ret = np.zeros(nenvs)
        self.md = moments_dtype

    def obfilt(self, obs):                                                              # :34-40
        if not self.ob_rms:
            return obs
        self.ob_rms.update(obs, self.md)
        return np.clip((obs - self.ob_rms.mean) / np.sqrt(self.ob_rms.var + self.epsilon), -self.clipob, self.clipob)

    def step(self, obs, rews):                                                          # :26-32
        self.ret = self.ret * self.gamma + rews
        obs = self.obfilt(obs)
        if self.ret_rms:
            self.ret_rms.update(self.ret, self.md)
            rews = np.clip(rews / np.sqrt(self.ret_rms.var + self.epsilon), -self.cliprew, self.cliprew)
        return obs, rews


class FrameStack:
    """``baselines/common/vec_env/vec_frame_stack.py:9-35`` without the venv."""

    def __init__(self, nenvs, ob_shape, dtype, nstack):
        shape = tuple(ob_shape[:-1]) + (ob_shape[-1] * nstack,)
        self.stackedobs = np.zeros((nenvs,) + shape, dtype)

    def reset(self, obs):
        self.stackedobs[...] = 0
        self.stackedobs[..., -obs.shape[-1]:] = obs
        return self.stackedobs

    def step(self, obs, news):
        self.stackedobs = np.roll(self.stackedobs, shift=-1, axis=-1)                   # :21
        for (i, new) in enumerate(news):
            if new:
                self.stackedobs[i] = 0
        self.stackedobs[..., -obs.shape[-1]:] = obs
        return self.stackedobs


class MpiRunningMeanStd:
    """``baselines/common/mpi_running_mean_std.py:4-43`` with the all-reduce as a hook: ``allreduce(vec)``
    returns the sum of ``vec`` over the ranks (identity for one process)."""

    def __init__(self, epsilon=1e-2, shape=(), allreduce=None):
        self.shape = tuple(shape)
        self._sum = np.zeros(self.shape, 'float64')
        self._sumsq = np.full(self.shape, epsilon, 'float64')
        self._count = np.float64(epsilon)
        self.allreduce = allreduce or (lambda v: v)

    def update(self, x):
        x = np.asarray(x).astype('float64')                                               # :37
        n = int(np.prod(self.shape))
        addvec = np.concatenate([x.sum(axis=0).ravel(), np.square(x).sum(axis=0).ravel(),
                                 np.array([len(x)], dtype='float64')])                    # :40
        tot = self.allreduce(addvec)
        self._sum = self._sum + tot[0:n].reshape(self.shape)
        self._sumsq = self._sumsq + tot[n:2 * n].reshape(self.shape)
        self._count = self._count + tot[2 * n]

    @property
    def mean(self):
        return (self._sum / self._count).astype(np.float32)                               # :25

    @property
    def std(self):
        return np.sqrt(np.maximum((self._sumsq / self._count).astype(np.float32) - np.square(self.mean), np.float32(1e-2)))
