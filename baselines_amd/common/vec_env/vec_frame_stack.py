"""VecFrameStack of baselines/common/vec_env/vec_frame_stack.py with the stack kept in HBM."""
import numpy as np
import torch

from ... import hip, spaces
from . import VecEnvWrapper


class VecFrameStack(VecEnvWrapper):
    """``stackedobs`` lives on the device and is rolled in place by one kernel per step (:19-26; as in the
    reference the roll is by one element of the last axis, i.e. one frame for single-channel frames).
    NumPy observations from the venv are uploaded and the stack is returned as NumPy; device
    observations keep everything in HBM and the returned tensor IS the stack (copy it before the
    next step if it must outlive it, as with the reference's array)."""

    def __init__(self, venv, nstack):
        self.nstack = nstack
        wos = venv.observation_space
        low = np.repeat(wos.low, self.nstack, axis=-1)
        high = np.repeat(wos.high, self.nstack, axis=-1)
        if low.dtype.itemsize not in (1, 4):
            raise TypeError("frame stacking on the device needs 1- or 4-byte elements, got %s" % low.dtype)
        self._tdtype = torch.from_numpy(np.zeros(1, low.dtype)).dtype
        self.stackedobs = torch.zeros((venv.num_envs,) + low.shape, dtype=self._tdtype, device="cuda")
        observation_space = spaces.Box(low=low, high=high, dtype=low.dtype)
        VecEnvWrapper.__init__(self, venv, observation_space=observation_space)

    def _dev(self, a):
        if isinstance(a, torch.Tensor):
            return a.contiguous(), False
        return torch.from_numpy(np.ascontiguousarray(a, dtype=self.observation_space.low.dtype)).cuda(), True

    def step_wait(self):
        obs, rews, news, infos = self._venv_step_wait()
        o, host = self._dev(obs)
        n = news if isinstance(news, torch.Tensor) else torch.from_numpy(np.asarray(news, dtype=np.bool_).astype(np.uint8)).cuda()
        hip.frame_stack(self.stackedobs, o, n)
        return (self.stackedobs.cpu().numpy() if host else self.stackedobs), rews, news, infos

    def reset(self):
        o, host = self._dev(self.venv.reset())
        self.stackedobs.zero_()
        hip.frame_stack(self.stackedobs, o, None)      # zeros rolled by one frame stay zeros (:33-34)
        return self.stackedobs.cpu().numpy() if host else self.stackedobs
