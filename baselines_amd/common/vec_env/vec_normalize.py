"""VecNormalize of baselines/common/vec_env/vec_normalize.py with the running statistics and
both filters on the device."""
import numpy as np
import torch

from ... import hip
from ..running_mean_std import RunningMeanStd
from . import VecEnvWrapper


def _to_dev(x):
    if isinstance(x, torch.Tensor):
        return x.contiguous(), False
    x = np.asarray(x)
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    return torch.from_numpy(np.ascontiguousarray(x)).cuda(), True


class VecNormalize(VecEnvWrapper):
    """Same arguments as the reference's class (:9).  A venv that hands out device tensors gets
    device tensors back (float32, ready for the policy); NumPy in gives NumPy out in float64, as
    the reference's filters return."""

    def __init__(self, venv, ob=True, ret=True, clipob=10., cliprew=10., gamma=0.99, epsilon=1e-8):
        VecEnvWrapper.__init__(self, venv)
        self.ob_rms = RunningMeanStd(shape=self.observation_space.shape) if ob else None
        self.ret_rms = RunningMeanStd(shape=()) if ret else None
        self.clipob = clipob
        self.cliprew = cliprew
        self._ret = torch.zeros(self.num_envs, dtype=torch.float64, device="cuda")
        self.gamma = gamma
        self.epsilon = epsilon
        self._ws = None

    @property
    def ret(self):
        return self._ret.cpu().numpy()

    def step_wait(self):
        obs, rews, news, infos = self._venv_step_wait()
        r, host = _to_dev(rews)
        out_dtype = torch.float64 if host else torch.float32
        r = hip.vecnorm_returns(r.reshape(-1), self._ret, self.ret_rms.state if self.ret_rms else None, self.gamma,
                                self.epsilon, self.cliprew, out_dtype=out_dtype)         # :27-31
        obs = self._obfilt(obs)
        return obs, (r.cpu().numpy() if host else r), news, infos

    def _obfilt(self, obs):
        if not self.ob_rms:
            return obs
        x, host = _to_dev(obs)
        self.ob_rms.update(x)                                                            # :36
        out = hip.normalize_clip(x, self.ob_rms.state, self.epsilon, self.clipob,
                                 out_dtype=torch.float64 if host else torch.float32)    # :37
        return out.cpu().numpy() if host else out

    def reset(self):
        return self._obfilt(self.venv.reset())
