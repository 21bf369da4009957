"""Device-side forms of the vectorised-env wrappers that sit right before the rollout."""


class VecEnvWrapper(object):
    """The wrapper base of baselines/common/vec_env/__init__.py:90-113, for venvs that offer either
    ``step_async``/``step_wait`` or a plain ``step``."""

    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        self.num_envs = venv.num_envs
        self.observation_space = observation_space or venv.observation_space
        self.action_space = action_space or venv.action_space
        self._pending = None
        # an env that lives in HBM stays one under its wrappers: tensors in, tensors out
        self.device_resident = bool(getattr(venv, "device_resident", False))

    def step_async(self, actions):
        if hasattr(self.venv, "step_async"):
            self.venv.step_async(actions)
        else:
            self._pending = actions

    def _venv_step_wait(self):
        if hasattr(self.venv, "step_async"):
            return self.venv.step_wait()
        actions, self._pending = self._pending, None
        return self.venv.step(actions)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        return self.venv.close()
