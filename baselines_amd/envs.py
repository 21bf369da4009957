"""Small in-process vectorised environments with the VecEnv interface of
baselines/common/vec_env/__init__.py:22-88 (num_envs, observation_space, action_space,
reset, step, close), used by the tests, the smoke run and the CPU-runnable CartPole
configuration.  gym is not a dependency, so CartPole's dynamics are written out here.
"""
import numpy as np

from .spaces import Box, Discrete


class CartPoleVecEnv:
    """`nenvs` independent cart-poles (classic control equations, Euler integration,
    episode ends at |x| > 2.4, |theta| > 12 deg or 200 steps), auto-resetting on done."""

    def __init__(self, nenvs, seed=0, reward_columns=None):
        self.num_envs = nenvs
        self.observation_space = Box(-np.inf, np.inf, shape=(4,), dtype=np.float32)
        self.action_space = Discrete(2)
        self.rng = np.random.RandomState(seed)
        self.reward_columns = reward_columns
        self.state = np.zeros((nenvs, 4))
        self.steps = np.zeros(nenvs, np.int64)
        self.ep_ret = np.zeros(nenvs)

    def _reset_some(self, mask):
        n = int(mask.sum())
        self.state[mask] = self.rng.uniform(-0.05, 0.05, size=(n, 4))
        self.steps[mask] = 0
        self.ep_ret[mask] = 0.0

    def reset(self):
        self._reset_some(np.ones(self.num_envs, bool))
        return self.state.astype(np.float32)

    def step(self, actions):
        g, mc, mp, l, fm, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
        x, xd, th, thd = self.state.T
        force = np.where(np.asarray(actions).ravel() == 1, fm, -fm)
        ct, st = np.cos(th), np.sin(th)
        temp = (force + mp * l * thd ** 2 * st) / (mc + mp)
        thacc = (g * st - ct * temp) / (l * (4.0 / 3.0 - mp * ct ** 2 / (mc + mp)))
        xacc = temp - mp * l * thacc * ct / (mc + mp)
        self.state = np.stack([x + tau * xd, xd + tau * xacc, th + tau * thd, thd + tau * thacc], axis=1)
        self.steps += 1
        self.ep_ret += 1.0
        dones = (np.abs(self.state[:, 0]) > 2.4) | (np.abs(self.state[:, 2]) > 12 * np.pi / 180) | (self.steps >= 200)
        infos = [{} for _ in range(self.num_envs)]
        for i in np.nonzero(dones)[0]:
            infos[i] = {"episode": {"r": np.array([self.ep_ret[i]]), "l": int(self.steps[i])}}
        self._reset_some(dones)
        rews = np.ones(self.num_envs, np.float32)
        if self.reward_columns:
            rews = np.repeat(rews[:, None], self.reward_columns, axis=1)
        return self.state.astype(np.float32), rews, dones, infos

    def close(self):
        pass


class SyntheticVecEnv:
    """Random observations / rewards / dones of a given shape — a stand-in for MuJoCo- or
    Atari-shaped environments when only the update path is under test.  With
    ``reward_columns = R + 1`` it returns the fork's reward layout (R channels + total score,
    baselines/ppo2/common_functions.py:262)."""

    def __init__(self, nenvs, ob_shape, action_space, reward_columns=None, done_p=0.01, ob_dtype=np.float32, seed=0):
        self.num_envs = nenvs
        self.ob_shape, self.ob_dtype = tuple(ob_shape), np.dtype(ob_dtype)
        lo, hi = (0, 255) if self.ob_dtype == np.uint8 else (-np.inf, np.inf)
        self.observation_space = Box(lo, hi, shape=self.ob_shape, dtype=self.ob_dtype)
        self.action_space = action_space
        self.reward_columns, self.done_p = reward_columns, done_p
        self.rng = np.random.RandomState(seed)

    def _obs(self):
        if self.ob_dtype == np.uint8:
            return self.rng.randint(0, 256, size=(self.num_envs,) + self.ob_shape).astype(np.uint8)
        return self.rng.randn(self.num_envs, *self.ob_shape).astype(np.float32)

    def reset(self):
        return self._obs()

    def step(self, actions):
        shape = (self.num_envs, self.reward_columns) if self.reward_columns else (self.num_envs,)
        rews = self.rng.randn(*shape).astype(np.float32)
        dones = self.rng.rand(self.num_envs) < self.done_p
        return self._obs(), rews, dones, [{} for _ in range(self.num_envs)]

    def close(self):
        pass


class DevicePointVecEnv:
    """`nenvs` point masses simulated in HBM (torch tensors): the observation is the position
    x in R^dim, an action moves it by 0.1*clip(a, -1, 1), the reward is -|x|^2 and an episode lasts
    `horizon` steps before the point is put back at a random place.  ``device_resident`` tells the
    runner that reset / step speak device tensors, so a whole rollout runs without leaving the GPU.
    Env side of the boundary: a test and benchmark fixture, not part of the update path."""

    device_resident = True

    def __init__(self, nenvs, dim=2, horizon=20, seed=0, reward_columns=None):
        import torch
        self.torch = torch
        self.num_envs, self.dim, self.horizon = nenvs, dim, horizon
        self.observation_space = Box(-np.inf, np.inf, shape=(dim,), dtype=np.float32)
        self.action_space = Box(-1.0, 1.0, shape=(dim,), dtype=np.float32)
        self.reward_columns = reward_columns
        self.gen = torch.Generator(device="cuda")
        self.gen.manual_seed(seed)
        self.x = torch.zeros((nenvs, dim), device="cuda")
        self.age = torch.zeros(nenvs, dtype=torch.int32, device="cuda")
        self.ep_ret = torch.zeros(nenvs, device="cuda")
        self.finished_returns = []          # device tensors of finished episodes' returns, one per step

    def reset(self):
        t = self.torch
        self.x = t.randn((self.num_envs, self.dim), device="cuda", generator=self.gen)
        self.age.zero_()
        self.ep_ret.zero_()
        return self.x.clone()

    def step(self, actions):
        t = self.torch
        self.x = self.x + 0.1 * actions.reshape(self.num_envs, self.dim).clamp(-1.0, 1.0)
        rew = -(self.x * self.x).sum(dim=1)
        self.age += 1
        self.ep_ret += rew
        dones = self.age >= self.horizon
        self.finished_returns.append(t.where(dones, self.ep_ret, t.full_like(self.ep_ret, float("nan"))))
        fresh = t.randn((self.num_envs, self.dim), device="cuda", generator=self.gen)
        self.x = t.where(dones[:, None], fresh, self.x)
        self.age = t.where(dones, t.zeros_like(self.age), self.age)
        self.ep_ret = t.where(dones, t.zeros_like(self.ep_ret), self.ep_ret)
        rews = rew if not self.reward_columns else rew[:, None].repeat(1, self.reward_columns).contiguous()
        return self.x.clone(), rews, dones, []

    def mean_finished_return(self, last=None):
        """Mean return of the episodes that ended in the last ``last`` steps (synchronising)."""
        t = self.torch
        rows = self.finished_returns[-last:] if last else self.finished_returns
        return float(t.nanmean(t.stack(rows)).item())

    def close(self):
        pass


class IdentityVecEnv:
    """`nenvs` copies of the reference's IdentityEnv (baselines/common/identity_env.py:5-30): the state is a
    random integer in [0, dim), the reward is 1 when the action repeats it, episodes never end.  The
    observation is the one-hot encoding the reference's input layer builds for a Discrete observation space."""

    def __init__(self, nenvs, dim, seed=0):
        self.num_envs, self.dim = nenvs, dim
        self.observation_space = Box(0.0, 1.0, shape=(dim,), dtype=np.float32)
        self.action_space = Discrete(dim)
        self.rng = np.random.RandomState(seed)
        self.state = np.zeros(nenvs, np.int64)

    def _obs(self):
        self.state = self.rng.randint(0, self.dim, size=self.num_envs)
        return np.eye(self.dim, dtype=np.float32)[self.state]

    def reset(self):
        return self._obs()

    def step(self, actions):
        rew = (np.asarray(actions).ravel() == self.state).astype(np.float32)
        return self._obs(), rew, np.zeros(self.num_envs, bool), [{} for _ in range(self.num_envs)]

    def close(self):
        pass
