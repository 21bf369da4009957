"""Generate the golden vectors under tests/golden/ by RUNNING the reference's own
NumPy code (build container only; /root/reference never travels).

The reference's TensorFlow / gym / mpi4py imports are satisfied by empty stub
modules placed in ``sys.modules`` — only the NumPy half of the reference
executes (Runner.run bookkeeping + GAE, compute_advs, sf01, discount_with_dones,
ortho_init, add_vtarg_and_adv, discount_with_boundaries).  Permutation vectors
come from NumPy's legacy ``np.random`` itself, which is what the reference calls.

Usage:  python tests/golden/make_fixtures.py      (writes tests/golden/*.npz)
"""
import contextlib
import importlib
import importlib.util
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class _Stub(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        s = _Stub(self.__name__ + "." + n)
        setattr(self, n, s)
        return s

    def __call__(self, *a, **k):
        return _Stub("call")


def load_reference():
    for m in ["tensorflow", "gym", "gym.spaces", "mpi4py", "mpi4py.MPI"]:
        sys.modules[m] = _Stub(m)
    sys.path.insert(0, REF)
    mods = types.SimpleNamespace()
    mods.cf = importlib.import_module("baselines.ppo2.common_functions")
    mods.a2c_utils = importlib.import_module("baselines.a2c.utils")
    mods.ppo1 = importlib.import_module("baselines.ppo1.pposgd_simple")
    mods.math_util = importlib.import_module("baselines.common.math_util")
    try:
        mods.a2c = importlib.import_module("baselines.a2c.a2c")
    except Exception as e:  # pragma: no cover
        print("a2c.a2c not importable:", e)
        mods.a2c = None
    return mods


class FakeSpace:
    def __init__(self, shape, dtype=np.float32):
        self.shape = shape
        self.dtype = np.dtype(dtype)


class FakeVecEnv:
    """Synthetic VecEnv: N(0,1) observations, R+1 reward columns, Bernoulli dones."""

    def __init__(self, rng, nenv, obdim, nrew, done_p):
        self.rng, self.num_envs, self.obdim, self.nrew, self.done_p = rng, nenv, obdim, nrew, done_p
        self.observation_space = FakeSpace((obdim,))
        self.last_dones = None

    def reset(self):
        return self.rng.randn(self.num_envs, self.obdim).astype(np.float32)

    def step(self, actions):
        obs = self.rng.randn(self.num_envs, self.obdim).astype(np.float32)
        rews = self.rng.randn(self.num_envs, self.nrew + 1).astype(np.float32)
        dones = self.rng.rand(self.num_envs) < self.done_p
        self.last_dones = dones.copy()
        return obs, rews, dones, [{} for _ in range(self.num_envs)]


class FakeModel:
    """Stands in for the TF model: values/actions from the RandomState."""

    class _X:
        class dtype:
            name = "float32"

    def __init__(self, rng, nrew, acdim):
        self.rng, self.nrew, self.acdim = rng, nrew, acdim
        self.train_model = types.SimpleNamespace(X=self._X)
        self.initial_state = None
        self.value_calls = []

    def step(self, obs, states, dones):
        n = obs.shape[0]
        return (self.rng.randn(n, self.acdim).astype(np.float32),
                self.rng.randn(n, self.nrew).astype(np.float32), None,
                self.rng.rand(n).astype(np.float32))

    step_max = step

    def value(self, obs, *a):
        v = self.rng.randn(obs.shape[0], self.nrew).astype(np.float32)
        self.value_calls.append(v)
        return v


def un_sf01(flat, T, E):
    return flat.reshape(E, T, *flat.shape[1:]).swapaxes(0, 1).copy()


def gen_gae(mods, out):
    cases = [("small", 6, 4, 0.2, [.99] * 5), ("mid", 16, 32, 0.2, [.99, .97, .97, .95, .95]),
             ("cartpole_shape", 128, 8, 0.04, [.99] * 5), ("nodone", 12, 5, 0.0, [.99] * 5),
             ("alldone", 9, 3, 1.0, [.99, .97, .97, .95, .95])]
    for i, (name, T, E, p, gamma) in enumerate(cases):
        rng = np.random.RandomState(100 + i)
        env = FakeVecEnv(rng, E, 3, 5, p)
        model = FakeModel(rng, 5, 2)
        with contextlib.redirect_stdout(io.StringIO()):
            runner = mods.cf.Runner(env=env, model=model, nsteps=T, gamma=gamma, lam=0.95)
            res = runner.run()
        obs, returns, masks, actions, values, rewards, neglogpacs = res[:7]
        last_values = model.value_calls[-1]
        d = dict(T=T, E=E, gamma=np.asarray(gamma), lam=0.95,
                 rewards=un_sf01(rewards, T, E), values=un_sf01(values, T, E),
                 dones=un_sf01(masks, T, E), last_values=last_values, last_dones=env.last_dones,
                 returns_flat=returns, values_flat=values, obs_flat=obs, obs_tm=un_sf01(obs, T, E))
        # F2: fork advantage scalarisation on the same rollout (ppo2.py:92-95)
        advs_ori = returns - values
        with contextlib.redirect_stdout(io.StringIO()):
            d["advs_scalar"] = mods.cf.compute_advs([200, 120, 0, 0, 0], values, advs_ori, [], [])
        for k, v in d.items():
            out["gae_%s_%s" % (name, k)] = np.asarray(v)
    out["gae_cases"] = np.array([c[0] for c in cases])


def gen_gae_scalar(mods, out):
    """F1b: single-env scalar GAE of ppo1 (R=1 cross-check)."""
    rng = np.random.RandomState(7)
    for name, T, p in [("a", 20, 0.15), ("b", 64, 0.05), ("c", 5, 0.0)]:
        rew = rng.randn(T).astype(np.float32)
        vpred = rng.randn(T).astype(np.float32)
        new = (rng.rand(T) < p).astype(np.int32)
        last_done = bool(rng.rand() < 0.5)
        last_value = np.float32(rng.randn())
        seg = dict(rew=rew, vpred=vpred, new=new, nextvpred=last_value * (1 - last_done))
        mods.ppo1.add_vtarg_and_adv(seg, 0.99, 0.95)
        for k, v in dict(rew=rew, vpred=vpred, new=new, last_done=last_done, last_value=last_value,
                         adv=seg["adv"], tdlamret=seg["tdlamret"]).items():
            out["gae1_%s_%s" % (name, k)] = np.asarray(v)


def gen_discount(mods, out):
    rng = np.random.RandomState(11)
    dwd = mods.a2c_utils.discount_with_dones
    out["dwd_kat"] = np.asarray(dwd([1, 2, 3, 4], [0, 0, 1, 0], 0.9))
    E, T = 6, 5
    rewards = rng.randn(E, T).astype(np.float32)
    dones = rng.rand(E, T) < 0.3
    dones[0, -1] = True
    dones[1, -1] = False
    last_values = rng.randn(E).astype(np.float32)
    exp = np.empty_like(rewards)
    if mods.a2c is not None:
        # drive the real a2c Runner.run (a2c/a2c.py:91-131) with recorded env/model outputs
        class Env:
            num_envs = E
            observation_space = FakeSpace((2,), np.uint8)

            def __init__(self):
                self.t = 0

            def reset(self):
                return np.zeros((E, 2), np.uint8)

            def step(self, actions):
                r, d = rewards[:, self.t], dones[:, self.t]
                self.t += 1
                return np.zeros((E, 2), np.uint8), r, d, [{}] * E

        class Model:
            initial_state = None

            def step(self, obs, states, dn):
                return np.zeros(E, np.int32), np.zeros(E, np.float32), None, None

            def value(self, obs, states, dn):
                return last_values

        r = mods.a2c.Runner(Env(), Model(), nsteps=T, gamma=0.99)
        res = r.run()
        exp = res[2].reshape(E, T)
        out["a2c_source"] = np.array("a2c.Runner.run")
    else:
        for n in range(E):
            rw, dn = rewards[n].tolist(), dones[n].tolist()
            if dn[-1] == 0:
                rw = dwd(rw + [float(last_values[n])], dn + [0], 0.99)[:-1]
            else:
                rw = dwd(rw, dn, 0.99)
            exp[n] = rw
        out["a2c_source"] = np.array("discount_with_dones + restated bootstrap rule")
    out["a2c_rewards"], out["a2c_dones"], out["a2c_last_values"], out["a2c_returns"] = rewards, dones, last_values, exp
    # F6: the reference's own known-answer test, math_util.py:75-85
    x = np.array([1.0, 2.0, 3.0, 4.0], "float32")
    out["dwb_x"], out["dwb_starts"] = x, np.array([1.0, 0.0, 0.0, 1.0])
    out["dwb_y"] = mods.math_util.discount_with_boundaries(x, [1.0, 0.0, 0.0, 1.0], 0.9)
    mods.math_util.test_discount_with_boundaries()


def gen_perm(out):
    """F4: what ppo2.py:110-116 and :124-136 produce under np.random.seed(s)."""
    for seed in (0, 1, 1234):
        for n, b in ((1024, 256), (4096, 1024)):
            np.random.seed(seed)
            inds = np.arange(n)
            eps = []
            for _ in range(4):
                np.random.shuffle(inds)
                eps.append(inds.copy())
            out["perm_s%d_n%d" % (seed, n)] = np.stack(eps)
        np.random.seed(seed)
        E, T = 8, 16
        envinds = np.arange(E)
        flatinds = np.arange(E * T).reshape(E, T)
        eps, flat = [], []
        for _ in range(4):
            np.random.shuffle(envinds)
            eps.append(envinds.copy())
            flat.append(np.stack([flatinds[envinds[s:s + 2]].ravel() for s in range(0, E, 2)]))
        out["rperm_s%d_env" % seed] = np.stack(eps)
        out["rperm_s%d_flat" % seed] = np.stack(flat)
    np.random.seed(42)
    x = np.arange(100000)
    np.random.shuffle(x)
    out["perm_s42_n100000_head"] = x[:64].copy()
    out["perm_s42_n100000_sum"] = np.array(int((x * np.arange(100000)).sum() % (1 << 61)))
    out["perm_s42_after_u32"] = np.array(np.random.randint(0, 2 ** 31 - 1))


def gen_ortho(mods, out):
    """F5: ortho_init under np.random.seed(0), a2c/utils.py:25-40."""
    np.random.seed(0)
    for name, shape, scale in [("fc_4_64", (4, 64), np.sqrt(2)), ("fc_64_64", (64, 64), np.sqrt(2)),
                               ("fc_64_2", (64, 2), 0.01), ("conv_8_8_4_32", (8, 8, 4, 32), np.sqrt(2)),
                               ("fc_64_1", (64, 1), 1.0)]:
        out["ortho_" + name] = mods.a2c_utils.ortho_init(scale)(shape, np.float32)


def main():
    mods = load_reference()
    out = {}
    gen_gae(mods, out)
    gen_gae_scalar(mods, out)
    gen_discount(mods, out)
    gen_perm(out)
    gen_ortho(mods, out)
    path = os.path.join(HERE, "reference_numpy_half.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays; numpy", np.__version__)


if __name__ == "__main__":
    main()
