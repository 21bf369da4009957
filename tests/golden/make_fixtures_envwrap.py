"""Golden vectors for the env-wrapper filters, made by RUNNING the reference's own NumPy classes
(RunningMeanStd, VecNormalize, VecFrameStack) on scripted venv outputs.  Build container only;
the reference never travels.  gym is satisfied by an empty stub: only ``spaces.Box(...)`` is called
on it, and its result is not used.

Usage:  python tests/golden/make_fixtures_envwrap.py   (writes tests/golden/reference_env_wrappers.npz)
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_fixtures import REF, _Stub  # noqa: E402


class Space:
    def __init__(self, low, high, dtype):
        self.low, self.high, self.dtype, self.shape = low, high, np.dtype(dtype), low.shape


class ScriptedVecEnv:
    """Plays back a prepared sequence of (obs, rews, news)."""

    def __init__(self, obs0, steps, space):
        self.num_envs = obs0.shape[0]
        self.observation_space, self.action_space = space, None
        self.obs0, self.steps, self.k = obs0, steps, 0

    def reset(self):
        return self.obs0

    def step_async(self, actions):
        pass

    def step_wait(self):
        o, r, n = self.steps[self.k]
        self.k += 1
        return o, r, n, [{} for _ in range(self.num_envs)]

    def close(self):
        pass


def main():
    for m in ["tensorflow", "gym", "gym.spaces", "mpi4py", "mpi4py.MPI"]:
        sys.modules[m] = _Stub(m)
    sys.path.insert(0, REF)
    rms_mod = importlib.import_module("baselines.common.running_mean_std")
    vn_mod = importlib.import_module("baselines.common.vec_env.vec_normalize")
    fs_mod = importlib.import_module("baselines.common.vec_env.vec_frame_stack")
    out = {}
    rng = np.random.RandomState(5)

    # RunningMeanStd: three updates of different sizes, far-off mean in one column
    rms = rms_mod.RunningMeanStd(shape=(3,))
    for k, n in enumerate((7, 64, 1)):
        x = (rng.randn(n, 3) * [1.0, 0.01, 30.0] + [0.0, 1000.0, -5.0]).astype(np.float32 if k < 2 else np.float64)
        rms.update(x)
        out["rms_x%d" % k], out["rms_mean%d" % k], out["rms_var%d" % k] = x, rms.mean.copy(), rms.var.copy()
        out["rms_count%d" % k] = np.array(rms.count)

    # VecNormalize: 6 steps, 5 envs, 4-dim float32 obs, float64 rewards
    E, D, K = 5, 4, 6
    space = Space(np.full((D,), -np.inf, np.float32), np.full((D,), np.inf, np.float32), np.float32)
    obs0 = (rng.randn(E, D) * 3 + 1).astype(np.float32)
    steps = [((rng.randn(E, D) * 3 + 1).astype(np.float32), rng.randn(E) * 2, rng.rand(E) < 0.2) for _ in range(K)]
    steps[3] = (steps[3][0] * 100, steps[3][1] * 100, steps[3][2])          # drives both clips
    venv = vn_mod.VecNormalize(ScriptedVecEnv(obs0, steps, space), clipob=3.0, cliprew=2.0)
    out["vn_first_obs"], out["vn_reset"] = obs0, venv.reset()
    for k in range(K):
        o, r, n, _ = venv.step(None)
        out["vn_in_obs%d" % k], out["vn_in_rew%d" % k], out["vn_in_new%d" % k] = steps[k]
        out["vn_obs%d" % k], out["vn_rew%d" % k] = o, r
    out["vn_ob_mean"], out["vn_ob_var"], out["vn_ob_count"] = venv.ob_rms.mean, venv.ob_rms.var, np.array(venv.ob_rms.count)
    out["vn_ret_mean"], out["vn_ret_var"], out["vn_ret"] = np.array(venv.ret_rms.mean), np.array(venv.ret_rms.var), venv.ret

    # VecFrameStack: uint8 frames [E, 6, 5, 1] stacked 4 deep, and float32 vectors [E, 3] stacked 2 deep
    for tag, shape, dtype, nstack in (("u8", (6, 5, 1), np.uint8, 4), ("f32", (3,), np.float32, 2)):
        E, K = 3, 6
        hi = 255 if dtype == np.uint8 else 1
        space = Space(np.zeros(shape, dtype), np.full(shape, hi, dtype), dtype)
        draw = lambda: (rng.randint(0, 256, (E,) + shape).astype(dtype) if dtype == np.uint8
                        else rng.randn(*((E,) + shape)).astype(dtype))
        obs0 = draw()
        steps = [(draw(), rng.randn(E), rng.rand(E) < 0.3) for _ in range(K)]
        venv = fs_mod.VecFrameStack(ScriptedVecEnv(obs0, steps, space), nstack)
        out["fs_%s_obs0" % tag], out["fs_%s_reset" % tag] = obs0, venv.reset().copy()
        for k in range(K):
            o, _, _, _ = venv.step(None)
            out["fs_%s_in_obs%d" % (tag, k)], out["fs_%s_in_new%d" % (tag, k)] = steps[k][0], steps[k][2]
            out["fs_%s_out%d" % (tag, k)] = o.copy()
    rms_mod.test_runningmeanstd()
    path = os.path.join(HERE, "reference_env_wrappers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays; numpy", np.__version__)


if __name__ == "__main__":
    main()
