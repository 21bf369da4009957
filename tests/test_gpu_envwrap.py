"""Device forms of RunningMeanStd / VecNormalize / VecFrameStack against the oracle and against the
vectors the reference's classes produced.  GPU only.

Tolerances: the reference accumulates a float32 batch's mean and variance in float32 (NumPy's own
reduction), the device in float64; against the float64-moment oracle the statistics agree to 1e-12
relative, against the reference's vectors to what float32 accumulation leaves (stated per check)."""
import os

import numpy as np
import pytest
import torch

from oracle import envwrap

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gw():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return np.load(os.path.join(HERE, "golden", "reference_env_wrappers.npz"))


class Scripted:
    def __init__(self, obs0, steps, ob_space, device=False):
        self.num_envs, self.observation_space, self.action_space = obs0.shape[0], ob_space, None
        self.obs0, self.steps, self.k, self.device = obs0, steps, 0, device

    def _o(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda() if self.device else a

    def reset(self):
        return self._o(self.obs0)

    def step(self, actions):
        o, r, n = self.steps[self.k]
        self.k += 1
        if self.device:
            return self._o(o), self._o(np.asarray(r, np.float32)), self._o(np.asarray(n)), [{}] * self.num_envs
        return o, r, n, [{}] * self.num_envs

    def close(self):
        pass


def test_running_mean_std_device(gw):
    from baselines_amd.common.running_mean_std import RunningMeanStd
    rms = RunningMeanStd(shape=(3,))
    o64 = envwrap.RunningMeanStd(shape=(3,))
    for k in range(3):
        x = gw["rms_x%d" % k]
        rms.update(x)
        o64.update(x, np.float64)
        np.testing.assert_allclose(rms.mean, o64.mean, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(rms.var, o64.var, rtol=1e-10)
        # float32 accumulation in the reference: column 1 sits at 1000 +- 0.01, at float32's resolution
        np.testing.assert_allclose(rms.mean, gw["rms_mean%d" % k], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(rms.var, gw["rms_var%d" % k], rtol=2e-3)
        assert rms.count == float(gw["rms_count%d" % k])


@pytest.mark.parametrize("n,d", [(1, 1), (5, 300), (4096, 17), (100000, 1), (257, 1000), (2048, 6)])
def test_rms_shapes(n, d):
    from baselines_amd import hip
    rng = np.random.RandomState(n + d)
    o = envwrap.RunningMeanStd(shape=(d,))
    st = hip.rms_state(d)
    for _ in range(3):
        x = (rng.randn(n, d) * rng.rand(d) * 5 + rng.randn(d) * 50).astype(np.float32)
        hip.rms_update(st, torch.from_numpy(x).cuda())
        o.update(x, np.float64)
    got = st.cpu().numpy()
    np.testing.assert_allclose(got[:d], o.mean, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(got[d:2 * d], o.var, rtol=1e-9)
    np.testing.assert_allclose(got[2 * d:], o.count, rtol=1e-15)
    from baselines_amd._lib import Ppo2HipError
    with pytest.raises(Ppo2HipError, match="empty batch"):
        hip.rms_update(st, torch.zeros((0, d), device="cuda"))


@pytest.mark.parametrize("device", [False, True])
def test_vec_normalize_device(gw, device):
    from baselines_amd import spaces
    from baselines_amd.common.vec_env.vec_normalize import VecNormalize
    steps = [(gw["vn_in_obs%d" % k], gw["vn_in_rew%d" % k], gw["vn_in_new%d" % k]) for k in range(6)]
    venv = VecNormalize(Scripted(gw["vn_first_obs"], steps, spaces.Box(-np.inf, np.inf, shape=(4,)), device),
                        clipob=3.0, cliprew=2.0)
    f64 = envwrap.VecNormalizeFilter(5, (4,), clipob=3.0, cliprew=2.0, moments_dtype=np.float64)
    host = lambda a: a.cpu().numpy() if isinstance(a, torch.Tensor) else a
    # device venvs hand float32 rewards over, so the oracle sees the same rounded rewards there
    rew_in = (lambda r: np.asarray(r, np.float32).astype(np.float64)) if device else (lambda r: r)
    tol = dict(rtol=1e-6, atol=1e-6) if device else dict(rtol=1e-12, atol=1e-12)
    o0 = host(venv.reset())
    np.testing.assert_allclose(o0, f64.obfilt(gw["vn_first_obs"]), **tol)
    np.testing.assert_allclose(o0, gw["vn_reset"], rtol=1e-5, atol=1e-5)
    for k in range(6):
        o, r, n, _ = venv.step(None)
        assert isinstance(o, torch.Tensor) == device and isinstance(r, torch.Tensor) == device
        eo, er = f64.step(steps[k][0], rew_in(steps[k][1]))
        np.testing.assert_allclose(host(o), eo, **tol)
        np.testing.assert_allclose(host(r), er, **tol)
        np.testing.assert_allclose(host(o), gw["vn_obs%d" % k], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(host(r), gw["vn_rew%d" % k], rtol=1e-5, atol=1e-5)
    assert np.abs(host(o)).max() <= 3.0 and np.abs(host(r)).max() <= 2.0
    np.testing.assert_allclose(venv.ob_rms.var, gw["vn_ob_var"], rtol=1e-5)
    np.testing.assert_allclose(venv.ret, f64.ret, rtol=1e-12)
    if not device:
        np.testing.assert_allclose(venv.ret_rms.var, gw["vn_ret_var"], rtol=1e-12)


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("tag,shape,dtype,nstack", [("u8", (6, 5, 1), np.uint8, 4), ("f32", (3,), np.float32, 2)])
def test_frame_stack_device(gw, tag, shape, dtype, nstack, device):
    from baselines_amd import spaces
    from baselines_amd.common.vec_env.vec_frame_stack import VecFrameStack
    steps = [(gw["fs_%s_in_obs%d" % (tag, k)], np.zeros(3), gw["fs_%s_in_new%d" % (tag, k)]) for k in range(6)]
    space = spaces.Box(0, 255 if dtype == np.uint8 else 1, shape=shape, dtype=dtype)
    venv = VecFrameStack(Scripted(gw["fs_%s_obs0" % tag], steps, space, device), nstack)
    assert venv.observation_space.shape == shape[:-1] + (shape[-1] * nstack,)
    host = lambda a: a.cpu().numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_array_equal(host(venv.reset()), gw["fs_%s_reset" % tag])
    for k in range(6):
        o, _, _, _ = venv.step(None)
        assert host(o).dtype == dtype
        np.testing.assert_array_equal(host(o), gw["fs_%s_out%d" % (tag, k)])


def test_frame_stack_atari_size():
    """84x84 frames, 4 deep, 256 envs: the roll against NumPy's at the size the Atari rollout uses."""
    from baselines_amd import hip
    rng = np.random.RandomState(1)
    E = 256
    fs = envwrap.FrameStack(E, (84, 84, 1), np.uint8, 4)
    st = torch.zeros((E, 84, 84, 4), dtype=torch.uint8, device="cuda")
    for k in range(5):
        obs = rng.randint(0, 256, (E, 84, 84, 1)).astype(np.uint8)
        news = rng.rand(E) < 0.1
        want = fs.step(obs, news)
        hip.frame_stack(st, torch.from_numpy(obs).cuda(), torch.from_numpy(news).cuda())
        np.testing.assert_array_equal(st.cpu().numpy(), want)
