"""The env-wrapper oracle against vectors the reference's own classes produced
(tests/golden/make_fixtures_envwrap.py)."""
import os

import numpy as np
import pytest

from oracle import envwrap

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gw():
    return np.load(os.path.join(HERE, "golden", "reference_env_wrappers.npz"))


def test_running_mean_std(gw):
    rms = envwrap.RunningMeanStd(shape=(3,))
    for k in range(3):
        rms.update(gw["rms_x%d" % k])
        np.testing.assert_array_equal(rms.mean, gw["rms_mean%d" % k])
        np.testing.assert_array_equal(rms.var, gw["rms_var%d" % k])
        assert rms.count == float(gw["rms_count%d" % k])
    # the reference's own test (running_mean_std.py:31-46): three updates equal one pass over the whole
    rng = np.random.RandomState(0)
    xs = [rng.randn(3, 2), rng.randn(4, 2), rng.randn(5, 2)]
    r2 = envwrap.RunningMeanStd(epsilon=0.0, shape=(2,))
    for x in xs:
        r2.update(x)
    allx = np.concatenate(xs)
    assert np.allclose([allx.mean(0), allx.var(0)], [r2.mean, r2.var])


def test_vec_normalize(gw):
    f = envwrap.VecNormalizeFilter(5, (4,), clipob=3.0, cliprew=2.0)
    np.testing.assert_array_equal(f.obfilt(gw["vn_first_obs"]), gw["vn_reset"])
    for k in range(6):
        o, r = f.step(gw["vn_in_obs%d" % k], gw["vn_in_rew%d" % k])
        np.testing.assert_array_equal(o, gw["vn_obs%d" % k])
        np.testing.assert_array_equal(r, gw["vn_rew%d" % k])
    np.testing.assert_array_equal(f.ret, gw["vn_ret"])
    np.testing.assert_array_equal(f.ob_rms.var, gw["vn_ob_var"])
    assert np.abs(gw["vn_obs3"]).max() == 3.0 and np.abs(gw["vn_rew3"]).max() == 2.0     # both clips were hit


@pytest.mark.parametrize("tag,shape,dtype,nstack", [("u8", (6, 5, 1), np.uint8, 4), ("f32", (3,), np.float32, 2)])
def test_frame_stack(gw, tag, shape, dtype, nstack):
    fs = envwrap.FrameStack(3, shape, dtype, nstack)
    np.testing.assert_array_equal(fs.reset(gw["fs_%s_obs0" % tag]), gw["fs_%s_reset" % tag])
    for k in range(6):
        out = fs.step(gw["fs_%s_in_obs%d" % (tag, k)], gw["fs_%s_in_new%d" % (tag, k)])
        np.testing.assert_array_equal(out, gw["fs_%s_out%d" % (tag, k)])
