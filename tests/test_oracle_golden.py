"""Pins the CPU oracle against vectors produced by the reference's own NumPy code
(tests/golden/make_fixtures.py) and against NumPy's legacy RNG."""
import numpy as np
import pytest

from oracle import advs as oadvs
from oracle import clib, perm, scan
from oracle.init import ortho_init

CASES = ["small", "mid", "cartpole_shape", "nodone", "alldone"]


def _case(golden, name):
    g = lambda k: golden["gae_%s_%s" % (name, k)]
    return g


@pytest.mark.parametrize("name", CASES)
def test_gae_matches_reference_runner(golden, name):
    g = _case(golden, name)
    advs, rets = scan.gae(g("rewards"), g("values"), g("dones"), g("last_values"), g("last_dones"),
                          g("gamma"), float(g("lam")))
    np.testing.assert_array_equal(scan.sf01(rets), g("returns_flat"))      # bit-exact
    np.testing.assert_array_equal(scan.sf01(g("values")), g("values_flat"))
    np.testing.assert_array_equal(scan.sf01(g("obs_tm")), g("obs_flat"))
    a2, r2 = clib.gae(g("rewards"), g("values"), g("dones"), g("last_values"), g("last_dones"),
                      g("gamma"), float(g("lam")))
    np.testing.assert_array_equal(r2, rets)
    np.testing.assert_array_equal(a2, advs)


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_gae_scalar_matches_ppo1(golden, name):
    g = lambda k: golden["gae1_%s_%s" % (name, k)]
    rew, vpred, new = g("rew"), g("vpred"), g("new")
    T = len(rew)
    # map the single-env segment onto the [T, E=1, R=1] layout of the ppo2 scan
    advs, rets = scan.gae(rew.reshape(T, 1, 1), vpred.reshape(T, 1, 1), new.reshape(T, 1).astype(bool),
                          np.array([[g("last_value")]], np.float32), np.array([bool(g("last_done"))]),
                          0.99, 0.95)
    np.testing.assert_allclose(advs.ravel(), g("adv"), rtol=0, atol=1e-6)
    np.testing.assert_allclose(rets.ravel(), g("tdlamret"), rtol=0, atol=1e-6)
    a1, r1 = scan.gae_single_env(rew, vpred, new, g("last_value") * (1 - bool(g("last_done"))), 0.99, 0.95)
    np.testing.assert_array_equal(a1, g("adv"))
    np.testing.assert_array_equal(r1, g("tdlamret"))


@pytest.mark.parametrize("name", CASES)
def test_compute_advs_matches_reference(golden, name, capsys):
    g = _case(golden, name)
    values, returns = g("values_flat"), g("returns_flat")
    out = oadvs.compute_advs([200, 120, 0, 0, 0], values, returns - values)
    np.testing.assert_array_equal(out, g("advs_scalar"))


def test_discount_with_dones(golden):
    assert np.allclose(scan.discount_with_dones([1, 2, 3, 4], [0, 0, 1, 0], 0.9), golden["dwd_kat"])
    assert np.allclose(golden["dwd_kat"], [5.23, 4.7, 3.0, 4.0])
    exp = golden["a2c_returns"]
    got = scan.a2c_returns(golden["a2c_rewards"], golden["a2c_dones"], golden["a2c_last_values"], 0.99)
    np.testing.assert_array_equal(got, exp)
    got_c = clib.discount(golden["a2c_rewards"], golden["a2c_dones"], golden["a2c_last_values"], 0.99)
    np.testing.assert_array_equal(got_c, exp)


def test_discount_with_boundaries_kat(golden):
    # the reference's own known-answer test, baselines/common/math_util.py:75-85
    gamma = 0.9
    y = scan.discount_with_boundaries(golden["dwb_x"], golden["dwb_starts"], gamma)
    assert np.allclose(y, [1 + gamma * 2 + gamma ** 2 * 3, 2 + gamma * 3, 3, 4])
    np.testing.assert_array_equal(y, golden["dwb_y"])


@pytest.mark.parametrize("seed", [0, 1, 1234])
@pytest.mark.parametrize("n,b", [(1024, 256), (4096, 1024)])
def test_permutation_matches_numpy(golden, seed, n, b):
    exp = golden["perm_s%d_n%d" % (seed, n)]
    # pure-Python restatement of the legacy RNG
    gen = perm.MT19937(seed)
    x = list(range(n))
    for e in range(2):
        perm.legacy_shuffle(gen, x)
        np.testing.assert_array_equal(np.asarray(x), exp[e])
    # C restatement, all four cumulative epochs
    np.testing.assert_array_equal(clib.shuffle_epochs(seed, n, 4), exp)
    # the learn() loop structure on NumPy itself
    np.random.seed(seed)
    eps = perm.minibatch_indices(n, b, 4)
    for e in range(4):
        np.testing.assert_array_equal(np.concatenate(eps[e]), exp[e])
        assert all(len(m) == b for m in eps[e])


@pytest.mark.parametrize("seed", [0, 1, 1234])
def test_recurrent_indices(golden, seed):
    np.random.seed(seed)
    eps = perm.recurrent_minibatch_indices(8, 16, 2 * 16, 4)
    for e in range(4):
        np.testing.assert_array_equal(np.concatenate([m[0] for m in eps[e]]), golden["rperm_s%d_env" % seed][e])
        np.testing.assert_array_equal(np.stack([m[1] for m in eps[e]]), golden["rperm_s%d_flat" % seed][e])


def test_permutation_large_and_stream_position(golden):
    x = clib.shuffle_epochs(42, 100000, 1)[0]
    np.testing.assert_array_equal(x[:64], golden["perm_s42_n100000_head"])
    assert int((x * np.arange(100000)).sum() % (1 << 61)) == int(golden["perm_s42_n100000_sum"])
    # MT state hand-off: after the shuffle NumPy's next draw must be reproducible from its state
    np.random.seed(42)
    y = np.arange(100000)
    np.random.shuffle(y)
    gen = perm.MT19937.from_numpy_state(np.random.get_state())
    np.random.seed(42)
    np.random.shuffle(np.arange(100000))
    assert gen.next_u32() == np.random.randint(0, 2 ** 32, dtype=np.uint64)


def test_ortho_init_matches_reference(golden):
    np.random.seed(0)
    for name, shape, scale in [("fc_4_64", (4, 64), np.sqrt(2)), ("fc_64_64", (64, 64), np.sqrt(2)),
                               ("fc_64_2", (64, 2), 0.01), ("conv_8_8_4_32", (8, 8, 4, 32), np.sqrt(2)),
                               ("fc_64_1", (64, 1), 1.0)]:
        np.testing.assert_array_equal(ortho_init(shape, scale), golden["ortho_" + name])


def test_single_env_gae_float64_promotion(golden):
    """The [T,1,1] case of the rollout scan IS ppo1's segment scan under value-based scalar promotion
    (float64 carry); the fixtures, made under a NumPy with weak Python scalars, lie within 1e-6."""
    for name in ["a", "b", "c"]:
        g = lambda k: golden["gae1_%s_%s" % (name, k)]
        rew, vpred, new = g("rew"), g("vpred"), g("new")
        T = len(rew)
        nextv = g("last_value") * (1 - bool(g("last_done")))
        a64, r64 = scan.gae_single_env_f64(rew, vpred, new, nextv, 0.99, 0.95)
        np.testing.assert_allclose(a64, g("adv"), rtol=0, atol=1e-6)
        advs, rets = scan.gae(rew.reshape(T, 1, 1), vpred.reshape(T, 1, 1), new.reshape(T, 1).astype(bool),
                              np.array([[nextv]], np.float32), np.array([False]), 0.99, 0.95)
        np.testing.assert_array_equal(advs.ravel(), a64)
        np.testing.assert_array_equal(rets.ravel(), r64)


@pytest.mark.parametrize("shape,dtype", [((1,), np.float32), ((50,), np.float32), ((33, 4), np.float64), ((7, 2, 3), np.float32)])
def test_discount_is_scipy_lfilter(shape, dtype):
    """math_util.discount is one scipy call; the restatement must reproduce it bit for bit."""
    import scipy.signal
    x = np.random.RandomState(len(shape)).randn(*shape).astype(dtype)
    ref = scipy.signal.lfilter([1], [1, -0.97], x[::-1], axis=0)[::-1]
    got = scan.discount(x, 0.97)
    assert got.dtype == ref.dtype == np.float64
    np.testing.assert_array_equal(got, ref)


def test_discount_with_boundaries_legacy_matches_kat(golden):
    y = scan.discount_with_boundaries_legacy(golden["dwb_x"], golden["dwb_starts"], 0.9)
    np.testing.assert_allclose(y, golden["dwb_y"], rtol=0, atol=1e-6)
