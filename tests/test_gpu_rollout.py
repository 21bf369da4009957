"""The act half of a rollout step on the device: sampling kernel against the oracle's generator and
formulas, the fork's done-bootstrap, and the device-resident rollout loop against the host loop.  GPU only.

Parity statement for sampling: the reference draws inside TensorFlow, which nothing reproduces; the
generator here is held to its published known-answer vectors (tests/test_sampling_oracle.py) and the
device to the oracle's stream and formulas: uniforms bit-exact, Box-Muller normals to 2e-6 absolute,
actions / neglogp to 1e-5, argmax equal wherever the oracle's two best Gumbel scores are 1e-4 apart."""
import numpy as np
import pytest
import torch

from oracle import sampling, scan, tfgraph
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from baselines_amd import _lib, envs, hip, spaces
    from baselines_amd.ppo2 import common_functions as cf, policies, ppo2
    return dict(hip=hip, lib=_lib, spaces=spaces, envs=envs, cf=cf, policies=policies, ppo2=ppo2)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("B,A", [(1, 1), (7, 6), (1000, 3), (4096, 17), (65, 32)])
def test_gaussian_sampling(mods, B, A):
    hip, lib = mods["hip"], mods["lib"]
    spec = hip.PolicySpec(lib.POLICY_MLP, lib.DIST_GAUSSIAN, A, 1, ob_dim=5, hidden=(64, 64), order=lib.ORDER_UPSTREAM)
    rng = np.random.RandomState(B + A)
    theta = (rng.randn(spec.nparams) * 0.3).astype(np.float32)
    logstd = next(theta[off:off + A] for name, off, shp in spec.vars if name == "logstd")
    pi = rng.randn(B, A).astype(np.float32)
    seed, step = 0x1234567890ABCDEF, 41
    a, nlp, z = hip.sample_actions(spec, dev(theta), dev(pi), seed, step, want_noise=True)
    a, nlp, z = a.cpu().numpy(), nlp.cpu().numpy(), z.cpu().numpy()
    np.testing.assert_allclose(z, sampling.normals(B, A, seed, step), rtol=0, atol=2e-6)
    ea, enlp = sampling.sample_gaussian(pi, logstd, z)
    np.testing.assert_allclose(a, ea, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(nlp, enlp, rtol=1e-5, atol=1e-5)
    # the neglogp kernel of the update path sees the stored actions the same way
    np.testing.assert_allclose(hip.neglogp(spec, dev(theta), dev(pi), dev(a)).cpu().numpy(), nlp, rtol=1e-6, atol=1e-6)
    a2, _ = hip.sample_actions(spec, dev(theta), dev(pi), seed, step)
    assert np.array_equal(a2.cpu().numpy(), a), "(seed, step) names the noise"
    a3, _ = hip.sample_actions(spec, dev(theta), dev(pi), seed, step + 1)
    assert not np.array_equal(a3.cpu().numpy(), a)
    am, nlpm = hip.sample_actions(spec, dev(theta), dev(pi), seed, step, deterministic=True)
    assert np.array_equal(am.cpu().numpy(), pi)
    np.testing.assert_allclose(nlpm.cpu().numpy(), 0.5 * np.log(2 * np.pi) * A + logstd.sum(), rtol=1e-6)


@pytest.mark.parametrize("B,A", [(1, 2), (9, 5), (5000, 6), (300, 18)])
def test_categorical_sampling(mods, B, A):
    hip, lib = mods["hip"], mods["lib"]
    spec = hip.PolicySpec(lib.POLICY_MLP, lib.DIST_CATEGORICAL, A, 1, ob_dim=4, hidden=(64, 64), order=lib.ORDER_UPSTREAM)
    rng = np.random.RandomState(B * 3 + A)
    theta = dev((rng.randn(spec.nparams) * 0.1).astype(np.float32))
    logits = (rng.randn(B, A) * 2).astype(np.float32)
    seed, step = 99, 2 ** 33 + 5
    a, nlp, u = hip.sample_actions(spec, theta, dev(logits), seed, step, want_noise=True)
    a, nlp, u = a.cpu().numpy(), nlp.cpu().numpy(), u.cpu().numpy()
    np.testing.assert_array_equal(u, sampling.uniforms(B, A, seed, step))
    assert u.min() > 0.0 and u.max() < 1.0
    ea, _ = sampling.sample_categorical(logits, u)
    sc = np.sort(sampling.gumbel_scores(logits, u), axis=-1)
    clear = (sc[:, -1] - sc[:, -2] > 1e-4) if A > 1 else np.ones(B, bool)
    np.testing.assert_array_equal(a[clear], ea[clear])
    assert clear.mean() > 0.99
    l = logits.astype(np.float64)
    ref_nlp = np.log(np.exp(l - l.max(-1, keepdims=True)).sum(-1)) - (l[np.arange(B), a] - l.max(-1))
    np.testing.assert_allclose(nlp, ref_nlp, rtol=1e-5, atol=1e-5)
    am, _ = hip.sample_actions(spec, theta, dev(logits), seed, step, deterministic=True)
    np.testing.assert_array_equal(am.cpu().numpy(), logits.argmax(-1))


def test_sampling_statistics(mods):
    """Moments of the normal stream and a chi-square of Gumbel-max frequencies against softmax."""
    hip, lib = mods["hip"], mods["lib"]
    B, A = 200000, 6
    spec = hip.PolicySpec(lib.POLICY_MLP, lib.DIST_GAUSSIAN, A, 1, ob_dim=3, hidden=(64, 64), order=lib.ORDER_UPSTREAM)
    theta = torch.zeros(spec.nparams, device="cuda")
    _, _, z = hip.sample_actions(spec, theta, torch.zeros((B, A), device="cuda"), 7, 0, want_noise=True)
    z = z.double()
    n = z.numel()
    assert abs(z.mean().item()) < 4 / np.sqrt(n) and abs(z.var().item() - 1) < 6 * np.sqrt(2 / n)
    assert abs((z ** 3).mean().item()) < 6 * np.sqrt(15 / n) and abs((z ** 4).mean().item() - 3) < 6 * np.sqrt(96 / n)
    assert abs(torch.corrcoef(z.T)[0, 1].item()) < 5 / np.sqrt(B)
    cspec = hip.PolicySpec(lib.POLICY_MLP, lib.DIST_CATEGORICAL, A, 1, ob_dim=3, hidden=(64, 64), order=lib.ORDER_UPSTREAM)
    logits = np.array([0.5, -1.0, 2.0, 0.0, 1.0, -3.0], np.float32)
    p = np.exp(logits - logits.max())
    p /= p.sum()
    a, _ = hip.sample_actions(cspec, torch.zeros(cspec.nparams, device="cuda"), dev(np.tile(logits, (B, 1))), 11, 3)
    counts = np.bincount(a.cpu().numpy(), minlength=A)
    chi2 = ((counts - B * p) ** 2 / (B * p)).sum()
    assert chi2 < 30.0, (chi2, counts)         # 5 degrees of freedom: P(chi2 > 30) ~ 1.5e-5


@pytest.mark.parametrize("E,R", [(1, 5), (300, 5), (64, 3), (10, 1), (5000, 8)])
def test_done_bootstrap(mods, E, R):
    hip = mods["hip"]
    rng = np.random.RandomState(E + R)
    rewards = rng.randn(E, R).astype(np.float32)
    dones = rng.rand(E) < 0.4
    v = rng.randn(E, R).astype(np.float32)
    gamma = np.linspace(0.99, 0.9, R)
    want = sampling.done_bootstrap(rewards, dones, v, gamma)
    got = hip.done_bootstrap(dev(rewards), dev(dones), dev(v), gamma).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    assert not np.array_equal(got, rewards) or not dones.any()


class Replay:
    """Plays back prepared (obs, rewards, dones); as host arrays, or as device tensors with the
    ``device_resident`` mark."""

    def __init__(self, obs0, steps, ob_space, ac_space, device):
        self.num_envs, self.observation_space, self.action_space = obs0.shape[0], ob_space, ac_space
        self.obs0, self.steps, self.k, self.device_resident = obs0, steps, 0, device
        self.actions_seen = []

    def _o(self, a):
        return dev(a) if self.device_resident else a.copy()

    def reset(self):
        return self._o(self.obs0)

    def step(self, actions):
        self.actions_seen.append(actions.cpu().numpy() if isinstance(actions, torch.Tensor) else np.asarray(actions))
        o, r, d = self.steps[self.k % len(self.steps)]
        self.k += 1
        return self._o(o), self._o(r), self._o(d), ([] if self.device_resident else [{} for _ in range(self.num_envs)])

    def close(self):
        pass


@pytest.mark.parametrize("kind", ["fork_mlp", "upstream_mlp", "upstream_cat", "lstm"])
def test_device_rollout_equals_host_rollout(mods, kind):
    """Same env data, same sampling seed: the loop that never leaves HBM and the host loop of the
    reference's Runner collect identical rollouts (two consecutive runs, so carried obs / dones /
    recurrent states are covered)."""
    sp, pol, cf = mods["spaces"], mods["policies"], mods["cf"]
    rng = np.random.RandomState(3)
    T, E = 6, 8
    if kind == "lstm":
        ob_space, ac_space = sp.Box(0, 255, shape=(84, 84, 4), dtype=np.uint8), sp.Discrete(4)
        policy, draw = pol.LstmPolicy, lambda: rng.randint(0, 256, (E, 84, 84, 4)).astype(np.uint8)
    else:
        ob_space = sp.Box(-np.inf, np.inf, shape=(5,))
        ac_space = sp.Discrete(3) if kind == "upstream_cat" else sp.Box(-1, 1, shape=(2,))
        policy, draw = (pol.MlpPolicy if kind == "fork_mlp" else pol.MlpPolicy64), lambda: rng.randn(E, 5).astype(np.float32)
    fork = kind == "fork_mlp"
    R = 5 if fork else 1
    rew = lambda: rng.randn(E, R + 1).astype(np.float32) if fork else rng.randn(E).astype(np.float32)
    obs0 = draw()
    steps = [(draw(), rew(), rng.rand(E) < 0.3) for _ in range(2 * T)]
    gamma = [0.99, 0.98, 0.97, 0.96, 0.95] if fork else 0.99
    got = []
    for device in (False, True):
        pol.reset_scopes()
        np.random.seed(0)
        kw = dict(policy=policy, ob_space=ob_space, ac_space=ac_space, nbatch_act=E, nbatch_train=E * T, nsteps=T,
                  ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5)
        model = cf.Model(upstream_loss=not fork, **kw)
        pol.set_sample_seed(5)
        env = Replay(obs0, steps, ob_space, ac_space, device)
        runner = cf.Runner(env=env, model=model, nsteps=T, gamma=gamma, lam=0.95)
        runs = []
        for _ in range(2):
            out = runner.run()
            ro = runner.rollout
            runs.append({k: getattr(ro, k).cpu().numpy().copy() for k in
                         ("obs", "actions", "rewards", "values", "neglogpacs", "dones", "returns", "advs_gae", "last_values")})
            if device:
                assert all(x is None for x in out[:7])
            if kind == "lstm":
                runs[-1]["states"] = np.asarray(out[7])
        got.append((runs, env.actions_seen))
    (host_runs, host_actions), (dev_runs, dev_actions) = got
    for a, b in zip(host_actions, dev_actions):
        np.testing.assert_array_equal(np.asarray(a).reshape(b.shape), b)
    for hr, dr in zip(host_runs, dev_runs):
        for k in hr:
            np.testing.assert_array_equal(hr[k].reshape(dr[k].shape), dr[k], err_msg=k)
    assert dev_runs[0]["dones"].any() and not np.array_equal(dev_runs[0]["actions"], dev_runs[1]["actions"])


def test_learn_on_device_env_improves_return(mods):
    """ppo2.learn against an env that lives in HBM: no array of the rollout visits the host, and the
    point-mass task gets solved (the mean episode return rises well above the random policy's)."""
    envs, pol, ppo2 = mods["envs"], mods["policies"], mods["ppo2"]
    pol.reset_scopes()
    np.random.seed(1)
    pol.set_sample_seed(2)
    env = envs.DevicePointVecEnv(256, dim=2, horizon=20, seed=3)
    ppo2.learn(policy=pol.MlpPolicy64, env=env, nsteps=40, total_timesteps=256 * 40 * 40, ent_coef=0.0, lr=1e-3,
               nminibatches=4, noptepochs=4, cliprange=0.2, log_interval=1000)
    t = torch.stack(env.finished_returns)
    early = float(torch.nanmean(t[:80]).item())
    late = float(torch.nanmean(t[-80:]).item())
    print("device env: mean episode return %.2f (first two rollouts) -> %.2f (last two)" % (early, late))
    assert late > early + 10.0 and late > -25.0


def test_probtypes_consistency(mods):
    """The reference's own distribution test (baselines/common/distributions.py:259-307) on the device sampler
    and neglogp kernel, same pdparams and N: the mean negative log-likelihood of the samples equals the
    entropy, and -H[p] - E_p[log q] equals KL[p, q], each within three standard errors."""
    hip, lib = mods["hip"], mods["lib"]
    N = 100000
    rng = np.random.RandomState(0)

    def set_logstd(spec, logstd):
        theta = np.zeros(spec.nparams, np.float32)
        off = next(o for name, o, shp in spec.vars if name == "logstd")
        theta[off:off + len(logstd)] = logstd
        return dev(theta)

    # DiagGaussian: pdparam = [mean | logstd] (:263)
    pd = np.array([-.2, .3, .4, -.5, .1, -.5, .1, 0.8])
    mean, logstd = pd[:4], pd[4:]
    spec = hip.PolicySpec(lib.POLICY_MLP, lib.DIST_GAUSSIAN, 4, 1, ob_dim=3, hidden=(64, 64), order=lib.ORDER_UPSTREAM)
    theta_p = set_logstd(spec, logstd)
    pi_p = dev(np.repeat(mean[None].astype(np.float32), N, 0))
    x, nlp_p = hip.sample_actions(spec, theta_p, pi_p, 17, 0)
    ll = -nlp_p.double().cpu().numpy()
    ent = np.sum(logstd + 0.5 * np.log(2.0 * np.pi * np.e))                               # :209-210
    assert abs(ent - (-ll.mean())) < 3 * ll.std() / np.sqrt(N)
    q = pd + rng.randn(pd.size) * 0.1
    qm, qls = q[:4], q[4:]
    llq = -hip.neglogp(spec, set_logstd(spec, qls), dev(np.repeat(qm[None].astype(np.float32), N, 0)), x).double().cpu().numpy()
    kl = np.sum(qls - logstd + (np.exp(2 * logstd) + (mean - qm) ** 2) / (2.0 * np.exp(2 * qls)) - 0.5)   # :206-208
    assert abs(kl - (-ent - llq.mean())) < 3 * llq.std() / np.sqrt(N)

    # Categorical: pdparam = logits (:267)
    logits = np.array([-.2, .3, .5])
    cspec = hip.PolicySpec(lib.POLICY_MLP, lib.DIST_CATEGORICAL, 3, 1, ob_dim=3, hidden=(64, 64), order=lib.ORDER_UPSTREAM)
    theta_c = torch.zeros(cspec.nparams, device="cuda")
    xs, nlp_c = hip.sample_actions(cspec, theta_c, dev(np.repeat(logits[None].astype(np.float32), N, 0)), 23, 0)
    ll = -nlp_c.double().cpu().numpy()
    lsm = lambda l: l - l.max() - np.log(np.exp(l - l.max()).sum())
    p = np.exp(lsm(logits))
    ent = -(p * lsm(logits)).sum()                                                        # :157-162
    assert abs(ent - (-ll.mean())) < 3 * ll.std() / np.sqrt(N)
    ql = logits + rng.randn(3) * 0.1
    llq = -hip.neglogp(cspec, theta_c, dev(np.repeat(ql[None].astype(np.float32), N, 0)), xs).double().cpu().numpy()
    kl = (p * (lsm(logits) - lsm(ql))).sum()                                              # :147-156
    assert abs(kl - (-ent - llq.mean())) < 3 * llq.std() / np.sqrt(N)


def test_learn_through_device_wrappers(mods):
    """VecNormalize around an env that lives in HBM: observations and rewards are filtered on the device, the
    rollout loop stays device-resident, and learning still solves the task (returns are the env's own,
    unnormalised)."""
    envs, pol, ppo2 = mods["envs"], mods["policies"], mods["ppo2"]
    from baselines_amd.common.vec_env.vec_normalize import VecNormalize
    pol.reset_scopes()
    np.random.seed(1)
    pol.set_sample_seed(2)
    inner = envs.DevicePointVecEnv(256, dim=2, horizon=20, seed=3)
    env = VecNormalize(inner)
    assert env.device_resident
    ppo2.learn(policy=pol.MlpPolicy64, env=env, nsteps=40, total_timesteps=256 * 40 * 40, ent_coef=0.0, lr=1e-3,
               nminibatches=4, noptepochs=4, cliprange=0.2, log_interval=1000)
    t = torch.stack(inner.finished_returns)
    early, late = float(torch.nanmean(t[:80]).item()), float(torch.nanmean(t[-80:]).item())
    print("device env under VecNormalize: mean episode return %.2f -> %.2f" % (early, late))
    assert late > early + 10.0 and late > -25.0
    assert env.ob_rms.count > 256 * 40 * 39 and abs(float(env.ob_rms.var.mean())) > 0.0
