"""Parity at the full sizes of the benchmark configurations.  GPU only.

The scan, the permutation stream and the 64-64 MLP update have vectorised NumPy oracles that finish in
seconds even at these sizes, so the comparison is direct; on top of it come the size-independent
properties the multi-GPU layout relies on (env shards concatenate bit-exactly, composed index streams
stay permutations, a shuffled epoch visits every sample once)."""
import numpy as np
import pytest
import torch

from oracle import perm, scan, tfgraph
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from baselines_amd import hip, spaces
    from baselines_amd.ppo2 import common_functions as cf, policies
    return dict(hip=hip, spaces=spaces, cf=cf, policies=policies)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("T,E,shards", [(256, 2048, 2), (128, 65536, 8)])
def test_gae_full_size_and_env_shards(mods, T, E, shards):
    """[nsteps, nenvs] of the MuJoCo-shaped and of the 65536-env configuration: bit-exact against
    the oracle, and the scan over all envs equals the scans over env shards laid side by side (what
    sharding by env index over GPUs assumes: no value crosses an env column)."""
    hip = mods["hip"]
    rng = np.random.RandomState(T + E)
    rewards = rng.randn(T, E, 1).astype(np.float32)
    values = rng.randn(T, E, 1).astype(np.float32)
    dones = rng.rand(T, E) < 0.02
    last_values = rng.randn(E, 1).astype(np.float32)
    last_dones = rng.rand(E) < 0.02
    advs, rets, ori = hip.gae(dev(rewards), dev(values), dev(dones), dev(last_values), dev(last_dones), 0.99, 0.95)
    o_advs, o_rets = scan.gae(rewards, values, dones, last_values, last_dones, 0.99, 0.95)
    np.testing.assert_array_equal(advs.cpu().numpy(), o_advs)
    np.testing.assert_array_equal(rets.cpu().numpy(), o_rets)
    np.testing.assert_array_equal(ori.cpu().numpy(), o_rets - values)
    w = E // shards
    for s in range(shards):
        c = slice(s * w, (s + 1) * w)
        a_s, r_s, _ = hip.gae(dev(rewards[:, c]), dev(values[:, c]), dev(dones[:, c]), dev(last_values[c]),
                              dev(last_dones[c]), 0.99, 0.95)
        assert torch.equal(a_s, advs[:, c]) and torch.equal(r_s, rets[:, c])
    # lam = 0 leaves the one-step residual, all-done leaves r - v: closed forms with no carry
    a0, _, _ = hip.gae(dev(rewards), dev(values), dev(dones), dev(last_values), dev(last_dones), 0.99, 0.0)
    nextv = np.concatenate([values[1:], last_values[None]], 0).astype(np.float64)
    nextnt = 1.0 - np.concatenate([dones[1:], last_dones[None]], 0).astype(np.float64)[..., None]
    td = rewards.astype(np.float64) + 0.99 * nextv * nextnt - values.astype(np.float64)
    np.testing.assert_array_equal(a0.cpu().numpy(), td.astype(np.float32))
    ones = np.ones_like(dones)
    a1, _, _ = hip.gae(dev(rewards), dev(values), dev(ones), dev(last_values), dev(np.ones_like(last_dones)), 0.99, 0.95)
    np.testing.assert_array_equal(a1.cpu().numpy(), (rewards.astype(np.float64) - values.astype(np.float64)).astype(np.float32))


# 64 K columns and more take the column-pair kernel (two batches of 16 steps in flight) when the column count is even and
# the arrays are 8-byte aligned, the one-lane-per-column kernel otherwise: R = 1 and R = 5 (a pair may straddle two envs),
# step counts that end in a part-filled batch or fit inside one, an odd column count, an odd-offset done array, and the
# scan's own advantage array left out (Runner.run does not return it, baselines/ppo2/common_functions.py:344)
@pytest.mark.parametrize("T,E,R,case", [(40, 65536, 1, "pair"), (7, 70000, 1, "pair"), (33, 16384, 5, "pair"), (16, 13108, 5, "pair"),
                                        (19, 65537, 1, "odd columns"), (21, 65536, 1, "odd done offset"), (32, 13109, 5, "odd columns")])
def test_gae_wide_rollout_routes(mods, T, E, R, case):
    hip = mods["hip"]
    rng = np.random.RandomState(T * 7 + E + R)
    rewards = rng.randn(T, E, R).astype(np.float32)
    values = rng.randn(T, E, R).astype(np.float32)
    dones = rng.rand(T, E) < 0.05
    last_values = rng.randn(E, R).astype(np.float32)
    last_dones = rng.rand(E) < 0.05
    gamma = 0.99 if R == 1 else [0.99, 0.97, 0.97, 0.95, 0.95]
    d_dones = dev(dones)
    if case == "odd done offset":
        buf = torch.zeros(T * E + 1, dtype=torch.bool, device="cuda")
        d_dones = buf[1:].view(T, E)
        d_dones.copy_(dev(dones))
        assert d_dones.data_ptr() % 2 == 1
    o_advs, o_rets = scan.gae(rewards, values, dones, last_values, last_dones, gamma, 0.95)
    advs, rets, ori = hip.gae(dev(rewards), dev(values), d_dones, dev(last_values), dev(last_dones), gamma, 0.95)
    np.testing.assert_array_equal(advs.cpu().numpy(), o_advs)
    np.testing.assert_array_equal(rets.cpu().numpy(), o_rets)
    np.testing.assert_array_equal(ori.cpu().numpy(), o_rets - values)
    none, rets2, ori2 = hip.gae(dev(rewards), dev(values), d_dones, dev(last_values), dev(last_dones), gamma, 0.95, want_advs=False)
    assert none is None and torch.equal(rets2, rets) and torch.equal(ori2, ori)
    _, rets3, none = hip.gae(dev(rewards), dev(values), d_dones, dev(last_values), dev(last_dones), gamma, 0.95,
                             want_advs_ori=False, want_advs=False)
    assert none is None and torch.equal(rets3, rets)


@pytest.mark.parametrize("n,nepochs", [(2048 * 256, 4), (65536 * 128, 2)])
def test_index_stream_full_size(mods, n, nepochs):
    """The worker ring draws each epoch's shuffle relative to the identity and the device composes
    them; the result must be the array np.random.shuffle leaves after the same number of in-place
    shuffles from the same seed, and the generator must end in the same state."""
    hip = mods["hip"]
    sh = hip.LegacyShuffler(123)
    pipe = hip.PermPipeline(sh, n, nepochs)
    pipe.begin(nepochs, relative=True)
    ref_rng = np.random.RandomState(123)
    want = np.arange(n)
    cur = None
    for e in range(nepochs):
        sigma = pipe.wait(e)
        ref_rng.shuffle(want)
        sig_dev = sigma.cuda()
        if cur is None:
            cur = sig_dev.clone()
        else:
            nxt = torch.empty_like(cur)
            hip.compose(cur, sig_dev, nxt)
            cur = nxt
        got = cur.cpu().numpy()
        np.testing.assert_array_equal(got, want)
        assert np.array_equal(np.sort(got), np.arange(n)), "each epoch visits every sample exactly once"
    pipe.wait_all()
    key, pos = sh.get_state()
    st = ref_rng.get_state()
    assert pos == st[2] and np.array_equal(key, st[1])


def test_mujoco_shaped_update_full_size(mods):
    """obs_dim 17, act_dim 6 Gaussian MLP(64,64), 2048 envs x 256 steps, one epoch of 16 minibatches
    of 32768: losses and post-update parameters against the oracle's update loop on the same
    np.random stream, with the one-epoch parameter diff reported."""
    sp, pol, cf, hip = mods["spaces"], mods["policies"], mods["cf"], mods["hip"]
    D, A, T, E, B, nep = 17, 6, 256, 2048, 32768, 1
    ob_space, ac_space = sp.Box(-1, 1, shape=(D,)), sp.Box(-1, 1, shape=(A,))
    shapes = tfgraph.mlp_param_shapes(D, (64, 64), A, 1, True, "upstream")
    params = util.make_params(shapes, 41, pi_scale=0.5, jitter=0.02)
    rng = np.random.RandomState(42)
    ro = util.synthetic_mlp_rollout(rng, T, E, D, A, 1, True, params, nlp_noise=0.2)
    pol.reset_scopes()
    model = cf.Model(policy=pol.MlpPolicy64, ob_space=ob_space, ac_space=ac_space, nbatch_act=E, nbatch_train=B,
                     nsteps=T, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, upstream_loss=True)
    model.arena.set_dict(params)
    d = cf.DeviceRollout(T, E, dev(ro["obs"]), dev(ro["actions"]), dev(ro["rewards"]), dev(ro["values"]),
                         dev(ro["neglogpacs"]), dev(ro["dones"]), dev(ro["last_values"]), dev(ro["last_dones"]))
    d.advs_gae, d.returns, d.advs_ori = hip.gae(d.rewards, d.values, d.dones, d.last_values, d.last_dones, 0.99, 0.95)
    d.advs = d.advs_ori.reshape(T, E)
    np.random.seed(11)
    losses = model.train_on_rollout(3e-4, 0.2, d, nep, B).cpu().numpy()
    after_hip = np.random.randint(0, 2 ** 31 - 1)

    o_advs, o_rets = scan.gae(ro["rewards"], ro["values"], ro["dones"], ro["last_values"], ro["last_dones"], 0.99, 0.95)
    flat = {k: scan.sf01(v) for k, v in dict(obs=ro["obs"], ret=o_rets, act=ro["actions"], val=ro["values"],
                                             nlp=ro["neglogpacs"]).items()}
    adv = (flat["ret"] - flat["val"])[:, 0]
    oracle = tfgraph.TrainOracle(params, "mlp", "gaussian", tfgraph.LossCfg(0.0, 0.5, 0.5, True))
    np.random.seed(11)
    rows = []
    for ep in perm.minibatch_indices(T * E, B, nep):
        assert len(ep) == 16
        for mb in ep:
            rows.append(oracle.train(3e-4, 0.2, flat["obs"][mb], flat["ret"][mb], None, flat["act"][mb], flat["val"][mb],
                                     adv[mb], flat["nlp"][mb]))
    assert after_hip == np.random.randint(0, 2 ** 31 - 1)
    want = oracle.numpy_params()
    np.testing.assert_allclose(losses, np.mean(rows, axis=0), rtol=5e-5, atol=1e-7)
    got = model.arena.get_list()
    diff = max(float(np.abs(g.reshape(want[name].shape) - want[name]).max()) for (name, _, _), g in zip(model.spec.vars, got))
    moved = max(float(np.abs(want[k] - params[k]).max()) for k in params)
    print("one-epoch parameter diff vs oracle at 2048x256: max %.3e (parameters moved by up to %.3e)" % (diff, moved))
    assert diff <= 5e-6
    assert moved > 1e-3
