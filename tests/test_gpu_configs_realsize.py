"""The benchmark configurations at their real sizes under a checker.  GPU only.

  * MLP(64,64) updates of the 2048-env x 256-step rollout (minibatches of 131 072) and of one GPU's shard of the
    65536-env x 128-step rollout (8192 envs, minibatches of 262 144): losses and post-update parameters against
    the float64 oracle's update loop on the same np.random stream.
  * nature_cnn, minibatch of 8192 out of a 256-env x 128-step uint8 rollout: the whole minibatch against the
    oracle, and against the fixed-order sum of the same call on 8 chunks of 1024 rows (different tile / slab
    choices in every kernel).
  * nature_cnn + LSTM(256), env-sequence minibatches with the full 128-step depth and done-mask resets at
    p = 1/20: 16 envs x 128 steps against the oracle, 128 envs x 128 steps against the sum over 8 env chunks.

The nature_cnn comparisons hand the implementation's own rectifier on/off pattern to the restatement (see
_rectifier_pattern) and check the rectified activations themselves to 2e-5.

Tolerances: losses rtol 5e-5; gradients 5e-5 of the per-variable max; post-update parameters atol 5e-6."""
import numpy as np
import pytest
import torch

from oracle import perm, scan, tfgraph
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from baselines_amd import _lib, hip, spaces
    from baselines_amd.ppo2 import common_functions as cf, policies
    return dict(hip=hip, spaces=spaces, cf=cf, policies=policies, lib=_lib)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("T,E,B,nep", [(256, 2048, 131072, 2), (128, 8192, 262144, 1)])
def test_mlp64_update_at_benchmark_minibatch_sizes(mods, T, E, B, nep):
    sp, pol, cf, hip = mods["spaces"], mods["policies"], mods["cf"], mods["hip"]
    D, A = 17, 6
    shapes = tfgraph.mlp_param_shapes(D, (64, 64), A, 1, True, "upstream")
    params = util.make_params(shapes, 51, pi_scale=0.5, jitter=0.02)
    ro = util.synthetic_mlp_rollout(np.random.RandomState(52), T, E, D, A, 1, True, params, nlp_noise=0.2)
    pol.reset_scopes()
    model = cf.Model(policy=pol.MlpPolicy64, ob_space=sp.Box(-1, 1, shape=(D,)), ac_space=sp.Box(-1, 1, shape=(A,)),
                     nbatch_act=E, nbatch_train=B, nsteps=T, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, upstream_loss=True)
    model.arena.set_dict(params)
    d = cf.DeviceRollout(T, E, dev(ro["obs"]), dev(ro["actions"]), dev(ro["rewards"]), dev(ro["values"]),
                         dev(ro["neglogpacs"]), dev(ro["dones"]), dev(ro["last_values"]), dev(ro["last_dones"]))
    d.advs_gae, d.returns, d.advs_ori = hip.gae(d.rewards, d.values, d.dones, d.last_values, d.last_dones, 0.99, 0.95)
    d.advs = d.advs_ori.reshape(T, E)
    np.random.seed(13)
    losses = model.train_on_rollout(3e-4, 0.2, d, nep, B).cpu().numpy()

    _, o_rets = scan.gae(ro["rewards"], ro["values"], ro["dones"], ro["last_values"], ro["last_dones"], 0.99, 0.95)
    flat = {k: scan.sf01(v) for k, v in dict(obs=ro["obs"], ret=o_rets, act=ro["actions"], val=ro["values"],
                                             nlp=ro["neglogpacs"]).items()}
    adv = (flat["ret"] - flat["val"])[:, 0]
    oracle = tfgraph.TrainOracle(params, "mlp", "gaussian", tfgraph.LossCfg(0.0, 0.5, 0.5, True))
    np.random.seed(13)
    rows = [oracle.train(3e-4, 0.2, flat["obs"][mb], flat["ret"][mb], None, flat["act"][mb], flat["val"][mb], adv[mb],
                         flat["nlp"][mb]) for ep in perm.minibatch_indices(T * E, B, nep) for mb in ep]
    assert len(rows) == nep * (T * E // B)
    np.testing.assert_allclose(losses, np.mean(rows, axis=0), rtol=5e-5, atol=1e-7)
    want = oracle.numpy_params()
    diff = max(float(np.abs(g.reshape(want[name].shape) - want[name]).max())
               for (name, _, _), g in zip(model.spec.vars, model.arena.get_list()))
    print("parameter diff vs oracle after %d minibatches of %d: max %.3e" % (len(rows), B, diff))
    assert diff <= 5e-6


def _normalised(adv, stats):
    """(a - mean) / (std + 1e-8) with the minibatch's {sum, sumsq, count} (baselines/ppo2/common_functions.py:67)."""
    s, ss, n = [float(x) for x in stats]
    mean = s / n
    std = np.sqrt(max(ss / n - mean * mean, 0.0))
    return (adv.astype(np.float64) - mean) / (std + 1e-8)


def _check_grads(spec, got, want, what, tol=5e-5):
    for name, off, shp in spec.vars:
        n = int(np.prod(shp))
        w, g = want[off:off + n], got[off:off + n]
        bound = tol * max(float(np.abs(w).max()), 1e-6)
        err = float(np.abs(g - w).max())
        assert err <= bound, (what, name, err, bound)


def _device_image_rollout(hip, spec, theta, T, E, hwc, nact, seed, done_p, nlstm=0):
    """uint8 frames drawn on the device; actions, values and old neglogp from the policy itself (stepped through
    time with its own state hand-over for the recurrent policy), plus noise on values and neglogp so that the
    clipping branches of the loss are all taken."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    H, W, C = hwc
    obs = torch.randint(0, 256, (T, E, H, W, C), generator=g, device="cuda", dtype=torch.uint8)
    dones = (torch.rand((T, E), generator=g, device="cuda") < done_p).to(torch.uint8)
    ws = torch.empty(spec.forward_workspace_bytes(E), dtype=torch.uint8, device="cuda")
    actions = torch.empty((T, E), dtype=torch.int32, device="cuda")
    values = torch.empty((T, E, 1), device="cuda")
    nlp = torch.empty((T, E), device="cuda")
    states0 = (0.3 * torch.randn((E, 2 * nlstm), generator=g, device="cuda")) if nlstm else None
    state = states0
    for t in range(T):
        if nlstm:
            pi, vf, state = hip.policy_forward(spec, theta, obs[t], ws, masks=dones[t], states=state)
        else:
            pi, vf, _ = hip.policy_forward(spec, theta, obs[t], ws)
        u = torch.rand((E, nact), generator=g, device="cuda").clamp_(1e-12, 1.0)
        a = torch.argmax(pi - torch.log(-torch.log(u)), dim=1).to(torch.int32)
        actions[t] = a
        values[t] = vf.reshape(E, 1) + 0.3 * torch.randn((E, 1), generator=g, device="cuda")
        nlp[t] = hip.neglogp(spec, theta, pi, a) + 0.2 * torch.randn((E,), generator=g, device="cuda")
    returns = values + torch.randn((T, E, 1), generator=g, device="cuda")
    advs = (returns - values).reshape(T, E).contiguous()
    return dict(obs=obs, dones=dones, actions=actions, values=values, nlp=nlp, returns=returns, advs=advs, states=states0)


def _rectifier_pattern(hip, spec, ws, B, nsteps_mb=0):
    """The rectified activations ppo_grad left in its workspace -> (on/off pattern per layer, the activations).  The
    float64 restatement is then evaluated with the same units switched on: among ~10^7 pre-activations a handful lie
    within float32 rounding of zero, the two sides take different signs there, and every such unit opens or closes a
    whole gradient path (1e-2 of a bias gradient at these sizes) without either side being wrong."""
    acts = [hip.grad_workspace_region(spec, ws, B, name, nsteps_mb).cpu().numpy() for name in ("c1", "c2", "c3", "fc1")]
    return [a > 0 for a in acts], acts


def _check_activations(oracle, acts):
    for name, mine, theirs in zip(("c1", "c2", "c3", "fc1"), acts, oracle.last_acts):
        t = theirs.detach().numpy().reshape(-1)
        err = float(np.abs(mine - t).max())
        assert err <= 2e-5, (name, err)


def _host_rows(ro, idx_dev, T, E):
    """The minibatch rows the env-major flat indices name, on the host (what ``arr[mbinds]`` hands to Model.train)."""
    s = idx_dev.long()
    env, t = s // T, s % T
    pick = lambda a: a[t, env].cpu().numpy()
    return dict(obs=pick(ro["obs"]), ret=pick(ro["returns"])[:, 0], act=pick(ro["actions"]), val=pick(ro["values"])[:, 0],
                adv=pick(ro["advs"]), nlp=pick(ro["nlp"]), masks=pick(ro["dones"]))


def test_nature_cnn_minibatch_of_8192(mods):
    """configs[2]: 256 envs x 128 steps of 84x84x4 uint8, minibatch 8192 (4 per epoch)."""
    hip, L = mods["hip"], mods["lib"]
    hwc, nact, T, E, B, nchunk = (84, 84, 4), 6, 128, 256, 8192, 8
    spec = hip.PolicySpec(L.POLICY_CNN, L.DIST_CATEGORICAL, nact, 1, ob_hwc=hwc)
    shapes = tfgraph.cnn_param_shapes(hwc[2], nact, None, flat=util.cnn_flat(hwc[0], hwc[1]))
    params = util.make_params(shapes, 61, pi_scale=0.5, jitter=0.002)
    theta = dev(util.flatten_params(params, spec))
    ro = _device_image_rollout(hip, spec, theta, T, E, hwc, nact, 62, 1.0 / 1000)
    rref = hip.RolloutRef(spec, ro["obs"], ro["actions"], ro["returns"], ro["values"], ro["nlp"], ro["advs"], T, E)
    idx = torch.randperm(T * E, generator=torch.Generator().manual_seed(63))[:B].to(torch.int32).cuda()
    stats = hip.adv_stats(ro["advs"], idx, B, 1, T, E)
    loss_desc = L.LossDesc(0.01, 0.5, 0.5, 1)
    ws = torch.empty(spec.grad_workspace_bytes(B), dtype=torch.uint8, device="cuda")
    grads = torch.full((spec.nparams,), float("nan"), device="cuda")
    losses = torch.zeros(8, device="cuda")
    hip.ppo_grad(spec, loss_desc, theta, rref, idx, B, stats, 0.1, 1.0 / B, grads, losses, ws)
    full_g, full_l = grads.cpu().numpy().astype(np.float64), losses.cpu().numpy()[:5].astype(np.float64)
    assert np.all(np.isfinite(full_g))
    relu_on, acts = _rectifier_pattern(hip, spec, ws, B)
    # the same minibatch as 8 chunks of 1024 rows, each scaled by 1/8192 and normalised with the whole minibatch's statistics
    sum_g, sum_l = np.zeros_like(full_g), np.zeros(5)
    cb = B // nchunk
    for c in range(nchunk):
        hip.ppo_grad(spec, loss_desc, theta, rref, idx[c * cb:(c + 1) * cb].contiguous(), cb, stats, 0.1, 1.0 / B, grads, losses, ws)
        sum_g += grads.cpu().numpy()
        sum_l += losses.cpu().numpy()[:5]
    _check_grads(spec, full_g, sum_g, "8192 rows at once vs 8 x 1024")
    np.testing.assert_allclose(full_l, sum_l, rtol=5e-5, atol=1e-7)
    # the whole minibatch against the float64 restatement
    rows = _host_rows(ro, idx, T, E)
    oracle = tfgraph.TrainOracle(params, "cnn", "categorical", tfgraph.LossCfg(0.01, 0.5, 0.5, True))
    # (before the oracle's own optimiser step moves its parameters)
    nflip = util.check_rectifiers_against_the_oracles_own(oracle, relu_on, acts, rows["obs"])
    print("nature_cnn minibatch 8192: %d of %d rectified units take the other side of zero in float64" % (nflip, sum(a.size for a in acts)))
    o_losses = oracle.train(2.5e-4, 0.1, rows["obs"], rows["ret"], None, rows["act"], rows["val"],
                            _normalised(rows["adv"], stats.cpu().numpy()[0]), rows["nlp"], normalize=False, relu_on=relu_on)
    _check_activations(oracle, acts)
    np.testing.assert_allclose(full_l, o_losses, rtol=5e-5, atol=1e-7)
    want = np.concatenate([oracle.last_grads[name].numpy().ravel() for name, _, _ in spec.vars])
    _check_grads(spec, full_g, want, "8192 rows vs oracle")

    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    hip.ppo_grad(spec, loss_desc, theta, rref, idx, B, stats, 0.1, 1.0 / B, grads, losses, ws)
    hip.adam_step(theta, grads, m, v, 2.5e-4, 1, hip.adam_scratch(spec.nparams, "cuda"), max_grad_norm=0.5)
    new = util.unflatten(theta.cpu().numpy(), spec, params)
    o_params = oracle.numpy_params()
    diff = max(float(np.abs(new[k] - o_params[k]).max()) for k in new)
    print("nature_cnn minibatch 8192: post-update parameter diff vs oracle %.3e" % diff)
    assert diff <= 5e-6


def test_cnn_lstm_full_depth_env_sequences(mods):
    """configs[4]: 512 envs x 128 steps, minibatches of 128 envs x 128 steps (baselines/ppo2/ppo2.py:124-136), BPTT over
    all 128 steps with the done-mask resets of baselines/a2c/utils.py:96-97 falling inside the sequences."""
    hip, L = mods["hip"], mods["lib"]
    hwc, nact, T, E, nlstm = (84, 84, 4), 6, 128, 512, 256
    spec = hip.PolicySpec(L.POLICY_CNN_LSTM, L.DIST_CATEGORICAL, nact, 1, ob_hwc=hwc, nlstm=nlstm)
    shapes = tfgraph.cnn_param_shapes(hwc[2], nact, "lstm", nlstm, flat=util.cnn_flat(hwc[0], hwc[1]))
    params = util.make_params(shapes, 71, pi_scale=0.5, jitter=0.002)
    theta = dev(util.flatten_params(params, spec))
    ro = _device_image_rollout(hip, spec, theta, T, E, hwc, nact, 72, 1.0 / 20, nlstm=nlstm)
    assert int(ro["dones"].sum()) > T * E // 40, "resets must fall inside the sequences"
    rref = hip.RolloutRef(spec, ro["obs"], ro["actions"], ro["returns"], ro["values"], ro["nlp"], ro["advs"], T, E,
                          masks=ro["dones"], states=ro["states"])
    loss_desc = L.LossDesc(0.01, 0.5, 0.5, 1)
    envperm = torch.randperm(E, generator=torch.Generator().manual_seed(73)).to(torch.int32)
    tgrid = torch.arange(T, dtype=torch.int32)[None, :]

    def minibatch(envs):
        flat = (envs[:, None] * T + tgrid).reshape(-1).contiguous()
        return envs.cuda(), flat.cuda()

    grads = torch.full((spec.nparams,), float("nan"), device="cuda")
    losses = torch.zeros(8, device="cuda")
    # (i) 16 envs x 128 steps against the oracle
    envs, flat = minibatch(envperm[:16])
    B = flat.numel()
    ws = torch.empty(spec.grad_workspace_bytes(128 * T), dtype=torch.uint8, device="cuda")
    stats = hip.adv_stats(ro["advs"], flat, B, 1, T, E)
    hip.ppo_grad(spec, loss_desc, theta, rref, flat, B, stats, 0.1, 1.0 / B, grads, losses, ws, mb_envs=envs, nsteps_mb=T)
    got_g, got_l = grads.cpu().numpy().astype(np.float64), losses.cpu().numpy()[:5].astype(np.float64)
    relu_on, acts = _rectifier_pattern(hip, spec, ws, B, T)
    rows = _host_rows(ro, flat, T, E)
    assert rows["masks"].reshape(16, T)[:, 1:].sum() >= 16, "several resets inside these sequences"
    oracle = tfgraph.TrainOracle(params, "lstm", "categorical", tfgraph.LossCfg(0.01, 0.5, 0.5, True))
    nflip = util.check_rectifiers_against_the_oracles_own(oracle, relu_on, acts, rows["obs"], rows["masks"].astype(bool),
                                                          ro["states"][envs.long()].cpu().numpy())
    print("cnn_lstm 16 envs x 128 steps: %d of %d rectified units take the other side of zero in float64" % (nflip, sum(a.size for a in acts)))
    o_losses = oracle.train(2.5e-4, 0.1, rows["obs"], rows["ret"], rows["masks"].astype(bool), rows["act"], rows["val"],
                            _normalised(rows["adv"], stats.cpu().numpy()[0]), rows["nlp"],
                            states=ro["states"][envs.long()].cpu().numpy(), normalize=False, relu_on=relu_on)
    _check_activations(oracle, acts)
    np.testing.assert_allclose(got_l, o_losses, rtol=5e-5, atol=1e-7)
    want = np.concatenate([oracle.last_grads[name].numpy().ravel() for name, _, _ in spec.vars])
    _check_grads(spec, got_g, want, "16 envs x 128 steps vs oracle")

    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    th = theta.clone()
    hip.adam_step(th, grads, m, v, 2.5e-4, 1, hip.adam_scratch(spec.nparams, "cuda"), max_grad_norm=0.5)
    new = util.unflatten(th.cpu().numpy(), spec, params)
    o_params = oracle.numpy_params()
    diff = max(float(np.abs(new[k] - o_params[k]).max()) for k in new)
    print("cnn_lstm 16 envs x 128 steps: post-update parameter diff vs oracle %.3e" % diff)
    assert diff <= 5e-6
    # (ii) the configuration's own minibatch, 128 envs x 128 steps, against the sum over 8 chunks of 16 envs
    envs, flat = minibatch(envperm[:128])
    B = flat.numel()
    stats = hip.adv_stats(ro["advs"], flat, B, 1, T, E)
    hip.ppo_grad(spec, loss_desc, theta, rref, flat, B, stats, 0.1, 1.0 / B, grads, losses, ws, mb_envs=envs, nsteps_mb=T)
    full_g, full_l = grads.cpu().numpy().astype(np.float64), losses.cpu().numpy()[:5].astype(np.float64)
    assert np.all(np.isfinite(full_g))
    sum_g, sum_l = np.zeros_like(full_g), np.zeros(5)
    for c in range(8):
        e_c, f_c = minibatch(envperm[c * 16:(c + 1) * 16])
        hip.ppo_grad(spec, loss_desc, theta, rref, f_c, f_c.numel(), stats, 0.1, 1.0 / B, grads, losses, ws, mb_envs=e_c, nsteps_mb=T)
        sum_g += grads.cpu().numpy()
        sum_l += losses.cpu().numpy()[:5]
    _check_grads(spec, full_g, sum_g, "128 envs at once vs 8 x 16 envs")
    np.testing.assert_allclose(full_l, sum_l, rtol=5e-5, atol=1e-7)


@pytest.mark.parametrize("B", [131072, 100003])
def test_fork_network_at_benchmark_minibatch_size(mods, B, monkeypatch):
    """The fork's own network (two 256-128 tanh trunks, five value channels, baselines/ppo2/policies.py:139-151) on a
    minibatch of the benchmark's size (and on one that is no multiple of any tile): the fused chain + fused
    weight-gradient route against the float64 restatement, and against the layer-by-layer route of the same library."""
    hip, lib = mods["hip"], mods["lib"]
    D, A, R, T, E = 17, 6, 5, 64, 2048
    spec = hip.PolicySpec(lib.POLICY_MLP, lib.DIST_GAUSSIAN, A, R, ob_dim=D, hidden=(256, 128), order=lib.ORDER_FORK)
    shapes = tfgraph.mlp_param_shapes(D, (256, 128), A, R, True, "fork")
    params = util.make_params(shapes, 61, pi_scale=0.5, jitter=0.02)
    ro = util.synthetic_mlp_rollout(np.random.RandomState(62), T, E, D, A, R, True, params, nlp_noise=0.2)
    _, rets = scan.gae(ro["rewards"], ro["values"], ro["dones"], ro["last_values"], ro["last_dones"], 0.99, 0.95)
    advs_tm = (rets - ro["values"]).sum(axis=2).astype(np.float32)
    idx = np.random.RandomState(63).permutation(T * E)[:B].astype(np.int32)
    lossd = lib.LossDesc(0.0, 0.5, 0.5, 0)

    f = lambda a: scan.sf01(a)[idx]
    oracle = tfgraph.TrainOracle(params, "mlp", "gaussian", tfgraph.LossCfg(0.0, 0.5, 0.5, False))
    o_losses = oracle.train(3e-4, 0.2, f(ro["obs"]), f(rets), None, f(ro["actions"]), f(ro["values"]), f(advs_tm),
                            f(ro["neglogpacs"]), apply=False)
    want = np.concatenate([oracle.last_grads[name].numpy().ravel() for name, _, _ in spec.vars])

    theta = dev(util.flatten_params(params, spec))
    ws = torch.empty(spec.grad_workspace_bytes(B), dtype=torch.uint8, device="cuda")
    rref = hip.RolloutRef(spec, dev(ro["obs"]), dev(ro["actions"]), dev(rets), dev(ro["values"]), dev(ro["neglogpacs"]),
                          dev(advs_tm), T, E)
    d_idx = dev(idx)
    stats = hip.adv_stats(rref.keep[5], d_idx, B, 1, T, E)
    got = {}
    for route in ("default", "layered"):
        if route == "layered":
            monkeypatch.setenv("PPO2HIP_MLP_PATH", "layered")
        grads = torch.full((spec.nparams,), float("nan"), device="cuda")
        losses = torch.zeros(8, device="cuda")
        hip.ppo_grad(spec, lossd, theta, rref, d_idx, B, stats, 0.2, 1.0 / B, grads, losses, ws)
        torch.cuda.synchronize()
        got[route] = (grads.cpu().numpy(), losses.cpu().numpy()[:5])
    for route, (g, l) in got.items():
        assert np.all(np.isfinite(g)), route
        np.testing.assert_allclose(l, o_losses, rtol=5e-5, atol=1e-7, err_msg=route)
        _check_grads(spec, g, want, route)
    _check_grads(spec, got["default"][0], got["layered"][0], "fused against layered", tol=2e-5)
