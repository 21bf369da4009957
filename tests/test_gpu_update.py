"""Minibatch gradient, loss scalars and the clip+Adam step of the HIP path against the
float64 restatement of the reference's TF graph (oracle/tfgraph.py).  GPU only.

Stated tolerance (north star: fp32, ~1e-5 relative): losses rtol 2e-5; gradients
|g - g_ref| <= 2e-5 * max|g_ref| per variable; post-update parameters atol 2e-6
(one Adam step moves a parameter by at most ~lr = 3e-4).
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import scan, tfgraph
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available()
    from baselines_amd import hip as h
    return h


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def mlp_case(hip, gaussian, upstream, D, hidden, A, R, T, E, B, seed, ent_coef):
    from baselines_amd import _lib
    order = _lib.ORDER_UPSTREAM if upstream else _lib.ORDER_FORK
    spec = hip.PolicySpec(_lib.POLICY_MLP, _lib.DIST_GAUSSIAN if gaussian else _lib.DIST_CATEGORICAL, A, R,
                          ob_dim=D, hidden=hidden, order=order)
    shapes = tfgraph.mlp_param_shapes(D, hidden, A, R, gaussian, "upstream" if upstream else "fork")
    params = util.make_params(shapes, seed, pi_scale=0.5, jitter=0.05)
    rng = np.random.RandomState(seed + 1)
    ro = util.synthetic_mlp_rollout(rng, T, E, D, A, R, gaussian, params)
    o_advs, o_rets = scan.gae(ro["rewards"], ro["values"], ro["dones"], ro["last_values"], ro["last_dones"], 0.99, 0.95)
    advs_tm = (o_rets - ro["values"]).sum(axis=2)          # any scalar advantage will do for the update test
    return spec, params, ro, o_rets, advs_tm.astype(np.float32)


CONFIGS = [
    # gaussian, upstream, D, hidden, A, R, T, E, B, ent_coef
    (True, True, 17, (64, 64), 6, 1, 16, 64, 512, 0.0),
    (True, True, 17, (64, 64), 6, 1, 8, 40, 300, 0.01),       # ragged sizes
    (False, True, 4, (64, 64), 2, 1, 128, 8, 256, 0.01),      # CartPole shape, Categorical
    (True, False, 17, (256, 128), 6, 5, 8, 64, 256, 0.0),     # fork-literal network, 5 value channels
    (True, False, 11, (40, 24), 3, 5, 4, 50, 200, 0.02),      # widths that are not multiples of 32
]


@pytest.mark.parametrize("cfg", CONFIGS)
def test_mlp_grad_and_step(hip, cfg):
    from baselines_amd import _lib
    gaussian, upstream, D, hidden, A, R, T, E, B, ent_coef = cfg
    spec, params, ro, rets, advs_tm = mlp_case(hip, gaussian, upstream, D, hidden, A, R, T, E, B, 11, ent_coef)
    N = T * E
    rng = np.random.RandomState(5)
    idx = rng.permutation(N)[:B].astype(np.int32)
    lossd = _lib.LossDesc(ent_coef, 0.5, 0.5, 1 if upstream else 0)
    cliprange, lr = 0.2, 3e-4

    # ---- oracle on the env-major flat arrays, as Model.train receives them
    f = lambda a: scan.sf01(a)[idx]
    oracle = tfgraph.TrainOracle(params, "mlp", "gaussian" if gaussian else "categorical",
                                 tfgraph.LossCfg(ent_coef, 0.5, 0.5, upstream))
    o_losses = oracle.train(lr, cliprange, f(ro["obs"]), f(rets), None, f(ro["actions"]), f(ro["values"]),
                            f(advs_tm), f(ro["neglogpacs"]))
    o_grads = oracle.last_grads
    o_params = oracle.numpy_params()

    # ---- HIP path reading the time-major rollout in place
    theta = dev(util.flatten_params(params, spec))
    grads = torch.full((spec.nparams,), float("nan"), device="cuda")
    losses = torch.zeros(8, device="cuda")
    ws = torch.empty(spec.grad_workspace_bytes(B), dtype=torch.uint8, device="cuda")
    rref = hip.RolloutRef(spec, dev(ro["obs"]), dev(ro["actions"]), dev(rets), dev(ro["values"]),
                          dev(ro["neglogpacs"]), dev(advs_tm), T, E)
    d_idx = dev(idx)
    stats = hip.adv_stats(rref.keep[5], d_idx, B, 1, T, E)
    hip.ppo_grad(spec, lossd, theta, rref, d_idx, B, stats, cliprange, 1.0 / B, grads, losses, ws)
    torch.cuda.synchronize()
    g = grads.cpu().numpy()
    assert np.all(np.isfinite(g)), "every parameter must receive a gradient"
    np.testing.assert_allclose(losses.cpu().numpy()[:5], o_losses, rtol=2e-5, atol=1e-7)
    for name, off, shp in spec.vars:
        ref = o_grads[name].numpy().ravel()
        got = g[off:off + ref.size]
        tol = 2e-5 * max(np.abs(ref).max(), 1e-6)
        assert np.abs(got - ref).max() <= tol, (name, np.abs(got - ref).max(), tol)

    # ---- clip + Adam
    m = torch.zeros_like(theta)
    v = torch.zeros_like(theta)
    scratch = hip.adam_scratch(spec.nparams, "cuda")
    hip.adam_step(theta, grads, m, v, lr, 1, scratch, max_grad_norm=0.5 if upstream else 0.0)
    torch.cuda.synchronize()
    new = util.unflatten(theta.cpu().numpy(), spec, params)
    diffs = {k: float(np.abs(new[k] - o_params[k]).max()) for k in new}
    assert max(diffs.values()) <= 2e-6, diffs
    if upstream:
        gn = scratch[:4].view(torch.float32).item()
        assert abs(gn - oracle.last_grad_norm) <= 2e-5 * oracle.last_grad_norm
    # the step must actually have moved the parameters
    assert max(float(np.abs(new[k] - params[k]).max()) for k in new) > 1e-5


def test_flat_rows_without_index(hip):
    """T = 0: arrays already in minibatch order, idx NULL, advantages pre-normalised (stats NULL)."""
    from baselines_amd import _lib
    gaussian, upstream, D, hidden, A, R, T, E, B, ent_coef = CONFIGS[0]
    spec, params, ro, rets, advs_tm = mlp_case(hip, gaussian, upstream, D, hidden, A, R, T, E, B, 3, ent_coef)
    f = lambda a: scan.sf01(a)[:B]
    advn = f(advs_tm)
    advn = ((advn - advn.mean()) / (advn.std() + 1e-8)).astype(np.float32)
    oracle = tfgraph.TrainOracle(params, "mlp", "gaussian", tfgraph.LossCfg(0.0, 0.5, 0.5, True))
    o_losses = oracle.train(3e-4, 0.2, f(ro["obs"]), f(rets), None, f(ro["actions"]), f(ro["values"]), advn,
                            f(ro["neglogpacs"]), normalize=False, apply=False)
    theta = dev(util.flatten_params(params, spec))
    grads = torch.zeros(spec.nparams, device="cuda")
    losses = torch.zeros(8, device="cuda")
    ws = torch.empty(spec.grad_workspace_bytes(B), dtype=torch.uint8, device="cuda")
    rref = hip.RolloutRef(spec, dev(f(ro["obs"])), dev(f(ro["actions"])), dev(f(rets)), dev(f(ro["values"])),
                          dev(f(ro["neglogpacs"])), dev(advn), 0, 0)
    hip.ppo_grad(spec, _lib.LossDesc(0.0, 0.5, 0.5, 1), theta, rref, None, B, None, 0.2, 1.0 / B, grads, losses, ws)
    np.testing.assert_allclose(losses.cpu().numpy()[:5], o_losses, rtol=2e-5, atol=1e-7)
    g = grads.cpu().numpy()
    for name, off, shp in spec.vars:
        ref = oracle.last_grads[name].numpy().ravel()
        assert np.abs(g[off:off + ref.size] - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-6), name


def test_workspace_too_small_is_refused(hip):
    from baselines_amd import _lib
    gaussian, upstream, D, hidden, A, R, T, E, B, ent_coef = CONFIGS[0]
    spec, params, ro, rets, advs_tm = mlp_case(hip, gaussian, upstream, D, hidden, A, R, T, E, B, 3, ent_coef)
    theta = dev(util.flatten_params(params, spec))
    rref = hip.RolloutRef(spec, dev(ro["obs"]), dev(ro["actions"]), dev(rets), dev(ro["values"]),
                          dev(ro["neglogpacs"]), dev(advs_tm), T, E)
    with pytest.raises(ValueError):
        hip.ppo_grad(spec, _lib.LossDesc(0, .5, .5, 1), theta, rref, None, B, None, 0.2, 1.0 / B,
                     torch.zeros(spec.nparams, device="cuda"), torch.zeros(8, device="cuda"),
                     torch.empty(1024, dtype=torch.uint8, device="cuda"))


def test_policy_forward_and_neglogp(hip):
    from baselines_amd import _lib
    gaussian, upstream, D, hidden, A, R, T, E, B, ent_coef = CONFIGS[3]
    spec, params, ro, rets, advs_tm = mlp_case(hip, gaussian, upstream, D, hidden, A, R, T, E, B, 4, ent_coef)
    obs = ro["obs"].reshape(-1, D)[:77]
    act = ro["actions"].reshape(-1, A)[:77]
    theta = dev(util.flatten_params(params, spec))
    ws = torch.empty(spec.forward_workspace_bytes(77), dtype=torch.uint8, device="cuda")
    pi, vf, _ = hip.policy_forward(spec, theta, dev(obs), ws)
    p = tfgraph.to_torch(params, torch.float64, requires_grad=False)
    o_pi, o_vf = tfgraph.mlp_forward(p, torch.as_tensor(obs).double())
    np.testing.assert_allclose(pi.cpu().numpy(), o_pi.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(vf.cpu().numpy(), o_vf.numpy(), rtol=1e-5, atol=1e-6)
    nlp = hip.neglogp(spec, theta, pi, dev(act)).cpu().numpy()
    o_nlp = tfgraph.gaussian_neglogp(o_pi, p["logstd"], torch.as_tensor(act).double()).numpy()
    np.testing.assert_allclose(nlp, o_nlp, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("cfg", [CONFIGS[0], CONFIGS[2], (True, False, 9, (64, 64), 3, 5, 8, 64, 500, 0.01)])
def test_fused_and_layered_mlp_paths_agree(hip, cfg, monkeypatch):
    """The single-launch fused kernel and the layer-by-layer GEMM path are two routes to the
    same gradient: they must agree to fp32 rounding (and each is checked against the oracle
    above through whichever route the shape selects)."""
    from baselines_amd import _lib
    gaussian, upstream, D, hidden, A, R, T, E, B, ent_coef = cfg
    spec, params, ro, rets, advs_tm = mlp_case(hip, gaussian, upstream, D, hidden, A, R, T, E, B, 13, ent_coef)
    rng = np.random.RandomState(6)
    idx = dev(rng.permutation(T * E)[:B].astype(np.int32))
    lossd = _lib.LossDesc(ent_coef, 0.5, 0.5, 1 if upstream else 0)
    theta = dev(util.flatten_params(params, spec))
    ws = torch.empty(spec.grad_workspace_bytes(B), dtype=torch.uint8, device="cuda")
    rref = hip.RolloutRef(spec, dev(ro["obs"]), dev(ro["actions"]), dev(rets), dev(ro["values"]),
                          dev(ro["neglogpacs"]), dev(advs_tm), T, E)
    stats = hip.adv_stats(rref.keep[5], idx, B, 1, T, E)
    out = {}
    for path in ("fused16", "fused32", "layered"):
        monkeypatch.setenv("PPO2HIP_MLP_PATH", path)
        grads = torch.full((spec.nparams,), float("nan"), device="cuda")
        losses = torch.zeros(8, device="cuda")
        hip.ppo_grad(spec, lossd, theta, rref, idx, B, stats, 0.2, 1.0 / B, grads, losses, ws)
        torch.cuda.synchronize()
        out[path] = (grads.cpu().numpy(), losses.cpu().numpy())
    gl = out["layered"][0]
    for path in ("fused16", "fused32"):
        gf = out[path][0]
        assert np.all(np.isfinite(gf))
        np.testing.assert_allclose(out[path][1][:5], out["layered"][1][:5], rtol=2e-5, atol=1e-7)
        for name, off, shp in spec.vars:
            n = int(np.prod(shp))
            a, b = gf[off:off + n], gl[off:off + n]
            assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-6), (path, name, np.abs(a - b).max(), np.abs(b).max())
