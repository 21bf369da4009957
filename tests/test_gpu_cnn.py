"""nature_cnn and cnn+LSTM policies: minibatch gradient, losses and the optimiser step of the
HIP path (implicit-GEMM convolutions, LSTM with done-mask resets, env-sequence minibatches)
against the float64 restatement.  GPU only.  Tolerances: losses rtol 5e-5; gradients 5e-5 of
the per-variable max (contractions up to K = 3136 in fp32); post-update parameters atol 2e-6."""
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


def _check(hip, spec, params, oracle, o_losses, theta, grads, losses, upstream=True, lr=2.5e-4):
    torch.cuda.synchronize()
    g = grads.cpu().numpy()
    assert np.all(np.isfinite(g))
    np.testing.assert_allclose(losses.cpu().numpy()[:5], o_losses, rtol=5e-5, atol=1e-7)
    for name, off, shp in spec.vars:
        ref = oracle.last_grads[name].numpy().ravel()
        got = g[off:off + ref.size]
        tol = 5e-5 * max(np.abs(ref).max(), 1e-6)
        assert np.abs(got - ref).max() <= tol, (name, float(np.abs(got - ref).max()), tol)
    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    scratch = hip.adam_scratch(spec.nparams, "cuda")
    hip.adam_step(theta, grads, m, v, lr, 1, scratch, max_grad_norm=0.5)
    new = util.unflatten(theta.cpu().numpy(), spec, params)
    o_params = oracle.numpy_params()
    assert max(float(np.abs(new[k] - o_params[k]).max()) for k in new) <= 2e-6


# (36, 36, 2): the first layer as a float32 product; (44, 52, 4), (84, 84, 4): its uint8 kernels (bf16 products of exact
# pieces) — 24 frames end in a part-filled 128-pixel tile and an odd number of pixel groups per frame; 300 frames make the
# workgroups of both kernels walk several tiles / frames each, unevenly
@pytest.mark.parametrize("hwc,nact,T,E,B", [((36, 36, 2), 4, 4, 6, 20), ((44, 52, 4), 6, 3, 8, 24), ((84, 84, 4), 6, 2, 8, 16),
                                            ((84, 84, 4), 6, 3, 100, 300), ((84, 84, 4), 6, 1, 3, 3)])
def test_cnn_policy_grad(hip, hwc, nact, T, E, B):
    from baselines_amd import _lib
    spec = hip.PolicySpec(_lib.POLICY_CNN, _lib.DIST_CATEGORICAL, nact, 1, ob_hwc=hwc)
    shapes = tfgraph.cnn_param_shapes(hwc[2], nact, None, flat=util.cnn_flat(hwc[0], hwc[1]))
    params = util.make_params(shapes, 5, pi_scale=0.5, jitter=0.002)
    rng = np.random.RandomState(6)
    ro = util.synthetic_cnn_rollout(rng, T, E, hwc, nact, params)
    _, rets = scan.gae(ro["rewards"], ro["values"], ro["dones"], ro["last_values"], ro["last_dones"], 0.99, 0.95)
    advs_tm = (rets - ro["values"])[:, :, 0]
    idx = rng.permutation(T * E)[:B].astype(np.int32)
    f = lambda a: scan.sf01(a)[idx]
    oracle = tfgraph.TrainOracle(params, "cnn", "categorical", tfgraph.LossCfg(0.01, 0.5, 0.5, True))
    o_losses = oracle.train(2.5e-4, 0.1, f(ro["obs"]), f(rets), None, f(ro["actions"]), f(ro["values"]), f(advs_tm),
                            f(ro["neglogpacs"]))
    theta = dev(util.flatten_params(params, spec))
    grads = torch.full((spec.nparams,), float("nan"), device="cuda")
    losses = torch.zeros(8, device="cuda")
    ws = torch.empty(spec.grad_workspace_bytes(B), dtype=torch.uint8, device="cuda")
    rref = hip.RolloutRef(spec, dev(ro["obs"]), dev(ro["actions"]), dev(rets), dev(ro["values"]), dev(ro["neglogpacs"]),
                          dev(advs_tm), T, E)
    d_idx = dev(idx)
    stats = hip.adv_stats(rref.keep[5], d_idx, B, 1, T, E)
    hip.ppo_grad(spec, _lib.LossDesc(0.01, 0.5, 0.5, 1), theta, rref, d_idx, B, stats, 0.1, 1.0 / B, grads, losses, ws)
    _check(hip, spec, params, oracle, o_losses, theta, grads, losses)
    # forward entry point used by the runner
    fws = torch.empty(spec.forward_workspace_bytes(E), dtype=torch.uint8, device="cuda")
    pi, vf, _ = hip.policy_forward(spec, dev(util.flatten_params(params, spec)), dev(ro["obs"][0]), fws)
    p = tfgraph.to_torch(params, torch.float64, requires_grad=False)
    o_pi, o_vf = tfgraph.cnn_forward(p, torch.as_tensor(ro["obs"][0]))
    np.testing.assert_allclose(pi.cpu().numpy(), o_pi.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(vf.cpu().numpy(), o_vf.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("cell", ["lstm", "lnlstm"])
@pytest.mark.parametrize("hwc,nact,T,E,nenv_mb,nlstm", [((36, 36, 2), 4, 6, 6, 3, 32), ((44, 52, 4), 5, 5, 8, 4, 64),
                                                        ((84, 84, 4), 6, 4, 4, 2, 256),
                                                        # two-step sequences over two env groups, the second part-filled
                                                        # (whole-sequence launches); one-step sequences (step launches)
                                                        ((36, 36, 2), 4, 2, 20, 18, 128), ((36, 36, 2), 4, 1, 6, 3, 64)])
def test_cnn_lstm_policy_grad(hip, hwc, nact, T, E, nenv_mb, nlstm, cell):
    """Env-sequence minibatch (baselines/ppo2/ppo2.py:124-136): whole env rows, masks reset the state;
    plain cell (a2c/utils.py:86-108) and layer-normalised cell (:110-147, gains and biases perturbed
    away from their 1 / 0 initial values so that every term of their gradients is exercised)."""
    from baselines_amd import _lib
    kind = _lib.POLICY_CNN_LSTM if cell == "lstm" else _lib.POLICY_CNN_LNLSTM
    spec = hip.PolicySpec(kind, _lib.DIST_CATEGORICAL, nact, 1, ob_hwc=hwc, nlstm=nlstm)
    shapes = tfgraph.cnn_param_shapes(hwc[2], nact, cell, nlstm, flat=util.cnn_flat(hwc[0], hwc[1]))
    params = util.make_params(shapes, 7, pi_scale=0.5, jitter=0.002)
    rng = np.random.RandomState(8)
    if cell == "lnlstm":
        for name in ("lstm1/gx", "lstm1/bx", "lstm1/gh", "lstm1/bh", "lstm1/b", "lstm1/gc", "lstm1/bc"):
            params[name] = (params[name] + 0.1 * rng.randn(*params[name].shape)).astype(np.float32)
    ro = util.synthetic_cnn_rollout(rng, T, E, hwc, nact, params, kind=cell, nlstm=nlstm)
    ro["states"] = (0.3 * rng.randn(E, 2 * nlstm)).astype(np.float32)     # non-trivial initial states
    _, rets = scan.gae(ro["rewards"], ro["values"], ro["dones"], ro["last_values"], ro["last_dones"], 0.99, 0.95)
    advs_tm = (rets - ro["values"])[:, :, 0]
    envs = rng.permutation(E)[:nenv_mb].astype(np.int32)
    flat = (envs[:, None] * T + np.arange(T)[None, :]).ravel().astype(np.int32)
    B = flat.size
    f = lambda a: scan.sf01(a)[flat]
    oracle = tfgraph.TrainOracle(params, cell, "categorical", tfgraph.LossCfg(0.01, 0.5, 0.5, True))
    o_losses = oracle.train(2.5e-4, 0.1, f(ro["obs"]), f(rets), f(ro["dones"]), f(ro["actions"]), f(ro["values"]),
                            f(advs_tm), f(ro["neglogpacs"]), states=ro["states"][envs])
    theta = dev(util.flatten_params(params, spec))
    grads = torch.full((spec.nparams,), float("nan"), device="cuda")
    losses = torch.zeros(8, device="cuda")
    ws = torch.empty(spec.grad_workspace_bytes(B), dtype=torch.uint8, device="cuda")
    rref = hip.RolloutRef(spec, dev(ro["obs"]), dev(ro["actions"]), dev(rets), dev(ro["values"]), dev(ro["neglogpacs"]),
                          dev(advs_tm), T, E, masks=dev(ro["dones"]), states=dev(ro["states"]))
    d_idx = dev(flat)
    stats = hip.adv_stats(rref.keep[5], d_idx, B, 1, T, E)
    hip.ppo_grad(spec, _lib.LossDesc(0.01, 0.5, 0.5, 1), theta, rref, d_idx, B, stats, 0.1, 1.0 / B, grads, losses, ws,
                 mb_envs=dev(envs), nsteps_mb=T)
    _check(hip, spec, params, oracle, o_losses, theta, grads, losses)
    # single-step forward with state hand-over, as Runner.run drives it
    fws = torch.empty(spec.forward_workspace_bytes(E), dtype=torch.uint8, device="cuda")
    th = dev(util.flatten_params(params, spec))
    pi, vf, snew = hip.policy_forward(spec, th, dev(ro["obs"][0]), fws, masks=dev(ro["dones"][0]), states=dev(ro["states"]))
    p = tfgraph.to_torch(params, torch.float64, requires_grad=False)
    o_pi, o_vf, o_s = tfgraph.cnn_lstm_forward(p, torch.as_tensor(ro["obs"][0]), torch.as_tensor(ro["dones"][0].astype(np.float64)),
                                               torch.as_tensor(ro["states"]).double(), E, 1, layer_norm=(cell == "lnlstm"))
    np.testing.assert_allclose(pi.cpu().numpy(), o_pi.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(snew.cpu().numpy(), o_s.numpy(), rtol=2e-5, atol=2e-6)


def test_lnlstm_width_limit_is_an_error_not_a_fallback(hip):
    """The layer-normalised cell kernels cover nlstm <= 1024; beyond that the library says so."""
    from baselines_amd import _lib
    spec = hip.PolicySpec(_lib.POLICY_CNN_LNLSTM, _lib.DIST_CATEGORICAL, 4, 1, ob_hwc=(36, 36, 2), nlstm=1100)
    fws = torch.empty(spec.forward_workspace_bytes(2), dtype=torch.uint8, device="cuda")
    with pytest.raises(_lib.Ppo2HipError):
        hip.policy_forward(spec, torch.zeros(spec.nparams, device="cuda"), torch.zeros((2, 36, 36, 2), dtype=torch.uint8, device="cuda"),
                           fws, masks=torch.zeros(2, dtype=torch.uint8, device="cuda"), states=torch.zeros((2, 2200), device="cuda"))


_ROUTE_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from baselines_amd import hip, _lib
hwc, nact, T, E, B = (84, 84, 4), 6, 2, 80, 160
spec = hip.PolicySpec(_lib.POLICY_CNN, _lib.DIST_CATEGORICAL, nact, 1, ob_hwc=hwc)
g = torch.Generator(device="cuda"); g.manual_seed(11)
theta = (torch.randn(spec.nparams, device="cuda", generator=g) * 0.03)
obs = torch.randint(0, 256, (T, E) + hwc, device="cuda", generator=g, dtype=torch.uint8)
f = lambda *s: torch.randn(*s, device="cuda", generator=g)
acts = torch.randint(0, nact, (T, E), device="cuda", generator=g, dtype=torch.int32)
rets, vals, nlp, advs = f(T, E, 1), f(T, E, 1), f(T, E).abs() + 1.0, f(T, E)
rref = hip.RolloutRef(spec, obs, acts, rets, vals, nlp, advs, T, E)
idx = torch.randperm(T * E, device="cuda", generator=g)[:B].to(torch.int32)
stats = hip.adv_stats(rref.keep[5], idx, B, 1, T, E)
grads = torch.zeros(spec.nparams, device="cuda"); losses = torch.zeros(8, device="cuda")
ws = torch.empty(spec.grad_workspace_bytes(B), dtype=torch.uint8, device="cuda")
hip.ppo_grad(spec, _lib.LossDesc(0.01, 0.5, 0.5, 1), theta, rref, idx, B, stats, 0.1, 1.0 / B, grads, losses, ws)
torch.cuda.synchronize()
np.savez(sys.argv[2], grads=grads.cpu().numpy(), losses=losses.cpu().numpy())
"""


def test_uint8_conv_kernels_agree_with_the_float32_route(tmp_path):
    """conv1 on uint8 frames: the bf16-piece kernels (frames read in place, rectifier bits) against the float32 product on a
    gathered copy, same inputs, two fresh processes (the switch is read once per process).  Every product term is exact in
    both, so the routes differ by the rounding of running sums only: 2e-6 of each variable's largest gradient."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "route.py"
    script.write_text(_ROUTE_SCRIPT)
    out = {}
    for name, env in (("pieces", {}), ("plain", {"PPO2HIP_CONV_U8_PLAIN": "1"})):
        e = dict(os.environ)
        e.update(env)
        path = str(tmp_path / (name + ".npz"))
        subprocess.run([sys.executable, str(script), root, path], check=True, env=e, timeout=240)
        out[name] = np.load(path)
    from baselines_amd import hip as h, _lib
    spec = h.PolicySpec(_lib.POLICY_CNN, _lib.DIST_CATEGORICAL, 6, 1, ob_hwc=(84, 84, 4))
    a, b = out["pieces"]["grads"], out["plain"]["grads"]
    assert np.all(np.isfinite(a)) and np.abs(a).max() > 0
    for name, off, shp in spec.vars:
        n = int(np.prod(shp))
        ref = b[off:off + n]
        assert np.abs(a[off:off + n] - ref).max() <= 2e-6 * max(np.abs(ref).max(), 1e-6), name
    np.testing.assert_allclose(out["pieces"]["losses"][:5], out["plain"]["losses"][:5], rtol=2e-6, atol=1e-7)


def test_piece_products_agree_with_the_float32_pipe(tmp_path):
    """The wide convolution products on the bf16 pipe (six exact piece products per multiply-add) against the same launches
    on the float32 pipe (PPO2HIP_GEMM_FP32=1, read once per process): same inputs, two fresh processes.  Both are exact
    up to the rounding of running sums, so the nature_cnn gradients agree to 5e-6 of each variable's largest entry."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "route.py"
    script.write_text(_ROUTE_SCRIPT)
    out = {}
    for name, env in (("pieces", {}), ("fp32", {"PPO2HIP_GEMM_FP32": "1"})):
        e = dict(os.environ)
        e.update(env)
        path = str(tmp_path / (name + ".npz"))
        subprocess.run([sys.executable, str(script), root, path], check=True, env=e, timeout=240)
        out[name] = np.load(path)
    from baselines_amd import hip as h, _lib
    spec = h.PolicySpec(_lib.POLICY_CNN, _lib.DIST_CATEGORICAL, 6, 1, ob_hwc=(84, 84, 4))
    a, b = out["pieces"]["grads"], out["fp32"]["grads"]
    assert np.all(np.isfinite(a)) and np.abs(a).max() > 0
    assert not np.array_equal(a, b)          # the switch did change the route
    for name, off, shp in spec.vars:
        n = int(np.prod(shp))
        ref = b[off:off + n]
        assert np.abs(a[off:off + n] - ref).max() <= 5e-6 * max(np.abs(ref).max(), 1e-6), name
    np.testing.assert_allclose(out["pieces"]["losses"][:5], out["fp32"]["losses"][:5], rtol=5e-6, atol=1e-7)


_LSTM_ROUTE_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from baselines_amd import hip, _lib
hwc, nact, nlstm = (36, 36, 2), 4, 128
T, E, nenv_mb = (int(v) for v in sys.argv[3:6])
spec = hip.PolicySpec(_lib.POLICY_CNN_LSTM, _lib.DIST_CATEGORICAL, nact, 1, ob_hwc=hwc, nlstm=nlstm)
g = torch.Generator(device="cuda"); g.manual_seed(5)
theta = (torch.randn(spec.nparams, device="cuda", generator=g) * 0.05)
obs = torch.randint(0, 256, (T, E) + hwc, device="cuda", generator=g, dtype=torch.uint8)
f = lambda *s: torch.randn(*s, device="cuda", generator=g)
acts = torch.randint(0, nact, (T, E), device="cuda", generator=g, dtype=torch.int32)
rets, vals, nlp, advs = f(T, E, 1), f(T, E, 1), f(T, E).abs() + 1.0, f(T, E)
dones = (torch.rand(T, E, device="cuda", generator=g) < 0.15)
states = f(E, 2 * nlstm) * 0.3
rref = hip.RolloutRef(spec, obs, acts, rets, vals, nlp, advs, T, E, dones, states)
envs = torch.randperm(E, device="cuda", generator=g)[:nenv_mb].to(torch.int32)
idx = (envs[:, None].to(torch.int64) * T + torch.arange(T, device="cuda")[None, :]).to(torch.int32).reshape(-1).contiguous()
B = nenv_mb * T
stats = hip.adv_stats(rref.keep[5], idx, B, 1, T, E)
grads = torch.zeros(spec.nparams, device="cuda"); losses = torch.zeros(8, device="cuda")
ws = torch.empty(spec.grad_workspace_bytes(B), dtype=torch.uint8, device="cuda")
hip.ppo_grad(spec, _lib.LossDesc(0.01, 0.5, 0.5, 1), theta, rref, idx, B, stats, 0.1, 1.0 / B, grads, losses, ws,
             mb_envs=envs, nsteps_mb=T)
hip.take_faults()
torch.cuda.synchronize()
np.savez(sys.argv[2], grads=grads.cpu().numpy(), losses=losses.cpu().numpy())
"""


# 128 steps with 40 and 24 envs: several env groups, the last part-filled, over a sequence long enough for env groups to
# drift whole steps apart (rows past the last env must not wait on another group's words); take_faults raises on a give-up
@pytest.mark.parametrize("T,E,nenv_mb", [(12, 48, 40), (128, 48, 40), (128, 24, 24)])
def test_lstm_sequence_kernels_agree_with_one_launch_per_step(tmp_path, T, E, nenv_mb):
    """Plain LSTM recurrence: the whole-sequence launches (workgroups handing h and partial dhm sums round) against one
    launch per step and direction, same inputs, two fresh processes (the switch is read once per process).  40 envs leave a
    part-filled env group.  The forward adds the same terms in the same order; the backward adds the sixteen producers'
    partial sums of dhm in another order than the per-step kernel's four wave sums: 5e-6 of each variable's largest gradient."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "route.py"
    script.write_text(_LSTM_ROUTE_SCRIPT)
    out = {}
    for name, env in (("seq", {}), ("steps", {"PPO2HIP_LSTM_SEQ_OFF": "1"})):
        e = dict(os.environ)
        e.update(env)
        path = str(tmp_path / (name + ".npz"))
        subprocess.run([sys.executable, str(script), root, path, str(T), str(E), str(nenv_mb)], check=True, env=e, timeout=240)
        out[name] = np.load(path)
    from baselines_amd import hip as h, _lib
    spec = h.PolicySpec(_lib.POLICY_CNN_LSTM, _lib.DIST_CATEGORICAL, 4, 1, ob_hwc=(36, 36, 2), nlstm=128)
    a, b = out["seq"]["grads"], out["steps"]["grads"]
    assert np.all(np.isfinite(a)) and np.abs(a).max() > 0
    for name, off, shp in spec.vars:
        n = int(np.prod(shp))
        ref = b[off:off + n]
        assert np.abs(a[off:off + n] - ref).max() <= 5e-6 * max(np.abs(ref).max(), 1e-6), name
    np.testing.assert_allclose(out["seq"]["losses"][:5], out["steps"]["losses"][:5], rtol=2e-6, atol=1e-7)
