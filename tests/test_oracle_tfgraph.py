"""The TF-graph restatement has no reference vectors to pin it (TensorFlow is absent), so
its autograd gradients are cross-checked against central finite differences of the same
float64 objective, and its Adam against the closed form of the first step."""
import numpy as np
import torch

from oracle import tfgraph
from tests import util


def _objective(oracle, batch_args):
    lr, clip, obs, rets, masks, acts, vals, advs, nlps, states = batch_args
    dt = oracle.dtype
    nb = obs.shape[0]
    batch = {"ADV": torch.as_tensor(advs.astype(np.float32)).to(dt),
             "R": torch.as_tensor(rets.reshape(nb, -1)).to(dt),
             "OLDNEGLOGPAC": torch.as_tensor(nlps).to(dt),
             "OLDVPRED": torch.as_tensor(vals.reshape(nb, -1)).to(dt)}
    pi, vf, _ = oracle.forward(obs, masks, states)
    nlp, ent = oracle.neglogp_entropy(pi, acts)
    total, _ = tfgraph.ppo_losses(nlp, ent, vf, batch, clip, oracle.cfg)
    return float(total.detach())


def _fd_check(oracle, batch_args, nprobe=12, eps=1e-6, rtol=2e-5):
    lr, clip, obs, rets, masks, acts, vals, advs, nlps, states = batch_args
    oracle.train(lr, clip, obs, rets, masks, acts, vals, advs, nlps, states=states, normalize=False, apply=False)
    rng = np.random.RandomState(0)
    for name, p in oracle.p.items():
        g = oracle.last_grads[name]
        flat = p.detach().view(-1)
        for _ in range(min(nprobe, flat.numel())):
            i = int(rng.randint(flat.numel()))
            old = float(flat[i])
            with torch.no_grad():
                flat[i] = old + eps
            up = _objective(oracle, batch_args)
            with torch.no_grad():
                flat[i] = old - eps
            dn = _objective(oracle, batch_args)
            with torch.no_grad():
                flat[i] = old
            fd = (up - dn) / (2 * eps)
            an = float(g.reshape(-1)[i])
            assert abs(fd - an) <= rtol * max(abs(an), abs(fd)) + 1e-8, (name, i, fd, an)


def _mlp_batch(gaussian, upstream, R, seed=0):
    D, A, hidden = 5, 3, (8, 6)
    shapes = tfgraph.mlp_param_shapes(D, hidden, A, R, gaussian, "upstream" if upstream else "fork")
    params = util.make_params(shapes, seed, pi_scale=0.5, jitter=0.1)
    rng = np.random.RandomState(seed + 1)
    ro = util.synthetic_mlp_rollout(rng, 4, 6, D, A, R, gaussian, params)
    n = 24
    f = lambda a: a.reshape(n, *a.shape[2:])
    advs = rng.randn(n).astype(np.float32)
    rets = (f(ro["values"]) + 0.5 * rng.randn(n, R)).astype(np.float32)
    args = (3e-4, 0.2, f(ro["obs"]), rets, None, f(ro["actions"]), f(ro["values"]), advs, f(ro["neglogpacs"]), None)
    return params, args


def test_mlp_gradients_against_finite_differences():
    for gaussian, upstream, R in [(True, True, 1), (False, True, 1), (True, False, 5)]:
        params, args = _mlp_batch(gaussian, upstream, R)
        oracle = tfgraph.TrainOracle(params, "mlp", "gaussian" if gaussian else "categorical",
                                     tfgraph.LossCfg(0.01, 0.5, 0.5, upstream))
        _fd_check(oracle, args)


HWC, NACT, NLSTM = (36, 36, 2), 3, 8


def _image_batch(kind, seed=0):
    """A tiny image minibatch: 36x36x2 frames (8x8x32 -> 3x3x64 -> 1x1x64 -> fc 512), 2 envs x 3 steps."""
    cell = None if kind == "cnn" else kind
    shapes = tfgraph.cnn_param_shapes(HWC[2], NACT, cell, NLSTM, flat=util.cnn_flat(HWC[0], HWC[1]))
    params = util.make_params(shapes, seed, pi_scale=0.5, jitter=0.01)
    rng = np.random.RandomState(seed + 1)
    if kind == "lnlstm":
        for name in ("lstm1/gx", "lstm1/bx", "lstm1/gh", "lstm1/bh", "lstm1/b", "lstm1/gc", "lstm1/bc"):
            params[name] = (params[name] + 0.1 * rng.randn(*params[name].shape)).astype(np.float32)
    T, E = 3, 2
    ro = util.synthetic_cnn_rollout(rng, T, E, HWC, NACT, params, kind=kind, nlstm=NLSTM if cell else 0)
    n = T * E
    em = lambda a: np.swapaxes(a, 0, 1).reshape(n, *a.shape[2:])        # env-major rows, as batch_to_seq expects
    advs = rng.randn(n).astype(np.float32)
    vals = em(ro["values"])
    rets = (vals + 0.5 * rng.randn(n, 1)).astype(np.float32)
    masks = em(ro["dones"])
    masks[1] = True                                                     # a reset inside env 0's sequence
    states = (0.3 * rng.randn(E, 2 * NLSTM)).astype(np.float32) if cell else None
    args = (2.5e-4, 0.1, em(ro["obs"]), rets, masks if cell else None, em(ro["actions"]), vals, advs, em(ro["neglogpacs"]), states)
    return params, args


def test_image_policy_gradients_against_finite_differences():
    """nature_cnn, cnn + LSTM and cnn + layer-normalised LSTM: autograd of the restated graph against central differences
    of the same float64 objective (every variable, a dozen entries each; the step of 1e-6 keeps clear of rectifier kinks)."""
    for kind in ("cnn", "lstm", "lnlstm"):
        params, args = _image_batch(kind)
        oracle = tfgraph.TrainOracle(params, kind, "categorical", tfgraph.LossCfg(0.01, 0.5, 0.5, True))
        _fd_check(oracle, args, nprobe=8, rtol=5e-5)


def _naive_conv(x, w, b, stride):
    """tf.nn.conv2d(x, w, strides=[1, s, s, 1], padding='VALID') + b from its definition: NHWC input, [kh, kw, cin, cout]
    weights, out[n, i, j, o] = sum_{di, dj, c} x[n, s i + di, s j + dj, c] w[di, dj, c, o]  (baselines/a2c/utils.py:42-61)."""
    n, H, W, C = x.shape
    kh, kw, _, O = w.shape
    oh, ow = (H - kh) // stride + 1, (W - kw) // stride + 1
    out = np.zeros((n, oh, ow, O))
    for i in range(oh):
        for j in range(ow):
            patch = x[:, i * stride:i * stride + kh, j * stride:j * stride + kw, :]
            out[:, i, j, :] = np.tensordot(patch, w, axes=([1, 2, 3], [0, 1, 2]))
    return out + b.reshape(1, 1, 1, O)


def _naive_forward(params, kind, obs, masks, states):
    """The policies written out a second time, element formulas in NumPy float64: nature_cnn (ppo2/policies.py:6-16), the
    LSTM step with gate order i, f, o, u and state [c | h] (a2c/utils.py:86-108) and its layer-normalised form (:110-147)."""
    p = {k: v.astype(np.float64) for k, v in params.items()}
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    h = obs.astype(np.float64) / 255.0
    for name, s in (("c1", 4), ("c2", 2), ("c3", 1)):
        h = np.maximum(_naive_conv(h, p[name + "/w"], p[name + "/b"], s), 0.0)
    h = np.maximum(h.reshape(h.shape[0], -1) @ p["fc1/w"] + p["fc1/b"], 0.0)
    if kind != "cnn":
        nenv, nh = states.shape[0], p["lstm1/wh"].shape[0]
        T = h.shape[0] // nenv

        def ln(x, g, b):
            u = x.mean(axis=1, keepdims=True)
            v = ((x - u) ** 2).mean(axis=1, keepdims=True)
            return (x - u) / np.sqrt(v + 1e-5) * g + b
        x, m = h.reshape(nenv, T, -1), masks.reshape(nenv, T).astype(np.float64)
        c, hh = states[:, :nh].astype(np.float64), states[:, nh:].astype(np.float64)
        rows = np.zeros((nenv, T, nh))
        for t in range(T):
            keep = (1.0 - m[:, t])[:, None]
            c, hh = c * keep, hh * keep
            if kind == "lnlstm":
                z = ln(x[:, t] @ p["lstm1/wx"], p["lstm1/gx"], p["lstm1/bx"]) + ln(hh @ p["lstm1/wh"], p["lstm1/gh"], p["lstm1/bh"]) + p["lstm1/b"]
            else:
                z = x[:, t] @ p["lstm1/wx"] + hh @ p["lstm1/wh"] + p["lstm1/b"]
            i, f, o, u = sig(z[:, :nh]), sig(z[:, nh:2 * nh]), sig(z[:, 2 * nh:3 * nh]), np.tanh(z[:, 3 * nh:])
            c = f * c + i * u
            hh = o * np.tanh(ln(c, p["lstm1/gc"], p["lstm1/bc"]) if kind == "lnlstm" else c)
            rows[:, t] = hh
        h = rows.reshape(nenv * T, nh)
    return h @ p["pi/w"] + p["pi/b"], h @ p["v/w"] + p["v/b"]


def test_image_policy_forward_against_a_second_derivation():
    """The torch restatement of the three image policies against the formulas written out in NumPy (a convolution from its
    definition, the LSTM gate by gate): logits and values to float64 rounding."""
    for kind in ("cnn", "lstm", "lnlstm"):
        params, args = _image_batch(kind, seed=3)
        _, _, obs, _, masks, _, _, _, _, states = args
        oracle = tfgraph.TrainOracle(params, kind, "categorical", tfgraph.LossCfg(0.01, 0.5, 0.5, True))
        with torch.no_grad():
            pi, vf, _ = oracle.forward(obs, masks, states)
        n_pi, n_vf = _naive_forward(params, kind, obs, masks, states)
        np.testing.assert_allclose(pi.numpy(), n_pi, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(vf.numpy(), n_vf, rtol=1e-10, atol=1e-12)


def test_rectifier_check_catches_a_zeroed_live_unit():
    """tests/util.check_rectifiers_against_the_oracles_own is what keeps the real-size GPU tests honest about the on/off
    pattern they borrow from the product: it passes on the oracle's own pattern and fails when one live unit is switched
    off (or a dead one on) in the pattern and activations handed to it."""
    import pytest
    params, args = _image_batch("cnn", seed=5)
    obs = args[2]
    oracle = tfgraph.TrainOracle(params, "cnn", "categorical", tfgraph.LossCfg(0.01, 0.5, 0.5, True))
    with torch.no_grad():
        oracle.forward(obs)
    acts = [a.numpy().reshape(-1).astype(np.float32) for a in oracle.last_acts]
    relu_on = [a > 0 for a in acts]
    assert util.check_rectifiers_against_the_oracles_own(oracle, relu_on, acts, obs) == 0
    layer = 1
    live = int(np.argmax(acts[layer]))                    # the most active unit of conv2's output
    bad_on, bad_acts = [r.copy() for r in relu_on], [a.copy() for a in acts]
    bad_on[layer][live] = False
    bad_acts[layer][live] = 0.0
    with pytest.raises(AssertionError):
        util.check_rectifiers_against_the_oracles_own(oracle, bad_on, bad_acts, obs)
    dead = int(np.argmin(oracle.last_pre[layer].numpy().reshape(-1)))
    bad_on = [r.copy() for r in relu_on]
    bad_on[layer][dead] = True
    with pytest.raises(AssertionError):
        util.check_rectifiers_against_the_oracles_own(oracle, bad_on, acts, obs)


def test_adam_first_step_closed_form():
    # after one step from m = v = 0:  theta -= lr * g / (|g| + eps * sqrt(1 - beta2))  (mpi_adam.py:30-35)
    p = [torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)]
    g = [torch.tensor([0.3, -1e-7, 0.0], dtype=torch.float64)]
    opt = tfgraph.Adam(p)
    opt.update(p, g, 1e-2)
    a = 1e-2 * np.sqrt(1 - 0.999) / (1 - 0.9)
    exp = np.array([1.0, -2.0, 0.5]) - a * (0.1 * g[0].numpy()) / (np.sqrt(0.001 * g[0].numpy() ** 2) + 1e-5)
    np.testing.assert_allclose(p[0].numpy(), exp, rtol=1e-12)


def test_clip_by_global_norm():
    gs = [torch.tensor([3.0, 0.0]), torch.tensor([[4.0]])]
    out, gn = tfgraph.clip_by_global_norm(gs, 0.5)
    assert abs(gn - 5.0) < 1e-12
    np.testing.assert_allclose(out[0].numpy(), [0.3, 0.0], rtol=1e-6)
    out, gn = tfgraph.clip_by_global_norm(gs, 10.0)     # below the threshold: unchanged
    np.testing.assert_allclose(out[1].numpy(), [[4.0]])


def test_distribution_consistency():
    """entropy == -E[logp] within 3 sigma, the property the reference's own (TF-only)
    test checks (baselines/common/distributions.py:281-295), on the same pdparams."""
    rng = np.random.RandomState(0)
    N = 100000
    mean = torch.tensor([[-.2, .3, .4, -.5]], dtype=torch.float64).repeat(N, 1)
    logstd = torch.tensor([[.1, -.5, .1, 0.8]], dtype=torch.float64)
    x = mean + torch.exp(logstd) * torch.as_tensor(rng.randn(N, 4))
    nlp = tfgraph.gaussian_neglogp(mean, logstd, x).numpy()
    ent = float(tfgraph.gaussian_entropy(mean, logstd)[0])
    assert abs(ent - nlp.mean()) < 3 * nlp.std() / np.sqrt(N)
    logits = torch.tensor([[-.2, .3, .5]], dtype=torch.float64).repeat(N, 1)
    u = torch.as_tensor(rng.rand(N, 3))
    a = torch.argmax(logits - torch.log(-torch.log(u)), dim=1)
    nlp = tfgraph.categorical_neglogp(logits, a).numpy()
    ent = float(tfgraph.categorical_entropy(logits)[0])
    assert abs(ent - nlp.mean()) < 3 * nlp.std() / np.sqrt(N)
