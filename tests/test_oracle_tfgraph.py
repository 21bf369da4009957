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
    return float(total)


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
            an = float(g.view(-1)[i])
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
