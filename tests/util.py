"""Shared helpers for the parity tests: synthetic rollouts and oracle <-> arena plumbing."""
from collections import OrderedDict

import numpy as np

from oracle import tfgraph


def make_params(shapes, seed, pi_scale=0.01, jitter=0.0):
    rng = np.random.RandomState(seed)
    p = tfgraph.init_params(shapes, rng, pi_scale=pi_scale)
    if jitter:
        for k in p:
            p[k] = (p[k] + jitter * rng.randn(*p[k].shape)).astype(np.float32)
    return p


def flatten_params(params, spec):
    """Oracle dict -> flat float32 arena in the library's variable order (names must agree)."""
    flat = np.zeros(spec.nparams, np.float32)
    names = [v[0] for v in spec.vars]
    assert names == list(params.keys()), (names, list(params.keys()))
    for name, off, shp in spec.vars:
        a = np.asarray(params[name], np.float32)
        assert a.size == int(np.prod(shp)), (name, a.shape, shp)
        flat[off:off + a.size] = a.ravel()
    return flat


def unflatten(flat, spec, like):
    out = OrderedDict()
    for name, off, shp in spec.vars:
        n = int(np.prod(shp))
        out[name] = np.asarray(flat[off:off + n]).reshape(like[name].shape)
    return out


def synthetic_mlp_rollout(rng, T, E, D, A, R, gaussian, params, nlp_noise=0.3):
    """Time-major synthetic rollout whose actions/values/neglogpacs come from `params`
    (so ratio ~ 1 as in real PPO) with noise on the old neglogp so clipping is exercised."""
    import torch
    N = T * E
    obs = rng.randn(T, E, D).astype(np.float32)
    p = tfgraph.to_torch(params, torch.float64, requires_grad=False)
    pi, vf = tfgraph.mlp_forward(p, torch.as_tensor(obs.reshape(N, D)).double())
    pi, vf = pi.numpy(), vf.numpy()
    if gaussian:
        std = np.exp(params["logstd"].astype(np.float64))
        actions = (pi + std * rng.randn(N, A)).astype(np.float32)
        nlp = tfgraph.gaussian_neglogp(torch.as_tensor(pi), torch.as_tensor(params["logstd"]).double(),
                                       torch.as_tensor(actions).double()).numpy()
        actions = actions.reshape(T, E, A)
    else:
        u = rng.rand(N, A)
        actions = np.argmax(pi - np.log(-np.log(u)), axis=1).astype(np.int32)
        nlp = tfgraph.categorical_neglogp(torch.as_tensor(pi), torch.as_tensor(actions)).numpy()
        actions = actions.reshape(T, E)
    nlp = (nlp + nlp_noise * rng.randn(N)).astype(np.float32).reshape(T, E)
    values = (vf + 0.3 * rng.randn(N, R)).astype(np.float32).reshape(T, E, R)
    rewards = rng.randn(T, E, R).astype(np.float32)
    dones = rng.rand(T, E) < 0.05
    last_values = rng.randn(E, R).astype(np.float32)
    last_dones = rng.rand(E) < 0.05
    return dict(obs=obs, actions=actions, neglogpacs=nlp, values=values, rewards=rewards, dones=dones,
                last_values=last_values, last_dones=last_dones)


def conv_out(n, k, s):
    return (n - k) // s + 1


def cnn_flat(H, W):
    h, w = conv_out(conv_out(conv_out(H, 8, 4), 4, 2), 3, 1), conv_out(conv_out(conv_out(W, 8, 4), 4, 2), 3, 1)
    return h * w * 64


def synthetic_cnn_rollout(rng, T, E, hwc, nact, params, kind="cnn", nlstm=0, nlp_noise=0.2):
    """uint8 image rollout; actions / values / neglogpacs from `params` (recurrent policies are
    stepped through time with done-mask resets, as the runner would)."""
    import torch
    H, W, C = hwc
    obs = rng.randint(0, 256, size=(T, E, H, W, C)).astype(np.uint8)
    dones = rng.rand(T, E) < 0.1
    p = tfgraph.to_torch(params, torch.float64, requires_grad=False)
    actions = np.zeros((T, E), np.int32)
    values = np.zeros((T, E, 1), np.float32)
    nlp = np.zeros((T, E), np.float32)
    state = np.zeros((E, 2 * nlstm), np.float32) if nlstm else None
    states0 = None if state is None else state.copy()
    with torch.no_grad():
        for t in range(T):
            if kind == "cnn":
                pi, vf = tfgraph.cnn_forward(p, torch.as_tensor(obs[t]))
            else:
                pi, vf, snew = tfgraph.cnn_lstm_forward(p, torch.as_tensor(obs[t]), torch.as_tensor(dones[t].astype(np.float64)),
                                                        torch.as_tensor(state).double(), E, 1, layer_norm=(kind == "lnlstm"))
                state = snew.numpy().astype(np.float32)
            pi = pi.numpy()
            u = rng.rand(E, nact)
            a = np.argmax(pi - np.log(-np.log(u)), axis=1).astype(np.int32)
            actions[t] = a
            values[t, :, 0] = vf.numpy()[:, 0] + 0.3 * rng.randn(E)
            nlp[t] = tfgraph.categorical_neglogp(torch.as_tensor(pi), torch.as_tensor(a)).numpy() + nlp_noise * rng.randn(E)
    rewards = rng.randn(T, E, 1).astype(np.float32)
    return dict(obs=obs, actions=actions, neglogpacs=nlp, values=values, rewards=rewards, dones=dones,
                last_values=rng.randn(E, 1).astype(np.float32), last_dones=rng.rand(E) < 0.1, states=states0)


def check_rectifiers_against_the_oracles_own(oracle, relu_on, acts, obs, masks=None, states=None):
    """The pattern handed to the oracle is the product's own, so a unit the kernels wrongly switch OFF would be off on
    both sides.  Here the float64 restatement runs its forward once more with its OWN rectifiers: the two on/off patterns
    may differ on fewer than 1e-5 of the units, every unit they differ on must have a float64 pre-activation within 1e-5
    of zero, and the product's activations must be the oracle's own to 2e-5 — a kernel that zeroes (or invents) live
    units at its production tiling fails all three."""
    import torch
    with torch.no_grad():
        oracle.forward(obs, masks, states, relu_on=None)
    for name, on, mine, z, theirs in zip(("c1", "c2", "c3", "fc1"), relu_on, acts, oracle.last_pre, oracle.last_acts):
        z = z.numpy().reshape(-1)
        differ = (z > 0) != on.reshape(-1)
        frac = float(differ.mean())
        worst = float(np.abs(z[differ]).max()) if differ.any() else 0.0
        off_here = int(((z > 0) & ~on.reshape(-1)).sum())      # live in float64, off in the pattern under test
        assert frac < 1e-5, (name, "rectifier patterns differ on %.2e of the units (%d off that float64 has on, %d on that it has "
                                   "off; largest |pre-activation| among them %.3e)" % (frac, off_here, int(differ.sum()) - off_here, worst))
        assert worst < 1e-5, (name, "a unit far from zero (%.3e) is on in one pattern and off in the other" % worst)
        err = float(np.abs(mine.reshape(-1) - theirs.numpy().reshape(-1)).max())
        assert err <= 2e-5, (name, err)
    return sum(int(((z.numpy().reshape(-1) > 0) != on.reshape(-1)).sum()) for on, z in zip(relu_on, oracle.last_pre))


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(target, world, args=(), timeout=240):
    """Spawn ``world`` processes running ``target(rank, world, port, *args, out)`` and return what rank 0 puts on
    ``out``.  The queue is polled against a deadline and the children's exit codes, so a rank that dies before
    reporting (import error, failed rendezvous, an assertion) fails the test at once instead of blocking it."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (out,)) for r in range(world)]
    for p in procs:
        p.start()
    deadline = time.time() + timeout
    res = None
    try:
        while res is None:
            if not out.empty():
                res = out.get()
                break
            dead = [p for p in procs if p.exitcode not in (None, 0)]
            if dead:
                raise AssertionError("rank process exited with code %s before reporting" % dead[0].exitcode)
            if all(p.exitcode == 0 for p in procs) and out.empty():
                raise AssertionError("all ranks exited without a result")
            if time.time() > deadline:
                raise AssertionError("ranks did not report within %d s" % timeout)
            time.sleep(0.05)
        for p in procs:
            p.join(max(1.0, deadline - time.time()))
            assert p.exitcode == 0, "rank exited with code %s" % p.exitcode
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    return res
