"""Caller-supplied index lists with an entry that names no sample: the kernels replace it by sample 0 instead of reading
out of bounds, and the library reports it (ppo2hip_take_faults) where NumPy would raise IndexError in the reference
(baselines/ppo2/ppo2.py:117).  GPU only."""
import numpy as np
import pytest
import torch

from oracle import scan, tfgraph
from tests import util

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _case(hidden, R, upstream):
    from baselines_amd import _lib, hip
    D, A, T, E = 17, 6, 8, 32
    spec = hip.PolicySpec(_lib.POLICY_MLP, _lib.DIST_GAUSSIAN, A, R, ob_dim=D, hidden=hidden,
                          order=_lib.ORDER_UPSTREAM if upstream else _lib.ORDER_FORK)
    shapes = tfgraph.mlp_param_shapes(D, hidden, A, R, True, "upstream" if upstream else "fork")
    params = util.make_params(shapes, 5, pi_scale=0.5, jitter=0.02)
    ro = util.synthetic_mlp_rollout(np.random.RandomState(6), T, E, D, A, R, True, params)
    _, rets = scan.gae(ro["rewards"], ro["values"], ro["dones"], ro["last_values"], ro["last_dones"], 0.99, 0.95)
    advs = (rets - ro["values"]).sum(axis=2).astype(np.float32)
    rref = hip.RolloutRef(spec, dev(ro["obs"]), dev(ro["actions"]), dev(rets), dev(ro["values"]), dev(ro["neglogpacs"]),
                          dev(advs), T, E)
    theta = dev(util.flatten_params(params, spec))
    return hip, _lib, spec, rref, theta, T, E


@pytest.mark.parametrize("hidden,R,upstream,route", [((64, 64), 1, True, None), ((64, 64), 1, True, "layered"),
                                                      ((64, 64), 1, True, "fused32"), ((256, 128), 5, False, None)])
def test_out_of_range_index_is_reported_not_followed(hidden, R, upstream, route, monkeypatch):
    hip, lib, spec, rref, theta, T, E = _case(hidden, R, upstream)
    if route:
        monkeypatch.setenv("PPO2HIP_MLP_PATH", route)
    B = 128
    assert hip.take_faults() == 0
    good = np.random.RandomState(1).permutation(T * E)[:B].astype(np.int32)
    ws = torch.empty(spec.grad_workspace_bytes(B), dtype=torch.uint8, device="cuda")
    grads, losses = torch.zeros(spec.nparams, device="cuda"), torch.zeros(8, device="cuda")
    lossd = lib.LossDesc(0.0, 0.5, 0.5, 1 if upstream else 0)

    def run(idx_host):
        d_idx = dev(idx_host)
        stats = hip.adv_stats(rref.keep[5], d_idx, B, 1, T, E)
        hip.ppo_grad(spec, lossd, theta, rref, d_idx, B, stats, 0.2, 1.0 / B, grads, losses, ws)

    run(good)
    assert hip.take_faults() == 0                       # a good list raises nothing
    for bad_value in (T * E, -1, 2 ** 31 - 1):
        bad = good.copy()
        bad[77] = bad_value
        run(bad)                                         # completes: no out-of-bounds access
        with pytest.raises(lib.Ppo2HipError, match="out-of-range"):
            hip.take_faults()
        assert hip.take_faults() == 0                    # collected and cleared
        assert torch.isfinite(grads).all()
    run(good)
    assert hip.take_faults() == 0


def test_compose_reports_a_bad_permutation_entry():
    from baselines_amd import hip
    n = 1000
    prev = dev(np.random.RandomState(2).permutation(n).astype(np.int32))
    sigma = np.random.RandomState(3).permutation(n).astype(np.int32)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    hip.compose(prev, dev(sigma), out)
    assert hip.take_faults() == 0
    sigma[5] = n
    hip.compose(prev, dev(sigma), out)
    assert hip.take_faults(raise_on_fault=False) == hip.FAULT_COMPOSE_RANGE
    assert hip.take_faults() == 0


def test_out_of_range_index_on_image_rollouts_is_reported_not_followed():
    """nature_cnn: conv1's kernels read the minibatch's frames where they lie in the rollout, through the frame list the
    gather leaves — a bad index must become frame 0 there, not an address 2^31 frames away."""
    from baselines_amd import _lib, hip
    hwc, nact, T, E, B = (84, 84, 4), 6, 2, 8, 16
    spec = hip.PolicySpec(_lib.POLICY_CNN, _lib.DIST_CATEGORICAL, nact, 1, ob_hwc=hwc)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    theta = torch.randn(spec.nparams, device="cuda", generator=g) * 0.03
    obs = torch.randint(0, 256, (T, E) + hwc, device="cuda", generator=g, dtype=torch.uint8)
    f = lambda *s: torch.randn(*s, device="cuda", generator=g)
    acts = torch.randint(0, nact, (T, E), device="cuda", generator=g, dtype=torch.int32)
    rref = hip.RolloutRef(spec, obs, acts, f(T, E, 1), f(T, E, 1), f(T, E).abs() + 1.0, f(T, E), T, E)
    ws = torch.empty(spec.grad_workspace_bytes(B), dtype=torch.uint8, device="cuda")
    grads, losses = torch.zeros(spec.nparams, device="cuda"), torch.zeros(8, device="cuda")
    good = np.random.RandomState(1).permutation(T * E)[:B].astype(np.int32)

    def run(idx_host):
        d_idx = dev(idx_host)
        stats = hip.adv_stats(rref.keep[5], d_idx, B, 1, T, E)
        hip.ppo_grad(spec, _lib.LossDesc(0.01, 0.5, 0.5, 1), theta, rref, d_idx, B, stats, 0.1, 1.0 / B, grads, losses, ws)

    assert hip.take_faults() == 0
    run(good)
    assert hip.take_faults() == 0
    for bad_value in (T * E, -1, 2 ** 31 - 1):
        bad = good.copy()
        bad[5] = bad_value
        run(bad)
        with pytest.raises(_lib.Ppo2HipError, match="out-of-range"):
            hip.take_faults()
        assert hip.take_faults() == 0
        assert torch.isfinite(grads).all()
