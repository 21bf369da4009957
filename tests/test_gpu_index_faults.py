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


_DIRECT_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from baselines_amd import _lib, hip, spaces
from baselines_amd.ppo2 import policies
from baselines_amd.ppo2.common_functions import DeviceRollout, Model
D, A, T, E, B = 17, 6, 16, 64, 256
g = torch.Generator(device="cuda"); g.manual_seed(2)
f = lambda *s: torch.randn(*s, device="cuda", generator=g)
model = Model(policy=policies.MlpPolicy64, ob_space=spaces.Box(-1, 1, shape=(D,)), ac_space=spaces.Box(-1, 1, shape=(A,)),
              nbatch_act=E, nbatch_train=B, nsteps=T, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5)
d = DeviceRollout(T, E, f(T, E, D), f(T, E, A), f(T, E, 1), f(T, E, 1), f(T, E).abs() + 1.0,
                  torch.zeros((T, E), dtype=torch.uint8, device="cuda"), f(E, 1), torch.zeros(E, dtype=torch.uint8, device="cuda"))
_, d.returns, d.advs_ori = hip.gae(d.rewards, d.values, d.dones, d.last_values, d.last_dones, 0.99, 0.95, want_advs=False)
d.advs = d.advs_ori.reshape(T, E)
np.random.seed(1)
out = model.train_on_rollout(3e-4, 0.2, d, 2, B)          # queued; nothing has been waited for yet
seen = []
try:
    model.check_faults()
except _lib.Ppo2HipError as e:
    seen.append("check_faults: " + str(e)[:60])
model.train_on_rollout(3e-4, 0.2, d, 2, B)
torch.cuda.synchronize()
try:
    model.train_on_rollout(3e-4, 0.2, d, 2, B)               # the update before this one faulted: raised on entry
except _lib.Ppo2HipError as e:
    seen.append("next update: " + str(e)[:60])
model.train_on_rollout(3e-4, 0.2, d, 2, B)
try:
    model.save(sys.argv[2])
except _lib.Ppo2HipError as e:
    seen.append("save: " + str(e)[:60])
print("SEEN", len(seen), seen)
"""


@pytest.mark.parametrize("forced", [False, True])
def test_direct_callers_of_train_on_rollout_see_a_faulted_launch(tmp_path, forced):
    """Model.train_on_rollout outside learn(): the fault word is read behind every update without stalling the stream, and
    a launch that gave up waiting for its peer workgroups (forced here: PPO2HIP_GRID_SYNC_SPINS=-1 makes the optimiser
    step folded into the epoch call raise the bit on every launch) surfaces from check_faults(), from the next update's
    entry and from save() — parameters left by such a launch cannot be trained on or written out unnoticed.  Without the
    forcing the same calls raise nothing."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "direct.py"
    script.write_text(_DIRECT_SCRIPT)
    env = dict(os.environ)
    if forced:
        env["PPO2HIP_GRID_SYNC_SPINS"] = "-1"
    res = subprocess.run([sys.executable, str(script), root, str(tmp_path / "model.pkl")], env=env, capture_output=True, text=True,
                         timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("SEEN")][0]
    if forced:
        assert line.startswith("SEEN 3"), line
        assert "gave up waiting" in line or "hand values" in line
        assert not os.path.exists(tmp_path / "model.pkl")
    else:
        assert line.startswith("SEEN 0"), line
        assert os.path.exists(tmp_path / "model.pkl")
