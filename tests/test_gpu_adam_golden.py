"""The optimiser kernel and the policies' initial values against what the reference's own code produces.  GPU only.

``ppo2hip_adam_step`` runs the 10 recorded ``MpiAdam.update`` calls of tests/golden/reference_mpi_adam.npz (the
reference's unmodified NumPy code, see tests/test_adam_golden.py for the file and the tolerance): gradients are
the float32 SUM over ranks, ``grad_scale`` = 1/size where the reference divides (baselines/common/mpi_adam.py:27-28).
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def adam_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_mpi_adam.npz"), allow_pickle=False)


def test_adam_kernel_matches_the_reference_update(adam_golden):
    from baselines_amd import hip
    z = adam_golden
    worst = 0.0
    for name in z["cases"]:
        size = int(name.split("_")[1][1:])
        eps = float(name.split("_")[2][3:])
        mean = name.endswith("_mean")
        theta = torch.from_numpy(z[name + "_theta0"].copy()).cuda()
        m, v = torch.zeros_like(theta), torch.zeros_like(theta)
        scratch = hip.adam_scratch(theta.numel(), "cuda")
        for t, lr in enumerate(z[name + "_stepsizes"]):
            g = torch.from_numpy(z[name + "_grads"][t].sum(axis=0, dtype=np.float32)).cuda()
            hip.adam_step(theta, g, m, v, float(lr), t + 1, scratch, grad_scale=(1.0 / size if mean else 1.0), epsilon=eps)
            want = z[name + "_theta"][t]
            err = float(np.abs(theta.cpu().numpy() - want).max())
            worst = max(worst, err)
            assert err <= 3e-7 * max(1.0, np.abs(want).max() / 2), (name, t, err)
            wm, wv = z[name + "_m"][t], z[name + "_v"][t]
            np.testing.assert_allclose(m.cpu().numpy(), wm, rtol=2e-6, atol=1e-6 * np.abs(wm).max())
            np.testing.assert_allclose(v.cpu().numpy(), wv, rtol=2e-6, atol=1e-6 * np.abs(wv).max())
    print("adam kernel vs the reference's MpiAdam.update over 16 cases x 10 updates: max parameter diff %.2e" % worst)


def test_policies_start_from_the_reference_initialisers():
    """Every weight draws ortho_init normals from the global np.random stream in variable creation order
    (baselines/a2c/utils.py:25-40, pinned by the golden ortho_* vectors) at the reference's scales: sqrt(2) trunks and
    convs, 0.01 for a ``pi`` head built with init_scale=0.01 (MlpPolicy, CnnPolicy, the 64-64 MlpPolicy), fc's default
    1.0 for the recurrent policies' ``pi`` (baselines/ppo2/policies.py:33,75) and for every value head and LSTM matrix;
    biases and logstd zero, layer-norm gains one."""
    from baselines_amd import spaces
    from baselines_amd.a2c.utils import ortho_init
    from baselines_amd.ppo2 import policies as pol
    img, box, disc = spaces.Box(0, 255, shape=(84, 84, 4), dtype=np.uint8), spaces.Box(-1, 1, shape=(6,)), spaces.Discrete(6)
    ob = spaces.Box(-1, 1, shape=(17,))
    cases = [
        (lambda: pol.MlpPolicy(None, ob, box, 5, reuse=False, model_name="m0"), 0.01),
        (lambda: pol.MlpPolicy64(None, ob, box, 8, 1, reuse=False), 0.01),
        (lambda: pol.CnnPolicy(None, img, disc, 8, 1, reuse=False), 0.01),
        (lambda: pol.LstmPolicy(None, img, disc, 8, 1, reuse=False), 1.0),
        (lambda: pol.LnLstmPolicy(None, img, disc, 8, 1, reuse=False), 1.0),
    ]
    for build, pi_scale in cases:
        pol.reset_scopes()
        np.random.seed(17)
        p = build()
        got = p.arena.get_list()
        np.random.seed(17)
        for (name, _, shp), g in zip(p.spec.vars, got):
            leaf = name.split("/")[-1]
            if leaf in ("w", "wx", "wh"):
                scale = pi_scale if name == "pi/w" else (1.0 if name in ("vf/w", "v/w") or leaf in ("wx", "wh") else math.sqrt(2.0))
                want = ortho_init(scale)(shp)
                assert np.array_equal(g, want), (type(p).__name__, name)
            elif leaf in ("gx", "gh", "gc"):
                assert np.all(g == 1.0), name
            else:
                assert np.all(g == 0.0), name
        # the head's scale shows in the logits' size: 100x between the two kinds of policy
        w = dict(zip([n for n, _, _ in p.spec.vars], got))["pi/w"]
        col = float(np.linalg.norm(w, axis=0).max())
        assert (col < 0.02) if pi_scale == 0.01 else (0.5 < col <= 1.01), (type(p).__name__, col)
