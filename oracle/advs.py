"""Advantage post-processing on the host side of the reference.

Test infrastructure (see ``oracle/__init__.py``).
"""
import numpy as np


def compute_advs(max_return, values, advs_ori, action_diff=(), action_std=()):
    """Fork-only scalarisation of the 5-channel advantage.

    Follows ``baselines/ppo2/common_functions.py:359-401`` statement by
    statement (both Python loops over N kept), including the dead terms whose
    weight is 0 (:395), so that dtype promotion is the reference's own:
    ``max_return`` is a Python list of ints, hence ``dir_v`` is float64;
    ``advs_dir_norm`` is float32 (``zeros_like(advs_ori)``, :364) and its mean
    (:367) a float32 reduction; the result array is float64 (:386).

    Note on NumPy versions: ``mean_dir_length + 1e-8`` (:392) is float32 + Python
    float.  NumPy >= 2 keeps float32 there, the NumPy 1.x the reference was
    written for promoted to float64.  The two differ by at most one float32 ulp
    of the result; parity tests use rtol 1e-6 for this function.
    """
    dir_v = np.asarray(max_return - values)
    dir_v_length = np.sqrt(np.sum(np.multiply(dir_v, dir_v), axis=1))

    advs_dir_norm = np.zeros_like(advs_ori)
    for t in range(len(values)):
        advs_dir_norm[t] = dir_v[t] / (dir_v_length[t] + 1e-8)
    mean_dir = np.mean(advs_dir_norm, axis=0)
    mean_dir_length = np.sqrt(np.sum(mean_dir * mean_dir))

    advs_dir = np.zeros(len(values))
    advs_grad = np.zeros(len(values))
    advs = np.zeros(len(values))
    for t in range(len(values)):
        advs_grad[t] = np.sum(np.multiply(advs_ori[t], mean_dir_length)) / (mean_dir_length + 1e-8)
        advs_dir[t] = np.sum(np.multiply(advs_ori[t], dir_v[t])) / (dir_v_length[t] + 1e-8)
        advs[t] = 0.0 * advs_dir[t] + 1.0 * advs_grad[t]
        if len(action_diff) != 0:
            if np.all(action_diff[t] < action_std[t] * 0.6):
                advs[t] = -1
    return advs


def normalize_advs(advs):
    """Per-minibatch normalisation, ``baselines/ppo2/common_functions.py:67``.

    Population std (ddof 0), epsilon added outside the square root; runs in the
    dtype that arrives (float64 from ``compute_advs``, float32 upstream).
    """
    return (advs - advs.mean()) / (advs.std() + 1e-8)
