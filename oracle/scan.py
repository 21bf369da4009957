"""Reverse scans of the rollout: GAE/returns, A2C discounted returns, flatten.

Test infrastructure (see ``oracle/__init__.py``).  NumPy restatements that keep
the reference's dtype flow: float32 inputs, float64 carry, float32 store.
"""
import numpy as np


def gae(rewards, values, dones, last_values, last_dones, gamma, lam):
    """GAE(lambda) advantages and returns over a time-major rollout.

    Follows ``baselines/ppo2/common_functions.py:308-328``.

    rewards, values : float32 [T, E, R]   (``mb_rewards``, ``mb_values`` :298,:300)
    dones           : bool    [T, E]      done flag observed BEFORE step t (:258)
    last_values     : float32 [E, R]      ``model.value(obs)`` after the last step (:304)
    last_dones      : bool    [E]         ``self.dones`` after the last step (:319)
    gamma           : scalar or length-R sequence (``self.gamma = np.array(gamma)`` :232)
    lam             : scalar

    The reference writes the five reward channels out by hand (:319,:322); the
    loop below broadcasts the mask over R instead, which is the same arithmetic
    for R == 5 and extends it to any R.  The carry ``lastgaelam`` stays float64
    while ``mb_advs[t]`` rounds to float32 (chained assignment at :327);
    ``mb_returns = mb_advs + mb_values`` is a float32 add (:328).
    """
    rewards = np.asarray(rewards, dtype=np.float32)
    values = np.asarray(values, dtype=np.float32)
    T, E, R = rewards.shape
    gamma = np.broadcast_to(np.asarray(gamma, dtype=np.float64), (R,))
    last_values = np.asarray(last_values, dtype=np.float32).reshape(E, R)
    dones = np.asarray(dones).astype(bool).reshape(T, E)
    last_dones = np.asarray(last_dones).astype(bool).reshape(E)
    advs = np.zeros_like(rewards)
    lastgaelam = 0
    for t in reversed(range(T)):
        nextnonterminal = np.zeros((E, R))
        if t == T - 1:
            nextnonterminal[:, :] = (1.0 - last_dones)[:, None]
            nextvalues = last_values
        else:
            nextnonterminal[:, :] = (1.0 - dones[t + 1])[:, None]
            nextvalues = values[t + 1]
        delta = rewards[t] + gamma * nextvalues * nextnonterminal - values[t]
        advs[t] = lastgaelam = delta + gamma * lam * nextnonterminal * lastgaelam
    returns = advs + values
    return advs, returns


def gae_single_env(rew, vpred, new, nextvpred, gamma, lam):
    """Single-env scalar GAE, ``baselines/ppo1/pposgd_simple.py:64-78``.

    ``new[t]`` is the episode-start flag of step t; ``nextvpred`` already holds
    ``last_value * (1 - last_done)``.  Returns (adv float32 [T], tdlamret).
    """
    new = np.append(new, 0)
    vpred_ext = np.append(vpred, nextvpred)
    T = len(rew)
    adv = np.empty(T, 'float32')
    lastgaelam = 0
    for t in reversed(range(T)):
        nonterminal = 1 - new[t + 1]
        delta = rew[t] + gamma * vpred_ext[t + 1] * nonterminal - vpred_ext[t]
        adv[t] = lastgaelam = delta + gamma * lam * nonterminal * lastgaelam
    return adv, adv + vpred


def gae_single_env_f64(rew, vpred, new, nextvpred, gamma, lam):
    """``baselines/ppo1/pposgd_simple.py:64-78`` with the scalar promotion of the NumPy the
    reference was written for: a Python-float ``gamma`` times a float32 ELEMENT gives float64
    there (value-based casting), so ``delta`` and the carry are float64 and only the store into
    ``gaelam`` rounds.  NumPy >= 2 keeps such products in float32; ``gae_single_env`` above
    follows whichever NumPy runs it."""
    new = np.append(np.asarray(new), 0)
    vpred_ext = np.append(np.asarray(vpred, np.float32), np.float32(nextvpred)).astype(np.float64)
    rew = np.asarray(rew, np.float32).astype(np.float64)
    T = len(rew)
    adv = np.empty(T, 'float32')
    lastgaelam = 0.0
    for t in reversed(range(T)):
        nonterminal = float(1 - new[t + 1])
        delta = rew[t] + gamma * vpred_ext[t + 1] * nonterminal - vpred_ext[t]
        adv[t] = lastgaelam = delta + gamma * lam * nonterminal * lastgaelam
    return adv, adv + np.asarray(vpred, np.float32)


def discount(x, gamma):
    """``baselines/common/math_util.py:5-23``: ``scipy.signal.lfilter([1], [1, -gamma], x[::-1],
    axis=0)[::-1]``.  lfilter's transposed direct form II with these coefficients is
    ``y[n] = x[n] + z``, ``z = gamma * y[n]`` in float64, i.e. a reverse scan over axis 0."""
    x = np.asarray(x)
    y = np.zeros(x.shape, np.float64)
    carry = np.zeros(x.shape[1:], np.float64)
    for t in reversed(range(x.shape[0])):
        carry = x[t].astype(np.float64) + gamma * carry
        y[t] = carry
    return y


def discount_with_boundaries_legacy(X, New, gamma):
    """``baselines/common/math_util.py:63-73`` with float64 scalar arithmetic and a store in
    ``X``'s dtype each step (``Y = np.zeros_like(X)``), the carry being the stored value."""
    X = np.asarray(X)
    Y = np.zeros_like(X)
    T = X.shape[0]
    Y[T - 1] = X[T - 1]
    for t in range(T - 2, -1, -1):
        Y[t] = X[t].astype(np.float64) + gamma * Y[t + 1].astype(np.float64) * float(1 - New[t + 1])
    return Y


def discount_with_dones(rewards, dones, gamma):
    """``baselines/a2c/utils.py:154-160`` — Python-float (f64) reverse scan."""
    discounted = []
    r = 0
    for reward, done in zip(rewards[::-1], dones[::-1]):
        r = reward + gamma * r * (1. - done)
        discounted.append(r)
    return discounted[::-1]


def a2c_returns(rewards, dones, last_values, gamma):
    """Bootstrapped n-step returns of the A2C runner, ``baselines/a2c/a2c.py:117-126``.

    rewards     : float32 [E, T]   env-major (after the swapaxes at :111)
    dones       : bool    [E, T]   done flag AFTER each step (``mb_dones[:, 1:]`` :116)
    last_values : [E]
    Returns float32 [E, T] (the scan runs in Python floats and is stored back
    into the float32 ``mb_rewards`` row, :126).
    """
    rewards = np.asarray(rewards, dtype=np.float32)
    out = np.empty_like(rewards)
    last_values = np.asarray(last_values).tolist()
    for n, (rew, dn, value) in enumerate(zip(rewards, np.asarray(dones), last_values)):
        rew = rew.tolist()
        dn = dn.tolist()
        if dn[-1] == 0:
            rew = discount_with_dones(rew + [value], dn + [0], gamma)[:-1]
        else:
            rew = discount_with_dones(rew, dn, gamma)
        out[n] = rew
    return out


def discount_with_boundaries(X, New, gamma):
    """``baselines/common/math_util.py:63-73``."""
    Y = np.zeros_like(X)
    T = X.shape[0]
    Y[T - 1] = X[T - 1]
    for t in range(T - 2, -1, -1):
        Y[t] = X[t] + gamma * Y[t + 1] * (1 - New[t + 1])
    return Y


def sf01(arr):
    """Swap and flatten axes 0 and 1, ``baselines/ppo2/common_functions.py:347-352``."""
    s = arr.shape
    return arr.swapaxes(0, 1).reshape(s[0] * s[1], *s[2:])
