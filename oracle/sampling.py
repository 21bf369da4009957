"""Counter-based noise and the action-sampling formulas of the act graph.

Test infrastructure (see ``oracle/__init__.py``).

The reference samples inside TensorFlow (``tf.random_normal`` / ``tf.random_uniform``,
``baselines/common/distributions.py:165,212``), whose stream nothing outside TensorFlow reproduces;
what can be held to parity is (i) the generator against its published known-answer vectors and
(ii) the formulas applied to given noise: ``mean + std * z`` (:212), Gumbel-max
``argmax(logits - log(-log(u)))`` (:165-166) and ``neglogp`` of the drawn action (:141-156, :202-205).

Generator: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11),
restated from the paper: ten rounds of two 32x32->64 multiplies with the constants below, the key
bumped by the Weyl constants between rounds.
"""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: (k0, k1) -> uint32 [..., 4]."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & MASK, int(key[1]) & MASK
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(MASK)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(MASK)
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return np.stack(c, axis=-1).astype(np.uint32)


def _counters(B, ncalls, step):
    """One generator call per (row, call): counter = (row, call, step low, step high)."""
    ctr = np.zeros((B, ncalls, 4), np.uint32)
    ctr[..., 0] = np.arange(B, dtype=np.uint32)[:, None]
    ctr[..., 1] = np.arange(ncalls, dtype=np.uint32)[None, :]
    ctr[..., 2] = np.uint32(step & MASK)
    ctr[..., 3] = np.uint32((step >> 32) & MASK)
    return ctr


def uniforms(B, A, seed, step):
    """float32 U(0,1) [B, A], open at both ends: (x >> 8) * 2^-24 + 2^-25."""
    ncalls = (A + 3) // 4
    x = philox4x32_10(_counters(B, ncalls, step), (seed & MASK, (seed >> 32) & MASK)).reshape(B, ncalls * 4)[:, :A]
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24) + np.float32(2.0 ** -25)).astype(np.float32)


def normals(B, A, seed, step):
    """float32 N(0,1) [B, A]: Box-Muller on the uniform pairs (0,1) and (2,3) of each call; element
    4j+0 / 4j+1 are the cosine / sine branch of the first pair, 4j+2 / 4j+3 of the second."""
    ncalls = (A + 3) // 4
    u = uniforms(B, ncalls * 4, seed, step).reshape(B, ncalls, 2, 2).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(u[..., 0]))
    th = 2.0 * np.pi * u[..., 1]
    z = np.stack([r * np.cos(th), r * np.sin(th)], axis=-1)       # [B, ncalls, 2 pairs, 2 branches]
    return z.reshape(B, ncalls * 4)[:, :A].astype(np.float32)


def sample_gaussian(mean, logstd, z):
    """``mean + std * random_normal`` (distributions.py:212) and neglogp of the result (:202-205), float32."""
    mean, z = np.asarray(mean, np.float32), np.asarray(z, np.float32)
    logstd = np.asarray(logstd, np.float32).reshape(1, -1)
    std = np.exp(logstd)
    a = (mean + std * z).astype(np.float32)
    nlp = (0.5 * np.sum(np.square((a - mean) / std), axis=-1) + 0.5 * np.log(2.0 * np.pi) * mean.shape[-1]
           + np.sum(logstd, axis=-1))
    return a, nlp.astype(np.float32)


def gumbel_scores(logits, u):
    return np.asarray(logits, np.float64) - np.log(-np.log(np.asarray(u, np.float64)))


def sample_categorical(logits, u):
    """``argmax(logits - log(-log(u)))`` (distributions.py:165-166) and neglogp (:141-156)."""
    logits = np.asarray(logits, np.float32)
    a = np.argmax(gumbel_scores(logits, u), axis=-1).astype(np.int32)
    l = logits.astype(np.float64)
    mx = l.max(-1, keepdims=True)
    nlp = np.log(np.exp(l - mx).sum(-1)) - (np.take_along_axis(l, a[:, None].astype(np.int64), -1)[:, 0] - mx[:, 0])
    return a, nlp.astype(np.float32)


def done_bootstrap(rewards, dones, v_preds, gamma):
    """The fork's bootstrap off the reset observation, ``baselines/ppo2/common_functions.py:265-276``:
    where an env is done, a positive first reward channel adds ``gamma * V`` to every channel, otherwise
    only channels 2.. receive it.  ``rewards`` [E, R] float32 is updated as NumPy would (float64 product,
    float32 store)."""
    rewards = np.array(rewards, np.float32)
    gamma = np.broadcast_to(np.asarray(gamma, np.float64), (rewards.shape[1],))
    for i in range(len(dones)):
        if dones[i]:
            if rewards[i][0] > 0:
                rewards[i] += gamma * v_preds[i]
            else:
                rewards[i][2:] += gamma[2:] * v_preds[i][2:]
    return rewards
