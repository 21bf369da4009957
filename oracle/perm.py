"""Minibatch index generation of ``ppo2.learn`` and the RNG underneath it.

Test infrastructure (see ``oracle/__init__.py``).

The reference shuffles with the global ``np.random`` (``baselines/ppo2/ppo2.py:113``,
``:130``), i.e. NumPy's legacy ``RandomState``: MT19937 seeded by
``init_genrand`` and a Fisher-Yates pass whose bounded draws use the
mask-and-reject rule.  ``MT19937``/``legacy_shuffle`` below restate that
published algorithm in pure Python so that the C++ generator of the product can
be checked against something that is not NumPy; ``tests/test_oracle_golden.py``
in turn pins this restatement against ``np.random.seed``/``np.random.shuffle``
itself and against the committed fixtures.
"""
import numpy as np


class MT19937:
    """Mersenne Twister (Matsumoto & Nishimura 1998), 32-bit outputs."""

    N, M = 624, 397

    def __init__(self, seed):
        mt = [0] * self.N
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, self.N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.mt = mt
        self.pos = self.N

    @classmethod
    def from_numpy_state(cls, state):
        """Build from ``np.random.get_state()`` (legacy tuple form)."""
        g = cls.__new__(cls)
        g.mt = [int(x) for x in state[1]]
        g.pos = int(state[2])
        return g

    def _regen(self):
        mt, N, M = self.mt, self.N, self.M
        for k in range(N):
            y = (mt[k] & 0x80000000) | (mt[(k + 1) % N] & 0x7FFFFFFF)
            mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ (0x9908B0DF if (y & 1) else 0)
        self.pos = 0

    def next_u32(self):
        if self.pos >= self.N:
            self._regen()
        y = self.mt[self.pos]
        self.pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def interval(self, mx):
        """Uniform integer in [0, mx], legacy ``rk_interval`` (mx < 2**32)."""
        if mx == 0:
            return 0
        mask = mx
        for s in (1, 2, 4, 8, 16):
            mask |= mask >> s
        while True:
            v = self.next_u32() & mask
            if v <= mx:
                return v


def legacy_shuffle(gen, x):
    """In-place Fisher-Yates of legacy ``RandomState.shuffle`` on a 1-D sequence."""
    for i in range(len(x) - 1, 0, -1):
        j = gen.interval(i)
        x[i], x[j] = x[j], x[i]


def minibatch_indices(nbatch, nbatch_train, noptepochs, shuffle=None):
    """Index lists of the non-recurrent update, ``baselines/ppo2/ppo2.py:110-116``.

    ``inds`` is created once and shuffled IN PLACE every epoch, so epoch k
    permutes epoch k-1's order.  ``shuffle`` defaults to ``np.random.shuffle``
    (the global stream, as in the reference).  Returns a list over epochs of
    lists over minibatches of int64 index arrays (copies).
    """
    shuffle = shuffle or np.random.shuffle
    inds = np.arange(nbatch)
    out = []
    for _ in range(noptepochs):
        shuffle(inds)
        ep = []
        for start in range(0, nbatch, nbatch_train):
            end = start + nbatch_train
            ep.append(inds[start:end].copy())
        out.append(ep)
    return out


def recurrent_minibatch_indices(nenvs, nsteps, nbatch_train, noptepochs, shuffle=None):
    """Index lists of the recurrent update, ``baselines/ppo2/ppo2.py:124-136``.

    ``envsperbatch = nbatch_train // nsteps`` (:128 overrides :125).  Returns a
    list over epochs of lists over minibatches of (mbenvinds, mbflatinds).
    """
    shuffle = shuffle or np.random.shuffle
    envinds = np.arange(nenvs)
    flatinds = np.arange(nenvs * nsteps).reshape(nenvs, nsteps)
    envsperbatch = nbatch_train // nsteps
    out = []
    for _ in range(noptepochs):
        shuffle(envinds)
        ep = []
        for start in range(0, nenvs, envsperbatch):
            end = start + envsperbatch
            mbenvinds = envinds[start:end].copy()
            ep.append((mbenvinds, flatinds[mbenvinds].ravel()))
        out.append(ep)
    return out
