"""Action-distribution descriptors (host side).

The arithmetic of the distributions — neglogp, entropy and their gradients — lives in
csrc/loss.hip; these classes only carry what the policy builders need to know:
which family an action space maps to and the sizes/dtypes of its parameters and samples
(baselines/common/distributions.py:49-109, make_pdtype :238-250).
"""
import numpy as np

from .. import _lib
from ..spaces import is_discrete


class PdType:
    dist = None

    def param_shape(self):
        raise NotImplementedError

    def sample_shape(self):
        raise NotImplementedError

    def sample_dtype(self):
        raise NotImplementedError


class DiagGaussianPdType(PdType):
    dist = _lib.DIST_GAUSSIAN

    def __init__(self, size):
        self.size = int(size)

    def param_shape(self):
        return [2 * self.size]

    def sample_shape(self):
        return [self.size]

    def sample_dtype(self):
        return np.float32


class CategoricalPdType(PdType):
    dist = _lib.DIST_CATEGORICAL

    def __init__(self, ncat):
        self.ncat = int(ncat)
        self.size = self.ncat

    def param_shape(self):
        return [self.ncat]

    def sample_shape(self):
        return []

    def sample_dtype(self):
        return np.int32


def make_pdtype(ac_space):
    """Box -> DiagGaussian, Discrete -> Categorical; other spaces are not on this path."""
    if is_discrete(ac_space):
        return CategoricalPdType(ac_space.n)
    if hasattr(ac_space, "shape") and len(ac_space.shape) == 1:
        return DiagGaussianPdType(ac_space.shape[0])
    raise NotImplementedError("action space %r" % (ac_space,))
