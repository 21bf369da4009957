"""RunningMeanStd of baselines/common/mpi_running_mean_std.py: running sums kept in HBM, the
per-update [sum | sumsq | count] vector exchanged with a SUM all-reduce over the data-parallel group
(RCCL in place of MPI.COMM_WORLD.Allreduce, :41)."""
import numpy as np
import torch

from .. import hip


class RunningMeanStd(object):
    """Same constructor, ``update`` and ``mean`` / ``std`` as the reference's class (:4-43).  The
    totals live in one device float64 vector ``[sum | sumsq | count]``; ``comm`` is a
    ``DataParallel`` context (None: single process)."""

    def __init__(self, epsilon=1e-2, shape=(), comm=None):
        self.shape = tuple(shape)
        self.n = int(np.prod(self.shape)) if self.shape else 1
        st = np.concatenate([np.zeros(self.n), np.full(self.n, epsilon), [epsilon]])      # :7-22
        self.state = torch.from_numpy(st).cuda()
        self.comm = comm
        self._zeros = torch.zeros(self.n, dtype=torch.float64, device="cuda")

    def update(self, x):
        if not isinstance(x, torch.Tensor):
            x = np.asarray(x)
            if x.dtype not in (np.float32, np.float64):
                x = x.astype(np.float64)
            x = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        addvec = hip.column_moments(x.contiguous(), self._zeros)                          # :40
        if self.comm is not None and self.comm.size > 1:
            self.comm.allreduce_sum_(addvec)                                              # :41
        hip.add_f64_(self.state, addvec)                                                  # :42, :29-32

    @property
    def mean(self):
        """tf.to_float(sum / count), :25."""
        st = self.state.cpu().numpy()
        return (st[:self.n] / st[2 * self.n]).astype(np.float32).reshape(self.shape)

    @property
    def std(self):
        """sqrt(max(to_float(sumsq / count) - mean^2, 1e-2)) in float32, :26."""
        st = self.state.cpu().numpy()
        mean = (st[:self.n] / st[2 * self.n]).astype(np.float32)
        ex2 = (st[self.n:2 * self.n] / st[2 * self.n]).astype(np.float32)
        return np.sqrt(np.maximum(ex2 - np.square(mean), np.float32(1e-2))).reshape(self.shape)
