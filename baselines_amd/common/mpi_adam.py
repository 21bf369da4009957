"""Data-parallel gradient exchange: the role of MpiAdam (baselines/common/mpi_adam.py) with
RCCL over xGMI in place of host MPI.  One process per GPU; ``torch.distributed`` (backend
"nccl" = RCCL on ROCm, "gloo" on CPU) carries the flat float32 gradient arena, so the
message is one contiguous buffer per minibatch (44 KB for MLP(64,64): latency-bound).

Two sharding modes (envs shard by index in both, so GAE needs no communication):
  "mpi"    — the reference's semantics: every rank shuffles and normalises its own shard,
             gradients are SUM-all-reduced and divided by the number of ranks
             (mpi_adam.py:24-28).
  "global" — every rank draws the SAME global permutation, keeps the rows whose env it owns,
             and the ranks all-reduce {sum a, sum a^2, n} and un-normalised gradient sums, so
             N ranks reproduce the single-process update up to summation order.
"""
import numpy as np
import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, mode="mpi", group=None):
        if mode not in ("mpi", "global"):
            raise ValueError("mode must be 'mpi' or 'global'")
        self.mode = mode
        self.group = group
        self.size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._gloo = dist.is_initialized() and dist.get_backend(group) == "gloo"

    @property
    def grad_scale(self):
        """1/size after a SUM all-reduce of per-rank means (mpi_adam.py:27-28); 1 when the
        ranks already scaled their sums by 1/B_global."""
        return 1.0 / self.size if self.mode == "mpi" else 1.0

    def _staged(self, t):
        """gloo carries host buffers only: device tensors are staged through the host then
        (test rigs without RCCL, e.g. several ranks sharing one GPU)."""
        return t.is_cuda and self._gloo

    def allreduce_sum_(self, t):
        if self.size > 1:
            if self._staged(t):
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def bcast_(self, t, root=0):
        if self.size > 1:
            if self._staged(t):
                h = t.cpu()
                dist.broadcast(h, src=root, group=self.group)
                t.copy_(h)
            else:
                dist.broadcast(t, src=root, group=self.group)
        return t

    def check_synced(self, theta):
        """MpiAdam.check_synced (mpi_adam.py:42-50): every rank's parameters must equal rank
        0's exactly.  Implemented as an all-reduce of (max, -min): equal iff max == min."""
        if self.size == 1:
            return True
        hi = theta.detach().cpu().clone() if self._staged(theta) else theta.clone()
        lo = -hi.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(lo, op=dist.ReduceOp.MAX, group=self.group)
        ok = bool(torch.equal(hi, -lo))
        assert ok, "replicas diverged"
        return ok

    # ---- env sharding and ownership-filtered minibatches ("global" mode) -------------
    def env_range(self, nenvs_global):
        """Contiguous block of env indices owned by this rank."""
        per = nenvs_global // self.size
        if per * self.size != nenvs_global:
            raise ValueError("nenvs must divide evenly over the ranks")
        return self.rank * per, (self.rank + 1) * per

    def local_minibatch(self, mbinds, nsteps, nenvs_global):
        """Global env-major flat sample indices of one minibatch -> the ones this rank owns,
        re-based to its local [E_local * nsteps) index space."""
        e0, e1 = self.env_range(nenvs_global)
        env = mbinds // nsteps
        keep = (env >= e0) & (env < e1)
        return (mbinds[keep] - e0 * nsteps).astype(np.int32)


class MpiAdam:
    """The reference's optimiser object (mpi_adam.py:6-50) over a device parameter arena:
    ``update(localg, stepsize)`` all-reduces, scales and applies Adam in the HIP kernel."""

    def __init__(self, theta, *, beta1=0.9, beta2=0.999, epsilon=1e-08, scale_grad_by_procs=True, comm=None):
        from .. import hip
        self._hip = hip
        self.theta = theta
        self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon
        self.scale_grad_by_procs = scale_grad_by_procs
        self.m = torch.zeros_like(theta)
        self.v = torch.zeros_like(theta)
        self.t = 0
        self.comm = comm or DataParallel("mpi")
        self._scratch = hip.adam_scratch(theta.numel(), theta.device)

    def update(self, localg, stepsize):
        if self.t % 100 == 0:
            self.check_synced()
        g = localg.to(torch.float32).clone()
        self.comm.allreduce_sum_(g)
        scale = 1.0 / self.comm.size if self.scale_grad_by_procs else 1.0
        self.t += 1
        self._hip.adam_step(self.theta, g, self.m, self.v, stepsize, self.t, self._scratch, grad_scale=scale,
                            beta1=self.beta1, beta2=self.beta2, epsilon=self.epsilon)

    def sync(self):
        self.comm.bcast_(self.theta)

    def check_synced(self):
        self.comm.check_synced(self.theta)
