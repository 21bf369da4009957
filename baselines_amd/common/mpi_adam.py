"""Data-parallel gradient exchange: the role of MpiAdam (baselines/common/mpi_adam.py) with
RCCL over xGMI in place of host MPI.  One process per GPU; the flat float32 gradient arena is
the message (44 KB for MLP(64,64): latency-bound), one contiguous buffer per minibatch.

Transports:
  "rccl" — the library's own communicator (``ppo2hip_comm``: RCCL bound inside libppo2hip, calls queued on the
           compute stream, so that an epoch of gradient -> all-reduce -> Adam steps is one library call).  Needs one
           GPU per rank.  ``torch.distributed`` is only the control plane that hands rank 0's unique id round.
  "host" — ``torch.distributed`` itself; with the gloo backend device tensors are staged through the host (test
           rigs in which several ranks share one GPU, where RCCL cannot run).
  "torch" — ``torch.distributed``'s own RCCL process group on device tensors (one Python call per collective); what
           "rccl" falls back to, on every rank together, when some rank cannot join the library's communicator.
  "auto" — "rccl" when torch.distributed runs on nccl (= RCCL: every rank has its GPU), else "host";
           ``PPO2HIP_DP_TRANSPORT`` overrides the choice.

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
    def __init__(self, mode="mpi", group=None, transport="auto"):
        if mode not in ("mpi", "global"):
            raise ValueError("mode must be 'mpi' or 'global'")
        if transport not in ("auto", "rccl", "host", "torch"):
            raise ValueError("transport must be 'auto', 'rccl', 'host' or 'torch'")
        self.mode = mode
        self.group = group
        self.size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        backend = dist.get_backend(group) if dist.is_initialized() else ""
        self._gloo = backend == "gloo"
        import os
        if transport == "auto":
            transport = os.environ.get("PPO2HIP_DP_TRANSPORT") or ("rccl" if "nccl" in backend else "host")
        self.transport = transport
        self._native = None
        self._pending_checks = []

    @property
    def grad_scale(self):
        """1/size after a SUM all-reduce of per-rank means (mpi_adam.py:27-28); 1 when the
        ranks already scaled their sums by 1/B_global."""
        return 1.0 / self.size if self.mode == "mpi" else 1.0

    # ---- the library's communicator ---------------------------------------------------
    def native(self):
        """The ppo2hip communicator of this process (created on first use, on the current device), or None
        when the transport is torch.distributed."""
        if self.transport != "rccl" or self.size == 1:
            return None
        if self._native is None:
            from .. import hip
            box = [hip.NativeComm.unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                       group=self.group)
            comm, err = None, None
            try:
                comm = hip.NativeComm(box[0], self.rank, self.size)
            except Exception as e:          # every rank must take the same transport: agree before using it
                err = e
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok.item()) == 0:
                # some rank could not join the library's communicator: all ranks carry on over torch.distributed's own
                # RCCL process group (same collectives, one Python call per all-reduce)
                import sys
                print("baselines_amd: library communicator unavailable on rank %d (%s); using torch.distributed"
                      % (self.rank, err), file=sys.stderr)
                self.transport = "torch"
                return None
            self._native = comm
        return self._native

    def _staged(self, t):
        """gloo carries host buffers only: device tensors are staged through the host then
        (test rigs without RCCL, e.g. several ranks sharing one GPU)."""
        return t.is_cuda and self._gloo

    def allreduce_sum_(self, t):
        if self.size > 1:
            nat = self.native() if t.is_cuda and t.dtype in (torch.float32, torch.float64) else None
            if nat is not None:
                nat.allreduce_sum_(t)
            elif self._staged(t):
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def bcast_(self, t, root=0):
        if self.size > 1:
            nat = self.native() if t.is_cuda and t.dtype == torch.float32 else None
            if nat is not None:
                nat.bcast_(t, root)
            elif self._staged(t):
                h = t.cpu()
                dist.broadcast(h, src=root, group=self.group)
                t.copy_(h)
            else:
                dist.broadcast(t, src=root, group=self.group)
        return t

    def check_synced(self, theta, wait=True):
        """MpiAdam.check_synced (mpi_adam.py:42-50): every rank's parameters must equal rank
        0's exactly.  On the library's communicator: MAX all-reduce of two 64-bit digests of the parameter
        bits and their complements (equal iff max == min), queued on the stream; with ``wait=False`` the
        verdict is collected by a later call (``poll_checks``), so that the check never stalls the stream.
        Over torch.distributed: all-reduce of (max, -min) of the parameters themselves."""
        if self.size == 1:
            return True
        nat = self.native() if theta.is_cuda else None
        if nat is not None:
            flag = nat.replica_checksum(theta)
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending_checks.append((ev, host, flag))
            if wait:
                ev.synchronize()
            return self.poll_checks()
        hi = theta.detach().cpu().clone() if self._staged(theta) else theta.clone()
        lo = -hi.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(lo, op=dist.ReduceOp.MAX, group=self.group)
        ok = bool(torch.equal(hi, -lo))
        assert ok, "replicas diverged"
        return ok

    def poll_checks(self, wait=False):
        """Collect the verdicts of queued replica checks that have completed (all of them with ``wait``)."""
        keep = []
        for ev, host, flag in self._pending_checks:
            if wait:
                ev.synchronize()
            if ev.query():
                assert int(host.item()) == 1, "replicas diverged"
            else:
                keep.append((ev, host, flag))
        self._pending_checks = keep
        return True

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

    def local_envs(self, mbenvinds, nenvs_global):
        """Global env numbers of one env-sequence minibatch (baselines/ppo2/ppo2.py:131-133) -> the envs this rank
        owns, in list order, re-based to its own numbering."""
        e0, e1 = self.env_range(nenvs_global)
        mbenvinds = np.asarray(mbenvinds)
        return (mbenvinds[(mbenvinds >= e0) & (mbenvinds < e1)] - e0).astype(np.int32)


class MpiAdam:
    """The reference's optimiser object (mpi_adam.py:6-50) over a device parameter arena:
    ``update(localg, stepsize)`` checks the replicas every 100 calls (:22-23), all-reduces, scales by
    1/size (``scale_grad_by_procs``, :27-28) and applies Adam in the HIP kernel; ``sync`` broadcasts
    rank 0's parameters (:37-40)."""

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
