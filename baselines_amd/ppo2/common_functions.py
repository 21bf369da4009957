"""``Model`` and ``Runner`` of the fork's PPO2 (baselines/ppo2/common_functions.py) on the
MI355X-native kernels, plus ``sf01``, ``constfn``, ``compute_advs`` and ``REWARD_NUM``.

Everything numerical runs in libppo2hip: GAE, flatten, advantage scalarisation and
normalisation, gather, forward/loss/backward, clip + Adam.  This module keeps the
reference's call signatures, moves host arrays to HBM and back, and adds a device-resident
fast path (``Model.train_on_rollout``) that ``learn`` uses to run all epochs of an update
without a host round trip per minibatch.
"""
import time

import os

import numpy as np
import torch

from .. import _lib, hip

REWARD_NUM = 5   # baselines/ppo2/common_functions.py:6


class DeviceRollout:
    """Time-major rollout buffers in HBM ([T, E, ...], env fastest) as the kernels read them."""

    def __init__(self, T, E, obs, actions, rewards, values, neglogpacs, dones, last_values, last_dones,
                 states=None):
        self.T, self.E = T, E
        self.obs, self.actions, self.rewards, self.values = obs, actions, rewards, values
        self.neglogpacs, self.dones, self.last_values, self.last_dones = neglogpacs, dones, last_values, last_dones
        self.states = states
        self.advs_gae = self.returns = self.advs_ori = None
        self.advs = None   # scalar advantage fed to the update: float32 or float64 [T, E]


_SIDE_STREAMS = {}


def _side_streams(device):
    """(upload stream, side compute stream) of a device, created once per process and shared by every model: the runtime
    maps streams onto a handful of hardware queues in creation order, and side streams created anew per model end up
    sharing a queue with the stream that carries the gradient kernels — their copies and small kernels then queue up
    behind the very launches they are meant to run beside."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    return _SIDE_STREAMS[key]


class _IndexUploader:
    """Index lists of successive updates, from the permutation pipelines' page-locked buffers to HBM, off the thread
    that feeds the GPU.

    Several pipelines on ONE generator take the updates' jobs in turn (job k: pipeline and buffer set k mod their
    number).  A scheduler thread begins job k as soon as (i) the caller has asked for it (``request``), (ii) job k - 1
    has drawn all its swap targets — the generator is free, although that job's last swaps still run, so consecutive
    jobs overlap and the period is the serial draw time, not the job time — and (iii) the caller has finished queueing
    the update whose buffers job k reuses.  An uploader thread with a stream of its own copies each list into a staging
    buffer as the pipeline finishes it.  hipMemcpyAsync from page-locked memory now and then holds its calling thread
    for 3-6 ms (the runtime call itself: torch's copy_ and a bare call alike); with the lists generated up to two updates
    ahead that delays a copy nobody is waiting for yet.  (A copy kernel instead of the copy engine avoids the stall but
    costs more than it saves: the gradient kernel's workgroups take every vector register of a CU, so they cannot start
    on a CU while any other kernel's wave is resident there — measured +0.14 ms per update.)
    ``stage[s][e]`` receives epoch e of the job in set s; ``ready`` is the event after its copy, ``free`` the event (set
    by the consumer) after which the staging buffer may be overwritten."""

    def __init__(self, device, pps, n, nepochs):
        import queue
        import threading
        self.device, self.pps, self.nepochs = device, pps, nepochs
        self.stream = _side_streams(device)[0]
        nsets = len(pps)
        self.stage = [[torch.empty(n, dtype=torch.int32, device=device) for _ in range(nepochs)] for _ in range(nsets)]
        self.ready = [[None] * nepochs for _ in range(nsets)]
        self.free = [[None] * nepochs for _ in range(nsets)]
        self.issued = [[threading.Event() for _ in range(nepochs)] for _ in range(nsets)]
        self.last_copy = [None] * nsets          # event behind the last copy out of each set's page-locked buffers
        self.error = None
        self.cv = threading.Condition()
        self.jobs_wanted = 0                     # jobs the caller has asked for
        self.jobs_begun = 0                      # jobs handed to a pipeline
        self.jobs_used = 0                       # updates the caller has finished queueing
        self.quit = False
        self._jobs = queue.Queue()
        self._thread = threading.Thread(target=self._run, name="ppo2hip-index-upload", daemon=True)
        self._sched = threading.Thread(target=self._schedule, name="ppo2hip-index-schedule", daemon=True)
        self._thread.start()
        self._sched.start()

    # ---- caller's side ------------------------------------------------------------------
    def request(self, upto):
        """Jobs 0 .. upto - 1 are to be generated (each as soon as the conditions above allow)."""
        with self.cv:
            if upto > self.jobs_wanted:
                self.jobs_wanted = upto
                self.cv.notify_all()

    def consumed(self, k):
        """Update k has been queued in full: its buffer set may be reused by job k + 2."""
        with self.cv:
            self.jobs_used = max(self.jobs_used, k + 1)
            self.cv.notify_all()

    def wait(self, s, e):
        """(device list, event after its copy) of epoch e of the job in set s; blocks until the copy has been issued."""
        self.issued[s][e].wait()
        if self.error is not None:
            raise self.error
        return self.stage[s][e], self.ready[s][e]

    def close(self, timeout=10.0):
        """Ends both threads (after the jobs already begun) and waits for them, so that neither is inside a HIP call
        when the interpreter or its owner goes away; the staging buffers are released with the object."""
        if self._thread is not None:
            with self.cv:
                self.quit = True
                self.cv.notify_all()
            self._sched.join(timeout)
            self._jobs.put(None)
            self._thread.join(timeout)
            self._thread = self._sched = None

    # ---- threads -------------------------------------------------------------------------
    def _schedule(self):
        torch.cuda.set_device(self.device)
        try:
            while True:
                with self.cv:
                    self.cv.wait_for(lambda: self.quit or (self.jobs_begun < self.jobs_wanted and
                                                           self.jobs_used >= self.jobs_begun - (len(self.pps) - 1)))
                    if self.quit:
                        return
                    k = self.jobs_begun
                s = k % len(self.pps)
                if k >= 1:
                    self.pps[(k - 1) % len(self.pps)].wait_drawn()     # the generator has moved past job k - 1 (no GIL held)
                if self.last_copy[s] is not None:
                    self.last_copy[s].synchronize()                   # copies out of this set (two jobs ago) have run
                for e in range(self.nepochs):
                    self.issued[s][e].clear()
                self.pps[s].begin(self.nepochs, relative=True)
                self._jobs.put((s, self.nepochs))
                with self.cv:
                    self.jobs_begun = k + 1
                    self.cv.notify_all()
        except BaseException as ex:          # hand the failure to the thread that waits
            self._fail(ex)

    def _fail(self, ex):
        self.error = ex
        for row in self.issued:
            for ev in row:
                ev.set()

    def _run(self):
        torch.cuda.set_device(self.device)
        while True:
            job = self._jobs.get()
            if job is None:
                return
            s, nepochs = job
            try:
                for e in range(nepochs):
                    hip.check(_lib.lib.ppo2hip_perm_wait(self.pps[s]._h, e), "perm_wait")      # blocks without the GIL
                    with torch.cuda.stream(self.stream):
                        if self.free[s][e] is not None:
                            self.stream.wait_event(self.free[s][e])
                        self.stage[s][e].copy_(self.pps[s].bufs[e], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                    self.ready[s][e] = ev
                    if e == nepochs - 1:
                        self.last_copy[s] = ev
                    self.issued[s][e].set()
            except BaseException as ex:
                self._fail(ex)


class Model(object):
    """baselines/ppo2/common_functions.py:7-220.

    Build-only keyword arguments (reference-literal defaults):
      upstream_loss : None -> fork loss/optimiser for fork-signature policies, the upstream
                      variant (value clipping, .5 factor, joint loss, clip_by_global_norm, and
                      ``advs = returns - values`` inside ``train``) for upstream-signature ones.
      reward_num    : value-head outputs for fork-signature policies (default REWARD_NUM).
      comm          : a ``DataParallel`` context; gradients are all-reduced before Adam.
    """

    def __init__(self, *, policy, ob_space, ac_space, nbatch_act, nbatch_train, nsteps, ent_coef, vf_coef,
                 max_grad_norm, model_name="model", lr=None, need_summary=False, upstream_loss=None,
                 reward_num=None, comm=None):
        sess = None
        self.signature = getattr(policy, "signature", "fork")
        if self.signature == "fork":
            rn = REWARD_NUM if reward_num is None else reward_num
            act_model = policy(sess, ob_space, ac_space, rn, reuse=False, model_name=model_name)
            train_model = policy(sess, ob_space, ac_space, rn, reuse=True, model_name=model_name)
        else:
            act_model = policy(sess, ob_space, ac_space, nbatch_act, 1, reuse=False)
            train_model = policy(sess, ob_space, ac_space, nbatch_train, nsteps, reuse=True)
        self.upstream_loss = (self.signature != "fork") if upstream_loss is None else bool(upstream_loss)
        self.spec = train_model.spec
        self.arena = train_model.arena
        self.device = self.arena.theta.device
        self.comm = comm
        self.nsteps = nsteps
        self.nbatch_train = nbatch_train
        self.loss_desc = _lib.LossDesc(float(ent_coef), float(vf_coef), float(max_grad_norm or 0.0),
                                       1 if self.upstream_loss else 0)
        self.max_grad_norm = float(max_grad_norm or 0.0) if self.upstream_loss else 0.0
        P = self.spec.nparams
        self.grads = torch.zeros(P, dtype=torch.float32, device=self.device)
        self.adam_m = torch.zeros(P, dtype=torch.float32, device=self.device)
        self.adam_v = torch.zeros(P, dtype=torch.float32, device=self.device)
        self.adam_t = 0
        self._adam_scratch = hip.adam_scratch(P, self.device)
        self._ws = None
        self._ws_rows = 0
        self._stat_ws = None
        self._losses = torch.zeros((1, 8), dtype=torch.float32, device=self.device)
        self._faults = hip.FaultWatch()
        if comm is not None:
            comm.bcast_(self.arena.theta)          # MpiAdam.sync, baselines/common/mpi_adam.py:37-40
            from . import policies as _policies
            _policies.set_sample_rank(comm.rank)   # every rank samples its own action noise

        self.loss_names = ['policy_loss', 'value_loss', 'policy_entropy', 'approxkl', 'clipfrac']
        self.train_model = train_model
        self.act_model = act_model
        self.step = act_model.step
        if hasattr(act_model, "step_max"):
            self.step_max = act_model.step_max
        self.value = act_model.value
        self.initial_state = act_model.initial_state
        self.train = self._train_fork if self.signature == "fork" else self._train_upstream
        if need_summary:
            self.summaries = []
            self.write_summary = lambda tag, value, step: self.summaries.append((tag, float(value), step))
            self.log_histogram = lambda tag, values, step, bins=1000: None

    # ------------------------------------------------------------------ helpers
    def _workspace(self, B):
        if self._ws is None or self._ws_rows < B:
            self._ws = torch.empty(self.spec.grad_workspace_bytes(B), dtype=torch.uint8, device=self.device)
            self._ws_rows = B
        return self._ws

    def _stats_workspace(self, nmb):
        need = int(_lib.lib.ppo2hip_adv_stats_workspace_bytes(nmb))
        if self._stat_ws is None or self._stat_ws.numel() < need:
            self._stat_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._stat_ws

    def _apply(self, lr):
        """All-reduce (data parallel) + clip + Adam on the gradient arena."""
        scale = 1.0
        if self.comm is not None and self.comm.size > 1:
            if self.adam_t % 100 == 0:       # MpiAdam.update, baselines/common/mpi_adam.py:22-23
                self.comm.check_synced(self.arena.theta, wait=False)
            self.comm.allreduce_sum_(self.grads)
            scale = self.comm.grad_scale
        self.adam_t += 1
        hip.adam_step(self.arena.theta, self.grads, self.adam_m, self.adam_v, lr, self.adam_t, self._adam_scratch,
                      max_grad_norm=self.max_grad_norm, grad_scale=scale)

    def _up(self, a, dtype):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(self.device)

    # ------------------------------------------------------------------ Model.train
    def _train_fork(self, lr, cliprange, obs, returns, masks, actions, values, advs, neglogpacs,
                    actions_suggest_list=[], advs_suggest_list=[], states=None):   # noqa: B006 (reference signature)
        """baselines/ppo2/common_functions.py:65-106 (the suggestion lists are dead code there)."""
        return self._train_host(lr, cliprange, obs, returns, masks, actions, values, advs, neglogpacs, states)

    def _train_upstream(self, lr, cliprange, obs, returns, masks, actions, values, neglogpacs, states=None):
        """Upstream arity: ``advs = returns - values`` inside train."""
        returns = np.asarray(returns, np.float32)
        values = np.asarray(values, np.float32)
        advs = (returns - values).reshape(len(returns), -1)[:, 0]
        return self._train_host(lr, cliprange, obs, returns, masks, actions, values, advs, neglogpacs, states)

    def _train_host(self, lr, cliprange, obs, returns, masks, actions, values, advs, neglogpacs, states):
        spec = self.spec
        B = len(obs)
        advs = np.asarray(advs)
        d_adv = self._up(advs, np.float64 if advs.dtype == np.float64 else np.float32)
        d_obs = self._up(np.asarray(obs).reshape((B,) + tuple(spec.ob_shape())),
                         np.float32 if spec.kind == _lib.POLICY_MLP else np.uint8)
        d_act = self._up(actions, np.float32 if spec.gaussian else np.int32)
        d_ret = self._up(np.asarray(returns).reshape(B, spec.nvalue), np.float32)
        d_val = self._up(np.asarray(values).reshape(B, spec.nvalue), np.float32)
        d_nlp = self._up(neglogpacs, np.float32)
        d_masks = d_states = mb_envs = None
        nsteps_mb = 0
        if states is not None:
            d_masks = self._up(np.asarray(masks).astype(np.uint8), np.uint8)
            d_states = self._up(states, np.float32)
            nenv_mb = d_states.shape[0]
            nsteps_mb = B // nenv_mb
            mb_envs = torch.arange(nenv_mb, dtype=torch.int32, device=self.device)
        ro = hip.RolloutRef(spec, d_obs, d_act, d_ret, d_val, d_nlp, d_adv, 0, 0, d_masks, d_states)
        stats = hip.adv_stats(d_adv, None, B, 1, 0, 0, self._stats_workspace(1))
        inv_count = 1.0 / B
        if self.comm is not None and self.comm.size > 1 and self.comm.mode == "global":
            self.comm.allreduce_sum_(stats)
            inv_count = 1.0 / float(stats[0, 2].item())
        hip.ppo_grad(spec, self.loss_desc, self.arena.theta, ro, None, B, stats, cliprange, inv_count, self.grads,
                     self._losses[0], self._workspace(B), mb_envs, nsteps_mb)
        self._apply(lr)
        out = self._losses[0, :5].clone()
        if self.comm is not None and self.comm.size > 1:
            self.comm.allreduce_sum_(out)
            if self.comm.mode != "global":
                out /= self.comm.size
        return [np.float32(x) for x in out.cpu().numpy()]

    # ------------------------------------------------------------------ device fast path
    def seed_shuffle(self, seed):
        """Give the model its own MT19937 stream for minibatch shuffles (same stream as
        ``np.random.seed(seed)``); without it the global ``np.random`` stream is used, as in
        the reference."""
        self.close()                       # lists generated ahead from an earlier stream are dropped with their pipelines
        self._shuffler = hip.LegacyShuffler(seed)
        self._own_stream = True

    INDEX_SETS = 3      # permutation pipelines / buffer sets: index lists are generated up to INDEX_SETS - 1 updates ahead

    def _index_pipeline(self, N, noptepochs):
        key = (N, noptepochs)
        if getattr(self, "_pp_key", None) != key:
            if not hasattr(self, "_shuffler"):
                self._shuffler = hip.LegacyShuffler(0)
                self._own_stream = False
            if getattr(self, "_uploader", None) is not None:
                self._uploader_finalizer()
            # three pipelines on the one generator: consecutive updates' jobs overlap and run up to two updates ahead
            # (see _IndexUploader)
            self._pps = [hip.PermPipeline(self._shuffler, N, noptepochs, nsets=1) for _ in range(self.INDEX_SETS)]
            self._pp_key = key
            self._job = 0                  # number of the job (= update with this shape) the next call consumes
            self._idx_dev = [torch.empty(N, dtype=torch.int32, device=self.device) for _ in range(noptepochs)]
            self._uploader = _IndexUploader(self.device, self._pps, N, noptepochs)
            # the uploader's threads and staging buffers go with the model: the threads reference the uploader only
            import weakref
            self._uploader_finalizer = weakref.finalize(self, self._uploader.close)
            self._idx_free = [None] * noptepochs      # main-stream event: kernels reading _idx_dev[e] done
            self._copy_stream = _side_streams(self.device)[1]
            self._side_stats = [None] * noptepochs    # per epoch: (workspace, out) of the advantage statistics taken on the side stream
        return self._uploader

    def _open_index_job(self, more_updates):
        """This update's job (and, when the model owns its shuffle stream and another update follows, the next one's:
        it is generated while the GPU works through this update) -> (job number, its buffer set)."""
        up, k = self._uploader, self._job
        if not self._own_stream:
            # the shared np.random stream is never drawn from ahead of time: other users of it would see the draws reordered
            self._shuffler.adopt_numpy()
            up.request(k + 1)
        else:
            up.request(k + self.INDEX_SETS if more_updates else k + 1)
        return k, k % self.INDEX_SETS

    def _close_index_job(self, k):
        if not self._own_stream:
            self._pps[k % self.INDEX_SETS].wait_all()
            self._shuffler.release_numpy()
        self._uploader.consumed(k)
        self._job = k + 1

    def train_on_rollout(self, lr, cliprange, ro, noptepochs, nbatch_train, events=None, more_updates=False):
        """All epochs x minibatches of one update from device-resident rollout buffers.

        Index generation follows baselines/ppo2/ppo2.py:110-116 (``inds`` shuffled in place
        every epoch) or :124-136 for recurrent policies (env sequences), drawn from the
        global ``np.random`` stream (or the model's own after ``seed_shuffle``).  The index
        lists are produced by the C++ pipeline while the GPU works on earlier epochs and
        reach HBM through an uploader thread with a stream of its own (each epoch's permutation of
        arange(N) is copied into a staging buffer as soon as it is finished and chained onto the
        previous order by a device gather on the side stream).  Nothing here waits for the GPU: the result is a device
        tensor holding the mean of the per-minibatch loss rows (:139).  ``events``: optional list
        that receives a (start, end) pair of timing events around every gradient evaluation.
        ``more_updates``: another update of the same shape follows; when the model owns its
        shuffle stream (``seed_shuffle``) the next update's index lists are then generated
        ahead, while the GPU works through this one (the shared ``np.random`` stream is never
        drawn from ahead of time: other users of it would see the draws reordered).

        Faults (an index list naming a sample that does not exist, a launch that gave up waiting for its peer
        workgroups) invalidate the update's results.  A read-and-clear of the GPU's fault word is queued behind every
        update; this call raises ``Ppo2HipError`` for every earlier update whose check has completed with a non-zero
        word, and ``check_faults()`` (also run by ``save``) waits for all checks — so a caller that never goes through
        ``learn`` still cannot keep training on, or save, parameters a faulted launch left behind.
        """
        self._faults.poll()
        try:
            return self._train_on_rollout(lr, cliprange, ro, noptepochs, nbatch_train, events, more_updates)
        finally:
            self._faults.post()

    def check_faults(self):
        """Waits for the fault checks queued behind the updates so far; raises ``Ppo2HipError`` if any launch faulted."""
        self._faults.poll(wait=True)

    def _train_on_rollout(self, lr, cliprange, ro, noptepochs, nbatch_train, events=None, more_updates=False):
        spec = self.spec
        T, E = ro.T, ro.E
        N = T * E
        main = torch.cuda.current_stream()
        # Keep at most two updates queued: a host that runs further ahead fills the launch queue, and the
        # launch that then blocks holds the host until the queue has drained, after which the GPU starves.
        inflight = self.__dict__.setdefault("_updates_inflight", [])
        if len(inflight) >= int(os.environ.get("PPO2HIP_UPDATES_INFLIGHT", "2")):
            inflight.pop(0).synchronize()
        rref = hip.RolloutRef(spec, ro.obs, ro.actions, ro.returns, ro.values, ro.neglogpacs, ro.advs, T, E,
                              ro.dones if spec.recurrent else None, ro.states)
        comm = self.comm if (self.comm is not None and self.comm.size > 1) else None
        if comm is not None and comm.mode == "global":
            return self._train_on_rollout_global(lr, cliprange, ro, rref, noptepochs, nbatch_train, more_updates)
        if not spec.recurrent:
            if N % nbatch_train != 0:
                raise AssertionError("nbatch %% nbatch_train != 0")   # baselines/ppo2/ppo2.py:74
            nmb = N // nbatch_train
            self._index_pipeline(N, noptepochs)
            job, pset = self._open_index_job(more_updates)
        else:
            envsperbatch = nbatch_train // T
            if envsperbatch < 1 or E % envsperbatch != 0:
                raise AssertionError("nenvs must be a multiple of the envs per minibatch")
            nmb = E // envsperbatch
            envinds = np.arange(E, dtype=np.int32)
            tgrid = np.arange(T, dtype=np.int32)[None, :]
        total = nmb * noptepochs
        if self._losses.shape[0] < total:
            self._losses = torch.zeros((total, 8), dtype=torch.float32, device=self.device)
        ws = self._workspace(nbatch_train)
        stat_ws = self._stats_workspace(nmb)
        row = 0
        # The advantage statistics of epochs 1.. need the rollout's advantages and that epoch's index list only, so they
        # are taken on the side stream right behind the list, while the main stream still works through the epoch
        # before (its kernels leave room on every CU); epoch 0's have nothing to hide behind and stay on the main stream.
        side_stats = not spec.recurrent and os.environ.get("PPO2HIP_STATS_ON_MAIN") is None
        if side_stats:
            advs_ready = torch.cuda.Event()
            advs_ready.record(main)
        for e in range(noptepochs):
            stats = None
            if not spec.recurrent:
                # this epoch's own permutation of arange(N), uploaded by the side thread (normally long since)
                sig, sig_ready = self._uploader.wait(pset, e)
                d_idx = sig if e == 0 else self._idx_dev[e]
                with torch.cuda.stream(self._copy_stream):
                    self._copy_stream.wait_event(sig_ready)
                    if self._idx_free[e] is not None:
                        self._copy_stream.wait_event(self._idx_free[e])      # also frees this epoch's statistics buffers
                    if e > 0:       # the running order is a device gather through the order of the epoch before
                        hip.compose(d_prev, sig, d_idx)
                        if side_stats:
                            if self._side_stats[e] is None or self._side_stats[e][1].shape[0] != nmb:
                                self._side_stats[e] = (torch.empty(int(_lib.lib.ppo2hip_adv_stats_workspace_bytes(nmb)), dtype=torch.uint8,
                                                                   device=self.device),
                                                       torch.empty((nmb, 3), dtype=torch.float64, device=self.device))
                            if e == 1:
                                self._copy_stream.wait_event(advs_ready)
                            stats = hip.adv_stats(ro.advs, d_idx, nbatch_train, nmb, T, E, self._side_stats[e][0],
                                                  out=self._side_stats[e][1])
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                d_prev = d_idx
                main.wait_event(ev)
                d_envs = None
            else:
                np.random.shuffle(envinds)                       # :130
                flat = (envinds[:, None].astype(np.int64) * T + tgrid).astype(np.int32).ravel()   # :134
                d_idx = torch.from_numpy(flat).to(self.device)
                d_envs = torch.from_numpy(envinds.copy()).to(self.device)
            if stats is None:
                stats = hip.adv_stats(ro.advs, d_idx, nbatch_train, nmb, T, E, stat_ws)
            native = comm.native() if comm is not None else None
            if (comm is None or native is not None) and not spec.recurrent and events is None:
                # the whole epoch (gradient, all-reduce over the ranks, Adam per minibatch) is one library call
                if comm is not None and -(-self.adam_t // 100) * 100 < self.adam_t + nmb:
                    comm.check_synced(self.arena.theta, wait=False)      # an Adam step count of this epoch is a multiple of 100
                hip.ppo_epoch(spec, self.loss_desc, self.arena.theta, self.grads, self.adam_m, self.adam_v, rref, d_idx,
                              nbatch_train, stats, cliprange, lr, self.adam_t + 1, self._losses[row:row + nmb], ws,
                              self._adam_scratch, max_grad_norm=self.max_grad_norm, comm=native,
                              grad_scale=comm.grad_scale if comm is not None else 1.0)
                self.adam_t += nmb
                row += nmb
                done = torch.cuda.Event()
                done.record(main)
                self._idx_free[e] = done
                self._uploader.free[pset][e] = done
                continue
            for j in range(nmb):
                lo = j * nbatch_train
                mb_envs = d_envs[j * envsperbatch:(j + 1) * envsperbatch] if spec.recurrent else None
                if events is not None:
                    ev0 = torch.cuda.Event(enable_timing=True)
                    ev0.record(main)
                hip.ppo_grad(spec, self.loss_desc, self.arena.theta, rref, d_idx[lo:lo + nbatch_train], nbatch_train,
                             stats[j], cliprange, 1.0 / nbatch_train, self.grads, self._losses[row], ws, mb_envs,
                             T if spec.recurrent else 0)
                if events is not None:
                    ev1 = torch.cuda.Event(enable_timing=True)
                    ev1.record(main)
                    events.append((ev0, ev1))
                self._apply(lr)
                row += 1
            if not spec.recurrent:
                done = torch.cuda.Event()
                done.record(main)
                self._idx_free[e] = done
                self._uploader.free[pset][e] = done
        if not spec.recurrent:
            self._close_index_job(job)
        # mean of the loss rows (baselines/ppo2/ppo2.py:139); over the ranks too: 1/size folded in before the SUM
        out = hip.row_mean(self._losses, total, 5, 1.0 / (total * (comm.size if comm is not None else 1)))
        if comm is not None:
            comm.allreduce_sum_(out)
            comm.poll_checks()
        queued = torch.cuda.Event()
        queued.record(main)
        inflight.append(queued)
        return out

    def _train_on_rollout_global(self, lr, cliprange, ro, rref, noptepochs, nbatch_train, more_updates=False):
        """Data-parallel update that reproduces the single-process one: every rank draws the same GLOBAL minibatch
        order (baselines/ppo2/ppo2.py:110-116 over all ranks' envs; ``nbatch_train`` = global minibatch size) with the
        same index pipeline as the single-process path, and works on the entries whose env it owns.

        Per epoch, on the side stream behind the index list: the rank's {sum a, sum a^2, n} per minibatch straight off
        the global list, the ownership filter (compact + count) and a copy of the counts to page-locked memory.  On the
        main stream: ONE all-reduce of the [nmb, 3] statistics per epoch, then per minibatch gradient sums scaled by
        1/B_global, the SUM all-reduce of the gradient arena and Adam — one library call per epoch on the library's
        communicator.  The host waits for the side stream's counts only (queued ahead of the main stream's work), never
        for the main stream."""
        spec, comm = self.spec, self.comm
        if spec.recurrent:
            return self._train_on_rollout_global_recurrent(lr, cliprange, ro, rref, noptepochs, nbatch_train)
        T, E, G = ro.T, ro.E, comm.size
        Nl = T * E
        Ng = Nl * G
        if Ng >= 2 ** 31:
            raise ValueError("global sample numbers are int32: %d samples over all ranks is too many" % Ng)
        if Ng % nbatch_train != 0:
            raise AssertionError("nbatch %% nbatch_train != 0")   # baselines/ppo2/ppo2.py:74
        nmb = Ng // nbatch_train
        lo, hi = comm.rank * Nl, (comm.rank + 1) * Nl
        main = torch.cuda.current_stream()
        self._index_pipeline(Ng, noptepochs)
        job, pset = self._open_index_job(more_updates)
        total = nmb * noptepochs
        if self._losses.shape[0] < total:
            self._losses = torch.zeros((total, 8), dtype=torch.float32, device=self.device)
        key = (Ng, nbatch_train, noptepochs)
        if getattr(self, "_share_key", None) != key:
            dv = self.device
            self._share = [dict(local=torch.empty(Ng, dtype=torch.int32, device=dv),
                                counts=torch.empty(nmb, dtype=torch.int32, device=dv),
                                counts_host=torch.empty(nmb, dtype=torch.int32, pin_memory=True),
                                stats=torch.empty((nmb, 3), dtype=torch.float64, device=dv),
                                stat_ws=torch.empty(int(_lib.lib.ppo2hip_adv_stats_owned_workspace_bytes(nmb)), dtype=torch.uint8, device=dv),
                                filter_ws=torch.empty(int(_lib.lib.ppo2hip_filter_owned_workspace_bytes(nbatch_train, nmb)),
                                                      dtype=torch.uint8, device=dv))
                           for _ in range(noptepochs)]
            self._share_key = key
        native = comm.native()
        advs_ready = torch.cuda.Event()
        advs_ready.record(main)
        row = 0
        for e in range(noptepochs):
            sig, sig_ready = self._uploader.wait(pset, e)
            d_idx = sig if e == 0 else self._idx_dev[e]
            b = self._share[e]
            with torch.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(sig_ready)
                if self._idx_free[e] is not None:
                    self._copy_stream.wait_event(self._idx_free[e])      # also frees this epoch's share buffers
                if e == 0:
                    self._copy_stream.wait_event(advs_ready)
                else:           # the running order is a device gather through the order of the epoch before
                    hip.compose(d_prev, sig, d_idx)
                hip.adv_stats_owned(ro.advs, d_idx, nbatch_train, nmb, lo, hi, T, E, b["stat_ws"], out=b["stats"])
                hip.filter_owned(d_idx, nbatch_train, nmb, lo, hi, b["local"], nbatch_train, b["counts"], b["filter_ws"])
                b["counts_host"].copy_(b["counts"], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
            d_prev = d_idx
            main.wait_event(ev)
            comm.allreduce_sum_(b["stats"])          # rows become {sum a, sum a^2, B_global}: one collective per epoch
            ev.synchronize()                         # the sizes of this rank's shares (side stream only)
            counts = [int(c) for c in b["counts_host"].tolist()]
            ws = self._workspace(max(max(counts), 1))
            if native is not None:
                if -(-self.adam_t // 100) * 100 < self.adam_t + nmb:
                    comm.check_synced(self.arena.theta, wait=False)      # an Adam step count of this epoch is a multiple of 100
                hip.ppo_epoch_shares(spec, self.loss_desc, self.arena.theta, self.grads, self.adam_m, self.adam_v, rref,
                                     b["local"], nbatch_train, counts, b["stats"], cliprange, 1.0 / nbatch_train, lr,
                                     self.adam_t + 1, self._losses[row:row + nmb], ws, self._adam_scratch,
                                     max_grad_norm=self.max_grad_norm, comm=native)
                self.adam_t += nmb
            else:       # torch.distributed carries the gradients: one minibatch per round of Python calls
                for j in range(nmb):
                    c = counts[j]
                    if c:
                        hip.ppo_grad(spec, self.loss_desc, self.arena.theta, rref, b["local"][j * nbatch_train:j * nbatch_train + c],
                                     c, b["stats"][j], cliprange, 1.0 / nbatch_train, self.grads, self._losses[row + j], ws)
                    else:
                        self.grads.zero_()
                        self._losses[row + j].zero_()
                    self._apply(lr)
            row += nmb
            done = torch.cuda.Event()
            done.record(main)
            self._idx_free[e] = done
            self._uploader.free[pset][e] = done
        self._close_index_job(job)
        out = hip.row_mean(self._losses, total, 5)
        comm.allreduce_sum_(out)       # per-rank rows are shares of the global means
        comm.poll_checks()
        queued = torch.cuda.Event()
        queued.record(main)
        self.__dict__.setdefault("_updates_inflight", []).append(queued)
        return out

    def _train_on_rollout_global_recurrent(self, lr, cliprange, ro, rref, noptepochs, nbatch_train):
        """Env-sequence minibatches (baselines/ppo2/ppo2.py:124-136) over env-sharded ranks with one global env order:
        every rank shuffles ``arange(E_global)`` from the same ``np.random`` state and keeps the envs it owns; the
        whole update's {sum a, sum a^2, n} rows are all-reduced in one collective, gradient sums (scaled by
        1 / global minibatch size) once per minibatch."""
        spec, comm = self.spec, self.comm
        T, E, G = ro.T, ro.E, comm.size
        Eg = E * G
        envsperbatch = nbatch_train // T
        if envsperbatch < 1 or Eg % envsperbatch != 0:
            raise AssertionError("nenvs must be a multiple of the envs per minibatch")
        nmb = Eg // envsperbatch
        total = nmb * noptepochs
        if self._losses.shape[0] < total:
            self._losses = torch.zeros((total, 8), dtype=torch.float32, device=self.device)
        envinds = np.arange(Eg, dtype=np.int32)
        lists = []
        for _ in range(noptepochs):
            np.random.shuffle(envinds)                       # :130, over all ranks' envs
            for start in range(0, Eg, envsperbatch):
                lists.append(comm.local_envs(envinds[start:start + envsperbatch], Eg))
        env_cat = np.concatenate(lists)
        flat_cat = (env_cat[:, None].astype(np.int64) * T + np.arange(T, dtype=np.int64)[None, :]).astype(np.int32).ravel()   # :134
        d_env = torch.from_numpy(env_cat).to(self.device)
        d_flat = torch.from_numpy(flat_cat).to(self.device)
        stats = torch.zeros((total, 3), dtype=torch.float64, device=self.device)
        stat_ws = self._stats_workspace(1)
        off = 0
        for k, l in enumerate(lists):
            if l.size:
                hip.adv_stats(ro.advs, d_flat[off * T:(off + l.size) * T], l.size * T, 1, T, E, stat_ws, out=stats[k:k + 1])
            off += l.size
        comm.allreduce_sum_(stats)
        ws = self._workspace(max(max(l.size for l in lists), 1) * T)
        off = 0
        for k, l in enumerate(lists):
            n = int(l.size)
            if n:
                hip.ppo_grad(spec, self.loss_desc, self.arena.theta, rref, d_flat[off * T:(off + n) * T], n * T, stats[k], cliprange,
                             1.0 / nbatch_train, self.grads, self._losses[k], ws, d_env[off:off + n], T)
            else:
                self.grads.zero_()
                self._losses[k].zero_()
            self._apply(lr)
            off += n
        out = hip.row_mean(self._losses, total, 5)
        comm.allreduce_sum_(out)
        comm.poll_checks()
        return out

    def close(self):
        """Ends the index-upload thread of the device path and frees its staging buffers (also runs when the model is
        garbage-collected or the interpreter exits)."""
        if getattr(self, "_uploader", None) is not None:
            self._uploader_finalizer()
            self._uploader = None
            self._pps = None
            self._pp_key = None

    # ------------------------------------------------------------------ misc surface
    def get_values(self, obs):
        return self.train_model.value(obs)

    def get_actions_dist(self, obs):
        """[mode actions, std] (baselines/ppo2/common_functions.py:112-113)."""
        return [self.train_model.sample(obs), self.train_model.std]

    def save(self, save_path, save_path_base=None):
        """joblib list of arrays in variable creation order, written twice
        (baselines/ppo2/common_functions.py:115-119)."""
        import joblib
        self.check_faults()           # never write out parameters a faulted launch left behind
        ps = self.arena.get_list()
        joblib.dump(ps, save_path)
        if save_path_base is not None:
            joblib.dump(ps, save_path_base)

    def load(self, load_path, load_value=True):
        """baselines/ppo2/common_functions.py:121-142: skip variables whose name contains 'vf'
        when ``load_value`` is False."""
        import joblib
        loaded = joblib.load(load_path)
        self.arena.set_list(loaded, select=None if load_value else (lambda name: name.find('vf') == -1))

    def load_value(self, load_path):
        """baselines/ppo2/common_functions.py:145-157: only the variables whose name contains 'vf'."""
        import joblib
        loaded = joblib.load(load_path)
        self.arena.set_list(loaded, select=lambda name: name.find('vf') != -1)


class Runner(object):
    """baselines/ppo2/common_functions.py:223-345.

    The rollout loop and its bookkeeping are host code as in the reference; the collected
    time-major arrays go to HBM once, GAE and the sf01 flatten run there.  The device
    buffers of the last call stay available as ``self.rollout`` for ``Model.train_on_rollout``.
    """

    def __init__(self, *, env, model, nsteps, gamma, lam):
        self.env = env
        self.model = model
        nenv = env.num_envs
        self.nenv = nenv
        self.device_env = bool(getattr(env, "device_resident", False))
        if self.device_env:
            self._init_device_env(env, model)
        else:
            self.obs = np.zeros((nenv,) + env.observation_space.shape, dtype=model.train_model.X.dtype.name)
            self.obs[:] = env.reset()
        self.gamma = np.array(gamma)
        self.lam = lam
        self.nsteps = nsteps
        self.states = model.initial_state
        self.dones = [False for _ in range(nenv)]
        self.fork = getattr(model, "signature", "fork") == "fork"
        self.rollout = None
        self.verbose = False
        self._arena = None

    def _init_device_env(self, env, model):
        """Envs that simulate in HBM (``env.device_resident``) keep observations, done flags and recurrent
        states as device tensors between steps."""
        self.obs_dev = env.reset().contiguous()
        self.dones_dev = torch.zeros(self.nenv, dtype=torch.uint8, device=model.device)
        self.states_dev = (torch.from_numpy(np.ascontiguousarray(model.initial_state, np.float32)).to(model.device)
                           if model.initial_state is not None else None)

    def run_dev(self, use_deterministic=False):
        """The rollout loop of ``run`` (baselines/ppo2/common_functions.py:247-304) with nothing leaving
        the device: each step's forward, sampling and neglogp write into row t of the time-major arena,
        the env steps on device tensors, the fork's done-bootstrap (:265-276) and GAE (:308-328) run in
        their kernels.  Returns ``run``'s 10-tuple with None in place of the seven flattened host arrays;
        the collected rollout is ``self.rollout``."""
        model, spec, T, E = self.model, self.model.spec, self.nsteps, self.nenv
        pol = model.act_model
        dev, R = model.device, spec.nvalue
        if self._arena is None:
            f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            self._arena = dict(
                obs=torch.empty((T, E) + tuple(spec.ob_shape()), dtype=spec.ob_dtype(), device=dev),
                actions=f32(T, E, spec.ac_dim) if spec.gaussian else torch.empty((T, E), dtype=torch.int32, device=dev),
                rewards=f32(T, E, R), values=f32(T, E, R), nlp=f32(T, E),
                dones=torch.empty((T, E), dtype=torch.uint8, device=dev))
        a = self._arena
        mb_states = self.states_dev.clone() if self.states_dev is not None else None
        epinfos = []
        for t in range(T):
            a["obs"][t].copy_(self.obs_dev)
            a["dones"][t].copy_(self.dones_dev)
            actions, _, snew, _ = pol.step_dev(self.obs_dev, self.states_dev, self.dones_dev, use_deterministic,
                                               actions=a["actions"][t], neglogp=a["nlp"][t], values=a["values"][t])
            if snew is not None:
                self.states_dev = snew
            obs, rewards, dones, infos = self.env.step(actions)
            self.obs_dev = obs.contiguous()
            self.dones_dev = dones.view(torch.uint8) if dones.dtype == torch.bool else dones
            rewards = rewards.to(torch.float32)
            if self.fork:
                rewards = rewards[:, :-1].contiguous()                                  # :262
                # V(reset observation) for every env; the kernel applies it where dones is set (:265-276)
                hip.done_bootstrap(rewards, self.dones_dev, pol.value_dev(self.obs_dev), self.gamma)
            a["rewards"][t].copy_(rewards.reshape(E, R))
            for info in infos:
                maybeepinfo = info.get('episode')
                if maybeepinfo:
                    epinfos.append(maybeepinfo)
        last_values = pol.value_dev(self.obs_dev, self.states_dev, self.dones_dev).reshape(E, R)   # :304
        gamma = np.broadcast_to(self.gamma.astype(np.float64), (R,)) if self.gamma.ndim else float(self.gamma)
        advs, returns, advs_ori = hip.gae(a["rewards"], a["values"], a["dones"], last_values, self.dones_dev, gamma, self.lam)
        ro = DeviceRollout(T, E, a["obs"], a["actions"], a["rewards"], a["values"], a["nlp"], a["dones"], last_values,
                           self.dones_dev, states=mb_states)
        ro.advs_gae, ro.returns, ro.advs_ori = advs, returns, advs_ori
        self.rollout = ro
        states_host = mb_states.cpu().numpy() if mb_states is not None else None
        return (None, None, None, None, None, None, None, states_host, epinfos, False)

    def run(self, is_test=False, use_deterministic=False):
        if self.device_env:
            return self.run_dev(use_deterministic)
        mb_obs, mb_rewards, mb_actions, mb_values, mb_dones, mb_neglogpacs = [], [], [], [], [], []
        mb_states = self.states
        epinfos = []
        t_s = time.time()
        ret = False
        model, R = self.model, self.model.spec.nvalue
        for t in range(self.nsteps):
            if use_deterministic:
                actions, values, self.states, neglogpacs = model.step_max(self.obs, self.states, self.dones)
            else:
                actions, values, self.states, neglogpacs = model.step(self.obs, self.states, self.dones)
            mb_obs.append(self.obs.copy())
            mb_actions.append(actions)
            mb_values.append(values)
            mb_neglogpacs.append(neglogpacs)
            mb_dones.append(self.dones)
            self.obs[:], rewards, self.dones, infos = self.env.step(actions)
            if self.fork:
                rewards = np.asarray(rewards)[:, :-1]                           # :262
                # done-bootstrap off the reset observation (:265-276)
                v_preds = None
                for i in range(len(self.dones)):
                    if self.dones[i]:
                        if v_preds is None:
                            v_preds = model.value(self.obs)
                        if rewards[i][0] > 0:
                            rewards[i] += self.gamma * v_preds[i]
                        else:
                            rewards[i][2:] += self.gamma[2:] * v_preds[i][2:]
            if self.verbose and t % 100 == 0:
                print(t / self.nsteps, time.time() - t_s)
            for info in infos:
                maybeepinfo = info.get('episode')
                if maybeepinfo:
                    epinfos.append(maybeepinfo)
            mb_rewards.append(rewards)
            if is_test:
                if any(self.dones):
                    ret = True
                    break
        T, E = len(mb_values), self.nenv
        dev = model.device
        spec = model.spec
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        ob_dt = np.float32 if spec.kind == _lib.POLICY_MLP else np.uint8
        d_obs = up(np.asarray(mb_obs).reshape((T, E) + tuple(spec.ob_shape())), ob_dt)
        d_rew = up(np.asarray(mb_rewards, dtype=np.float32).reshape(T, E, R), np.float32)
        d_act = up(np.asarray(mb_actions), np.float32 if spec.gaussian else np.int32)
        d_val = up(np.asarray(mb_values, dtype=np.float32).reshape(T, E, R), np.float32)
        d_nlp = up(np.asarray(mb_neglogpacs, dtype=np.float32), np.float32)
        d_dones = up(np.asarray(mb_dones, dtype=np.bool_).astype(np.uint8), np.uint8)
        last_values = model.value(self.obs, self.states, self.dones)            # :304
        d_lastv = up(np.asarray(last_values, np.float32).reshape(E, R), np.float32)
        d_lastd = up(np.asarray(self.dones, dtype=np.bool_).astype(np.uint8), np.uint8)
        gamma = np.broadcast_to(self.gamma.astype(np.float64), (R,)) if self.gamma.ndim else float(self.gamma)
        advs, returns, advs_ori = hip.gae(d_rew, d_val, d_dones, d_lastv, d_lastd, gamma, self.lam)   # :308-328
        ro = DeviceRollout(T, E, d_obs, d_act, d_rew, d_val, d_nlp, d_dones, d_lastv, d_lastd,
                           states=(up(mb_states, np.float32) if mb_states is not None else None))
        ro.advs_gae, ro.returns, ro.advs_ori = advs, returns, advs_ori
        self.rollout = ro
        squeeze = (lambda a: a) if self.fork else (lambda a: a.reshape(a.shape[0]))
        flat = lambda d: hip.sf01(d).cpu().numpy()                              # :344, :347-352
        out = (flat(d_obs), squeeze(flat(returns)), flat(d_dones).astype(np.bool_), flat(d_act),
               squeeze(flat(d_val)), squeeze(flat(d_rew)), flat(d_nlp))
        return (*out, mb_states, epinfos, ret)


def sf01(arr):
    """swap and then flatten axes 0 and 1 (baselines/ppo2/common_functions.py:347-352);
    NumPy arrays go through HBM and the HIP transpose, device tensors stay on the device."""
    if isinstance(arr, torch.Tensor):
        return hip.sf01(arr.contiguous())
    a = np.ascontiguousarray(arr)
    out = hip.sf01(torch.from_numpy(a.view(np.uint8) if a.dtype == np.bool_ else a).cuda()).cpu().numpy()
    return out.view(np.bool_) if a.dtype == np.bool_ else out


def constfn(val):
    def f(_):
        return val
    return f


def compute_advs(max_return, values, advs_ori, action_diff, action_std):
    """Advantage scalarisation of the fork (baselines/ppo2/common_functions.py:359-401) in
    the HIP kernel; returns float64 [N].  The ``action_diff`` override (:396-398) is dead in
    the reference's driver (it passes ``[]``) and is not supported."""
    if len(action_diff) != 0:
        raise NotImplementedError("compute_advs: action_diff override is not on the PPO2 path")
    if isinstance(values, torch.Tensor):
        return hip.compute_advs(max_return, values, advs_ori)
    v = torch.from_numpy(np.ascontiguousarray(values, np.float32)).cuda()
    a = torch.from_numpy(np.ascontiguousarray(advs_ori, np.float32)).cuda()
    return hip.compute_advs(max_return, v, a).cpu().numpy()
