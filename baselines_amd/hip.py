"""Tensor-level wrappers over the C ABI: shape/dtype checks on the host, then one
ppo2hip_* call on the current HIP stream.  torch tensors are only containers for
device memory here; no torch operator computes anything on the path.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a device tensor" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _as_u8(t):
    return t.view(torch.uint8) if t.dtype == torch.bool else t


# --------------------------------------------------------------------------------------
# scans
# --------------------------------------------------------------------------------------

def gae(rewards, values, dones, last_values, last_dones, gamma, lam, want_advs_ori=True, want_advs=True):
    """GAE over a time-major rollout ([T,E,R] float32); returns (advs, returns, advs_ori).  ``want_advs=False`` skips the
    scan's own advantage array (None in its place): Runner.run hands on returns only
    (baselines/ppo2/common_functions.py:344) and the driver recomputes ``returns - values`` (advs_ori)."""
    rewards = _dev(rewards, torch.float32, "rewards")
    values = _dev(values, torch.float32, "values")
    if rewards.dim() != 3 or values.shape != rewards.shape:
        raise ValueError("rewards/values must both be [T, E, R]")
    T, E, R = rewards.shape
    dones = _dev(_as_u8(dones), torch.uint8, "dones")
    last_dones = _dev(_as_u8(last_dones), torch.uint8, "last_dones")
    last_values = _dev(last_values, torch.float32, "last_values")
    if tuple(dones.shape) != (T, E) or last_dones.numel() != E or last_values.numel() != E * R:
        raise ValueError("dones [T,E], last_dones [E], last_values [E,R] expected")
    g = np.ascontiguousarray(np.broadcast_to(np.asarray(gamma, np.float64), (R,)))
    advs = torch.empty_like(rewards) if want_advs else None
    rets = torch.empty_like(rewards)
    ori = torch.empty_like(rewards) if want_advs_ori else None
    check(lib.ppo2hip_gae_f32(_ptr(rewards), _ptr(values), _ptr(dones), _ptr(last_values), _ptr(last_dones),
                              g.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), float(lam), T, E, R,
                              _ptr(advs), _ptr(rets), _ptr(ori), _stream()), "gae")
    return advs, rets, ori


def discount_with_dones(rewards, dones, last_values, gamma):
    """A2C n-step returns over time-major [T,E]; last_values None = no bootstrap."""
    rewards = _dev(rewards, torch.float32, "rewards")
    if rewards.dim() != 2:
        raise ValueError("rewards must be [T, E]")
    T, E = rewards.shape
    dones = _dev(_as_u8(dones), torch.uint8, "dones")
    if tuple(dones.shape) != (T, E):
        raise ValueError("dones must be [T, E]")
    if last_values is not None:
        last_values = _dev(last_values, torch.float32, "last_values")
        if last_values.numel() != E:
            raise ValueError("last_values must be [E]")
    out = torch.empty_like(rewards)
    check(lib.ppo2hip_discount_with_dones_f32(_ptr(rewards), _ptr(dones), _ptr(last_values), float(gamma), T, E,
                                              _ptr(out), _stream()), "discount_with_dones")
    return out


def discount(x, gamma, starts=None, out_dtype=None):
    """Discounted sums along axis 0 of a device tensor [T, ...] (float32 or float64).  Without
    ``starts`` the output is float64 (what scipy's lfilter returns); with episode-start flags
    [T] it is ``x``'s dtype unless ``out_dtype`` says otherwise."""
    if not isinstance(x, torch.Tensor) or not x.is_cuda or not x.is_contiguous() or x.dim() < 1:
        raise ValueError("x must be a contiguous device tensor with at least one dim")
    if x.dtype not in (torch.float32, torch.float64):
        raise TypeError("x must be float32 or float64")
    T = x.shape[0]
    C = x.numel() // T if T else 0
    if out_dtype is None:
        out_dtype = torch.float64 if starts is None else x.dtype
    if out_dtype not in (torch.float32, torch.float64) or (x.dtype == torch.float64 and out_dtype != torch.float64):
        raise TypeError("out_dtype must be float64, or float32 for float32 input")
    if starts is not None:
        starts = _dev(_as_u8(starts), torch.uint8, "starts")
        if starts.numel() != T:
            raise ValueError("starts must have one flag per time step")
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    check(lib.ppo2hip_discount(_ptr(x), int(x.dtype == torch.float64), _ptr(starts), float(gamma), T, C, _ptr(out),
                               int(out_dtype == torch.float64), _stream()), "discount")
    return out


def sf01(arr):
    """[T, E, ...] -> [E*T, ...] on device."""
    if not arr.is_cuda or not arr.is_contiguous() or arr.dim() < 2:
        raise ValueError("sf01 needs a contiguous device tensor with at least 2 dims")
    T, E = arr.shape[0], arr.shape[1]
    src = _as_u8(arr)
    row_bytes = src.element_size()
    for d in src.shape[2:]:
        row_bytes *= d
    out = torch.empty((T * E,) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
    check(lib.ppo2hip_sf01(_ptr(src), T, E, row_bytes, _ptr(out), _stream()), "sf01")
    return out.view(arr.dtype) if arr.dtype == torch.bool else out


# --------------------------------------------------------------------------------------
# env-wrapper filters
# --------------------------------------------------------------------------------------

def _float_dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or not t.is_contiguous():
        raise TypeError("%s must be a contiguous device tensor" % name)
    if t.dtype not in (torch.float32, torch.float64):
        raise TypeError("%s must be float32 or float64" % name)
    return t


def rms_state(d, epsilon=1e-4, device="cuda"):
    """Device statistics of one RunningMeanStd: double[3*d] = mean 0, var 1, count epsilon."""
    st = torch.empty(3 * int(d), dtype=torch.float64, device=device)
    st[:d] = 0.0
    st[d:2 * d] = 1.0
    st[2 * d:] = float(epsilon)
    return st


def rms_update(state, x, workspace=None):
    """Merge the batch ``x`` [n, ...] into ``state`` (RunningMeanStd.update)."""
    x = _float_dev(x, "x")
    n = x.shape[0] if x.dim() else 0
    d = state.numel() // 3
    if x.numel() != n * d:
        raise ValueError("x must be [n, %d]" % d)
    need = lib.ppo2hip_rms_workspace_bytes(n, d)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
    check(lib.ppo2hip_rms_update(_ptr(x), int(x.dtype == torch.float64), n, d, _ptr(_dev(state, torch.float64, "state")),
                                 _ptr(workspace), workspace.numel(), _stream()), "rms_update")
    return workspace


def column_moments(x, zeros=None, workspace=None):
    """[x.sum(0) | (x*x).sum(0) | len(x)] in float64 on the device, for x [n, ...]."""
    x = _float_dev(x, "x")
    n = x.shape[0]
    d = x.numel() // n if n else 0
    if n < 1 or d < 1:
        raise ValueError("x must hold at least one row")
    if zeros is None or zeros.numel() < d:
        zeros = torch.zeros(d, dtype=torch.float64, device=x.device)
    need = lib.ppo2hip_rms_workspace_bytes(n, d)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
    out = torch.empty(2 * d + 1, dtype=torch.float64, device=x.device)
    check(lib.ppo2hip_column_moments(_ptr(x), int(x.dtype == torch.float64), n, d, _ptr(zeros), _ptr(out), _ptr(workspace),
                                     workspace.numel(), _stream()), "column_moments")
    return out


def add_f64_(dst, src):
    _dev(dst, torch.float64, "dst")
    _dev(src, torch.float64, "src")
    if dst.numel() != src.numel():
        raise ValueError("dst and src differ in size")
    check(lib.ppo2hip_add_f64(_ptr(dst), _ptr(src), dst.numel(), _stream()), "add_f64")
    return dst


def normalize_clip(x, state, epsilon, clip, center=True, out_dtype=torch.float32):
    """clip((x - mean) / sqrt(var + epsilon), -clip, clip) with the statistics in ``state``."""
    x = _float_dev(x, "x")
    d = state.numel() // 3
    n = x.numel() // d if d else 0
    if x.numel() != n * d:
        raise ValueError("x must be [n, %d]" % d)
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean, var = state[:d], state[d:2 * d]
    check(lib.ppo2hip_normalize_clip(_ptr(x), int(x.dtype == torch.float64), n, d, _ptr(mean) if center else None,
                                     _ptr(var), float(epsilon), float(clip), _ptr(out), int(out_dtype == torch.float64),
                                     _stream()), "normalize_clip")
    return out


def vecnorm_returns(rews, ret, state, gamma, epsilon, clip, out_dtype=torch.float32, workspace=None):
    """ret = ret*gamma + rews; update ``state`` with ret; return clip(rews / sqrt(var + epsilon))."""
    rews = _float_dev(rews, "rews")
    E = rews.numel()
    _dev(ret, torch.float64, "ret")
    if ret.numel() != E:
        raise ValueError("ret must hold one float64 per env")
    if state is None:
        out_dtype = rews.dtype
    need = lib.ppo2hip_rms_workspace_bytes(E, 1)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=rews.device)
    out = torch.empty(rews.shape, dtype=out_dtype, device=rews.device)
    check(lib.ppo2hip_vecnorm_returns(_ptr(rews), int(rews.dtype == torch.float64), E, float(gamma), _ptr(ret), _ptr(state),
                                      float(epsilon), float(clip), _ptr(out), int(out_dtype == torch.float64),
                                      _ptr(workspace), workspace.numel(), _stream()), "vecnorm_returns")
    return out


def frame_stack(stacked, obs, news=None):
    """Roll ``stacked`` [E, ..., nstack*c] in place as np.roll(shift=-1) on the last axis does and
    write ``obs`` [E, ..., c] into the last c places."""
    if not (isinstance(stacked, torch.Tensor) and stacked.is_cuda and stacked.is_contiguous() and obs.is_cuda
            and obs.is_contiguous() and stacked.dtype == obs.dtype and stacked.dim() >= 2 and obs.dim() == stacked.dim()):
        raise TypeError("stacked and obs must be contiguous device tensors of one dtype and rank")
    if stacked.element_size() not in (1, 4):
        raise TypeError("elements of 1 or 4 bytes")
    E, c, sc = stacked.shape[0], obs.shape[-1], stacked.shape[-1]
    if obs.shape[:-1] != stacked.shape[:-1] or c == 0 or sc % c:
        raise ValueError("obs %s does not stack into %s" % (tuple(obs.shape), tuple(stacked.shape)))
    pixels = stacked.numel() // (E * sc) if E else 0
    if news is not None:
        news = _dev(_as_u8(news), torch.uint8, "news")
        if news.numel() != E:
            raise ValueError("news must hold one flag per env")
    check(lib.ppo2hip_frame_stack(_ptr(stacked), _ptr(obs), _ptr(news), E, pixels, c, sc // c, stacked.element_size(),
                                  _stream()), "frame_stack")
    return stacked


# --------------------------------------------------------------------------------------
# advantages
# --------------------------------------------------------------------------------------

def compute_advs(max_return, values, advs_ori):
    """Fork scalarisation: values/advs_ori float32 [N,R] -> float64 [N]."""
    values = _dev(values, torch.float32, "values")
    advs_ori = _dev(advs_ori, torch.float32, "advs_ori")
    if values.dim() != 2 or values.shape != advs_ori.shape:
        raise ValueError("values/advs_ori must both be [N, R]")
    N, R = values.shape
    mr = np.ascontiguousarray(np.asarray(max_return, np.float64))
    if mr.shape != (R,):
        raise ValueError("max_return must have R entries")
    wsb = lib.ppo2hip_compute_advs_workspace_bytes(N, R)
    ws = torch.empty(wsb, dtype=torch.uint8, device=values.device)
    out = torch.empty(N, dtype=torch.float64, device=values.device)
    check(lib.ppo2hip_compute_advs(_ptr(values), _ptr(advs_ori), mr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                   N, R, _ptr(ws), wsb, _ptr(out), _stream()), "compute_advs")
    return out


def adv_stats(advs, idx, B, nmb, T, E, workspace=None, out=None):
    """{sum, sumsq, count} per minibatch (float64 [nmb,3], device)."""
    if advs.dtype not in (torch.float32, torch.float64) or not advs.is_cuda or not advs.is_contiguous():
        raise TypeError("advs must be a contiguous float32/float64 device tensor")
    n_src = advs.numel()
    if idx is not None:
        idx = _dev(idx, torch.int32, "idx")
        if idx.numel() < B * nmb:
            raise ValueError("idx shorter than nmb*B")
    elif n_src < B * nmb:
        raise ValueError("advs shorter than nmb*B")
    if T and T * E != n_src:
        raise ValueError("advs must hold T*E entries")
    wsb = lib.ppo2hip_adv_stats_workspace_bytes(nmb)
    if workspace is None:
        workspace = torch.empty(wsb, dtype=torch.uint8, device=advs.device)
    elif workspace.numel() * workspace.element_size() < wsb:
        raise ValueError("adv_stats workspace too small")
    if out is None:
        out = torch.empty((nmb, 3), dtype=torch.float64, device=advs.device)
    check(lib.ppo2hip_adv_stats(_ptr(advs), int(advs.dtype == torch.float64), _ptr(idx), B, nmb, T, E,
                                _ptr(workspace), _ptr(out), _stream()), "adv_stats")
    return out


# --------------------------------------------------------------------------------------
# permutation (host)
# --------------------------------------------------------------------------------------

class LegacyShuffler:
    """np.random.shuffle-compatible permutation source running in C++.

    ``shuffle(inds)`` permutes a host int32/int64 NumPy array in place exactly as
    ``np.random.shuffle`` would from the same MT19937 state.  ``adopt_numpy()`` /
    ``release_numpy()`` move the global ``np.random`` stream in and out, so that a
    caller mixing both sees one continuous stream, as the reference does.
    """

    def __init__(self, seed=0):
        self._h = lib.ppo2hip_rng_create(int(seed) & 0xFFFFFFFF)
        if not self._h:
            raise MemoryError("ppo2hip_rng_create failed")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:      # module globals may be gone at interpreter exit
            lib.ppo2hip_rng_destroy(h)

    def seed(self, seed):
        lib.ppo2hip_rng_seed(self._h, int(seed) & 0xFFFFFFFF)

    def set_state(self, key, pos):
        key = np.ascontiguousarray(key, np.uint32)
        if key.shape != (624,):
            raise ValueError("MT19937 key must have 624 words")
        check(lib.ppo2hip_rng_set_state(self._h, key.ctypes.data_as(ctypes.c_void_p), int(pos)), "rng_set_state")

    def get_state(self):
        key = np.empty(624, np.uint32)
        pos = ctypes.c_int(0)
        check(lib.ppo2hip_rng_get_state(self._h, key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos)), "rng_get_state")
        return key, pos.value

    def adopt_numpy(self):
        st = np.random.get_state()
        if st[0] != "MT19937":
            raise RuntimeError("global np.random is not the legacy MT19937 stream")
        self._np_tail = st[3:]
        self.set_state(st[1], st[2])

    def release_numpy(self):
        key, pos = self.get_state()
        np.random.set_state(("MT19937", key, pos) + tuple(self._np_tail))

    def shuffle(self, inds):
        if not isinstance(inds, np.ndarray) or inds.ndim != 1 or not inds.flags.c_contiguous:
            raise TypeError("inds must be a contiguous 1-D NumPy array")
        p = inds.ctypes.data_as(ctypes.c_void_p)
        if inds.dtype == np.int64:
            check(lib.ppo2hip_shuffle_i64(self._h, p, inds.size), "shuffle")
        elif inds.dtype == np.int32:
            check(lib.ppo2hip_shuffle_i32(self._h, p, inds.size), "shuffle")
        else:
            raise TypeError("inds must be int32 or int64")
        return inds


class PermPipeline:
    """All epochs' index lists of an update, generated in the background (C++ worker threads)
    into pinned host buffers, from a LegacyShuffler's stream.  ``nsets`` buffer sets let a job
    for the next update be started while copies out of the previous set are still queued."""

    def __init__(self, shuffler, n, max_epochs, pinned=True, nsets=1):
        self.shuffler, self.n, self.max_epochs = shuffler, int(n), int(max_epochs)
        self._h = lib.ppo2hip_perm_create(shuffler._h, self.n, self.max_epochs)
        if not self._h:
            check(_lib.EINVAL, "perm_create")
        pin = bool(pinned and torch.cuda.is_available())
        self.sets = [[torch.empty(self.n, dtype=torch.int32, pin_memory=pin) for _ in range(self.max_epochs)]
                     for _ in range(int(nsets))]
        self._set_ptrs = [(ctypes.c_void_p * self.max_epochs)(*[b.data_ptr() for b in s]) for s in self.sets]
        self.bufs = self.sets[0]

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:      # module globals may be gone at interpreter exit
            lib.ppo2hip_perm_destroy(h)

    def begin(self, nepochs, relative=False, buffer_set=0):
        """Start the job; with ``relative`` the buffers receive each epoch's own permutation of
        arange(n), to be chained with :func:`compose` on the device."""
        fn = lib.ppo2hip_perm_begin_relative if relative else lib.ppo2hip_perm_begin
        check(fn(self._h, int(nepochs), self._set_ptrs[buffer_set]), "perm_begin")
        self.bufs = self.sets[buffer_set]

    def wait(self, epoch):
        check(lib.ppo2hip_perm_wait(self._h, int(epoch)), "perm_wait")
        return self.bufs[epoch]

    def wait_all(self):
        check(lib.ppo2hip_perm_wait_all(self._h), "perm_wait_all")

    def wait_drawn(self):
        """Blocks until the running job has drawn all its swap targets (the generator is free for another pipeline)."""
        check(lib.ppo2hip_perm_wait_drawn(self._h), "perm_wait_drawn")


def compose(prev, sigma, out):
    """out[k] = prev[sigma[k]] on the device (int32): the minibatch order after one more shuffle."""
    for t, name in ((prev, "prev"), (sigma, "sigma"), (out, "out")):
        _dev(t, torch.int32, name)
    n = sigma.numel()
    if prev.numel() != n or out.numel() != n:
        raise ValueError("compose: prev, sigma and out must hold the same number of entries")
    check(lib.ppo2hip_compose_i32(_ptr(prev), _ptr(sigma), _ptr(out), n, _stream()), "compose")
    return out


FAULT_INDEX_RANGE, FAULT_COMPOSE_RANGE, FAULT_GRID_SYNC = 1, 2, 4


def take_faults(raise_on_fault=True):
    """Waits for the stream and collects (and clears) this GPU's fault word: which caller-supplied index lists used since
    the last call named a sample that does not exist (the kernels replaced those entries by sample 0 instead of
    reading out of bounds).  NumPy raises IndexError where the reference indexes with such a list
    (baselines/ppo2/ppo2.py:117); this raises Ppo2HipError(EINVAL)."""
    bits = ctypes.c_uint32(0)
    check(lib.ppo2hip_take_faults(ctypes.byref(bits), _stream()), "take_faults")
    if bits.value and raise_on_fault:
        raise fault_error(bits.value)
    return bits.value


def fault_error(bits):
    """The exception a non-zero fault word stands for."""
    if bits & FAULT_GRID_SYNC:
        return _lib.Ppo2HipError("a launch whose workgroups hand values to each other (whole-sequence LSTM, or the optimiser step folded "
                                 "into the epoch call) gave up waiting for its peers (is the GPU shared with another job?); results since "
                                 "the last check are invalid — PPO2HIP_LSTM_SEQ_OFF=1 takes one LSTM launch per step")
    what = [name for bit, name in ((FAULT_INDEX_RANGE, "a minibatch index list"), (FAULT_COMPOSE_RANGE, "a permutation handed to compose"))
            if bits & bit]
    return _lib.Ppo2HipError("index check failed (code %d): out-of-range entry in %s; results since the last check are invalid"
                             % (_lib.EINVAL, " and ".join(what)))


class FaultWatch:
    """Fault-word checks that do not stall the stream: ``post()`` queues a read-and-clear of this GPU's fault word behind
    the work issued so far (the word lands in page-locked memory); ``poll()`` raises for every posted check that has
    completed with a non-zero word; ``poll(wait=True)`` waits for all of them first."""

    SLOTS = 64

    def __init__(self):
        self._pending = []
        self._ring = None        # page-locked words the one-thread kernel stores into, used in turn
        self._next = 0

    def post(self):
        if self._ring is None:
            self._ring = torch.zeros(self.SLOTS, dtype=torch.int32, pin_memory=True)
        if len(self._pending) >= self.SLOTS - 1:
            self.poll(wait=True)                    # never more checks in flight than words to hold them
        host = self._ring[self._next:self._next + 1]
        self._next = (self._next + 1) % self.SLOTS
        check(lib.ppo2hip_post_faults(ctypes.c_void_p(host.data_ptr()), _stream()), "post_faults")
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((ev, host))

    def poll(self, wait=False):
        keep, bits = [], 0
        for ev, host in self._pending:
            if wait:
                ev.synchronize()
            if ev.query():
                bits |= int(host.item()) & 0xFFFFFFFF
            else:
                keep.append((ev, host))
        self._pending = keep
        if bits:
            raise fault_error(bits)


# --------------------------------------------------------------------------------------
# policy arena, forward, gradient, optimiser
# --------------------------------------------------------------------------------------

class PolicySpec:
    """Python view of ppo2hip_policy_desc plus the arena layout it implies."""

    def __init__(self, kind, dist, ac_dim, nvalue, ob_dim=0, ob_hwc=(0, 0, 0), hidden=(0, 0), order=_lib.ORDER_FORK,
                 nlstm=0):
        self.desc = _lib.PolicyDesc(kind, dist, order, ob_dim, ob_hwc[0], ob_hwc[1], ob_hwc[2], hidden[0], hidden[1],
                                    ac_dim, nvalue, nlstm)
        n = lib.ppo2hip_param_count(ctypes.byref(self.desc))
        if n < 0:
            check(int(n), "param_count")
        self.nparams = int(n)
        names = ((ctypes.c_char * 32) * 64)()
        offs = (ctypes.c_int64 * 64)()
        shapes = (ctypes.c_int64 * 256)()
        nv = lib.ppo2hip_param_layout(ctypes.byref(self.desc), 64, names, offs, shapes)
        if nv < 0:
            check(nv, "param_layout")
        self.vars = []
        for i in range(nv):
            shp = tuple(int(shapes[i * 4 + k]) for k in range(4) if shapes[i * 4 + k] != 0)
            self.vars.append((names[i].value.decode(), int(offs[i]), shp))
        self.kind, self.dist, self.ac_dim, self.nvalue, self.nlstm = kind, dist, ac_dim, nvalue, nlstm
        self.ob_dim, self.ob_hwc = ob_dim, tuple(ob_hwc)
        self.recurrent = kind in (_lib.POLICY_CNN_LSTM, _lib.POLICY_CNN_LNLSTM)

    @property
    def gaussian(self):
        return self.dist == _lib.DIST_GAUSSIAN

    def ob_shape(self):
        return (self.ob_dim,) if self.kind == _lib.POLICY_MLP else self.ob_hwc

    def ob_dtype(self):
        return torch.float32 if self.kind == _lib.POLICY_MLP else torch.uint8

    def views(self, arena):
        """name -> view of `arena` (device or host tensor / ndarray) in creation order."""
        out = {}
        for name, off, shp in self.vars:
            n = int(np.prod(shp))
            out[name] = arena[off:off + n].reshape(shp)
        return out

    def act_workspace_bytes(self, B):
        return int(lib.ppo2hip_act_workspace_bytes(ctypes.byref(self.desc), B))

    def forward_workspace_bytes(self, B):
        return int(lib.ppo2hip_forward_workspace_bytes(ctypes.byref(self.desc), B))

    def grad_workspace_bytes(self, B):
        return int(lib.ppo2hip_grad_workspace_bytes(ctypes.byref(self.desc), B))


def policy_forward(spec, params, obs, workspace, masks=None, states=None, vf_out=None):
    """obs [B, ...] -> (pi [B, ac_dim], vf [B, nvalue], new_states or None); ``vf_out`` may be a
    preallocated [B, nvalue] row of a rollout arena."""
    params = _dev(params, torch.float32, "params")
    if params.numel() != spec.nparams:
        raise ValueError("params has %d entries, policy needs %d" % (params.numel(), spec.nparams))
    obs = _dev(obs, spec.ob_dtype(), "obs")
    if tuple(obs.shape[1:]) != tuple(spec.ob_shape()):
        raise ValueError("obs rows must have shape %s, got %s" % (spec.ob_shape(), tuple(obs.shape[1:])))
    B = obs.shape[0]
    need = spec.forward_workspace_bytes(B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        raise ValueError("forward workspace too small: need %d bytes" % need)
    pi = torch.empty((B, spec.ac_dim), dtype=torch.float32, device=obs.device)
    if vf_out is None:
        vf = torch.empty((B, spec.nvalue), dtype=torch.float32, device=obs.device)
    else:
        vf = _dev(vf_out, torch.float32, "vf_out")
        if vf.numel() != B * spec.nvalue:
            raise ValueError("vf_out must hold [B, %d]" % spec.nvalue)
    snew = None
    if spec.recurrent:
        masks = _dev(_as_u8(masks), torch.uint8, "masks")
        states = _dev(states, torch.float32, "states")
        if masks.numel() != B or tuple(states.shape) != (B, 2 * spec.nlstm):
            raise ValueError("recurrent forward needs masks [B] and states [B, 2*nlstm]")
        snew = torch.empty_like(states)
    check(lib.ppo2hip_policy_forward(ctypes.byref(spec.desc), _ptr(params), _ptr(obs), B, _ptr(masks), _ptr(states),
                                     _ptr(pi), _ptr(vf), _ptr(snew), _ptr(workspace),
                                     workspace.numel() * workspace.element_size(), _stream()), "policy_forward")
    return pi, vf, snew


def neglogp(spec, params, pi, actions):
    pi = _dev(pi, torch.float32, "pi")
    B = pi.shape[0]
    actions = _dev(actions, torch.float32 if spec.gaussian else torch.int32, "actions")
    if actions.numel() != B * (spec.ac_dim if spec.gaussian else 1):
        raise ValueError("actions has the wrong number of entries")
    out = torch.empty(B, dtype=torch.float32, device=pi.device)
    check(lib.ppo2hip_neglogp(ctypes.byref(spec.desc), _ptr(params), _ptr(pi), _ptr(actions), B, _ptr(out), _stream()),
          "neglogp")
    return out


def sample_actions(spec, params, pi, seed, step, deterministic=False, want_noise=False, actions=None, neglogp=None):
    """Draw actions from ``pi`` [B, ac_dim] and evaluate their neglogp; the noise is a pure function of
    (seed, step, row, element).  ``actions`` / ``neglogp`` may be preallocated rows of a rollout arena."""
    pi = _dev(pi, torch.float32, "pi")
    B = pi.shape[0]
    if pi.dim() != 2 or pi.shape[1] != spec.ac_dim:
        raise ValueError("pi must be [B, %d]" % spec.ac_dim)
    if actions is None:
        actions = (torch.empty((B, spec.ac_dim), dtype=torch.float32, device=pi.device) if spec.gaussian
                   else torch.empty(B, dtype=torch.int32, device=pi.device))
    else:
        _dev(actions, torch.float32 if spec.gaussian else torch.int32, "actions")
        if actions.numel() != (B * spec.ac_dim if spec.gaussian else B):
            raise ValueError("actions has the wrong size")
    if neglogp is None:
        neglogp = torch.empty(B, dtype=torch.float32, device=pi.device)
    elif _dev(neglogp, torch.float32, "neglogp").numel() != B:
        raise ValueError("neglogp has the wrong size")
    noise = torch.empty((B, spec.ac_dim), dtype=torch.float32, device=pi.device) if want_noise else None
    check(lib.ppo2hip_sample_actions(ctypes.byref(spec.desc), _ptr(params), _ptr(pi), B, int(seed) & (2 ** 64 - 1),
                                     int(step) & (2 ** 64 - 1), int(bool(deterministic)), _ptr(actions), _ptr(neglogp),
                                     _ptr(noise), _stream()), "sample_actions")
    return (actions, neglogp, noise) if want_noise else (actions, neglogp)


def act_step(spec, params, obs, workspace, seed, step, masks=None, states=None, deterministic=False, actions=None,
             values=None, neglogp=None):
    """Forward + sampling + neglogp of one rollout step in one library call.  ``actions`` / ``values`` /
    ``neglogp`` may be rows of a rollout arena.  Returns (actions, values, new_states or None, neglogp)."""
    obs = _dev(obs, spec.ob_dtype(), "obs")
    if tuple(obs.shape[1:]) != tuple(spec.ob_shape()):
        raise ValueError("obs rows must have shape %s, got %s" % (spec.ob_shape(), tuple(obs.shape[1:])))
    B = obs.shape[0]
    dev_ = obs.device
    if actions is None:
        actions = (torch.empty((B, spec.ac_dim), dtype=torch.float32, device=dev_) if spec.gaussian
                   else torch.empty(B, dtype=torch.int32, device=dev_))
    elif _dev(actions, torch.float32 if spec.gaussian else torch.int32, "actions").numel() != (B * spec.ac_dim if spec.gaussian else B):
        raise ValueError("actions has the wrong size")
    if values is None:
        values = torch.empty((B, spec.nvalue), dtype=torch.float32, device=dev_)
    elif _dev(values, torch.float32, "values").numel() != B * spec.nvalue:
        raise ValueError("values has the wrong size")
    if neglogp is None:
        neglogp = torch.empty(B, dtype=torch.float32, device=dev_)
    elif _dev(neglogp, torch.float32, "neglogp").numel() != B:
        raise ValueError("neglogp has the wrong size")
    snew = None
    if spec.recurrent:
        masks = _dev(_as_u8(masks), torch.uint8, "masks")
        states = _dev(states, torch.float32, "states")
        if masks.numel() != B or tuple(states.shape) != (B, 2 * spec.nlstm):
            raise ValueError("recurrent step needs masks [B] and states [B, 2*nlstm]")
        snew = torch.empty_like(states)
    else:
        masks = states = None
    wbytes = workspace.numel() * workspace.element_size()
    check(lib.ppo2hip_act_step(ctypes.byref(spec.desc), _ptr(_dev(params, torch.float32, "params")), _ptr(obs), B, _ptr(masks),
                               _ptr(states), int(seed) & (2 ** 64 - 1), int(step) & (2 ** 64 - 1), int(bool(deterministic)),
                               _ptr(actions), _ptr(values), _ptr(neglogp), _ptr(snew), _ptr(workspace), wbytes, _stream()),
          "act_step")
    return actions, values, snew, neglogp


def done_bootstrap(rewards, dones, v_reset, gamma):
    """In place: rewards[e] += gamma * v_reset[e] for finished envs, by the fork's channel rule."""
    rewards = _dev(rewards, torch.float32, "rewards")
    v_reset = _dev(v_reset, torch.float32, "v_reset")
    if rewards.dim() != 2 or v_reset.shape != rewards.shape:
        raise ValueError("rewards and v_reset must both be [E, R]")
    E, R = rewards.shape
    dones = _dev(_as_u8(dones), torch.uint8, "dones")
    if dones.numel() != E:
        raise ValueError("dones must be [E]")
    g = np.ascontiguousarray(np.broadcast_to(np.asarray(gamma, np.float64), (R,)))
    check(lib.ppo2hip_done_bootstrap(_ptr(rewards), _ptr(dones), _ptr(v_reset), g.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                     E, R, _stream()), "done_bootstrap")
    return rewards


class RolloutRef:
    """Checked view of the arrays a minibatch is drawn from (ppo2hip_rollout)."""

    def __init__(self, spec, obs, actions, returns, values, neglogpacs, advs, T, E, masks=None, states=None):
        n = T * E if T else obs.shape[0]
        R = spec.nvalue
        self.keep = (obs, actions, returns, values, neglogpacs, advs, masks, states)
        _dev(obs, spec.ob_dtype(), "obs")
        _dev(actions, torch.float32 if spec.gaussian else torch.int32, "actions")
        _dev(returns, torch.float32, "returns")
        _dev(values, torch.float32, "values")
        _dev(neglogpacs, torch.float32, "neglogpacs")
        if advs.dtype not in (torch.float32, torch.float64) or not advs.is_cuda or not advs.is_contiguous():
            raise TypeError("advs must be a contiguous float32/float64 device tensor")
        ob_elems = int(np.prod(spec.ob_shape()))
        for t, cnt, name in ((obs, n * ob_elems, "obs"), (actions, n * (spec.ac_dim if spec.gaussian else 1), "actions"),
                             (returns, n * R, "returns"), (values, n * R, "values"), (neglogpacs, n, "neglogpacs"),
                             (advs, n, "advs")):
            if t.numel() != cnt:
                raise ValueError("%s holds %d entries, rollout needs %d" % (name, t.numel(), cnt))
        if spec.recurrent:
            masks = _dev(_as_u8(masks), torch.uint8, "masks")
            states = _dev(states, torch.float32, "states")
            if masks.numel() != n:
                raise ValueError("masks must hold one flag per sample")
        self.n = n
        self.c = _lib.Rollout(obs.data_ptr(), actions.data_ptr(), returns.data_ptr(), values.data_ptr(),
                              neglogpacs.data_ptr(), advs.data_ptr(),
                              masks.data_ptr() if masks is not None else None,
                              states.data_ptr() if states is not None else None,
                              int(advs.dtype == torch.float64), T, E, 0, n)


def ppo_grad(spec, loss_desc, params, rollout, idx, B, stats, cliprange, inv_count, grads, losses, workspace,
             mb_envs=None, nsteps_mb=0):
    """One minibatch: gather + forward + loss + backward into `grads` / `losses` (device)."""
    _dev(params, torch.float32, "params")
    _dev(grads, torch.float32, "grads")
    _dev(losses, torch.float32, "losses")
    if params.numel() != spec.nparams or grads.numel() != spec.nparams or losses.numel() < 8:
        raise ValueError("params/grads must have %d entries and losses at least 8" % spec.nparams)
    if idx is not None:
        _dev(idx, torch.int32, "idx")
        if idx.numel() < B:
            raise ValueError("idx shorter than B")
    elif rollout.n < B:
        raise ValueError("rollout shorter than B")
    if stats is not None:
        _dev(stats, torch.float64, "stats")
        if stats.numel() < 3:
            raise ValueError("stats needs 3 entries")
    need = spec.grad_workspace_bytes(B)
    have = workspace.numel() * workspace.element_size()
    if have < need:
        raise ValueError("grad workspace too small: need %d bytes, have %d" % (need, have))
    if mb_envs is not None:
        _dev(mb_envs, torch.int32, "mb_envs")
    check(lib.ppo2hip_ppo_grad(ctypes.byref(spec.desc), ctypes.byref(loss_desc), _ptr(params), ctypes.byref(rollout.c),
                               _ptr(idx), B, _ptr(mb_envs), nsteps_mb, _ptr(stats), float(cliprange),
                               float(inv_count), _ptr(grads), _ptr(losses), _ptr(workspace), have, _stream()),
          "ppo_grad")


def grad_workspace_region(spec, workspace, B, name, nsteps_mb=0):
    """View of one of the nature_cnn trunk's activation blocks ("c1", "c2", "c3", "fc1") inside a gradient
    workspace, as the last ``ppo_grad`` call with this (B, nsteps_mb) left it (for tests)."""
    off, n = ctypes.c_size_t(0), ctypes.c_int64(0)
    check(lib.ppo2hip_grad_workspace_region(ctypes.byref(spec.desc), int(B), int(nsteps_mb), name.encode(), ctypes.byref(off),
                                            ctypes.byref(n)), "grad_workspace_region")
    if off.value + 4 * n.value > workspace.numel() * workspace.element_size():
        raise ValueError("workspace smaller than the region")
    return workspace.view(torch.uint8)[off.value:off.value + 4 * n.value].view(torch.float32)


def adam_scratch(P, device):
    return torch.zeros(int(lib.ppo2hip_adam_scratch_bytes(P)), dtype=torch.uint8, device=device)


def adam_step(params, grads, m, v, lr, t, scratch, max_grad_norm=0.0, grad_scale=1.0, beta1=0.9, beta2=0.999,
              epsilon=1e-5):
    for x, name in ((params, "params"), (grads, "grads"), (m, "m"), (v, "v")):
        _dev(x, torch.float32, name)
    P = params.numel()
    if grads.numel() != P or m.numel() != P or v.numel() != P:
        raise ValueError("params/grads/m/v sizes differ")
    if scratch.numel() * scratch.element_size() < lib.ppo2hip_adam_scratch_bytes(P):
        raise ValueError("adam scratch too small")
    check(lib.ppo2hip_adam_step(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), P, float(lr), int(t), beta1, beta2,
                                epsilon, float(grad_scale), float(max_grad_norm or 0.0), _ptr(scratch), _stream()),
          "adam_step")


class NativeComm:
    """ppo2hip_comm: the library's own RCCL communicator (one per process, on the current device).  ``unique_id``
    None makes a single-rank communicator that touches no collective library."""

    def __init__(self, unique_id, rank, size):
        if unique_id is not None and len(unique_id) != _lib.UNIQUE_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % _lib.UNIQUE_ID_BYTES)
        buf = ctypes.create_string_buffer(bytes(unique_id), _lib.UNIQUE_ID_BYTES) if unique_id is not None else None
        self._h = lib.ppo2hip_comm_create(buf, int(rank), int(size))
        if not self._h:
            check(_lib.ECOMM, "comm_create")
        self.rank, self.size = int(rank), int(size)
        self._digest_ws = None

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        check(lib.ppo2hip_comm_unique_id(buf), "comm_unique_id")
        return buf.raw

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:      # module globals may be gone at interpreter exit
            lib.ppo2hip_comm_destroy(h)

    def allreduce_sum_(self, t):
        if not t.is_cuda or not t.is_contiguous():
            raise TypeError("allreduce needs a contiguous device tensor")
        if t.dtype == torch.float32:
            check(lib.ppo2hip_allreduce_grads(self._h, _ptr(t), t.numel(), _stream()), "allreduce_grads")
        elif t.dtype == torch.float64:
            check(lib.ppo2hip_allreduce_f64(self._h, _ptr(t), t.numel(), _stream()), "allreduce_f64")
        else:
            raise TypeError("allreduce carries float32 or float64")
        return t

    def bcast_(self, t, root=0):
        _dev(t, torch.float32, "params")
        check(lib.ppo2hip_bcast_params(self._h, _ptr(t), t.numel(), int(root), _stream()), "bcast_params")
        return t

    def replica_checksum(self, theta):
        """Queues the digest exchange; returns a device int32 tensor that becomes 1 (replicas bit-identical) or 0."""
        _dev(theta, torch.float32, "params")
        if self._digest_ws is None:
            self._digest_ws = torch.zeros(4, dtype=torch.int64, device=theta.device)
        flag = torch.zeros(1, dtype=torch.int32, device=theta.device)
        check(lib.ppo2hip_replica_checksum(self._h, _ptr(theta), theta.numel(), _ptr(self._digest_ws), _ptr(flag), _stream()),
              "replica_checksum")
        return flag


def ppo_epoch(spec, loss_desc, params, grads, m, v, rollout, idx, nbatch_train, stats, cliprange, lr, t_first,
              losses, workspace, adam_scratch, max_grad_norm=0.0, beta1=0.9, beta2=0.999, epsilon=1e-5, comm=None,
              grad_scale=1.0):
    """All minibatches of one epoch, gradient + Adam each, in one library call; with ``comm`` (a NativeComm) the
    gradient arena is all-reduced over the ranks between the two and scaled by ``grad_scale``."""
    for x, name in ((params, "params"), (grads, "grads"), (m, "m"), (v, "v"), (losses, "losses")):
        _dev(x, torch.float32, name)
    _dev(idx, torch.int32, "idx")
    P, N = spec.nparams, idx.numel()
    if params.numel() != P or grads.numel() != P or m.numel() != P or v.numel() != P:
        raise ValueError("params/grads/m/v must have %d entries" % P)
    if N % nbatch_train != 0:
        raise ValueError("idx must hold a whole number of minibatches")
    nmb = N // nbatch_train
    if losses.numel() < 8 * nmb:
        raise ValueError("losses needs 8 entries per minibatch")
    if stats is not None:
        _dev(stats, torch.float64, "stats")
        if stats.numel() < 3 * nmb:
            raise ValueError("stats needs 3 entries per minibatch")
    have = workspace.numel() * workspace.element_size()
    if have < spec.grad_workspace_bytes(nbatch_train):
        raise ValueError("grad workspace too small")
    if adam_scratch.numel() * adam_scratch.element_size() < lib.ppo2hip_adam_scratch_bytes(P):
        raise ValueError("adam scratch too small")
    if comm is not None:
        check(lib.ppo2hip_ppo_epoch_dp(ctypes.byref(spec.desc), ctypes.byref(loss_desc), _ptr(params), _ptr(grads), _ptr(m),
                                       _ptr(v), ctypes.byref(rollout.c), _ptr(idx), N, int(nbatch_train), _ptr(stats),
                                       float(cliprange), float(lr), int(t_first), beta1, beta2, epsilon,
                                       float(max_grad_norm or 0.0), _ptr(losses), _ptr(workspace), have, _ptr(adam_scratch),
                                       comm._h, float(grad_scale), _stream()), "ppo_epoch_dp")
        return
    check(lib.ppo2hip_ppo_epoch(ctypes.byref(spec.desc), ctypes.byref(loss_desc), _ptr(params), _ptr(grads), _ptr(m),
                                _ptr(v), ctypes.byref(rollout.c), _ptr(idx), N, int(nbatch_train), _ptr(stats),
                                float(cliprange), float(lr), int(t_first), beta1, beta2, epsilon,
                                float(max_grad_norm or 0.0), _ptr(losses), _ptr(workspace), have, _ptr(adam_scratch),
                                _stream()), "ppo_epoch")


# --------------------------------------------------------------------------------------
# one global permutation over env-sharded ranks: ownership filter, statistics and epoch call over a rank's shares
# --------------------------------------------------------------------------------------

def filter_owned(idx, nbatch_train, nmb, lo, hi, out, out_stride, counts, workspace=None):
    """Minibatch j of the global list ``idx`` ([nmb * nbatch_train] int32) -> this rank's entries (sample numbers in
    [lo, hi), re-based to 0) at ``out[j * out_stride:]`` in list order, their number in ``counts[j]`` (device int32)."""
    idx = _dev(idx, torch.int32, "idx")
    out = _dev(out, torch.int32, "out")
    counts = _dev(counts, torch.int32, "counts")
    if idx.numel() < nmb * nbatch_train or out.numel() < nmb * out_stride or counts.numel() < nmb:
        raise ValueError("filter_owned: idx, out or counts too short")
    wsb = lib.ppo2hip_filter_owned_workspace_bytes(int(nbatch_train), int(nmb))
    if workspace is None:
        workspace = torch.empty(wsb, dtype=torch.uint8, device=idx.device)
    have = workspace.numel() * workspace.element_size()
    check(lib.ppo2hip_filter_owned_i32(_ptr(idx), int(nbatch_train), int(nmb), int(lo), int(hi), _ptr(out), int(out_stride),
                                       _ptr(counts), _ptr(workspace), have, _stream()), "filter_owned")
    return out, counts


def adv_stats_owned(advs, idx, nbatch_train, nmb, lo, hi, T, E, workspace=None, out=None):
    """{sum a, sum a^2, entries owned} of this rank's share of every global minibatch (float64 [nmb, 3], device);
    ``advs`` is the rank's own time-major [T, E] array."""
    if advs.dtype not in (torch.float32, torch.float64) or not advs.is_cuda or not advs.is_contiguous():
        raise TypeError("advs must be a contiguous float32/float64 device tensor")
    idx = _dev(idx, torch.int32, "idx")
    if idx.numel() < nmb * nbatch_train:
        raise ValueError("idx shorter than nmb * nbatch_train")
    if T * E != advs.numel() or hi - lo > T * E:
        raise ValueError("advs must hold T*E entries and the owned range must fit them")
    wsb = lib.ppo2hip_adv_stats_owned_workspace_bytes(int(nmb))
    if workspace is None:
        workspace = torch.empty(wsb, dtype=torch.uint8, device=advs.device)
    elif workspace.numel() * workspace.element_size() < wsb:
        raise ValueError("adv_stats_owned workspace too small")
    if out is None:
        out = torch.empty((nmb, 3), dtype=torch.float64, device=advs.device)
    check(lib.ppo2hip_adv_stats_owned(_ptr(advs), int(advs.dtype == torch.float64), _ptr(idx), int(nbatch_train), int(nmb),
                                      int(lo), int(hi), int(T), int(E), _ptr(workspace), _ptr(out), _stream()), "adv_stats_owned")
    return out


def ppo_epoch_shares(spec, loss_desc, params, grads, m, v, rollout, idx, idx_stride, counts, stats, cliprange, inv_count, lr,
                     t_first, losses, workspace, adam_scratch, max_grad_norm=0.0, beta1=0.9, beta2=0.999, epsilon=1e-5,
                     comm=None):
    """An epoch over this rank's shares of the global minibatches (``counts``: host ints, one per minibatch): gradient
    sums scaled by ``inv_count`` = 1 / global minibatch size, SUM all-reduce over ``comm`` (a NativeComm or None), Adam."""
    for x, name in ((params, "params"), (grads, "grads"), (m, "m"), (v, "v"), (losses, "losses")):
        _dev(x, torch.float32, name)
    _dev(idx, torch.int32, "idx")
    _dev(stats, torch.float64, "stats")
    nmb, P = len(counts), spec.nparams
    if idx.numel() < nmb * idx_stride or losses.numel() < 8 * nmb or stats.numel() < 3 * nmb:
        raise ValueError("ppo_epoch_shares: idx, losses or stats too short")
    have = workspace.numel() * workspace.element_size()
    if have < spec.grad_workspace_bytes(max(max(counts), 1)):
        raise ValueError("grad workspace too small")
    c_counts = (ctypes.c_int64 * nmb)(*[int(c) for c in counts])
    check(lib.ppo2hip_ppo_epoch_shares(ctypes.byref(spec.desc), ctypes.byref(loss_desc), _ptr(params), _ptr(grads), _ptr(m),
                                       _ptr(v), ctypes.byref(rollout.c), _ptr(idx), int(idx_stride), c_counts, nmb, _ptr(stats),
                                       float(cliprange), float(inv_count), float(lr), int(t_first), beta1, beta2, epsilon,
                                       float(max_grad_norm or 0.0), _ptr(losses), _ptr(workspace), have, _ptr(adam_scratch),
                                       comm._h if comm is not None else None, _stream()), "ppo_epoch_shares")

def row_mean(rows, nrows, ncols, scale=None):
    """scale * column sums of the first ``nrows`` rows and ``ncols`` columns of a float32 [*, stride] tensor
    (default scale 1/nrows: the mean)."""
    rows = _dev(rows, torch.float32, "rows")
    if rows.dim() != 2 or nrows < 1 or nrows > rows.shape[0] or not 1 <= ncols <= min(16, rows.shape[1]):
        raise ValueError("row_mean: bad shape")
    out = torch.empty(ncols, dtype=torch.float32, device=rows.device)
    check(lib.ppo2hip_row_mean(_ptr(rows), int(nrows), int(rows.shape[1]), int(ncols),
                               float(1.0 / nrows if scale is None else scale), _ptr(out), _stream()), "row_mean")
    return out


def gemm(A, B):
    A = _dev(A, torch.float32, "A")
    B = _dev(B, torch.float32, "B")
    M, K = A.shape
    K2, N = B.shape
    if K != K2:
        raise ValueError("inner dimensions differ")
    C = torch.empty((M, N), dtype=torch.float32, device=A.device)
    check(lib.ppo2hip_gemm_f32(_ptr(A), _ptr(B), _ptr(C), M, N, K, _stream()), "gemm")
    return C
