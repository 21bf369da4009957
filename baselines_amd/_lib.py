"""ctypes binding of libppo2hip.so — the only door from Python to the HIP kernels.

There is no fallback: if the shared library is missing or a symbol does not
resolve, importing this module raises.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_uint32, c_uint64, c_void_p)

import torch  # noqa: F401  (loads the process's HIP runtime before ours resolves against it)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libppo2hip.so")

OK, EINVAL, ENOSPACE, EUNSUPPORTED, ECOMM = 0, -1, -2, -3, -4
UNIQUE_ID_BYTES = 128
POLICY_MLP, POLICY_CNN, POLICY_CNN_LSTM, POLICY_CNN_LNLSTM = 0, 1, 2, 3
DIST_GAUSSIAN, DIST_CATEGORICAL = 0, 1
ORDER_FORK, ORDER_UPSTREAM = 0, 1


class PolicyDesc(Structure):
    _fields_ = [("kind", c_int32), ("dist", c_int32), ("order", c_int32), ("ob_dim", c_int32),
                ("ob_h", c_int32), ("ob_w", c_int32), ("ob_c", c_int32), ("hidden1", c_int32),
                ("hidden2", c_int32), ("ac_dim", c_int32), ("nvalue", c_int32), ("nlstm", c_int32)]


class LossDesc(Structure):
    _fields_ = [("ent_coef", c_float), ("vf_coef", c_float), ("max_grad_norm", c_float), ("upstream", c_int32)]


class Rollout(Structure):
    _fields_ = [("obs", c_void_p), ("actions", c_void_p), ("returns", c_void_p), ("values", c_void_p),
                ("neglogpacs", c_void_p), ("advs", c_void_p), ("masks", c_void_p), ("states", c_void_p),
                ("adv_f64", c_int32), ("T", c_int32), ("E", c_int32), ("_pad", c_int32), ("n", c_int64)]


class Ppo2HipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libppo2hip.so not found at %s — build it with `make -C baselines_amd/csrc` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback" % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

_P = c_void_p
_SIGS = {
    "ppo2hip_version": (c_int, []),
    "ppo2hip_last_error": (c_char_p, []),
    "ppo2hip_gae_f32": (c_int, [_P, _P, _P, _P, _P, POINTER(c_double), c_double, c_int, c_int, c_int, _P, _P, _P, _P]),
    "ppo2hip_discount_with_dones_f32": (c_int, [_P, _P, _P, c_double, c_int, c_int, _P, _P]),
    "ppo2hip_rms_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ppo2hip_rms_update": (c_int, [_P, c_int, c_int64, c_int64, _P, _P, c_size_t, _P]),
    "ppo2hip_column_moments": (c_int, [_P, c_int, c_int64, c_int64, _P, _P, _P, c_size_t, _P]),
    "ppo2hip_add_f64": (c_int, [_P, _P, c_int64, _P]),
    "ppo2hip_normalize_clip": (c_int, [_P, c_int, c_int64, c_int64, _P, _P, c_double, c_double, _P, c_int, _P]),
    "ppo2hip_vecnorm_returns": (c_int, [_P, c_int, c_int64, c_double, _P, _P, c_double, c_double, _P, c_int, _P,
                                        c_size_t, _P]),
    "ppo2hip_frame_stack": (c_int, [_P, _P, _P, c_int64, c_int64, c_int, c_int, c_int, _P]),
    "ppo2hip_sample_actions": (c_int, [POINTER(PolicyDesc), _P, _P, c_int64, c_uint64, c_uint64, c_int, _P, _P, _P, _P]),
    "ppo2hip_act_workspace_bytes": (c_size_t, [POINTER(PolicyDesc), c_int64]),
    "ppo2hip_act_step": (c_int, [POINTER(PolicyDesc), _P, _P, c_int64, _P, _P, c_uint64, c_uint64, c_int, _P, _P, _P, _P, _P,
                                 c_size_t, _P]),
    "ppo2hip_done_bootstrap": (c_int, [_P, _P, _P, POINTER(c_double), c_int64, c_int, _P]),
    "ppo2hip_row_mean": (c_int, [_P, c_int, c_int, c_int, c_double, _P, _P]),
    "ppo2hip_discount": (c_int, [_P, c_int, _P, c_double, c_int, c_int64, _P, c_int, _P]),
    "ppo2hip_sf01": (c_int, [_P, c_int, c_int, c_int64, _P, _P]),
    "ppo2hip_compute_advs_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "ppo2hip_compute_advs": (c_int, [_P, _P, POINTER(c_double), c_int64, c_int, _P, c_size_t, _P, _P]),
    "ppo2hip_adv_stats_workspace_bytes": (c_size_t, [c_int]),
    "ppo2hip_adv_stats": (c_int, [_P, c_int, _P, c_int64, c_int, c_int, c_int, _P, _P, _P]),
    "ppo2hip_rng_create": (c_void_p, [c_uint32]),
    "ppo2hip_rng_destroy": (None, [c_void_p]),
    "ppo2hip_rng_seed": (None, [c_void_p, c_uint32]),
    "ppo2hip_rng_set_state": (c_int, [c_void_p, _P, c_int]),
    "ppo2hip_rng_get_state": (c_int, [c_void_p, _P, POINTER(c_int)]),
    "ppo2hip_shuffle_i64": (c_int, [c_void_p, _P, c_int64]),
    "ppo2hip_shuffle_i32": (c_int, [c_void_p, _P, c_int64]),
    "ppo2hip_perm_create": (c_void_p, [c_void_p, c_int64, c_int]),
    "ppo2hip_perm_destroy": (None, [c_void_p]),
    "ppo2hip_perm_begin": (c_int, [c_void_p, c_int, _P]),
    "ppo2hip_perm_begin_relative": (c_int, [c_void_p, c_int, _P]),
    "ppo2hip_compose_i32": (c_int, [_P, _P, _P, c_int64, _P]),
    "ppo2hip_take_faults": (c_int, [ctypes.POINTER(ctypes.c_uint32), _P]),
    "ppo2hip_post_faults": (c_int, [_P, _P]),
    "ppo2hip_perm_wait": (c_int, [c_void_p, c_int]),
    "ppo2hip_perm_wait_all": (c_int, [c_void_p]),
    "ppo2hip_perm_wait_drawn": (c_int, [c_void_p]),
    "ppo2hip_param_count": (c_int64, [POINTER(PolicyDesc)]),
    "ppo2hip_param_layout": (c_int, [POINTER(PolicyDesc), c_int, _P, _P, _P]),
    "ppo2hip_forward_workspace_bytes": (c_size_t, [POINTER(PolicyDesc), c_int64]),
    "ppo2hip_policy_forward": (c_int, [POINTER(PolicyDesc), _P, _P, c_int64, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "ppo2hip_neglogp": (c_int, [POINTER(PolicyDesc), _P, _P, _P, c_int64, _P, _P]),
    "ppo2hip_grad_workspace_bytes": (c_size_t, [POINTER(PolicyDesc), c_int64]),
    "ppo2hip_grad_workspace_region": (c_int, [POINTER(PolicyDesc), c_int64, c_int32, c_char_p, POINTER(c_size_t),
                                              POINTER(c_int64)]),
    "ppo2hip_ppo_grad": (c_int, [POINTER(PolicyDesc), POINTER(LossDesc), _P, POINTER(Rollout), _P, c_int64, _P,
                                 c_int32, _P, c_float, c_double, _P, _P, _P, c_size_t, _P]),
    "ppo2hip_adam_scratch_bytes": (c_size_t, [c_int64]),
    "ppo2hip_ppo_epoch": (c_int, [POINTER(PolicyDesc), POINTER(LossDesc), _P, _P, _P, _P, POINTER(Rollout), _P, c_int64,
                                  c_int64, _P, c_float, c_double, c_int64, c_double, c_double, c_double, c_double, _P,
                                  _P, c_size_t, _P, _P]),
    "ppo2hip_adam_step": (c_int, [_P, _P, _P, _P, c_int64, c_double, c_int64, c_double, c_double, c_double,
                                  c_double, c_double, _P, _P]),
    "ppo2hip_adam_step_static": (c_int, [_P, _P, _P, _P, c_int64, c_double, c_int64, c_double, c_double, c_double,
                                         c_double, c_double, _P, _P]),
    "ppo2hip_comm_unique_id": (c_int, [_P]),
    "ppo2hip_comm_create": (c_void_p, [_P, c_int, c_int]),
    "ppo2hip_comm_destroy": (None, [c_void_p]),
    "ppo2hip_comm_rank": (c_int, [c_void_p]),
    "ppo2hip_comm_size": (c_int, [c_void_p]),
    "ppo2hip_allreduce_grads": (c_int, [c_void_p, _P, c_int64, _P]),
    "ppo2hip_allreduce_f64": (c_int, [c_void_p, _P, c_int64, _P]),
    "ppo2hip_bcast_params": (c_int, [c_void_p, _P, c_int64, c_int, _P]),
    "ppo2hip_replica_checksum": (c_int, [c_void_p, _P, c_int64, _P, _P, _P]),
    "ppo2hip_ppo_epoch_dp": (c_int, [POINTER(PolicyDesc), POINTER(LossDesc), _P, _P, _P, _P, POINTER(Rollout), _P, c_int64,
                                     c_int64, _P, c_float, c_double, c_int64, c_double, c_double, c_double, c_double, _P,
                                     _P, c_size_t, _P, c_void_p, c_double, _P]),
    "ppo2hip_filter_owned_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "ppo2hip_filter_owned_i32": (c_int, [_P, c_int64, c_int, c_int32, c_int32, _P, c_int64, _P, _P, c_size_t, _P]),
    "ppo2hip_adv_stats_owned_workspace_bytes": (c_size_t, [c_int]),
    "ppo2hip_adv_stats_owned": (c_int, [_P, c_int, _P, c_int64, c_int, c_int32, c_int32, c_int, c_int, _P, _P, _P]),
    "ppo2hip_ppo_epoch_shares": (c_int, [POINTER(PolicyDesc), POINTER(LossDesc), _P, _P, _P, _P, POINTER(Rollout), _P, c_int64,
                                         POINTER(c_int64), c_int, _P, c_float, c_double, c_double, c_int64, c_double, c_double,
                                         c_double, c_double, _P, _P, c_size_t, _P, c_void_p, _P]),
    "ppo2hip_gemm_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
}

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)   # AttributeError here = header and library disagree: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = sorted(_SIGS)


def check(rc, what=""):
    """Raise on a non-zero return code of a ppo2hip_* call."""
    if rc != 0:
        msg = lib.ppo2hip_last_error()
        raise Ppo2HipError("%s failed (code %d): %s" % (what or "ppo2hip call", rc, msg.decode() if msg else ""))
