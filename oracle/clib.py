"""ctypes loader for the plain-C checker (``oracle/scan_perm_c.c``).  Test infrastructure."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    return lib


def shuffle_epochs(seed, n, nepochs):
    lib = load()
    out = np.empty((nepochs, n), np.int64)
    lib.oracle_shuffle_epochs(ctypes.c_uint32(seed), ctypes.c_int64(n), ctypes.c_int(nepochs),
                              out.ctypes.data_as(ctypes.c_void_p))
    return out


def gae(rewards, values, dones, last_values, last_dones, gamma, lam):
    lib = load()
    rewards = np.ascontiguousarray(rewards, np.float32)
    values = np.ascontiguousarray(values, np.float32)
    T, E, R = rewards.shape
    dones = np.ascontiguousarray(np.asarray(dones).astype(np.uint8))
    last_dones = np.ascontiguousarray(np.asarray(last_dones).astype(np.uint8))
    last_values = np.ascontiguousarray(last_values, np.float32)
    gamma = np.ascontiguousarray(np.broadcast_to(np.asarray(gamma, np.float64), (R,)))
    advs = np.empty_like(rewards)
    rets = np.empty_like(rewards)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.oracle_gae(p(rewards), p(values), p(dones), p(last_values), p(last_dones), p(gamma),
                   ctypes.c_double(lam), T, E, R, p(advs), p(rets))
    return advs, rets


def discount(rewards, dones, last_values, gamma):
    lib = load()
    rewards = np.ascontiguousarray(rewards, np.float32)
    E, T = rewards.shape
    dones = np.ascontiguousarray(np.asarray(dones).astype(np.uint8))
    last_values = np.ascontiguousarray(last_values, np.float32)
    out = np.empty_like(rewards)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.oracle_discount(p(rewards), p(dones), p(last_values), ctypes.c_double(gamma), E, T, p(out))
    return out
