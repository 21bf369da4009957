"""The advantage step of baselines/ppo1/pposgd_simple.py on the rollout scan kernel."""
import numpy as np
import torch

from .. import hip


def add_vtarg_and_adv(seg, gamma, lam):
    """TD(lambda) value targets and GAE(lambda) advantages of one actor's segment, written into
    ``seg["adv"]`` and ``seg["tdlamret"]`` (baselines/ppo1/pposgd_simple.py:64-78; the same
    function is baselines/trpo_mpi/trpo_mpi.py:69-80).

    The segment is the [T, 1, 1] case of the time-major scan: ``seg["new"][t]`` is the flag seen
    BEFORE step t, and ``seg["nextvpred"]`` already carries ``vpred * (1 - new)``, so the
    bootstrap value is used as it is.  The carry runs in float64 as with NumPy's value-based
    scalar promotion (Python-float ``gamma`` times a float32 element), the stores are float32."""
    rew = np.ascontiguousarray(seg["rew"], np.float32)
    vpred = np.ascontiguousarray(seg["vpred"], np.float32)
    T = rew.shape[0]
    if T == 0:
        seg["adv"] = np.empty(0, np.float32)
        seg["tdlamret"] = seg["adv"] + vpred
        return
    dev = lambda a: torch.from_numpy(a).cuda()
    new = dev(np.ascontiguousarray(np.asarray(seg["new"]) != 0).astype(np.uint8).reshape(T, 1))
    last_v = dev(np.asarray(seg["nextvpred"], np.float32).reshape(1, 1))
    last_d = torch.zeros(1, dtype=torch.uint8, device="cuda")
    advs, rets, _ = hip.gae(dev(rew.reshape(T, 1, 1)), dev(vpred.reshape(T, 1, 1)), new, last_v, last_d,
                            float(gamma), float(lam), want_advs_ori=False)
    seg["adv"] = advs.cpu().numpy().reshape(T)
    seg["tdlamret"] = rets.cpu().numpy().reshape(T)
