"""The product's data-parallel update on the GPU: two ranks (sharing the one GPU of the test
box, gradients exchanged over gloo staged through the host — RCCL needs one GPU per rank)
must reproduce the single-process update when they shard the envs and draw the same global
permutation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import tfgraph
from tests import util

pytestmark = pytest.mark.gpu

D, A, T, E, B, NEP = 17, 6, 8, 32, 64, 2


def _data():
    shapes = tfgraph.mlp_param_shapes(D, (64, 64), A, 1, True, "upstream")
    params = util.make_params(shapes, 41, pi_scale=0.5, jitter=0.02)
    ro = util.synthetic_mlp_rollout(np.random.RandomState(42), T, E, D, A, 1, True, params, nlp_noise=0.2)
    return params, ro


def _run_model(params, ro, e0, e1, comm, mode_seed):
    from baselines_amd import hip, spaces
    from baselines_amd.ppo2 import policies
    from baselines_amd.ppo2.common_functions import DeviceRollout, Model
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    policies.reset_scopes()
    model = Model(policy=policies.MlpPolicy64, ob_space=spaces.Box(-1, 1, shape=(D,)), ac_space=spaces.Box(-1, 1, shape=(A,)),
                  nbatch_act=e1 - e0, nbatch_train=B, nsteps=T, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, comm=comm)
    model.arena.set_dict(params)
    sl = lambda a: a[:, e0:e1] if a.ndim >= 2 else a[e0:e1]
    d = DeviceRollout(T, e1 - e0, dev(sl(ro["obs"])), dev(sl(ro["actions"])), dev(sl(ro["rewards"])), dev(sl(ro["values"])),
                      dev(sl(ro["neglogpacs"])), dev(sl(ro["dones"])), dev(ro["last_values"][e0:e1]), dev(ro["last_dones"][e0:e1]))
    d.advs_gae, d.returns, d.advs_ori = hip.gae(d.rewards, d.values, d.dones, d.last_values, d.last_dones, 0.99, 0.95)
    d.advs = d.advs_ori.reshape(T, e1 - e0)
    np.random.seed(mode_seed)
    losses = model.train_on_rollout(3e-4, 0.2, d, NEP, B).cpu().numpy()
    theta = model.arena.theta.cpu().numpy()
    return losses, theta


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from baselines_amd.common.mpi_adam import DataParallel
    comm = DataParallel("global")
    params, ro = _data()
    e0, e1 = comm.env_range(E)
    losses, theta = _run_model(params, ro, e0, e1, comm, 5)
    assert comm.check_synced(torch.from_numpy(theta).cuda())
    if rank == 0:
        out.put((losses, theta))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_single_process_update():
    assert torch.cuda.is_available()
    params, ro = _data()
    l1, t1 = _run_model(params, ro, 0, E, None, 5)
    l2, t2 = util.run_ranks(_worker, 2)
    np.testing.assert_allclose(l2, l1, rtol=5e-5, atol=1e-7)
    moved = np.abs(t1 - util.flatten_params(params, type("S", (), {"nparams": t1.size, "vars": _vars()})())).max()
    assert np.abs(t1 - t2).max() <= 2e-6, np.abs(t1 - t2).max()
    assert moved > 1e-4


def _vars():
    from baselines_amd import _lib, hip
    return hip.PolicySpec(_lib.POLICY_MLP, _lib.DIST_GAUSSIAN, A, 1, ob_dim=D, hidden=(64, 64), order=_lib.ORDER_UPSTREAM).vars


# ---- running statistics exchanged over the data-parallel group -----------------------------------------
def _rms_data(rank):
    rng = np.random.RandomState(0)
    p = [rng.randn(3, 2), rng.randn(4, 2), rng.randn(5, 2)]
    q = [rng.randn(6, 2) * 3 + 1, rng.randn(7, 2) * 3 + 1, rng.randn(8, 2) * 3 + 1]
    return (p, q)[rank], p + q


def _rms_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from baselines_amd.common.mpi_adam import DataParallel
    from baselines_amd.common.mpi_running_mean_std import RunningMeanStd
    rms = RunningMeanStd(epsilon=0.0, shape=(2,), comm=DataParallel("mpi"))
    mine, _ = _rms_data(rank)
    for x in mine:
        rms.update(x)
    if rank == 0:
        out.put((rms.mean, rms.std, rms.state.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_mpi_running_mean_std_two_ranks():
    """The reference's own distributed check (baselines/common/mpi_running_mean_std.py:85-107): two ranks fed
    different data end with the statistics of the concatenated data."""
    from oracle import envwrap
    mean, std, state = util.run_ranks(_rms_worker, 2)
    _, everything = _rms_data(0)
    allx = np.concatenate(everything)
    # the reference compares with allclose against mean / std of the whole; std floors its variance at 1e-2
    assert np.allclose(mean, allx.mean(0)) and np.allclose(std, np.sqrt(np.maximum(allx.var(0), 1e-2)), atol=1e-6)
    o = envwrap.MpiRunningMeanStd(epsilon=0.0, shape=(2,))
    o.update(allx)
    np.testing.assert_allclose(state, np.concatenate([o._sum, o._sumsq, [o._count]]), rtol=1e-13)


def test_mpi_running_mean_std_single_process():
    """:46-63 of the same file: three updates equal one pass; totals against the oracle to float64 rounding,
    mean / std (float32 in the reference's graph) exactly where the totals agree."""
    from baselines_amd.common.mpi_running_mean_std import RunningMeanStd
    from oracle import envwrap
    rng = np.random.RandomState(1)
    for shape in ((), (2,), (5, 3)):
        xs = [(rng.randn(*((n,) + shape)) * 2 + 0.5).astype(np.float32 if n != 4 else np.float64) for n in (3, 4, 500)]
        rms, o = RunningMeanStd(shape=shape), envwrap.MpiRunningMeanStd(shape=shape)
        for x in xs:
            rms.update(x)
            o.update(x)
        st = rms.state.cpu().numpy()
        np.testing.assert_allclose(st, np.concatenate([o._sum.ravel(), o._sumsq.ravel(), [o._count]]), rtol=1e-13)
        np.testing.assert_allclose(rms.mean, o.mean, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rms.std, o.std, rtol=1e-6)
        assert rms.mean.dtype == np.float32 and rms.mean.shape == shape
