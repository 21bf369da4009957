"""The data-parallel exchange of the C ABI on the GPU.

  * The RCCL binding inside libppo2hip (unique id, communicator, SUM all-reduce of float32 / float64, broadcast, the
    replica digest) on a one-rank communicator — RCCL refuses two ranks on one GPU, so this is how its path runs on a
    one-GPU box — and ppo2hip_ppo_epoch_dp against ppo2hip_ppo_epoch, bit for bit.
  * The product's MpiAdam class against the vectors the reference's own MpiAdam.update produced
    (tests/golden/reference_mpi_adam.npz), two ranks played by a communicator that adds the other rank's recorded
    gradient.
  * Two ranks sharing this box's GPU in "mpi" mode (per-rank shuffles and normalisation, SUM all-reduce, 1/size,
    baselines/common/mpi_adam.py:24-28; gradients staged through the host over gloo) against the oracle run the same way,
    for the MLP and for cnn + LSTM with env-sequence minibatches (baselines/ppo2/ppo2.py:124-137).
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import perm, scan, tfgraph
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, A, T, E, B, NEP = 17, 6, 8, 32, 64, 2


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def test_rccl_one_rank_communicator():
    from baselines_amd import hip
    uid = hip.NativeComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = hip.NativeComm(uid, 0, 1)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    x = torch.randn(11085, generator=g, device="cuda")
    keep = x.clone()
    comm.allreduce_sum_(x)                       # the sum over one rank
    d = torch.randn(12, generator=g, device="cuda", dtype=torch.float64)
    keepd = d.clone()
    comm.allreduce_sum_(d)
    comm.bcast_(x, 0)
    flag = comm.replica_checksum(x)
    torch.cuda.synchronize()
    assert torch.equal(x, keep) and torch.equal(d, keepd) and int(flag.item()) == 1
    # a communicator without an id touches no collective library and answers the same
    local = hip.NativeComm(None, 0, 1)
    local.allreduce_sum_(x)
    assert int(local.replica_checksum(x).item()) == 1 and torch.equal(x, keep)
    with pytest.raises(Exception):
        hip.NativeComm(None, 0, 2)              # several ranks need rank 0's id


def _model_and_rollout(comm=None):
    from baselines_amd import hip, spaces
    from baselines_amd.ppo2 import policies
    from baselines_amd.ppo2.common_functions import DeviceRollout, Model
    shapes = tfgraph.mlp_param_shapes(D, (64, 64), A, 1, True, "upstream")
    params = util.make_params(shapes, 41, pi_scale=0.5, jitter=0.02)
    ro = util.synthetic_mlp_rollout(np.random.RandomState(42), T, E, D, A, 1, True, params, nlp_noise=0.2)
    policies.reset_scopes()
    model = Model(policy=policies.MlpPolicy64, ob_space=spaces.Box(-1, 1, shape=(D,)), ac_space=spaces.Box(-1, 1, shape=(A,)),
                  nbatch_act=E, nbatch_train=B, nsteps=T, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, comm=comm)
    model.arena.set_dict(params)
    d = DeviceRollout(T, E, dev(ro["obs"]), dev(ro["actions"]), dev(ro["rewards"]), dev(ro["values"]),
                      dev(ro["neglogpacs"]), dev(ro["dones"]), dev(ro["last_values"]), dev(ro["last_dones"]))
    d.advs_gae, d.returns, d.advs_ori = hip.gae(d.rewards, d.values, d.dones, d.last_values, d.last_dones, 0.99, 0.95)
    d.advs = d.advs_ori.reshape(T, E)
    return model, d, params, ro


def test_epoch_dp_through_rccl_equals_epoch():
    from baselines_amd import hip
    results = []
    for use_comm in (False, True):
        model, d, _, _ = _model_and_rollout()
        rref = hip.RolloutRef(model.spec, d.obs, d.actions, d.returns, d.values, d.neglogpacs, d.advs, T, E)
        idx = dev(np.random.RandomState(3).permutation(T * E).astype(np.int32))
        nmb = T * E // B
        stats = hip.adv_stats(d.advs, idx, B, nmb, T, E)
        losses = torch.zeros((nmb, 8), device="cuda")
        comm = hip.NativeComm(hip.NativeComm.unique_id(), 0, 1) if use_comm else None
        hip.ppo_epoch(model.spec, model.loss_desc, model.arena.theta, model.grads, model.adam_m, model.adam_v, rref, idx, B,
                      stats, 0.2, 3e-4, 1, losses, model._workspace(B), model._adam_scratch, max_grad_norm=0.5, comm=comm,
                      grad_scale=1.0)
        torch.cuda.synchronize()
        results.append((model.arena.theta.clone(), losses.clone(), model.adam_v.clone()))
    for a, b in zip(*results):
        assert torch.equal(a, b)


class _TwoRankComm:
    """Rank 0 of a two-rank job whose other rank's gradients are on record (the product's communicator interface)."""

    def __init__(self, others):
        self.size, self.rank, self.others, self.step, self.checks = 2, 0, others, 0, 0

    def allreduce_sum_(self, t):
        t += self.others[self.step]
        self.step += 1
        return t

    def bcast_(self, t, root=0):
        return t

    def check_synced(self, theta, wait=True):
        self.checks += 1
        return True


def test_product_mpi_adam_matches_the_reference_fixture():
    from baselines_amd.common.mpi_adam import MpiAdam
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_mpi_adam.npz"), allow_pickle=False)
    for name in z["cases"]:
        size = int(name.split("_")[1][1:])
        eps = float(name.split("_")[2][3:])
        mean = name.endswith("_mean")
        grads = z[name + "_grads"]
        comm = _TwoRankComm([dev(grads[t, 1]) for t in range(grads.shape[0])]) if size == 2 else None
        theta = dev(z[name + "_theta0"].copy())
        adam = MpiAdam(theta, epsilon=eps, scale_grad_by_procs=mean, comm=comm)
        for t, lr in enumerate(z[name + "_stepsizes"]):
            adam.update(dev(grads[t, 0]), float(lr))
            want = z[name + "_theta"][t]
            err = float(np.abs(theta.cpu().numpy() - want).max())
            assert err <= 3e-7 * max(1.0, np.abs(want).max() / 2), (name, t, err)
        if comm is not None:
            assert comm.checks == 1 and comm.step == 10         # check_synced at t = 0 only (t % 100, mpi_adam.py:22)
        wm = z[name + "_m"][-1]
        np.testing.assert_allclose(adam.m.cpu().numpy(), wm, rtol=2e-6, atol=1e-6 * np.abs(wm).max())


def _mpi_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from baselines_amd import hip
    from baselines_amd.common.mpi_adam import DataParallel
    from baselines_amd.ppo2.common_functions import DeviceRollout
    comm = DataParallel("mpi")
    assert comm.transport == "host" and comm.native() is None
    model, d_all, params, ro = _model_and_rollout(comm)
    e0, e1 = comm.env_range(E)
    Es = e1 - e0
    sl = lambda a: a[:, e0:e1].contiguous()
    d = DeviceRollout(T, Es, sl(d_all.obs), sl(d_all.actions), sl(d_all.rewards), sl(d_all.values), sl(d_all.neglogpacs),
                      sl(d_all.dones), d_all.last_values[e0:e1].contiguous(), d_all.last_dones[e0:e1].contiguous())
    d.returns, d.advs = sl(d_all.returns), d_all.advs[:, e0:e1].contiguous()
    np.random.seed(5)                       # every rank shuffles its own shard (here: from the same seed)
    losses = model.train_on_rollout(3e-4, 0.2, d, NEP, B // 2).cpu().numpy()
    theta = model.arena.theta
    assert comm.check_synced(theta)
    if rank == 0:
        out.put((losses, theta.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_mpi_mode_matches_the_oracle():
    losses, theta = util.run_ranks(_mpi_worker, 2)
    shapes = tfgraph.mlp_param_shapes(D, (64, 64), A, 1, True, "upstream")
    params = util.make_params(shapes, 41, pi_scale=0.5, jitter=0.02)
    ro = util.synthetic_mlp_rollout(np.random.RandomState(42), T, E, D, A, 1, True, params, nlp_noise=0.2)
    _, rets = scan.gae(ro["rewards"], ro["values"], ro["dones"], ro["last_values"], ro["last_dones"], 0.99, 0.95)
    Es, Bs = E // 2, B // 2
    shards = []
    for r in range(2):
        c = slice(r * Es, (r + 1) * Es)
        flat = {k: scan.sf01(v[:, c]) for k, v in dict(obs=ro["obs"], ret=rets, act=ro["actions"], val=ro["values"],
                                                       nlp=ro["neglogpacs"]).items()}
        flat["adv"] = (flat["ret"] - flat["val"])[:, 0]
        shards.append(flat)
    oracle = tfgraph.TrainOracle(params, "mlp", "gaussian", tfgraph.LossCfg(0.0, 0.5, 0.5, True))
    plist = list(oracle.p.values())
    np.random.seed(5)
    rows = []
    for ep in perm.minibatch_indices(T * Es, Bs, NEP):
        for mb in ep:
            gs, ls = [], []
            for f in shards:        # both ranks draw the same local indices; each evaluates its own rows at the shared parameters
                ls.append(oracle.train(3e-4, 0.2, f["obs"][mb], f["ret"][mb], None, f["act"][mb], f["val"][mb], f["adv"][mb],
                                       f["nlp"][mb], apply=False))
                gs.append([g.clone() for g in oracle.last_grads.values()])
            mean_g = [(a + b) / 2 for a, b in zip(*gs)]                      # Allreduce(SUM); /= size  (mpi_adam.py:26-28)
            clipped, _ = tfgraph.clip_by_global_norm(mean_g, 0.5)
            oracle.adam.update(plist, clipped, 3e-4)
            rows.append(np.mean(ls, axis=0))
    want = np.concatenate([p.detach().numpy().ravel() for p in plist])
    np.testing.assert_allclose(losses, np.mean(rows, axis=0), rtol=5e-5, atol=1e-7)
    diff = float(np.abs(theta - want).max())
    print("two ranks, mpi mode: parameter diff vs oracle %.3e" % diff)
    assert diff <= 2e-6


# ---- cnn + LSTM, env-sequence minibatches, two ranks in "mpi" mode --------------------------------------
R_HWC, R_NACT, R_T, R_E, R_NLSTM = (36, 36, 2), 4, 6, 8, 32


def _recurrent_data():
    shapes = tfgraph.cnn_param_shapes(R_HWC[2], R_NACT, "lstm", R_NLSTM, flat=util.cnn_flat(R_HWC[0], R_HWC[1]))
    params = util.make_params(shapes, 51, pi_scale=0.5, jitter=0.002)
    rng = np.random.RandomState(52)
    ro = util.synthetic_cnn_rollout(rng, R_T, R_E, R_HWC, R_NACT, params, kind="lstm", nlstm=R_NLSTM)
    ro["states"] = (0.2 * rng.randn(R_E, 2 * R_NLSTM)).astype(np.float32)
    return params, ro


def _mpi_recurrent_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from baselines_amd import hip, spaces
    from baselines_amd.common.mpi_adam import DataParallel
    from baselines_amd.ppo2 import policies
    from baselines_amd.ppo2.common_functions import DeviceRollout, Model
    comm = DataParallel("mpi")
    params, ro = _recurrent_data()
    e0, e1 = comm.env_range(R_E)
    Es = e1 - e0
    nbt = (Es // 2) * R_T

    class SmallLstm(policies.LstmPolicy):
        def __init__(self, *a, **k):
            super().__init__(*a, nlstm=R_NLSTM, **k)

    policies.reset_scopes()
    model = Model(policy=SmallLstm, ob_space=spaces.Box(0, 255, shape=R_HWC, dtype=np.uint8), ac_space=spaces.Discrete(R_NACT),
                  nbatch_act=Es, nbatch_train=nbt, nsteps=R_T, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, upstream_loss=True,
                  comm=comm)
    model.arena.set_dict(params)
    sl = lambda a: dev(a[:, e0:e1])
    d = DeviceRollout(R_T, Es, sl(ro["obs"]), sl(ro["actions"]), sl(ro["rewards"]), sl(ro["values"]), sl(ro["neglogpacs"]),
                      sl(ro["dones"]), dev(ro["last_values"][e0:e1]), dev(ro["last_dones"][e0:e1]), states=dev(ro["states"][e0:e1]))
    _, d.returns, d.advs_ori = hip.gae(d.rewards, d.values, d.dones, d.last_values, d.last_dones, 0.99, 0.95, want_advs=False)
    d.advs = d.advs_ori.reshape(R_T, Es)
    np.random.seed(5)                       # every rank shuffles its own envs (here: from the same seed)
    losses = model.train_on_rollout(2.5e-4, 0.1, d, NEP, nbt).cpu().numpy()
    model.check_faults()
    assert comm.check_synced(model.arena.theta)
    if rank == 0:
        out.put((losses, model.arena.theta.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_mpi_mode_recurrent_matches_the_oracle():
    """cnn + LSTM over two ranks: each rank runs the env-sequence minibatch loop of baselines/ppo2/ppo2.py:124-137 on its own
    envs; per minibatch the gradients are summed over the ranks and divided by their number (MpiAdam.update,
    baselines/common/mpi_adam.py:24-28) before clip + Adam.  The oracle evaluates both ranks' minibatches at the shared
    parameters and applies the mean gradient."""
    losses, theta = util.run_ranks(_mpi_recurrent_worker, 2)
    params, ro = _recurrent_data()
    _, rets = scan.gae(ro["rewards"], ro["values"], ro["dones"], ro["last_values"], ro["last_dones"], 0.99, 0.95)
    Es = R_E // 2
    nbt = (Es // 2) * R_T
    shards = []
    for r in range(2):
        c = slice(r * Es, (r + 1) * Es)
        flat = {k: scan.sf01(v[:, c]) for k, v in dict(obs=ro["obs"], ret=rets, act=ro["actions"], val=ro["values"],
                                                       nlp=ro["neglogpacs"], msk=ro["dones"]).items()}
        flat["adv"] = (flat["ret"] - flat["val"])[:, 0]
        flat["states"] = ro["states"][c]
        shards.append(flat)
    oracle = tfgraph.TrainOracle(params, "lstm", "categorical", tfgraph.LossCfg(0.01, 0.5, 0.5, True))
    plist = list(oracle.p.values())
    np.random.seed(5)
    rows = []
    for ep in perm.recurrent_minibatch_indices(Es, R_T, nbt, NEP):
        for envs_mb, fl in ep:
            gs, ls = [], []
            for f in shards:        # both ranks draw the same local env order; each evaluates its own envs at the shared parameters
                ls.append(oracle.train(2.5e-4, 0.1, f["obs"][fl], f["ret"][fl], f["msk"][fl], f["act"][fl], f["val"][fl],
                                       f["adv"][fl], f["nlp"][fl], states=f["states"][envs_mb], apply=False))
                gs.append([g.clone() for g in oracle.last_grads.values()])
            mean_g = [(a + b) / 2 for a, b in zip(*gs)]                      # Allreduce(SUM); /= size  (mpi_adam.py:26-28)
            clipped, _ = tfgraph.clip_by_global_norm(mean_g, 0.5)
            oracle.adam.update(plist, clipped, 2.5e-4)
            rows.append(np.mean(ls, axis=0))
    want = np.concatenate([p.detach().numpy().ravel() for p in plist])
    np.testing.assert_allclose(losses, np.mean(rows, axis=0), rtol=1e-4, atol=1e-7)
    diff = float(np.abs(theta - want).max())
    print("two ranks, mpi mode, cnn + LSTM: parameter diff vs oracle %.3e" % diff)
    assert diff <= 1e-5
