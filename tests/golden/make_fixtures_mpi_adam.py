"""Golden vectors for Adam and its data-parallel exchange, recorded by RUNNING the reference's own
``MpiAdam.update`` / ``sync`` / ``check_synced`` (baselines/common/mpi_adam.py:21-50), unmodified, in the build
container.  TensorFlow and mpi4py are satisfied by the stub modules of make_fixtures.py; the optimiser object is
made with ``object.__new__`` (its constructor needs TF variables) and given NumPy ``getflat`` / ``setfromflat``
and a fake communicator that plays the other ranks' recorded gradients:

    Allreduce(local, global, op) -> global[:] = local + sum of the other ranks' gradients of this step
    Bcast(buf, root=0)           -> non-root ranks receive rank 0's parameters
    Get_size / Get_rank

Cases: 10 updates of a 13-element vector (3 + 2*5, the shape of the reference's own test_MpiAdam, :57-58) and of a
1000-element one, with 1 and 2 ranks, epsilon 1e-8 (MpiAdam's default) and 1e-5 (what PPO2 passes to
tf.train.AdamOptimizer, baselines/ppo2/common_functions.py:51-52), scale_grad_by_procs on and off.

NumPy note: ``a = stepsize * np.sqrt(...)`` (:31) is a float64 NumPy scalar; under NumPy >= 2 it promotes ``step``
(:34) to float64 before the final float32 store, under the NumPy of the reference's time it stayed float32.  The
vectors below are what THIS NumPy (version recorded in the file) computes; consumers compare within float32
rounding of one update (a few 1e-8 relative to the step), not bit for bit.

Usage:  python tests/golden/make_fixtures_mpi_adam.py      (writes tests/golden/reference_mpi_adam.npz)
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_fixtures import REF, _Stub  # noqa: E402


class FakeComm:
    """One rank's view of a job whose other ranks' per-step gradients are known in advance."""

    def __init__(self, rank, size, others, root_theta):
        self.rank, self.size, self.others, self.root_theta = rank, size, others, root_theta
        self.step = 0
        self.bcasts = 0

    def Get_size(self):
        return self.size

    def Get_rank(self):
        return self.rank

    def Allreduce(self, local, out, op=None):
        out[:] = local
        for g in self.others[self.step]:
            out += g
        self.step += 1

    def Bcast(self, buf, root=0):
        self.bcasts += 1
        if self.rank != root:
            buf[:] = self.root_theta()


def make_adam(mod, theta, comm, epsilon, scale):
    adam = object.__new__(mod.MpiAdam)
    adam.var_list = None
    adam.beta1, adam.beta2, adam.epsilon, adam.scale_grad_by_procs = 0.9, 0.999, epsilon, scale
    adam.m = np.zeros(theta.size, 'float32')
    adam.v = np.zeros(theta.size, 'float32')
    adam.t = 0
    store = {"theta": theta.astype(np.float32).copy()}
    adam.getflat = lambda: store["theta"].copy()
    adam.setfromflat = lambda x: store.__setitem__("theta", np.asarray(x).astype(np.float32))   # tf.assign casts
    adam.comm = comm
    adam._store = store
    return adam


def main():
    for m in ["tensorflow", "gym", "gym.spaces", "mpi4py", "mpi4py.MPI"]:
        sys.modules.setdefault(m, _Stub(m))
    sys.path.insert(0, REF)
    mod = importlib.import_module("baselines.common.mpi_adam")
    out = {"numpy_version": np.array(np.__version__)}
    rng = np.random.RandomState(5)
    nsteps = 10
    names = []
    for P in (13, 1000):
        for size in (1, 2):
            for epsilon in (1e-8, 1e-5):
                for scale in (True, False):
                    name = "P%d_n%d_eps%g_%s" % (P, size, epsilon, "mean" if scale else "sum")
                    names.append(name)
                    theta0 = rng.randn(P).astype(np.float32)
                    grads = (rng.randn(nsteps, size, P) * np.exp(rng.randn(nsteps, 1, 1))).astype(np.float32)
                    stepsizes = (3e-4 * (1 - np.arange(nsteps) / (2.0 * nsteps))).astype(np.float64)
                    adams = []
                    snap = {}       # rank 0's parameters at the start of the step, as a collective Bcast would deliver them
                    for r in range(size):
                        others = [[grads[t, q] for q in range(size) if q != r] for t in range(nsteps)]
                        comm = FakeComm(r, size, others, lambda: snap["root"])
                        adams.append(make_adam(mod, theta0, comm, epsilon, scale))
                    thetas, ms, vs = [], [], []
                    for t in range(nsteps):
                        snap["root"] = adams[0]._store["theta"].copy()
                        for r in range(size):      # ranks run in turn; non-root ranks compare against rank 0 (:42-50)
                            adams[r].update(grads[t, r], float(stepsizes[t]))
                        for r in range(1, size):   # replicas stay identical, as check_synced demands
                            assert np.array_equal(adams[r]._store["theta"], adams[0]._store["theta"])
                        thetas.append(adams[0]._store["theta"].copy())
                        ms.append(np.asarray(adams[0].m).copy())
                        vs.append(np.asarray(adams[0].v).copy())
                    assert adams[0].comm.bcasts == 1        # check_synced ran at t = 0 only (t % 100, :22)
                    out[name + "_theta0"] = theta0
                    out[name + "_grads"] = grads
                    out[name + "_stepsizes"] = stepsizes
                    out[name + "_theta"] = np.stack(thetas)
                    out[name + "_m"] = np.stack(ms)
                    out[name + "_v"] = np.stack(vs)
                    out[name + "_m_dtype"] = np.array(str(np.asarray(adams[0].m).dtype))
    out["cases"] = np.array(names)
    # sync(): rank 1 starts from other parameters and receives rank 0's (:37-40); check_synced then passes, and fails
    # with an AssertionError when a replica has drifted (:50)
    root = rng.randn(13).astype(np.float32)
    holder = {"root": root}
    a1 = make_adam(mod, rng.randn(13).astype(np.float32), FakeComm(1, 2, [], lambda: holder["root"]), 1e-8, True)
    before = a1._store["theta"].copy()
    a1.sync()
    assert np.array_equal(a1._store["theta"], root) and not np.array_equal(before, root)
    a1.check_synced()
    a1._store["theta"][3] += 1e-3
    try:
        a1.check_synced()
        raised = False
    except AssertionError:
        raised = True
    out["sync_root"], out["sync_before"], out["sync_after"] = root, before, root.copy()
    out["check_synced_raises_on_drift"] = np.array(raised)
    path = os.path.join(HERE, "reference_mpi_adam.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays; numpy", np.__version__)


if __name__ == "__main__":
    main()
