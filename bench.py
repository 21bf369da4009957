#!/usr/bin/env python
"""PPO2 samples/sec (GAE + update) on synthetic device-resident rollouts.

One "step" = one pass of the rollout-to-update path over one synthetic rollout batch:
GAE/returns, advantage preparation, then noptepochs x (N / nbatch_train) minibatch
updates (bit-exact index generation, gather, normalisation, forward, loss, backward,
clip + Adam, and the gradient all-reduce when N > 1 GPUs).  Environment time is excluded
(inputs are resident in HBM before the timed region), as the reference's `fps`
(baselines/ppo2/ppo2.py:141) minus the rollout.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32-input MFMA dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # bf16-input MFMA dense peak

WORKLOADS = {
    # BASELINE.json configs[1]: synthetic MuJoCo-shaped rollout, Gaussian MLP(64,64)
    "mujoco_mlp64": dict(D=17, A=6, E=2048, T=256, hidden=(64, 64), R=1, upstream=True, nmb=4, noptepochs=4,
                         lr=3e-4, cliprange=0.2, ent_coef=0.0, done_p=1.0 / 200),
    # BASELINE.json configs[3]: one GPU's shard (8192 envs) of the 65536-env x 128-step env-sharded MLP rollout;
    # run with --gpus 8 for the configuration itself
    "mlp_env_shard": dict(D=17, A=6, E=8192, T=128, hidden=(64, 64), R=1, upstream=True, nmb=4, noptepochs=4,
                          lr=3e-4, cliprange=0.2, ent_coef=0.0, done_p=1.0 / 200),
    # the same shape through the fork's own network/loss (256-128, 5 value channels)
    "mujoco_fork": dict(D=17, A=6, E=2048, T=256, hidden=(256, 128), R=5, upstream=False, nmb=4, noptepochs=4,
                        lr=3e-4, cliprange=0.2, ent_coef=0.0, done_p=1.0 / 200),
    # BASELINE.json configs[2]: synthetic Atari-shaped rollout, nature_cnn, Categorical(6)
    "atari_cnn": dict(hwc=(84, 84, 4), A=6, E=256, T=128, R=1, upstream=True, nmb=4, noptepochs=4, lr=2.5e-4,
                      cliprange=0.1, ent_coef=0.01, done_p=1.0 / 1000),
    # BASELINE.json configs[4]: cnn_lstm, env-sequence minibatches of E/4 envs x T steps
    "atari_lstm": dict(hwc=(84, 84, 4), A=6, E=512, T=128, R=1, upstream=True, nmb=4, noptepochs=4, lr=2.5e-4,
                       cliprange=0.1, ent_coef=0.01, done_p=1.0 / 1000, nlstm=256),
}


def dense_flops_per_sample(D, hidden, A, R):
    """fwd 2*sum(in*out) + dW (same) + dX (same minus first layer), both trunks."""
    h1, h2 = hidden
    tot = 0
    for out in (A, R):
        fwd = 2 * (D * h1 + h1 * h2 + h2 * out)
        dx = 2 * (h1 * h2 + h2 * out)
        tot += fwd + fwd + dx
    return tot


def measured_traffic(workload, kernels):
    """HBM bytes per launch of `kernels` (one name or several that make up a call: their sum) from the committed PMC
    passes (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 runs of this script), or None when no profile of this
    workload is on file."""
    if isinstance(kernels, (tuple, list)):
        parts = [measured_traffic(workload, k) for k in kernels]
        return None if any(x is None for x in parts) else sum(parts)
    kernel = kernels
    path = os.path.join(ROOT, "profiles", "r03_%s_hbm_traffic.json" % workload)
    try:
        with open(path) as fh:
            prof = json.load(fh)
    except OSError:
        return None
    k = prof.get("kernels", {}).get(kernel)
    if prof.get("workload") != workload or not k or "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
        return None
    return k["FETCH_SIZE"] + k["WRITE_SIZE"]


def cnn_flops_per_sample(hwc, nact, nlstm=0):
    """nature_cnn on HxWxC (+ LSTM(nlstm) on its 512 features): forward + weight gradients (same) +
    input gradients (same minus conv1)."""
    H, W, C = hwc
    o = lambda n, k, s: (n - k) // s + 1
    h1, w1 = o(H, 8, 4), o(W, 8, 4)
    h2, w2 = o(h1, 4, 2), o(w1, 4, 2)
    h3, w3 = o(h2, 3, 1), o(w2, 3, 1)
    conv1 = 2 * h1 * w1 * 32 * 8 * 8 * C
    feat = nlstm if nlstm else 512
    fwd = conv1 + 2 * h2 * w2 * 64 * 4 * 4 * 32 + 2 * h3 * w3 * 64 * 3 * 3 * 64 + 2 * h3 * w3 * 64 * 512 + 2 * feat * (nact + 1)
    if nlstm:
        fwd += 2 * (512 + nlstm) * 4 * nlstm
    return 3 * fwd - conv1


def lstm_recurrent_flops_per_sample(nlstm):
    """The part of cnn_flops_per_sample that stays on the float32 pipe: the h . Wh products inside the whole-sequence
    kernels, forward and input gradient (their weight gradient is one wide product over all steps and runs as pieces)."""
    return 2 * (2 * nlstm * 4 * nlstm)


def cnn_conv1_flops_per_sample(hwc):
    """The part of cnn_flops_per_sample that runs as bf16 products (three per float32 multiply-add: the uint8 operand is
    exact in bf16, the float32 one is cut into three exact bf16 pieces): conv1 forward and weight gradient."""
    H, W, C = hwc
    o = lambda n, k, s: (n - k) // s + 1
    return 2 * (2 * o(H, 8, 4) * o(W, 8, 4) * 32 * 8 * 8 * C)


def make_image_rollout(model, cfg, seed, device):
    """Synthetic Atari-shaped rollout in HBM: uniform uint8 frames, Gumbel-max actions, values and
    neglogpacs from the initial policy."""
    from baselines_amd import hip
    from baselines_amd.ppo2.common_functions import DeviceRollout
    T, E, A = cfg["T"], cfg["E"], cfg["A"]
    H, W, C = cfg["hwc"]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    obs = torch.randint(0, 256, (T, E, H, W, C), generator=g, device=device, dtype=torch.uint8)
    rewards = torch.randn((T, E, 1), generator=g, device=device)
    dones = (torch.rand((T, E), generator=g, device=device) < cfg["done_p"]).to(torch.uint8)
    last_dones = (torch.rand((E,), generator=g, device=device) < cfg["done_p"]).to(torch.uint8)
    spec, theta = model.spec, model.arena.theta
    ws = torch.empty(spec.forward_workspace_bytes(E), dtype=torch.uint8, device=device)
    actions = torch.empty((T, E), dtype=torch.int32, device=device)
    values = torch.empty((T, E, 1), device=device)
    nlp = torch.empty((T, E), device=device)
    nlstm = cfg.get("nlstm", 0)
    states0 = torch.zeros((E, 2 * nlstm), device=device) if nlstm else None
    state = states0
    for t in range(T):
        if nlstm:
            pi, vf, state = hip.policy_forward(spec, theta, obs[t], ws, masks=dones[t], states=state)
        else:
            pi, vf, _ = hip.policy_forward(spec, theta, obs[t], ws)
        u = torch.rand((E, A), generator=g, device=device).clamp_(1e-12, 1.0)
        a = torch.argmax(pi - torch.log(-torch.log(u)), dim=1).to(torch.int32)
        actions[t] = a
        values[t] = vf.reshape(E, 1)
        nlp[t] = hip.neglogp(spec, theta, pi, a)
    last_values = torch.randn((E, 1), generator=g, device=device)
    return DeviceRollout(T, E, obs, actions, rewards, values, nlp, dones, last_values, last_dones, states=states0)


def make_rollout(model, cfg, seed, device):
    """Synthetic rollout in HBM: N(0,1) observations and rewards, Bernoulli dones, actions /
    values / neglogpacs from the initial policy (so ratio ~ 1 at the first minibatch)."""
    from baselines_amd import hip
    from baselines_amd.ppo2.common_functions import DeviceRollout
    T, E, D, A, R = cfg["T"], cfg["E"], cfg["D"], cfg["A"], cfg["R"]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    obs = torch.randn((T, E, D), generator=g, device=device)
    rewards = torch.randn((T, E, R), generator=g, device=device)
    dones = (torch.rand((T, E), generator=g, device=device) < cfg["done_p"]).to(torch.uint8)
    last_dones = (torch.rand((E,), generator=g, device=device) < cfg["done_p"]).to(torch.uint8)
    spec, theta = model.spec, model.arena.theta
    ws = torch.empty(spec.forward_workspace_bytes(T * E), dtype=torch.uint8, device=device)
    pi, vf, _ = hip.policy_forward(spec, theta, obs.reshape(T * E, D), ws)
    eps = torch.randn((T * E, A), generator=g, device=device)
    actions = (pi + eps).contiguous()            # logstd = 0 at init
    nlp = hip.neglogp(spec, theta, pi, actions)
    last_values = torch.randn((E, R), generator=g, device=device)
    del ws
    return DeviceRollout(T, E, obs, actions.reshape(T, E, A), rewards, vf.reshape(T, E, R).contiguous(),
                         nlp.reshape(T, E), dones, last_values, last_dones)


def cpu_baseline(model, ro, cfg, budget_s=20.0):
    """Reference-structured CPU path on the host cores (oracle = CPU restatement, "port"):
    NumPy GAE loop over t, sf01, np.random.shuffle + fancy-index gathers, per-minibatch NumPy
    normalisation, and the TF step restated with torch-CPU float32 eager ops."""
    from oracle import scan, tfgraph
    T, E, R = cfg["T"], cfg["E"], cfg["R"]
    N = T * E
    host = {k: getattr(ro, k).cpu().numpy() for k in ("obs", "actions", "rewards", "values", "neglogpacs", "dones",
                                                       "last_values", "last_dones")}
    params = {name: arr for (name, _, _), arr in zip(model.spec.vars, model.arena.get_list())}
    image = "hwc" in cfg
    recurrent = bool(cfg.get("nlstm"))
    oracle = tfgraph.TrainOracle(params, ("lstm" if recurrent else "cnn") if image else "mlp", "categorical" if image else "gaussian",
                                 tfgraph.LossCfg(cfg["ent_coef"], 0.5, 0.5, cfg["upstream"]), dtype=torch.float32)
    B = N // cfg["nmb"]
    np.random.seed(0)

    def one_update():
        advs, rets = scan.gae(host["rewards"], host["values"], host["dones"].astype(bool), host["last_values"],
                              host["last_dones"].astype(bool), 0.99, 0.95)
        obs, returns, actions, values, nlps = (scan.sf01(a) for a in (host["obs"], rets, host["actions"],
                                                                      host["values"], host["neglogpacs"]))
        adv = (returns - values).sum(axis=1)
        if recurrent:                 # env-sequence minibatches, baselines/ppo2/ppo2.py:124-136
            masks = scan.sf01(host["dones"])
            states = ro.states.cpu().numpy()
            envsperbatch = E // cfg["nmb"]
            envinds = np.arange(E)
            flatinds = np.arange(N).reshape(E, T)
            for _ in range(cfg["noptepochs"]):
                np.random.shuffle(envinds)
                for start in range(0, E, envsperbatch):
                    mbenv = envinds[start:start + envsperbatch]
                    mb = flatinds[mbenv].ravel()
                    oracle.train(cfg["lr"], cfg["cliprange"], obs[mb], returns[mb], masks[mb], actions[mb], values[mb],
                                 adv[mb], nlps[mb], states=states[mbenv])
            return
        inds = np.arange(N)
        for _ in range(cfg["noptepochs"]):
            np.random.shuffle(inds)
            for start in range(0, N, B):
                mb = inds[start:start + B]
                oracle.train(cfg["lr"], cfg["cliprange"], obs[mb], returns[mb], None, actions[mb], values[mb], adv[mb],
                             nlps[mb])

    t0 = time.perf_counter()
    one_update()                      # warm-up, also sizes the sample
    t_first = time.perf_counter() - t0
    if t_first > budget_s:            # one update already spends the budget: its own time stands (first-call costs included)
        reps, dt = 1, t_first
    else:
        reps = int(max(1, min(5, budget_s // max(t_first, 1e-3) - 1)))
        t0 = time.perf_counter()
        for _ in range(reps):
            one_update()
        dt = (time.perf_counter() - t0) / reps
    cores = torch.get_num_threads()
    # the same update on ONE thread (bounded: one update, skipped when the multi-thread figure says it would take minutes)
    one_thread = None
    if dt * cores < 90.0:
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        one_update()
        one_thread = N / (time.perf_counter() - t0)
        torch.set_num_threads(cores)
    model_name = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model_name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(value=N / dt, unit="samples/s", cores=cores, kind="port", cpu_model=model_name,
                value_one_thread=one_thread,
                sample="%d full updates of the same workload (N=%d, %d minibatches x %d epochs), float32 torch-CPU + NumPy; "
                       "value_one_thread: one such update with torch.set_num_threads(1)"
                       % (reps, N, cfg["nmb"], cfg["noptepochs"]))


def build_workload(cfg, device, comm, rank):
    """Model, synthetic rollout in HBM and the step function (one pass of the hot path over it) of one workload."""
    from baselines_amd import hip, spaces
    from baselines_amd.ppo2 import policies
    from baselines_amd.ppo2.common_functions import Model
    T, E, A, R = cfg["T"], cfg["E"], cfg["A"], cfg["R"]
    image = "hwc" in cfg
    D = None if image else cfg["D"]
    N = T * E
    B = N // cfg["nmb"]
    np.random.seed(1)                     # parameter init stream (ortho_init)
    policies.reset_scopes()
    if image:
        policy = policies.LstmPolicy if cfg.get("nlstm") else policies.CnnPolicy
        ob_space, ac_space = spaces.Box(0, 255, shape=cfg["hwc"], dtype=np.uint8), spaces.Discrete(A)
    else:
        policy = policies.MlpPolicy64 if cfg["upstream"] else policies.MlpPolicy
        ob_space, ac_space = spaces.Box(-np.inf, np.inf, shape=(D,)), spaces.Box(-1, 1, shape=(A,))
    model = Model(policy=policy, ob_space=ob_space, ac_space=ac_space,
                  nbatch_act=E, nbatch_train=B, nsteps=T, ent_coef=cfg["ent_coef"], vf_coef=0.5, max_grad_norm=0.5,
                  upstream_loss=cfg["upstream"], comm=comm)
    model.seed_shuffle(0)                 # same index stream on every rank
    ro = (make_image_rollout if image else make_rollout)(model, cfg, seed=rank, device=device)
    gamma = 0.99 if R == 1 else [0.99] * R

    def step(events=None):
        if events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # the scan's own advantage array is left out: the update consumes returns and `returns - values` only, as the
        # reference's Runner.run hands on mb_returns but not mb_advs (baselines/ppo2/common_functions.py:344, ppo2.py:92)
        ro.advs_gae, ro.returns, ro.advs_ori = hip.gae(ro.rewards, ro.values, ro.dones, ro.last_values, ro.last_dones,
                                                       gamma, 0.95, want_advs=False)
        if events is not None:
            e1.record()
            events["gae"].append((e0, e1))
        if cfg["upstream"]:
            ro.advs = ro.advs_ori.reshape(T, E)
        else:
            ro.advs = hip.compute_advs([200, 120, 0, 0, 0][:R], ro.values.reshape(N, R), ro.advs_ori.reshape(N, R)).reshape(T, E)
        return model.train_on_rollout(cfg["lr"], cfg["cliprange"], ro, cfg["noptepochs"], B,
                                      events=None if events is None else events["grad"], more_updates=True)
    return model, ro, step


def gradient_roofline(name, cfg, B, grad_ms):
    """The minibatch gradient call against the matrix-pipe peak: algorithmic FLOPs of one call over its duration."""
    image = "hwc" in cfg
    A, R = cfg["A"], cfg["R"]
    flops_mb = (cnn_flops_per_sample(cfg["hwc"], A, cfg.get("nlstm", 0)) if image
                else dense_flops_per_sample(cfg["D"], cfg["hidden"], A, R)) * B
    fused = (not image) and cfg["hidden"] == (64, 64)
    fused256 = (not image) and cfg["hidden"] == (256, 128)
    achieved_tf = flops_mb / (grad_ms * 1e-3) / 1e12
    peak_tf = MFMA_F32_PEAK_TFLOPS
    if fused:
        # layer 2's forward, input-gradient and weight-gradient products of both trunks (3 x 2 x 2 H H flops per sample) run
        # on the bf16 pipe as six piece products per multiply-add, the rest on the float32 pipe: the peak is the
        # algorithmic flops over the time both pipes need at their dense rates
        h = cfg["hidden"][0]
        f_bf16 = 3 * 2 * (2 * h * h) * B
        peak_tf = flops_mb / ((flops_mb - f_bf16) / MFMA_F32_PEAK_TFLOPS + 6 * f_bf16 / MFMA_BF16_PEAK_TFLOPS)
    if fused256:
        # layer 2's weight gradient of both trunks (2 x 2 H1 H2 flops per sample) runs on the bf16 pipe as six piece
        # products per multiply-add, the rest on the float32 pipe
        h1, h2 = cfg["hidden"]
        f_bf16 = 2 * (2 * h1 * h2) * B
        peak_tf = flops_mb / ((flops_mb - f_bf16) / MFMA_F32_PEAK_TFLOPS + 6 * f_bf16 / MFMA_BF16_PEAK_TFLOPS)
    if image:
        # every wide product runs on the bf16 pipe: conv1 at three piece products per multiply-add (its uint8 operand is
        # one piece), the others at six; the recurrent h . Wh products of the sequence kernels stay on the float32
        # pipe.  The peak is the algorithmic flops over the time the two pipes need at their dense rates.
        f_conv1 = cnn_conv1_flops_per_sample(cfg["hwc"]) * B if cfg["hwc"][2] * 8 == 32 else 0
        f_rec = lstm_recurrent_flops_per_sample(cfg["nlstm"]) * B if cfg.get("nlstm") else 0
        peak_tf = flops_mb / (f_rec / MFMA_F32_PEAK_TFLOPS + (3 * f_conv1 + 6 * (flops_mb - f_conv1 - f_rec)) / MFMA_BF16_PEAK_TFLOPS)
    return {"bound": "mfma",
            "kernel": ("ppo_grad = mlp_fused16_kernel + fused_finalize_kernel (minibatch gather, forward, loss, backward and weight "
                       "gradients; fp32 MFMA except layer 2's three products, which run on the bf16 pipe as six "
                       "exact piece products per multiply-add; peak = algorithmic flops over the time the two matrix pipes need at "
                       "their dense rates)") if fused else
                      ("ppo_grad = gather + mlp_fused256_kernel (forward, loss, input gradients) + mlp_wgrad256_kernel "
                       "(all weight gradients; layer 2's on the bf16 pipe as six exact piece products per multiply-add) + slab "
                       "reduction, otherwise fp32 MFMA; peak = algorithmic flops over the time the two matrix pipes need at "
                       "their dense rates") if fused256 else
                      ("ppo_grad = gather + layered GEMM / implicit-GEMM conv launches (bf16 pipe: six exact piece products per "
                       "float32 multiply-add, three for conv1 on uint8 frames; float32 pipe for the recurrent products of the "
                       "sequence kernels) + loss + slab reductions; peak = algorithmic flops over the time the two matrix "
                       "pipes need at their dense rates"),
            "achieved": achieved_tf, "peak": peak_tf, "unit": "TFLOP/s",
            "frac": achieved_tf / peak_tf, "frac_of_fp32_mfma_peak": achieved_tf / MFMA_F32_PEAK_TFLOPS,
            "traffic": measured_traffic(name, "ppo2hip::mlp_fused16_kernel<5, false, 2>") if fused else
                       measured_traffic(name, ("ppo2hip::mlp_fused256_kernel", "ppo2hip::mlp_wgrad256_kernel")) if fused256 else None,
            "flops_per_launch": flops_mb, "ms_per_launch": grad_ms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 100 updates of 1.5 ms behind a warm-up that covers the card's clock ramp (the first dozen updates after an
    # idle spell run up to 10 % slower) — 0.2 s for the headline region
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="mujoco_mlp64", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the short runs of the other workloads that follow the headline region")
    args = ap.parse_args()
    cfg = WORKLOADS[args.workload]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started bare with --gpus N: this process becomes the launcher of N fresh rank processes (one per GPU) and
        # has not touched the GPU itself; it passes their output through and exits with their status
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # PPO2HIP_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share
    # devices, gradients staged through the host); the measured configuration is one rank per GPU over RCCL
    backend = os.environ.get("PPO2HIP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit("rank %d has no GPU of its own (%d visible)" % (local_rank, ndev))
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    comm = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        from baselines_amd.common.mpi_adam import DataParallel
        comm = DataParallel(os.environ.get("PPO2HIP_BENCH_DP_MODE", "mpi"))

    from baselines_amd import hip

    T, E, A, R = cfg["T"], cfg["E"], cfg["A"], cfg["R"]
    image = "hwc" in cfg
    D = None if image else cfg["D"]
    N = T * E
    B = N // cfg["nmb"]
    model, ro, step = build_workload(cfg, device, comm, rank)
    comm_info = None
    if world > 1:
        # Proof that the collective the timed region uses spans every rank: each rank contributes {its id, 1} to a SUM
        # all-reduce through the same communicator object (the library's own RCCL communicator when the transport is
        # "rccl": ppo2hip_allreduce_f64) and must get {N(N-1)/2, N} back.
        probe = torch.tensor([float(rank), 1.0], dtype=torch.float64, device=device)
        comm.allreduce_sum_(probe)
        id_sum, ranks_seen = (int(round(x)) for x in probe.cpu().tolist())
        if ranks_seen != world or id_sum != world * (world - 1) // 2:
            raise SystemExit("rank %d: the all-reduce reached %d ranks (id sum %d), expected %d (%d)"
                             % (rank, ranks_seen, id_sum, world, world * (world - 1) // 2))
        native = comm.native()
        comm_info = {"transport": comm.transport, "mode": comm.mode, "ranks_seen": ranks_seen, "rank_id_sum": id_sum,
                     "library_communicator": native is not None,
                     "library_comm_size": native.size if native is not None else None,
                     "epoch_call": "ppo_epoch_dp" if (native is not None and not cfg.get("nlstm")) else "ppo_grad + allreduce + adam_step per minibatch",
                     "allreduce_bytes_per_minibatch": 4 * model.spec.nparams}

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # Optional extra untimed updates before the caller's warmup (off by default; reported as "priming_steps").
    priming = int(os.environ.get("PPO2HIP_BENCH_PRIMING", "0"))
    for _ in range(priming):
        step()
    for _ in range(args.warmup):
        step()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    # The interpreter's cyclic collector stays out of the timed regions: with torch imported a full pass takes 35-48 ms
    # (30 updates of the headline workload), and it used to land on the first steps after the warm-up in two runs of
    # three.  ppo2.learn does the same for its loop (gc.freeze() once the model and runner stand).
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        last = step()
        marks[k + 1].record()            # GPU-side step boundaries: the spread of the per-step times, beside the wall mean
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    step_seq = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)]
    slowest_step = max(range(args.steps), key=step_seq.__getitem__)
    step_ms = sorted(step_seq)
    dt = max_over_ranks(dt)
    losses = last.cpu().numpy()
    assert np.all(np.isfinite(losses)), "update diverged: %s" % losses
    model.check_faults()       # raises if an index list was out of range or a launch gave up waiting for its peer workgroups

    # instrumented pass: HIP events around the GAE launch and every minibatch gradient evaluation
    events = {"gae": [], "grad": []}
    for _ in range(3):
        step(events)
    torch.cuda.synchronize()
    # the GAE launch alone: launches straight through the C-ABI into preallocated outputs, back to back
    # between two events behind a kernel that keeps the queue busy, so that the interval holds kernel time
    # and not the host's launch latency
    import ctypes
    from baselines_amd._lib import lib, check
    gv = (ctypes.c_double * R)(*([0.99] * R))
    outs = [None] + [torch.empty_like(ro.rewards) for _ in range(2)]      # returns and returns - values, as in step()
    dones_u8, last_dones_u8 = ro.dones.to(torch.uint8), ro.last_dones.to(torch.uint8)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def gae_raw():
        check(lib.ppo2hip_gae_f32(ptr(ro.rewards), ptr(ro.values), ptr(dones_u8), ptr(ro.last_values), ptr(last_dones_u8), gv,
                                  0.95, T, E, R, None, ptr(outs[1]), ptr(outs[2]), stream), "gae")

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    gae_raw()
    torch.empty(64 << 20, device=device).normal_()
    e0.record()
    for _ in range(reps):
        gae_raw()
    e1.record()
    torch.cuda.synchronize()
    gae_ms = e0.elapsed_time(e1) / reps
    assert torch.equal(outs[1], ro.returns) and torch.equal(outs[2], ro.advs_ori)
    # the same kernel family on a rollout large enough to be bandwidth- rather than latency-bound: the 65536-env x
    # 128-step rollout of the env-sharded configuration, whole, on this one GPU (R = 1)
    def gae_large():
        Tl, El, nsets = 128, 65536, 4
        gen = torch.Generator(device=device)
        gen.manual_seed(3)
        g1 = (ctypes.c_double * 1)(0.99)
        # four distinct sets of inputs and outputs (143 MB each, 570 MB in all) visited in turn, so that no launch finds
        # its operands in the 256 MB Infinity Cache left there by the launch before it
        sets = []
        for _ in range(nsets):
            rw = torch.randn(Tl, El, 1, generator=gen, device=device)
            vl = torch.randn(Tl, El, 1, generator=gen, device=device)
            dn = (torch.rand(Tl, El, generator=gen, device=device) < 0.005).to(torch.uint8)
            lv = torch.randn(El, 1, generator=gen, device=device)
            ld = torch.zeros(El, dtype=torch.uint8, device=device)
            sets.append((rw, vl, dn, lv, ld, [torch.empty_like(rw) for _ in range(2)]))

        def call(k):
            rw, vl, dn, lv, ld, o = sets[k % nsets]
            check(lib.ppo2hip_gae_f32(ptr(rw), ptr(vl), ptr(dn), ptr(lv), ptr(ld), g1, 0.95, Tl, El, 1, None, ptr(o[0]),
                                      ptr(o[1]), stream), "gae")
        for k in range(nsets):
            call(k)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.empty(64 << 20, device=device).normal_()
        a.record()
        for k in range(reps):
            call(k)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        nbytes = Tl * El * 17 + El * 5           # rewards 4 + values 4 + done 1 read, two float32 outputs written
        return {"bound": "hbm", "kernel": "gae_vec2_kernel<true>", "workload": "65536 envs x 128 steps, R=1 (configs[3], whole, on one GPU)",
                "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": nbytes, "ms_per_launch": ms,
                "traffic": measured_traffic("mujoco_mlp64", "ppo2hip::gae_vec2_kernel<true>"),
                "buffer_sets": nsets, "note": "outputs: returns and returns - values (what the update reads); %d buffer sets of "
                                              "143 MB rotate so the 256 MB Infinity Cache holds none of a launch's operands" % nsets}
    gae_large_roofline = gae_large() if rank == 0 else None
    # median over the minibatches: the first evaluation of an epoch can sit behind the index upload, which is not
    # launch time
    grad_ms = float(np.median([a.elapsed_time(b) for a, b in events["grad"]]))

    # what this card delivers on a plain device copy and on the library's fp32 GEMM, for reading the fractions below
    # against achievable rather than nominal peaks
    def timed(fn, reps):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3
    src = torch.empty(1 << 28, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    copy_gbs = 2 * src.numel() * 4 / timed(lambda: dst.copy_(src), 10) / 1e9
    del dst
    ga, gb = src[:8192 * 8192].view(8192, 8192), src[8192 * 8192:2 * 8192 * 8192].view(8192, 8192)
    sgemm_tf = 2 * 8192 ** 3 / timed(lambda: torch.matmul(ga, gb), 5) / 1e12
    del src, ga, gb

    # the other workloads, a few steps each behind the headline region: the same step function, timed the same way
    # (barrier + synchronize on both sides, max over ranks), so that their figures come out of this run too.  With
    # several ranks only the two configurations that name several GPUs run (the env-sharded MLP and cnn + LSTM), and a
    # watchdog guards them: should an extra hang (a rank lost inside a collective), every rank gives up after
    # EXTRA_TIMEOUT_S, rank 0 prints the headline line it already has, and the processes exit.
    extra = {}
    cpu_line = cpu_baseline(model, ro, cfg) if (world == 1 and rank == 0 and not args.no_cpu_baseline) else None
    headline = None
    if rank == 0:
        headline = headline_line(args, cfg, world, N, B, T, E, A, R, D, image, dt, step_ms, slowest_step, priming, grad_ms, gae_ms,
                                 gae_large_roofline, copy_gbs, sgemm_tf, losses, comm_info, cpu_line)
    if not args.no_extra_workloads:
        import threading
        fired = threading.Event()

        def give_up():
            fired.set()
            if rank == 0:
                headline["workloads"] = dict(extra, error="an extra workload did not finish within %d s; abandoned" % EXTRA_TIMEOUT_S)
                print(json.dumps(headline), flush=True)
            os._exit(0)
        watchdog = threading.Timer(EXTRA_TIMEOUT_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        model.close()
        del model, ro, step, outs
        torch.cuda.empty_cache()
        names = ("mlp_env_shard", "mujoco_fork", "atari_cnn", "atari_lstm", "mujoco_mlp64") if world == 1 else ("mlp_env_shard", "atari_lstm")
        for name in names:
            if name == args.workload:
                continue
            xcfg = WORKLOADS[name]
            xN, xB = xcfg["T"] * xcfg["E"], xcfg["T"] * xcfg["E"] // xcfg["nmb"]
            # (a model built later in a process runs its first twenty-odd updates about 1 ms slower — new worker threads
            # and buffers settling — so the millisecond-scale MLP updates get a warm-up that covers it)
            xsteps, xwarm = (20, 30) if "hwc" not in xcfg else (3, 2)
            try:
                xmodel, xro, xstep = build_workload(xcfg, device, comm, rank)
                for _ in range(xwarm):
                    xstep()
                gc.collect()
                gc.disable()
                barrier()
                t0 = time.perf_counter()
                for _ in range(xsteps):
                    xlast = xstep()
                barrier()
                xdt = max_over_ranks(time.perf_counter() - t0)
                gc.enable()
                xlosses = xlast.cpu().numpy()
                xmodel.check_faults()
                xev = {"gae": [], "grad": []}
                xstep(xev)
                torch.cuda.synchronize()
                xgrad_ms = float(np.median([a.elapsed_time(b) for a, b in xev["grad"]]))
                roof = gradient_roofline(name, xcfg, xB, xgrad_ms)
                extra[name] = {"samples_per_s": world * xN * xsteps / xdt, "ms_per_step": xdt / xsteps * 1e3, "steps": xsteps,
                               "warmup": xwarm, "samples_per_step_per_gpu": xN, "nbatch_train": xB,
                               "finite_losses": bool(np.all(np.isfinite(xlosses))),
                               "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_of_fp32_mfma_peak", "ms_per_launch")}}
                xmodel.close()
                del xmodel, xro, xstep, xlast, xev
            except Exception as ex:          # a failing extra must not take the headline line with it
                extra[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
                if world > 1:
                    break                    # the ranks may no longer be in step: no further collectives
            torch.cuda.empty_cache()
        watchdog.cancel()
        if fired.is_set():
            time.sleep(3600)                 # the watchdog is printing and exiting

    if rank == 0:
        if extra:
            headline["workloads"] = extra
        print(json.dumps(headline))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


EXTRA_TIMEOUT_S = 240


def headline_line(args, cfg, world, N, B, T, E, A, R, D, image, dt, step_ms, slowest_step, priming, grad_ms, gae_ms, gae_large_roofline,
                  copy_gbs, sgemm_tf, losses, comm_info, cpu_line):
    """The JSON object of the headline workload (rank 0)."""
    if image:
        shape = "Atari-shaped rollout: %dx%dx%d uint8 obs, nature_cnn%s Categorical(%d)" % (
            cfg["hwc"] + (" + LSTM(%d), env-sequence minibatches," % cfg["nlstm"] if cfg.get("nlstm") else "", A))
        which = "configs[4]" if cfg.get("nlstm") else "configs[2]"
    else:
        shape = "MuJoCo-shaped rollout: obs_dim=%d act_dim=%d Gaussian MLP%s" % (D, A, str(cfg["hidden"]).replace(" ", ""))
        which = "configs[3] shard" if args.workload == "mlp_env_shard" else "configs[1]"
    gae_bytes = T * E * (16 * R + 1) + E * (4 * R + 1)
    gae_gbs = gae_bytes / (gae_ms * 1e-3) / 1e9
    out = {
        "metric": "PPO2 samples/sec (GAE+update)", "value": world * N * args.steps / dt, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "priming_steps": priming, "ms_per_step": dt / args.steps * 1e3,
        "ms_per_step_gpu_events": {"median": step_ms[len(step_ms) // 2], "min": step_ms[0], "max": step_ms[-1], "slowest_step": slowest_step},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s synthetic %s nenvs=%d nsteps=%d per GPU, %d minibatches x %d epochs, %s loss"
                               % (which, shape, E, T, cfg["nmb"], cfg["noptepochs"], "upstream" if cfg["upstream"] else "fork"),
                   "samples_per_step_per_gpu": N, "nbatch_train": B,
                   "parallelism": "env-sharded dp%d, RCCL grad all-reduce (MpiAdam semantics)" % world if world > 1 else "single GPU"},
        "roofline": gradient_roofline(args.workload, cfg, B, grad_ms),
        "gae_roofline": {"bound": "hbm", "kernel": "gae_tiled_kernel", "achieved": gae_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gae_gbs / HBM_PEAK_GBS, "bytes_per_launch": gae_bytes,
                         "traffic": measured_traffic(args.workload, "ppo2hip::gae_tiled_kernel"),
                         "ms_per_launch": gae_ms},
        "gae_roofline_large": gae_large_roofline,
        "device_measured": {"copy_gbs": copy_gbs, "gae_large_vs_copy": (gae_large_roofline["achieved"] / copy_gbs) if gae_large_roofline else None,
                            "library_sgemm_tflops": sgemm_tf,
                            "note": "1 GiB device-to-device copy (read + write) and an 8192^3 fp32 torch.matmul on this card"},
        "losses": [float(x) for x in losses],
    }
    if comm_info is not None:
        out["comm"] = comm_info
    if cpu_line is not None:
        out["cpu_baseline"] = cpu_line
    return out


if __name__ == "__main__":
    main()
