"""bench.py prints ONE JSON line with the agreed keys (short run, CPU baseline leg skipped).  GPU only."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         cwd=ROOT, check=True, capture_output=True, text=True, timeout=280).stdout
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None or r["traffic"] > 0
    g = d["gae_roofline"]
    assert g["bound"] == "hbm" and g["unit"] == "GB/s" and 0 < g["frac"] < 1
    assert d["value"] > 1e6 and abs(d["value"] - 524288 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    # the other four workloads ride in the same line, a few steps each
    w = d["workloads"]
    assert sorted(w) == ["atari_cnn", "atari_lstm", "mlp_env_shard", "mujoco_fork"]
    for name, x in w.items():
        assert "error" not in x, (name, x)
        assert x["finite_losses"] and x["samples_per_s"] > 1e4 and 0 < x["roofline"]["frac"] < 1
        assert abs(x["samples_per_s"] - x["samples_per_step_per_gpu"] / (x["ms_per_step"] * 1e-3)) / x["samples_per_s"] < 1e-6


def test_bench_two_ranks_rehearsal():
    """The N > 1 launch (torch.distributed.run, one process per rank; bench.py starts it itself when it is run without
    a launcher), rehearsed with two
    ranks on this box's one GPU: gloo carries the gradients through the host (PPO2HIP_BENCH_BACKEND=gloo), so
    the throughput means nothing, but rank handling, barriers, the max-over-ranks clock and the single JSON
    line from rank 0 are the measured path's."""
    env = dict(os.environ, PPO2HIP_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    # started bare, as `python bench.py --gpus 2`: the script launches its own two rank processes
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-extra-workloads"],
                         cwd=ROOT, env=env, check=True, capture_output=True, text=True, timeout=280).stdout
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "dp2" in d["config"]["parallelism"]
    assert "cpu_baseline" not in d                       # rank 0 at N = 1 only
    # the collective that the timed region uses reached both ranks: SUM of {rank id, 1} came back as {1, 2}
    c = d["comm"]
    assert c["ranks_seen"] == 2 and c["rank_id_sum"] == 1 and c["transport"] == "host" and c["mode"] == "mpi"
    assert c["library_communicator"] is False           # two ranks share this box's one GPU: RCCL cannot run here
    assert abs(d["value"] - 2 * 524288 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
