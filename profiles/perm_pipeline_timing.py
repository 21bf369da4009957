"""Host time of the bit-exact permutation pipeline (MT19937 + NumPy's legacy shuffle, one permutation of arange(N) per
epoch, epochs on worker threads with the generator handed on) for one update's four epochs, at one GPU's shard of the
env-sharded configuration (N = 1 M) and at the global sample count every rank would have to draw in "global" mode on
eight GPUs (N = 8.4 M).  Run on the GPU box: python profiles/perm_pipeline_timing.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baselines_amd import hip  # noqa: E402

out = {}
for N in (524288, 1048576, 8388608):
    sh = hip.LegacyShuffler(0)
    pp = hip.PermPipeline(sh, N, 4, pinned=True)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        pp.begin(4, relative=True)
        pp.wait_all()
        best = min(best, time.perf_counter() - t0)
    out[str(N)] = {"ms_per_update_of_4_epochs": best * 1e3}
    del pp, sh
print(json.dumps(out))
