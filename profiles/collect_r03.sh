#!/bin/bash
# Profile collection for the files profiles/r03_*, run on the GPU box from the repo root:  bash profiles/collect_r03.sh [workloads...]
# Each pass is its own rocprofv3 run (kernel trace + stats; FETCH_SIZE; WRITE_SIZE); the summaries land in gpurun_out/
# and are copied into profiles/ by hand.
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
mkdir -p $OUT
WL=${@:-mujoco_mlp64 mlp_env_shard mujoco_fork atari_cnn atari_lstm}
for w in $WL; do
    rm -rf $OUT/stats_$w
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-extra-workloads > $OUT/stats_$w.log 2>&1 || { tail -5 $OUT/stats_$w.log; exit 1; }
    db=$(find $OUT/stats_$w -name "*results.db" | head -1)
    python3 profiles/collect_traffic.py stats "$db" gpurun_out/r03_${w}_kernel_stats.csv || exit 1
    grep '"metric"' $OUT/stats_$w.log > gpurun_out/r03_${w}_bench_under_profiler.json
    rm -rf $OUT/stats_$w                      # the raw database stays on the box (the merge back is capped at 64 MiB)
    echo "stats $w done"
done
for w in $WL; do
    case $w in mujoco_*|atari_cnn) ;; *) continue;; esac
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf $OUT/${c}_$w
        timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $c -d $OUT/${c}_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-extra-workloads > $OUT/${c}_$w.log 2>&1 || { tail -5 $OUT/${c}_$w.log; exit 1; }
    done
    python3 profiles/collect_traffic.py traffic $w "$(find $OUT/FETCH_SIZE_$w -name '*results.db' | head -1)" "$(find $OUT/WRITE_SIZE_$w -name '*results.db' | head -1)" gpurun_out/r03_${w}_hbm_traffic.json || exit 1
    rm -rf $OUT/FETCH_SIZE_$w $OUT/WRITE_SIZE_$w
    echo "traffic $w done"
done
