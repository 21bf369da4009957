"""Fold rocprofv3 result databases (rocpd SQLite, the tool's default output) into the committed summaries.

  python profiles/collect_traffic.py stats   RESULTS.db OUT.csv
      per-kernel calls / total / average duration of a `--kernel-trace --stats` run (view top_kernels)
  python profiles/collect_traffic.py traffic WORKLOAD FETCH.db WRITE.db OUT.json
      bytes per launch and kernel from two `--pmc` runs (FETCH_SIZE and WRITE_SIZE cannot share a pass)

The counters are reported in KB; bytes = value x 1024.  On gfx950 FETCH_SIZE tallies a 128-byte request of
a wide streaming read as 64 bytes; no correction is applied here (the dominant kernel gathers 4 bytes per
lane, an access width for which the counter is uncalibrated); the note travels in the JSON instead.
"""
import csv
import json
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def stats(db, out):
    rows = sqlite3.connect(db).execute(
        "select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for name, calls, total, avg, pct in rows:
            w.writerow([short(name), calls, "%.3f" % total, "%.3f" % avg, "%.3f" % pct])
    print("wrote", out, len(rows), "kernels")


def fold(db, counter):
    q = ("select kernel_name, sum(value) * 1024.0 / count(distinct dispatch_id), count(distinct dispatch_id) "
         "from counters_collection where counter_name = ? group by kernel_name")
    return {short(k): (v, n) for k, v, n in sqlite3.connect(db).execute(q, (counter,))}


def traffic(workload, db_fetch, db_write, out):
    fetch, write = fold(db_fetch, "FETCH_SIZE"), fold(db_write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        e = kernels.setdefault(k, {})
        if k in fetch:
            e["FETCH_SIZE"], e["launches_FETCH_SIZE"] = int(round(fetch[k][0])), fetch[k][1]
        if k in write:
            e["WRITE_SIZE"], e["launches_WRITE_SIZE"] = int(round(write[k][0])), write[k][1]
    doc = {
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a second run, WRITE_SIZE) -- python3 bench.py "
                   "--workload W --steps 5 --warmup 2 --no-cpu-baseline --no-extra-workloads",
        "unit": "bytes per launch (counter KB x 1024, as reported; FETCH_SIZE is uncalibrated for the 4-byte-per-lane "
                "gathers of the dominant kernel and would need doubling for wide streaming reads)",
        "workload": workload,
        "kernels": kernels,
    }
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(*sys.argv[2:4])
    else:
        traffic(*sys.argv[2:6])
