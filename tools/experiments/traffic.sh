#!/bin/bash
# HBM traffic of the tick / B1 / solver launches of 16 steady-state ticks on one stream (FETCH_SIZE, WRITE_SIZE: separate passes, KiB)
set -e
export HK_SPLIT=0 TMPDIR=/tmp
out=gpurun_out/traffic; mkdir -p $out
python3 tools/experiments/region_cost.py dump /tmp/rc_state.npz
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/$c -o pmc --output-format csv -- python3 tools/experiments/region_cost.py run /tmp/rc_state.npz > $out/$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            key = "tick" if "env_run_kernel" in k else "b1" if "env_b1_kernel" in k else "lqn" if "lqn_round" in k else None
            if key and r["Counter_Name"] == c: acc[key][c].append(float(r["Counter_Value"]) * 1024)
for key, d in acc.items():
    for c, v in d.items():
        print("%-5s %-10s launches %2d  MB per launch: %s" % (key, c, len(v), " ".join("%.0f" % (x / 1e6) for x in v)))
PY
