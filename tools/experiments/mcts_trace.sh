#!/bin/bash
# kernel totals of the configs[2] planner workload (rocprofv3 --kernel-trace --stats)
export TMPDIR=/tmp
O=gpurun_out/mcts_kt; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload mcts --steps 800 --warmup 256 --no-cpu-baseline > $O/bench.json 2> $O/err.log
python3 - <<PY
import csv, glob, json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print("value", round(d["value"] / 1e6, 1))
for r in csv.DictReader(open(glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0])):
    if float(r["Percentage"]) > 0.5: print("  %-70s calls %5s total %8.2f ms avg %8.1f us %5s%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
