#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace) of 16 steady-state ticks on one stream, per variant library: bash tools/experiments/ktime.sh base nostore ...
set -e
export HK_SPLIT=0 TMPDIR=/tmp
out=gpurun_out/ktime; mkdir -p $out
python3 tools/experiments/region_cost.py dump /tmp/rc_state.npz
for v in "$@"; do
  export HK_LIB_PATH=build/libhk_$v.so
  rocprofv3 --kernel-trace -d $out/$v -o t --output-format csv -- python3 tools/experiments/region_cost.py run /tmp/rc_state.npz > $out/$v.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/$v/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        key = "tick" if "env_run_kernel" in k else "b1" if "env_b1_kernel" in k else "lqn" if "lqn_round" in k else None
        if key: acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("$v", {k: " ".join("%.1f" % x for x in v) for k, v in acc.items()})
PY
done
