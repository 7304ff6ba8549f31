#!/bin/bash
# instruction-fetch stalls of the tick / B1 launches (16 steady-state ticks, one stream)
set -e
export HK_SPLIT=0 TMPDIR=/tmp
out=gpurun_out/ifetch; mkdir -p $out
python3 tools/experiments/region_cost.py dump /tmp/rc_state.npz
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_IFETCH -d $out/a -o pmc --output-format csv -- python3 tools/experiments/region_cost.py run /tmp/rc_state.npz > $out/a.log 2>&1
rocprofv3 --pmc SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_REQ SQC_ICACHE_HITS -d $out/b -o pmc --output-format csv -- python3 tools/experiments/region_cost.py run /tmp/rc_state.npz > $out/b.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        key = "tick" if "env_run_kernel" in k else "b1" if "env_b1_kernel" in k else "lqn" if "lqn_round" in k else None
        if key: acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key].add(r["Dispatch_Id"])
for key, d in acc.items():
    wc = d["SQ_WAVE_CYCLES"]
    print(key, "launches", len(n[key]) // 2, {c: "%.3e" % v for c, v in d.items()})
    print("   waiting for instructions %.1f %% of wave cycles; waiting (any) %.1f %%; issuing %.1f %%" % (100 * d["SQ_WAIT_INST_ANY"] / wc, 100 * d["SQ_WAIT_ANY"] / wc, 100 * d["SQ_ACTIVE_INST_ANY"] / wc))
PY
