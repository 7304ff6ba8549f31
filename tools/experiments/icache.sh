#!/bin/bash
# instruction-cache behaviour of the tick / B1 / fused kernels under three schedules (run on the GPU box from the repo root)
set -e
R=$PWD; out=$R/gpurun_out/icache; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
run() {  # name, env...
  n=$1; shift
  env "$@" rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $out/$n.a -o pmc --output-format csv -- python3 $R/bench.py --steps 256 --warmup 0 --no-cpu-baseline --no-secondary > $out/$n.a.log 2>&1
  env "$@" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU -d $out/$n.b -o pmc --output-format csv -- python3 $R/bench.py --steps 256 --warmup 0 --no-cpu-baseline --no-secondary > $out/$n.b.log 2>&1
}
run split HK_DUMMY=1
run one HK_SPLIT=0
# (run park ...: HK_PARK was retired in round 6 with its kernel; profiles/r05_a_backend_flags.txt keeps the numbers)
cd $R
python3 - <<PY
import csv, glob, collections
for n in ("split", "one", "park"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for f in glob.glob("$out/%s.*/**/*counter_collection.csv" % n, recursive=True):
        rows = list(csv.DictReader(open(f)))
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) < last * 0.7: continue          # the timed window at the end of the run (pre-roll before it)
            k = r["Kernel_Name"]
            key = "tick/fused" if "env_run_kernel" in k else "b1" if "env_b1_kernel" in k else "lqn" if "lqn_round" in k else None
            if key: acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key].add(r["Dispatch_Id"])
    for key, d in acc.items():
        wc = d["SQ_WAVE_CYCLES"] or 1.0
        print(n, key, "launches", len(cnt[key]), {c: "%.3e" % v for c, v in sorted(d.items())})
        print("    icache hit rate %.4f; misses per launch %.0f; waiting for instructions %.1f %% of wave cycles; waiting (any) %.1f %%; issuing %.1f %%" % (
            d["SQC_ICACHE_HITS"] / max(d["SQC_ICACHE_REQ"], 1), d["SQC_ICACHE_MISSES"] / max(len(cnt[key]), 1) * 2, 100 * d["SQ_WAIT_INST_ANY"] / wc, 100 * d["SQ_WAIT_ANY"] / wc, 100 * d["SQ_ACTIVE_INST_ANY"] / wc))
PY
