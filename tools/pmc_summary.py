#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection.csv files (FETCH_SIZE and WRITE_SIZE collected in SEPARATE passes, as
MI355X_MICROARCH.md prescribes) into profiles/pmc_summary.json: HBM bytes per launch for each kernel.
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction: FETCH_SIZE reads exactly 1/2 of the bytes of a wide coalesced
streaming read; our kernels gather 4-byte fields of 420-byte records, which is NOT that pattern, so both the raw and the
doubled figure are recorded and the doubled one is used as the (conservative) traffic."""
import csv, json, re, sys, collections, glob

def fold(path, counter, last=None):
    per = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hk::", "")
            name = re.sub(r"^g[48]::", "", name)                               # lane-group namespace (hk_env_ga.h)
            name = re.sub(r"^env_run_kernel<.*>$", "env_run_kernel", name)     # the headline instantiation <false, false>
            per[name].append(float(r["Counter_Value"]))
    return {k: (sum(v[-last:]) / len(v[-last:]) if last else sum(v) / len(v), len(v)) for k, v in per.items()}

fetch = fold(sys.argv[1], "FETCH_SIZE", 200)
write = fold(sys.argv[2], "WRITE_SIZE", 200)
out = {"note": __doc__.strip().split("\n\n")[0], "window": "last 200 launches of each kernel (steady state)"}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, (0.0, 0))[0] * 1024.0
    w = write.get(k, (0.0, 0))[0] * 1024.0
    out[k] = {"fetch_bytes_raw_per_launch": f, "fetch_bytes_x2_per_launch": 2 * f, "write_bytes_per_launch": w,
              "hbm_bytes_per_launch": 2 * f + w, "launches_seen": fetch.get(k, (0, 0))[1]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
