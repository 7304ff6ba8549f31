#!/usr/bin/env python3
"""Summary of tools/experiments/region_cost.sh: per variant and kernel, vector instructions per wave over the 16 ticks and the lanes per instruction."""
import csv, glob, json, os, sys
out, names = sys.argv[1], sys.argv[2:]
res = {}
for v in names:
    acc = {}
    for f in glob.glob(os.path.join(out, v, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            key = "tick" if "env_run_kernel" in k else "b1" if "env_b1_kernel" in k else None
            if key is None: continue
            a = acc.setdefault(key, {"dispatches": set()})
            a["dispatches"].add(r["Dispatch_Id"])
            a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    res[v] = {}
    for key, a in acc.items():
        waves = 65536 * 4 / 64
        res[v][key] = {"launches": len(a["dispatches"]), "valu_insts_per_wave": a["SQ_INSTS_VALU"] / waves,
                       "lanes_per_inst": a["SQ_THREAD_CYCLES_VALU"] / a["SQ_INSTS_VALU"],
                       "wave_cycles": a["SQ_WAVE_CYCLES"], "active_inst_valu": a.get("SQ_ACTIVE_INST_VALU", 0.0)}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for v in names:
    for key, d in sorted(res[v].items()):
        print("%-10s %-5s launches %3d  insts/wave %9.0f  lanes %5.1f  wave_cycles %.3e" % (v, key, d["launches"], d["valu_insts_per_wave"], d["lanes_per_inst"], d["wave_cycles"]))
