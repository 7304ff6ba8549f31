#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection.csv files into profiles/<name>_pmc_summary.json.

  python tools/pmc_summary.py <fetch csv> <write csv> <out json> [--sq <sq csv> [<sq csv 2> ...]] [--commit <sha>] [--command "<what ran>"] [--env-steps-per-launch N]

FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes, as MI355X_MICROARCH.md prescribes (they do not fit one pass), in KiB.
gfx950 correction: FETCH_SIZE reads exactly 1/2 of the bytes of a wide coalesced streaming read; our kernels gather 4-byte fields
of 448-byte records, which is NOT that pattern, so both the raw and the doubled figure are recorded and the doubled one is used as
the (conservative) traffic.  Per kernel the FULL-SIZE launches are averaged: the 30 launches with the largest value of the counter
(a long hk_step ends with short launches for a few laggard envs, which say nothing about the kernel).
SQ passes (any of SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU, SQ_ACTIVE_INST_VALU, SQ_ACTIVE_INST_ANY, SQ_WAVE_CYCLES, SQ_WAIT_ANY,
SQ_WAIT_INST_ANY, SQ_INSTS_SALU, SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 ...): the same selection by SQ_WAVE_CYCLES / SQ_INSTS_SALU, plus
the derived fractions bench.py reports as roofline.binding."""
import csv, json, re, sys, collections


def rows(path):
    per = collections.defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hk::", "")
            name = re.sub(r"^g[48]::", "", name)                               # lane-group namespace (hk_env_ga.h)
            name = re.sub(r"<.*$", "", name)                                   # template arguments
            per[(name, int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
    return per


LAST = 0          # --last N: average the LAST N launches of each kernel (the steady state at the end of the profiled run) instead of the 30 largest


def top_mean(per, key, top=30):
    by = collections.defaultdict(list)
    for (name, disp), v in sorted(per.items(), key=lambda kv: kv[0][1]):
        if key in v:
            by[name].append(v)
    out = {}
    for name, lst in by.items():
        if LAST:
            sel = lst[-LAST:]
        else:
            lst.sort(key=lambda v: -v[key])
            sel = lst[:top]
        out[name] = ({c: sum(v.get(c, 0.0) for v in sel) / len(sel) for c in sel[0]}, len(lst))
    return out


def main():
    a = sys.argv[1:]
    fetch_csv, write_csv, out_path = a[0], a[1], a[2]
    sq, meta = [], {}
    i = 3
    while i < len(a):
        if a[i] == "--sq":
            i += 1
            while i < len(a) and not a[i].startswith("--"):
                sq.append(a[i]); i += 1
        elif a[i] in ("--commit", "--command"):
            meta[a[i][2:]] = a[i + 1]; i += 2
        elif a[i] == "--last":
            global LAST
            LAST = int(a[i + 1]); i += 2
        elif a[i] == "--waves-per-simd":                # waves of the tick / B1 kernels that share a SIMD (their launch bounds: HK_FIS_OCC / HK_B1_OCC)
            meta["waves_per_simd"] = int(a[i + 1]); i += 2
        elif a[i] == "--env-steps-per-launch":          # what one full-size launch of the tick kernel advances in the profiled window (E x ticks per launch)
            meta["env_steps_per_launch"] = float(a[i + 1]); i += 2
        else:
            i += 1
    fetch = top_mean(rows(fetch_csv), "FETCH_SIZE")
    write = top_mean(rows(write_csv), "WRITE_SIZE")
    import os, sys as _sys
    _sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from source_hash import source_hash
    meta["sources_sha16"] = source_hash()                 # the tree the counters were measured on (bench.py checks it)
    out = {"note": "HBM bytes per launch from separate --pmc FETCH_SIZE / WRITE_SIZE passes (KiB -> bytes; fetch doubled: gfx950 correction); "
                   "SQ counters from their own passes; per kernel the mean over " + ("its last %d launches (the steady state at the end of the run)" % LAST if LAST else "its 30 largest launches"), **meta}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, ({}, 0))[0].get("FETCH_SIZE", 0.0) * 1024.0
        w = write.get(k, ({}, 0))[0].get("WRITE_SIZE", 0.0) * 1024.0
        out[k] = {"fetch_bytes_raw_per_launch": f, "fetch_bytes_x2_per_launch": 2 * f, "write_bytes_per_launch": w,
                  "hbm_bytes_per_launch": 2 * f + w, "launches_seen": fetch.get(k, ({}, 0))[1]}
    for path in sq:
        per = rows(path)
        key = "SQ_WAVE_CYCLES" if any("SQ_WAVE_CYCLES" in v for v in per.values()) else "SQ_INSTS_SALU"
        for k, (v, n) in top_mean(per, key).items():
            d = out.setdefault(k, {}).setdefault("sq", {})
            d.update({c: x for c, x in v.items()})
    for k, v in out.items():
        s = v.get("sq") if isinstance(v, dict) else None
        if not s:
            continue
        g = s.get
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            s["derived"] = {
                # per wave: share of its resident cycles in which it issues any / a vector instruction, waits (s_waitcnt, barrier) or stalls on issue
                "wave_issuing_any_frac": g("SQ_ACTIVE_INST_ANY", 0) / wc, "wave_issuing_valu_frac": g("SQ_ACTIVE_INST_VALU", 0) / wc,
                "wave_waiting_frac": g("SQ_WAIT_ANY", 0) / wc, "wave_issue_stall_frac": g("SQ_WAIT_INST_ANY", 0) / wc,
                # lanes switched on in an average vector instruction (SQ_THREAD_CYCLES_VALU counts one per active lane and instruction:
                # the 2-player solver, whose lanes all work, reads 49.5 of 64 with its 3-player games on 60 lanes)
                "valu_lanes_active_of_64": g("SQ_THREAD_CYCLES_VALU", 0) / g("SQ_INSTS_VALU", 1) if g("SQ_INSTS_VALU") else None,
                "valu_insts_per_launch": g("SQ_INSTS_VALU")}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k in ("env_run_kernel", "lqn_round_kernel")}, indent=1))


if __name__ == "__main__":
    main()
