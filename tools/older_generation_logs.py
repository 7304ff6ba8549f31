#!/usr/bin/env python3
"""A second hold-out for the engine restatement: the reference's OLDER generation of ExperimentLogs (the files without
the 2 / 3 suffix), which were not looked at when the restatement was built.  The Compete scenes hold only the latest set-ups, so an older log is
compared with our races of the set-up that carries its name + suffix — meaningful where the agents are LQNG only (no trained actor whose checkpoint
changed between the generations); the reference's own drift between its two generations is printed beside it as the yardstick.

  python tools/older_generation_logs.py [--update]      (--update writes tests/golden/older_generation_log_stats.json: statistics, no reference text)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hierarchicalkarting_amd import telemetry as T

LOGS = "/root/reference/ExperimentLogs"
GOLD = os.path.join(ROOT, "tests", "golden")
KEYS = ("median_best_lap", "mean_total_time", "mean_lane_difference", "illegal_lane_changes_per_race", "collisions_per_race", "wins", "dnfs", "races")


def main():
    ours = json.load(open(os.path.join(GOLD, "experiment_oracle.json")))
    new = json.load(open(os.path.join(GOLD, "reference_log_stats.json")))
    out = {}
    for name in sorted(ours):
        base = name.rstrip("23")
        path = os.path.join(LOGS, base + ".txt")
        if base == name or not os.path.exists(path):
            continue
        st = T.summarize_log(T.read_experiment_log(path))
        if set(st) != set(ours[name]["stats"]):
            continue                                  # another cast of agents under the old name
        out[base] = {"latest": name, "lqng_only": all("RL" not in t and "E2E" not in t for t in st), "stats": {t: {k: st[t][k] for k in KEYS} for t in st}}
    print("%-34s %-10s %22s %22s %16s" % ("older log (vs latest set-up)", "agent", "best lap old/new/ours", "total time old/new/ours", "wins old/new/ours"))
    for base, rec in out.items():
        for t, o in rec["stats"].items():
            n, u = new[rec["latest"]]["stats"][t], ours[rec["latest"]]["stats"][t]
            print("%-34s %-10s %7.2f %6.2f %6.2f %8.2f %6.2f %6.2f %6d %4d %4d %s" % (base, t, o["median_best_lap"], n["median_best_lap"], u["median_best_lap"], o["mean_total_time"],
                  n["mean_total_time"], u["mean_total_time"], o["wins"], n["wins"], u["wins"], "" if rec["lqng_only"] else "(trained actors: checkpoints may differ)"))
    if "--update" in sys.argv:
        json.dump(out, open(os.path.join(GOLD, "older_generation_log_stats.json"), "w"), indent=1, sort_keys=True)
        print("wrote tests/golden/older_generation_log_stats.json")


if __name__ == "__main__":
    main()
