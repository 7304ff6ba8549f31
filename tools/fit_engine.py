#!/usr/bin/env python3
"""Fit / evaluate the engine restatement's parameters (include/hk.h hk_engine_params) against the reference's ExperimentLogs.

The reference's engine (Unity / PhysX) has no source; what it does between two FixedUpdates is restated from the prefabs' Rigidbody /
collider / WheelCollider / KartAnimation data with ONE fitted number (side_slope0).  This tool races the CPU oracle on a set of the
22 experiment set-ups for given parameter values and scores the statistics against tests/golden/reference_log_stats.json (which
tools/compare_experiment_logs.py --update wrote from the reference's logs: data, no reference code runs).

  python tools/fit_engine.py --set fit  side_slope0=1.5                      one evaluation on the fitting half
  python tools/fit_engine.py --set holdout side_slope0=1.5                   ... on the held-out half
  python tools/fit_engine.py --set fit --sweep side_slope0=0.5,1,1.5,2,3     a sweep

The fitting half / held-out half split is fixed here (alternating set-ups in sorted order within each scene family), so that the
numbers in DESIGN.md section 4 can be regenerated."""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))

REF = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_log_stats.json")))


def split():
    names = sorted(REF)
    fam = {}
    for n in names:
        key = ("Complex" if "Complex" in n else "Oval") + ("Duos" if "Duos" in n else "")
        fam.setdefault(key, []).append(n)
    fit, hold = [], []
    for key in sorted(fam):
        for k, n in enumerate(fam[key]):
            (fit if k % 2 == 0 else hold).append(n)
    return fit, hold


def score(ref, ours):
    """sum over the agent rows of squared log ratios (pace, lane error) and squared rate differences (DNFs, wins)"""
    tot, parts = 0.0, {}
    for typ, o in ours.items():
        r = ref[typ]
        races = max(o["races"], 1)
        p = {}
        finished = o["dnfs"] < races
        p["lap"] = math.log(o["median_best_lap"] / r["median_best_lap"]) ** 2 * 100 if (finished and r["median_best_lap"] and o["median_best_lap"]) else 1.0
        p["time"] = math.log(o["mean_total_time"] / r["mean_total_time"]) ** 2 * 100 if (finished and o["mean_total_time"]) else 1.0
        p["dnf"] = ((o["dnfs"] - r["dnfs"]) / races) ** 2 * 4
        p["win"] = ((o["wins"] - r["wins"]) / races) ** 2
        if r["mean_lane_difference"] and o["mean_lane_difference"]:
            p["lane"] = math.log(o["mean_lane_difference"] / r["mean_lane_difference"]) ** 2 * 0.25
        parts[typ] = p
        tot += sum(p.values())
    return tot, parts


def evaluate(names, eng, verbose=True, iters=128):
    # (the candidate constants replace the module's defaults for THIS process only: the product reads no environment override of its physics)
    from hierarchicalkarting_amd import config as _cfg
    _cfg.ENGINE_PARAMS.update({k: float(v) for k, v in eng.items()})
    import compare_experiment_logs as CE
    import oracle_lib as O
    total = 0.0
    rows = []
    for n in names:
        t0 = time.time()
        _, ours = CE.run_ours(n, O.OracleEnv, iters)
        sc, parts = score(REF[n]["stats"], ours)
        total += sc
        for typ, o in ours.items():
            r = REF[n]["stats"][typ]
            rows.append((n, typ, r, o, parts[typ]))
            if verbose:
                print("%-36s %-9s wins %2d/%2d dnf %2d/%2d  lap %6.2f/%6.2f (x%.3f)  time %6.2f/%6.2f (x%.3f)  lane %.2f/%.2f (x%.2f)  coll %.2f/%.2f  ilc %.2f/%.2f  dvel %.2f/%.2f  [%.3f, %.0fs]" % (
                    n, typ, r["wins"], o["wins"], r["dnfs"], o["dnfs"], r["median_best_lap"], o["median_best_lap"],
                    o["median_best_lap"] / r["median_best_lap"] if r["median_best_lap"] else 0, r["mean_total_time"], o["mean_total_time"],
                    o["mean_total_time"] / r["mean_total_time"] if r["mean_total_time"] else 0,
                    r["mean_lane_difference"], o["mean_lane_difference"], o["mean_lane_difference"] / r["mean_lane_difference"] if r["mean_lane_difference"] else 0,
                    r["collisions_per_race"], o["collisions_per_race"], r["illegal_lane_changes_per_race"], o["illegal_lane_changes_per_race"],
                    r["mean_vel_difference"], o["mean_vel_difference"], sum(parts[typ].values()), time.time() - t0))
                sys.stdout.flush()
    return total, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="fit", help="fit | holdout | all | comma-separated substrings")
    ap.add_argument("--sweep", default="")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("params", nargs="*")
    a = ap.parse_args()
    fit, hold = split()
    if a.set == "fit":
        names = fit
    elif a.set == "holdout":
        names = hold
    elif a.set == "all":
        names = sorted(REF)
    else:
        names = [n for n in sorted(REF) if any(s in n for s in a.set.split(","))]
    eng = dict(kv.split("=") for kv in a.params)
    if a.sweep:
        k, vals = a.sweep.split("=")
        for v in vals.split(","):
            e = dict(eng); e[k] = v
            tot, _ = evaluate(names, e, verbose=not a.quiet)
            print("## %s=%s  score %.4f" % (k, v, tot)); sys.stdout.flush()
    else:
        tot, _ = evaluate(names, eng, verbose=not a.quiet)
        print("## %s  score %.4f over %d set-ups" % (eng, tot, len(names)))


if __name__ == "__main__":
    main()
