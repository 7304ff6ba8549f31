#!/usr/bin/env python3
"""Distributional pin against the ONLY reference-held outcome data: ExperimentLogs/*.txt.

Runs the CPU oracle on the reference's experiment set-ups (same agents, wiring, laps, orderings e % A!, no start jitter),
writes the races in the reference's own log grammar, reads both logs back with the same in-repo reader
(hierarchicalkarting_amd/telemetry.py) and prints the statistics side by side: mean total time, median best lap,
collisions and illegal lane changes per race, wins, DNFs.  No reference code is executed; the reference logs are read as data
(build container only — /root/reference does not exist on the GPU box).  With --update the reference-side statistics are
stored in tests/golden/reference_log_stats.json, which tests/test_reference_logs.py compares the oracle with on any box.

  python tools/compare_experiment_logs.py [--update] [--only oval_1v1,...] [--mcts-iterations 128]"""
import argparse, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hierarchicalkarting_amd import _lib, telemetry as T          # noqa: E402
from hierarchicalkarting_amd.config import make_config            # noqa: E402

REF_LOGS = "/root/reference/ExperimentLogs"
MC, FX, LQ = _lib.HK_HIGH_MCTS, _lib.HK_HIGH_FIXED, _lib.HK_LOW_LQR
# name -> (reference log, track, agent names in Agents[] order, high modes, tree depths, experiments)
SETUPS = {
    # CompeteAgents-Oval.unity env MCTS_LQR_vs_Fixed_LQR_Oval2: Agents = [MCTS-LQR, Fixed-LQR], 50 experiments, laps 4
    "oval_1v1": ("MCTS_LQR_vs_Fixed_LQR_Oval2", "oval", ["MCTS-LQR", "Fixed-LQR"], [MC, FX], [8, 5], 50),
    # CompeteAgents-OvalDuosAll.unity env 1132641209: Agents = [M0, M1, F0, F1], 48 experiments
    "oval_2v2": ("MCTS_LQR_vs_Fixed_LQR_OvalDuos2", "oval", ["MCTS-LQR(M0)", "MCTS-LQR(M1)", "Fixed-LQR(F0)", "Fixed-LQR(F1)"], [MC, MC, FX, FX], [8, 8, 5, 5], 48),
    # CompeteAgents-Complex.unity: laps 3, MaxLaneChanges 4, 41 sections
    "complex_1v1": ("MCTS_LQR_vs_Fixed_LQR_Complex3", "complex", ["MCTS-LQR", "Fixed-LQR"], [MC, FX], [8, 5], 50),
}


def run_ours(track, names, high, depth, n_exp, iters, log_path):
    import oracle_lib as O
    A = len(names)
    b = make_config(n_exp, A, track=track, high_mode=high, low_mode=[LQ] * A, tree_search_depth=depth, jitter_seed=0, auto_reset=0,
                    mcts_iterations=iters)
    o = O.OracleEnv(b)
    o.reset()                                    # experiment e starts from ordering e % A! (REC:528-530)
    for _ in range(80):
        o.step(100)
        if (o.env_state()["inactive_mask"] == (1 << A) - 1).all():
            break
    o.step(1)                                    # the tick on which REC.FixedUpdate writes the block
    res = o.episode_results()
    log = T.ExperimentLog(log_path, names, b.cfg.laps)
    for e in range(n_exp):
        log.append(e, res[e])
    return T.summarize_log(T.read_experiment_log(log_path))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--update", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--mcts-iterations", type=int, default=128)
    a = ap.parse_args()
    golden = os.path.join(ROOT, "tests", "golden", "reference_log_stats.json")
    ref_all = json.load(open(golden)) if os.path.exists(golden) else {}
    out = {}
    for name, (logname, track, names, high, depth, n_exp) in SETUPS.items():
        if a.only and name not in a.only.split(","):
            continue
        path = os.path.join(REF_LOGS, logname + ".txt")
        if os.path.exists(path):
            ref_all[name] = {"log": "ExperimentLogs/%s.txt" % logname, "stats": T.summarize_log(T.read_experiment_log(path))}
        with tempfile.TemporaryDirectory() as d:
            ours = run_ours(track, names, high, depth, n_exp, a.mcts_iterations, os.path.join(d, "ours.txt"))
        out[name] = ours
        print("== %s  (reference: %s)" % (name, ref_all.get(name, {}).get("log", "not available on this box")))
        for typ in ours:
            r = ref_all.get(name, {}).get("stats", {}).get(typ, {})
            print("  %-10s %-30s %12s %12s" % (typ, "", "reference", "oracle"))
            for k in ("races", "wins", "dnfs", "mean_total_time", "median_best_lap", "collisions_per_race", "illegal_lane_changes_per_race"):
                rv, ov = r.get(k), ours[typ][k]
                fmt = lambda v: "-" if v is None else ("%d" % v if isinstance(v, int) else "%.3f" % v)
                ratio = "" if not isinstance(rv, (int, float)) or not isinstance(ov, (int, float)) or not rv else "  x%.3f" % (ov / rv)
                print("  %-10s %-30s %12s %12s%s" % ("", k, fmt(rv), fmt(ov), ratio))
    if a.update:
        json.dump(ref_all, open(golden, "w"), indent=1, sort_keys=True)
        print("wrote", golden)
    return out


if __name__ == "__main__":
    main()
