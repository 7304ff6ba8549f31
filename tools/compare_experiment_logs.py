#!/usr/bin/env python3
"""Closed-loop pin against the ONLY outcome data the reference holds: ExperimentLogs/<ExperimentName>.txt.

For every experiment set-up of the reference's Compete scenes whose agents are all HierarchicalKartAgents
(tests/golden/reference_experiments.json, 22 set-ups: {Fixed-RL, MCTS-RL, MCTS-LQR} x {Fixed-LQR, MCTS-LQR, Fixed-RL} on Oval,
OvalDuos, Complex, ComplexDuos) this runs the CPU oracle on the same set-up — same Agents[] order, wiring, gameParams, rules,
orderings e % A!, and for LowMode == RL agents the reference's own trained actors (tests/golden/reference_actors.npz) —
writes the races in the reference's log grammar, reads both logs back with the same in-repo reader
(hierarchicalkarting_amd/telemetry.py) and prints the statistics side by side.  No reference code is executed; the logs,
scenes and .onnx files are read as data (build container only — /root/reference does not exist on the GPU box).

  --update     store the reference-side statistics in tests/golden/reference_log_stats.json and the oracle-side results
               (statistics + a hash of every race's hk_episode_result) in tests/golden/experiment_oracle.json; the GPU
               parity test compares libhk's races with those hashes, the CPU test re-derives them for a subset
  --markdown   print the residual table of DESIGN.md
  --only a,b   substrings of experiment names"""
import argparse, hashlib, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hierarchicalkarting_amd import telemetry as T          # noqa: E402
import experiments as X                                     # noqa: E402

REF_LOGS = "/root/reference/ExperimentLogs"
STATS = ("races", "wins", "dnfs", "mean_total_time", "median_best_lap", "collisions_per_race", "illegal_lane_changes_per_race",
         "mean_lane_difference", "mean_vel_difference")


def setup_names():
    """the set-ups that have a log of the same name in the reference (the latest generation: suffix 2 / 3)"""
    return [n for n in sorted(X.experiments()) if n.endswith(("2", "3"))]


def results_hash(res):
    h = hashlib.sha256()
    for name in res.dtype.names:
        h.update(res[name].tobytes())
    return h.hexdigest()


def run_ours(name, env_cls, mcts_iterations=128):
    s = X.Setup(name, mcts_iterations=mcts_iterations)
    res = s.run(env_cls)
    with tempfile.TemporaryDirectory() as d:
        stats = s.stats(res, os.path.join(d, "ours.txt"))
    return res, stats


def fmt(v):
    return "-" if v is None else ("%d" % v if isinstance(v, int) else "%.3f" % v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--update", action="store_true")
    ap.add_argument("--markdown", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--mcts-iterations", type=int, default=128)
    a = ap.parse_args()
    import oracle_lib as O
    gref = os.path.join(ROOT, "tests", "golden", "reference_log_stats.json")
    gora = os.path.join(ROOT, "tests", "golden", "experiment_oracle.json")
    ref_all = json.load(open(gref)) if os.path.exists(gref) else {}
    ora_all = json.load(open(gora)) if os.path.exists(gora) else {}
    rows = []
    for name in setup_names():
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        path = os.path.join(REF_LOGS, name + ".txt")
        if os.path.exists(path):
            ref_all[name] = {"log": "ExperimentLogs/%s.txt" % name, "stats": T.summarize_log(T.read_experiment_log(path))}
        if name not in ref_all:
            continue
        t0 = time.time()
        res, ours = run_ours(name, O.OracleEnv, a.mcts_iterations)
        ora_all[name] = {"mcts_iterations": a.mcts_iterations, "results_sha256": results_hash(res), "stats": ours}
        ref = ref_all[name]["stats"]
        print("== %s  (reference: %s; oracle %.1f s)" % (name, ref_all[name]["log"], time.time() - t0))
        for typ in ours:
            r = ref.get(typ, {})
            print("  %-10s %-32s %12s %12s" % (typ, "", "reference", "oracle"))
            for k in STATS:
                rv, ov = r.get(k), ours[typ][k]
                ratio = "" if not isinstance(rv, (int, float)) or not isinstance(ov, (int, float)) or not rv else "  x%.3f" % (ov / rv)
                print("  %-10s %-32s %12s %12s%s" % ("", k, fmt(rv), fmt(ov), ratio))
            rows.append((name, typ, r, ours[typ]))
        sys.stdout.flush()
    if a.markdown:
        print("| experiment | agent | wins | DNFs | mean total time [s] | median best lap [s] | collisions / race | illegal lane changes / race | lane difference [m] |")
        print("|---|---|---|---|---|---|---|---|---|")
        for name, typ, r, o in rows:
            c = lambda k: "%s / %s" % (fmt(r.get(k)), fmt(o.get(k)))
            print("| %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (name, typ, c("wins"), c("dnfs"), c("mean_total_time"), c("median_best_lap"),
                                                                   c("collisions_per_race"), c("illegal_lane_changes_per_race"), c("mean_lane_difference")))
    if a.update:
        json.dump(ref_all, open(gref, "w"), indent=1, sort_keys=True)
        json.dump(ora_all, open(gora, "w"), indent=1, sort_keys=True)
        print("wrote", gref, "and", gora)


if __name__ == "__main__":
    main()
