"""Experiment-log wire format of the reference: the text block TelemetryViewer.Update builds
(TelemetryViewer.cs:90-104) and RacingEnvController appends per finished experiment (REC:249-265,289-305), so that the
reference's own experiment_log_parser.py reads our races unchanged.  Pure host-side formatting of hk_episode_result."""
import numpy as np


def _f(x):
    """System.Single.ToString(): shortest round-trip decimal of a float32"""
    x = np.float32(x)
    if x == 0:
        return "0"
    s = np.format_float_positional(x, unique=True, trim="-")
    if len(s.replace("-", "").replace(".", "").strip("0")) > 9 or abs(float(x)) >= 1e7 or (abs(float(x)) < 1e-4):
        s = np.format_float_scientific(x, unique=True, trim="-", exp_digits=2).replace("e", "E")
    return s


def winner_of(names, res):
    """TelemetryViewer.cs:53-88 (note the reference compares lastEpisodeSteps against seconds)"""
    min_times = 1000.0
    winner = ""
    for i, name in enumerate(names):
        inactive = not bool(res["active"][i])
        if inactive and res["lap_end_step"][i] < min_times and winner != "Tie":
            winner = name
            min_times = float(res["total_time"][i])
        elif inactive and float(res["total_time"][i]) == min_times:
            winner = "Tie"
    return winner


def telemetry_block(names, res, laps):
    """res: hk_episode_result row [A] of one env -> the TelemetryViewer text (ends with a newline, like AppendLine)"""
    out = []
    for i, n in enumerate(names):
        out.append("%s Speed: %s" % (n, _f(res["speed"][i])))
        out.append("%s Reward: %s" % (n, _f(res["reward"][i])))
        out.append("%s Last Lap: %s" % (n, _f(res["last_lap"][i])))
        out.append("%s Best Lap: %s" % (n, _f(res["best_lap"][i])))
        out.append("%s Total Time: %s" % (n, _f(res["total_time"][i])))
        out.append("%s Laps Completed: %d/%d" % (n, int(res["laps_completed"][i]), laps))
        out.append("%s Illegal Lane Changes: %d" % (n, int(res["illegal_lane_changes"][i])))
        out.append("%s Collisions: %d" % (n, int(res["forward_collisions"][i])))
        out.append("%s Avg Target Lane Difference: %s" % (n, _f(res["avg_lane_diff"][i])))
        out.append("%s Avg Target Vel Difference: %s" % (n, _f(res["avg_vel_diff"][i])))
    out.append("Winner: " + winner_of(names, res))
    return "\n".join(out) + "\n"


class ExperimentLog:
    """REC:261-264: StreamWriter(append): WriteLine("Experiment " + experimentNum); WriteLine(tm.uiText.text)"""

    def __init__(self, path, names, laps, truncate=True):
        self.path, self.names, self.laps = path, list(names), laps
        if truncate:                      # REC:271-275: the file is emptied when the first experiment starts
            open(path, "w").close()

    def append(self, experiment_num, res_row):
        with open(self.path, "a") as f:
            f.write("Experiment %d\n" % experiment_num)
            f.write(telemetry_block(self.names, res_row, self.laps) + "\n")


# ---- reading the format back (ours or the reference's own ExperimentLogs/*.txt: the same grammar) -----------------------
_FIELDS = {"Speed": float, "Reward": float, "Last Lap": float, "Best Lap": float, "Total Time": float, "Overall Time": float,
           "Illegal Lane Changes": int, "Collisions": int, "Avg Target Lane Difference": float, "Avg Target Vel Difference": float}


def read_experiment_log(path):
    """-> [ {"experiment": k, "winner": str, "agents": {name: {field: value, "laps": (done, total)}}} ] in file order.
    Grammar (REC:261-264 + TelemetryViewer.cs:90-104): a line "Experiment <k>", then per agent lines "<name> <Field>: <value>"
    (the name has no spaces), then "Winner: <name or empty>", then a blank line.  Older logs say "Overall Time" for "Total Time"."""
    out, cur = [], None
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("Experiment "):
                cur = {"experiment": int(line.split()[1]), "winner": "", "agents": {}}
                out.append(cur)
                continue
            if cur is None or not line.strip():
                continue
            if line.startswith("Winner:"):
                cur["winner"] = line[len("Winner:"):].strip()
                continue
            name, _, rest = line.partition(" ")
            field, _, val = rest.rpartition(": ")
            ag = cur["agents"].setdefault(name, {})
            if field == "Laps Completed":
                done, _, tot = val.partition("/")
                ag["laps"] = (int(done), int(tot))
            elif field in _FIELDS:
                ag["Total Time" if field == "Overall Time" else field] = _FIELDS[field](float(val))
    return out


def summarize_log(records):
    """per agent type (the name up to "("): races, finishes, wins (fastest Total Time among the finishers of a race), DNFs,
    mean Total Time / Best Lap of the finishers, collisions and illegal lane changes per race, and the plan-tracking averages
    (KartAgent.AverageLaneDifference / AverageVelDifference, KA:226-239) over all races"""
    import statistics
    acc = {}
    for rec in records:
        fin = {n: a for n, a in rec["agents"].items() if a.get("laps", (0, 1))[0] == a.get("laps", (0, 1))[1]}
        win = min(fin, key=lambda n: fin[n]["Total Time"]) if fin else None
        for n, a in rec["agents"].items():
            t = acc.setdefault(n.split("(")[0], {"races": 0, "wins": 0, "dnfs": 0, "total": [], "best": [], "coll": [], "illegal": [], "lane": [], "vel": []})
            t["races"] += 1
            t["coll"].append(a.get("Collisions", 0)); t["illegal"].append(a.get("Illegal Lane Changes", 0))
            t["lane"].append(a.get("Avg Target Lane Difference", 0.0)); t["vel"].append(a.get("Avg Target Vel Difference", 0.0))
            if n in fin:
                t["total"].append(a["Total Time"]); t["best"].append(a["Best Lap"])
            else:
                t["dnfs"] += 1
            if n == win:
                t["wins"] += 1
    return {k: {"races": v["races"], "wins": v["wins"], "dnfs": v["dnfs"],
                "mean_total_time": statistics.fmean(v["total"]) if v["total"] else None,
                "median_best_lap": statistics.median(v["best"]) if v["best"] else None,
                "collisions_per_race": statistics.fmean(v["coll"]), "illegal_lane_changes_per_race": statistics.fmean(v["illegal"]),
                "mean_lane_difference": statistics.fmean(v["lane"]), "mean_vel_difference": statistics.fmean(v["vel"])}
            for k, v in acc.items()}
