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
