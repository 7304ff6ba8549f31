"""Closed-loop pin against the only outcome data the reference holds: ExperimentLogs/<ExperimentName>.txt.

What is compared.  The reference's Compete scenes define 22 experiment set-ups whose agents are all HierarchicalKartAgents
({Fixed-RL, MCTS-RL, MCTS-LQR} against {Fixed-LQR, MCTS-LQR, Fixed-RL} on Oval, OvalDuos, Complex, ComplexDuos), each with a log
of 50 (1v1) or 48 (2v2) races.  tests/golden/reference_experiments.json holds the set-ups as resolved from the scenes
(tools/extract_experiments.py), tests/golden/reference_actors.npz the trained actors the LowMode == RL agents run
(tools/make_actor_fixtures.py), tests/golden/reference_log_stats.json the statistics of the reference's logs and
tests/golden/experiment_oracle.json the CPU oracle's races of the same set-ups (statistics + a hash of every
hk_episode_result; tools/compare_experiment_logs.py --update, in the build container).  No reference code runs; the logs,
scenes and .onnx files are data.

Here (CPU): (i) the stored oracle statistics of ALL 22 set-ups must fall in bands around the reference's (+-3 % on pace (BANDS), the
lane-tracking metric within 0.75 - 1.5 x (LANE_DIFF_BAND), the speed at the Triggers within -0.15 / +0.45 m/s (FIXED_VEL_DIFF_BAND): no written-down residual since round 4); (ii) for 12 set-ups the oracle is run again
and must reproduce the stored hashes, so the stored statistics are the oracle's; (iii) the headline facts the reference's
own actors establish: a trained actor driven through our observation layout and kart model beats the LQNG controller as it
does in the reference (43 / 7 there), and laps within 1 % of its reference pace.
On the GPU (tests/test_experiments_gpu.py) libhk runs all 22 set-ups and must reproduce the stored hashes bit for bit.

Why the residuals are not zero: the engine is a restatement (PhysX has no source), the planner's budget is
iterations, not wall-clock, and Barracuda's random stream is not reproducible."""
import json
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_log_stats.json")))
ORA = json.load(open(os.path.join(ROOT, "tests", "golden", "experiment_oracle.json")))
ALL = sorted(ORA)

# statistic -> (low, high) of oracle / reference.  Round 4 (engine restatement with the WheelColliders' tire forces and contact yaw
# response, KartAgent.Sensors[] in the scenes' order, the planner's full action list): every one of the 44 agent rows inside +-3 % (worst 2.54 %).
BANDS = {"median_best_lap": (0.97, 1.03), "mean_total_time": (0.97, 1.03)}
# The reference's lane-tracking metric (KartAgent.AverageLaneDifference, KA:226-239: distance to the target lane marker when a
# Trigger is entered, minus 1.3 m): 0.8 - 1.45 x the reference's over the 44 rows (rounds 1 - 3, without tire side forces: 1.5 - 6 x)
LANE_DIFF_BAND = (0.75, 1.5)
# AverageVelDifference (KA:235-239: speed minus planned speed at the Triggers) of the agents that follow a FIXED plan (15 m/s everywhere):
# how fast the karts really are where the reference measures it; oracle - reference, m/s
FIXED_VEL_DIFF_BAND = (-0.15, 0.45)
# Known residuals (experiment, agent type) -> what is asserted instead, and why.  None: the stalls of rounds 1 - 3 were the wrong Sensors[] order.
RESIDUALS = {}
# set-ups the CPU suite re-runs on the oracle (all six 1v1 Oval, three 1v1 Complex, three 2v2): ~2.5 min on 8 cores
CPU_RERUN = ["Fixed_RL_vs_Fixed_LQR_Oval2", "Fixed_RL_vs_MCTS_LQR_Oval2", "MCTS_LQR_vs_Fixed_LQR_Oval2", "MCTS_RL_vs_Fixed_LQR_Oval2",
             "MCTS_RL_vs_Fixed_RL_Oval2", "MCTS_RL_vs_MCTS_LQR_Oval2", "Fixed_RL_vs_Fixed_LQR_Complex2", "MCTS_LQR_vs_Fixed_LQR_Complex3",
             "MCTS_RL_vs_Fixed_RL_Complex2", "MCTS_LQR_vs_Fixed_LQR_OvalDuos2", "MCTS_LQR_vs_Fixed_LQR_ComplexDuos2", "MCTS_RL_vs_Fixed_LQR_OvalDuos2"]


def test_the_22_setups_of_the_reference_scenes_have_logs_and_oracle_races():
    assert len(ALL) == 22 and set(ALL) <= set(REF)
    assert len([n for n in ALL if "Complex" in n]) == 11 and len([n for n in ALL if "Duos" in n]) == 10
    for n in ALL:
        assert set(ORA[n]["stats"]) == set(REF[n]["stats"]), n
        for typ, s in ORA[n]["stats"].items():
            assert s["races"] == REF[n]["stats"][typ]["races"], (n, typ)


def test_golden_reference_stats_are_the_survey_numbers():
    """the committed reference-side statistics reproduce SURVEY §6 / BASELINE.md §1 (so the reader parses the reference's logs as the
    reference's own parser does)"""
    s = REF["MCTS_LQR_vs_Fixed_LQR_Oval2"]["stats"]
    assert s["Fixed-LQR"]["races"] == 50 and s["MCTS-LQR"]["wins"] == 24 and s["Fixed-LQR"]["wins"] == 26
    assert abs(s["Fixed-LQR"]["mean_total_time"] - 79.45) < 0.01 and abs(s["Fixed-LQR"]["median_best_lap"] - 18.62) < 0.005
    assert abs(s["MCTS-LQR"]["mean_total_time"] - 81.40) < 0.01 and abs(s["Fixed-LQR"]["collisions_per_race"] - 0.24) < 1e-9
    d = REF["MCTS_LQR_vs_Fixed_LQR_OvalDuos2"]["stats"]
    assert d["Fixed-LQR"]["races"] == 96 and abs(d["Fixed-LQR"]["mean_total_time"] - 82.54) < 0.01 and abs(d["MCTS-LQR"]["median_best_lap"] - 19.88) < 0.005
    c = REF["MCTS_LQR_vs_Fixed_LQR_Complex3"]["stats"]
    assert abs(c["Fixed-LQR"]["mean_total_time"] - 101.90) < 0.01 and abs(c["Fixed-LQR"]["median_best_lap"] - 32.84) < 0.005
    r = REF["Fixed_RL_vs_Fixed_LQR_Oval2"]["stats"]
    assert r["Fixed-RL"]["wins"] == 43 and r["Fixed-LQR"]["wins"] == 7


@pytest.mark.parametrize("name", ALL)
def test_oracle_races_fall_in_the_reference_bands(name):
    ref, ours = REF[name]["stats"], ORA[name]["stats"]
    assert not RESIDUALS
    for typ in ours:
        o, r = ours[typ], ref[typ]
        # every agent finishes at least as reliably as in the reference (whose DNFs are karts stuck on PhysX geometry: 0 - 11 of 50 on Complex)
        assert o["dnfs"] <= max(r["dnfs"], 2), (name, typ, o["dnfs"], r["dnfs"])
        for k, (lo, hi) in BANDS.items():
            ratio = o[k] / r[k]
            assert lo <= ratio <= hi, (name, typ, k, o[k], r[k])
        # forward collisions stay rare events, as in the reference (0.18 .. 2.4 per race there)
        assert o["collisions_per_race"] < 3.5
        # illegal lane changes: a handful per race at most, as in the reference (0 .. 3 there)
        assert o["illegal_lane_changes_per_race"] < 6.0
        ratio = o["mean_lane_difference"] / r["mean_lane_difference"]
        assert LANE_DIFF_BAND[0] <= ratio <= LANE_DIFF_BAND[1], (name, typ, o["mean_lane_difference"], r["mean_lane_difference"])
        if typ.startswith("Fixed"):
            d = o["mean_vel_difference"] - r["mean_vel_difference"]
            assert FIXED_VEL_DIFF_BAND[0] <= d <= FIXED_VEL_DIFF_BAND[1], (name, typ, o["mean_vel_difference"], r["mean_vel_difference"])
    # every race has one winner here (the reference's totals fall short of the race count where every kart of a race got stuck)
    races = max(o["races"] for o in ours.values()) // (2 if "Duos" in name else 1)
    assert sum(ours[t]["wins"] for t in ours) == races


def test_planned_speed_is_followed_as_in_the_reference():
    """AverageVelDifference of the MCTS-RL agents — the actor is told the planned speed of the next sections (HKA:530-552) and the metric is
    its speed minus that plan at the Triggers: +1.3 .. +2.4 m/s in the reference's logs.  With the planner's action list cut at 20 entries
    (rounds 1 - 3; these agents' velocityBucketSize is 1: 9 speeds x 4 lanes = 36 actions, KDG:329-338) no plan exceeded 11 m/s and the
    metric read +4.9; with the full list it is the reference's."""
    n = 0
    for name in ALL:
        for typ, o in ORA[name]["stats"].items():
            if typ == "MCTS-RL" and "Oval" in name:
                r = REF[name]["stats"][typ]
                assert abs(o["mean_vel_difference"] - r["mean_vel_difference"]) < 0.6, (name, o["mean_vel_difference"], r["mean_vel_difference"])
                n += 1
    assert n >= 4


def test_trained_actors_of_the_reference_drive_and_win_as_in_the_reference():
    """The end-to-end check of the observation layout (HKA:485-604), the action decoding (HKA:1371-1379) and the kart model: the
    reference's own trained actors, fed our observations, race at their reference pace and beat the LQNG agents as they do there."""
    def g(name, typ, side=ORA):
        return side[name]["stats"][typ]
    # Oval 1v1, Fixed-RL (actor FixedHierarchicalAgent-NonLSTM-allsolo10): reference 43 wins of 50, mean 80.28 s, best lap 19.18 s
    assert g("Fixed_RL_vs_Fixed_LQR_Oval2", "Fixed-RL")["wins"] >= 33
    assert abs(g("Fixed_RL_vs_Fixed_LQR_Oval2", "Fixed-RL")["mean_total_time"] / g("Fixed_RL_vs_Fixed_LQR_Oval2", "Fixed-RL", REF)["mean_total_time"] - 1) < 0.02
    # Oval 1v1, MCTS-RL (actor HierarchicalAgent-NonLSTM-allsolo6): reference 47 / 3 against Fixed-LQR
    assert g("MCTS_RL_vs_Fixed_LQR_Oval2", "MCTS-RL")["wins"] >= 40
    for n in ("MCTS_RL_vs_Fixed_LQR_Oval2", "MCTS_RL_vs_Fixed_RL_Oval2", "MCTS_RL_vs_MCTS_LQR_Oval2"):
        o, r = g(n, "MCTS-RL"), g(n, "MCTS-RL", REF)
        assert abs(o["median_best_lap"] / r["median_best_lap"] - 1) < 0.012 and o["dnfs"] == 0
        assert o["illegal_lane_changes_per_race"] < 0.5
    # the sampled team actors (sigma 0.76 - 0.85) finish every Oval race, as in the reference (rounds 1 - 3: they stalled against a wall in
    # 71 - 96 of 96 races: their ray inputs were in the prefab's order, not the scenes')
    for n in ("Fixed_RL_vs_Fixed_LQR_OvalDuos2", "Fixed_RL_vs_MCTS_LQR_OvalDuos2"):
        assert g(n, "Fixed-RL")["dnfs"] == 0 and abs(g(n, "Fixed-RL")["median_best_lap"] / g(n, "Fixed-RL", REF)["median_best_lap"] - 1) < 0.02
        assert abs(g(n, "Fixed-RL")["wins"] - g(n, "Fixed-RL", REF)["wins"]) <= 8
    # Complex 1v1: MCTS-RL wins all 50 against Fixed-RL on both sides; 2v2 Oval: the MCTS-RL team (actor TeamDOE-all28) takes 37 of 48 there, 38 here
    assert g("MCTS_RL_vs_Fixed_RL_Complex2", "MCTS-RL")["wins"] == 50 == g("MCTS_RL_vs_Fixed_RL_Complex2", "MCTS-RL", REF)["wins"]
    for n in ("MCTS_RL_vs_Fixed_LQR_OvalDuos2", "MCTS_RL_vs_MCTS_LQR_OvalDuos2"):
        assert abs(g(n, "MCTS-RL")["wins"] - g(n, "MCTS-RL", REF)["wins"]) <= 6 and g(n, "MCTS-RL")["dnfs"] == 0
    # the planner is worth something against the fixed plan, as in the reference: a third of the races at least
    for n in ("MCTS_LQR_vs_Fixed_LQR_Oval2", "MCTS_LQR_vs_Fixed_LQR_OvalDuos2", "MCTS_LQR_vs_Fixed_LQR_Complex3", "MCTS_LQR_vs_Fixed_LQR_ComplexDuos2"):
        tot = g(n, "MCTS-LQR")["wins"] + g(n, "Fixed-LQR")["wins"]
        assert g(n, "MCTS-LQR")["wins"] >= tot // 3
    # The fixed plan follows each scene's own DiscretePositionTracker.optimalLane (tests/experiments.py: the 1v1 scenes' copies of the Oval keep
    # to lanes 3 / 2 where the track fixture says 4 / 3).  With the fixture's lanes the Fixed-LQR kart swerved on every lap of the 1v1 Oval
    # (3.5 - 3.9 illegal lane changes a race; reference 0.00 - 0.02) and lost 35 : 15 to the planner; with the scene's: 0.0, and 31 : 19
    # (reference 24 : 26).  The 2v2 Oval scenes do use lanes 4 / 3, and there the reference swerves too (2.1 - 3.0 a race; here 3.2 - 3.5).
    for n in ("Fixed_RL_vs_Fixed_LQR_Oval2", "MCTS_LQR_vs_Fixed_LQR_Oval2", "MCTS_RL_vs_Fixed_LQR_Oval2"):
        assert g(n, "Fixed-LQR")["illegal_lane_changes_per_race"] <= 0.1, (n, g(n, "Fixed-LQR")["illegal_lane_changes_per_race"])
    assert abs(g("MCTS_LQR_vs_Fixed_LQR_Oval2", "MCTS-LQR")["wins"] - g("MCTS_LQR_vs_Fixed_LQR_Oval2", "MCTS-LQR", REF)["wins"]) <= 8
    # win counts over all 22 set-ups: the side that wins in the reference wins here in all but the closest match-ups (a handful of races apart)
    agree = sum((ORA[n]["stats"][a]["wins"] > ORA[n]["stats"][b]["wins"]) == (REF[n]["stats"][a]["wins"] > REF[n]["stats"][b]["wins"])
                for n in ALL for a, b in [tuple(ORA[n]["stats"])])
    assert agree >= 18, agree


@pytest.mark.parametrize("name", CPU_RERUN)
def test_oracle_reproduces_the_stored_races(name):
    import compare_experiment_logs as CE
    import oracle_lib as O
    res, stats = CE.run_ours(name, O.OracleEnv, ORA[name]["mcts_iterations"])
    assert CE.results_hash(res) == ORA[name]["results_sha256"]
    assert stats == ORA[name]["stats"]


def test_older_generation_logs_as_a_second_hold_out():
    """The reference's OLDER ExperimentLogs (no 2 / 3 suffix; tools/older_generation_logs.py) were not looked at when the engine restatement was
    built.  Their scenes are gone, so each is compared with our races of the latest set-up of the same name — meaningful for the four LQNG-only
    match-ups (no trained actor whose checkpoint changed).  The reference's own two generations differ by up to 3 % in best lap (Fixed-LQR on
    the Oval: 19.19 -> 18.62 s); ours sits within 2.5 % of the older generation (best lap 2.1 %, total time 2.5 %) and inside a 4 % band of both."""
    old = json.load(open(os.path.join(ROOT, "tests", "golden", "older_generation_log_stats.json")))
    n = 0
    for base, rec in old.items():
        if not rec["lqng_only"]:
            continue
        for typ, o in rec["stats"].items():
            u, r = ORA[rec["latest"]]["stats"][typ], REF[rec["latest"]]["stats"][typ]
            for k in ("median_best_lap", "mean_total_time"):
                assert 0.96 <= u[k] / o[k] <= 1.04, (base, typ, k, u[k], o[k])
                assert 0.96 <= r[k] / o[k] <= 1.04, ("the reference against itself", base, typ, k)
            assert u["dnfs"] <= max(o["dnfs"], 2)
            n += 1
    assert n == 8
