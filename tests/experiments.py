"""The reference's experiment set-ups (Experiment mode of RacingEnvController, REC:239-321, 499-719) as data, and one runner
for them that works with either side of the parity tests (the CPU oracle or libhk's RacingEnv).

tests/golden/reference_experiments.json  the set-ups resolved from the reference's Compete scenes (tools/extract_experiments.py):
                                         Agents[] order, HighMode / LowMode, gameParams, team wiring, rules, the actor each
                                         LowMode == RL agent runs and its StackingSensor depth, DecisionPeriod
tests/golden/reference_actors.npz        those trained actors as float32 arrays (tools/make_actor_fixtures.py)
tests/golden/reference_log_stats.json    statistics of the reference's own ExperimentLogs/<ExperimentName>.txt

Nothing here reads /root/reference (it does not exist on the GPU box)."""
import copy
import json
import os
import numpy as np
from hierarchicalkarting_amd import _lib, telemetry as T
from hierarchicalkarting_amd.config import make_config, load_track
from hierarchicalkarting_amd.policy import Policy

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_EXPS = None
_ACTORS = None


def experiments():
    """ExperimentName -> env record, for the set-ups every agent of which is a HierarchicalKartAgent (EndToEnd agents are
    out of scope).  Several scenes may hold an env of the same name (the *All scenes and the single ones):
    the *All scene's copy — the one the logs with the same suffix were written from — wins."""
    global _EXPS
    if _EXPS is None:
        out = {}
        for e in json.load(open(os.path.join(GOLD, "reference_experiments.json"))):
            name = e.get("ExperimentName")
            if not name or any(a.get("script") != "HierarchicalKartAgent.cs" for a in e["agents"]):
                continue
            if name not in out or e["scene"].endswith("All.unity"):
                out[name] = e
        _EXPS = out
    return _EXPS


def actor(model, stack, seed):
    global _ACTORS
    if _ACTORS is None:
        _ACTORS = np.load(os.path.join(GOLD, "reference_actors.npz"))
    return Policy.from_arrays(_ACTORS, model + "/", stack=stack, deterministic=False, seed=seed)


class Setup:
    def __init__(self, name, mcts_iterations=128, n_exp=None, seed=0):
        e = experiments()[name]
        ag = e["agents"]
        self.name, self.env = name, e
        self.A = len(ag)
        self.track = "oval" if "Oval" in e["scene"] else "complex"
        # every RacingEnvController of the reference's scenes owns a copy of the track, and the copies differ in DiscretePositionTracker.optimalLane
        # (what planFixed follows, HKA:145-152): the 1v1 set-ups of CompeteAgents-OvalAll / -ComplexAll are not the track fixture's lanes
        track = copy.deepcopy(load_track(self.track))
        if e.get("optimal_lanes"):
            assert len(e["optimal_lanes"]) == len(track["sections"])
            for sec, lane in zip(track["sections"], e["optimal_lanes"]):
                sec["optimalLane"] = int(lane)
        self.names = [a["name"] for a in ag]
        self.n_exp = int(n_exp if n_exp is not None else e["TotalExperiments"])
        team_of = [0] * self.A
        for t, members in enumerate(e["teams"]):
            for m in members:
                team_of[m] = t
        gp = [a["gameParams"] for a in ag]
        self.built = make_config(
            self.n_exp, self.A, track=track, high_mode=[a["HighMode"] for a in ag], low_mode=[a["LowMode"] for a in ag],
            tree_search_depth=[g["treeSearchDepth"] for g in gp], velocity_bucket_size=[g["velocityBucketSize"] for g in gp],
            time_precision=[g["timePrecision"] for g in gp], section_window=[g["sectionWindow"] for g in gp],
            wiring=(team_of, [a["teamAgents"] for a in ag], [a["otherAgents"] for a in ag]),
            laps=e["laps"], max_episode_steps=e["maxEpisodeSteps"], max_lane_changes=e["MaxLaneChanges"], disable_on_end=e["disableOnEnd"],
            jitter_seed=0, auto_reset=0, mcts_iterations=mcts_iterations, mcts_seed=0x4D435453 + seed, sensors=ag[0]["sensors"])
        assert all(a["sensors"] == ag[0]["sensors"] for a in ag)     # one Sensors[] layout per scene (hk_config holds one)
        assert self.built.cfg.section_horizon == e["sectionHorizon"]
        # one attached policy per distinct (actor, stack, period): BehaviorParameters.m_Model / DecisionRequester of each RL agent
        groups = {}
        for i, a in enumerate(ag):
            if a["LowMode"] == _lib.HK_LOW_RL:
                b = a["behavior"]
                groups.setdefault((b["model"], int(b["stacked"]), int(a["decision_period"])), []).append(i)
        self.policies = [(actor(m, st, seed * 16 + k + 1), slots, period) for k, ((m, st, period), slots) in enumerate(sorted(groups.items()))]

    def start(self, env_cls):
        env = env_cls(self.built)
        for pol, slots, period in self.policies:
            env.attach_policy(pol, slots, period)
        env.reset()                               # experiment e starts from ordering e % A! (REC:528-530)
        return env

    def run(self, env_cls, chunk=100):
        """-> hk_episode_result[n_exp][A] of the finished races (REC:249-265: read on the tick after the last agent went inactive)"""
        env = self.start(env_cls)
        full = (1 << self.A) - 1
        for _ in range((self.built.cfg.max_episode_steps + 200) // chunk + 1):
            env.step(chunk)
            if (env.env_state()["inactive_mask"] == full).all():
                break
        env.step(1)
        res = env.episode_results()
        env.close()
        return res

    def stats(self, res, log_path):
        log = T.ExperimentLog(log_path, self.names, self.built.cfg.laps)
        for e in range(self.n_exp):
            log.append(e, res[e])
        return T.summarize_log(T.read_experiment_log(log_path))
