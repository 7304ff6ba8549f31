"""Build an hk_config (include/hk.h) from the committed track table + the constants of SURVEY App. A.

Pure host-side data plumbing; shared by the product host API (env.py) and, in tests, by the oracle bindings."""
import ctypes as C
import json
import os
from . import _lib

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# KartAgent.Sensors[] as the Compete SCENES hold it: index -> yaw of the MLAgent_Sensors child the entry points at (positive = right).
# The scenes re-point the entries of every HierarchicalKartAgent (113 of 115; tools/extract_experiments.py resolves the references;
# tests/golden/reference_experiments.json "sensors"), so the order is NOT the prefab's {0, 30, 60, 90, -30, -60, -90, 45, -45} that
# SURVEY App. A lists: SolveLQR's rays 0, 2, 4, 8, 6 (HKA:834-844) are 0, +45, +90, -90, -45 degrees, KA.FixedUpdate's 0, 1, 5
# (KA:138-143) are 0, +30, -30, and the validation distances below are mirror-symmetric as they should be.  The running means of the
# reference's trained actors' input normalisers (ray 4: 5.0 m, ray 8: 4.7 m = half the track width) say the same.
SENSOR_YAW_DEG = [0.0, 30.0, 45.0, 60.0, 90.0, -30.0, -45.0, -60.0, -90.0]
RAY_DISTANCE = [20.0] * 9
WALL_HIT_VALIDATION = [0.8, 0.9, 1.0, 0.8, 0.6, 0.9, 1.0, 0.8, 0.6]
AGENT_HIT_VALIDATION = [1.5] * 9

# effective ArcadeKart.Stats of the Compete scenes (class defaults <- prefabs <- scene overrides)
KART_STATS = dict(TopSpeed=15.0, Acceleration=7.0, ReverseSpeed=10.0, ReverseAcceleration=3.0, AccelerationCurve=0.5,
                  Braking=16.0, CoastingDrag=5.0, Grip=0.97, MaxSteer=4.0, MinSteer=1.0, TireWearFactor=0.001,
                  MinGs=0.5, MaxGs=2.0, AddedGravity=1.0, TireWearRate=10000.0, AngularDrag=0.05)


# Engine restatement (include/hk.h hk_engine_params): Rigidbody / CapsuleCollider / WheelCollider / KartAnimation values of
# BaseKartClassic.prefab; inertia_y = the capsule's (r 0.45, h 2) transverse inertia at mass 250; side_slope0 is the one fitted number
ENGINE_PARAMS = dict(mass=250.0, inertia_y=74.71, gravity=9.81, axle_zf=0.58625615, axle_zr=-0.68122816, max_steer_deg=30.0,
                     steer_damping=10.0, side_ext_slip=0.2, side_ext_value=1.0, side_asy_slip=0.5, side_asy_value=0.75,
                     side_stiffness=1.0, side_slope0=1.0, slip_min_speed=1.0, wheel_mass=20.0, wheel_radius_f=0.1372984,
                     wheel_radius_r=0.1630791, wheel_damping=0.56, fwd_ext_slip=0.4, fwd_ext_value=1.0, fwd_asy_slip=0.8, fwd_asy_value=0.5,
                     fwd_stiffness=1.0, long_slip_min_speed=4.0, wheel_friction=1, contact_yaw=1, wheel_rolling=1)


def load_track(name="oval"):
    with open(os.path.join(_DATA, "%s_track.json" % name)) as f:
        return json.load(f)


def default_wiring(num_agents):
    """teamAgents / otherAgents as in the reference scenes: 1v1 (CompeteAgents-Oval) or 2v2 duos
    (CompeteAgents-OvalDuosAll: Agents = [M0, M1, F0, F1], Teams[1] = {M0, M1}, Teams[0] = {F0, F1})."""
    if num_agents == 1:
        return [0], [[]], [[]]
    if num_agents == 2:
        return [0, 1], [[], []], [[1], [0]]
    if num_agents == 4:
        return [1, 1, 0, 0], [[1], [0], [3], [2]], [[2, 3], [2, 3], [0, 1], [0, 1]]
    if num_agents == 8:
        # synthetic 4v4 (BASELINE configs[4]; the reference has no 8-agent scene): the duos pattern with four karts a side
        a, b = [0, 1, 2, 3], [4, 5, 6, 7]
        return ([1] * 4 + [0] * 4, [[j for j in (a if i < 4 else b) if j != i] for i in range(8)],
                [list(b if i < 4 else a) for i in range(8)])
    team_of = list(range(num_agents))
    return team_of, [[] for _ in range(num_agents)], [[j for j in range(num_agents) if j != i] for i in range(num_agents)]


class BuiltConfig:
    """hk_config plus the ctypes arrays it points into (kept alive with it)."""

    def __init__(self, cfg, sections, walls, track):
        self.cfg, self.sections, self.walls, self.track = cfg, sections, walls, track


# RacingEnvController.cs:65-108 field initialisers
REWARD_DEFAULTS = dict(
    WallHitPenalty=-0.05, OpponentHitPenalty=-2.0, HitByOpponentPenalty=-2.0, PassCheckpointLaneReward=4.0,
    PassCheckpointVelocityReward=4.0, PassCheckpointBase=20.0, PassCheckpointTimeMultiplier=5.0, TeamPassCheckpointBase=20.0,
    TeamPassCheckpointTimeMultiplier=5.0, BeingBehindOpponentCheckpointPenalty=-0.06, BeingBehindTeammateCheckpointPenalty=-0.02,
    TeamScoreRewardMultiplier=0.75, ReversePenalty=-0.5, SwervingPenalty=-0.5, ReachGoalCheckpointRewardMultplier=5.0,
    ReachGoalCheckpointRewardBase=3.0, TowardsCheckpointReward=0.008, SpeedReward=0.07, SlowMovingPenalty=-3.0,
    AccelerationReward=0.002, NotAtGoalPenalty=-0.001)


def make_config(num_envs, num_agents=4, track="oval", high_mode=_lib.HK_HIGH_FIXED, low_mode=_lib.HK_LOW_LQR,
                tree_search_depth=5, jitter_seed=0, jitter_pos=0.5, jitter_yaw=0.05, auto_reset=1, env_id_base=0,
                device_id=0, wiring=None, env_mode=_lib.HK_MODE_EXPERIMENT, max_episode_steps=None, laps=None,
                stats=None, time_precision=100, section_window=2, mcts_iterations=128, mcts_initial_iterations=None,
                mcts_latency_ticks=45, mcts_initial_latency_ticks=75, mcts_seed=0x4D435453, rewards=0, training_agents=None,
                reward_params=None, disable_on_end=None, train_seed=0x54524149, velocity_bucket_size=2, max_lane_changes=None,
                engine=None, sensors=None):
    tr = load_track(track) if isinstance(track, str) else track
    secs = tr["sections"]
    L = len(secs)
    sec_arr = (_lib.Section * L)()
    for i, s in enumerate(secs):
        d = sec_arr[i]
        d.trig_x, d.trig_z = s["Trigger"]["x"], s["Trigger"]["z"]
        d.yaw_deg = s["waypoint"]["yaw_deg"]
        d.marker_y = s["Trigger"]["y"]
        for l in range(4):
            d.lane_x[l] = s["Lane%d" % (l + 1)]["x"]
            d.lane_z[l] = s["Lane%d" % (l + 1)]["z"]
        d.track_inside_radius = s["trackInsideRadius"]
        d.track_length = s["trackLength"]
        d.track_width = s["trackWidth"]
        d.turn_degrees = s["turnDegrees"]
        d.left_turn = int(s["leftTurn"])
        d.optimal_lane = int(s["optimalLane"])
    segs = []
    for w in tr["walls"]:
        pts = w["points"]
        for a, b in zip(pts[:-1], pts[1:]):
            segs.append((a[0], a[1], b[0], b[1]))
    wall_arr = (_lib.WallSeg * max(len(segs), 1))()
    for i, (x0, z0, x1, z1) in enumerate(segs):
        wall_arr[i].x0, wall_arr[i].z0, wall_arr[i].x1, wall_arr[i].z1 = x0, z0, x1, z1
    cfg = _lib.Config()
    cfg.abi_version = _lib.HK_ABI_VERSION
    cfg.num_envs, cfg.num_agents, cfg.device_id = num_envs, num_agents, device_id
    team_of, team, other = wiring if wiring is not None else default_wiring(num_agents)
    for i in range(num_agents):
        cfg.team_of[i] = team_of[i]
        cfg.n_team[i] = len(team[i])
        for j, t in enumerate(team[i]):
            cfg.team_agents[i][j] = t
        cfg.n_other[i] = len(other[i])
        for j, t in enumerate(other[i]):
            cfg.other_agents[i][j] = t
        cfg.high_mode[i] = high_mode[i] if isinstance(high_mode, (list, tuple)) else high_mode
        cfg.low_mode[i] = low_mode[i] if isinstance(low_mode, (list, tuple)) else low_mode
        cfg.tree_search_depth[i] = tree_search_depth[i] if isinstance(tree_search_depth, (list, tuple)) else tree_search_depth
        per = lambda v: v[i] if isinstance(v, (list, tuple)) else v          # gameParams are per agent (HKA:38-52)
        cfg.velocity_bucket_size[i] = per(velocity_bucket_size)
        cfg.time_precision[i] = per(time_precision)
        cfg.section_window[i] = per(section_window)
    st = dict(KART_STATS)
    if stats:
        st.update(stats)
    for k, v in st.items():
        setattr(cfg.stats, k, v)
    eng = dict(ENGINE_PARAMS)
    if engine:
        eng.update(engine)
    for k, v in eng.items():
        setattr(cfg.engine, k, int(v) if k in ("wheel_friction", "contact_yaw", "wheel_rolling") else v)
    rules = tr["rules"]
    cfg.laps = int(laps if laps is not None else rules["laps"])
    cfg.max_episode_steps = int(max_episode_steps if max_episode_steps is not None else rules["maxEpisodeSteps"])
    cfg.max_lane_changes = int(rules["MaxLaneChanges"] if max_lane_changes is None else max_lane_changes)
    cfg.section_horizon = int(rules["sectionHorizon"])
    cfg.disable_on_end = int(rules["disableOnEnd"] if disable_on_end is None else disable_on_end)
    cfg.env_mode = env_mode
    # StartRaceAfterDelay waits 1.5 s except in Training mode (REC:723-724)
    cfg.start_hold_ticks = 0 if env_mode == _lib.HK_MODE_TRAINING else 75
    cfg.train_seed = train_seed
    cfg.auto_reset = auto_reset
    cfg.dt = 0.02
    cfg.kart_y = 0.28
    for i in range(_lib.HK_NUM_SENSORS):
        sn = sensors[i] if sensors else None     # one scene entry: {"yaw_deg", "RayDistance", "WallHitValidationDistance", "AgentHitValidationDistance"}
        cfg.sensor_yaw_deg[i] = sn["yaw_deg"] if sn else SENSOR_YAW_DEG[i]
        cfg.ray_distance[i] = sn["RayDistance"] if sn else RAY_DISTANCE[i]
        cfg.wall_hit_validation[i] = sn["WallHitValidationDistance"] if sn else WALL_HIT_VALIDATION[i]
        cfg.agent_hit_validation[i] = sn["AgentHitValidationDistance"] if sn else AGENT_HIT_VALIDATION[i]
    # MCTS planner budget: iterations stand for the reference's wall-clock T (0.9 s per replan, 1.5 s at reset)
    cfg.mcts_iterations = int(mcts_iterations)
    cfg.mcts_initial_iterations = int(mcts_initial_iterations if mcts_initial_iterations is not None else (mcts_iterations * 5 + 2) // 3)
    cfg.mcts_latency_ticks, cfg.mcts_initial_latency_ticks = int(mcts_latency_ticks), int(mcts_initial_latency_ticks)
    cfg.mcts_seed = mcts_seed
    # reward shaping (REC:65-108 defaults; scenes may override)
    cfg.rewards = int(rewards)
    for i in range(num_agents):
        cfg.training_agent[i] = int(training_agents[i]) if training_agents is not None else 0
    rw = dict(REWARD_DEFAULTS)
    if reward_params:
        rw.update(reward_params)
    for k, v in rw.items():
        setattr(cfg.rw, k, v)
    cfg.jitter_seed = jitter_seed
    cfg.jitter_pos, cfg.jitter_yaw = jitter_pos, jitter_yaw
    cfg.env_id_base = env_id_base
    cfg.num_sections, cfg.num_walls = L, len(segs)
    cfg.sections = C.cast(sec_arr, C.POINTER(_lib.Section))
    cfg.walls = C.cast(wall_arr, C.POINTER(_lib.WallSeg))
    return BuiltConfig(cfg, sec_arr, wall_arr, tr)
