"""ctypes binding of libhk.so (the C ABI in include/hk.h).  The library is the product; there is no Python or
CPU fallback: if it is missing or no GPU is present, calls raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HK_LIB_PATH") or os.path.join(_HERE, "libhk.so")   # HK_LIB_PATH: kernel-variant experiments (tools/)

HK_MAX_AGENTS = 8
HK_MAX_SECTIONS = 64
HK_NUM_SENSORS = 9
HK_ABI_VERSION = 5
HK_PROF_STAGES = 6
PROF_STAGE_NAMES = ("env_run_kernel", "lqn_kernel<2,3,4>", "lq_batch_kernel", "policy_mlp_kernel", "observe+stack", "env_b1_kernel")
HK_COMM_ID_BYTES = 128
HK_MAX_POLICIES = 4
HK_POLICY_MAX_LAYERS = 4
HK_POLICY_MAX_IN = 1280
HK_POLICY_MAX_HIDDEN = 256

HK_OK, HK_ERR_INVALID, HK_ERR_NO_DEVICE, HK_ERR_HIP, HK_ERR_UNSUPPORTED, HK_ERR_SINGULAR = 0, -1, -2, -3, -4, -5
HK_LOW_RL, HK_LOW_MPC, HK_LOW_LQR = 0, 1, 2
HK_HIGH_MCTS, HK_HIGH_FIXED = 0, 1
HK_MODE_RACE, HK_MODE_TRAINING, HK_MODE_EXPERIMENT = 0, 1, 2
HK_F_ACCEL, HK_F_BRAKE, HK_F_ACTIVE, HK_F_FORWARD_COLLISION, HK_F_HAS_COLLISION, HK_F_CAN_MOVE, HK_F_ENABLED = (1 << i for i in range(7))


class KartStats(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "TopSpeed", "Acceleration", "ReverseSpeed", "ReverseAcceleration", "AccelerationCurve", "Braking", "CoastingDrag",
        "Grip", "MaxSteer", "MinSteer", "TireWearFactor", "MinGs", "MaxGs", "AddedGravity", "TireWearRate", "AngularDrag")]


class Section(C.Structure):
    _fields_ = [("trig_x", C.c_float), ("trig_z", C.c_float), ("yaw_deg", C.c_float), ("marker_y", C.c_float),
                ("lane_x", C.c_float * 4), ("lane_z", C.c_float * 4),
                ("track_inside_radius", C.c_float), ("track_length", C.c_float), ("track_width", C.c_float),
                ("turn_degrees", C.c_float), ("left_turn", C.c_int32), ("optimal_lane", C.c_int32)]


class WallSeg(C.Structure):
    _fields_ = [("x0", C.c_float), ("z0", C.c_float), ("x1", C.c_float), ("z1", C.c_float)]


_I8 = C.c_int32 * HK_MAX_AGENTS


class RewardParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "WallHitPenalty", "OpponentHitPenalty", "HitByOpponentPenalty", "PassCheckpointLaneReward", "PassCheckpointVelocityReward",
        "PassCheckpointBase", "PassCheckpointTimeMultiplier", "TeamPassCheckpointBase", "TeamPassCheckpointTimeMultiplier",
        "BeingBehindOpponentCheckpointPenalty", "BeingBehindTeammateCheckpointPenalty", "TeamScoreRewardMultiplier",
        "ReversePenalty", "SwervingPenalty", "ReachGoalCheckpointRewardMultplier", "ReachGoalCheckpointRewardBase",
        "TowardsCheckpointReward", "SpeedReward", "SlowMovingPenalty", "AccelerationReward", "NotAtGoalPenalty")]


class EngineParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "mass", "inertia_y", "gravity", "axle_zf", "axle_zr", "max_steer_deg", "steer_damping", "side_ext_slip", "side_ext_value",
        "side_asy_slip", "side_asy_value", "side_stiffness", "side_slope0", "slip_min_speed", "wheel_mass", "wheel_radius_f",
        "wheel_radius_r", "wheel_damping", "fwd_ext_slip", "fwd_ext_value", "fwd_asy_slip", "fwd_asy_value", "fwd_stiffness",
        "long_slip_min_speed")] + [
        ("wheel_friction", C.c_int32), ("contact_yaw", C.c_int32), ("wheel_rolling", C.c_int32)]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("num_envs", C.c_int32), ("num_agents", C.c_int32), ("device_id", C.c_int32),
        ("team_of", _I8), ("n_team", _I8), ("team_agents", _I8 * HK_MAX_AGENTS), ("n_other", _I8),
        ("other_agents", _I8 * HK_MAX_AGENTS), ("high_mode", _I8), ("low_mode", _I8), ("tree_search_depth", _I8),
        ("velocity_bucket_size", _I8),
        ("stats", KartStats),
        ("laps", C.c_int32), ("max_episode_steps", C.c_int32), ("max_lane_changes", C.c_int32),
        ("section_horizon", C.c_int32), ("disable_on_end", C.c_int32), ("env_mode", C.c_int32),
        ("start_hold_ticks", C.c_int32), ("auto_reset", C.c_int32), ("dt", C.c_float), ("kart_y", C.c_float),
        ("sensor_yaw_deg", C.c_float * HK_NUM_SENSORS), ("ray_distance", C.c_float * HK_NUM_SENSORS),
        ("wall_hit_validation", C.c_float * HK_NUM_SENSORS), ("agent_hit_validation", C.c_float * HK_NUM_SENSORS),
        ("jitter_seed", C.c_uint32), ("jitter_pos", C.c_float), ("jitter_yaw", C.c_float), ("env_id_base", C.c_int32),
        ("num_sections", C.c_int32), ("num_walls", C.c_int32),
        ("sections", C.POINTER(Section)), ("walls", C.POINTER(WallSeg)),
        ("time_precision", _I8), ("section_window", _I8), ("mcts_iterations", C.c_int32),
        ("mcts_initial_iterations", C.c_int32), ("mcts_latency_ticks", C.c_int32), ("mcts_initial_latency_ticks", C.c_int32),
        ("mcts_seed", C.c_uint32), ("rewards", C.c_int32), ("training_agent", _I8), ("rw", RewardParams),
        ("train_seed", C.c_uint32), ("debug_taps", C.c_int32), ("engine", EngineParams),
    ]


HK_MCTS_MAX_DEPTH = 8
HK_MCTS_MAX_ACTIONS = 36
HK_MCTS_SECTIME_RING = 8
HK_MCTS_MAX_ROOT_PHASES = 3
_U8A = C.c_uint8 * HK_MAX_AGENTS


class MctsPlan(C.Structure):
    _fields_ = [("n_states", C.c_int32), ("n_players", C.c_int32), ("section", C.c_int32 * HK_MCTS_MAX_DEPTH),
                ("player_agent", _U8A), ("lane", _U8A * HK_MCTS_MAX_DEPTH), ("vel", _U8A * HK_MCTS_MAX_DEPTH)]


class MctsState(C.Structure):
    _fields_ = [("sec_time", C.c_int32 * HK_MCTS_SECTIME_RING), ("ready_step", C.c_int32), ("searches", C.c_int32),
                ("root_live", C.c_int32), ("root_cycles", C.c_int32), ("pend_kind", C.c_int32), ("root_phases", C.c_int32),
                ("best", MctsPlan), ("pend", MctsPlan),
                ("belief_lane", (C.c_uint8 * HK_MAX_SECTIONS) * HK_MAX_AGENTS),
                ("belief_vel", (C.c_uint8 * HK_MAX_SECTIONS) * HK_MAX_AGENTS)]


class AgentState(C.Structure):
    _fields_ = [
        ("px", C.c_float), ("pz", C.c_float), ("yaw", C.c_float), ("vx", C.c_float), ("vz", C.c_float), ("wy", C.c_float),
        ("acc_ang_v", C.c_float), ("steering", C.c_float), ("avg_lane_diff", C.c_float), ("avg_vel_diff", C.c_float),
        ("cum_reward", C.c_float), ("contact_nx", C.c_float), ("contact_nz", C.c_float),
        ("section_index", C.c_int32), ("lane", C.c_int32), ("lane_changes", C.c_int32), ("illegal_lane_changes", C.c_int32),
        ("forward_collisions", C.c_int32), ("last_collision_time", C.c_int32), ("time_steps", C.c_int32),
        ("init_checkpoint_index", C.c_int32),
        ("flags", C.c_uint32), ("trig_lo", C.c_uint32), ("trig_hi", C.c_uint32), ("final_steer", C.c_float),
        ("tele_completed_laps", C.c_int32), ("tele_lap_end_step", C.c_int32), ("tele_last_lap", C.c_float),
        ("tele_best_lap", C.c_float), ("tele_total_time", C.c_float),
        ("plan_lane", C.c_uint8 * HK_MAX_SECTIONS), ("plan_vel", C.c_float * HK_MAX_SECTIONS),
        ("step_reward", C.c_float), ("group_reward", C.c_float), ("steer_smoothed", C.c_float),
        ("wheel_uf", C.c_float), ("wheel_ur", C.c_float),
    ]


class EnvState(C.Structure):
    _fields_ = [("episode_steps", C.c_int32), ("inactive_mask", C.c_uint32), ("experiment_num", C.c_int32),
                ("episodes_done", C.c_int32), ("status", C.c_uint32), ("initial_started", C.c_int32), ("reserved", C.c_int32 * 2)]


class EpisodeResult(C.Structure):
    _fields_ = [("time_steps", C.c_int32), ("section_index", C.c_int32), ("illegal_lane_changes", C.c_int32),
                ("forward_collisions", C.c_int32), ("avg_lane_diff", C.c_float), ("avg_vel_diff", C.c_float),
                ("reward", C.c_float), ("episode", C.c_int32), ("last_lap", C.c_float), ("best_lap", C.c_float),
                ("total_time", C.c_float), ("laps_completed", C.c_int32), ("lap_end_step", C.c_int32), ("speed", C.c_float),
                ("active", C.c_int32), ("group_reward", C.c_float)]


class LqDebug(C.Structure):
    _fields_ = [("n_players", C.c_int32), ("player_agent", _I8), ("branch", _I8),
                ("initial", (C.c_double * 4) * HK_MAX_AGENTS), ("target", (C.c_double * 4) * HK_MAX_AGENTS),
                ("target_w", (C.c_double * 4) * HK_MAX_AGENTS), ("control_w", C.c_double * HK_MAX_AGENTS),
                ("u0", C.c_double * 2)]


# every symbol include/hk.h declares: (restype, argtypes)
_dp = C.POINTER(C.c_double)
_H = C.c_void_p
_fp = C.POINTER(C.c_float)


class PolicyDesc(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("stack", C.c_int32), ("hidden", C.c_int32), ("n_layers", C.c_int32),
                ("n_branch", C.c_int32), ("normalize", C.c_int32), ("deterministic", C.c_int32), ("seed", C.c_uint32),
                ("norm_mean", _fp), ("norm_std", _fp), ("W", _fp * HK_POLICY_MAX_LAYERS), ("b", _fp * HK_POLICY_MAX_LAYERS),
                ("W_mu", _fp), ("b_mu", _fp), ("log_sigma", _fp), ("W_branch", _fp), ("b_branch", _fp)]


SYMBOLS = {
    "hk_create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "hk_destroy": (None, [_H]),
    "hk_last_error": (C.c_char_p, [_H]),
    "hk_reset": (C.c_int, [_H, C.POINTER(C.c_int32), C.c_int, C.c_int]),
    "hk_set_actions": (C.c_int, [_H, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "hk_step": (C.c_int, [_H, C.c_int]),
    "hk_obs_dim": (C.c_int, [_H]),
    "hk_get_observations": (C.c_int, [_H, C.POINTER(C.c_float)]),
    "hk_get_agent_state": (C.c_int, [_H, C.POINTER(AgentState)]),
    "hk_set_agent_state": (C.c_int, [_H, C.POINTER(AgentState)]),
    "hk_get_env_state": (C.c_int, [_H, C.POINTER(EnvState)]),
    "hk_set_env_state": (C.c_int, [_H, C.POINTER(EnvState)]),
    "hk_get_episode_results": (C.c_int, [_H, C.POINTER(EpisodeResult)]),
    "hk_get_rewards": (C.c_int, [_H, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "hk_get_lq_debug": (C.c_int, [_H, C.c_int, C.c_int, C.POINTER(LqDebug)]),
    "hk_get_mcts_state": (C.c_int, [_H, C.POINTER(MctsState)]),
    "hk_lq_solve_batch": (C.c_int, [_H, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp]),
    "hk_lq_solve_batch_device": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hk_device_results_ptr": (C.c_void_p, [_H]),
    "hk_device_agents_ptr": (C.c_void_p, [_H]),
    "hk_observe": (C.c_int, [_H]),
    "hk_rewards_device": (C.c_int, [_H]),
    "hk_device_obs_ptr": (C.c_void_p, [_H]),
    "hk_device_reward_ptr": (C.c_void_p, [_H]),
    "hk_device_group_reward_ptr": (C.c_void_p, [_H]),
    "hk_device_act_steer_ptr": (C.c_void_p, [_H]),
    "hk_device_act_branch_ptr": (C.c_void_p, [_H]),
    "hk_stream": (C.c_void_p, [_H]),
    "hk_synchronize": (C.c_int, [_H]),
    "hk_build_info": (C.c_char_p, []),
    "hk_schedule_info": (C.c_char_p, [_H]),
    "hk_policy_attach": (C.c_int, [_H, C.POINTER(PolicyDesc), C.POINTER(C.c_int32), C.c_int, C.c_int]),
    "hk_policy_forward": (C.c_int, [_H, C.c_int, C.c_int, _fp, _fp, _fp]),
    "hk_get_actions": (C.c_int, [_H, _fp, C.POINTER(C.c_int32)]),
    "hk_comm_unique_id": (C.c_int, [C.c_void_p]),
    "hk_comm_init": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "hk_gather_results": (C.c_int, [_H, C.POINTER(EpisodeResult)]),
    "hk_comm_destroy": (C.c_int, [_H]),
    "hk_prof_enable": (C.c_int, [_H, C.c_int]),
    "hk_prof_reset": (C.c_int, [_H]),
    "hk_prof_read": (C.c_int, [_H, _dp, C.POINTER(C.c_int64)]),
    "hk_prof_games": (C.c_int, [_H, C.POINTER(C.c_int64)]),
    "hk_gather_count": (C.c_int, [_H, C.POINTER(C.c_int64)]),
}

_lib = None


class HkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libhk error %d: %s" % (code, msg))
        self.code = code


def load():
    """dlopen libhk.so and bind every declared symbol; raises if the HIP extension is missing (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libhk.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc, h=None):
    if rc != 0:
        msg = load().hk_last_error(h)
        raise HkError(rc, msg.decode() if msg else "")
    return rc
