// HkNative.cs — P/Invoke binding of libhk.so (include/hk.h, HK_ABI_VERSION 5) for the reference's Unity C# host.
//
// Drop it under Assets/Karting/Scripts/AI/Native/ with libhk.so in Assets/Plugins/x86_64.  Every struct mirrors its C
// twin field for field (tests/test_csharp_layout.py parses this file and checks field order, types, array lengths and
// sizes against the ctypes mirror of hk.h, so the two cannot drift apart unnoticed); every [DllImport] names an exported symbol.
// dotnet / mono / Unity are not part of the build image: this file is source, compiled only where the host lives.
using System;
using System.Runtime.InteropServices;

namespace KartGame.AI.Native
{
    public static class HkConst
    {
        public const int HK_ABI_VERSION = 5;
        public const int HK_MAX_AGENTS = 8;
        public const int HK_MAX_SECTIONS = 64;
        public const int HK_NUM_SENSORS = 9;
        public const int HK_MCTS_MAX_DEPTH = 8;
        public const int HK_MCTS_SECTIME_RING = 8;
        public const int HK_MCTS_MAX_ROOT_PHASES = 3;
        public const int HK_COMM_ID_BYTES = 128;
        public const int HK_PROF_STAGES = 6;
        // HierarchicalKartAgent.cs:21-33
        public const int HK_LOW_RL = 0, HK_LOW_MPC = 1, HK_LOW_LQR = 2;
        public const int HK_HIGH_MCTS = 0, HK_HIGH_FIXED = 1;
        // RacingEnvController.cs:24-29
        public const int HK_MODE_RACE = 0, HK_MODE_TRAINING = 1, HK_MODE_EXPERIMENT = 2;
        // hk_agent_state.flags
        public const uint HK_F_ACCEL = 1u << 0, HK_F_BRAKE = 1u << 1, HK_F_ACTIVE = 1u << 2, HK_F_FORWARD_COLLISION = 1u << 3,
                          HK_F_HAS_COLLISION = 1u << 4, HK_F_CAN_MOVE = 1u << 5, HK_F_ENABLED = 1u << 6;
    }

    // hk_kart_stats: ArcadeKart.Stats (KartSystems/ArcadeKart.cs:20-69)
    [StructLayout(LayoutKind.Sequential)]
    public struct HkKartStats
    {
        public float TopSpeed;
        public float Acceleration;
        public float ReverseSpeed;
        public float ReverseAcceleration;
        public float AccelerationCurve;
        public float Braking;
        public float CoastingDrag;
        public float Grip;
        public float MaxSteer;
        public float MinSteer;
        public float TireWearFactor;
        public float MinGs;
        public float MaxGs;
        public float AddedGravity;
        public float TireWearRate;
        public float AngularDrag;
    }

    // hk_section: one DiscretePositionTracker (DiscretePositionTracker.cs:20-44) with its Waypoint prefab geometry
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct HkSection
    {
        public float trig_x;
        public float trig_z;
        public float yaw_deg;
        public float marker_y;
        public fixed float lane_x[4];
        public fixed float lane_z[4];
        public float track_inside_radius;
        public float track_length;
        public float track_width;
        public float turn_degrees;
        public int left_turn;
        public int optimal_lane;
    }

    // hk_wall_seg
    [StructLayout(LayoutKind.Sequential)]
    public struct HkWallSeg
    {
        public float x0;
        public float z0;
        public float x1;
        public float z1;
    }

    // hk_reward_params: RacingEnvController reward fields (REC:65-108)
    [StructLayout(LayoutKind.Sequential)]
    public struct HkRewardParams
    {
        public float WallHitPenalty;
        public float OpponentHitPenalty;
        public float HitByOpponentPenalty;
        public float PassCheckpointLaneReward;
        public float PassCheckpointVelocityReward;
        public float PassCheckpointBase;
        public float PassCheckpointTimeMultiplier;
        public float TeamPassCheckpointBase;
        public float TeamPassCheckpointTimeMultiplier;
        public float BeingBehindOpponentCheckpointPenalty;
        public float BeingBehindTeammateCheckpointPenalty;
        public float TeamScoreRewardMultiplier;
        public float ReversePenalty;
        public float SwervingPenalty;
        public float ReachGoalCheckpointRewardMultplier;
        public float ReachGoalCheckpointRewardBase;
        public float TowardsCheckpointReward;
        public float SpeedReward;
        public float SlowMovingPenalty;
        public float AccelerationReward;
        public float NotAtGoalPenalty;
    }

    // hk_engine_params: the engine restatement's constants (Rigidbody / CapsuleCollider / WheelCollider / KartAnimation values of the kart prefab)
    [StructLayout(LayoutKind.Sequential)]
    public struct HkEngineParams
    {
        public float mass;
        public float inertia_y;
        public float gravity;
        public float axle_zf;
        public float axle_zr;
        public float max_steer_deg;
        public float steer_damping;
        public float side_ext_slip;
        public float side_ext_value;
        public float side_asy_slip;
        public float side_asy_value;
        public float side_stiffness;
        public float side_slope0;
        public float slip_min_speed;
        public float wheel_mass;
        public float wheel_radius_f;
        public float wheel_radius_r;
        public float wheel_damping;
        public float fwd_ext_slip;
        public float fwd_ext_value;
        public float fwd_asy_slip;
        public float fwd_asy_value;
        public float fwd_stiffness;
        public float long_slip_min_speed;
        public int wheel_friction;
        public int contact_yaw;
        public int wheel_rolling;
    }

    // hk_config
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct HkConfig
    {
        public int abi_version;
        public int num_envs;
        public int num_agents;
        public int device_id;
        public fixed int team_of[8];
        public fixed int n_team[8];
        public fixed int team_agents[64];
        public fixed int n_other[8];
        public fixed int other_agents[64];
        public fixed int high_mode[8];
        public fixed int low_mode[8];
        public fixed int tree_search_depth[8];
        public fixed int velocity_bucket_size[8];
        public HkKartStats stats;
        public int laps;
        public int max_episode_steps;
        public int max_lane_changes;
        public int section_horizon;
        public int disable_on_end;
        public int env_mode;
        public int start_hold_ticks;
        public int auto_reset;
        public float dt;
        public float kart_y;
        public fixed float sensor_yaw_deg[9];
        public fixed float ray_distance[9];
        public fixed float wall_hit_validation[9];
        public fixed float agent_hit_validation[9];
        public uint jitter_seed;
        public float jitter_pos;
        public float jitter_yaw;
        public int env_id_base;
        public int num_sections;
        public int num_walls;
        public HkSection* sections;
        public HkWallSeg* walls;
        public fixed int time_precision[8];
        public fixed int section_window[8];
        public int mcts_iterations;
        public int mcts_initial_iterations;
        public int mcts_latency_ticks;
        public int mcts_initial_latency_ticks;
        public uint mcts_seed;
        public int rewards;
        public fixed int training_agent[8];
        public HkRewardParams rw;
        public uint train_seed;
        public int debug_taps;
        public HkEngineParams engine;
    }

    // hk_mcts_plan: bestStates (HKA:70,252)
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct HkMctsPlan
    {
        public int n_states;
        public int n_players;
        public fixed int section[8];
        public fixed byte player_agent[8];
        public fixed byte lane[64];
        public fixed byte vel[64];
    }

    // hk_mcts_state
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct HkMctsState
    {
        public fixed int sec_time[8];
        public int ready_step;
        public int searches;
        public int root_live;
        public int root_cycles;
        public int pend_kind;
        public int root_phases;
        public HkMctsPlan best;
        public HkMctsPlan pend;
        public fixed byte belief_lane[512];
        public fixed byte belief_vel[512];
    }

    // hk_agent_state: the fields of KartAgent (KA:102-128), ArcadeKart (AK:190-205) and the Rigidbody that survive a tick
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct HkAgentState
    {
        public float px;
        public float pz;
        public float yaw;
        public float vx;
        public float vz;
        public float wy;
        public float acc_ang_v;
        public float steering;
        public float avg_lane_diff;
        public float avg_vel_diff;
        public float cum_reward;
        public float contact_nx;
        public float contact_nz;
        public int section_index;
        public int lane;
        public int lane_changes;
        public int illegal_lane_changes;
        public int forward_collisions;
        public int last_collision_time;
        public int time_steps;
        public int init_checkpoint_index;
        public uint flags;
        public uint trig_lo;
        public uint trig_hi;
        public float final_steer;
        public int tele_completed_laps;
        public int tele_lap_end_step;
        public float tele_last_lap;
        public float tele_best_lap;
        public float tele_total_time;
        public fixed byte plan_lane[64];
        public fixed float plan_vel[64];
        public float step_reward;
        public float group_reward;
        public float steer_smoothed;
        public float wheel_uf;
        public float wheel_ur;
    }

    // hk_env_state
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct HkEnvState
    {
        public int episode_steps;
        public uint inactive_mask;
        public int experiment_num;
        public int episodes_done;
        public uint status;
        public int initial_started;
        public fixed int reserved[2];
    }

    // hk_episode_result: TelemetryViewer / experiment-log quantities (TelemetryViewer.cs:49-108)
    [StructLayout(LayoutKind.Sequential)]
    public struct HkEpisodeResult
    {
        public int time_steps;
        public int section_index;
        public int illegal_lane_changes;
        public int forward_collisions;
        public float avg_lane_diff;
        public float avg_vel_diff;
        public float reward;
        public int episode;
        public float last_lap;
        public float best_lap;
        public float total_time;
        public int laps_completed;
        public int lap_end_step;
        public float speed;
        public int active;
        public float group_reward;
    }

    // hk_lq_debug
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct HkLqDebug
    {
        public int n_players;
        public fixed int player_agent[8];
        public fixed int branch[8];
        public fixed double initial[32];
        public fixed double target[32];
        public fixed double target_w[32];
        public fixed double control_w[8];
        public fixed double u0[2];
    }

    // hk_policy_desc: the ML-Agents actor read from a BehaviorParameters.Model asset (NNModel -> ONNX initialisers)
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct HkPolicyDesc
    {
        public int in_dim;
        public int stack;
        public int hidden;
        public int n_layers;
        public int n_branch;
        public int normalize;
        public int deterministic;
        public uint seed;
        public float* norm_mean;
        public float* norm_std;
        public float* W0;
        public float* W1;
        public float* W2;
        public float* W3;
        public float* b0;
        public float* b1;
        public float* b2;
        public float* b3;
        public float* W_mu;
        public float* b_mu;
        public float* log_sigma;
        public float* W_branch;
        public float* b_branch;
    }

    public static unsafe class Hk
    {
        const string Lib = "hk";   // libhk.so next to the player binary / in Assets/Plugins/x86_64

        // scene instantiation of RacingEnvController + KartAgents (REC.Start :148-168, HKA.Awake :413-426)
        [DllImport(Lib)] public static extern int hk_create(HkConfig* cfg, out IntPtr handle);
        [DllImport(Lib)] public static extern void hk_destroy(IntPtr h);
        [DllImport(Lib)] public static extern IntPtr hk_last_error(IntPtr h);
        // RacingEnvController.ResetGame (REC:499-719)
        [DllImport(Lib)] public static extern int hk_reset(IntPtr h, int* envIds, int n, int experimentNum);
        // KartAgent.OnActionReceived / InterpretDiscreteActions (KA:440-478, HKA:1371-1379)
        [DllImport(Lib)] public static extern int hk_set_actions(IntPtr h, float* steer, int* branch);
        // n Unity FixedUpdate ticks
        [DllImport(Lib)] public static extern int hk_step(IntPtr h, int nTicks);
        // HierarchicalKartAgent.CollectObservations (HKA:485-604)
        [DllImport(Lib)] public static extern int hk_obs_dim(IntPtr h);
        [DllImport(Lib)] public static extern int hk_get_observations(IntPtr h, float* obs);
        [DllImport(Lib)] public static extern int hk_get_agent_state(IntPtr h, HkAgentState* outState);
        [DllImport(Lib)] public static extern int hk_set_agent_state(IntPtr h, HkAgentState* inState);
        [DllImport(Lib)] public static extern int hk_get_env_state(IntPtr h, HkEnvState* outState);
        [DllImport(Lib)] public static extern int hk_set_env_state(IntPtr h, HkEnvState* inState);
        [DllImport(Lib)] public static extern int hk_get_episode_results(IntPtr h, HkEpisodeResult* outRes);
        // Agent.SendInfo: m_Reward / m_GroupReward since the last call (read and reset)
        [DllImport(Lib)] public static extern int hk_get_rewards(IntPtr h, float* reward, float* groupReward);
        [DllImport(Lib)] public static extern int hk_get_lq_debug(IntPtr h, int env, int ego, HkLqDebug* outDbg);
        [DllImport(Lib)] public static extern int hk_get_mcts_state(IntPtr h, HkMctsState* outStates);
        // KartLQR.solveFeedbackLQR (AI/LQR/KartLQR.cs:17), batched
        [DllImport(Lib)] public static extern int hk_lq_solve_batch(IntPtr h, int batch, int N, double* A, double* B, double* Q, double* q,
                                                                  double* R, double* x0, int horizon, double* u0);
        [DllImport(Lib)] public static extern int hk_lq_solve_batch_device(IntPtr h, int batch, int N, IntPtr dA, IntPtr dB, IntPtr dQ, IntPtr dq,
                                                                         IntPtr dR, IntPtr dx0, int horizon, IntPtr du0, IntPtr stream);
        // device-resident surface (an external trainer that keeps its tensors on the GPU)
        [DllImport(Lib)] public static extern IntPtr hk_device_results_ptr(IntPtr h);
        [DllImport(Lib)] public static extern IntPtr hk_device_agents_ptr(IntPtr h);
        [DllImport(Lib)] public static extern int hk_observe(IntPtr h);
        [DllImport(Lib)] public static extern int hk_rewards_device(IntPtr h);
        [DllImport(Lib)] public static extern IntPtr hk_device_obs_ptr(IntPtr h);
        [DllImport(Lib)] public static extern IntPtr hk_device_reward_ptr(IntPtr h);
        [DllImport(Lib)] public static extern IntPtr hk_device_group_reward_ptr(IntPtr h);
        [DllImport(Lib)] public static extern IntPtr hk_device_act_steer_ptr(IntPtr h);
        [DllImport(Lib)] public static extern IntPtr hk_device_act_branch_ptr(IntPtr h);
        [DllImport(Lib)] public static extern IntPtr hk_stream(IntPtr h);
        [DllImport(Lib)] public static extern int hk_synchronize(IntPtr h);
        [DllImport(Lib)] public static extern IntPtr hk_build_info();               // const char* (JSON), owned by the library: Marshal.PtrToStringAnsi
        [DllImport(Lib)] public static extern IntPtr hk_schedule_info(IntPtr h);     // const char* (JSON), owned by the handle
        // LowMode == RL: the actor runs on device every DecisionPeriod ticks instead of Barracuda (KA:440, HKA:1371-1379)
        [DllImport(Lib)] public static extern int hk_policy_attach(IntPtr h, HkPolicyDesc* desc, int* agentSlots, int nSlots, int decisionPeriod);
        [DllImport(Lib)] public static extern int hk_policy_forward(IntPtr h, int policy, int rows, float* obs, float* mu, float* logits);
        [DllImport(Lib)] public static extern int hk_get_actions(IntPtr h, float* steer, int* branch);
        // multi-GPU: one process per GPU, envs sharded by HkConfig.env_id_base; the only exchange is this all-gather over RCCL
        [DllImport(Lib)] public static extern int hk_comm_unique_id(byte* id128);
        [DllImport(Lib)] public static extern int hk_comm_init(IntPtr h, int worldSize, int rank, byte* id128);
        [DllImport(Lib)] public static extern int hk_gather_count(IntPtr h, long* totalEnvs);
        [DllImport(Lib)] public static extern int hk_gather_results(IntPtr h, HkEpisodeResult* all);
        [DllImport(Lib)] public static extern int hk_comm_destroy(IntPtr h);
        // timing taps
        [DllImport(Lib)] public static extern int hk_prof_enable(IntPtr h, int on);
        [DllImport(Lib)] public static extern int hk_prof_reset(IntPtr h);
        [DllImport(Lib)] public static extern int hk_prof_read(IntPtr h, double* ms, long* launches);
        [DllImport(Lib)] public static extern int hk_prof_games(IntPtr h, long* games);

        public static void Check(int rc, IntPtr h)
        {
            if (rc < 0) throw new InvalidOperationException("libhk error " + rc + ": " + Marshal.PtrToStringAnsi(hk_last_error(h)));
        }
    }
}
