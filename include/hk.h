/*
 * hk.h — C ABI of libhk.so: the MI355X-native batched kart-racing step + feedback LQ Nash-game solve.
 *
 * Drop-in boundary for ONE hot path of ribsthakkar/HierarchicalKarting (Unity C#).  The reference has no FFI for
 * this path (everything is in-process managed calls), so each entry point below names the managed surface it
 * replaces (reference file:line, paths relative to Assets/Karting/Scripts/).  A C# host binds these with
 * [DllImport("hk")]; all types are blittable (int32 / uint32 / float / double / pointers).
 *
 * Conventions: return 0 on success, negative hk_status on error (never throws across the ABI);
 * host buffers are caller-owned, device buffers library-owned; one handle = one GPU context, NOT thread-safe
 * (Unity drives it from FixedUpdate on the main thread, like the reference); row-major arrays.
 * There is NO CPU fallback: without a HIP device hk_create / hk_lq_solve_batch fail with HK_ERR_NO_DEVICE.
 */
#ifndef HK_H
#define HK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HK_ABI_VERSION 5
#define HK_MAX_AGENTS 8      /* the largest reference scene has 4; 5..8 agents per env is the synthetic extension of BASELINE configs[4] (start grid continued row by row) */
#define HK_MAX_SECTIONS 64   /* Oval 24, Complex 41 */
#define HK_NUM_SENSORS 9     /* MLAgent_Sensors.prefab */

typedef enum hk_status {
    HK_OK = 0,
    HK_ERR_INVALID = -1,      /* bad argument / config */
    HK_ERR_NO_DEVICE = -2,    /* no HIP device: the product path has no CPU fallback */
    HK_ERR_HIP = -3,          /* HIP runtime error (see hk_last_error) */
    HK_ERR_UNSUPPORTED = -4,  /* valid in the reference, not built yet (num_agents > 4, LQ games of > 8 players) */
    HK_ERR_SINGULAR = -5      /* LQ: zero pivot in the m x m solve */
} hk_status;

/* HierarchicalKartAgent.cs:21-33 */
enum { HK_LOW_RL = 0, HK_LOW_MPC = 1, HK_LOW_LQR = 2 };
enum { HK_HIGH_MCTS = 0, HK_HIGH_FIXED = 1 };
/* RacingEnvController.cs:24-29 */
enum { HK_MODE_RACE = 0, HK_MODE_TRAINING = 1, HK_MODE_EXPERIMENT = 2 };

/* agent flag bits (hk_agent_state.flags) */
enum {
    HK_F_ACCEL = 1u << 0,             /* KartAgent.m_Acceleration      KA:104 */
    HK_F_BRAKE = 1u << 1,             /* KartAgent.m_Brake             KA:105 */
    HK_F_ACTIVE = 1u << 2,            /* KartAgent.is_active           KA:123 */
    HK_F_FORWARD_COLLISION = 1u << 3, /* KartAgent.forwardCollision    KA:125 */
    HK_F_HAS_COLLISION = 1u << 4,     /* ArcadeKart.m_HasCollision     AK:202 */
    HK_F_CAN_MOVE = 1u << 5,          /* ArcadeKart.m_CanMove          AK:195 */
    HK_F_ENABLED = 1u << 6            /* GameObject.activeSelf (Deactivate(disable) KA:405-416) */
};

/* ArcadeKart.Stats (KartSystems/ArcadeKart.cs:20-69); values used: SURVEY App. A */
typedef struct hk_kart_stats {
    float TopSpeed, Acceleration, ReverseSpeed, ReverseAcceleration, AccelerationCurve, Braking, CoastingDrag, Grip;
    float MaxSteer, MinSteer, TireWearFactor, MinGs, MaxGs, AddedGravity;
    float TireWearRate;      /* AK:191 */
    float AngularDrag;       /* Rigidbody.angularDrag 0.05 (BaseKartClassic.prefab:173) */
} hk_kart_stats;

/* one DiscretePositionTracker (DiscretePositionTracker.cs:20-44) with its Waypoint prefab geometry */
typedef struct hk_section {
    float trig_x, trig_z;        /* Trigger box centre (box 10 x 1 x 1, local z +0.407 from the waypoint) */
    float yaw_deg;               /* Unity Y rotation of the waypoint (0 = +z, 90 = +x) */
    float marker_y;              /* world y of Trigger / lane markers (0.75): the constant dy of quirk Q13 */
    float lane_x[4], lane_z[4];  /* Lane1..Lane4 markers */
    float track_inside_radius, track_length, track_width, turn_degrees;
    int32_t left_turn, optimal_lane;
} hk_section;

typedef struct hk_wall_seg { float x0, z0, x1, z1; } hk_wall_seg; /* road-side wall face, world metres */

/* RacingEnvController reward fields (REC:65-108, defaults as in the source) */
typedef struct hk_reward_params {
    float WallHitPenalty, OpponentHitPenalty, HitByOpponentPenalty, PassCheckpointLaneReward, PassCheckpointVelocityReward;
    float PassCheckpointBase, PassCheckpointTimeMultiplier, TeamPassCheckpointBase, TeamPassCheckpointTimeMultiplier;
    float BeingBehindOpponentCheckpointPenalty, BeingBehindTeammateCheckpointPenalty, TeamScoreRewardMultiplier;
    float ReversePenalty, SwervingPenalty, ReachGoalCheckpointRewardMultplier, ReachGoalCheckpointRewardBase;
    float TowardsCheckpointReward, SpeedReward, SlowMovingPenalty, AccelerationReward, NotAtGoalPenalty;
} hk_reward_params;

/* Engine restatement: what Unity / PhysX does between two FixedUpdates and the C# never spells out.  Every
 * constant is read from the reference's prefabs / assets; `side_slope0` is the one fitted number.
 *   rigid body   Rigidbody mass 250, free rotation (m_Constraints 0; BaseKartClassic.prefab:164-178); centre of mass = the
 *                WheelColliders transform, kart-local (0, 0.12, 0) (AK:254, BaseKartClassic.prefab:203); inertia about y = that of the
 *                capsule collider (r 0.45, h 2, axis z) at mass 250 = 74.7 kg m^2 (PhysX computes it from the shapes)
 *   contacts     kart capsule: PhysicsMaterials/NoFriction (friction 0, bounciness 0); track MeshColliders: CarWheels (friction 0,
 *                combine Minimum) -> frictionless, inelastic contacts; an off-centre contact impulse turns the body
 *   wheels       four WheelColliders (BaseKartClassic.prefab:35-63,...): axles at kart-local z +0.586 / -0.681, sidewaysFriction
 *                {extremum (0.2, 1), asymptote (0.5, 0.75), stiffness 1}; the front pair is steered by KartAnimation.FixedUpdate
 *                (KartAnimation.cs:54-63): steerAngle = MoveTowards(smoothed, TurnInput, 10 dt) * maxSteeringAngle 30 */
typedef struct hk_engine_params {
    float mass, inertia_y, gravity;
    float axle_zf, axle_zr;          /* front / rear axle, metres ahead of the centre of mass */
    float max_steer_deg;             /* KartAnimation.maxSteeringAngle */
    float steer_damping;             /* KartAnimation.steeringAnimationDamping (1 / s) */
    float side_ext_slip, side_ext_value, side_asy_slip, side_asy_value, side_stiffness;   /* WheelCollider.sidewaysFriction */
    float side_slope0;               /* FITTED: slope of the friction curve at zero slip, in units of ext_value / ext_slip (0 .. 3) */
    float slip_min_speed;            /* m/s added to |longitudinal speed| in the slip ratio (PhysX vehicle tire model: 1) */
    /* rolling: no script drives or brakes the wheels (motorTorque = brakeTorque = 0), so their spin follows the ground through the
     * forwardFriction curve and is damped by wheelDampingRate; the force that keeps them turning is a drag on the body, and every change
     * of the kart's speed has to spin 4 x 20 kg of wheel up or down */
    float wheel_mass;                /* WheelCollider.mass 20 (moment of inertia m r^2 / 2) */
    float wheel_radius_f, wheel_radius_r;   /* 0.1372984 / 0.1630791 */
    float wheel_damping;             /* WheelCollider.wheelDampingRate 0.56 (kg m^2 / s) */
    float fwd_ext_slip, fwd_ext_value, fwd_asy_slip, fwd_asy_value, fwd_stiffness;        /* WheelCollider.forwardFriction */
    float long_slip_min_speed;       /* m/s added to |longitudinal speed| in the longitudinal slip ratio (PhysX minLongSlipDenominator: 4) */
    int32_t wheel_friction;          /* 0: no tire forces (the model of ABI <= 4) */
    int32_t contact_yaw;             /* 0: contacts change the linear velocity only (the model of ABI <= 4) */
    int32_t wheel_rolling;           /* 0: no longitudinal tire forces */
} hk_engine_params;

typedef struct hk_config {
    int32_t abi_version;         /* HK_ABI_VERSION */
    int32_t num_envs;            /* E: independent RacingEnvControllers on this device */
    int32_t num_agents;          /* A: RacingEnvController.Agents.Length (REC:49) */
    int32_t device_id;
    /* wiring: KartAgent.teamAgents / otherAgents (KA:63-65), RacingEnvController.Teams (REC:46) */
    int32_t team_of[HK_MAX_AGENTS];
    int32_t n_team[HK_MAX_AGENTS];
    int32_t team_agents[HK_MAX_AGENTS][HK_MAX_AGENTS];
    int32_t n_other[HK_MAX_AGENTS];
    int32_t other_agents[HK_MAX_AGENTS][HK_MAX_AGENTS];
    int32_t high_mode[HK_MAX_AGENTS];          /* HK_HIGH_*  (HKA:64) */
    int32_t low_mode[HK_MAX_AGENTS];           /* HK_LOW_*   (HKA:62) */
    int32_t tree_search_depth[HK_MAX_AGENTS];  /* gameParams.treeSearchDepth (HKA:48) */
    int32_t velocity_bucket_size[HK_MAX_AGENTS];
    hk_kart_stats stats;                       /* identical on every kart of the Compete scenes */
    /* rules (REC:110-130) */
    int32_t laps, max_episode_steps, max_lane_changes, section_horizon, disable_on_end, env_mode;
    int32_t start_hold_ticks;    /* WaitForSeconds(1.5f) at dt 0.02 = 75 ticks (REC:721-724) */
    int32_t auto_reset;          /* REC.FixedUpdate resets when every agent is inactive (REC:241-270) */
    float dt;                    /* Time.fixedDeltaTime 0.02 */
    float kart_y;                /* spawn height 0.28 (REC:715) */
    /* sensors (MLAgent_Sensors.prefab, KA:20-26): yaw in degrees, positive = to the right */
    float sensor_yaw_deg[HK_NUM_SENSORS];
    float ray_distance[HK_NUM_SENSORS];
    float wall_hit_validation[HK_NUM_SENSORS];
    float agent_hit_validation[HK_NUM_SENSORS];
    /* synthetic de-synchronisation of the start grid (NOT in the reference): Philox-4x32 keyed by
     * jitter_seed + env_id;  dx,dz ~ U(-jitter_pos, jitter_pos), dyaw ~ U(-jitter_yaw, jitter_yaw) rad. 0 = off. */
    uint32_t jitter_seed;
    float jitter_pos, jitter_yaw;
    int32_t env_id_base;         /* global id of local env 0 (multi-GPU sharding: contiguous ranges per rank) */
    /* track */
    int32_t num_sections;
    int32_t num_walls;
    const hk_section* sections;
    const hk_wall_seg* walls;
    /* MCTS high-level planner (HighMode == HK_HIGH_MCTS; KartMCTS.cs, KartDiscreteGame.cs, HKA:172-284,330-402).
     * gameParams (HKA:38-52), scene values {timePrecision 100, sectionWindow 2, treeSearchDepth 8, velocityBucketSize 2}.
     * The reference searches on a background thread under a WALL-CLOCK budget (T = 1.5 s at reset, 0.9 s every 100
     * ticks) with System.Random / MathNet draws; here the budget is an iteration count, the plan becomes visible a fixed
     * number of ticks after it was requested, and the draws are Philox-4x32 keyed by mcts_seed ("parity unpinned").  Root reuse
     * (HKA:265-283) and the sectionTimes back-fill of ResetGame (REC:679-702) are restated; see hk_mcts_state. */
    int32_t time_precision[HK_MAX_AGENTS];
    int32_t section_window[HK_MAX_AGENTS];
    int32_t mcts_iterations;          /* search iterations of a 100-tick replan (stands for T = 0.9 s) */
    int32_t mcts_initial_iterations;  /* of the plan made at reset (T = 1.5 s) */
    int32_t mcts_latency_ticks;       /* ticks until a replan's result is used (0.9 s = 45 ticks) */
    int32_t mcts_initial_latency_ticks;/* same for the plan made at reset (1.5 s = 75 ticks = the start hold) */
    uint32_t mcts_seed;
    /* reward shaping: 0 = off (hk_agent_state reward fields stay 0) */
    int32_t rewards;
    int32_t training_agent[HK_MAX_AGENTS];   /* KartAgent.Mode == AgentMode.Training (KA:34-38): only these enter AddGoalTimingRewards (REC:217) */
    hk_reward_params rw;
    /* env_mode == HK_MODE_TRAINING: ResetGame scatters the karts at random (REC:520-668); training_agent[i]: the agent plans
     * with planRandomly (HKA:109-143) instead of planFixed / MCTS.  UnityEngine.Random / System.Random draws become
     * Philox-4x32 keyed by train_seed, the global env id and the episode ("parity unpinned"). */
    uint32_t train_seed;
    int32_t debug_taps;          /* bit 0: record the LQ debug taps read by hk_get_lq_debug (costs a few %).  The environment variable
                                  * HK_LQ_DEBUG=<bits>, read once by hk_create, ORs into this field (tests switch the taps on without
                                  * rebuilding their configs) */
    hk_engine_params engine;     /* ABI 5 */
} hk_config;

#define HK_MCTS_MAX_DEPTH 8      /* gameParams.treeSearchDepth <= 8 */
#define HK_MCTS_MAX_ACTIONS 36   /* velocity buckets x 4 lanes (KartDiscreteGame.cs:329-338: for (i = 6; i < 15; i += velocityBucketSize)): 5 x 4 at
                                  * velocityBucketSize 2 (the MCTS-LQR agents of the Compete scenes), 9 x 4 at 1 (their MCTS-RL / Fixed-RL agents) */
#define HK_MCTS_SECTIME_RING 8   /* sectionTimes entries kept per kart (sectionWindow <= 4) */
#define HK_MCTS_MAX_ROOT_PHASES 3 /* searches one tree can receive (the reference: CyclesRootProcessed < 3); see hk_mcts_state.root_phases */

/* Planner state of one agent (kept beside hk_agent_state; HighMode MCTS agents only).  `best` is what the reference
 * holds in bestStates (HKA:70,252): for each future section the game reached by every player, each player's lane and
 * max_velocity; `pend` is a finished search that becomes `best` at episode step ready_step. */
typedef struct hk_mcts_plan {
    int32_t n_states;                                  /* 0..HK_MCTS_MAX_DEPTH */
    int32_t n_players;
    int32_t section[HK_MCTS_MAX_DEPTH];                /* lastCompletedSection of the state */
    uint8_t player_agent[HK_MAX_AGENTS];               /* Agents[] index of each player */
    uint8_t lane[HK_MCTS_MAX_DEPTH][HK_MAX_AGENTS];
    uint8_t vel[HK_MCTS_MAX_DEPTH][HK_MAX_AGENTS];     /* max_velocity */
} hk_mcts_plan;

typedef struct hk_mcts_state {
    int32_t sec_time[HK_MCTS_SECTIME_RING];            /* KartAgent.sectionTimes (KA:124), ring over section & 7: the planner reads the
                                                        * entries of karts at most 2 * sectionWindow - 2 <= 6 sections apart (HKA:221-224) */
    int32_t ready_step;                                /* episode step at which `pend` replaces `best`; -1: nothing pending */
    int32_t searches;                                  /* searches started since hk_create (debug / RNG stream id) */
    /* root reuse (HKA:66-67,175,265-283,660-669): a replan re-searches the tree of the previous plan (up to three searches per
     * tree) unless the kart entered a section, or drove backwards through one, after that plan was finished */
    int32_t root_live;                                 /* currentRoot != null */
    int32_t root_cycles;                               /* CyclesRootProcessed */
    int32_t pend_kind;                                 /* the search in flight: 0 none, 1 a new tree (HKA:175), 2 the existing root again (HKA:265) */
    int32_t root_phases;                               /* searches the current tree has received (set at the request; <= HK_MCTS_MAX_ROOT_PHASES) */
    hk_mcts_plan best, pend;
    /* opponentUpcomingLanes / opponentUpcomingVelocities (HKA:77-78): this agent's belief about every other agent's
     * plan, keyed by section % L; 0 = no entry */
    uint8_t belief_lane[HK_MAX_AGENTS][HK_MAX_SECTIONS];
    uint8_t belief_vel[HK_MAX_AGENTS][HK_MAX_SECTIONS];
} hk_mcts_state;


/* Per-agent persistent state: the fields of KartAgent (KA:102-128), ArcadeKart (AK:190-205) and the Rigidbody that
 * survive a tick.  hk_get/set_agent_state copy every field, which gives snapshot/restore for free.  (On the device the fields that
 * change every tick live in a wave-tiled structure-of-arrays layout; this record is the host-facing view.) */
typedef struct hk_agent_state {
    float px, pz;                 /* transform.position (x, z); y is kart_y */
    float yaw;                    /* Unity Y rotation, radians, kept in [0, 2pi) */
    float vx, vz;                 /* Rigidbody.velocity */
    float wy;                     /* Rigidbody.angularVelocity.y (rad/s) */
    float acc_ang_v;              /* ArcadeKart.m_AccumulatedAngularV (AK:190) */
    float steering;               /* KartAgent.m_Steering (KA:106) */
    float avg_lane_diff;          /* KA:116 */
    float avg_vel_diff;           /* KA:115 */
    float cum_reward;             /* Agent.GetCumulativeReward(): sum of AddReward since the episode began (0 unless cfg.rewards) */
    float contact_nx, contact_nz; /* m_LastCollisionNormal (AK:201), x/z */
    int32_t section_index;        /* m_SectionIndex KA:107 */
    int32_t lane;                 /* m_Lane KA:108 */
    int32_t lane_changes;         /* KA:109 */
    int32_t illegal_lane_changes; /* KA:110 */
    int32_t forward_collisions;   /* KA:126 */
    int32_t last_collision_time;  /* KA:127 */
    int32_t time_steps;           /* m_timeSteps KA:112 */
    int32_t init_checkpoint_index;/* KA:49 */
    uint32_t flags;               /* HK_F_* */
    uint32_t trig_lo, trig_hi;    /* which Trigger boxes the kart overlapped after the last tick (OnTriggerEnter edge) */
    float final_steer;            /* ArcadeKart.m_FinalStats.Steer as of the last UpdateStats (AK:300) */
    /* TelemetryViewer per-agent arrays (TelemetryViewer.cs:12-16), advanced once per tick after the engine step */
    int32_t tele_completed_laps;  /* lastCompletedLaps */
    int32_t tele_lap_end_step;    /* lastEpisodeSteps */
    float tele_last_lap;          /* lastLapTimes (s) */
    float tele_best_lap;          /* bestLapTimes (s) */
    float tele_total_time;        /* lastOverallTimes (s) */
    uint8_t plan_lane[HK_MAX_SECTIONS]; /* m_UpcomingLanes keyed by section % L (KA:117); 0 = no entry */
    float plan_vel[HK_MAX_SECTIONS];    /* m_UpcomingVelocities (KA:118) */
    /* ML-Agents per-decision accumulators: what Agent.AddReward / AddGroupReward collected since hk_get_rewards last
     * read them (Agent.SendInfo resets m_Reward / m_GroupReward the same way) */
    float step_reward;
    float group_reward;
    float steer_smoothed;         /* KartAnimation.m_SmoothedSteeringInput (KartAnimation.cs:43,56): the front WheelColliders' steerAngle / 30 deg (ABI 5) */
    float wheel_uf, wheel_ur;     /* rim speed (angular velocity x radius, m/s) of the front / rear WheelCollider pair (ABI 5) */
} hk_agent_state;

typedef struct hk_env_state {
    int32_t episode_steps;     /* REC:125 */
    uint32_t inactive_mask;    /* REC.inactiveAgents (REC:118) as a bit set over Agents[] */
    int32_t experiment_num;    /* REC:63 */
    int32_t episodes_done;     /* finished episodes since hk_create */
    uint32_t status;           /* bit0 NaN/Inf seen in a kart state, bit1 timeout ended last episode */
    int32_t initial_started;   /* REC.initialStarted (REC:132): the very first all-inactive tick resets without logging */
    int32_t reserved[2];       /* library-internal progress words of hk_step: [0] ticks left, [1] bits 0..3 phase inside a tick (both 0 between calls), bit 4 a scheduling hint */
} hk_env_state;

/* TelemetryViewer / experiment-log quantities of the last finished episode (TelemetryViewer.cs:49-108) */
typedef struct hk_episode_result {
    int32_t time_steps;            /* m_timeSteps at finish; 0 = did not finish (REC:470,476) */
    int32_t section_index;         /* laps = section_index / L */
    int32_t illegal_lane_changes;
    int32_t forward_collisions;
    float avg_lane_diff;
    float avg_vel_diff;
    float reward;                  /* Agent.GetCumulativeReward() at the end of the episode */
    int32_t episode;               /* index of the episode these numbers belong to (-1: none finished yet) */
    float last_lap, best_lap, total_time;   /* TelemetryViewer.cs:58-79 (seconds) */
    int32_t laps_completed;        /* lastCompletedLaps */
    int32_t lap_end_step;          /* lastEpisodeSteps (enters the reference's winner rule, TelemetryViewer.cs:80) */
    float speed;                   /* Rigidbody.velocity.magnitude when the block was written */
    int32_t active;                /* is_active when the block was written */
    float group_reward;            /* m_GroupReward when the episode ended, AddGoalTimingRewards (REC:174-237) included: the record is
                                      rewritten by ResetGame in the same tick, so this is where a trainer finds the terminal group reward */
} hk_episode_result;

/* debug tap: the LQ game one ego assembled on its last solve tick (HKA:699-1201), ego-local player order */
typedef struct hk_lq_debug {
    int32_t n_players;
    int32_t player_agent[HK_MAX_AGENTS]; /* agent index of each player, order [this, team..., other...] filtered (HKA:702-725) */
    int32_t branch[HK_MAX_AGENTS];       /* heading-heuristic branch id B1..B7 per player */
    double initial[HK_MAX_AGENTS][4];
    double target[HK_MAX_AGENTS][4];
    double target_w[HK_MAX_AGENTS][4];
    double control_w[HK_MAX_AGENTS];
    double u0[2];
} hk_lq_debug;

typedef struct hk_context* hk_handle;

/* replaces: scene instantiation of RacingEnvController + KartAgents (REC.Start :148-168, HKA.Awake :413-426) */
int hk_create(const hk_config* cfg, hk_handle* out);
void hk_destroy(hk_handle h);
const char* hk_last_error(hk_handle h);

/* RacingEnvController.ResetGame (REC:499-719) for the listed envs (env_ids == NULL: all), Experiment/Race grid:
 * ordering = allOrderings[experiment_num % A!] (REC:528-530).  experiment_num < 0: per-env (env_id_base + env) % A! */
int hk_reset(hk_handle h, const int32_t* env_ids, int n, int experiment_num);

/* KartAgent.OnActionReceived / InterpretDiscreteActions (KA:440-478, HKA:1371-1379): only read by LowMode == RL agents.
 * steer[E][A] continuous action 0, branch[E][A] discrete action 0 in {0 brake, 1 coast, 2 accelerate} */
int hk_set_actions(hk_handle h, const float* steer, const int32_t* branch);

/* n_ticks Unity FixedUpdate ticks: REC.FixedUpdate -> [HKA.FixedUpdate incl. SolveLQR] ->
 * ArcadeKart.FixedUpdate -> engine step (integrate, contacts, triggers). Asynchronous on the handle's stream.
 * Completion.  hk_step returns when the work is ISSUED.  Calls of fewer than 64 ticks, and every call of a handle with a planner or
 * an attached actor, issue a fixed number of kernel rounds (the solve ticks the call can hold + 1) and need no host sync.  Calls of
 * >= 64 ticks of plain handles issue the rounds a field without queued games needs and leave the rest to the NEXT entry point that
 * touches the state (any hk_get_* / hk_set_* / hk_reset / hk_step / hk_prof_read / hk_gather_results, or hk_synchronize): it waits
 * for a two-word report of the device and issues what the laggard envs still need ("lazy completion").  hk_synchronize is the
 * completion point: a host that overlaps its own work with hk_step, or times it, calls hk_synchronize where it needs the ticks done.
 * Round 6: a lazily completed call of >= 512 ticks paces itself: the host stays at most 32 rounds (128 ticks) of launches ahead of the GPU (a marker event
 * every 16 rounds, a wait for the marker two back) so that it can look at the games meter and change where multi-player games are solved while the call runs —
 * such a call returns when its LAST 32 rounds are issued, not at once.  Shorter calls return as soon as they are issued, as before.
 * Round 5: a fixed-round call of a plain handle whose envs are all believed to stand on the same episode step (a reset of every env, then only hk_step
 * calls) issues exactly the launches such a field needs — a one-tick call off a solve tick is one launch — and the completion guard verifies the
 * belief: the next entry point other than hk_step looks at it and, if an env fell behind (it finished its race, a time-out), finishes that env the
 * lazy way before anything is read.  Successive hk_step calls do not look; hk_synchronize does.
 * Error surface: hk_step only reports launch errors.  A completion guard (env_check_kernel, or the last tick launch of a fixed-round call) runs after the rounds of a call; if an
 * env still had ticks to run (an internal scheduling error) the next hk_get_agent_state / hk_get_env_state fails ONCE with
 * HK_ERR_HIP "an env did not complete its ticks" instead of returning stale state; the flag is sticky until that report (or
 * hk_reset), and the unfinished envs keep their leftover ticks, which the next hk_step runs first.  A zero pivot in an LQ solve is
 * sticky in the same way (status bit 0) but does not fail the getters: the reference throws nothing there either (MathNet returns
 * inf / NaN), and hk_env_state.status bit 0 flags the karts whose state went non-finite.
 * Scheduling switches, read from the environment ONCE in hk_create (none changes a result bit; hk_schedule_info() reports what a call ran):
 * HK_FISSION (0: every handle on the fused tick kernel instead of the tick kernel without phase B1 + env_b1_kernel per solve cadence), HK_SPLIT (1: two halves on
 * two streams in every call of a plain handle whatever its size, 0: one stream always; unset: every call of a plain handle of >= 8 192 envs), HK_LAZY_JOIN (0: the
 * halves of a split call are joined into hk_stream at the end of every call instead of by the next entry point that needs the whole state), HK_INWAVE (0: multi-player games through the queues and a solver launch, 1: solved by the
 * B1 waves that assembled them in every round; unset: in-wave once the field has spread), HK_LQN (pair: the pair / matrix-core solver launch also for a spread
 * field), HK_FIXED_ROUNDS, HK_NO_OPTIMISTIC (fixed-round calls issue the worst-case round count instead of the verified plan of a field in lock-step),
 * HK_OPTIMISTIC_SKEW (tests), HK_MCTS_NO_PAUSE, HK_MCTS_NO_OVERLAP / HK_MCTS_SIDE_WAVES (a replan's searches on the handle's stream after the stretch / search
 * workgroup size beside the ticks), HK_DEBUG_MAX_ROUNDS (diagnostic); read at table / buffer set-up: HK_MCTS_PERSIST_GB, HK_NO_HOLD_DEDUPE, HK_LQ_DEBUG,
 * HK_TAB_GLOBAL (track tables read from global memory, as for tracks that exceed the LDS budget).  Round 6 retired the switches whose A/B was settled
 * (profiles/README.md keeps their numbers).
 * Planner handles (any HighMode MCTS agent): a call of more than 38 ticks without attached actors synchronises with the host
 * between stretches of ~100 ticks — envs wait at the tick boundary after a search request so that the searches of a stretch run
 * as ONE batch (results do not depend on it; environment variable HK_MCTS_NO_PAUSE=1 restores the fully asynchronous schedule).
 * Where the planner's trees live is decided at hk_create: one arena slice per agent if that fits HK_MCTS_PERSIST_GB (environment,
 * default 64), else one per resident search lane with re-searched roots rebuilt by replay; plans are bit-identical either way. */
int hk_step(hk_handle h, int n_ticks);

/* HierarchicalKartAgent.CollectObservations (HKA:485-604): obs[E][A][hk_obs_dim], order exactly as the reference adds them */
int hk_obs_dim(hk_handle h);
int hk_get_observations(hk_handle h, float* obs);

int hk_get_agent_state(hk_handle h, hk_agent_state* out /*[E][A]*/);
int hk_set_agent_state(hk_handle h, const hk_agent_state* in /*[E][A]*/);   /* (MCTS agents: bestStates are copied into the new records on the next tick) */
int hk_get_env_state(hk_handle h, hk_env_state* out /*[E]*/);
int hk_set_env_state(hk_handle h, const hk_env_state* in /*[E]*/);            /* (reserved[0] is stored as 0 and reserved[1] keeps its hint bit only: the progress words are the library's) */
int hk_get_episode_results(hk_handle h, hk_episode_result* out /*[E][A]*/);
/* Agent.SendInfo: reward[E][A] = m_Reward, group_reward[E][A] = m_GroupReward collected since the last call; both reset to 0 */
int hk_get_rewards(hk_handle h, float* reward, float* group_reward);
int hk_get_lq_debug(hk_handle h, int env, int ego, hk_lq_debug* out);
/* planner state of every agent, [E][A] (zeros for agents that are not HighMode MCTS).  Searches are batched over hk_step calls
 * (at most 32 armed ticks, well inside the 41+ ticks a plan has until it is due): this call runs whatever is still queued first,
 * so `pend` always holds the result of the last request. */
int hk_get_mcts_state(hk_handle h, hk_mcts_state* out /*[E][A]*/);

/* KartLQR.solveFeedbackLQR (AI/LQR/KartLQR.cs:17) batched, 1:1 incl. quirks Q1 (block-transposed LHS) and Q2.
 * A[b][N][4][4], B[b][N][4][2], Q[b][N][n][n], q[b][N][n], R[b][N][2][2], x0[b][n], n = 4N; u0_out[b][2].
 * Host pointers; h may be NULL (a temporary context on device 0 is used). N <= 8
 * (the env path uses N <= 4 = agents per env; 5..8 run the same generic core, untuned). */
int hk_lq_solve_batch(hk_handle h, int batch, int N, const double* A, const double* B, const double* Q, const double* q,
                      const double* R, const double* x0, int horizon, double* u0_out);

/* device-resident variants used by bench.py / torch (pointers are HIP device pointers; stream = hipStream_t or NULL) */
int hk_lq_solve_batch_device(hk_handle h, int batch, int N, const double* dA, const double* dB, const double* dQ,
                             const double* dq, const double* dR, const double* dx0, int horizon, double* du0, void* stream);
/* The three state pointers below first settle the handle as any getter does (laggards of a lazily completed call, the completion guard of optimistic
 * short calls, a search launch on the planner's side stream — each may synchronise with the device), then return the buffer; NULL on failure
 * (hk_last_error).  What a pointer shows is complete once hk_stream has been synchronised; after further hk_step calls request it again. */
void* hk_device_results_ptr(hk_handle h);  /* hk_episode_result[E][A] on device: the payload of the RCCL all-gather */
void* hk_device_agents_ptr(hk_handle h);   /* hk_agent_state[E][A] on device: a SNAPSHOT as of this call (the library keeps the per-tick fields in a
                                            * wave-tiled layout of its own and gathers them into these records here, asynchronously on hk_stream;
                                            * write kart states with hk_set_agent_state, not through this pointer) */
/* Device-resident RL loop (an external trainer that keeps its tensors on the GPU): hk_observe runs CollectObservations into
 * the library's device buffer (asynchronous, on the handle's stream; raises the HitWall / HitOpponent reward events like
 * hk_get_observations); hk_rewards_device moves m_Reward / m_GroupReward into the two device buffers and zeroes the
 * accumulators (Agent.SendInfo); the trainer writes its actions straight into the action buffers before hk_step.
 * All four buffers are [E][A] (obs: [E][A][hk_obs_dim]); synchronise with hk_stream / hk_synchronize. */
int hk_observe(hk_handle h);
int hk_rewards_device(hk_handle h);
void* hk_device_obs_ptr(hk_handle h);           /* float[E][A][hk_obs_dim] */
void* hk_device_reward_ptr(hk_handle h);        /* float[E][A], valid after hk_rewards_device */
void* hk_device_group_reward_ptr(hk_handle h);  /* float[E][A], valid after hk_rewards_device */
void* hk_device_act_steer_ptr(hk_handle h);     /* float[E][A]: continuous action 0 */
void* hk_device_act_branch_ptr(hk_handle h);    /* int32[E][A]: discrete action 0 */
void* hk_stream(hk_handle h);              /* hipStream_t the handle launches on.  Short hk_step calls of a large plain handle leave half of the batch on a second
                                            * stream (joined lazily); this getter — like every entry point but hk_step — orders hk_stream behind it first, so work
                                            * a caller enqueues on the returned stream AFTER this call sees every tick issued so far.  Call it again after later
                                            * hk_step calls rather than caching the ordering (the handle itself never needs it: its own entry points join). */
/* What built this libhk.so (round 6): a JSON string compiled into the library at link time — hipcc / clang version, the flags, the hidden back-end switches the
 * toolchain accepted and, per translation unit, the code-generation guard variant it shipped with.  bench.py prints it in config.build. */
const char* hk_build_info(void);
/* The schedule the last hk_step of this handle ran (round 6; a JSON string owned by the handle, valid until the next hk_step): fixed / lazy / pause rounds,
 * fission or fused kernel, streams, the optimistic plan, where multi-player games are solved.  bench.py prints it in config.schedule. */
const char* hk_schedule_info(hk_handle h);
int hk_synchronize(hk_handle h);

/* ---- RL low-level policy inference on device ---------------------------------------------------------
 * Replaces the Barracuda execution of the ML-Agents 2.0.1 PPO actor that drives LowMode == RL agents
 * (BehaviorParameters.Model -> Agent.OnActionReceived KA:440 -> InterpretDiscreteActions HKA:1371-1379).  The network
 * is what mlagents-learn exports (the .onnx files under Assets/Karting/Prefabs/AI/, graph probed with
 * tools/onnx_read.py):  x = clip((obs - mean) / std, -5, 5);  n_layers x { x = swish(W x + b) };
 * mu = W_mu x + b_mu;  logits = W_br x + b_br;
 *   continuous_actions[0] = clip(mu + eps * exp(log_sigma), -3, 3) / 3,  eps ~ N(0,1)   (deterministic: eps = 0)
 *   discrete_actions[0]   ~ Categorical(softmax(logits))                                (deterministic: argmax)
 * obs = the agent's last `stack` CollectObservations vectors, oldest first, zero-filled after an episode reset
 * (ML-Agents StackingSensor).  A decision is taken on every tick t of the handle with t % decision_period == 0
 * (DecisionRequester, scene override DecisionPeriod = 2) and the action is repeated in between (TakeActionsBetweenDecisions).
 * Barracuda's random stream is not reproducible outside Unity: eps and the categorical draw come from Philox-4x32
 * keyed by `seed` (counter = decision index, row) instead — "parity unpinned" for the sampled outputs, exact for mu/logits.
 * All arithmetic is fp32: the GEMMs run on the f32-input MFMA, whose result is bit-for-bit a k-ascending fmaf chain
 * seeded with the bias, which is what the CPU oracle computes. */
#define HK_MAX_POLICIES 4
#define HK_POLICY_MAX_LAYERS 4
#define HK_POLICY_MAX_IN 1280     /* obs_dim * stack; HierarchicalKartAgent models: 216 (A = 2), 312 (A = 4), 624 (A = 4, stack 8) */
#define HK_POLICY_MAX_HIDDEN 256

typedef struct hk_policy_desc {
    int32_t in_dim;            /* obs_dim * stack; even */
    int32_t stack;             /* NumStackedVectorObservations (4) */
    int32_t hidden;            /* 32..256, multiple of 32 */
    int32_t n_layers;          /* 1..HK_POLICY_MAX_LAYERS */
    int32_t n_branch;          /* size of discrete branch 0 (3: brake / coast / accelerate) */
    int32_t normalize;         /* 1: apply (obs - mean) / std and the +-5 clip */
    int32_t deterministic;     /* 1: deterministic_continuous_actions / deterministic_discrete_actions */
    uint32_t seed;
    const float* norm_mean;    /* [in_dim] */
    const float* norm_std;     /* [in_dim] the exported divisor sqrt(variance / steps) */
    const float* W[HK_POLICY_MAX_LAYERS];  /* [hidden][k] row-major (torch Linear.weight); k = in_dim for layer 0, else hidden */
    const float* b[HK_POLICY_MAX_LAYERS];  /* [hidden] */
    const float* W_mu;         /* [hidden] */
    const float* b_mu;         /* [1] */
    const float* log_sigma;    /* [1] */
    const float* W_branch;     /* [n_branch][hidden] */
    const float* b_branch;     /* [n_branch] */
} hk_policy_desc;

/* Upload a policy and bind it to the agent slots listed (all must be LowMode == RL; hk_obs_dim * stack must equal
 * in_dim).  Returns the policy index (>= 0) or a negative hk_status.  decision_period (DecisionRequester) is per handle:
 * a second attach with a different period is refused with HK_ERR_INVALID. */
int hk_policy_attach(hk_handle h, const hk_policy_desc* desc, const int32_t* agent_slots, int n_slots, int decision_period);
/* The MLP alone on caller-supplied stacked observations (host pointers): mu[rows], logits[rows][n_branch]. */
int hk_policy_forward(hk_handle h, int policy, int rows, const float* obs /*[rows][in_dim]*/, float* mu, float* logits);
/* The actions currently latched for every agent (what the policies / hk_set_actions wrote): steer[E][A], branch[E][A] */
int hk_get_actions(hk_handle h, float* steer, int32_t* branch);

/* ---- multi-GPU: the path's ONE exchange step ---------------------------------------------------------------
 * Race instances are independent (one RacingEnvController owns its own Agents[] / Sections[], REC:46-52): every rank steps its
 * own contiguous env-id range (hk_config.env_id_base) and nothing crosses GPUs on the data path.  What a host wants at the end
 * of an episode batch — the reference writes it to ExperimentLogs/ (REC:249-265) — is every race's result: one all-gather of
 * hk_episode_result[E][A] over RCCL (xGMI inside a node).  librccl.so is loaded on the first hk_comm_* call, so single-GPU users
 * do not need it.  Bootstrap as with NCCL: rank 0 obtains an id, the host passes it to every rank by whatever channel it has
 * (MPI, a file, torch.distributed's store, ...), every rank calls hk_comm_init.  One communicator per handle. */
#define HK_COMM_ID_BYTES 128
int hk_comm_unique_id(void* id_out /*[HK_COMM_ID_BYTES]*/);
int hk_comm_init(hk_handle h, int world_size, int rank, const void* id /*[HK_COMM_ID_BYTES]*/);
/* all[E_total][A] (host pointer), the ranks' rows in rank order.  Ranks may hold different env counts (a contiguous split of a
 * total the world size does not divide): the byte counts are exchanged first and the contributions padded on the wire.
 * hk_gather_count returns E_total (the sum of every rank's num_envs) so the caller can size `all`. */
int hk_gather_count(hk_handle h, int64_t* total_envs);
int hk_gather_results(hk_handle h, hk_episode_result* all);
int hk_comm_destroy(hk_handle h);

/* timing taps for bench.py's roofline object: accumulated HIP-event time (ms) and launch count per kernel stage since
 * the last hk_prof_reset.  Stages: [0] env_run_kernel (the fused tick kernel), [1] the lqn_kernel<2,3,4> launches of a
 * round (one bracket), [2] lq_batch_kernel, [3] policy_mlp_kernel,
 * [4] env_observe_kernel + policy_stack_kernel of a decision tick, [5] env_b1_kernel (phase B1 of the solve tick when the tick kernel runs without it).  Events are recorded on the handle's own stream around every launch
 * (no host sync per launch; the tick kernel and the solver launch of a round share the event between them) and folded when read. */
#define HK_PROF_STAGES 6
int hk_prof_enable(hk_handle h, int on);
int hk_prof_reset(hk_handle h);
int hk_prof_read(hk_handle h, double* ms /*[HK_PROF_STAGES]*/, int64_t* launches /*[HK_PROF_STAGES]*/);
/* multi-player LQ games (KartLQR.solveFeedbackLQR calls with N >= 2 players) the solver kernels ran since the last hk_prof_reset,
 * by player count: games[N], N = 2 .. HK_MAX_AGENTS (single-player games are solved inside the tick / B1 kernel and not counted).  games[0], games[1] (round 6):
 * the solver passes the waves of env_b1_kernel ran in-wave, and the waves that ran any — their ratio says how well the regroup keeps the envs that hold
 * games apart (1.00: one pass per wave). */
int hk_prof_games(hk_handle h, int64_t* games /*[HK_MAX_AGENTS + 1]*/);

#ifdef __cplusplus
}
#endif
#endif /* HK_H */
