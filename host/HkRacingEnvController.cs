// HkRacingEnvController.cs — RacingEnvController's role (REC:44-797) for E race instances stepped on the GPU by libhk.so.
//
// Start        scene -> HkConfig (Sections[] -> HkSection, wall polylines -> HkWallSeg, Agents[].teamAgents / otherAgents ->
//              wiring, rules and reward fields as serialised on the reference's controller) -> hk_create -> hk_reset
// FixedUpdate  RL actions -> hk_set_actions; hk_step(1) = one Unity tick of every instance (REC.FixedUpdate, the agents'
//              FixedUpdate incl. SolveLQR, ArcadeKart.FixedUpdate, the engine step); poses of DisplayEnv -> GameObjects;
//              finished episodes -> ExperimentLogs/<ExperimentName>.txt in TelemetryViewer's text format (REC:249-265)
// Source only: dotnet / Unity are not part of the build image.
using System;
using System.IO;
using System.Text;
using UnityEngine;
using KartGame.AI;
using KartGame.AI.Native;

public unsafe class HkRacingEnvController : MonoBehaviour
{
    [Header("Batch")]
    public int NumEnvs = 4096;
    public int DeviceId = 0;
    public int EnvIdBase = 0;              // multi-GPU: global id of this process's env 0
    public int DisplayEnv = 0;             // the instance mirrored into the scene's GameObjects
    [Header("Scene (as RacingEnvController)")]
    public HkKartAgent[] Agents;
    public DiscretePositionTracker[] Sections;
    public TextAsset WallPolylines;        // road-side wall faces, one "x0 z0 x1 z1" per line (tools/extract_track.py writes them)
    public EnvironmentMode mode = EnvironmentMode.Experiment;
    public string ExperimentName = "hk";
    [Header("Rules (REC:110-130)")]
    public int MaxLaneChanges = 3;
    public int laps = 4;
    public int maxEpisodeSteps = 6000;
    public bool disableOnEnd = true;
    public int sectionHorizon = 5;
    [Header("Planner budget (stands for the reference's wall-clock T = 0.9 s / 1.5 s)")]
    public int mctsIterations = 128;
    public bool rewards = false;

    IntPtr h = IntPtr.Zero;
    HkAgentState[] state;
    HkEpisodeResult[] results;
    float[] actSteer;
    int[] actBranch;
    int[] seenEpisode;
    bool actionsDirty;
    public float[] Observations { get; private set; }
    public int ObsDim { get; private set; }

    void Start()
    {
        int A = Agents.Length, L = Sections.Length;
        var secs = new HkSection[L];
        for (int i = 0; i < L; i++)
        {
            DiscretePositionTracker s = Sections[i];
            secs[i].trig_x = s.Trigger.transform.position.x; secs[i].trig_z = s.Trigger.transform.position.z;
            secs[i].yaw_deg = s.transform.eulerAngles.y; secs[i].marker_y = s.Trigger.transform.position.y;
            BoxCollider[] lanes = { s.Lane1, s.Lane2, s.Lane3, s.Lane4 };
            for (int l = 0; l < 4; l++) { secs[i].lane_x[l] = lanes[l].transform.position.x; secs[i].lane_z[l] = lanes[l].transform.position.z; }
            secs[i].track_inside_radius = s.trackInsideRadius; secs[i].track_length = s.trackLength; secs[i].track_width = s.trackWidth;
            secs[i].turn_degrees = s.turnDegrees; secs[i].left_turn = s.leftTurn ? 1 : 0; secs[i].optimal_lane = s.optimalLane;
        }
        string[] lines = WallPolylines.text.Split(new[] { '\n' }, StringSplitOptions.RemoveEmptyEntries);
        var walls = new HkWallSeg[lines.Length];
        for (int i = 0; i < lines.Length; i++)
        {
            string[] f = lines[i].Split(' ');
            walls[i].x0 = float.Parse(f[0]); walls[i].z0 = float.Parse(f[1]); walls[i].x1 = float.Parse(f[2]); walls[i].z1 = float.Parse(f[3]);
        }
        var cfg = new HkConfig();
        cfg.abi_version = HkConst.HK_ABI_VERSION; cfg.num_envs = NumEnvs; cfg.num_agents = A; cfg.device_id = DeviceId;
        for (int i = 0; i < A; i++)
        {
            HkKartAgent a = Agents[i];
            a.envController = this; a.agentIndex = i;
            cfg.n_team[i] = a.teamAgents.Length; cfg.n_other[i] = a.otherAgents.Length;
            for (int j = 0; j < a.teamAgents.Length; j++) cfg.team_agents[i * 8 + j] = Array.IndexOf(Agents, a.teamAgents[j]);
            for (int j = 0; j < a.otherAgents.Length; j++) cfg.other_agents[i * 8 + j] = Array.IndexOf(Agents, a.otherAgents[j]);
            cfg.high_mode[i] = (int)a.HighMode; cfg.low_mode[i] = (int)a.LowMode;
            cfg.tree_search_depth[i] = a.treeSearchDepth; cfg.velocity_bucket_size[i] = a.velocityBucketSize;
            cfg.time_precision[i] = a.timePrecision; cfg.section_window[i] = a.sectionWindow;
            cfg.training_agent[i] = a.Mode == AgentMode.Training ? 1 : 0;
        }
        // team ids: agents that list each other as teamAgents share one (REC.Teams, REC:46)
        int teams = 0;
        for (int i = 0; i < A; i++)
        {
            int t = -1;
            for (int j = 0; j < i; j++) if (Array.IndexOf(Agents[i].teamAgents, Agents[j]) >= 0) t = cfg.team_of[j];
            cfg.team_of[i] = t >= 0 ? t : teams++;
        }
        // effective ArcadeKart.Stats of the Compete scenes (SURVEY App. A)
        cfg.stats = new HkKartStats { TopSpeed = 15f, Acceleration = 7f, ReverseSpeed = 10f, ReverseAcceleration = 3f, AccelerationCurve = 0.5f,
                                      Braking = 16f, CoastingDrag = 5f, Grip = 0.97f, MaxSteer = 4f, MinSteer = 1f, TireWearFactor = 0.001f,
                                      MinGs = 0.5f, MaxGs = 2f, AddedGravity = 1f, TireWearRate = 10000f, AngularDrag = 0.05f };
        cfg.laps = laps; cfg.max_episode_steps = maxEpisodeSteps; cfg.max_lane_changes = MaxLaneChanges; cfg.section_horizon = sectionHorizon;
        cfg.disable_on_end = disableOnEnd ? 1 : 0; cfg.env_mode = (int)mode;
        cfg.start_hold_ticks = mode == EnvironmentMode.Training ? 0 : 75;       // WaitForSeconds(1.5f), REC:721-724
        cfg.auto_reset = 1; cfg.dt = Time.fixedDeltaTime; cfg.kart_y = 0.28f;  // REC:715
        // KartAgent.Sensors[] as the Compete scenes hold it (the scenes re-point the prefab's entries): 0, then +30 .. +90, then -30 .. -90.
        // A host with the reference's agents in the scene reads Sensors[k].Transform.localEulerAngles.y instead.
        float[] yaw = { 0f, 30f, 45f, 60f, 90f, -30f, -45f, -60f, -90f };
        float[] wallVal = { 0.8f, 0.9f, 1f, 0.8f, 0.6f, 0.9f, 1f, 0.8f, 0.6f };
        for (int k = 0; k < HkConst.HK_NUM_SENSORS; k++)
        { cfg.sensor_yaw_deg[k] = yaw[k]; cfg.ray_distance[k] = 20f; cfg.wall_hit_validation[k] = wallVal[k]; cfg.agent_hit_validation[k] = 1.5f; }
        // engine restatement (hk.h hk_engine_params): Rigidbody / CapsuleCollider / WheelCollider / KartAnimation values of BaseKartClassic.prefab
        cfg.engine = new HkEngineParams { mass = 250f, inertia_y = 74.71f, gravity = 9.81f, axle_zf = 0.58625615f, axle_zr = -0.68122816f,
                                          max_steer_deg = 30f, steer_damping = 10f, side_ext_slip = 0.2f, side_ext_value = 1f, side_asy_slip = 0.5f,
                                          side_asy_value = 0.75f, side_stiffness = 1f, side_slope0 = 1f, slip_min_speed = 1f, wheel_mass = 20f,
                                          wheel_radius_f = 0.1372984f, wheel_radius_r = 0.1630791f, wheel_damping = 0.56f, fwd_ext_slip = 0.4f, fwd_ext_value = 1f,
                                          fwd_asy_slip = 0.8f, fwd_asy_value = 0.5f, fwd_stiffness = 1f, long_slip_min_speed = 4f,
                                          wheel_friction = 1, contact_yaw = 1, wheel_rolling = 1 };
        cfg.env_id_base = EnvIdBase; cfg.num_sections = L; cfg.num_walls = walls.Length;
        cfg.mcts_iterations = mctsIterations; cfg.mcts_initial_iterations = (mctsIterations * 5 + 2) / 3;
        cfg.mcts_latency_ticks = 45; cfg.mcts_initial_latency_ticks = 75; cfg.mcts_seed = 0x4D435453;
        cfg.rewards = rewards ? 1 : 0; cfg.train_seed = 0x54524149;
        cfg.rw = new HkRewardParams { WallHitPenalty = -0.05f, OpponentHitPenalty = -2f, HitByOpponentPenalty = -2f, PassCheckpointLaneReward = 4f,
                                      PassCheckpointVelocityReward = 4f, PassCheckpointBase = 20f, PassCheckpointTimeMultiplier = 5f,
                                      TeamPassCheckpointBase = 20f, TeamPassCheckpointTimeMultiplier = 5f, BeingBehindOpponentCheckpointPenalty = -0.06f,
                                      BeingBehindTeammateCheckpointPenalty = -0.02f, TeamScoreRewardMultiplier = 0.75f, ReversePenalty = -0.5f,
                                      SwervingPenalty = -0.5f, ReachGoalCheckpointRewardMultplier = 5f, ReachGoalCheckpointRewardBase = 3f,
                                      TowardsCheckpointReward = 0.008f, SpeedReward = 0.07f, SlowMovingPenalty = -3f, AccelerationReward = 0.002f,
                                      NotAtGoalPenalty = -0.001f };
        fixed (HkSection* ps = secs) fixed (HkWallSeg* pw = walls)
        {
            cfg.sections = ps; cfg.walls = pw;                                   // copied by hk_create
            Hk.Check(Hk.hk_create(&cfg, out h), IntPtr.Zero);
        }
        ObsDim = Hk.hk_obs_dim(h);
        int n = NumEnvs * A;
        state = new HkAgentState[n]; results = new HkEpisodeResult[n];
        actSteer = new float[n]; actBranch = new int[n]; seenEpisode = new int[NumEnvs];
        for (int i = 0; i < n; i++) actBranch[i] = 1;                           // "coast"
        for (int e = 0; e < NumEnvs; e++) seenEpisode[e] = -1;
        Observations = new float[n * ObsDim];
        Hk.Check(Hk.hk_reset(h, null, 0, -1), h);                                // Experiment grid, ordering (env_id_base + env) % A!
    }

    // HkKartAgent.OnActionReceived of the displayed instance; a trainer that drives every instance writes the arrays directly
    public void SetAction(int agent, float steer, int branch)
    {
        int k = DisplayEnv * Agents.Length + agent;
        actSteer[k] = steer; actBranch[k] = branch; actionsDirty = true;
    }

    void FixedUpdate()
    {
        if (h == IntPtr.Zero) return;
        if (actionsDirty)
        {
            fixed (float* s = actSteer) fixed (int* b = actBranch) Hk.Check(Hk.hk_set_actions(h, s, b), h);
            actionsDirty = false;
        }
        Hk.Check(Hk.hk_step(h, 1), h);
        fixed (HkAgentState* p = state) Hk.Check(Hk.hk_get_agent_state(h, p), h);
        for (int i = 0; i < Agents.Length; i++) Agents[i].ApplyState(state[DisplayEnv * Agents.Length + i], 0.28f);
        fixed (float* o = Observations) Hk.Check(Hk.hk_get_observations(h, o), h);      // what CollectObservations hands to ML-Agents
        if (mode == EnvironmentMode.Experiment) LogFinishedEpisodes();
    }

    // REC:249-265 + TelemetryViewer.Update :49-108: one text block per finished race
    void LogFinishedEpisodes()
    {
        fixed (HkEpisodeResult* r = results) Hk.Check(Hk.hk_get_episode_results(h, r), h);
        int A = Agents.Length;
        for (int e = 0; e < NumEnvs; e++)
        {
            if (results[e * A].episode == seenEpisode[e]) continue;
            seenEpisode[e] = results[e * A].episode;
            var sb = new StringBuilder();
            sb.AppendLine("Experiment " + seenEpisode[e]);
            for (int i = 0; i < A; i++)
            {
                HkEpisodeResult q = results[e * A + i];
                string n = Agents[i].name;
                sb.AppendLine(n + " Speed: " + q.speed); sb.AppendLine(n + " Reward: " + q.reward);
                sb.AppendLine(n + " Last Lap: " + q.last_lap); sb.AppendLine(n + " Best Lap: " + q.best_lap);
                sb.AppendLine(n + " Total Time: " + q.total_time); sb.AppendLine(n + " Laps Completed: " + q.laps_completed + "/" + laps);
                sb.AppendLine(n + " Illegal Lane Changes: " + q.illegal_lane_changes); sb.AppendLine(n + " Collisions: " + q.forward_collisions);
                sb.AppendLine(n + " Avg Target Lane Difference: " + q.avg_lane_diff); sb.AppendLine(n + " Avg Target Vel Difference: " + q.avg_vel_diff);
            }
            sb.AppendLine("Winner: ");
            File.AppendAllText(Path.Combine("ExperimentLogs", ExperimentName + "_env" + (EnvIdBase + e) + ".txt"), sb.ToString() + "\n");
        }
    }

    void OnDestroy()
    {
        if (h != IntPtr.Zero) { Hk.hk_destroy(h); h = IntPtr.Zero; }
    }
}
