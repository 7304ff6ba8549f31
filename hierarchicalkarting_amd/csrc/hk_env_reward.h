// hk_env_reward.h — reward shaping inside the fused tick kernel (only in the env_run_kernel<*, true>
// instantiations, i.e. when hk_config.rewards != 0).
//
// Reference: KartAgent.cs :165 (NotAtGoalPenalty), :380-400, :440-470 (OnActionReceived), HierarchicalKartAgent.cs :457-480
// (reward dividers), :611-675 (swerving / reverse penalties), RacingEnvController.cs :174-237 (AddGoalTimingRewards),
// :359-433 (ApplySectionRewardsAndPenalties).  The three ML-Agents accumulators of an agent (m_CumulativeReward, m_Reward,
// m_GroupReward) ride in registers (RwAcc) next to the Hot fields for the ticks of a launch.
// Order inside a tick: Academy step (OnActionReceived) -> REC.FixedUpdate (goal-timing rewards before ResetGame) ->
// KA.FixedUpdate (NotAtGoalPenalty) -> triggers.  Trigger rewards touch state shared by the env's agents
// (minSectionTimes / agentsPastSection, group rewards): they are recorded per lane during the trigger loop and replayed
// agent by agent afterwards (rw_replay_events), which is the order the engine restatement dispatches the triggers in.
// CollectObservations raises HitWall / HitOpponent from its ray distances (HKA:580-598): env_observe_kernel records a code
// per sensor, reward_hits_kernel replays them per env in agent / sensor order (REC.ResolveEvent :444-462).
// (included once per lane-group width by hk_env_ga.h: no include guard, namespace hk::HK_GA_NS)
#include "hk_env_device.h"

namespace hk { namespace HK_GA_NS {

struct RwAcc { float cum, step, group; };

struct RwEvent {        // one OnTriggerEnter outcome of this lane's kart, replayed in rw_replay_events
    int kind;           // 1: reached a section (forward), 2: drove back through one
    int section;        // kind 1: the new m_SectionIndex;  kind 2: old section - index + 1
    int swerve;         // kind 1: SwervingPenalty applies (HKA:636-640)
    float lane_div, vel_div;
};
constexpr int RW_MAX_EVENTS = 2;

__device__ __forceinline__ void rw_add(RwAcc& r, float x) { r.step += x; r.cum += x; }     // Agent.AddReward

// Academy step -> KartAgent.OnActionReceived KA:440-470 (active agents)
__device__ inline void rw_academy(const EnvParams& P, const TabView& T, const int env, const int i, const Hot& h, const float hfx,
                                  const float hfz, const hk_agent_state* a, const int* act_branch, RwAcc& r)
{
    const uint32_t fl = h.flags;
    if (!(fl & HK_F_ENABLED) || !(fl & HK_F_ACTIVE)) return;
    bool accel = (fl & HK_F_ACCEL) != 0, brake = (fl & HK_F_BRAKE) != 0;
    if (P.low_mode[i] == HK_LOW_RL) {                                // InterpretDiscreteActions ran first (KA:448)
        const int br = act_branch[(size_t)env * P.A + i];
        accel = br > 1; brake = br < 1;
    }
    const int next = (h.section_index + 1) % P.L;
    float cx, cz;
    lane_marker(T, next, a->plan_lane[next], cx, cz);                // plan lane marker, or the Trigger when there is no entry
    float dx = cx - h.px, dy = T.sec[next].marker_y - P.kart_y, dz = cz - h.pz;
    const float dm = sqrtf(dx * dx + dy * dy + dz * dz);
    if (dm > 1e-5f) { dx = dx / dm; dy = dy / dm; dz = dz / dm; } else { dx = 0.0f; dy = 0.0f; dz = 0.0f; }   // Vector3.normalized
    float vx = h.vx, vy = 0.0f, vz = h.vz;
    const float vm = sqrtf(vx * vx + vy * vy + vz * vz);
    if (vm > 1e-5f) { vx = vx / vm; vy = vy / vm; vz = vz / vm; } else { vx = 0.0f; vy = 0.0f; vz = 0.0f; }
    const float reward = vx * dx + vy * dy + vz * dz;
    rw_add(r, reward * P.rw.TowardsCheckpointReward);
    rw_add(r, (accel && !brake ? 1.0f : 0.0f) * P.rw.AccelerationReward);
    float ls = 0.0f;                                                 // ArcadeKart.LocalSpeed AK:325-342
    if (fl & HK_F_CAN_MOVE) {
        const float dot = hfx * h.vx + hfz * h.vz;
        if (f_abs(dot) > 0.1f) {
            const float speed = sqrtf(h.vx * h.vx + 0.0f * 0.0f + h.vz * h.vz);
            ls = dot < 0 ? -(speed / P.st.ReverseSpeed) : (speed / P.st.TopSpeed);
        }
    }
    const float speedProportion = 0.00f;
    rw_add(r, (ls - speedProportion) / (1 - speedProportion) * P.rw.SpeedReward);
}

__device__ __forceinline__ void rw_not_at_goal(const EnvParams& P, const Hot& h, RwAcc& r)
{   // KA:165
    if ((h.flags & HK_F_ACTIVE) || h.section_index != P.laps * P.L + 1) rw_add(r, P.rw.NotAtGoalPenalty);
}

// HKA.setLaneDifferenceDivider :457-468, setVelocityDifferenceDivider :473-480 (the plan entry of `key` still exists)
__device__ inline void rw_dividers(const EnvParams& P, const TabView& T, const int i, const int key, const int lane, const int plan_lane,
                                   const float plan_vel, const float px, const float pz, const float vx, const float vz,
                                   float& lane_div, float& vel_div)
{
    lane_div = 1.0f; vel_div = 1.0f;
    if (lane != -1) {
        float lx, lz;
        lane_marker(T, key, plan_lane, lx, lz);
        const float dx = lx - px, dy = T.sec[key].marker_y - P.kart_y, dz = lz - pz;
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        if ((double)d > 1.3) lane_div = (float)hk_exp((double)(1.0f * d) * hk_log((double)1.3f));      // Mathf.Pow(1.3f, d)
    }
    if (P.high_mode[i] == HK_HIGH_FIXED) return;
    const float velocity = sqrtf(vx * vx + 0.0f * 0.0f + vz * vz);
    const float dv = f_abs(velocity - plan_vel);
    if (dv > (float)P.vbucket[i] / 2.0f) vel_div = (float)hk_exp((double)(1.0f * dv) * hk_log((double)1.1f));
}

__device__ __forceinline__ int rw_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rw_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// REC.ApplySectionRewardsAndPenalties :359-433 for this lane's kart; returns the group reward its team receives
__device__ inline float rw_section(const EnvParams& P, const RwDev& R, const int env, const int i, const int s, const int steps,
                                   const float lane_div, const float vel_div, RwAcc& r)
{
    rw_add(r, P.rw.PassCheckpointLaneReward / lane_div);             // KA:397
    rw_add(r, P.rw.PassCheckpointVelocityReward / vel_div);          // KA:398
    if (s < 0 || s >= R.S) return 0.0f;
    const int team = P.team_of[i];
    int* mt = R.sec_time + (size_t)env * P.A * R.S;
    int* cn = R.sec_cnt + (size_t)env * P.A * R.S;
    int total = 0;
    const int mine = rw_ld(&mt[team * R.S + s]);
    if (mine < 0) {
        rw_st(&mt[team * R.S + s], steps); rw_st(&cn[team * R.S + s], 1);
        for (int t = 0; t < P.n_teams; t++) {
            if (t == team) continue;
            const int ot = rw_ld(&mt[t * R.S + s]);
            if (ot < 0) continue;
            const int oc = rw_ld(&cn[t * R.S + s]);
            rw_add(r, P.rw.BeingBehindOpponentCheckpointPenalty * ((float)steps - (float)ot) * (float)oc / (1.0f * (float)(P.A - P.team_size[t])));
            total += oc;
        }
        total += 1;
    } else {
        for (int t = 0; t < P.n_teams; t++) {
            const int ot = t == team ? mine : rw_ld(&mt[t * R.S + s]);
            if (t == team) {
                const int oc = rw_ld(&cn[t * R.S + s]);
                rw_add(r, P.rw.BeingBehindTeammateCheckpointPenalty * ((float)steps - (float)ot) * (float)oc / (1.0f * (float)P.team_size[t]));
            } else if (ot >= 0) {
                const int oc = rw_ld(&cn[t * R.S + s]);
                rw_add(r, P.rw.BeingBehindOpponentCheckpointPenalty * ((float)steps - (float)ot) * (float)oc / (1.0f * (float)(P.A - P.team_size[t])));
                total += oc;
            }
        }
        rw_st(&cn[team * R.S + s], rw_ld(&cn[team * R.S + s]) + 1);
        total += 1;
    }
    const int q = (total - 1) < 3 ? (total - 1) : 3;
    const float m4 = q == 1 ? 0.75f : (q == 2 ? 0.6f : 0.4f);
    const float aMult = q == 0 ? P.rw.PassCheckpointTimeMultiplier : P.rw.PassCheckpointTimeMultiplier * m4;
    const float aBase = q == 0 ? P.rw.PassCheckpointBase : P.rw.PassCheckpointBase * m4;
    rw_add(r, aBase + aMult * (float)(P.max_steps - steps) / (1.0f * (float)P.max_steps));
    const float gMult = q == 0 ? P.rw.TeamPassCheckpointTimeMultiplier : P.rw.TeamPassCheckpointTimeMultiplier * m4;
    const float gBase = q == 0 ? P.rw.TeamPassCheckpointBase : P.rw.TeamPassCheckpointBase * m4;
    return gBase + gMult * (float)(P.max_steps - steps) / (1.0f * (float)P.max_steps);
}

// Replay the trigger outcomes of the env's agents in agent order (every lane of the quad must call this).
// en_before / en_after: this lane's GameObject enabled before / after its own trigger callbacks of this tick.
__device__ inline void rw_replay_events(const EnvParams& P, const RwDev& R, const int env, const int i, const int steps,
                                        const RwEvent* ev, const int nev, const bool en_before, const bool en_after, RwAcc& r)
{
    if (!group_or(nev)) return;
    for (int j = 0; j < P.A; j++) {
        float gadd[RW_MAX_EVENTS] = {0.0f, 0.0f};
        int gn = 0;
        if (i == j) {
            for (int k = 0; k < nev; k++) {
                if (ev[k].kind == 1) {
                    if (ev[k].swerve) rw_add(r, P.rw.SwervingPenalty);                           // HKA:638
                    gadd[gn++] = rw_section(P, R, env, i, ev[k].section, steps, ev[k].lane_div, ev[k].vel_div, r);
                } else if (ev[k].kind == 2) {
                    rw_add(r, P.rw.ReversePenalty * (float)ev[k].section);                       // HKA:666
                }
            }
        }
        __threadfence();                 // lane j's table updates are visible to the lanes replayed after it
        const int gn_j = quad_get(gn, j);
        for (int k = 0; k < RW_MAX_EVENTS; k++) {
            const float g = quad_get(gadd[k], j);
            // SimpleMultiAgentGroup.AddGroupReward: the team's registered (= enabled) agents at that moment
            if (k < gn_j && i < P.A && P.team_of[i] == P.team_of[j] && (i < j ? en_after : en_before)) r.group += g;
        }
    }
}

// REC.AddGoalTimingRewards :174-237 (all lanes; `enabled` = this lane's GameObject after the Deactivate-all of :243-247)
__device__ inline void rw_goal_timing(const EnvParams& P, const int i, const int time_steps, const bool enabled, RwAcc& r)
{
    const int A = P.A, maxs = P.max_steps;
    int ts[GA];
    for (int j = 0; j < GA; j++) { const int t = quad_get(time_steps, j); ts[j] = t == 0 ? 5 * maxs : t; }
    if (i >= A) return;
    if (A == 1) {
        if (time_steps != 0) rw_add(r, P.rw.ReachGoalCheckpointRewardMultplier * (1.0f - (float)time_steps * 1.0f / (float)maxs) + P.rw.ReachGoalCheckpointRewardBase);
        return;
    }
    const float maxReward = 1.0f, minReward = -1.0f;
    float mine = 0.0f;                                   // groupRewards[team of this lane], accumulated in agent order
    for (int k = 0; k < A; k++) {
        if (!P.training_agent[k] || P.team_of[k] != P.team_of[i]) continue;
        int teamScore = 0, oppScore = 0;
        for (int j = 0; j < P.n_other[k]; j++) oppScore += ts[P.other[k][j]];
        for (int j = 0; j < P.n_team[k]; j++) teamScore += ts[P.team[k][j]];
        const float finalCur = (float)ts[k] + (float)teamScore * P.rw.TeamScoreRewardMultiplier;
        const float finalOpp = (float)oppScore * (1.0f + (float)P.n_team[k] * P.rw.TeamScoreRewardMultiplier) / ((float)P.n_other[k] * 1.0f);
        const float gt = ((finalOpp - finalCur) / (1.0f + (float)P.n_team[k] * P.rw.TeamScoreRewardMultiplier)) / (float)maxs;
        mine += P.rw.ReachGoalCheckpointRewardBase + P.rw.ReachGoalCheckpointRewardMultplier * ((gt - minReward) * 1.0f / (maxReward - minReward));
    }
    if (enabled) r.group += mine / (float)P.team_size[P.team_of[i]];
}

// REC.ResolveEvent :438-462 for the HitWall / HitOpponent events of the last observation pass; one thread per env
__global__ __launch_bounds__(128) void reward_hits_kernel(EnvParams P, hk_agent_state* agents, const uint32_t* hot, const int* slot_of, const unsigned char* hit_code)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= P.E) return;
    hk_agent_state* ags = agents + (size_t)env * P.A;
    const uint32_t* hrow = hot + hot_base<GA>(slot_of[env], 0);
    for (int i = 0; i < P.A; i++) {
        if (!(hot_get<uint32_t>(hrow + i, HF_flags) & HK_F_ACTIVE)) continue;             // :441
        for (int si = 0; si < HK_NUM_SENSORS; si++) {
            const int code = hit_code[((size_t)env * P.A + i) * HK_NUM_SENSORS + si];
            if (code == 0) continue;
            if (code == 1) { const float x = 1.0f * P.rw.WallHitPenalty; ags[i].step_reward += x; ags[i].cum_reward += x; continue; }
            const int v = code - 2;
            { const float x = 1.0f * P.rw.OpponentHitPenalty; ags[i].step_reward += x; ags[i].cum_reward += x; }
            if (P.team_of[i] == P.team_of[v]) {
                { const float x = 1.5f * P.rw.OpponentHitPenalty; ags[i].step_reward += x; ags[i].cum_reward += x; }
                { const float x = 1.15f * P.rw.HitByOpponentPenalty; ags[v].step_reward += x; ags[v].cum_reward += x; }
            } else { const float x = 1.0f * P.rw.HitByOpponentPenalty; ags[v].step_reward += x; ags[v].cum_reward += x; }
        }
    }
}

// hk_get_rewards: Agent.SendInfo reads and zeroes m_Reward / m_GroupReward
__global__ __launch_bounds__(256) void rewards_read_kernel(hk_agent_state* agents, int n, float* reward, float* group_reward)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    reward[t] = agents[t].step_reward; group_reward[t] = agents[t].group_reward;
    agents[t].step_reward = 0.0f; agents[t].group_reward = 0.0f;
}

} }  // namespace hk::HK_GA_NS
