/*
 * hk_oracle_reward.c — CPU ORACLE (test infrastructure): reward shaping of the racing environment, restated from
 *   KA  = AI/KartAgent.cs            :165 (NotAtGoalPenalty), :380-400 (hit / section reward helpers), :440-470 (OnActionReceived)
 *   HKA = AI/HierarchicalKartAgent.cs :457-480 (reward dividers), :611-675 (OnTriggerEnter: swerving / reverse penalties)
 *   REC = RacingEnvController.cs     :174-237 (AddGoalTimingRewards), :359-433 (ApplySectionRewardsAndPenalties), :438-486
 * ML-Agents bookkeeping (Agent.AddReward: m_Reward and m_CumulativeReward; SimpleMultiAgentGroup.AddGroupReward: m_GroupReward of
 * every registered = enabled member; Agent.SendInfo zeroes m_Reward / m_GroupReward) is the published 2.0.1 behaviour; the
 * package source is not in the reference checkout, so that boundary is "parity unpinned".
 * Order inside a tick: Academy step (OnActionReceived rewards, state as the previous tick left it) -> REC.FixedUpdate
 * (AddGoalTimingRewards before ResetGame) -> KA.FixedUpdate (NotAtGoalPenalty) -> physics triggers (section rewards).
 * CollectObservations raises HitWall / HitOpponent from its ray distances (HKA:580-598 -> REC.ResolveEvent :444-462): applied
 * whenever observations are collected (a decision tick of an attached policy, or hko_get_observations).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "hk_oracle.h"
#include "../include/hk_detmath.h"
#include "hk_oracle_internal.h"

static inline void add_reward(hk_agent_state* a, float r) { a->step_reward += r; a->cum_reward += r; }   /* Agent.AddReward */

static void add_group_reward(hko_env* e, int env, int team, float r)
{   /* SimpleMultiAgentGroup.AddGroupReward: registered agents = the team's enabled GameObjects, in Racers order */
    hk_agent_state* ags = &e->ag[(size_t)env * e->A];
    for (int i = 0; i < e->A; i++)
        if (e->cfg.team_of[i] == team && (ags[i].flags & HK_F_ENABLED)) ags[i].group_reward += r;
}

static int team_size(const hko_env* e, int team)
{
    int n = 0;
    for (int i = 0; i < e->A; i++) n += e->cfg.team_of[i] == team;
    return n;
}
static int n_teams(const hko_env* e)
{
    int m = 0;
    for (int i = 0; i < e->A; i++) if (e->cfg.team_of[i] + 1 > m) m = e->cfg.team_of[i] + 1;
    return m;
}

int hko_rw_table_len(const hko_env* e) { return e->cfg.laps * e->L + 2; }

void hko_rw_reset_env(hko_env* e, int env)
{   /* REC.ResetGame :508-512: minSectionTimes / agentsPastSection cleared */
    const size_t n = (size_t)e->A * hko_rw_table_len(e);
    memset(e->sec_min_time + (size_t)env * n, 0xFF, n * sizeof(int32_t));      /* -1 = key absent */
    memset(e->sec_count + (size_t)env * n, 0, n);
}

/* Academy step -> KartAgent.OnActionReceived KA:440-470, for every active agent */
void hko_rw_academy(hko_env* e, int env)
{
    const hk_config* cfg = &e->cfg;
    const hk_reward_params* rw = &cfg->rw;
    hk_agent_state* ags = &e->ag[(size_t)env * e->A];
    for (int i = 0; i < e->A; i++) {
        hk_agent_state* a = &ags[i];
        if (!(a->flags & HK_F_ENABLED) || !(a->flags & HK_F_ACTIVE)) continue;          /* :443 */
        int accel = (a->flags & HK_F_ACCEL) != 0, brake = (a->flags & HK_F_BRAKE) != 0;
        if (cfg->low_mode[i] == HK_LOW_RL) {                                             /* InterpretDiscreteActions ran first :448 */
            int br = e->act_branch[(size_t)env * e->A + i];
            accel = br > 1; brake = br < 1;
        }
        const int next = (a->section_index + 1) % e->L;                                  /* :451 */
        const hk_section* s = &e->sec[next];
        float cx = s->trig_x, cz = s->trig_z;
        if (a->plan_lane[next] != 0) { cx = s->lane_x[a->plan_lane[next] - 1]; cz = s->lane_z[a->plan_lane[next] - 1]; }   /* :452 */
        float dx = cx - a->px, dy = s->marker_y - cfg->kart_y, dz = cz - a->pz;          /* :453 */
        float dm = sqrtf(dx * dx + dy * dy + dz * dz);
        if (dm > 1e-5f) { dx = dx / dm; dy = dy / dm; dz = dz / dm; } else { dx = dy = dz = 0.0f; }   /* Vector3.normalized */
        float vx = a->vx, vy = 0.0f, vz = a->vz;
        float vm = sqrtf(vx * vx + vy * vy + vz * vz);
        if (vm > 1e-5f) { vx = vx / vm; vy = vy / vm; vz = vz / vm; } else { vx = vy = vz = 0.0f; }
        float reward = vx * dx + vy * dy + vz * dz;                                      /* :454 Vector3.Dot */
        add_reward(a, reward * rw->TowardsCheckpointReward);                             /* :460 */
        add_reward(a, (accel && !brake ? 1.0f : 0.0f) * rw->AccelerationReward);         /* :461 */
        float ls = 0.0f;                                                                 /* ArcadeKart.LocalSpeed AK:325-342 */
        if (a->flags & HK_F_CAN_MOVE) {
            float fx = hk_sinf(a->yaw), fz = hk_cosf(a->yaw);
            float dot = fx * a->vx + fz * a->vz;
            if (fabsf(dot) > 0.1f) {
                float speed = sqrtf(a->vx * a->vx + 0.0f * 0.0f + a->vz * a->vz);
                ls = dot < 0 ? -(speed / cfg->stats.ReverseSpeed) : (speed / cfg->stats.TopSpeed);
            }
        }
        const float speedProportion = 0.00f;                                             /* :459, so the slow-moving branch is dead */
        add_reward(a, (ls - speedProportion) / (1 - speedProportion) * rw->SpeedReward); /* :468 */
    }
}

void hko_rw_not_at_goal(hko_env* e, hk_agent_state* a)
{   /* KA:165 */
    if ((a->flags & HK_F_ACTIVE) || a->section_index != e->cfg.laps * e->L + 1) add_reward(a, e->cfg.rw.NotAtGoalPenalty);
}

/* HKA.setLaneDifferenceDivider :457-468 / setVelocityDifferenceDivider :473-480 (called while the plan entry still exists) */
void hko_rw_dividers(hko_env* e, int agent, const hk_agent_state* a, int index, int lane, float* lane_div, float* vel_div)
{
    const int key = index % e->L;
    *lane_div = 1.0f; *vel_div = 1.0f;
    if (lane != -1) {
        const hk_section* s = &e->sec[key];
        const int tl = a->plan_lane[key];
        float dx = s->lane_x[tl - 1] - a->px, dy = s->marker_y - e->cfg.kart_y, dz = s->lane_z[tl - 1] - a->pz;
        float d = sqrtf(dx * dx + dy * dy + dz * dz);
        if ((double)d > 1.3) *lane_div = (float)hk_exp((double)(1.0f * d) * hk_log((double)1.3f));       /* Mathf.Pow(1.3f, d) */
    }
    if (e->cfg.high_mode[agent] == HK_HIGH_FIXED) return;
    float velocity = sqrtf(a->vx * a->vx + 0.0f * 0.0f + a->vz * a->vz);
    float dv = fabsf(velocity - a->plan_vel[key]);
    if (dv > (float)e->cfg.velocity_bucket_size[agent] / 2.0f) *vel_div = (float)hk_exp((double)(1.0f * dv) * hk_log((double)1.1f));
}

void hko_rw_swerve(hko_env* e, hk_agent_state* a) { add_reward(a, e->cfg.rw.SwervingPenalty); }                       /* HKA:638 */
void hko_rw_reverse(hko_env* e, hk_agent_state* a, int old_section, int index)
{   /* HKA:666 */
    add_reward(a, e->cfg.rw.ReversePenalty * (float)(old_section - index + 1));
}

/* REC.ApplySectionRewardsAndPenalties :359-433 (agent already moved to its new section) */
void hko_rw_section(hko_env* e, int env, int ai, float lane_div, float vel_div)
{
    const hk_config* cfg = &e->cfg;
    const hk_reward_params* rw = &cfg->rw;
    hk_agent_state* a = &e->ag[(size_t)env * e->A + ai];
    const int steps = e->es[env].episode_steps, maxs = cfg->max_episode_steps;
    const int S = hko_rw_table_len(e), T = n_teams(e);
    int32_t* mt = e->sec_min_time + (size_t)env * e->A * S;
    uint8_t* cn = e->sec_count + (size_t)env * e->A * S;
    add_reward(a, rw->PassCheckpointLaneReward / lane_div);                              /* KA:397 */
    add_reward(a, rw->PassCheckpointVelocityReward / vel_div);                           /* KA:398 */
    const int team = cfg->team_of[ai], s = a->section_index;
    int total = 0;
    if (s < 0 || s >= S) return;
    if (mt[team * S + s] < 0) {                                                          /* :366 */
        mt[team * S + s] = steps; cn[team * S + s] = 1;
        for (int i = 0; i < T; i++)
            if (i != team && mt[i * S + s] >= 0) {
                add_reward(a, rw->BeingBehindOpponentCheckpointPenalty * ((float)steps - (float)mt[i * S + s]) * (float)cn[i * S + s] /
                                  (1.0f * (float)(e->A - team_size(e, i))));            /* :375 */
                total += cn[i * S + s];
            }
        total += 1;
    } else {
        for (int i = 0; i < T; i++) {
            if (i == team)
                add_reward(a, rw->BeingBehindTeammateCheckpointPenalty * ((float)steps - (float)mt[i * S + s]) * (float)cn[i * S + s] /
                                  (1.0f * (float)team_size(e, i)));                     /* :388 */
            else if (mt[i * S + s] >= 0) {
                add_reward(a, rw->BeingBehindOpponentCheckpointPenalty * ((float)steps - (float)mt[i * S + s]) * (float)cn[i * S + s] /
                                  (1.0f * (float)(e->A - team_size(e, i))));            /* :392 */
                total += cn[i * S + s];
            }
        }
        cn[team * S + s] += 1;
        total += 1;
    }
    const float m4[4] = {1.0f, 0.75f, 0.6f, 0.4f};
    const int q = (total - 1) < 3 ? (total - 1) : 3;
    const float aMult = q == 0 ? rw->PassCheckpointTimeMultiplier : rw->PassCheckpointTimeMultiplier * m4[q];   /* :413 */
    const float aBase = q == 0 ? rw->PassCheckpointBase : rw->PassCheckpointBase * m4[q];
    add_reward(a, aBase + aMult * (float)(maxs - steps) / (1.0f * (float)maxs));         /* :418 */
    const float gMult = q == 0 ? rw->TeamPassCheckpointTimeMultiplier : rw->TeamPassCheckpointTimeMultiplier * m4[q];
    const float gBase = q == 0 ? rw->TeamPassCheckpointBase : rw->TeamPassCheckpointBase * m4[q];
    add_group_reward(e, env, team, gBase + gMult * (float)(maxs - steps) / (1.0f * (float)maxs));   /* :424 */
}

/* REC.AddGoalTimingRewards :174-237 */
void hko_rw_goal_timing(hko_env* e, int env)
{
    const hk_config* cfg = &e->cfg;
    const hk_reward_params* rw = &cfg->rw;
    hk_agent_state* ags = &e->ag[(size_t)env * e->A];
    const int A = e->A, maxs = cfg->max_episode_steps;
    if (A == 1) {
        if (ags[0].time_steps != 0)
            add_reward(&ags[0], rw->ReachGoalCheckpointRewardMultplier * (1.0f - (float)ags[0].time_steps * 1.0f / (float)maxs) + rw->ReachGoalCheckpointRewardBase);
        return;
    }
    int ts[HK_MAX_AGENTS];
    for (int i = 0; i < A; i++) ts[i] = ags[i].time_steps == 0 ? 5 * maxs : ags[i].time_steps;   /* :187-191 (a local copy: the field is
                                                                                         rewritten by ResetGame right after) */
    float gt[HK_MAX_AGENTS];
    const float maxReward = 1.0f, minReward = -1.0f;
    for (int i = 0; i < A; i++) {
        const int opponentAgents = cfg->n_other[i], teamAgents = cfg->n_team[i];
        int teamScore = 0, oppScore = 0;
        for (int j = 0; j < opponentAgents; j++) oppScore += ts[cfg->other_agents[i][j]];
        for (int j = 0; j < teamAgents; j++) teamScore += ts[cfg->team_agents[i][j]];
        const float finalCur = (float)ts[i] + (float)teamScore * rw->TeamScoreRewardMultiplier;
        const float finalOpp = (float)oppScore * (1.0f + (float)teamAgents * rw->TeamScoreRewardMultiplier) / ((float)opponentAgents * 1.0f);
        gt[i] = ((finalOpp - finalCur) / (1.0f + (float)teamAgents * rw->TeamScoreRewardMultiplier)) / (float)maxs;
    }
    float groupRewards[HK_MAX_AGENTS];
    const int T = n_teams(e);
    for (int t = 0; t < T; t++) groupRewards[t] = 0.0f;
    for (int i = 0; i < A; i++) {
        if (!cfg->training_agent[i]) continue;                                           /* :217 Mode != Training */
        groupRewards[cfg->team_of[i]] += rw->ReachGoalCheckpointRewardBase +
            rw->ReachGoalCheckpointRewardMultplier * ((gt[i] - minReward) * 1.0f / (maxReward - minReward));   /* :220 */
    }
    for (int t = 0; t < T; t++) add_group_reward(e, env, t, groupRewards[t] / (float)team_size(e, t));          /* :233 */
}

/* REC.ResolveEvent :438-462 for Event.HitWall (victim < 0) / Event.HitOpponent raised by `agent` from CollectObservations */
void hko_rw_hit(hko_env* e, int env, int agent, int victim)
{
    const hk_reward_params* rw = &e->cfg.rw;
    hk_agent_state* ags = &e->ag[(size_t)env * e->A];
    hk_agent_state* a = &ags[agent];
    if (!(a->flags & HK_F_ACTIVE)) return;                                               /* :441 */
    if (victim < 0) { add_reward(a, 1.0f * rw->WallHitPenalty); return; }               /* ApplyHitWallPenalty KA:380 */
    add_reward(a, 1.0f * rw->OpponentHitPenalty);                                        /* :448 */
    if (e->cfg.team_of[agent] == e->cfg.team_of[victim]) {                               /* :452-456 crashing into a team mate */
        add_reward(a, 1.5f * rw->OpponentHitPenalty);
        add_reward(&ags[victim], 1.15f * rw->HitByOpponentPenalty);
    } else add_reward(&ags[victim], 1.0f * rw->HitByOpponentPenalty);                    /* :460 */
}

int hko_get_rewards(hko_env* e, float* reward, float* group_reward)
{
    const size_t na = (size_t)e->E * e->A;
    for (size_t i = 0; i < na; i++) {
        reward[i] = e->ag[i].step_reward; group_reward[i] = e->ag[i].group_reward;
        e->ag[i].step_reward = 0.0f; e->ag[i].group_reward = 0.0f;
    }
    return 0;
}
