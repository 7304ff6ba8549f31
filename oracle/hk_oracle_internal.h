/*
 * hk_oracle_internal.h — CPU ORACLE (test infrastructure): the environment record shared by hk_oracle_env.c and
 * hk_oracle_policy.c.  Not part of any public interface.
 */
#ifndef HK_ORACLE_INTERNAL_H
#define HK_ORACLE_INTERNAL_H
#include <stdint.h>
#include "hk_oracle.h"

struct hko_policy;

typedef struct {
    float fx, fz;     /* forward = (sin yaw, cos yaw) */
    float yaw_rad;
} sec_pre;

/* the engine restatement's derived constants (hk_oracle_env.c hko_engine_derive; the kernels: EngDerived, formed by the same expressions) */
typedef struct { float inv_ext, inv_span, ext, asy, a3, a2, a1, b3, b2, b0, flat; } hko_curve;
typedef struct {
    float inv_m, inv_i, inv_mw;
    float side_kf, side_kr, fwd_kf, fwd_kr;     /* stiffness * axle load * dt: the impulse of friction coefficient 1 in one tick */
    float inv_damp_f, inv_damp_r;               /* 1 / (1 + dt * wheelDampingRate / (m_wheel r^2 / 2)) */
    float jden_r, jlden_r;                      /* rear axle (not steered: levers zr and 0): 1 / (1/m + zr^2 / I),  1 / (1/m_wheel + 1/m) */
    hko_curve side, fwd;
} hko_engine;

struct hko_env {
    hk_config cfg;
    hko_engine eng;
    hk_section* sec;
    sec_pre* sp;
    hk_wall_seg* walls;
    int E, A, L, NW;
    hk_agent_state* ag;    /* [E][A] */
    hk_env_state* es;      /* [E] */
    hk_episode_result* res;/* [E][A] */
    hk_lq_debug* dbg;      /* [E][A] */
    float* act_steer;      /* [E][A] */
    int32_t* act_branch;
    float init_acc_ang_v;
    float max_speed;       /* ArcadeKart.GetMaxSpeed AK:210 */
    int nperm;
    int* perms;            /* [A!][A] lexicographic (REC:137-145,166) */
    float ray_agent_r;     /* stadium radius of a kart capsule sliced at the sensor height */
    float sens_c[HK_NUM_SENSORS], sens_s[HK_NUM_SENSORS];   /* cos / sin of the sensors' local yaw */
    /* MCTS planner (hk_oracle_mcts.c): [E][A], NULL when no agent is HighMode MCTS */
    hk_mcts_state* mcts;
    struct hko_tree* trees; /* [E][A] the persisted search trees (currentRoot, HKA:66) */
    /* reward shaping (hk_oracle_reward.c): minSectionTimes / agentsPastSection, [E][A teams][laps * L + 2]; NULL when off */
    int32_t* sec_min_time;
    uint8_t* sec_count;
    /* RL policies (hk_oracle_policy.c) */
    int n_policies;
    struct hko_policy* policy[HK_MAX_POLICIES];
    int decision_period;
    int64_t academy_step;  /* ticks stepped since creation (Academy.StepCount) */
    float* obs_scratch;    /* [E][A][obs_dim] */
};

/* ------------------------------------------------------------------ Philox-4x32-10 (synthetic start jitter) */
static inline void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }


/* hk_oracle_mcts.c */
struct hko_tree;
size_t hko_mcts_tree_bytes(void);
void hko_mcts_search(hko_env* e, int env, int ego, int iterations, int reuse, const float* steer_seen, hk_mcts_plan* plan);
void hko_mcts_request(hko_env* e, int env, int ego, int iterations, int latency, const float* steer_seen);
void hko_mcts_promote(hk_mcts_state* m);
void hko_mcts_consume(hko_env* e, int env, int i);
void hko_mcts_tree_drop(hko_env* e, int env, int ego);
void hko_mcts_trees_free(hko_env* e);
void hko_mcts_backfill_section_times(hko_env* e, int env);

/* hk_oracle_reward.c */
int hko_rw_table_len(const hko_env* e);
void hko_rw_reset_env(hko_env* e, int env);
void hko_rw_academy(hko_env* e, int env);
void hko_rw_not_at_goal(hko_env* e, hk_agent_state* a);
void hko_rw_dividers(hko_env* e, int agent, const hk_agent_state* a, int index, int lane, float* lane_div, float* vel_div);
void hko_rw_swerve(hko_env* e, hk_agent_state* a);
void hko_rw_reverse(hko_env* e, hk_agent_state* a, int old_section, int index);
void hko_rw_section(hko_env* e, int env, int ai, float lane_div, float vel_div);
void hko_rw_goal_timing(hko_env* e, int env);
void hko_rw_hit(hko_env* e, int env, int agent, int victim);

/* hk_oracle_policy.c */
void hko_policy_decide(hko_env* e);           /* observe -> stack -> infer -> latch actions, if this tick is a decision tick */
void hko_policy_invalidate(hko_env* e, int env);
void hko_policy_free(hko_env* e);
void hko_observe_env(hko_env* e, int env, float* obs /*[A][dim]*/);
int hko_obs_dim(const hko_env* e);

#endif
