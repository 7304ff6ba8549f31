/*
 * hk_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's hot path (ribsthakkar/HierarchicalKarting, Unity C#), each function
 * citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (libhk.so) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" at two third-party boundaries:
 *   - MathNet.Numerics 4.15.0 (LU solve, sparse products): no golden vectors exist in the reference; the LU here
 *     restates MathNet's published JAMA-style algorithm (UserLU: left-looking Doolittle, partial pivoting, first max).
 *   - Unity 2020.3 / PhysX 4.1 (pose integration, contacts, raycasts, triggers): closed source; this oracle DEFINES
 *     a flat 2-D analytic model (see DESIGN.md "Engine restatement").
 * Everything that IS in /root/reference as C# source is restated line by line, quirks Q1-Q14 included.
 * The reference cannot be compiled here (no dotnet/mono); the LQ core is additionally pinned by an independent
 * numpy mirror (oracle/lq_numpy.py) and the golden vectors it emitted (tests/golden/lq_*.json).
 */
#ifndef HK_ORACLE_H
#define HK_ORACLE_H
#include <stdint.h>
#include "../include/hk.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HKO_MAX_PLAYERS 8
#define HKO_MAX_N (4 * HKO_MAX_PLAYERS)
#define HKO_MAX_M (2 * HKO_MAX_PLAYERS)

/* a1: KartLQR.solveFeedbackLQR (AI/LQR/KartLQR.cs:17-128).  Layout as hk_lq_solve_batch (batch = 1).
 * trace (optional, may be NULL): per sweep t = horizon..0, [P (m*n) | alpha (m)] appended. returns 0 / <0 singular */
int hko_lq_solve(int N, const double* A, const double* B, const double* Q, const double* q, const double* R,
                 const double* x0, int horizon, double* u0, double* trace);

/* a2: LinearizedBicycle.getA/getB (AI/LQR/KartLQRDynamics.cs:40-62). initial = (x, z, v, h). A[16], B[8] row-major */
void hko_bicycle_AB(double dt, const double initial[4], double* A, double* B);

/* a3: LQRCheckpointReachAvoidCost.getQMatrix/getQVec/getRMatrix (AI/LQR/KartLQRCosts.cs:57-140).
 * M = number of avoid dynamics (others); n = 4 + 4*M.  avoid_w[2][M] (state x then z; avoidIndices are always the
 * same state index, HKA:1013-1022), opp_target[M][4], opp_w[M][3] (x,z,v).  Q[n*n], qv[n], R[4]. */
void hko_cost_build(int M, const double target[4], const double target_w[4], double control_w, const double* avoid_w,
                    const double* opp_target, const double* opp_w, double* Q, double* qv, double* R);

/* deterministic math wrappers (so tests can bound include/hk_detmath.h against mpmath) */
double hko_sin(double x);
double hko_cos(double x);
double hko_atan2(double y, double x);
double hko_exp(double x);
double hko_log(double x);
float hko_expf_fast(float x);
float hko_swishf(float x);
void hko_sincos_near0(double x, double* s, double* c);   /* the kernels' small-angle path: also bit-identical */
void hko_sincos(double x, double* s, double* c);   /* the kernels' fused pair: must equal (hko_sin, hko_cos) bit for bit */   /* the actor's Swish exp (fp32) */

/* ---- whole-environment oracle (components a4-a11) ---- */
typedef struct hko_env hko_env;
hko_env* hko_create(const hk_config* cfg);
void hko_destroy(hko_env*);
int hko_reset(hko_env*, const int32_t* env_ids, int n, int experiment_num);
int hko_step(hko_env*, int n_ticks);
int hko_set_threads(int n);   /* OpenMP threads hko_step uses (n <= 0: leave as is); returns the current count */
int hko_set_actions(hko_env*, const float* steer, const int32_t* branch);
int hko_get_agent_state(hko_env*, hk_agent_state* out /*[E][A]*/);
int hko_set_agent_state(hko_env*, const hk_agent_state* in);
int hko_get_env_state(hko_env*, hk_env_state* out /*[E]*/);
int hko_set_env_state(hko_env*, const hk_env_state* in);
int hko_get_observations(hko_env*, float* obs);
int hko_get_episode_results(hko_env*, hk_episode_result* out);
int hko_get_mcts_state(hko_env*, hk_mcts_state* out /*[E][A]*/);
int hko_get_rewards(hko_env*, float* reward, float* group_reward);   /* read and reset, as hk_get_rewards */
/* RL policy (hk_oracle_policy.c): same contracts as hk_policy_attach / hk_policy_forward / hk_get_actions */
int hko_policy_attach(hko_env*, const hk_policy_desc* desc, const int32_t* agent_slots, int n_slots, int decision_period);
int hko_policy_forward(hko_env*, int policy, int rows, const float* obs, float* mu, float* logits);
int hko_get_actions(hko_env*, float* steer, int32_t* branch);
/* debug taps for single-step fixtures: last LQ game assembled for (env, ego) */
int hko_debug_last_game(hko_env*, int env, int ego, hk_lq_debug* out);
/* raycast against the track walls (analytic Physics.Raycast vs TrackMask): returns hit distance or -1 */
float hko_raycast_track(hko_env*, float ox, float oz, float dx, float dz, float maxdist);

#ifdef __cplusplus
}
#endif
#endif
