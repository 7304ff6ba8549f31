// hk_env_mcts.h — the MCTS high-level planner on device (contract: include/hk.h "MCTS high-level planner").
//
// Reference: AI/MCTS/KartMCTS.cs (tree search), AI/MCTS/KartDiscreteGame.cs (the discrete racing game),
// HierarchicalKartAgent.planWithMCTS HKA:172-263 (game construction) and HKA:366-402 (how bestStates feed the
// low-level planner).  The reference plans on a background thread under a wall-clock budget; here (see hk.h):
//   * when an agent's planning tick comes (reset, or episodeSteps % 100 == 0) the fused tick kernel SNAPSHOTS what the
//     planner reads (section, lane, lane changes, tire age, section times of every kart) into an MctsReq, queues the
//     (env, agent) pair and keeps ticking;
//   * mcts_search_kernel, launched after every round of the tick kernel, runs the queued searches (iteration budget)
//     and stores the result as hk_mcts_state.pend;
//   * mcts_latency ticks after the request the tick kernel promotes pend to best and, every tick, copies best into the
//     agent's plan and its beliefs about the other karts (HKA:366-402).
// One lane per search in this first version.  Data layout differs from the reference (and from the CPU oracle, which
// follows the C#): ONE running game state per search instead of a state per tree node, nodes are 32-byte records in a
// first-child / next-sibling arena, the legal-move scan of a position is done once and shared by isOver, nextMoves and
// the rollout ordering (the reference recomputes it three times), and what a move costs (time, tire load) comes from
// tables filled once at hk_create by the very functions that restate applyAction / computeTOC.  The draws (Philox) and every float expression
// are the same, so the plans agree with the oracle's bit for bit.
// (included once per lane-group width by hk_env_ga.h: no include guard, namespace hk::HK_GA_NS)
#include "hk_env_device.h"

namespace hk { namespace HK_GA_NS {

constexpr int MC_MAXP = GA;                   // players of a discrete game (<= agents of an env)
constexpr int MC_MAXA = HK_MCTS_MAX_ACTIONS;  // 36: capacity; a class of agents has C.nact = 4 x its velocity buckets (20 at bucket size 2, 36 at 1)
#ifndef HK_MC_SPW
#define HK_MC_SPW 64
#endif
constexpr int MC_SPW = HK_MC_SPW;             // searches per wave (see mcts_search_kernel)

// Root reuse (HKA:66-67,175,265-283): the reference keeps the tree of an agent's last plan (currentRoot) and a replan searches it
// again, up to three searches per tree, unless the kart entered a section in between.  No tree is kept here — the arena belongs
// to the resident lanes of the search kernel, not to the agents.  What IS kept is the recipe: the root position (the kart
// snapshots of the request that started the tree) and the (episode step, iteration budget) of every search it received; the
// draws are counter-based (Philox keyed by episode step / agent / episode), so replaying the earlier searches rebuilds exactly
// the tree the reference would still hold, and the new search continues on it.  The CPU oracle really keeps its trees.
struct MctsReq {
    int episode_steps, epoch, iterations, gen;      // of the latest request.  gen: bumped by every request of this ego (stale queue entries are skipped)
    int last_sec, n_phases, tree_nodes, done_gen;   // last_sec: m_SectionIndex at the last copy of bestStates (mcts_consume); tree_nodes: nodes of the kept
                                                    // tree (M.persist); done_gen: gen of the last request whose search has run (P.mcts_pause)
    int ph_step[HK_MCTS_MAX_ROOT_PHASES];           // the searches the current tree has received, oldest first: episode step of the request ...
    int ph_iter[HK_MCTS_MAX_ROOT_PHASES];           // ... and its iteration budget
    MctsKartSnap k[MC_MAXP];                        // the root position: every kart as the request that started the tree saw it
};
__device__ __forceinline__ MctsReq* mc_reqs(const MctsDev& M) { return static_cast<MctsReq*>(M.req); }
// ---------------------------------------------------------------------------------------------------------------------
// pieces used by the tick kernel

// prepareForReuse (HKA:428-452, KA:212): beliefs and bestStates cleared, sectionTimes[m_SectionIndex] = 0
__device__ inline void mcts_reset_state(hk_mcts_state* m)
{
    const int searches = m->searches;
    __builtin_memset(m, 0, sizeof(hk_mcts_state));
    m->searches = searches;
    m->ready_step = -1;
}


__device__ inline int mcts_tire_age(const EnvParams& P, float final_steer)
{   // HKA:236
    return (int)((P.st.MaxSteer - final_steer) / (P.st.MaxSteer - P.st.MinSteer) * 10000.0f);
}

// The replan decision of planWithMCTS (HKA:175 / :265) for the calling ego: 1 = a new tree, 2 = the existing root again,
// 0 = no search (the tree has had its three).  `t` is always finished here: a search lasts less than the 100 ticks between replans.
__device__ __forceinline__ int mcts_request_kind(const hk_mcts_state* m)
{
    if (!m->root_live) return 1;
    if (m->root_cycles < 3) return m->root_phases < HK_MCTS_MAX_ROOT_PHASES ? 2 : 1;
    return 0;
}

// Every lane of the env's group calls this with the same masks (bit e of req_mask: ego e searches on this tick; bit e of
// new_mask: its search starts a new tree).  Lane i contributes its own kart's snapshot to the record of every ego that starts a
// tree; a requesting ego also writes the header and queues itself.  old_steer: m_FinalStats.Steer before ResetGame — at a reset
// the plans are made agent by agent between the prepareForReuse calls that refresh it (REC:705-710), so ego e sees the stale
// value of every agent behind it in Agents[] order; elsewhere old_steer = final_steer.
__device__ inline void mcts_post_request(const EnvParams& P, const MctsDev& M, int set, int env, int i, uint32_t req_mask, uint32_t new_mask,
                                         int episode_steps, int epoch, int iterations, int ready_step,
                                         int section, int lane, int lane_changes, float final_steer, float old_steer)
{
    if (i >= P.A) return;
    hk_mcts_state* mine = &M.st[(size_t)env * P.A + i];
    MctsKartSnap s;
    s.section = section; s.lane = lane; s.lane_changes = lane_changes;
    const int age_new = mcts_tire_age(P, final_steer), age_old = mcts_tire_age(P, old_steer);
    for (int q = 0; q < HK_MCTS_SECTIME_RING; q++) s.sec_time[q] = mine->sec_time[q];
    for (int e = 0; e < P.A; e++)
        if (new_mask & (1u << e)) { s.tire_age = i > e ? age_old : age_new; mc_reqs(M)[(size_t)env * P.A + e].k[i] = s; }
    const bool requesting = (req_mask & (1u << i)) != 0;
    const int slot = wave_agg_inc(&M.qcnt[set * 2], requesting);          // one atomic per wave, not per ego
    if (requesting) {
        MctsReq* r = &mc_reqs(M)[(size_t)env * P.A + i];
        r->episode_steps = episode_steps; r->epoch = epoch; r->iterations = iterations; r->gen += 1;
        const bool fresh = (new_mask & (1u << i)) != 0;
        const int ph = fresh ? 0 : r->n_phases;                           // (mcts_request_kind keeps ph < HK_MCTS_MAX_ROOT_PHASES)
        r->ph_step[ph] = episode_steps; r->ph_iter[ph] = iterations; r->n_phases = ph + 1;
        mine->root_phases = ph + 1;
        mine->pend_kind = fresh ? 1 : 2;
        mine->searches += 1;
        mine->ready_step = ready_step;
        // an ego can post twice in one launch (a replan tick, then the episode ends and the reset plans again): the queue set
        // holds 2 entries per agent and the entry carries the request generation, so the search kernel skips the stale one
        M.queue[(size_t)set * 2 * P.E * P.A + slot] = (env * P.A + i) | ((r->gen & 0xFF) << 24);
    }
}

// HKA.FixedUpdate :366-402, every tick: promote a finished search, then copy bestStates into the own plan and beliefs
// The reference repeats the copy on every tick; it is idempotent while bestStates and m_SectionIndex stand still (the set of
// own-plan entries it writes only shrinks as the section index grows, nothing else writes beliefs, and a passed checkpoint
// clears an entry outside that set), so here it runs when a search is promoted or the section index changed — two loads per
// tick instead of re-reading the plan and re-storing ~100 B of plan / belief entries per agent and tick.
__device__ inline void mcts_consume(const EnvParams& P, hk_mcts_state* m, hk_agent_state* a, int i, int episode_steps, int section_index,
                                    MctsReq* r)
{
    bool promoted = false;
    if (m->ready_step >= 0 && episode_steps >= m->ready_step) {
        // the search thread ends (HKA:250-253 / :271-273): bestStates, currentRoot and CyclesRootProcessed are written
        m->best = m->pend; m->ready_step = -1; promoted = true;
        if (m->pend_kind == 1) { m->root_live = 1; m->root_cycles = 1; }
        else if (m->pend_kind == 2) { m->root_live = 1; m->root_cycles += 1; }
        m->pend_kind = 0;
    }
    if (!promoted && r->last_sec == section_index) return;
    r->last_sec = section_index;
    const hk_mcts_plan& b = m->best;
    const int L = P.L;
    for (int q = 0; q < b.n_states; q++) {
        const int sec = b.section[q];
        for (int p = 0; p < b.n_players; p++) {
            const int who = b.player_agent[p];
            if (who == i) {
                if (sec > section_index + (section_index == 0 ? 0 : 1)) {
                    a->plan_lane[sec % L] = b.lane[q][p];
                    a->plan_vel[sec % L] = (float)b.vel[q][p];
                }
            } else {
                m->belief_lane[who][sec % L] = b.lane[q][p];
                m->belief_vel[who][sec % L] = b.vel[q][p];
            }
        }
    }
}

// HKA.FixedUpdate :330-402 for one tick (after SolveLQR): request a replan every 100 ticks, consume bestStates.
// `flags` / `section` ... are the calling lane's own kart; every lane of the quad must call this.
__device__ inline void phase_plan(const EnvParams& P, const MctsDev& M, int set, int env, int i, const hk_env_state& es,
                                  uint32_t flags, int section, int lane, int lane_changes, float final_steer, hk_agent_state* arec)
{
    const bool me = i < P.A;
    const bool enabled = me && (flags & HK_F_ENABLED);
    const bool inactive = (es.inactive_mask >> i) & 1u;
    uint32_t req = 0, fresh = 0;
    if (enabled && P.high_mode[i] == HK_HIGH_MCTS && !P.training_agent[i] && es.episode_steps % 100 == 0 && es.episode_steps < P.max_steps &&
        es.episode_steps > 0 && !inactive) {
        const int kind = mcts_request_kind(&M.st[(size_t)env * P.A + i]);
        if (kind) req = 1u << i;
        if (kind == 1) fresh = 1u << i;
    }
    req = (uint32_t)group_or((int)req);
    fresh = (uint32_t)group_or((int)fresh);
    if (req) mcts_post_request(P, M, set, env, i, req, fresh, es.episode_steps, es.episodes_done, P.mcts_iter, es.episode_steps + P.mcts_lat,
                               section, lane, lane_changes, final_steer, final_steer);
    if (enabled && P.high_mode[i] == HK_HIGH_MCTS) mcts_consume(P, &M.st[(size_t)env * P.A + i], arec, i, es.episode_steps, section, &mc_reqs(M)[(size_t)env * P.A + i]);
}

// ---------------------------------------------------------------------------------------------------------------------
// the discrete game (KartDiscreteGame.cs), one running state per search

struct DKart { int section, time, minv, maxv, lane, tire, lchg; };
// The karts are NAMED members, not an array: every access is either mc_k<I>(g) with a compile-time I or a field-wise value
// select (mc_get / mc_set).  With an array the compiler keeps the state in scratch and turns the selects into address
// selects; a scratch round trip costs hundreds of cycles at this kernel's occupancy and there were hundreds per position.
#if HK_GA > 4
#define MC_EACH(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define MC_EACH2(X, S) X(S, 0) X(S, 1) X(S, 2) X(S, 3) X(S, 4) X(S, 5) X(S, 6) X(S, 7)
#else
#define MC_EACH(X) X(0) X(1) X(2) X(3)
#define MC_EACH2(X, S) X(S, 0) X(S, 1) X(S, 2) X(S, 3)
#endif
struct DGame {
    int P;
#define MC_MEMBER(I) DKart k##I; int t##I;        /* kart and team of player I */
    MC_EACH(MC_MEMBER)
#undef MC_MEMBER
    int last, fin;              // lastCompletedSection, finalSection
};
template <int I> __device__ __forceinline__ DKart& mc_k(DGame& g)
{
#define MC_REF(J) if constexpr (I == J) return g.k##J;
    MC_EACH(MC_REF)
#undef MC_REF
    __builtin_unreachable();
}
template <int I> __device__ __forceinline__ const DKart& mc_k(const DGame& g)
{
#define MC_REF(J) if constexpr (I == J) return g.k##J;
    MC_EACH(MC_REF)
#undef MC_REF
    __builtin_unreachable();
}
template <int I> __device__ __forceinline__ int& mc_t(DGame& g)
{
#define MC_REF(J) if constexpr (I == J) return g.t##J;
    MC_EACH(MC_REF)
#undef MC_REF
    __builtin_unreachable();
}
template <int I> __device__ __forceinline__ int mc_t(const DGame& g)
{
#define MC_REF(J) if constexpr (I == J) return g.t##J;
    MC_EACH(MC_REF)
#undef MC_REF
    __builtin_unreachable();
}
// The legal moves of the player who is up next: a bit per canonical action (velocity-major) and the time each adds.
// Every loop over dt[] is fully unrolled, so the array lives in registers (no scratch: a scratch round trip costs
// hundreds of cycles at the one-wave-per-SIMD occupancy this kernel runs at, and there were hundreds per position).
struct MoveEval {
    int n;
    unsigned long long legal;
    int row;                    // table row of the position: (section % L, lane, velocity bucket) of the player who is up next
    DKart cur;                  // that player's kart (mc_get: ~50 mask operations) — the move that follows is made from the same position
};
struct MctsCtx {
    const EnvParams* P;
    const TabView* T;
    int bucket, precision, vmax, nact;
    uint32_t bucket_magic;      // ceil(2^16 / bucket): x / bucket == (x * bucket_magic) >> 16 exactly for 0 <= x < 64 (frac(x / b) <= 1 - 1/64 < 1 - 64 / 65536)
    uint32_t key0, key1, c1, c2, draw;
    const short* dt_tab; const float* load_tab; const float* rad_tab; int nv;      // the move tables (the search kernel's copy in LDS)
    const unsigned long long* mask_tab; const unsigned char* order_tab;      // per table row: feasible actions (dt >= 0), the nact actions in rollout order
    const unsigned char* sec_flags;         // per section (mod L): bit 0 straight, bits 1.. optimal lane (LDS; nullptr: read the track table)
};

__device__ __forceinline__ void mc_draw(MctsCtx& C, uint32_t r[4]) { philox4x32(C.draw++, C.c1, C.c2, 0x4D435453u, C.key0, C.key1, r); }
__device__ __forceinline__ int mc_rand_next(MctsCtx& C, int n)
{
    uint32_t r[4]; mc_draw(C, r);
    return (int)(((unsigned long long)r[0] * (unsigned long long)n) >> 32);
}
__device__ __forceinline__ float mc_normal(MctsCtx& C)
{
    uint32_t r[4]; mc_draw(C, r);
    const float u1 = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f);
    const float u2 = u01(r[1]);
    return sqrtf(-2.0f * hk_logf(u1)) * hk_cosf((2.0f * HK_PI_F) * u2);
}
__device__ __forceinline__ float mc_gauss_bounded(MctsCtx& C, float mean, float sd, float lo, float hi)
{   // KartMCTS.NextGaussian(mean, sd, min, max) :225-240
    float x; int attempts = 0;
    do { x = mean + mc_normal(C) * sd; attempts += 1; } while ((x < lo || x > hi) && attempts < 10);
    if (attempts == 10 && (x < lo || x > hi)) return mean;
    return x;
}

__device__ __forceinline__ void mc_action(const MctsCtx& C, int a, int& minv, int& maxv, int& lane)
{   // KDG:325-338: for (i = 6; i < maxSpeed; i += bucket) for (lane = 1..4)
    const int vi = a >> 2;
    minv = 6 + vi * C.bucket;
    maxv = (minv + C.bucket) < C.vmax ? (minv + C.bucket) : C.vmax;
    lane = (a & 3) + 1;
}

__device__ inline float mc_lane_radius(const SecDev& s, const SecGeo& G, int lane)
{   // DPT.Start :72-88
    const int q = G.left_turn ? (lane - 1) : (4 - lane);
    return s.inside_radius + G.track_width * ((float)q / 4.0f);
}
__device__ inline float mc_radius(const MctsCtx& C, int section, int l0, int l1)
{   // DPT.radiusOfLane :153-158
    const SecDev& s = C.T->sec[section % C.P->L];
    if (s.inside_radius == 0.0f) return 0.0f;
    const SecGeo& G = C.P->sec_geo[section % C.P->L];
    return (mc_lane_radius(s, G, l0) + mc_lane_radius(s, G, l1)) / 2.0f;
}
__device__ inline float mc_distance(const MctsCtx& C, int section, int l0, int l1)
{   // DPT.distanceToTravel :163-175
    const SecDev& s = C.T->sec[section % C.P->L];
    const SecGeo& G = C.P->sec_geo[section % C.P->L];
    const int dl = l0 > l1 ? l0 - l1 : l1 - l0;
    if (s.inside_radius == 0.0f) {
        const float w = ((float)dl * 1.0f / 3.0f) * G.track_width;
        return sqrtf(w * w + G.track_length * G.track_length);
    }
    return (HK_PI_F / 180.0f) * G.turn_degrees * mc_radius(C, section, l0, l1);
}
__device__ inline float mc_tire_load(const MctsCtx& C, int section, float velocity, int l0, int l1)
{   // DPT.tireLoad :180-192
    const SecDev& s = C.T->sec[section % C.P->L];
    if (s.inside_radius == 0.0f) return mc_distance(C, section, l0, l1) * 0.01f;
    const float gs = (velocity * velocity) / mc_radius(C, section, l0, l1);
    return gs * mc_distance(C, section, l0, l1) * 0.01f;
}
__device__ __forceinline__ float mc_max_speed(const MctsCtx& C, float radius, float wear)
{   // ArcadeKart.getMaxSpeedForRadiusAndWear :536-547
    const hk_kart_stats& st = C.P->st;
    if (radius == 0.0f) return st.TopSpeed;
    const float gsw = (1 - wear) * (st.MaxGs - st.MinGs) + st.MinGs;
    float v = sqrtf(gsw * 9.81f * f_abs(radius));
    if (__builtin_isinf(v) || __builtin_isnan(v)) v = st.TopSpeed;
    return v < 0.0001f ? 0.0001f : (v > st.TopSpeed ? st.TopSpeed : v);
}
// section % L for the sections of a discrete game (0 <= section < 2^32 / L, hk_env_device.h mod_L): the run-time `%` was a ~30-instruction
// integer division, and a ply asks five times
__device__ __forceinline__ int mc_mod_L(const MctsCtx& C, int section) { return mod_L(*C.P, section); }
__device__ __forceinline__ bool mc_straight(const MctsCtx& C, int section)
{   // (asked several times per move: from the kernel's LDS copy, not the track table in global memory)
    if (C.sec_flags) return (C.sec_flags[mc_mod_L(C, section)] & 1) != 0;
    return C.T->sec[mc_mod_L(C, section)].inside_radius == 0.0f;
}
__device__ inline float mc_avgv(int minv, int maxv) { return (1.0f * (float)(minv + maxv)) / 2.0f; }

__device__ inline float mc_toc(const MctsCtx& C, float distance, float radius, float wear, float initV, float finalV)
{   // DiscreteKartState.computeTOC KDG:66-117
    const float acc = C.P->st.Acceleration, brk = C.P->st.Braking;
    if (finalV > initV && (finalV * finalV - initV * initV) / (2 * acc) > distance) return -1.0f;
    if (initV > finalV && (initV * initV - finalV * finalV) / (2 * brk) > distance) return -1.0f;
    const float vmax = mc_max_speed(C, radius, wear);
    float t1 = vmax >= initV ? (vmax - initV) / acc : (initV - vmax) / brk;
    float t3 = vmax >= finalV ? (vmax - finalV) / brk : (finalV - vmax) / acc;
    const float x1 = 0.5f * (initV + vmax) * t1;
    const float x3 = 0.5f * (finalV + vmax) * t3;
    const float x2 = distance - x1 - x3;
    const float t2 = x2 / vmax;
    if ((double)t2 > 0.001) return t1 + t2 + t3;
    if (initV <= vmax) {
        const float ms = sqrtf((2 * distance * -brk * acc + -brk * initV * initV - acc * finalV * finalV) / (-acc - brk));
        t1 = (ms - initV) / acc;
        t3 = (ms - finalV) / brk;
        return t1 + t3;
    }
    return -1.0f;
}

// applyAction KDG:122-167 on kart k; returns false when the move is infeasible.  dt = the time it adds.
__device__ inline bool mc_apply(const MctsCtx& C, const DKart& k, int minv, int maxv, int lane, DKart& n, int& dt)
{
    n.section = k.section + 1;
    n.minv = minv; n.maxv = maxv; n.lane = lane;
    const int dl = lane > k.lane ? lane - k.lane : k.lane - lane;
    if (mc_straight(C, k.section) != mc_straight(C, k.section + 1)) n.lchg = 0;
    else if (lane != k.lane) n.lchg = k.lchg + dl;
    else n.lchg = k.lchg;
    const float dist = mc_distance(C, k.section, k.lane, lane);
    const float rad = mc_radius(C, k.section, k.lane, lane);
    // the reference reads newState.tireAge here, which is still 0 (KDG:150)
    dt = (int)(mc_toc(C, dist, rad, 0.0f / 10000.0f, mc_avgv(k.minv, k.maxv), mc_avgv(minv, maxv)) * (float)C.precision);
    n.time = k.time + dt;
    const float load = mc_tire_load(C, k.section, (float)maxv, k.lane, lane);
    n.tire = (int)(((float)k.tire / 10000.0f + load * C.P->st.TireWearFactor) * 10000.0f);
    return dt >= 0;
}

__device__ __forceinline__ int mc_cmp(const DKart& a, const DKart& b)
{   // the Comparison of upNext KDG:186-223
    if (a.section < b.section) return -1;
    if (a.section > b.section) return 1;
    if (a.time < b.time) return -1;
    if (a.time == b.time) {
        const float va = mc_avgv(a.minv, a.maxv), vb = mc_avgv(b.minv, b.maxv);
        if (va > vb) return -1;
        if (va == vb) return 0;
        return 1;
    }
    return 1;
}

// Selection by bit masks, not by ?: — a conditional between two member reads becomes a select of ADDRESSES followed by one
// load (in C++ `c ? lvalue : lvalue` is an lvalue, and LLVM forms the same thing from selects of loads), which pins the
// whole game state in scratch.  All fields are ints, so and / or with all-ones masks is exact.
__device__ __forceinline__ DKart mc_get(const DGame& g, int p)
{
    // player 0 also answers for any p outside 1 .. MC_MAXP - 1
    int rest = 0;
#define MC_REST(I) if (I > 0) rest |= -(int)(p == I);
    MC_EACH(MC_REST)
#undef MC_REST
    const int m0 = ~rest;
    DKart r;
    r.section = g.k0.section & m0; r.time = g.k0.time & m0; r.minv = g.k0.minv & m0; r.maxv = g.k0.maxv & m0;
    r.lane = g.k0.lane & m0; r.tire = g.k0.tire & m0; r.lchg = g.k0.lchg & m0;
#define MC_SELK(I)                                                                           \
    if (I > 0) {                                                                             \
        const int m = -(int)(p == I);                                                        \
        r.section |= mc_k<I>(g).section & m; r.time |= mc_k<I>(g).time & m; r.minv |= mc_k<I>(g).minv & m; r.maxv |= mc_k<I>(g).maxv & m; \
        r.lane |= mc_k<I>(g).lane & m; r.tire |= mc_k<I>(g).tire & m; r.lchg |= mc_k<I>(g).lchg & m; \
    }
    MC_EACH(MC_SELK)
#undef MC_SELK
    return r;
}
__device__ __forceinline__ void mc_set(DGame& g, int p, const DKart& v)
{
#define MC_PUT(I, f) mc_k<I>(g).f = (mc_k<I>(g).f & ~(-(int)(p == I))) | (v.f & -(int)(p == I));
#define MC_PUTK(I) MC_PUT(I, section) MC_PUT(I, time) MC_PUT(I, minv) MC_PUT(I, maxv) MC_PUT(I, lane) MC_PUT(I, tire) MC_PUT(I, lchg)
    MC_EACH(MC_PUTK)
#undef MC_PUTK
#undef MC_PUT
}
__device__ __forceinline__ int mc_team(const DGame& g, int p)
{
    int rest = 0, r = 0;
#define MC_TSEL(I) if (I > 0) { const int m = -(int)(p == I); rest |= m; r |= mc_t<I>(g) & m; }
    MC_EACH(MC_TSEL)
#undef MC_TSEL
    return r | (g.t0 & ~rest);
}

// upNext KDG:183-238: the first kart, in the order List.Sort leaves them, that has not completed section last + 1.
// Every kart of a game state sits at section `last` or `last + 1`, so those still to move sort before the others and the
// answer is their minimum under the comparison; .NET's small-partition sorts (2: one compare-swap, 3: (0,1)(0,2)(1,2),
// 4..16: insertion sort) all leave the lowest-indexed of several equal minima in front, which is the tie rule used here
// (the CPU oracle runs the sorts themselves).
__device__ __forceinline__ int mc_up_next(const DGame& g)
{
    int best = -1;
    DKart bk = g.k0;
#define MC_STEP(I)                                                                           \
    if (I < g.P && mc_k<I>(g).section != g.last + 1) {                                       \
        if (best < 0 || mc_cmp(mc_k<I>(g), bk) < 0) { best = I; bk = mc_k<I>(g); }           \
    }
    MC_EACH(MC_STEP)
#undef MC_STEP
    return best;
}

__device__ __forceinline__ int mc_vb(const MctsCtx& C, int minv)
{   // minv == 0 ? 0 : 1 + (minv - 6) / bucket, with minv in {0} u [6, max speed): the division by the magic number (MctsCtx::bucket_magic)
    return minv == 0 ? 0 : 1 + (int)(((uint32_t)(minv - 6) * C.bucket_magic) >> 16);
}

// nextMoves KDG:318-411 for the player who is up next (np): a bit per canonical action.  An action is legal when (i) it exists and is
// feasible from this row of the move tables (dt >= 0: a precomputed 20-bit mask per row), (ii) on a straight the lane change it needs
// fits the budget (KDG:355-361: a mask over the target lanes, the same for every velocity bucket), (iii) its min_velocity does not
// exceed the lateral-g speed limit of its target lane at the kart's tire age (KDG:362-370: per lane the buckets below the limit).
__device__ __forceinline__ void mc_eval_moves(const MctsCtx& C, const DGame& g, int np, MoveEval& mv)
{
    const DKart cur = mc_get(g, np);
    const int sm = mc_mod_L(C, cur.section);
    const bool str = mc_straight(C, cur.section);
    const float wear = (float)cur.tire / 10000.0f;
    const float* rp = C.rad_tab + (sm * 4 + (cur.lane - 1)) * 4;
    float vl[4];
#pragma unroll
    for (int l = 0; l < 4; l++) vl[l] = mc_max_speed(C, rp[l], wear);                       // lateral-g speed limit per target lane
    const int row = (sm * 4 + (cur.lane - 1)) * (C.nv + 1) + mc_vb(C, cur.minv);
    unsigned long long m = C.mask_tab[row];
    uint32_t lanes = 0;                                                                     // bit l: target lane l + 1 passes (ii) and bucket 0 .. of (iii)
    unsigned long long speed = 0;                                                           // bit 4 vi + l: bucket vi of lane l + 1 passes (iii)
    const int nvb = C.nact >> 2;
#pragma unroll
    for (int l = 0; l < 4; l++) {
        const int dl = (l + 1) > cur.lane ? (l + 1) - cur.lane : cur.lane - (l + 1);
        if (!(str && cur.lchg + dl > C.P->max_lane_changes)) lanes |= 1u << l;
    }
    // (bucket-major with an early exit: a launch holds one gameParams class, so nvb is the same in every lane — 5 buckets at bucket size 2 — and
    // the 9-bucket capacity used to cost 36 predicated compares per position)
#pragma unroll
    for (int vi = 0; vi < MC_MAXA / 4; vi++) {
        if (vi >= nvb) break;
        const float th = (float)(6 + vi * C.bucket);
        uint32_t nib = 0;
#pragma unroll
        for (int l = 0; l < 4; l++) nib |= (!(vl[l] < th) ? 1u : 0u) << l;
        speed |= (unsigned long long)nib << (4 * vi);
    }
    m &= ((unsigned long long)lanes * 0x111111111ull) & speed;
    mv.legal = m; mv.n = __builtin_popcountll(m); mv.row = row; mv.cur = cur;
}

// makeMove KDG:416-443 on the running state (applyAction through the move tables)
// (k: player np's kart as mc_get(g, np) returns it — the caller has it from mc_eval_moves of the same position)
__device__ __forceinline__ void mc_make_move(const MctsCtx& C, DGame& g, int np, int a, DKart k)
{
    const int sm = mc_mod_L(C, k.section);
    int minv, maxv, lane;
    mc_action(C, a, minv, maxv, lane);
    const int dl = lane > k.lane ? lane - k.lane : k.lane - lane;
    int lchg;
    if (mc_straight(C, k.section) != mc_straight(C, k.section + 1)) lchg = 0;
    else if (lane != k.lane) lchg = k.lchg + dl;
    else lchg = k.lchg;
    const int dt = C.dt_tab[((sm * 4 + (k.lane - 1)) * (C.nv + 1) + mc_vb(C, k.minv)) * C.nact + a];
    const float load = C.load_tab[(sm * 4 + (k.lane - 1)) * C.nact + a];
    k.tire = (int)(((float)k.tire / 10000.0f + load * C.P->st.TireWearFactor) * 10000.0f);
    k.time += dt;
    k.section += 1; k.minv = minv; k.maxv = maxv; k.lane = lane; k.lchg = lchg;
    mc_set(g, np, k);
    bool allAhead = true;
#define MC_AHEAD(I) if (I < g.P) allAhead = allAhead && (mc_k<I>(g).section > g.last);
    MC_EACH(MC_AHEAD)
#undef MC_AHEAD
    if (allAhead) g.last += 1;
}
__device__ __forceinline__ void mc_make_move(const MctsCtx& C, DGame& g, int np, int a) { mc_make_move(C, g, np, a, mc_get(g, np)); }

// The move the rollout picks (KM:255-270): the index-th of the legal moves ordered by
// OrderBy(time added).ThenByDescending(max_velocity).ThenBy(|lane change|).ThenBy(sign * lane), a stable sort.  Every key is a
// function of the table row — the time added and the lane change by definition, and sign comes from the optimal lane of section
// `last`, which IS the section of the player who is up next — so the order of all 20 actions is precomputed per row
// (mcts_order_kernel) and the pick walks it, counting the legal ones.
__device__ __forceinline__ int mc_pick_move(const MctsCtx& C, const MoveEval& mv, int index)
{
    const uint32_t* ord = reinterpret_cast<const uint32_t*>(C.order_tab + (size_t)mv.row * C.nact);      // nact bytes = 5 or 9 words (rows are 4-byte aligned)
    int move = 0, seen = 0;
    const int nw = C.nact >> 2;
#pragma unroll
    for (int w = 0; w < MC_MAXA / 4; w++) {
        if (w >= nw || seen > index) break;                  // (nw is uniform: every search of a launch belongs to one gameParams class; the rollout's index is small — |N(0, n / 6)| — so most lanes are done after two words)
        const uint32_t four = ord[w];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int a = (int)((four >> (8 * b)) & 0xFFu);
            const bool ok = (mv.legal >> a) & 1ull;
            move = (ok && seen == index) ? a : move;
            seen += ok ? 1 : 0;
        }
    }
    return move;
}

// isOver KDG:246-313 given the legal-move count of the position.  scores: the reference's List<float> (can exceed P)
__device__ __forceinline__ bool mc_is_over(const MctsCtx& C, const DGame& g, int np, int n_legal, float (&scores)[2 * MC_MAXP])
{
    if (n_legal == 0) {
        // for each player: [0.0 if it is the stuck player or a team mate of it], then 0.5 (the reference has no `else`)
        const int tnp = mc_team(g, np);
        int n = 0;
#pragma unroll
        for (int q = 0; q < 2 * MC_MAXP; q++) scores[q] = 0.0f;
#define MC_NOMOVE(I)                                                                         \
        if (I < g.P) {                                                                       \
            const bool zero_first = (I == np) || (mc_t<I>(g) == tnp);                        \
            const int pos = n + (zero_first ? 1 : 0);   /* entry n = 0.0 (only when zero_first), then 0.5 */ \
            _Pragma("unroll") for (int q = 0; q < 2 * MC_MAXP; q++) if (q == pos) scores[q] = 0.5f;          \
            n = pos + 1;                                                                     \
        }
        MC_EACH(MC_NOMOVE)
#undef MC_NOMOVE
        return true;
    }
    if (g.last != g.fin) return false;
    if (g.P > 1) {
        const float tsm = 0.75f;                                    // RacingEnvController.TeamScoreRewardMultiplier
        float maxScore = (float)C.precision * -1000.0f, minScore = (float)C.precision * 1000.0f;
        float teamScore = 0.0f, opponentScore = 0.0f;
        int teamCount = 0, opponentCount = 0;                       // accumulate over players (not reset, KDG:273-276)
#define MC_RAW(I) float raw##I = 0.0f;
        MC_EACH(MC_RAW)
#undef MC_RAW
#define MC_PAIR(S, O)                                                                        \
        if (O < g.P) {                                                                       \
            if (S == O) teamScore += (float)mc_k<O>(g).time;                                 \
            else if (mc_t<S>(g) == mc_t<O>(g)) { teamScore += (float)mc_k<O>(g).time * tsm; teamCount += 1; } \
            else { opponentScore += (float)mc_k<O>(g).time; opponentCount += 1; }            \
        }
#define MC_SCORE(S)                                                                          \
        if (S < g.P) {                                                                       \
            MC_EACH2(MC_PAIR, S)                                                             \
            const float score = opponentScore * (((float)teamCount * tsm + 1.0f) / ((float)opponentCount * 1.0f)) - teamScore; \
            raw##S = score;                                                                  \
            const bool nn = __builtin_isnan(score);   /* Math.Max / Math.Min propagate NaN */ \
            maxScore = (__builtin_isnan(maxScore) || nn) ? __builtin_nanf("") : (maxScore > score ? maxScore : score); \
            minScore = (__builtin_isnan(minScore) || nn) ? __builtin_nanf("") : (minScore < score ? minScore : score); \
        }
        MC_EACH(MC_SCORE)
#undef MC_SCORE
#undef MC_PAIR
#define MC_NORM(S)                                                                           \
        if (S < g.P) {                                                                       \
            const int si = (raw##S >= -2147483648.0f && raw##S < 2147483648.0f) ? (int)raw##S : (-2147483647 - 1); \
            scores[S] = ((float)si - minScore) * 1.0f / (maxScore - minScore);               \
        }
        MC_EACH(MC_NORM)
#undef MC_NORM
        return true;
    }
    scores[0] = (float)(C.P->max_steps - g.k0.time / C.P->max_steps);
    return true;
}

__device__ __forceinline__ float mc_score_at(const float (&scores)[2 * MC_MAXP], int q)
{
    float r = scores[0];
#pragma unroll
    for (int j = 1; j < 2 * MC_MAXP; j++) if (q == j) r = scores[j];
    return r;
}

__device__ inline void mc_ctx_init(MctsCtx& C, const EnvParams& P, const TabView& T, const MctsDev& M, int ego)
{
    C.P = &P; C.T = &T;
    C.bucket = P.vbucket[ego]; C.precision = P.time_precision[ego];
    C.vmax = (int)P.max_speed;
    C.nact = 0;
    for (int i = 6; i < C.vmax; i += C.bucket) C.nact += 4;
    if (C.nact > MC_MAXA) C.nact = MC_MAXA;                // (hk_create refuses such a list)
    C.bucket_magic = (65536u + (uint32_t)C.bucket - 1u) / (uint32_t)C.bucket;
    C.sec_flags = nullptr; C.mask_tab = nullptr; C.order_tab = nullptr;
    C.dt_tab = nullptr; C.load_tab = M.load_tab; C.rad_tab = M.rad_tab; C.nv = M.nv;       // (the search kernel points these at its LDS copies)
    C.key0 = 0; C.key1 = 0; C.c1 = 0; C.c2 = 0; C.draw = 0;
}

// the host rewrote agent records (hk_set_agent_state): the next tick copies bestStates again whatever the section index is
__global__ __launch_bounds__(256) void mcts_invalidate_copy_kernel(MctsDev M, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) mc_reqs(M)[t].last_sec = -1;
}

// fills the move tables (hk_create): one thread per (section, lane, velocity bucket, action)
__global__ __launch_bounds__(256) void mcts_table_kernel(EnvParams P, MctsDev M, int ego)
{
    const TabView T = tab_view(P, P.tab);
    MctsCtx C;
    mc_ctx_init(C, P, T, M, ego);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = M.nv + 1;
    const int na = M.na;
    if (t >= P.L * 4 * nb * na) return;
    const int a = t % na, vb = (t / na) % nb, l0 = (t / (na * nb)) % 4, sec = t / (na * nb * 4);
    int dt = -1;
    float load = 0.0f;
    if (a < C.nact) {
        DKart cur;
        cur.section = sec; cur.time = 0; cur.lane = l0 + 1; cur.tire = 0; cur.lchg = 0;
        cur.minv = vb == 0 ? 0 : 6 + (vb - 1) * C.bucket;
        cur.maxv = vb == 0 ? (C.bucket < C.vmax ? C.bucket : C.vmax) : ((cur.minv + C.bucket) < C.vmax ? (cur.minv + C.bucket) : C.vmax);
        int minv, maxv, lane;
        mc_action(C, a, minv, maxv, lane);
        DKart nk;
        if (!mc_apply(C, cur, minv, maxv, lane, nk, dt)) dt = dt < 0 ? dt : -1;
        load = mc_tire_load(C, sec, (float)maxv, l0 + 1, lane);
    }
    M.dt_tab[t] = dt;
    if (vb == 0) M.load_tab[(sec * 4 + l0) * na + a] = load;
    if (vb == 0 && a < 4) M.rad_tab[(sec * 4 + l0) * 4 + a] = mc_radius(C, sec, l0 + 1, a + 1);
}

// per table row: which actions are feasible, and the 20 actions in the order the rollout ranks them (see mc_pick_move).  The keys are
// those of KM:255-270 for a kart in this row — time added, max_velocity descending, |lane change|, sign * lane — ties by the
// canonical action index (a stable sort); infeasible actions (dt < 0) sort last and are never legal.
__global__ __launch_bounds__(256) void mcts_order_kernel(EnvParams P, MctsDev M, int ego, int rows)
{
    const TabView T = tab_view(P, P.tab);
    MctsCtx C;
    mc_ctx_init(C, P, T, M, ego);
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int nb = M.nv + 1;
    const int cur_lane = (row / nb) % 4 + 1, sm = row / (nb * 4);
    const int ol = T.sec[sm].optimal_lane;
    const int sign = ol == 1 ? 1 : (ol == 4 ? -1 : 0);                       // DPT:221-231
    const int na = M.na;
    uint32_t key[MC_MAXA];
    unsigned long long mask = 0;
    for (int a = 0; a < na; a++) {
        const int dt = M.dt_tab[(size_t)row * na + a];
        const int lane = (a & 3) + 1, minv = 6 + (a >> 2) * C.bucket;
        const int maxv = (minv + C.bucket) < C.vmax ? (minv + C.bucket) : C.vmax;
        const int dl = lane > cur_lane ? lane - cur_lane : cur_lane - lane;
        key[a] = ((uint32_t)dt << 12) | ((uint32_t)(C.vmax - maxv) << 6) | ((uint32_t)dl << 4) | (uint32_t)(sign * lane + 4);
        if (dt >= 0) mask |= 1ull << a;
    }
    unsigned long long left = (1ull << na) - 1ull;
    for (int r = 0; r < na; r++) {                                           // selection by repeated minimum, the lowest action wins ties
        uint32_t best = 0xFFFFFFFFu;
        int move = -1;
        for (int a = 0; a < na; a++)
            if (((left >> a) & 1ull) && (move < 0 || key[a] < best)) { best = key[a]; move = a; }
        left &= ~(1ull << move);
        M.order_tab[(size_t)row * na + r] = (unsigned char)move;
    }
    M.mask_tab[row] = mask;
}

// ---------------------------------------------------------------------------------------------------------------------
// the tree (KartMCTS.cs)

// upperConfidenceStrategy :167-193 (children in insertion order) with UCTWeight :162-165 (integer division inside the log).
// The reference starts from the child at a random index and lets any child with a strictly larger weight replace it, walking
// the list from the front: the result is the FIRST child holding the maximum if that maximum exceeds the random child's
// weight, else the random child — computed here in ONE walk over the sibling list (each hop is a dependent global load).
__device__ __forceinline__ int mc_ucs(MctsCtx& C, const MNode* nd, int first_child, int n_children, int parent_episodes, MNode& chosen)
{   // (`chosen`: the record of the returned child — the walk has loaded it already; loading it again was one more dependent round trip per level)
    const int index = mc_rand_next(C, n_children);
    int best_first = -1, idx_child = first_child;
    float best = -__builtin_inff(), u_idx = 0.0f;
    MNode best_n, idx_n;
    int c = first_child;
    for (int q = 0; c >= 0; q++) {
        const MNode n = nd[c];
        if (q == 0) { best_n = n; idx_n = n; }
        const float u = (n.totalValue / (float)n.numEpisodes) + sqrtf(1.0f) * hk_logf((float)(parent_episodes / n.numEpisodes));
        if (q == index) { idx_child = c; u_idx = u; idx_n = n; }
        if (u > best) { best = u; best_first = c; best_n = n; }
        c = n.next_sibling;
    }
    if (best > u_idx) { chosen = best_n; return best_first; }
    chosen = idx_n;
    return idx_child;
}

struct alignas(8) MStat { int n; float tv; };                    // MNode::numEpisodes, totalValue as one 8-byte access
static_assert(offsetof(MNode, numEpisodes) % 8 == 0 && offsetof(MNode, totalValue) == offsetof(MNode, numEpisodes) + 4, "MStat overlays MNode");
constexpr int MC_MAXPATH = HK_MCTS_MAX_DEPTH * MC_MAXP + 2;     // nodes on one root-to-leaf path (a move per player and depth level)
constexpr int MC_ROOT_WORDS = (int)(sizeof(DGame) / sizeof(int));
static_assert(sizeof(DGame) % sizeof(int) == 0, "DGame is all ints");
// The root position is needed once per iteration (and for the final read-out) but would sit in ~35 (8 karts: ~70) registers for
// the whole search: it waits in global memory instead, word-major over the arena's lanes (M.roots[word][lane]: coalesced).
__device__ __forceinline__ void mc_root_store(int* rootl, const int stride, const DGame& g)
{
    int w[MC_ROOT_WORDS];
    __builtin_memcpy(w, &g, sizeof(DGame));
#pragma unroll
    for (int f = 0; f < MC_ROOT_WORDS; f++) rootl[(size_t)f * stride] = w[f];
}
__device__ __forceinline__ DGame mc_root_load(const int* rootl, const int stride)
{
    int w[MC_ROOT_WORDS];
#pragma unroll
    for (int f = 0; f < MC_ROOT_WORDS; f++) w[f] = rootl[(size_t)f * stride];
    DGame g;
    __builtin_memcpy(&g, w, sizeof(DGame));
    return g;
}

// one queued search (queue entry q of `set`), its tree in the arena slice `nd`
struct MctsTabs { const short* dt; const float* load; const float* rad; const unsigned char* flags; const unsigned long long* mask; const unsigned char* order; };
__device__ __forceinline__ void mcts_search_one(const EnvParams& P, const MctsDev& M, const TabView& T, const MctsTabs& tabs, const int set, const int q, const int lane0,
                                                unsigned short* path /* LDS, [MC_MAXPATH][64], this lane's column */, unsigned char* pup, const uint32_t cls_agents)
{
    const unsigned ent = (unsigned)M.queue[(size_t)set * 2 * P.E * P.A + q];
    const int pair = (int)(ent & 0xFFFFFFu);
    const int gen0 = mc_reqs(M)[pair].gen;
    if (((unsigned)gen0 & 0xFFu) != (ent >> 24)) return;                           // superseded by a later request of the same ego
    const int env = pair / P.A, ego = pair % P.A;
    if (!((cls_agents >> ego) & 1u)) return;                                       // another gameParams class: its own launch, with its tables
    MctsReq& R = mc_reqs(M)[pair];
    const int slot = M.persist ? pair : lane0;                                    // whose arena slice: the agent's, or the resident lane's
    MNode* nd = M.nodes + (size_t)slot * M.pool_cap;
    int* rootl = M.roots + slot;                                                  // the root position, word-major (stride M.slots)
    hk_mcts_state* mst = &M.st[pair];

    MctsCtx C;
    mc_ctx_init(C, P, T, M, ego);
    C.dt_tab = tabs.dt; C.load_tab = tabs.load; C.rad_tab = tabs.rad; C.sec_flags = tabs.flags; C.mask_tab = tabs.mask; C.order_tab = tabs.order;
    C.key0 = P.mcts_seed; C.key1 = (uint32_t)(P.env_id_base + env) * (uint32_t)P.A + (uint32_t)ego;
    C.c1 = (uint32_t)R.ph_step[0]; C.c2 = (uint32_t)R.epoch; C.draw = 0;

    // planWithMCTS HKA:172-263: the discrete game of the karts within sectionWindow sections of the ego
    DGame root;
    uint32_t agent_of = 0;                                           // player p is agent (agent_of >> 4p) & 15
    root.P = 0;
    int initialSection = R.k[ego].section, furthest = ego;
    for (int i = 0; i < P.A; i++) {
        int d = R.k[i].section - R.k[ego].section; d = d < 0 ? -d : d;
        if (d < P.section_window[ego]) {
            agent_of |= (uint32_t)i << (4 * root.P); root.P++;
            if (R.k[i].section > initialSection) initialSection = R.k[i].section;
            if (initialSection == R.k[i].section) furthest = i;
        }
    }
#define MC_ROOT(I)                                                                           \
    {                                                                                        \
        DKart& k = mc_k<I>(root);                                                            \
        k.section = 0; k.time = 0; k.minv = 0; k.maxv = 0; k.lane = 1; k.tire = 0; k.lchg = 0; mc_t<I>(root) = 0; \
        if (I < root.P) {                                                                    \
            const int ap = (int)((agent_of >> (4 * I)) & 15u);                              \
            const MctsKartSnap& s = R.k[ap];                                                 \
            mc_t<I>(root) = P.team_of[ap];                                                   \
            k.minv = 0;                                /* HKA:211-219: the bucket loop breaks at i = 0 */ \
            k.maxv = C.bucket < C.vmax ? C.bucket : C.vmax;                                  \
            k.section = initialSection;                                                      \
            if (s.section != initialSection)           /* HKA:221-224 */                      \
                k.time = (int)((float)(s.sec_time[s.section & (HK_MCTS_SECTIME_RING - 1)] - R.k[furthest].sec_time[s.section & (HK_MCTS_SECTIME_RING - 1)]) * P.dt * (float)C.precision); \
            k.lane = s.lane; k.tire = s.tire_age; k.lchg = s.lane_changes;                   \
        }                                                                                    \
    }
    MC_EACH(MC_ROOT)
#undef MC_ROOT
    root.last = initialSection;
    root.fin = initialSection + P.depth[ego];

    const int rootP = root.P;
    const int n_phases = R.n_phases;
    // a re-searched root (n_phases > 1): with per-agent slices the tree is still there and the new search continues on it; without,
    // the earlier searches are replayed first (each with its own draw stream) — see MctsReq
    const bool kept = M.persist && n_phases > 1;
    int n_nodes = 1;
    if (kept) n_nodes = R.tree_nodes;
    else {
        mc_root_store(rootl, M.slots, root);
        nd[0].parent = -1; nd[0].first_child = -1; nd[0].last_child = -1; nd[0].next_sibling = -1;
        nd[0].numEpisodes = 0; nd[0].totalValue = 0.0f; nd[0].action = 0; nd[0].n_children = 0; nd[0].pad = 0; nd[0].pad2 = 0;
        nd[0].upnext = (unsigned char)mc_up_next(root);
    }

    // The tree lives in global memory and a search is one lane: every dependent load is a round trip nothing hides.  So the
    // node the search stands on is kept in registers (`cur`), the path is remembered in LDS instead of being re-walked through
    // the parent links, and nodes created in this iteration are only ever stored: once a rollout leaves the existing tree
    // every deeper node is new, so the rest of the rollout and most of the back-propagation load nothing at all.
    float scores[2 * MC_MAXP];
    MoveEval mv;
    // every search the tree has received, oldest first (a re-searched root: the earlier ones are replayed, see MctsReq); each has
    // its own draw stream, and the last one's continues into the read-out below
    int ph = kept ? n_phases - 1 : 0, it_left = R.ph_iter[ph];
    C.c1 = (uint32_t)R.ph_step[ph];
    while (true) {
        if (it_left == 0) {
            if (++ph >= n_phases) break;
            C.c1 = (uint32_t)R.ph_step[ph]; C.draw = 0;
            it_left = R.ph_iter[ph];
            continue;
        }
        it_left--;
        // findLeaf :195-202 on a running copy of the root state
        DGame g = mc_root_load(rootl, M.slots);
        int node = 0, depth = 0;
        MNode cur = nd[0];
        path[0] = 0; pup[0] = cur.upnext;
        int np = cur.upnext;
        mc_eval_moves(C, g, np, mv);
        while (cur.n_children > 0 && cur.n_children == mv.n) {
            { MNode pick; node = mc_ucs(C, nd, cur.first_child, cur.n_children, cur.numEpisodes, pick); cur = pick; }
            mc_make_move(C, g, np, cur.action, mv.cur);
            np = cur.upnext;
            depth++;
            path[depth * 64] = (unsigned short)node; pup[depth * 64] = cur.upnext;
            mc_eval_moves(C, g, np, mv);
        }
        // simulate :242-283 (mv / np describe the position of `node`)
        bool out_of_nodes = false;
        int first_new = MC_MAXPATH;                    // depth of the first node created in this iteration
        while (true) {
            if (mc_is_over(C, g, np, mv.n, scores)) break;
            int index;
            if (mv.n > 2) index = (int)__builtin_rintf(f_abs(mc_gauss_bounded(C, 0.0f, (float)mv.n / 6.0f, -(float)mv.n + 1.0f, (float)mv.n - 1.0f)));
            else index = mc_rand_next(C, mv.n);
            const int move = mc_pick_move(C, mv, index);
            int c = -1;
            MNode nxt;
            if (cur.n_children > 0) {
                for (int ch = cur.first_child; ch >= 0;) {
                    const MNode n = nd[ch];
                    if (n.action == move) { c = ch; nxt = n; break; }
                    ch = n.next_sibling;
                }
            }
            mc_make_move(C, g, np, move, mv.cur);
            const int np2 = mc_up_next(g);
            if (c < 0) {
                if (n_nodes >= M.pool_cap || depth + 1 >= MC_MAXPATH) { out_of_nodes = true; break; }
                c = n_nodes++;
                nxt.parent = node; nxt.first_child = -1; nxt.last_child = -1; nxt.next_sibling = -1;
                nxt.numEpisodes = 0; nxt.totalValue = 0.0f; nxt.action = (unsigned char)move; nxt.upnext = (unsigned char)np2;
                nxt.n_children = 0; nxt.pad = 0; nxt.pad2 = 0;
                nd[c] = nxt;
                if (cur.last_child >= 0) nd[cur.last_child].next_sibling = c; else cur.first_child = c;
                cur.last_child = c;
                cur.n_children += 1;
                nd[node] = cur;                        // (numEpisodes / totalValue of `cur` are still those of the tree)
                if (first_new == MC_MAXPATH) first_new = depth + 1;
            }
            node = c;
            cur = nxt;
            np = np2;
            depth++;
            path[depth * 64] = (unsigned short)node; pup[depth * 64] = (unsigned char)np2;
            mc_eval_moves(C, g, np, mv);
        }
        if (out_of_nodes) break;                       // (cannot happen: hk_create sizes the arena for the worst case of every search a tree can receive)
        // backpropagate :285-293 along the remembered path, leaf to root; a node created in this iteration holds 0 / 0.0f
        // (the nodes of a path are distinct, so the order of the updates is free: the tree's old nodes — a load, an add, a store each — go four at a
        // time, their loads in flight together, instead of one dependent round trip per node)
        {
            int d = depth;
            for (; d >= first_new; d--) {
                const int b = path[d * 64];
                MStat w; w.n = 1; w.tv = 0.0f + mc_score_at(scores, pup[d * 64]);
                *reinterpret_cast<MStat*>(&nd[b].numEpisodes) = w;
            }
            for (; d >= 3; d -= 4) {
                const int b0 = path[d * 64], b1 = path[(d - 1) * 64], b2 = path[(d - 2) * 64], b3 = path[(d - 3) * 64];
                MStat w0 = *reinterpret_cast<const MStat*>(&nd[b0].numEpisodes), w1 = *reinterpret_cast<const MStat*>(&nd[b1].numEpisodes);
                MStat w2 = *reinterpret_cast<const MStat*>(&nd[b2].numEpisodes), w3 = *reinterpret_cast<const MStat*>(&nd[b3].numEpisodes);
                w0.tv += mc_score_at(scores, pup[d * 64]); w0.n += 1;
                w1.tv += mc_score_at(scores, pup[(d - 1) * 64]); w1.n += 1;
                w2.tv += mc_score_at(scores, pup[(d - 2) * 64]); w2.n += 1;
                w3.tv += mc_score_at(scores, pup[(d - 3) * 64]); w3.n += 1;
                *reinterpret_cast<MStat*>(&nd[b0].numEpisodes) = w0; *reinterpret_cast<MStat*>(&nd[b1].numEpisodes) = w1;
                *reinterpret_cast<MStat*>(&nd[b2].numEpisodes) = w2; *reinterpret_cast<MStat*>(&nd[b3].numEpisodes) = w3;
            }
            for (; d >= 0; d--) {
                const int b = path[d * 64];
                MStat w = *reinterpret_cast<const MStat*>(&nd[b].numEpisodes);
                w.tv += mc_score_at(scores, pup[d * 64]); w.n += 1;
                *reinterpret_cast<MStat*>(&nd[b].numEpisodes) = w;
            }
        }
    }

    // getBestStatesSequence :108-123
    hk_mcts_plan plan;
    __builtin_memset(&plan, 0, sizeof(plan));
    plan.n_players = rootP;
    for (int p = 0; p < rootP; p++) plan.player_agent[p] = (uint8_t)((agent_of >> (4 * p)) & 15u);
    {
        DGame g = mc_root_load(rootl, M.slots);
        int node = 0;
        while (nd[node].n_children > 0) {
            const MNode rec = nd[node];
            const int np = rec.upnext;
            MNode pick;
            node = mc_ucs(C, nd, rec.first_child, rec.n_children, rec.numEpisodes, pick);
            mc_make_move(C, g, np, pick.action);
            bool all_at = true;
#define MC_AT(I) if (I < g.P) all_at = all_at && (mc_k<I>(g).section == g.last);
            MC_EACH(MC_AT)
#undef MC_AT
            if (all_at && plan.n_states < HK_MCTS_MAX_DEPTH) {
                const int s = plan.n_states++;
                plan.section[s] = g.last;
#define MC_OUT(I) if (I < g.P) { plan.lane[s][I] = (uint8_t)mc_k<I>(g).lane; plan.vel[s][I] = (uint8_t)mc_k<I>(g).maxv; }
                MC_EACH(MC_OUT)
#undef MC_OUT
            }
        }
    }
    // A search that runs on the side stream BESIDE tick launches (hk_api.hip: the replan chunk of a planner + actor handle, pause = 0) can be overtaken by
    // its own env: the race ends or times out, the reset posts a new request for this ego (gen + 1) and rewrites the root snapshot this search has been
    // reading.  Whatever it computed from a torn snapshot is dropped here — every field of the record is a valid value of one of the two requests, so the
    // search itself stays in range — and the new request's own search (queued by the reset, launched behind this one: they share the arena) supplies the plan.
    if (__hip_atomic_load(&R.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen0) return;
    mst->pend = plan;
    if (M.persist) R.tree_nodes = n_nodes;
    // the plan first, device-wide, then "done": a tick launch may run BESIDE this kernel (hk_api.hip: the search of a replan overlaps the ticks up
    // to the plan's deadline) and reads the plan only after it has read done_gen == gen
    __threadfence();
    R.done_gen = R.gen;
}

// P.mcts_pause: does this lane's agent wait for a search that has been requested but has not run yet?
__device__ __forceinline__ bool mcts_search_outstanding(const EnvParams& P, const MctsDev& M, int env, int i)
{
    if (i >= P.A || P.high_mode[i] != HK_HIGH_MCTS) return false;
    const volatile MctsReq& R = mc_reqs(M)[(size_t)env * P.A + i];      // (volatile: the search kernel may write done_gen while this launch runs)
    return R.gen != R.done_gen;
}
// P.mcts_pause: the first episode step this lane's agent must not BEGIN while its requested search has not run — the step that promotes the
// search's result (hk_mcts_state.ready_step; mcts_consume) — or INT_MAX when nothing is outstanding.  Until then the env keeps stepping: the
// reference's search thread runs beside FixedUpdate for T seconds (HKA:172-284), and here the search kernel runs beside the tick launches.
__device__ __forceinline__ int mcts_hold_step(const EnvParams& P, const MctsDev& M, int env, int i)
{
    if (!mcts_search_outstanding(P, M, env, i)) return 0x7FFFFFFF;
    const int rs = M.st[(size_t)env * P.A + i].ready_step;
    return rs >= 0 ? rs : 0x7FFFFFFF;
}

// The search kernel: a fixed grid of waves (<= MCTS_ARENA_WAVES) walks the queue with a grid stride, so every wave ends when the queue
// is exhausted.  A workgroup is 4 or 8 waves that share ONE copy of the move tables in LDS (dt as int16: hk_create checks the range):
// a position evaluation reads 24 table entries per lane, and from L2 those loads kept the wave waiting 44 % of the time
// (SQ_WAIT_ANY, round 2).  One workgroup per CU: 4 waves = one per SIMD while the batch is small (<= 1 024 waves), 8 = two per SIMD
// beyond that (the kernel is held to 256 registers) — two resident waves hide each other's tree loads: 65 536 x 4 searches of 64
// iterations 93 -> 59 ms before the tables moved.
#define HK_MC_BOUNDS __launch_bounds__(512)
constexpr int MC_PATH_BYTES = MC_MAXPATH * 64 * 3;          // per wave: node indices (16 bit) + up-next players (8 bit) of one root-to-leaf path
__host__ __device__ inline size_t mcts_table_lds_bytes(int ntab, int L, int na, int tier)
{   // dt (int16) | rad | section flags | [tier bit 0] row masks | [bit 1] load | [bit 2] row orders
    size_t b = (((size_t)ntab * sizeof(short) + 15) & ~(size_t)15) + (size_t)L * 4 * 4 * sizeof(float) + HK_MAX_SECTIONS;
    if (tier & 1) b += (size_t)(ntab / na) * sizeof(unsigned long long);
    if (tier & 2) b += (size_t)L * 4 * na * sizeof(float);
    if (tier & 4) b += ((size_t)ntab + 15) & ~(size_t)15;
    return (b + 15) & ~(size_t)15;
}
inline size_t mcts_search_lds_bytes(int ntab, int L, int na, int tier, int waves) { return mcts_table_lds_bytes(ntab, L, na, tier) + (size_t)waves * MC_PATH_BYTES; }
// ALL_LDS: every table rides in LDS (tier 7).  false (tier 0: a long track at velocityBucketSize 1, whose tables exceed the CU's LDS): only
// the time table, the radii and the section flags do; masks, loads and orders are read from global memory.  A COMPILE-TIME choice: with a
// run-time one the table pointers are generic and every LDS read becomes a flat_load (see tab_stage, hk_env_device.h).
template <bool ALL_LDS>
__global__ HK_MC_BOUNDS void mcts_search_kernel(EnvParams P, MctsDev M, int set, int ntab, uint32_t cls_agents)
{
    HK_DYN_SHARED(mc_smem);
    const int count = M.qcnt[set * 2];
    if (count == 0) return;                                  // (uniform: nothing queued, no table copy either)
    constexpr int tier = ALL_LDS ? 7 : 0;
    const int na = M.na, rows = ntab / na;
    short* dt_s = reinterpret_cast<short*>(mc_smem);
    float* rad_s = reinterpret_cast<float*>(mc_smem + (((size_t)ntab * sizeof(short) + 15) & ~(size_t)15));
    unsigned char* flags_s = reinterpret_cast<unsigned char*>(rad_s + P.L * 4 * 4);          // [HK_MAX_SECTIONS]
    unsigned char* nextp = flags_s + HK_MAX_SECTIONS;                                         // (8-byte aligned: 16 + L * 64 + 64)
    unsigned long long* mask_s = reinterpret_cast<unsigned long long*>(nextp);                // [rows]
    if (tier & 1) nextp += (size_t)rows * sizeof(unsigned long long);
    float* load_s = reinterpret_cast<float*>(nextp);                                          // [L][4][na]
    if (tier & 2) nextp += (size_t)P.L * 4 * na * sizeof(float);
    unsigned char* order_s = nextp;                                                           // [rows][na] = ntab bytes
    if (tier & 4) nextp += ((size_t)ntab + 15) & ~(size_t)15;
    unsigned char* paths = mc_smem + mcts_table_lds_bytes(ntab, P.L, na, tier);
    for (int k = threadIdx.x; k < ntab; k += blockDim.x) { const int v = M.dt_tab[k]; dt_s[k] = (short)(v < 0 ? -1 : v); }     // < 0: infeasible (only the sign is read)
    for (int k = threadIdx.x; k < P.L * 4 * 4; k += blockDim.x) rad_s[k] = M.rad_tab[k];
    if (tier & 1) for (int k = threadIdx.x; k < rows; k += blockDim.x) mask_s[k] = M.mask_tab[k];
    if (tier & 2) for (int k = threadIdx.x; k < P.L * 4 * na; k += blockDim.x) load_s[k] = M.load_tab[k];
    if (tier & 4) for (int k = threadIdx.x; k < ntab / 4; k += blockDim.x) reinterpret_cast<uint32_t*>(order_s)[k] = reinterpret_cast<const uint32_t*>(M.order_tab)[k];
    {
        const TabView Tg = tab_view(P, P.tab);
        for (int k = threadIdx.x; k < P.L; k += blockDim.x) flags_s[k] = (unsigned char)((Tg.sec[k].inside_radius == 0.0f ? 1 : 0) | (Tg.sec[k].optimal_lane << 1));
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = blockDim.x >> 6;
    const int lane0 = (blockIdx.x * waves + wave) * 64 + lane;
    if (lane0 >= M.grid_lanes) return;
    const TabView T = tab_view(P, P.tab);
    unsigned short* path = reinterpret_cast<unsigned short*>(paths + (size_t)wave * MC_PATH_BYTES);       // [MC_MAXPATH][64]: node indices fit 16 bits
    unsigned char* pup = reinterpret_cast<unsigned char*>(path + MC_MAXPATH * 64);                         //   (hk_create refuses pools beyond 65 535 nodes)
    MctsTabs tabs;
    if constexpr (ALL_LDS) tabs = MctsTabs{dt_s, load_s, rad_s, flags_s, mask_s, order_s};
    else tabs = MctsTabs{dt_s, M.load_tab, rad_s, flags_s, M.mask_tab, M.order_tab};
    for (int q = lane0; q < count; q += gridDim.x * waves * 64) mcts_search_one(P, M, T, tabs, set, q, lane0, path + lane, pup + lane, cls_agents);
}

#undef MC_EACH
#undef MC_EACH2
#undef HK_MC_BOUNDS

} }  // namespace hk::HK_GA_NS
