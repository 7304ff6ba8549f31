// hk_env_kernels.h — host side of the batched kart environment: device tables, candidate wall lists, the round scheduler.
// (API translation unit only; the kernels live in hk_ga4.hip / hk_ga8.hip behind hk_env_host.h's GaOps table.)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "hk_env_host.h"
#include "hk_env_params.h"

namespace hk {

// forward to the kernels of the handle's lane-group width (hk_env_host.h: GaOps)
#define HK_GA_CALL(d, call) (ga_ops(d).call)

namespace detail {

template <class T>
int upload(T** dst, const std::vector<T>& v, std::string& err)
{
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    hipError_t e = hipMalloc((void**)dst, bytes);
    if (e != hipSuccess) { err = std::string("hipMalloc: ") + hipGetErrorString(e); return HK_ERR_HIP; }
    if (!v.empty()) {
        e = hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) { err = std::string("hipMemcpy: ") + hipGetErrorString(e); return HK_ERR_HIP; }
    }
    return HK_OK;
}

}  // namespace detail

__global__ void status_and_kernel(int* status, int mask) { atomicAnd(status, mask); }

// point d.mcts (the kernel argument) at the move tables of gameParams class c
inline void mcts_use_class(EnvDevice& d, int c)
{
    const EnvDevice::MctsClass& K = d.mcls[c];
    d.mcts.dt_tab = K.dt_tab; d.mcts.load_tab = K.load_tab; d.mcts.rad_tab = K.rad_tab; d.mcts.mask_tab = K.mask_tab; d.mcts.order_tab = K.order_tab;
    d.mcts.nv = K.nv; d.mcts.na = K.na; d.mcts.lds_tier = K.lds_tier; d.mcts.ntab = K.ntab;
}

inline void env_destroy(EnvDevice& d)
{
    void* ptrs[] = {d.agents, d.envs, d.results, d.lq_debug, d.obs, d.act_steer, d.act_branch, d.reward_out, d.status, d.game_stats, d.games, d.queue_cnt, d.queue, d.env_ids, d.tab, d.perms,
                    d.rw.sec_time, d.rw.sec_cnt, d.rw.hit_code, d.mcts.st, d.mcts.req, d.mcts.qcnt, d.mcts.queue, d.mcts.nodes, d.mcts.roots, d.sec_geo, d.perm, d.perm_counts,
                    d.hot, d.hot_alt, d.envs_alt, d.envs_stage, d.slot_of, d.perm_alt};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int c = 0; c < d.n_mcls; c++) { void* t[] = {d.mcls[c].dt_tab, d.mcls[c].load_tab, d.mcls[c].rad_tab, d.mcls[c].mask_tab, d.mcls[c].order_tab}; for (void* p : t) if (p) (void)hipFree(p); }
    d = EnvDevice{};
}

inline int env_create(hk_config& cfg, std::vector<hk_section>& sections, std::vector<hk_wall_seg>& walls, EnvDevice& d,
                      hipStream_t stream, std::string& err)
{
    const int A = cfg.num_agents, L = cfg.num_sections, E = cfg.num_envs;
    EnvParams& P = d.P;
    std::vector<int> perms;
    {
        std::vector<unsigned char> pk;
        int rc0;
        if ((rc0 = env_build_params(cfg, sections, walls, P, pk, perms, err))) return rc0;
        if ((rc0 = detail::upload(&d.tab, pk, err))) return rc0;
        P.tab = d.tab;
        // Oval: 29 KB per block, Complex 47.5 KB (round 5: the tight Trigger candidates are two bytes per cell), all of it in LDS; a longer track keeps
        // the candidates in global memory; beyond that the global-memory instantiation (HK_TAB_GLOBAL=1 forces it, for tests)
        const int stage = P.tab_bytes <= 48 * 1024 ? P.tab_bytes : P.o_tmask2;
        d.tab_lds = (stage <= 48 * 1024 && !std::getenv("HK_TAB_GLOBAL")) ? stage : 0;
        P.tab_stage_bytes = d.tab_lds ? d.tab_lds : P.tab_bytes;
    }
    int rc;
    if ((rc = detail::upload(&d.perms, perms, err))) return rc;
    P.perms = d.perms;
    const size_t na = (size_t)E * A;
    const int obs_dim = HK_NUM_SENSORS + cfg.section_horizon * 5 + 8 + 12 * (A - 1);
    hipError_t e;
#define HK_ALLOC(ptr, bytes)                                                                              \
    if ((e = hipMalloc((void**)&(ptr), (bytes))) != hipSuccess) { err = std::string("hipMalloc: ") + hipGetErrorString(e); return HK_ERR_HIP; } \
    if ((e = hipMemsetAsync((ptr), 0, (bytes), stream)) != hipSuccess) { err = std::string("hipMemset: ") + hipGetErrorString(e); return HK_ERR_HIP; }
    HK_ALLOC(d.agents, na * sizeof(hk_agent_state));
    HK_ALLOC(d.envs, (size_t)E * sizeof(hk_env_state));
    HK_ALLOC(d.envs_alt, (size_t)E * sizeof(hk_env_state));
    HK_ALLOC(d.envs_stage, (size_t)E * sizeof(hk_env_state));
    HK_ALLOC(d.hot, HK_GA_CALL(d, hot_tile_words(E)) * sizeof(uint32_t));
    HK_ALLOC(d.hot_alt, HK_GA_CALL(d, hot_tile_words(E)) * sizeof(uint32_t));
    HK_ALLOC(d.slot_of, (size_t)E * sizeof(int));
    HK_ALLOC(d.perm_alt, (size_t)E * sizeof(int));
    HK_ALLOC(d.results, na * sizeof(hk_episode_result));
    HK_ALLOC(d.lq_debug, na * sizeof(hk_lq_debug));
    HK_ALLOC(d.obs, na * obs_dim * sizeof(float));
    HK_ALLOC(d.act_steer, na * sizeof(float));
    HK_ALLOC(d.act_branch, na * sizeof(int32_t));
    HK_ALLOC(d.reward_out, 2 * na * sizeof(float));
    HK_ALLOC(d.status, 4 * sizeof(int));
    HK_ALLOC(d.game_stats, GAME_STATS_N * sizeof(unsigned long long));     // (hk_env_device.h: GAME_STATS_N)
    HK_ALLOC(d.games, na * HK_GA_CALL(d, game_doubles_per_ego()) * sizeof(double));
    HK_ALLOC(d.queue_cnt, 2 * SPLIT_WAYS_MAX * 16 * sizeof(int));
    HK_ALLOC(d.perm, (size_t)E * sizeof(int));
    HK_ALLOC(d.perm_counts, 32 * sizeof(int));
    HK_ALLOC(d.queue, 2 * SPLIT_WAYS_MAX * HK_GA_CALL(d, queue_ints_per_set(na)) * sizeof(int));
    if (cfg.rewards) {
        d.rw.S = cfg.laps * L + 2;
        const size_t n = na * (size_t)d.rw.S;
        HK_ALLOC(d.rw.sec_time, n * sizeof(int));
        HK_ALLOC(d.rw.sec_cnt, n * sizeof(int));
        HK_ALLOC(d.rw.hit_code, na * HK_NUM_SENSORS);
        if ((e = hipMemsetAsync(d.rw.sec_time, 0xFF, n * sizeof(int), stream)) != hipSuccess) { err = "hipMemset"; return HK_ERR_HIP; }
    }
    if (P.any_mcts) {
        if (na >= (1u << 24)) { err = "hk_create: MCTS planner queues address at most 2^24 agents per device"; return HK_ERR_UNSUPPORTED; }
        std::vector<SecGeo> geo(L);
        for (int i = 0; i < L; i++) geo[i] = SecGeo{sections[i].track_width, sections[i].track_length, sections[i].turn_degrees, sections[i].left_turn};
        if ((rc = detail::upload(&d.sec_geo, geo, err))) return rc;
        P.sec_geo = d.sec_geo;
        int max_depth = 1;
        for (int i = 0; i < A; i++) if (cfg.high_mode[i] == HK_HIGH_MCTS) max_depth = std::max(max_depth, cfg.tree_search_depth[i]);
        // every search a tree can receive (root reuse: the first one plus HK_MCTS_MAX_ROOT_PHASES - 1 replans), each adding at most
        // (depth x players) + 1 nodes per iteration
        d.mcts.pool_cap = 1 + (std::max(cfg.mcts_iterations, cfg.mcts_initial_iterations) + (HK_MCTS_MAX_ROOT_PHASES - 1) * cfg.mcts_iterations) * (max_depth * A + 1);
        {   // the search kernel is a fixed grid walking the queue, at most MCTS_ARENA_WAVES waves (2 per SIMD of the 256 CUs) however
            // many envs there are.  Where do the trees live?  If one arena slice per AGENT fits the budget (HK_MCTS_PERSIST_GB,
            // default 64 of the 288 GB), the tree of an agent's last plan simply stays in its slice and a re-searched root (HKA:265)
            // continues on it; otherwise the arena holds one tree per resident lane and a re-searched root is rebuilt by replaying the
            // searches it received (hk_env_mcts.h, MctsReq) — same bits, more iterations for the few lanes that reuse a root.
            const int spw = HK_GA_CALL(d, mcts_searches_per_wave());
            const long long want = (((long long)E * P.any_mcts + spw - 1) / spw) * spw;
            d.mcts.grid_lanes = (int)std::min<long long>(want, (long long)MCTS_ARENA_WAVES * spw);
            double budget_gb = 64.0;
            if (const char* ev = std::getenv("HK_MCTS_PERSIST_GB")) budget_gb = std::atof(ev);
            const double per_agent_gb = (double)na * d.mcts.pool_cap * sizeof(MNode) / 1e9;
            d.mcts.persist = per_agent_gb <= budget_gb ? 1 : 0;
            d.mcts.slots = d.mcts.persist ? (int)na : d.mcts.grid_lanes;
        }
        HK_ALLOC(d.mcts.st, na * sizeof(hk_mcts_state));
        HK_ALLOC(d.mcts.req, na * HK_GA_CALL(d, mcts_req_bytes()));
        HK_ALLOC(d.mcts.queue, 2 * 2 * na * sizeof(int));
        if (d.mcts.pool_cap > 65535) { err = "hk_create: MCTS iteration budget too large (the tree pool is limited to 65 535 nodes per search)"; return HK_ERR_UNSUPPORTED; }
        HK_ALLOC(d.mcts.nodes, (size_t)d.mcts.slots * d.mcts.pool_cap * sizeof(MNode));
        HK_ALLOC(d.mcts.roots, (size_t)d.mcts.slots * HK_GA_CALL(d, mcts_root_words()) * sizeof(int));
        HK_ALLOC(d.mcts.qcnt, 4 * sizeof(int));
        // move tables: one set per gameParams class (velocityBucketSize, timePrecision) of the MCTS agents
        for (int i = 0; i < A; i++) {
            if (cfg.high_mode[i] != HK_HIGH_MCTS) continue;
            int c = 0;
            for (; c < d.n_mcls; c++) {
                const int rep = __builtin_ctz(d.mcls[c].agents);
                if (cfg.velocity_bucket_size[i] == cfg.velocity_bucket_size[rep] && cfg.time_precision[i] == cfg.time_precision[rep]) break;
            }
            if (c == d.n_mcls) {
                if (d.n_mcls == HK_MCTS_MAX_CLASSES) { err = "hk_create: more than two (velocityBucketSize, timePrecision) classes among the MCTS agents"; return HK_ERR_UNSUPPORTED; }
                d.n_mcls++;
            }
            d.mcls[c].agents |= 1u << i;
        }
        for (int c = 0; c < d.n_mcls; c++) {
            EnvDevice::MctsClass& K = d.mcls[c];
            const int ego0 = __builtin_ctz(K.agents);
            int nv = 0;
            for (int v = 6; v < (int)P.max_speed; v += cfg.velocity_bucket_size[ego0]) nv++;       // KDG:329
            if (nv < 1 || 4 * nv > HK_MCTS_MAX_ACTIONS) { err = "hk_create: the planner's action list (4 lanes x velocity buckets of velocity_bucket_size from 6 m/s up to the top speed) must hold 4 .. 36 actions"; return HK_ERR_UNSUPPORTED; }
            K.nv = nv; K.na = 4 * nv; K.lds_tier = 7;
            const int ntab = L * 4 * (nv + 1) * K.na;
            K.ntab = ntab;
            HK_ALLOC(K.dt_tab, (size_t)ntab * sizeof(int));
            HK_ALLOC(K.load_tab, (size_t)L * 4 * K.na * sizeof(float));
            HK_ALLOC(K.rad_tab, (size_t)L * 4 * 4 * sizeof(float));
            HK_ALLOC(K.mask_tab, (size_t)(ntab / K.na) * sizeof(unsigned long long));
            HK_ALLOC(K.order_tab, (size_t)ntab);
            mcts_use_class(d, c);
            if ((rc = HK_GA_CALL(d, launch_mcts_table(d, ego0, ntab, stream, err)))) return rc;
            {   // the search kernel keeps the tables in LDS, dt as int16, 8 waves a workgroup: check the range and the fit
                std::vector<int> hdt((size_t)ntab);
                if ((e = hipMemcpyAsync(hdt.data(), K.dt_tab, (size_t)ntab * sizeof(int), hipMemcpyDeviceToHost, stream)) != hipSuccess ||
                    (e = hipStreamSynchronize(stream)) != hipSuccess) { err = std::string("hk_create: move tables: ") + hipGetErrorString(e); return HK_ERR_HIP; }
                for (int v : hdt) if (v > 32767) { err = "hk_create: a move of the discrete game takes more than 32 767 time units (timePrecision too fine for the planner's tables)"; return HK_ERR_UNSUPPORTED; }
                // what rides in LDS beside the int16 time table: everything if it fits (tier 7), else only the radii and section flags (tier 0:
                // rollout orders, tire loads and row masks are read from global memory / L2: a long track at velocityBucketSize 1)
                const int tiers[2] = {7, 0};
                int t = 0;
                while (t < 2 && HK_GA_CALL(d, mcts_lds_bytes(ntab, L, K.na, tiers[t], 4)) > 160 * 1024) t++;
                if (t == 2) { err = "hk_create: the planner's move tables do not fit the LDS (track too long)"; return HK_ERR_UNSUPPORTED; }
                K.lds_tier = tiers[t];
            }
        }
        mcts_use_class(d, 0);
    }
#undef HK_ALLOC
    // REC.Start :148-168: every agent starts inactive; results carry episode = -1; RL branch defaults to "coast"
    {
        std::vector<hk_env_state> es(E);
        std::memset(es.data(), 0, sizeof(hk_env_state) * E);
        for (int i = 0; i < E; i++) es[i].inactive_mask = (1u << A) - 1u;
        std::vector<hk_episode_result> rs(na);
        std::memset(rs.data(), 0, sizeof(hk_episode_result) * na);
        for (size_t i = 0; i < na; i++) rs[i].episode = -1;
        std::vector<int32_t> br(na, 1);
        std::vector<int> ident(E);                      // slot <-> env: the identity until the first regroup
        for (int i = 0; i < E; i++) ident[i] = i;
        if ((e = hipMemcpyAsync(d.perm, ident.data(), sizeof(int) * E, hipMemcpyHostToDevice, stream)) != hipSuccess ||
            (e = hipMemcpyAsync(d.slot_of, ident.data(), sizeof(int) * E, hipMemcpyHostToDevice, stream)) != hipSuccess) {
            err = std::string("hipMemcpy: ") + hipGetErrorString(e); return HK_ERR_HIP;
        }
        if ((e = hipMemcpyAsync(d.envs, es.data(), sizeof(hk_env_state) * E, hipMemcpyHostToDevice, stream)) != hipSuccess ||
            (e = hipMemcpyAsync(d.results, rs.data(), sizeof(hk_episode_result) * na, hipMemcpyHostToDevice, stream)) != hipSuccess ||
            (e = hipMemcpyAsync(d.act_branch, br.data(), sizeof(int32_t) * na, hipMemcpyHostToDevice, stream)) != hipSuccess ||
            (e = hipStreamSynchronize(stream)) != hipSuccess) {
            err = std::string("hipMemcpy: ") + hipGetErrorString(e); return HK_ERR_HIP;
        }
    }
    return HK_OK;
}

inline int env_flush_mcts(EnvDevice& d, hipStream_t stream, std::string& err) { return HK_GA_CALL(d, flush_mcts(d, stream, err)); }
inline int env_flush_mcts_on(EnvDevice& d, hipStream_t stream, hipStream_t side, std::string& err) { return HK_GA_CALL(d, flush_mcts_on(d, stream, side, err)); }
inline int env_mcts_invalidate(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err) { return HK_GA_CALL(d, launch_mcts_invalidate(d, cfg, stream, err)); }

inline int env_reset(EnvDevice& d, const hk_config& cfg, const int32_t* env_ids, int n, int experiment_num, hipStream_t stream,
                     std::string& err)
{
    const int E = cfg.num_envs;
    int cnt = env_ids ? n : E;
    if (cnt < 0) { err = "hk_reset: n < 0"; return HK_ERR_INVALID; }
    if (cnt == 0) return HK_OK;
    const int* dids = nullptr;
    if (env_ids) {
        for (int i = 0; i < n; i++) if (env_ids[i] < 0 || env_ids[i] >= E) { err = "hk_reset: env id out of range"; return HK_ERR_INVALID; }
        if (n > d.env_ids_cap) {
            if (d.env_ids) (void)hipFree(d.env_ids);
            d.env_ids = nullptr; d.env_ids_cap = 0;
            if (hipMalloc((void**)&d.env_ids, sizeof(int) * n) != hipSuccess) { err = "hipMalloc env ids"; return HK_ERR_HIP; }
            d.env_ids_cap = n;
        }
        if (hipMemcpyAsync(d.env_ids, env_ids, sizeof(int) * n, hipMemcpyHostToDevice, stream) != hipSuccess) { err = "hipMemcpy env ids"; return HK_ERR_HIP; }
        dids = d.env_ids;
    }
    int rc = HK_GA_CALL(d, launch_reset(d, dids, cnt, experiment_num, stream, err));
    if (rc) return rc;
    if (!env_ids) { d.ticks_since_reset = 0; d.call_ticks = 0; d.call_ticks_issued = 0; }       // every env stands on the start grid again
    if (hipStreamSynchronize(stream) != hipSuccess) { err = "hk_reset: sync failed"; return HK_ERR_HIP; }
    return HK_OK;
}

// arm hk_step(n): every env gets n ticks to run
inline int env_launch_arm(EnvDevice& d, const hk_config& cfg, int n_ticks, hipStream_t stream, std::string& err)
{
    return HK_GA_CALL(d, launch_arm(d, cfg, n_ticks, stream, err));
}

inline int env_launch_check(EnvDevice& d, const hk_config& cfg, bool lazy, hipStream_t stream, std::string& err)
{
    return HK_GA_CALL(d, launch_done_check(d, cfg, lazy ? 1 : 0, stream, err));
}

// Number of {run, lqn} rounds issued up front for n ticks when the host does not look at the device in between (short calls,
// planner / actor handles).  An env's part in a round ends when it finishes its ticks, when it parks at a solve tick (it queued a
// multi-player game, or — eager assembly — it assembled the games of the solve tick its budget ends on), or when its budget of
// `cap` ticks is used up.  So rounds <= parks + budget ends + 1.  Parks: at most one per solve tick.  n consecutive ticks hold at most
// ceil(n / cadence) solve ticks per episode segment, and every reset starts a segment with a solve tick of its own (episode_steps = 0
// whatever the phase of the old episode was): S(n) = min(n, ceil(n / cadence) + resets), resets <= 1 + n / 32 (an episode outlasts the
// start hold or, in Training mode, the ride to its first Trigger).  Budget ends: with the eager assembly (cap = cadence) they ARE
// parks; otherwise at most ceil(n / cap) - 1.  A one-tick call is 2 rounds, a 20-tick call of a plain handle 7 (round 2 issued 4 and 8
// by a looser bound).  env_check_kernel still guards the result.
inline int env_rounds_for(const hk_config& cfg, int n_ticks, int cap = RUN_CAP, bool eager = false)
{
    const int cadence = cfg.num_agents > 2 ? 4 : 1;
    static_assert(RUN_CAP > 4, "RUN_CAP must exceed the solve cadence");
    const int solve_ticks = std::min(n_ticks, (n_ticks + cadence - 1) / cadence + 1 + n_ticks / 32);
    return eager ? solve_ticks + 1 : solve_ticks + (n_ticks + cap - 1) / cap;
}

// Rounds a field needs for n ticks when nothing is queued: every env retires the launch's budget of ticks per round (long calls of
// plain handles issue these, look at the device and finish the laggards: hk_api.hip finish_ticks).
inline int env_rounds_min(const hk_config& cfg, int n_ticks, int cfg_run_cap = RUN_CAP)
{
    (void)cfg;
    return (n_ticks + cfg_run_cap - 1) / cfg_run_cap;
}

inline int env_launch_run(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err)
{
    if (++d.rounds_since_regroup >= d.regroup_rounds) {
        d.rounds_since_regroup = 0;
        int rc = HK_GA_CALL(d, launch_regroup(d, cfg, stream, err));
        if (rc) return rc;
    }
    return HK_GA_CALL(d, launch_run(d, cfg, stream, err));
}
// (the split batch: no periodic regroup between the halves' launches — the caller regroups where both streams are joined)
inline int env_launch_run_only(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err) { return HK_GA_CALL(d, launch_run(d, cfg, stream, err)); }
inline int env_launch_b1(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err) { return HK_GA_CALL(d, launch_b1(d, cfg, stream, err)); }
inline int env_launch_lqn(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err) { return HK_GA_CALL(d, launch_lqn(d, cfg, stream, err)); }
// pack the envs that still have ticks to run into the first lane groups (the tail of a call; see env_regroup_count_kernel)
inline int env_launch_regroup(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err)
{
    d.rounds_since_regroup = 0;
    return HK_GA_CALL(d, launch_regroup(d, cfg, stream, err));
}
// agent_mask: the agent slots whose observations are needed (all of them for the host's hk_get_observations / hk_observe)
inline int env_launch_observe(EnvDevice& d, const hk_config& cfg, uint32_t agent_mask, hipStream_t stream, std::string& err) { return HK_GA_CALL(d, launch_observe(d, cfg, agent_mask, stream, err)); }

}  // namespace hk
