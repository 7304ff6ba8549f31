// hk_env_kernels.h — host side of the batched kart environment: device tables, candidate wall lists, the round scheduler.
// (API translation unit only; the kernels live in hk_ga4.hip / hk_ga8.hip behind hk_env_host.h's GaOps table.)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "hk_env_host.h"

namespace hk {

// forward to the kernels of the handle's lane-group width (hk_env_host.h: GaOps)
#define HK_GA_CALL(d, call) (ga_ops(d).call)

namespace detail {

inline double seg_seg_dist(double p1x, double p1z, double q1x, double q1z, double p2x, double p2z, double q2x, double q2z)
{   // host-only, used for culling radii (not on the parity path)
    auto pt_seg = [](double px, double pz, double ax, double az, double bx, double bz) {
        double dx = bx - ax, dz = bz - az, l2 = dx * dx + dz * dz;
        double t = l2 > 0 ? ((px - ax) * dx + (pz - az) * dz) / l2 : 0.0;
        t = std::min(1.0, std::max(0.0, t));
        double cx = ax + t * dx - px, cz = az + t * dz - pz;
        return std::sqrt(cx * cx + cz * cz);
    };
    auto cross = [](double ax, double az, double bx, double bz) { return ax * bz - az * bx; };
    double d1 = cross(q1x - p1x, q1z - p1z, p2x - p1x, p2z - p1z), d2 = cross(q1x - p1x, q1z - p1z, q2x - p1x, q2z - p1z);
    double d3 = cross(q2x - p2x, q2z - p2z, p1x - p2x, p1z - p2z), d4 = cross(q2x - p2x, q2z - p2z, q1x - p2x, q1z - p2z);
    if (((d1 > 0) != (d2 > 0)) && ((d3 > 0) != (d4 > 0))) return 0.0;
    return std::min(std::min(pt_seg(p1x, p1z, p2x, p2z, q2x, q2z), pt_seg(q1x, q1z, p2x, p2z, q2x, q2z)),
                    std::min(pt_seg(p2x, p2z, p1x, p1z, q1x, q1z), pt_seg(q2x, q2z, p1x, p1z, q1x, q1z)));
}

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
    d.mcts.nv = K.nv; d.mcts.ntab = K.ntab;
}

inline void env_destroy(EnvDevice& d)
{
    void* ptrs[] = {d.agents, d.envs, d.results, d.lq_debug, d.obs, d.act_steer, d.act_branch, d.reward_out, d.status, d.game_stats, d.games, d.queue_cnt, d.queue, d.env_ids, d.tab, d.perms,
                    d.rw.sec_time, d.rw.sec_cnt, d.rw.hit_code, d.mcts.st, d.mcts.req, d.mcts.qcnt, d.mcts.queue, d.mcts.nodes, d.mcts.roots, d.sec_geo, d.perm, d.perm_counts};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int c = 0; c < d.n_mcls; c++) { void* t[] = {d.mcls[c].dt_tab, d.mcls[c].load_tab, d.mcls[c].rad_tab, d.mcls[c].mask_tab, d.mcls[c].order_tab}; for (void* p : t) if (p) (void)hipFree(p); }
    d = EnvDevice{};
}

inline int env_create(hk_config& cfg, std::vector<hk_section>& sections, std::vector<hk_wall_seg>& walls, EnvDevice& d,
                      hipStream_t stream, std::string& err)
{
    const int A = cfg.num_agents, L = cfg.num_sections, E = cfg.num_envs;
    if (E < 1 || A < 1 || L < 1 || L > HK_MAX_SECTIONS || cfg.num_walls < 0 || !cfg.sections || (cfg.num_walls > 0 && !cfg.walls)) {
        err = "hk_create: bad num_envs / num_agents / track table"; return HK_ERR_INVALID;
    }
    if (A > ENV_MAXA) { err = "hk_create: num_agents > 8"; return HK_ERR_UNSUPPORTED; }
    // The round scheduler (env_rounds_for) and the planner's request queues (two requests per agent between flushes) assume an
    // episode outlasts the start hold and a 100-tick replan period; every reference scene uses 6 000.
    if (cfg.max_episode_steps < 100 || cfg.max_episode_steps <= cfg.start_hold_ticks) {
        err = "hk_create: max_episode_steps must be >= 100 and exceed start_hold_ticks"; return HK_ERR_INVALID;
    }
    if (cfg.start_hold_ticks < 0 || cfg.laps < 1 || cfg.section_horizon < 1 || cfg.section_horizon > 16 || !(cfg.dt > 0.0f)) {
        err = "hk_create: bad start_hold_ticks / laps / section_horizon / dt"; return HK_ERR_INVALID;
    }
    for (int i = 0; i < A; i++) {
        if (cfg.n_team[i] < 0 || cfg.n_other[i] < 0 || cfg.n_team[i] + cfg.n_other[i] != A - 1) {
            err = "hk_create: teamAgents + otherAgents of every agent must list every other agent exactly once"; return HK_ERR_INVALID;
        }
        // every entry in range, not the agent itself, no duplicates (the kernels index kart arrays with them)
        uint32_t seen = 1u << i;
        for (int j = 0; j < cfg.n_team[i] + cfg.n_other[i]; j++) {
            const int o = j < cfg.n_team[i] ? cfg.team_agents[i][j] : cfg.other_agents[i][j - cfg.n_team[i]];
            if (o < 0 || o >= A || ((seen >> o) & 1u)) {
                err = "hk_create: teamAgents / otherAgents entries must be distinct agent indices in [0, num_agents) other than the agent itself"; return HK_ERR_INVALID;
            }
            seen |= 1u << o;
        }
        if (cfg.high_mode[i] != HK_HIGH_FIXED && cfg.high_mode[i] != HK_HIGH_MCTS) { err = "hk_create: bad high_mode"; return HK_ERR_INVALID; }
        if (cfg.high_mode[i] == HK_HIGH_MCTS) {
            if (cfg.tree_search_depth[i] < 1 || cfg.tree_search_depth[i] > HK_MCTS_MAX_DEPTH || cfg.velocity_bucket_size[i] < 1 ||
                cfg.time_precision[i] < 1 || cfg.section_window[i] < 1 || cfg.section_window[i] > 4) { err = "hk_create: bad MCTS gameParams (depth 1..8, bucket, precision, window 1..4)"; return HK_ERR_INVALID; }
            // a search requested on tick t runs between two launches of the tick kernel (<= RUN_CAP ticks each) and must be
            // finished before tick t + latency; it must also have been consumed before the next request (every 100 ticks)
            if (cfg.mcts_iterations < 1 || cfg.mcts_initial_iterations < 1 || cfg.mcts_latency_ticks <= MCTS_MIN_LATENCY || cfg.mcts_latency_ticks >= 100 ||
                cfg.mcts_initial_latency_ticks <= MCTS_MIN_LATENCY || cfg.mcts_initial_latency_ticks >= 100) {
                err = "hk_create: MCTS budget / latency out of range (iterations >= 1, 40 < latency ticks < 100)"; return HK_ERR_INVALID;
            }
        }
        if (cfg.low_mode[i] != HK_LOW_LQR && cfg.low_mode[i] != HK_LOW_RL) { err = "hk_create: LowMode MPC is dead code in the reference"; return HK_ERR_UNSUPPORTED; }
        if (cfg.tree_search_depth[i] < 0 || cfg.tree_search_depth[i] > L) { err = "hk_create: bad tree_search_depth"; return HK_ERR_INVALID; }
    }
    sections.assign(cfg.sections, cfg.sections + L);
    walls.assign(cfg.walls, cfg.walls + cfg.num_walls);
    cfg.sections = sections.data();
    cfg.walls = walls.data();
    EnvParams& P = d.P;
    std::memset(&P, 0, sizeof(P));
    P.E = E; P.A = A; P.L = L; P.NW = cfg.num_walls;
    P.dt = cfg.dt; P.kart_y = cfg.kart_y; P.st = cfg.stats;
    P.laps = cfg.laps; P.max_steps = cfg.max_episode_steps; P.max_lane_changes = cfg.max_lane_changes;
    P.H = cfg.section_horizon; P.disable_on_end = cfg.disable_on_end; P.hold = cfg.start_hold_ticks; P.auto_reset = cfg.auto_reset;
    for (int i = 0; i < A; i++) {
        P.high_mode[i] = cfg.high_mode[i]; P.low_mode[i] = cfg.low_mode[i]; P.depth[i] = cfg.tree_search_depth[i];
        P.vbucket[i] = cfg.velocity_bucket_size[i];
        P.team_of[i] = cfg.team_of[i]; P.time_precision[i] = cfg.time_precision[i]; P.section_window[i] = cfg.section_window[i];
        if (cfg.high_mode[i] == HK_HIGH_MCTS) P.any_mcts += 1;
        P.n_team[i] = cfg.n_team[i]; P.n_other[i] = cfg.n_other[i];
        for (int j = 0; j < ENV_MAXA; j++) { P.team[i][j] = cfg.team_agents[i][j]; P.other[i][j] = cfg.other_agents[i][j]; }
    }
    for (int i = 0; i < HK_NUM_SENSORS; i++) {
        const float dl = cfg.sensor_yaw_deg[i] * DEG2RAD_F;
        P.sens_c[i] = hk_cosf(dl); P.sens_s[i] = hk_sinf(dl); P.ray_dist[i] = cfg.ray_distance[i];
    }
    P.training_reset = cfg.env_mode == HK_MODE_TRAINING ? 1 : 0; P.train_seed = cfg.train_seed;
    P.hold_dedupe = 0;          // set below, once the planner count is known
    P.rewards = cfg.rewards; P.rw = cfg.rw;
    for (int i = 0; i < HK_NUM_SENSORS; i++) { P.wall_val[i] = cfg.wall_hit_validation[i]; P.agent_val[i] = cfg.agent_hit_validation[i]; }
    for (int i = 0; i < A; i++) {
        if (cfg.team_of[i] < 0 || cfg.team_of[i] >= A) { err = "hk_create: team_of out of range"; return HK_ERR_INVALID; }
        P.training_agent[i] = cfg.training_agent[i];
        P.team_size[cfg.team_of[i]] += 1;
        if (cfg.team_of[i] + 1 > P.n_teams) P.n_teams = cfg.team_of[i] + 1;
    }
    P.hold_dedupe = (P.any_mcts == 0 && !P.training_reset && !std::getenv("HK_NO_HOLD_DEDUPE")) ? 1 : 0;
    P.run_cap = RUN_CAP;
    P.eager = 0;
    P.mcts_iter = cfg.mcts_iterations; P.mcts_iter0 = cfg.mcts_initial_iterations; P.mcts_lat = cfg.mcts_latency_ticks;
    P.mcts_lat0 = cfg.mcts_initial_latency_ticks; P.mcts_seed = cfg.mcts_seed;
    P.jitter_seed = cfg.jitter_seed; P.jitter_pos = cfg.jitter_pos; P.jitter_yaw = cfg.jitter_yaw; P.env_id_base = cfg.env_id_base;
    P.max_speed = cfg.stats.TopSpeed > cfg.stats.ReverseSpeed ? cfg.stats.TopSpeed : cfg.stats.ReverseSpeed;   // AK:210
    P.init_acc = -cfg.stats.TireWearRate * hk_logf(1 - ((cfg.stats.MaxSteer - cfg.stats.MinSteer) * 0.25f / cfg.stats.MaxSteer));  // REC:588
    {
        const float dyc = 0.5f - 0.582f;   // sensor height - capsule centre height (kart-local)
        P.ray_agent_r = sqrtf(CAP_R * CAP_R - dyc * dyc);
    }
    { const char* dbg = std::getenv("HK_LQ_DEBUG"); P.debug = cfg.debug_taps | (dbg ? std::atoi(dbg) : 0); }   // hk_get_lq_debug taps (hk.h: debug_taps)
    // sections with forward precomputed (same float expressions as everywhere else)
    std::vector<SecDev> sd(L);
    for (int i = 0; i < L; i++) {
        const hk_section& s = sections[i];
        SecDev& o = sd[i];
        o.trig_x = s.trig_x; o.trig_z = s.trig_z;
        o.yaw_rad = s.yaw_deg * DEG2RAD_F;
        o.fx = hk_sinf(o.yaw_rad); o.fz = hk_cosf(o.yaw_rad);
        o.marker_y = s.marker_y; o.inside_radius = s.track_inside_radius; o.optimal_lane = s.optimal_lane;
        for (int l = 0; l < 4; l++) { o.lane_x[l] = s.lane_x[l]; o.lane_z[l] = s.lane_z[l]; }
    }
    // uniform wall grid: per cell every wall segment within GRID_REACH of the cell rectangle, ascending wall index
    std::vector<unsigned short> goff, gidx;
    {
        float x0 = 0, x1 = 1, z0 = 0, z1 = 1;
        for (size_t w = 0; w < walls.size(); w++) {
            const hk_wall_seg& s = walls[w];
            float lx = std::min(s.x0, s.x1), hx = std::max(s.x0, s.x1), lz = std::min(s.z0, s.z1), hz = std::max(s.z0, s.z1);
            if (w == 0) { x0 = lx; x1 = hx; z0 = lz; z1 = hz; }
            x0 = std::min(x0, lx); x1 = std::max(x1, hx); z0 = std::min(z0, lz); z1 = std::max(z1, hz);
        }
        P.grid_x0 = std::floor(x0) - 1.0f; P.grid_z0 = std::floor(z0) - 1.0f;
        P.grid_inv = 1.0f / GRID_CELL;
        P.grid_nx = (int)std::ceil((x1 + 1.0f - P.grid_x0) / GRID_CELL) + 1;
        P.grid_nz = (int)std::ceil((z1 + 1.0f - P.grid_z0) / GRID_CELL) + 1;
        const size_t ncell = (size_t)P.grid_nx * P.grid_nz;
        goff.assign(ncell + 1, 0);
        for (int iz = 0; iz < P.grid_nz; iz++)
            for (int ix = 0; ix < P.grid_nx; ix++) {
                const double cx0 = P.grid_x0 + ix * (double)GRID_CELL, cz0 = P.grid_z0 + iz * (double)GRID_CELL;
                const double cx1 = cx0 + GRID_CELL, cz1 = cz0 + GRID_CELL;
                const size_t c = (size_t)iz * P.grid_nx + ix;
                if (gidx.size() > 65000) { err = "hk_create: wall grid too large"; return HK_ERR_UNSUPPORTED; }
                goff[c] = (unsigned short)gidx.size();
                for (size_t w = 0; w < walls.size(); w++) {
                    const hk_wall_seg& s = walls[w];
                    auto inside = [&](double x, double z) { return x >= cx0 && x <= cx1 && z >= cz0 && z <= cz1; };
                    double dmin = (inside(s.x0, s.z0) || inside(s.x1, s.z1)) ? 0.0 : 1e30;
                    const double ex[5] = {cx0, cx1, cx1, cx0, cx0}, ez[5] = {cz0, cz0, cz1, cz1, cz0};
                    for (int q = 0; q < 4 && dmin > 0.0; q++)
                        dmin = std::min(dmin, detail::seg_seg_dist(s.x0, s.z0, s.x1, s.z1, ex[q], ez[q], ex[q + 1], ez[q + 1]));
                    if (dmin <= (double)GRID_REACH) gidx.push_back((unsigned short)w);
                }
            }
        goff[ncell] = (unsigned short)gidx.size();
    }
    // cut table: Physics.Raycast(lane marker -> next lane marker) vs every wall (HKA:832), static geometry
    std::vector<unsigned char> cut((size_t)L * 25, 0);
    for (int s = 0; s < L; s++)
        for (int la = 0; la < 5; la++)
            for (int lb = 0; lb < 5; lb++) {
                const hk_section& a = sections[s];
                const hk_section& b = sections[(s + 1) % L];
                float lx = la ? a.lane_x[la - 1] : a.trig_x, lz = la ? a.lane_z[la - 1] : a.trig_z;
                float nx = lb ? b.lane_x[lb - 1] : b.trig_x, nz = lb ? b.lane_z[lb - 1] : b.trig_z;
                float cdx = nx - lx, cdz = nz - lz;
                float clen = sqrtf((lx - nx) * (lx - nx) + 0.0f * 0.0f + (lz - nz) * (lz - nz));
                bool hit = false;
                if (clen > 0.0f) {
                    float dx = cdx / clen, dz = cdz / clen;
                    for (size_t w = 0; w < walls.size() && !hit; w++) {
                        float t = ray_seg_host(lx, lz, dx, dz, walls[w]);
                        if (t >= 0.0f && t <= clen) hit = true;
                    }
                }
                cut[((size_t)s * 5 + la) * 5 + lb] = hit ? 1 : 0;
            }
    // lexicographic permutations (REC:137-145,166)
    std::vector<int> perm(A), perms;
    for (int i = 0; i < A; i++) perm[i] = i;
    do { perms.insert(perms.end(), perm.begin(), perm.end()); } while (std::next_permutation(perm.begin(), perm.end()));
    P.nperm = (int)(perms.size() / A);
    // pack the tables: [SecDev L][walls NW][grid_off u16][grid_idx u16][cut u8], 16-B aligned
    {
        if (walls.size() > 65535) { err = "hk_create: more than 65535 wall segments"; return HK_ERR_UNSUPPORTED; }
        std::vector<unsigned char> pk;
        auto seg = [&](const void* src, size_t bytes) {
            size_t off = (pk.size() + 15) & ~size_t(15);
            pk.resize(off + bytes);
            if (bytes) std::memcpy(pk.data() + off, src, bytes);
            return (int)off;
        };
        seg(sd.data(), sd.size() * sizeof(SecDev));
        P.o_walls = seg(walls.data(), walls.size() * sizeof(hk_wall_seg));
        P.o_goff = seg(goff.data(), goff.size() * sizeof(unsigned short));
        P.o_gidx = seg(gidx.data(), gidx.size() * sizeof(unsigned short));
        P.o_cut = seg(cut.data(), cut.size());
        {   // Trigger candidate masks over a coarse grid whose box holds the wall grid and every Trigger centre
            double bx0 = P.grid_x0, bz0 = P.grid_z0, bx1 = P.grid_x0 + P.grid_nx * (double)GRID_CELL, bz1 = P.grid_z0 + P.grid_nz * (double)GRID_CELL;
            for (int t = 0; t < L; t++) {
                bx0 = std::min(bx0, (double)sections[t].trig_x - 1.0); bx1 = std::max(bx1, (double)sections[t].trig_x + 1.0);
                bz0 = std::min(bz0, (double)sections[t].trig_z - 1.0); bz1 = std::max(bz1, (double)sections[t].trig_z + 1.0);
            }
            P.tgrid_x0 = (float)std::floor(bx0); P.tgrid_z0 = (float)std::floor(bz0);
            P.tgrid_inv = 1.0f / TRIG_CELL;
            P.tgrid_nx = (int)std::ceil((bx1 - P.tgrid_x0) / TRIG_CELL) + 1;
            P.tgrid_nz = (int)std::ceil((bz1 - P.tgrid_z0) / TRIG_CELL) + 1;
            if ((long long)P.tgrid_nx * P.tgrid_nz > 65536) { err = "hk_create: track too large for the Trigger grid"; return HK_ERR_UNSUPPORTED; }
            std::vector<uint32_t> tm((size_t)P.tgrid_nx * P.tgrid_nz * 2, 0u);
            for (int iz = 0; iz < P.tgrid_nz; iz++)
                for (int ix = 0; ix < P.tgrid_nx; ix++) {
                    // 0.25 m slack on the cell rectangle for the float rounding of the cell index
                    const double cx0 = P.tgrid_x0 + ix * (double)TRIG_CELL - 0.25, cz0 = P.tgrid_z0 + iz * (double)TRIG_CELL - 0.25;
                    const double cx1 = cx0 + TRIG_CELL + 0.5, cz1 = cz0 + TRIG_CELL + 0.5;
                    for (int t = 0; t < L; t++) {
                        const double tx = sections[t].trig_x, tz = sections[t].trig_z;
                        const double ddx = tx < cx0 ? cx0 - tx : (tx > cx1 ? tx - cx1 : 0.0), ddz = tz < cz0 ? cz0 - tz : (tz > cz1 ? tz - cz1 : 0.0);
                        if (ddx * ddx + ddz * ddz <= (double)TRIG_REACH * TRIG_REACH) tm[((size_t)iz * P.tgrid_nx + ix) * 2 + (t >> 5)] |= 1u << (t & 31);
                    }
                }
            P.o_tmask = seg(tm.data(), tm.size() * sizeof(uint32_t));
        }
        pk.resize((pk.size() + 15) & ~size_t(15));
        P.tab_bytes = (int)pk.size();
        int rc0;
        if ((rc0 = detail::upload(&d.tab, pk, err))) return rc0;
        P.tab = d.tab;
        d.tab_lds = (P.tab_bytes <= 48 * 1024 && !std::getenv("HK_TAB_GLOBAL")) ? P.tab_bytes : 0;      // Oval: ~20 KB per block (HK_TAB_GLOBAL=1: the global-memory instantiation, for tests)
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
    HK_ALLOC(d.results, na * sizeof(hk_episode_result));
    HK_ALLOC(d.lq_debug, na * sizeof(hk_lq_debug));
    HK_ALLOC(d.obs, na * obs_dim * sizeof(float));
    HK_ALLOC(d.act_steer, na * sizeof(float));
    HK_ALLOC(d.act_branch, na * sizeof(int32_t));
    HK_ALLOC(d.reward_out, 2 * na * sizeof(float));
    HK_ALLOC(d.status, 4 * sizeof(int));
    HK_ALLOC(d.game_stats, 64 * sizeof(unsigned long long));     // [0, 16) games by player count, [16, 64) diagnostic stamps
    HK_ALLOC(d.games, na * HK_GA_CALL(d, game_doubles_per_ego()) * sizeof(double));
    HK_ALLOC(d.queue_cnt, 4 * 16 * sizeof(int));
    HK_ALLOC(d.perm, (size_t)E * sizeof(int));
    HK_ALLOC(d.perm_counts, 32 * sizeof(int));
    HK_ALLOC(d.queue, 4 * HK_GA_CALL(d, queue_ints_per_set(na)) * sizeof(int));
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
            for (int v = 6; v < (int)P.max_speed && nv < 5; v += cfg.velocity_bucket_size[ego0]) nv++;
            K.nv = nv;
            const int ntab = L * 4 * (nv + 1) * HK_MCTS_MAX_ACTIONS;
            K.ntab = ntab;
            HK_ALLOC(K.dt_tab, (size_t)ntab * sizeof(int));
            HK_ALLOC(K.load_tab, (size_t)L * 4 * HK_MCTS_MAX_ACTIONS * sizeof(float));
            HK_ALLOC(K.rad_tab, (size_t)L * 4 * 4 * sizeof(float));
            HK_ALLOC(K.mask_tab, (size_t)(ntab / HK_MCTS_MAX_ACTIONS) * sizeof(uint32_t));
            HK_ALLOC(K.order_tab, (size_t)ntab);
            mcts_use_class(d, c);
            if ((rc = HK_GA_CALL(d, launch_mcts_table(d, ego0, ntab, stream, err)))) return rc;
            {   // the search kernel keeps the tables in LDS, dt as int16, 8 waves a workgroup: check the range and the fit
                std::vector<int> hdt((size_t)ntab);
                if ((e = hipMemcpyAsync(hdt.data(), K.dt_tab, (size_t)ntab * sizeof(int), hipMemcpyDeviceToHost, stream)) != hipSuccess ||
                    (e = hipStreamSynchronize(stream)) != hipSuccess) { err = std::string("hk_create: move tables: ") + hipGetErrorString(e); return HK_ERR_HIP; }
                for (int v : hdt) if (v > 32767) { err = "hk_create: a move of the discrete game takes more than 32 767 time units (timePrecision too fine for the planner's tables)"; return HK_ERR_UNSUPPORTED; }
                if (HK_GA_CALL(d, mcts_lds_bytes(ntab, L, 4)) > 160 * 1024) { err = "hk_create: the planner's move tables do not fit the LDS (track too long)"; return HK_ERR_UNSUPPORTED; }
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
