// hk_env_params.h — hk_config -> EnvParams + the packed track tables, on the host with no device call (hk_create uploads the
// result; the host emulation of the tick kernel in tests/ builds the same tables from the same code).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include "hk_env_device.h"

namespace hk {

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
}  // namespace detail

// Validates cfg, copies the track into sections / walls (cfg then points at the copies), fills P (every field but the device
// pointers tab / perms / sec_geo) and packs the tables: pk = [SecDev L][walls NW][grid_off u16][grid_idx u16][cut u8][tmask u32].
inline int env_build_params(hk_config& cfg, std::vector<hk_section>& sections, std::vector<hk_wall_seg>& walls, EnvParams& P,
                            std::vector<unsigned char>& pk, std::vector<int>& perms, std::string& err)
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
        if (cfg.low_mode[i] == HK_LOW_LQR) P.any_lqr += 1;
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
    {   // engine restatement (hk.h hk_engine_params)
        const hk_engine_params& g = cfg.engine;
        if (g.wheel_friction || g.contact_yaw) {
            if (!(g.mass > 0.0f) || !(g.inertia_y > 0.0f)) { err = "hk_create: engine.mass / engine.inertia_y must be positive"; return HK_ERR_INVALID; }
        }
        if (g.wheel_friction) {
            if (!(g.axle_zf > g.axle_zr) || !(g.side_ext_slip > 0.0f) || !(g.side_asy_slip > g.side_ext_slip) || !(g.slip_min_speed > 0.0f) ||
                !(g.side_slope0 >= 0.0f && g.side_slope0 <= 3.0f)) {
                err = "hk_create: engine wheel parameters out of range (axle_zf > axle_zr, 0 < ext_slip < asy_slip, slip_min_speed > 0, 0 <= side_slope0 <= 3)"; return HK_ERR_INVALID;
            }
        }
        P.eng = g;
        EngDerived& D = P.engd;
        std::memset(&D, 0, sizeof(D));
        auto curve = [](EngCurve& c, float ext_slip, float ext_value, float asy_slip, float asy_value, float slope0) {
            c.ext = ext_slip; c.asy = asy_slip;
            c.inv_ext = 1.0f / ext_slip; c.inv_span = 1.0f / (asy_slip - ext_slip);
            c.a3 = ext_value * (slope0 - 2.0f); c.a2 = ext_value * (3.0f - 2.0f * slope0); c.a1 = ext_value * slope0;
            c.b3 = -2.0f * (asy_value - ext_value); c.b2 = 3.0f * (asy_value - ext_value); c.b0 = ext_value;
            c.flat = asy_value;
        };
        if (g.mass > 0.0f && g.inertia_y > 0.0f) {
            D.inv_m = 1.0f / g.mass; D.inv_i = 1.0f / g.inertia_y;
            const float span = g.axle_zf - g.axle_zr;
            if (g.wheel_friction) {
                const float load_f = (-g.axle_zr / span) * g.mass * g.gravity, load_r = (g.axle_zf / span) * g.mass * g.gravity;
                D.side_kf = g.side_stiffness * load_f * cfg.dt; D.side_kr = g.side_stiffness * load_r * cfg.dt;
                D.jden_r = 1.0f / (D.inv_m + g.axle_zr * g.axle_zr * D.inv_i);
                curve(D.side, g.side_ext_slip, g.side_ext_value, g.side_asy_slip, g.side_asy_value, g.side_slope0);
                if (g.wheel_rolling) {
                    if (!(g.wheel_mass > 0.0f) || !(g.wheel_radius_f > 0.0f) || !(g.wheel_radius_r > 0.0f) || !(g.wheel_damping >= 0.0f) ||
                        !(g.fwd_ext_slip > 0.0f) || !(g.fwd_asy_slip > g.fwd_ext_slip) || !(g.long_slip_min_speed > 0.0f)) {
                        err = "hk_create: engine rolling parameters out of range (wheel_mass, radii > 0, damping >= 0, 0 < fwd_ext_slip < fwd_asy_slip, long_slip_min_speed > 0)"; return HK_ERR_INVALID;
                    }
                    D.inv_mw = 1.0f / g.wheel_mass;
                    D.fwd_kf = g.fwd_stiffness * load_f * cfg.dt; D.fwd_kr = g.fwd_stiffness * load_r * cfg.dt;
                    D.inv_damp_f = 1.0f / (1.0f + cfg.dt * g.wheel_damping / (0.5f * g.wheel_mass * g.wheel_radius_f * g.wheel_radius_f));
                    D.inv_damp_r = 1.0f / (1.0f + cfg.dt * g.wheel_damping / (0.5f * g.wheel_mass * g.wheel_radius_r * g.wheel_radius_r));
                    D.jlden_r = 1.0f / (D.inv_mw + D.inv_m);
                    curve(D.fwd, g.fwd_ext_slip, g.fwd_ext_value, g.fwd_asy_slip, g.fwd_asy_value, g.side_slope0);
                }
            }
        }
    }
    P.max_speed = cfg.stats.TopSpeed > cfg.stats.ReverseSpeed ? cfg.stats.TopSpeed : cfg.stats.ReverseSpeed;   // AK:210
    // phase_assemble walks at most 6 two-metre samples of the forward ray: hits up to 11.25 m are found, and the ray is compared with
    // speed / 2 (HKA:834) — exact for speeds up to 22.5 m/s (the reference's karts: 15)
    if (P.max_speed * 0.5f > 11.25f) { err = "hk_create: TopSpeed / ReverseSpeed above 22.5 m/s is outside the forward ray's sampled range"; return HK_ERR_UNSUPPORTED; }
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
    std::vector<unsigned char> ncnt;           // walls of the cell's list within NEAR_REACH: they come first in the list
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
        goff.assign(ncell + 1, 0); ncnt.assign(ncell, 0);
        for (int iz = 0; iz < P.grid_nz; iz++)
            for (int ix = 0; ix < P.grid_nx; ix++) {
                const double cx0 = P.grid_x0 + ix * (double)GRID_CELL, cz0 = P.grid_z0 + iz * (double)GRID_CELL;
                const double cx1 = cx0 + GRID_CELL, cz1 = cz0 + GRID_CELL;
                const size_t c = (size_t)iz * P.grid_nx + ix;
                if (gidx.size() > 65000) { err = "hk_create: wall grid too large"; return HK_ERR_UNSUPPORTED; }
                goff[c] = (unsigned short)gidx.size();
                std::vector<unsigned short> far;
                for (size_t w = 0; w < walls.size(); w++) {
                    const hk_wall_seg& s = walls[w];
                    auto inside = [&](double x, double z) { return x >= cx0 && x <= cx1 && z >= cz0 && z <= cz1; };
                    double dmin = (inside(s.x0, s.z0) || inside(s.x1, s.z1)) ? 0.0 : 1e30;
                    const double ex[5] = {cx0, cx1, cx1, cx0, cx0}, ez[5] = {cz0, cz0, cz1, cz1, cz0};
                    for (int q = 0; q < 4 && dmin > 0.0; q++)
                        dmin = std::min(dmin, detail::seg_seg_dist(s.x0, s.z0, s.x1, s.z1, ex[q], ez[q], ex[q + 1], ez[q + 1]));
                    if (dmin <= (double)NEAR_REACH) gidx.push_back((unsigned short)w);
                    else if (dmin <= (double)GRID_REACH) far.push_back((unsigned short)w);
                }
                const size_t nn = gidx.size() - goff[c];
                if (nn > 255) { err = "hk_create: more than 255 wall segments within 1.3 m of a grid cell"; return HK_ERR_UNSUPPORTED; }
                ncnt[c] = (unsigned char)nn;
                gidx.insert(gidx.end(), far.begin(), far.end());
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
    std::vector<int> perm(A);
    perms.clear(); pk.clear();
    for (int i = 0; i < A; i++) perm[i] = i;
    do { perms.insert(perms.end(), perm.begin(), perm.end()); } while (std::next_permutation(perm.begin(), perm.end()));
    P.nperm = (int)(perms.size() / A);
    // pack the tables: [SecDev L][walls NW][grid_off u16][grid_idx u16][cut u8], 16-B aligned
    {
        if (walls.size() > 65535) { err = "hk_create: more than 65535 wall segments"; return HK_ERR_UNSUPPORTED; }
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
        P.o_ncnt = seg(ncnt.data(), ncnt.size());
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
        {   // tight Trigger candidates per cell of the wall grid.  A kart overlaps Trigger t only if its origin lies within the box (half extents
            // TRIG_HX x TRIG_HZ about the Trigger, turned by the section's yaw) grown by the capsule's reach 1.107 m; + 3 cm for the float rounding
            // of the cell index.  A border cell also stands for every position beyond it (grid_cell clamps).
            const double grow = 1.107 + 0.03;
            std::vector<uint32_t> tm2((size_t)P.grid_nx * P.grid_nz * 2, 0u);
            for (int iz = 0; iz < P.grid_nz; iz++)
                for (int ix = 0; ix < P.grid_nx; ix++) {
                    double cx0 = P.grid_x0 + ix * (double)GRID_CELL, cz0 = P.grid_z0 + iz * (double)GRID_CELL;
                    double cx1 = cx0 + GRID_CELL, cz1 = cz0 + GRID_CELL;
                    if (ix == 0) cx0 = -1e9;
                    if (iz == 0) cz0 = -1e9;
                    if (ix == P.grid_nx - 1) cx1 = 1e9;
                    if (iz == P.grid_nz - 1) cz1 = 1e9;
                    for (int t = 0; t < L; t++) {
                        // conservative: the cell rectangle against the bounding circle ... no: against the grown box's own axes (separating axes)
                        const double tx = sections[t].trig_x, tz = sections[t].trig_z;
                        const double yaw = (double)sd[t].yaw_rad, fxx = std::sin(yaw), fzz = std::cos(yaw);      // box forward; right = (fzz, -fxx)
                        const double hx = TRIG_HX + grow, hz = TRIG_HZ + grow;
                        // clamp the infinite border cells to something finite around the box for the axis tests
                        const double bx0 = std::max(cx0, tx - 100.0), bx1 = std::min(cx1, tx + 100.0), bz0 = std::max(cz0, tz - 100.0), bz1 = std::min(cz1, tz + 100.0);
                        if (bx0 > bx1 || bz0 > bz1) continue;
                        // axes of the cell (x, z): the box's projection
                        const double ex = std::fabs(fzz) * hx + std::fabs(fxx) * hz, ez = std::fabs(fxx) * hx + std::fabs(fzz) * hz;
                        if (tx + ex < bx0 || tx - ex > bx1 || tz + ez < bz0 || tz - ez > bz1) continue;
                        // axes of the box: the cell's projection
                        const double ccx = 0.5 * (bx0 + bx1) - tx, ccz = 0.5 * (bz0 + bz1) - tz, hcx = 0.5 * (bx1 - bx0), hcz = 0.5 * (bz1 - bz0);
                        const double pr = ccx * fzz - ccz * fxx, pf = ccx * fxx + ccz * fzz;                      // cell centre in the box frame (right, forward)
                        const double rr = hcx * std::fabs(fzz) + hcz * std::fabs(fxx), rf = hcx * std::fabs(fxx) + hcz * std::fabs(fzz);
                        if (std::fabs(pr) > hx + rr || std::fabs(pf) > hz + rf) continue;
                        tm2[((size_t)iz * P.grid_nx + ix) * 2 + (t >> 5)] |= 1u << (t & 31);
                    }
                }
            // Stored as TWO Trigger indices per cell, a byte each (0xFF: none): 2 bytes instead of a 64-bit mask, a third of the staged tables less
            // (Oval 42 -> 29 KB per block and launch) and the Complex track's candidates fit the LDS at all (30 -> 7.5 KB; round 4 kept them in
            // global memory).  Triggers are 10 m apart along the track, so a 2 m cell meets one, at a corner two; a cell that meets more (a hairpin of a
            // track not seen yet) is marked 0xFEFE and the kernel falls back to the coarse masks above — a superset, slower, never wrong.
            const size_t ncell = (size_t)P.grid_nx * P.grid_nz;
            std::vector<unsigned short> tc2(ncell, (unsigned short)0xFFFF);
            for (size_t c = 0; c < ncell; c++) {
                int found[3], nf = 0;
                for (int t = 0; t < L && nf < 3; t++) if (tm2[c * 2 + (t >> 5)] & (1u << (t & 31))) found[nf++] = t;
                if (nf > 2) tc2[c] = (unsigned short)0xFEFE;
                else if (nf == 2) tc2[c] = (unsigned short)(found[0] | (found[1] << 8));
                else if (nf == 1) tc2[c] = (unsigned short)(found[0] | 0xFF00);
            }
            if (L > 0xFD) { err = "hk_create: more than 253 sections (the Trigger candidate table holds section indices in a byte)"; return HK_ERR_UNSUPPORTED; }
            P.o_tmask2 = seg(tc2.data(), tc2.size() * sizeof(unsigned short));
        }
        P.L_magic = (uint32_t)((1ull << 32) / (unsigned long long)L) + 1u;
        pk.resize((pk.size() + 15) & ~size_t(15));
        P.tab_bytes = (int)pk.size();
        P.tab_stage_bytes = P.tab_bytes;      // (env_create narrows it when the tight Trigger masks do not fit the LDS budget)
    }
    return HK_OK;
}

}  // namespace hk
