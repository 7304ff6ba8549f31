// hk_env_step.h — the non-LQ part of a tick: episode controller, sensing, ArcadeKart model, engine restatement.
//
// Thread mapping: GA consecutive lanes (a "quad" for GA = 4: up to 4 agents per env; 8 lanes for the synthetic 8-agent
// configuration) own one race instance, lane q = agent q; a 256-thread block holds 256 / GA envs.  Kart-to-kart data moves
// with group-wide shuffles; per-env words (episode_steps, inactive set) are recomputed identically by the lanes of the group
// and written by lane 0.  Wall queries read the uniform wall grid that
// hk_create builds (per 2 m cell: every segment within 2.2 m), staged in LDS with the other track tables; the lists
// are supersets of what can be reached, so results equal a brute force scan over every wall (what the CPU oracle does).
// (included once per lane-group width by hk_env_ga.h: no include guard, namespace hk::HK_GA_NS)
#include "hk_env_device.h"

namespace hk { namespace HK_GA_NS {

// REC.ResetGame :499-719 for one agent (Experiment / Race grid); see oracle reset_env for the line map
template <bool TRAIN>
__device__ inline void reset_agent(const EnvParams& P, const TabView& T, int env, int i, int experiment_num, int episodes_done, hk_agent_state* a)
{
    const int pi = ((experiment_num % P.nperm) + P.nperm) % P.nperm;
    const int* ord = P.perms + (size_t)pi * P.A;
    int j = 0;
    for (int q = 0; q < P.A; q++) if (ord[q] == i) j = q;
    // zero the whole record (plans, counters, trigger overlap set) except the TelemetryViewer arrays, which the
    // reference never resets with the game (the viewer notices the lower lap count itself)
    const int t_laps = a->tele_completed_laps, t_step = a->tele_lap_end_step;
    const float t_last = a->tele_last_lap, t_best = a->tele_best_lap, t_total = a->tele_total_time;
    __builtin_memset(a, 0, sizeof(hk_agent_state));     // (not a uint32_t* loop: that would violate type-based aliasing)
    a->tele_completed_laps = t_laps; a->tele_lap_end_step = t_step;
    a->tele_last_lap = t_last; a->tele_best_lap = t_best; a->tele_total_time = t_total;
    // expSectionChoices {0, 0, 1, 1} / expLaneChoices {2, 3, 2, 3} (REC:526-527), continued row by row for more than 4 agents
    int sec = j >> 1;
    int lane = 2 + (j & 1);
    float spawn = 3.0f, acc0 = P.init_acc;
    if (TRAIN && P.training_reset) {                         // Training mode: random scatter (REC:520-668)
        float twp = 0.25f;
        training_layout(P, T, env, experiment_num, episodes_done, ord, i, sec, lane, twp, spawn);
        acc0 = -P.st.TireWearRate * hk_logf(1 - ((P.st.MaxSteer - P.st.MinSteer) * twp / P.st.MaxSteer));
    }
    a->section_index = sec;
    a->init_checkpoint_index = sec;
    a->acc_ang_v = acc0;
    a->lane = lane;
    const SecDev& s = T.sec[mod_L(P, sec)];
    float yaw = s.yaw_rad;
    float px = s.lane_x[lane - 1] + s.fx * spawn;
    float pz = s.lane_z[lane - 1] + s.fz * spawn;
    if (P.jitter_seed != 0u) {
        uint32_t r[4];
        philox4x32((uint32_t)experiment_num, (uint32_t)i, 0u, 0u, P.jitter_seed + (uint32_t)(P.env_id_base + env), 0u, r);
        px += (2.0f * u01(r[0]) - 1.0f) * P.jitter_pos;
        pz += (2.0f * u01(r[1]) - 1.0f) * P.jitter_pos;
        yaw += (2.0f * u01(r[2]) - 1.0f) * P.jitter_yaw;
        if (yaw < 0.0f) yaw += TWO_PI_F;
        if (yaw >= TWO_PI_F) yaw -= TWO_PI_F;
    }
    a->px = px; a->pz = pz; a->yaw = yaw;
    a->final_steer = kart_steer(P, a->acc_ang_v);
    if (TRAIN && P.training_agent[i]) plan_randomly(P, T, env, i, a->section_index, 0, episodes_done, a);     // Mode == Training (HKA:101-103)
    else if (P.high_mode[i] == HK_HIGH_FIXED) plan_fixed(P, T, i, a->section_index, a);
    a->flags = HK_F_ACTIVE | HK_F_ENABLED;
}

// REC.ResetGame :679-702 for agent i — "Generate times for reaching certain sections": made-up, descending section times (int
// Random.Range(earliest, 0), here Philox keyed by mcts_seed / env / agent / episode / section) for the sections between the rearmost
// kart's and its own, 0 for its own (:697).  planWithMCTS reads them as the time offsets of karts behind the furthest one
// (HKA:221-224).  The agentsPastSection counts of the same loop are dead — ApplySectionRewardsAndPenalties reads a count only where
// minSectionTimes holds the section, and the statement that creates that key assigns the count (REC:366-370) — and not restated.
template <bool TRAIN>
__device__ inline void mcts_backfill_section_times(const EnvParams& P, const TabView& T, int env, int i, int experiment_num, int episodes_done,
                                                   int own_section, hk_mcts_state* m)
{
    int back = 0;                                           // Experiment / Race grid: slot 0 stands in section 0
    if (TRAIN && P.training_reset) {
        const int pi = ((experiment_num % P.nperm) + P.nperm) % P.nperm;
        const int* ord = P.perms + (size_t)pi * P.A;
        back = own_section;
        for (int q = 0; q < P.A; q++) {
            int sec = 0, lane = 0; float twp = 0.0f, spawn = 0.0f;
            training_layout(P, T, env, experiment_num, episodes_done, ord, q, sec, lane, twp, spawn);
            back = sec < back ? sec : back;
        }
    }
    int earliest = -P.max_steps;
    for (int tp = back; tp < own_section; tp++) {
        uint32_t r[4];
        philox4x32((uint32_t)tp, (uint32_t)i, (uint32_t)episodes_done, 0x53454354u, P.mcts_seed, (uint32_t)(P.env_id_base + env), r);
        const int v = earliest + (int)(((unsigned long long)r[0] * (unsigned long long)(uint32_t)(-earliest)) >> 32);   // int Random.Range(earliest, 0)
        m->sec_time[tp & (HK_MCTS_SECTIME_RING - 1)] = v;
        earliest = v;
    }
    m->sec_time[own_section & (HK_MCTS_SECTIME_RING - 1)] = 0;
}

__device__ inline void snapshot_result(const Hot& h, float cum_reward, float group_reward, hk_episode_result* r, int episode)
{
    r->time_steps = h.time_steps;
    r->section_index = h.section_index;
    r->illegal_lane_changes = h.illegal_lane_changes;
    r->forward_collisions = h.forward_collisions;
    r->avg_lane_diff = h.avg_lane_diff;
    r->avg_vel_diff = h.avg_vel_diff;
    r->reward = cum_reward;
    r->episode = episode;
    r->last_lap = h.tele_last_lap; r->best_lap = h.tele_best_lap; r->total_time = h.tele_total_time;
    r->laps_completed = h.tele_completed_laps; r->lap_end_step = h.tele_lap_end_step;
    r->speed = mag3(h.vx, 0.0f, h.vz);
    r->active = (h.flags & HK_F_ACTIVE) ? 1 : 0;
    r->group_reward = group_reward;
}

// explicit hk_reset
__global__ __launch_bounds__(256) void env_reset_kernel(EnvParams P, hk_agent_state* agents, uint32_t* hot, const int* slot_of, hk_env_state* envs_by_slot,
                                                        const int* env_ids, int n, int experiment_num, MctsDev M, int set, RwDev RD, int* status)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid == 0) atomicAnd(status, ~4);       // (see env_arm_kernel)
    const int idx = gid / GA, i = gid % GA;
    if (idx >= n || i >= P.A) return;
    const int env = env_ids ? env_ids[idx] : idx;
    const int slot = slot_of[env];
    hk_env_state* const esp = &envs_by_slot[slot];             // the env's words live at its slot
    const int ex = experiment_num >= 0 ? experiment_num : (P.env_id_base + env) % P.nperm;
    const TabView T = tab_view(P, P.tab);
    hk_agent_state* ar = &agents[(size_t)env * P.A + i];
    uint32_t* const htile = hot + hot_base<GA>(slot, i);
    // the AoS record is the staging copy of the reset (reset_agent rewrites all of it, plans included, and keeps the telemetry fields it finds)
    store_hot(ar, load_hot_tile(htile));
    const float old_steer = ar->final_steer;        // m_FinalStats.Steer as the previous episode left it (mcts_post_request)
    reset_agent<true>(P, T, env, i, ex, esp->episodes_done, ar);
    store_hot_tile(htile, load_hot(ar));
    if (RD.sec_time) {
        const int nn = P.A * RD.S;
        for (int q = i; q < nn; q += P.A) { RD.sec_time[(size_t)env * nn + q] = -1; RD.sec_cnt[(size_t)env * nn + q] = 0; }
    }
    if (M.st) {
        mcts_reset_state(&M.st[(size_t)env * P.A + i]);
        mc_reqs(M)[(size_t)env * P.A + i].last_sec = -1;
        mcts_backfill_section_times<true>(P, T, env, i, ex, esp->episodes_done, ar->section_index, &M.st[(size_t)env * P.A + i]);
        uint32_t req = 0;
        for (int e = 0; e < P.A; e++) if (P.high_mode[e] == HK_HIGH_MCTS && !P.training_agent[e]) req |= 1u << e;
        mcts_post_request(P, M, set, env, i, req, req, 0, esp->episodes_done, P.mcts_iter0, P.mcts_lat0,
                          ar->section_index, ar->lane, ar->lane_changes, ar->final_steer, old_steer);
    }
    if (i == 0) {
        hk_env_state* es = esp;
        es->experiment_num = ex;
        es->status = 0;
        es->initial_started = 1;
        es->episode_steps = 0;
        es->inactive_mask = 0;
        es->reserved[0] = 0; es->reserved[1] = 0;      // an env left mid-tick by a failed hk_step starts clean
    }
}

// ---------------------------------------------------------------------------------------------------------------
// phase A of a tick: REC.FixedUpdate (:239-311), StartRaceAfterDelay (:721-744), KA.FixedUpdate forward-collision
// rays (:135-167).  `es` is the quad's register copy of the env words (identical in the 4 lanes).  Returns true when the
// env is parked (auto_reset off and the episode is over): nothing else happens on this tick.
// ---------------------------------------------------------------------------------------------------------------
template <bool RWT, bool TRAIN>
__device__ inline bool phase_begin(const EnvParams& P, const int env, const int i, const bool env_ok, hk_env_state& es,
                                   Hot& h, float& hfx, float& hfz, hk_agent_state* agents, hk_episode_result* results,
                                   const MctsDev& M, const int set, const RwDev& RD, RwAcc& rwv, const int* act_branch)
{
    const bool RW = RWT && P.rewards != 0;     // the <true, true, true> instantiation also serves handles without rewards
    const bool me = env_ok && i < P.A;
    HK_LP(1);
    hk_agent_state* a = me ? &agents[(size_t)env * P.A + i] : nullptr;
    const uint32_t all_mask = (1u << P.A) - 1u;
    bool skip = !env_ok;       // envs that stay parked (auto_reset off and finished)
    bool finish = false, timeout = false;
    if (RW && me && !(!P.auto_reset && (es.inactive_mask & all_mask) == all_mask)) {
        // Academy step: OnActionReceived rewards on the state the previous tick left
        const TabView Tg = tab_view(P, P.tab);
        rw_academy(P, Tg, env, i, h, hfx, hfz, a, act_branch, rwv);
    }
    if (env_ok) {
        if ((es.inactive_mask & all_mask) == all_mask) {
            if (!P.auto_reset) {
                if (!(es.status & 4u)) {
                    if (me) snapshot_result(h, RW ? rwv.cum : a->cum_reward, RW ? rwv.group : a->group_reward, &results[(size_t)env * P.A + i], es.episodes_done);
                    es.episodes_done += 1; es.status |= 4u;
                }
                skip = true;
            } else finish = true;
        } else {
            es.episode_steps += 1;
            if (es.episode_steps >= P.max_steps) {
                if (!P.auto_reset) {
                    if (me) {
                        uint32_t fl = h.flags;
                        if (fl & HK_F_ACTIVE) h.flags = deactivate_fields(P, h, fl);
                        snapshot_result(h, RW ? rwv.cum : a->cum_reward, RW ? rwv.group : a->group_reward, &results[(size_t)env * P.A + i], es.episodes_done);
                    }
                    es.inactive_mask = all_mask;
                    es.episodes_done += 1; es.status |= 2u | 4u;
                    skip = true;
                } else { finish = true; timeout = true; }
            }
        }
    }
    if (finish) {
        if (me) {
            uint32_t fl = h.flags;
            if (fl & HK_F_ACTIVE) h.flags = deactivate_fields(P, h, fl);
        }
        if (es.initial_started || timeout) {
            if (me) snapshot_result(h, RW ? rwv.cum : a->cum_reward, RW ? rwv.group : a->group_reward, &results[(size_t)env * P.A + i], es.episodes_done);
            es.episodes_done += 1;
            es.status = (es.status & ~2u) | (timeout ? 2u : 0u);
            es.experiment_num += 1;
        }
        if (RW) {
            // AddGoalTimingRewards (REC:267 / :306) before ResetGame; then ResetGame clears the section tables (:508-512)
            rw_goal_timing(P, i, me ? h.time_steps : 0, me && (h.flags & HK_F_ENABLED), rwv);
            if (me) results[(size_t)env * P.A + i].group_reward = rwv.group;
            if (env_ok) {
                const int n = P.A * RD.S;
                for (int q = i; q < n; q += GA) { RD.sec_time[(size_t)env * n + q] = -1; RD.sec_cnt[(size_t)env * n + q] = 0; }
            }
            __threadfence();
            rwv.cum = 0.0f; rwv.step = 0.0f; rwv.group = 0.0f;      // the record is rewritten below (EndGroupEpisode)
        }
        const float old_steer = h.final_steer;      // m_FinalStats.Steer as the episode that just ended left it (mcts_post_request)
        if (me) {
            // REC.ResetGame rewrites the whole record (plans included): through memory, then back into registers
            store_hot(a, h);
            const TabView T = tab_view(P, P.tab);
            reset_agent<TRAIN>(P, T, env, i, es.experiment_num, es.episodes_done, a);
#ifdef HK_STAMPS
            { Hot nh = load_hot(a); nh.st_t = h.st_t; for (int k = 0; k < HK_NSTAMP; k++) nh.st_acc[k] = h.st_acc[k]; h = nh; }
#else
            h = load_hot(a);
#endif
            hk_sincosf(h.yaw, &hfx, &hfz);
        }
        es.episode_steps = 0;
        es.inactive_mask = 0;
        es.initial_started = 1;
        if (M.st) {
            // prepareForReuse + initialPlan (HKA:84-96, 428-452): planner state cleared, first plan requested (T = 1.5 s)
            if (me) {
                mcts_reset_state(&M.st[(size_t)env * P.A + i]); mc_reqs(M)[(size_t)env * P.A + i].last_sec = -1;
                const TabView T = tab_view(P, P.tab);
                mcts_backfill_section_times<TRAIN>(P, T, env, i, es.experiment_num, es.episodes_done, h.section_index, &M.st[(size_t)env * P.A + i]);
            }
            uint32_t req = 0;
            for (int e = 0; e < P.A; e++) if (P.high_mode[e] == HK_HIGH_MCTS && !P.training_agent[e]) req |= 1u << e;
            mcts_post_request(P, M, set, env, i, req, req, 0, es.episodes_done, P.mcts_iter0, P.mcts_lat0,
                              h.section_index, h.lane, h.lane_changes, h.final_steer, old_steer);
        }
    }
    // own pose / flags (after a possible reset)
    float px = 0, pz = 0, fx = 0, fz = 1;
    uint32_t fl = 0;
    if (me && !skip) { px = h.px; pz = h.pz; fx = hfx; fz = hfz; fl = h.flags; }
    // StartRaceAfterDelay
    if (me && !skip && es.episode_steps >= P.hold && (fl & HK_F_ACTIVE) && !(fl & HK_F_CAN_MOVE)) fl |= HK_F_CAN_MOVE;
    // KA.FixedUpdate: three rays against the other karts' capsules
    bool hitAgent = false;
    {
        const float ox = px + SENSOR_LZ * fx, oz = pz + SENSOR_LZ * fz;
        const int csens[3] = {0, 1, 5};
        const float clen[3] = {0.8f, 0.9f, 0.9f};
        float ddx[3], ddz[3];
#pragma unroll
        for (int q = 0; q < 3; q++) sensor_dir(P, csens[q], fx, fz, ddx[q], ddz[q]);
        for_each_lane([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const float jpx = group_get<j>(px), jpz = group_get<j>(pz), jfx = group_get<j>(fx), jfz = group_get<j>(fz);
            const uint32_t jfl = group_get<j>(fl);
            if (j >= P.A || j == i || !(jfl & HK_F_ENABLED)) return;
            // exact cull: a 0.9 m ray from 0.1 m ahead of this kart cannot reach a capsule whose origin is > 2.2 m away
            // (0.1 + 0.9 + core half length 0.657 + slice radius 0.4425 = 2.10)
            if ((jpx - px) * (jpx - px) + (jpz - pz) * (jpz - pz) > 2.2f * 2.2f) return;
            HK_LP(2);
#pragma unroll
            for (int q = 0; q < 3; q++) {
                float t = ray_stadium(ox, oz, ddx[q], ddz[q], jpx, jpz, jfx, jfz, P.ray_agent_r);
                if (t >= 0.0f && t <= clen[q]) hitAgent = true;
            }
        });
    }
    if (me && !skip) {
        if (fl & HK_F_ENABLED) {
            const bool fc = (fl & HK_F_FORWARD_COLLISION) != 0;
            const int lct = h.last_collision_time;
            if (hitAgent && !fc && (lct == 0 || es.episode_steps - lct > 75)) {
                fl |= HK_F_FORWARD_COLLISION; h.forward_collisions += 1; h.last_collision_time = es.episode_steps;
            } else if (hitAgent) {
                fl |= HK_F_FORWARD_COLLISION; h.last_collision_time = es.episode_steps;
            } else {
                fl &= ~HK_F_FORWARD_COLLISION;
            }
        }
        h.flags = fl;
        if (RW && (fl & HK_F_ENABLED)) rw_not_at_goal(P, h, rwv);     // KA:165
    }
    return skip;
}

// ---------------------------------------------------------------------------------------------------------------
// phase C: RL actions, planFixed (HKA:331-355), ArcadeKart.FixedUpdate (AK:243-503), engine restatement, triggers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rot_y(float ang_rad, float& x, float& z)
{
    float c, s;
    hk_sincosf_near0(ang_rad, &s, &c);        // a few degrees per tick: the reduction-free path of the same function
    float nx = x * c + z * s;
    float nz = z * c - x * s;
    x = nx; z = nz;
}

__device__ inline int calculate_lane(const EnvParams& P, const SecDev& s, float px, float pz)
{   // DPT.CalculateLane :116-148
    float dy = P.kart_y - s.marker_y;
    float d[4];
#pragma unroll
    for (int l = 0; l < 4; l++) d[l] = mag3(px - s.lane_x[l], dy, pz - s.lane_z[l]);
    float mn = f_min(f_min(d[0], d[1]), f_min(d[2], d[3]));
#pragma unroll
    for (int l = 0; l < 4; l++) if (mn == d[l]) return l + 1;
    return -1;
}

// phase C of a tick (after every ego's controls are known)
template <bool RWT, bool TRAIN>
__device__ inline void phase_move(const EnvParams& P, const TabView& T, const int env, const int i, const bool env_ok,
                                  hk_env_state& es, Hot& h, float& hfx, float& hfz, hk_agent_state* agents, const float* act_steer,
                                  const int* act_branch, hk_mcts_state* mcts_all, const RwDev& RD, RwAcc& rwv, const int my_low_mode,
                                  const int my_high_mode)
{
    const bool RW = RWT && P.rewards != 0;
    const bool me = env_ok && i < P.A;
    const int episode_steps = es.episode_steps;
    const uint32_t inactive_mask = es.inactive_mask, status = es.status;
    const uint32_t all_mask = (1u << P.A) - 1u;
    // parked env (auto_reset off, finished): nothing moves
    const bool parked = env_ok && !P.auto_reset && (inactive_mask & all_mask) == all_mask && (status & 4u);
    hk_agent_state* a = me ? &agents[(size_t)env * P.A + i] : nullptr;
    uint32_t fl = 0;
    float px = 0, pz = 0, yaw = 0, vx = 0, vz = 0, wy = 0;
    float cfx = hfx, cfz = hfz;          // forward of the kart (sin yaw, cos yaw), refreshed after the integration
    bool live = me && !parked;
    if (live) { fl = h.flags; px = h.px; pz = h.pz; yaw = h.yaw; vx = h.vx; vz = h.vz; wy = h.wy; }
    const bool enabled = live && (fl & HK_F_ENABLED);
    const bool inactive_before = (inactive_mask >> i) & 1u;
    HK_LP(12);
    if (enabled) {
        // KA.OnActionReceived / InterpretDiscreteActions (HKA:1371-1379) for RL agents
        if (my_low_mode == HK_LOW_RL && (fl & HK_F_ACTIVE)) {
            h.steering = act_steer[(size_t)env * P.A + i];
            int br = act_branch[(size_t)env * P.A + i];
            if (br > 1) fl |= HK_F_ACCEL; else fl &= ~HK_F_ACCEL;
            if (br < 1) fl |= HK_F_BRAKE; else fl &= ~HK_F_BRAKE;
        }
        if (episode_steps % 100 == 0 && episode_steps < P.max_steps && episode_steps > 0 && !inactive_before) {
            HK_LP(21);
            if (TRAIN && P.training_agent[i]) plan_randomly(P, T, env, i, h.section_index, episode_steps, es.episodes_done, a);   // HKA:357-360
            else if (my_high_mode == HK_HIGH_FIXED) plan_fixed(P, T, i, h.section_index, a);
        }
        // ---- ArcadeKart.FixedUpdate
        bool accelerate = false, brake = false; float turnInput = 0.0f;
        if (fl & HK_F_ACTIVE) { accelerate = (fl & HK_F_ACCEL) != 0; brake = (fl & HK_F_BRAKE) != 0; turnInput = h.steering; }
        float acc_ang_v = h.acc_ang_v;
        const float final_steer = kart_steer(P, acc_ang_v);               // UpdateStats AK:295-302
        h.final_steer = final_steer;
        {   // KartAnimation.FixedUpdate (KartAnimation.cs:54-63, after ArcadeKart's): the front WheelColliders' steerAngle / maxSteeringAngle
            const float maxDelta = P.eng.steer_damping * P.dt;
            float ss = h.steer_smoothed;
            if (f_abs(turnInput - ss) <= maxDelta) ss = turnInput;
            else ss = ss + f_sign(turnInput - ss) * maxDelta;
            h.steer_smoothed = ss;
        }
#ifdef HK_DUMMY_NO_MOVE
        if (false) {
#else
        if (fl & HK_F_CAN_MOVE) {                                         // MoveVehicle AK:363-503
#endif
            HK_LP(13);
            const float dt = P.dt;
            const float fx = cfx, fz = cfz;
            float accelInput = (accelerate ? 1.0f : 0.0f) - (brake ? 1.0f : 0.0f);
            float localVelZ = vx * fx + vz * fz;
            bool accelDirectionIsFwd = accelInput >= 0;
            bool localVelDirectionIsFwd = localVelZ >= 0;
            // (kernel-argument fields are read into values BEFORE a conditional picks one: `c ? P.a : P.b` is a conditional between two
            // lvalues, i.e. a select of the ADDRESS and one per-lane global_load from the argument segment, on every tick)
            const float st_top = P.st.TopSpeed, st_rev = P.st.ReverseSpeed, st_acc = P.st.Acceleration, st_racc = P.st.ReverseAcceleration;
            float maxSpeed = localVelDirectionIsFwd ? st_top + 0.0f : st_rev + 0.0f;
            float wear = tire_wear(P, final_steer);
            float maxAllowedSpeed = sqrtf(max_lat_gs(P, wear) * 9.81f * f_abs(turning_radius(vx, vz, fx, fz, wy)));
            if (!(isinf(maxAllowedSpeed) || isnan(maxAllowedSpeed)))
                maxSpeed = f_clamp(maxSpeed, 0.001f, f_max(maxAllowedSpeed, 0.001f));
            float accelPower = accelDirectionIsFwd ? st_acc + 0.0f : st_racc + 0.0f;
            float currentSpeed = mag3(vx, 0.0f, vz);
            float accelRampT = currentSpeed / maxSpeed;
            float multipliedAccelerationCurve = P.st.AccelerationCurve * 5;
            float tt = accelRampT * accelRampT;
            tt = f_clamp(tt, 0.0f, 1.0f);
            float accelRamp = multipliedAccelerationCurve + (1 - multipliedAccelerationCurve) * tt;
            bool isBraking = (localVelDirectionIsFwd && brake) || (!localVelDirectionIsFwd && accelerate);
            float finalAccelPower = isBraking ? P.st.Braking : accelPower;
            float finalAcceleration = finalAccelPower * accelRamp;
            float turningPower = turnInput * final_steer * (f_abs(currentSpeed) > 0.5f ? 1.0f : 0.0f);
            float fwx = fx, fwz = fz;
            rot_y(turningPower * DEG2RAD_F, fwx, fwz);
            float accx = fwx * accelInput * finalAcceleration * 1.0f;
            float accz = fwz * accelInput * finalAcceleration * 1.0f;
            bool wasOverMaxSpeed = currentSpeed >= maxSpeed;
            if (wasOverMaxSpeed && !isBraking) { accx *= 0.0f; accz *= 0.0f; }
            float nvx = vx + accx * dt, nvz = vz + accz * dt;
            if (wasOverMaxSpeed) {
                float sq = nvx * nvx + nvz * nvz;
                if (sq > maxSpeed * maxSpeed) {
                    float mg = sqrtf(sq);
                    nvx = (nvx / mg) * maxSpeed; nvz = (nvz / mg) * maxSpeed;
                }
            }
            if (f_abs(accelInput) < 0.01f) {
                float maxDelta = dt * P.st.CoastingDrag;
                float tx = 0.0f - nvx, tz = 0.0f - nvz;
                float sq = tx * tx + tz * tz;
                if (sq == 0.0f || sq <= maxDelta * maxDelta) { nvx = 0.0f; nvz = 0.0f; }
                else { float d = sqrtf(sq); nvx = nvx + tx / d * maxDelta; nvz = nvz + tz / d * maxDelta; }
            }
            vx = nvx; vz = nvz;
            float angularVelocitySteering = 0.4f;
            if (!localVelDirectionIsFwd && !accelDirectionIsFwd) angularVelocitySteering *= -1.0f;
            {
                float target = turningPower * angularVelocitySteering, maxDelta = dt * 20.0f;
                if (f_abs(target - wy) <= maxDelta) wy = target;
                else wy = wy + f_sign(target - wy) * maxDelta;
            }
            acc_ang_v += f_abs(wy);
            h.acc_ang_v = acc_ang_v;
            rot_y(turningPower * f_sign(localVelZ) * 25.0f * P.st.Grip * dt * DEG2RAD_F, vx, vz);
            // ---- engine: the wheels' sideways friction, then integrate
#ifndef HK_DUMMY_NO_ENGINE          /* HK_DUMMY_*: region-cost experiments only (tools/experiments/region_cost.py); never in the product */
            if (P.eng.wheel_friction)
#else
            if (false)
#endif
                engine_wheels(P, cfx, cfz, h.steer_smoothed, vx, vz, wy, h.wheel_uf, h.wheel_ur);
            wy = wy * (1.0f - P.st.AngularDrag * dt);
            yaw = yaw + wy * dt;
            if (yaw < 0.0f) yaw += TWO_PI_F;
            if (yaw >= TWO_PI_F) yaw -= TWO_PI_F;
            px = px + vx * dt;
            pz = pz + vz * dt;
            hk_sincosf(yaw, &cfx, &cfz);
        }
    }
    HK_ST(h, 8);                       // [8] actions, planFixed, ArcadeKart model, integration
    // ---- kart-kart contacts (Jacobi over one snapshot)
#ifndef HK_DUMMY_NO_KART
    {
        float ax = 0, az = 0, bx = 0, bz = 0;
        kart_core(cfx, cfz, px, pz, ax, az, bx, bz);
        float cpx = 0, cpz = 0, cvx = 0, cvz = 0, cwy = 0;
        bool touched = false;
        for_each_lane([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const float jpx = group_get<j>(px), jpz = group_get<j>(pz), jfx = group_get<j>(cfx), jfz = group_get<j>(cfz);
            const float jvx = group_get<j>(vx), jvz = group_get<j>(vz), jwy = group_get<j>(wy);
            const uint32_t jfl = group_get<j>(fl);
            if (j >= P.A || j == i || !enabled || !(jfl & HK_F_ENABLED)) return;
            // exact cull: two capsules (reach 0.657 + 0.45 from their origins) cannot touch when the origins are > 2.3 m apart
            if ((jpx - px) * (jpx - px) + (jpz - pz) * (jpz - pz) > 2.3f * 2.3f) return;
            HK_LP(14);
            float cx, cz, dx, dz, c1x, c1z, c2x, c2z;
            kart_core(jfx, jfz, jpx, jpz, cx, cz, dx, dz);
            float d2 = seg_seg_closest(ax, az, bx, bz, cx, cz, dx, dz, c1x, c1z, c2x, c2z);
            const float rr = 2.0f * CAP_R;
            if (d2 < rr * rr) {
                float d = sqrtf(d2);
                float nx, nz;
                if (d > 1e-6f) { nx = (c1x - c2x) / d; nz = (c1z - c2z) / d; }
                else {
                    float ex = px - jpx, ez = pz - jpz;
                    float el = sqrtf(ex * ex + ez * ez);
                    if (el > 1e-6f) { nx = ex / el; nz = ez / el; } else { nx = (i < j) ? 1.0f : -1.0f; nz = 0.0f; }
                }
                float pen = rr - d;
                float share = (jfl & HK_F_CAN_MOVE) ? 0.5f : 1.0f;
                cpx += nx * (pen * share); cpz += nz * (pen * share);
                if (!P.eng.contact_yaw) {
                    float vrel = (vx - jvx) * nx + (vz - jvz) * nz;
                    if (vrel < 0.0f) { cvx -= nx * (vrel * share); cvz -= nz * (vrel * share); }
                } else {
                    // frictionless inelastic contact of two free rigid bodies at the surface points (c1 - R n on this kart, c2 + R n on j)
                    const float inv_m = P.engd.inv_m, inv_i = P.engd.inv_i;
                    const float rix = (c1x - nx * CAP_R) - px, riz = (c1z - nz * CAP_R) - pz;
                    const float rjx = (c2x + nx * CAP_R) - jpx, rjz = (c2z + nz * CAP_R) - jpz;
                    const float ki = riz * nx - rix * nz, kj = rjz * nx - rjx * nz;
                    float vrel = (vx - jvx) * nx + (vz - jvz) * nz + wy * ki - jwy * kj;
                    if (vrel < 0.0f) {
                        float den = inv_m + ki * ki * inv_i;
                        if (jfl & HK_F_CAN_MOVE) den = den + (inv_m + kj * kj * inv_i);
                        const float jn = -vrel / den;
                        cvx += nx * (jn * inv_m); cvz += nz * (jn * inv_m); cwy += jn * ki * inv_i;
                    }
                }
                touched = true;
            }
        });
        if (live) {
            if (touched && (fl & HK_F_CAN_MOVE)) { px += cpx; pz += cpz; vx += cvx; vz += cvz; wy += cwy; }
            if (touched) fl |= HK_F_HAS_COLLISION; else fl &= ~HK_F_HAS_COLLISION;
        }
    }
#endif
    HK_ST(h, 9);                       // [9] kart-kart contacts
    // ---- kart-wall contacts: deepest penetration, two passes
#ifndef HK_DUMMY_NO_WALL
    if (enabled && (fl & HK_F_CAN_MOVE)) {
        for (int pass = 0; pass < 2; pass++) {
            HK_LP(15);
            float ax, az, bx, bz;
            kart_core(cfx, cfz, px, pz, ax, az, bx, bz);
            // a contact needs a wall within CAP_R of the core, i.e. within 1.11 m of the kart origin: the cell's near list
            const int cell = grid_cell(P, px, pz);
            const int w0 = T.grid_off[cell], w1 = w0 + T.near_cnt[cell];
            float bestpen = 0.0f, bnx = 0.0f, bnz = 0.0f, bcx = 0.0f, bcz = 0.0f;
            bool found = false;
            // Two passes over the cell's list, 32 walls at a time: a cheap bounding-box test marks the walls that can be within
            // CAP_R of the core at all (a bit per wall), then only those get the closest-point computation, in ascending list
            // order.  The lanes of a wave sit in different cells: with the full test on every listed wall the wave paid the
            // longest list (a curve's ~7 one-metre segments) times the full test on every tick, although almost no kart
            // touches a wall.  The box test is conservative (1 cm margin >> float rounding), so the result is unchanged.
            // (round 4: the cull is the wall's distance from the kart ORIGIN — every point of the capsule lies within 0.657 + 0.45 = 1.107 m of it —
            // not a bounding-box overlap: the one-metre diagonal segments of a curve's inner wall, 1.2 - 1.5 m from a kart on the racing line, passed
            // the box test on most ticks and went through the closest-point computation for nothing)
            for (int base = w0; base < w1; base += 32) {
                const int nq = (w1 - base) < 32 ? (w1 - base) : 32;
                uint32_t cand = 0;
                for (int q = 0; q < nq; q += 4) {
                    HK_LP(16);
                    // four independent index -> wall load chains in flight (the LDS round trips, not the arithmetic, are what
                    // this pass costs); slots past the end re-read the last wall and are masked out
                    hk_wall_seg ws[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) ws[j] = T.walls[T.grid_idx[base + ((q + j) < nq ? (q + j) : (nq - 1))]];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const bool apart = !wall_within(ws[j], px, pz, 1.107f + 0.02f);
                        cand |= ((apart || (q + j) >= nq) ? 0u : 1u) << (q + j);
                    }
                }
                while (cand) {
                    const int q = __ffs((int)cand) - 1;
                    cand &= cand - 1u;
                    HK_LP(17);
                    const hk_wall_seg ws = T.walls[T.grid_idx[base + q]];
                    float c1x, c1z, c2x, c2z;
                    float d2 = seg_seg_closest(ax, az, bx, bz, ws.x0, ws.z0, ws.x1, ws.z1, c1x, c1z, c2x, c2z);
                    if (d2 < CAP_R * CAP_R) {
                        float d = sqrtf(d2);
                        float pen = CAP_R - d;
                        float nx, nz;
                        if (d > 1e-6f) { nx = (c1x - c2x) / d; nz = (c1z - c2z) / d; }
                        else {
                            float ex = ws.x1 - ws.x0, ez = ws.z1 - ws.z0;
                            float el = sqrtf(ex * ex + ez * ez);
                            nx = -ez / el; nz = ex / el;
                            if ((px - ws.x0) * nx + (pz - ws.z0) * nz < 0.0f) { nx = -nx; nz = -nz; }
                        }
                        // the near part of grid_idx is ascending per cell, so "first strictly deeper" == the oracle's lowest-index tie break
                        if (!found || pen > bestpen) { found = true; bestpen = pen; bnx = nx; bnz = nz; bcx = c1x; bcz = c1z; }
                    }
                }
            }
            if (!found) break;
            if (!P.eng.contact_yaw) {
                float vn = vx * bnx + vz * bnz;
                if (vn < 0.0f) { vx -= bnx * vn; vz -= bnz * vn; }
            } else {
                // frictionless inelastic contact at the capsule's surface point c1 - R n (lever about the centre of mass = the kart origin,
                // taken before the push-out): the impulse stops the POINT's approach and turns the kart along the wall
                const float inv_m = P.engd.inv_m, inv_i = P.engd.inv_i;
                const float rx = (bcx - bnx * CAP_R) - px, rz = (bcz - bnz * CAP_R) - pz;
                const float k = rz * bnx - rx * bnz;
                const float vn = vx * bnx + vz * bnz + wy * k;
                if (vn < 0.0f) {
                    const float jn = -vn / (inv_m + k * k * inv_i);
                    vx += bnx * (jn * inv_m); vz += bnz * (jn * inv_m); wy += jn * k * inv_i;
                }
            }
            px += bnx * bestpen; pz += bnz * bestpen;
            fl |= HK_F_HAS_COLLISION;
            h.contact_nx = bnx; h.contact_nz = bnz;
        }
    }
#endif
    HK_ST(h, 10);                      // [10] kart-wall contacts
    bool bad = false;
    if (live) bad = !f_finite(px) || !f_finite(pz) || !f_finite(vx) || !f_finite(vz) || !f_finite(yaw) || !f_finite(wy);
    // ---- trigger dispatch (HKA.OnTriggerEnter :611-675) on the post-contact pose
    uint32_t newly_inactive = 0;
    RwEvent ev[RW_MAX_EVENTS];
    int nev = 0;
    if (enabled) {
        h.px = px; h.pz = pz; h.yaw = yaw; h.vx = vx; h.vz = vz; h.wy = wy;
#ifndef HK_DUMMY_NO_TRIG
        float ax, az, bx, bz;
        kart_core(cfx, cfz, px, pz, ax, az, bx, bz);
        uint32_t lo = 0, hi = 0;
        // only the Triggers listed for the kart's coarse cell can be within reach (the others fail the distance cull below
        // by construction): a couple of trips instead of one per section, each a dependent LDS round trip
        const unsigned tcand = trig_candidates_tight(P, T, px, pz);
        HK_ST(h, 26);                  // [26] (of [11]) finite checks, capsule core, candidate cell
        auto trig_test = [&](const int t, const float tx, const float tz, const float tfx, const float tfz) {
            HK_LP(18);
            // exact cull: box half diagonal 5.03 + capsule reach 1.11 < 6.5
            if ((px - tx) * (px - tx) + (pz - tz) * (pz - tz) > 6.5f * 6.5f) return;
#ifndef HK_NO_TRIG_SLAB_CULL
            // exact cull: the box is 1 m thick along its forward axis and every point of the capsule lies within 1.107 m of the kart
            // origin, so an origin further than 0.5 + 1.107 (+ 1 cm >> float rounding) from the mid plane cannot overlap — true on most
            // of the ticks that pass the distance cull (a kart is within 6.5 m of the next Trigger for a third of every section)
            if (f_abs((px - tx) * tfx + (pz - tz) * tfz) > TRIG_HZ + 1.117f) return;
#endif
            HK_LP(19);
            float rax = ax - tx, raz = az - tz, rbx = bx - tx, rbz = bz - tz;
            float lax = rax * tfz - raz * tfx, laz = rax * tfx + raz * tfz;
            float lbx = rbx * tfz - rbz * tfx, lbz = rbx * tfx + rbz * tfz;
            float zlo = f_min(laz, lbz) - CAP_R, zhi = f_max(laz, lbz) + CAP_R;
            float xlo = f_min(lax, lbx) - CAP_R, xhi = f_max(lax, lbx) + CAP_R;
            if (zlo <= TRIG_HZ && zhi >= -TRIG_HZ && xlo <= TRIG_HX && xhi >= -TRIG_HX) {
                if (t < 32) lo |= 1u << t; else hi |= 1u << (t - 32);
            }
        };
        if (tcand == 0xFEFEu) {
            // (a cell that meets more than two Triggers: the coarse masks, every candidate in section order)
            const uint2 tc = trig_candidates(P, T, px, pz);
#pragma unroll 1
            for (int half = 0; half < 2; half++) {
                uint32_t bits = half ? tc.y : tc.x;
                while (bits) {
                    const int t = (__ffs((int)bits) - 1) + 32 * half;
                    bits &= bits - 1u;
                    const SecDev& s = T.sec[t];
                    trig_test(t, s.trig_x, s.trig_z, s.fx, s.fz);
                }
            }
        } else if (tcand != 0xFFFFu) {
            const int t0 = (int)(tcand & 0xFFu), t1 = (int)(tcand >> 8);
            { const SecDev& s = T.sec[t0]; trig_test(t0, s.trig_x, s.trig_z, s.fx, s.fz); }
            if (t1 != 0xFF) { const SecDev& s = T.sec[t1]; trig_test(t1, s.trig_x, s.trig_z, s.fx, s.fz); }
        }
        HK_ST(h, 27);                  // [27] (of [11]) Trigger overlap tests
        const uint32_t nlo = lo & ~h.trig_lo, nhi = hi & ~h.trig_hi;
        h.trig_lo = lo; h.trig_hi = hi;
#ifdef HK_DUMMY_NO_ENTER
        if (false) {
#else
        if (nlo | nhi) {
#endif
            uint32_t elo = nlo, ehi = nhi;
            while (elo | ehi) {                                        // newly entered Triggers in section order
                int t;
                if (elo) { t = __ffs((int)elo) - 1; elo &= elo - 1u; } else { t = (__ffs((int)ehi) - 1) + 32; ehi &= ehi - 1u; }
                if (!(fl & HK_F_ACTIVE)) continue;
                HK_LP(20);
                // this Trigger's plan entry, asked for before anything else: the two global loads used to sit behind each other and behind
                // the section search, and the wave waited for both (two lanes of it are here on most ticks)
#ifdef HK_DUMMY_NO_PLANMEM              // (timing experiments only: wrong results)
                const int pl_t = 1 + (t & 1);
                const float pv_t = 10.0f;
#else
                const int pl_t = a->plan_lane[t];
                const float pv_t = a->plan_vel[t];
#endif
                const int L = P.L, H = P.H;
                const int sec = h.section_index, init = h.init_checkpoint_index;
                int index = -1, lane = -1;
                int lo_i = sec - H; if (lo_i < init) lo_i = init;
                // KA.FindSectionIndex :348-364: the first ii in [lo_i, sec + H) whose section (ii mod L) is this Trigger's.  The loop of the
                // C# costs a run-time modulo per step (typically seven steps) for the two lanes of a wave that are here: closed form.
                if (lo_i >= 0) {
                    int delta = t - mod_L(P, lo_i);
                    if (delta < 0) delta += L;
                    if (lo_i + delta < sec + H) index = lo_i + delta;
                } else {
                    for (int ii = lo_i; ii < sec + H; ii++) {
                        int idx = ii < 0 ? ii + L : ii;
                        if (idx % L == t) { index = idx; break; }
                    }
                }
                if (index != -1) lane = calculate_lane(P, T.sec[t], px, pz);        // (index mod L == t)
                const int secm = mod_L(P, sec);
                const bool sec_straight = T.sec[secm].inside_radius == 0.0f;       // is_straight(sec)
                if (index != -1 && ((index > sec) || (t == 0 && secm == L - 1))) {
                    const int key = t;
                    const int pl = pl_t;                                   // (key == t)
                    float lane_div = 1.0f, vel_div = 1.0f;                 // HKA:618-619
                    if (pl != 0) {
                        if (RW) rw_dividers(P, T, i, key, lane, pl, pv_t, px, pz, vx, vz, lane_div, vel_div);
                        float lmx, lmz;
                        lane_marker(T, key, pl, lmx, lmz);
                        float dist = mag3(px - lmx, P.kart_y - T.sec[key].marker_y, pz - lmz);
                        h.avg_lane_diff = (f_max(dist - 1.3f, 0.0f) + h.avg_lane_diff * (index - init - 1)) / (index - init);
                        float velocity = mag3(vx, 0.0f, vz);
                        h.avg_vel_diff = ((velocity - pv_t) + h.avg_vel_diff * (index - init - 1)) / (index - init);
#ifndef HK_DUMMY_NO_PLANMEM
                        a->plan_lane[key] = 0; a->plan_vel[key] = 0.0f;
#endif
                    }
                    const int cur_lane = h.lane;
                    int dl = cur_lane - lane; if (dl < 0) dl = -dl;
                    int lc = h.lane_changes;
                    const bool swerve = lc + dl > P.max_lane_changes && sec_straight;
                    if (swerve) h.illegal_lane_changes += 1;
                    if (RW && nev < RW_MAX_EVENTS) {
                        ev[nev].kind = 1; ev[nev].section = index; ev[nev].swerve = swerve ? 1 : 0;
                        ev[nev].lane_div = lane_div; ev[nev].vel_div = vel_div; nev++;
                    }
                    if (sec_straight != (T.sec[t].inside_radius == 0.0f)) lc = 0;
                    else if (cur_lane != lane) lc += dl;
                    h.lane_changes = lc;
                    h.section_index = index; h.lane = lane;
                    if (mcts_all) {
                        hk_mcts_state* mst = &mcts_all[(size_t)env * P.A + i];
                        mst->sec_time[index & (HK_MCTS_SECTIME_RING - 1)] = episode_steps;   // sectionTimes HKA:651
                        mst->root_live = 0; mst->root_cycles = 0;                          // currentRoot = null HKA:660-661
                    }
                    if (index == P.laps * L + 1) {                         // ReachGoalSection REC:469-474
                        h.time_steps = episode_steps;
                        fl = deactivate_fields(P, h, fl);
                        vx = 0; vz = 0; wy = 0;
                        newly_inactive |= 1u << i;
                    }
                } else if (index != -1 && ((index < sec) || (secm == 0 && t == L - 1))) {
                    if (RW && nev < RW_MAX_EVENTS) {
                        ev[nev].kind = 2; ev[nev].section = sec - index + 1; ev[nev].swerve = 0;
                        ev[nev].lane_div = 1.0f; ev[nev].vel_div = 1.0f; nev++;
                    }
                    h.section_index = index;
                    if (mcts_all) { hk_mcts_state* mst = &mcts_all[(size_t)env * P.A + i]; mst->root_live = 0; mst->root_cycles = 0; }   // HKA:668-669
                } else if (index == -1) {                                  // DroveReverseLimit REC:475-479
                    h.time_steps = P.max_steps * 6;
                    fl = deactivate_fields(P, h, fl);
                    vx = 0; vz = 0; wy = 0;
                    newly_inactive |= 1u << i;
                }
            }
        }
    #endif
    }
    HK_ST(h, 11);                      // [11] Trigger overlap / enter, section and lane rules
    hfx = cfx; hfz = cfz;
    if (RW && !parked) rw_replay_events(P, RD, env, i, episode_steps, ev, nev, enabled, live && (fl & HK_F_ENABLED), rwv);
    if (live) {
        h.flags = fl;
        // TelemetryViewer.Update :49-88 (once per tick)
        const int currentLap = div_L(P, h.section_index);
        const int done = h.tele_completed_laps;
        if (currentLap > done) {
            h.tele_completed_laps = currentLap;
            const float last = P.dt * (episode_steps - h.tele_lap_end_step);
            h.tele_last_lap = last;
            const float best = h.tele_best_lap;
            if (best < 10 || last < best) h.tele_best_lap = last;
            h.tele_lap_end_step = episode_steps;
        } else if (currentLap < done) {
            h.tele_completed_laps = currentLap;
            h.tele_last_lap = 0.0f; h.tele_best_lap = 0.0f; h.tele_lap_end_step = 0;
        }
        if (fl & HK_F_ACTIVE) h.tele_total_time = episode_steps * P.dt;
    }
    // per-env words: OR over the quad
    uint32_t ni = newly_inactive, bd = bad ? 1u : 0u;
    ni = (uint32_t)group_or((int)ni);
    bd = (uint32_t)group_or((int)bd);
    es.inactive_mask = inactive_mask | ni;      // identical in the lanes of the group
    if (bd) es.status = status | 1u;
    HK_ST(h, 12);                      // [12] telemetry, per-env words
}

} }  // namespace hk::HK_GA_NS
