// hk_env_training.h — Training mode of the episode controller: RacingEnvController.ResetGame scatters the karts at random
// (REC:520-668) and Mode == Training agents plan with HierarchicalKartAgent.planRandomly (HKA:109-143) instead of planFixed /
// MCTS.  The reference draws from UnityEngine.Random / System.Random / MathNet Normal (global streams shared by every env in
// the scene, so not reproducible outside Unity): here Philox-4x32 keyed by hk_config.train_seed, the global env id / agent row
// and the episode — "parity unpinned" (include/hk.h).  Resets are rare, so every lane of a quad simply recomputes the whole
// env's layout (the placement of one kart depends on the karts placed before it) and keeps its own row.  This code is compiled
// only into the env_run_kernel<true, true, true> instantiation: inlined into the headline kernel it cost 9 % (register pressure in
// a path that never runs there), and as noinline calls even more (call ABI spills).
// (included once per lane-group width by hk_env_ga.h: no include guard, namespace hk::HK_GA_NS)
#include "hk_env_device.h"

namespace hk { namespace HK_GA_NS {

struct TrRng { uint32_t k0, k1, c1, c2, c3, n; };
__device__ inline uint32_t tr_u32(TrRng& g) { uint32_t r[4]; philox4x32(g.n++, g.c1, g.c2, g.c3, g.k0, g.k1, r); return r[0]; }
__device__ inline int tr_range_i(TrRng& g, int lo, int hi)            // Random.Range(int, int): [lo, hi)
{
    if (hi <= lo) return lo;
    return lo + (int)(((unsigned long long)tr_u32(g) * (unsigned long long)(hi - lo)) >> 32);
}
__device__ inline float tr_range_f(TrRng& g, float lo, float hi) { return lo + u01(tr_u32(g)) * (hi - lo); }
__device__ inline float tr_normal(TrRng& g)
{
    uint32_t r[4]; philox4x32(g.n++, g.c1, g.c2, g.c3, g.k0, g.k1, r);
    const float u1 = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f), u2 = u01(r[1]);
    return sqrtf(-2.0f * hk_logf(u1)) * hk_cosf((2.0f * HK_PI_F) * u2);
}
__device__ inline float tr_gauss_bounded(TrRng& g, float mean, float sd, float lo, float hi)
{   // KartMCTS.NextGaussian(mean, sd, min, max) KM:225-240
    float x; int attempts = 0;
    do { x = mean + tr_normal(g) * sd; attempts += 1; } while ((x < lo || x > hi) && attempts < 10);
    if (attempts == 10 && (x < lo || x > hi)) return mean;
    return x;
}

// HKA.planRandomly :109-143
__device__ inline void plan_randomly(const EnvParams& P, const TabView& T, int env, int agent, int sec, int episode_steps,
                                     int episodes_done, hk_agent_state* a)
{
    TrRng g = {P.train_seed ^ 0x504C414Eu, (uint32_t)(P.env_id_base + env) * (uint32_t)P.A + (uint32_t)agent,
               (uint32_t)episode_steps, (uint32_t)episodes_done, 0u, 0u};
    int hi = sec + P.depth[agent]; if (hi > 1000) hi = 1000;
    for (int i = sec + 1; i < hi + 1; i++) {
        const int key = i % P.L;
        if (a->plan_lane[key] != 0) continue;
        const int index = (int)__builtin_rintf(f_abs(tr_gauss_bounded(g, 0.0f, 1.0f, -(float)4 + 1.0f, (float)4 - 1.0f)));
        const int ol = T.sec[(i - 1) % P.L].optimal_lane;
        const int sign = ol == 1 ? 1 : (ol == 4 ? -1 : 0);
        const int lane = sign < 0 ? 4 - index : 1 + index;          // Enumerable.Range(1, 4).OrderBy(l => sign * l)[index]
        a->plan_lane[key] = (uint8_t)lane;
        if (P.high_mode[agent] == HK_HIGH_FIXED) a->plan_vel[key] = P.max_speed;
        else a->plan_vel[key] = P.max_speed - f_abs(tr_gauss_bounded(g, 0.0f, 1.5f, -8.0f, 8.0f));
    }
}

// REC.ResetGame :520-668 (Training mode): section, lane, tire-wear proportion and spawn distance of agent `me`
__device__ inline void training_layout(const EnvParams& P, const TabView& T, int env, int experiment_num, int episodes_done,
                                       const int* ord, int me, int& sec_out, int& lane_out, float& twp_out, float& dist_out)
{
    TrRng g = {P.train_seed, (uint32_t)(P.env_id_base + env), (uint32_t)episodes_done, (uint32_t)experiment_num, 0x54524E47u, 0u};
    const int A = P.A, L = P.L, goal = P.laps * L + 1;
    const bool headToHead = tr_range_i(g, 0, 9) >= 3;
    int used_sec[GA], used_lane[GA], n_added = 0, initialSection = -1;
    for (int j = 0; j < A; j++) {
        const int i = ord[j];
        int s_i, l_i;
        float twp;
        if (!headToHead) {
            while (true) {
                s_i = tr_range_i(g, 0, goal);
                l_i = tr_range_i(g, 1, 5);
                bool clash = false;
                for (int q = 0; q < n_added; q++) clash = clash || (used_sec[q] == s_i % L && used_lane[q] == l_i);
                if (!clash) break;
            }
            twp = tr_range_f(g, 0.0f, 1.0f);
        } else if (n_added == 0) {
            s_i = tr_range_i(g, 0, goal);
            initialSection = s_i;
            twp = tr_range_f(g, 0.0f, 1.0f);
            l_i = tr_range_i(g, 1, 5);
        } else {
            const int lo = initialSection - 1 > 0 ? initialSection - 1 : 0, hi = initialSection + 2 < goal ? initialSection + 2 : goal;
            while (true) {
                s_i = tr_range_i(g, lo, hi);
                l_i = tr_range_i(g, 1, 5);
                bool clash = false;
                for (int q = 0; q < n_added; q++) clash = clash || (used_sec[q] == s_i % L && used_lane[q] == l_i);
                if (!clash) break;
            }
            twp = tr_range_f(g, 0.0f, 1.0f);
        }
        float d = tr_range_f(g, 1.0f, 4.0f);
        if (tr_range_f(g, 0.0f, 1.0f) < 0.3f) {                      // start close behind a wall: Physics.Raycast(marker, forward, 10)
            const SecDev& s = T.sec[s_i % L];
            float hit = -1.0f;
            for (int w = 0; w < P.NW; w++) {
                const float t = ray_seg(s.lane_x[l_i - 1], s.lane_z[l_i - 1], s.fx, s.fz, T.walls[w]);
                if (t >= 0.0f && t <= 10.0f && (hit < 0.0f || t < hit)) hit = t;
            }
            if (hit >= 0.0f) d = hit - 1.0f;
        }
        if (i == me) { sec_out = s_i; lane_out = l_i; twp_out = twp; dist_out = d; }
        used_sec[n_added] = s_i % L; used_lane[n_added] = l_i; n_added++;
    }
}

} }  // namespace hk::HK_GA_NS
