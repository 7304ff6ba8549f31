// hk_env_observe.h — HierarchicalKartAgent.CollectObservations (HKA:485-604): one thread per (env, agent).
// Layout = the order the reference calls sensor.AddObservation: 8 own, 12 per teammate, 12 per opponent,
// 5 per upcoming section (sectionHorizon), 9 ray distances (up to 20 m long: walked through the wall grid).
// (included once per lane-group width by hk_env_ga.h: no include guard, namespace hk::HK_GA_NS)
#include "hk_env_device.h"

namespace hk { namespace HK_GA_NS {

// what CollectObservations reads of a kart's per-tick state: a view of its hot-tile rows (hk_env_device.h)
struct KartObs {
    const uint32_t* p;
    __device__ __forceinline__ float px() const { return hot_get<float>(p, HF_px); }
    __device__ __forceinline__ float pz() const { return hot_get<float>(p, HF_pz); }
    __device__ __forceinline__ float vx() const { return hot_get<float>(p, HF_vx); }
    __device__ __forceinline__ float vz() const { return hot_get<float>(p, HF_vz); }
    __device__ __forceinline__ float yaw() const { return hot_get<float>(p, HF_yaw); }
    __device__ __forceinline__ float final_steer() const { return hot_get<float>(p, HF_final_steer); }
    __device__ __forceinline__ uint32_t flags() const { return hot_get<uint32_t>(p, HF_flags); }
    __device__ __forceinline__ int lane() const { return hot_get<int>(p, HF_lane); }
    __device__ __forceinline__ int lane_changes() const { return hot_get<int>(p, HF_lane_changes); }
    __device__ __forceinline__ int section_index() const { return hot_get<int>(p, HF_section_index); }
};

// (fx, fz) = the kart's forward (sin yaw, cos yaw): evaluated once per kart by the caller, not per use
__device__ inline float local_speed(const EnvParams& P, const KartObs a, float fx, float fz)
{   // AK:325-342
    if (!(a.flags() & HK_F_CAN_MOVE)) return 0.0f;
    const float vx = a.vx(), vz = a.vz();
    float dot = fx * vx + fz * vz;
    if (f_abs(dot) > 0.1f) {
        float speed = mag3(vx, 0.0f, vz);
        return dot < 0 ? -(speed / P.st.ReverseSpeed) : (speed / P.st.TopSpeed);
    }
    return 0.0f;
}

__device__ inline void inv_transform_point(const float apx, const float apz, float fx, float fz, float wx, float wy, float wz, float ky, float out[3])
{   // Transform.InverseTransformPoint of a yaw-only transform
    float rx = wx - apx, rz = wz - apz;
    out[0] = rx * fz + rz * (-fx);
    out[1] = wy - ky;
    out[2] = rx * fx + rz * fz;
}

// 16 lanes per agent: lanes 0-8 cast one sensor ray each (Physics.Raycast vs TrackMask and vs AgentMask), lanes 9-13 write the
// upcoming sections (strided when sectionHorizon > 5), lane 14 the agent's own block; the blocks of the other karts are written by
// lanes 0 .. A - 2 (one kart each) before they turn to their ray or section.
// Lane l of the group first evaluates the forward vector of kart l % A (one fp64 sin/cos pair per lane instead of A per
// agent) and the group shares them by shuffles.
constexpr int OBS_LANES = 16;
// BLOCK: threads per workgroup = 16 x the agents that share one LDS copy of the tables (round 5: 1 024 for a long track, whose 47.5 KB per 16 agents
// cost more to stage than they saved)
template <bool TAB_LDS, int BLOCK = 256>
__global__ __launch_bounds__(BLOCK) void env_observe_kernel(EnvParams P, const hk_agent_state* agents, const uint32_t* hot, const int* slot_of, float* obs, unsigned char* hit_code,
                                                          uint32_t agent_mask /* bit i: agent slot i is observed */)
{
    // the track tables (wall grid walked by nine 20 m rays per agent) staged in LDS, as in the tick kernel
    HK_DYN_SHARED(smem);
    const TabView T = tab_stage<TAB_LDS>(P, smem);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int gid0 = tid / OBS_LANES, l = tid % OBS_LANES;
    const int n_agents = P.E * P.A;
    const bool valid = gid0 < n_agents;
    const int gid = valid ? gid0 : n_agents - 1;       // idle groups shadow the last agent (shuffles stay convergent), writes off
    const int env = gid / P.A, i = gid % P.A;
    const int A = P.A, L = P.L, H = P.H;
    const int dim = HK_NUM_SENSORS + H * 5 + 8 + 12 * (A - 1);
    const int goal = P.laps * L + 1;
    const hk_agent_state* cold = &agents[(size_t)env * A + i];      // the plan arrays
    const uint32_t* const hrow = hot + hot_base<GA>(slot_of[env], 0);   // the env's lane group: kart j is hrow + j
    const KartObs a{hrow + i};
    const float apx = a.px(), apz = a.pz();
    float* o = obs + (size_t)gid * dim;
    // forward vectors of the env's karts: lane l evaluates kart l % A, the group shares them
    float kfx[GA], kfz[GA];
    {
        const float yl = KartObs{hrow + l % A}.yaw();
        float sx, cz;
        hk_sincosf(yl, &sx, &cz);
        const int base = (threadIdx.x & 63) & ~(OBS_LANES - 1);
#pragma unroll
        for (int j = 0; j < GA; j++) {
            kfx[j] = __shfl(sx, base + (j < A ? j : 0), 64);
            kfz[j] = __shfl(cz, base + (j < A ? j : 0), 64);
        }
    }
    float fx = kfx[0], fz = kfz[0];
#pragma unroll
    for (int j = 1; j < GA; j++) if (i == j) { fx = kfx[j]; fz = kfz[j]; }
    if (!valid || !((agent_mask >> i) & 1u)) return;
    if (l == 14) {                                                           // own block HKA:489-496
        o[0] = local_speed(P, a, fx, fz);
        o[1] = (a.flags() & HK_F_ACCEL) ? 1.0f : 0.0f;
        o[2] = (float)a.lane();
        o[3] = a.lane_changes() * 1.0f / P.max_lane_changes;
        o[4] = (a.flags() & HK_F_ACTIVE) ? 1.0f : 0.0f;
        o[5] = a.section_index() * 1.0f / goal;
        o[6] = is_straight(P, T, a.section_index()) ? 1.0f : 0.0f;
        o[7] = tire_wear(P, a.final_steer());
    }
    // team mates, then opponents HKA:500-527: block q of the list is written by lane q of the group (one kart each, side by side,
    // instead of one lane walking all of them while the other fifteen wait)
    if (l < P.n_team[i] + P.n_other[i]) {
        const int nt = P.n_team[i];
        const int bj = l < nt ? P.team[i][l] : P.other[i][l - nt];
        int p = 8 + 12 * l;
        const KartObs b{hrow + bj};
        float bfx = kfx[0], bfz = kfz[0];
#pragma unroll
        for (int q = 1; q < GA; q++) if (bj == q) { bfx = kfx[q]; bfz = kfz[q]; }
        o[p++] = local_speed(P, b, bfx, bfz);
        o[p++] = (b.flags() & HK_F_ACCEL) ? 1.0f : 0.0f;
        o[p++] = (float)b.lane();
        o[p++] = b.lane_changes() * 1.0f / P.max_lane_changes;
        o[p++] = (b.flags() & HK_F_ACTIVE) ? 1.0f : 0.0f;
        o[p++] = is_straight(P, T, b.section_index()) ? 1.0f : 0.0f;
        o[p++] = tire_wear(P, b.final_steer());
        o[p++] = b.section_index() * 1.0f / goal;
        const float bpx = b.px(), bpz = b.pz();
        o[p++] = mag3(bpx - apx, 0.0f, bpz - apz);
        float lp[3];
        inv_transform_point(apx, apz, fx, fz, bpx, P.kart_y, bpz, P.kart_y, lp);
        o[p++] = lp[0]; o[p++] = lp[1]; o[p++] = lp[2];
    }
    if (l == 14 || l == 15) {
        // (lane 14 wrote the own block above)
    } else if (l >= 9) {                                                     // upcoming sections HKA:530-552
        for (int q = l - 9; q < H; q += 5) {
            const int next = (a.section_index() + 1 + q) % L;
            float* os = o + 8 + 12 * (A - 1) + 5 * q;
            float lp[3];
            const int pl = cold->plan_lane[next];
            if (pl != 0) {
                float mx, mz;
                lane_marker(T, next, pl, mx, mz);
                inv_transform_point(apx, apz, fx, fz, mx, T.sec[next].marker_y, mz, P.kart_y, lp);
                os[0] = lp[0]; os[1] = lp[1]; os[2] = lp[2];
                os[3] = cold->plan_vel[next] / P.max_speed;
            } else {
                inv_transform_point(apx, apz, fx, fz, T.sec[next].trig_x, T.sec[next].marker_y, T.sec[next].trig_z, P.kart_y, lp);
                os[0] = lp[0]; os[1] = lp[1]; os[2] = lp[2];
                os[3] = 1.0f;
            }
            os[4] = is_straight(P, T, next) ? 1.0f : 0.0f;
        }
    } else {                                                                 // sensor l HKA:553-603
        const int si = l;
        const float ox = apx + SENSOR_LZ * fx, oz = apz + SENSOR_LZ * fz;
        const bool see = (a.flags() & HK_F_ENABLED) != 0;
        float dx, dz;
        sensor_dir(P, si, fx, fz, dx, dz);
        const float maxd = P.ray_dist[si];
        // Physics.Raycast vs TrackMask through the wall grid: a hit at distance t lies within 1 m of one of the samples
        // o + {0, 2, 4, ...} d, whose cells list every wall within GRID_REACH (2.2 m) of them; once the samples up to
        // distance sd are done, every hit with t <= sd + 1 has been seen, so a best hit that near is final.  The lists
        // are supersets and the minimum over them is what a scan of every wall (the oracle) returns.
        float ht = -1.0f;
        int prev = -1;
        for (float sd = 0.0f; sd < maxd + 1.0f; sd += GRID_CELL) {
            const int cell = grid_cell(P, ox + dx * sd, oz + dz * sd);
            if (cell != prev) {
                prev = cell;
                const int w0 = T.grid_off[cell], w1 = T.grid_off[cell + 1];
                for (int q = w0; q < w1; q++) {
                    float t = ray_seg(ox, oz, dx, dz, T.walls[T.grid_idx[q]]);
                    if (t >= 0.0f && t <= maxd && (ht < 0.0f || t < ht)) ht = t;
                }
            }
            if (ht >= 0.0f && ht <= sd + 1.0f) break;
        }
        float ha = -1.0f;
        int who = -1;
        if (see) {
#pragma unroll
            for (int j = 0; j < GA; j++) {
                const KartObs kj{hrow + j};
                if (j >= A || j == i || !(kj.flags() & HK_F_ENABLED)) continue;
                float t = ray_stadium(ox, oz, dx, dz, kj.px(), kj.pz(), kfx[j], kfz[j], P.ray_agent_r);
                if (t >= 0.0f && t <= maxd && (ha < 0.0f || t < ha)) { ha = t; who = j; }
            }
        }
        int code = 0;                                                        // HitWall / HitOpponent events (HKA:580-598)
        float val;
        if (ht >= 0.0f && (ha < 0.0f || ht < ha)) { val = ht; if (ht < P.wall_val[si]) code = 1; }
        else if (ha >= 0.0f) { val = ha; if (ha < P.agent_val[si]) code = 2 + who; }
        else val = maxd;
        o[8 + 12 * (A - 1) + 5 * H + si] = val;
        if (hit_code) hit_code[(size_t)gid * HK_NUM_SENSORS + si] = (unsigned char)code;
    }
}

} }  // namespace hk::HK_GA_NS
