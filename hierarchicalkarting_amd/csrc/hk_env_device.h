// hk_env_device.h — device-side pieces of the kart environment shared by the env kernels.
//
// Reference (paths under Assets/Karting/Scripts/): AK = KartSystems/ArcadeKart.cs, HKA = AI/HierarchicalKartAgent.cs,
// KA = AI/KartAgent.cs, REC = RacingEnvController.cs, DPT = DiscretePositionTracker.cs.
// Float arithmetic follows the C# expression order; Mathf.* transcendental calls go through include/hk_detmath.h
// (double, rounded to float).  The TU is compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hk.h"
#include "../../include/hk_detmath.h"

// the block's dynamic LDS (tests/host_emu substitutes a host buffer)
#ifdef HK_HOST_EMU
#define HK_DYN_SHARED(name) unsigned char* name = hk_emu::dyn_shared
#else
#define HK_DYN_SHARED(name) extern __shared__ __align__(16) unsigned char name[]
#endif

namespace hk {

constexpr int ENV_MAXA = HK_MAX_AGENTS; // capacity of the per-agent parameter arrays; the kernels are compiled per lane-group width GA (hk_env_ga.h)
#ifndef HK_RUN_CAP
#define HK_RUN_CAP 8
#endif
constexpr int RUN_CAP = HK_RUN_CAP;
constexpr int MCTS_MIN_LATENCY = 40;   // (MCTS_FLUSH_ROUNDS + 1) * RUN_CAP: see flush_mcts (hk_env_launch.h)
// Once the field has spread out (few queued games per round) longer launches pay: fewer rounds, fewer solver launches (each one
// solve's latency).  Measured with the lazily completed long calls of round 2: 8 -> 909 M env-steps/s, 12 -> 925 M, 16 -> 911 M in the
// steady state, but 437 / 397 / 368 M over the first 512 ticks of a race — so 12 only after BULK_TICKS.  (Same box, same call: 8 -> 893 M,
// 12 -> 906 M; 10 and 14 -> 750 M: the cap must be a multiple of the solve cadence, or the envs of a wave end their launches in different
// solve phases and the wave runs the solve path on every tick.)  With the eager assembly (hk_env_run.h): 4 -> 1 125 M, 8 -> 1 217 M,
// 12 -> 1 217 M, 16 -> 1 194 M, 24 -> 1 123 M, 32 -> 1 050 M.
constexpr int RUN_CAP_SPREAD = 12;      // ticks per env per launch (> cadence).  Measured at E = 65 536, 4-agent Oval: 8 -> 533 M env-steps/s,
                                // 6 -> 456 M, 5 -> 408 M (misaligned with the 4-tick cadence), 16 -> 503 M, 32 -> 450 M, 128 -> 264 M: a quad that
                                // queues a game idles its lanes until the launch ends, so long launches waste lanes
// hk_env_state.reserved[1]: bits 0..3 the env's phase inside a tick (0 between calls), bit 4 a scheduling hint that survives calls
constexpr int ENV_PHASE_MASK = 15, ENV_PACK_HINT = 16;
constexpr float DEG2RAD_F = 0.0174532924f;
constexpr float TWO_PI_F = 2.0f * HK_PI_F;
constexpr float CAP_R = 0.45f;          // kart capsule (BaseKartClassic.prefab): radius, core segment in kart-local z
constexpr float CAP_Z0 = -0.107f - 0.55f;
constexpr float CAP_Z1 = -0.107f + 0.55f;
constexpr float SENSOR_LZ = 0.1f;       // MLAgent_Sensors origin (0, 0.5, 0.1)
constexpr float TRIG_HX = 5.0f, TRIG_HZ = 0.5f;

struct SecDev {                         // one DiscretePositionTracker + Waypoint geometry, with forward precomputed
    float trig_x, trig_z, fx, fz;
    float yaw_rad, marker_y, inside_radius;
    int optimal_lane;
    float lane_x[4], lane_z[4];
};

struct SecGeo { float track_width, track_length, turn_degrees; int left_turn; };   // DPT fields only the MCTS planner reads

// the engine restatement's derived constants, formed once on the host (env_build_params); the CPU oracle forms them with the same expressions
struct EngCurve { float inv_ext, inv_span, ext, asy, a3, a2, a1, b3, b2, b0, flat; };
struct EngDerived {
    float inv_m, inv_i, inv_mw;
    float side_kf, side_kr, fwd_kf, fwd_kr;     // stiffness * axle load * dt: the impulse of friction coefficient 1 in one tick
    float inv_damp_f, inv_damp_r;               // 1 / (1 + dt * wheelDampingRate / (m_wheel r^2 / 2))
    float jden_r, jlden_r;                      // rear axle (not steered: levers zr and 0): 1 / (1/m + zr^2 / I),  1 / (1/m_wheel + 1/m)
    EngCurve side, fwd;
};

struct EnvParams {
    int E, A, L, NW;
    float dt, kart_y;
    hk_kart_stats st;
    int laps, max_steps, max_lane_changes, H, disable_on_end, hold, auto_reset;
    int high_mode[ENV_MAXA], low_mode[ENV_MAXA], depth[ENV_MAXA], vbucket[ENV_MAXA];
    int n_team[ENV_MAXA], team[ENV_MAXA][ENV_MAXA], n_other[ENV_MAXA], other[ENV_MAXA][ENV_MAXA];
    float sens_c[HK_NUM_SENSORS], sens_s[HK_NUM_SENSORS];   // cos / sin of the sensors' local yaw (MLAgent_Sensors.prefab)
    float ray_dist[HK_NUM_SENSORS];
    uint32_t jitter_seed;
    float jitter_pos, jitter_yaw;
    int env_id_base, nperm;
    float init_acc, max_speed, ray_agent_r;
    int debug;
    // the read-only track tables live in ONE packed buffer (16-B aligned segments); kernels copy it to LDS
    const unsigned char* tab;   // packed tables in global memory
    int tab_bytes, o_walls, o_goff, o_gidx, o_cut, o_tmask, o_ncnt;
    // uniform grid over the walls' bounding box: cell (ix, iz) lists every wall segment within GRID_REACH of the cell; those within
    // NEAR_REACH come first (the "near" list: the first near_cnt[cell] entries), each part in ascending wall index
    float grid_x0, grid_z0, grid_inv;   // origin and 1 / cell size
    int grid_nx, grid_nz;
    // coarse grid over the same box: per TRIG_CELL x TRIG_CELL cell a 64-bit mask of the section Triggers a kart in that cell can overlap
    float tgrid_x0, tgrid_z0, tgrid_inv;      // its box contains every Trigger centre
    int tgrid_nx, tgrid_nz;
    const int* perms;      // [A!][A]
    // MCTS planner (hk_env_mcts.h)
    const SecGeo* sec_geo; // [L]
    int team_of[ENV_MAXA], time_precision[ENV_MAXA], section_window[ENV_MAXA];
    int mcts_iter, mcts_iter0, mcts_lat, mcts_lat0, any_mcts;
    int any_lqr;            // agents whose low level is the LQ game (none: phase B1 has nothing to solve, the fission tick kernel never parks for it)
    int eager;          // 1: an env whose budget ends at a solve tick still assembles that tick's games before the launch ends (hk_env_run.h)
    int run_cap;        // ticks an env may run per launch (RUN_CAP; RUN_CAP_SPREAD for long calls on a spread field, hk_api.hip step_ticks)
    int mcts_pause;     // set per hk_step call: an env that requested a planner search stops at the next tick boundary until the search has run (hk_api.hip step_ticks)
    uint32_t mcts_seed;
    // reward shaping (hk_env_reward.h)
    int rewards, n_teams, team_size[ENV_MAXA], training_agent[ENV_MAXA];
    float wall_val[HK_NUM_SENSORS], agent_val[HK_NUM_SENSORS];   // Sensor.WallHitValidationDistance / AgentHitValidationDistance
    hk_reward_params rw;
    // 1: during the start hold the solves after the first cadence are skipped (see env_run_kernel): the karts are frozen, so every
    // one of them would reproduce the controls the first one decoded
    int hold_dedupe;
    // Training mode (hk_env_training.h)
    int training_reset;
    uint32_t train_seed;
    // engine restatement (hk.h hk_engine_params), with the derived constants the oracle computes per call: 1 / mass, 1 / inertia and
    // the axles' static loads share * mass * gravity (front: -zr / (zf - zr), rear: zf / (zf - zr))
    hk_engine_params eng;
    EngDerived engd;
    // floor(2^32 / L) + 1: x / L == (x * L_magic) >> 32 exactly for 0 <= x < 2^32 / L (sections count in the thousands); the hardware has no
    // integer divide, and the tick loop divides by the section count on every tick (TelemetryViewer's lap number) and several times per Trigger
    uint32_t L_magic;
    // tight Trigger candidates: per cell of the wall grid the Triggers whose box, grown by the capsule's reach, meets the cell
    int o_tmask2;
    // pinned host word the completion guards raise beside status bit 2 (hk_api.hip verify_optimistic: the host looks at it after a stream sync instead of
    // copying the status words back — one round trip less at the end of every short call); nullptr: none
    int* guard_flag;
    int tab_stage_bytes;    // bytes of the packed tables a block copies to LDS: all of them, or (a long track) all but the last segment, tmask2, which is
                            // then read from global memory (one 8-byte load per kart and tick)
};
// ---- device buffers of the MCTS planner (hk_env_mcts.h) and of the reward shaping (hk_env_reward.h): the same for every GA
struct MctsKartSnap { int section, lane, lane_changes, tire_age; int sec_time[HK_MCTS_SECTIME_RING]; };
struct MNode {
    int parent, first_child, last_child, next_sibling;
    int numEpisodes;
    float totalValue;
    unsigned char action, upnext, n_children, pad;
    int pad2;
};
static_assert(sizeof(MNode) == 32, "MNode layout");
constexpr int HK_MCTS_MAX_CLASSES = 2;      // distinct (velocityBucketSize, timePrecision) pairs among the MCTS agents of one handle
struct MctsDev {
    hk_mcts_state* st;      // [E][A]; nullptr: no agent plans with MCTS
    void* req;              // [E][A] MctsReq (its size depends on GA: one kart snapshot per lane of the group)
    int* qcnt;              // [2 sets][2]: {queued searches, unused}; the host flips the set when it launches
                            // the search kernel (every few rounds of the tick kernel, see env_launch_lqn) and clears the new one
    int* queue;             // [2][2*E*A]: (env * A + agent) | generation << 24
    MNode* nodes;           // [slots][pool_cap]
    int* roots;             // [words of the root position][slots]: each search's root game state (hk_env_mcts.h)
    // move tables, filled once by mcts_table_kernel with the same device functions the search would call (so they are
    // bit-identical to evaluating applyAction on the spot): what a move costs depends only on the section (mod L), the
    // kart's lane and velocity bucket and the action — not on the tree
    int* dt_tab;            // [L][4 lanes][nv + 1 buckets][na actions]: time added (x timePrecision); < 0 = infeasible
    float* load_tab;        // [L][4][na]: tireLoad of the move
    float* rad_tab;         // [L][4][4]: radiusOfLane(section, from, to)
    unsigned long long* mask_tab;   // [rows = L * 4 * (nv + 1)]: bit a: action a exists and is feasible from this row (dt >= 0)
    unsigned char* order_tab;   // [rows][na]: the canonical actions in the rollout's order for this row (hk_env_mcts.h mcts_order_kernel)
    int nv;                 // velocity buckets of the action list (<= 9)
    int na;                 // actions of the list = 4 nv: the row stride of dt_tab / load_tab / order_tab
    int lds_tier;           // which tables the search kernel stages in LDS beside dt / rad / flags: bit 0 masks, bit 1 loads, bit 2 orders
    int ntab;               // entries of dt_tab
    int lds_attr_set;       // host: the search kernel's dynamic-LDS limit has been raised on this device
    int pool_cap;
    int slots;              // trees the arena holds: one per resident lane of the search kernel (persist = 0) or one per agent (persist = 1)
    int grid_lanes;         // lanes of the search kernel's grid (<= MCTS_ARENA_WAVES waves)
    int persist;            // 1: every agent owns an arena slice and its tree survives between searches (root reuse costs nothing
                            // extra); 0: the arena belongs to the resident lanes and a re-searched root is rebuilt by replay (MctsReq)
};
struct RwDev {
    int* sec_time;      // [E][A teams][S]  minSectionTimes (episode step; -1 = key absent)
    int* sec_cnt;       // [E][A teams][S]  agentsPastSection
    int S;              // laps * L + 2
    unsigned char* hit_code;   // [E][A][sensors]: what the last CollectObservations saw closer than the validation distance
                               // (0 nothing, 1 wall, 2 + j agent j); replayed by reward_hits_kernel
};

constexpr float TRIG_CELL = 8.0f;       // coarse cell of the Trigger candidate masks
constexpr float TRIG_REACH = 6.6f;      // a kart overlaps a Trigger only within 6.5 m of its centre (box half diagonal 5.03 + capsule reach 1.11)
constexpr float GRID_CELL = 2.0f;       // cell size (m)
constexpr float GRID_REACH = 2.2f;      // list radius: the 2 m side rays, plus slack for float rounding of the cell index
constexpr float NEAR_REACH = 1.3f;      // the near list: 1 m half-spacing of the long-ray samples / 1.11 m contact reach, plus slack.
                                        // On the racing line it is empty: a tick's wall-contact pass then costs two LDS reads.

// where a kernel reads the track tables from: the packed global buffer, or its per-block LDS copy
struct TabView {
    const SecDev* sec;            // [L]
    const hk_wall_seg* walls;     // [NW]
    const unsigned short* grid_off;   // [nx*nz + 1] candidate wall segments per grid cell (ascending wall index)
    const unsigned short* grid_idx;
    const unsigned char* near_cnt;    // [nx*nz] how many entries at the head of the cell's list lie within NEAR_REACH
    const unsigned char* cut;     // [L][5][5]: does the ray lane marker -> next lane marker hit a wall (HKA:832)
    const uint2* tmask;           // [tgrid_nx * tgrid_nz] Trigger candidates per coarse cell (bit t = section t)
    const unsigned short* tmask2; // [grid_nx * grid_nz] the (at most two) Triggers a kart in this 2 m cell of the wall grid can overlap: section indices in the
                                  // low / high byte, 0xFF = none; 0xFEFE = more than two, use `tmask` (hk_env_params.h)
};
__host__ __device__ inline TabView tab_view(const EnvParams& P, const unsigned char* base)
{
    TabView T;
    T.sec = reinterpret_cast<const SecDev*>(base);
    T.walls = reinterpret_cast<const hk_wall_seg*>(base + P.o_walls);
    T.grid_off = reinterpret_cast<const unsigned short*>(base + P.o_goff);
    T.grid_idx = reinterpret_cast<const unsigned short*>(base + P.o_gidx);
    T.near_cnt = base + P.o_ncnt;
    T.cut = base + P.o_cut;
    T.tmask = reinterpret_cast<const uint2*>(base + P.o_tmask);
    T.tmask2 = reinterpret_cast<const unsigned short*>(base + P.o_tmask2);
    return T;
}
// copy the packed tables into dynamic LDS (all threads of the block); TAB_LDS false: the launch passed no dynamic LDS (tables
// larger than the budget) and they are read from global memory.  A COMPILE-TIME choice: with a run-time one the table pointers
// are generic, every table read is a flat_load that finds its way to LDS through the vector-memory path, and the dependent
// reads of the wall lists (cell -> list -> segment) pay that latency in series — the four short sensor rays alone took 11 % of
// the tick kernel; as ds_reads the headline gains 5 % (903 -> 952 M env-steps/s, same box).
// (stage_bytes: what the caller needs — phase B1 reads neither Trigger mask table, the last two segments)
template <bool TAB_LDS>
__device__ inline TabView tab_stage(const EnvParams& P, unsigned char* smem, const int stage_bytes = 0)
{
    if (!TAB_LDS) return tab_view(P, P.tab);
    const int n16 = (stage_bytes ? stage_bytes : P.tab_stage_bytes) >> 4;
    const uint4* src = reinterpret_cast<const uint4*>(P.tab);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    // eight loads in flight per lane: written as `dst[i] = src[i]` the loop waits for every load before its LDS write (one L2 round trip per
    // 16 bytes and lane — 12 of them in a row on the Oval, ~12 us at the head of EVERY launch, with all the launch's waves in it at once)
    int i = threadIdx.x;
    const int bd = blockDim.x;
    for (; i + 7 * bd < n16; i += 8 * bd) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = src[i + k * bd];
#pragma unroll
        for (int k = 0; k < 8; k++) dst[i + k * bd] = v[k];
    }
    if (i + 3 * bd < n16) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = src[i + k * bd];
#pragma unroll
        for (int k = 0; k < 4; k++) dst[i + k * bd] = v[k];
        i += 4 * bd;
    }
    for (; i < n16; i += bd) dst[i] = src[i];
    __syncthreads();
    TabView T = tab_view(P, smem);
    if (P.tab_stage_bytes < P.tab_bytes) T.tmask2 = reinterpret_cast<const unsigned short*>(P.tab + P.o_tmask2);
    return T;
}

// ------------------------------------------------------------------ float helpers (Unity Mathf semantics, Q10)
__device__ __forceinline__ float f_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float f_max(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float f_abs(float a) { return a < 0.0f ? -a : a; }
__device__ __forceinline__ float f_clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float f_sign(float v) { return v >= 0.0f ? 1.0f : -1.0f; }
__device__ __forceinline__ float mag2(float x, float z) { return sqrtf(x * x + 0.0f * 0.0f + z * z); }
__device__ __forceinline__ float mag3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }
__device__ __forceinline__ bool f_finite(float x) { return !(isinf(x) || isnan(x)); }

// x / L and x % L for 0 <= x < 2^32 / L (exact: see EnvParams::L_magic)
__device__ __forceinline__ int div_L(const EnvParams& P, int x) { return (int)(((unsigned long long)(uint32_t)x * (unsigned long long)P.L_magic) >> 32); }
__device__ __forceinline__ int mod_L(const EnvParams& P, int x) { return x - div_L(P, x) * P.L; }
__device__ __forceinline__ bool is_straight(const EnvParams& P, const TabView& T, int section) { return T.sec[mod_L(P, section)].inside_radius == 0.0f; }

__device__ __forceinline__ float kart_steer(const EnvParams& P, float acc_ang_v)
{   // AK:300
    return f_clamp(P.st.MaxSteer * hk_expf(-acc_ang_v / P.st.TireWearRate), P.st.MinSteer, P.st.MaxSteer);
}
__device__ __forceinline__ float tire_wear(const EnvParams& P, float steer)
{   // AK:304-307
    return (P.st.MaxSteer - steer) / (P.st.MaxSteer - P.st.MinSteer);
}
__device__ __forceinline__ float max_lat_gs(const EnvParams& P, float wear)
{   // AK:517-520
    return (1 - wear) * (P.st.MaxGs - P.st.MinGs) + P.st.MinGs;
}
__device__ __forceinline__ float turning_radius(float vx, float vz, float fx, float fz, float wy)
{   // AK:522-529
    float out = (vx * fx + vz * fz) / wy;
    if (isinf(out) || isnan(out)) return 1000.0f;
    return out;
}
__device__ inline float max_speed_for_state(const EnvParams& P, float fx, float fz, float vx, float vz, float wy, float final_steer)
{   // AK:531-547
    float radius = turning_radius(vx, vz, fx, fz, wy);
    float wear = tire_wear(P, final_steer);
    if (radius == 0) return P.st.TopSpeed;
    float allowed = sqrtf(max_lat_gs(P, wear) * 9.81f * f_abs(radius));
    if (isinf(allowed) || isnan(allowed)) allowed = P.st.TopSpeed;
    return f_clamp(allowed, 0.0001f, P.st.TopSpeed);
}

// grid cell of a point (clamped to the grid): every wall within GRID_REACH of the point is in the cell's list
__device__ __forceinline__ int grid_cell(const EnvParams& P, float x, float z)
{
    int ix = (int)((x - P.grid_x0) * P.grid_inv), iz = (int)((z - P.grid_z0) * P.grid_inv);
    ix = ix < 0 ? 0 : (ix >= P.grid_nx ? P.grid_nx - 1 : ix);
    iz = iz < 0 ? 0 : (iz >= P.grid_nz ? P.grid_nz - 1 : iz);
    return iz * P.grid_nx + ix;
}

// Triggers a kart at (x, z) can overlap: every Trigger within TRIG_REACH of the kart's coarse cell (a superset of those within
// 6.5 m of the kart; a position outside the grid's box clamps to the border cell, and since every Trigger centre lies inside
// the box the clamped point is nearer to each of them than the kart is)
// The tight version: the kart's cell of the 2 m wall grid lists the Triggers whose box, grown by the capsule's reach (1.107 m + margin),
// meets the cell — a kart is inside such a region for a sixth of every section, so most ticks find no candidate at all (the coarse
// masks below: 2.5 candidates per kart and tick).  Border cells stand for everything outside the grid on their side.
__device__ __forceinline__ unsigned trig_candidates_tight(const EnvParams& P, const TabView& T, float x, float z)
{
    return T.tmask2[grid_cell(P, x, z)];
}
__device__ __forceinline__ uint2 trig_candidates(const EnvParams& P, const TabView& T, float x, float z)
{
    int ix = (int)((x - P.tgrid_x0) * P.tgrid_inv), iz = (int)((z - P.tgrid_z0) * P.tgrid_inv);
    ix = ix < 0 ? 0 : (ix >= P.tgrid_nx ? P.tgrid_nx - 1 : ix);
    iz = iz < 0 ? 0 : (iz >= P.tgrid_nz ? P.tgrid_nz - 1 : iz);
    return T.tmask[iz * P.tgrid_nx + ix];
}

// ------------------------------------------------------------------ analytic Physics.Raycast pieces
// ray (o, unit d) vs one wall segment.  With e = p1 - p0, w = p0 - o:  den = d x e,  s = (w x d) / den (position along the
// segment),  t = (w x e) / den (distance along the ray).  The crossing test 0 <= s <= 1, t >= 0 is made on the cross products
// themselves (signs and |s numerator| <= |den|): no rounded quotient is compared, and only an actual hit pays for a division.
#define HK_RAY_SEG_CROSS                                     \
    const float ex = w.x1 - w.x0, ez = w.z1 - w.z0;          \
    den = dx * ez - dz * ex;                                 \
    const float wx = w.x0 - ox, wz = w.z0 - oz;              \
    tn = wx * ez - wz * ex;                                  \
    const float sn = wx * dz - wz * dx;                      \
    return den > 0.0f ? (sn >= 0.0f && sn <= den && tn >= 0.0f) : (den < 0.0f && sn <= 0.0f && sn >= den && tn <= 0.0f);
// host twin (hk_create precomputes the static lane->lane "cut" rays with the same arithmetic)
inline bool ray_seg_hits_host(float ox, float oz, float dx, float dz, const hk_wall_seg& w, float& tn, float& den) { HK_RAY_SEG_CROSS }
inline float ray_seg_host(float ox, float oz, float dx, float dz, const hk_wall_seg& w)
{
    float tn, den;
    return ray_seg_hits_host(ox, oz, dx, dz, w, tn, den) ? tn / den : -1.0f;
}
// does the ray cross the segment?  tn / den is then the distance
__device__ __forceinline__ bool ray_seg_hits(float ox, float oz, float dx, float dz, const hk_wall_seg& w, float& tn, float& den) { HK_RAY_SEG_CROSS }
__device__ __forceinline__ float ray_seg(float ox, float oz, float dx, float dz, const hk_wall_seg& w)
{
    float tn, den;
    return ray_seg_hits(ox, oz, dx, dz, w, tn, den) ? tn / den : -1.0f;
}

// ray vs another kart's capsule sliced at the ray height (stadium); origin inside -> no hit (Q10)
__device__ inline float ray_stadium(float ox, float oz, float dx, float dz, float kpx, float kpz, float fx, float fz, float r)
{   // (fx, fz) = the other kart's forward
    float rx = fz, rz = -fx;
    float relx = ox - kpx, relz = oz - kpz;
    float lx = relx * rx + relz * rz;
    float lz = relx * fx + relz * fz;
    float ldx = dx * rx + dz * rz;
    float ldz = dx * fx + dz * fz;
    float cz = f_clamp(lz, CAP_Z0, CAP_Z1);
    float ddz = lz - cz;
    if (lx * lx + ddz * ddz <= r * r) return -1.0f;
    float best = -1.0f;
    {
        float tmin = 0.0f, tmax = 3.0e38f;
        bool ok = true;
        if (ldx == 0.0f) { if (lx < -r || lx > r) ok = false; }
        else {
            float t1 = (-r - lx) / ldx, t2 = (r - lx) / ldx;
            if (t1 > t2) { float tt = t1; t1 = t2; t2 = tt; }
            tmin = f_max(tmin, t1); tmax = f_min(tmax, t2);
        }
        if (ok) {
            if (ldz == 0.0f) { if (lz < CAP_Z0 || lz > CAP_Z1) ok = false; }
            else {
                float t1 = (CAP_Z0 - lz) / ldz, t2 = (CAP_Z1 - lz) / ldz;
                if (t1 > t2) { float tt = t1; t1 = t2; t2 = tt; }
                tmin = f_max(tmin, t1); tmax = f_min(tmax, t2);
            }
        }
        if (ok && tmin <= tmax) best = tmin;
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
        float czc = c == 0 ? CAP_Z0 : CAP_Z1;
        float mx = lx, mz = lz - czc;
        float b = mx * ldx + mz * ldz;
        float cc = mx * mx + mz * mz - r * r;
        float disc = b * b - cc;
        if (disc < 0.0f) continue;
        float t = -b - sqrtf(disc);
        if (t >= 0.0f && (best < 0.0f || t < best)) best = t;
    }
    return best;
}

// closest points between 2-D segments (p1->q1, p2->q2); returns squared distance
__device__ inline float seg_seg_closest(float p1x, float p1z, float q1x, float q1z, float p2x, float p2z, float q2x, float q2z,
                                        float& c1x, float& c1z, float& c2x, float& c2z)
{
    float d1x = q1x - p1x, d1z = q1z - p1z;
    float d2x = q2x - p2x, d2z = q2z - p2z;
    float rx = p1x - p2x, rz = p1z - p2z;
    float a = d1x * d1x + d1z * d1z;
    float ee = d2x * d2x + d2z * d2z;
    float f = d2x * rx + d2z * rz;
    float s, t;
    const float EPS = 1e-12f;
    if (a <= EPS && ee <= EPS) { s = 0.0f; t = 0.0f; }
    else if (a <= EPS) { s = 0.0f; t = f_clamp(f / ee, 0.0f, 1.0f); }
    else {
        float c = d1x * rx + d1z * rz;
        if (ee <= EPS) { t = 0.0f; s = f_clamp(-c / a, 0.0f, 1.0f); }
        else {
            float b = d1x * d2x + d1z * d2z;
            float den = a * ee - b * b;
            if (den != 0.0f) s = f_clamp((b * f - c * ee) / den, 0.0f, 1.0f); else s = 0.0f;
            t = (b * s + f) / ee;
            if (t < 0.0f) { t = 0.0f; s = f_clamp(-c / a, 0.0f, 1.0f); }
            else if (t > 1.0f) { t = 1.0f; s = f_clamp((b - c) / a, 0.0f, 1.0f); }
        }
    }
    c1x = p1x + d1x * s; c1z = p1z + d1z * s;
    c2x = p2x + d2x * t; c2z = p2z + d2z * t;
    float ddx = c1x - c2x, ddz = c1z - c2z;
    return ddx * ddx + ddz * ddz;
}

__device__ __forceinline__ void kart_core(float fx, float fz, float px, float pz, float& ax, float& az, float& bx, float& bz)
{
    ax = px + CAP_Z0 * fx; az = pz + CAP_Z0 * fz;
    bx = px + CAP_Z1 * fx; bz = pz + CAP_Z1 * fz;
}

// Is the wall segment within `reach` of the point?  (no division: the three cases of the closest point — beyond either end, or the foot of the
// perpendicular, where dist^2 = cross^2 / |e|^2 is compared as cross^2 <= reach^2 |e|^2.)  Used as an exact CULL with a reach a centimetre or
// two beyond what the following test needs, so float rounding cannot drop a wall that matters.
__device__ __forceinline__ bool wall_within(const hk_wall_seg& w, const float px, const float pz, const float reach)
{
    const float ex = w.x1 - w.x0, ez = w.z1 - w.z0, rx = px - w.x0, rz = pz - w.z0;
    const float u = rx * ex + rz * ez, ee = ex * ex + ez * ez, r2 = reach * reach;
    if (u <= 0.0f) return rx * rx + rz * rz <= r2;
    if (u >= ee) { const float qx = px - w.x1, qz = pz - w.z1; return qx * qx + qz * qz <= r2; }
    const float cr = rx * ez - rz * ex;
    return cr * cr <= r2 * ee;
}

// ------------------------------------------------------------------ engine restatement: tire forces (hk.h hk_engine_params)
// Arithmetic contract shared with oracle/hk_oracle_env.c engine_wheels: divisions by configuration constants are multiplications by
// reciprocals formed once on the host (env_build_params); a WheelFrictionCurve is ONE cubic in Horner form on its piece.
__device__ __forceinline__ float curve_eval(const EngCurve& c, float slip)
{
    // Both pieces are evaluated and the VALUE is selected.  (`in1 ? c.a3 : c.b3` is a conditional between two lvalues: C++ selects the
    // ADDRESS and loads once, and with `c` in the kernel-argument segment that load is a per-lane global_load — eight of them per call,
    // four calls a tick, made the tire forces the most expensive part of the tick: 11.7 of 39 kcycles, profiles/r04_c_*.)
    const float t1 = slip * c.inv_ext, t2 = (slip - c.ext) * c.inv_span;
    const float v1 = ((c.a3 * t1 + c.a2) * t1 + c.a1) * t1 + 0.0f;
    const float v2 = ((c.b3 * t2 + c.b2) * t2 + 0.0f) * t2 + c.b0;
    const float v = slip <= c.ext ? v1 : v2;
    const float flat = c.flat;
    return slip <= c.asy ? v : flat;
}
// One tick of the four WheelColliders' tire forces on a free rigid body (centre of mass at the kart origin), both axles from the same
// velocities; the front pair is steered by KartAnimation (steer_smoothed * max_steer_deg).
//   sideways  the lateral impulse mu(slip) * load * dt against the axle's sideways motion, capped at what stops that motion
//   rolling   nothing drives or brakes the wheels: the pair's rim speed u follows the ground speed through forwardFriction of the slip
//             (u - v_long) / (|v_long| + 4); the reaction acts on the body; wheelDampingRate slows the spin (implicit, as PhysX does)
__device__ __forceinline__ void engine_wheels(const EnvParams& P, const float fx, const float fz, const float steer_smoothed,
                                              float& vx, float& vz, float& wy, float& uf, float& ur)
{
    const hk_engine_params& g = P.eng;
    const EngDerived& E = P.engd;
    const float rx = fz, rz = -fx;
    float sd, cd;
    hk_sincosf_near0(steer_smoothed * g.max_steer_deg * DEG2RAD_F, &sd, &cd);      // |delta| <= 30 degrees: the reduction-free path of the same function
    float dvx = 0.0f, dvz = 0.0f, dw = 0.0f;
    {   // front axle: levers zf cos(delta) (sideways impulse) and zf sin(delta) (rolling impulse)
        const float zk = g.axle_zf;
        const float wfx = cd * fx + sd * rx, wfz = cd * fz + sd * rz;
        const float wlx = cd * rx - sd * fx, wlz = cd * rz - sd * fz;
        const float vkx = vx + wy * zk * rx, vkz = vz + wy * zk * rz;
        const float vlong = vkx * wfx + vkz * wfz, vlat = vkx * wlx + vkz * wlz;
        const float slip = f_abs(vlat) / (f_abs(vlong) + g.slip_min_speed);
        float jn = curve_eval(E.side, slip) * E.side_kf;
        const float lev = zk * cd;
        const float jmax = f_abs(vlat) / (E.inv_m + lev * lev * E.inv_i);
        if (jn > jmax) jn = jmax;
        if (vlat > 0.0f) jn = -jn;
        dvx += wlx * (jn * E.inv_m); dvz += wlz * (jn * E.inv_m); dw += jn * lev * E.inv_i;
        if (g.wheel_rolling) {
            const float du = uf - vlong;
            const float ls = du / (f_abs(vlong) + g.long_slip_min_speed);
            float jl = curve_eval(E.fwd, f_abs(ls)) * E.fwd_kf;
            const float levl = zk * sd;
            const float jlmax = f_abs(du) / (E.inv_mw + E.inv_m + levl * levl * E.inv_i);
            if (jl > jlmax) jl = jlmax;
            if (ls < 0.0f) jl = -jl;
            dvx += wfx * (jl * E.inv_m); dvz += wfz * (jl * E.inv_m); dw += jl * levl * E.inv_i;
            uf = (uf - jl * E.inv_mw) * E.inv_damp_f;
        }
    }
    {   // rear axle (not steered): levers zr and 0
        const float zk = g.axle_zr;
        const float vkx = vx + wy * zk * rx, vkz = vz + wy * zk * rz;
        const float vlong = vkx * fx + vkz * fz, vlat = vkx * rx + vkz * rz;
        const float slip = f_abs(vlat) / (f_abs(vlong) + g.slip_min_speed);
        float jn = curve_eval(E.side, slip) * E.side_kr;
        const float jmax = f_abs(vlat) * E.jden_r;
        if (jn > jmax) jn = jmax;
        if (vlat > 0.0f) jn = -jn;
        dvx += rx * (jn * E.inv_m); dvz += rz * (jn * E.inv_m); dw += jn * zk * E.inv_i;
        if (g.wheel_rolling) {
            const float du = ur - vlong;
            const float ls = du / (f_abs(vlong) + g.long_slip_min_speed);
            float jl = curve_eval(E.fwd, f_abs(ls)) * E.fwd_kr;
            const float jlmax = f_abs(du) * E.jlden_r;
            if (jl > jlmax) jl = jlmax;
            if (ls < 0.0f) jl = -jl;
            dvx += fx * (jl * E.inv_m); dvz += fz * (jl * E.inv_m);
            ur = (ur - jl * E.inv_mw) * E.inv_damp_r;
        }
    }
    vx += dvx; vz += dvz; wy += dw;
}

// Wave-aggregated queue slot allocation: the active lanes whose `pred` holds get consecutive slots from ONE atomicAdd per
// wave (the lowest such lane adds their count, the others take base + their rank).  With a lane-level atomicAdd every queued
// ego hit the same counter: 131 072 serialised atomics per tick in the 2-agent configuration (every tick queues a 2-player game
// per ego) — 1.3 ms of a 1.5 ms launch; the start of a 4-agent race (every ego in a 4-player game) paid the same.
__device__ __forceinline__ int wave_agg_inc(int* counter, bool pred)
{
    const unsigned long long mask = __ballot(pred);
#ifdef HK_HOST_EMU          // the host stand-in's __shfl is an exchange among ALL lanes of the group: run it before the lanes part
    if (mask == 0ull) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader, 64);
    if (!pred) return -1;
#else
    if (!pred) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader, 64);
#endif
    return base + __popcll(mask & ((1ull << lane) - 1ull));
}

// Philox-4x32-10: synthetic start-grid jitter (not in the reference)
__device__ inline void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// The games-per-launch meter (round 6; the host's choice between in-wave solves and queues + the pair solver's launch, hk_api.hip).  Per part of the batch p
// (issue_rounds_split; 0 for an unsplit launch) four words at game_stats[GAME_METER + 4 p]: [0 .. 2] the multi-player games the B1 launches of rounds
// = 0, 1, 2 mod 3 assembled — launch k counts into slot k % 3, clears slot (k + 1) % 3 and reads the finished launch's total from slot (k + 2) % 3 — and
// [3] a decaying maximum of those totals (x 3/4 per launch: the handful of laggards' launches at the end of a call say nothing about the field).
constexpr int GAME_METER = 160, GAME_METER_PARTS = 4;
constexpr int GAME_STATS_N = 160 + 4 * GAME_METER_PARTS;      // game_stats: [0, 16) games by player count, [16, 64) cycle stamps, [64, 160) lane-participation probes
// Diagnostic build only (-DHK_LANEPROF, tools/lane_profile.py): probe k counts the waves that reach it and the lanes switched on when they do
// (game_stats[64 + 2k] lanes, [65 + 2k] waves) — where in the tick kernel the masked half of the average vector instruction lives.
#ifdef HK_LANEPROF
__device__ unsigned long long* hk_lp_ptr;
#define HK_LP(k) do { const unsigned long long m_ = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) { \
    atomicAdd(&hk_lp_ptr[64 + 2 * (k)], (unsigned long long)__popcll(m_)); atomicAdd(&hk_lp_ptr[65 + 2 * (k)], 1ull); } } while (0)
#else
#define HK_LP(k) do { } while (0)
#endif
// Diagnostic build only (-DHK_STAMPS, tools/stamp_profile.py): cycle stamps at the phase boundaries of the fused tick kernel,
// accumulated per lane, reduced to the wave's maximum and added to game_stats[16 + k] when the kernel ends.
#ifdef HK_STAMPS
#define HK_NSTAMP 30
#define HK_ST(h, k) do { const unsigned long long n_ = __builtin_readcyclecounter(); (h).st_acc[k] += (unsigned)(n_ - (h).st_t); (h).st_t = n_; } while (0)
#else
#define HK_ST(h, k) do { } while (0)
#endif

// The fields of hk_agent_state that change every tick ("hot" fields).  On the device they do NOT live in the 460-byte AoS records:
// their home is the HOT TILE array — one tile per wave of the tick kernel, field-major, 64 consecutive dwords per field (AoSoA) — so that
// the head and the tail of a launch are 32 coalesced 256-byte rows per wave instead of 32 x 64 scattered dwords (round 5; the scattered
// form kept a CU's address path busy for ~14 us per launch: 2 x 32 instructions x 64 lines x 8 waves).  Tiles are indexed by the lane-group
// SLOT, not by the env id: a regroup moves tiles and env words physically (env_regroup_scatter_kernel), `perm[slot] = env` and
// `slot_of[env] = slot` translate for everything that is indexed by env id (cold records, results, queues, planner state).  The hot
// fields of the AoS record are a staging copy only: written by hot_gather_kernel for hk_get_agent_state, read by hot_scatter_kernel
// after hk_set_agent_state, and used as scratch by the reset path (reset_agent rewrites the whole record, plans included).
#define HK_HOT_FIELDS(X)                                                                                                              \
    X(float, px) X(float, pz) X(float, yaw) X(float, vx) X(float, vz) X(float, wy) X(float, acc_ang_v) X(float, steering)             \
    X(float, avg_lane_diff) X(float, avg_vel_diff) X(float, final_steer) X(float, contact_nx) X(float, contact_nz)                    \
    X(int, section_index) X(int, lane) X(int, lane_changes) X(int, illegal_lane_changes) X(int, forward_collisions)                   \
    X(int, last_collision_time) X(int, time_steps) X(int, init_checkpoint_index) X(uint32_t, flags) X(uint32_t, trig_lo)              \
    X(uint32_t, trig_hi) X(int, tele_completed_laps) X(int, tele_lap_end_step) X(float, tele_last_lap) X(float, tele_best_lap)        \
    X(float, tele_total_time) X(float, steer_smoothed) X(float, wheel_uf) X(float, wheel_ur)
enum HotField {
#define HK_X(T, n) HF_##n,
    HK_HOT_FIELDS(HK_X)
#undef HK_X
    HF_N
};
static_assert(HF_N == 32, "a hot tile is 32 rows of 64 dwords");
constexpr int HOT_TILE_WORDS = HF_N * 64;
// words of the tile array for E lane groups of GA lanes (whole tiles)
__host__ __device__ inline size_t hot_words(int E, int GA) { const int gpw = 64 / GA; return (size_t)((E + gpw - 1) / gpw) * HOT_TILE_WORDS; }
// word index of field 0 of lane (slot, i); field f sits f * 64 words further.  For the tick kernel's own lane this is
// (gid / 64) * HOT_TILE_WORDS + gid % 64 with gid = slot * GA + i: a wave's row is one contiguous 256-byte run.
template <int GA>
__host__ __device__ __forceinline__ size_t hot_base(int slot, int i)
{
    constexpr int GPW = 64 / GA;
    return (size_t)(slot / GPW) * HOT_TILE_WORDS + (size_t)((slot % GPW) * GA + i);
}
template <class T> __host__ __device__ __forceinline__ T hot_get(const uint32_t* p, int f) { return __builtin_bit_cast(T, p[f * 64]); }
template <class T> __host__ __device__ __forceinline__ void hot_put(uint32_t* p, int f, T v) { p[f * 64] = __builtin_bit_cast(uint32_t, v); }

struct Hot {
#ifdef HK_STAMPS
    unsigned long long st_t;
    unsigned st_acc[HK_NSTAMP];
#endif
#define HK_X(T, n) T n;
    HK_HOT_FIELDS(HK_X)
#undef HK_X
};
// AoS record <-> registers (the staging copy: reset path, hk_get / hk_set_agent_state)
__host__ __device__ __forceinline__ Hot load_hot(const hk_agent_state* a)
{
    Hot h;
#define HK_X(T, n) h.n = a->n;
    HK_HOT_FIELDS(HK_X)
#undef HK_X
    return h;
}
__host__ __device__ __forceinline__ void store_hot(hk_agent_state* a, const Hot& h)
{
#define HK_X(T, n) a->n = h.n;
    HK_HOT_FIELDS(HK_X)
#undef HK_X
}
// hot tile <-> registers; p = tiles + hot_base<GA>(slot, i)
__host__ __device__ __forceinline__ Hot load_hot_tile(const uint32_t* p)
{
    Hot h;
#define HK_X(T, n) h.n = hot_get<T>(p, HF_##n);
    HK_HOT_FIELDS(HK_X)
#undef HK_X
    return h;
}
__host__ __device__ __forceinline__ void store_hot_tile(uint32_t* p, const Hot& h)
{
#define HK_X(T, n) hot_put<T>(p, HF_##n, h.n);
    HK_HOT_FIELDS(HK_X)
#undef HK_X
}


// Sensor.Transform.forward: the kart's forward (fx, fz) turned by the sensor's local yaw (Unity Y rotation, +z toward +x)
__device__ __forceinline__ void sensor_dir(const EnvParams& P, int si, float fx, float fz, float& dx, float& dz)
{
    dx = fx * P.sens_c[si] + fz * P.sens_s[si];
    dz = fz * P.sens_c[si] - fx * P.sens_s[si];
}

// HKA.planFixed :145-166
__device__ inline void plan_fixed(const EnvParams& P, const TabView& T, int agent, int sec, hk_agent_state* a)
{
    int hi = sec + P.depth[agent]; if (hi > 1000) hi = 1000;
    for (int i = sec + 1; i < hi + 1; i++) {
        int key = mod_L(P, i);
        if (a->plan_lane[key] == 0) {
            a->plan_lane[key] = (uint8_t)T.sec[mod_L(P, i - 1)].optimal_lane;
            a->plan_vel[key] = P.max_speed;
        }
    }
}

// KA.Deactivate :405-416 (+ SetZeroInputs :480-486): returns the new flags, zeroes the motion fields
__device__ inline uint32_t deactivate_fields(const EnvParams& P, Hot& h, uint32_t flags)
{
    h.steering = 0.0f;
    h.vx = 0.0f; h.vz = 0.0f; h.wy = 0.0f;
    flags &= ~(HK_F_ACCEL | HK_F_BRAKE | HK_F_CAN_MOVE | HK_F_ACTIVE);
    if (P.disable_on_end) flags &= ~HK_F_ENABLED;
    return flags;
}

// DPT.getBoxColliderForLane :99-111 (lane 0 -> Trigger)
__device__ __forceinline__ void lane_marker(const TabView& T, int idx, int lane, float& x, float& z)
{
    const SecDev& s = T.sec[idx];
    if (lane >= 1 && lane <= 4) { x = s.lane_x[lane - 1]; z = s.lane_z[lane - 1]; }
    else { x = s.trig_x; z = s.trig_z; }
}

}  // namespace hk
