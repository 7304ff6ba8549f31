// hk_env_run.h — the fused tick kernel.
//
// env_run_kernel advances every race instance by up to RUN_CAP Unity FixedUpdate ticks in ONE launch: a group of GA lanes
// (a quad for up to 4 agents; lane = agent) loops  phase A (episode controller + kart-vs-kart rays)  ->  phase B1 (wall rays, game assembly,
// single-player Riccati solve)  ->  phase C (ArcadeKart model, engine restatement, triggers)  with the env words and
// the per-tick agent fields (Hot) in registers; only the plan arrays are touched in memory.  The only thing a quad cannot do alone is a multi-player LQ game
// (2-4 players, 8-16 cooperating lanes): it writes the game, queues it by player count, stores its progress and
// leaves the loop; lqn_round_kernel (hk_lq2_pair.h / lqn_body<N>) solves the queues, and the next launch resumes the env at phase C.  Once the field
// has spread out ~99 % of all games are single-player, so almost every env runs its RUN_CAP ticks without leaving.
//
// Progress words live in hk_env_state.reserved[]: [0] = ticks still to run for the current hk_step, [1] = phase
// (0: at a tick boundary, 1: waiting for a queued game of this tick).  hk_step(n) arms [0] = n and launches
// ceil(n / cadence) + 1 rounds of {env_run_kernel, lqn_round_kernel}: a round always retires at least one solve
// cadence (4 ticks for A > 2, 1 tick otherwise) of every env that is not finished, so that many rounds always suffice;
// rounds that find nothing to do cost a few microseconds.  (RUN_CAP must exceed the cadence: a resumed env finishes its
// pending tick and must be able to reach its next solve tick within the same launch.)  env_check_kernel flags any env
// that still has ticks left after the last round (a bug guard: hk_get_* then fail instead of returning stale state).
// (included once per lane-group width by hk_env_ga.h: no include guard, namespace hk::HK_GA_NS)
#include "hk_env_device.h"
#include "hk_regroup_pos.h"

namespace hk { namespace HK_GA_NS {

__global__ __launch_bounds__(256) void env_arm_kernel(hk_env_state* envs, int E, int n_ticks, int* status)
{
    // ADD to the ticks left: every env is at 0 here unless an earlier call failed its completion guard, and then its leftover ticks are
    // not dropped.  The "did not complete" flag (status bit 2) is sticky: only the getter that reports it (hk_get_agent_state / hk_get_env_state) and hk_reset clear it.
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env < E) envs[env].reserved[0] += n_ticks;
    (void)status;
}

__device__ __forceinline__ void raise_guard(int* status, int* guard_flag)
{
    atomicOr(status, 4);
    if (guard_flag) __hip_atomic_store(guard_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(256) void env_check_kernel(const hk_env_state* envs, int E, int* status, int lazy, int* guard_flag)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= E) return;
    const int left = envs[env].reserved[0], phase = envs[env].reserved[1] & ENV_PHASE_MASK;
    if (left != 0 || phase != 0) {
        if (!lazy) raise_guard(status, guard_flag);          // sticky error for the getters (fixed-rounds mode)
        // what the lazy completion of hk_step needs (hk_api.hip finish_ticks): the host cleared both words before this launch
        atomicMax(status + 1, left);
        if (phase != 0) atomicOr(status + 2, 1);
    }
}

// Regrouping by solve phase.  SolveLQR runs on ticks with episodeSteps % 4 == 0 (A > 2, Q9): during the first episode every env
// of a wave is in the same phase, but auto-resets happen at different ticks, and from then on a wave would run the (expensive)
// solve path for some of its lanes on EVERY tick.  Every few hundred ticks the lane groups are therefore re-assigned so that the
// envs of a wave share their phase: perm[slot] = env, grouped by key = (episode_steps + ticks still to run) & 3, which does not
// change while the call runs.  Envs are independent, so which lanes run an env changes nothing but the speed.
// Second use (round 2): the tail of a call.  Envs that met multi-player games lag behind, and the last rounds of an hk_step run for
// them alone — scattered one or two to a wave, each such wave costing full ticks.  The key therefore also says whether an env is done
// with the call (keys 8..15): unfinished envs are packed, by phase, into the first lane groups; blocks that hold only finished envs
// leave at once.  (A 20-tick call at 8 ticks per launch: 8 rounds of which 5 are tail; with the eager assembly short calls have none.)
// Third use (round 3): packs.  An env whose karts race within 8 m of each other queues a multi-player game at (nearly) every solve
// tick and leaves the launch there; scattered one to a wave, such envs left their 15 wave neighbours' lanes... running, but their own
// idle for the rest of the launch, and the wave no shorter (in-kernel stamps: 22 % of the wave time was lanes waiting for the other
// lane groups of their wave).  The tick kernel therefore leaves a hint in the env's phase word — did its last solve tick queue a game —
// and the key carries it (keys 4..7): packs share waves with packs, which leave their launches together.
__device__ __forceinline__ int regroup_key(const hk_env_state& e)
{
    const int left = e.reserved[0];
    return ((e.episode_steps + left) & 3) + ((e.reserved[1] & ENV_PACK_HINT) ? 4 : 0) + ((left == 0 && (e.reserved[1] & ENV_PHASE_MASK) == 0) ? 8 : 0);
}
// Both kernels aggregate per block: the waves' ballots go through LDS, then threads 0..7 issue ONE atomic per key, side by side
// (per-wave atomics with their return values in series made the pair 125 us for 65 536 envs: 6 % of a 20-tick call).
constexpr int REGROUP_KEYS = 16;
struct RegroupLds { int cnt[4][REGROUP_KEYS]; int base[REGROUP_KEYS]; int cstart[REGROUP_KEYS]; int cn[REGROUP_KEYS]; int ch[REGROUP_KEYS]; };
__device__ __forceinline__ int regroup_block_counts(RegroupLds& L, int key, int& lane_rank)
{   // fills L.cnt[wave][k]; returns nothing useful for key < 0; lane_rank = rank of this lane among its wave's lanes of the same key
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lane_rank = 0;
#pragma unroll
    for (int k = 0; k < REGROUP_KEYS; k++) {
        const unsigned long long m = __ballot(key == k);
        if (lane == 0) L.cnt[wave][k] = __popcll(m);
        if (key == k) lane_rank = __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    return wave;
}
// (both kernels walk the lane-group SLOTS: envs[] is stored by slot)
__global__ __launch_bounds__(256) void env_regroup_count_kernel(const hk_env_state* envs, int E, int* counts /*[2 * REGROUP_KEYS]: counts, cursors*/)
{
    __shared__ RegroupLds L;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int key = slot < E ? regroup_key(envs[slot]) : -1;
    int rank;
    (void)regroup_block_counts(L, key, rank);
    if (threadIdx.x < REGROUP_KEYS) {
        const int k = threadIdx.x;
        const int tot = L.cnt[0][k] + L.cnt[1][k] + L.cnt[2][k] + L.cnt[3][k];
        if (tot) atomicAdd(&counts[k], tot);
    }
}
// The regroup is PHYSICAL (round 5): the lane group that sat in `slot` moves to `pos` with its env words and its rows of the hot tiles
// (GA consecutive dwords of each of the 32 fields: one 16-byte piece per field for a quad), from the old buffers to the new ones — the host
// swaps the pointers after this launch — so that a wave of the tick kernel keeps reading ONE contiguous tile.  perm / slot_of follow.
struct alignas(16) HotPiece { uint32_t v[4]; };
__global__ __launch_bounds__(256) void env_regroup_scatter_kernel(const hk_env_state* envs_old, hk_env_state* envs_new, const uint32_t* hot_old,
                                                                  uint32_t* hot_new, const int* perm_old, int* perm_new, int* slot_of, int E, int* counts, int spread)
{
    __shared__ RegroupLds L;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    hk_env_state es{};
    if (slot < E) es = envs_old[slot];
    const int key = slot < E ? regroup_key(es) : -1;
    int rank;
    const int wave = regroup_block_counts(L, key, rank);
    if (threadIdx.x < REGROUP_KEYS) {
        const int k = threadIdx.x;
        const int tot = L.cnt[0][k] + L.cnt[1][k] + L.cnt[2][k] + L.cnt[3][k];
        int start = 0;                                   // where key k begins in the new order: the counts of the keys before it
        if (!spread) {
            for (int j = 0; j < k; j++) start += counts[j];
        } else {
            // SPREAD (round 6, the games solved in-wave: hk_regroup_pos.h): a class = the lane groups of one phase, done or not done, hinted or not; the
            // classes in the order (done, phase), and inside a class the hinted ones evenly spaced among the plain ones.  `base` is then the RANK inside
            // the key, the class's start / size / hinted count ride beside it.
            const int ck = (k & 3) | ((k & 8) >> 1), kp = k & ~4;
            int cs = 0;
            for (int j = 0; j < REGROUP_KEYS; j++) if (((j & 3) | ((j & 8) >> 1)) < ck) cs += counts[j];
            L.cstart[k] = cs; L.cn[k] = counts[kp] + counts[kp | 4]; L.ch[k] = counts[kp | 4];
        }
        L.base[k] = start + (tot ? atomicAdd(&counts[REGROUP_KEYS + k], tot) : 0);
    }
    __syncthreads();
    if (key >= 0) {
        int pos = L.base[key] + rank;
        for (int w = 0; w < wave; w++) pos += L.cnt[w][key];
        if (spread) {
            const long long N = L.cn[key], H = L.ch[key];
            const bool hinted = (key & 4) != 0;
            if (regroup_spreadable(N, H)) pos = L.cstart[key] + (int)regroup_spread_pos(pos, hinted, N, H);
            else pos = L.cstart[key] + (hinted ? (int)(N - H) + pos : pos);          // (no hinted ones, or too many to keep apart: plain first, packs behind)
        }
        const int env = perm_old[slot];
        perm_new[pos] = env;
        slot_of[env] = pos;              // (one writer per env; nobody reads slot_of in this launch)
        envs_new[pos] = es;
        const HotPiece* src = reinterpret_cast<const HotPiece*>(hot_old + hot_base<GA>(slot, 0));
        HotPiece* dst = reinterpret_cast<HotPiece*>(hot_new + hot_base<GA>(pos, 0));
#pragma unroll 8
        for (int f = 0; f < HF_N; f++) {
#pragma unroll
            for (int q = 0; q < GA / 4; q++) dst[f * 16 + q] = src[f * 16 + q];        // a row is 64 dwords = 16 pieces
        }
    }
}

// The host-facing views ("a host-facing view is produced only by hk_get_*"): hk_get_agent_state / hk_device_agents_ptr gather the hot
// rows into the AoS records, hk_set_agent_state scatters them back; hk_get / hk_set_env_state translate the slot order into env order.
__global__ __launch_bounds__(256) void hot_gather_kernel(hk_agent_state* agents, const uint32_t* hot, const int* slot_of, int n, int A)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int env = t / A, i = t - env * A;
    store_hot(&agents[t], load_hot_tile(hot + hot_base<GA>(slot_of[env], i)));
}
__global__ __launch_bounds__(256) void hot_scatter_kernel(const hk_agent_state* agents, uint32_t* hot, const int* slot_of, int n, int A)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int env = t / A, i = t - env * A;
    store_hot_tile(hot + hot_base<GA>(slot_of[env], i), load_hot(&agents[t]));
}
__global__ __launch_bounds__(256) void envs_gather_kernel(hk_env_state* by_env, const hk_env_state* by_slot, const int* slot_of, int E)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env < E) by_env[env] = by_slot[slot_of[env]];
}
// hk_set_env_state: the progress words of hk_step are the library's (both 0 between calls but for the scheduling hint); a record the host
// saved in the middle of nothing — or filled by hand — must not arm ticks or resume a phase (arming ADDS to reserved[0])
__global__ __launch_bounds__(256) void envs_scatter_kernel(const hk_env_state* by_env, hk_env_state* by_slot, const int* slot_of, int E)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= E) return;
    hk_env_state es = by_env[env];
    es.reserved[0] = 0; es.reserved[1] &= ENV_PACK_HINT;
    by_slot[slot_of[env]] = es;
}

// <false, false, false> is the headline path: the planner hooks (request / consume / beliefs), reward shaping and the
// Training-mode code compile away entirely.  Instantiated: every (HAS_MCTS, HAS_RW) pair without Training code, and
// <true, true, true> for any handle that uses Training mode (its planner / reward parts are also guarded at run time).
// FISSION (round 4): the tick kernel WITHOUT phase B1.  An env that reaches a solve tick parks there (phase 2) after its phase A;
// env_b1_kernel — the next launch on the stream — runs the sensor rays, the assembly and the single-player solves for every parked env
// at once (phase 2 -> 1; multi-player games go to the queues as before), then the solver launch, and the next tick launch resumes the
// envs at phase C.  What it buys: the fused body's register peak is phase B1 (the 4 x 4 Riccati recursion in fp64) ON TOP of the
// per-tick state; without it the tick loop fits its registers and runs more waves per SIMD.  What it costs: a launch per solve
// cadence (4 ticks) instead of per RUN_CAP_SPREAD (12), and the kart records travel through HBM in between.
#ifndef HK_RUN_OCC
#define HK_RUN_OCC 2
#endif
#ifndef HK_FIS_OCC
#define HK_FIS_OCC 3          // three waves per SIMD (168 VGPRs) since round 5: with the back-end switches of __graft_entry__.BACKEND_FLAGS the tick
#endif                        // kernel needs 162 registers and no scratch; 1 632 -> 1 762 M env-steps/s (profiles/r05_a_backend_flags.txt)
// Phase B1 as a real CALL, for the fused Training-mode instantiation only (round 5).  Inlined, phase B1's register peak makes the allocator split the long
// live ranges of the tick loop around it (save copy before, copy back after), and this toolchain puts such save copies at the top of a join block, AHEAD of
// the instruction that restores EXEC: lanes of the other side then get the copy back without ever having made the save (a finished kart's tele_total_time
// read 4.6e-41; DESIGN.md section 10, the second pattern).  Across a call the ABI does the parking itself — callee-saved registers in the callee's prologue,
// the rest next to the call — under the call site's own EXEC.  Slower (the whole kart state goes through the stack around every solve tick) and only here:
// the instantiation serves 2-agent Training fields; everything else stays inline.
__device__ __attribute__((noinline)) int phase_assemble_call(const EnvParams& P, const TabView& T, KartS* ks, const int env, const int ego, const bool act,
                                                             const hk_env_state& es, Hot& h, const float hfx, const float hfz, hk_agent_state* agents,
                                                             const GameSoA games, int* queue_cnt, int* queue, hk_lq_debug* dbg_out, int* status,
                                                             const hk_mcts_state* mcts_all, const LaneCfg& LC)
{
    return phase_assemble(P, T, ks, env, ego, act, es, h, hfx, hfz, agents, games, queue_cnt, queue, dbg_out, status, mcts_all, LC);
}

template <bool HAS_MCTS, bool HAS_RW, bool HAS_TRAIN, bool TAB_LDS, bool FISSION = false>
__global__ __launch_bounds__(256, FISSION ? HK_FIS_OCC : HK_RUN_OCC) void env_run_kernel(EnvParams P, hk_agent_state* agents, uint32_t* hot, hk_env_state* envs,
                                                      hk_episode_result* results, GameSoA games, int* queue_cnt_all,
                                                      int* queue_all, int round, const float* act_steer, const int* act_branch,
                                                      hk_lq_debug* dbg_out, int* status, MctsDev Marg, int mset, RwDev RD, const int* perm,
                                                      unsigned long long* stats, int slot0, int slot1, int qbase, int arm_ticks, int guard)
{
    MctsDev M{};
    if (HAS_MCTS) M = Marg;
#ifdef HK_STAMPS
    const unsigned long long st_entry = __builtin_readcyclecounter();
#endif
#ifdef HK_LANEPROF
    hk_lp_ptr = stats;                 // (every thread stores the same value)
#endif
    __shared__ KartS ks[256];
    HK_DYN_SHARED(smem);
    // this launch runs the lane groups [slot0, slot1) and uses the queue sets qbase, qbase + 1 (one launch for every env: 0, E, 0;
    // plain handles split the batch in two halves on two streams so that one half's solver launch hides behind the other's ticks)
    const int gid = slot0 * GA + blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = gid / GA, i = gid % GA;
    const bool env_ok = slot < slot1;
    // Hot tiles and env words are stored by SLOT (a regroup moves them physically): neither load waits for the env id, which only
    // addresses what is indexed by env (cold records, results, queues, planner state)
    const int env = env_ok ? perm[slot] : 0;
    uint32_t* const htile = hot + (size_t)(gid >> 6) * HOT_TILE_WORDS + (gid & 63);      // == hot + hot_base<GA>(slot, i)
    // the game queues are double buffered over rounds: this launch fills `set`, and clears the other one, which the
    // previous round's lqn kernels have finished reading
    const int set = qbase + (round & 1);
    int* queue_cnt = queue_cnt_all + set * 16;
    int* queue = queue_all + (size_t)set * (GA - 1) * P.E * P.A;
    if (blockIdx.x == 0 && threadIdx.x < 16) queue_cnt_all[(set ^ 1) * 16 + threadIdx.x] = 0;
    hk_env_state es;
    if (env_ok) es = envs[slot];
    else { es.episode_steps = 0; es.inactive_mask = 0; es.experiment_num = 0; es.episodes_done = 0; es.status = 0; es.initial_started = 0; es.reserved[0] = 0; es.reserved[1] = 0; }
    if (env_ok) es.reserved[0] += arm_ticks;             // the first launch of a fixed-round call arms the envs (else env_arm_kernel did)
    // nothing to do in this block? (every env finished its ticks): skip the table staging too
    if (__syncthreads_or(es.reserved[0] > 0 || (es.reserved[1] & ENV_PHASE_MASK) != 0) == 0) return;
#ifdef HK_STAMPS
    const unsigned long long st_a = __builtin_readcyclecounter();
#endif
    // (the kart's rows of the hot tile are asked for BEFORE the table staging and its barrier: 32 loads in flight behind the copy instead of after it)
    const bool has_rec = env_ok && i < P.A;
    Hot h;
    if (has_rec) h = load_hot_tile(htile); else { Hot z = {}; h = z; }
    const TabView T = tab_stage<TAB_LDS>(P, smem);
#ifdef HK_STAMPS
    const unsigned long long st_b = __builtin_readcyclecounter();
#endif
    hk_agent_state* arec = has_rec ? &agents[(size_t)env * P.A + i] : nullptr;
    const int cadence = P.A > 2 ? 4 : 1;
    const int cmask = cadence - 1;                       // cadence is 1 or 4 and episode_steps >= 0: x % cadence == x & cmask
    const uint32_t all_mask = (1u << P.A) - 1u;
    int budget = P.run_cap;
    const LaneCfg LC = lane_cfg(P, i);                   // this lane's agent: modes and player list, read once
    int left = es.reserved[0];
    int phase = es.reserved[1] & ENV_PHASE_MASK;         // 0: at a tick boundary; 1: phases A / B1 of a tick done, waiting for (or holding) its controls
    // an env still parked at its solve tick (phase 2): the round plan of the call before did not foresee that tick (hk_api.hip: the optimistic plan of a
    // field believed to be in lock-step) and launched no env_b1_kernel behind it.  It keeps its ticks and stays parked — the next B1 launch serves it, the
    // completion guard reports it, and the next entry point that looks at the state finishes it.
    const bool stuck = FISSION && phase == 2;
    bool pack = (es.reserved[1] & ENV_PACK_HINT) != 0;   // did the env's last solve tick queue a multi-player game (regroup_key)
    bool dirty = false;
#ifdef HK_STAMPS
    for (int k = 0; k < HK_NSTAMP; k++) h.st_acc[k] = 0;
    h.st_acc[20] = (unsigned)(st_a - st_entry); h.st_acc[21] = (unsigned)(st_b - st_a);     // the head of the launch: lane group + env words; table staging
    h.st_t = st_b;
#endif
    float hfx, hfz;                                      // the kart's forward, carried across ticks (changes only when yaw does)
    hk_sincosf(h.yaw, &hfx, &hfz);
    RwAcc rwv = {0.0f, 0.0f, 0.0f};
    if (HAS_RW && P.rewards && arec) { rwv.cum = arec->cum_reward; rwv.step = arec->step_reward; rwv.group = arec->group_reward; }
    HK_ST(h, 0);                       // [0] prologue: (table staging,) state load
    // P.mcts_pause (long hk_step calls of planner handles): an env with a requested, not yet run search waits at the tick boundary —
    // the host launches the searches of a whole stretch of rounds in ONE batch (a launch lasts as long as one search however few it holds)
    // (round 5: the env does not stop at once — it runs on until the step that would use the search's result, a replan tick or a reset, so that
    // the search launch overlaps up to mcts_latency_ticks of stepping.  hold_at: the first episode step the env may not begin, group-uniform.)
    int hold_at = 0x7FFFFFFF;
    if (HAS_MCTS && P.mcts_pause && env_ok) hold_at = group_min(mcts_hold_step(P, M, env, i));
    auto held_now = [&]() -> bool {
        if (!(HAS_MCTS && P.mcts_pause) || hold_at == 0x7FFFFFFF) return false;
        const int nxt = es.episode_steps + 1;
        // (the step that promotes the result; a replan tick; the tick of an auto-reset — each would touch the request record the search kernel works on)
        return nxt >= hold_at || nxt % 100 == 0 || (P.auto_reset && (nxt >= P.max_steps || (es.inactive_mask & all_mask) == all_mask));
    };
    bool held = held_now();
    // Eager assembly (P.eager, cadence 4).  A budget that ends on a solve tick would leave that tick's games to the NEXT launch:
    // an env that then meets a multi-player game parks at once and sits out the whole launch.  Instead the budget is trimmed so
    // that it ends on a solve tick (first launch of a call, after a reset), and the env runs phases A / B1 of that tick before
    // the launch ends: its games go to this round's solver launch, and every env — racing alone or in a pack — advances one
    // cadence per round at least.  The env waits in phase 1 ("controls ready"), exactly as if it had queued a game.
    // (es.episode_steps is the index of the tick last BEGUN: phase_begin increments it.  An env at a tick boundary runs ticks
    // e + 1 .. e + budget, one resuming its parked tick e runs e .. e + budget - 1; the tick after those should be a solve tick.)
    if (P.eager && cadence > 1) {
        const int over = (es.episode_steps + budget + (phase == 0 ? 1 : 0)) & cmask;
        if (over < budget) budget -= over;
    }
    bool eager_it = false;           // this iteration is the assembly-only one at the end of the budget
    // The loop is wave-uniform: a lane group whose env has nothing (more) to run in this launch stays in it, idle.  `go` depends on
    // the env's words only, so a lane group is in or out as a whole, and the group-wide exchanges below (group_get / group_or:
    // DPP quad permutes, which read 0 from a lane that is switched off) always run with whole groups.
    bool go = env_ok && !stuck && (phase != 0 || (left > 0 && budget > 0 && !held));
    while (__ballot(go) != 0ull) {
        int qn = 0;                  // player count of the multi-player game this ego assembled on this tick (0: none)
        bool began = false;          // this env ran phases A / B1 in this iteration (it was at a tick boundary and not parked)
        bool solved = false;         // ... and the tick was a solve tick
        bool b1_pending = false;     // FISSION: ... whose phase B1 is left to env_b1_kernel
        bool moving = go;            // this env runs phase C in this iteration
        HK_LP(23);                   // every lane of a wave that is still in the loop
        if (go) {
            HK_LP(0);
            dirty = true;
            if (phase == 0) {
                const bool parked = phase_begin<HAS_RW, HAS_TRAIN>(P, env, i, env_ok, es, h, hfx, hfz, agents, results, M, mset, RD, rwv, act_branch);
                HK_ST(h, 1);               // [1] phase A: episode controller + kart-vs-kart rays
                if (!parked) {
                    // Start hold (REC:721-744): the karts cannot move before episodeSteps reaches `hold`, nothing a solve reads changes, and
                    // each SolveLQR overwrites the controls of the one before: the solves after the first cadence of the hold
                    // (ticks 0 and `cadence`: the reset tick of an auto-reset, the first solve tick after hk_reset) would decode
                    // bit-identical controls and are skipped (P.hold_dedupe: no planner whose first plan lands inside the hold,
                    // and the host has not written kart states by hand).  18 of the 128 solve ticks of a race start.
                    const bool act = (es.episode_steps & cmask) == 0 &&                                    // HKA:317 (Q9): episodeSteps % cadence
                                     !(!P.auto_reset && (es.inactive_mask & all_mask) == all_mask && (es.status & 4u)) &&
                                     !(P.hold_dedupe && es.episode_steps > cadence && es.episode_steps < P.hold);
                    if constexpr (FISSION) b1_pending = act && P.any_lqr != 0;      // (no LQ agent: nothing for env_b1_kernel to do, the env moves on)
                    else {
                        if constexpr (HAS_TRAIN) qn = phase_assemble_call(P, T, ks, env, i, act, es, h, hfx, hfz, agents, games, queue_cnt, queue, dbg_out, status, M.st, LC);
                        else qn = phase_assemble(P, T, ks, env, i, act, es, h, hfx, hfz, agents, games, queue_cnt, queue, dbg_out, status, M.st, LC);
                    }
                    began = true;
                    solved = act;
                } else {
                    left -= 1; budget -= 1;         // a parked env lets the tick pass
                    moving = false;
                }
            }
        }
        if (began) {
            HK_LP(22);
            // bin the games for the solver kernels by player count, one atomic per wave and count
            if constexpr (!FISSION) {
#pragma unroll
                for (int n = 2; n <= GA; n++) {
                    const int pos = wave_agg_inc(&queue_cnt[n], qn == n);
                    if (qn == n) queue[(size_t)(n - 2) * P.E * P.A + pos] = env * P.A + i;
                }
            }
            // (FISSION: on a solve tick the planner hook follows SolveLQR, i.e. it runs in env_b1_kernel)
            if (M.st && !(FISSION && b1_pending)) phase_plan(P, M, mset, env, i, es, h.flags, h.section_index, h.lane, h.lane_changes, h.final_steer, arec);
            // requests are posted on replan ticks (phase_plan) and on the reset tick (phase_begin): look again after those
            if (HAS_MCTS && P.mcts_pause && es.episode_steps % 100 == 0) hold_at = group_min(mcts_hold_step(P, M, env, i));
            HK_ST(h, 6);           // [6] queue binning (+ planner hooks)
            // does any ego of this env wait for a queued multi-player solve?  (or was this the assembly at the end of the budget)
            const bool queued = FISSION ? false : group_or(qn ? 1 : 0) != 0;
            if (!FISSION && solved) pack = queued;
            if (queued || eager_it) { phase = 1; moving = false; go = false; }
            if (FISSION && b1_pending) { phase = 2; moving = false; go = false; }      // (quad-uniform: act depends on the env words only)
        }
        if (moving) {
            phase_move<HAS_RW, HAS_TRAIN>(P, T, env, i, env_ok, es, h, hfx, hfz, agents, act_steer, act_branch, M.st, RD, rwv, LC.low_mode, LC.high_mode);
            phase = 0;
            left -= 1; budget -= 1;
        }
        {
            // the assembly-only iteration at the end of the budget (see above) is taken once, by an env that just ran out of budget on a
            // solve tick; it ends the env's part in this launch
            held = held_now();
            const bool again = left > 0 && budget > 0 && !held;
            const bool last = P.eager != 0 && cadence > 1 && !eager_it && phase == 0 && left > 0 && budget <= 0 && !held && ((es.episode_steps + 1) & cmask) == 0;
#ifdef HK_LOOP_IFELSE
            // the same truth table written as a chain of branches: the form that made round 2's Training-mode instantiation fail its
            // parity test.  Built and run by tests/test_loop_form_gpu.py only.
            if (eager_it) { go = false; eager_it = false; }
            else if (last) eager_it = true;
            else go = go && again;
#else
            go = go && !eager_it && (again || last);
            eager_it = last;
#endif
        }
    }
    HK_ST(h, 13);                      // [13] waiting for the other lane groups of the wave to leave the loop
#ifdef HK_DUMMY_NO_STORE
    if (arec && dirty && h.px == 12345.678f) {       // (timing experiments only: the record stores compiled to a branch never taken)
#else
    if (arec && dirty) {
#endif
        store_hot_tile(htile, h);
        if (HAS_RW && P.rewards) { arec->cum_reward = rwv.cum; arec->step_reward = rwv.step; arec->group_reward = rwv.group; }
    }
    // (arm_ticks: the first launch of a folded call arms in registers — an env that does not enter the loop (`stuck`: parked at a solve tick the
    // call before did not plan for) must still keep the call's ticks, or the completion guard would only ever finish what was left before them)
    if (env_ok && (dirty || arm_ticks != 0) && i == 0) {
        es.reserved[0] = left;
        es.reserved[1] = phase | (pack ? ENV_PACK_HINT : 0);
        envs[slot] = es;
    }
    // the last launch of a fixed-round call is its completion guard (what env_check_kernel does for the other calls)
    if (guard && env_ok && i == 0 && (left != 0 || phase != 0)) raise_guard(status, P.guard_flag);
#ifdef HK_STAMPS
    __builtin_amdgcn_s_waitcnt(0); HK_ST(h, 22);        // [22] the record stores, waited for
    h.st_acc[24] = (unsigned)(h.st_t - st_entry);       // [24] the wave's whole life in this launch
    for (int k = 0; k < HK_NSTAMP; k++) {
        unsigned v = h.st_acc[k];
        for (int o = 32; o > 0; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = w > v ? w : v; }
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&stats[16 + k], (unsigned long long)v);
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(&stats[16 + HK_NSTAMP], 1ull);     // waves that entered the loop
#endif
}

// FISSION, part 2: phase B1 of the solve tick for every env the tick launch parked there (phase 2).  Same lane groups, same device
// functions (phase_assemble: sensor rays through the LDS wall grid, players within 8 m, assembly, lq1_solve in the ego's lane, multi-player
// games to GameSoA + queues); of the kart record only what B1 reads is loaded and only what it decodes (flags, steering) is stored.
#ifndef HK_B1_OCC
#define HK_B1_OCC 3
#endif
template <bool TAB_LDS, bool HAS_MCTS = false>
__global__ __launch_bounds__(256, HK_B1_OCC) void env_b1_kernel(EnvParams P, hk_agent_state* agents, uint32_t* hot, hk_env_state* envs, GameSoA games, int* queue_cnt_all,
                                                        int* queue_all, int round, hk_lq_debug* dbg_out, int* status, MctsDev Marg, const int* perm,
                                                        unsigned long long* stats, int slot0, int slot1, int qbase, int mset, int inwave)
{
    MctsDev M{};
    if (HAS_MCTS) M = Marg;
#ifdef HK_LANEPROF
    hk_lp_ptr = stats;
#endif
#ifdef HK_STAMPS
    const unsigned long long st_entry = __builtin_readcyclecounter();
#endif
    __shared__ __align__(16) KartS ks[256];
#ifndef HK_HOST_EMU          // (the solvers are not part of the host emulation of the tick kernels: there the games always go to the queues)
    static_assert(sizeof(KartS) * 64 == LQS_WAVE_LDS, "the in-wave solver borrows the wave's slice of the kart staging area");
#endif
    HK_DYN_SHARED(smem);
    const int gid = slot0 * GA + blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = gid / GA, i = gid % GA;
    const bool env_ok = slot < slot1;
    const int env = env_ok ? perm[slot] : 0;
    uint32_t* const htile = hot + (size_t)(gid >> 6) * HOT_TILE_WORDS + (gid & 63);
    const int set = qbase + (round & 1);
    int* queue_cnt = queue_cnt_all + set * 16;
    int* queue = queue_all + (size_t)set * (GA - 1) * P.E * P.A;
    hk_env_state es;
    if (env_ok) es = envs[slot];
    else { es.episode_steps = 0; es.inactive_mask = 0; es.experiment_num = 0; es.episodes_done = 0; es.status = 0; es.initial_started = 0; es.reserved[0] = 0; es.reserved[1] = 0; }
    const bool pend = env_ok && (es.reserved[1] & ENV_PHASE_MASK) == 2;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // the meter: the launch before this one (same part, same stream) is complete — fold its total into the decaying maximum, clear the next launch's slot
        unsigned long long* mt = stats + GAME_METER + 4 * (qbase >> 1);
        const unsigned sl = (unsigned)round % 3u;
        // (bit 1 of `inwave`: this part has not launched for a long time — the batch changed shape, hk_api.hip step_ticks — and its words are old: start over.
        // The slot this launch counts into is clean either way: the part's round counter carries on where its last launch, which cleared it, left off)
        const unsigned long long prev = mt[(sl + 2u) % 3u], dec = mt[3] - (mt[3] >> 2);
        mt[3] = (inwave & 2) ? 0ull : (prev > dec ? prev : dec);
        mt[(sl + 1u) % 3u] = 0ull;
    }
    if (__syncthreads_or(pend ? 1 : 0) == 0) return;
    hk_agent_state* arec = (pend && i < P.A) ? &agents[(size_t)env * P.A + i] : nullptr;
    Hot h = {};
#ifdef HK_STAMPS
    h.st_t = __builtin_readcyclecounter();
    h.st_acc[23] = (unsigned)(h.st_t - st_entry);      // [23] B1 kernel: lane group + env words
#endif
    // (the kart's fields are asked for before the table staging and its barrier: the loads fly behind the copy)
    if (arec) {
#define HK_B1_LOAD(T, n) h.n = hot_get<T>(htile, HF_##n)
        HK_B1_LOAD(float, px); HK_B1_LOAD(float, pz); HK_B1_LOAD(float, yaw); HK_B1_LOAD(float, vx); HK_B1_LOAD(float, vz); HK_B1_LOAD(float, wy);
        HK_B1_LOAD(float, final_steer); HK_B1_LOAD(int, section_index); HK_B1_LOAD(uint32_t, flags); HK_B1_LOAD(float, steering);
        if (HAS_MCTS) { HK_B1_LOAD(int, lane); HK_B1_LOAD(int, lane_changes); }
#undef HK_B1_LOAD
    }
    const TabView T = tab_stage<TAB_LDS>(P, smem, P.o_tmask);      // (the segments before the Trigger masks: sections, walls, wall grid, cut table)
    const LaneCfg LC = lane_cfg(P, i);
    float hfx, hfz;
    hk_sincosf(h.yaw, &hfx, &hfz);
    HK_ST(h, 19);                      // [19] B1 kernel: table staging, record loads, sincos
    const int qn = phase_assemble(P, T, ks, env, i, pend, es, h, hfx, hfz, agents, games, queue_cnt, queue, dbg_out, status, M.st, LC);
    // the games-per-launch meter of the host's schedule (hk_env_device.h GAME_METER), this launch's share
    {
        const unsigned long long mg = __ballot(qn != 0);
        if ((threadIdx.x & 63) == 0 && mg != 0ull) atomicAdd(&stats[GAME_METER + 4 * (qbase >> 1) + (unsigned)round % 3u], (unsigned long long)__popcll(mg));
    }
    bool queued = false;
#ifndef HK_HOST_EMU
    if (inwave & 1) {
        // IN-WAVE (round 6): a spread field holds a few dozen multi-player games per solve tick; the waves that assembled them solve them here, behind
        // their own phase_assemble — no queue, no solver launch between this launch and the next tick launch, no barrier (hk_lq_spread.h lqs_inwave).
        double ua = 0.0, ub = 0.0;
        lqs_inwave(P, games, qn, env * P.A + i, reinterpret_cast<unsigned char*>(&ks[threadIdx.x & ~63]), ua, ub, status, stats);
        if (qn != 0) decode_controls(h.final_steer, h.flags, h.steering, ua, ub, (dbg_out && (P.debug & 1)) ? &dbg_out[(size_t)env * P.A + i] : nullptr);
    } else
#endif
    {
        (void)inwave;
#pragma unroll
        for (int n = 2; n <= GA; n++) {
            const int pos = wave_agg_inc(&queue_cnt[n], qn == n);
            if (qn == n) queue[(size_t)(n - 2) * P.E * P.A + pos] = env * P.A + i;
        }
        queued = group_or(qn ? 1 : 0) != 0;
    }
#ifndef HK_HOST_EMU
    // (in-wave the hint says the same thing — this env's last solve tick held a game — and the regroup reads it the other way round: such envs are kept
    // APART, a wave solves its games a pass at a time; env_regroup_scatter_kernel, hk_regroup_pos.h)
    if (inwave & 1) queued = group_or(qn ? 1 : 0) != 0;
#endif
    // the planner hook of a solve tick (HKA:330-402, after SolveLQR; every lane of the group calls it): replan request, bestStates -> plan entries
    if (HAS_MCTS && M.st && pend) phase_plan(P, M, mset, env, i, es, h.flags, h.section_index, h.lane, h.lane_changes, h.final_steer, arec);
    if (arec) { hot_put<uint32_t>(htile, HF_flags, h.flags); hot_put<float>(htile, HF_steering, h.steering); }
    if (pend && i == 0) envs[slot].reserved[1] = 1 | (queued ? ENV_PACK_HINT : 0);
#ifdef HK_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    HK_ST(h, 18);                      // [18] B1 kernel: binning, stores (waited for)
    h.st_acc[25] = (unsigned)(h.st_t - st_entry);
    if ((threadIdx.x & 63) == 0) atomicAdd(&stats[16 + HK_NSTAMP + 1], 1ull);     // waves of the B1 kernel that had work
    for (int k = 0; k < HK_NSTAMP; k++) {
        unsigned v = h.st_acc[k];
        for (int o = 32; o > 0; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = w > v ? w : v; }
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&stats[16 + k], (unsigned long long)v);
    }
#endif
}

} }  // namespace hk::HK_GA_NS
