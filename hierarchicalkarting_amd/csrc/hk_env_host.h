// hk_env_host.h — what the host side of the batched kart environment and the per-width kernel translation units share:
// the device-buffer block (EnvDevice), scheduling constants and the table of launch entry points (GaOps) each width exports.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../include/hk.h"
#include "hk_env_device.h"
#include "hk_lq_core.h"

namespace hk {

struct EnvDevice {
    hk_agent_state* agents = nullptr;   // [E][A] by env id: the cold fields (plans, rewards); its hot fields are a staging copy (hk_env_device.h)
    // Stored by lane-group SLOT and moved physically by a regroup (double buffered; launch_regroup swaps the pairs):
    uint32_t* hot = nullptr;       // hot tiles [ceil(E / (64 / GA))][32 fields][64 lanes]
    uint32_t* hot_alt = nullptr;
    hk_env_state* envs = nullptr;  // [E] env words of the lane group in each slot
    hk_env_state* envs_alt = nullptr;
    hk_env_state* envs_stage = nullptr;   // [E] by env id: what hk_get / hk_set_env_state copy
    int* slot_of = nullptr;        // [E] env id -> slot (perm below is the inverse)
    hk_episode_result* results = nullptr;
    hk_lq_debug* lq_debug = nullptr;
    float* obs = nullptr;
    float* act_steer = nullptr;
    int32_t* act_branch = nullptr;
    float* reward_out = nullptr;   // [2][E][A]: m_Reward, m_GroupReward as read by hk_get_rewards / hk_rewards_device
    int* status = nullptr;
    unsigned long long* game_stats = nullptr;   // [16]: multi-player games solved by the lqn kernels, by player count (hk_prof_games)
    double* games = nullptr;       // queued multi-player games, structure-of-arrays (GameSoA, hk_env_solve.h)
    int* queue_cnt = nullptr;      // [4 sets][16] number of queued multi-player games per player count (sets 2, 3: the second half of a split batch)
    int slot0 = 0, slot1 = 0, qbase = 0;   // the lane groups and queue sets of the next {tick, solver} launches (hk_api.hip issue_rounds; 0, E, 0 = everything)
    int* queue = nullptr;          // [2 sets][GA - 1][E*A] game ids with N = 2 .. GA
    int round = 0;                 // launches so far: round & 1 selects the queue set (double buffered over rounds)
    long long ticks_since_reset = 1ll << 40;   // ticks stepped since the last hk_reset of every env, up to the current hk_step call (launch_lqn: bulk or sparse)
    int call_ticks = 0, call_ticks_issued = 0; // the current call: its ticks, and a lower bound of those its rounds so far have retired
    int* env_ids = nullptr;
    int env_ids_cap = 0;
    // tables
    unsigned char* tab = nullptr;  // packed track tables (EnvParams::tab)
    int* perms = nullptr;
    int tab_lds = 0;               // dynamic LDS bytes the env kernels are launched with (0: read tables from global)
    // MCTS planner (hk_env_mcts.h): all null / 0 when no agent is HighMode MCTS
    MctsDev mcts{};
    // gameParams classes: MCTS agents that share (velocityBucketSize, timePrecision) share one set of move tables; the reference's
    // MCTS-RL vs MCTS-LQR set-ups run two classes in one env (bucket 1 against 2).  d.mcts carries class 0's tables; a search launch
    // runs once per class with that class's tables and agent mask (flush_mcts).
    struct MctsClass { int* dt_tab; float* load_tab; float* rad_tab; unsigned long long* mask_tab; unsigned char* order_tab; int nv, na, ntab, lds_tier; uint32_t agents; };
    MctsClass mcls[HK_MCTS_MAX_CLASSES] = {};
    int n_mcls = 0;
    RwDev rw{};                    // reward shaping tables (null when hk_config.rewards == 0)
    int mset = 0;                  // planner queue set the tick kernel currently fills
    int mcts_rounds = 0;           // rounds of the tick kernel since the last search launch
    int mcts_ticks = 0;            // ticks armed by short hk_step calls since the last search launch (see step_ticks)
    bool mcts_defer = false;       // the current hk_step call is short: its rounds do not launch searches themselves
    SecGeo* sec_geo = nullptr;
    // lane-group -> env assignment of the tick kernel, regrouped by solve phase every REGROUP_ROUNDS rounds (hk_env_run.h)
    int* perm = nullptr;           // [E] slot -> env id (identity until the first regroup)
    int* perm_alt = nullptr;
    bool b1_due = false;           // FISSION: the tick launch just issued parked its envs at their solve tick: env_b1_kernel is next on that stream
    int lqn_sparse_blocks = 1024;  // workgroups per queue of a solver launch once the field has spread (HK_LQN_SPARSE_BLOCKS)
    bool lqn_spread = true;        // the solver launch of a spread field runs the lane-per-(player, row) solver (hk_lq_spread.h); HK_LQN=pair: the pair / matrix-core kernel always
    bool dense = false;            // the games-per-launch meter's last word: many multi-player games per launch — a solver launch takes the pair / matrix-core kernel (32 games a wave), not the spread solver
    bool inwave_ok = false;        // the current call may solve in-wave (hk_api.hip step_ticks: the handle's shape, the switches, the games-per-launch meter)
    bool inwave_always = false;    // HK_INWAVE=1 (tests): also while the field stands close
    unsigned meter_fresh = 0;      // bit p: part p's meter words are old (the batch changed shape): its next B1 launch starts them over
    bool inwave = false;           // the B1 launches of the current rounds solve their multi-player games themselves (hk_lq_spread.h lqs_inwave): no queue, no solver launch
    bool lqn_launched = false;     // the last launch_lqn launched a kernel (or skipped a provably empty one): there is a solver stage to time
    bool b1_small = false;         // the rounds issued while a search launch runs on the side stream in 4-wave workgroups (every CU): B1 reads its tables from global memory, the solver launch is the <= 256-register form (hk_env_launch.h)
    int mcts_side_waves = 8;       // waves per workgroup of a search launch that runs beside tick launches (HK_MCTS_SIDE_WAVES; 0: as alone)
    bool exact_plan = false;       // the current fixed-round call follows the exact plan of a field in lock-step: its last round is the tick launch alone (hk_api.hip step_ticks)
    bool fission = false;          // the current call runs the tick kernel without phase B1 + env_b1_kernel (hk_env_run.h FISSION; hk_api.hip step_ticks)
    bool fold_split = false;       // a folded call on the two-stream schedule: each part's last tick launch is its completion guard (hk_api.hip issue_rounds_split)
    int arm_ticks = 0;             // > 0: the next tick launch adds these ticks to every env's count (a fixed-round call arms itself)
    bool last_solve_skippable = false;   // fixed-round call of a plain handle: no env can park in its last round, so that round queues no game (launch_lqn)
    int guard_rounds_left = 0;     // > 0: fixed-round call; the tick launch that brings it to 0 flags the envs that are not done (the guard)
    int* perm_counts = nullptr;    // [2 * REGROUP_KEYS]: counts, cursors
    int regroup_rounds = 48;       // rounds between two periodic re-assignments (REGROUP_ROUNDS; HK_REGROUP_ROUNDS)
    int rounds_since_regroup = 0;
    int regroup_mode = -1;         // how the last regroup ordered the envs that hold multi-player games: 0 packed (queues), 1 spread (in-wave solves); -1: none yet
    EnvParams P{};
};

constexpr int LQN_BULK_GAMES = 2048; // a round with more 3- (4-) player games than this runs them 5 (4) to a wave (hk_lq2_pair.h lqn_round_kernel)
constexpr int MCTS_FLUSH_ROUNDS = MCTS_MIN_LATENCY / RUN_CAP - 1;      // 4 at RUN_CAP 8
constexpr int MCTS_ARENA_WAVES = 2048;
constexpr int SPLIT_WAYS_MAX = 4;   // parts a split batch can have (hk_api.hip issue_rounds_split): each owns a pair of queue sets
constexpr int BULK_TICKS = 384;        // after a full reset the field needs about this long to spread out (launch_lqn)
// do the B1 launches issued now solve their games in-wave?  (the call may — inwave_ok — and the field has had the time to spread, or HK_INWAVE=1)
inline bool inwave_now(const EnvDevice& d) { return d.inwave_ok && (d.inwave_always || !(d.ticks_since_reset + d.call_ticks_issued < BULK_TICKS)); }
constexpr int REGROUP_ROUNDS = 48;     // the tick kernel's lane groups are re-assigned by solve phase every so many rounds (~200 ticks)
constexpr int MCTS_DEFER_TICKS = 38;   // short hk_step calls share one search launch until this many ticks have been armed
static_assert(MCTS_DEFER_TICKS < MCTS_MIN_LATENCY, "a deferred search must still finish before its plan is due");
static_assert((MCTS_FLUSH_ROUNDS + 1) * RUN_CAP <= MCTS_MIN_LATENCY, "a queued search must finish before its plan is due");

inline int launch_check(std::string& err, const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string(what) + ": " + hipGetErrorString(e); return HK_ERR_HIP; }
    return HK_OK;
}

}  // namespace hk


namespace hk {

// Launch entry points of ONE lane-group width.  The env kernels are compiled per width in their own translation units
// (hk_ga4.hip: quads, up to 4 agents per env; hk_ga8.hip: 8 lanes, the synthetic 8-agent configuration) so that they build in
// parallel and a change to one width does not recompile the other; the API translation unit reaches them through this table.
struct GaOps {
    size_t (*mcts_req_bytes)();
    int (*mcts_searches_per_wave)();
    size_t (*mcts_lds_bytes)(int ntab, int L, int na, int tier, int waves);
    int (*mcts_root_words)();
    size_t (*game_doubles_per_ego)();
    size_t (*queue_ints_per_set)(size_t na);
    int (*launch_mcts_table)(EnvDevice& d, int ego0, int ntab, hipStream_t stream, std::string& err);
    int (*launch_mcts_invalidate)(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err);
    int (*flush_mcts)(EnvDevice& d, hipStream_t stream, std::string& err);
    int (*flush_mcts_on)(EnvDevice& d, hipStream_t stream, hipStream_t side, std::string& err);
    int (*launch_reset)(EnvDevice& d, const int* dids, int cnt, int experiment_num, hipStream_t stream, std::string& err);
    int (*launch_regroup)(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err);
    int (*launch_run)(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err);
    int (*launch_b1)(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err);
    int (*launch_lqn)(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err);
    int (*launch_observe)(EnvDevice& d, const hk_config& cfg, uint32_t agent_mask, hipStream_t stream, std::string& err);
    int (*launch_arm)(EnvDevice& d, const hk_config& cfg, int n_ticks, hipStream_t stream, std::string& err);
    int (*launch_done_check)(EnvDevice& d, const hk_config& cfg, int lazy, hipStream_t stream, std::string& err);
    int (*launch_rewards_read)(EnvDevice& d, int cnt, float* reward, float* group_reward, hipStream_t stream, std::string& err);
    int (*launch_hot_gather)(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err);    // hot tiles -> the AoS records' hot fields
    int (*launch_hot_scatter)(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err);   // ... and back
    int (*launch_envs_gather)(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err);   // env words by slot -> envs_stage by env id
    int (*launch_envs_scatter)(EnvDevice& d, const hk_config& cfg, hipStream_t stream, std::string& err);  // ... and back (progress words sanitized)
    size_t (*hot_tile_words)(int E);
};
const GaOps& ga_ops_g4();      // hk_ga4.hip
const GaOps& ga_ops_g8();      // hk_ga8.hip
inline const GaOps& ga_ops(const EnvDevice& d) { return d.P.A > 4 ? ga_ops_g8() : ga_ops_g4(); }

}  // namespace hk
