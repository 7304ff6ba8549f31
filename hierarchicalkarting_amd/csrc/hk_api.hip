// hk_api.hip — C ABI of libhk.so (include/hk.h) on top of the gfx950 kernels.  No CPU fallback anywhere:
// without a HIP device every compute entry point returns HK_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <utility>
#include <algorithm>
#include <new>
#include "../../include/hk.h"
#include "hk_lq_core.h"
#include "hk_env_kernels.h"
#include "hk_policy.h"
#include <dlfcn.h>

namespace hk {
// hk_lq_batch.hip
int lq_batch_launch(int batch, int N, const double* dA, const double* dB, const double* dQ, const double* dq, const double* dR,
                    const double* dx0, int horizon, double* du0, int* d_status, hipStream_t st);
}

namespace {

thread_local std::string g_last_error;

// HIP-event timing of the kernels on the handle's own stream, without a host sync per launch: every profiled launch is
// bracketed by an event pair taken from a pool; hk_prof_read synchronises once and folds the elapsed times.
struct Prof {
    bool on = false;
    std::vector<hipEvent_t> pool;                                         // free events
    struct Span { hipEvent_t first, second; bool owns_first; };           // owns_first false: `first` is the `second` of the span before
    std::vector<Span> rec[HK_PROF_STAGES];                                // recorded, not yet folded
    double ms[HK_PROF_STAGES] = {};
    int64_t n[HK_PROF_STAGES] = {};
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return nullptr;   // (timing stays on; no system-scope fence per record)
        return e;
    }
    // bracket helpers: begin() records the first event of a pair on `st`, end() the second
    hipEvent_t begin(hipStream_t st)
    {
        if (!on) return nullptr;
        hipEvent_t e = get();
        if (e && hipEventRecord(e, st) != hipSuccess) { pool.push_back(e); return nullptr; }
        return e;
    }
    void end(int stage, hipEvent_t e0, hipStream_t st)
    {
        if (!e0) return;
        hipEvent_t e1 = get();
        if (!e1 || hipEventRecord(e1, st) != hipSuccess) { pool.push_back(e0); if (e1) pool.push_back(e1); return; }
        rec[stage].push_back(Span{e0, e1, true});
    }
    // back-to-back launches on one stream share their boundary event: chain() closes the span that began at `prev` and returns the
    // event it recorded as the start of the next one (an event between two kernels costs ~5 us of the GPU's time: a round of
    // {tick kernel, solver kernel} carries two of them this way, not four).  `first`: prev came from begin().
    hipEvent_t chain(int stage, hipEvent_t prev, bool first, hipStream_t st)
    {
        if (!prev) return nullptr;
        hipEvent_t e1 = get();
        if (!e1 || hipEventRecord(e1, st) != hipSuccess) { if (first) pool.push_back(prev); if (e1) pool.push_back(e1); return nullptr; }
        rec[stage].push_back(Span{prev, e1, first});
        return e1;
    }
    void fold()
    {
        for (int s = 0; s < HK_PROF_STAGES; s++) {
            for (auto& pr : rec[s]) {
                float t = 0;
                if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) { ms[s] += t; n[s]++; }
                if (pr.owns_first) pool.push_back(pr.first);
                pool.push_back(pr.second);
            }
            rec[s].clear();
        }
    }
};

}  // namespace

// Scheduling switches: read from the environment ONCE, in hk_create, into the handle (listed in include/hk.h).  None of them changes a result bit; they
// exist for the same-box A/B measurements under profiles/ and for the parity tests that run every schedule against the oracle.  Round 6 retired the ones whose
// A/B is settled (HK_PARK, HK_NO_EAGER, HK_TAIL_WORST_CASE, HK_KEEP_LAST_SOLVE, HK_RUN_CAP_SHORT / _SPREAD, HK_SPLIT_WAYS, HK_SPLIT_MIN_TICKS, HK_LAZY_MIN_TICKS,
// HK_REGROUP_ROUNDS, HK_LQN_SPARSE_BLOCKS, HK_NO_SPLIT, HK_NO_FISSION_SHAPED / _MCTS / _CHUNKS, HK_DEBUG_NO_CHECK, HK_STAMPS_DUMP) with their code paths; the numbers
// they were settled with are in profiles/README.md.  hk_schedule_info() reports what a call ran.
constexpr int LAZY_MIN_TICKS = 64;        // calls at least this long issue the rounds a spread field needs and finish the laggards after a look at the device
constexpr int SPLIT_MIN_TICKS = 8;        // with HK_LAZY_JOIN=0: calls of plain handles at least this long run as two halves on two streams (a 20-tick call: 1 050 -> 1 120 M env-steps/s, round 4).
                                          // Since round 6 the parts are joined lazily (split_join) and EVERY call of a plain handle of >= 8 192 envs is split: hk_step(1) 57.5 -> 45.7 us
                                          // per call, hk_step(2) 98.3 -> 81.0 (joined at the end of every call the one-tick call took 72.6 us: the reason for the threshold)
constexpr int SPLIT_WAYS = 2;             // parts of a split batch (three / four parts on as many streams: 1 504 / 1 107 M against 1 532, round 4; three with round 6's in-wave solves: 1 410 against 2 270)
constexpr int LQN_SPARSE_BLOCKS = 1024;   // workgroups per queue of a solver launch once the field has spread
struct Tuning {
    bool fission = true;         // HK_FISSION=0: every handle on the fused tick kernel (phase B1 inside the tick loop) instead of tick kernel + env_b1_kernel per solve cadence
    int split = -1;              // HK_SPLIT=1: the batch as two halves on two streams in EVERY call of a plain handle; 0: one stream always; unset: every call of a plain handle of >= 8 192 envs (with HK_LAZY_JOIN=0: calls of >= SPLIT_MIN_TICKS ticks, and while the field stands close)
    int inwave = -1;             // HK_INWAVE=0: multi-player games always go through the queues and a solver launch (the schedule before round 6); 1: env_b1_kernel solves them in-wave in every round (tests); unset: in-wave while the games-per-launch meter says the field has spread
    bool lqn_spread = true;      // HK_LQN=pair: the solver launch of a spread field stays on the pair / matrix-core kernel (the schedule before round 6)
    bool lazy = true;            // HK_FIXED_ROUNDS=1: every call issues the worst-case round count up front (no look at the device)
    bool optimistic = true;      // HK_NO_OPTIMISTIC=1: fixed-round calls always issue the worst-case round count (the schedule before round 5)
    int optimistic_skew = 0;     // HK_OPTIMISTIC_SKEW=k (tests): the believed episode step is off by k, so the exact plans are wrong and the recovery path runs
    bool mcts_pause = true;      // HK_MCTS_NO_PAUSE=1: long calls of planner handles keep the deadline schedule
    bool mcts_overlap = true;    // HK_MCTS_NO_OVERLAP=1: long calls of planner handles launch a replan's searches when its stretch of rounds has ended, on the handle's stream (the schedule before round 5)
    int mcts_side_waves = -1;    // HK_MCTS_SIDE_WAVES=4 / 8 / 0: waves per workgroup of a search launch that runs beside tick launches (unset: 4 where a tick block fits beside one, else 8)
    int debug_max_rounds = 0;    // HK_DEBUG_MAX_ROUNDS (diagnostic): cap on the rounds of a call, to look at the state in between
    void read()
    {
        auto flag = [](const char* n) { return std::getenv(n) != nullptr; };
        auto num = [](const char* n, int dflt, int lo, int hi) { const char* e = std::getenv(n); const int v = e ? std::atoi(e) : dflt; return v >= lo && v <= hi ? v : dflt; };
        fission = num("HK_FISSION", 1, 0, 1) != 0;
        split = num("HK_SPLIT", -1, 0, 1);
        inwave = num("HK_INWAVE", -1, 0, 1);
        { const char* e = std::getenv("HK_LQN"); lqn_spread = !(e && std::strcmp(e, "pair") == 0); }
        lazy = !flag("HK_FIXED_ROUNDS");
        optimistic = !flag("HK_NO_OPTIMISTIC"); optimistic_skew = num("HK_OPTIMISTIC_SKEW", 0, 0, 3);
        mcts_pause = !flag("HK_MCTS_NO_PAUSE"); mcts_overlap = !flag("HK_MCTS_NO_OVERLAP");
        { const char* e = std::getenv("HK_MCTS_SIDE_WAVES"); mcts_side_waves = e ? (std::atoi(e) == 4 ? 4 : (std::atoi(e) == 0 ? 0 : 8)) : -1; }
        debug_max_rounds = num("HK_DEBUG_MAX_ROUNDS", 0, 0, 1 << 20);
    }
};

struct hk_context {
    int device = 0;
    Tuning tune;
    hipStream_t stream = nullptr;
    hipStream_t qstream[hk::SPLIT_WAYS_MAX - 1] = {};   // the other parts of a split batch run here (issue_rounds_split)
    bool env_ready = false;
    hk_config cfg{};
    std::vector<hk_section> sections;
    std::vector<hk_wall_seg> walls;
    hk::EnvDevice dev{};           // device-side tables + state
    int* d_status = nullptr;       // LQ singular flag etc.
    std::string err;
    Prof prof;
    // scratch for the host-pointer LQ entry point
    void* lq_scratch = nullptr;
    size_t lq_scratch_bytes = 0;
    // RL policies (hk_policy.h)
    hk::PolicyDevice policy[HK_MAX_POLICIES];
    int n_policies = 0;
    int decision_period = 1;
    long long academy_step = 0;    // ticks stepped since hk_create (Academy.StepCount)
    // lazy completion of hk_step (handles without planner / attached actors): the call issues the rounds a field without
    // multi-player games needs and a guard kernel that reports what is left; the NEXT entry point that touches the state
    // finishes the stragglers (finish_ticks)
    // The optimistic round plan (step_ticks): lock_tick = the episode step every env is believed to stand on (-1: not known) — 0 after a reset of every env,
    // + n per hk_step(n), unknown after anything else that moves episode steps.  A fixed-round call of a plain handle then issues exactly the launches a field
    // in lock-step needs (a tick launch per stretch between solve ticks, a B1 + solver launch per solve tick inside the call) instead of the worst case, and
    // the completion guard VERIFIES it: opt_pending = such a call has been issued and its guard not looked at yet.  The next entry point other than hk_step
    // looks (verify_optimistic): an env the plan missed kept its ticks, the belief is dropped and the laggards are finished like those of a long call.
    long long lock_tick = -1;
    bool opt_pending = false;
    int exact_idx = 0, exact_total = 0;      // round counter of the current exact plan (issue_rounds is called in pieces)
    long long mcts_async_deadline = -1;      // short calls of planner handles: the believed episode step at which the search launch running on mcts_stream is first used (-1: none in flight)
    bool step_pending = false;
    bool split = false;            // the current call runs the batch as two halves on two streams (issue_rounds)
    int round_half[hk::SPLIT_WAYS_MAX] = {};    // each part's own round counter (the parity picks its queue set)
    hipEvent_t ev_fork = nullptr, ev_join[hk::SPLIT_WAYS_MAX - 1] = {};
    hipStream_t mcts_stream = nullptr;                  // the search launch of a replan runs here, beside the tick launches up to the plans' deadline (step_ticks, pause mode)
    hipEvent_t ev_mcts_go = nullptr, ev_mcts_done = nullptr;
    int* done_host = nullptr;      // pinned: [0] max ticks left over the envs, [1] an env waits for a queued game
    // the games-per-launch meter (hk_env_device.h GAME_METER): env_b1_kernel keeps, per part of the batch, a decaying maximum of the multi-player games its
    // launches assembled.  A copy travels to pinned memory with every look at the device (lazy completion, the stretches of a planner handle's long call) and,
    // WITHOUT a sync, after a short call — a heuristic may be a call late.  Few games per launch: the B1 waves solve their own (in-wave); many (the Complex
    // track's traffic, envs that reset and bring packs back): queues + the pair solver's launch, 32 games a wave.
    unsigned long long* meter_host = nullptr;     // pinned [4 * GAME_METER_PARTS]
    hipStream_t meter_stream = nullptr;
    int meter_ticks = 0;           // ticks issued since the last copy of a short call
    // Long lazily completed calls (hk_step of thousands of ticks): the host issues far ahead of the GPU and would hold the schedule it chose on entry through
    // whatever the field turns into (second episodes: the resets bring every pack back at once).  It therefore stays at most THROTTLE_AHEAD rounds ahead — a
    // marker event every THROTTLE_EVERY rounds, a wait for the marker two back — and looks at the meter at every marker.  The GPU never drains.
    bool throttle = false;
    hipEvent_t ev_thr[4] = {};
    bool thr_valid[4] = {};
    bool lazy_join = true;         // HK_LAZY_JOIN=0: every split call joins its parts at its end (the schedule before)
    bool split_open = false;       // the parts of the last split call have not been joined into the handle's stream yet (short folded calls: split_join)
    bool meter_was_split = false;  // the call before ran as SPLIT_WAYS parts (their words are the current ones)
    bool meter_looked = false;     // meter_look has run (its band needs a previous answer)
    bool meter_sparse = true;      // what the last copy said (until one arrives: sparse once the field has had BULK_TICKS to spread — launch_b1's rule)
    bool meter_dense = false;      // ... so many games per launch that a solver launch wants the pair solver's 32 games a wave
    int meter_games = 0;           // ... the decaying maximum itself (sizes the spread solver's grid)
    std::string sched;             // hk_schedule_info: the schedule of the last hk_step (written by step_ticks)
    void* pol_scratch = nullptr;   // hk_policy_forward staging
    size_t pol_scratch_bytes = 0;
    // RCCL communicator for hk_gather_results (librccl.so loaded lazily)
    void* comm = nullptr;
    int comm_world = 0, comm_rank = 0;
    void* gather_buf = nullptr;
    size_t gather_bytes = 0;
    void* gather_cnt = nullptr;    // per-rank byte counts of the gather (ranks may hold different env counts)
    size_t gather_cnt_bytes = 0;
};

static int finish_ticks(hk_context* h);      // lazy completion of the last hk_step (defined with step_ticks)
static int meter_copy(hk_context* h, bool in_order);      // the games-per-launch meter on its way to pinned memory (defined with step_ticks)
static int throttle_mark(hk_context* h, int r);           // long lazy calls: stay a bounded number of rounds ahead of the GPU, look at the meter (defined with step_ticks)
static int verify_optimistic(hk_context* h); // the completion guard of optimistic fixed-round calls, looked at; laggards finished (defined with step_ticks)
// a search launch that runs on the side stream beside the chunks of a planner + actor handle (step_ticks): the handle's stream waits for it — before another
// search launch (they share the tree arena), before the chunk that uses its plans, before anything reads the planner state
// Two halves on two streams, JOINED LAZILY (round 6).  A short folded call used to end with its parts' streams joined into the handle's stream — and the next
// call forked them again: a host that steps tick by tick (the reference's FixedUpdate) paid an event pair each way per call and, worse, a GPU-side barrier
// between the calls, so the halves could never drift apart the way they do inside a long call (one half's B1 tail behind the other half's ticks).  Now a
// folded split call leaves its parts open; the next such call just continues on both streams.  Everything else that touches the state — every entry point
// but hk_step, an unsplit or lazily completed call, a regroup — joins first.  (The completion guard's flag is a host word, the meter's copy orders nothing.)
static inline int split_join(hk_context* h)
{
    if (!h->split_open) return 0;
    h->split_open = false;
    for (int k = 0; k < hk::SPLIT_WAYS_MAX - 1; k++) {
        if (!h->qstream[k] || !h->ev_join[k]) continue;
        if (hipEventRecord(h->ev_join[k], h->qstream[k]) != hipSuccess) return -1;
        if (hipStreamWaitEvent(h->stream, h->ev_join[k], 0) != hipSuccess) return -1;
    }
    return 0;
}
static inline int mcts_join_async(hk_context* h)
{
    if (h->mcts_async_deadline < 0) return 0;
    h->mcts_async_deadline = -1;
    return hipStreamWaitEvent(h->stream, h->ev_mcts_done, 0) == hipSuccess ? 0 : -1;
}

namespace {

int fail(hk_context* h, int code, const std::string& msg)
{
    g_last_error = msg;
    if (h) h->err = msg;
    return code;
}

#define HK_HIP(h, call)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail((h), HK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

int device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ensure_ctx_basics(hk_context* h)
{
    HK_HIP(h, hipSetDevice(h->device));
    if (!h->stream) HK_HIP(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    if (!h->d_status) {
        HK_HIP(h, hipMalloc(&h->d_status, sizeof(int) * 4));
        HK_HIP(h, hipMemsetAsync(h->d_status, 0, sizeof(int) * 4, h->stream));
    }
    return HK_OK;
}

hk_context* g_default_ctx = nullptr;   // used by hk_lq_solve_batch(NULL, ...)

// The five RCCL entry points hk_gather_results needs, resolved from librccl.so at first use (signatures as in rccl.h;
// ncclUniqueId is a 128-byte struct passed by value, ncclChar = 0, ncclSuccess = 0).
struct RcclId { char internal[HK_COMM_ID_BYTES]; };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load(std::string& err)
    {
        if (lib) return true;
        lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { err = std::string("hk_comm: cannot load librccl.so: ") + dlerror(); return false; }
        GetUniqueId = (int (*)(RcclId*))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(lib, "ncclCommInitRank");
        AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(lib, "ncclAllGather");
        CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
        GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy) { err = "hk_comm: librccl.so lacks an expected symbol"; lib = nullptr; return false; }
        return true;
    }
    std::string why(int rc) const { return GetErrorString ? std::string(GetErrorString(rc)) : std::to_string(rc); }
};
RcclApi g_rccl;

}  // namespace

extern "C" {

const char* hk_last_error(hk_handle h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int hk_create(const hk_config* cfg, hk_handle* out)
{
    if (!out) return fail(nullptr, HK_ERR_INVALID, "hk_create: out is NULL");
    *out = nullptr;
    if (device_count() <= 0) return fail(nullptr, HK_ERR_NO_DEVICE, "hk_create: no HIP device (libhk has no CPU fallback)");
    hk_context* h = new (std::nothrow) hk_context();
    if (!h) return fail(nullptr, HK_ERR_INVALID, "hk_create: out of memory");
    if (cfg) {
        if (cfg->abi_version != HK_ABI_VERSION) { delete h; return fail(nullptr, HK_ERR_INVALID, "hk_create: abi_version mismatch"); }
        h->device = cfg->device_id;
    }
    int rc = ensure_ctx_basics(h);
    if (rc) { g_last_error = h->err; delete h; return rc; }
    h->tune.read();
    if (cfg) {
        h->cfg = *cfg;
        h->dev.regroup_rounds = hk::REGROUP_ROUNDS;
        rc = hk::env_create(h->cfg, h->sections, h->walls, h->dev, h->stream, h->err);
        if (rc) { g_last_error = h->err; hk_destroy(h); return rc; }
        h->env_ready = true;
        if (hipHostMalloc((void**)&h->done_host, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) h->done_host = nullptr;   // (no pinned memory: fixed rounds)
        else {
            h->done_host[0] = 0; h->done_host[1] = 0; h->done_host[2] = 0; h->done_host[3] = 0;
            // word 2: the completion guards' flag, written by the kernels themselves (EnvParams::guard_flag)
            void* dp = nullptr;
            h->dev.P.guard_flag = hipHostGetDevicePointer(&dp, h->done_host + 2, 0) == hipSuccess ? (int*)dp : nullptr;
        }
        if (hipHostMalloc((void**)&h->meter_host, 4 * hk::GAME_METER_PARTS * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) h->meter_host = nullptr;
        else std::memset(h->meter_host, 0, 4 * hk::GAME_METER_PARTS * sizeof(unsigned long long));
        (void)hipStreamCreateWithFlags(&h->meter_stream, hipStreamNonBlocking);          // (here, not inside a call: creating a stream takes milliseconds)
    }
    *out = h;
    return HK_OK;
}

void hk_destroy(hk_handle h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)split_join(h);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->done_host) (void)hipHostFree(h->done_host);
    if (h->meter_stream) { (void)hipStreamSynchronize(h->meter_stream); (void)hipStreamDestroy(h->meter_stream); }
    for (hipEvent_t e : h->ev_thr) if (e) (void)hipEventDestroy(e);
    if (h->meter_host) (void)hipHostFree(h->meter_host);
    hk::env_destroy(h->dev);
    if (h->d_status) (void)hipFree(h->d_status);
    if (h->lq_scratch) (void)hipFree(h->lq_scratch);
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    if (h->gather_buf) (void)hipFree(h->gather_buf);
    if (h->gather_cnt) (void)hipFree(h->gather_cnt);
    if (h->pol_scratch) (void)hipFree(h->pol_scratch);
    for (int p = 0; p < HK_MAX_POLICIES; p++) hk::policy_free(h->policy[p]);
    h->prof.fold();
    for (hipEvent_t e : h->prof.pool) (void)hipEventDestroy(e);
    for (hipStream_t q : h->qstream) if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
    if (h->mcts_stream) { (void)hipStreamSynchronize(h->mcts_stream); (void)hipStreamDestroy(h->mcts_stream); }
    if (h->ev_mcts_go) (void)hipEventDestroy(h->ev_mcts_go);
    if (h->ev_mcts_done) (void)hipEventDestroy(h->ev_mcts_done);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (hipEvent_t e : h->ev_join) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h == g_default_ctx) g_default_ctx = nullptr;
    delete h;
}

void* hk_stream(hk_handle h)
{
    if (!h) return nullptr;
    if (h->split_open && (hipSetDevice(h->device) != hipSuccess || split_join(h))) return nullptr;       // (work the caller puts on the stream must come behind both parts)
    return (void*)h->stream;
}
const char* hk_schedule_info(hk_handle h) { return h ? h->sched.c_str() : ""; }

int hk_synchronize(hk_handle h)
{
    if (!h) return HK_ERR_INVALID;
    HK_HIP(h, hipSetDevice(h->device));
    if (split_join(h)) return fail(h, HK_ERR_HIP, "hipStreamWaitEvent (the parts of a split call)");
    if (h->step_pending) { int rc = finish_ticks(h); if (rc) return rc; }
    if (h->opt_pending) { int rc = verify_optimistic(h); if (rc) return rc; }
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

// ------------------------------------------------------------------ LQ batch
int hk_lq_solve_batch_device(hk_handle h, int batch, int N, const double* dA, const double* dB, const double* dQ,
                             const double* dq, const double* dR, const double* dx0, int horizon, double* du0, void* stream)
{
    if (!h) return fail(nullptr, HK_ERR_INVALID, "hk_lq_solve_batch_device: handle required");
    if (batch < 0 || N < 1 || horizon < 0) return fail(h, HK_ERR_INVALID, "hk_lq_solve_batch: bad batch/N/horizon");
    if (N > hk::LQ_BATCH_MAXP) return fail(h, HK_ERR_UNSUPPORTED, "hk_lq_solve_batch: N > 8 players not built");
    if (batch == 0) return HK_OK;
    HK_HIP(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    hipEvent_t pe0 = h->prof.begin(st);
    const int rc = hk::lq_batch_launch(batch, N, dA, dB, dQ, dq, dR, dx0, horizon, du0, h->d_status, st);
    const hipError_t le = hipGetLastError();
    if (rc != HK_OK || le != hipSuccess) {
        if (pe0) h->prof.pool.push_back(pe0);          // the opening event of the span goes back to the pool
        if (rc == HK_ERR_UNSUPPORTED) return fail(h, HK_ERR_UNSUPPORTED, "hk_lq_solve_batch: N > 8 players not built");
        return fail(h, HK_ERR_HIP, std::string("hk_lq_solve_batch: kernel launch: ") + hipGetErrorString(le));
    }
    h->prof.end(2, pe0, st);
    return HK_OK;
}

int hk_lq_solve_batch(hk_handle h, int batch, int N, const double* A, const double* B, const double* Q, const double* q,
                      const double* R, const double* x0, int horizon, double* u0_out)
{
    if (!h) {
        if (!g_default_ctx) {
            int rc = hk_create(nullptr, &g_default_ctx);
            if (rc) return rc;
        }
        h = g_default_ctx;
    }
    if (batch < 0 || N < 1) return fail(h, HK_ERR_INVALID, "hk_lq_solve_batch: bad batch/N");
    if (N > hk::LQ_BATCH_MAXP) return fail(h, HK_ERR_UNSUPPORTED, "hk_lq_solve_batch: N > 8 players not built");
    if (batch == 0) return HK_OK;
    if (!A || !B || !Q || !q || !R || !x0 || !u0_out) return fail(h, HK_ERR_INVALID, "hk_lq_solve_batch: NULL pointer");
    HK_HIP(h, hipSetDevice(h->device));
    const size_t n = 4 * (size_t)N, b = (size_t)batch;
    const size_t szA = b * N * 16, szB = b * N * 8, szQ = b * N * n * n, szq = b * N * n, szR = b * N * 4, szx = b * n, szu = b * 2;
    const size_t total = (szA + szB + szQ + szq + szR + szx + szu) * sizeof(double);
    if (total > h->lq_scratch_bytes) {
        if (h->lq_scratch) HK_HIP(h, hipFree(h->lq_scratch));
        h->lq_scratch = nullptr; h->lq_scratch_bytes = 0;
        HK_HIP(h, hipMalloc(&h->lq_scratch, total));
        h->lq_scratch_bytes = total;
    }
    double* d = (double*)h->lq_scratch;
    double *dA = d, *dB = dA + szA, *dQ = dB + szB, *dq = dQ + szQ, *dR = dq + szq, *dx = dR + szR, *du = dx + szx;
    HK_HIP(h, hipMemcpyAsync(dA, A, szA * 8, hipMemcpyHostToDevice, h->stream));
    HK_HIP(h, hipMemcpyAsync(dB, B, szB * 8, hipMemcpyHostToDevice, h->stream));
    HK_HIP(h, hipMemcpyAsync(dQ, Q, szQ * 8, hipMemcpyHostToDevice, h->stream));
    HK_HIP(h, hipMemcpyAsync(dq, q, szq * 8, hipMemcpyHostToDevice, h->stream));
    HK_HIP(h, hipMemcpyAsync(dR, R, szR * 8, hipMemcpyHostToDevice, h->stream));
    HK_HIP(h, hipMemcpyAsync(dx, x0, szx * 8, hipMemcpyHostToDevice, h->stream));
    HK_HIP(h, hipMemsetAsync(h->d_status, 0, sizeof(int), h->stream));
    int rc = hk_lq_solve_batch_device(h, batch, N, dA, dB, dQ, dq, dR, dx, horizon, du, h->stream);
    if (rc) return rc;
    int st = 0;
    HK_HIP(h, hipMemcpyAsync(u0_out, du, szu * 8, hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipMemcpyAsync(&st, h->d_status, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    if (st & 1) return fail(h, HK_ERR_SINGULAR, "hk_lq_solve_batch: zero pivot in the m x m solve");
    return HK_OK;
}

// ------------------------------------------------------------------ environment
#define HK_NEED_ENV(h)                                                                     \
    do {                                                                                   \
        if (!(h)) return fail(nullptr, HK_ERR_INVALID, "NULL handle");                     \
        if (!(h)->env_ready) return fail((h), HK_ERR_INVALID, "handle has no environment"); \
        HK_HIP((h), hipSetDevice((h)->device));                                            \
        if (split_join(h)) return fail((h), HK_ERR_HIP, "hipStreamWaitEvent (the parts of a split call)"); \
        if ((h)->step_pending) { int rc_ = finish_ticks(h); if (rc_) return rc_; }         \
        if ((h)->opt_pending) { int rc_ = verify_optimistic(h); if (rc_) return rc_; }     \
        if (mcts_join_async(h)) return fail((h), HK_ERR_HIP, "hipStreamWaitEvent (planner side stream)"); \
    } while (0)
// hk_step itself: the calls of a host that steps tick by tick follow each other without a look at the device
#define HK_NEED_ENV_STEP(h)                                                                \
    do {                                                                                   \
        if (!(h)) return fail(nullptr, HK_ERR_INVALID, "NULL handle");                     \
        if (!(h)->env_ready) return fail((h), HK_ERR_INVALID, "handle has no environment"); \
        HK_HIP((h), hipSetDevice((h)->device));                                            \
        if ((h)->step_pending) { int rc_ = finish_ticks(h); if (rc_) return rc_; }         \
    } while (0)

int hk_reset(hk_handle h, const int32_t* env_ids, int n, int experiment_num)
{
    HK_NEED_ENV(h);
    int rc = hk::env_reset(h->dev, h->cfg, env_ids, n, experiment_num, h->stream, h->err);
    if (rc) { g_last_error = h->err; return rc; }
    h->lock_tick = env_ids ? -1 : 0;            // every env stands on episode step 0 again / some do: the field is no longer known to be in lock-step
    // the agents were reset: their observation stacks start from zeros again (env_reset left the ids in dev.env_ids)
    for (int p = 0; p < h->n_policies; p++) {
        const int cnt = (env_ids ? n : h->cfg.num_envs) * h->policy[p].q.n_slots;
        if (cnt <= 0) continue;
        hipLaunchKernelGGL(hk::policy_invalidate_kernel, dim3((cnt + 255) / 256), dim3(256), 0, h->stream, h->policy[p].q,
                           env_ids ? h->dev.env_ids : nullptr, env_ids ? n : h->cfg.num_envs);
        HK_HIP(h, hipGetLastError());
    }
    return HK_OK;
}

int hk_set_actions(hk_handle h, const float* steer, const int32_t* branch)
{
    HK_NEED_ENV(h);
    if (!steer || !branch) return fail(h, HK_ERR_INVALID, "hk_set_actions: NULL pointer");
    const size_t cnt = (size_t)h->cfg.num_envs * h->cfg.num_agents;
    HK_HIP(h, hipMemcpyAsync(h->dev.act_steer, steer, cnt * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HK_HIP(h, hipMemcpyAsync(h->dev.act_branch, branch, cnt * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

// `rounds` rounds of {fused tick kernel (up to RUN_CAP ticks per env), queued multi-player solves}
static int issue_rounds_split(hk_handle h, int rounds);
static int issue_rounds(hk_handle h, int rounds)
{
    if (h->split && rounds > 0) return issue_rounds_split(h, rounds);
    hipEvent_t e = h->prof.begin(h->stream);
    bool first = true;
    for (int r = 0; r < rounds; r++) {
        int rc = throttle_mark(h, r);
        if (rc) return rc;
        rc = hk::env_launch_run(h->dev, h->cfg, h->stream, h->err);
        if (rc) { g_last_error = h->err; return rc; }
        e = h->prof.chain(0, e, first, h->stream);
        if (h->dev.exact_plan && ++h->exact_idx == h->exact_total) { h->dev.b1_due = false; h->dev.round += 1; first = false; continue; }     // (the plan's last round: no solve tick is left in the call)
        if (h->dev.b1_due) {
            rc = hk::env_launch_b1(h->dev, h->cfg, h->stream, h->err);
            if (rc) { g_last_error = h->err; return rc; }
            e = h->prof.chain(5, e, false, h->stream);
        }
        rc = hk::env_launch_lqn(h->dev, h->cfg, h->stream, h->err);
        if (rc) { g_last_error = h->err; return rc; }
        if (h->dev.lqn_launched) e = h->prof.chain(1, e, false, h->stream);
        first = false;
    }
    if (first && e) h->prof.pool.push_back(e);          // no round issued: the opening event goes back
    return HK_OK;
}

// The batch as two halves (lane groups [0, E/2) and [E/2, E), each with its own pair of queue sets) on two streams: a round of
// one half is {tick kernel, solver kernel} back to back as before, but while one half waits for its handful of solves (one
// solve's latency: 35 - 59 us of an otherwise idle GPU per round — a fifth of a 20-tick call) the other half's tick kernel has
// the whole GPU.  Nothing is deferred: a queued game still costs its env one round.  Both streams are joined before anything
// else touches the state (the guard kernel, a regroup, a getter).
static int issue_rounds_split(hk_handle h, int rounds)
{
    const int K = SPLIT_WAYS;
    for (int k = 0; k < K - 1; k++) {
        if (!h->qstream[k]) HK_HIP(h, hipStreamCreateWithFlags(&h->qstream[k], hipStreamNonBlocking));
        if (!h->ev_join[k]) HK_HIP(h, hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming));
    }
    if (!h->ev_fork) HK_HIP(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    const int E = h->cfg.num_envs;
    int cut[hk::SPLIT_WAYS_MAX + 1];                       // part k: lane groups [cut[k], cut[k + 1]) (a block of the tick kernel holds 64 lane groups)
    for (int k = 0; k <= K; k++) cut[k] = k == K ? E : (int)(((long long)E * k / K + 127) / 128 * 128);     // (whole blocks of the 512-thread form too)
    if ((h->dev.rounds_since_regroup += rounds) >= h->dev.regroup_rounds) {  // the periodic regroup by solve phase, here where the streams are joined
        if (split_join(h)) return fail(h, HK_ERR_HIP, "hipStreamWaitEvent (the parts of a split call)");
        int rcg = hk::env_launch_regroup(h->dev, h->cfg, h->stream, h->err);
        if (rcg) { g_last_error = h->err; return rcg; }
        h->dev.rounds_since_regroup = rounds;      // (the rounds issued below count toward the next one)
    }
    hipStream_t st[hk::SPLIT_WAYS_MAX];
    hipEvent_t e[hk::SPLIT_WAYS_MAX];
    st[0] = h->stream;
    for (int k = 1; k < K; k++) st[k] = h->qstream[k - 1];
    h->round_half[0] = h->dev.round;          // sets 0 / 1 are also the unsplit launches' sets: continue their parity
    if (!h->split_open) {          // (open: the call before left its parts on these streams — this one continues them)
        HK_HIP(h, hipEventRecord(h->ev_fork, h->stream));
        for (int k = 1; k < K; k++) HK_HIP(h, hipStreamWaitEvent(st[k], h->ev_fork, 0));
    }
    h->split_open = true;
    for (int k = 0; k < K; k++) e[k] = h->prof.begin(st[k]);
    bool first = true;
    int rc = HK_OK;
    // Round 6, from the kernel trace of the driver's 20-tick call (profiles/r06_b_short_call_trace.txt): the HOST is what the GPU waits for at the start of a
    // short call — a launch costs it 6 - 8 us, and with one part's whole round issued before the other's first launch the second stream began 93 us into a
    // 895 us call.  The launches of a round are therefore issued kind by kind: every part's tick launch, then every part's B1 launch, then the solver launches.
    // A folded call (step_ticks: fold_split) arms inside each part's first tick launch and makes each part's last tick launch its completion guard: no
    // env_arm_kernel in front of the fork, no env_check_kernel behind the join.
    const int arm = h->dev.arm_ticks;
    h->dev.arm_ticks = 0;
    bool b1due[hk::SPLIT_WAYS_MAX] = {}, inw[hk::SPLIT_WAYS_MAX] = {};
    auto part = [&](int k) { h->dev.slot0 = cut[k]; h->dev.slot1 = cut[k + 1]; h->dev.qbase = 2 * k; h->dev.round = h->round_half[k]; };
    for (int r = 0; r < rounds && rc == HK_OK; r++) {
        if ((rc = throttle_mark(h, r))) break;
        const bool plan_last = h->dev.exact_plan && h->exact_idx + 1 == h->exact_total;      // (the plan's last round: the tick launches alone)
        for (int k = 0; k < K && rc == HK_OK; k++) {
            part(k);
            if (r == 0) h->dev.arm_ticks = arm;
            if (h->dev.fold_split && r == rounds - 1) h->dev.guard_rounds_left = 1;
            rc = hk::env_launch_run_only(h->dev, h->cfg, st[k], h->err);
            if (rc) break;
            e[k] = h->prof.chain(0, e[k], first, st[k]);
            b1due[k] = h->dev.b1_due; h->dev.b1_due = false;
            if (plan_last) h->round_half[k] = h->dev.round + 1;
        }
        if (h->dev.exact_plan) h->exact_idx += 1;
        first = false;
        if (plan_last || rc) continue;
        for (int k = 0; k < K && rc == HK_OK; k++) {
            inw[k] = false;
            if (!b1due[k]) continue;
            part(k);
            h->dev.b1_due = true;
            rc = hk::env_launch_b1(h->dev, h->cfg, st[k], h->err);          // (decides whether this round's games are solved in-wave)
            if (rc) break;
            inw[k] = h->dev.inwave;
            e[k] = h->prof.chain(5, e[k], false, st[k]);
        }
        for (int k = 0; k < K && rc == HK_OK; k++) {
            part(k);
            h->dev.inwave = inw[k];
            rc = hk::env_launch_lqn(h->dev, h->cfg, st[k], h->err);          // (advances dev.round)
            if (rc) break;
            if (h->dev.lqn_launched) e[k] = h->prof.chain(1, e[k], false, st[k]);
            h->round_half[k] = h->dev.round;
        }
    }
    h->dev.inwave = false;
    h->dev.slot0 = 0; h->dev.slot1 = 0; h->dev.qbase = 0; h->dev.round = h->round_half[0];
    if (first) for (int k = 0; k < K; k++) if (e[k]) h->prof.pool.push_back(e[k]);
    // a folded call (nothing follows its rounds on the handle's stream: no guard kernel, no report) leaves its parts open for the next one
    if (!(h->dev.fold_split && h->lazy_join) || rc) { if (split_join(h) && !rc) rc = fail(h, HK_ERR_HIP, "hipStreamWaitEvent (the parts of a split call)"); }
    if (rc) g_last_error = h->err;
    return rc;
}

// guard kernel + (lazy mode) its report on the way to pinned host memory
static int issue_check(hk_handle h, bool lazy)
{
    if (lazy) HK_HIP(h, hipMemsetAsync(h->dev.status + 1, 0, 2 * sizeof(int), h->stream));
    int rc = hk::env_launch_check(h->dev, h->cfg, lazy, h->stream, h->err);
    if (rc) { g_last_error = h->err; return rc; }
    if (lazy) HK_HIP(h, hipMemcpyAsync(h->done_host, h->dev.status + 1, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (lazy) return meter_copy(h, true);          // (the caller synchronises before it looks at either)
    return HK_OK;
}

// Lazy completion: wait for the last hk_step's report; while some env still has ticks to run (it met multi-player games, each
// of which costs it a round), issue the rounds the laggard needs and look again.
static int finish_ticks(hk_handle h)
{
    const int cadence = h->cfg.num_agents > 2 ? 4 : 1;
    h->dev.last_solve_skippable = false; h->dev.guard_rounds_left = 0; h->dev.exact_plan = false; h->throttle = false;      // (the laggards' rounds are plain rounds)
    bool packed = false;
    for (int guard = 0; guard < 1024 && h->step_pending; guard++) {
        HK_HIP(h, hipStreamSynchronize(h->stream));
        const int maxleft = h->done_host[0], waiting = h->done_host[1];
        if (maxleft <= 0 && !waiting) { h->step_pending = false; break; }
        int rc = hk::env_launch_regroup(h->dev, h->cfg, h->stream, h->err);      // the laggards into the first lane groups
        if (rc) { g_last_error = h->err; return rc; }
        packed = true;
        h->split = false;                // ... which all lie in the first half: the tail runs as one batch on one stream
        // Rounds for the slowest env if it met no further multi-player game (+ 1), not for the worst case (a round per cadence): the
        // batch ends with a look at the device anyway, and two thirds of the worst-case rounds used to find nothing to do (94 of 141 in
        // the headline's 3 072-tick call, ~18 us each).  An env that does park on every solve tick still gets a third of its ticks per batch.
        const int cap = (h->dev.fission && cadence == 1) ? 1 : std::max(h->dev.P.run_cap, cadence);      // (2-agent fission: a tick per round)
        rc = issue_rounds(h, (maxleft + cap - 1) / cap + 1);
        if (rc) return rc;
        rc = issue_check(h, true);
        if (rc) return rc;
    }
    if (h->step_pending) { h->step_pending = false; return fail(h, HK_ERR_HIP, "hk_step: an env did not complete its ticks (internal scheduling error)"); }
    // The laggards — the envs that met the most multi-player games, and will meet the next ones — now sit side by side in the first lane groups.  With the
    // games solved in-wave that is the worst order there is (a wave solves its games a pass at a time, and the whole front of the batch is one half of a
    // split call): the next round regroups again, and with every env done that regroup spreads the envs that hold games evenly (hk_regroup_pos.h).
    // Found in the driver's window: B1 launches of 91 us where the same games, never packed, took 73 (profiles/r06_f_spread_regroup.txt).
    // (whatever the schedule is now: the next call decides again, and its regroup packs or spreads accordingly)
    if (packed) h->dev.rounds_since_regroup = h->dev.regroup_rounds;
    return HK_OK;
}

// may the B1 launches of this call solve their games in-wave?  (the fission schedule of a quad handle with LQ agents; per round launch_b1 still keeps
// the queues while the field stands close after a reset of every env)
static bool inwave_allowed(hk_handle h)
{
    if (!h->dev.fission || h->dev.P.any_lqr == 0 || h->cfg.num_agents < 3 || h->cfg.num_agents > 4 || h->tune.inwave == 0) return false;
    return h->tune.inwave == 1 || h->meter_sparse;
}
constexpr int METER_TICKS = 16;
// the meter's copy to pinned memory: on the handle's stream (the caller synchronises: the copy is current) or, after a short call, on a stream of its own
// (the copy orders nothing and must not stand between two launches of the handle's stream; it shows whatever the device has reached)
static int meter_copy(hk_handle h, bool in_order)
{
    if (!h->meter_host || !h->dev.game_stats) return HK_OK;
    hipStream_t st = h->stream;
    if (!in_order) {
        if (!h->meter_stream) return HK_OK;
        st = h->meter_stream;
    }
    HK_HIP(h, hipMemcpyAsync(h->meter_host, h->dev.game_stats + hk::GAME_METER, 4 * hk::GAME_METER_PARTS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    return HK_OK;
}
// what the last copy says (the three regimes and their borders: below)
static void meter_look(hk_handle h)
{
    if (!h->meter_host) return;
    // (the parts the call before ran as: a part that no longer launches keeps its last word for ever — a host that steps tick by tick after a race start on two
    // streams would read the start's counts from the idle part for the rest of the race; a part that launches again after a change of shape starts its words over)
    const int parts = h->meter_was_split ? SPLIT_WAYS : 1;
    unsigned long long worst = 0;
    for (int p = 0; p < parts; p++) worst = std::max(worst, h->meter_host[4 * p + 3]);
    const double envs_per_launch = h->cfg.num_envs / (double)parts;
    // Three regimes, by the games of a launch of `envs_per_launch` envs (same-box A/Bs: profiles/r06_f_spread_regroup.txt — 20-tick windows along the race
    // start with the games solved in-wave and by a solver launch —, r06_b_short_call_trace.txt, r06_c_dense_fields.txt):
    //   up to 1 per 40 envs    in-wave (per half-batch launch of 32 768 envs: 2 - 7 % ahead of a solver launch at 54 .. 722 games, 11 % behind at 2 779; a wave
    //                          that holds games runs one pass while the regroup keeps such envs apart)
    //   up to 1 per 8 envs     queues + lqn_spread_kernel, its grid sized for the count
    //   beyond                 queues + the pair / matrix-core kernel, 32 games a wave (the Complex track under the planner: 101.8 against 57.5 M; second episodes: 1 412 against 772 M)
    // (with a band: a field that hovers at a border would change sides call by call, and every change of sides is a regroup)
    h->meter_sparse = (double)worst <= envs_per_launch / 40.0 || (h->meter_sparse && h->meter_looked && (double)worst <= envs_per_launch / 28.0);
    h->meter_looked = true;
    h->meter_dense = (double)worst > envs_per_launch / 8.0;
    h->meter_games = (int)std::min<unsigned long long>(worst, 1u << 30);
}

static bool inwave_allowed(hk_handle h);
// what the meter's last look means for the launches from here on: where the multi-player games are solved, and — the two are coupled — how the regroup
// orders the envs that hold them.  Queues want such envs PACKED (they leave their tick launches together), in-wave solves want them APART (a wave solves
// its games a pass at a time; hk_regroup_pos.h): when the schedule changes sides the order of the last regroup is the wrong one, so the next round regroups.
static void apply_meter(hk_handle h)
{
    h->dev.inwave_ok = inwave_allowed(h);
    h->dev.dense = h->meter_dense || !h->dev.fission;      // (the meter lives in env_b1_kernel: a handle on the fused kernel keeps round 5's solver launch)
    h->dev.lqn_sparse_blocks = std::min(4096, std::max(LQN_SPARSE_BLOCKS, h->meter_games));
    if (h->dev.regroup_mode >= 0 && h->dev.regroup_mode != (hk::inwave_now(h->dev) ? 1 : 0)) h->dev.rounds_since_regroup = h->dev.regroup_rounds;
}
// (round 6, end: a marker every 4 rounds and in every lazily completed call — every 16 rounds and in calls of >= 512 ticks before.  The races of a batch that
// was reset together END together: within ~150 ticks the field goes from a few dozen games per launch to a game for every ego, and a host that looked every
// 16 rounds from up to 48 rounds ahead kept the in-wave schedule through ~60 rounds of that — B1 launches of milliseconds (tools/experiments/schedule_trace.py:
// 110 ms of B1 in the 200-tick call that held the onset, 22.7 ms now; second episodes 1 409 -> 1 626 M, the windows at ticks 20 000 / 40 000 1 342 / 1 336 ->
// 1 565 / 1 544 M; profiles/r06_g_long_run.txt).  The host issues a round in ~25 us, the GPU runs one in ~115: 8 - 12 rounds of queued work never drain —
// the protocol window, the race start and the short calls read the same.  Sending the games of three and more players of such a field to a solver launch of
// their own (four of them sit in ONE quad when an env restarts) was built, parity-green, and lost to this: 1 445 - 1 463 M against 1 544 - 1 565.)
// (a marker every 1 / 2 / 4 / 8 rounds, same box: second episodes 1 656 / 1 674 / 1 632 / 1 489 M, ticks 20 000 .. 26 000 1 589 / 1 586 / 1 575 / 1 493 M, protocol window
// 2 279 / 2 303 / 2 314 / 2 321 M: 4 keeps a millisecond of queued work between a descheduled host thread and an idle GPU)
constexpr int THROTTLE_EVERY = 4, THROTTLE_MIN_TICKS = 64;
static int throttle_mark(hk_handle h, int r)
{
    if (!h->throttle || (r % THROTTLE_EVERY) != THROTTLE_EVERY - 1) return HK_OK;
    const int i = (r / THROTTLE_EVERY) & 3, back = (i + 2) & 3;
    if (h->thr_valid[back]) HK_HIP(h, hipEventSynchronize(h->ev_thr[back]));          // the GPU has passed the marker of 2 x THROTTLE_EVERY rounds ago
    meter_look(h);
    apply_meter(h);
    if (!h->ev_thr[i]) HK_HIP(h, hipEventCreateWithFlags(&h->ev_thr[i], hipEventDisableTiming));
    HK_HIP(h, hipEventRecord(h->ev_thr[i], h->stream));
    h->thr_valid[i] = true;
    return meter_copy(h, false);
}

// hk_schedule_info: what step_ticks decided for the call it just issued, in one place (the decisions themselves are spread over the function for
// historical reasons; this record is what bench.py prints, so a measured number names its schedule)
static void record_schedule(hk_handle h, int n_ticks, const char* rounds_mode, int rounds, bool fold, bool planner)
{
    char buf[640];
    const char* kern = h->dev.fission ? "fission (tick kernel + env_b1_kernel per solve cadence)" : "fused";
    const char* games = !h->dev.fission || h->dev.P.any_lqr == 0 ? "queues + solver launch"
                        : (h->dev.inwave_ok ? (h->dev.inwave_always ? "in-wave (env_b1_kernel), every round" : "in-wave (env_b1_kernel) once the field has spread, queues + solver launch before")
                                            : (h->dev.lqn_spread ? "queues + lqn_spread_kernel (lqn_round_kernel while the field stands close)" : "queues + lqn_round_kernel"));
    std::snprintf(buf, sizeof(buf),
                  "{\"call_ticks\": %d, \"rounds\": \"%s\", \"rounds_issued\": %d, \"kernel\": \"%s\", \"streams\": %d, \"ticks_per_launch\": %d, "
                  "\"optimistic_plan\": %s, \"armed_in_first_launch\": %s, \"multi_player_games\": \"%s\", \"games_meter\": \"%s\", \"planner\": %s, \"actors\": %d}",
                  n_ticks, rounds_mode, rounds, kern, h->split ? SPLIT_WAYS : 1, h->dev.P.run_cap, h->dev.exact_plan ? "true" : "false", fold ? "true" : "false",
                  games, h->meter_sparse ? "sparse" : (h->meter_dense ? "dense" : "medium"), planner ? "true" : "false", h->n_policies);
    h->sched = buf;
}

// n_ticks of every env: arm, rounds of {fused tick kernel, queued multi-player solves}, check
static int step_ticks(hk_handle h, int n_ticks)
{
    int rc;
    meter_look(h);
    // Planner searches: a launch of the search kernel lasts as long as one search however few it holds, so requests are
    // batched.  A long call launches them every MCTS_FLUSH_ROUNDS rounds and once more before it returns; short calls (a
    // Unity host stepping tick by tick, or the chunks between two RL decisions) share ONE launch until MCTS_DEFER_TICKS
    // ticks have been armed since the last one — early enough, because a plan is due > MCTS_MIN_LATENCY ticks after its
    // request.  (hk_get_mcts_state launches what is pending before it reads.)
    const bool planner = h->dev.mcts.st != nullptr;
    // (the parts of the call before are still open — split_join: only a short folded call of a plain handle may continue them; whether THIS one is split
    // and folded is known further down, everything else joins here, before it puts anything on the handle's stream)
    if (h->split_open && (planner || h->n_policies > 0 || n_ticks >= LAZY_MIN_TICKS || !h->tune.lazy)) { if (split_join(h)) return fail(h, HK_ERR_HIP, "hipStreamWaitEvent (the parts of a split call)"); }
    const long long T0 = h->lock_tick;                           // the episode step the field is believed to stand on before this call (-1: not known)
    if (h->lock_tick >= 0) h->lock_tick += n_ticks;
    h->dev.exact_plan = false;
    // (how long a request may wait for its launch: a request posted on armed tick 1 is searched before armed tick defer + 1 runs, and its
    // plan is due `latency` ticks after the request — so defer = the handle's smaller latency - 1, at least MCTS_DEFER_TICKS)
    const int defer = std::max(hk::MCTS_DEFER_TICKS, std::min(h->cfg.mcts_latency_ticks, h->cfg.mcts_initial_latency_ticks) - 1);
    const bool short_call = planner && n_ticks <= defer;
    // Round 5: the searches of a replan beside the CHUNKS that follow it (handles with attached actors step in decision chunks; round 4 launched a replan's
    // searches up to `defer` ticks late, on the handle's stream, and every chunk behind them waited ~10 - 100 ms).  If the field is believed to be in lock-step
    // the host knows the chunk that holds the replan step (a multiple of 100): whatever is queued is searched BEFORE that chunk, so that the launch after it
    // holds that chunk's requests only — none older than the chunk — and may therefore run on the side stream until the chunk that reaches request +
    // latency.  A wrong belief costs the overlap only: requests posted at other times are served by the `defer` rule exactly as before, and every
    // search launch first waits for the one in flight (they share the tree arena).
    const int lat_min = std::min(h->cfg.mcts_latency_ticks, h->cfg.mcts_initial_latency_ticks);
    long long t_req = -1;
    if (short_call && T0 >= 0 && h->tune.mcts_overlap && h->n_policies > 0) { t_req = (T0 / 100 + 1) * 100; if (t_req > T0 + n_ticks) t_req = -1; }
    if (h->mcts_async_deadline >= 0 && (!short_call || T0 < 0 || T0 + n_ticks >= h->mcts_async_deadline)) { if (mcts_join_async(h)) return fail(h, HK_ERR_HIP, "hipStreamWaitEvent (planner side stream)"); }
    if (planner && h->dev.mcts_ticks > 0 && (!short_call || h->dev.mcts_ticks + n_ticks > defer || t_req >= 0)) {
        if (mcts_join_async(h)) return fail(h, HK_ERR_HIP, "hipStreamWaitEvent (planner side stream)");
        rc = hk::env_flush_mcts(h->dev, h->stream, h->err);
        if (rc) { g_last_error = h->err; return rc; }
    }
    if (short_call) h->dev.mcts_ticks += n_ticks;
    h->dev.mcts_defer = short_call;
    h->dev.ticks_since_reset += h->dev.call_ticks;      // the previous call's ticks
    h->dev.lqn_sparse_blocks = LQN_SPARSE_BLOCKS;
    h->dev.lqn_spread = h->tune.lqn_spread;
    h->dev.inwave_always = h->tune.inwave == 1;
    { const char* e = std::getenv("HK_LAZY_JOIN"); h->lazy_join = !(e && std::atoi(e) == 0); }
    h->dev.call_ticks = n_ticks; h->dev.call_ticks_issued = 0;
    // Long calls of a planner handle without attached actors run in PAUSE mode: an env that requests a search stops at the next
    // tick boundary until the search has run, the host runs a stretch of rounds (every env reaches its replan tick or the end of
    // the call), launches ALL the searches of the stretch in one batch, looks at what is left and repeats.  Without the pause
    // the requests of one replan wave trickle in over many rounds (envs that queue multi-player games advance 4 ticks a round,
    // the others 8) and every partial batch costs a full search latency: 4-agent Complex, 16 384 envs: 11.8 -> see profiles/.
    const bool pause = planner && !short_call && h->n_policies == 0 && h->done_host != nullptr && h->tune.mcts_pause;
    h->dev.P.mcts_pause = pause ? 1 : 0;
    // the eager assembly (hk_env_run.h) in pause mode too: requests are posted on the same ticks, a round earlier at most
    // (configs[2]: 61.4 -> 63.6 M env-steps/s)
    if (planner) h->dev.P.eager = (pause && h->cfg.num_agents > 2 && h->cfg.num_agents <= 4) ? 1 : 0;
    // Arming: a kernel of its own, except in fixed-round calls that are not split, where the first tick launch adds the ticks itself and the
    // last one raises the "did not complete" flag the guard kernel would (a one-tick call: 4 launches instead of 9 with round 2's tail regroup)
    const bool lazy_call = !planner && h->n_policies == 0 && h->done_host != nullptr && n_ticks >= LAZY_MIN_TICKS && h->tune.lazy;
    const bool split_req = h->tune.split == 1 || (h->tune.split < 0 && (lazy_call || (!planner && h->n_policies == 0 && n_ticks >= (h->lazy_join ? 1 : SPLIT_MIN_TICKS))));      // (HK_SPLIT=1: every call)
    // (round 6: split calls fold too — each part's first tick launch arms its lane groups, each part's last one is the guard: issue_rounds_split)
    const bool fold = !pause && !lazy_call;
    if (fold) h->dev.arm_ticks = n_ticks;
    else {
        rc = hk::env_launch_arm(h->dev, h->cfg, n_ticks, h->stream, h->err);
        if (rc) { g_last_error = h->err; return rc; }
    }
    if (pause) {
        h->throttle = false;
        // the fission schedule for planner handles too (long calls, LQNG low levels; round 4): the same two kernels with the planner hooks
        bool shaped_p = h->cfg.rewards != 0 || h->cfg.env_mode == HK_MODE_TRAINING;
        for (int i = 0; i < h->cfg.num_agents; i++) shaped_p = shaped_p || h->cfg.training_agent[i] != 0;
        shaped_p = false;      // (reward shaping and the Training-mode reset live in phases A / C: the tick kernel's instantiations carry them)
        h->dev.fission = h->tune.fission && !shaped_p && h->dev.P.eager && h->cfg.num_agents > 2 && h->cfg.num_agents <= 4;
        if (h->dev.fission) h->dev.P.run_cap = 4;
        apply_meter(h);
        h->dev.mcts_defer = true;                       // the rounds do not launch searches themselves
        const int cadence = h->cfg.num_agents > 2 ? 4 : 1;
        int maxleft = n_ticks;
        for (int guard = 0; guard < 4096; guard++) {
            // a stretch: enough rounds for EVERY env to reach its next replan (<= 100 ticks away) or the end of the call, also one
            // that meets a multi-player game on every solve tick (a solve cadence per round).  The rounds in which most envs already
            // wait cost tens of microseconds; a second search launch for the late ones would cost a full search latency
            int reach = std::min(maxleft, 100);
            // Round 5: a replan's searches run BESIDE the ticks that follow it.  The reference's search thread works for T = 0.9 s while FixedUpdate
            // goes on (HKA:172-284) and its plan is used mcts_latency_ticks after the request; an env with an outstanding search now steps on until
            // that tick (hk_env_run.h held_now) instead of stopping at once, so the search launch — as long as ONE search whatever the batch, ~10 ms —
            // can share the GPU with up to 44 ticks of every env.  The host does not know when requests are posted; it guesses from the ticks since the
            // last reset of every env (a field that was reset together replans together, on episode steps that are multiples of 100): the stretch is cut
            // to end 40 ticks after the next such step, the searches are launched on a side stream right after the round that is expected to post the
            // requests, and the handle's stream waits for them only before the rounds that could reach the deadline.  A wrong guess costs the overlap,
            // nothing else: every stretch still ends with a search launch for whatever is queued, and no env passes its deadline unserved (device side).
            const int lat = std::min(h->cfg.mcts_latency_ticks, h->cfg.mcts_initial_latency_ticks);
            {
                // Search workgroups beside the ticks: 4 waves (one per SIMD) on EVERY CU when a tick block still fits a CU's LDS beside one — the searches then run
                // a wave to a SIMD (~7 ms instead of ~10.6 at two waves per SIMD on half the CUs); phase B1 reads its tables from global memory and the solver
                // launch takes its <= 256-register form for those rounds (b1_small; hk_env_launch.h).  Otherwise 8 waves on half the CUs (round 5's first form).
                int sw = 4;
                for (int c = 0; c < h->dev.n_mcls; c++) {
                    const auto& K = h->dev.mcls[c];
                    if (hk::ga_ops(h->dev).mcts_lds_bytes(K.ntab, h->dev.P.L, K.na, K.lds_tier, 4) + (size_t)h->dev.tab_lds + 1024 > 160 * 1024) sw = 8;
                }
                if (!h->dev.tab_lds) sw = 8;
                if (h->tune.mcts_side_waves >= 0) sw = h->tune.mcts_side_waves;
                h->dev.mcts_side_waves = sw;
            }
            const bool overlap = h->tune.mcts_overlap && h->dev.fission && cadence == 4 && lat >= 24;
            int r_post = -1, r_free = 0;
            if (overlap) {
                const long long T = (long long)h->dev.ticks_since_reset + (n_ticks - maxleft);      // episode step of a field in lock-step
                const int k = 100 - (int)(T % 100);                                                 // ticks to the next replan step (1 .. 100)
                const int after = ((lat - 1) / cadence - 1) * cadence;                              // whole rounds an env runs on before its deadline (45 -> 40 ticks)
                reach = std::min(maxleft, (k + after - 1) % 100 + 1);
                if (k <= reach) { r_post = (k + cadence - 1) / cadence + 1; r_free = after / cadence; }
            }
            const int stretch_rounds = (reach + cadence - 1) / cadence + 2;
            if (r_post > 0 && r_post < stretch_rounds) {
                if (!h->mcts_stream) HK_HIP(h, hipStreamCreateWithFlags(&h->mcts_stream, hipStreamNonBlocking));
                if (!h->ev_mcts_go) HK_HIP(h, hipEventCreateWithFlags(&h->ev_mcts_go, hipEventDisableTiming));
                if (!h->ev_mcts_done) HK_HIP(h, hipEventCreateWithFlags(&h->ev_mcts_done, hipEventDisableTiming));
                rc = issue_rounds(h, r_post);
                if (rc) return rc;
                HK_HIP(h, hipEventRecord(h->ev_mcts_go, h->stream));
                HK_HIP(h, hipStreamWaitEvent(h->mcts_stream, h->ev_mcts_go, 0));
                rc = hk::env_flush_mcts_on(h->dev, h->stream, h->mcts_stream, h->err);
                if (rc) { g_last_error = h->err; return rc; }
                HK_HIP(h, hipEventRecord(h->ev_mcts_done, h->mcts_stream));
                const int r2 = std::min(stretch_rounds - r_post, r_free);
                h->dev.b1_small = h->dev.mcts_side_waves == 4;      // (4-wave search workgroups sit on EVERY CU: what runs beside them must share its LDS and registers)
                rc = issue_rounds(h, r2);
                h->dev.b1_small = false;
                if (rc) return rc;
                HK_HIP(h, hipStreamWaitEvent(h->stream, h->ev_mcts_done, 0));
                rc = issue_rounds(h, stretch_rounds - r_post - r2);
                if (rc) return rc;
            } else {
                rc = issue_rounds(h, stretch_rounds);
                if (rc) return rc;
            }
            rc = hk::env_flush_mcts(h->dev, h->stream, h->err);
            if (rc) { g_last_error = h->err; return rc; }
            rc = issue_check(h, true);
            if (rc) return rc;
            HK_HIP(h, hipStreamSynchronize(h->stream));
            maxleft = h->done_host[0];
            if (maxleft <= 0 && !h->done_host[1]) break;
            meter_look(h); apply_meter(h);      // (the copy of this stretch's check is current)
        }
        h->dev.mcts_defer = false;
        h->dev.P.mcts_pause = 0;
        record_schedule(h, n_ticks, "pause (stretches of rounds between search launches)", 0, false, true);
        if (maxleft > 0) return fail(h, HK_ERR_HIP, "hk_step: an env did not complete its ticks (internal scheduling error)");
        return HK_OK;
    }
    // Rounds.  An env that meets no multi-player game retires RUN_CAP ticks per round; one that does retires at least a solve
    // cadence.  Handles with a planner or attached actors issue the worst-case count up front (they step in short chunks and
    // must not stall on the host).  Everything else — the LQNG races of the headline — issues what a field without
    // multi-player games needs, and the stragglers are finished lazily by the next call that touches the state
    // (finish_ticks): most of the worst-case rounds found nothing to do, and on a 20-tick call they were 5 launches of 8.
    // (short calls — a host stepping tick by tick — keep the fixed count too: a handful of rounds, no host sync)
    const bool lazy = !planner && h->n_policies == 0 && h->done_host != nullptr && n_ticks >= LAZY_MIN_TICKS && h->tune.lazy;
    // ticks per launch: longer launches once the field has spread out (see RUN_CAP_SPREAD)
    const int spread_cap = hk::RUN_CAP_SPREAD;
    // short calls of plain LQNG handles: one solve cadence per launch — with the eager assembly every env, in a pack or not, retires
    // it, so a 20-tick call is 6 equal rounds and no tail (at 8 ticks per launch: 3 rounds + a regroup + 5 rounds for the laggards)
    const int short_cap = 4;
    const bool eager = true;
    const bool plain = !planner && h->n_policies == 0;
    // (planner / actor handles keep their deadline arithmetic as it was; the 8-lane groups run the older loop without it)
    h->dev.P.eager = (eager && plain && h->cfg.num_agents > 2 && h->cfg.num_agents <= 4) ? 1 : 0;
    // FISSION: every env parks at every solve tick, so a round retires exactly one cadence
    bool shaped = h->cfg.rewards != 0 || h->cfg.env_mode == HK_MODE_TRAINING;       // reward shaping / Training mode: their own instantiations of the fused kernel
    for (int i = 0; i < h->cfg.num_agents; i++) shaped = shaped || h->cfg.training_agent[i] != 0;
    shaped = false;    // round 5: reward shaping and the Training-mode reset live in phases A / C — the tick kernel's <.., HAS_RW, HAS_TRAIN, .., FISSION> instantiations carry them
    h->dev.fission = h->tune.fission && plain && !shaped && h->dev.P.eager && h->cfg.num_agents > 2 && h->cfg.num_agents <= 4;
    // handles without an LQ agent (every low level an RL actor, attached or driven through hk_set_actions): the tick kernel of the fission
    // schedule alone — phase B1 has nothing to solve, no env parks for it, no B1 launch
    if (h->tune.fission && !shaped && h->dev.P.any_lqr == 0 && h->cfg.num_agents > 2 && h->cfg.num_agents <= 4) h->dev.fission = true;     // (with a planner too: its hook stays in the tick loop)
    // handles that step in short fixed-round chunks (attached actors: a chunk per decision; planners outside their long calls): the same two
    // kernels without the eager assembly — an env parks at its solve tick, B1 + solver run, the next round resumes it; the rounds issued
    // are the worst case the fused kernel was given too (a round per solve tick of the chunk + 1)
    if (h->tune.fission && !shaped && (planner || h->n_policies > 0) && h->cfg.num_agents > 2 && h->cfg.num_agents <= 4) h->dev.fission = true;
    // 2-agent fields (cadence 1: every tick is a solve tick and both egos hold the 2-player game, HKA:317,709) stay on the fused kernel.  Round 6 measured
    // the fission schedule for them (bit-equal; three launches per tick, no eager assembly): 303 M env-steps/s against the fused kernel's 331 M — with a game per
    // ego and tick the round is the pair solver's 131 072 games (98 us) and the GameSoA round trip of the assembly (B1 82 us), which a split does not shrink.
    const bool fission_a2 = false;
    apply_meter(h);
    h->throttle = lazy && n_ticks >= THROTTLE_MIN_TICKS && h->dev.fission && h->dev.P.any_lqr != 0;
    for (bool& v : h->thr_valid) v = false;
    const int run_cap = (h->dev.fission && h->dev.P.any_lqr != 0) ? 4 : (lazy && h->cfg.num_agents > 2 && h->dev.ticks_since_reset >= hk::BULK_TICKS) ? spread_cap
                        : (!lazy && plain && h->dev.P.eager) ? short_cap : hk::RUN_CAP;
    h->dev.P.run_cap = run_cap;
    // two halves on two streams (issue_rounds_split): the default for the long calls of plain handles since round 4 (1 472 vs 1 392 M
    // env-steps/s in round 3's protocol window; bench.py then reports the roofline fraction of the whole job, see there), on request
    // (HK_SPLIT=1) for every call.  The notes of the rounds before:  Measured: headline 1 221 -> 1 292 M env-steps/s, race start 439 -> 458 M,
    // a 20-tick call unchanged (the solver kernel needs a SIMD's whole register file and finds none while the other half's tick kernel
    // fills the GPU, so on short launches its latency is not hidden but moved).  On a spread field it is off unless HK_SPLIT=1: two tick
    // kernels that share the GPU each take longer, and bench.py's per-launch roofline (bytes of a launch / its duration) would no longer
    // describe the kernel (hk_prof's stage totals then add up the spans of two concurrent streams).  While the field stands close (BULK_TICKS after a reset of every env: every ego holds a multi-player game and a round's solver
    // launch lasts hundreds of microseconds) the split is used without being asked: race start 440 -> 458 M.
    const bool want_split = split_req, no_split = h->tune.split == 0;
    const bool close_field = h->dev.ticks_since_reset < hk::BULK_TICKS;
    h->split = (want_split || (close_field && !no_split)) && h->dev.P.eager && h->cfg.num_envs >= 8192;
    if (h->split != h->meter_was_split && h->dev.game_stats) {
        // the batch changes shape: the parts that stop launching (or start again after a long time) must not be read with their old words
        // (no launch for it: the parts' next B1 launches start their words over, the host reads only the parts a call ran as — meter_look)
        h->dev.meter_fresh |= ((1u << hk::GAME_METER_PARTS) - 1u) & ~1u;
        h->meter_was_split = h->split;
    }
    // (2-agent fission: a round retires exactly one tick whatever the launch's budget)
    int rounds = lazy ? hk::env_rounds_min(h->cfg, n_ticks, fission_a2 ? 1 : run_cap) : hk::env_rounds_for(h->cfg, n_ticks, run_cap, h->dev.P.eager != 0 || fission_a2);
    // The optimistic plan of a fixed-round call (round 5).  If every env stands on episode step T0, the call's ticks T0 + 1 .. T0 + n hold S solve ticks
    // (multiples of the cadence) and the field needs exactly S rounds of {tick launch up to the solve tick, B1, solver} and one more tick launch: a one-tick
    // call off a solve tick is ONE launch, the driver's 20-tick window 6 + 5 + 5 launches per half instead of 7 + 7 + 7.  The plan is a belief, not a proof
    // (envs finish and reset on their own; a time-out inside the call would add a solve tick): the call's completion guard verifies it, an env the plan
    // missed keeps its ticks (and stays parked at its solve tick: hk_env_run.h `stuck`), and the next entry point that looks at the state drops the belief
    // and finishes it (verify_optimistic).  Nothing is ever wrong, a wrong belief is only slow — so it is used only where it is cheap to check.
    if (!lazy && plain && h->dev.fission && h->dev.P.any_lqr != 0 && h->tune.optimistic && T0 >= 0 && T0 + n_ticks < h->cfg.max_episode_steps && h->tune.debug_max_rounds == 0) {
        const long long Tb = T0 + h->tune.optimistic_skew;
        const int cad = h->cfg.num_agents > 2 ? 4 : 1;
        const int S = (int)((Tb + n_ticks) / cad - Tb / cad);               // multiples of the cadence in (Tb, Tb + n]
        rounds = S + 1;
        h->dev.exact_plan = true;
        h->exact_idx = 0; h->exact_total = rounds;
    }
    if (h->tune.debug_max_rounds > 0) rounds = std::min(rounds, h->tune.debug_max_rounds);     // (diagnostic: look at the state between two rounds)
    {
        // the rounds every env needs at RUN_CAP ticks a round, then — the laggards packed into the first lane groups — the tail
        const int main_rounds = std::min(rounds, fission_a2 ? n_ticks : (n_ticks + run_cap - 1) / run_cap);
        h->dev.guard_rounds_left = (fold && !h->split) ? rounds : 0;          // the tick launch that brings this to 0 is the call's last: it is the guard
        h->dev.fold_split = fold && h->split;
        if (h->split_open && !h->dev.fold_split) { if (split_join(h)) return fail(h, HK_ERR_HIP, "hipStreamWaitEvent (the parts of a split call)"); }
        h->dev.last_solve_skippable = fold && plain && !h->split && h->tune.debug_max_rounds == 0;
        // (with the eager assembly there is nothing to regroup between the two: ONE issue — a split call used to join its streams and fork them again for
        // the tail, which left the first stream idle for ~100 us of the driver's 20-tick call)
        rc = issue_rounds(h, h->dev.P.eager ? rounds : main_rounds);
        if (rc) return rc;
        if (rounds > main_rounds && !h->dev.P.eager) {
            // (with the eager assembly every env retires a cadence per round: there are no laggards to pack, and a one-tick call
            // would pay three more GPU operations for the regroup than for its tick)
            if (!planner && h->n_policies == 0 && !h->dev.P.eager) {
                rc = hk::env_launch_regroup(h->dev, h->cfg, h->stream, h->err);
                if (rc) { g_last_error = h->err; return rc; }
            }
            rc = issue_rounds(h, rounds - main_rounds);
            if (rc) return rc;
        }
    }
    if (t_req >= 0) {
        // this chunk held the replan step: its requests (and nothing older) go to the side stream
        if (mcts_join_async(h)) return fail(h, HK_ERR_HIP, "hipStreamWaitEvent (planner side stream)");
        if (!h->mcts_stream) HK_HIP(h, hipStreamCreateWithFlags(&h->mcts_stream, hipStreamNonBlocking));
        if (!h->ev_mcts_go) HK_HIP(h, hipEventCreateWithFlags(&h->ev_mcts_go, hipEventDisableTiming));
        if (!h->ev_mcts_done) HK_HIP(h, hipEventCreateWithFlags(&h->ev_mcts_done, hipEventDisableTiming));
        HK_HIP(h, hipEventRecord(h->ev_mcts_go, h->stream));
        HK_HIP(h, hipStreamWaitEvent(h->mcts_stream, h->ev_mcts_go, 0));
        h->dev.mcts_side_waves = h->tune.mcts_side_waves >= 0 ? h->tune.mcts_side_waves : 8;      // (decision chunks: 8-wave workgroups on half the CUs)
        rc = hk::env_flush_mcts_on(h->dev, h->stream, h->mcts_stream, h->err);
        if (rc) { g_last_error = h->err; return rc; }
        HK_HIP(h, hipEventRecord(h->ev_mcts_done, h->mcts_stream));
        // the oldest request in that launch was posted on step T0 + 1 at the earliest: its plan is first used on that step + the smaller latency
        h->mcts_async_deadline = T0 + 1 + lat_min;
    }
    if (planner && !short_call) {
        // searches requested in the last rounds of a long call run before it returns
        rc = hk::env_flush_mcts(h->dev, h->stream, h->err);
        if (rc) { g_last_error = h->err; return rc; }
    }
    if (!fold) {
        rc = issue_check(h, lazy);
        if (rc) return rc;
    }
    h->step_pending = lazy;
    if (!lazy && h->dev.fission && (h->meter_ticks += n_ticks) >= METER_TICKS) { h->meter_ticks = 0; if ((rc = meter_copy(h, false))) return rc; }
    record_schedule(h, n_ticks, lazy ? "lazy" : "fixed", rounds, fold, planner);
    if (h->dev.exact_plan) h->opt_pending = true;
    h->dev.exact_plan = false;
    // (an exact plan's last round is the tick launch alone and never reaches the solver launch that consumes this flag: left set, the first solver launch
    // of the rounds that finish a missed env — verify_optimistic — would be skipped and the env would resume on stale controls; found by the fold + skew
    // modes of tests/test_optimistic_plan_gpu.py, round 6)
    h->dev.last_solve_skippable = false;
    h->dev.guard_rounds_left = 0;
    return HK_OK;
}

// The completion guard of the optimistic fixed-round calls issued since the last look (status bit 2: the last tick launch of a folded call, env_check_kernel
// of the others).  Set: some env did not finish — the belief that the field is in lock-step was wrong (an env finished its race and reset, a time-out) — it
// kept its leftover ticks; the belief is dropped and the laggards are finished as those of a long call are (a guard launch that reports what is left,
// finish_ticks).  Not an error: the state every getter sees afterwards is the state the worst-case schedule would have produced.
static int verify_optimistic(hk_handle h)
{
    h->opt_pending = false;
    int st[4] = {0, 0, 0, 0};
    if (h->dev.P.guard_flag) {
        // the guards raise a pinned host word beside the status bit: nothing to copy back when — as good as always — the plan held (the copy to a pageable
        // buffer was a second round trip behind the stream's completion, ~40 us of a 20-tick call's 820)
        HK_HIP(h, hipStreamSynchronize(h->stream));
        if (__atomic_load_n(h->done_host + 2, __ATOMIC_ACQUIRE) == 0) return HK_OK;
        __atomic_store_n(h->done_host + 2, 0, __ATOMIC_RELEASE);
    }
    HK_HIP(h, hipMemcpyAsync(st, h->dev.status, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    if (!(st[0] & 4)) return HK_OK;
    h->lock_tick = -1;
    hipLaunchKernelGGL(hk::status_and_kernel, dim3(1), dim3(1), 0, h->stream, h->dev.status, ~4);
    HK_HIP(h, hipGetLastError());
    if (h->done_host == nullptr) return fail(h, HK_ERR_HIP, "hk_step: an env did not complete its ticks and the handle has no completion buffer");
    int rc = issue_check(h, true);
    if (rc) return rc;
    h->step_pending = true;
    return finish_ticks(h);
}

// Academy step of a decision tick: CollectObservations -> StackingSensor -> actor -> OnActionReceived latch
static int policy_decide(hk_handle h)
{
    hipEvent_t eo = h->prof.begin(h->stream);
    uint32_t need = 0;                      // only the agents some attached actor drives are observed here
    for (int p = 0; p < h->n_policies; p++)
        for (int j = 0; j < h->policy[p].q.n_slots; j++) need |= 1u << h->policy[p].q.slots[j];
    int rc = hk::env_launch_observe(h->dev, h->cfg, need, h->stream, h->err);
    if (rc) { g_last_error = h->err; return rc; }
    const unsigned long long decision = (unsigned long long)(h->academy_step / h->decision_period);
    const int E = h->cfg.num_envs, A = h->cfg.num_agents;
    for (int p = 0; p < h->n_policies; p++) {
        const hk::PolicyDevice& pd = h->policy[p];
        const int pairs = E * pd.q.n_slots;
        const int w = (int)(decision % (unsigned long long)pd.q.stack);
        hipLaunchKernelGGL(hk::policy_stack_kernel, dim3((pairs + 3) / 4), dim3(256), 0, h->stream, pd.q, E, A, h->dev.envs, h->dev.slot_of,
                           h->dev.obs, w);
        HK_HIP(h, hipGetLastError());
    }
    h->prof.end(4, eo, h->stream);
    for (int p = 0; p < h->n_policies; p++) {
        const hk::PolicyDevice& pd = h->policy[p];
        const int pairs = E * pd.q.n_slots;
        const int w = (int)(decision % (unsigned long long)pd.q.stack);
        hipEvent_t e = h->prof.begin(h->stream);
        rc = hk::policy_launch_mlp(pd, pairs, pd.q.ring, w, decision, h->cfg.env_id_base, A, nullptr, nullptr, h->dev.act_steer,
                                   h->dev.act_branch, h->stream, h->err);
        if (rc) { g_last_error = h->err; return rc; }
        h->prof.end(3, e, h->stream);
    }
    return HK_OK;
}

int hk_step(hk_handle h, int n_ticks)
{
    HK_NEED_ENV_STEP(h);
    if (n_ticks < 0) return fail(h, HK_ERR_INVALID, "hk_step: n_ticks < 0");
    if (n_ticks == 0) return HK_OK;
    if (h->n_policies == 0) {
        h->academy_step += n_ticks;
        return step_ticks(h, n_ticks);
    }
    // with policies attached the Academy steps first in a decision tick; the ticks up to the next decision run fused
    int left = n_ticks;
    while (left > 0) {
        const int phase = (int)(h->academy_step % h->decision_period);
        if (phase == 0) { int rc = policy_decide(h); if (rc) return rc; }
        int chunk = h->decision_period - phase;
        if (chunk > left) chunk = left;
        int rc = step_ticks(h, chunk);
        if (rc) return rc;
        h->academy_step += chunk;
        left -= chunk;
    }
    return HK_OK;
}

int hk_policy_attach(hk_handle h, const hk_policy_desc* desc, const int32_t* agent_slots, int n_slots, int decision_period)
{
    HK_NEED_ENV(h);
    int rc = hk::policy_validate(desc, h->err);
    if (rc) { g_last_error = h->err; return rc; }
    if (!agent_slots || n_slots < 1 || n_slots > h->cfg.num_agents || decision_period < 1)
        return fail(h, HK_ERR_INVALID, "hk_policy_attach: bad agent_slots / decision_period");
    if (h->n_policies >= HK_MAX_POLICIES) return fail(h, HK_ERR_INVALID, "hk_policy_attach: HK_MAX_POLICIES policies already attached");
    if (desc->in_dim != hk_obs_dim(h) * desc->stack)
        return fail(h, HK_ERR_INVALID, "hk_policy_attach: in_dim != hk_obs_dim * stack (a model trained for another agent count / horizon)");
    for (int j = 0; j < n_slots; j++) {
        const int a = agent_slots[j];
        if (a < 0 || a >= h->cfg.num_agents || h->cfg.low_mode[a] != HK_LOW_RL)
            return fail(h, HK_ERR_INVALID, "hk_policy_attach: agent slot out of range or not LowMode RL");
        for (int p = 0; p < h->n_policies; p++)
            for (int q = 0; q < h->policy[p].q.n_slots; q++)
                if (h->policy[p].q.slots[q] == a) return fail(h, HK_ERR_INVALID, "hk_policy_attach: agent slot already has a policy");
        for (int q = 0; q < j; q++) if (agent_slots[q] == a) return fail(h, HK_ERR_INVALID, "hk_policy_attach: duplicate agent slot");
    }
    if (h->n_policies > 0 && decision_period != h->decision_period)
        return fail(h, HK_ERR_INVALID, "hk_policy_attach: decision_period differs from the policies already attached (DecisionRequester is per handle)");
    const int idx = h->n_policies;
    rc = hk::policy_upload(h->policy[idx], desc, idx, hk_obs_dim(h), agent_slots, n_slots, h->cfg.num_envs, h->stream, h->err);
    if (rc) { hk::policy_free(h->policy[idx]); g_last_error = h->err; return rc; }
    h->decision_period = decision_period;
    h->n_policies = idx + 1;
    return idx;
}

int hk_policy_forward(hk_handle h, int policy, int rows, const float* obs, float* mu, float* logits)
{
    HK_NEED_ENV(h);
    if (policy < 0 || policy >= h->n_policies || rows < 0) return fail(h, HK_ERR_INVALID, "hk_policy_forward: bad policy / rows");
    if (rows == 0) return HK_OK;
    if (!obs || !mu || !logits) return fail(h, HK_ERR_INVALID, "hk_policy_forward: NULL pointer");
    const hk::PolicyDevice& pd = h->policy[policy];
    const size_t n_in = (size_t)rows * pd.q.in_dim, n_lg = (size_t)rows * pd.q.n_branch;
    const size_t total = (n_in + rows + n_lg) * sizeof(float);
    if (total > h->pol_scratch_bytes) {
        if (h->pol_scratch) HK_HIP(h, hipFree(h->pol_scratch));
        h->pol_scratch = nullptr; h->pol_scratch_bytes = 0;
        HK_HIP(h, hipMalloc(&h->pol_scratch, total));
        h->pol_scratch_bytes = total;
    }
    float* d_in = (float*)h->pol_scratch;
    float* d_mu = d_in + n_in;
    float* d_lg = d_mu + rows;
    HK_HIP(h, hipMemcpyAsync(d_in, obs, n_in * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipEvent_t e = h->prof.begin(h->stream);
    int rc = hk::policy_launch_mlp(pd, rows, d_in, pd.q.stack - 1, 0ull, 0, h->cfg.num_agents, d_mu, d_lg, nullptr, nullptr, h->stream, h->err);
    if (rc) { g_last_error = h->err; return rc; }
    h->prof.end(3, e, h->stream);
    HK_HIP(h, hipMemcpyAsync(mu, d_mu, rows * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipMemcpyAsync(logits, d_lg, n_lg * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

int hk_get_actions(hk_handle h, float* steer, int32_t* branch)
{
    HK_NEED_ENV(h);
    if (!steer || !branch) return fail(h, HK_ERR_INVALID, "hk_get_actions: NULL pointer");
    const size_t cnt = (size_t)h->cfg.num_envs * h->cfg.num_agents;
    HK_HIP(h, hipMemcpyAsync(steer, h->dev.act_steer, cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipMemcpyAsync(branch, h->dev.act_branch, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

int hk_obs_dim(hk_handle h)
{
    if (!h || !h->env_ready) return HK_ERR_INVALID;
    // HKA:424  Sensors.Length + sectionHorizon*5 + 8 + 12*(others + team)
    return HK_NUM_SENSORS + h->cfg.section_horizon * 5 + 8 + 12 * (h->cfg.num_agents - 1);
}

int hk_get_observations(hk_handle h, float* obs)
{
    HK_NEED_ENV(h);
    if (!obs) return fail(h, HK_ERR_INVALID, "hk_get_observations: NULL pointer");
    int rc = hk::env_launch_observe(h->dev, h->cfg, 0xFFFFFFFFu, h->stream, h->err);
    if (rc) { g_last_error = h->err; return rc; }
    const size_t cnt = (size_t)h->cfg.num_envs * h->cfg.num_agents * hk_obs_dim(h);
    HK_HIP(h, hipMemcpyAsync(obs, h->dev.obs, cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

// after a sync: did every env finish the ticks of the last hk_step, did an LQ solve hit a zero pivot?
static int check_device_status(hk_handle h)
{
    int st[4] = {0, 0, 0, 0};
    HK_HIP(h, hipMemcpyAsync(st, h->dev.status, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    if (st[0] & 4) {
        // reported once: the flag is cleared here (and by hk_reset), the unfinished envs keep their leftover ticks for the next hk_step
        const int keep = ~4;
        hipLaunchKernelGGL(hk::status_and_kernel, dim3(1), dim3(1), 0, h->stream, h->dev.status, keep);
        HK_HIP(h, hipGetLastError());
        return fail(h, HK_ERR_HIP, "hk_step: an env did not complete its ticks (internal scheduling error)");
    }
    return HK_OK;
}

int hk_get_agent_state(hk_handle h, hk_agent_state* out)
{
    HK_NEED_ENV(h);
    if (!out) return fail(h, HK_ERR_INVALID, "NULL pointer");
    { int rc = check_device_status(h); if (rc) return rc; }
    const size_t cnt = (size_t)h->cfg.num_envs * h->cfg.num_agents;
    // the per-tick fields live in the hot tiles (hk_env_device.h): gather them into the records first
    { int rc = hk::ga_ops(h->dev).launch_hot_gather(h->dev, h->cfg, h->stream, h->err); if (rc) { g_last_error = h->err; return rc; } }
    HK_HIP(h, hipMemcpyAsync(out, h->dev.agents, cnt * sizeof(hk_agent_state), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

int hk_set_agent_state(hk_handle h, const hk_agent_state* in)
{
    HK_NEED_ENV(h);
    if (!in) return fail(h, HK_ERR_INVALID, "NULL pointer");
    const size_t cnt = (size_t)h->cfg.num_envs * h->cfg.num_agents;
    // the device divides by the section count with a multiply-shift that is exact for 0 <= x < 2^32 / L (hk_env_device.h div_L): a rewound or
    // hand-built record outside that range would index the track tables with garbage
    {
        const int64_t lim = (int64_t)(0x100000000ull / (uint64_t)std::max(1, h->cfg.num_sections)) - 2048;
        for (size_t t = 0; t < cnt; t++)
            if (in[t].section_index < 0 || in[t].init_checkpoint_index < 0 || in[t].section_index >= lim || in[t].init_checkpoint_index >= lim)
                return fail(h, HK_ERR_INVALID, "hk_set_agent_state: section_index / init_checkpoint_index out of range (negative, or beyond 2^32 / num_sections)");
    }
    h->dev.P.hold_dedupe = 0;          // the host moves karts by hand: a held kart is no longer guaranteed to be where its last solve saw it
    HK_HIP(h, hipMemcpyAsync(h->dev.agents, in, cnt * sizeof(hk_agent_state), hipMemcpyHostToDevice, h->stream));
    { int rc = hk::ga_ops(h->dev).launch_hot_scatter(h->dev, h->cfg, h->stream, h->err); if (rc) { g_last_error = h->err; return rc; } }
    { int rc = hk::env_mcts_invalidate(h->dev, h->cfg, h->stream, h->err); if (rc) { g_last_error = h->err; return rc; } }   // plans were rewritten
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

int hk_get_env_state(hk_handle h, hk_env_state* out)
{
    HK_NEED_ENV(h);
    if (!out) return fail(h, HK_ERR_INVALID, "NULL pointer");
    { int rc = check_device_status(h); if (rc) return rc; }
    { int rc = hk::ga_ops(h->dev).launch_envs_gather(h->dev, h->cfg, h->stream, h->err); if (rc) { g_last_error = h->err; return rc; } }
    HK_HIP(h, hipMemcpyAsync(out, h->dev.envs_stage, (size_t)h->cfg.num_envs * sizeof(hk_env_state), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

int hk_set_env_state(hk_handle h, const hk_env_state* in)
{
    HK_NEED_ENV(h);
    if (!in) return fail(h, HK_ERR_INVALID, "NULL pointer");
    h->dev.P.hold_dedupe = 0;          // (as hk_set_agent_state: episode_steps may be rewound into a hold whose solves were skipped)
    h->lock_tick = -1;                 // (the host wrote episode steps)
    HK_HIP(h, hipMemcpyAsync(h->dev.envs_stage, in, (size_t)h->cfg.num_envs * sizeof(hk_env_state), hipMemcpyHostToDevice, h->stream));
    { int rc = hk::ga_ops(h->dev).launch_envs_scatter(h->dev, h->cfg, h->stream, h->err); if (rc) { g_last_error = h->err; return rc; } }   // (progress words sanitized on the way)
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

int hk_get_episode_results(hk_handle h, hk_episode_result* out)
{
    HK_NEED_ENV(h);
    if (!out) return fail(h, HK_ERR_INVALID, "NULL pointer");
    const size_t cnt = (size_t)h->cfg.num_envs * h->cfg.num_agents;
    HK_HIP(h, hipMemcpyAsync(out, h->dev.results, cnt * sizeof(hk_episode_result), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

int hk_comm_unique_id(void* id_out)
{
    if (!id_out) return fail(nullptr, HK_ERR_INVALID, "hk_comm_unique_id: NULL pointer");
    std::string err;
    if (!g_rccl.load(err)) return fail(nullptr, HK_ERR_UNSUPPORTED, err);
    RcclId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return fail(nullptr, HK_ERR_HIP, "ncclGetUniqueId: " + g_rccl.why(rc));
    std::memcpy(id_out, id.internal, HK_COMM_ID_BYTES);
    return HK_OK;
}

int hk_comm_init(hk_handle h, int world_size, int rank, const void* id)
{
    HK_NEED_ENV(h);
    if (!id || world_size < 1 || rank < 0 || rank >= world_size) return fail(h, HK_ERR_INVALID, "hk_comm_init: bad world_size / rank / id");
    if (h->comm) return fail(h, HK_ERR_INVALID, "hk_comm_init: this handle already has a communicator");
    if (!g_rccl.load(h->err)) { g_last_error = h->err; return HK_ERR_UNSUPPORTED; }
    RcclId rid;
    std::memcpy(rid.internal, id, HK_COMM_ID_BYTES);
    const int rc = g_rccl.CommInitRank(&h->comm, world_size, rid, rank);
    if (rc != 0) { h->comm = nullptr; return fail(h, HK_ERR_HIP, "ncclCommInitRank: " + g_rccl.why(rc)); }
    h->comm_world = world_size; h->comm_rank = rank;
    return HK_OK;
}

int hk_gather_results(hk_handle h, hk_episode_result* all)
{
    HK_NEED_ENV(h);
    if (!all) return fail(h, HK_ERR_INVALID, "hk_gather_results: NULL pointer");
    if (!h->comm) return fail(h, HK_ERR_INVALID, "hk_gather_results: call hk_comm_init first");
    { int rc = check_device_status(h); if (rc) return rc; }
    const int W = h->comm_world;
    const size_t local = (size_t)h->cfg.num_envs * h->cfg.num_agents * sizeof(hk_episode_result);
    // Ranks may hold different env counts (a contiguous split of a total that the world size does not divide): exchange the
    // byte counts first (8 bytes per rank), pad every contribution to the largest, gather, and trim on the way to the host.
    const size_t cnt_bytes = (size_t)W * sizeof(unsigned long long);
    if (cnt_bytes + sizeof(unsigned long long) > h->gather_cnt_bytes) {
        if (h->gather_cnt) HK_HIP(h, hipFree(h->gather_cnt));
        h->gather_cnt = nullptr; h->gather_cnt_bytes = 0;
        HK_HIP(h, hipMalloc(&h->gather_cnt, cnt_bytes + sizeof(unsigned long long)));
        h->gather_cnt_bytes = cnt_bytes + sizeof(unsigned long long);
    }
    unsigned long long* d_cnt = (unsigned long long*)h->gather_cnt;          // [W] gathered, [W] = this rank's word
    const unsigned long long mine = (unsigned long long)local;
    HK_HIP(h, hipMemcpyAsync(d_cnt + W, &mine, sizeof(mine), hipMemcpyHostToDevice, h->stream));
    int rc = g_rccl.AllGather(d_cnt + W, d_cnt, sizeof(unsigned long long), /*ncclChar*/ 0, h->comm, h->stream);
    if (rc != 0) return fail(h, HK_ERR_HIP, "ncclAllGather (sizes): " + g_rccl.why(rc));
    std::vector<unsigned long long> cnt((size_t)W);
    HK_HIP(h, hipMemcpyAsync(cnt.data(), d_cnt, cnt_bytes, hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    size_t per = 0;
    for (int r = 0; r < W; r++) {
        if (cnt[r] % ((size_t)h->cfg.num_agents * sizeof(hk_episode_result)) != 0)
            return fail(h, HK_ERR_INVALID, "hk_gather_results: a rank holds a different num_agents");
        per = cnt[r] > per ? (size_t)cnt[r] : per;
    }
    const size_t total = per * (size_t)W + per;                              // [W] padded slots + this rank's padded send slot
    if (total > h->gather_bytes) {
        if (h->gather_buf) HK_HIP(h, hipFree(h->gather_buf));
        h->gather_buf = nullptr; h->gather_bytes = 0;
        HK_HIP(h, hipMalloc(&h->gather_buf, total));
        h->gather_bytes = total;
    }
    char* buf = (char*)h->gather_buf;
    const void* send = h->dev.results;
    if (local < per) {                                                       // pad this rank's contribution
        HK_HIP(h, hipMemsetAsync(buf + per * W, 0, per, h->stream));
        HK_HIP(h, hipMemcpyAsync(buf + per * W, h->dev.results, local, hipMemcpyDeviceToDevice, h->stream));
        send = buf + per * W;
    }
    // bytes on the wire (ncclChar): the records are plain data, identical layout on every rank
    rc = g_rccl.AllGather(send, buf, per, /*ncclChar*/ 0, h->comm, h->stream);
    if (rc != 0) return fail(h, HK_ERR_HIP, "ncclAllGather: " + g_rccl.why(rc));
    char* out = (char*)all;
    for (int r = 0; r < W; r++) {                                            // rank r's rows, trimmed to what it holds
        if (cnt[r]) HK_HIP(h, hipMemcpyAsync(out, buf + per * r, (size_t)cnt[r], hipMemcpyDeviceToHost, h->stream));
        out += cnt[r];
    }
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

int hk_gather_count(hk_handle h, int64_t* total_envs)
{
    HK_NEED_ENV(h);
    if (!total_envs) return fail(h, HK_ERR_INVALID, "hk_gather_count: NULL pointer");
    if (!h->comm) return fail(h, HK_ERR_INVALID, "hk_gather_count: call hk_comm_init first");
    const int W = h->comm_world;
    const size_t cnt_bytes = (size_t)W * sizeof(unsigned long long);
    if (cnt_bytes + sizeof(unsigned long long) > h->gather_cnt_bytes) {
        if (h->gather_cnt) HK_HIP(h, hipFree(h->gather_cnt));
        h->gather_cnt = nullptr; h->gather_cnt_bytes = 0;
        HK_HIP(h, hipMalloc(&h->gather_cnt, cnt_bytes + sizeof(unsigned long long)));
        h->gather_cnt_bytes = cnt_bytes + sizeof(unsigned long long);
    }
    unsigned long long* d_cnt = (unsigned long long*)h->gather_cnt;
    const unsigned long long mine = (unsigned long long)h->cfg.num_envs;
    HK_HIP(h, hipMemcpyAsync(d_cnt + W, &mine, sizeof(mine), hipMemcpyHostToDevice, h->stream));
    const int rc = g_rccl.AllGather(d_cnt + W, d_cnt, sizeof(unsigned long long), /*ncclChar*/ 0, h->comm, h->stream);
    if (rc != 0) return fail(h, HK_ERR_HIP, "ncclAllGather (sizes): " + g_rccl.why(rc));
    std::vector<unsigned long long> cnt((size_t)W);
    HK_HIP(h, hipMemcpyAsync(cnt.data(), d_cnt, cnt_bytes, hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    int64_t t = 0;
    for (int r = 0; r < W; r++) t += (int64_t)cnt[r];
    *total_envs = t;
    return HK_OK;
}

int hk_comm_destroy(hk_handle h)
{
    if (!h) return fail(nullptr, HK_ERR_INVALID, "NULL handle");
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    h->comm = nullptr; h->comm_world = 0;
    return HK_OK;
}

int hk_get_rewards(hk_handle h, float* reward, float* group_reward)
{
    HK_NEED_ENV(h);
    if (!reward || !group_reward) return fail(h, HK_ERR_INVALID, "NULL pointer");
    { int rc = check_device_status(h); if (rc) return rc; }
    const size_t cnt = (size_t)h->cfg.num_envs * h->cfg.num_agents;
    float* d_r = h->dev.reward_out;
    { int rc = hk::ga_ops(h->dev).launch_rewards_read(h->dev, (int)cnt, d_r, d_r + cnt, h->stream, h->err); if (rc) { g_last_error = h->err; return rc; } }
    HK_HIP(h, hipMemcpyAsync(reward, d_r, cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipMemcpyAsync(group_reward, d_r + cnt, cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

int hk_get_mcts_state(hk_handle h, hk_mcts_state* out)
{
    HK_NEED_ENV(h);
    if (!out) return fail(h, HK_ERR_INVALID, "NULL pointer");
    { int rc = check_device_status(h); if (rc) return rc; }
    const size_t cnt = (size_t)h->cfg.num_envs * h->cfg.num_agents;
    if (!h->dev.mcts.st) { std::memset(out, 0, cnt * sizeof(hk_mcts_state)); return HK_OK; }
    if (h->dev.mcts_ticks > 0) {          // searches deferred by short hk_step calls: `pend` must be there when the host looks
        int rc = hk::env_flush_mcts(h->dev, h->stream, h->err);
        if (rc) { g_last_error = h->err; return rc; }
    }
    HK_HIP(h, hipMemcpyAsync(out, h->dev.mcts.st, cnt * sizeof(hk_mcts_state), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

int hk_get_lq_debug(hk_handle h, int env, int ego, hk_lq_debug* out)
{
    HK_NEED_ENV(h);
    if (!out || env < 0 || env >= h->cfg.num_envs || ego < 0 || ego >= h->cfg.num_agents)
        return fail(h, HK_ERR_INVALID, "hk_get_lq_debug: bad arguments");
    if (!h->dev.lq_debug) return fail(h, HK_ERR_INVALID, "hk_get_lq_debug: debug taps are off");
    HK_HIP(h, hipMemcpyAsync(out, h->dev.lq_debug + ((size_t)env * h->cfg.num_agents + ego), sizeof(hk_lq_debug),
                             hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    return HK_OK;
}

// The device-pointer getters hand out the library's own buffers for zero-copy consumers (torch, the C# host's compute buffers).  Each one first SETTLES
// the handle like every other getter — laggards of a lazily completed call, the completion guard of optimistic short calls, a search launch on the
// planner's side stream — so that what the pointer shows, once the handle's stream has been synchronised, is the state hk_synchronize would leave.
// The work is issued on the handle's stream; a pointer taken BEFORE a later hk_step shows that step's results only after the next getter / hk_synchronize.
static bool settle_for_pointer(hk_handle h)
{
    if (!h || !h->env_ready) return false;
    if (hipSetDevice(h->device) != hipSuccess) return false;
    if (split_join(h)) return false;
    if (h->step_pending && finish_ticks(h) != HK_OK) return false;
    if (h->opt_pending && verify_optimistic(h) != HK_OK) return false;
    if (mcts_join_async(h)) return false;
    return true;
}
void* hk_device_results_ptr(hk_handle h) { return settle_for_pointer(h) ? (void*)h->dev.results : nullptr; }
void* hk_device_agents_ptr(hk_handle h)
{
    // a SNAPSHOT: the per-tick fields are gathered from the hot tiles into the records (asynchronously, on the handle's stream) by this call;
    // after further hk_step calls the pointer has to be requested again
    if (!settle_for_pointer(h)) return nullptr;
    if (hk::ga_ops(h->dev).launch_hot_gather(h->dev, h->cfg, h->stream, h->err) != HK_OK) { g_last_error = h->err; return nullptr; }
    return (void*)h->dev.agents;
}
void* hk_device_obs_ptr(hk_handle h) { return settle_for_pointer(h) ? (void*)h->dev.obs : nullptr; }
void* hk_device_act_steer_ptr(hk_handle h) { return (h && h->env_ready) ? (void*)h->dev.act_steer : nullptr; }
void* hk_device_act_branch_ptr(hk_handle h) { return (h && h->env_ready) ? (void*)h->dev.act_branch : nullptr; }
void* hk_device_reward_ptr(hk_handle h) { return (h && h->env_ready) ? (void*)h->dev.reward_out : nullptr; }
void* hk_device_group_reward_ptr(hk_handle h)
{
    return (h && h->env_ready && h->dev.reward_out) ? (void*)(h->dev.reward_out + (size_t)h->cfg.num_envs * h->cfg.num_agents) : nullptr;
}

int hk_observe(hk_handle h)
{
    HK_NEED_ENV(h);
    int rc = hk::env_launch_observe(h->dev, h->cfg, 0xFFFFFFFFu, h->stream, h->err);
    if (rc) g_last_error = h->err;
    return rc;
}

int hk_rewards_device(hk_handle h)
{
    HK_NEED_ENV(h);
    const size_t cnt = (size_t)h->cfg.num_envs * h->cfg.num_agents;
    int rc = hk::ga_ops(h->dev).launch_rewards_read(h->dev, (int)cnt, h->dev.reward_out, h->dev.reward_out + cnt, h->stream, h->err);
    if (rc) g_last_error = h->err;
    return rc;
}

int hk_prof_enable(hk_handle h, int on)
{
    if (!h) return HK_ERR_INVALID;
    HK_HIP(h, hipSetDevice(h->device));
    h->prof.on = on != 0;
    return HK_OK;
}

int hk_prof_reset(hk_handle h)
{
    if (!h) return HK_ERR_INVALID;
    HK_HIP(h, hipSetDevice(h->device));
    if (split_join(h)) return fail(h, HK_ERR_HIP, "hipStreamWaitEvent (the parts of a split call)");
    if (h->step_pending) { int rc = finish_ticks(h); if (rc) return rc; }
    HK_HIP(h, hipStreamSynchronize(h->stream));
    h->prof.fold();
    for (int s = 0; s < HK_PROF_STAGES; s++) { h->prof.ms[s] = 0; h->prof.n[s] = 0; }
    if (h->env_ready && h->dev.game_stats) {
        HK_HIP(h, hipMemsetAsync(h->dev.game_stats, 0, hk::GAME_METER * sizeof(unsigned long long), h->stream));      // (not the games-per-launch meter behind the statistics: the schedule lives on it)
        HK_HIP(h, hipStreamSynchronize(h->stream));
    }
    return HK_OK;
}

int hk_prof_games(hk_handle h, int64_t* games)
{
    HK_NEED_ENV(h);
    if (!games) return fail(h, HK_ERR_INVALID, "hk_prof_games: NULL pointer");
    unsigned long long g[hk::GAME_STATS_N];
    HK_HIP(h, hipMemcpyAsync(g, h->dev.game_stats, sizeof(g), hipMemcpyDeviceToHost, h->stream));
    HK_HIP(h, hipStreamSynchronize(h->stream));
    for (int n = 0; n <= HK_MAX_AGENTS; n++) games[n] = (int64_t)g[n];
#ifdef HK_STAMPS
    {       // diagnostic builds (-DHK_STAMPS): the phase cycle counters of the tick kernel
        std::fprintf(stderr, "HK_STAMPS");
        for (int k = 16; k < hk::GAME_STATS_N; k++) std::fprintf(stderr, " %llu", g[k]);
        std::fprintf(stderr, "\n");
    }
#endif
    return HK_OK;
}

int hk_prof_read(hk_handle h, double* ms, int64_t* launches)
{
    if (!h) return HK_ERR_INVALID;
    HK_HIP(h, hipSetDevice(h->device));
    if (split_join(h)) return fail(h, HK_ERR_HIP, "hipStreamWaitEvent (the parts of a split call)");
    if (h->step_pending) { int rc = finish_ticks(h); if (rc) return rc; }
    HK_HIP(h, hipStreamSynchronize(h->stream));
    h->prof.fold();
    for (int s = 0; s < HK_PROF_STAGES; s++) {
        if (ms) ms[s] = h->prof.ms[s];
        if (launches) launches[s] = h->prof.n[s];
    }
    return HK_OK;
}

}  // extern "C"
