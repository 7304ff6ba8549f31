// Test infrastructure: the tick kernel (csrc/hk_env_run.h: env_run_kernel, with everything it inlines — episode controller, rays, game
// assembly, single-player Riccati solve, kart model, triggers, rewards) and env_reset_kernel compiled for the HOST through the stand-in
// <hip/hip_runtime.h> in tests/host_emu, so that the device code runs under -fsanitize=address,undefined.  One lane group (a quad) is
// four host threads; a cross-lane primitive reached by only part of the group ends the run (see the stand-in header).
//
// The only thing the tick kernel does not do itself is a multi-player LQ game (it queues it for lqn_round_kernel and parks the env).
// This driver plays that kernel's part with recorded controls: the caller (tests/test_env_run_host.py) hands in, per tick, the
// (flags, steering) every agent had after that tick in the C oracle; a queued ego gets its ACCEL / BRAKE bits and steering from there —
// exactly the two fields the solver kernel writes — and the env resumes.  Everything else in the state dumps comes from the device code.
//
// in : <file>  int32 header[10] = {magic, sizeof(hk_config), L, NW, n_ticks, chunk, eager, E, A, run_cap}, hk_config, sections, walls,
//              ctl[n_ticks][E][A] = {uint32 flags, float steering}
// out: <file>  per chunk: agents[E][A] (hk_agent_state), envs[E] (hk_env_state)
#include <hip/hip_runtime.h>
#include <string>
#include <thread>
#include <vector>

thread_local hk_emu_dim3 threadIdx;
hk_emu_dim3 blockIdx, blockDim, gridDim;
namespace hk_emu { Barrier bar; uint64_t slot[LANES]; unsigned char* dyn_shared; }

#ifndef HK_GA
#define HK_GA 4
#define HK_GA_NS g4
#endif
#include "hk_env_ga.h"
#include "hk_env_params.h"

using namespace hk;
using namespace hk::HK_GA_NS;

struct Ctl { uint32_t flags; float steering; };

template <class T> static std::vector<T> read_n(FILE* f, size_t n)
{
    std::vector<T> v(n);
    if (n && std::fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "short input\n"); std::exit(2); }
    return v;
}

struct World {
    EnvParams P;
    std::vector<hk_agent_state> agents;   // cold fields + the staging copy of the hot ones (gathered from the tiles before every dump)
    std::vector<uint32_t> hot;            // the hot tiles (hk_env_device.h), identity slot order: slot == env
    std::vector<int> ident;               // perm == slot_of == identity
    std::vector<hk_env_state> envs;
    std::vector<hk_episode_result> results;
    std::vector<double> games;
    std::vector<int> queue_cnt, queue, sec_time, sec_cnt;
    std::vector<unsigned char> hit_code;
    std::vector<unsigned long long> stats;
    RwDev RD{};
    int status[4] = {0, 0, 0, 0};
};

template <bool HAS_MCTS, bool HAS_RW, bool HAS_TRAIN>
static void launch_quad(World& W, int env, int round, int arm)
{
    GameSoA G{W.games.data(), W.agents.size()};
    std::thread th[GA];
    for (unsigned l = 0; l < (unsigned)GA; l++)
        th[l] = std::thread([&, l] {
            threadIdx = {l, 0, 0};
            env_run_kernel<HAS_MCTS, HAS_RW, HAS_TRAIN, false>(W.P, W.agents.data(), W.hot.data(), W.envs.data(), W.results.data(), G, W.queue_cnt.data(), W.queue.data(), round,
                                                        nullptr, nullptr, nullptr, W.status, MctsDev{}, 0, W.RD, W.ident.data(), W.stats.data(), env, env + 1, 0, arm, 0);
        });
    for (auto& t : th) t.join();
}

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: env_run_host_check <in> <out>\n"); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    const auto hd = read_n<int32_t>(f, 10);
    if (hd[0] != 0x484b4531 || hd[1] != (int)sizeof(hk_config)) { std::fprintf(stderr, "bad header (sizeof(hk_config) = %zu)\n", sizeof(hk_config)); return 2; }
    const int L = hd[2], NW = hd[3], n_ticks = hd[4], chunk = hd[5], eager = hd[6], E = hd[7], A = hd[8], run_cap = hd[9];
    hk_config cfg = read_n<hk_config>(f, 1)[0];
    auto sec_in = read_n<hk_section>(f, L);
    auto wall_in = read_n<hk_wall_seg>(f, NW);
    const auto ctl = read_n<Ctl>(f, (size_t)n_ticks * E * A);
    std::fclose(f);
    if (cfg.num_envs != E || cfg.num_agents != A || cfg.num_sections != L || cfg.num_walls != NW || A > GA) { std::fprintf(stderr, "header / config mismatch\n"); return 2; }
    cfg.sections = sec_in.data(); cfg.walls = wall_in.data();

    World W;
    std::vector<hk_section> sections; std::vector<hk_wall_seg> walls; std::vector<unsigned char> pk; std::vector<int> perms; std::string err;
    if (int rc = env_build_params(cfg, sections, walls, W.P, pk, perms, err)) { std::fprintf(stderr, "env_build_params: %d %s\n", rc, err.c_str()); return 2; }
    if (W.P.any_mcts) { std::fprintf(stderr, "planner handles are not emulated\n"); return 2; }
    W.P.tab = pk.data(); W.P.perms = perms.data();
    W.P.eager = eager; W.P.run_cap = run_cap;
    const size_t na = (size_t)E * A;
    W.agents.assign(na, hk_agent_state{}); W.envs.assign(E, hk_env_state{}); W.results.assign(na, hk_episode_result{});
    W.hot.assign(hot_words(E, GA), 0u); W.ident.resize(E);
    for (int e = 0; e < E; e++) W.ident[e] = e;
    W.games.assign(na * GA * GP_FIELDS, 0.0);
    W.queue_cnt.assign(4 * 16, 0); W.queue.assign(4 * (GA - 1) * na, 0); W.stats.assign(64, 0ull);
    if (cfg.rewards) {
        W.RD.S = cfg.laps * L + 2;
        W.sec_time.assign(na * W.RD.S, -1); W.sec_cnt.assign(na * W.RD.S, 0); W.hit_code.assign(na * HK_NUM_SENSORS, 0);
        W.RD.sec_time = W.sec_time.data(); W.RD.sec_cnt = W.sec_cnt.data(); W.RD.hit_code = W.hit_code.data();
    }
    std::vector<unsigned char> lds(64 * 1024, 0);
    hk_emu::dyn_shared = lds.data();
    blockIdx = {0, 0, 0}; blockDim = {(unsigned)GA, 1, 1}; gridDim = {1, 1, 1};

    // hk_reset of every env: env_reset_kernel is lane-local, one host thread walks the lanes
    blockDim = {(unsigned)(E * GA), 1, 1};
    for (unsigned t = 0; t < (unsigned)(E * GA); t++) {
        threadIdx = {t, 0, 0};
        env_reset_kernel(W.P, W.agents.data(), W.hot.data(), W.ident.data(), W.envs.data(), nullptr, E, -1, MctsDev{}, 0, W.RD, W.status);
    }
    blockDim = {(unsigned)GA, 1, 1};

    FILE* out = std::fopen(argv[2], "wb");
    if (!out) { std::perror(argv[2]); return 2; }
    std::vector<int> done(E, 0);          // ticks each env has finished
    int round = 0;
    long launches = 0, parked_games = 0;
    for (int t0 = 0; t0 < n_ticks; t0 += chunk) {
        const int n = std::min(chunk, n_ticks - t0);
        bool first = true, busy = true;
        int guard_rounds = 0;
        while (busy) {
            if (++guard_rounds > 4 * n + 8) { std::fprintf(stderr, "chunk at tick %d did not finish in %d launches\n", t0, guard_rounds); return 4; }
            const int set = round & 1;
            for (int k = 0; k < 16; k++) W.queue_cnt[set * 16 + k] = 0;
            for (int env = 0; env < E; env++) {
                // the instantiation hk_step picks (hk_env_launch.h: launch_run): Training handles run <true, true, true>
                if (W.P.training_reset) launch_quad<true, true, true>(W, env, round, first ? n : 0);
                else if (cfg.rewards) launch_quad<false, true, false>(W, env, round, first ? n : 0);
                else launch_quad<false, false, false>(W, env, round, first ? n : 0);
                launches++;
            }
            first = false;
            // the solver kernel's part: controls of the queued egos, from the recorded oracle run
            for (int np = 2; np <= GA; np++) {
                const int cnt = W.queue_cnt[set * 16 + np];
                if (cnt < 0 || cnt > (int)na) { std::fprintf(stderr, "queue count %d out of range\n", cnt); return 4; }
                for (int q = 0; q < cnt; q++) {
                    const int game = W.queue[(size_t)set * (GA - 1) * na + (size_t)(np - 2) * na + q];
                    if (game < 0 || game >= (int)na) { std::fprintf(stderr, "queued game %d out of range\n", game); return 4; }
                    const int env = game / A;
                    const hk_env_state& es = W.envs[env];
                    if ((es.reserved[1] & ENV_PHASE_MASK) != 1) { std::fprintf(stderr, "env %d queued a game without parking\n", env); return 4; }
                    const int tick = t0 + n - es.reserved[0];         // the tick in progress (0-based): ticks finished so far
                    if (tick < 0 || tick >= n_ticks) { std::fprintf(stderr, "env %d: tick %d out of range\n", env, tick); return 4; }
                    const Ctl& c = ctl[((size_t)tick * E) * A + game];
                    uint32_t* me = W.hot.data() + hot_base<GA>(env, game - env * A);      // what decode_store writes (hk_env_solve.h)
                    hot_put<uint32_t>(me, HF_flags, (hot_get<uint32_t>(me, HF_flags) & ~(uint32_t)(HK_F_ACCEL | HK_F_BRAKE)) | (c.flags & (uint32_t)(HK_F_ACCEL | HK_F_BRAKE)));
                    hot_put<float>(me, HF_steering, c.steering);
                    parked_games++;
                }
            }
            round++;
            busy = false;
            for (int env = 0; env < E; env++) busy = busy || W.envs[env].reserved[0] != 0 || (W.envs[env].reserved[1] & ENV_PHASE_MASK) != 0;
        }
        for (size_t t = 0; t < na; t++) store_hot(&W.agents[t], load_hot_tile(W.hot.data() + hot_base<GA>((int)(t / A), (int)(t % A))));   // hot_gather_kernel's part
        std::fwrite(W.agents.data(), sizeof(hk_agent_state), na, out);
        std::fwrite(W.envs.data(), sizeof(hk_env_state), E, out);
    }
    std::fclose(out);
    std::printf("launches %ld queued_games %ld status %d\n", launches, parked_games, W.status[0]);
    return 0;
}
