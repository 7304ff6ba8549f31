// Test infrastructure (tests/test_lq_spread_host.py): hk_lq_spread.h compiled for the HOST.  One thread per lane of a game's lane set (four
// 2-player games side by side, as in a wave); the wave-level LDS ordering X::sync becomes a barrier, LDS is plain memory.  The arithmetic is the
// header's own, so the controls must equal the C oracle's bit for bit.
// stdin: NP n_games sw, then the GameSoA doubles [NP * GP_FIELDS][n_games]; stdout: u0 of each game as hex floats + the singular flag.
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <thread>
#include <barrier>
#include <vector>
#include <memory>
#define __device__
#define __forceinline__ inline
#define __global__
#define HK_LQS_HOST_CHECK 1
struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline double fma64(double a, double b, double c) { return std::fma(a, b, c); }
using std::fabs;
constexpr int GP_NO = 3;
constexpr int GP_X0 = 0, GP_A4 = 4, GP_TW = 8, GP_TGT = 12, GP_RC = 16, GP_AW = 17, GP_OPW = GP_AW + GP_NO, GP_OPT = GP_OPW + 3 * GP_NO,
              GP_M = GP_OPT + 3 * GP_NO, GP_FIELDS = (GP_M + 2) & ~1;
struct GameSoA { double* d; size_t ng; double get(int game, int i, int f) const { return d[((size_t)(i * GP_FIELDS + f)) * ng + game]; } };
static std::unique_ptr<std::barrier<>> bar;
struct HostSync { static void sync() { bar->arrive_and_wait(); } };
#include "hk_lq_spread.h"

template <int NP, bool SW> static int run(int ng, GameSoA G)
{
    constexpr int LANES = LqSpreadDims<NP>::G, GPW = LqSpreadDims<NP>::GPW;
    static LqSpreadLds<NP, SW> L[GPW];
    const int threads = LANES * GPW;
    bar = std::make_unique<std::barrier<>>(threads);
    for (int base = 0; base < ng; base += GPW) {
        std::memset(L, 0xFF, sizeof(L));            // (stale LDS contents must not matter)
        std::vector<double> u((size_t)threads * 2);
        std::vector<int> sing(threads);
        std::vector<std::thread> th;
        for (int l = 0; l < threads; l++)
            th.emplace_back([&, l] {
                const int slot = l / LANES, ln = l % LANES;
                const int game = base + slot < ng ? base + slot : ng - 1;          // idle slots recompute the last game
                lq_spread_solve<NP, HostSync, SW>(ln, game, (double)0.02f, G, L[slot], &u[(size_t)l * 2], sing[l]);
            });
        for (auto& t : th) t.join();
        for (int s = 0; s < GPW && base + s < ng; s++) {
            for (int l = 1; l < LANES; l++)
                if (std::memcmp(&u[(size_t)(s * LANES + l) * 2], &u[(size_t)(s * LANES) * 2], 16) != 0) { fprintf(stderr, "lanes disagree on u0 (game %d lane %d)\n", base + s, l); return 3; }
            printf("%a %a %d\n", u[(size_t)(s * LANES) * 2], u[(size_t)(s * LANES) * 2 + 1], sing[s * LANES]);
        }
    }
    return 0;
}

int main()
{
    int NP = 0, ng = 0, sw = 0;          // sw: the players take ONE W block in turn (LqSpreadLds<NP, true>: the in-wave form of env_b1_kernel's 4-player games)
    if (scanf("%d %d %d", &NP, &ng, &sw) != 3 || NP < 2 || NP > 4 || ng < 1) return 1;
    std::vector<double> d((size_t)NP * GP_FIELDS * ng);
    for (auto& x : d) { if (scanf("%lf", &x) != 1) return 2; }
    GameSoA G{d.data(), (size_t)ng};
    if (sw) return NP == 2 ? run<2, true>(ng, G) : (NP == 3 ? run<3, true>(ng, G) : run<4, true>(ng, G));
    return NP == 2 ? run<2, false>(ng, G) : (NP == 3 ? run<3, false>(ng, G) : run<4, false>(ng, G));
}
