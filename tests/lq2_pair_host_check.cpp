// Test infrastructure (tests/test_lq2_pair_host.py): hk_lq2_pair.h compiled for the HOST.  Two threads stand for the two lanes of a
// pair; the DPP exchange pair_get<> becomes a barrier-guarded swap; LDS is plain memory.  The arithmetic is the header's own, so the
// controls must equal the C oracle's bit for bit.  stdin: n_games, then the GameSoA doubles; stdout: u0 of each game as hex floats.
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <thread>
#include <barrier>
#include <vector>
#define __device__
#define __forceinline__ inline
#define __global__
#define HK_LQ2_HOST_CHECK 1
static inline double fma64(double a, double b, double c) { return std::fma(a, b, c); }
constexpr int GP_NO = 3;
constexpr int GP_X0 = 0, GP_A4 = 4, GP_TW = 8, GP_TGT = 12, GP_RC = 16, GP_AW = 17, GP_OPW = GP_AW + GP_NO, GP_OPT = GP_OPW + 3 * GP_NO,
              GP_M = GP_OPT + 3 * GP_NO, GP_FIELDS = (GP_M + 2) & ~1;
struct GameSoA { double* d; size_t ng; double get(int game, int i, int f) const { return d[((size_t)(i * GP_FIELDS + f)) * ng + game]; } };
static std::barrier<> bar(2);
static double xch[2];
static thread_local int tl_lane;
template <int SEL> static double pair_get_host(double v)
{
    xch[tl_lane] = v; bar.arrive_and_wait();
    const double r = SEL == 2 ? xch[1 - tl_lane] : xch[SEL];
    bar.arrive_and_wait();
    return r;
}
#define pair_get pair_get_host
#define HK_LQ2_NO_DPP 1
#include "hk_lq2_pair.h"
int main(int argc, char** argv)
{
    int ng = 0;
    if (scanf("%d", &ng) != 1) return 1;
    std::vector<double> d((size_t)2 * GP_FIELDS * ng);
    for (auto& x : d) { if (scanf("%lf", &x) != 1) return 2; }
    static Lq2PairLds S;
    GameSoA G{d.data(), (size_t)ng};
    for (int g = 0; g < ng; g++) {
        double u[2][2]; int sing[2];
        std::thread t0([&] { tl_lane = 0; lq2_pair_solve(g, 0, 0, 0.02f, G, S, u[0], sing[0]); });
        std::thread t1([&] { tl_lane = 1; lq2_pair_solve(g, 1, 1, 0.02f, G, S, u[1], sing[1]); });
        t0.join(); t1.join();
        printf("%a %a\n", u[0][0], u[0][1]);
    }
}
