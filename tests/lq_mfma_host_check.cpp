// Test infrastructure (tests/test_lq_mfma_host.py): hk_lq_mfma.h compiled for the HOST.  64 threads stand for the lanes of the wave
// that owns one game; the three wave primitives become barrier-guarded exchanges — the f64 MFMA as the k-ascending fma chain the
// hardware was measured to compute (tools/experiments/mfma_f64_check.hip), v_readlane as a shared slot, the wave-level LDS ordering as a
// barrier.  The arithmetic is the header's own, so the controls must equal the C oracle's bit for bit.
// stdin: NP n_games, then per game A[NP][16] B[NP][8] Q[NP][n][n] q[NP][n] R[NP][4] x0[n]; stdout: u0 of each game as hex floats.
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
#define HK_LQM_HOST_CHECK 1
struct double2 { double x, y; };
namespace hk {
static inline double fma64(double a, double b, double c) { return std::fma(a, b, c); }
}
// hk_lq_core.h is device code; the two things the MFMA solver takes from it:
#define HK_LQ_CORE_FOR_HOST 1
static std::barrier<> bar(64);
static double xa[64], xb[64];
static int xi[64];
static thread_local int tl_lane;
struct lqm_d4_h { double v[4]; double& operator[](int i) { return v[i]; } const double& operator[](int i) const { return v[i]; } };
struct LqmHost {
    template <class D4> static D4 mfma(double a, double b, D4 c)
    {
        xa[tl_lane] = a; xb[tl_lane] = b;
        bar.arrive_and_wait();
        const int g = tl_lane >> 4, col = tl_lane & 15;
        D4 d;
        for (int j = 0; j < 4; j++) {
            const int row = g + 4 * j;
            double s = c[j];
            for (int k = 0; k < 4; k++) s = std::fma(xa[row + 16 * k], xb[col + 16 * k], s);      // A[row][k] = lane (k, row), B[k][col] = lane (k, col)
            d[j] = s;
        }
        bar.arrive_and_wait();
        return d;
    }
    static double readlane(double v, int k) { xa[tl_lane] = v; bar.arrive_and_wait(); const double r = xa[k]; bar.arrive_and_wait(); return r; }
    static int readlane_i(int v, int k) { xi[tl_lane] = v; bar.arrive_and_wait(); const int r = xi[k]; bar.arrive_and_wait(); return r; }
    static void sync() { bar.arrive_and_wait(); }
};
#include "hk_lq_mfma.h"

struct QDenseH {
    const double* Qg; const double* qg; int n;
    double Q(int i, int r, int c) const { return Qg[((size_t)i * n + r) * n + c]; }
    double q(int i, int r) const { return qg[(size_t)i * n + r]; }
};

template <int NP> static int run(int ng)
{
    constexpr int n = 4 * NP;
    static hk::LqMfmaLds<NP> L;
    for (int gme = 0; gme < ng; gme++) {
        std::vector<double> A(NP * 16), B(NP * 8), Q((size_t)NP * n * n), q(NP * n), R(NP * 4), x0(n);
        auto rd = [](std::vector<double>& v) { for (auto& x : v) if (scanf("%lf", &x) != 1) return false; return true; };
        if (!rd(A) || !rd(B) || !rd(Q) || !rd(q) || !rd(R) || !rd(x0)) return 2;
        std::memset(&L, 0, sizeof(L));
        for (int i = 0; i < NP; i++) {
            for (int e = 0; e < 16; e++) L.Ab[i][e] = A[i * 16 + e];
            for (int e = 0; e < 8; e++) L.Bb[i][e] = B[i * 8 + e];
            for (int e = 0; e < 4; e++) L.Rb[i][e] = R[i * 4 + e];
        }
        for (int r = 0; r < n; r++) L.x0[r] = x0[r];
        QDenseH qp{Q.data(), q.data(), n};
        double u[64][2]; int sing[64];
        std::vector<std::thread> th;
        for (int l = 0; l < 64; l++) th.emplace_back([&, l] { tl_lane = l; hk::lq_solve_game_mfma<NP, QDenseH, LqmHost>(l, L, qp, 3, u[l], sing[l]); });
        for (auto& t : th) t.join();
        for (int l = 1; l < 64; l++) if (std::memcmp(u[l], u[0], 16) != 0) { fprintf(stderr, "lanes disagree on u0 (game %d lane %d)\n", gme, l); return 3; }
        printf("%a %a %d\n", u[0][0], u[0][1], sing[0]);
    }
    return 0;
}

int main()
{
    int NP = 0, ng = 0;
    if (scanf("%d %d", &NP, &ng) != 2) return 1;
    return NP == 3 ? run<3>(ng) : NP == 4 ? run<4>(ng) : 1;
}
