// host check of csrc/hk_regroup_pos.h: for every (N, H) tried the map (rank, hinted) -> slot is a bijection onto [0, N), monotone within each kind,
// and the hinted slots are evenly spaced.  Prints "ok <cases>".
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hk_regroup_pos.h"
int main()
{
    long long cases = 0;
    std::vector<long long> Ns = {1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 64, 100, 127, 128, 1000, 4096, 8191, 65536};
    for (long long N : Ns) {
        std::vector<long long> Hs;
        for (long long H = 1; H <= N / 2 && H <= (N <= 1000 ? 40 : 3); H++) Hs.push_back(H);
        for (long long H : {N / 2, N / 2 - 1, N / 3, N / 5, N / 16, N / 17, N / 100, N / 128}) if (H > 40 && 2 * H <= N) Hs.push_back(H);
        for (long long H : Hs) {
            if (!hk::regroup_spreadable(N, H)) { std::printf("not spreadable N %lld H %lld\n", N, H); return 1; }
            std::vector<char> seen((size_t)N, 0);
            long long prev = -1;
            for (long long r = 0; r < H; r++) {
                const long long p = hk::regroup_spread_pos(r, true, N, H);
                if (p < 0 || p >= N || seen[(size_t)p] || p <= prev) { std::printf("hinted N %lld H %lld r %lld -> %lld\n", N, H, r, p); return 1; }
                if (prev >= 0 && (p - prev < N / H || p - prev > N / H + 1)) { std::printf("spacing N %lld H %lld r %lld: %lld\n", N, H, r, p - prev); return 1; }
                seen[(size_t)p] = 1; prev = p;
            }
            prev = -1;
            for (long long r = 0; r < N - H; r++) {
                const long long p = hk::regroup_spread_pos(r, false, N, H);
                if (p < 0 || p >= N || seen[(size_t)p] || p <= prev) { std::printf("plain N %lld H %lld r %lld -> %lld\n", N, H, r, p); return 1; }
                seen[(size_t)p] = 2; prev = p;
            }
            for (long long p = 0; p < N; p++) if (!seen[(size_t)p]) { std::printf("hole N %lld H %lld at %lld\n", N, H, p); return 1; }
            cases++;
        }
    }
    std::printf("ok %lld\n", cases);
    return 0;
}
