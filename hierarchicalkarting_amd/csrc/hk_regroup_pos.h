// Where a lane group goes inside its class (same solve phase, same done / not done) when the regroup SPREADS the envs that hold multi-player games
// instead of packing them (round 6).  With the games solved in-wave (hk_lq_spread.h lqs_inwave) a wave solves its games three or four to a pass, one
// pass after the other: packs that share a wave — what round 3's pack hint arranges for the queues, and what the tail of a lazy call leaves behind —
// make the slowest wave of a B1 launch run several passes while nine waves in ten have none (B1 launch of the driver's window: 91 us against 73 us
// for the same games where no regroup had packed them; profiles/r06_f_spread_regroup.txt).  The class of N lane groups holds H hinted ones: hinted
// rank r goes to slot floor(r N / H) — evenly spaced, at most one per wave while H <= N / 16 —, the plain ones fill the free slots in rank order, so
// neighbours stay neighbours and the scatter's writes stay coalesced.  Plain C++ (host + device); tests/regroup_pos_host_check.cpp proves the bijection.
#pragma once
#ifdef __HIPCC__
#define HK_RGP_HD __host__ __device__
#else
#define HK_RGP_HD
#endif
namespace hk {
// hinted slots strictly below p: #{r in [0, H) : floor(r N / H) < p} = ceil(p H / N)
HK_RGP_HD inline long long regroup_hinted_below(long long p, long long N, long long H) { return (p * H + N - 1) / N; }
// may the class be spread?  (hinted slots at least two apart, so that the slot after a hinted one is free)
HK_RGP_HD inline bool regroup_spreadable(long long N, long long H) { return H > 0 && 2 * H <= N; }
// rank r among the class's hinted (hinted = true) or plain lane groups -> slot in [0, N)
HK_RGP_HD inline long long regroup_spread_pos(long long r, bool hinted, long long N, long long H)
{
    if (hinted) return (r * N) / H;
    // the r-th free slot: the least p with p - hinted_below(p) = r that is not itself a hinted slot.  p = r + hinted_below(p) from below: the start
    // r N / (N - H) rounded down is at most the answer and the map is monotone, so the iteration climbs to the least fixed point (a few steps: the
    // error shrinks by H / N each time)
    long long p = r + (r * H) / (N - H);
    for (int it = 0; it < 64; it++) {
        const long long pn = r + regroup_hinted_below(p, N, H);
        if (pn == p) break;
        p = pn;
    }
    if (regroup_hinted_below(p + 1, N, H) - regroup_hinted_below(p, N, H) == 1) p += 1;      // the fixed point is a hinted slot: the free one behind it
    return p;
}
}  // namespace hk
