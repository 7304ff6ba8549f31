// hk_env_solve.h — HierarchicalKartAgent.SolveLQR (HKA:699-1236) on gfx950, inside the fused tick kernel (hk_env_run.h):
//
//   phase_assemble      lane = ego.  Sensor rays of the ego's own kart through the LDS wall grid, the players within 8 m, per
//                       player initial / target / weights and the 7-branch heading heuristic (HKA:726-1198).
//   lq1_solve           a single-player game (n = 4, m = 2; ~99.9 % of the games once the field has spread out) is solved by
//                       the assembling lane itself: the whole Riccati recursion in registers.
//   lqn_body<N>         multi-player games (N = 2 .. GA) are written to a structure-of-arrays buffer (GameSoA), queued by
//                       player count and solved between two launches of the tick kernel by lqn_all_kernel / lqn_single_kernel /
//                       lqn_big_kernel: 64 / (4 N) games per wave through hk_lq_core.h.
//   both decode u0 -> (Accelerate, Brake, Steering) (HKA:1206-1224).
//
// Results are bit-identical to the CPU oracle's dense restatement: see the arithmetic contract in hk_lq_core.h.
// (included once per lane-group width by hk_env_ga.h: no include guard, namespace hk::HK_GA_NS)
#include "hk_env_device.h"
#include "hk_lq_core.h"

namespace hk { namespace HK_GA_NS {

struct GamePlayer {            // one player's share of a game, ego-local order (Q3)
    double x0[4];              // initial (x, z, v, heading)                                  HKA:730-736
    double a4[4];              // A[x,v], A[z,v], A[x,h], A[z,h]                              KartLQRDynamics.cs:45-48
    double tw[4];              // target weights                                              HKA:930-964
    double tgt[4];             // target state                                                HKA:808-926
    double rc;                 // control weight                                              HKA:1192-1196
    double aw[GA - 1];         // avoid weight per other player (x and z share it)            HKA:1019,1114
    double opw[GA - 1][3];     // opponent-target weights (x, z, v)                           HKA:1073-1093,1168-1188
    double opt[GA - 1][3];     // opponent target (x, z, v); heading entry is never set -> 0  HKA:1065-1067
    int M, agent, branch, pad_;
};
// Queued multi-player games live in ONE array of doubles, structure-of-arrays over the games: field f of player i of game g
// (g = env * A + ego) sits at d[(i * GP_FIELDS + f) * ng + g].  The egos of a wave are consecutive games, so every store of
// the assembly and every load of the solver is one contiguous run per wave instruction (the array-of-structs this replaces
// cost 64 scattered sectors per instruction: the 2-agent configuration, where every tick queues a 2-player game per ego,
// spent most of its time there).
constexpr int GP_NO = GA - 1;     // other players of a game, at most
constexpr int GP_X0 = 0, GP_A4 = 4, GP_TW = 8, GP_TGT = 12, GP_RC = 16, GP_AW = 17, GP_OPW = GP_AW + GP_NO, GP_OPT = GP_OPW + 3 * GP_NO,
              GP_M = GP_OPT + 3 * GP_NO, GP_FIELDS = (GP_M + 2) & ~1;      // GA = 4: 20, 29, 38, 40
struct GameSoA {
    double* d;
    size_t ng;
    __device__ __forceinline__ double get(int game, int i, int f) const { return d[((size_t)(i * GP_FIELDS + f)) * ng + game]; }
    __device__ __forceinline__ void put(int game, int i, int f, double v) const { d[((size_t)(i * GP_FIELDS + f)) * ng + game] = v; }
};

// What the parameter block says about THIS lane's agent, read once per launch: indexing the by-value kernel argument with the
// lane's agent index makes every use a vector load from the kernarg segment (and small local arrays filled with run-time
// counts live in scratch), inside the tick loop.  order: the agent's player list [this, teamAgents..., otherAgents...]
// (HKA:702), 4 bits per entry.
struct LaneCfg { int low_mode, high_mode, vbucket, nall; uint32_t order; };
__device__ __forceinline__ LaneCfg lane_cfg(const EnvParams& P, const int i)
{
    LaneCfg c;
    c.low_mode = P.low_mode[i]; c.high_mode = P.high_mode[i]; c.vbucket = P.vbucket[i];
    c.order = (uint32_t)i; c.nall = 1;
    for (int j = 0; j < P.n_team[i]; j++) { c.order |= (uint32_t)P.team[i][j] << (4 * c.nall); c.nall++; }
    for (int j = 0; j < P.n_other[i]; j++) { c.order |= (uint32_t)P.other[i][j] << (4 * c.nall); c.nall++; }
    return c;
}

struct KartS {                 // per-kart staging (LDS), filled by the kart's own lane
    float px, pz, yaw, fx, fz, speed, heading, msfs, dC;
    float ray[5];              // nearest wall distance along sensors 0, 2, 4, 8, 6 (3e38 = none)
    int sec, straight, pl1, pl2;
    float pv1, pv2;
    uint32_t flags;
    int pad_;
};

__device__ __forceinline__ double angle_difference(double a1, double a2)
{   // HKA:1341-1344
    double sd, cd;
    hk_sincos(a2 - a1, &sd, &cd);
    return hk_atan2(sd, cd);
}

// HKA:1206-1224
__device__ inline void decode_controls(const float fs /*final_steer*/, uint32_t& flags, float& steering, double u0a, double u0b,
                                       hk_lq_debug* dbg)
{
    const float maxAng = fs * 0.4f;                                                   // getMaxAngularVelocity AK:505
    float angVel = f_clamp((float)u0b, -maxAng, maxAng);
    uint32_t fl = flags;
    if (u0a < 0) { fl &= ~HK_F_ACCEL; fl |= HK_F_BRAKE; }
    else if (u0a > 0) { fl |= HK_F_ACCEL; fl &= ~HK_F_BRAKE; }
    else { fl &= ~(HK_F_ACCEL | HK_F_BRAKE); angVel = 0.0f; }                         // Q7
    flags = fl;
    steering = angVel / (0.4f * fs);
    if (dbg) { dbg->u0[0] = u0a; dbg->u0[1] = u0b; }
}

// The solver kernels decode into the ego's hot-tile rows (flags, steering), found through slot_of: a queued game is named by env * A + ego
struct HotRef { uint32_t* w; const int* slot_of; };
__device__ __forceinline__ void decode_store(const EnvParams& P, const HotRef& HR, const int game, const double u0a, const double u0b, hk_lq_debug* dbg)
{
    const int env = game / P.A, ego = game - env * P.A;
    uint32_t* p = HR.w + hot_base<GA>(HR.slot_of[env], ego);
    uint32_t fl = hot_get<uint32_t>(p, HF_flags);
    float st = hot_get<float>(p, HF_steering);
    decode_controls(hot_get<float>(p, HF_final_steer), fl, st, u0a, u0b, dbg);
    hot_put<uint32_t>(p, HF_flags, fl); hot_put<float>(p, HF_steering, st);
}

// ---------------------------------------------------------------------------------------------------------------
// single-player games, solved by the assembling thread itself (lq1_solve).  Dense 4x4 / 2x2 algebra with the SAME fma chains as the generic
// algorithm (k ascending, seeded +0.0); entries of A, B that are structural zeros are skipped (exact), entries
// that are 1.0 enter through fma(z, 1.0, s) like any other value.
// ---------------------------------------------------------------------------------------------------------------
#ifndef HK_LQ1_ATTR
#define HK_LQ1_ATTR inline
#endif
__device__ HK_LQ1_ATTR void lq1_solve(const EnvParams& P, const GamePlayer& gp, Hot& h, hk_lq_debug* dbg, int* status)
{
    const double dt = (double)P.dt;
    double A[4][4] = {{1.0, 0.0, gp.a4[0], gp.a4[2]}, {0.0, 1.0, gp.a4[1], gp.a4[3]}, {0.0, 0.0, 1.0, 0.0}, {0.0, 0.0, 0.0, 1.0}};
    const double R[2][2] = {{gp.rc, 0.0}, {0.0, gp.rc}};
    double Q[4], qv[4], x0[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        double d = 0.0;                       // total of the (empty) avoid list, KartLQRCosts.cs:67-79
        d += gp.tw[s];                        // :81-84
        Q[s] = d;
        double t = -gp.tgt[s];                // getQVec :109-113
        qv[s] = t * gp.tw[s];
        x0[s] = gp.x0[s];
    }
    double Z[4][4], eta[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int c = 0; c < 4; c++) Z[r][c] = (r == c) ? Q[r] : 0.0;
        eta[r] = qv[r];
    }
    double Pm[2][4], alpha[2];
    int singular = 0;
#pragma unroll 1
    for (int t = 3; t >= 0; t--) {
        // T1 = Z B (4x2): B[2][0] = B[3][1] = dt
        double T1[4][2];
#pragma unroll
        for (int r = 0; r < 4; r++) { T1[r][0] = fma64(Z[r][2], dt, 0.0); T1[r][1] = fma64(Z[r][3], dt, 0.0); }
        // LHS = R + B'(ZB)
        double L00 = R[0][0] + fma64(dt, T1[2][0], 0.0), L01 = R[0][1] + fma64(dt, T1[2][1], 0.0);
        double L10 = R[1][0] + fma64(dt, T1[3][0], 0.0), L11 = R[1][1] + fma64(dt, T1[3][1], 0.0);
        // ZA rows 2, 3 (the only rows B' picks), then RHS = B'(ZA) (2x4) and B' eta
        double b0[5], b1[5];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            double s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool nz = (k == c) || (k < 2 && c >= 2);      // structural non-zeros of A
                if (nz) { s2 = fma64(Z[2][k], A[k][c], s2); s3 = fma64(Z[3][k], A[k][c], s3); }
            }
            b0[c] = fma64(dt, s2, 0.0);
            b1[c] = fma64(dt, s3, 0.0);
        }
        b0[4] = fma64(dt, eta[2], 0.0);
        b1[4] = fma64(dt, eta[3], 0.0);
        // 2x2 LU, MathNet/JAMA order: column 0 pivot, then column 1 with s = L10*u01 subtracted once
        if (fabs(L10) > fabs(L00)) {
            double tmp = L00; L00 = L10; L10 = tmp;
            tmp = L01; L01 = L11; L11 = tmp;
#pragma unroll
            for (int c = 0; c < 5; c++) { tmp = b0[c]; b0[c] = b1[c]; b1[c] = tmp; }
        }
        if (L00 == 0.0) singular = 1;
        if (L00 != 0.0) L10 = L10 / L00;
        {
            double s = 0.0;
            s += L10 * L01;
            L11 = L11 - s;
        }
        if (L11 == 0.0) singular = 1;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            double temp = b0[c] * L10;
            b1[c] = b1[c] - temp;
            b1[c] = b1[c] / L11;
            temp = b1[c] * L01;
            b0[c] = b0[c] - temp;
            b0[c] = b0[c] / L00;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) { Pm[0][c] = b0[c]; Pm[1][c] = b1[c]; }
        alpha[0] = b0[4]; alpha[1] = b1[4];
        // The last sweep's value update (KartLQR.cs:113-119 at t = 0) feeds nothing: u0 below reads this sweep's P and alpha only.  A quarter of
        // the recursion's fp64 work, skipped (exact: no output depends on it; the oracle computes it and throws it away as the C# does).
        if (t == 0) break;
        // F = A - (0 + B P): rows x, z keep A; rows v, h subtract dt * P
        double F[4][4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            F[0][c] = A[0][c] - 0.0;
            F[1][c] = A[1][c] - 0.0;
            F[2][c] = A[2][c] - (0.0 + fma64(dt, Pm[0][c], 0.0));
            F[3][c] = A[3][c] - (0.0 + fma64(dt, Pm[1][c], 0.0));
        }
        double beta[4];
        beta[0] = 0.0; beta[1] = 0.0;
        beta[2] = 0.0 - fma64(dt, alpha[0], 0.0);
        beta[3] = 0.0 - fma64(dt, alpha[1], 0.0);
        // Z <- (Q + P'(R P)) + F'(Z F)
        double W[4][4], RP[2][4];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 4; k++) s = fma64(Z[r][k], F[k][c], s);
                W[r][c] = s;
            }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                double s = 0.0;
                s = fma64(R[a][0], Pm[0][c], s);
                s = fma64(R[a][1], Pm[1][c], s);
                RP[a][c] = s;
            }
        double Zn[4][4];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                double o = 0.0;
#pragma unroll
                for (int k = 0; k < 4; k++) o = fma64(F[k][r], W[k][c], o);
                double t2 = 0.0;
                t2 = fma64(Pm[0][r], RP[0][c], t2);
                t2 = fma64(Pm[1][r], RP[1][c], t2);
                const double qq = (r == c) ? Q[r] : 0.0;
                Zn[r][c] = (qq + t2) + o;
            }
        // eta <- (q + P'(R alpha)) + F'(eta + Z_new beta)
        double v1[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double zb = 0.0;
#pragma unroll
            for (int k = 0; k < 4; k++) zb = fma64(Zn[r][k], beta[k], zb);
            v1[r] = eta[r] + zb;
        }
        double ra0 = 0.0, ra1 = 0.0;
        ra0 = fma64(R[0][0], alpha[0], ra0); ra0 = fma64(R[0][1], alpha[1], ra0);
        ra1 = fma64(R[1][0], alpha[0], ra1); ra1 = fma64(R[1][1], alpha[1], ra1);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v3 = 0.0;
#pragma unroll
            for (int k = 0; k < 4; k++) v3 = fma64(F[k][r], v1[k], v3);
            double v2 = 0.0;
            v2 = fma64(Pm[0][r], ra0, v2);
            v2 = fma64(Pm[1][r], ra1, v2);
            eta[r] = (qv[r] + v2) + v3;
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) Z[r][c] = Zn[r][c];
    }
    double u0[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 4; c++) s = fma64(-Pm[a][c], x0[c], s);
        u0[a] = s - alpha[a];
    }
    if (singular) atomicOr(status, 1);
    decode_controls(h.final_steer, h.flags, h.steering, u0[0], u0[1], dbg);
}

// one player's share of the ego's game (HKA:726-1198); `out` is a register struct for single-player games and the
// global descriptor otherwise
template <bool SINGLE>
__device__ __forceinline__ void assemble_player(const EnvParams& P, const TabView& T, const int env, const int ego, const int i, const int N,
                                                const int nearbyAgents, const uint32_t plm /* players, 4 bits each */, const KartS* kq, const bool fixed, const int vbucket,
                                                const float dy, GamePlayer* gp, hk_lq_debug* dbg_out, const hk_mcts_state* bel,
                                                const GameSoA& games, const int game)
{
    // Single-player games stay in the register struct `gp` (the others loop below never runs for them).  Multi-player games
    // go straight to the structure-of-arrays buffer: staging them in a local struct would put it in scratch (the per-other
    // fields are indexed by a run-time M), and 80 scratch round trips per player dominated the 2-agent configuration.
    const int A = P.A, L = P.L;
    const KartS& mek = kq[ego];
        const int ki = (int)((plm >> (4 * i)) & 15u);
        const KartS& kk = kq[ki];
        const float speed = kk.speed;
        double initial[4];
        initial[0] = kk.px; initial[1] = kk.pz; initial[2] = speed; initial[3] = kk.heading;
        {   // LinearizedBicycle (KartLQRDynamics.cs:40-62), dt = Time.fixedDeltaTime widened to double (HKA:707)
            const double dt = (double)P.dt;
            double sh, ch;
            hk_sincos(initial[3], &sh, &ch);
            const double a40 = ch * dt, a41 = sh * dt, a42 = -sh * dt * initial[2], a43 = ch * dt * initial[2];
            if (SINGLE) { gp->a4[0] = a40; gp->a4[1] = a41; gp->a4[2] = a42; gp->a4[3] = a43; }
            else { games.put(game, i, GP_A4 + 0, a40); games.put(game, i, GP_A4 + 1, a41); games.put(game, i, GP_A4 + 2, a42); games.put(game, i, GP_A4 + 3, a43); }
        }
        const int s = kk.sec + 1;                                                     // :746
        const int idx = mod_L(P, s), idx2 = mod_L(P, s + 1);
        int laneSel = 0, nextSel = 0;
        double vel = P.max_speed, nextVel = P.max_speed;
        if (ki == ego) {                                                              // :752-764, :782-794
            if (mek.pl1 != 0) {
                laneSel = mek.pl1;
                double pv = mek.pv1 + (fixed ? 0 : vbucket * 2);
                vel = (double)P.max_speed < pv ? (double)P.max_speed : pv;
            }
            if (mek.pl2 != 0) {
                nextSel = mek.pl2;
                double pv = mek.pv2 + (fixed ? 0 : vbucket * 2);
                nextVel = (double)P.max_speed < pv ? (double)P.max_speed : pv;
            }
        } else if (bel) {                                                             // :767-771, :797-801: the EGO's beliefs
            if (bel->belief_lane[ki][idx] != 0) {
                laneSel = bel->belief_lane[ki][idx];
                double pv = (float)bel->belief_vel[ki][idx] + (fixed ? 0 : vbucket * 2);
                vel = (double)P.max_speed < pv ? (double)P.max_speed : pv;
            }
            if (bel->belief_lane[ki][idx2] != 0) {
                nextSel = bel->belief_lane[ki][idx2];
                double pv = (float)bel->belief_vel[ki][idx2] + (fixed ? 0 : vbucket * 2);
                nextVel = (double)P.max_speed < pv ? (double)P.max_speed : pv;
            }
        }   // no belief (only the MCTS planner fills them) -> Trigger / max speed
        float lx, lz, nx, nz, cx, cz;
        lane_marker(T, idx, laneSel, lx, lz);
        lane_marker(T, idx2, nextSel, nx, nz);
        lane_marker(T, idx, 0, cx, cz);
        double target[4];
        target[0] = lx; target[1] = lz;
        target[2] = (speed <= 5.0f) ? 0.0f : vel;                                     // :810-817
        // The 7-branch heading heuristic (:819-926; table in SURVEY 8 a4).  The lanes of a wave sit in different branches, and a
        // wave executes every branch any of its lanes takes: written branch by branch (as the C# is) it paid for up to five
        // atan2 and nine AngleDifference evaluations (all fp64) per player.  Here the branch is decided first with cheap tests,
        // then each transcendental is evaluated ONCE on branch-selected arguments: atan2(L - p), atan2(C - p), one more atan2
        // (B4: N - p, B5: N - L), one inner AngleDifference (B2, B5) and the final one.  Every lane computes exactly the values
        // its own branch computed before (same functions, same arguments, same operation order), so nothing changes numerically.
        const float h1raw = hk_atan2f(lz - kk.pz, lx - kk.px);                        // :821 targetHeading, :827 heading1
        float h1w = h1raw;
        if (h1w < 0) h1w += TWO_PI_F;
        const float h5raw = hk_atan2f(cz - kk.pz, cx - kk.px);                        // heading5 (:831), B6's heading (:909)
        float h5w = h5raw;
        if (h5w < 0) h5w += TWO_PI_F;
        int branch;
        if (mag3(lx - kk.px, dy, lz - kk.pz) <= (kk.straight ? 10.5f : 7.5f)) {       // :823
            const bool cutTrack = T.cut[(idx * 5 + laneSel) * 5 + nextSel] != 0;      // :832 (static geometry)
            const bool hit0 = kk.ray[0] <= speed * 0.5f;                              // :834
            const bool side = (kk.ray[1] <= 2.0f) || (kk.ray[2] <= 1.5f) || (kk.ray[3] <= 1.5f) || (kk.ray[4] <= 2.0f);
            const float dC = kk.dC;
            if (cutTrack && dC > 4.0f) branch = 1;                                    // :846
            else if ((side && (f_sign(h1raw) == f_sign(h5raw))) || hit0) branch = 2;  // :857 (Q12: signs of the un-wrapped values)
            else if (side && (f_sign(h1raw) != f_sign(h5raw))) branch = 3;            // :867
            else if (dC <= 4.0f) branch = 4;                                          // :876
            else branch = 5;                                                          // :891
        } else {
            branch = (kk.ray[0] <= (kk.straight ? 8.0f : 5.0f)) ? 6 : 7;              // :906
        }
        float h3w = 0.0f;                                                             // B4: heading6 = atan2(N - p) :878;  B5: heading2 = atan2(N - L) :893
        if (branch == 4 || branch == 5) {
            const float ay = branch == 4 ? nz - kk.pz : nz - lz;
            const float ax = branch == 4 ? nx - kk.px : nx - lx;
            h3w = hk_atan2f(ay, ax);
            if (h3w < 0) h3w += TWO_PI_F;
        }
        double inner = 0.0;                                                           // B2: AD(heading1, heading5) :860;  B5: AD(heading2, heading1) :898
        if (branch == 2 || branch == 5) {
            const double pa = branch == 2 ? (double)h1raw : (double)h3w;
            const double qa = branch == 2 ? (double)h5w : (double)h1w;
            inner = angle_difference(pa, qa);
        }
        double fth;
        if (branch == 2) fth = h5w - inner * 0.7f;
        else if (branch == 5) fth = h1w - inner * 0.4f;
        else if (branch == 4) fth = h3w;
        else if (branch == 7) fth = h1w;
        else fth = h5w;                                                               // B1, B3, B6
        if (branch == 4) {
            target[0] = nx; target[1] = nz;
            if (speed > 5.0f) target[2] = nextVel;
        }
        if (fth < 0) fth += TWO_PI_F;                                                 // (B6 / B7 values are already in [0, 2 pi))
        {
            const double ad = angle_difference(initial[3], fth);
            fth = initial[3] - (branch == 6 ? ad * 0.85f : ad);                       // :852,862,872,887,900,921 / :918
        }
        target[3] = fth;                                                              // :926
        double tw[4];                                                                 // :930-964
        if (N > 2) tw[3] = (fixed ? 2.5 : 3.5) * nearbyAgents; else tw[3] = (fixed ? 1.9 : 3.5);
        if (speed <= 5.0f) {
            tw[0] = nearbyAgents * 0.3 * 3.1; tw[1] = nearbyAgents * 0.3 * 3.1; tw[2] = nearbyAgents * -2;
        } else {
            double mx = initial[2] > 1 ? initial[2] : 1;
            tw[0] = nearbyAgents * 0.3 * 3.1 / mx; tw[1] = nearbyAgents * 0.3 * 3.1 / mx; tw[2] = nearbyAgents * 5e-4;
        }
        float multiplier;                                                             // :976-1003
        if (A > 2 && N > 2) multiplier = (ki == ego ? (fixed ? 0.55f : 1.0f) : 1.7f) / nearbyAgents;
        else multiplier = (ki == ego ? (fixed ? 0.45f : 1.0f) : 1.3f);
        int M = 0, nearbyOpponents = 0;
        const int no = P.n_other[ki], nt = P.n_team[ki];
        // a single-player game has no other members: the loop below would skip every j
        for (int j = 0; j < (SINGLE ? 0 : no + nt); j++) {                                           // :1004-1190, k's own order (Q3)
            const bool isteam = j >= no;
            const int oi = isteam ? P.team[ki][j - no] : P.other[ki][j];
            bool member = false;
            for (int q = 0; q < N; q++) if ((int)((plm >> (4 * q)) & 15u) == oi) member = true;
            if (!member) continue;
            const KartS& o = kq[oi];
            const float dist = mag3(o.px - kk.px, 0.0f, o.pz - kk.pz);
            const bool far = (dist > 8) || !(o.flags & HK_F_ACTIVE);
            double w = 0.0;
            if (!far) {
                float mult = isteam ? multiplier / 2.0f : multiplier;
                float pw = (float)((double)dist * sqrt((double)dist));                // Mathf.Pow(d, 1.5f)
                w = 1.0f / (pw * mult);
                if (!isteam) nearbyOpponents += 1;
            }
            if (SINGLE) gp->aw[M] = w; else games.put(game, i, GP_AW + M, w);
            const int io = mod_L(P, o.sec + 1);
            float olx, olz; double ov;
            if (oi == ego) {
                lane_marker(T, io, mek.pl1, olx, olz);
                if (isteam) ov = mek.msfs;
                else if (mek.pl1 != 0) {
                    double pv = mek.pv1 + (fixed ? 0 : vbucket * 2);
                    ov = (double)P.max_speed < pv ? (double)P.max_speed : pv;
                } else ov = P.max_speed;
            } else {
                const int bl = bel ? bel->belief_lane[oi][io] : 0;                    // :1054 / :1150
                lane_marker(T, io, bl, olx, olz);
                if (isteam) ov = o.msfs;
                else if (bl != 0) {
                    double pv = (float)bel->belief_vel[oi][io] + (fixed ? 0 : vbucket * 2);
                    ov = (double)P.max_speed < pv ? (double)P.max_speed : pv;
                } else ov = P.max_speed;
            }
            if (SINGLE) { gp->opt[M][0] = olx; gp->opt[M][1] = olz; gp->opt[M][2] = ov; }
            else { games.put(game, i, GP_OPT + 3 * M + 0, olx); games.put(game, i, GP_OPT + 3 * M + 1, olz); games.put(game, i, GP_OPT + 3 * M + 2, ov); }
            double mx = initial[2] > 1 ? initial[2] : 1;
            double w0, w1, w2;
            if (!isteam) {
                if (far) { w0 = 0.0; w1 = 0.0; w2 = 0; }
                else if (N > 2) { w0 = (fixed ? 0.1 : 0.2) / (mx * nearbyAgents); w1 = (fixed ? 0.1 : 0.2) / (mx * nearbyAgents); w2 = 0.08 / nearbyAgents; }
                else { w0 = (fixed ? 0.1 : 0.2) / mx; w1 = (fixed ? 0.1 : 0.2) / mx; w2 = 0.08; }
            } else {
                if (far || nearbyOpponents < 1) { w0 = 0.0; w1 = 0.0; w2 = 0; }
                else if (N > 2) { w0 = -(fixed ? 0 : 3e-5) / (mx * nearbyAgents); w1 = -(fixed ? 0 : 3e-5) / (mx * nearbyAgents); w2 = 0 / nearbyAgents; }
                else { w0 = -(fixed ? 1e-4 : 2e-4) / mx; w1 = -(fixed ? 1e-4 : 2e-4) / mx; w2 = 0; }
            }
            if (SINGLE) { gp->opw[M][0] = w0; gp->opw[M][1] = w1; gp->opw[M][2] = w2; }
            else { games.put(game, i, GP_OPW + 3 * M + 0, w0); games.put(game, i, GP_OPW + 3 * M + 1, w1); games.put(game, i, GP_OPW + 3 * M + 2, w2); }
            M++;
        }
        double controlcost = 0.115;                                                   // :1192-1196
        if (N > 2) controlcost = fixed ? 0.135 : 0.25;
        if (SINGLE) {
            gp->rc = 1.0 * controlcost;                                               // getRMatrix: SparseIdentity * w
            gp->M = M; gp->agent = ki; gp->branch = branch;
#pragma unroll
            for (int c = 0; c < 4; c++) { gp->x0[c] = initial[c]; gp->tw[c] = tw[c]; gp->tgt[c] = target[c]; }
        } else {
            games.put(game, i, GP_RC, 1.0 * controlcost);
            games.put(game, i, GP_M, (double)M);
#pragma unroll
            for (int c = 0; c < 4; c++) { games.put(game, i, GP_X0 + c, initial[c]); games.put(game, i, GP_TW + c, tw[c]); games.put(game, i, GP_TGT + c, target[c]); }
        }
        if (dbg_out && (P.debug & 1)) {
            hk_lq_debug* d = &dbg_out[(size_t)env * A + ego];
            d->player_agent[i] = ki; d->branch[i] = branch; d->control_w[i] = controlcost;
            for (int c = 0; c < 4; c++) { d->initial[i][c] = initial[c]; d->target[i][c] = target[c]; d->target_w[i][c] = tw[c]; }
        }
    }

// LDS writes by the lanes of a quad, then reads by the other lanes of the SAME quad: one wave, so ordering the
// compiler is all that is needed (the LDS serves a wave's requests in order).  No block barrier: quads leave the
// fused tick loop at different times.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------------------------
// phase B1 of a tick: the ego's sensor rays, game assembly; a single-player game is solved on the spot, a multi-player
// game is written to games[] (the caller queues it by player count).  Returns the player count of a game to queue, else 0.
// `act`: this env solves on this tick (cadence, not parked); it is quad-uniform.
// ---------------------------------------------------------------------------------------------------------------
#ifndef HK_ASM_ATTR
#define HK_ASM_ATTR inline
#endif
__device__ HK_ASM_ATTR int phase_assemble(const EnvParams& P, const TabView& T, KartS* ks, const int env, const int ego,
                                      const bool act, const hk_env_state& es, Hot& h, const float hfx, const float hfz, hk_agent_state* agents,
                                      const GameSoA games,
                                      int* queue_cnt, int* queue, hk_lq_debug* dbg_out, int* status, const hk_mcts_state* mcts_all,
                                      const LaneCfg& LC)
{
    const int A = P.A, L = P.L;
    const bool me = act && ego < A;
    HK_LP(3);
    const hk_agent_state* a = me ? &agents[(size_t)env * A + ego] : nullptr;
    KartS k;
    k.px = k.pz = k.yaw = k.fx = k.fz = k.speed = k.heading = k.msfs = k.dC = 0.0f;
    k.sec = 0; k.straight = 1; k.pl1 = k.pl2 = 0; k.pv1 = k.pv2 = 0.0f; k.flags = 0; k.pad_ = 0;
#pragma unroll
    for (int q = 0; q < 5; q++) k.ray[q] = 3.0e38f;
    if (me) {
        HK_LP(4);
        k.px = h.px; k.pz = h.pz; k.yaw = h.yaw;
        k.fx = hfx; k.fz = hfz;
        const float vx = h.vx, vz = h.vz;
        k.speed = mag3(vx, 0.0f, vz);
        float heading = hk_atan2f(k.fz, k.fx);                                        // HKA:734
        if (heading < 0) heading += TWO_PI_F;
        k.heading = heading;
        k.msfs = max_speed_for_state(P, k.fx, k.fz, vx, vz, h.wy, h.final_steer);
        k.sec = h.section_index;
        k.straight = is_straight(P, T, k.sec) ? 1 : 0;
        k.flags = h.flags;
        const int i1 = mod_L(P, k.sec + 1), i2 = mod_L(P, k.sec + 2);
        k.pl1 = a->plan_lane[i1]; k.pv1 = a->plan_vel[i1];
        k.pl2 = a->plan_lane[i2]; k.pv2 = a->plan_vel[i2];
        {   // BoxCollider.ClosestPoint distance to the next section's Trigger (HKA:846,876)
            const SecDev& s = T.sec[i1];
            float relx = k.px - s.trig_x, relz = k.pz - s.trig_z;
            float lx = relx * s.fz + relz * (-s.fx);
            float lz = relx * s.fx + relz * s.fz;
            float dx = lx - f_clamp(lx, -TRIG_HX, TRIG_HX);
            float dz = lz - f_clamp(lz, -TRIG_HZ, TRIG_HZ);
            k.dC = sqrtf(dx * dx + dz * dz);
        }
        // sensor rays of the own kart (Physics.Raycast vs TrackMask, HKA:834-844,906) through the wall grid.  Sensors
        // 2, 4, 8, 6 are at most 2 m long: every wall they can hit is in the list of the origin's cell.  Sensor 0 is
        // compared with up to 8 m (and with speed / 2): a hit at distance t <= 2 (ns - 1) + 1.3 lies within 1.3 m of one of the
        // samples o + {0, 2, .., 2 (ns - 1)} d, so the union of those cells' NEAR lists (NEAR_REACH 1.3 m) contains it.  Lists are
        // supersets; below the largest threshold the minimum is what a scan of every wall (the oracle) returns.
        const float ox = k.px + SENSOR_LZ * k.fx, oz = k.pz + SENSOR_LZ * k.fz;
        HK_ST(h, 14);
#ifndef HK_DUMMY_NO_RAYS          /* (timing / counter experiments only: tools/build_variant.py) */
        {
            float d0x, d0z;
            sensor_dir(P, 0, k.fx, k.fz, d0x, d0z);
            float best = 3.0e38f;
            int prev = -1;
            // the largest distance this kart's ray is ever compared with (assemble_player: speed / 2, 8 m on a straight, 5 m in a curve):
            // samples up to 2 (ns - 1) >= thr - 1.25 cover every hit that can decide a comparison
            const float thr = f_max(k.speed * 0.5f, k.straight ? 8.0f : 5.0f);
            int ns = 1 + (int)ceilf((thr - 1.25f) * 0.5f);
            ns = ns > 6 ? 6 : ns;
#pragma unroll 1
            for (int sm = 0; sm < ns; sm++) {
                const float sd = 2.0f * (float)sm;
                // every hit up to 2 (sm - 1) + 1 m lies within 1 m of one of the samples walked so far, i.e. in their near lists: a hit that
                // close is final (the walls a later sample adds can only be hit further out)
                if (best <= sd - 1.0f) break;
                const int cell = grid_cell(P, ox + d0x * sd, oz + d0z * sd);
                if (cell == prev) continue;
                prev = cell;
                HK_LP(5);
                const int w0 = T.grid_off[cell], w1 = w0 + T.near_cnt[cell];
                // two walls per trip: their index -> segment load chains are in flight together (the LDS round trips, not the arithmetic, are
                // what a trip costs); an odd list tests its last wall twice, which a minimum does not notice
                for (int w = w0; w < w1; w += 2) {
                    HK_LP(6);
                    const hk_wall_seg wa = T.walls[T.grid_idx[w]], wb = T.walls[T.grid_idx[w + 1 < w1 ? w + 1 : w]];
                    const float ta = ray_seg(ox, oz, d0x, d0z, wa), tb = ray_seg(ox, oz, d0x, d0z, wb);
                    if (ta >= 0.0f && ta < best) best = ta;
                    if (tb >= 0.0f && tb < best) best = tb;
                }
            }
            k.ray[0] = best;
        }
        HK_ST(h, 15);
        {
            const int ssel[4] = {2, 4, 8, 6};
            float ddx[4], ddz[4], best[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                sensor_dir(P, ssel[q], k.fx, k.fz, ddx[q], ddz[q]);
                best[q] = 3.0e38f;
            }
            const int cell = grid_cell(P, ox, oz);
            const int w0 = T.grid_off[cell], w1 = T.grid_off[cell + 1];
            // Two passes over the cell's list, 32 walls at a time (round 6; the wall-contact pass of phase_move has had this form since round 2): a cheap
            // distance test marks the walls that can matter (a bit per wall), then only those get the four ray tests, in list order.  The four rays are
            // compared with 2 m and 1.5 m only (side, below): a wall further than that from the ray origin cannot change any of the comparisons
            // (1 - 2 cm margin >> float rounding).  In one pass — test, then the rays behind an `if` — the wave ran the four ray tests for nearly every
            // listed wall, because with 64 lanes in as many places SOME lane's wall passes on almost every trip; now it runs them as often as the lane with
            // the most walls in reach needs (a minimum does not depend on the order, and the order is the list's anyway).
            for (int base = w0; base < w1; base += 32) {
                const int nq = (w1 - base) < 32 ? (w1 - base) : 32;
                uint32_t cand = 0;
                for (int q = 0; q < nq; q += 4) {
                    HK_LP(7);
                    hk_wall_seg ws[4];         // (four index -> wall load chains in flight; slots past the end re-read the last wall and are masked out)
#pragma unroll
                    for (int j = 0; j < 4; j++) ws[j] = T.walls[T.grid_idx[base + ((q + j) < nq ? (q + j) : (nq - 1))]];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
#ifndef HK_NO_SHORT_RAY_CULL
                        const bool apart = !wall_within(ws[j], ox, oz, 2.0f + 0.02f);
#else
                        const bool apart = false;
#endif
                        cand |= ((apart || (q + j) >= nq) ? 0u : 1u) << (q + j);
                    }
                }
                while (cand) {
                    const int q = __ffs((int)cand) - 1;
                    cand &= cand - 1u;
                    HK_LP(8);
                    const hk_wall_seg ws = T.walls[T.grid_idx[base + q]];
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) {
                        const float t = ray_seg(ox, oz, ddx[r4], ddz[r4], ws);
                        if (t >= 0.0f && t < best[r4]) best[r4] = t;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) k.ray[1 + q] = best[q];
        }
#endif
    }
    HK_ST(h, 16);
    ks[threadIdx.x] = k;
    HK_ST(h, 17);
    wave_lds_sync();
    HK_ST(h, 2);                       // [2] own-kart staging + the five wall rays
    const KartS* kq = &ks[threadIdx.x & ~(GA - 1)];   // the env's karts
    const bool solving = me && (k.flags & HK_F_ENABLED) && LC.low_mode == HK_LOW_LQR && !((es.inactive_mask >> ego) & 1u);
    const float dy = T.sec[0].marker_y - P.kart_y;                                    // Q13
    // my multi-player game, if I hold one: players (4 bits each), their count, HKA:721's count
    uint32_t plm = 0;
    int Nmp = 0, nearbyAgents = -1;
    const bool fixed = LC.high_mode == HK_HIGH_FIXED;
    if (solving) {
        // ---- players (HKA:702-725)
        int N = 0;
        for (int q = 0; q < LC.nall; q++) {
            const uint32_t who = (LC.order >> (4 * q)) & 15u;
            bool in = true;
            if (A > 2) {
                const KartS& o = kq[who];
                in = mag3(o.px - k.px, 0.0f, o.pz - k.pz) < 8;
                if (in) nearbyAgents += 1;
            }
            if (in) { plm |= who << (4 * N); N++; }
        }
        nearbyAgents = nearbyAgents > 1 ? nearbyAgents : 1;
        HK_ST(h, 3);                       // [3] players within 8 m
        if (N == 1) {
            // single-player game: assemble into registers and run the whole Riccati recursion right here
            const hk_mcts_state* bel = mcts_all ? &mcts_all[(size_t)env * A + ego] : nullptr;
            GamePlayer loc;
            HK_LP(9);
            assemble_player<true>(P, T, env, ego, 0, N, nearbyAgents, plm, kq, fixed, LC.vbucket, dy, &loc, dbg_out, bel, games, 0);
            HK_ST(h, 4);                   // [4] single-player assembly (heading heuristic, weights)
            if (dbg_out && (P.debug & 1)) dbg_out[(size_t)env * A + ego].n_players = 1;
            HK_LP(10);
#ifndef HK_DUMMY_NO_LQ1
            lq1_solve(P, loc, h, (dbg_out && (P.debug & 1)) ? &dbg_out[(size_t)env * A + ego] : nullptr, status);
#else
            h.steering = (float)loc.tgt[3] * 0.01f; h.flags |= HK_F_ACCEL;
#endif
            HK_ST(h, 5);                   // [5] lq1_solve
        } else {
            Nmp = N;
            if (dbg_out && (P.debug & 1)) dbg_out[(size_t)env * A + ego].n_players = N;
        }
    }
    // ---- multi-player games: every (ego, player) share of the lane group's games is an item, and the group's lanes take the items in turn.
    // An ego used to assemble its N players one after the other in its own lane while the lanes without a game waited: once the field has
    // spread, a game is a rare event (77 two-player games per solve launch of 262 144 egos), but the few waves that hold one finished
    // 10 us after the others and the launch ended when they did (B1 kernel 106 -> 83 us with the assembly compiled out).  The two egos of a
    // two-player game now take one share each of both games: half the time.  Same function, same arguments: the descriptors are unchanged.
    // (The exchanges are among the lanes of the group, and the trip count is the same in all of them.)
    {
        const int lane0 = threadIdx.x & ~(GA - 1), li = threadIdx.x & (GA - 1);
        int offs[GA], tot = 0;
#pragma unroll
        for (int q = 0; q < GA; q++) { offs[q] = tot; tot += quad_get(Nmp, q); }
        for (int it0 = 0; it0 < tot; it0 += GA) {
            HK_LP(11);
            const int it = it0 + li;
            int e = 0;
#pragma unroll
            for (int q = 0; q < GA; q++) if (it >= offs[q]) e = q;          // the last ego whose first item is not beyond `it` (egos without a game share their successor's offset)
            int off_e = 0;
#pragma unroll
            for (int q = 0; q < GA; q++) if (e == q) off_e = offs[q];
            const uint32_t plm_e = (uint32_t)__shfl((int)plm, lane0 | e, 64);
            const int N_e = __shfl(Nmp, lane0 | e, 64), nb_e = __shfl(nearbyAgents, lane0 | e, 64);
            const int fixed_e = __shfl(fixed ? 1 : 0, lane0 | e, 64), vb_e = __shfl(LC.vbucket, lane0 | e, 64);
            if (it < tot)
                assemble_player<false>(P, T, env, e, it - off_e, N_e, nb_e, plm_e, kq, fixed_e != 0, vb_e, dy, nullptr, dbg_out,
                                       mcts_all ? &mcts_all[(size_t)env * A + e] : nullptr, games, env * A + e);
        }
    }
    HK_ST(h, 7);                       // [7] multi-player assembly
    return Nmp;              // the caller bins the game by its player count (wave-aggregated slot allocation needs every queued lane together)
}

#ifndef HK_HOST_EMU          // the solver kernels are not part of the host emulation of the tick kernel
// ---------------------------------------------------------------------------------------------------------------
// K_B2b: multi-player games of one size NP from queue[NP-2], 64/(4*NP) per wave
// ---------------------------------------------------------------------------------------------------------------
template <int NP>
struct CostRows {              // compact cost rows of one game: QC[i][b'][r] = Q_i[r][4b' + (r&3)]
    double QC[NP][NP][4 * NP];
    double QV[NP][4 * NP];
};
template <int NP>
struct QCompact {
    const CostRows<NP>* C;
    __device__ double Q(int i, int r, int c) const { return ((c & 3) == (r & 3)) ? C->QC[i][c >> 2][r] : 0.0; }
    __device__ double q(int i, int r) const { return C->QV[i][r]; }
};

// one queued game: inputs from the GameSoA buffer -> the game's LDS slice LG (any type with Ab / Bb / Rb / x0) and its compact cost rows;
// lane r = row r of the game, r < 4 NP
template <int NP, class LDS>
__device__ __forceinline__ void lqn_stage_inputs(const int game, const int r, const double dt, const GameSoA& games, LDS& LG, CostRows<NP>& CR)
{
    constexpr int n = LqDims<NP>::n;
    // game inputs -> LDS
#pragma unroll
    for (int i = 0; i < NP; i++) {
        for (int e = r; e < 16; e += n) {
            const int rr = e >> 2, cc = e & 3;
            double av = 0.0;
            if (rr == cc) av = 1.0;
            else if (rr == 0 && cc == 2) av = games.get(game, i, GP_A4 + 0);
            else if (rr == 1 && cc == 2) av = games.get(game, i, GP_A4 + 1);
            else if (rr == 0 && cc == 3) av = games.get(game, i, GP_A4 + 2);
            else if (rr == 1 && cc == 3) av = games.get(game, i, GP_A4 + 3);
            LG.Ab[i][e] = av;
        }
        for (int e = r; e < 8; e += n) LG.Bb[i][e] = (e == 4 || e == 7) ? dt : 0.0;      // B[v][0] = B[h][1] = dt
        if (r < 4) LG.Rb[i][r] = (r == 0 || r == 3) ? games.get(game, i, GP_RC) : 0.0;
    }
    LG.x0[r] = games.get(game, r >> 2, GP_X0 + (r & 3));
    // compact reach-avoid cost rows (KartLQRCosts.cs:57-127): lane r = row r
    {
        const int b = r >> 2, sidx = r & 3;
#pragma unroll
        for (int i = 0; i < NP; i++) {
            double qc[NP];
#pragma unroll
            for (int q = 0; q < NP; q++) qc[q] = 0.0;
            double qv = 0.0;
            const int M = (int)games.get(game, i, GP_M);
            auto AW = [&](int j) { return games.get(game, i, GP_AW + j); };
            auto TW = [&](int c) { return games.get(game, i, GP_TW + c); };
            auto TGT = [&](int c) { return games.get(game, i, GP_TGT + c); };
            auto OPW = [&](int j, int c) { return games.get(game, i, GP_OPW + 3 * j + c); };
            auto OPT = [&](int j, int c) { return games.get(game, i, GP_OPT + 3 * j + c); };
            if (b == 0) {
                double d = 0.0;
                if (sidx < 2) {
                    double total = 0.0;                                    // :67-79
                    for (int j = 0; j < M; j++) total -= AW(j);
                    d = total;
                }
                d += TW(sidx);                                          // :81-84
                qc[0] = d;
                if (sidx < 2) {
#pragma unroll
                    for (int q = 1; q < NP; q++) if (M > q - 1) qc[q] = AW(q - 1);
                }
                double t = -TGT(sidx);                                  // getQVec :109-113
                qv = t * TW(sidx);
            } else {
                const int j = b - 1;
                if (sidx < 2) qc[0] = AW(j);                            // :74
                double dg = 0.0;
                if (sidx < 3) dg = -OPW(j, sidx);                       // :91 assignment (Q4)
#pragma unroll
                for (int q = 1; q < NP; q++) if (b == q) qc[q] = dg;
                if (sidx < 3) { qv = OPT(j, sidx); qv = qv * -OPW(j, sidx); }   // :117,:121 (heading entry 0)
            }
#pragma unroll
            for (int q = 0; q < NP; q++) CR.QC[i][q][r] = qc[q];
            CR.QV[i][r] = qv;
        }
    }
}

// ... then the coupled Riccati recursion on the lane-per-row core (hk_lq_core.h; games of 5 .. 8 players)
template <int NP, class SYNC, bool MFMA = false>
__device__ __forceinline__ void lqn_stage_solve(const int game, const int r, const double dt, const GameSoA& games, LqGameLds<NP>& LG,
                                                CostRows<NP>& CR, double u0[2], int& singular, LqGameLds<NP>* slots = nullptr)
{
    SYNC::sync();              // the slice may still be read by the previous game's last sweep
    lqn_stage_inputs<NP>(game, r, dt, games, LG, CR);
    SYNC::sync();
    QCompact<NP> qp;
    qp.C = &CR;
    lq_solve_game<NP, QCompact<NP>, true, SYNC, MFMA>(r, LG, qp, 3, u0, singular, slots);      // HKA:1201 horizon literal 3 (Q6)
}

// 3- and 4-player games when a round holds a handful of them (a spread field): ONE game per wave, whole solve in 27 us instead of the
// lane-per-row core's 59 (hk_lq_mfma.h).  When a round holds thousands (the race start) lqn_body<NP, true> serves 64 / n games per wave.
template <int NP>
__device__ __forceinline__ void lqn_body_mfma(const int block, const int nblocks, const EnvParams& P, const HotRef hr, const GameSoA games,
                                              const int* queue_cnt, const int* queue, hk_lq_debug* dbg_out, int* status, unsigned char* smem,
                                              unsigned long long* gstats)
{
    constexpr int n = 4 * NP;
    LqMfmaLds<NP>& LG = *reinterpret_cast<LqMfmaLds<NP>*>(smem);
    CostRows<NP>& CR = *reinterpret_cast<CostRows<NP>*>(smem + sizeof(LqMfmaLds<NP>));
    const int lane = threadIdx.x & 63;
    const int count = queue_cnt[NP];
    if (block == 0 && threadIdx.x == 0 && count > 0) atomicAdd(&gstats[NP], (unsigned long long)count);   // hk_prof_games
    const int* qbase = queue + (size_t)(NP - 2) * P.E * P.A;
    for (int slot = block; slot < count; slot += nblocks) {
        const int game = qbase[slot];
        LqmDev::sync();              // the slice may still be read by the previous game's last sweep
        if (lane < n) lqn_stage_inputs<NP>(game, lane, (double)P.dt, games, LG, CR);
        LqmDev::sync();
        QCompact<NP> qp;
        qp.C = &CR;
        double u0[2];
        int singular = 0;
        lq_solve_game_mfma<NP, QCompact<NP>, LqmDev>(lane, LG, qp, 3, u0, singular);       // HKA:1201 horizon literal 3 (Q6)
        if (lane == 0) {
            if (singular) atomicOr(status, 1);
            decode_store(P, hr, game, u0[0], u0[1], (dbg_out && (P.debug & 1)) ? &dbg_out[game] : nullptr);
        }
    }
}

#ifndef HK_LQN_SYNC
#define HK_LQN_SYNC LqBlockSync
#endif
template <int NP, bool MFMA = false>
__device__ __forceinline__ void lqn_body(const int block, const int nblocks, const EnvParams& P, const HotRef hr, const GameSoA games,
                                         const int* queue_cnt, const int* queue, hk_lq_debug* dbg_out, int* status, unsigned char* smem,
                                         unsigned long long* gstats)
{
    constexpr int n = LqDims<NP>::n, GPW = LqDims<NP>::GPW, SLOTS = LqDims<NP>::SLOTS;
    LqGameLds<NP>* lds = reinterpret_cast<LqGameLds<NP>*>(smem);                                     // [SLOTS]
    CostRows<NP>* rows = reinterpret_cast<CostRows<NP>*>(smem + sizeof(LqGameLds<NP>) * SLOTS);     // [SLOTS]
    const int lane = threadIdx.x & 63;
    const int gs = lane / n, r = lane % n;
    const int count = queue_cnt[NP];
    if (block == 0 && threadIdx.x == 0 && count > 0) atomicAdd(&gstats[NP], (unsigned long long)count);   // hk_prof_games
    const int* qbase = queue + (size_t)(NP - 2) * P.E * P.A;
    LqGameLds<NP>& LG = lds[gs];
    CostRows<NP>& CR = rows[gs];
    for (int base = block * GPW; base < count; base += nblocks * GPW) {
        const int slot = base + gs;
        const bool live = gs < GPW && slot < count;
        const int game = qbase[live ? slot : count - 1];      // idle slots recompute the last game and discard it
        double u0[2];
        int singular = 0;
        lqn_stage_solve<NP, HK_LQN_SYNC, MFMA>(game, r, (double)P.dt, games, LG, CR, u0, singular, lds);
        if (live && r == 0) {
            if (singular) atomicOr(status, 1);
            decode_store(P, hr, game, u0[0], u0[1], (dbg_out && (P.debug & 1)) ? &dbg_out[game] : nullptr);
        }
    }
}

// The solver kernel of a round (lqn_round_kernel: 2-player games on pairs of lanes, 3- / 4-player games on the generic core) lives
// in hk_lq2_pair.h.  (Round 1 / early round 2 went through three stages here, all retired: three launches per round, one merged
// launch at the 4-player body's occupancy — 0.6 ms for the 262 144 games of a race-start tick — and a 2-player kernel of its own at
// two waves per SIMD, 0.38 ms per 100 000 games.)
#include "hk_lq2_pair.h"
#include "hk_lq_spread.h"

#if HK_GA > 4
// Games with 5..8 players (only the synthetic 8-agent configuration has them).  The generic core is the same; beyond 4 players a
// lane's value-matrix rows (NP x 4 NP doubles) exceed the register file and spill — the more players the more — so the sizes
// are compiled in two kernels ({5, 6} and {7, 8}: a kernel's register / scratch budget is that of its largest size; the common
// 5- and 6-player games would otherwise run with the 8-player allocation).  Blocks [0, nb) take the lower size, [nb, 2 nb) the
// higher one.  Functional, not tuned: such games need more than 4 karts within 8 m of each other.
template <int NA>
__global__ __launch_bounds__(64) void lqn_big_kernel(EnvParams P, const HotRef hr, const GameSoA games, const int* queue_cnt,
                                                     const int* queue, hk_lq_debug* dbg_out, int* status, int nb, unsigned long long* gstats)
{
    constexpr int NB = NA + 1;
    constexpr size_t BA = (sizeof(LqGameLds<NA>) + sizeof(CostRows<NA>)) * LqDims<NA>::SLOTS;
    constexpr size_t BB = (sizeof(LqGameLds<NB>) + sizeof(CostRows<NB>)) * LqDims<NB>::SLOTS;
    constexpr size_t BMAX = BA > BB ? BA : BB;
    static_assert(BMAX <= 160 * 1024, "one workgroup's games must fit the CU's LDS");
    __shared__ __align__(16) unsigned char smem[BMAX];
    const int which = blockIdx.x / nb, b = blockIdx.x - which * nb;
    if (which == 0) lqn_body<NA>(b, nb, P, hr, games, queue_cnt, queue, dbg_out, status, smem, gstats);
    else lqn_body<NB>(b, nb, P, hr, games, queue_cnt, queue, dbg_out, status, smem, gstats);
}
#endif

#endif  // HK_HOST_EMU

} }  // namespace hk::HK_GA_NS
