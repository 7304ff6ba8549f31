// hk_lq2_pair.h — TWO-player feedback LQ Nash games (KartLQR.solveFeedbackLQR with N = 2: n = 8 states, m = 4 controls) solved by a
// PAIR of lanes per game: lane p owns player p's value matrix Z_p (all 8 rows) and that player's whole update.
//
// Why.  Two karts within 8 m of each other are by far the most common multi-player game (every game of a 1v1 race, the start
// grid's row mates, every overtake).  The generic core (hk_lq_core.h) gives a game 8 lanes, one per row, and stages everything
// that crosses rows through LDS between workgroup barriers: ~85 barriers per solve, a serial pivot loop, and a wave that issues
// fp64 FMAs less than a tenth of the time (100 000 games took 0.38 ms: 8 % of the fp64 vector rate).  A player's update
//     Z_p <- Q_p + P_p' R_p P_p + F' Z_p F,   eta_p <- q_p + P_p' R_p alpha_p + F' (eta_p + Z_p beta)
// needs nothing of the other player but the shared 4 x 4 solve, so here the two players of a game run side by side in two lanes
// and talk only there: 26 doubles each way for [LHS | RHS], 16 for the other half of P — DPP quad permutes, no barrier anywhere.
// Z_p lives in the lane's private LDS column (64 doubles, read and written once per sweep, conflict-free), W = Z_p F and the dense
// rows of F in registers, every loop unrolled.  32 games per wave, one wave per SIMD (40 KB of LDS per wave).
//
// ARITHMETIC: every value is produced by exactly the operations, in exactly the order, of lq_solve_game<2, QCompact<2>, true> on
// the game lqn_stage_solve would stage (k-ascending fma chains seeded with +0.0; JAMA-order LU with plain mul / add; the
// reference's quirks Q1, Q2, Q4).  Terms whose factor is a structural +0.0 of the linearised bicycle (B has only its two dt
// entries; the position rows of A and of F = A - sum B_k P_k are zero outside their own player's block) are left out —
// fma(z, +0.0, s) = s for finite z — and factors of 1.0 are kept as fma(z, 1.0, s), as lq1_solve does for the single-player game.
// The parity tests compare the decoded controls and every kart field with the CPU oracle bit for bit.
// (included by hk_env_solve.h inside namespace hk::HK_GA_NS)
#pragma once

struct Lq2PairLds {
    double z[64][64];          // [8 r + c][lane]: Z_p, private to the lane
    double pp[16][64];         // [8 a + c][lane]: rows 2 p + a of P (the lane's own player), kept from the solve to the end of the sweep
};

// SEL 0 / 1: the value held by lane 0 / 1 of the calling lane's pair; 2: the value of the pair's other lane.  (quad_perm: every
// lane of the quad must be active)
#ifndef HK_LQ2_NO_DPP
template <int SEL> __device__ __forceinline__ double pair_get(const double v)
{
    constexpr int ctrl = SEL == 0 ? 0xA0 : (SEL == 1 ? 0xF5 : 0xB1);          // [0,0,2,2] / [1,1,3,3] / [1,0,3,2]
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
#endif

// `game`: index into the GameSoA buffer (the same in both lanes of the pair); p = lane & 1; returns player 0's control at t = 0
__device__ __forceinline__ void lq2_pair_solve(const int game, const int p, const int lane, const double dt, const GameSoA& games,
                                               Lq2PairLds& S, double u0[2], int& singular)
{
    constexpr int NP = 2, n = 8, m = 4;
    // ---- the game's constants.  A entries of both players; control weight and compact cost rows of the lane's own player
    double a4[NP][4];
#pragma unroll
    for (int j = 0; j < NP; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) a4[j][e] = games.get(game, j, GP_A4 + e);
    const double rc = games.get(game, p, GP_RC);
    // compact cost rows of player p (KartLQRCosts.cs:57-127): Q_p(r, c) = ((c & 3) == (r & 3)) ? qc[c >> 2][r] : 0, q_p(r) = qv[r]
    double qc0[n], qc1[n], qv[n];
    {
        const int M = (int)games.get(game, p, GP_M);
        const double aw0 = games.get(game, p, GP_AW + 0);
        double total = 0.0;                                                   // :67-79
        if (M > 0) total -= aw0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const double tw = games.get(game, p, GP_TW + s), tg = games.get(game, p, GP_TGT + s);
            double d = s < 2 ? total : 0.0;
            d += tw;                                                          // :81-84
            qc0[s] = d;
            qc1[s] = (s < 2 && M > 0) ? aw0 : 0.0;
            const double t = -tg;                                             // getQVec :109-113
            qv[s] = t * tw;
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            double opw = 0.0, opt = 0.0;
            if (s < 3) { opw = games.get(game, p, GP_OPW + s); opt = games.get(game, p, GP_OPT + s); }
            qc0[4 + s] = s < 2 ? aw0 : 0.0;                                   // :74
            qc1[4 + s] = s < 3 ? -opw : 0.0;                                  // :91 assignment (Q4)
            double v = 0.0;
            if (s < 3) { v = opt; v = v * -opw; }                             // :117, :121 (heading entry 0)
            qv[4 + s] = v;
        }
    }
    auto Qrc = [&](const int r, const int c) -> double { return ((c & 3) == (r & 3)) ? ((c >> 2) == 0 ? qc0[r] : qc1[r]) : 0.0; };
    double eta[n];
#pragma unroll
    for (int r = 0; r < n; r++) {
#pragma unroll
        for (int c = 0; c < n; c++) S.z[8 * r + c][lane] = Qrc(r, c);                 // KartLQR.cs:62
        eta[r] = qv[r];                                                              // :63
    }
    singular = 0;
    double alpha[m];
#pragma unroll 1
    for (int t = 3; t >= 0; t--) {                                                   // :64 (HKA:1201 horizon literal 3)
        // ---------------- S1 / S2: the lane's share of [LHS | RHSMat | RHSVec]: rows 4 p + 2 + a (v, heading) of Z_p B_j and Z_p A ----------------
        double loc[NP][2][2];                    // [j][a][cb]: LHS[2 j + a][2 p + cb]
        double rl[2][n], rvl[2];                 // RHSMat[2 p + a][c], RHSVec[2 p + a]
#pragma unroll
        for (int a = 0; a < 2; a++) {
            double zo[n];
#pragma unroll
            for (int c = 0; c < n; c++) zo[c] = S.z[8 * (4 * p + 2 + a) + c][lane];
#pragma unroll
            for (int j = 0; j < NP; j++) {
                const double t10 = fma64(zo[4 * j + 2], dt, 0.0);                    // (Z_p B_j)[row][0]: B_j[2][0] = dt
                const double t11 = fma64(zo[4 * j + 3], dt, 0.0);                    // B_j[3][1] = dt
#pragma unroll
                for (int cb = 0; cb < 2; cb++) {
                    const double s = fma64(dt, cb ? t11 : t10, 0.0);                 // B_p'(Z_p B_j): only B_p[2 + a][a] = dt
                    const double rb = a == cb ? rc : 0.0;
                    loc[j][a][cb] = (p == j) ? (rb + s) : s;                         // :78
                }
                double za[4];
                za[0] = fma64(zo[4 * j + 0], 1.0, 0.0);
                za[1] = fma64(zo[4 * j + 1], 1.0, 0.0);
                { double s = fma64(zo[4 * j + 0], a4[j][0], 0.0); s = fma64(zo[4 * j + 1], a4[j][1], s); za[2] = fma64(zo[4 * j + 2], 1.0, s); }
                { double s = fma64(zo[4 * j + 0], a4[j][2], 0.0); s = fma64(zo[4 * j + 1], a4[j][3], s); za[3] = fma64(zo[4 * j + 3], 1.0, s); }
#pragma unroll
                for (int cc = 0; cc < 4; cc++) rl[a][4 * j + cc] = fma64(dt, za[cc], 0.0);   // B_p'(Z_p A)
            }
            rvl[a] = fma64(dt, p ? eta[6 + a] : eta[2 + a], 0.0);                             // B_p' eta_p
        }
        // both lanes: the whole 4 x 4 LHS, the right-hand sides of the lane's own four columns (c = 4 p + lc) and RHSVec
        double L[m][m], sacc[m][m], bb[m][4], bv[m];
#pragma unroll
        for (int j = 0; j < NP; j++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int cb = 0; cb < 2; cb++) {
                    L[2 * j + a][0 + cb] = pair_get<0>(loc[j][a][cb]);
                    L[2 * j + a][2 + cb] = pair_get<1>(loc[j][a][cb]);
                }
#pragma unroll
        for (int i = 0; i < m; i++)
#pragma unroll
            for (int c = 0; c < m; c++) sacc[i][c] = 0.0;
#pragma unroll
        for (int a = 0; a < 2; a++) {
            bv[0 + a] = pair_get<0>(rvl[a]);
            bv[2 + a] = pair_get<1>(rvl[a]);
#pragma unroll
            for (int lc = 0; lc < 4; lc++) {
                // row 2 j + a of RHSMat is held by lane j; this lane wants its columns 4 p + lc
                // (every permute is executed by BOTH lanes, then each keeps what it wants: a permute inside a conditional would run with
                // the pair's other lane switched off)
                const double lo = rl[a][lc], hi = rl[a][4 + lc];
                const double lo0 = pair_get<0>(lo), hi0 = pair_get<0>(hi), lo1 = pair_get<1>(lo), hi1 = pair_get<1>(hi);
                bb[0 + a][lc] = p ? hi0 : lo0;
                bb[2 + a][lc] = p ? hi1 : lo1;
            }
        }
        // ---------------- S3: LU (JAMA order, every lane the whole 4 x 4) + forward elimination of the lane's right-hand sides ----------------
#pragma unroll
        for (int k = 0; k < m; k++) {
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i >= k) L[i][k] = L[i][k] - sacc[i][k];
            int pv = k;
            double best = fabs(L[k][k]);
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i > k && fabs(L[i][k]) > best) { best = fabs(L[i][k]); pv = i; }
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i > k && i == pv) {
#pragma unroll
                    for (int c = 0; c < m; c++) {
                        double tmp = L[i][c]; L[i][c] = L[k][c]; L[k][c] = tmp;
                        if (c != k) { tmp = sacc[i][c]; sacc[i][c] = sacc[k][c]; sacc[k][c] = tmp; }
                    }
#pragma unroll
                    for (int lc = 0; lc < 4; lc++) { const double tmp = bb[i][lc]; bb[i][lc] = bb[k][lc]; bb[k][lc] = tmp; }
                    { const double tmp = bv[i]; bv[i] = bv[k]; bv[k] = tmp; }
                }
            const double ck = L[k][k];
            if (ck == 0.0) singular = 1;
            double lm[m];
#pragma unroll
            for (int i = 0; i < m; i++) {
                lm[i] = 0.0;
                if (i > k) {
                    if (ck != 0.0) L[i][k] = L[i][k] / ck;
                    lm[i] = L[i][k];
                }
            }
#pragma unroll
            for (int c = 0; c < m; c++)
                if (c > k) {
                    L[k][c] = L[k][c] - sacc[k][c];
#pragma unroll
                    for (int i = 0; i < m; i++)
                        if (i > k) sacc[i][c] += lm[i] * L[k][c];
                }
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i > k) {
#pragma unroll
                    for (int lc = 0; lc < 4; lc++) { const double temp = bb[k][lc] * lm[i]; bb[i][lc] = bb[i][lc] - temp; }
                    { const double tempv = bv[k] * lm[i]; bv[i] = bv[i] - tempv; }
                }
        }
        // back substitution  U X = Y  (k descending)
#pragma unroll
        for (int kk = 0; kk < m; kk++) {
            const int k = m - 1 - kk;
            const double ukk = L[k][k];
#pragma unroll
            for (int lc = 0; lc < 4; lc++) bb[k][lc] = bb[k][lc] / ukk;
            bv[k] = bv[k] / ukk;
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i < k) {
                    const double uik = L[i][k];
#pragma unroll
                    for (int lc = 0; lc < 4; lc++) { const double temp = bb[k][lc] * uik; bb[i][lc] = bb[i][lc] - temp; }
                    { const double tempv = bv[k] * uik; bv[i] = bv[i] - tempv; }
                }
        }
        // ---------------- S4: P (all columns: the other four come from the pair's other lane), alpha ----------------
        double Pfull[m][n];
#pragma unroll
        for (int i = 0; i < m; i++) {
            alpha[i] = bv[i];
#pragma unroll
            for (int lc = 0; lc < 4; lc++) {
                const double mine = bb[i][lc], theirs = pair_get<2>(bb[i][lc]);
                Pfull[i][lc] = p ? theirs : mine;
                Pfull[i][4 + lc] = p ? mine : theirs;
            }
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int c = 0; c < n; c++) S.pp[8 * a + c][lane] = p ? Pfull[2 + a][c] : Pfull[a][c];

        // The value update of the last sweep feeds nothing (u0 reads this sweep's P and alpha): skipped, as in lq1_solve.  (Written as a
        // conditional around the rest of the sweep: a `break` here made the allocator spill 22 - 32 registers in every sweep.)
        if (t > 0) {
        // ---------------- S5: F = A - sum_k B_k P_k; beta = -sum_k B_k alpha_k ----------------
        // position rows (x, z) of block k: F[4 k + 0 / 1][c] = A_k entries in the block's own columns, +0.0 elsewhere; dense rows
        // Fd[2 k + a][c] = F[4 k + 2 + a][c] = [c == 4 k + 2 + a] - (0.0 + dt P[2 k + a][c])
        double Fd[m][n];
#pragma unroll
        for (int k = 0; k < NP; k++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int c = 0; c < n; c++) {
                    const double tt = fma64(dt, Pfull[2 * k + a][c], 0.0);
                    const double av = (c == 4 * k + 2 + a) ? 1.0 : 0.0;
                    Fd[2 * k + a][c] = av - (0.0 + tt);
                }
        // F[4 b + 0][4 b + cs] and F[4 b + 1][4 b + cs] (own block only), exactly as lq_solve_game forms them: av - (0.0 + 0.0)
        auto Fx = [&](const int b, const int cs) -> double { const double av = cs == 0 ? 1.0 : (cs == 2 ? a4[b][0] : (cs == 3 ? a4[b][2] : 0.0)); return av - (0.0 + 0.0); };
        auto Fz = [&](const int b, const int cs) -> double { const double av = cs == 1 ? 1.0 : (cs == 2 ? a4[b][1] : (cs == 3 ? a4[b][3] : 0.0)); return av - (0.0 + 0.0); };
        double beta[n];
#pragma unroll
        for (int k = 0; k < NP; k++) {
            beta[4 * k + 0] = 0.0 - 0.0;
            beta[4 * k + 1] = 0.0 - 0.0;
            beta[4 * k + 2] = 0.0 - fma64(dt, alpha[2 * k + 0], 0.0);
            beta[4 * k + 3] = 0.0 - fma64(dt, alpha[2 * k + 1], 0.0);
        }
        // ---------------- S6 for the lane's own player: W = Z_p F, then Z_p and eta_p (:113-119) ----------------
        double W[n][n];
#pragma unroll
        for (int r = 0; r < n; r++) {
            double zr[n];
#pragma unroll
            for (int k = 0; k < n; k++) zr[k] = S.z[8 * r + k][lane];
#pragma unroll
            for (int c = 0; c < n; c++) {
                const int b = c >> 2, cs = c & 3;
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < n; k++) {
                    if ((k & 3) < 2 && (k >> 2) != b) continue;                      // structural +0.0 (BICYCLE)
                    const double f = (k & 3) == 0 ? Fx(b, cs) : ((k & 3) == 1 ? Fz(b, cs) : Fd[2 * (k >> 2) + (k & 3) - 2][c]);
                    s = fma64(zr[k], f, s);
                }
                W[r][c] = s;
            }
        }
        // Z_p <- (Q_p + P_p'(R_p P_p)) + F'(Z_p F), row by row, straight into the lane's LDS column (every W is in registers)
        double vec[n];
#pragma unroll
        for (int r = 0; r < n; r++) {
            const int br = r >> 2, rs = r & 3;
            const double p0r = S.pp[8 * 0 + r][lane], p1r = S.pp[8 * 1 + r][lane];    // P[2 p + a][r]
            double zn[n];
#pragma unroll
            for (int c = 0; c < n; c++) {
                double o = 0.0;
#pragma unroll
                for (int k = 0; k < n; k++) {
                    if ((k & 3) < 2 && (k >> 2) != br) continue;                     // F[k][r] is a structural +0.0
                    const double f = (k & 3) == 0 ? Fx(br, rs) : ((k & 3) == 1 ? Fz(br, rs) : Fd[2 * (k >> 2) + (k & 3) - 2][r]);
                    o = fma64(f, W[k][c], o);
                }
                const double rp0 = fma64(rc, S.pp[8 * 0 + c][lane], 0.0);            // (R_p P_p)[a][c]
                const double rp1 = fma64(rc, S.pp[8 * 1 + c][lane], 0.0);
                double t2 = fma64(p0r, rp0, 0.0);
                t2 = fma64(p1r, rp1, t2);
                zn[c] = (Qrc(r, c) + t2) + o;
                S.z[8 * r + c][lane] = zn[c];
            }
            // eta_p <- (q_p + P_p'(R_p alpha_p)) + F'(eta_p + Z_p beta)    with the NEW Z_p (Q2)
            double zb = 0.0;
#pragma unroll
            for (int k = 0; k < n; k++) zb = fma64(zn[k], beta[k], zb);
            vec[r] = eta[r] + zb;
        }
        const double ra0 = fma64(rc, p ? alpha[2] : alpha[0], 0.0);
        const double ra1 = fma64(rc, p ? alpha[3] : alpha[1], 0.0);
#pragma unroll
        for (int r = 0; r < n; r++) {
            const int br = r >> 2, rs = r & 3;
            double v3 = 0.0;
#pragma unroll
            for (int k = 0; k < n; k++) {
                if ((k & 3) < 2 && (k >> 2) != br) continue;
                const double f = (k & 3) == 0 ? Fx(br, rs) : ((k & 3) == 1 ? Fz(br, rs) : Fd[2 * (k >> 2) + (k & 3) - 2][r]);
                v3 = fma64(f, vec[k], v3);
            }
            double v2 = fma64(S.pp[8 * 0 + r][lane], ra0, 0.0);
            v2 = fma64(S.pp[8 * 1 + r][lane], ra1, v2);
            eta[r] = (qv[r] + v2) + v3;
        }
        }       // t > 0
    }
    // :121-126 u0 = -P_0 x0 - alpha_0: rows 0, 1 of the last sweep's P are lane 0's own rows (S.pp); lane 1's value is not used
#pragma unroll
    for (int a = 0; a < 2; a++) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < n; c++) s = fma64(-S.pp[8 * a + c][lane], games.get(game, c >> 2, GP_X0 + (c & 3)), s);
        u0[a] = s - alpha[a];
    }
}

#ifndef HK_LQ2_HOST_CHECK
// The 2-player queue of a round, 32 games per wave (one workgroup = one wave), grid-stride over blocks [0, nblocks)
__device__ __forceinline__ void lq2_pair_body(const int block, const int nblocks, const EnvParams& P, const HotRef hr, const GameSoA& games,
                                              const int* queue_cnt, const int* queue, hk_lq_debug* dbg_out, int* status, Lq2PairLds& S,
                                              unsigned long long* gstats)
{
    const int lane = threadIdx.x & 63, p = lane & 1, gs = lane >> 1;
    const int count = queue_cnt[2];
    if (block == 0 && threadIdx.x == 0 && count > 0) atomicAdd(&gstats[2], (unsigned long long)count);   // hk_prof_games
    for (int base = block * 32; base < count; base += nblocks * 32) {
        const int slot = base + gs;
        const bool live = slot < count;
        const int game = queue[live ? slot : count - 1];      // idle pairs recompute the last game and discard it (DPP needs full quads)
        double u0[2];
        int singular = 0;
        lq2_pair_solve(game, p, lane, (double)P.dt, games, S, u0, singular);
        if (live && p == 0) {
            if (singular) atomicOr(status, 1);
            decode_store(P, hr, game, u0[0], u0[1], (dbg_out && (P.debug & 1)) ? &dbg_out[game] : nullptr);
        }
    }
}

// ONE launch per round for every queued game of up to 4 players: blocks [0, n34) take the 3- and 4-player queues (the generic core,
// n34 / sizes blocks per size; first in the grid: they are the long ones), blocks [n34, n34 + n2) the 2-player queue (pairs of
// lanes).  Both bodies run one wave per SIMD (458 registers / 40 KB of LDS), so they share a kernel at no cost — and in a spread
// field, where a round holds a handful of games of each kind, their latencies overlap instead of adding up.
// SMALL: the form that runs BESIDE a planner's search launch (round 5).  A search wave holds 232 of a SIMD's 512 registers on every CU for ~7 ms; the
// one-game-per-wave matrix-core solver needs a SIMD's whole register file and would wait for the search to end.  This instantiation solves the 3- / 4-player
// queues on the lane-per-row core whatever their length (the same arithmetic by contract: hk_lq_core.h) and is held to two waves per SIMD (amdgpu_waves_per_eu 2).
constexpr size_t lqn_round_small_lds()
{
    constexpr size_t B2 = (sizeof(LqGameLds<2>) + sizeof(CostRows<2>)) * LqDims<2>::SLOTS, B3 = (sizeof(LqGameLds<3>) + sizeof(CostRows<3>)) * LqDims<3>::SLOTS,
                     B4 = (sizeof(LqGameLds<4>) + sizeof(CostRows<4>)) * LqDims<4>::SLOTS;
    return (B2 > B3 ? (B2 > B4 ? B2 : B4) : (B3 > B4 ? B3 : B4));
}
template <bool SMALL>
__device__ __forceinline__ void lqn_round_body(const EnvParams& P, const HotRef hr, const GameSoA games, const int* queue_cnt,
                                               const int* queue, hk_lq_debug* dbg_out, int* status, int n34, int sizes, int n2,
                                               unsigned long long* gstats, int bulk34)
{
    constexpr size_t B3a = sizeof(LqMfmaLds<3>) + sizeof(CostRows<3>), B3b = (sizeof(LqGameLds<3>) + sizeof(CostRows<3>)) * LqDims<3>::SLOTS;
    constexpr size_t B4a = sizeof(LqMfmaLds<4>) + sizeof(CostRows<4>), B4b = (sizeof(LqGameLds<4>) + sizeof(CostRows<4>)) * LqDims<4>::SLOTS;
    constexpr size_t B3 = B3a > B3b ? B3a : B3b, B4 = B4a > B4b ? B4a : B4b;
    constexpr size_t B34 = B3 > B4 ? B3 : B4;
    constexpr size_t B2 = (sizeof(LqGameLds<2>) + sizeof(CostRows<2>)) * LqDims<2>::SLOTS;
    constexpr size_t BP = SMALL ? B2 : sizeof(Lq2PairLds);
    constexpr size_t B34s = SMALL ? (B3b > B4b ? B3b : B4b) : B34;
    constexpr size_t BMAX = B34s > BP ? B34s : BP;
    // (SMALL: dynamic LDS of lqn_round_small_lds() bytes — with the size in the kernel's own metadata the back end derives one wave per SIMD from it and hands
    // the kernel the whole register file whatever amdgpu_waves_per_eu says)
    __shared__ __align__(16) unsigned char smem_static[SMALL ? 16 : BMAX];
    HK_DYN_SHARED(smem_dyn);
    unsigned char* smem = SMALL ? smem_dyn : smem_static;
    const int b = blockIdx.x;
    if (b >= n34) {
        // (SMALL: the pair solver keeps a game in ~450 registers; beside a search wave the 2-player queue goes to the lane-per-row core too, 8 games a wave)
        if constexpr (SMALL) lqn_body<2, false>(b - n34, n2, P, hr, games, queue_cnt, queue, dbg_out, status, smem, gstats);
        else lq2_pair_body(b - n34, n2, P, hr, games, queue_cnt, queue, dbg_out, status, *reinterpret_cast<Lq2PairLds*>(smem), gstats);
        return;
    }
    const int per = n34 / sizes, which = b / per, bb = b - which * per;
    // Which solver?  The queue length decides, on the device: a round that holds a handful of games (a spread field) wants the shortest
    // latency — one game per wave; one that holds thousands (the race start, packs) wants 64 / n games per instruction of the m x m solve.
    const int cnt = queue_cnt[which == 0 ? 3 : 4];
    if (SMALL || cnt > bulk34) {
        if (which == 0) lqn_body<3, true>(bb, per, P, hr, games, queue_cnt, queue, dbg_out, status, smem, gstats);
        else lqn_body<4, true>(bb, per, P, hr, games, queue_cnt, queue, dbg_out, status, smem, gstats);
    } else if constexpr (!SMALL) {
        if (which == 0) lqn_body_mfma<3>(bb, per, P, hr, games, queue_cnt, queue, dbg_out, status, smem, gstats);
        else lqn_body_mfma<4>(bb, per, P, hr, games, queue_cnt, queue, dbg_out, status, smem, gstats);
    }
}
__global__ __launch_bounds__(64) void lqn_round_kernel(EnvParams P, const HotRef hr, const GameSoA games, const int* queue_cnt, const int* queue, hk_lq_debug* dbg_out,
                                                       int* status, int n34, int sizes, int n2, unsigned long long* gstats, int bulk34)
{
    lqn_round_body<false>(P, hr, games, queue_cnt, queue, dbg_out, status, n34, sizes, n2, gstats, bulk34);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void lqn_round_small_kernel(EnvParams P, const HotRef hr, const GameSoA games, const int* queue_cnt,
                                                                                                   const int* queue, hk_lq_debug* dbg_out, int* status, int n34, int sizes, int n2,
                                                                                                   unsigned long long* gstats, int bulk34)
{
    lqn_round_body<true>(P, hr, games, queue_cnt, queue, dbg_out, status, n34, sizes, n2, gstats, bulk34);
}
#endif
