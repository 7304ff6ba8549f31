// hk_lq2_quad.h — a TWO-player feedback LQ Nash game (KartLQR.solveFeedbackLQR with N = 2: n = 8 states, m = 4 controls) solved
// by the four lanes of a quad in registers: no LDS, no barriers, every exchange a DPP quad broadcast.
//
// Why: two karts within 8 m of each other are by far the most common multi-player game (the start grid's row mates, every
// overtake), and the generic core (hk_lq_core.h: a lane per row, everything that crosses lanes staged through LDS between
// barriers) needs ~60 us of latency for one of them.  Here lane q of the quad owns rows 2q, 2q + 1 of both players' value
// matrices — lane 0: (x, z) of player 0, lane 1: (v, heading) of player 0, lane 2 / 3: the same for player 1 — and COLUMNS 2q,
// 2q + 1 of [RHSMat], P and F.  The 4 x 4 system matrix is small enough that every lane factorises it redundantly.
//
// ARITHMETIC: every value is produced by exactly the operations, in exactly the order, of lq_solve_game<2, QCompact<2>, true> on
// the game lqn_stage_solve would stage (k-ascending fma chains seeded with +0.0; JAMA-order LU with plain mul / add; the
// reference's quirks Q1, Q2, Q4).  Terms whose factor is a structural zero of the linearised bicycle (B has only the two dt
// entries, A is the identity plus four entries) are left out — fma(z, +-0.0, s) = s — and terms whose factor is 1.0 are kept
// as fma(z, 1.0, s), exactly as lq1_solve does for the single-player game.  The parity tests compare the decoded controls and
// every kart field with the CPU oracle bit for bit.
// (included by hk_env_solve.h inside namespace hk::HK_GA_NS; quads only)
#pragma once

template <int J> __device__ __forceinline__ double qb(const double v)
{   // value of lane J of the calling lane's quad (every lane of the quad must be active)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, J * 0x55, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, J * 0x55, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// `game`: index into the GameSoA buffer (quad-uniform); q = lane & 3; returns player 0's control at t = 0 in every lane
__device__ HK_INWAVE_CALL void lq2_quad_solve(const int game, const int q, const double dt, const GameSoA games, double u0[2], int& singular)
{
    constexpr int NP = 2, n = 8, m = 4;
    const int ib = q >> 1;                       // player block of this lane's rows
    // ---- inputs (as lqn_stage_solve stages them).  The game's constants — A entries, control weights, the compact cost rows of the
    // lane's two rows — are NOT kept in registers across a sweep: they are re-read from the GameSoA buffer (L2) where a sweep
    // uses them.  Held for the whole recursion they pushed the kernel 170 VGPRs past the 256 of two waves per SIMD, and the
    // spill traffic made a solve 10x slower than its arithmetic.  `opaque` keeps the compiler from hoisting the loads back out.
    auto opaque = [](int g) { asm volatile("" : "+v"(g)); return g; };
    auto load_a4 = [&](const int g, const int j, double out[4]) {
#pragma unroll
        for (int e = 0; e < 4; e++) out[e] = games.get(g, j, GP_A4 + e);
    };
    // compact cost rows of player i for the rows r = 2q + lr: qc[b'][lr] = Q_i[r][4 b' + (r & 3)], qv[lr] = q_i[r]
    auto load_cost = [&](const int g, const int i, double qc[NP][2], double qv[2]) {
#pragma unroll
        for (int lr = 0; lr < 2; lr++) {
            const int r = 2 * q + lr;
            const int b = r >> 2, sidx = r & 3;
            double c0 = 0.0, c1 = 0.0, v = 0.0;
            const int M = (int)games.get(g, i, GP_M);
            if (b == 0) {
                double d = 0.0;
                if (sidx < 2) {
                    double total = 0.0;                                    // KartLQRCosts.cs:67-79
                    for (int j = 0; j < M; j++) total -= games.get(g, i, GP_AW + j);
                    d = total;
                }
                d += games.get(g, i, GP_TW + sidx);                        // :81-84
                c0 = d;
                if (sidx < 2 && M > 0) c1 = games.get(g, i, GP_AW + 0);
                const double t = -games.get(g, i, GP_TGT + sidx);          // getQVec :109-113
                v = t * games.get(g, i, GP_TW + sidx);
            } else {
                if (sidx < 2) c0 = games.get(g, i, GP_AW + 0);             // :74
                double dg = 0.0;
                if (sidx < 3) dg = -games.get(g, i, GP_OPW + sidx);        // :91 assignment (Q4)
                c1 = dg;
                if (sidx < 3) { v = games.get(g, i, GP_OPT + sidx); v = v * -games.get(g, i, GP_OPW + sidx); }   // :117, :121
            }
            qc[0][lr] = c0; qc[1][lr] = c1; qv[lr] = v;
        }
    };
    // Q_i(r, c) = ((c & 3) == (r & 3)) ? qc[c >> 2] : 0
    auto Qrc = [&](const double qc[NP][2], const int lr, const int c) -> double {
        const int sidx = (2 * q + lr) & 3;
        return ((c & 3) == sidx) ? qc[c >> 2][lr] : 0.0;
    };
    double Z[NP][2][n], eta[NP][2];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        double qc[NP][2], qv[2];
        load_cost(game, i, qc, qv);
#pragma unroll
        for (int lr = 0; lr < 2; lr++) {
#pragma unroll
            for (int c = 0; c < n; c++) Z[i][lr][c] = Qrc(qc, lr, c);                 // KartLQR.cs:62
            eta[i][lr] = qv[lr];                                                     // :63
        }
    }
    singular = 0;
    double P[m][2], alpha[m];                    // P[i][lc]: column 2q + lc of P; alpha in every lane
#pragma unroll 1
    for (int t = 3; t >= 0; t--) {                                                   // :64 (HKA:1201 horizon literal 3)
        // ---------------- S1: this lane's rows of Z_ib B_j (T1) and Z_ib A ----------------
        double T1[2][NP][2], ZA[2][n];           // [lr][j][b], [lr][c]
        const int g1 = opaque(game);
        double a4[NP][4];
        load_a4(g1, 0, a4[0]); load_a4(g1, 1, a4[1]);
        const double rcb = games.get(g1, ib, GP_RC);     // control weight of this lane's player block
#pragma unroll
        for (int lr = 0; lr < 2; lr++) {
            double zo[n];
#pragma unroll
            for (int c = 0; c < n; c++) zo[c] = ib ? Z[1][lr][c] : Z[0][lr][c];
#pragma unroll
            for (int j = 0; j < NP; j++) {
                T1[lr][j][0] = fma64(zo[4 * j + 2], dt, 0.0);                        // B_j[2][0] = dt
                T1[lr][j][1] = fma64(zo[4 * j + 3], dt, 0.0);                        // B_j[3][1] = dt
                ZA[lr][4 * j + 0] = fma64(zo[4 * j + 0], 1.0, 0.0);
                ZA[lr][4 * j + 1] = fma64(zo[4 * j + 1], 1.0, 0.0);
                { double s = fma64(zo[4 * j + 0], a4[j][0], 0.0); s = fma64(zo[4 * j + 1], a4[j][1], s); ZA[lr][4 * j + 2] = fma64(zo[4 * j + 2], 1.0, s); }
                { double s = fma64(zo[4 * j + 0], a4[j][2], 0.0); s = fma64(zo[4 * j + 1], a4[j][3], s); ZA[lr][4 * j + 3] = fma64(zo[4 * j + 3], 1.0, s); }
            }
        }
        // ---------------- S2: [LHS | RHSMat | RHSVec].  Everything comes from the (v, heading) lanes 1 and 3 ----------------
        // LHS[2j + a][2ci + cb] = B_ci'(Z_ci B_j) (+ R_ci): lane 2ci + 1 holds T1[a][j][cb] of player ci's rows
        double L[m][m], sacc[m][m];
        {
            double loc[NP][2][2];                // [j][a][cb] of THIS lane (meaningful in lanes 1 and 3)
#pragma unroll
            for (int j = 0; j < NP; j++)
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int cb = 0; cb < 2; cb++) {
                        const double s = fma64(dt, T1[a][j][cb], 0.0);
                        const double rb = a == cb ? rcb : 0.0;
                        loc[j][a][cb] = (ib == j) ? (rb + s) : s;                    // :78
                    }
#pragma unroll
            for (int j = 0; j < NP; j++)
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int cb = 0; cb < 2; cb++) {
                        L[2 * j + a][0 + cb] = qb<1>(loc[j][a][cb]);
                        L[2 * j + a][2 + cb] = qb<3>(loc[j][a][cb]);
                    }
#pragma unroll
            for (int i = 0; i < m; i++)
#pragma unroll
                for (int c = 0; c < m; c++) sacc[i][c] = 0.0;
        }
        // RHSMat[2j + a][c] = dt * (Z_j A)[4j + 2 + a][c] for the lane's columns c = 2q, 2q + 1; RHSVec[2j + a] = dt * eta_j[4j + 2 + a]
        double bb[m][2], bv[m];
        {
            double rl[2][n], rvl[2];             // this lane's rows, scaled (meaningful in lanes 1 and 3)
#pragma unroll
            for (int a = 0; a < 2; a++) {
#pragma unroll
                for (int c = 0; c < n; c++) rl[a][c] = fma64(dt, ZA[a][c], 0.0);
                rvl[a] = fma64(dt, ib ? eta[1][a] : eta[0][a], 0.0);
            }
#pragma unroll
            for (int a = 0; a < 2; a++) {
                bv[0 + a] = qb<1>(rvl[a]);
                bv[2 + a] = qb<3>(rvl[a]);
#pragma unroll
                for (int lc = 0; lc < 2; lc++) {
                    // destination lane d takes columns 2d + lc: four rounds, each lane keeps its own
                    double x1 = 0.0, x3 = 0.0;
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const double y1 = qb<1>(rl[a][2 * d + lc]), y3 = qb<3>(rl[a][2 * d + lc]);
                        x1 = q == d ? y1 : x1;
                        x3 = q == d ? y3 : x3;
                    }
                    bb[0 + a][lc] = x1;
                    bb[2 + a][lc] = x3;
                }
            }
        }
        // ---------------- S3: LU (JAMA order, every lane the whole 4 x 4) + forward elimination of the lane's right-hand sides ----------------
#pragma unroll
        for (int k = 0; k < m; k++) {
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i >= k) L[i][k] = L[i][k] - sacc[i][k];
            int p = k;
            double best = fabs(L[k][k]);
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i > k && fabs(L[i][k]) > best) { best = fabs(L[i][k]); p = i; }
            // row swap k <-> p: the whole row of L, the accumulators of the other columns, the right-hand sides
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i > k && i == p) {
#pragma unroll
                    for (int c = 0; c < m; c++) {
                        double tmp = L[i][c]; L[i][c] = L[k][c]; L[k][c] = tmp;
                        if (c != k) { tmp = sacc[i][c]; sacc[i][c] = sacc[k][c]; sacc[k][c] = tmp; }
                    }
#pragma unroll
                    for (int lc = 0; lc < 2; lc++) { const double tmp = bb[i][lc]; bb[i][lc] = bb[k][lc]; bb[k][lc] = tmp; }
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
                    for (int lc = 0; lc < 2; lc++) { const double temp = bb[k][lc] * lm[i]; bb[i][lc] = bb[i][lc] - temp; }
                    { const double tempv = bv[k] * lm[i]; bv[i] = bv[i] - tempv; }
                }
        }
        // back substitution  U X = Y  (k descending)
#pragma unroll
        for (int kk = 0; kk < m; kk++) {
            const int k = m - 1 - kk;
            const double ukk = L[k][k];
#pragma unroll
            for (int lc = 0; lc < 2; lc++) bb[k][lc] = bb[k][lc] / ukk;
            bv[k] = bv[k] / ukk;
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i < k) {
                    const double uik = L[i][k];
#pragma unroll
                    for (int lc = 0; lc < 2; lc++) { const double temp = bb[k][lc] * uik; bb[i][lc] = bb[i][lc] - temp; }
                    { const double tempv = bv[k] * uik; bv[i] = bv[i] - tempv; }
                }
        }
        // ---------------- S4: P (the lane's columns), alpha ----------------
#pragma unroll
        for (int i = 0; i < m; i++) { P[i][0] = bb[i][0]; P[i][1] = bb[i][1]; alpha[i] = bv[i]; }
        // ---------------- S5: F = A - sum_k B_k P_k (columns 2q, 2q + 1), beta ----------------
        double Fc[2][n];                         // Fc[lc][row] = F[row][2q + lc]
        const int g5 = opaque(game);
        double a4b[4];                           // A entries of the player block of this lane's columns
        load_a4(g5, ib, a4b);
#pragma unroll
        for (int lc = 0; lc < 2; lc++) {
            const int cs = (2 * q + lc) & 3;     // state index of the column inside its player block (block = ib)
#pragma unroll
            for (int k = 0; k < NP; k++) {
                // rows x, z: B rows are zero -> acc = 0.0 + 0.0; f = av - 0.0
                const double ax = (ib == k) ? (cs == 0 ? 1.0 : (cs == 2 ? a4b[0] : (cs == 3 ? a4b[2] : 0.0))) : 0.0;
                const double az = (ib == k) ? (cs == 1 ? 1.0 : (cs == 2 ? a4b[1] : (cs == 3 ? a4b[3] : 0.0))) : 0.0;
                Fc[lc][4 * k + 0] = ax - (0.0 + 0.0);
                Fc[lc][4 * k + 1] = az - (0.0 + 0.0);
                { const double tt = fma64(dt, P[2 * k + 0][lc], 0.0); const double av = (ib == k && cs == 2) ? 1.0 : 0.0; Fc[lc][4 * k + 2] = av - (0.0 + tt); }
                { const double tt = fma64(dt, P[2 * k + 1][lc], 0.0); const double av = (ib == k && cs == 3) ? 1.0 : 0.0; Fc[lc][4 * k + 3] = av - (0.0 + tt); }
            }
        }
        double beta[n];
#pragma unroll
        for (int k = 0; k < NP; k++) {
            beta[4 * k + 0] = 0.0 - 0.0;
            beta[4 * k + 1] = 0.0 - 0.0;
            beta[4 * k + 2] = 0.0 - fma64(dt, alpha[2 * k + 0], 0.0);
            beta[4 * k + 3] = 0.0 - fma64(dt, alpha[2 * k + 1], 0.0);
        }
        // ---------------- S6: per player Z_i, eta_i update (:113-119) ----------------
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int g6 = opaque(game);
            double qc[NP][2], qv[2];
            load_cost(g6, i, qc, qv);
            const double rci = games.get(g6, i, GP_RC);
            // W = Z_i F, this lane's two rows (position rows of the OTHER player's block of F are structural zeros: BICYCLE)
            double W[2][n];
#pragma unroll
            for (int c = 0; c < n; c++) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int k = 0; k < n; k++) {
                    if ((k & 3) < 2 && (k >> 2) != (c >> 2)) continue;
                    const double f = (c >> 1) == 0 ? qb<0>(Fc[c & 1][k]) : ((c >> 1) == 1 ? qb<1>(Fc[c & 1][k]) : ((c >> 1) == 2 ? qb<2>(Fc[c & 1][k]) : qb<3>(Fc[c & 1][k])));
                    s0 = fma64(Z[i][0][k], f, s0);
                    s1 = fma64(Z[i][1][k], f, s1);
                }
                W[0][c] = s0; W[1][c] = s1;
            }
            // R_i P_i for the lane's columns
            double RPl[2][2];                    // [a][lc]
#pragma unroll
            for (int lc = 0; lc < 2; lc++) {
                RPl[0][lc] = fma64(rci, P[2 * i + 0][lc], 0.0);
                RPl[1][lc] = fma64(rci, P[2 * i + 1][lc], 0.0);
            }
            // Z_i <- (Q_i + P_i'(R_i P_i)) + F'(Z_i F), this lane's rows r = 2q + lr (F[k][r] = Fc[lr][k], P[.][r] = P[.][lr])
#pragma unroll
            for (int c = 0; c < n; c++) {
                const double rp0 = (c >> 1) == 0 ? qb<0>(RPl[0][c & 1]) : ((c >> 1) == 1 ? qb<1>(RPl[0][c & 1]) : ((c >> 1) == 2 ? qb<2>(RPl[0][c & 1]) : qb<3>(RPl[0][c & 1])));
                const double rp1 = (c >> 1) == 0 ? qb<0>(RPl[1][c & 1]) : ((c >> 1) == 1 ? qb<1>(RPl[1][c & 1]) : ((c >> 1) == 2 ? qb<2>(RPl[1][c & 1]) : qb<3>(RPl[1][c & 1])));
                double o0 = 0.0, o1 = 0.0;
#pragma unroll
                for (int k = 0; k < n; k++) {
                    const double w = (k >> 1) == 0 ? qb<0>(W[k & 1][c]) : ((k >> 1) == 1 ? qb<1>(W[k & 1][c]) : ((k >> 1) == 2 ? qb<2>(W[k & 1][c]) : qb<3>(W[k & 1][c])));
                    o0 = fma64(Fc[0][k], w, o0);
                    o1 = fma64(Fc[1][k], w, o1);
                }
                double t20 = fma64(P[2 * i + 0][0], rp0, 0.0); t20 = fma64(P[2 * i + 1][0], rp1, t20);
                double t21 = fma64(P[2 * i + 0][1], rp0, 0.0); t21 = fma64(P[2 * i + 1][1], rp1, t21);
                Z[i][0][c] = (Qrc(qc, 0, c) + t20) + o0;
                Z[i][1][c] = (Qrc(qc, 1, c) + t21) + o1;
            }
            // eta_i <- (q_i + P_i'(R_i alpha_i)) + F'(eta_i + Z_i beta)    with the NEW Z_i (Q2)
            double vecl[2];
#pragma unroll
            for (int lr = 0; lr < 2; lr++) {
                double zb = 0.0;
#pragma unroll
                for (int k = 0; k < n; k++) zb = fma64(Z[i][lr][k], beta[k], zb);
                vecl[lr] = eta[i][lr] + zb;
            }
            double v30 = 0.0, v31 = 0.0;
#pragma unroll
            for (int k = 0; k < n; k++) {
                const double vk = (k >> 1) == 0 ? qb<0>(vecl[k & 1]) : ((k >> 1) == 1 ? qb<1>(vecl[k & 1]) : ((k >> 1) == 2 ? qb<2>(vecl[k & 1]) : qb<3>(vecl[k & 1])));
                v30 = fma64(Fc[0][k], vk, v30);
                v31 = fma64(Fc[1][k], vk, v31);
            }
            const double ra0 = fma64(rci, alpha[2 * i + 0], 0.0);
            const double ra1 = fma64(rci, alpha[2 * i + 1], 0.0);
#pragma unroll
            for (int lr = 0; lr < 2; lr++) {
                double v2 = fma64(P[2 * i + 0][lr], ra0, 0.0);
                v2 = fma64(P[2 * i + 1][lr], ra1, v2);
                eta[i][lr] = (qv[lr] + v2) + (lr ? v31 : v30);
            }
        }
    }
    // :121-126 u0 = -P_0 x0 - alpha_0
#pragma unroll
    for (int a = 0; a < 2; a++) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < n; c++) {
            const double pac = (c >> 1) == 0 ? qb<0>(P[a][c & 1]) : ((c >> 1) == 1 ? qb<1>(P[a][c & 1]) : ((c >> 1) == 2 ? qb<2>(P[a][c & 1]) : qb<3>(P[a][c & 1])));
            s = fma64(-pac, games.get(game, c >> 2, GP_X0 + (c & 3)), s);
        }
        u0[a] = s - alpha[a];
    }
}
