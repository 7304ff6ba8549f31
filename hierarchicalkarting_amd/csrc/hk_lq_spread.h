// hk_lq_spread.h — the feedback LQ Nash game of 2, 3 and 4 players (KartLQR.solveFeedbackLQR, reference AI/LQR/KartLQR.cs:17-128) SPREAD over
// the lanes of a wave: lane (i, r) owns ROW r of player i's value matrix Z_i — NP * 4 NP lanes per game (16 / 36 / 64), four 2-player games or
// one 3- / 4-player game per wave.
//
// Why (round 6).  Once the field has spread a solve tick holds a few dozen multi-player games, each a short overtake, and what they cost is
// not arithmetic but LATENCY and RESIDENCY: the pair solver (hk_lq2_pair.h) runs a whole game in two lanes — ~6 000 dependent vector
// instructions — and keeps W = Z_p F, the dense rows of F and a 4 x 4 LU in ~450 registers plus 40 KB of LDS per wave, so its waves need a SIMD
// to themselves and start only where a CU has drained of the tick kernel's three waves per SIMD (profiles/r05_a_backend_flags.txt, sections 5 - 9).
// Here a lane keeps one row of one Z_i (n doubles), one row of W and one column of F: every value update is n-wide instead of n x n-wide, the
// solve is ~4 x shorter in instructions per lane, and a wave needs ~1/3 of the registers and 2 - 11 KB of LDS — it fits beside resident tick
// waves, and inside the B1 kernel behind phase_assemble (hk_env_run.h), where the registers of the assembly are dead.
//
// Per sweep (t = horizon .. 0):
//   S1  the 2 NP "control lanes" (i, 4 i + 2 + a) hold the rows of Z_i that B_i' picks: they write their share of [LHS | RHSMat | RHSVec] to LDS;
//   S2-S3  lane (., c) takes column c of [LHS | RHSMat] and every lane RHSVec; the m x m LU with partial pivoting in JAMA order runs once per
//       player group (redundantly: the groups publish identical pivots and multipliers), the step's pivot row and multipliers go through LDS;
//   S4-S5  lane (., c) now holds column c of P and every lane alpha: the dense rows of F = A - sum B_k P_k go to LDS (one copy), column c of F
//       stays in the lane;
//   S6  W = Z_i F, row r (F's dense rows as LDS broadcasts, its position rows are constants of the game) -> LDS;
//   S7  Z_i <- (Q_i + P_i'(R_i P_i)) + F' W, row r: the lane's own column of F against W_i's rows (LDS broadcasts within the player group);
//   S8  eta_i likewise, through a vector in LDS.
// No workgroup barrier, no DPP, no readlane: lanes of ONE wave talk through LDS, which serves a wave's requests in order (X::sync orders the
// compiler).  Works on any subset of a wave's lanes, several games side by side.
//
// ARITHMETIC: the contract of hk_lq_core.h, unchanged — every value is produced by exactly the operations, in exactly the order, of
// lq_solve_game<NP, QCompact<NP>, true> on the game lqn_stage_inputs would stage (k-ascending fma chains seeded with +0.0; JAMA-order LU with
// plain mul / add; quirks Q1, Q2, Q4), on another lane.  As in hk_lq2_pair.h, terms whose factor is a structural +0.0 of the linearised
// bicycle are left out (fma(z, +0.0, s) = s for finite z and s never -0.0: s starts at +0.0 and x + y is -0.0 only if both are), factors of
// 1.0 are kept as fma(z, 1.0, s).  Where a lane's own position rows of F enter a chain at a lane-dependent place, every candidate place is
// evaluated with the factor switched to +0.0 elsewhere — the same exact no-op.  tests/test_lq_spread_host.py runs this header on the host
// (64 threads = the lanes) against the C oracle; on the GPU every env parity test holds bit for bit.
// (included by hk_env_solve.h inside namespace hk::HK_GA_NS; the host check defines the few names it needs itself)
#pragma once

// The chains of one column block (S6) / column pair (S7) read 2 - 3 dozen LDS words; left alone the scheduler hoists the loads of ALL blocks to the top
// of the phase (186 / 237 / 269 registers for 2 / 3 / 4 players).  A scheduling fence between blocks keeps a lane's live set at one block's operands.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HK_LQS_NO_FENCE)
#define HK_LQS_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define HK_LQS_FENCE() ((void)0)
#endif

// SW ("serial W"): ONE W block that the players use in turn (S6 - S7 run player by player) — a 4-player game in 5.3 KB instead of 11.2, for the wave of
// env_b1_kernel that solves inside its own 5 632-byte slice of the kart staging area (lqs_inwave)
template <int NP, bool SW = false>
struct __attribute__((aligned(16))) LqSpreadLds {
    static constexpr int n = 4 * NP, m = 2 * NP, CW = m + n + 1;
    static constexpr int WS = n * n + 2;          // doubles per player's W block (+2: the NP blocks a wave reads side by side start 4 banks apart)
    static constexpr int SCR = (m * CW + m * m + m + 2 + 1) & ~1;      // the solve phase's scratch
    // value-update phase: the W rows of every player.  Solve phase (aliased — nothing of W is live then): rows of [LHS | RHSMat | RHSVec], U,
    // and the pivot step's broadcast ([0] pivot row, [1 .. m - 1] multipliers, [m] singular flag)
    double wbuf[SW ? (WS > SCR ? WS : SCR) : NP * WS];
    double Fd[m][n];            // dense rows of F: Fd[2 k + a][c] = F[4 k + 2 + a][c]
    double Pm[m][n];            // P rows
    double vec[NP][n];          // eta_i + Z_i beta
    double al[m];               // alpha
    double beta[m];             // dense entries of beta: beta[2 k + a] = beta_full[4 k + 2 + a]
    double a4[NP][4];           // A[x,v], A[z,v], A[x,h], A[z,h] of every player (KartLQRDynamics.cs:45-48)
    double rc[NP + (NP & 1)];   // control weights (an even count: the struct stays a multiple of 16 bytes)
    __device__ __forceinline__ double* W(int i) { return SW ? wbuf : wbuf + i * WS; }
    __device__ __forceinline__ double* Cm() { return wbuf; }                        // [m][CW]
    __device__ __forceinline__ double* U() { return wbuf + m * CW; }                // [m][m]
    __device__ __forceinline__ double* lu() { return wbuf + m * CW + m * m; }       // [m + 2]
};
static_assert(LqSpreadLds<2>::SCR <= 2 * LqSpreadLds<2>::WS && LqSpreadLds<3>::SCR <= 3 * LqSpreadLds<3>::WS && LqSpreadLds<4>::SCR <= 4 * LqSpreadLds<4>::WS, "solve-phase scratch fits the W area");

template <int NP> struct LqSpreadDims { static constexpr int n = 4 * NP, G = NP * 4 * NP, GPW = 64 / G; };

// the compact cost row of lane (i, r) (KartLQRCosts.cs:57-127; the expressions of lqn_stage_inputs): Q_i(r, c) = ((c & 3) == (r & 3)) ? qc[c >> 2] : 0, q_i(r) = qv
template <int NP>
__device__ __forceinline__ void lqs_cost_row(const GameSoA& games, const int game, const int i, const int r, double qc[NP], double& qv)
{
    const int b = r >> 2, sidx = r & 3;
#pragma unroll
    for (int q = 0; q < NP; q++) qc[q] = 0.0;
    qv = 0.0;
    const int M = (int)games.get(game, i, GP_M);
    if (b == 0) {
        double d = 0.0;
        if (sidx < 2) {
            double total = 0.0;                                    // :67-79
#pragma unroll
            for (int j = 0; j < NP - 1; j++) if (j < M) total -= games.get(game, i, GP_AW + j);
            d = total;
        }
        const double tw = games.get(game, i, GP_TW + sidx);
        d += tw;                                                   // :81-84
        qc[0] = d;
        if (sidx < 2) {
#pragma unroll
            for (int q = 1; q < NP; q++) if (M > q - 1) qc[q] = games.get(game, i, GP_AW + q - 1);
        }
        const double t = -games.get(game, i, GP_TGT + sidx);       // getQVec :109-113
        qv = t * tw;
    } else {
        const int j = b - 1;
        if (sidx < 2) qc[0] = games.get(game, i, GP_AW + j);       // :74
        double dg = 0.0, opw = 0.0;
        if (sidx < 3) { opw = games.get(game, i, GP_OPW + 3 * j + sidx); dg = -opw; }      // :91 assignment (Q4)
#pragma unroll
        for (int q = 1; q < NP; q++) if (b == q) qc[q] = dg;
        if (sidx < 3) { qv = games.get(game, i, GP_OPT + 3 * j + sidx); qv = qv * -opw; }  // :117, :121 (heading entry 0)
    }
}

// ---- the m x m solve  LHS [P | alpha] = [RHSMat | RHSVec]  (KartLQR.cs:104-105: MathNet's LU with partial pivoting, JAMA order), one right-hand
// side per lane: column cl of [RHSMat | RHSVec] in, the same column of [P | alpha] out.  Cm: the rows of [LHS | RHSMat | RHSVec] in LDS.
// NP = 2: every lane factors the whole 4 x 4 LHS in its registers — the steps of lq2_pair_solve, which both lanes of a pair run redundantly too — no
// exchange at all.
__device__ __forceinline__ void lqs_solve_column_2(const double* Cm, const int cl, double x[4], int& singular)
{
    constexpr int m = 4, CW = 13;
    double Lm[m][m], sacc[m][m];
#pragma unroll
    for (int q = 0; q < m; q++) {
#pragma unroll
        for (int c = 0; c < m; c++) { Lm[q][c] = Cm[q * CW + c]; sacc[q][c] = 0.0; }
        x[q] = Cm[q * CW + m + cl];
    }
#pragma unroll
    for (int k = 0; k < m; k++) {
#pragma unroll
        for (int q = 0; q < m; q++)
            if (q >= k) Lm[q][k] = Lm[q][k] - sacc[q][k];
        int pv = k;
        double best = fabs(Lm[k][k]);
#pragma unroll
        for (int q = 0; q < m; q++)
            if (q > k && fabs(Lm[q][k]) > best) { best = fabs(Lm[q][k]); pv = q; }
#pragma unroll
        for (int q = 0; q < m; q++)
            if (q > k && q == pv) {
#pragma unroll
                for (int c = 0; c < m; c++) {
                    double tmp = Lm[q][c]; Lm[q][c] = Lm[k][c]; Lm[k][c] = tmp;
                    if (c != k) { tmp = sacc[q][c]; sacc[q][c] = sacc[k][c]; sacc[k][c] = tmp; }
                }
                const double tmp = x[q]; x[q] = x[k]; x[k] = tmp;
            }
        const double ck = Lm[k][k];
        if (ck == 0.0) singular = 1;
        double lm[m];
#pragma unroll
        for (int q = 0; q < m; q++) {
            lm[q] = 0.0;
            if (q > k) {
                if (ck != 0.0) Lm[q][k] = Lm[q][k] / ck;
                lm[q] = Lm[q][k];
            }
        }
#pragma unroll
        for (int c = 0; c < m; c++)
            if (c > k) {
                Lm[k][c] = Lm[k][c] - sacc[k][c];
#pragma unroll
                for (int q = 0; q < m; q++)
                    if (q > k) sacc[q][c] += lm[q] * Lm[k][c];
            }
#pragma unroll
        for (int q = 0; q < m; q++)
            if (q > k) { const double temp = x[k] * lm[q]; x[q] = x[q] - temp; }
    }
    // back substitution  U X = Y  (k descending)
#pragma unroll
    for (int kk = 0; kk < m; kk++) {
        const int k = m - 1 - kk;
        const double ukk = Lm[k][k];
        x[k] = x[k] / ukk;
#pragma unroll
        for (int q = 0; q < m; q++)
            if (q < k) { const double uik = Lm[q][k]; const double temp = x[k] * uik; x[q] = x[q] - temp; }
    }
}
// NP = 3, 4: lane c < m also owns column c of the LHS; the pivot row and the multipliers of each step go through LDS (the steps of lq_solve_game's S3,
// the right-looking elimination that reproduces JAMA's accumulation order).  Lanes that duplicate a column publish identical values.
template <int NP, class X, class LDS>
__device__ __forceinline__ void lqs_solve_column_n(LDS& L, const int cl, double x[2 * NP], int& singular)
{
    constexpr int n = 4 * NP, m = 2 * NP, CW = m + n + 1;
    double col[m], sacc[m];
    {
        const double* Cm = L.Cm();
#pragma unroll
        for (int row = 0; row < m; row++) {
            const double lv = Cm[row * CW + (cl < m ? cl : 0)];
            col[row] = (cl < m) ? lv : 0.0;
            x[row] = Cm[row * CW + m + cl];
            sacc[row] = 0.0;
        }
    }
    double* lu = L.lu();
#pragma unroll
    for (int k = 0; k < m; k++) {
        if (cl == k) {
            // finalize rows >= k of column k: col[i] -= s_i   (rows < k were finalized at their own step)
#pragma unroll
            for (int q = 0; q < m; q++)
                if (q >= k) col[q] = col[q] - sacc[q];
            int p = k;
            double best = fabs(col[k]);
#pragma unroll
            for (int q = 0; q < m; q++)
                if (q > k && fabs(col[q]) > best) { best = fabs(col[q]); p = q; }
            double ck = col[k];
#pragma unroll
            for (int q = 0; q < m; q++)
                if (q > k && q == p) { ck = col[q]; col[q] = col[k]; }
            col[k] = ck;
            lu[0] = (double)p;
            if (ck == 0.0) lu[m] = 1.0;
#pragma unroll
            for (int q = 0; q < m; q++)
                if (q > k) {
                    if (ck != 0.0) col[q] = col[q] / ck;
                    lu[q] = col[q];
                }
        }
        X::sync();
        const int p = (int)lu[0];
        double lm[m];
#pragma unroll
        for (int q = 0; q < m; q++) lm[q] = (q > k) ? lu[q] : 0.0;
        if (p != k) {
            // row swap k <-> p in every other column, accumulators and right-hand sides (lane k did its own)
#pragma unroll
            for (int q = 0; q < m; q++)
                if (q > k && q == p) {
                    double tmp;
                    if (cl != k) { tmp = col[q]; col[q] = col[k]; col[k] = tmp; }
                    if (cl != k) { tmp = sacc[q]; sacc[q] = sacc[k]; sacc[k] = tmp; }
                    tmp = x[q]; x[q] = x[k]; x[k] = tmp;
                }
        }
        if (cl > k && cl < m) {
            // column cl > k: u[k] of this column becomes final, then accumulate s_i += L[i][k]*u[k]
            col[k] = col[k] - sacc[k];
#pragma unroll
            for (int q = 0; q < m; q++)
                if (q > k) sacc[q] += lm[q] * col[k];
        }
#pragma unroll
        for (int q = 0; q < m; q++)
            if (q > k) { const double temp = x[k] * lm[q]; x[q] = x[q] - temp; }
        X::sync();
    }
    // publish U (upper triangle incl. diagonal): U[q][c] = col[q] of lane c
    double* U = L.U();
    if (cl < m) {
#pragma unroll
        for (int q = 0; q < m; q++) U[q * m + cl] = col[q];
    }
    X::sync();
    // back substitution  U X = Y  (k descending)
#pragma unroll
    for (int kk = 0; kk < m; kk++) {
        const int k = m - 1 - kk;
        const double ukk = U[k * m + k];
        x[k] = x[k] / ukk;
#pragma unroll
        for (int q = 0; q < m; q++)
            if (q < k) { const double uik = U[q * m + k]; const double temp = x[k] * uik; x[q] = x[q] - temp; }
    }
    if (lu[m] != 0.0) singular = 1;
}

// `ln`: the calling lane's index within its game's lane set, 0 .. NP * 4 NP - 1 (lanes beyond it pass the last index: exact duplicates that
// write the same values to the same places).  L: the game's LDS slice, not in use by any other lane of the wave.  Every lane of the set
// must call this (and, where several sets share a wave, all of them together: X::sync is wave-wide).  u0: player 0's control at t = 0.
template <int NP, class X, bool SW = false>
__device__ __forceinline__ void lq_spread_solve(const int ln, const int game, const double dt, const GameSoA& games, LqSpreadLds<NP, SW>& L, double u0[2], int& singular)
{
    constexpr int n = 4 * NP, m = 2 * NP, CW = m + n + 1;
    const int i = ln / n, r = ln - i * n;
    const int br = r >> 2, rs = r & 3;
    // ---- the game's constants -> LDS; the lane's cost row
    X::sync();                  // (the slice may still be read by the previous game's last sweep)
    if (ln < n) L.a4[ln >> 2][ln & 3] = games.get(game, ln >> 2, GP_A4 + (ln & 3));
    if (ln < NP) L.rc[ln] = games.get(game, ln, GP_RC);
    double qc[NP], qv;
    lqs_cost_row<NP>(games, game, i, r, qc, qv);
    double z[n];
#pragma unroll
    for (int c = 0; c < n; c++) z[c] = ((c & 3) == rs) ? qc[c >> 2] : 0.0;            // KartLQR.cs:62
    double eta = qv;                                                                   // :63
    X::sync();
    const double rci = L.rc[i];
    // the position rows of F in the lane's column r (block br): F[4 br][r], F[4 br + 1][r], formed as lq_solve_game forms them: av - (0.0 + 0.0).
    // fxs / fzs[kb]: the same with the factor switched to +0.0 outside the column's own block (see ARITHMETIC above)
    double fxs[NP], fzs[NP];
    {
        const double a0 = L.a4[br][0], a1 = L.a4[br][1], a2 = L.a4[br][2], a3 = L.a4[br][3];
        const double avx = rs == 0 ? 1.0 : (rs == 2 ? a0 : (rs == 3 ? a2 : 0.0));
        const double avz = rs == 1 ? 1.0 : (rs == 2 ? a1 : (rs == 3 ? a3 : 0.0));
        const double fx = avx - (0.0 + 0.0), fz = avz - (0.0 + 0.0);
#pragma unroll
        for (int kb = 0; kb < NP; kb++) { fxs[kb] = (kb == br) ? fx : 0.0; fzs[kb] = (kb == br) ? fz : 0.0; }
    }
    singular = 0;
    const int cl = ln < n ? ln : n;             // the lane's column of [RHSMat | RHSVec] in the m x m solve (lanes beyond n duplicate lane n)
#pragma unroll 1
    for (int t = 3; t >= 0; t--) {                                                     // :64 (HKA:1201 horizon literal 3, Q6)
        // ---------------- S1: control lanes (i, 4 i + 2 + a): their share of [LHS | RHSMat | RHSVec] ----------------
        if (br == i && rs >= 2) {
            const int a = rs - 2;
            double* Cm = L.Cm();
#pragma unroll
            for (int j = 0; j < NP; j++) {
#pragma unroll
                for (int cb = 0; cb < 2; cb++) {
                    const double t1 = fma64(z[4 * j + 2 + cb], dt, 0.0);               // (Z_i B_j)[row][cb]: B_j[2 + cb][cb] = dt
                    const double s = fma64(dt, t1, 0.0);                               // B_i'(Z_i B_j)[a][cb]: only B_i[2 + a][a] = dt
                    const double rb = (a == cb) ? rci : 0.0;
                    Cm[(2 * j + a) * CW + 2 * i + cb] = (i == j) ? (rb + s) : s;       // :78 (Q1: block [j][i])
                }
                const double aj0 = L.a4[j][0], aj1 = L.a4[j][1], aj2 = L.a4[j][2], aj3 = L.a4[j][3];
                double za[4];
                za[0] = fma64(z[4 * j + 0], 1.0, 0.0);
                za[1] = fma64(z[4 * j + 1], 1.0, 0.0);
                { double s = fma64(z[4 * j + 0], aj0, 0.0); s = fma64(z[4 * j + 1], aj1, s); za[2] = fma64(z[4 * j + 2], 1.0, s); }
                { double s = fma64(z[4 * j + 0], aj2, 0.0); s = fma64(z[4 * j + 1], aj3, s); za[3] = fma64(z[4 * j + 3], 1.0, s); }
#pragma unroll
                for (int cc = 0; cc < 4; cc++) Cm[(2 * i + a) * CW + m + 4 * j + cc] = fma64(dt, za[cc], 0.0);      // B_i'(Z_i A), :89/:95
            }
            Cm[(2 * i + a) * CW + m + n] = fma64(dt, eta, 0.0);                        // B_i' eta_i, :96
        }
        if (NP > 2 && ln == 0) L.lu()[m] = 0.0;
        X::sync();
        // ---------------- S2 - S3: the lane's column of the m x m solve: column cl of P (cl < n) or alpha (cl = n) ----------------
        double x[m];
        {
            int sing = 0;
            if constexpr (NP == 2) lqs_solve_column_2(L.Cm(), cl, x, sing);
            else lqs_solve_column_n<NP, X, LqSpreadLds<NP, SW>>(L, cl, x, sing);
            if (sing) singular = 1;
        }
        // ---------------- S4: P rows, alpha and the dense entries of beta = -sum_k B_k alpha_k to LDS (one copy) ----------------
        if (ln <= n) {
#pragma unroll
            for (int q = 0; q < m; q++) {
                if (cl < n) L.Pm[q][cl] = x[q];
                else { L.al[q] = x[q]; L.beta[q] = 0.0 - fma64(dt, x[q], 0.0); }
            }
        }
        X::sync();
        // the value update of the last sweep (KartLQR.cs:113-119 at t = 0) feeds nothing: u0 below reads this sweep's P and alpha only
        if (t > 0) {
        // ---------------- S5: the lane's column of F = A - sum_k B_k P_k (dense rows); what the chains need of the player's own P and alpha ----------------
        double fd[m];                            // fd[2 k + a] = F[4 k + 2 + a][r] = [r == 4 k + 2 + a] - (0.0 + dt P[2 k + a][r])
        double p0r = 0.0, p1r = 0.0;             // P[2 i + a][r]
#pragma unroll
        for (int q = 0; q < m; q++) {
            const double pq = L.Pm[q][r];
            const double tt = fma64(dt, pq, 0.0);
            const double av = (r == 4 * (q >> 1) + 2 + (q & 1)) ? 1.0 : 0.0;
            fd[q] = av - (0.0 + tt);
            if (i == 0) L.Fd[q][r] = fd[q];
            if (q == 2 * i) p0r = pq;
            if (q == 2 * i + 1) p1r = pq;
        }
        const double ra0 = fma64(rci, L.al[2 * i], 0.0), ra1 = fma64(rci, L.al[2 * i + 1], 0.0);      // (R_i alpha_i)[a]
        X::sync();                               // Fd, beta, Pm are visible; nothing reads U / lu any more: the W area may be written
        // (SW: the players take the one W block in turn)
#pragma unroll 1
        for (int turn = 0; turn < (SW ? NP : 1); turn++) {
        const bool mine = !SW || i == turn;
        // ---------------- S6: W = Z_i F, row r (:113-116) -> LDS ----------------
        if (mine) {
            double* Wi = L.W(i) + r * n;
#pragma unroll
            for (int b = 0; b < NP; b++) {       // columns 4 b .. 4 b + 3 = the states of player b
                const double ab0 = L.a4[b][0], ab1 = L.a4[b][1], ab2 = L.a4[b][2], ab3 = L.a4[b][3];
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                for (int kb = 0; kb < NP; kb++) {
                    if (kb == b) {
                        // k = 4 b: F[4 b][4 b + (0, 1, 2, 3)] = (1, +0, A[x,v], A[x,h]);  k = 4 b + 1: (+0, 1, A[z,v], A[z,h])
                        s0 = fma64(z[4 * b], 1.0, s0); s2 = fma64(z[4 * b], ab0, s2); s3 = fma64(z[4 * b], ab2, s3);
                        s1 = fma64(z[4 * b + 1], 1.0, s1); s2 = fma64(z[4 * b + 1], ab1, s2); s3 = fma64(z[4 * b + 1], ab3, s3);
                    }
#pragma unroll
                    for (int a = 0; a < 2; a++) {
                        const double2 f01 = *reinterpret_cast<const double2*>(&L.Fd[2 * kb + a][4 * b]);
                        const double2 f23 = *reinterpret_cast<const double2*>(&L.Fd[2 * kb + a][4 * b + 2]);
                        const double zk = z[4 * kb + 2 + a];
                        s0 = fma64(zk, f01.x, s0); s1 = fma64(zk, f01.y, s1); s2 = fma64(zk, f23.x, s2); s3 = fma64(zk, f23.y, s3);
                    }
                }
                *reinterpret_cast<double2*>(&Wi[4 * b]) = make_double2(s0, s1);
                *reinterpret_cast<double2*>(&Wi[4 * b + 2]) = make_double2(s2, s3);
                HK_LQS_FENCE();
            }
        }
        X::sync();
        // ---------------- S7: Z_i <- (Q_i + P_i'(R_i P_i)) + F'(Z_i F), row r ----------------
        if (mine) {
            const double* Wi = L.W(i);
            const double* wp0 = Wi + (4 * br) * n;             // the rows of the lane's own position entries of F
            const double* wp1 = Wi + (4 * br + 1) * n;
            const double* pr0 = &L.Pm[2 * i][0];
            const double* pr1 = &L.Pm[2 * i + 1][0];
#pragma unroll
            for (int c = 0; c < n; c += 2) {
                double o0 = 0.0, o1 = 0.0;
                const double2 w0 = *reinterpret_cast<const double2*>(&wp0[c]), w1 = *reinterpret_cast<const double2*>(&wp1[c]);
#pragma unroll
                for (int kb = 0; kb < NP; kb++) {
                    o0 = fma64(fxs[kb], w0.x, o0); o1 = fma64(fxs[kb], w0.y, o1);
                    o0 = fma64(fzs[kb], w1.x, o0); o1 = fma64(fzs[kb], w1.y, o1);
#pragma unroll
                    for (int a = 0; a < 2; a++) {
                        const double2 w = *reinterpret_cast<const double2*>(&Wi[(4 * kb + 2 + a) * n + c]);
                        o0 = fma64(fd[2 * kb + a], w.x, o0); o1 = fma64(fd[2 * kb + a], w.y, o1);
                    }
                }
                const double2 q0 = *reinterpret_cast<const double2*>(&pr0[c]), q1 = *reinterpret_cast<const double2*>(&pr1[c]);
                double t20 = fma64(p0r, fma64(rci, q0.x, 0.0), 0.0);
                t20 = fma64(p1r, fma64(rci, q1.x, 0.0), t20);
                double t21 = fma64(p0r, fma64(rci, q0.y, 0.0), 0.0);
                t21 = fma64(p1r, fma64(rci, q1.y, 0.0), t21);
                const double qa = ((c & 3) == rs) ? qc[c >> 2] : 0.0, qb = (((c + 1) & 3) == rs) ? qc[(c + 1) >> 2] : 0.0;
                z[c] = (qa + t20) + o0;
                z[c + 1] = (qb + t21) + o1;
                HK_LQS_FENCE();
            }
        }
        if (SW) X::sync();          // (the next player's rows go into the same block)
        }       // turn
        // ---------------- S8: eta_i <- (q_i + P_i'(R_i alpha_i)) + F'(eta_i + Z_i beta) with the NEW Z_i (Q2, :117) ----------------
        {
            double zb = 0.0;
#pragma unroll
            for (int kb = 0; kb < NP; kb++) {
                zb = fma64(z[4 * kb + 2], L.beta[2 * kb], zb);
                zb = fma64(z[4 * kb + 3], L.beta[2 * kb + 1], zb);
            }
            L.vec[i][r] = eta + zb;
            X::sync();
            const double* vi = &L.vec[i][0];
            const double vp0 = vi[4 * br], vp1 = vi[4 * br + 1];
            double v3 = 0.0;
#pragma unroll
            for (int kb = 0; kb < NP; kb++) {
                v3 = fma64(fxs[kb], vp0, v3);
                v3 = fma64(fzs[kb], vp1, v3);
                v3 = fma64(fd[2 * kb], vi[4 * kb + 2], v3);
                v3 = fma64(fd[2 * kb + 1], vi[4 * kb + 3], v3);
            }
            double v2 = fma64(p0r, ra0, 0.0);
            v2 = fma64(p1r, ra1, v2);
            eta = (qv + v2) + v3;
        }
        }       // t > 0
        X::sync();              // (t == 0: P is in LDS; t > 0: the W area and vec have been read)
    }
    // :121-126 u0 = -P_0 x0 - alpha_0   (every lane computes it)
    {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int c = 0; c < n; c++) { const double xc = games.get(game, c >> 2, GP_X0 + (c & 3)); s0 = fma64(-L.Pm[0][c], xc, s0); s1 = fma64(-L.Pm[1][c], xc, s1); }
        u0[0] = s0 - L.al[0];
        u0[1] = s1 - L.al[1];
    }
}

#ifndef HK_LQS_HOST_CHECK
// One queue of a round on the spread solver: GPW games per wave (one workgroup = one wave), grid-stride over blocks [0, nblocks)
template <int NP>
__device__ __forceinline__ void lqs_body(const int block, const int nblocks, const EnvParams& P, const HotRef hr, const GameSoA& games, const int* queue_cnt,
                                         const int* queue, hk_lq_debug* dbg_out, int* status, unsigned char* smem, unsigned long long* gstats)
{
    constexpr int G = LqSpreadDims<NP>::G, GPW = LqSpreadDims<NP>::GPW;
    LqSpreadLds<NP>* lds = reinterpret_cast<LqSpreadLds<NP>*>(smem);               // [GPW]
    const int lane = threadIdx.x & 63;
    const int grp = lane / G;
    const bool real = grp < GPW;                         // (3 players: lanes 36 .. 63 duplicate the set's last lane)
    const int slot = real ? grp : GPW - 1, ln = real ? lane - grp * G : G - 1;
    const int count = queue_cnt[NP];
    if (block == 0 && threadIdx.x == 0 && count > 0) atomicAdd(&gstats[NP], (unsigned long long)count);   // hk_prof_games
    const int* qbase = queue + (size_t)(NP - 2) * P.E * P.A;
    for (int base = block * GPW; base < count; base += nblocks * GPW) {
        const int s = base + slot;
        const bool live = s < count;
        const int game = qbase[live ? s : count - 1];    // idle slots recompute the last game and discard it
        double u0[2];
        int singular = 0;
        lq_spread_solve<NP, LqWaveSync>(ln, game, (double)P.dt, games, lds[slot], u0, singular);
        if (live && real && ln == 0) {
            if (singular) atomicOr(status, 1);
            decode_store(P, hr, game, u0[0], u0[1], (dbg_out && (P.debug & 1)) ? &dbg_out[game] : nullptr);
        }
    }
}

// IN-WAVE: the multi-player games the egos of ONE wave of env_b1_kernel have just assembled (qn = the lane's player count, 0: none), solved on the spot by
// that wave — no queue, no solver launch; the controls come back to the ego's own lane (ua, ub), which decodes them into the registers it is about to
// store.  Everything happens in `wave_lds`, the wave's own slice of the kart staging area (64 KartS = 5 632 B, dead once phase_assemble has returned):
// 2-player games three at a time, 3-player games one at a time, 4-player games one at a time with the players taking ONE W block in turn (SW).
// NO workgroup barrier anywhere (profiles/r06_b_short_call_trace.txt: a barrier at the end of the B1 kernel — the first form pooled a block's games behind
// one — cost the launch 30 us: the waves of a block finish 20 - 30 % apart, and with the barrier every one of them holds its SIMD slot until the last).
// mygame: env * A + ego of the calling lane.
struct LqsOut { double a, b; int sing; };
// The solves are real CALLS: inlined into env_b1_kernel the three solver bodies made the allocator spill ~300 registers of the kernel around them (the
// kernel is held to three waves per SIMD); across a call only the handful of values live at the call site are parked, and each body gets an allocation
// of its own (the 3- / 4-player bodies spill inside themselves: rare games).
template <int NP, bool SW>
__device__ __attribute__((noinline)) LqsOut lqs_solve_call(const int ln, const int game, const double dt, const GameSoA games, LqSpreadLds<NP, SW>* L)
{
    LqsOut o;
    double u0[2];
    int sg = 0;
    lq_spread_solve<NP, LqWaveSync, SW>(ln, game, dt, games, *L, u0, sg);
    o.a = u0[0]; o.b = u0[1]; o.sing = sg;
    return o;
}
constexpr size_t LQS_WAVE_LDS = 5632;
static_assert(3 * sizeof(LqSpreadLds<2>) <= LQS_WAVE_LDS && 4 * sizeof(LqSpreadLds<2, true>) <= LQS_WAVE_LDS && sizeof(LqSpreadLds<3>) <= LQS_WAVE_LDS && sizeof(LqSpreadLds<4, true>) <= LQS_WAVE_LDS,
              "the wave's slice holds three 2-player games, a 3-player game, or a 4-player game with one W block");
__device__ __forceinline__ double lqs_readlane(const double v, const int k)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void lqs_inwave(const EnvParams& P, const GameSoA& games, const int qn, const int mygame, unsigned char* wave_lds,
                                           double& ua, double& ub, int* status, unsigned long long* gstats)
{
    unsigned long long m2 = __ballot(qn == 2), m3 = __ballot(qn == 3), m4 = __ballot(qn == 4);
    if ((m2 | m3 | m4) == 0ull) return;
    const int lane = threadIdx.x & 63;
    const double dt = (double)P.dt;
    int sing = 0, passes = 0;
    if (lane == 0) {          // hk_prof_games
        if (m2) atomicAdd(&gstats[2], (unsigned long long)__popcll(m2));
        if (m3) atomicAdd(&gstats[3], (unsigned long long)__popcll(m3));
        if (m4) atomicAdd(&gstats[4], (unsigned long long)__popcll(m4));
    }
    if (__popcll(m2) > 3) {
        // two packs (or more) in one wave: FOUR games per pass with the players of a game taking one W block in turn (1 376 B per game) — a pass is a third
        // longer, but the wave is done in one where three sets would need two, and the launch ends with its slowest wave
        LqSpreadLds<2, true>* l2 = reinterpret_cast<LqSpreadLds<2, true>*>(wave_lds);
        const int set = lane >> 4, ln2 = lane & 15;
        while (m2 != 0ull) {
            int e[4];
            e[0] = __ffsll((long long)m2) - 1; m2 &= m2 - 1ull;
#pragma unroll
            for (int q = 1; q < 4; q++) { e[q] = e[0]; if (m2 != 0ull) { e[q] = __ffsll((long long)m2) - 1; m2 &= m2 - 1ull; } }
            const int game = __shfl(mygame, set == 0 ? e[0] : (set == 1 ? e[1] : (set == 2 ? e[2] : e[3])), 64);
            const LqsOut o = lqs_solve_call<2, true>(ln2, game, dt, games, &l2[set]);
            passes++;
#pragma unroll
            for (int q = 3; q >= 0; q--) {
                const double a = lqs_readlane(o.a, 16 * q), b = lqs_readlane(o.b, 16 * q);
                if (lane == e[q]) { ua = a; ub = b; }
            }
            sing |= o.sing;
        }
    } else {
        LqSpreadLds<2>* l2 = reinterpret_cast<LqSpreadLds<2>*>(wave_lds);
        const int grp = lane >> 4, set = grp < 3 ? grp : 2, ln2 = lane & 15;          // lanes 48 .. 63 duplicate the third set
        while (m2 != 0ull) {
            int e[3];
            e[0] = __ffsll((long long)m2) - 1; m2 &= m2 - 1ull;
            e[1] = e[0]; e[2] = e[0];                         // (a set without a game of its own solves the first one again, in its own slice)
            if (m2 != 0ull) { e[1] = __ffsll((long long)m2) - 1; m2 &= m2 - 1ull; }
            if (m2 != 0ull) { e[2] = __ffsll((long long)m2) - 1; m2 &= m2 - 1ull; }
            const int game = __shfl(mygame, set == 0 ? e[0] : (set == 1 ? e[1] : e[2]), 64);
            const LqsOut o = lqs_solve_call<2, false>(ln2, game, dt, games, &l2[set]);
            passes++;
#pragma unroll
            for (int q = 2; q >= 0; q--) {
                const double a = lqs_readlane(o.a, 16 * q), b = lqs_readlane(o.b, 16 * q);
                if (lane == e[q]) { ua = a; ub = b; }
            }
            sing |= o.sing;
        }
    }
#if HK_GA >= 4
    while (m3 != 0ull) {
        const int e = __ffsll((long long)m3) - 1; m3 &= m3 - 1ull;
        const LqsOut o = lqs_solve_call<3, false>(lane < 36 ? lane : 35, __shfl(mygame, e, 64), dt, games, reinterpret_cast<LqSpreadLds<3>*>(wave_lds));
        const double a = lqs_readlane(o.a, 0), b = lqs_readlane(o.b, 0);
        if (lane == e) { ua = a; ub = b; }
        sing |= o.sing; passes++;
    }
    while (m4 != 0ull) {
        const int e = __ffsll((long long)m4) - 1; m4 &= m4 - 1ull;
        const LqsOut o = lqs_solve_call<4, true>(lane, __shfl(mygame, e, 64), dt, games, reinterpret_cast<LqSpreadLds<4, true>*>(wave_lds));
        const double a = lqs_readlane(o.a, 0), b = lqs_readlane(o.b, 0);
        if (lane == e) { ua = a; ub = b; }
        sing |= o.sing; passes++;
    }
#endif
    // (hk_prof_games words 0 / 1: solver passes run in-wave, waves that ran any — their ratio says how well the regroup keeps the games apart)
    if (lane == 0) { atomicAdd(&gstats[0], (unsigned long long)passes); atomicAdd(&gstats[1], 1ull); }
    if (sing && lane == 0) atomicOr(status, 1);
}

// The solver launch of a round once the field has spread (hk_env_launch.h launch_lqn): blocks [0, n2) walk the 2-player queue, [n2, n2 + n3) the
// 3-player queue, the rest the 4-player queue.  Small in registers and LDS on purpose: its waves start beside the other half's resident tick / B1
// waves instead of waiting for a CU to drain.
constexpr size_t lqn_spread_lds()
{
    constexpr size_t b2 = sizeof(LqSpreadLds<2>) * LqSpreadDims<2>::GPW, b3 = sizeof(LqSpreadLds<3>), b4 = sizeof(LqSpreadLds<4>);
    return b2 > b3 ? (b2 > b4 ? b2 : b4) : (b3 > b4 ? b3 : b4);
}
#ifndef HK_LQS_WAVES
#define HK_LQS_WAVES 3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HK_LQS_WAVES))) void lqn_spread_kernel(EnvParams P, const HotRef hr, const GameSoA games, const int* queue_cnt,
                                                                                                    const int* queue, hk_lq_debug* dbg_out, int* status, int n2, int n3, int n4,
                                                                                                    unsigned long long* gstats)
{
    __shared__ __align__(16) unsigned char smem[lqn_spread_lds()];
    const int b = blockIdx.x;
    if (b < n2) lqs_body<2>(b, n2, P, hr, games, queue_cnt, queue, dbg_out, status, smem, gstats);
#if HK_GA >= 3
    else if (b < n2 + n3) lqs_body<3>(b - n2, n3, P, hr, games, queue_cnt, queue, dbg_out, status, smem, gstats);
#endif
#if HK_GA >= 4
    else lqs_body<4>(b - n2 - n3, n4, P, hr, games, queue_cnt, queue, dbg_out, status, smem, gstats);
#endif
    (void)n4;
}
#endif  // HK_LQS_HOST_CHECK
