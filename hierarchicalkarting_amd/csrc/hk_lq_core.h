// hk_lq_core.h — feedback LQ Nash game (coupled Riccati) for gfx950, templated on the player count NP.
//
// Replaces KartLQR.solveFeedbackLQR (reference AI/LQR/KartLQR.cs:17-128, fp64, MathNet sparse objects).
//
// Mapping (CDNA4, 64-wide waves): one game occupies n = 4*NP consecutive lanes, so a wave solves 64/n games at once
// (NP = 2: 8 games, NP = 3: 5, NP = 4: 4).  Lane r of a game owns ROW r of every player's value matrix Z_i in registers
// (Z[NP][n] doubles).  Everything that crosses lanes (F, W = Z_i F, P, alpha, ...) is staged through the game's LDS slice
// and read back as broadcasts.  The m x m solve (m = 2*NP) is a right-looking elimination with lane c owning column c of
// [LHS | RHSMat] and lane 0 also owning RHSVec; the pivot row and multipliers of each step are published through LDS.
// All loop bounds are compile-time, nothing is padded.
//
// ARITHMETIC CONTRACT (bit-exact with oracle/hk_oracle_lq.c): every inner product is a k-ascending chain
// s = fma(a_k, b_k, s) seeded with +0.0; terms whose factor is a structural zero (other players' blocks of the
// block-diagonal A and of B_i) are skipped, which leaves s unchanged exactly.  The LU follows MathNet's JAMA-style
// order: column j accumulates s_i = sum_k L[i][k]*u[k] (plain mul, add) and subtracts it once; right-hand sides are
// updated term by term (temp = b[k]*L[i][k]; b[i] -= temp).  Translation unit must be compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace hk {

constexpr int LQ_MAXP = 4;          // largest player count of the env path (agents per env)
constexpr int LQ_BATCH_MAXP = 8;    // largest player count of hk_lq_solve_batch (the core is generic in NP; beyond 4 players the
                                    // per-lane value-matrix rows no longer fit the register file and spill: functional, not tuned)

__device__ __forceinline__ double fma64(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <int NP>
struct LqDims {
    static constexpr int n = 4 * NP;            // states
    static constexpr int m = 2 * NP;            // controls
    static constexpr int LD = n + 2;            // LDS row stride in doubles (16-B aligned pairs, spreads rows over banks)
    static constexpr int GPW = 64 / n;          // games per wave
    static constexpr int SLOTS = (64 % n) ? GPW + 1 : GPW;   // + one scratch slot for the lanes that do not fill a game
};

template <int NP>
struct __attribute__((aligned(16))) LqGameLds {
    static constexpr int n = LqDims<NP>::n, m = LqDims<NP>::m, LD = LqDims<NP>::LD;
    double F[n][LD];            // F[k][c]; before F exists the same storage holds T1 = rows of Z_i B_j
    double W[n][LD];            // staging: (Z_i A) rows, then W = Z_i F rows, then new Z_i rows
    double Pm[m][LD];           // P rows; column n = alpha.  Also carries U during the back substitution
    double Ab[NP][16];          // A_i (4x4 row-major)
    double Bb[NP][8];           // B_i (4x2)
    double Rb[NP][4];           // R_i (2x2)
    double RP[2][n];            // R_i P_i rows
    double vec[n];              // eta_i rows / (eta_i + Z_i beta)
    double beta[n];
    double x0[n];
    double lu[m + 2];           // [0] pivot row, [1..m-1] multipliers of the current step, [m] singular flag
};

// How the lanes of a game synchronise their LDS traffic.  BlockSync: the solver kernels (one wave per workgroup; a workgroup
// barrier).  WaveSync: the solve runs inside a multi-wave kernel on a subset of one wave's lanes (the tick kernel's in-wave
// path): a workgroup barrier is not allowed there, and not needed — the LDS serves a wave's requests in order, so a compiler
// fence is all it takes.
struct LqBlockSync { static __device__ __forceinline__ void sync() { __syncthreads(); } };
struct LqWaveSync {
    static __device__ __forceinline__ void sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

// Q provider concept: double Q(int i, int r, int c); double q(int i, int r);   (player i's cost, ego-local order)
// With LqBlockSync all 64 lanes of the wave must call this (it contains block barriers); r in [0, n) for real lanes.
// BICYCLE: the game is one of SolveLQR's (block-diagonal A, B_k with zero position rows — the linearised bicycle of
// KartLQRDynamics.cs:40-62).  Then the position rows (x, z) of F = A - sum B_k P_k are exactly +0.0 outside their own player's
// block, whatever P is, and the chain W = Z_i F may leave those terms out (fma(z, +0.0, s) = s exactly): 37 % of the chain at
// N = 4.  The generic hk_lq_solve_batch (arbitrary A, B) keeps every term.
// MFMA (NP = 3, 4; `slots` = the LDS slices of all the wave's games, this game's first): the two dense products of the value update run
// on the fp64 matrix core, one game of the wave after the other with all 64 lanes (lane l = (g = l >> 4, cc = l & 15) supplies
// Z_i[cc][4 s + g] and F[4 s + g][cc], K-step s), while everything else — the m x m solve, the small chains — keeps the lane-per-row
// layout that serves 64 / n games per instruction.  v_mfma_f64_16x16x4_f64 is bit for bit the k-ascending fma chain of this contract
// (tools/experiments/mfma_f64_check.hip), the rows / columns beyond n are zero padding, so the results are those of the chains below.
// The chains read F and W back from LDS for every term (1 024 ds_read_b128 per game and sweep at N = 4, which bound the kernel);
// the matrix core takes its operands from eight registers.
typedef double lq_d4 __attribute__((ext_vector_type(4)));
template <int NP, class QP, bool BICYCLE = false, class SYNC = LqBlockSync, bool MFMA = false>
__device__ void lq_solve_game(const int r, LqGameLds<NP>& L, const QP& qp, const int horizon, double u0[2], int& singular,
                              LqGameLds<NP>* slots = nullptr)
{
    constexpr int n = LqDims<NP>::n, m = LqDims<NP>::m;
    static_assert(!MFMA || NP == 3 || NP == 4, "one 16 x 16 tile per value matrix");
    const int ib = (r >> 2) < NP ? (r >> 2) : NP - 1;       // player block this lane's row belongs to

    double Z[NP][n];
    double eta[NP];
    double Fcol[n];
    double pc[m];
#pragma unroll
    for (int i = 0; i < NP; i++) {
#pragma unroll
        for (int c = 0; c < n; c++) Z[i][c] = qp.Q(i, r, c);                                  // KartLQR.cs:62
        eta[i] = qp.q(i, r);                                                                  // :63
    }
    singular = 0;
    double alpha[m];

    for (int t = horizon; t >= 0; t--) {                                                      // :64
        // ---------------- S1: rows of Z_{ib} B_j (T1) and Z_{ib} A, eta_{ib} -> LDS ----------------
        {
            double zo[n];
#pragma unroll
            for (int c = 0; c < n; c++) {
                double v = Z[0][c];
#pragma unroll
                for (int i = 1; i < NP; i++) v = (ib == i) ? Z[i][c] : v;
                zo[c] = v;
            }
            double eo = eta[0];
#pragma unroll
            for (int i = 1; i < NP; i++) eo = (ib == i) ? eta[i] : eo;
#pragma unroll
            for (int j = 0; j < NP; j++) {
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    double s = 0.0;                                   // (Z_i B_j)[r][b], :78/:82 Zs[i].Multiply(Bs[j])
#pragma unroll
                    for (int k = 0; k < 4; k++) s = fma64(zo[4 * j + k], L.Bb[j][k * 2 + b], s);
                    (&L.F[0][0])[(j * 2 + b) * n + r] = s;            // T1[j][b][r]
                }
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    double s = 0.0;                                   // (Z_i A)[r][4j+cc], :89/:95 Zs[i].Multiply(A)
#pragma unroll
                    for (int k = 0; k < 4; k++) s = fma64(zo[4 * j + k], L.Ab[j][k * 4 + cc], s);
                    L.W[r][4 * j + cc] = s;
                }
            }
            L.vec[r] = eo;
            if (r == 0) L.lu[m] = 0.0;
        }
        SYNC::sync();
        // ---------------- S2: columns of [LHS | RHSMat | RHSVec] ----------------
        double col[m], sacc[m], bb[m], bv[m];
        {
            const int ci = (r >> 1) < NP ? (r >> 1) : NP - 1, cb = r & 1;   // LHS column r = 2*ci + cb (player ci, control cb)
#pragma unroll
            for (int j = 0; j < NP; j++) {
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const int row = 2 * j + a;
                    double v = 0.0;
                    if (r < m) {
                        // T2[a][cb] = sum_k B_ci[k][a] * T1_{ci,j}[4ci+k][cb]      (:78/:82 Bs[i]' * (.))
                        double s = 0.0;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            s = fma64(L.Bb[ci][k * 2 + a], (&L.F[0][0])[(j * 2 + cb) * n + 4 * ci + k], s);
                        v = (ci == j) ? (L.Rb[ci][a * 2 + cb] + s) : s;                  // :78
                    }
                    double rb = 0.0;                                                     // :89/:95 Bs[j]'(Z_j A), column r
#pragma unroll
                    for (int k = 0; k < 4; k++) rb = fma64(L.Bb[j][k * 2 + a], L.W[4 * j + k][r], rb);
                    double rv = 0.0;                                                     // :96 Bs[j]' eta_j
#pragma unroll
                    for (int k = 0; k < 4; k++) rv = fma64(L.Bb[j][k * 2 + a], L.vec[4 * j + k], rv);
                    col[row] = v; sacc[row] = 0.0; bb[row] = rb; bv[row] = rv;
                }
            }
        }
        SYNC::sync();
        // ---------------- S3: LU (JAMA order) + forward elimination of the right-hand sides ----------------
#pragma unroll
        for (int k = 0; k < m; k++) {
            if (r == k) {
                // finalize rows >= k of column k: col[i] -= s_i   (rows < k were finalized at their own step)
#pragma unroll
                for (int i = 0; i < m; i++)
                    if (i >= k) col[i] = col[i] - sacc[i];
                int p = k;
                double best = fabs(col[k]);
#pragma unroll
                for (int i = 0; i < m; i++)
                    if (i > k && fabs(col[i]) > best) { best = fabs(col[i]); p = i; }
                double ck = col[k];
#pragma unroll
                for (int i = 0; i < m; i++)
                    if (i > k && i == p) { ck = col[i]; col[i] = col[k]; }
                col[k] = ck;
                L.lu[0] = (double)p;
                if (ck == 0.0) L.lu[m] = 1.0;
#pragma unroll
                for (int i = 0; i < m; i++)
                    if (i > k) {
                        if (ck != 0.0) col[i] = col[i] / ck;
                        L.lu[i] = col[i];
                    }
            }
            SYNC::sync();
            const int p = (int)L.lu[0];
            double lm[m];
#pragma unroll
            for (int i = 0; i < m; i++) lm[i] = (i > k) ? L.lu[i] : 0.0;
            if (p != k) {
                // row swap k <-> p in every other column, accumulators and right-hand sides (lane k did its own)
#pragma unroll
                for (int i = 0; i < m; i++)
                    if (i > k && i == p) {
                        double tmp;
                        if (r != k) { tmp = col[i]; col[i] = col[k]; col[k] = tmp; }
                        if (r != k) { tmp = sacc[i]; sacc[i] = sacc[k]; sacc[k] = tmp; }
                        tmp = bb[i]; bb[i] = bb[k]; bb[k] = tmp;
                        tmp = bv[i]; bv[i] = bv[k]; bv[k] = tmp;
                    }
            }
            if (r > k && r < m) {
                // column r > k: u[k] of this column becomes final, then accumulate s_i += L[i][k]*u[k]
                col[k] = col[k] - sacc[k];
#pragma unroll
                for (int i = 0; i < m; i++)
                    if (i > k) sacc[i] += lm[i] * col[k];
            }
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i > k) {
                    double temp = bb[k] * lm[i];
                    bb[i] = bb[i] - temp;
                    double tempv = bv[k] * lm[i];
                    bv[i] = bv[i] - tempv;
                }
            SYNC::sync();
        }
        // publish U (upper triangle incl. diagonal): U[i][c] = col[i] of lane c
        if (r < m) {
#pragma unroll
            for (int i = 0; i < m; i++) L.Pm[i][r] = col[i];
        }
        SYNC::sync();
        // back substitution  U X = Y  (k descending)
#pragma unroll
        for (int kk = 0; kk < m; kk++) {
            const int k = m - 1 - kk;
            const double ukk = L.Pm[k][k];
            bb[k] = bb[k] / ukk;
            bv[k] = bv[k] / ukk;
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i < k) {
                    const double uik = L.Pm[i][k];
                    double temp = bb[k] * uik;
                    bb[i] = bb[i] - temp;
                    double tempv = bv[k] * uik;
                    bv[i] = bv[i] - tempv;
                }
        }
        if (L.lu[m] != 0.0) singular = 1;
        SYNC::sync();
        // ---------------- S4: publish P (column r) and alpha ----------------
#pragma unroll
        for (int i = 0; i < m; i++) {
            pc[i] = bb[i];
            L.Pm[i][r] = bb[i];
            if (r == 0) L.Pm[i][n] = bv[i];
        }
        SYNC::sync();
#pragma unroll
        for (int i = 0; i < m; i++) alpha[i] = L.Pm[i][n];
        // the value update of the last sweep (KartLQR.cs:113-119 at t = 0) feeds nothing: u0 below reads this sweep's P and alpha only
        if (t == 0) break;
        // ---------------- S5: F = A - sum_k B_k P_k (column r), beta = -sum_k B_k alpha_k (row r) ----------------
#pragma unroll
        for (int k = 0; k < NP; k++) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int row = 4 * k + rr;
                double tt = 0.0;                                          // :110 Bs[k] * P_k
                tt = fma64(L.Bb[k][rr * 2 + 0], pc[2 * k + 0], tt);
                tt = fma64(L.Bb[k][rr * 2 + 1], pc[2 * k + 1], tt);
                const double acc = 0.0 + tt;                              // Aggregate seed (zero matrix) + B_k P_k
                const double av = (ib == k) ? L.Ab[k][rr * 4 + (r & 3)] : 0.0;
                const double f = av - acc;
                Fcol[row] = f;
                L.F[row][r] = f;
            }
        }
        {
            double a0 = alpha[0], a1 = alpha[1];
#pragma unroll
            for (int i = 1; i < NP; i++) { a0 = (ib == i) ? alpha[2 * i] : a0; a1 = (ib == i) ? alpha[2 * i + 1] : a1; }
            double tt = 0.0;                                              // :111 beta row r (block ib)
            tt = fma64(L.Bb[ib][(r & 3) * 2 + 0], a0, tt);
            tt = fma64(L.Bb[ib][(r & 3) * 2 + 1], a1, tt);
            L.beta[r] = 0.0 - tt;
        }
        SYNC::sync();
        // ---------------- S6: per player Z_i, eta_i update (:113-119) ----------------
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (MFMA) {
                // rows of Z_i -> this game's W slice
#pragma unroll
                for (int c = 0; c < n; c += 2) *reinterpret_cast<double2*>(&L.W[r][c]) = make_double2(Z[i][c], Z[i][c + 1]);
            } else {
            // W = Z_i F (row r): four independent fma chains per pass (columns c..c+3)
            for (int c = 0; c < n; c += 4) {                // columns c .. c + 3 = the states of player c / 4
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                for (int k = 0; k < n; k++) {
                    // (BICYCLE) rows k = 4 j, 4 j + 1 of F are zero in these columns unless j is the columns' player
                    if (BICYCLE && (k & 3) < 2 && (k >> 2) != (c >> 2)) continue;
                    const double2 f = *reinterpret_cast<const double2*>(&L.F[k][c]);
                    const double2 g = *reinterpret_cast<const double2*>(&L.F[k][c + 2]);
                    s0 = fma64(Z[i][k], f.x, s0);
                    s1 = fma64(Z[i][k], f.y, s1);
                    s2 = fma64(Z[i][k], g.x, s2);
                    s3 = fma64(Z[i][k], g.y, s3);
                }
                *reinterpret_cast<double2*>(&L.W[r][c]) = make_double2(s0, s1);
                *reinterpret_cast<double2*>(&L.W[r][c + 2]) = make_double2(s2, s3);
            }
            }
            // R_i P_i (column r): (RP)[a][r] = R[a][0] P[2i][r] + R[a][1] P[2i+1][r]
#pragma unroll
            for (int a = 0; a < 2; a++) {
                double s = 0.0;
                s = fma64(L.Rb[i][a * 2 + 0], pc[2 * i + 0], s);
                s = fma64(L.Rb[i][a * 2 + 1], pc[2 * i + 1], s);
                L.RP[a][r] = s;
            }
            SYNC::sync();
            if (MFMA) {
#ifdef __HIP_DEVICE_COMPILE__
                // every game of the wave in turn, all 64 lanes: W = Z_i F, then W' F = (F' W)', back into the game's W slice
                const int ml = threadIdx.x & 63, mg = ml >> 4, mc = ml & 15;
#ifndef HK_LQ_MFMA_UNROLL
#define HK_LQ_MFMA_UNROLL 1
#endif
#pragma unroll HK_LQ_MFMA_UNROLL
                for (int q = 0; q < LqDims<NP>::GPW; q++) {
                    LqGameLds<NP>& G = slots[q];
                    double za[4], fb[4];
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) {
                        const int kk = 4 * s4 + mg;
                        const bool in = mc < n && kk < n;
                        za[s4] = in ? G.W[mc][kk] : 0.0;              // A operand: Z_i[mc][kk]
                        fb[s4] = in ? G.F[kk][mc] : 0.0;              // B operand: F[kk][mc]
                    }
                    lq_d4 w4 = {0.0, 0.0, 0.0, 0.0}, o4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) w4 = __builtin_amdgcn_mfma_f64_16x16x4f64(za[s4], fb[s4], w4, 0, 0, 0);   // w4[j] = W[mg + 4 j][mc]
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) o4 = __builtin_amdgcn_mfma_f64_16x16x4f64(w4[s4], fb[s4], o4, 0, 0, 0);   // o4[j] = (F' W)[mc][4 j + mg]
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) {
                        const int kk = 4 * s4 + mg;
                        if (mc < n && kk < n) G.W[mc][kk] = o4[s4];    // (the wave's LDS requests are served in order: the reads above are done)
                    }
                }
#endif
                SYNC::sync();
            }
            // Z_i <- (Q_i + P_i'(R_i P_i)) + F'(Z_i F)   (row r)
            if (MFMA) {
#pragma unroll
                for (int c = 0; c < n; c += 2) {
                    const double2 w = *reinterpret_cast<const double2*>(&L.W[r][c]);        // (F' W)[r][c], [r][c + 1] from the matrix core
                    double t20 = 0.0, t21 = 0.0;
                    t20 = fma64(pc[2 * i + 0], L.RP[0][c], t20);
                    t20 = fma64(pc[2 * i + 1], L.RP[1][c], t20);
                    t21 = fma64(pc[2 * i + 0], L.RP[0][c + 1], t21);
                    t21 = fma64(pc[2 * i + 1], L.RP[1][c + 1], t21);
                    Z[i][c] = (qp.Q(i, r, c) + t20) + w.x;
                    Z[i][c + 1] = (qp.Q(i, r, c + 1) + t21) + w.y;
                }
            } else {
            // ... written in place over W[r][*]
            for (int c = 0; c < n; c += 2) {
                double o0 = 0.0, o1 = 0.0;               // two independent chains (columns c, c+1)
#pragma unroll
                for (int k = 0; k < n; k++) {
                    const double2 w = *reinterpret_cast<const double2*>(&L.W[k][c]);
                    o0 = fma64(Fcol[k], w.x, o0);
                    o1 = fma64(Fcol[k], w.y, o1);
                }
                double t20 = 0.0, t21 = 0.0;
                t20 = fma64(pc[2 * i + 0], L.RP[0][c], t20);
                t20 = fma64(pc[2 * i + 1], L.RP[1][c], t20);
                t21 = fma64(pc[2 * i + 0], L.RP[0][c + 1], t21);
                t21 = fma64(pc[2 * i + 1], L.RP[1][c + 1], t21);
                const double z0 = (qp.Q(i, r, c) + t20) + o0;
                const double z1 = (qp.Q(i, r, c + 1) + t21) + o1;
                *reinterpret_cast<double2*>(&L.W[r][c]) = make_double2(z0, z1);
            }
            SYNC::sync();
#pragma unroll
            for (int c = 0; c < n; c++) Z[i][c] = L.W[r][c];
            }
            // eta_i <- (q_i + P_i'(R_i alpha_i)) + F'(eta_i + Z_i beta)    with the NEW Z_i (Q2)
            double zb = 0.0;
#pragma unroll
            for (int k = 0; k < n; k++) zb = fma64(Z[i][k], L.beta[k], zb);
            L.vec[r] = eta[i] + zb;
            SYNC::sync();
            double v3 = 0.0;
#pragma unroll
            for (int k = 0; k < n; k++) v3 = fma64(Fcol[k], L.vec[k], v3);
            double ra0 = 0.0, ra1 = 0.0;
            ra0 = fma64(L.Rb[i][0], alpha[2 * i + 0], ra0);
            ra0 = fma64(L.Rb[i][1], alpha[2 * i + 1], ra0);
            ra1 = fma64(L.Rb[i][2], alpha[2 * i + 0], ra1);
            ra1 = fma64(L.Rb[i][3], alpha[2 * i + 1], ra1);
            double v2 = 0.0;
            v2 = fma64(pc[2 * i + 0], ra0, v2);
            v2 = fma64(pc[2 * i + 1], ra1, v2);
            eta[i] = (qp.q(i, r) + v2) + v3;
            SYNC::sync();
        }
    }
    // :121-126 u0 = -P_0 x0 - alpha_0   (all lanes compute it redundantly)
#pragma unroll
    for (int a = 0; a < 2; a++) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < n; c++) s = fma64(-L.Pm[a][c], L.x0[c], s);
        u0[a] = s - L.Pm[a][n];
    }
}

}  // namespace hk
