// hk_lq_core.h — feedback LQ Nash game (coupled Riccati) for gfx950, one game per 16-lane group.
//
// Replaces KartLQR.solveFeedbackLQR (reference AI/LQR/KartLQR.cs:17-128, fp64, MathNet sparse objects).
//
// Mapping (CDNA4, 64-wide waves): a wave solves 4 independent games, one per 16-lane group.  Lane r of a group owns
// ROW r of every player's value matrix Z_i (n = 4N <= 16 rows, N <= 4 players) in registers (Z[4][16] doubles).
// Everything that crosses lanes (F, W = Z_i F, P, alpha, ...) is staged through the group's LDS slice and read back
// as broadcasts.  The m x m solve (m = 2N <= 8) is a right-looking elimination with lane c owning column c of
// [LHS | RHSMat] and lane 0 also owning RHSVec; the pivot row and multipliers of each step are published through LDS.
//
// ARITHMETIC CONTRACT (bit-exact with oracle/hk_oracle_lq.c): every inner product is a k-ascending chain
// s = fma(a_k, b_k, s) seeded with +0.0; terms whose factor is a structural zero (other players' blocks of the
// block-diagonal A and of B_i) are skipped, which leaves s unchanged exactly.  The LU follows MathNet's JAMA-style
// order: column j accumulates s_i = sum_k L[i][k]*u[k] (plain mul, add) and subtracts it once; right-hand sides are
// updated term by term (temp = b[k]*L[i][k]; b[i] -= temp).  Zero padding (rows/cols >= n, identity pivots >= m)
// only ever adds exact zeros.  Translation unit must be compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace hk {

constexpr int LQ_MAXP = 4;          // players per game handled by this core
constexpr int LQ_MAXN = 16;         // states
constexpr int LQ_MAXM = 8;          // controls
constexpr int LQ_LD = 18;           // LDS row stride in doubles: 144 B -> conflict-free b64 row writes, 16-B aligned pairs

struct __attribute__((aligned(16))) LqGroupLds {
    double F[LQ_MAXN][LQ_LD];       // F[k][c]; before F exists the same storage holds T1 (see t1())
    double W[LQ_MAXN][LQ_LD];       // staging: (Z_i A) rows, then W = Z_i F rows, then new Z_i rows
    double Pm[LQ_MAXM][LQ_LD];      // P rows; column 16 = alpha
    double Ab[LQ_MAXP][16];         // A_i (4x4 row-major)
    double Bb[LQ_MAXP][8];          // B_i (4x2)
    double Rb[LQ_MAXP][4];          // R_i (2x2)
    double RP[2][LQ_MAXN];          // R_i P_i rows
    double vec[LQ_MAXN];            // eta_i rows / (eta_i + Z_i beta)
    double beta[LQ_MAXN];
    double x0[LQ_MAXN];
    double lu[12];                  // [0] pivot row, [1..7] multipliers of the current step, [8] singular flag
    double pad_[2];                 // group stride = 7440 B = 29*256 + 16: groups land on different banks
};

__device__ __forceinline__ double fma64(double a, double b, double c) { return __builtin_fma(a, b, c); }

__device__ __forceinline__ void group_sync() { __syncthreads(); }

// Q provider concept: double Q(int i, int r, int c); double q(int i, int r);   (player i's cost, ego-local order)

template <class QP>
__device__ void lq_solve_group(const int r, const int N, const int Nmax, LqGroupLds& L, const QP& qp, const int horizon,
                               double u0[2], int& singular)
{
    const int n = 4 * N, m = 2 * N;
    const int nmax = 4 * Nmax, mmax = 2 * Nmax;
    const int ib = r >> 2;                 // block (player) this lane's row belongs to
    const bool row_ok = r < n;

    double Z[LQ_MAXP][LQ_MAXN];
    double eta[LQ_MAXP];
    double Fcol[LQ_MAXN];
    double pc[LQ_MAXM];
#pragma unroll
    for (int i = 0; i < LQ_MAXP; i++) {
        const bool ok = row_ok && i < N;
#pragma unroll
        for (int c = 0; c < LQ_MAXN; c++) Z[i][c] = (ok && c < n) ? qp.Q(i, r, c) : 0.0;     // KartLQR.cs:62
        eta[i] = ok ? qp.q(i, r) : 0.0;                                                      // :63
    }
    singular = 0;
    double alpha[LQ_MAXM];

    for (int t = horizon; t >= 0; t--) {                                                     // :64
        // ---------------- S1: rows of Z_{ib} B_j (T1) and Z_{ib} A, eta_{ib} -> LDS ----------------
        double zo[LQ_MAXN];
#pragma unroll
        for (int c = 0; c < LQ_MAXN; c++)
            zo[c] = ib == 0 ? Z[0][c] : ib == 1 ? Z[1][c] : ib == 2 ? Z[2][c] : Z[3][c];
        const double eo = ib == 0 ? eta[0] : ib == 1 ? eta[1] : ib == 2 ? eta[2] : eta[3];
#pragma unroll
        for (int j = 0; j < LQ_MAXP; j++) {
            if (j < Nmax) {
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    double s = 0.0;                                   // (Z_i B_j)[r][b], :78/:82 Zs[i].Multiply(Bs[j])
#pragma unroll
                    for (int k = 0; k < 4; k++) s = fma64(zo[4 * j + k], L.Bb[j][k * 2 + b], s);
                    (&L.F[0][0])[(j * 2 + b) * LQ_MAXN + r] = s;      // T1[j][b][r]
                }
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    double s = 0.0;                                   // (Z_i A)[r][4j+cc], :89/:95 Zs[i].Multiply(A)
#pragma unroll
                    for (int k = 0; k < 4; k++) s = fma64(zo[4 * j + k], L.Ab[j][k * 4 + cc], s);
                    L.W[r][4 * j + cc] = s;
                }
            }
        }
        L.vec[r] = eo;
        if (r == 0) L.lu[8] = 0.0;
        group_sync();
        // ---------------- S2: columns of [LHS | RHSMat | RHSVec] ----------------
        double col[LQ_MAXM], sacc[LQ_MAXM], bb[LQ_MAXM], bv[LQ_MAXM];
        {
            const int ci = r >> 1, cb = r & 1;       // LHS column r = 2*ci + cb (player ci, control cb)
#pragma unroll
            for (int j = 0; j < LQ_MAXP; j++) {
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const int row = 2 * j + a;
                    double v = 0.0, rb = 0.0, rv = 0.0;
                    if (j < Nmax) {
                        if (r < m && j < N) {
                            // T2[a][cb] = sum_k B_ci[k][a] * T1_{ci,j}[4ci+k][cb]      (:78/:82 Bs[i]' * (.))
                            double s = 0.0;
#pragma unroll
                            for (int k = 0; k < 4; k++)
                                s = fma64(L.Bb[ci][k * 2 + a], (&L.F[0][0])[(j * 2 + cb) * LQ_MAXN + 4 * ci + k], s);
                            v = (ci == j) ? (L.Rb[ci][a * 2 + cb] + s) : s;                  // :78
                        }
                        if (row_ok && j < N) {
                            double s = 0.0;                                                  // :89/:95 Bs[j]'(Z_j A)
#pragma unroll
                            for (int k = 0; k < 4; k++) s = fma64(L.Bb[j][k * 2 + a], L.W[4 * j + k][r], s);
                            rb = s;
                        }
                        if (j < N) {
                            double s = 0.0;                                                  // :96 Bs[j]' eta_j
#pragma unroll
                            for (int k = 0; k < 4; k++) s = fma64(L.Bb[j][k * 2 + a], L.vec[4 * j + k], s);
                            rv = s;
                        }
                    }
                    // identity padding for rows/cols >= m keeps the elimination well defined and exact
                    if (!(r < m && j < N)) v = (row == r) ? 1.0 : 0.0;
                    col[row] = v; sacc[row] = 0.0; bb[row] = rb; bv[row] = rv;
                }
            }
        }
        group_sync();
        // ---------------- S3: LU (JAMA order) + forward elimination of the right-hand sides ----------------
#pragma unroll
        for (int k = 0; k < LQ_MAXM; k++) {
            if (k < mmax) {
                if (r == k) {
                    // finalize rows >= k of column k: col[i] -= s_i   (rows < k were finalized at their own step)
#pragma unroll
                    for (int i = 0; i < LQ_MAXM; i++)
                        if (i >= k) col[i] = col[i] - sacc[i];
                    int p = k;
                    double best = fabs(col[k]);
#pragma unroll
                    for (int i = 0; i < LQ_MAXM; i++)
                        if (i > k && fabs(col[i]) > best) { best = fabs(col[i]); p = i; }
                    // swap rows k <-> p of this column, then multipliers
                    double ck = col[k];
#pragma unroll
                    for (int i = 0; i < LQ_MAXM; i++)
                        if (i > k && i == p) { ck = col[i]; col[i] = col[k]; }
                    col[k] = ck;
                    L.lu[0] = (double)p;
                    if (ck == 0.0) L.lu[8] = 1.0;
#pragma unroll
                    for (int i = 0; i < LQ_MAXM; i++)
                        if (i > k) {
                            if (ck != 0.0) col[i] = col[i] / ck;
                            L.lu[i] = col[i];
                        }
                }
                group_sync();
                const int p = (int)L.lu[0];
                double lm[LQ_MAXM];
#pragma unroll
                for (int i = 0; i < LQ_MAXM; i++) lm[i] = (i > k) ? L.lu[i] : 0.0;
                if (p != k) {
                    // row swap k <-> p in every other column, accumulators and right-hand sides (lane k did its own)
#pragma unroll
                    for (int i = 0; i < LQ_MAXM; i++)
                        if (i > k && i == p) {
                            double tmp;
                            if (r != k) { tmp = col[i]; col[i] = col[k]; col[k] = tmp; }
                            if (r != k) { tmp = sacc[i]; sacc[i] = sacc[k]; sacc[k] = tmp; }
                            tmp = bb[i]; bb[i] = bb[k]; bb[k] = tmp;
                            tmp = bv[i]; bv[i] = bv[k]; bv[k] = tmp;
                        }
                }
                if (r > k && r < LQ_MAXM) {
                    // column r > k: u[k] of this column becomes final, then accumulate s_i += L[i][k]*u[k]
                    col[k] = col[k] - sacc[k];
#pragma unroll
                    for (int i = 0; i < LQ_MAXM; i++)
                        if (i > k) sacc[i] += lm[i] * col[k];
                }
#pragma unroll
                for (int i = 0; i < LQ_MAXM; i++)
                    if (i > k) {
                        double temp = bb[k] * lm[i];
                        bb[i] = bb[i] - temp;
                        double tempv = bv[k] * lm[i];
                        bv[i] = bv[i] - tempv;
                    }
                group_sync();
            }
        }
        // publish U (upper triangle incl. diagonal) column by column through Pm scratch: U[i][c] = col[i] of lane c
        if (r < LQ_MAXM) {
#pragma unroll
            for (int i = 0; i < LQ_MAXM; i++) L.Pm[i][r] = col[i];
        }
        group_sync();
        // back substitution  U X = Y  (k descending)
#pragma unroll
        for (int kk = 0; kk < LQ_MAXM; kk++) {
            const int k = LQ_MAXM - 1 - kk;
            if (k < mmax) {
                const double ukk = L.Pm[k][k];
                bb[k] = bb[k] / ukk;
                bv[k] = bv[k] / ukk;
#pragma unroll
                for (int i = 0; i < LQ_MAXM; i++)
                    if (i < k) {
                        const double uik = L.Pm[i][k];
                        double temp = bb[k] * uik;
                        bb[i] = bb[i] - temp;
                        double tempv = bv[k] * uik;
                        bv[i] = bv[i] - tempv;
                    }
            }
        }
        if (L.lu[8] != 0.0) singular = 1;
        group_sync();
        // ---------------- S4: publish P (column r) and alpha ----------------
#pragma unroll
        for (int i = 0; i < LQ_MAXM; i++) {
            pc[i] = bb[i];
            L.Pm[i][r] = bb[i];
            if (r == 0) L.Pm[i][16] = bv[i];
        }
        group_sync();
#pragma unroll
        for (int i = 0; i < LQ_MAXM; i++) alpha[i] = L.Pm[i][16];
        // ---------------- S5: F = A - sum_k B_k P_k (column r), beta = -sum_k B_k alpha_k (row r) ----------------
#pragma unroll
        for (int k = 0; k < LQ_MAXP; k++) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int row = 4 * k + rr;
                double f = 0.0;
                if (k < Nmax) {
                    double tt = 0.0;                                          // :110 Bs[k] * P_k
                    tt = fma64(L.Bb[k][rr * 2 + 0], pc[2 * k + 0], tt);
                    tt = fma64(L.Bb[k][rr * 2 + 1], pc[2 * k + 1], tt);
                    const double acc = 0.0 + tt;                              // Aggregate seed (zero matrix) + B_k P_k
                    const double av = (ib == k) ? L.Ab[k][rr * 4 + (r & 3)] : 0.0;
                    f = av - acc;
                    if (!(row_ok && k < N)) f = 0.0;
                }
                Fcol[row] = f;
                L.F[row][r] = f;
            }
        }
        {
            double tt = 0.0;                                                  // :111 beta row r (block ib)
            const int kb = ib < LQ_MAXP ? ib : 0;
            const double a0 = kb == 0 ? alpha[0] : kb == 1 ? alpha[2] : kb == 2 ? alpha[4] : alpha[6];
            const double a1 = kb == 0 ? alpha[1] : kb == 1 ? alpha[3] : kb == 2 ? alpha[5] : alpha[7];
            tt = fma64(L.Bb[kb][(r & 3) * 2 + 0], a0, tt);
            tt = fma64(L.Bb[kb][(r & 3) * 2 + 1], a1, tt);
            L.beta[r] = row_ok ? (0.0 - tt) : 0.0;
        }
        group_sync();
        // ---------------- S6: per player Z_i, eta_i update (:113-119) ----------------
#pragma unroll
        for (int i = 0; i < LQ_MAXP; i++) {
            if (i < Nmax) {
                const bool pl_ok = i < N;
                // W = Z_i F (row r)
                for (int c = 0; c < nmax; c += 2) {
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int k = 0; k < LQ_MAXN; k++) {
                        if (k < nmax) {
                            const double2 f = *reinterpret_cast<const double2*>(&L.F[k][c]);
                            s0 = fma64(Z[i][k], f.x, s0);
                            s1 = fma64(Z[i][k], f.y, s1);
                        }
                    }
                    *reinterpret_cast<double2*>(&L.W[r][c]) = make_double2(s0, s1);
                }
                // R_i P_i (column r): (RP)[a][r] = R[a][0] P[2i][r] + R[a][1] P[2i+1][r]
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    double s = 0.0;
                    s = fma64(L.Rb[i][a * 2 + 0], pc[2 * i + 0], s);
                    s = fma64(L.Rb[i][a * 2 + 1], pc[2 * i + 1], s);
                    L.RP[a][r] = s;
                }
                group_sync();
                // Z_i <- (Q_i + P_i'(R_i P_i)) + F'(Z_i F)   (row r), written in place over W[r][*]
                for (int c = 0; c < nmax; c++) {
                    double o = 0.0;
#pragma unroll
                    for (int k = 0; k < LQ_MAXN; k++)
                        if (k < nmax) o = fma64(Fcol[k], L.W[k][c], o);
                    double t2 = 0.0;
                    t2 = fma64(pc[2 * i + 0], L.RP[0][c], t2);
                    t2 = fma64(pc[2 * i + 1], L.RP[1][c], t2);
                    const double qv = (pl_ok && row_ok && c < n) ? qp.Q(i, r, c) : 0.0;
                    L.W[r][c] = (qv + t2) + o;
                }
                group_sync();
#pragma unroll
                for (int c = 0; c < LQ_MAXN; c++) Z[i][c] = (pl_ok && row_ok && c < n) ? L.W[r][c] : 0.0;
                // eta_i <- (q_i + P_i'(R_i alpha_i)) + F'(eta_i + Z_i beta)    with the NEW Z_i (Q2)
                double zb = 0.0;
#pragma unroll
                for (int k = 0; k < LQ_MAXN; k++)
                    if (k < nmax) zb = fma64(Z[i][k], L.beta[k], zb);
                group_sync();           // everyone finished reading W before vec is reused? (vec is separate) keep order simple
                L.vec[r] = eta[i] + zb;
                group_sync();
                double v3 = 0.0;
#pragma unroll
                for (int k = 0; k < LQ_MAXN; k++)
                    if (k < nmax) v3 = fma64(Fcol[k], L.vec[k], v3);
                double ra0 = 0.0, ra1 = 0.0;
                ra0 = fma64(L.Rb[i][0], alpha[2 * i + 0], ra0);
                ra0 = fma64(L.Rb[i][1], alpha[2 * i + 1], ra0);
                ra1 = fma64(L.Rb[i][2], alpha[2 * i + 0], ra1);
                ra1 = fma64(L.Rb[i][3], alpha[2 * i + 1], ra1);
                double v2 = 0.0;
                v2 = fma64(pc[2 * i + 0], ra0, v2);
                v2 = fma64(pc[2 * i + 1], ra1, v2);
                const double qi = (pl_ok && row_ok) ? qp.q(i, r) : 0.0;
                eta[i] = (pl_ok && row_ok) ? ((qi + v2) + v3) : 0.0;
                group_sync();
            }
        }
    }
    // :121-126 u0 = -P_0 x0 - alpha_0   (all lanes compute it redundantly)
#pragma unroll
    for (int a = 0; a < 2; a++) {
        double s = 0.0;
        for (int c = 0; c < n; c++) s = fma64(-L.Pm[a][c], L.x0[c], s);
        u0[a] = s - L.Pm[a][16];
    }
}

}  // namespace hk
