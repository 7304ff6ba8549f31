// hk_lq_mfma.h — the feedback LQ Nash game of 3 and 4 players (n = 12 / 16 states) with ONE game per wave and the two dense products of
// the value update, W = Z_i F and F' W, on the fp64 matrix core (v_mfma_f64_16x16x4_f64).
//
// Replaces, for NP = 3, 4, the lane-per-row core of hk_lq_core.h (KartLQR.solveFeedbackLQR, reference AI/LQR/KartLQR.cs:17-128).
// That core kept row r of every Z_i on lane r of a 4 NP-lane group and read F and W back from LDS for every term of every chain:
// 4 096 ds_read_b128 and ~85 workgroup barriers per solve, 59 us for one solve and 10.7 % of the fp64 vector peak in bulk.  Here:
//
//   lane l = (g = l >> 4, c = l & 15).  Z_i lives in the A-operand layout of the 16x16x4 MFMA: Zr[i][s] = Z_i[c][4 s + g].  With
//   Freg[s] = F[4 s + g][c] (the B-operand layout) four MFMAs give W = Z_i F in the result layout Wr[j] = W[g + 4 j][c] — which is
//   the A-operand layout of W' for K-step j — so four more MFMAs with the SAME Freg give (W' F)[g + 4 j][c] = (F' W)[c][4 j + g]:
//   the new Z_i, already in the A-operand layout of the next sweep.  No transposition, no LDS, no barrier in the recursion itself.
//   The m x m solve (m = 2 NP) runs redundantly in the four 16-lane groups (lane c owns column c of [LHS | RHSMat], every lane the
//   RHSVec column); the pivot row and the multipliers of a step come from lane k by v_readlane.  What still crosses lanes through
//   LDS is small: the rows of block i of Z_i (S1), the eta / vec vectors and P.
//
// ARITHMETIC: the contract of hk_lq_core.h, unchanged.  The f64 MFMA is bit for bit the k-ascending chain
// fma(a_3, b_3, fma(a_2, b_2, fma(a_1, b_1, fma(a_0, b_0, c)))) with k = lane >> 4 (tools/experiments/mfma_f64_check.hip: 0 of 512 000
// outputs differ on operands spread over 60 binades), K-steps ascending, seeded with +0.0; the padding of a 3-player game (rows and
// columns 12 .. 15) and the structural zeros of F contribute fma(x, +-0, s) = s exactly.  Every other expression is the one of
// lq_solve_game, evaluated on another lane.  tests/test_lq_mfma_host.py runs this header on the host (64 threads = the lanes, the MFMA
// as that chain) against the C oracle; on the GPU the golden vectors and every env parity test hold bit for bit.
#pragma once
#ifndef HK_LQM_HOST_CHECK
#include "hk_lq_core.h"
#endif

namespace hk {

#ifdef HK_LQM_HOST_CHECK
struct lqm_d4 { double v[4]; double& operator[](int i) { return v[i]; } const double& operator[](int i) const { return v[i]; } };
#else
typedef double lqm_d4 __attribute__((ext_vector_type(4)));
#endif

#ifndef HK_LQM_HOST_CHECK
struct LqmDev {
    static __device__ __forceinline__ lqm_d4 mfma(double a, double b, lqm_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // value of lane `k` (wave-uniform k) in every lane
    static __device__ __forceinline__ double readlane(double v, int k)
    {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ int readlane_i(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
    // LDS written by some lanes of the wave, read by others: the LDS serves a wave's requests in order, so ordering the compiler is enough
    static __device__ __forceinline__ void sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};
#endif

template <int NP>
struct __attribute__((aligned(16))) LqMfmaLds {
    static constexpr int ZLD = 18;      // row stride of the Z copies in doubles (16-B aligned pairs, rows spread over banks)
    double Zl[NP][16][ZLD];             // Z_i[r][k] (rows / columns >= n are zero)
    double T1[NP][4][2 * NP];           // (Z_i B_j)[4 i + rr][b] at [i][rr][2 j + b]
    double ZA[NP][4][16];               // (Z_i A)[4 i + rr][col]
    double Pl[2 * NP][ZLD];             // P rows (columns >= n are zero)
    double eta[NP][16];
    double vec[NP][16];                 // eta_i + Z_i beta
    double Ab[NP][16];                  // A_i (4x4 row-major)
    double Bb[NP][8];                   // B_i (4x2)
    double Rb[NP][4];                   // R_i (2x2)
    double x0[16];
};

// Q provider concept as in hk_lq_core.h.  All 64 lanes of the wave must call this; the game's constants (Ab, Bb, Rb, x0) are in L and
// visible (the caller synchronised).  X supplies the three wave primitives (LqmDev; the host check substitutes threads and barriers).
template <int NP, class QP, class X>
__device__ void lq_solve_game_mfma(const int lane, LqMfmaLds<NP>& L, const QP& qp, const int horizon, double u0[2], int& singular)
{
    static_assert(NP == 3 || NP == 4, "one 16 x 16 tile per value matrix: 3 or 4 players");
    constexpr int n = 4 * NP, m = 2 * NP;
    const int g = lane >> 4, c = lane & 15;
    const bool rowok = c < n;

    double Qr[NP][4], Zr[NP][4], qv[NP], eta[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int col = 4 * s + g;
            Qr[i][s] = (rowok && col < n) ? qp.Q(i, c, col) : 0.0;                             // KartLQR.cs:62
            Zr[i][s] = Qr[i][s];
            L.Zl[i][c][col] = Qr[i][s];
        }
        qv[i] = rowok ? qp.q(i, c) : 0.0;                                                     // :63
        eta[i] = qv[i];
        if (g == 0) L.eta[i][c] = eta[i];
    }
    singular = 0;
    double alpha[m], pc[m];
    X::sync();

    for (int t = horizon; t >= 0; t--) {                                                      // :64
        // ---------------- S1: rows 4 i .. 4 i + 3 of Z_i B_j and Z_i A (lane = (i, rr, j)) ----------------
        {
            const int i = g, rr = c >> 2, j = c & 3;
            if (i < NP && j < NP) {
                double z[4];
#pragma unroll
                for (int k = 0; k < 4; k++) z[k] = L.Zl[i][4 * i + rr][4 * j + k];
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    double s = 0.0;                                   // (Z_i B_j)[4 i + rr][b], :78/:82 Zs[i].Multiply(Bs[j])
#pragma unroll
                    for (int k = 0; k < 4; k++) s = fma64(z[k], L.Bb[j][k * 2 + b], s);
                    L.T1[i][rr][2 * j + b] = s;
                }
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    double s = 0.0;                                   // (Z_i A)[4 i + rr][4 j + cc], :89/:95 Zs[i].Multiply(A)
#pragma unroll
                    for (int k = 0; k < 4; k++) s = fma64(z[k], L.Ab[j][k * 4 + cc], s);
                    L.ZA[i][rr][4 * j + cc] = s;
                }
            }
        }
        X::sync();
        // ---------------- S2: column c of [LHS | RHSMat], and RHSVec (every lane) ----------------
        double col[m], sacc[m], bb[m], bv[m];
        {
            const int ci = (c >> 1) < NP ? (c >> 1) : NP - 1, cb = c & 1;   // LHS column c = 2 ci + cb (player ci, control cb)
#pragma unroll
            for (int j = 0; j < NP; j++) {
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const int row = 2 * j + a;
                    double v = 0.0;
                    if (c < m) {
                        double s = 0.0;                                                  // :78/:82 Bs[i]' * (Z_i B_j)
#pragma unroll
                        for (int k = 0; k < 4; k++) s = fma64(L.Bb[ci][k * 2 + a], L.T1[ci][k][2 * j + cb], s);
                        v = (ci == j) ? (L.Rb[ci][a * 2 + cb] + s) : s;                  // :78
                    }
                    double rb = 0.0;                                                     // :89/:95 Bs[j]'(Z_j A), column c
#pragma unroll
                    for (int k = 0; k < 4; k++) rb = fma64(L.Bb[j][k * 2 + a], rowok ? L.ZA[j][k][c] : 0.0, rb);
                    double rv = 0.0;                                                     // :96 Bs[j]' eta_j
#pragma unroll
                    for (int k = 0; k < 4; k++) rv = fma64(L.Bb[j][k * 2 + a], L.eta[j][4 * j + k], rv);
                    col[row] = v; sacc[row] = 0.0; bb[row] = rb; bv[row] = rv;
                }
            }
        }
        // ---------------- S3: LU (JAMA order) + forward elimination of the right-hand sides ----------------
        int sing = 0;
#pragma unroll
        for (int k = 0; k < m; k++) {
            int pk = k, sk = 0;
            if (c == k) {
                // finalize rows >= k of column k: col[i] -= s_i   (rows < k were finalized at their own step)
#pragma unroll
                for (int i = 0; i < m; i++)
                    if (i >= k) col[i] = col[i] - sacc[i];
                double best = fabs(col[k]);
#pragma unroll
                for (int i = 0; i < m; i++)
                    if (i > k && fabs(col[i]) > best) { best = fabs(col[i]); pk = i; }
                double ck = col[k];
#pragma unroll
                for (int i = 0; i < m; i++)
                    if (i > k && i == pk) { ck = col[i]; col[i] = col[k]; }
                col[k] = ck;
                if (ck == 0.0) sk = 1;
#pragma unroll
                for (int i = 0; i < m; i++)
                    if (i > k && ck != 0.0) col[i] = col[i] / ck;
            }
            const int p = X::readlane_i(pk, k);
            sing |= X::readlane_i(sk, k);
            double lm[m];
#pragma unroll
            for (int i = 0; i < m; i++) lm[i] = (i > k) ? X::readlane(col[i], k) : 0.0;
            if (p != k) {
                // row swap k <-> p in every other column, accumulators and right-hand sides (lane k did its own)
#pragma unroll
                for (int i = 0; i < m; i++)
                    if (i > k && i == p) {
                        double tmp;
                        if (c != k) { tmp = col[i]; col[i] = col[k]; col[k] = tmp; }
                        if (c != k) { tmp = sacc[i]; sacc[i] = sacc[k]; sacc[k] = tmp; }
                        tmp = bb[i]; bb[i] = bb[k]; bb[k] = tmp;
                        tmp = bv[i]; bv[i] = bv[k]; bv[k] = tmp;
                    }
            }
            if (c > k && c < m) {
                // column c > k: u[k] of this column becomes final, then accumulate s_i += L[i][k]*u[k]
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
        }
        // back substitution  U X = Y  (k descending); U[i][k] = col[i] of lane k
#pragma unroll
        for (int kk = 0; kk < m; kk++) {
            const int k = m - 1 - kk;
            const double ukk = X::readlane(col[k], k);
            bb[k] = bb[k] / ukk;
            bv[k] = bv[k] / ukk;
#pragma unroll
            for (int i = 0; i < m; i++)
                if (i < k) {
                    const double uik = X::readlane(col[i], k);
                    double temp = bb[k] * uik;
                    bb[i] = bb[i] - temp;
                    double tempv = bv[k] * uik;
                    bv[i] = bv[i] - tempv;
                }
        }
        if (sing) singular = 1;
        // ---------------- S4: P (column c), alpha (every lane); P also to LDS ----------------
#pragma unroll
        for (int i = 0; i < m; i++) {
            pc[i] = bb[i];
            alpha[i] = bv[i];
            if (g == 0) L.Pl[i][c] = bb[i];
        }
        // the value update of the last sweep feeds nothing (u0 reads this sweep's P and alpha): skipped — a quarter of the products at horizon 3
        if (t == 0) { X::sync(); break; }                                     // (P is in LDS)
        // ---------------- S5: F = A - sum_k B_k P_k (column c, all rows), beta = -sum_k B_k alpha_k (all rows) ----------------
        double Fcol[16], beta[16];
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int row = 4 * k + rr;
                if (k < NP) {
                    double tt = 0.0;                                          // :110 Bs[k] * P_k
                    tt = fma64(L.Bb[k][rr * 2 + 0], pc[2 * k + 0], tt);
                    tt = fma64(L.Bb[k][rr * 2 + 1], pc[2 * k + 1], tt);
                    const double acc = 0.0 + tt;                              // Aggregate seed (zero matrix) + B_k P_k
                    const double av = ((c >> 2) == k) ? L.Ab[k][rr * 4 + (c & 3)] : 0.0;
                    Fcol[row] = av - acc;
                    double tb = 0.0;                                          // :111 beta row (block k)
                    tb = fma64(L.Bb[k][rr * 2 + 0], alpha[2 * k + 0], tb);
                    tb = fma64(L.Bb[k][rr * 2 + 1], alpha[2 * k + 1], tb);
                    beta[row] = 0.0 - tb;
                } else { Fcol[row] = 0.0; beta[row] = 0.0; }
            }
        }
        double Freg[4];                                                       // F[4 s + g][c]: the B operand of both products
#pragma unroll
        for (int s = 0; s < 4; s++) {
            double f = Fcol[4 * s];
#pragma unroll
            for (int q = 1; q < 4; q++) f = (g == q) ? Fcol[4 * s + q] : f;
            Freg[s] = f;
        }
        X::sync();                                                            // P is in LDS
        // ---------------- S6: Z_i <- (Q_i + P_i'(R_i P_i)) + F'(Z_i F) on the matrix core (:113-116) ----------------
        lqm_d4 W[NP], O[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) { W[i] = lqm_d4{0.0, 0.0, 0.0, 0.0}; O[i] = lqm_d4{0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int i = 0; i < NP; i++) W[i] = X::mfma(Zr[i][s], Freg[s], W[i]);          // W = Z_i F: W[i][j] = W[g + 4 j][c]
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int i = 0; i < NP; i++) O[i] = X::mfma(W[i][s], Freg[s], O[i]);           // W' F: O[i][j] = (F' W)[c][4 j + g]
#pragma unroll
        for (int i = 0; i < NP; i++) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int colx = 4 * s + g;
                const double p0 = L.Pl[2 * i + 0][colx], p1 = L.Pl[2 * i + 1][colx];
                double rp0 = 0.0, rp1 = 0.0;                                  // (R_i P_i)[a][colx]
                rp0 = fma64(L.Rb[i][0], p0, rp0); rp0 = fma64(L.Rb[i][1], p1, rp0);
                rp1 = fma64(L.Rb[i][2], p0, rp1); rp1 = fma64(L.Rb[i][3], p1, rp1);
                double t2 = 0.0;                                              // (P_i'(R_i P_i))[c][colx]
                t2 = fma64(pc[2 * i + 0], rp0, t2);
                t2 = fma64(pc[2 * i + 1], rp1, t2);
                const double z = (Qr[i][s] + t2) + O[i][s];
                Zr[i][s] = z;
                L.Zl[i][c][colx] = z;
            }
        }
        X::sync();
        // ---------------- eta_i <- (q_i + P_i'(R_i alpha_i)) + F'(eta_i + Z_i beta) with the NEW Z_i (Q2, :117) ----------------
#pragma unroll
        for (int i = 0; i < NP; i++) {
            double zb = 0.0;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                const double2 zz = *reinterpret_cast<const double2*>(&L.Zl[i][c][k]);
                zb = fma64(zz.x, beta[k], zb);
                zb = fma64(zz.y, beta[k + 1], zb);
            }
            if (g == 0) L.vec[i][c] = eta[i] + zb;
        }
        X::sync();
#pragma unroll
        for (int i = 0; i < NP; i++) {
            double v3 = 0.0;
#pragma unroll
            for (int k = 0; k < 16; k++) v3 = fma64(Fcol[k], L.vec[i][k], v3);
            double ra0 = 0.0, ra1 = 0.0;
            ra0 = fma64(L.Rb[i][0], alpha[2 * i + 0], ra0);
            ra0 = fma64(L.Rb[i][1], alpha[2 * i + 1], ra0);
            ra1 = fma64(L.Rb[i][2], alpha[2 * i + 0], ra1);
            ra1 = fma64(L.Rb[i][3], alpha[2 * i + 1], ra1);
            double v2 = 0.0;
            v2 = fma64(pc[2 * i + 0], ra0, v2);
            v2 = fma64(pc[2 * i + 1], ra1, v2);
            eta[i] = (qv[i] + v2) + v3;
            if (g == 0) L.eta[i][c] = eta[i];
        }
        X::sync();
    }
    // :121-126 u0 = -P_0 x0 - alpha_0   (all lanes compute it redundantly)
#pragma unroll
    for (int a = 0; a < 2; a++) {
        double s = 0.0;
#pragma unroll
        for (int cc = 0; cc < n; cc++) s = fma64(-L.Pl[a][cc], L.x0[cc], s);
        u0[a] = s - alpha[a];
    }
}

}  // namespace hk
