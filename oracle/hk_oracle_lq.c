/*
 * hk_oracle_lq.c — CPU ORACLE (test infrastructure), LQ Nash core: components a1-a3 of SURVEY §8.
 * Dense, generic, line-by-line restatement; association order of every product follows the C# expression.
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: the HIP kernels reproduce these bits).
 */
#include <string.h>
#include <math.h>
#include "hk_oracle.h"
#include "../include/hk_detmath.h"

double hko_sin(double x) { return hk_sin(x); }
double hko_cos(double x) { return hk_cos(x); }
double hko_atan2(double y, double x) { return hk_atan2(y, x); }
double hko_exp(double x) { return hk_exp(x); }
double hko_log(double x) { return hk_log(x); }
float hko_expf_fast(float x) { return hk_expf_fast(x); }
float hko_swishf(float x) { return hk_swishf(x); }
void hko_sincos(double x, double* s, double* c) { hk_sincos(x, s, c); }
void hko_sincos_near0(double x, double* s, double* c) { hk_sincos_near0(x, s, c); }
/* the float entry points (Mathf.*: double evaluation to <= 2^-44, rounded once) */
float hko_sinf(float x) { return hk_sinf(x); }
float hko_cosf(float x) { return hk_cosf(x); }
void hko_sincosf(float x, float* s, float* c) { hk_sincosf(x, s, c); }
void hko_sincosf_near0(float x, float* s, float* c) { hk_sincosf_near0(x, s, c); }
float hko_atan2f(float y, float x) { return hk_atan2f(y, x); }
float hko_expf(float x) { return hk_expf(x); }

/* KartMPC index constants (AI/MPC/KartMPC.cs:15-18) */
enum { XI = 0, ZI = 1, VI = 2, HI = 3 };

/* ---- tiny dense helpers; C = A(ra x ca) * B(ca x cb) ----
 * ARITHMETIC CONTRACT (DESIGN.md "LQ arithmetic"): every inner product on the LQ path is a k-ascending chain of
 * fused multiply-adds seeded with +0.0, s = fma(a_k, b_k, s).  MathNet's managed loops round the product first;
 * the two differ by <= 0.5 ulp per term (~1e-16 relative, vs the 1e-4 tolerance), and the fused form is what
 * gfx950's v_fma_f64 executes at full rate, so the HIP kernels reproduce this oracle bit for bit.  Terms whose
 * factor is a structural zero leave s unchanged exactly, which lets the kernels skip them.
 * Build with -DHKO_NO_FMA to get the product-then-add form (tests bound the difference). */
#ifdef HKO_NO_FMA
#define HKO_FMA(a, b, c) ((a) * (b) + (c))
#else
#define HKO_FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif
static void mm(int ra, int ca, int cb, const double* A, const double* B, double* C)
{
    for (int i = 0; i < ra; i++)
        for (int j = 0; j < cb; j++) {
            double s = 0.0;
            for (int k = 0; k < ca; k++) s = HKO_FMA(A[i * ca + k], B[k * cb + j], s);
            C[i * cb + j] = s;
        }
}
/* C = A' * B where A is (ra x ca): result (ca x cb)  — MathNet TransposeThisAndMultiply */
static void mtm(int ra, int ca, int cb, const double* A, const double* B, double* C)
{
    for (int i = 0; i < ca; i++)
        for (int j = 0; j < cb; j++) {
            double s = 0.0;
            for (int k = 0; k < ra; k++) s = HKO_FMA(A[k * ca + i], B[k * cb + j], s);
            C[i * cb + j] = s;
        }
}

/* LinearizedBicycle.getA / getB — KartLQRDynamics.cs:40-62 */
void hko_bicycle_AB(double dt, const double initial[4], double* A, double* B)
{
    memset(A, 0, 16 * sizeof(double));
    memset(B, 0, 8 * sizeof(double));
    for (int i = 0; i < 4; i++) A[i * 4 + i] = 1.0;                      /* :44 SparseIdentity */
    A[XI * 4 + VI] = hk_cos(initial[HI]) * dt;                           /* :45 */
    A[ZI * 4 + VI] = hk_sin(initial[HI]) * dt;                           /* :46 */
    A[XI * 4 + HI] = -hk_sin(initial[HI]) * dt * initial[VI];            /* :47 */
    A[ZI * 4 + HI] = hk_cos(initial[HI]) * dt * initial[VI];             /* :48 */
    B[VI * 2 + 0] = dt;                                                  /* :58 */
    B[HI * 2 + 1] = dt;                                                  /* :59 */
}

/* LQRCheckpointReachAvoidCost — KartLQRCosts.cs:57-140 */
void hko_cost_build(int M, const double target[4], const double target_w[4], double control_w, const double* avoid_w,
                    const double* opp_target, const double* opp_w, double* Q, double* qv, double* R)
{
    int n = 4 + 4 * M;
    memset(Q, 0, sizeof(double) * n * n);
    /* :64-80 avoid terms; avoidWeights.Keys = {xIndex, zIndex} in insertion order (HKA:967-971) */
    for (int s = 0; s < 2; s++) {
        int cs = s; /* currStateIndex: xIndex=0 then zIndex=1; avoidIndices[...] are the same index (HKA:1013-1022) */
        int cur = 4;
        double total = 0.0;
        for (int i = 0; i < M; i++) {
            int t = cur + cs;
            double w = avoid_w[s * M + i];
            Q[cs * n + t] = w;          /* :73 */
            Q[t * n + cs] = w;          /* :74 */
            Q[t * n + t] = -w;          /* :75 */
            total -= w;                 /* :76 */
            cur += 4;
        }
        Q[cs * n + cs] = total;         /* :79 */
    }
    for (int k = 0; k < 4; k++) Q[k * n + k] += target_w[k];            /* :81-84 */
    {
        int cur = 4;
        for (int i = 0; i < M; i++) {                                    /* :86-94 ASSIGN, quirk Q4 */
            for (int k = 0; k < 3; k++) Q[(cur + k) * n + (cur + k)] = -opp_w[i * 3 + k];
            cur += 4;
        }
    }
    /* getQVec :103-127 */
    memset(qv, 0, sizeof(double) * n);
    for (int k = 0; k < 4; k++) qv[k] = -target[k];                      /* :109 Negate */
    for (int k = 0; k < 4; k++) qv[k] = qv[k] * target_w[k];             /* :112 */
    {
        int cur = 4;
        for (int i = 0; i < M; i++) {
            for (int k = 0; k < 4; k++) qv[cur + k] = opp_target[i * 4 + k];            /* :117 */
            for (int k = 0; k < 3; k++) qv[cur + k] = qv[cur + k] * -opp_w[i * 3 + k];  /* :121 */
            cur += 4;
        }
    }
    /* getRMatrix :132-140 SparseIdentity * controlWeight */
    R[0] = 1.0 * control_w; R[1] = 0.0; R[2] = 0.0; R[3] = 1.0 * control_w;
}

/*
 * MathNet.Numerics 4.15.0 Matrix<double>.Solve for a square (sparse-storage) matrix = UserLU (JAMA-style
 * left-looking Doolittle with partial pivoting, pivot = first strictly larger |.|), then L-forward / U-backward
 * substitution on the pivoted right-hand sides.  Restated from MathNet's published algorithm and checked against the IL of the
 * binary the reference ships (tools/mathnet_il.py, tests/test_mathnet_il.py: dispatch Solve -> LU() -> UserLU.Create for sparse
 * storage, accumulation s = s + a * b with k ascending, strict '>' pivot test, division by the pivot, substitution order).  The
 * products around it (MathNet's sparse multiply) stay unpinned; any backward-stable LU agrees to ~1e-14 here (cond <= ~8).
 * LU (m x m) row-major is overwritten by the factors; Bm (m x nb) row-major by the solution.
 */
static int lu_solve(int m, double* LU, int nb, double* Bm)
{
    int piv[HKO_MAX_M];
    double col[HKO_MAX_M];
    for (int i = 0; i < m; i++) piv[i] = i;
    for (int j = 0; j < m; j++) {
        for (int i = 0; i < m; i++) col[i] = LU[i * m + j];
        for (int i = 0; i < m; i++) {
            int kmax = i < j ? i : j;
            double s = 0.0;
            for (int k = 0; k < kmax; k++) s += LU[i * m + k] * col[k];
            col[i] -= s;
            LU[i * m + j] = col[i];
        }
        int p = j;
        for (int i = j + 1; i < m; i++)
            if (fabs(col[i]) > fabs(col[p])) p = i;
        if (p != j) {
            for (int k = 0; k < m; k++) {
                double t = LU[p * m + k]; LU[p * m + k] = LU[j * m + k]; LU[j * m + k] = t;
            }
            piv[j] = p;
        }
        if (LU[j * m + j] != 0.0)
            for (int i = j + 1; i < m; i++) LU[i * m + j] /= LU[j * m + j];
    }
    for (int j = 0; j < m; j++)
        if (LU[j * m + j] == 0.0) return HK_ERR_SINGULAR;
    /* UserLU.Solve: apply pivots, L*Y = P*B, U*X = Y */
    for (int i = 0; i < m; i++) {
        if (piv[i] == i) continue;
        int p = piv[i];
        for (int j = 0; j < nb; j++) {
            double t = Bm[p * nb + j]; Bm[p * nb + j] = Bm[i * nb + j]; Bm[i * nb + j] = t;
        }
    }
    for (int k = 0; k < m; k++)
        for (int i = k + 1; i < m; i++)
            for (int j = 0; j < nb; j++) {
                double temp = Bm[k * nb + j] * LU[i * m + k];
                Bm[i * nb + j] = Bm[i * nb + j] - temp;
            }
    for (int k = m - 1; k >= 0; k--) {
        for (int j = 0; j < nb; j++) Bm[k * nb + j] /= LU[k * m + k];
        for (int i = 0; i < k; i++)
            for (int j = 0; j < nb; j++) {
                double temp = Bm[k * nb + j] * LU[i * m + k];
                Bm[i * nb + j] = Bm[i * nb + j] - temp;
            }
    }
    return 0;
}

/* KartLQR.solveFeedbackLQR — KartLQR.cs:17-128 */
int hko_lq_solve(int N, const double* Ain, const double* Bin, const double* Qin, const double* qin, const double* Rin,
                 const double* x0, int horizon, double* u0, double* trace)
{
    if (N < 1 || N > HKO_MAX_PLAYERS) return HK_ERR_INVALID;
    const int n = 4 * N, m = 2 * N;                                       /* :22-23 */
    static __thread double A[HKO_MAX_N * HKO_MAX_N];
    static __thread double Bs[HKO_MAX_PLAYERS][HKO_MAX_N * 2];
    static __thread double Zs[HKO_MAX_PLAYERS][HKO_MAX_N * HKO_MAX_N];
    static __thread double etas[HKO_MAX_PLAYERS][HKO_MAX_N];
    static __thread double LHS[HKO_MAX_M * HKO_MAX_M];
    static __thread double RHS[HKO_MAX_M * (HKO_MAX_N + 1)];   /* [RHSMat | RHSVec]: solved with the same factors (:104-105) */
    static __thread double P[HKO_MAX_M * HKO_MAX_N], alpha[HKO_MAX_M];
    static __thread double F[HKO_MAX_N * HKO_MAX_N], beta[HKO_MAX_N];
    static __thread double T1[HKO_MAX_N * HKO_MAX_N], T2[HKO_MAX_N * HKO_MAX_N], T3[HKO_MAX_N * HKO_MAX_N];
    static __thread double v1[HKO_MAX_N], v2[HKO_MAX_N], v3[HKO_MAX_N];

    /* :33-37 A = blockdiag(A_i) */
    memset(A, 0, sizeof(double) * n * n);
    for (int i = 0; i < N; i++)
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) A[(4 * i + r) * n + (4 * i + c)] = Ain[i * 16 + r * 4 + c];
    /* :41-52 B_i = [0; ..; B_i^loc; ..; 0] (n x 2) */
    for (int i = 0; i < N; i++) {
        memset(Bs[i], 0, sizeof(double) * n * 2);
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 2; c++) Bs[i][(4 * i + r) * 2 + c] = Bin[i * 8 + r * 2 + c];
    }
    /* :62-63 terminal Z_i = Q_i, eta_i = q_i */
    for (int i = 0; i < N; i++) {
        memcpy(Zs[i], Qin + (size_t)i * n * n, sizeof(double) * n * n);
        memcpy(etas[i], qin + (size_t)i * n, sizeof(double) * n);
    }
    int tr = 0;
    for (int t = horizon; t >= 0; t--) {                                  /* :64 */
        /* :67-87 LHS: column block i = vstack_j( B_i' (Z_i B_j) [+ R_i if i == j] )   (quirk Q1) */
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                mm(n, n, 2, Zs[i], Bs[j], T1);          /* Zs[i].Multiply(Bs[j])  (n x 2) */
                mtm(n, 2, 2, Bs[i], T1, T2);            /* Bs[i]' * (.)           (2 x 2) */
                for (int a = 0; a < 2; a++)
                    for (int b = 0; b < 2; b++) {
                        double v = T2[a * 2 + b];
                        if (i == j) v = Rin[i * 4 + a * 2 + b] + v;      /* :78 R + B'ZB */
                        LHS[(2 * j + a) * m + (2 * i + b)] = v;
                    }
            }
        /* :89-98 RHSMat = vstack_i( B_i' (Z_i A) ), RHSVec = concat_i( B_i' eta_i ) */
        for (int i = 0; i < N; i++) {
            mm(n, n, n, Zs[i], A, T1);
            mtm(n, 2, n, Bs[i], T1, T2);                /* 2 x n */
            mtm(n, 2, 1, Bs[i], etas[i], v1);           /* 2 */
            for (int a = 0; a < 2; a++) {
                for (int c = 0; c < n; c++) RHS[(2 * i + a) * (n + 1) + c] = T2[a * n + c];
                RHS[(2 * i + a) * (n + 1) + n] = v1[a];
            }
        }
        /* :104-105 P = LHS.Solve(RHSMat); alpha = LHS.Solve(RHSVec) */
        int rc = lu_solve(m, LHS, n + 1, RHS);
        if (rc) return rc;
        for (int r = 0; r < m; r++) {
            for (int c = 0; c < n; c++) P[r * n + c] = RHS[r * (n + 1) + c];
            alpha[r] = RHS[r * (n + 1) + n];
        }
        if (trace) {
            memcpy(trace + tr, P, sizeof(double) * m * n); tr += m * n;
            memcpy(trace + tr, alpha, sizeof(double) * m); tr += m;
        }
        /* :110 F = A - sum_k B_k P_k ; :111 beta = - sum_k B_k alpha_k  (Aggregate from a zero seed) */
        memset(T3, 0, sizeof(double) * n * n);
        memset(v3, 0, sizeof(double) * n);
        for (int k = 0; k < N; k++) {
            mm(n, 2, n, Bs[k], P + (size_t)(2 * k) * n, T1);
            for (int e = 0; e < n * n; e++) T3[e] = T3[e] + T1[e];
            mm(n, 2, 1, Bs[k], alpha + 2 * k, v1);
            for (int e = 0; e < n; e++) v3[e] = v3[e] - v1[e];
        }
        for (int e = 0; e < n * n; e++) F[e] = A[e] - T3[e];
        for (int e = 0; e < n; e++) beta[e] = v3[e];
        for (int i = 0; i < N; i++) {                                     /* :113-119 */
            const double* Pi = P + (size_t)(2 * i) * n;                   /* 2 x n */
            const double* Ri = Rin + i * 4;
            /* :116 Z_i = Q_i + P_i'(R_i P_i) + F'(Z_i F) */
            mm(2, 2, n, Ri, Pi, T1);                    /* R P   (2 x n) */
            mtm(2, n, n, Pi, T1, T2);                   /* P'(RP) (n x n) */
            mm(n, n, n, Zs[i], F, T1);                  /* Z F */
            mtm(n, n, n, F, T1, T3);                    /* F'(ZF) */
            const double* Qi = Qin + (size_t)i * n * n;
            for (int e = 0; e < n * n; e++) Zs[i][e] = (Qi[e] + T2[e]) + T3[e];
            /* :117 eta_i = q_i + P_i'(R_i alpha_i) + F'(eta_i + Z_i beta)   — NEW Z_i (quirk Q2) */
            mm(2, 2, 1, Ri, alpha + 2 * i, v1);         /* R alpha (2) */
            mtm(2, n, 1, Pi, v1, v2);                   /* P'(R alpha) (n) */
            mm(n, n, 1, Zs[i], beta, v1);               /* Z beta */
            for (int e = 0; e < n; e++) v1[e] = etas[i][e] + v1[e];
            mtm(n, n, 1, F, v1, v3);
            const double* qi = qin + (size_t)i * n;
            for (int e = 0; e < n; e++) etas[i][e] = (qi[e] + v2[e]) + v3[e];
        }
    }
    /* :121-126 u0 = -P_0 * initial - alpha_0 */
    for (int a = 0; a < 2; a++) {
        double s = 0.0;
        for (int c = 0; c < n; c++) s = HKO_FMA(-P[a * n + c], x0[c], s);
        u0[a] = s - alpha[a];
    }
    return 0;
}
