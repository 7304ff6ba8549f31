// hk_lq_kernels.h — batched KartLQR.solveFeedbackLQR over explicit (A, B, Q, q, R, x0) inputs (hk_lq_solve_batch).
#pragma once
#include "hk_lq_core.h"
#include "hk_lq_mfma.h"

namespace hk {

// dense Q_i / q_i straight from global memory (generic costs, as the reference API allows)
struct QDenseP {
    const double* Qg;   // [N][n][n] of this game
    const double* qg;   // [N][n]
    int n;
    __device__ double Q(int i, int r, int c) const { return Qg[((size_t)i * n + r) * n + c]; }
    __device__ double q(int i, int r) const { return qg[(size_t)i * n + r]; }
};

// one wave (64 threads) per block, 64/(4*NP) games per wave
template <int NP, bool MFMA = false>
__global__ __launch_bounds__(64) void lq_batch_kernel(int batch, const double* __restrict__ A, const double* __restrict__ B,
                                                       const double* __restrict__ Q, const double* __restrict__ q,
                                                       const double* __restrict__ R, const double* __restrict__ x0, int horizon,
                                                       double* __restrict__ u0_out, int* __restrict__ status)
{
    constexpr int n = LqDims<NP>::n, GPW = LqDims<NP>::GPW, SLOTS = LqDims<NP>::SLOTS;
    __shared__ LqGameLds<NP> lds[SLOTS];
    const int lane = threadIdx.x & 63;
    const int gs = lane / n, r = lane % n;
    long game = (long)blockIdx.x * GPW + gs;
    const bool live = gs < GPW && game < batch;
    if (!live) game = batch - 1;                 // idle slots recompute the last game and discard it
    LqGameLds<NP>& L = lds[gs];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        for (int e = r; e < 16; e += n) L.Ab[i][e] = A[((size_t)game * NP + i) * 16 + e];
        for (int e = r; e < 8; e += n) L.Bb[i][e] = B[((size_t)game * NP + i) * 8 + e];
        if (r < 4) L.Rb[i][r] = R[((size_t)game * NP + i) * 4 + r];
    }
    L.x0[r] = x0[(size_t)game * n + r];
    __syncthreads();
    QDenseP qp;
    qp.Qg = Q + (size_t)game * NP * n * n;
    qp.qg = q + (size_t)game * NP * n;
    qp.n = n;
    double u0[2];
    int singular;
    lq_solve_game<NP, QDenseP, false, LqBlockSync, MFMA>(r, L, qp, horizon, u0, singular, lds);
    if (live && r == 0) {
        u0_out[game * 2 + 0] = u0[0];
        u0_out[game * 2 + 1] = u0[1];
        if (singular) atomicOr(status, 1);
    }
}

// 3 and 4 players: one game per wave on the fp64 matrix core (hk_lq_mfma.h); LQM_WPB waves (games) per workgroup, each with its own
// LDS slice, no workgroup barrier after the staging
constexpr int LQM_WPB = 4;
#ifndef HK_LQM_OCC
#define HK_LQM_OCC 2
#endif
template <int NP>
__global__ __launch_bounds__(64 * LQM_WPB, HK_LQM_OCC) void lq_batch_mfma_kernel(int batch, const double* __restrict__ A, const double* __restrict__ B,
                                                                      const double* __restrict__ Q, const double* __restrict__ q,
                                                                      const double* __restrict__ R, const double* __restrict__ x0, int horizon,
                                                                      double* __restrict__ u0_out, int* __restrict__ status)
{
    constexpr int n = 4 * NP;
    __shared__ LqMfmaLds<NP> lds[LQM_WPB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long game = (long)blockIdx.x * LQM_WPB + wave;
    const bool live = game < batch;
    if (!live) game = batch - 1;                 // idle waves recompute the last game and discard it
    LqMfmaLds<NP>& L = lds[wave];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        if (lane < 16) L.Ab[i][lane] = A[((size_t)game * NP + i) * 16 + lane];
        if (lane < 8) L.Bb[i][lane] = B[((size_t)game * NP + i) * 8 + lane];
        if (lane < 4) L.Rb[i][lane] = R[((size_t)game * NP + i) * 4 + lane];
    }
    if (lane < 16) L.x0[lane] = lane < n ? x0[(size_t)game * n + lane] : 0.0;
    LqmDev::sync();
    QDenseP qp;
    qp.Qg = Q + (size_t)game * NP * n * n;
    qp.qg = q + (size_t)game * NP * n;
    qp.n = n;
    double u0[2];
    int singular;
    lq_solve_game_mfma<NP, QDenseP, LqmDev>(lane, L, qp, horizon, u0, singular);
    if (live && lane == 0) {
        u0_out[game * 2 + 0] = u0[0];
        u0_out[game * 2 + 1] = u0[1];
        if (singular) atomicOr(status, 1);
    }
}

}  // namespace hk
