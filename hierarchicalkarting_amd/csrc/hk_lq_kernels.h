// hk_lq_kernels.h — batched KartLQR.solveFeedbackLQR over explicit (A, B, Q, q, R, x0) inputs (hk_lq_solve_batch).
#pragma once
#include "hk_lq_core.h"

namespace hk {

// dense Q_i / q_i straight from global memory (generic costs, as the reference API allows)
struct QDenseP {
    const double* Qg;   // [N][n][n] of this game
    const double* qg;   // [N][n]
    int n;
    __device__ double Q(int i, int r, int c) const { return Qg[((size_t)i * n + r) * n + c]; }
    __device__ double q(int i, int r) const { return qg[(size_t)i * n + r]; }
};

// one wave (64 threads) per block, 4 games per wave
__global__ __launch_bounds__(64) void lq_batch_kernel(int batch, int N, const double* __restrict__ A,
                                                       const double* __restrict__ B, const double* __restrict__ Q,
                                                       const double* __restrict__ q, const double* __restrict__ R,
                                                       const double* __restrict__ x0, int horizon,
                                                       double* __restrict__ u0_out, int* __restrict__ status)
{
    __shared__ LqGroupLds lds[4];
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, r = lane & 15;
    const long game = (long)blockIdx.x * 4 + g;
    const bool live = game < batch;
    const int n = 4 * N;
    LqGroupLds& L = lds[g];
    const int Ng = live ? N : 0;
#pragma unroll
    for (int i = 0; i < LQ_MAXP; i++) {
        const bool ok = live && i < N;
        L.Ab[i][r] = ok ? A[((size_t)game * N + i) * 16 + r] : 0.0;
        if (r < 8) L.Bb[i][r] = ok ? B[((size_t)game * N + i) * 8 + r] : 0.0;
        if (r < 4) L.Rb[i][r] = ok ? R[((size_t)game * N + i) * 4 + r] : 0.0;
    }
    L.x0[r] = (live && r < n) ? x0[(size_t)game * n + r] : 0.0;
    __syncthreads();
    QDenseP qp;
    qp.Qg = Q + (live ? (size_t)game * N * n * n : 0);
    qp.qg = q + (live ? (size_t)game * N * n : 0);
    qp.n = n;
    double u0[2];
    int singular;
    lq_solve_group(r, Ng, N, L, qp, horizon, u0, singular);
    if (live && r == 0) {
        u0_out[game * 2 + 0] = u0[0];
        u0_out[game * 2 + 1] = u0[1];
        if (singular) atomicOr(status, 1);
    }
}

}  // namespace hk
