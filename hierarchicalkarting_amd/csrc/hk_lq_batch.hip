// hk_lq_batch.hip — hk_lq_solve_batch's kernels (lq_batch_kernel<1..8>, hk_lq_kernels.h) in their own translation unit:
// the 5..8-player instantiations are the slowest things to compile in the library and change rarely.
#include <hip/hip_runtime.h>
#include "../../include/hk.h"
#include "hk_lq_kernels.h"

namespace hk {

int lq_batch_launch(int batch, int N, const double* dA, const double* dB, const double* dQ, const double* dq, const double* dR,
                    const double* dx0, int horizon, double* du0, int* d_status, hipStream_t st)
{
    switch (N) {
#define HK_LQ_CASE(NP)                                                                                                   \
    case NP: {                                                                                                           \
        const int gpw = LqDims<NP>::GPW;                                                                                 \
        hipLaunchKernelGGL(lq_batch_kernel<NP>, dim3((batch + gpw - 1) / gpw), dim3(64), 0, st, batch, dA, dB, dQ, dq, dR, \
                           dx0, horizon, du0, d_status);                                                                 \
    } break;
#ifndef HK_LQ_BATCH_MODE
#define HK_LQ_BATCH_MODE 1      /* 3 / 4 players: 1 = lane-per-row core with the dense products on the matrix core (64 / n games per wave); 2 = one game per wave */
#endif
#if HK_LQ_BATCH_MODE == 1
#define HK_LQM_CASE(NP)                                                                                                  \
    case NP: {                                                                                                           \
        const int gpw = LqDims<NP>::GPW;                                                                                 \
        hipLaunchKernelGGL((lq_batch_kernel<NP, true>), dim3((batch + gpw - 1) / gpw), dim3(64), 0, st, batch, dA, dB, dQ, dq, dR, \
                           dx0, horizon, du0, d_status);                                                                 \
    } break;
#else
#define HK_LQM_CASE(NP)                                                                                                  \
    case NP:                                                                                                             \
        hipLaunchKernelGGL(lq_batch_mfma_kernel<NP>, dim3((batch + LQM_WPB - 1) / LQM_WPB), dim3(64 * LQM_WPB), 0, st, batch, dA, dB, dQ, dq, dR, \
                           dx0, horizon, du0, d_status);                                                                 \
        break;
#endif
        HK_LQ_CASE(1) HK_LQ_CASE(2) HK_LQM_CASE(3) HK_LQM_CASE(4) HK_LQ_CASE(5) HK_LQ_CASE(6) HK_LQ_CASE(7) HK_LQ_CASE(8)
#undef HK_LQ_CASE
#undef HK_LQM_CASE
    default: return HK_ERR_UNSUPPORTED;
    }
    return HK_OK;          // (a launch failure stays in hipGetLastError for the caller to report)
}

}  // namespace hk
