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
        HK_LQ_CASE(1) HK_LQ_CASE(2) HK_LQ_CASE(3) HK_LQ_CASE(4) HK_LQ_CASE(5) HK_LQ_CASE(6) HK_LQ_CASE(7) HK_LQ_CASE(8)
#undef HK_LQ_CASE
    default: return HK_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? HK_OK : HK_ERR_HIP;
}

}  // namespace hk
