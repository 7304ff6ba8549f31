// Calibration of SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU ("lanes active per vector instruction") on gfx950: chains of fp32 / fp64 FMAs
// with all 64 lanes, and with 16 of 64 lanes, switched on.  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -- ./valu_lanes_calib
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int LANES>
__global__ void chain(T* out, T a, T b, int iters)
{
    T x = (T)threadIdx.x, y = (T)1, z = (T)2, w = (T)3;
    if ((int)(threadIdx.x & 63) < LANES) {
        for (int i = 0; i < iters; i++) {
            x = __builtin_fma(x, a, b); y = __builtin_fma(y, a, b); z = __builtin_fma(z, a, b); w = __builtin_fma(w, a, b);
            x = __builtin_fma(x, a, b); y = __builtin_fma(y, a, b); z = __builtin_fma(z, a, b); w = __builtin_fma(w, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w;
}
template <int LANES>
__global__ void chain32(float* out, float a, float b, int iters)
{
    float x = (float)threadIdx.x, y = 1.0f, z = 2.0f, w = 3.0f;
    if ((int)(threadIdx.x & 63) < LANES) {
        for (int i = 0; i < iters; i++) {
            x = __builtin_fmaf(x, a, b); y = __builtin_fmaf(y, a, b); z = __builtin_fmaf(z, a, b); w = __builtin_fmaf(w, a, b);
            x = __builtin_fmaf(x, a, b); y = __builtin_fmaf(y, a, b); z = __builtin_fmaf(z, a, b); w = __builtin_fmaf(w, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w;
}
// the same question for instructions that are not FMAs: compares + selects, integer ops, moves through DPP, divisions / square roots
__global__ void mix_sel(float* out, float a, float b, int iters)
{
    float x = (float)threadIdx.x, y = 1.0f, z = 2.0f, w = 3.0f;
    for (int i = 0; i < iters; i++) {
        x = x > a ? y : x + b; y = y < b ? z : y - a; z = z > w ? x : z + a; w = w < x ? y : w - b;
        x = x > y ? z : x + a; y = y < z ? w : y - b; z = z > x ? w : z + b; w = w < y ? x : w - a;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w;
}
__global__ void mix_int(int* out, int a, int b, int iters)
{
    int x = threadIdx.x, y = 1, z = 2, w = 3;
    for (int i = 0; i < iters; i++) {
        x = (x ^ a) + (y >> 1); y = (y & b) | (z << 1); z = (z + w) ^ x; w = (w - x) & 0xffff;
        x = x > y ? x - y : y - x; y = __builtin_amdgcn_update_dpp(0, y, 0x55, 0xF, 0xF, true) + z;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w;
}
__global__ void mix_div(float* out, float a, float b, int iters)
{
    float x = (float)threadIdx.x + 1.0f, y = 1.5f;
    for (int i = 0; i < iters; i++) { x = x / (y + a) + b; y = sqrtf(y * x + a); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
int main()
{
    float* f; double* d;
    hipMalloc(&f, 1024 * 256 * sizeof(float)); hipMalloc(&d, 1024 * 256 * sizeof(double));
    hipLaunchKernelGGL((chain32<64>), dim3(1024), dim3(256), 0, 0, f, 1.0001f, 0.5f, 2000);
    hipLaunchKernelGGL((chain32<16>), dim3(1024), dim3(256), 0, 0, f, 1.0001f, 0.5f, 2000);
    hipLaunchKernelGGL((chain<double, 64>), dim3(1024), dim3(256), 0, 0, d, 1.0001, 0.5, 2000);
    hipLaunchKernelGGL((chain<double, 16>), dim3(1024), dim3(256), 0, 0, d, 1.0001, 0.5, 2000);
    hipLaunchKernelGGL(mix_sel, dim3(1024), dim3(256), 0, 0, f, 1.0001f, 0.5f, 2000);
    hipLaunchKernelGGL(mix_int, dim3(1024), dim3(256), 0, 0, (int*)f, 12345, 0xff0f, 2000);
    hipLaunchKernelGGL(mix_div, dim3(1024), dim3(256), 0, 0, f, 1.0001f, 0.5f, 500);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
