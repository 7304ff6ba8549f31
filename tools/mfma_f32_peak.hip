// What the f32 matrix pipe of this GPU sustains: waves that do nothing but v_mfma_f32_32x32x2_f32 on NACC independent
// accumulators, WPS waves per SIMD on every CU.  The actor's roofline fraction is quoted against the guide's 157.3 TFLOP/s;
// this prints the rate a kernel with no other instruction reaches on the box at hand (clock under load included).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f32_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void spin(float* out, int iters, float a, float b)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = (float)(i + r);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run(int wps, const char* label)
{
    const int blocks = 256 * wps;                     // 256 threads = 4 waves = one per SIMD; wps blocks per CU
    float* out;
    if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) { printf("no device\n"); return; }
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<NACC>, dim3(blocks), dim3(256), 0, 0, out, 2000, 1.0f, 0.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(spin<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.0f);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * (double)iters * 8 * NACC * (32.0 * 32 * 2 * 2);
    printf("%s: %d accumulators, %d waves per SIMD: %.1f TFLOP/s (%.2f ms)\n", label, NACC, wps, flop / ms * 1e-9, ms);
    (void)hipFree(out);
}

int main()
{
    run<2>(1, "f32 32x32x2"); run<2>(2, "f32 32x32x2"); run<2>(4, "f32 32x32x2"); run<4>(2, "f32 32x32x2"); run<1>(4, "f32 32x32x2");
    return 0;
}
