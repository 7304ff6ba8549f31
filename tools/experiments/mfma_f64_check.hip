// Is v_mfma_f64_16x16x4_f64 bit-for-bit a k-ascending fma chain seeded with C?  (lane l supplies A[l&15][k = l>>4] and
// B[k = l>>4][l&15]; D[row = (l>>4) + 4 j][col = l&15] in result j = 0..3: cdna_hip_programming.md §3.)
// Compares D with  fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, c))))  on random operands with a wide exponent spread
// (cancellation makes any other association or a wider accumulator visible), chained over four K-steps like the solver does,
// and times a dependent / independent stream of them.   hipcc --offload-arch=gfx950 -O2 mfma_f64_check.hip -o mfma_f64_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_check(const double* A /*[16][16] row-major M x K*/, const double* B /*[16][16] K x N*/, const double* C /*[16][16]*/, double* D)
{
    const int l = threadIdx.x, g = l >> 4, c = l & 15;
    d4 acc;
    for (int j = 0; j < 4; j++) acc[j] = C[(g + 4 * j) * 16 + c];
    for (int s = 0; s < 4; s++) {
        const double a = A[c * 16 + 4 * s + g];          // A[M = c][k = 4 s + g]
        const double b = B[(4 * s + g) * 16 + c];        // B[k = 4 s + g][N = c]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int j = 0; j < 4; j++) D[(g + 4 * j) * 16 + c] = acc[j];
}

__global__ void k_time(double* out, int iters, int indep)
{
    const int l = threadIdx.x;
    d4 acc[4];
    for (int q = 0; q < 4; q++) for (int j = 0; j < 4; j++) acc[q][j] = (double)(l + q + j);
    const double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (indep) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[3], 0, 0, 0);
        } else {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int q = 0; q < 4; q++) for (int j = 0; j < 4; j++) s += acc[q][j];
    out[blockIdx.x * 64 + l] = s;
    if (l == 0 && blockIdx.x == 0) out[64 * gridDim.x] = (double)(t1 - t0);
}

int main()
{
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    std::uniform_int_distribution<int> ex(-30, 30);
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    long long bad = 0, total = 0;
    for (int trial = 0; trial < 2000; trial++) {
        std::vector<double> A(256), B(256), C(256), D(256);
        for (int i = 0; i < 256; i++) {
            A[i] = std::ldexp(u(rng), ex(rng)); B[i] = std::ldexp(u(rng), ex(rng)); C[i] = (trial & 1) ? 0.0 : std::ldexp(u(rng), ex(rng));
            if (trial % 7 == 3 && (i % 5) == 0) A[i] = 0.0;                // structural zeros, as in the solver's F
        }
        hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
        for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) {
            double s = C[r * 16 + c];
            for (int k = 0; k < 16; k++) s = std::fma(A[r * 16 + k], B[k * 16 + c], s);
            total++;
            if (std::memcmp(&s, &D[r * 16 + c], 8) != 0) { if (bad < 5) std::printf("mismatch trial %d (%d,%d): chain %a mfma %a\n", trial, r, c, s, D[r * 16 + c]); bad++; }
        }
    }
    std::printf("mfma_f64_16x16x4: %lld of %lld outputs differ from the k-ascending fma chain\n", bad, total);
    double* dout; hipMalloc(&dout, (64 * 1024 + 1) * 8);
    for (int indep = 0; indep < 2; indep++) {
        for (int blocks : {1, 1024}) {
            hipLaunchKernelGGL(k_time, dim3(blocks), dim3(64), 0, 0, dout, 1000, indep);
            hipDeviceSynchronize();
            double cyc; hipMemcpy(&cyc, dout + 64 * blocks, 8, hipMemcpyDeviceToHost);
            std::printf("%s, %4d waves: %.1f readcyclecounter ticks per MFMA (4000 MFMAs)\n", indep ? "4 independent accumulators" : "1 dependent accumulator  ", blocks, cyc / 4000.0);
        }
    }
    {   // wall-clock rate with every SIMD busy: 4096 waves x 4000 MFMAs
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_time, dim3(1024), dim3(64), 0, 0, dout, 1000, 1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_time, dim3(1024), dim3(64), 0, 0, dout, 4000, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::printf("1024 waves x 16000 MFMAs in %.3f ms = %.1f TFLOP/s (2048 flop per MFMA)\n", ms, 1024.0 * 16000 * 2048 / (ms * 1e-3) / 1e12);
    }
    return bad ? 1 : 0;
}
