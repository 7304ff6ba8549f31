// Exhaustive check of csrc/hk_swish.h against include/hk_detmath.h's hk_swishf on the GPU: every fp32 bit pattern.
// Built and run by tests/test_swish_device.py (hipcc --offload-arch=gfx950 -ffp-contract=off, the library's own flags).
// Prints "mismatches <n> first <hex>"; NaN results count as equal to NaN results.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../hierarchicalkarting_amd/csrc/hk_swish.h"

__global__ void check(unsigned long long* bad, unsigned* first)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride) {
        const float s = __uint_as_float((unsigned)u);
        const float a = hk::swish(s), b = hk_swishf(s);
        const bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
        if (!same) { if (!mine) atomicMin(first, (unsigned)u); mine++; }
    }
    if (mine) atomicAdd(bad, mine);
}

int main()
{
    unsigned long long* bad; unsigned* first;
    if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&first, 4) != hipSuccess) { printf("no device\n"); return 2; }
    (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0xff, 4);
    hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, bad, first);
    unsigned long long hb = 0; unsigned hf = 0;
    if (hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
    (void)hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("mismatches %llu first %08x\n", hb, hf);
    return hb ? 1 : 0;
}
