// which CU does a workgroup run on?  HW_REG_HW_ID / HW_REG_XCC_ID per block of a 512-thread, 68 KB-LDS launch (the actor kernel's shape)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(512) void probe(unsigned* out, int spin)
{
    extern __shared__ float sm[];
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        out[2 * blockIdx.x + 1] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
    sm[threadIdx.x] = threadIdx.x;
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(127);
    __syncthreads();
    if (sm[(threadIdx.x + 1) & 511] < 0) out[0] = 0;
}
int main()
{
    const int nb = 2048;
    unsigned* d; hipMalloc(&d, nb * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 68 * 1024, 0, d, 20);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, int> keys;
    for (int b = 0; b < nb; b++) keys[((h[2 * b + 1] & 0xF) << 8) | ((h[2 * b] >> 8) & 0xFF)]++;
    printf("blocks %d distinct keys %zu\n", nb, keys.size());
    for (int b = 0; b < 24; b++) printf("block %d hw %08x xcc %08x\n", b, h[2 * b], h[2 * b + 1]);
    for (int b = 512; b < 520; b++) printf("block %d hw %08x xcc %08x\n", b, h[2 * b], h[2 * b + 1]);
    int k = 0; for (auto& kv : keys) { if (k++ < 12) printf("key %03x: %d blocks\n", kv.first, kv.second); }
    return 0;
}
