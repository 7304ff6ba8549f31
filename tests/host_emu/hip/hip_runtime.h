// Test infrastructure: a HOST stand-in for <hip/hip_runtime.h>, just enough of it to compile the env kernels' device code with g++
// (tests/env_run_host_check.cpp, tests/test_env_run_host.py) and run it under -fsanitize=address,undefined.
// One lane group (a quad: 4 lanes = the 4 agents of one race instance) is 4 host threads; the "wave" and the "block" are that quad.
// Every cross-lane primitive is a barrier-guarded exchange among the 4 threads, so a primitive reached by a PART of the lane group —
// the hazard class a DPP quad_perm has on the GPU, where a switched-off lane reads as 0 — does not return: the barrier times out
// and the run fails.
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __align__(x) alignas(x)
#define __restrict__
#define HK_HOST_EMU 1
#define __shared__ static
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))

struct hk_emu_dim3 { unsigned x, y, z; };
extern thread_local hk_emu_dim3 threadIdx;
extern hk_emu_dim3 blockIdx, blockDim, gridDim;
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
using std::isinf; using std::isnan;
typedef int hipStream_t;

namespace hk_emu {
#ifndef HK_EMU_LANES
#define HK_EMU_LANES 4          /* the lanes of a group: 4, or 8 for the hk::g8 build of the harness (-DHK_GA=8 -DHK_GA_NS=g8 -DHK_EMU_LANES=8) */
#endif
constexpr int LANES = HK_EMU_LANES;
// a reusable barrier for LANES threads that reports a hang instead of deadlocking
struct Barrier {
    // sense-reversing spin barrier (the lanes meet thousands of times per tick: a condition variable's wake-up latency dominated the run)
    std::atomic<int> count{0};
    std::atomic<unsigned> gen{0};
    void wait(const char* what)
    {
        const unsigned g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == LANES) {
            count.store(0, std::memory_order_relaxed);
            gen.store(g + 1, std::memory_order_release);
            return;
        }
        unsigned long spins = 0;
        std::chrono::steady_clock::time_point t0;
        while (gen.load(std::memory_order_acquire) == g) {
            if (++spins < 200) continue;
            std::this_thread::yield();
            if (spins == 200) t0 = std::chrono::steady_clock::now();
            else if ((spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
                std::fprintf(stderr, "HK_EMU: cross-lane primitive '%s' reached by only part of the lane group (lane %u waited)\n", what, threadIdx.x);
                std::fflush(stderr);
                std::_Exit(3);
            }
        }
    }
};
extern Barrier bar;
extern uint64_t slot[LANES];
extern unsigned char* dyn_shared;          // the block's dynamic LDS (HK_DYN_SHARED)
template <class T> inline uint64_t bits(T v) { uint64_t b = 0; std::memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from(uint64_t b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }
// every lane publishes v, then reads lane `src` (its own choice)
template <class T> inline T exchange(T v, int src, const char* what)
{
    slot[threadIdx.x & (LANES - 1)] = bits(v);
    bar.wait(what);
    const T r = from<T>(slot[src & (LANES - 1)]);
    bar.wait(what);
    return r;
}
}  // namespace hk_emu

inline void __syncthreads() { hk_emu::bar.wait("__syncthreads"); }
inline int __syncthreads_or(int p)
{
    int r = 0;
    for (int l = 0; l < hk_emu::LANES; l++) r |= hk_emu::exchange<int>(p, l, "__syncthreads_or");
    return r;
}
inline unsigned long long __ballot(int p)
{
    unsigned long long m = 0;
    for (int l = 0; l < hk_emu::LANES; l++) if (hk_emu::exchange<int>(p ? 1 : 0, l, "__ballot")) m |= 1ull << l;
    return m;
}
template <class T> inline T __shfl(T v, int src, int = 64) { return hk_emu::exchange<T>(v, src, "__shfl"); }
template <class T> inline T __shfl_xor(T v, int mask, int = 64) { return hk_emu::exchange<T>(v, (int)(threadIdx.x ^ (unsigned)mask), "__shfl_xor"); }
// DPP quad_perm [J, J, J, J] (ctrl = J * 0x55): the value of lane J of the quad
inline int __builtin_amdgcn_update_dpp(int, int v, int ctrl, int, int, bool) { return hk_emu::exchange<int>(v, ctrl & 3, "update_dpp quad_perm"); }
inline void __builtin_amdgcn_fence(int, const char*) { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __builtin_amdgcn_wave_barrier() { hk_emu::bar.wait("wave_barrier"); }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline int atomicAnd(int* p, int v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
inline int atomicMax(int* p, int v) { int o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
inline float atomicAdd(float* p, float v) { float o, n; do { o = *p; n = o + v; } while (!__atomic_compare_exchange(reinterpret_cast<int*>(p), reinterpret_cast<int*>(&o), reinterpret_cast<int*>(&n), false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)); return o; }
inline unsigned long long __builtin_readcyclecounter_emu() { return 0; }
