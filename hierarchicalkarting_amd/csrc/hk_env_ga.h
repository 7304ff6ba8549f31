// hk_env_ga.h — the env kernels for ONE lane-group width GA (lanes per race instance).  Included once per width by
// hk_env_kernels.h with HK_GA (4 or 8) and HK_GA_NS (g4 / g8) defined; everything below lands in namespace hk::HK_GA_NS.
//   GA = 4: up to 4 agents per env (every reference scene; the headline path) — a quad of lanes per env.
//   GA = 8: the synthetic 8-agent configuration (BASELINE configs[4]; the reference has no such scene) — 8 lanes per env,
//           games of up to 8 players, 8 karts in the planner's discrete game.
// Sizes that follow GA are compile-time (register arrays, unrolled kart loops, named game-state members), which is why this is a
// second compilation of the same sources and not a run-time parameter: the 4-agent kernels stay exactly what they were.
// (No include guard on purpose.)
#if !defined(HK_GA) || !defined(HK_GA_NS)
#error "hk_env_ga.h needs HK_GA and HK_GA_NS"
#endif
#include <utility>
#include <type_traits>
#include "hk_env_device.h"
#include "hk_lq_core.h"
#ifndef HK_HOST_EMU
#include "hk_lq_mfma.h"
#endif

namespace hk { namespace HK_GA_NS {

constexpr int GA = HK_GA;
static_assert(GA == 4 || GA == 8, "lane groups of 4 or 8");

// value of lane q of the calling lane's group
__device__ __forceinline__ float quad_get(float v, int q) { return __shfl(v, (threadIdx.x & ~(GA - 1)) | q, 64); }
__device__ __forceinline__ uint32_t quad_get(uint32_t v, int q) { return (uint32_t)__shfl((int)v, (threadIdx.x & ~(GA - 1)) | q, 64); }
__device__ __forceinline__ int quad_get(int v, int q) { return __shfl(v, (threadIdx.x & ~(GA - 1)) | q, 64); }
// The same with a compile-time lane: for a quad this is one DPP move (quad_perm [J, J, J, J]) instead of a ds_bpermute round
// trip through the LDS crossbar — the per-tick kart-vs-kart loops broadcast a dozen values per kart.  Every lane of the group
// must be active (the callers are group-uniform); an 8-lane group has no single-instruction broadcast and keeps the shuffle.
template <int J> __device__ __forceinline__ int group_get(int v)
{
    if constexpr (GA == 4) return __builtin_amdgcn_update_dpp(0, v, J * 0x55, 0xF, 0xF, true);
    else return __shfl(v, (threadIdx.x & ~(GA - 1)) | J, 64);
}
template <int J> __device__ __forceinline__ uint32_t group_get(uint32_t v) { return (uint32_t)group_get<J>((int)v); }
template <int J> __device__ __forceinline__ float group_get(float v) { return __builtin_bit_cast(float, group_get<J>(__builtin_bit_cast(int, v))); }
// f(std::integral_constant<int, j>) for j = 0 .. GA - 1, unrolled
template <class F, int... Js> __device__ __forceinline__ void for_lanes_impl(F&& f, std::integer_sequence<int, Js...>) { (f(std::integral_constant<int, Js>{}), ...); }
template <class F> __device__ __forceinline__ void for_each_lane(F&& f) { for_lanes_impl(f, std::make_integer_sequence<int, GA>{}); }
// minimum over the lanes of the group (every lane gets the result)
__device__ __forceinline__ int group_min(int v)
{
    int o = __shfl_xor(v, 1, 64); v = o < v ? o : v;
    o = __shfl_xor(v, 2, 64); v = o < v ? o : v;
    if (GA > 4) { o = __shfl_xor(v, 4, 64); v = o < v ? o : v; }
    return v;
}
// OR over the lanes of the group (every lane gets the result)
__device__ __forceinline__ int group_or(int v)
{
    v |= __shfl_xor(v, 1, 64); v |= __shfl_xor(v, 2, 64);
    if (GA > 4) v |= __shfl_xor(v, 4, 64);
    return v;
}

} }  // namespace hk::HK_GA_NS

#include "hk_env_mcts.h"
#include "hk_env_training.h"
#include "hk_env_reward.h"
#include "hk_env_step.h"
#include "hk_env_solve.h"
#include "hk_env_run.h"
#ifndef HK_HOST_EMU          // (the host emulation of the tick kernel stops here: tests/env_run_host_check.cpp)
#include "hk_env_observe.h"
#include "hk_env_launch.h"
#endif
