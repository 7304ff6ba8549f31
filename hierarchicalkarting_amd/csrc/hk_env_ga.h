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
#include "hk_env_device.h"
#include "hk_lq_core.h"

namespace hk { namespace HK_GA_NS {

constexpr int GA = HK_GA;
static_assert(GA == 4 || GA == 8, "lane groups of 4 or 8");

// value of lane q of the calling lane's group
__device__ __forceinline__ float quad_get(float v, int q) { return __shfl(v, (threadIdx.x & ~(GA - 1)) | q, 64); }
__device__ __forceinline__ uint32_t quad_get(uint32_t v, int q) { return (uint32_t)__shfl((int)v, (threadIdx.x & ~(GA - 1)) | q, 64); }
__device__ __forceinline__ int quad_get(int v, int q) { return __shfl(v, (threadIdx.x & ~(GA - 1)) | q, 64); }
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
#include "hk_env_observe.h"
#include "hk_env_launch.h"
