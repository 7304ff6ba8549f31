// hk_env_kernels.h — batched kart environment on gfx950 (placeholder until the env kernels land).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/hk.h"

namespace hk {

struct EnvDevice {
    hk_agent_state* agents = nullptr;
    hk_env_state* envs = nullptr;
    hk_episode_result* results = nullptr;
    hk_lq_debug* lq_debug = nullptr;
    float* obs = nullptr;
    float* act_steer = nullptr;
    int32_t* act_branch = nullptr;
};

inline int env_create(const hk_config&, std::vector<hk_section>&, std::vector<hk_wall_seg>&, EnvDevice&, hipStream_t, std::string& err)
{ err = "environment kernels not built yet"; return HK_ERR_UNSUPPORTED; }
inline void env_destroy(EnvDevice&) {}
inline int env_reset(EnvDevice&, const hk_config&, const int32_t*, int, int, hipStream_t, std::string& err) { err = "n/a"; return HK_ERR_UNSUPPORTED; }
inline int env_launch_solve(EnvDevice&, const hk_config&, hipStream_t, std::string& err) { err = "n/a"; return HK_ERR_UNSUPPORTED; }
inline int env_launch_step(EnvDevice&, const hk_config&, hipStream_t, std::string& err) { err = "n/a"; return HK_ERR_UNSUPPORTED; }
inline int env_launch_observe(EnvDevice&, const hk_config&, hipStream_t, std::string& err) { err = "n/a"; return HK_ERR_UNSUPPORTED; }

}  // namespace hk
