// hk_ga4.hip — the env kernels for lane groups of 4 (a quad per race instance, up to 4 agents: every reference scene and the
// headline path), one translation unit (see hk_env_ga.h).
#include "hk_env_host.h"
#define HK_GA 4
#define HK_GA_NS g4
#include "hk_env_ga.h"
namespace hk { const GaOps& ga_ops_g4() { return g4::make_ops(); } }
