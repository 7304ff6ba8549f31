// hk_ga8.hip — the env kernels for lane groups of 8 (the synthetic 8-agent configuration, BASELINE configs[4]), one translation
// unit (see hk_env_ga.h).
#include "hk_env_host.h"
#define HK_GA 8
#define HK_GA_NS g8
#include "hk_env_ga.h"
namespace hk { const GaOps& ga_ops_g8() { return g8::make_ops(); } }
