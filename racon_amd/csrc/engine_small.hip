// engine_small.hip -- the small-window kernel (poa_small.hpp: one wave per window, graph in LDS) in a translation unit of
// its own: four waves per SIMD, its own register budget (see engine_deep.hip for why instances are compiled apart).
#define RCN_DEEP_TU 1
#define RCN_SMALL_TU 1
#include "poa_small.hpp"
