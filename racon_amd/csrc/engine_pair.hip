// engine_pair.hip -- the device pairwise aligner (pair_align.hpp: k_pair_align, k_ops_breaking_points) in a translation unit of its
// own (see engine_deep.hip for why instances are compiled apart; here it is compile time: the bit-vector passes are inlined in four
// carry variants per symbol-plane count).  No host code: engine.hip (window_build.hpp) launches the kernels.
#include <hip/hip_runtime.h>
#include <cstdint>
#define RCN_PAIR_TU 1
#include "pair_align.hpp"
