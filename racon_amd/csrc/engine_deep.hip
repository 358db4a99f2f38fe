// engine_deep.hip -- the instance of the window-consensus kernel for work-groups that have a CU to themselves
// (poa_window_kernel2_deep: the deep launch of engine.hip's split; poa_kernel2.hpp says why it is a translation unit of
// its own).  No host code: engine.hip launches it.
#include <hip/hip_runtime.h>
#define RCN_DEEP_TU 1
#include "poa_kernel2.hpp"
