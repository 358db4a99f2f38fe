// nw_path.hpp — global unit-cost pairwise alignment with path, standing in for
//   edlibAlign(q, ql, t, tl, edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_PATH, NULL, 0))
//   + edlibAlignmentToCigar(..., EDLIB_CIGAR_STANDARD)
// as called by reference src/overlap.cpp:205-224.  martinsos/edlib v1.2.7 is an
// un-vendored dependency of the reference (CMakeLists.txt:43-48), so this is a
// from-scratch implementation of its published behaviour: exact edit distance,
// and — what matters for byte identity — the SAME co-optimal path: plain
// traceback (prefer insertion, then deletion, then (mis)match, walking back from
// the end) for small problems, Hirschberg on the target axis with the smallest
// optimal query split above edlib's 1 MiB traceback-state threshold (SURVEY.md
// Appendix B; pinned by the PAF/MHAP goldens of test/racon_test.cpp).
// This stays on the HOST: BASELINE.json keeps the breakpoint pre-alignment on the CPU.
#pragma once
#include <cstdint>
#include <string>

namespace racon {
namespace nwpath {

// CIGAR (M/I/D, run-length encoded) of query (rows) against target (columns).
std::string align_cigar(const char* query, uint32_t query_length, const char* target, uint32_t target_length);

// Plain global edit distance (edlibAlign with edlibDefaultAlignConfig(), test/racon_test.cpp:14-23).
uint64_t edit_distance(const char* query, uint64_t query_length, const char* target, uint64_t target_length);

}  // namespace nwpath
}  // namespace racon
