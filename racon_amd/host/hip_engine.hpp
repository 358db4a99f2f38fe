// hip_engine.hpp — C++ face of the C ABI in include/racon_hip.h.
//
// racon::HipEngine stands where the reference has spoa::AlignmentEngine (one per
// worker, reference src/polisher.cpp:179-183) and CUDABatchProcessor (one per
// device batch, reference src/cuda/cudabatch.hpp:27-122): it owns one engine
// handle bound to one device + stream and is NOT thread safe.  The shared
// library libracon_hip.so is resolved at run time (dlopen), so this host layer
// builds and loads on machines without a GPU; any attempt to compute without the
// library or without a device is a fatal error — there is no CPU fallback.
//
// racon::PackedBatch is the flat `rcn_batch` the ABI consumes, filled from
// Window objects (what CUDABatchProcessor::addWindow extracts, reference
// src/cuda/cudabatch.cpp:80-122, plus positions_.second / type, which it drops).
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/racon_hip.h"

namespace racon {

class Window;

struct PackedBatch {
    std::vector<uint32_t> win_seq_off{0};
    std::vector<uint8_t> win_type;
    std::vector<uint64_t> seq_off{0};
    std::vector<uint8_t> seq_has_qual;
    std::vector<uint32_t> seq_begin, seq_end;
    std::vector<uint8_t> bases, quals;

    void add(const Window& w);
    void clear();
    uint32_t n_windows() const { return static_cast<uint32_t>(win_type.size()); }
    uint64_t n_bases() const { return bases.size(); }
    rcn_batch view() const;           // pointers into this object
};

// The same information as BORROWED pointers (rcn_window_refs): what a Window already holds, one entry per sequence.
// The engine packs straight from these into its pinned staging, so the host copies every base once, not twice.
struct WindowRefs {
    std::vector<uint32_t> win_seq_off{0};
    std::vector<uint8_t> win_type;
    std::vector<const uint8_t*> seq, qual;      // qual[i] == nullptr: no quality
    std::vector<uint32_t> seq_len, seq_begin, seq_end;
    uint64_t bases = 0;

    void add(const Window& w);
    void clear();
    uint32_t n_windows() const { return static_cast<uint32_t>(win_type.size()); }
    rcn_window_refs view(uint32_t flags) const;
};

class HipEngine {
public:
    // Fatal (racon::fatal) when the library or the device is missing.
    // `arena_bytes`: this engine's share of the device's HBM for its scratch (0 = 80 % of what is free when it runs;
    // several engines on one device must split it, see FreeMemory).
    static std::shared_ptr<HipEngine> Create(int32_t device, int8_t match, int8_t mismatch, int8_t gap, uint64_t arena_bytes = 0);
    static uint64_t FreeMemory(int32_t device);   // free HBM in bytes (0 on error)
    static uint64_t UsedMemory(int32_t device);   // HBM in use on the device, by anyone (0 on error)
    static int32_t DeviceCount();      // 0 when the library cannot be loaded or no device is visible
    ~HipEngine();

    // Consensus of every window of `batch` (inputs are copied to HBM, results copied back).
    void consensus(const PackedBatch& batch, bool trim, std::vector<std::string>* consensus,
                   std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric);
    // The same from borrowed pointers (rcn_engine_polish_refs); `queued`: the caller keeps further batches in flight.
    void consensus(const WindowRefs& refs, bool queued, bool trim, std::vector<std::string>* consensus,
                   std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric);
    // Allocation ahead of the first batch (rcn_engine_reserve; the role of AlignmentEngine::Prealloc, reference
    // src/polisher.cpp:180-182): arena, pinned staging, code object, copy engines.
    void reserve(uint32_t n_windows, uint32_t n_seqs, uint64_t n_bases, uint32_t window_length, uint32_t max_layer_length,
                 uint64_t max_window_bases = 0);
    // ... for one known batch: exactly what consensus(refs, queued, ...) will allocate (rcn_engine_reserve_refs)
    void reserve(const WindowRefs& refs, bool queued);
    // The same, with the windows built on the device from the flattened sequences / overlaps
    // (rcn_engine_build_windows: reference src/polisher.cpp:388-461 in HBM).
    void consensus(const rcn_read_set& reads, const rcn_overlap_set& overlaps, uint32_t window_length, double quality_threshold,
                   uint8_t window_type, bool trim, std::vector<std::string>* consensus,
                   std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric);
    // ... and with the breaking points found on the device as well (rcn_engine_build_windows_from_cigars)
    void consensus(const rcn_read_set& reads, const rcn_cigar_set& alignments, uint32_t window_length, double quality_threshold,
                   uint8_t window_type, bool trim, std::vector<std::string>* consensus,
                   std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric);
    // ... and with the overlaps aligned on the device first (rcn_engine_build_windows_from_pairs: the edlib-equivalent of
    // reference src/overlap.cpp:205-224, byte-identical paths)
    void consensus(const rcn_read_set& reads, const rcn_pair_set& pairs, uint32_t window_length, double quality_threshold,
                   uint8_t window_type, bool trim, std::vector<std::string>* consensus,
                   std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric);
    // The two halves of those three: windows built in HBM and left resident (what Polisher::initialize does with host memory,
    // reference src/polisher.cpp:388-461), reserve_run() = everything run() will allocate (rcn_engine_reserve_run), and run() =
    // the consensus of the resident windows (Polisher::polish).  Fatal like consensus(); last_rc() says why.
    void build(const rcn_read_set& reads, const rcn_overlap_set& overlaps, uint32_t window_length, double quality_threshold, uint8_t window_type);
    void build(const rcn_read_set& reads, const rcn_cigar_set& alignments, uint32_t window_length, double quality_threshold, uint8_t window_type);
    void build(const rcn_read_set& reads, const rcn_pair_set& pairs, uint32_t window_length, double quality_threshold, uint8_t window_type);
    void reserve_run();
    void run(bool trim, std::vector<std::string>* consensus, std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric);
    double last_kernel_ms() const { return last_kernel_ms_; }
    int last_rc() const { return last_rc_; }      // return code of the ABI call behind the last consensus() (RCN_OK, RCN_E_*)
    // Only windows [first, last) of the next consensus() calls are copied into strings (the others come back empty):
    // a shard of a device-built job owns a range of the windows its engine returns.  (0, ~0) = all.
    void set_fetch_range(uint64_t first, uint64_t last) { fetch_first_ = first; fetch_last_ = last; }
    // The result block of the last run (consensus bytes at prefix-summed offsets in pinned host memory, written by the kernel): valid
    // until this engine's next build / run / reset.  A caller that concatenates the windows anyway reads them from here instead of
    // having every window copied into a string first (set_fetch_range(0, 0)).
    const rcn_result& last_result() const { return last_result_; }
    // RACON_HIP_VERIFY (rcn_engine_verify, run behind every batch when set): the Polisher's ids of the next batch's windows, so that a
    // failed self-check names the window as the Polisher numbers it (empty: the index within the batch); windows checked so far
    void set_verify_ids(std::function<uint64_t(uint32_t)> ids) { verify_ids_ = std::move(ids); }
    uint64_t verified_windows() const { return verified_windows_; }

private:
    HipEngine() = default;
    HipEngine(const HipEngine&) = delete;
    void built(int rc);
    void fetch(int rc, std::vector<std::string>* consensus, std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric, bool run = true);
    rcn_engine* handle_ = nullptr;
    double last_kernel_ms_ = 0;
    int last_rc_ = 0;
    rcn_result last_result_{};
    uint64_t fetch_first_ = 0, fetch_last_ = ~uint64_t(0);
    std::function<uint64_t(uint32_t)> verify_ids_;
    uint64_t verified_windows_ = 0;
    std::string pool_key_;           // non-empty: the handle goes back to the process-wide pool when this object dies (hip_engine.cpp)
};

}  // namespace racon
