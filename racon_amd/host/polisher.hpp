// polisher.hpp — racon::Polisher (reference src/polisher.hpp:33-99,
// src/polisher.cpp): parse targets / reads / overlaps, filter overlaps, cut
// reads into per-window layers, then polish every window and stitch the
// consensi per target.  Same factory signature, same two public calls, same
// output naming as the reference, so callers (reference src/main.cpp:147-161,
// test/racon_test.cpp:25-51) keep working.  What differs is where
// Window::generate_consensus runs: polish() batches the windows to the MI355X
// engines (one or more per device) the way the reference's CUDAPolisher::polish
// does (src/cuda/cudapolisher.cpp:216-413) — with the CPU path's exact results
// and no CPU fallback.
#pragma once
#include <chrono>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/racon_hip.h"
#include "window.hpp"

namespace racon {

class Sequence;
class Overlap;
class HipEngine;
struct PackedBatch;
struct WindowRefs;

enum class PolisherType { kC, kF };   // contig polishing / fragment error correction

class Polisher;
// `cudapoa_batches` keeps its place in the signature (reference src/polisher.hpp:42-48):
// here it is the number of HIP engines (batches in flight) per device, 0 meaning 1.
// `cuda_banded_alignment`, `cudaaligner_batches`, `cudaaligner_band_width` are accepted and
// ignored: the DP is exact/unbanded and the pre-alignment stays on the host.
std::unique_ptr<Polisher> createPolisher(const std::string& sequences_path, const std::string& overlaps_path,
    const std::string& target_path, PolisherType type, uint32_t window_length, double quality_threshold,
    double error_threshold, bool trim, int8_t match, int8_t mismatch, int8_t gap, uint32_t num_threads,
    uint32_t cudapoa_batches = 0, bool cuda_banded_alignment = false, uint32_t cudaaligner_batches = 0,
    uint32_t cudaaligner_band_width = 0);

class Logger {                          // reference src/logger.cpp:20-54
public:
    void log();
    void log(const std::string& msg) const;
    void bar(const std::string& msg);
    void total(const std::string& msg) const;
private:
    double time_ = 0; uint32_t bar_ = 0;
    std::chrono::time_point<std::chrono::steady_clock> time_point_{};
};

class Polisher {
public:
    virtual ~Polisher();
    virtual void initialize();
    virtual void polish(std::vector<std::unique_ptr<Sequence>>& dst, bool drop_unpolished_sequences);

    // --- the two halves of polish(), exposed for the parity harness ------------------
    // All windows, flattened (the bytes every consensus backend consumes).
    void pack_windows(PackedBatch* out) const;

    // What the two loops at the end of initialize() consume, flattened (include/racon_hip.h: rcn_read_set /
    // rcn_overlap_set): every sequence on its forward strand and every kept overlap with its breaking points.
    // Recorded by initialize() when keep_layout(true) was called before it; the input of rcn_engine_build_windows.
    // (bytes that are about to be overwritten need no zero fill first: resize() of these leaves them uninitialised)
    template <class T> struct NoInit : std::allocator<T> {
        template <class U> struct rebind { using other = NoInit<U>; };
        template <class U, class... A> void construct(U* p, A&&... a) {
            if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U; else ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
        }
    };
    using ByteBuf = std::vector<uint8_t, NoInit<uint8_t>>;
    struct Layout {
        std::vector<uint64_t> seq_off{0};
        ByteBuf bases, quals;
        std::vector<uint8_t> seq_has_qual;
        uint64_t n_targets = 0;
        std::vector<uint32_t> q_id, t_id;
        std::vector<uint8_t> strand;
        std::vector<uint64_t> bp_off{0};
        std::vector<uint32_t> bp_t, bp_q;
        // the alignments the breaking points came from (rcn_cigar_set): CIGAR text, first query position on the
        // overlap's strand, target extent
        std::vector<uint64_t> cigar_off{0};
        ByteBuf cigar;
        std::vector<uint32_t> q_start, t_begin, t_end;
        // the aligned query segment on the forward read (rcn_pair_set): what the device aligner needs of an overlap
        // that came without a CIGAR
        std::vector<uint32_t> q_begin, q_end;
        uint8_t window_type = 0;
    };
    void keep_layout(bool on) { keep_layout_ = on; }
    // what initialize() does when RACON_HIP_DEVICE_WINDOWS is not set: "0" (the library's default: the caller may ask for windows())
    // or "auto" (the racon_hip binary)
    void set_default_device_mode(const std::string& mode) { default_device_mode_ = mode; }
    // Windows built on the device: initialize() then skips the serial add_layer loop (windows_ keep only their backbones,
    // which polish() needs for the stitching) and records the layout instead.  Also switched on by RACON_HIP_DEVICE_WINDOWS=1.
    // cigars: the CIGAR walk (Overlap::find_breaking_points, reference src/overlap.cpp:226-292) runs on the device too
    // (rcn_engine_build_windows_from_cigars; RACON_HIP_DEVICE_WINDOWS=2).
    // align: overlaps without a CIGAR (PAF / MHAP) are aligned on the device as well (rcn_engine_build_windows_from_pairs:
    // the edlib-equivalent of reference src/overlap.cpp:205-224 in HBM; RACON_HIP_DEVICE_WINDOWS=3); files whose overlaps
    // carry CIGARs (SAM) take the cigars path.
    void device_windows(bool on, bool cigars = false, bool align = false) {
        device_windows_ = on; device_cigars_ = on && (cigars || align); device_align_ = on && align; if (on) keep_layout_ = true;
    }
    const Layout& layout() const { return layout_; }
    // ---- windows built on the device (device_job.cpp) ------------------------------------------------------------------------
    // The cut of the job into shards: contiguous window ranges balanced by the overlaps over them (the reference hands window
    // ranges to per-device batches, src/cuda/cudapolisher.cpp:228-240), the targets behind every range, and every shard's overlaps
    // (indices into the layout, ascending; an overlap across a boundary is in both).  Pure host code.
    struct DevicePlan {
        uint32_t n_shards = 0;
        std::vector<uint64_t> cut;                  // [n_shards + 1]: shard s owns windows [cut[s], cut[s + 1])
        std::vector<uint64_t> target_lo, target_hi; // [n_shards]: ... which lie in targets [target_lo[s], target_hi[s])
        std::vector<uint64_t> first_window;         // [targets + 1]
        std::vector<uint64_t> bucket_off, bucket;   // overlaps of shard s: bucket[bucket_off[s] .. bucket_off[s + 1])
    };
    DevicePlan plan_device_job(uint32_t n_shards) const;
    // What rcn_engine_build_windows* takes for one shard (include/racon_hip.h): its targets, the reads its overlaps point into
    // (re-numbered), its overlaps' slices of the layout; `whole`: the one shard of an uncut job, views straight into the layout.
    struct ShardInput {
        uint32_t sidx = 0;
        uint64_t wa = 0, wb = 0;        // the shard's windows (the job's numbering)
        uint64_t window_base = 0;       // ... of which the engine's window 0 is this one
        uint64_t n_local = 0;           // windows the engine builds (all windows of the shard's targets)
        bool whole = false;
        std::vector<uint32_t> q_id, t_id, bp_t, bp_q, q_start, t_begin, t_end, q_begin, q_end;
        std::vector<uint8_t> strand, has_qual;
        std::vector<uint64_t> bp_off, cigar_off, seq_off;
        ByteBuf cigar, bases, quals;
        rcn_read_set reads{}; rcn_overlap_set overlaps{};
        const uint32_t *p_q_start = nullptr, *p_t_begin = nullptr, *p_t_end = nullptr, *p_q_begin = nullptr, *p_q_end = nullptr;
        const uint64_t* p_cigar_off = nullptr; const uint8_t* p_cigar = nullptr;
        std::vector<uint32_t> remap, old_of;        // (work arrays of make_shard_input)
        double t_begin_s = 0, t_sliced_s = 0;       // (timing lines)
    };
    void make_shard_input(const DevicePlan& plan, uint32_t sidx, ShardInput* out) const;
    uint32_t window_length() const { return window_length_; }
    double quality_threshold() const { return quality_threshold_; }
    // Per-target concatenation + tags from per-window results (reference src/polisher.cpp:505-537).
    // consensus(i) / polished(i) are indexed like windows(); consumes the windows.
    void assemble(const std::function<const std::string&(uint64_t)>& consensus,
                  const std::function<bool(uint64_t)>& polished,
                  std::vector<std::unique_ptr<Sequence>>& dst, bool drop_unpolished_sequences);
    // the same over (pointer, length) views: windows polished on resident device windows are concatenated straight from their engines'
    // result blocks (HipEngine::last_result), without a string per window in between
    typedef std::pair<const char*, size_t> ConsensusView;
    void assemble_views(const std::function<ConsensusView(uint64_t)>& consensus,
                        const std::function<bool(uint64_t)>& polished,
                        std::vector<std::unique_ptr<Sequence>>& dst, bool drop_unpolished_sequences);
    uint64_t num_windows() const { return windows_.size(); }
    const std::vector<std::shared_ptr<Window>>& windows() const { return windows_; }

    friend std::unique_ptr<Polisher> createPolisher(const std::string&, const std::string&, const std::string&, PolisherType,
        uint32_t, double, double, bool, int8_t, int8_t, int8_t, uint32_t, uint32_t, bool, uint32_t, uint32_t);

protected:
    Polisher(const std::string& sequences_path, const std::string& overlaps_path, const std::string& target_path,
             PolisherType type, uint32_t window_length, double quality_threshold, double error_threshold, bool trim,
             int8_t match, int8_t mismatch, int8_t gap, uint32_t num_threads, uint32_t hip_batches);
    Polisher(const Polisher&) = delete;
    Polisher& operator=(const Polisher&) = delete;
    virtual void find_overlap_breaking_points(std::vector<std::unique_ptr<Overlap>>& overlaps);

    std::string sequences_path_, overlaps_path_, target_path_;
    PolisherType type_;
    double quality_threshold_, error_threshold_;
    bool trim_;
    int8_t match_, mismatch_, gap_;
    uint32_t num_threads_, hip_batches_;

    std::vector<std::unique_ptr<Sequence>> sequences_;
    std::vector<uint32_t> targets_coverages_;
    std::string dummy_quality_;
    uint32_t window_length_;
    std::vector<std::shared_ptr<Window>> windows_;
    bool keep_layout_ = false;
    std::string default_device_mode_ = "0";
    bool device_cigars_ = false;
    bool device_align_ = false;     // overlaps without a CIGAR are aligned on the device (set back by initialize() when the file has CIGARs)
    bool device_windows_ = false;   // polish(): windows built in HBM (rcn_engine_build_windows) instead of packed from windows_
    Layout layout_;
    std::unique_ptr<Logger> logger_;
    // Started by initialize(): loads libracon_hip.so and brings up the HIP runtime and the devices' contexts while the input
    // files are parsed (0.2 s that polish() would otherwise spend before its first launch); joined by polish() / the destructor.
    std::thread device_warmup_;
    std::thread cleanup_;           // frees the windows and sequences polish() is done with (see assemble())
    std::thread cleanup2_;          // ... and the per-window result strings
    // The engines polish() drives: `2 * hip_batches_` per device, created ONCE by the warm-up thread together with their
    // arenas, pinned staging and the first use of the code object (HipEngine::reserve; the reference creates its
    // alignment engines in the constructor and Preallocs them, src/polisher.cpp:176-183), so that the interval the
    // Logger brackets around polish() (reference src/polisher.cpp:493 -> :539-543) holds consensus work only.
    std::vector<std::shared_ptr<HipEngine>> engines_;
    std::string engines_error_;     // what went wrong in the warm-up thread (reported by polish())
    int32_t n_devices_ = -1;
    double polish_seconds_ = 0;     // the Logger-bracketed interval of the last polish()
    uint32_t polish_chunks_ = 0, polish_engines_used_ = 0;   // chunks / engines that took one in the last polish() (host-built windows)
    void create_engines();          // (warm-up thread, or polish() when the warm-up was switched off)
    // windows built on the device (device_windows_): phase 1 at the end of initialize(), phase 2 in polish(); phase 0 = both in polish()
    void build_device_windows();
    void device_job(int phase, std::vector<std::string>* cons, std::vector<uint8_t>* pol, std::vector<uint8_t>* chim);
    // phase 2 (every shard resident on an engine of its own): the windows' consensus as views into the engines' result blocks
    std::vector<ConsensusView> cons_views_;
    void build_shard(HipEngine& engine, const ShardInput& in);           // rcn_engine_build_windows* by mode, host aligner as the way out
    void run_shard(bool views, HipEngine& engine, uint64_t window_base, uint64_t n_local, uint64_t wa, uint64_t wb,
                   std::vector<std::string>& cons, std::vector<uint8_t>& pol, std::vector<uint8_t>& chim);
    uint32_t device_shards() const;         // devices, RACON_HIP_DEVICE_SHARDS, or what a failed build raised it to
    bool device_built_ = false;             // initialize() left the windows resident on the engines
    DevicePlan device_plan_;                // ... cut like this
    uint32_t device_min_shards_ = 1;        // raised when a build found no room on the device: the job is cut into more, smaller shards
    void reserve_for_windows();     // end of initialize(): the arenas sized for the windows that were built
    // polish()'s work list: windows ranked deepest first, cut into chunks (planned once, by reserve_for_windows or polish)
    std::vector<uint32_t> rank_;
    std::vector<std::pair<uint64_t, uint64_t>> chunks_;     // [first, last) positions in rank_
    void plan_chunks();
    std::vector<WindowRefs> planned_refs_;                   // the pointer tables of the first chunks (built for the dry run, used by polish())
public:
    double polish_seconds() const { return polish_seconds_; }
    uint32_t polish_chunks() const { return polish_chunks_; }
    uint32_t polish_engines_used() const { return polish_engines_used_; }
};

}  // namespace racon
