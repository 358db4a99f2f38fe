// engine.hip — host side of the MI355X window-consensus engine and its C ABI
// (include/racon_hip.h).  Plays the role of the reference's CUDABatchProcessor
// (reference src/cuda/cudabatch.cpp:24-280) — pack windows, run the device POA,
// hand back consensus strings + per-window status — but with the CPU path's
// exact semantics (reference src/window.cpp:65-149), unbounded window shapes and
// no CPU fallback: a window that exceeds the first-pass capacities is re-run on
// the GPU with worst-case capacities.
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/racon_hip.h"
#include "poa_kernel2.hpp"
#include "poa_small.hpp"

namespace {

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t err__ = (expr);                                                          \
        if (err__ != hipSuccess) {                                                          \
            fprintf(stderr, "[racon_hip] HIP error %s at %s:%d: %s\n", hipGetErrorName(err__), \
                    __FILE__, __LINE__, #expr);                                             \
            return RCN_E_HIP;                                                               \
        }                                                                                   \
    } while (0)

// tests (Knobs::fail_alloc_above, RCN_FAIL_ALLOC_ABOVE behind RCN_EXPERIMENT=1): device allocations above this many bytes fail like
// an exhausted device -- the way the callers survive RCN_E_NOMEM can be exercised on a box with 288 GB
static std::atomic<uint64_t> g_fail_alloc_above{0};

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap && p) return RCN_OK;
        { const uint64_t lim = g_fail_alloc_above.load(std::memory_order_relaxed); if (lim && bytes > lim) return RCN_E_NOMEM; }
        // A buffer that GROWS gets an eighth more than asked for: a sequence of like-sized jobs on one engine (the window-range
        // shards of a job cut to fit one device: cfg5 whole, eight of them) differs by fractions of a percent, and every
        // regrowth of a multi-GB buffer is a hipFree + hipMalloc that waits for the driver's page clearing (0.6-2.4 s,
        // tools/probe/malloc_time.hip).  The first allocation is exact.
        // Round 6: so does a FIRST allocation of a GiB or more while the device has three times that free -- the one regrowth a
        // sharded job was left with (its second shard's arena a few per cent above the first's) cost cfg5 whole 2.1 s in one run
        // and 1.7 s in another, and nothing in a third (profiles/r06/o_cfg5_whole_one_gpu.json, c_, i_).
        bool regrow = p != nullptr;
        if (p) { if (hipFree(p) != hipSuccess) return RCN_E_HIP; p = nullptr; cap = 0; }
        size_t want = std::max<size_t>(bytes, 256);
        if (!regrow && bytes >= (size_t(1) << 30)) { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr >= 3 * want) regrow = true; else (void)hipGetLastError(); }
        if (regrow && bytes >= (64u << 20) && hipMalloc(&p, want + want / 8) == hipSuccess) { cap = want + want / 8; return RCN_OK; }
        (void)hipGetLastError();
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return RCN_E_NOMEM; }
        cap = want; return RCN_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct HostBuf {                       // pinned host staging (D2H of the consensus bytes at full PCIe rate)
    void* p = nullptr; size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap && p) return RCN_OK;
        // (a block that has to grow gets an eighth on top: the shards of a job differ by a few per cent, and every regrowth is a
        //  hipHostFree + hipHostMalloc of > 100 MB -- 45-95 ms with the device idle, profiles/r06/j_run_breakdown.txt)
        size_t want = std::max<size_t>(bytes, 256);
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; want += want / 8; }
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
            if (want == std::max<size_t>(bytes, 256) || hipHostMalloc(&p, want = std::max<size_t>(bytes, 256), hipHostMallocDefault) != hipSuccess) { p = nullptr; return RCN_E_NOMEM; }
        }
        cap = want; return RCN_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct WinShape { int32_t L, sum_l, lmax, nsym; };

// The metadata of a streamed batch (polish_view) as ONE block, laid out alike in pinned staging and in HBM: byte offsets of
// win_seq_off, win_type, the per-window flags, seq_off, has_qual, begin, end, layer order, full-span flags, output offsets.
// One block because of how the runtime copies: a copy below its blit threshold (tens of KB) is a shader kernel, and a
// shader kernel waits for a compute unit -- behind the persistent consensus launch of the device's OTHER engine that was
// 15-36 ms (profiles/r03/c_timeline_cfg3_share.txt, d_...), for 3 KB of flags.  Large copies go to the DMA engines.
struct MetaLayout { uint64_t wso, type, flags, so, hq, bg, en, ord, full, ooff, end; };
inline MetaLayout meta_layout(uint64_t nw, uint64_t ns) {
    MetaLayout m;
    m.wso = 0; m.type = m.wso + 4 * (nw + 1); m.flags = m.type + nw; m.so = (m.flags + nw + 15) & ~uint64_t(15);
    m.hq = m.so + 8 * (ns + 1); m.bg = (m.hq + ns + 15) & ~uint64_t(15); m.en = m.bg + 4 * ns; m.ord = m.en + 4 * ns;
    m.full = m.ord + 4 * ns; m.ooff = (m.full + ns + 15) & ~uint64_t(15); m.end = (m.ooff + 8 * (nw + 1) + 255) & ~uint64_t(255);
    return m;
}
constexpr uint64_t kDmaCopyBytes = 64 << 10;    // copies are padded up to this (with bytes the destination already holds)

// two HIP events for a timed interval, destroyed on every exit path
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    int create() {
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return RCN_E_HIP;
        return RCN_OK;
    }
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

// Host worker threads shared by every engine of the process: creating threads per call costs more (~40 us each) than the
// packing work of a small piece.  Chunks of different callers interleave on one queue; a caller runs one chunk itself and
// then helps with whatever is queued until its own chunks are done.
class HostPool {
public:
    static HostPool& get() { static HostPool p; return p; }
    unsigned workers() const { return static_cast<unsigned>(threads_.size()); }
    void submit(std::function<void()> f) {
        { std::lock_guard<std::mutex> g(m_); q_.push_back(std::move(f)); }
        cv_.notify_one();
    }
    bool run_one() {                     // a queued chunk on the calling thread; false if the queue is empty
        std::function<void()> f;
        { std::lock_guard<std::mutex> g(m_); if (q_.empty()) return false; f = std::move(q_.front()); q_.pop_front(); }
        f();
        return true;
    }
private:
    HostPool() {
        // fifteen helpers feed one device (a batch is packed by up to sixteen threads); a process that drives several
        // devices -- every one with two engines packing their chunks at once -- gets twelve more per further device
        int devices = 1;
        if (hipGetDeviceCount(&devices) != hipSuccess || devices < 1) devices = 1;
        const unsigned want = 15u + 12u * static_cast<unsigned>(devices - 1);
        const unsigned n = std::min(want, std::max(2u, std::thread::hardware_concurrency()) - 1u);
        for (unsigned t = 0; t < n; ++t) threads_.emplace_back([this]() { loop(); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& th : threads_) th.join();
    }
    void loop() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this]() { return stop_ || !q_.empty(); });
                if (q_.empty()) { if (stop_) return; continue; }
                f = std::move(q_.front()); q_.pop_front();
            }
            f();
        }
    }
    std::mutex m_; std::condition_variable cv_; std::deque<std::function<void()>> q_; std::vector<std::thread> threads_; bool stop_ = false;
};

// Runs fn(i) for i in [0, n) on up to `threads` host threads (contiguous blocks); exceptions are not expected from fn.
template <class F>
void host_parallel(size_t n, unsigned threads, F fn) {
    HostPool& pool = HostPool::get();
    threads = static_cast<unsigned>(std::min<size_t>(std::min(std::max(1u, threads), pool.workers() + 1u), std::max<size_t>(1, n)));
    if (threads <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<unsigned> left{threads - 1};
    for (unsigned t = 1; t < threads; ++t)
        pool.submit([&fn, &left, n, t, threads]() {
            for (size_t i = n * t / threads; i < n * (t + 1) / threads; ++i) fn(i);
            left.fetch_sub(1, std::memory_order_release);
        });
    for (size_t i = 0; i < n / threads; ++i) fn(i);                      // chunk 0 here
    while (left.load(std::memory_order_acquire) != 0) { if (!pool.run_one()) std::this_thread::yield(); }
}

}  // namespace

// Experiment and test switches, read ONCE when the engine is created and only when RCN_EXPERIMENT=1 is set (tests/conftest.py
// sets it; tools/exp scripts do): a stray RCN_* variable in a user's environment cannot change the product's kernel path.
// RCN_DEBUG (host timeline on stderr, no effect on results) is the one switch that needs no gate.
struct Knobs {
    bool debug = false, no_ptab = false, force_exact = false, force_slow_tb = false, no_band = false, force_band_fail = false,
         band_scores = false, no_code_wave = false, wide_only = false, no_stream = false, no_small = false, force_small = false, prof_layers = false,
         cigar_serial = false;
    int plant_fault = 0;                            // RCN_PLANT_FAULT (tests of the self-check): 1 = the sink-tie rule picks the LAST key instead of the first
    int force_tie = 0, wg_per_cu = 0, split = -1, split_deep_per_cu = 0, split_rest_per_cu = 0, split_deep = 0, split_deep_wide = -1,
        split_cus = 0, hrows_div = 0, small_per_cu = 0, split_mid = 0, split_mid_per_cu = 0, split_mid_cus = 0;
    double heavy_pct = 1.0;
    unsigned long long fail_alloc_above = 0;
};
static Knobs read_knobs() {
    Knobs k;
    k.debug = getenv("RCN_DEBUG") != nullptr;
    const char* gate = getenv("RCN_EXPERIMENT");
    if (!gate || atoi(gate) == 0) {
        // a script that exports RCN_* switches without the gate compares identical configurations: say so, once
        static bool warned = false;
        extern char** environ;
        for (char** ev = environ; ev && *ev && !warned; ++ev)
            if (!strncmp(*ev, "RCN_", 4) && strncmp(*ev, "RCN_DEBUG", 9) && strncmp(*ev, "RCN_EXPERIMENT", 14)) {
                fprintf(stderr, "[racon_hip] warning: %.*s is set but ignored (experiment switches need RCN_EXPERIMENT=1)\n", static_cast<int>(strcspn(*ev, "=")), *ev);
                warned = true;
            }
        return k;
    }
    auto flag = [](const char* n) { return getenv(n) != nullptr; };
    auto num = [](const char* n, int dflt) { const char* v = getenv(n); return v ? atoi(v) : dflt; };
    k.no_ptab = flag("RCN_NO_PTAB"); k.force_exact = flag("RCN_FORCE_EXACT"); k.force_slow_tb = flag("RCN_FORCE_SLOW_TB");
    k.no_band = flag("RCN_NO_BAND"); k.force_band_fail = flag("RCN_FORCE_BAND_FAIL"); k.band_scores = flag("RCN_BAND_SCORES");
    k.no_code_wave = flag("RCN_NO_CODE_WAVE"); k.wide_only = flag("RCN_WIDE_ONLY"); k.no_stream = flag("RCN_NO_STREAM");
    k.cigar_serial = flag("RCN_CIGAR_SERIAL"); k.no_small = flag("RCN_NO_SMALL"); k.force_small = flag("RCN_FORCE_SMALL"); k.prof_layers = flag("RCN_PROF_LAYERS");
    k.plant_fault = num("RCN_PLANT_FAULT", 0);
    k.force_tie = num("RCN_FORCE_TIE", 0); k.wg_per_cu = num("RCN_WG_PER_CU", 0); k.split = num("RCN_SPLIT", -1);
    k.split_deep_per_cu = num("RCN_SPLIT_DEEP_PER_CU", 0); k.split_rest_per_cu = num("RCN_SPLIT_REST_PER_CU", 0);
    k.split_deep = num("RCN_SPLIT_DEEP", 0); k.split_deep_wide = num("RCN_SPLIT_DEEP_WIDE", -1); k.split_cus = num("RCN_SPLIT_CUS", 0);
    k.hrows_div = num("RCN_HROWS_DIV", 0); k.small_per_cu = num("RCN_SMALL_PER_CU", 0);
    k.split_mid = num("RCN_SPLIT_MID", 0); k.split_mid_per_cu = num("RCN_SPLIT_MID_PER_CU", 0); k.split_mid_cus = num("RCN_SPLIT_MID_CUS", 0);
    if (const char* v = getenv("RCN_HEAVY_PCT")) k.heavy_pct = atof(v);
    if (const char* v = getenv("RCN_FAIL_ALLOC_ABOVE")) k.fail_alloc_above = strtoull(v, nullptr, 10);
    return k;
}

struct rcn_engine {
    rcn_engine_config cfg{};
    Knobs knobs;
    hipStream_t stream = nullptr;                   // main stream: resident-batch launches, retry pass, result copies
    hipStream_t copy_stream = nullptr;              // H2D of a streamed batch (rcn_engine_polish)
    // sub-launches of a streamed batch: each on its own stream so that they overlap.  Two pieces, three streams in use
    // (copy, piece 0, piece 1 = main): the device runs kernels of at most four hardware queues side by side, a third
    // piece was observed to wait for the second one to finish
    static constexpr int kSubLaunches = 2;
    hipStream_t sub_stream[kSubLaunches] = {nullptr, nullptr};
    // CU-masked pair for the split launch of a batch that is resident all at once (split_plan): the deepest windows on
    // `deep_stream` (the first split_cus bits of the mask = an even slice of every XCD, tools/probe/cu_mask.hip) with few
    // work-groups per CU, all the others on `rest_stream` (the complement).  Null when the runtime refused the masks.
    hipStream_t deep_stream = nullptr, rest_stream = nullptr;
    int split_cus = 0;
    // three tiers (RCN_SPLIT_MID*): the next-deepest windows at a few work-groups per CU on CUs of their own, the rest on what is left
    hipStream_t mid_stream = nullptr, rest3_stream = nullptr;
    int split_mid_cus = 0;
    static constexpr int kMaxLaunches = 3;          // launches of one first pass (split launch: deep, [middle,] rest; streamed batch: its pieces)
    bool warmed = false;                            // rcn_engine_reserve ran its warm-up launch
    bool lds_optin = false;                         // a work-group may ask for more than 64 KB of LDS (fewer than three per CU)
    int caps_level = 0;                             // first_pass_caps: raised when a batch needed many retries
    bool small_off = false;                         // the small-window kernel sent too many windows back: not for this engine's next batches
    bool pass_small = false;                        // the first pass of the current batch ran (partly) on the small-window kernel
    std::vector<uint8_t> item_small;                // ... and which work items did (collect: only those are re-run by poa_window_kernel2 first)
    // host preparation of the batch rcn_engine_reserve_refs made its dry run for (layer order, full-span flags; shapes / work order /
    // output offsets stay in the fields above): the polish call for the SAME batch takes it over instead of sorting 700 000 layers again
    bool pc_valid = false;
    uint64_t pc_key[4] = {0, 0, 0, 0};
    std::vector<uint32_t> pc_order;
    std::vector<uint8_t> pc_full;
    bool stats_pending = false;                     // the last run's device counters have not been read yet (rcn_engine_stats)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t sub_ev[kMaxLaunches][3] = {};        // per sub-launch: copy done, kernel begin, kernel end
    int n_cu = 256;
    uint64_t t_max = 0, t_sum = 0;      // bases of the deepest window / of all windows of the prepared batch (wg_per_cu)
    bool queued = false;                // RCN_REFS_QUEUED: the caller keeps several batches in flight (wg_per_cu: eight)
    size_t free_mem = 0;

    // resident batch
    uint32_t n_windows = 0, n_seqs = 0;
    uint64_t n_bases = 0;
    // lpt_layout: the device arrays hold the windows deepest first (work item k IS device window k; streamed upload);
    // otherwise they are in caller order and d_lpt_ids maps work items to windows
    bool lpt_layout = false;
    DevBuf d_win_seq_off, d_win_type, d_seq_off, d_has_qual, d_begin, d_end, d_bases, d_quals, d_order, d_full;
    DevBuf d_lpt_ids, d_win_ids, d_win_flags, d_scratch, d_out_cons, d_out_len, d_out_flags, d_out_off, d_ctr;
    DevBuf d_meta, d_retry_off;         // streamed batches (lpt_layout): the metadata block (MetaLayout ml); output offsets of a retry pass
    MetaLayout ml{};
    HostBuf h_out, h_stage;                         // pinned: outputs of a pass; inputs of a streamed batch
    std::vector<WinShape> shapes;                   // by window (caller order)
    int32_t heavy_ns = 0;
    std::vector<uint32_t> h_win_seq_off;
    std::vector<uint32_t> lpt;          // work item -> window, deepest windows first (longest processing time first)
    std::vector<uint64_t> out_off;      // [n_windows + 1] consensus byte offsets by work item (first pass)
    bool uploaded = false, ran = false;

    // results
    std::vector<uint64_t> cons_off;
    std::vector<uint8_t> cons, polished, chimeric;
    rcn_run_stats stats{};
    rcn_build_stats bstats{};
    DevBuf d_build[24];                 // rcn_engine_build_windows: resident reads / overlaps / work arrays
    DevBuf d_align[12];                 // rcn_engine_align_pairs: pair table, op bytes, distances, scratch
    rcn_align_stats astats{};
    uint64_t a_n_pairs = 0;             // pairs of the last alignment run (their op bytes are resident)
    std::vector<uint64_t> a_ops_off;

    // incremental builder (addWindow form)
    std::vector<uint32_t> b_win_seq_off{0};
    std::vector<uint8_t> b_win_type, b_has_qual, b_bases, b_quals;
    std::vector<uint64_t> b_seq_off{0};
    std::vector<uint32_t> b_begin, b_end;
};

namespace {

constexpr size_t kCtrBytes = 512;       // d_ctr: 16 work-queue counters (4 B each), then the statistics words at byte 64
constexpr size_t kStatsOff = 64;

int upload_vec(DevBuf& d, const void* src, size_t bytes, hipStream_t s) {
    int rc = d.reserve(bytes);
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, s));
    return RCN_OK;
}

struct Caps { int32_t ncap, ecap, ring, lmax, hstride, hrows; uint64_t slot_bytes; bool fast; bool small = false; uint32_t lds = 0, per_cu = 0; };

// fast = poa_window_kernel2 (4 waves per window, int16 Z matrix); else poa_window_kernel (1 wave, int32 H)
// hrows: rows of the DP matrix (0 = ncap + 1, one per node the graph arrays can hold)
Caps make_caps(int32_t ncap, int32_t ecap, int32_t ring, int32_t lmax, bool fast, int32_t hrows = 0) {
    Caps c; c.ncap = ncap; c.ecap = ecap; c.ring = ring; c.lmax = lmax; c.fast = fast;
    c.hrows = (hrows > 0 && hrows < ncap + 1) ? hrows : ncap + 1;
    // fast: row stride in int16 cells, a multiple of 512: every DP shape (64 or 256 lanes x 2..8 cells) then
    // covers whole rows only, so the row store needs no lane mask
    c.hstride = fast ? ((lmax + 1 + 511) / 512) * 512 : (lmax + 1 + 128 + 3) & ~3;
    rcn::Win tmp;
    c.slot_bytes = rcn::win_bind(tmp, nullptr, ncap, ecap, ring, lmax, c.hstride, fast ? 2 : 4, c.hrows);
    c.slot_bytes = (c.slot_bytes + 255) & ~uint64_t(255);
    return c;
}

// ---- the small-window kernel (poa_small.hpp): one wave per window, the graph in LDS ----
// A pass takes it when EVERY window of the pass has the shape: backbone and growth within the LDS layout, layers of at most
// 255 bases (what is not known before the bases are read -- symbols besides A, C, G, T -- and what only shows while the
// graph grows is the kernel's own business: it flags such a window and collect() re-runs it with poa_window_kernel2).
// Node capacity = backbone + the larger of 64 and 3/8 of it + 13 (BASELINE configs[3], 200-base windows at 60x: graphs end
// at 221 nodes, the largest of 5000 at ~270; 288 nodes = 10 080 bytes of LDS = sixteen windows per CU).
inline int small_ncap(int L) { return std::max<int>(rcn::kSmMinCap, (L + std::max(64, (3 * L) / 8 + 13) + 7) & ~7); }
inline bool small_shape(const WinShape& s) { return s.L >= 1 && small_ncap(s.L) <= rcn::kSmMaxCap && s.lmax <= rcn::kSmLen; }
template <class It>
bool small_caps(const rcn_engine* e, It first, It last, Caps& out);

// first-pass capacities of a set of windows: typical graph growth (a window that outgrows them is re-run by the retry
// pass with worst-case capacities)
// `level`: 0 = the estimates below; raised by collect() when a batch sent more than 2 % of its windows to the retry pass
// (reads noisier than the 10-15 % the estimates leave room for): every step gives the graph and the matrix more room
// for the NEXT batch -- a node per 3, then 2 layer bases, a matrix row per 4, then 2.  `hrows_div` (tests, Knobs::hrows_div: a
// small matrix forces retries) overrides the matrix divisor.
template <class It>
Caps first_pass_caps(It first, It last, bool fast, int level = 0, int hrows_div = 0) {
    // The graph arrays (~230 B per node) for a node per four layer bases; the DP matrix -- one row of 1 or 2 KB per node,
    // nine tenths of a slot -- for one per six: cfg2's windows end with a node per ~12 layer bases (10 % read error, most
    // errors shared by no other read), so both leave room, and the arena (slots x slot bytes, tens of GB that the driver
    // has to find and clear) shrinks by a third.
    const int hdiv = hrows_div > 0 ? hrows_div : (level <= 0 ? 6 : level == 1 ? 4 : 2);
    const int ndiv = level <= 0 ? 4 : level == 1 ? 3 : 2;
    int32_t ncap = 0, hrows = 0, lmax = 1, nsym = 2;
    for (It it = first; it != last; ++it) {
        const WinShape& s = *it;
        const int64_t worst = static_cast<int64_t>(s.L) + s.sum_l + 8;
        const int64_t est = static_cast<int64_t>(s.L) + s.sum_l / ndiv + 256;
        ncap = std::max<int32_t>(ncap, static_cast<int32_t>(std::min(worst, est)));
        hrows = std::max<int32_t>(hrows, static_cast<int32_t>(std::min(worst + 1, static_cast<int64_t>(s.L) + s.sum_l / hdiv + 129)));
        lmax = std::max(lmax, s.lmax); nsym = std::max(nsym, s.nsym);
    }
    return make_caps(ncap, 2 * ncap, std::max(1, nsym - 1), lmax, fast, fast ? hrows : 0);
}

// Consensus bytes reserved for a window in the first pass.  The consensus is a path of the graph: it can be as long as
// the graph in theory, about the backbone in practice; a longer one is flagged (kFlagOverflow) and comes from the retry pass.
inline uint64_t first_pass_out_cap(const WinShape& s) {
    const uint64_t worst = static_cast<uint64_t>(s.L) + s.sum_l + 8;
    return (std::min<uint64_t>(worst, 2ull * s.L + 64) + 15) & ~uint64_t(15);
}

// capacities of a pass on the small-window kernel, false when some window of [first, last) does not have the shape
template <class It>
bool small_caps(const rcn_engine* e, It first, It last, Caps& out) {
    if (first == last || e->knobs.no_small || e->knobs.wide_only || (e->small_off && !e->knobs.force_small) || e->cfg.gap >= 0) return false;
    // A pass takes the kernel when (nearly) all of its windows have the shape: up to one window in eight may not (real reads
    // on short windows: a few layers beyond 255 bases among thousands within) -- the kernel flags those at once (kSmCap /
    // kSmLong, before it touches anything sized by the capacities below) and collect() hands them to poa_window_kernel2
    // like every other window that leaves the kernel.  The capacities follow the windows that do have the shape.
    // Only windows that will be POLISHED count (three or more sequences = two or more layers = sum_l > lmax; the others are copied
    // through by either kernel, window.cpp:68-71): a window-range shard of a device-built job holds every OTHER range's windows as
    // bare backbones, and counting those as "has the shape" handed a batch of 250 000 ten-kilobase-read windows among 1.75 M bare
    // backbones to this kernel -- which flagged every real window and left them all to the retry tier (cfg5 whole, round 5).
    int32_t Lmax = 1;
    uint64_t shaped = 0, total = 0;
    for (It it = first; it != last; ++it) {
        if (it->sum_l <= it->lmax) { if (small_shape(*it)) Lmax = std::max(Lmax, it->L); continue; }
        ++total;
        if (!small_shape(*it)) continue;
        ++shaped;
        Lmax = std::max(Lmax, it->L);
    }
    if (shaped == 0 || (total - shaped) * 8 > total) return false;
    Caps c{};
    c.ncap = small_ncap(Lmax); c.ecap = 0; c.ring = rcn::kSmRing; c.lmax = rcn::kSmLen; c.hstride = 256; c.hrows = c.ncap + 1;
    c.fast = true; c.small = true;
    c.slot_bytes = rcn::small_slot_bytes(c.ncap);
    c.lds = rcn::small_layout(c.ncap).end;
    // LDS is handed out in 1280-byte granules, 128 per CU (a 200-base window: 13 216 bytes = eleven granules: eleven per CU);
    // twelve one-wave work-groups (three per SIMD) are what the kernel's register budget admits (__launch_bounds__(64, 3): up
    // to 168 VGPRs, nothing spilled)
    const uint32_t granules = (c.lds + 1279u) / 1280u;
    c.per_cu = std::max(1u, std::min(12u, 128u / granules));
    if (e->knobs.small_per_cu > 0) c.per_cu = static_cast<uint32_t>(std::min(32, e->knobs.small_per_cu));
    out = c;
    return true;
}

// What a pass may take for its slots: 80 % of what is free NOW (free_mem is refreshed at the start of every run by begin_run / rcn_engine_reserve,
// which fold the scratch this engine already holds into it -- nothing is added here), and never more than the caller's arena.  (An arena fixed when the engine was created -- the host
// layer splits a device's free memory between its engines then -- says nothing about what reads, overlaps and other engines
// have taken since: sized from it alone, a pass asked hipMalloc for memory that was no longer there.)
uint64_t scratch_budget(const rcn_engine* e) {
    const uint64_t now = static_cast<uint64_t>(e->free_mem * 0.80);
    return e->cfg.arena_bytes ? std::min<uint64_t>(e->cfg.arena_bytes, now) : now;
}
uint32_t max_slots(const rcn_engine* e) {
    return e->cfg.max_slots ? e->cfg.max_slots : static_cast<uint32_t>(e->n_cu) * 8u;    // 8 work-groups per CU (20 KiB LDS each) at most
}

// ---- how many work-groups of poa_window_kernel2 share a CU ----
// Eight fit (20.5 KiB of LDS each, 32 wave slots), and eight is what a long queue wants: 142k windows/s at 8000 windows
// against 119k with six and 92k with four (profiles/r02/f_occupancy.txt).  But a batch whose windows are all resident at
// once lasts as long as its DEEPEST window, and that window is faster with fewer neighbours on its CU: its DP wave
// shares a SIMD's VALU with at most one other (-11 % DP clocks at four per CU), its graph phases share the CU's memory
// pipeline (-30 %).  The estimate below says which case a batch is: per-window time goes with the window's bases (sum of
// its layers' lengths; measured ~ n_seqs^1.1), the launch is max(deepest window, all windows / slots), six per CU cost
// 1.23 x the second term (6 / 8 of the slots at 0.925 of the clocks) and save ~4 % of the first.
// Fewer than eight per CU is enforced by asking for more LDS per work-group (allocation granule 1280 B, 128 per CU).
// deepest window against an even share of all windows: does one window's serial chain rule the launch?
bool deepest_rules(const rcn_engine* e) {
    const double even = static_cast<double>(e->t_sum) / (8.0 * e->n_cu);           // bases per slot if all slots stayed busy
    return static_cast<double>(e->t_max) > 1.25 * even;
}
uint32_t wg_per_cu(const rcn_engine* e) {
    if (e->knobs.wg_per_cu > 0) return static_cast<uint32_t>(std::min(8, std::max(e->lds_optin ? 1 : 3, e->knobs.wg_per_cu)));   // experiments
    if (e->queued) return 8u;                       // the caller keeps the device busy with further batches: a long queue
    return deepest_rules(e) ? 6u : 8u;
}
uint32_t lds_bytes_for(uint32_t per_cu) {
    const uint32_t need = rcn::kLdsBytes + rcn::kCtxBytes;
    return per_cu >= 8 ? need : std::max(need, (128u / std::max(1u, per_cu)) * 1280u);
}

// ---- split launch: the deepest windows on CUs of their own ----
// A batch that is resident all at once ends with its deepest window (cfg2: 53 layers against a median of 30, ~45 % of
// the slot-time of a uniform launch is tail), and a window runs faster the fewer neighbours share its CU: -11 % DP
// clocks and -30 % in the four-wave graph phases at four work-groups per CU instead of eight
// (profiles/r02/f_occupancy.txt).  Uniform residency cannot use that, two launches on two CU-masked streams can
// (hipExtStreamCreateWithCUMask; tools/probe/cu_mask.hip: the first N mask bits are N/8 CUs of every XCD, the
// complement is honoured): the `n_deep` deepest windows -- the layer chains that cannot be reordered, reference
// src/window.cpp:88-119 -- at `deep_per_cu` work-groups per CU on `split_cus` CUs, all the others at eight per CU on the
// rest of the chip, persistent over their queue.  Same kernel, same results; the decision uses the cost proxy of
// wg_per_cu.  RCN_SPLIT=0 switches it off, RCN_SPLIT=1 forces it (tests), RCN_SPLIT_DEEP / RCN_SPLIT_DEEP_PER_CU /
// RCN_SPLIT_REST_PER_CU override the plan (experiments).
struct SplitPlan { bool on = false; uint32_t n_deep = 0, deep_per_cu = 1, rest_per_cu = 8, n_mid = 0, mid_per_cu = 4; };
SplitPlan split_plan(const rcn_engine* e, uint32_t nw, bool fast) {
    SplitPlan sp;
    if (!fast || !e->deep_stream || !e->rest_stream || e->cfg.max_slots || e->knobs.split == 0) return sp;
    const bool forced = e->knobs.split == 1;
    const uint32_t cus = static_cast<uint32_t>(e->split_cus);
    if (!forced && (e->queued || !deepest_rules(e) || nw < 256u || nw > static_cast<uint32_t>(e->n_cu) * 8u)) return sp;
    if (nw < 2) return sp;
    sp.on = true;
    sp.deep_per_cu = 1; sp.rest_per_cu = 8;
    if (e->knobs.split_deep_per_cu > 0) sp.deep_per_cu = static_cast<uint32_t>(std::min(8, e->knobs.split_deep_per_cu));
    if (e->knobs.split_rest_per_cu > 0) sp.rest_per_cu = static_cast<uint32_t>(std::min(8, e->knobs.split_rest_per_cu));
    if (!e->lds_optin) sp.deep_per_cu = std::max(sp.deep_per_cu, 3u);      // (residency is set through the LDS request: 64 KB at most without the opt-in)
    sp.n_deep = cus * sp.deep_per_cu;
    if (e->knobs.split_deep > 0) sp.n_deep = static_cast<uint32_t>(e->knobs.split_deep);
    sp.n_deep = std::min(sp.n_deep, nw - 1);
    // middle tier (experiment switches; resident batches): the next n_mid windows at mid_per_cu work-groups per CU
    if (e->mid_stream && e->knobs.split_mid > 0 && sp.n_deep + 1 < nw) {
        sp.mid_per_cu = static_cast<uint32_t>(std::min(8, std::max(1, e->knobs.split_mid_per_cu > 0 ? e->knobs.split_mid_per_cu : 4)));
        if (!e->lds_optin) sp.mid_per_cu = std::max(sp.mid_per_cu, 3u);
        sp.n_mid = std::min(static_cast<uint32_t>(e->knobs.split_mid), nw - 1 - sp.n_deep);
    }
    return sp;
}

struct ResultLayout { uint64_t off_flags, off_cons; };
inline ResultLayout result_layout(uint32_t nw) { ResultLayout r; r.off_flags = 4ull * nw; r.off_cons = (r.off_flags + nw + 15) & ~uint64_t(15); return r; }

struct Launch {
    Caps c;
    const uint32_t* d_ids = nullptr;    // work item -> device window, or nullptr: work_base + work item
    uint32_t n_work = 0, work_base = 0, out_base = 0, slots = 0;
    uint32_t per_cu = 0;                // work-groups per CU of the fast kernel (0: wg_per_cu)
    int32_t heavy_ns = -1;              // KParams::heavy_ns of this pass (-1: the engine's)
    bool host_out = true;               // results go straight into the pinned result block (first pass); false: d_out_* (retry pass)
    uint64_t scratch_off = 0;
    int ctr = 0;                        // which work-queue counter of d_ctr
    hipStream_t stream = nullptr;
};

// Enqueues one kernel pass (asynchronous).  Scratch for [scratch_off, scratch_off + slots * slot_bytes) must be reserved.
int launch_pass(rcn_engine* e, const Launch& L) {
    if (L.n_work == 0) return RCN_OK;
    HIP_TRY(hipMemsetAsync(e->d_ctr.as<uint8_t>() + 4 * L.ctr, 0, 4, L.stream));
    rcn::KParams P{};
    const uint8_t* win_flags;
    if (e->lpt_layout) {                // a streamed batch: everything but the bases sits in the metadata block
        const uint8_t* mb = e->d_meta.as<uint8_t>();
        const MetaLayout& m = e->ml;
        P.win_seq_off = reinterpret_cast<const uint32_t*>(mb + m.wso); P.win_type = mb + m.type;
        P.seq_off = reinterpret_cast<const uint64_t*>(mb + m.so); P.seq_has_qual = mb + m.hq;
        P.seq_begin = reinterpret_cast<const uint32_t*>(mb + m.bg); P.seq_end = reinterpret_cast<const uint32_t*>(mb + m.en);
        P.order = reinterpret_cast<const uint32_t*>(mb + m.ord); P.seq_full = mb + m.full;
        P.out_off = reinterpret_cast<const uint64_t*>(mb + m.ooff);
        win_flags = mb + m.flags;
    } else {
        P.win_seq_off = e->d_win_seq_off.as<uint32_t>(); P.win_type = e->d_win_type.as<uint8_t>();
        P.seq_off = e->d_seq_off.as<uint64_t>(); P.seq_has_qual = e->d_has_qual.as<uint8_t>();
        P.seq_begin = e->d_begin.as<uint32_t>(); P.seq_end = e->d_end.as<uint32_t>();
        P.order = e->d_order.as<uint32_t>(); P.seq_full = e->d_full.as<uint8_t>();
        P.out_off = e->d_out_off.as<uint64_t>();
        win_flags = e->d_win_flags.as<uint8_t>();
    }
    if (!L.host_out) P.out_off = e->d_retry_off.as<uint64_t>();         // a retry pass numbers its outputs by its own work items
    P.bases = e->d_bases.as<uint8_t>(); P.quals = e->d_quals.as<uint8_t>();
    P.win_ids = L.d_ids; P.n_work = L.n_work; P.work_base = L.work_base;
    const Knobs& K = e->knobs;
    P.win_flags = (K.no_ptab && !L.c.small) ? nullptr : win_flags;
    P.m = e->cfg.match; P.x = e->cfg.mismatch; P.g = e->cfg.gap; P.trim = e->cfg.trim;
    P.heavy_ns = L.heavy_ns >= 0 ? L.heavy_ns : e->heavy_ns; P.force_exact = K.force_exact ? 1 : 0;
    P.force_tie = K.force_tie;
    P.force_slow_tb = (K.force_slow_tb ? 1 : 0) | (K.plant_fault == 1 ? 2 : 0);       // (bit 1: the planted wrong tie rule, poa_k2_sinktie.hpp)
    // the band's exactness certificate (poa_band.hpp: a cell is alive when H' + m (len - j) >= T) assumes that a remaining
    // base adds at most m: any -m/-x/-g is legal on the command line (reference src/main.cpp:51-53,91-99), so score sets
    // with x > m or g > m take full rows
    const bool band_sound = e->cfg.match >= e->cfg.mismatch && e->cfg.match >= e->cfg.gap;
    P.band = (!band_sound || K.no_band) ? 0 : (K.force_band_fail ? 2 : (K.band_scores ? 3 : 1));
    P.scratch = e->d_scratch.as<uint8_t>() + L.scratch_off; P.slot_bytes = L.c.slot_bytes;
    P.ncap = L.c.ncap; P.ecap = L.c.ecap; P.ring = L.c.ring; P.lmax = L.c.lmax; P.hstride = L.c.hstride; P.hrows = L.c.hrows;
    P.out_base = L.out_base;
    P.lds_extra = (L.c.fast && !L.c.small) ? static_cast<int32_t>(lds_bytes_for(L.per_cu ? L.per_cu : wg_per_cu(e)) - (rcn::kLdsBytes + rcn::kCtxBytes)) : 0;
    P.no_help = K.no_code_wave ? 1 : 0;
    if (L.host_out) {
        // The first pass writes lengths, flags and consensus bytes straight into pinned host memory (hipHostMalloc memory is
        // device-visible): ~1 KB per window over PCIe from the kernel's own stores, and no device-to-host copy afterwards -- a
        // copy that, queued behind another engine's persistent launch or simply first of its kind on a stream, was observed
        // to take 8-9 ms for 2-3 MB (profiles/r03/b_timeline_*).  Layout: [lengths 4 nw][flags nw, padded][bytes].
        const ResultLayout rl = result_layout(e->n_windows);
        uint8_t* hb = e->h_out.as<uint8_t>();
        P.out_len = reinterpret_cast<uint32_t*>(hb); P.out_flags = hb + rl.off_flags; P.out_cons = hb + rl.off_cons;
    } else {
        P.out_cons = e->d_out_cons.as<uint8_t>(); P.out_len = e->d_out_len.as<uint32_t>(); P.out_flags = e->d_out_flags.as<uint8_t>();
    }
    P.next = e->d_ctr.as<unsigned int>() + L.ctr;
    P.stats = reinterpret_cast<unsigned long long*>(e->d_ctr.as<uint8_t>() + kStatsOff);
    const uint32_t per_cu = L.c.small ? L.c.per_cu : (L.per_cu ? L.per_cu : wg_per_cu(e));
    if (K.debug) fprintf(stderr, "[racon_hip] pass of %u windows on %u slots, %u work-groups per CU%s (deepest window %llu bases, all %llu)\n", L.n_work, L.slots,
                         L.c.fast ? per_cu : 8u, L.c.small ? " (small-window kernel)" : "", (unsigned long long)e->t_max, (unsigned long long)e->t_sum);
    if (L.c.fast) e->stats.wg_per_cu = per_cu;
    if (L.c.small) {
        // one wave per window, the graph in LDS (poa_small.hpp); a window outside its shape comes back flagged (collect())
        hipLaunchKernelGGL(rcn::poa_window_kernel_small, dim3(L.slots), dim3(64), L.c.lds, L.stream, P);
        HIP_TRY(hipGetLastError());
        return RCN_OK;
    }
    // one work-group per CU with the CU's whole LDS: the instance of the kernel that runs the banded DP with code waves
    const bool deep_kernel = L.c.fast && P.lds_extra >= rcn::kHelpLdsBytes && !P.no_help;
    if (deep_kernel) hipLaunchKernelGGL(rcn::poa_window_kernel2_deep, dim3(L.slots), dim3(rcn::kThreads2), lds_bytes_for(per_cu), L.stream, P);
    else if (L.c.fast) hipLaunchKernelGGL(rcn::poa_window_kernel2, dim3(L.slots), dim3(rcn::kThreads2), lds_bytes_for(per_cu), L.stream, P);
    else hipLaunchKernelGGL(rcn::poa_window_kernel, dim3(L.slots), dim3(64), rcn::kLdsBytes + rcn::kCtxBytes, L.stream, P);
    HIP_TRY(hipGetLastError());
    return RCN_OK;
}

// resident slots for a pass of n_work windows within the scratch budget (0: not even one slot fits)
// A queue of a few windows per slot ends when the slot with one window more than the others is done: for the small-window
// kernel (whose windows are alike) the slots are cut down to what fills whole rounds -- 5000 windows on 4096 slots are two
// rounds for 904 slots and one for the rest; on 2500 slots they are two rounds for all, each window with fewer neighbours.
inline uint32_t balanced_slots(uint32_t n_work, uint32_t max_slots) {
    if (max_slots == 0 || n_work <= max_slots) return n_work;
    const uint32_t rounds = (n_work + max_slots - 1) / max_slots;
    return rounds > 8 ? max_slots : (n_work + rounds - 1) / rounds;
}
uint32_t slots_for(const rcn_engine* e, const Caps& c, uint32_t n_work, uint64_t budget) {
    uint32_t slots = std::min(e->cfg.max_slots ? e->cfg.max_slots : static_cast<uint32_t>(e->n_cu) * (c.small ? c.per_cu : wg_per_cu(e)), n_work);
    if (c.small && !e->cfg.max_slots) slots = balanced_slots(n_work, slots);
    while (slots > 1 && static_cast<uint64_t>(slots) * c.slot_bytes > budget) slots = (slots + 1) / 2;
    return static_cast<uint64_t>(slots) * c.slot_bytes > budget ? 0 : slots;
}

// one synchronous kernel pass on the main stream (resident batch / retry pass), timed with HIP events
int run_pass(rcn_engine* e, const Caps& c, const uint32_t* d_ids, uint32_t n_work, bool host_out = true) {
    if (n_work == 0) return RCN_OK;
    Launch L; L.c = c; L.d_ids = d_ids; L.n_work = n_work; L.stream = e->stream; L.host_out = host_out;
    L.slots = slots_for(e, c, n_work, scratch_budget(e));
    if (L.slots == 0) return RCN_E_CAPACITY;
    int rc = e->d_scratch.reserve(static_cast<uint64_t>(L.slots) * c.slot_bytes);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    if ((rc = launch_pass(e, L))) return rc;
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->stats.kernel_ms += ms; e->stats.n_launches += 1;
    return RCN_OK;
}

// Host-side facts of a batch that the kernel takes as inputs (computed once per batch, caller window order):
// layer order (window.cpp:79-86: the reference's std::sort, unstable -- libstdc++'s introsort is what decides ties),
// full-span flags (window.cpp:88,93-94), shape statistics for the scratch capacities, the ACGT-only flag, and the
// deepest-first work order.  `bases` may be null (batch built on the device): every window then gets the batch-wide
// symbol count `nsym_all`.
struct HostPrep { std::vector<uint32_t> order; std::vector<uint8_t> full, wflags; };

// symbols of one window's sequences: how many distinct byte values, and whether they are all A/C/G/T
inline void scan_symbols(const uint8_t* bases, uint64_t a, uint64_t z, int32_t& nsym, uint8_t& acgt_only) {
    uint64_t present[4] = {0, 0, 0, 0};
    for (uint64_t k = a; k < z; ++k) present[bases[k] >> 6] |= 1ull << (bases[k] & 63);
    nsym = 0;
    for (uint64_t p : present) nsym += __builtin_popcountll(p);
    const uint64_t acgt = (1ull << ('A' & 63)) | (1ull << ('C' & 63)) | (1ull << ('G' & 63)) | (1ull << ('T' & 63));
    acgt_only = (present[0] == 0 && present[2] == 0 && present[3] == 0 && (present[1] & ~acgt) == 0) ? 1 : 0;
}

// `scan` = false (streamed upload): the symbol statistics (shapes[].nsym, wflags) are left for the caller, which reads
// the bases anyway when it packs a piece.
int prepare_host(rcn_engine* e, HostPrep& hp, uint32_t nw, uint32_t ns, const uint32_t* win_seq_off, const uint64_t* seq_off,
                 const uint32_t* seq_begin, const uint32_t* seq_end, const uint8_t* bases, int32_t nsym_all, bool acgt_all, bool scan = true) {
    // (shapes, work order and output capacities below are what a dry run's cached preparation -- polish_view, pc_* -- stands on: whoever
    //  recomputes them for another batch invalidates it)
    e->pc_valid = false;
    hp.wflags.assign(nw, 0); hp.order.resize(ns); hp.full.assign(ns, 0);
    e->h_win_seq_off.assign(win_seq_off, win_seq_off + nw + 1);
    e->shapes.resize(nw);
    std::vector<int> bad(nw, 0);
    const unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    host_parallel(nw, nw >= 256 ? threads : 1, [&](size_t wi) {
        const uint32_t w = static_cast<uint32_t>(wi);
        const uint32_t s0 = win_seq_off[w], n = win_seq_off[w + 1] - s0;
        if (n == 0) { bad[w] = 1; return; }
        std::vector<uint32_t> rank(n);
        for (uint32_t i = 0; i < n; ++i) rank[i] = i;
        std::sort(rank.begin() + 1, rank.end(), [&](uint32_t lhs, uint32_t rhs) {
            return seq_begin[s0 + lhs] < seq_begin[s0 + rhs]; });
        const uint32_t L = static_cast<uint32_t>(seq_off[s0 + 1] - seq_off[s0]);
        if (L == 0) { bad[w] = 1; return; }                   // createWindow rejects empty backbones (window.cpp:19-23)
        const uint32_t offset = static_cast<uint32_t>(0.01 * L);
        WinShape sh{static_cast<int32_t>(L), 0, 0, 0};
        for (uint32_t i = 0; i < n; ++i) {
            hp.order[s0 + i] = rank[i];
            const uint32_t si = s0 + i;
            const uint64_t a = seq_off[si], z = seq_off[si + 1];
            if (i > 0) {
                const uint32_t bg = seq_begin[si], en = seq_end[si];
                if (z == a || bg >= en || bg > L || en > L) { bad[w] = 1; return; }   // add_layer contract (window.cpp:45-58)
                hp.full[si] = (bg < offset && en > L - offset) ? 1 : 0;
                sh.sum_l += static_cast<int32_t>(z - a);
                sh.lmax = std::max<int32_t>(sh.lmax, static_cast<int32_t>(z - a));
            }
        }
        if (bases && scan) scan_symbols(bases, seq_off[s0], seq_off[s0 + n], sh.nsym, hp.wflags[w]);
        else if (!bases) { sh.nsym = nsym_all; hp.wflags[w] = acgt_all ? 1 : 0; }
        e->shapes[w] = sh;
    });
    for (uint32_t w = 0; w < nw; ++w) if (bad[w]) return RCN_E_ARG;
    e->t_max = 0; e->t_sum = 0;                                   // wg_per_cu()
    for (uint32_t w = 0; w < nw; ++w) {
        const uint64_t tw = win_seq_off[w + 1] - win_seq_off[w] < 3 ? 0 : static_cast<uint64_t>(e->shapes[w].sum_l + e->shapes[w].L);
        e->t_max = std::max(e->t_max, tw); e->t_sum += tw;
    }
    // Longest processing time first: a window's cost grows with (layers x bases), and a launch ends with its
    // slowest window; when there are more windows than resident slots the deep ones must not start last.
    e->lpt.resize(nw);
    for (uint32_t w = 0; w < nw; ++w) e->lpt[w] = w;
    std::stable_sort(e->lpt.begin(), e->lpt.end(), [&](uint32_t a, uint32_t c) {
        const uint64_t ca = static_cast<uint64_t>(win_seq_off[a + 1] - win_seq_off[a]) * static_cast<uint64_t>(e->shapes[a].sum_l + e->shapes[a].L);
        const uint64_t cc = static_cast<uint64_t>(win_seq_off[c + 1] - win_seq_off[c]) * static_cast<uint64_t>(e->shapes[c].sum_l + e->shapes[c].L);
        return ca > cc; });
    // consensus output offsets by work item
    e->out_off.assign(static_cast<size_t>(nw) + 1, 0);
    for (uint32_t wi = 0; wi < nw; ++wi) e->out_off[wi + 1] = e->out_off[wi] + first_pass_out_cap(e->shapes[e->lpt[wi]]);
    return RCN_OK;
}

}  // namespace

// Preparation of a batch that is resident in caller order (rcn_engine_upload, rcn_engine_build_windows): prepare_host +
// upload of its products.
int prepare_resident(rcn_engine* e, uint32_t nw, uint32_t ns, const uint32_t* win_seq_off, const uint64_t* seq_off,
                            const uint32_t* seq_begin, const uint32_t* seq_end, const uint8_t* bases, int32_t nsym_all, bool acgt_all = false) {
    HostPrep hp;
    int rc = prepare_host(e, hp, nw, ns, win_seq_off, seq_off, seq_begin, seq_end, bases, nsym_all, acgt_all);
    if (rc) return rc;
    e->lpt_layout = false;
    if ((rc = upload_vec(e->d_lpt_ids, e->lpt.data(), 4ull * nw, e->stream))) return rc;
    if ((rc = upload_vec(e->d_order, hp.order.data(), 4ull * ns, e->stream))) return rc;
    if ((rc = upload_vec(e->d_full, hp.full.data(), ns, e->stream))) return rc;
    if ((rc = upload_vec(e->d_win_flags, hp.wflags.data(), nw, e->stream))) return rc;
    if ((rc = upload_vec(e->d_out_off, e->out_off.data(), 8ull * (nw + 1), e->stream))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));                 // hp is stack-scoped
    return RCN_OK;
}

#include "pair_align.hpp"
#include "window_build.hpp"

extern "C" {

int rcn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* rcn_version(void) { return "racon-hip 0.1.0 (gfx950)"; }

const char* rcn_strerror(int code) {
    switch (code) {
        case RCN_OK: return "ok";
        case RCN_BATCH_FULL: return "batch full";
        case RCN_E_NO_DEVICE: return "no HIP device (the engine has no CPU fallback)";
        case RCN_E_HIP: return "HIP runtime error";
        case RCN_E_ARG: return "invalid argument";
        case RCN_E_NOMEM: return "out of device memory";
        case RCN_E_STATE: return "call sequence error";
        case RCN_E_CAPACITY: return "window exceeds device scratch budget";
        case RCN_E_LAYER: return "layer begin and end positions are invalid (Window::add_layer contract)";
        default: return "unknown";
    }
}

int rcn_engine_create(const rcn_engine_config* cfg, rcn_engine** out) {
    if (!cfg || !out) return RCN_E_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RCN_E_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= n) return RCN_E_ARG;
    HIP_TRY(hipSetDevice(cfg->device));
    // the handle owns its streams / events from the first one on: every early return below destroys what exists
    struct Guard { rcn_engine* e; ~Guard() { if (e) rcn_engine_destroy(e); } } guard{new rcn_engine()};
    rcn_engine* e = guard.e;
    e->cfg = *cfg;
    e->knobs = read_knobs();
    g_fail_alloc_above.store(e->knobs.fail_alloc_above);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    e->n_cu = prop.multiProcessorCount;
    size_t fr = 0, tot = 0;
    HIP_TRY(hipMemGetInfo(&fr, &tot));
    e->free_mem = fr;
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&e->sub_stream[0], hipStreamNonBlocking));
    e->sub_stream[rcn_engine::kSubLaunches - 1] = e->stream;                 // the last piece runs on the main stream
    HIP_TRY(hipEventCreate(&e->ev0));
    HIP_TRY(hipEventCreate(&e->ev1));
    for (auto& evs : e->sub_ev) for (auto& ev : evs) HIP_TRY(hipEventCreate(&ev));
    {
        // the CU-masked stream pair of the split launch (split_plan); a runtime that refuses masks leaves them null
        // 32 of the 256 CUs for the deep launch.  cfg2 (profiles/r03/o_split_sweep.txt): the deep launch lasts as long as the
        // deepest window's own chain whatever it is given (19.3-19.5 ms with 16, 24, 32 or 64 CUs), the other launch gets
        // shorter the more CUs it keeps (19.6 ms on 192, 17.3 on 224): 19.3-19.5 ms per step with 16-32, 19.6 with 64,
        // 20.4 without the split.  (Masks of 40, 48, 56 or 72 bits ran the deep launch in TWO rounds -- 29 ms: fewer CUs
        // than bits took its work-groups -- so the count stays one of those measured.)
        int cus = 32;
        if (e->knobs.split_cus > 0) cus = e->knobs.split_cus;
        cus = (cus / 8) * 8;                                   // an even slice of the eight XCDs
        if (cus >= 8 && cus <= e->n_cu - 8 && e->n_cu <= 256 && e->knobs.split != 0) {
            uint32_t deep[8] = {0}, rest[8] = {0};
            for (int b = 0; b < e->n_cu; ++b) (b < cus ? deep : rest)[b >> 5] |= 1u << (b & 31);
            const uint32_t words = static_cast<uint32_t>((e->n_cu + 31) / 32);
            if (hipExtStreamCreateWithCUMask(&e->deep_stream, words, deep) != hipSuccess) e->deep_stream = nullptr;
            if (e->deep_stream && hipExtStreamCreateWithCUMask(&e->rest_stream, words, rest) != hipSuccess) e->rest_stream = nullptr;
            if (!e->rest_stream && e->deep_stream) { (void)hipStreamDestroy(e->deep_stream); e->deep_stream = nullptr; }
            (void)hipGetLastError();
            e->split_cus = e->deep_stream ? cus : 0;
            // the middle tier's CUs follow the deep launch's in the mask; the third stream gets what both leave
            const int mcus = (e->knobs.split_mid_cus / 8) * 8;
            if (e->split_cus && mcus >= 8 && cus + mcus <= e->n_cu - 8) {
                uint32_t mid[8] = {0}, rest3[8] = {0};
                for (int b = cus; b < e->n_cu; ++b) (b < cus + mcus ? mid : rest3)[b >> 5] |= 1u << (b & 31);
                if (hipExtStreamCreateWithCUMask(&e->mid_stream, words, mid) != hipSuccess) e->mid_stream = nullptr;
                if (e->mid_stream && hipExtStreamCreateWithCUMask(&e->rest3_stream, words, rest3) != hipSuccess) e->rest3_stream = nullptr;
                if (!e->rest3_stream && e->mid_stream) { (void)hipStreamDestroy(e->mid_stream); e->mid_stream = nullptr; }
                (void)hipGetLastError();
                e->split_mid_cus = e->mid_stream ? mcus : 0;
            }
        }
        // fewer than eight work-groups per CU are enforced through the LDS request (lds_bytes_for): up to the whole LDS
        e->lds_optin = hipFuncSetAttribute(reinterpret_cast<const void*>(rcn::poa_window_kernel2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        e->lds_optin = e->lds_optin && hipFuncSetAttribute(reinterpret_cast<const void*>(rcn::poa_window_kernel2_deep), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rcn::poa_window_kernel_small), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipGetLastError();
    }
    int rc = e->d_ctr.reserve(kCtrBytes);
    if (rc) return rc;
    guard.e = nullptr;
    *out = e;
    return RCN_OK;
}

void rcn_engine_destroy(rcn_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    for (DevBuf* d : {&e->d_win_seq_off, &e->d_win_type, &e->d_seq_off, &e->d_has_qual, &e->d_begin, &e->d_end,
                      &e->d_bases, &e->d_quals, &e->d_order, &e->d_full, &e->d_lpt_ids, &e->d_win_ids, &e->d_win_flags, &e->d_scratch,
                      &e->d_out_cons, &e->d_out_len, &e->d_out_flags, &e->d_out_off, &e->d_ctr, &e->d_meta, &e->d_retry_off})
        d->release();
    for (DevBuf& d : e->d_build) d.release();
    for (DevBuf& d : e->d_align) d.release();
    e->h_out.release(); e->h_stage.release();
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    for (auto& evs : e->sub_ev) for (auto& ev : evs) if (ev) (void)hipEventDestroy(ev);
    for (auto& st : e->sub_stream) if (st && st != e->stream) (void)hipStreamDestroy(st);
    if (e->deep_stream) (void)hipStreamDestroy(e->deep_stream);
    if (e->rest_stream) (void)hipStreamDestroy(e->rest_stream);
    if (e->mid_stream) (void)hipStreamDestroy(e->mid_stream);
    if (e->rest3_stream) (void)hipStreamDestroy(e->rest3_stream);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int rcn_device_free_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
    if (hipSetDevice(device) != hipSuccess) return RCN_E_ARG;
    size_t fr = 0, tot = 0;
    HIP_TRY(hipMemGetInfo(&fr, &tot));
    if (free_bytes) *free_bytes = fr;
    if (total_bytes) *total_bytes = tot;
    return RCN_OK;
}

static int check_batch(const rcn_batch* b) {
    if (b->n_windows && (!b->win_seq_off || !b->win_type || !b->seq_off || !b->seq_has_qual || !b->seq_begin ||
                         !b->seq_end || !b->bases || !b->quals))
        return RCN_E_ARG;
    return RCN_OK;
}

int rcn_engine_upload(rcn_engine* e, const rcn_batch* b) {
    if (!e || !b || check_batch(b)) return RCN_E_ARG;
    HIP_TRY(hipSetDevice(e->cfg.device));
    EventPair t;
    if (t.create()) return RCN_E_HIP;
    HIP_TRY(hipEventRecord(t.a, e->stream));
    const uint32_t nw = b->n_windows, ns = b->n_seqs;
    e->n_windows = nw; e->n_seqs = ns; e->n_bases = ns ? b->seq_off[ns] : 0;
    e->uploaded = false; e->ran = false; e->queued = false;
    int rc;
    if ((rc = prepare_resident(e, nw, ns, b->win_seq_off, b->seq_off, b->seq_begin, b->seq_end, b->bases, 0))) return rc;
    if ((rc = upload_vec(e->d_win_seq_off, b->win_seq_off, 4ull * (nw + 1), e->stream))) return rc;
    if ((rc = upload_vec(e->d_win_type, b->win_type, nw, e->stream))) return rc;
    if ((rc = upload_vec(e->d_seq_off, b->seq_off, 8ull * (ns + 1), e->stream))) return rc;
    if ((rc = upload_vec(e->d_has_qual, b->seq_has_qual, ns, e->stream))) return rc;
    if ((rc = upload_vec(e->d_begin, b->seq_begin, 4ull * ns, e->stream))) return rc;
    if ((rc = upload_vec(e->d_end, b->seq_end, 4ull * ns, e->stream))) return rc;
    if ((rc = upload_vec(e->d_bases, b->bases, e->n_bases, e->stream))) return rc;
    if ((rc = upload_vec(e->d_quals, b->quals, e->n_bases, e->stream))) return rc;
    HIP_TRY(hipEventRecord(t.b, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
    e->stats = rcn_run_stats{};
    e->stats.h2d_ms = ms;
    e->stats.bytes_in = 2 * e->n_bases + 17ull * ns + 5ull * nw;
    e->uploaded = true;
    return RCN_OK;
}

int rcn_engine_build_windows(rcn_engine* e, const rcn_read_set* reads, const rcn_overlap_set* ovl,
                             uint32_t window_length, double quality_threshold, uint8_t window_type) {
    if (!e || !reads || !ovl || window_length == 0) return RCN_E_ARG;
    return rcn::build_windows(e, *reads, *ovl, window_length, quality_threshold, window_type);
}

int rcn_engine_build_windows_from_cigars(rcn_engine* e, const rcn_read_set* reads, const rcn_cigar_set* al,
                                         uint32_t window_length, double quality_threshold, uint8_t window_type) {
    if (!e || !reads || !al || window_length == 0) return RCN_E_ARG;
    return rcn::build_windows_from_cigars(e, *reads, *al, window_length, quality_threshold, window_type);
}

int rcn_engine_align_pairs(rcn_engine* e, const rcn_read_set* reads, const rcn_pair_set* pairs) {
    if (!e || !reads || !pairs) return RCN_E_ARG;
    return rcn::align_pairs(e, *reads, *pairs, false);
}

int rcn_engine_build_windows_from_pairs(rcn_engine* e, const rcn_read_set* reads, const rcn_pair_set* pairs,
                                        uint32_t window_length, double quality_threshold, uint8_t window_type) {
    if (!e || !reads || !pairs || window_length == 0) return RCN_E_ARG;
    return rcn::build_windows_from_pairs(e, *reads, *pairs, window_length, quality_threshold, window_type);
}

int rcn_engine_align_stats(rcn_engine* e, rcn_align_stats* out) {
    if (!e || !out) return RCN_E_ARG;
    *out = e->astats;
    return RCN_OK;
}

// Op bytes -> CIGAR text (edlibAlignmentToCigar, EDLIB_CIGAR_STANDARD: M / I / D runs), on the host after one D2H copy.
int rcn_engine_alignment_cigars(rcn_engine* e, uint64_t* cigar_off, char* cigar, uint64_t cap, uint64_t* need, int32_t* distance) {
    if (!e || !cigar_off) return RCN_E_ARG;
    if (e->a_ops_off.empty() || e->a_ops_off.size() != e->a_n_pairs + 1) return RCN_E_STATE;
    HIP_TRY(hipSetDevice(e->cfg.device));
    const uint64_t n = e->a_n_pairs, bytes = e->a_ops_off[n];
    std::vector<uint8_t> ops(bytes + 1);
    if (bytes) HIP_TRY(hipMemcpy(ops.data(), e->d_align[rcn::kAOps].p, bytes, hipMemcpyDeviceToHost));
    if (distance && n) HIP_TRY(hipMemcpy(distance, e->d_align[rcn::kADist].p, 4 * n, hipMemcpyDeviceToHost));
    std::string all;
    cigar_off[0] = 0;
    for (uint64_t o = 0; o < n; ++o) {
        const uint64_t a = e->a_ops_off[o], z = e->a_ops_off[o + 1];
        uint8_t run_op = 0; uint64_t run = 0;
        for (uint64_t p = a; p < z; ++p) {
            const uint8_t op = ops[p];
            if (!op) continue;
            if (op == run_op) { ++run; continue; }
            if (run) { all += std::to_string(run); all += static_cast<char>(run_op); }
            run_op = op; run = 1;
        }
        if (run) { all += std::to_string(run); all += static_cast<char>(run_op); }
        cigar_off[o + 1] = all.size();
    }
    if (need) *need = all.size();
    if (!cigar || cap < all.size()) return cigar ? RCN_E_CAPACITY : RCN_OK;
    std::memcpy(cigar, all.data(), all.size());
    return RCN_OK;
}

int rcn_engine_build_stats(rcn_engine* e, rcn_build_stats* out) {
    if (!e || !out) return RCN_E_ARG;
    *out = e->bstats;
    return RCN_OK;
}

int rcn_engine_batch_dims(rcn_engine* e, rcn_batch_dims* out) {
    if (!e || !out) return RCN_E_ARG;
    if (!e->uploaded) return RCN_E_STATE;
    out->n_windows = e->n_windows; out->n_seqs = e->n_seqs; out->n_bases = e->n_bases;
    return RCN_OK;
}

int rcn_engine_export_batch(rcn_engine* e, uint32_t* win_seq_off, uint8_t* win_type, uint64_t* seq_off,
                            uint8_t* seq_has_qual, uint32_t* seq_begin, uint32_t* seq_end, uint8_t* bases, uint8_t* quals) {
    if (!e) return RCN_E_ARG;
    if (!e->uploaded) return RCN_E_STATE;
    HIP_TRY(hipSetDevice(e->cfg.device));
    const uint64_t nw = e->n_windows, ns = e->n_seqs;
    auto get = [&](void* dst, const DevBuf& src, size_t bytes) -> int {
        if (dst && bytes) HIP_TRY(hipMemcpy(dst, src.p, bytes, hipMemcpyDeviceToHost));
        return RCN_OK;
    };
    int rc;
    if (!e->lpt_layout) {
        if ((rc = get(win_seq_off, e->d_win_seq_off, 4 * (nw + 1)))) return rc;
        if ((rc = get(win_type, e->d_win_type, nw))) return rc;
        if ((rc = get(seq_off, e->d_seq_off, 8 * (ns + 1)))) return rc;
        if ((rc = get(seq_has_qual, e->d_has_qual, ns))) return rc;
        if ((rc = get(seq_begin, e->d_begin, 4 * ns))) return rc;
        if ((rc = get(seq_end, e->d_end, 4 * ns))) return rc;
        if ((rc = get(bases, e->d_bases, e->n_bases))) return rc;
        if ((rc = get(quals, e->d_quals, e->n_bases))) return rc;
        return RCN_OK;
    }
    // streamed batches (rcn_engine_polish*) are resident deepest first: device window k is caller window lpt[k].  The
    // copy handed out is in CALLER order, like every other view of the batch.
    std::vector<uint32_t> d_wso(nw + 1), d_bg(ns), d_en(ns);
    std::vector<uint64_t> d_so(ns + 1);
    std::vector<uint8_t> d_type(nw), d_hq(ns), d_b(bases ? e->n_bases : 0), d_q(quals ? e->n_bases : 0);
    {
        std::vector<uint8_t> meta(e->ml.end);
        if ((rc = get(meta.data(), e->d_meta, e->ml.end))) return rc;
        std::memcpy(d_wso.data(), meta.data() + e->ml.wso, 4 * (nw + 1)); std::memcpy(d_type.data(), meta.data() + e->ml.type, nw);
        std::memcpy(d_so.data(), meta.data() + e->ml.so, 8 * (ns + 1)); std::memcpy(d_hq.data(), meta.data() + e->ml.hq, ns);
        std::memcpy(d_bg.data(), meta.data() + e->ml.bg, 4 * ns); std::memcpy(d_en.data(), meta.data() + e->ml.en, 4 * ns);
    }
    if ((rc = get(d_b.data(), e->d_bases, d_b.size())) || (rc = get(d_q.data(), e->d_quals, d_q.size()))) return rc;
    std::vector<uint32_t> item_of(nw);
    for (uint32_t k = 0; k < nw; ++k) item_of[e->lpt[k]] = k;
    uint32_t so = 0; uint64_t bo = 0;
    if (win_seq_off) win_seq_off[0] = 0;
    if (seq_off) seq_off[0] = 0;
    for (uint32_t w = 0; w < nw; ++w) {
        const uint32_t k = item_of[w], s0 = d_wso[k], n = d_wso[k + 1] - s0;
        if (win_type) win_type[w] = d_type[k];
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t a = d_so[s0 + i], len = d_so[s0 + i + 1] - a;
            if (seq_has_qual) seq_has_qual[so] = d_hq[s0 + i];
            if (seq_begin) seq_begin[so] = d_bg[s0 + i];
            if (seq_end) seq_end[so] = d_en[s0 + i];
            if (bases && len) std::memcpy(bases + bo, d_b.data() + a, len);
            if (quals && len) std::memcpy(quals + bo, d_q.data() + a, len);
            bo += len; ++so;
            if (seq_off) seq_off[so] = bo;
        }
        if (win_seq_off) win_seq_off[w + 1] = so;
    }
    return RCN_OK;
}

// After the first pass over all work items: brings lengths / flags / consensus bytes back (one pinned staging buffer),
// re-runs overflowed windows with worst-case capacities (int32 kernel), reads the device counters and assembles the
// per-window results in caller order.
static int collect(rcn_engine* e) {
    const uint32_t nw = e->n_windows;
    const bool dbg = e->knobs.debug;
    const auto c0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count(); };
    const uint64_t cons_bytes = e->out_off[nw];
    // the kernel wrote its results into the pinned block (launch_pass): [lengths 4 nw][flags nw, padded][consensus bytes];
    // every launch of the pass has been waited for (events), so the bytes are here
    const ResultLayout rl = result_layout(nw);
    const uint64_t off_flags = rl.off_flags, off_cons = rl.off_cons;
    int rc;
    if (e->h_out.cap < off_cons + cons_bytes + 16) return RCN_E_STATE;
    uint8_t* hb = e->h_out.as<uint8_t>();
    if (dbg) fprintf(stderr, "[racon_hip] collect: results on the host (%.1f MB) after %.2f ms\n", (off_cons + cons_bytes) / 1e6, since());
    const uint32_t* len_item = reinterpret_cast<const uint32_t*>(hb);
    const uint8_t* flag_item = hb + off_flags;
    const uint8_t* raw = hb + off_cons;

    // outputs of the first pass are indexed by work item: back to window order
    std::vector<uint32_t> out_len(nw), item_of(nw);
    std::vector<uint8_t> flags(nw);
    for (uint32_t wi = 0; wi < nw; ++wi) { const uint32_t w = e->lpt[wi]; out_len[w] = len_item[wi]; flags[w] = flag_item[wi]; item_of[w] = wi; }
    // Windows the first pass flagged are re-run on the GPU, never on the CPU: a window the small-window kernel sent back
    // (outside its shape: poa_small.hpp) goes to poa_window_kernel2 with first-pass capacities, what that one flags -- or
    // what it flagged in the first place -- to the int32 kernel with worst-case capacities.
    // (per work item: a batch can be cut into pieces of which only some ran the small-window kernel -- what
    //  poa_window_kernel2 itself flagged goes straight to the int32 kernel, the same first-pass capacities again would overflow again)
    std::vector<uint32_t> retry, retry_k2;
    for (uint32_t w = 0; w < nw; ++w) {
        if (flags[w] & rcn::kFlagError) { fprintf(stderr, "[racon_hip] internal error on window %u\n", w); return RCN_E_STATE; }
        if (!(flags[w] & rcn::kFlagOverflow)) continue;
        if (e->pass_small && item_of[w] < e->item_small.size() && e->item_small[item_of[w]]) retry.push_back(w); else retry_k2.push_back(w);
    }
    std::vector<std::string> retry_cons;
    std::vector<uint32_t> retry_win;                 // windows whose bytes are in retry_cons, ascending
    {
        std::vector<std::pair<uint32_t, std::string>> redone;
        const uint32_t n_first_k2 = static_cast<uint32_t>(retry_k2.size());
        // (windows the small-window kernel sent back although they HAD its shape: a pass may hold up to an eighth of windows outside the
        //  shape from the start -- small_caps -- and those come back by design, they say nothing about the kernel's fit for the job)
        uint32_t n_first_small = 0;
        for (uint32_t w : retry) n_first_small += small_shape(e->shapes[w]) ? 1 : 0;
        uint32_t n_small_items = 0;
        for (uint8_t s : e->item_small) n_small_items += s;
        for (int tier = 0; tier < 2; ++tier) {
            if (tier == 1) { retry.insert(retry.end(), retry_k2.begin(), retry_k2.end()); std::sort(retry.begin(), retry.end()); }
            if (retry.empty()) continue;
            int32_t n2 = 0, l2 = 1, nsym = 2;
            const uint32_t nr = static_cast<uint32_t>(retry.size());
            std::vector<uint32_t> ids(nr);
            std::vector<uint64_t> off2(nr + 1, 0);
            std::vector<WinShape> sh(nr);
            for (size_t k = 0; k < nr; ++k) {
                const uint32_t w = retry[k];
                const auto& sw = e->shapes[w];
                sh[k] = sw;
                n2 = std::max<int32_t>(n2, sw.L + sw.sum_l + 8); l2 = std::max(l2, sw.lmax); nsym = std::max(nsym, sw.nsym);
                ids[k] = e->lpt_layout ? item_of[w] : w;                          // device window id
                off2[k + 1] = off2[k] + ((static_cast<uint64_t>(sw.L) + sw.sum_l + 8 + 15) & ~uint64_t(15));
            }
            const Caps c2 = tier == 0 ? first_pass_caps(sh.begin(), sh.end(), true, e->caps_level, e->knobs.hrows_div)
                                      : make_caps(n2, n2 + 8, std::max(1, nsym - 1), l2, false);     // int32 kernel, worst-case capacities
            // first-pass bytes are already on the host; a retry pass indexes its outputs by its own work items
            if ((rc = e->d_out_cons.reserve(off2[nr] + 16)) || (rc = e->d_out_len.reserve(4ull * nr)) || (rc = e->d_out_flags.reserve(nr))) return rc;
            if ((rc = upload_vec(e->d_retry_off, off2.data(), 8ull * (nr + 1), e->stream))) return rc;
            if ((rc = upload_vec(e->d_win_ids, ids.data(), 4ull * nr, e->stream))) return rc;
            if ((rc = run_pass(e, c2, e->d_win_ids.as<uint32_t>(), nr, /*host_out=*/false))) return rc;
            std::vector<uint32_t> len2(nr);
            std::vector<uint8_t> fl2(nr);
            HIP_TRY(hipMemcpyAsync(len2.data(), e->d_out_len.p, 4ull * nr, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipMemcpyAsync(fl2.data(), e->d_out_flags.p, nr, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            std::vector<uint32_t> again;
            const size_t first_new = redone.size();
            for (size_t k = 0; k < nr; ++k) {
                const uint32_t w = retry[k];
                if (fl2[k] & rcn::kFlagError) { fprintf(stderr, "[racon_hip] internal error on window %u (retry pass)\n", w); return RCN_E_STATE; }
                if (fl2[k] & rcn::kFlagOverflow) { if (tier == 1) return RCN_E_CAPACITY; again.push_back(w); continue; }
                redone.emplace_back(w, std::string(len2[k], '\0'));
                out_len[w] = len2[k]; flags[w] = fl2[k];
            }
            if (nr <= 64) {
                for (size_t k = 0, j = first_new; k < nr; ++k) {
                    if (fl2[k] & rcn::kFlagOverflow) continue;
                    if (len2[k]) HIP_TRY(hipMemcpyAsync(&redone[j].second[0], e->d_out_cons.as<uint8_t>() + off2[k], len2[k], hipMemcpyDeviceToHost, e->stream));
                    ++j;
                }
                HIP_TRY(hipStreamSynchronize(e->stream));
            } else {
                // many windows (a pass that the small-window kernel gave back wholesale): one copy of the block, not one per window
                // (250 000 of them were 4.5 s)
                std::vector<uint8_t> block(off2[nr] + 16);
                HIP_TRY(hipMemcpyAsync(block.data(), e->d_out_cons.p, off2[nr], hipMemcpyDeviceToHost, e->stream));
                HIP_TRY(hipStreamSynchronize(e->stream));
                for (size_t k = 0, j = first_new; k < nr; ++k) {
                    if (fl2[k] & rcn::kFlagOverflow) continue;
                    if (len2[k]) std::memcpy(&redone[j].second[0], block.data() + off2[k], len2[k]);
                    ++j;
                }
            }
            if (tier == 0) e->stats.n_small_bailed = nr;
            retry.swap(again);
        }
        std::sort(redone.begin(), redone.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        for (auto& r : redone) { retry_win.push_back(r.first); retry_cons.push_back(std::move(r.second)); }
        e->stats.n_retried = static_cast<uint32_t>(retry_win.size());
        // a batch whose windows keep leaving the small-window kernel (noisy reads on short windows: graphs that outgrow
        // the LDS) pays for them twice: not for this engine's next batches
        if (n_first_small * 8 > n_small_items && n_small_items >= 64) e->small_off = true;
        const uint32_t n_k2_items = nw - n_small_items;
        if (n_first_k2 * 50 > n_k2_items && n_k2_items >= 50 && e->caps_level < 2) ++e->caps_level;
    }

    e->stats_pending = true;                      // the device counters of this run: fetched by rcn_engine_stats on demand
#ifdef RCN_PROF_ROWS
    { static unsigned long long rp[256][20]; HIP_TRY(hipMemcpyFromSymbol(rp, HIP_SYMBOL(rcn::g_rowprof), sizeof(rp)));
      unsigned long long clk[9] = {0}, cnt[9] = {0}, allc = 0, alln = 0;
      for (int b = 0; b < 256; ++b) for (int k = 0; k < 9; ++k) { clk[k] += rp[b][k]; cnt[k] += rp[b][10 + k]; }
      for (int k = 0; k < 9; ++k) { allc += clk[k]; alln += cnt[k]; }
      static const char* names[9] = {"chain", "fast, 1 pred", "fast, 2 preds", "fast, 3-4 preds", "medium", "general", "window moves", "sink", "rows in octets"};
      fprintf(stderr, "[racon_hip] banded DP rows since the library was loaded (clocks include the probe itself):\n");
      for (int k = 0; k < 9; ++k) if (cnt[k]) fprintf(stderr, "  %-16s rows %5.1f %%  clocks %5.1f %%  %7.0f clocks/row\n", names[k], 100.0 * cnt[k] / std::max(1ull, alln),
                                                         100.0 * clk[k] / std::max(1ull, allc), (double)clk[k] / cnt[k]);
      fprintf(stderr, "  all              rows %llu  %7.0f clocks/row\n", alln, (double)allc / std::max(1ull, alln));
      static unsigned long long sp[256][40]; HIP_TRY(hipMemcpyFromSymbol(sp, HIP_SYMBOL(rcn::g_secprof), sizeof(sp)));
      static const char* secs[15] = {"shift: edge flush + certificate of the dropped cells", "shift: register window re-based (bpermutes)", "shift: cold state updated", "shift: columns (bases, thresholds)",
                                     "shift: profile tables", "shift: rest", "general row: cold state, descriptor", "general row: first predecessor", "general row: further predecessors", "general row: rest of the in-edge list",
                                     "general / medium tail: candidates + scan", "general / medium tail: codes + store", "general / medium tail: ring + window", "general / medium tail: sink part + end", "general row in one round trip (round 6)"};
      for (int k = 0; k < 15; ++k) { unsigned long long c_ = 0, n_ = 0; for (int b = 0; b < 256; ++b) { c_ += sp[b][k]; n_ += sp[b][20 + k]; }
        if (n_) fprintf(stderr, "  section %-56s %9llu times  %7.0f clocks each\n", secs[k], n_, (double)c_ / n_); } }
#endif
#ifdef RCN_PROF_WIN
    { unsigned long long ws[8]; HIP_TRY(hipMemcpyFromSymbol(ws, HIP_SYMBOL(rcn::g_wsub), sizeof(ws)));
      if (ws[4]) fprintf(stderr, "[racon_hip] Subgraph sweep (every 16th window since load): %llu calls, clocks per call: set-up %.0f, pass A %.0f, pass B %.0f (%.1f chunks), pass C %.0f; %.0f ranks swept of %.0f nodes\n",
                         ws[4], (double)ws[0] / ws[4], (double)ws[1] / ws[4], (double)ws[2] / ws[4], (double)ws[5] / ws[4], (double)ws[3] / ws[4], (double)ws[6] / ws[4], (double)ws[7] / ws[4]); }
    { static unsigned long long wt[4096][8]; HIP_TRY(hipMemcpyFromSymbol(wt, HIP_SYMBOL(rcn::g_wtb), sizeof(wt)));
      unsigned long long dg[8] = {0}; for (int w = 0; w < 4096; ++w) for (int k = 0; k < 8; ++k) dg[k] += wt[w][k];
      fprintf(stderr, "[racon_hip] code traceback, first 4096 work items since load: %llu tiles, clocks per tile: issue loads %.0f, wait + barrier %.0f, walk %.0f\n", dg[3],
              (double)dg[0] / std::max(1ull, dg[3]), (double)dg[1] / std::max(1ull, dg[3]), (double)dg[2] / std::max(1ull, dg[3]));
      fprintf(stderr, "[racon_hip]   %llu boxes, clocks per box: decode %.0f, walk %.0f, emit + next anchor %.0f\n", dg[7], (double)dg[4] / std::max(1ull, dg[7]),
              (double)dg[5] / std::max(1ull, dg[7]), (double)dg[6] / std::max(1ull, dg[7]));
      static unsigned long long w2[4096][8]; HIP_TRY(hipMemcpyFromSymbol(w2, HIP_SYMBOL(rcn::g_wtb2), sizeof(w2)));
      unsigned long long ex[8] = {0}; for (int w = 0; w < 4096; ++w) for (int k = 0; k < 8; ++k) ex[k] += w2[w][k];
      fprintf(stderr, "[racon_hip]   boxes left at: tile edge %.1f %%, origin %.1f %%, columns used up %.1f %%, climbed 1-2 box heights %.1f %%, below the skew line %.1f %%, climbed more %.1f %%; %.2f cells walked per box\n",
              100.0 * ex[0] / std::max(1ull, dg[7]), 100.0 * ex[1] / std::max(1ull, dg[7]), 100.0 * ex[2] / std::max(1ull, dg[7]), 100.0 * ex[3] / std::max(1ull, dg[7]),
              100.0 * ex[4] / std::max(1ull, dg[7]), 100.0 * ex[5] / std::max(1ull, dg[7]), (double)ex[6] / std::max(1ull, dg[7])); }
#endif
#ifdef RCN_PROF_WIN
    { static unsigned long long wc[4096][8]; HIP_TRY(hipMemcpyFromSymbol(wc, HIP_SYMBOL(rcn::g_wclk), sizeof(wc)));
      std::vector<std::pair<unsigned long long, int>> tot;
      for (int w = 0; w < 4096 && w < (int)nw; ++w) { unsigned long long s = 0; for (int k = 0; k < 7; ++k) s += wc[w][k]; tot.push_back({s, w}); }
      std::sort(tot.rbegin(), tot.rend());
      unsigned long long all = 0; for (auto& p : tot) all += p.first;
      fprintf(stderr, "[racon_hip] per-window clocks: mean %.3g\n", (double)all / std::max<size_t>(1, tot.size()));
      for (int k = 0; k < 3 && k < (int)tot.size(); ++k) { int w = tot[k].second; fprintf(stderr, "  work item %d (%u seqs): total %.3g | sub %.3g desc %.3g dp %.3g tb %.3g add %.3g merge %.3g cons %.3g | tiles %llu boxes %llu slow %llu\n", w,
          e->h_win_seq_off[e->lpt[w] + 1] - e->h_win_seq_off[e->lpt[w]], (double)tot[k].first, (double)wc[w][0], (double)wc[w][1], (double)wc[w][2], (double)wc[w][3], (double)wc[w][4], (double)wc[w][5], (double)wc[w][6], wc[w][7] >> 40, (wc[w][7] >> 20) & 0xfffff, wc[w][7] & 0xfffff); }
      { int w = tot[tot.size() / 2].second; fprintf(stderr, "  median work item %d: total %.3g | sub %.3g desc %.3g dp %.3g tb %.3g add %.3g merge %.3g cons %.3g\n", w, (double)tot[tot.size() / 2].first,
          (double)wc[w][0], (double)wc[w][1], (double)wc[w][2], (double)wc[w][3], (double)wc[w][4], (double)wc[w][5], (double)wc[w][6]); }
      { unsigned long long wt_[8]; HIP_TRY(hipMemcpyFromSymbol(wt_, HIP_SYMBOL(rcn::g_wtie), sizeof(wt_)));
        fprintf(stderr, "[racon_hip] sink ties since load: %llu alignments, %.0f clocks each; %llu past the rule, %llu closure sweeps, %llu full DFS\n", wt_[0], (double)wt_[1] / std::max(1ull, wt_[0]), wt_[2], wt_[3], wt_[4]); }
      { unsigned long long wh_[8]; HIP_TRY(hipMemcpyFromSymbol(wh_, HIP_SYMBOL(rcn::g_whelp), sizeof(wh_)));
        fprintf(stderr, "[racon_hip] code waves since load: %llu lap checks of wave 0 with %llu polls; %llu waits of the code waves with %llu polls\n", wh_[0], wh_[1], wh_[2], wh_[3]); }
      if (e->knobs.prof_layers) {
          static unsigned long long wl[4][128][5]; HIP_TRY(hipMemcpyFromSymbol(wl, HIP_SYMBOL(rcn::g_wlay), sizeof(wl)));
          for (int w = 0; w < 4; ++w) for (int j = 1; j < 128; ++j) if (wl[w][j][0])
              fprintf(stderr, "  item %d layer %3d: V %4llu len %4llu | dp %8llu tie %8llu tb %8llu | tied %llu level %llu band-flags %llu\n", w, j, wl[w][j][4] >> 32, wl[w][j][4] & 0xffffffffull,
                      wl[w][j][0], wl[w][j][1], wl[w][j][2], wl[w][j][3] >> 16, (wl[w][j][3] >> 8) & 255, wl[w][j][3] & 255);
      } }
#endif

    e->cons_off.assign(static_cast<size_t>(nw) + 1, 0);
    for (uint32_t w = 0; w < nw; ++w) e->cons_off[w + 1] = e->cons_off[w] + out_len[w];
    e->cons.resize(e->cons_off[nw] + 1);
    e->polished.assign(nw, 0); e->chimeric.assign(nw, 0);
    size_t rk = 0;
    for (uint32_t w = 0; w < nw; ++w) {
        uint8_t* dst = e->cons.data() + e->cons_off[w];
        if (rk < retry_win.size() && retry_win[rk] == w) { std::memcpy(dst, retry_cons[rk].data(), out_len[w]); ++rk; }
        else std::memcpy(dst, raw + e->out_off[item_of[w]], out_len[w]);
        e->polished[w] = (flags[w] & rcn::kFlagPolished) ? 1 : 0;
        e->chimeric[w] = (flags[w] & rcn::kFlagChimeric) ? 1 : 0;
    }
    e->stats.bytes_out = e->cons_off[nw] + 5ull * nw;
    e->ran = true;
    if (dbg) fprintf(stderr, "[racon_hip] collect: results in window order after %.2f ms\n", since());
    return RCN_OK;
}

static int begin_run(rcn_engine* e) {
    const uint32_t nw = e->n_windows;
    const double h2d = e->stats.h2d_ms; const uint64_t bin = e->stats.bytes_in;
    e->stats = rcn_run_stats{}; e->stats.h2d_ms = h2d; e->stats.bytes_in = bin;
    { size_t fr = 0, tot = 0; HIP_TRY(hipMemGetInfo(&fr, &tot)); e->free_mem = fr + e->d_scratch.cap; }
    int rc;
    if ((rc = e->h_out.reserve(result_layout(nw).off_cons + e->out_off[nw] + 16))) return rc;
    e->stats_pending = false;
    e->pass_small = false;
    e->item_small.assign(nw, 0);
    HIP_TRY(hipMemsetAsync(e->d_ctr.p, 0, kCtrBytes, e->stream));
    {
        // windows in the top tail of the depth distribution can be given the 4-wave DP (RCN_HEAVY_PCT; off by default:
        // measured slower).  Threshold = a percentile of sequences per window.
        const double pct = e->knobs.heavy_pct;
        e->heavy_ns = 0;
        if (pct < 1.0 && nw) {
            std::vector<uint32_t> depth(nw);
            for (uint32_t w = 0; w < nw; ++w) depth[w] = e->h_win_seq_off[w + 1] - e->h_win_seq_off[w];
            const size_t kth = std::min<size_t>(nw - 1, static_cast<size_t>(std::max(0.0, pct) * nw));
            std::nth_element(depth.begin(), depth.begin() + kth, depth.end());
            e->heavy_ns = pct <= 0.0 ? 1 : static_cast<int32_t>(std::max<uint32_t>(depth[kth], 3));
        }
    }
    return RCN_OK;
}

int rcn_engine_forget(rcn_engine* e) {
    if (!e) return RCN_E_ARG;
    e->caps_level = 0; e->small_off = false; e->pc_valid = false; e->queued = false;
    return RCN_OK;
}

}  // extern "C"

namespace {

// Internal view of a batch: one pointer per sequence.  rcn_batch (contiguous bytes) and rcn_window_refs (borrowed
// pointers, what a racon::Window holds) both map onto it.
struct SrcView {
    uint32_t nw = 0, ns = 0, flags = 0;
    const uint32_t* win_seq_off = nullptr; const uint8_t* win_type = nullptr;
    const uint64_t* seq_off = nullptr;                  // [ns + 1] prefix sums of the sequence lengths
    const uint8_t* const* seq = nullptr;                // [ns]
    const uint8_t* const* qual = nullptr;               // [ns] nullptr = no quality
    const uint32_t* begin = nullptr; const uint32_t* end = nullptr;
};

// contiguous copy of a view (the plain upload path of a batch that came as pointers)
struct HostBatch { std::vector<uint8_t> has_qual, bases, quals; rcn_batch b{}; };
void materialize(const SrcView& v, HostBatch& h) {
    const uint64_t nb = v.ns ? v.seq_off[v.ns] : 0;
    h.has_qual.resize(std::max<size_t>(1, v.ns)); h.bases.resize(nb + 1); h.quals.resize(nb + 1);
    for (uint32_t i = 0; i < v.ns; ++i) {
        const uint64_t a = v.seq_off[i], len = v.seq_off[i + 1] - a;
        if (len) std::memcpy(h.bases.data() + a, v.seq[i], len);
        h.has_qual[i] = v.qual[i] ? 1 : 0;
        if (!len) continue;
        if (v.qual[i]) std::memcpy(h.quals.data() + a, v.qual[i], len); else std::memset(h.quals.data() + a, '!', len);
    }
    h.b.n_windows = v.nw; h.b.n_seqs = v.ns; h.b.win_seq_off = v.win_seq_off; h.b.win_type = v.win_type; h.b.seq_off = v.seq_off;
    h.b.seq_has_qual = h.has_qual.data(); h.b.seq_begin = v.begin; h.b.seq_end = v.end; h.b.bases = h.bases.data(); h.b.quals = h.quals.data();
}

// The launches of a first pass: cut[c] .. cut[c + 1] are the work items of piece c (deepest first).  Fills the Launch
// records (capacities from the pieces' own shapes, slots, scratch placement); false when the scratch does not fit.
struct PassPlan { int n = 0; uint32_t cut[rcn_engine::kMaxLaunches + 1] = {}; Launch L[rcn_engine::kMaxLaunches]; int tier[rcn_engine::kMaxLaunches] = {0, 2, 2};   /* split launch: 0 deep, 1 middle, 2 rest */ uint64_t scratch = 0; bool split = false, copied = false; };

// `estimate`: a plan made for sizing only (every window taken for the usual alphabet): it leaves the engine's record of which work
// items the small-window kernel polishes alone -- collect() sends a flagged window to the retry tier by that record.
void plan_piece(rcn_engine* e, PassPlan& pp, int c, const SplitPlan& sp, bool fast, uint32_t slots_left, bool estimate = false) {
    std::vector<WinShape> sh;
    sh.reserve(pp.cut[c + 1] - pp.cut[c]);
    for (uint32_t k = pp.cut[c]; k < pp.cut[c + 1]; ++k) sh.push_back(e->shapes[e->lpt[k]]);
    Launch& L = pp.L[c];
    const bool small = fast && !sp.on && small_caps(e, sh.begin(), sh.end(), L.c);
    if (!small) L.c = first_pass_caps(sh.begin(), sh.end(), fast, e->caps_level, e->knobs.hrows_div);
    L.n_work = pp.cut[c + 1] - pp.cut[c]; L.work_base = pp.cut[c]; L.out_base = pp.cut[c]; L.ctr = c;
    if (!estimate) {
        if (small) e->pass_small = true;
        for (uint32_t k = pp.cut[c]; k < pp.cut[c + 1] && k < e->item_small.size(); ++k) e->item_small[k] = small ? 1 : 0;
    }
    if (small) {
        L.stream = e->sub_stream[c]; L.per_cu = L.c.per_cu;
        // (every piece may fill the device: its work-groups are persistent over a queue, the ones that find no room at first
        //  start as the earlier piece's retire -- a fixed share would idle once that piece is done)
        L.slots = e->cfg.max_slots ? std::min(L.n_work, slots_left) : balanced_slots(L.n_work, static_cast<uint32_t>(e->n_cu) * L.c.per_cu);
    } else if (sp.on) {
        const int tier = pp.tier[c];
        const bool three = pp.n == 3;
        L.stream = tier == 0 ? e->deep_stream : tier == 1 ? e->mid_stream : (three ? e->rest3_stream : e->rest_stream);
        L.per_cu = tier == 0 ? sp.deep_per_cu : tier == 1 ? sp.mid_per_cu : sp.rest_per_cu;
        if (tier == 0 && e->knobs.split_deep_wide >= 0) L.heavy_ns = e->knobs.split_deep_wide;
        const uint32_t cus = tier == 0 ? static_cast<uint32_t>(e->split_cus) : tier == 1 ? static_cast<uint32_t>(e->split_mid_cus)
                                       : static_cast<uint32_t>(e->n_cu - e->split_cus - (three ? e->split_mid_cus : 0));
        L.slots = std::min(L.n_work, cus * L.per_cu);
    } else {
        L.stream = e->sub_stream[c]; L.per_cu = 0;
        L.slots = std::min(L.n_work, slots_left);
    }
    L.scratch_off = pp.scratch;
    pp.scratch += static_cast<uint64_t>(L.slots) * L.c.slot_bytes;
}

// kernel begin / end events of the pieces -> run statistics (the overlapping launches as one interval)
int finish_pieces(rcn_engine* e, const PassPlan& pp, hipEvent_t ref) {
    float span0 = 0, span1 = 0;
    bool have = false;
    for (int c = 0; c < pp.n; ++c) {
        if (pp.L[c].n_work == 0) continue;
        HIP_TRY(hipEventSynchronize(e->sub_ev[c][2]));
        float a0 = 0, a1 = 0;       // begin / end of this launch relative to `ref`
        HIP_TRY(hipEventElapsedTime(&a0, ref, e->sub_ev[c][1]));
        HIP_TRY(hipEventElapsedTime(&a1, ref, e->sub_ev[c][2]));
        if (!have) { span0 = a0; span1 = a1; have = true; } else { span0 = std::min(span0, a0); span1 = std::max(span1, a1); }
        if (e->knobs.debug) {
            float cp = 0; if (pp.copied) (void)hipEventElapsedTime(&cp, ref, e->sub_ev[c][0]);
            fprintf(stderr, "[racon_hip] piece %d: work items [%u, %u) slots %u (%u per CU) ncap %d lmax %d | copy done %.2f ms, kernel %.2f .. %.2f ms\n",
                    c, pp.cut[c], pp.cut[c + 1], pp.L[c].slots, pp.L[c].per_cu, pp.L[c].c.ncap, pp.L[c].c.lmax, cp, a0, a1);
        }
        if (pp.split && pp.n == 3) { if (pp.tier[c] == 1) e->stats.launch_ms_mid = a1 - a0; else e->stats.launch_ms[pp.tier[c] == 0 ? 0 : 1] = a1 - a0; }
        else e->stats.launch_ms[c] = a1 - a0;
        e->stats.n_launches += 1;
    }
    e->stats.kernel_ms += span1 - span0;
    e->stats.split_deep = pp.split ? pp.L[0].n_work : 0;
    e->stats.split_cus = pp.split ? static_cast<uint32_t>(e->split_cus) : 0;
    e->stats.split_deep_per_cu = pp.split ? pp.L[0].per_cu : 0;
    e->stats.split_mid = (pp.split && pp.n == 3) ? pp.L[1].n_work : 0;
    e->stats.split_mid_cus = (pp.split && pp.n == 3) ? static_cast<uint32_t>(e->split_mid_cus) : 0;
    e->stats.split_mid_per_cu = (pp.split && pp.n == 3) ? pp.L[1].per_cu : 0;
    return RCN_OK;
}

}  // namespace

extern "C" {

}  // extern "C"
namespace { __global__ void k_warm_out(uint32_t* p); }
extern "C" {
// The resident batch (rcn_engine_upload / rcn_engine_build_windows*) through the consensus kernels.  `dry`: everything the run
// would allocate -- the scratch of its launches as it will plan them, the pinned result block -- and nothing else
// (rcn_engine_reserve_run).
static int run_resident(rcn_engine* e, bool dry) {
    if (!e) return RCN_E_ARG;
    if (!e->uploaded) return RCN_E_STATE;
    HIP_TRY(hipSetDevice(e->cfg.device));
    const uint32_t nw = e->n_windows;
    if (nw == 0) {
        if (dry) return RCN_OK;
        e->stats = rcn_run_stats{};
        e->cons_off.assign(1, 0); e->polished.clear(); e->chimeric.clear(); e->cons.assign(1, 0); e->ran = true;
        return RCN_OK;
    }
    int rc;
    const auto r0__ = std::chrono::steady_clock::now();
    if ((rc = begin_run(e))) return rc;
    if (e->knobs.debug && !dry) fprintf(stderr, "[racon_hip] run: begin_run (result block, counters) took %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0__).count());
    auto warm_out = [&]() -> int {
        // the arena's first touch (what AlignmentEngine::Prealloc does to its matrices, reference src/polisher.cpp:180-182): the
        // launches of the run then find its pages mapped and their translations fresh
        if (e->d_scratch.p && e->d_scratch.cap) HIP_TRY(hipMemsetAsync(e->d_scratch.p, 0, e->d_scratch.cap, e->stream));
        // the kernel's first stores into the (new) pinned result block, as polish_view's dry run makes them
        hipLaunchKernelGGL(k_warm_out, dim3(1), dim3(64), 0, e->stream, e->h_out.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(e->stream));
        return RCN_OK;
    };
    const bool fast = !e->knobs.wide_only;
    const uint32_t* ids = e->lpt_layout ? nullptr : e->d_lpt_ids.as<uint32_t>();
    {
        Caps cs;
        if (fast && small_caps(e, e->shapes.begin(), e->shapes.end(), cs)) {
            if (dry) {
                const uint32_t slots = slots_for(e, cs, nw, scratch_budget(e));
                if (slots && (rc = e->d_scratch.reserve(static_cast<uint64_t>(slots) * cs.slot_bytes))) return rc;
                return warm_out();
            }
            e->pass_small = true;
            e->item_small.assign(nw, 1);
            if ((rc = run_pass(e, cs, ids, nw))) return rc;
            return collect(e);
        }
    }
    const SplitPlan sp = split_plan(e, nw, fast);
    if (sp.on) {
        // the split pair: deepest windows on their own CUs, everything else on the rest of the chip
        PassPlan pp; pp.split = true; pp.cut[0] = 0; pp.cut[1] = sp.n_deep;
        if (sp.n_mid > 0) { pp.n = 3; pp.cut[2] = sp.n_deep + sp.n_mid; pp.cut[3] = nw; pp.tier[0] = 0; pp.tier[1] = 1; pp.tier[2] = 2; }
        else { pp.n = 2; pp.cut[2] = nw; pp.tier[0] = 0; pp.tier[1] = 2; }
        for (int c = 0; c < pp.n; ++c) { plan_piece(e, pp, c, sp, fast, 0); pp.L[c].d_ids = ids ? ids + pp.cut[c] : nullptr; }
        if (pp.scratch <= scratch_budget(e)) {
            if ((rc = e->d_scratch.reserve(pp.scratch))) return rc;
            if (dry) return warm_out();
            HIP_TRY(hipEventRecord(e->ev0, e->stream));             // the counters were zeroed on the main stream (begin_run)
            for (int c = 0; c < pp.n; ++c) {
                HIP_TRY(hipStreamWaitEvent(pp.L[c].stream, e->ev0, 0));
                HIP_TRY(hipEventRecord(e->sub_ev[c][1], pp.L[c].stream));
                if ((rc = launch_pass(e, pp.L[c]))) return rc;
                HIP_TRY(hipEventRecord(e->sub_ev[c][2], pp.L[c].stream));
            }
            if ((rc = finish_pieces(e, pp, e->ev0))) return rc;
            return collect(e);
        }
    }
    const Caps c1 = first_pass_caps(e->shapes.begin(), e->shapes.end(), fast, e->caps_level, e->knobs.hrows_div);
    if (dry) {
        const uint32_t slots = slots_for(e, c1, nw, scratch_budget(e));
        if (slots && (rc = e->d_scratch.reserve(static_cast<uint64_t>(slots) * c1.slot_bytes))) return rc;
        return warm_out();
    }
    if ((rc = run_pass(e, c1, ids, nw))) return rc;
    if (e->knobs.debug) fprintf(stderr, "[racon_hip] run: pass done %.2f ms after the call\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0__).count());
    return collect(e);
}

int rcn_engine_run(rcn_engine* e) { return run_resident(e, false); }

int rcn_engine_reserve_run(rcn_engine* e) {
    if (!e) return RCN_E_ARG;
    rcn_reserve_hint warm{};
    const int rc = rcn_engine_reserve(e, &warm);                        // first use of the code object / streams / copy engines
    return rc ? rc : run_resident(e, true);
}

}  // extern "C"

namespace {

__global__ void k_warm_out(uint32_t* p) { p[threadIdx.x] = 0u; }

// Which byte values occur (256 presence bits).  A/C/G/T-only stretches -- all of a window, usually -- are taken 32 bytes
// at a time with AVX2 where the host has it (four compares per block; a block with anything else goes byte by byte): the
// byte loop alone was most of what packing a batch cost (2 ms of the 2.9 ms before cfg2's second launch could start).
inline void symbols_add_bytes(uint64_t present[4], const uint8_t* p, uint64_t n) {
    for (uint64_t k = 0; k < n; ++k) present[p[k] >> 6] |= 1ull << (p[k] & 63);
}
__attribute__((target("avx2"))) void symbols_add_avx2(uint64_t present[4], const uint8_t* p, uint64_t n) {
    const __m256i A = _mm256_set1_epi8('A'), C = _mm256_set1_epi8('C'), G = _mm256_set1_epi8('G'), T = _mm256_set1_epi8('T');
    __m256i sa = _mm256_setzero_si256(), sc = sa, sg = sa, st = sa;
    uint64_t k = 0;
    for (; k + 32 <= n; k += 32) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + k));
        const __m256i ma = _mm256_cmpeq_epi8(v, A), mc = _mm256_cmpeq_epi8(v, C), mg = _mm256_cmpeq_epi8(v, G), mt = _mm256_cmpeq_epi8(v, T);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(ma, mc), _mm256_or_si256(mg, mt));
        if (static_cast<uint32_t>(_mm256_movemask_epi8(ok)) != 0xffffffffu) symbols_add_bytes(present, p + k, 32);
        sa = _mm256_or_si256(sa, ma); sc = _mm256_or_si256(sc, mc); sg = _mm256_or_si256(sg, mg); st = _mm256_or_si256(st, mt);
    }
    if (!_mm256_testz_si256(sa, sa)) present['A' >> 6] |= 1ull << ('A' & 63);
    if (!_mm256_testz_si256(sc, sc)) present['C' >> 6] |= 1ull << ('C' & 63);
    if (!_mm256_testz_si256(sg, sg)) present['G' >> 6] |= 1ull << ('G' & 63);
    if (!_mm256_testz_si256(st, st)) present['T' >> 6] |= 1ull << ('T' & 63);
    symbols_add_bytes(present, p + k, n - k);
}
inline void symbols_add(uint64_t present[4], const uint8_t* p, uint64_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 64) symbols_add_avx2(present, p, n); else symbols_add_bytes(present, p, n);
}
inline void symbols_finish(const uint64_t present[4], int32_t& nsym, uint8_t& acgt_only) {
    nsym = 0;
    for (int k = 0; k < 4; ++k) nsym += __builtin_popcountll(present[k]);
    const uint64_t acgt = (1ull << ('A' & 63)) | (1ull << ('C' & 63)) | (1ull << ('G' & 63)) | (1ull << ('T' & 63));
    acgt_only = (present[0] == 0 && present[2] == 0 && present[3] == 0 && (present[1] & ~acgt) == 0) ? 1 : 0;
}

// Upload + run of one batch with the copy hidden behind the kernel (what Polisher::polish pays per batch; reference
// src/cuda/cudapolisher.cpp:254-333 fills and runs a batch strictly one after the other).  The windows are packed
// DEEPEST FIRST into pinned staging by host threads -- straight from the caller's per-sequence pointers --, go to HBM in
// kSubLaunches pieces on a copy stream, and every piece is polished by its own launch on its own stream as soon as it
// has arrived: the launches overlap on the device, the deepest windows (which decide when the batch ends) start after
// the first, small piece.  With the split plan the first piece is the deep launch on its own CUs.  A batch with more
// windows than resident slots runs the same way, its launches persistent over their queues.  Same results as
// upload + run.
// `dry`: everything polish_view(v) would allocate -- device inputs, pinned staging, result buffers, the scratch of its
// launches (their slots and capacities planned from v's own shapes) -- and nothing else: rcn_engine_reserve_refs.
inline bool dbg_prep(const rcn_engine* e) { return e->knobs.debug; }
int polish_view(rcn_engine* e, const SrcView& v, bool dry = false) {
    const uint32_t nw = v.nw, ns = v.ns;
    HIP_TRY(hipSetDevice(e->cfg.device));
    e->queued = (v.flags & RCN_REFS_QUEUED) != 0;
    auto plain = [&]() -> int {
        HostBatch hb; materialize(v, hb);
        const int rc = rcn_engine_upload(e, &hb.b);
        e->queued = (v.flags & RCN_REFS_QUEUED) != 0;
        return rc ? rc : rcn_engine_run(e);
    };
    const bool fast = !e->knobs.wide_only;
    if (nw < 64 || e->knobs.no_stream) return dry ? RCN_OK : plain();
    e->uploaded = false; e->ran = false;
    e->n_windows = nw; e->n_seqs = ns; e->n_bases = v.seq_off[ns];
    const bool dbg = e->knobs.debug;
    const auto h0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count(); };
    EventPair t;
    if (t.create()) return RCN_E_HIP;
    HIP_TRY(hipEventRecord(t.a, e->copy_stream));
    HostPrep hp;
    int rc = RCN_OK;
    {
        // identity of the batch: sizes, total bases, and the sequence POINTERS at 64 sampled positions (a dry run and the call it was
        // made for pass the same tables; one use, then the entry is gone)
        uint64_t key[4] = {(static_cast<uint64_t>(nw) << 32) | ns, v.seq_off[ns], 1469598103934665603ull, reinterpret_cast<uintptr_t>(v.seq)};
        for (uint32_t q = 0; q < 64 && ns; ++q) {
            const uint32_t i = static_cast<uint32_t>((static_cast<uint64_t>(q) * (ns - 1)) / 63);
            key[2] = (key[2] ^ reinterpret_cast<uintptr_t>(v.seq[i]) ^ (static_cast<uint64_t>(v.begin[i]) << 48) ^ (v.seq_off[i + 1] - v.seq_off[i])) * 1099511628211ull;
        }
        // ... and everything the preparation is computed FROM, in full: the windows' sequence counts, every layer's length, begin and end
        // (a caller that edits its tables between the dry run and the call gets a fresh preparation, not the old one)
        {
            uint64_t h = 14695981039346656037ull;
            for (uint32_t w = 0; w <= nw; ++w) h = (h ^ v.win_seq_off[w]) * 1099511628211ull;
            for (uint32_t i = 0; i < ns; ++i) h = (h ^ (v.seq_off[i + 1] - v.seq_off[i]) ^ (static_cast<uint64_t>(v.begin[i]) << 20) ^ (static_cast<uint64_t>(v.end[i]) << 42)) * 1099511628211ull;
            key[3] ^= h;
        }
        const bool hit = !dry && e->pc_valid && std::memcmp(key, e->pc_key, sizeof(key)) == 0 && e->pc_order.size() == ns && e->shapes.size() == nw && e->lpt.size() == nw;
        e->pc_valid = false;
        if (hit) {
            hp.order.swap(e->pc_order); hp.full.swap(e->pc_full); hp.wflags.assign(nw, 0);
            e->h_win_seq_off.assign(v.win_seq_off, v.win_seq_off + nw + 1);
        } else {
            rc = prepare_host(e, hp, nw, ns, v.win_seq_off, v.seq_off, v.begin, v.end, nullptr, 0, false, /*scan=*/false);
            if (rc) return rc;
            if (dry) { e->pc_order = hp.order; e->pc_full = hp.full; std::memcpy(e->pc_key, key, sizeof(key)); e->pc_valid = true; }
        }
        if (dbg_prep(e)) fprintf(stderr, "[racon_hip] polish: host preparation %s\n", hit ? "taken over from the dry run" : "computed");
    }
    e->lpt_layout = true;
    e->stats = rcn_run_stats{};
    if (dbg) fprintf(stderr, "[racon_hip] polish: host preparation done at %.2f ms\n", since());
    if ((rc = begin_run(e))) return rc;

    // ---- device layout: window k of the device arrays = work item k = caller window lpt[k] ----
    const uint64_t nb = e->n_bases;
    // staging: metadata block first, then bases, then qualities (each in deepest-first order)
    const MetaLayout ml = meta_layout(nw, ns);
    const uint64_t o_wso = ml.wso, o_type = ml.type, o_flags = ml.flags, o_so = ml.so, o_hq = ml.hq, o_bg = ml.bg, o_en = ml.en, o_ord = ml.ord,
                   o_full = ml.full, o_ooff = ml.ooff, o_bases = ml.end, o_quals = (o_bases + nb + 255) & ~uint64_t(255), total = o_quals + nb + 256;
    e->ml = ml;
    if ((rc = e->h_stage.reserve(total))) return rc;
    if (dbg) fprintf(stderr, "[racon_hip] polish: pinned staging (%.1f MB) at %.2f ms\n", total / 1e6, since());
    if ((rc = e->d_meta.reserve(ml.end + kDmaCopyBytes)) || (rc = e->d_bases.reserve(nb + 16)) || (rc = e->d_quals.reserve(nb + 16))) return rc;
    if (dbg) fprintf(stderr, "[racon_hip] polish: device inputs reserved at %.2f ms\n", since());
    SplitPlan sp = split_plan(e, nw, fast);
    uint32_t slots_total = e->cfg.max_slots ? e->cfg.max_slots : static_cast<uint32_t>(e->n_cu) * wg_per_cu(e);
    {
        // a batch of small windows: no split (its windows are a few ms each, the launch does not hang on one of them), the
        // pieces go to the small-window kernel (plan_piece) with its own residency
        Caps cs;
        if (fast && small_caps(e, e->shapes.begin(), e->shapes.end(), cs)) {
            sp = SplitPlan{};
            if (!e->cfg.max_slots) slots_total = static_cast<uint32_t>(e->n_cu) * cs.per_cu;
        }
    }
    if (dry) {
        // the launches' scratch as polish_view will lay it out (usual alphabet: A, C, G, T and one more symbol), and the pinned
        // result buffer of collect()
        if (slots_total < 2) return RCN_OK;
        PassPlan est; est.n = rcn_engine::kSubLaunches; est.split = sp.on;
        est.cut[0] = 0; est.cut[2] = nw;
        if (sp.on) est.cut[1] = sp.n_deep;
        else {
            // (the same cut as the call itself makes below: a different plan is a different arena, grown inside polish())
            Caps cs_;
            const bool small_ = fast && small_caps(e, e->shapes.begin(), e->shapes.end(), cs_);
            const uint64_t q1 = small_ ? nb / 3 : nb / 24;
            uint64_t acc = 0; uint32_t k = 0;
            while (k < nw && acc < q1) { const uint32_t w = e->lpt[k], s0 = v.win_seq_off[w]; acc += v.seq_off[v.win_seq_off[w + 1]] - v.seq_off[s0]; ++k; }
            est.cut[1] = std::min(std::min(std::max(k, 1u), nw), std::max(1u, slots_total / 2));
            if (small_ && nw > slots_total && !e->cfg.max_slots) est.cut[1] = slots_total;
        }
        for (uint32_t w = 0; w < nw; ++w) e->shapes[w].nsym = 5;
        uint32_t left = slots_total;
        for (int c = 0; c < est.n; ++c) { if (est.cut[c + 1] == est.cut[c]) continue; plan_piece(e, est, c, sp, fast, left, /*estimate=*/true); left -= std::min(left, est.L[c].slots); }
        if (est.scratch <= scratch_budget(e) && (rc = e->d_scratch.reserve(est.scratch))) return rc;
        if (dbg) fprintf(stderr, "[racon_hip] reserve: scratch arena (%.2f GB, %u + %u slots) at %.2f ms\n", est.scratch / 1e9, est.L[0].slots, est.L[1].slots, since());
        if ((rc = e->h_out.reserve(result_layout(nw).off_cons + e->out_off[nw] + 16))) return rc;
        {
            // The first copies out of a new pinned block into new device buffers cost 8-30 ms on their stream (measured:
            // profiles/r03/c_timeline_*, the same copies take 0.05 ms on the second batch): make them here, with whatever
            // the staging block holds -- the real call overwrites every byte.
            uint8_t* hs = e->h_stage.as<uint8_t>();
            hipStream_t cs = e->copy_stream;
            const uint64_t part = std::min<uint64_t>(nb, 4ull << 20);
            HIP_TRY(hipMemcpyAsync(e->d_meta.p, hs, ml.end, hipMemcpyHostToDevice, cs));
            HIP_TRY(hipMemcpyAsync(e->d_bases.p, hs + o_bases, part, hipMemcpyHostToDevice, cs));
            HIP_TRY(hipMemcpyAsync(e->d_quals.p, hs + o_quals, part, hipMemcpyHostToDevice, cs));
            HIP_TRY(hipStreamSynchronize(cs));
            // ... the arena's first touch, and the kernel's first stores into the pinned result block
            if (e->d_scratch.p && e->d_scratch.cap) HIP_TRY(hipMemsetAsync(e->d_scratch.p, 0, e->d_scratch.cap, e->stream));
            hipLaunchKernelGGL(k_warm_out, dim3(1), dim3(64), 0, e->stream, e->h_out.as<uint32_t>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(e->stream));
        }
        if (dbg) fprintf(stderr, "[racon_hip] reserve: done at %.2f ms\n", since());
        return RCN_OK;
    }
    uint8_t* hs = e->h_stage.as<uint8_t>();
    uint32_t* s_wso = reinterpret_cast<uint32_t*>(hs + o_wso); uint8_t* s_type = hs + o_type; uint8_t* s_flags = hs + o_flags;
    uint64_t* s_so = reinterpret_cast<uint64_t*>(hs + o_so); uint8_t* s_hq = hs + o_hq;
    uint32_t* s_bg = reinterpret_cast<uint32_t*>(hs + o_bg); uint32_t* s_en = reinterpret_cast<uint32_t*>(hs + o_en);
    uint32_t* s_ord = reinterpret_cast<uint32_t*>(hs + o_ord); uint8_t* s_full = hs + o_full;
    std::vector<uint64_t> win_base(nw + 1, 0);         // first byte of device window k in the packed bases
    s_wso[0] = 0; s_so[0] = 0;
    for (uint32_t k = 0; k < nw; ++k) {
        const uint32_t w = e->lpt[k], s0 = v.win_seq_off[w], n = v.win_seq_off[w + 1] - s0;
        s_wso[k + 1] = s_wso[k] + n;
        win_base[k + 1] = win_base[k] + (v.seq_off[s0 + n] - v.seq_off[s0]);
        s_type[k] = v.win_type[w];
    }
    const unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    host_parallel(nw, threads, [&](size_t k) {
        const uint32_t w = e->lpt[k], s0 = v.win_seq_off[w], n = v.win_seq_off[w + 1] - s0, d0 = s_wso[k];
        const uint64_t src0 = v.seq_off[s0];
        for (uint32_t i = 0; i < n; ++i) {
            s_so[d0 + i + 1] = win_base[k] + (v.seq_off[s0 + i + 1] - src0);
            s_hq[d0 + i] = v.qual[s0 + i] ? 1 : 0; s_bg[d0 + i] = v.begin[s0 + i]; s_en[d0 + i] = v.end[s0 + i];
            s_ord[d0 + i] = hp.order[s0 + i]; s_full[d0 + i] = hp.full[s0 + i];
        }
    });
    if (dbg) fprintf(stderr, "[racon_hip] polish: metadata staged at %.2f ms\n", since());
    hipStream_t cs = e->copy_stream;
    // (the output offsets travel in the block like everything else: the first asynchronous copy out of PAGEABLE memory makes
    //  the runtime set up its own staging)
    std::memcpy(hs + o_ooff, e->out_off.data(), 8ull * (nw + 1));
    HIP_TRY(hipMemcpyAsync(e->d_meta.p, hs, ml.end, hipMemcpyHostToDevice, cs));        // (the flags follow piece by piece)

    // ---- pieces ----
    // split plan: the deep launch's windows, then the rest.  Otherwise: the deepest windows that hold 1/24 of the bases
    // (they decide when the batch ends and must start first; the fewer there are, the sooner the first launch is under
    // way), then the rest.
    if (slots_total < 2) { HIP_TRY(hipStreamSynchronize(cs)); return plain(); }
    PassPlan pp; pp.n = rcn_engine::kSubLaunches; pp.split = sp.on; pp.copied = true;
    pp.cut[0] = 0; pp.cut[2] = nw;
    if (sp.on) pp.cut[1] = sp.n_deep;
    else {
        // (a batch for the small-window kernel: a third of the bases first -- its windows are alike and short-lived, what
        //  counts is that the device has work while the host packs the rest: 130 MB take 6 ms)
        Caps cs_;
        const bool small_ = fast && small_caps(e, e->shapes.begin(), e->shapes.end(), cs_);
        const uint64_t q1 = small_ ? nb / 3 : nb / 24;
        uint32_t k = 0;
        while (k < nw && win_base[k] < q1) ++k;
        pp.cut[1] = std::min(std::min(std::max(k, 1u), nw), std::max(1u, slots_total / 2));
        // small windows, more of them than resident slots: the first piece is one whole round of the device (a window lasts
        // ~5 ms whatever the piece: 3494 windows behind a first piece of 1506 were two rounds on half the slots)
        if (small_ && nw > slots_total && !e->cfg.max_slots) pp.cut[1] = slots_total;
    }
    {   // reserve the scratch for the usual alphabet (A, C, G, T and one more symbol) before anything runs; a piece that
        // needs more grows the buffer below, behind a synchronisation.  (The symbol count of a window sizes its aligned
        // rings and is only known once its bases have been read, which happens while its piece is packed.)
        PassPlan est = pp;
        std::vector<int32_t> keep(nw);
        for (uint32_t w = 0; w < nw; ++w) { keep[w] = e->shapes[w].nsym; e->shapes[w].nsym = 5; }
        uint32_t left = slots_total;
        for (int c = 0; c < pp.n; ++c) { if (est.cut[c + 1] == est.cut[c]) continue; plan_piece(e, est, c, sp, fast, left, /*estimate=*/true); left -= std::min(left, est.L[c].slots); }
        for (uint32_t w = 0; w < nw; ++w) e->shapes[w].nsym = keep[w];
        if (est.scratch <= scratch_budget(e) && (rc = e->d_scratch.reserve(est.scratch))) return rc;
        if (dbg) fprintf(stderr, "[racon_hip] polish: scratch arena (%.2f GB) at %.2f ms\n", est.scratch / 1e9, since());
    }
    // the main stream zeroed the counters (begin_run): the sub-launches must not start before that
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    uint32_t slots_left = slots_total;
    for (int c = 0; c < pp.n; ++c) {
        const uint32_t k0 = pp.cut[c], k1 = pp.cut[c + 1];
        if (k1 == k0) continue;
        // pack the piece into pinned staging and collect its symbol statistics in the same pass over the bases -- a few MB
        // at a time, each part's copy queued as soon as it is packed, so that the copy engine works while the host packs
        // the next part (a piece is up to 60 MB at cfg2: ~2.5 ms of packing and ~1.5 ms of PCIe that used to run one after
        // the other before the piece's launch could start)
        constexpr uint64_t kPart = 6ull << 20;
        for (uint32_t ka = k0; ka < k1;) {
            uint32_t kb = ka + 1;
            while (kb < k1 && win_base[kb] - win_base[ka] < kPart) ++kb;
            host_parallel(kb - ka, threads, [&](size_t kk) {
                const uint32_t k = ka + static_cast<uint32_t>(kk), w = e->lpt[k], s0 = v.win_seq_off[w], n = v.win_seq_off[w + 1] - s0;
                uint8_t* db = hs + o_bases + win_base[k]; uint8_t* dq = hs + o_quals + win_base[k];
                uint8_t* const db0 = db;
                uint64_t present[4] = {0, 0, 0, 0};
                for (uint32_t i = 0; i < n; ++i) {
                    const uint64_t len = v.seq_off[s0 + i + 1] - v.seq_off[s0 + i];
                    if (!len) continue;
                    std::memcpy(db, v.seq[s0 + i], len);
                    if (v.qual[s0 + i]) std::memcpy(dq, v.qual[s0 + i], len); else std::memset(dq, '!', len);
                    db += len; dq += len;
                }
                // (one scan over the window's packed bytes: a short-read window is ~140 layers of ~90 bases, and a scan per
                //  layer is below the vector path's 64 bytes half of the time)
                symbols_add(present, db0, static_cast<uint64_t>(db - db0));
                symbols_finish(present, e->shapes[w].nsym, s_flags[k]);
            });
            const uint64_t pa = win_base[ka], pz = win_base[kb];
            if (pz > pa) {
                HIP_TRY(hipMemcpyAsync(e->d_bases.as<uint8_t>() + pa, hs + o_bases + pa, pz - pa, hipMemcpyHostToDevice, cs));
                HIP_TRY(hipMemcpyAsync(e->d_quals.as<uint8_t>() + pa, hs + o_quals + pa, pz - pa, hipMemcpyHostToDevice, cs));
            }
            ka = kb;
        }
        plan_piece(e, pp, c, sp, fast, slots_left);
        slots_left -= std::min(slots_left, pp.L[c].slots);
        if (pp.L[c].slots == 0 || pp.scratch > scratch_budget(e)) {
            // no slot left / does not fit next to each other: the plain path shares the slots (launches already made finish first)
            HIP_TRY(hipDeviceSynchronize());
            return plain();
        }
        // growing the scratch buffer would move it under a running launch: the pieces before this one must be done
        if (pp.scratch > e->d_scratch.cap && c > 0) HIP_TRY(hipDeviceSynchronize());
        if ((rc = e->d_scratch.reserve(pp.scratch))) return rc;
        const Launch& L = pp.L[c];
        {   // the piece's window flags, padded up to a DMA-sized copy with the bytes behind them (metadata the device block
            // already holds: the same bytes again)
            const uint64_t a = (o_flags + k0) & ~uint64_t(15);
            const uint64_t z = std::min<uint64_t>(ml.end, std::max<uint64_t>(o_flags + k1, a + kDmaCopyBytes));
            HIP_TRY(hipMemcpyAsync(e->d_meta.as<uint8_t>() + a, hs + a, z - a, hipMemcpyHostToDevice, cs));
        }
        HIP_TRY(hipEventRecord(e->sub_ev[c][0], cs));
        HIP_TRY(hipStreamWaitEvent(L.stream, e->sub_ev[c][0], 0));
        if (L.stream != e->stream) HIP_TRY(hipStreamWaitEvent(L.stream, e->ev0, 0));
        HIP_TRY(hipEventRecord(e->sub_ev[c][1], L.stream));
        if ((rc = launch_pass(e, L))) return rc;
        HIP_TRY(hipEventRecord(e->sub_ev[c][2], L.stream));
        if (dbg) fprintf(stderr, "[racon_hip] polish: piece %d enqueued at %.2f ms (host clock)\n", c, since());
    }
    HIP_TRY(hipEventRecord(t.b, cs));
    if ((rc = finish_pieces(e, pp, t.a))) return rc;
    if (dbg) fprintf(stderr, "[racon_hip] polish: launches done at %.2f ms (host clock)\n", since());
    HIP_TRY(hipEventSynchronize(t.b));          // (the last command of the copy stream: not necessarily flushed yet)
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
    e->stats.h2d_ms = ms;
    e->stats.bytes_in = 2 * nb + 17ull * ns + 5ull * nw;
    e->uploaded = true;
    rc = collect(e);
    if (dbg) fprintf(stderr, "[racon_hip] polish: results collected at %.2f ms (host clock)\n", since());
    return rc;
}

int refs_view(const rcn_window_refs* w, std::vector<uint64_t>& so, SrcView& v) {
    if (w->n_windows && (!w->win_seq_off || !w->win_type || !w->seq || !w->qual || !w->seq_len || !w->seq_begin || !w->seq_end)) return RCN_E_ARG;
    if (w->n_windows && w->win_seq_off[w->n_windows] != w->n_seqs) return RCN_E_ARG;
    const uint32_t ns = w->n_seqs;
    so.assign(static_cast<size_t>(ns) + 1, 0);
    for (uint32_t i = 0; i < ns; ++i) {
        if (w->seq_len[i] && !w->seq[i]) return RCN_E_ARG;
        so[i + 1] = so[i] + w->seq_len[i];
    }
    static const uint8_t kNoByte = 0; static const uint32_t kNoWord = 0; static const uint8_t* const kNoPtr = nullptr;
    v.nw = w->n_windows; v.ns = ns; v.flags = w->flags; v.seq_off = so.data();
    v.win_seq_off = w->n_windows ? w->win_seq_off : &kNoWord; v.win_type = w->n_windows ? w->win_type : &kNoByte;
    v.seq = ns ? w->seq : &kNoPtr; v.qual = ns ? w->qual : &kNoPtr; v.begin = ns ? w->seq_begin : &kNoWord; v.end = ns ? w->seq_end : &kNoWord;
    return RCN_OK;
}

__global__ void k_warm(unsigned* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 0; }

}  // namespace

extern "C" {

int rcn_engine_polish(rcn_engine* e, const rcn_batch* b) {
    if (!e || !b || check_batch(b)) return RCN_E_ARG;
    const uint32_t ns = b->n_seqs;
    std::vector<const uint8_t*> sp(std::max<uint32_t>(1, ns)), qp(std::max<uint32_t>(1, ns));
    for (uint32_t i = 0; i < ns; ++i) { sp[i] = b->bases + b->seq_off[i]; qp[i] = b->seq_has_qual[i] ? b->quals + b->seq_off[i] : nullptr; }
    SrcView v;
    v.nw = b->n_windows; v.ns = ns; v.win_seq_off = b->win_seq_off; v.win_type = b->win_type; v.seq_off = b->seq_off;
    v.seq = sp.data(); v.qual = qp.data(); v.begin = b->seq_begin; v.end = b->seq_end;
    if (v.nw == 0) { HIP_TRY(hipSetDevice(e->cfg.device)); const int rc = rcn_engine_upload(e, b); return rc ? rc : rcn_engine_run(e); }
    const int rc = polish_view(e, v);
    if (rc) (void)hipDeviceSynchronize();           // (a failed call may have launches in flight: the caller may retry with less)
    return rc;
}

int rcn_engine_polish_refs(rcn_engine* e, const rcn_window_refs* w) {
    if (!e || !w) return RCN_E_ARG;
    std::vector<uint64_t> so; SrcView v;
    const int rc0 = refs_view(w, so, v);
    if (rc0) return rc0;
    if (v.nw == 0) {
        HIP_TRY(hipSetDevice(e->cfg.device));
        HostBatch hb; materialize(v, hb);
        const int rc = rcn_engine_upload(e, &hb.b);
        return rc ? rc : rcn_engine_run(e);
    }
    const int rc = polish_view(e, v);
    if (rc) (void)hipDeviceSynchronize();           // (a failed call may have launches in flight: the caller may retry with less)
    return rc;
}

int rcn_engine_reserve_refs(rcn_engine* e, const rcn_window_refs* w) {
    if (!e || !w) return RCN_E_ARG;
    std::vector<uint64_t> so; SrcView v;
    int rc = refs_view(w, so, v);
    if (rc) return rc;
    rcn_reserve_hint warm{};
    if ((rc = rcn_engine_reserve(e, &warm))) return rc;                 // first use of the code object / copy engines
    if (v.nw == 0) return RCN_OK;
    return polish_view(e, v, /*dry=*/true);
}

int rcn_engine_reserve(rcn_engine* e, const rcn_reserve_hint* h) {
    if (!e || !h) return RCN_E_ARG;
    HIP_TRY(hipSetDevice(e->cfg.device));
    const uint64_t nw = h->n_windows, ns = std::max<uint64_t>(h->n_seqs, nw), nb = h->n_bases;
    int rc;
    if (!e->warmed) {
        // first use of the code object (loaded lazily by the runtime), of every stream's queue and of the copy engines
        (void)HostPool::get();
        for (hipStream_t st : {e->stream, e->copy_stream, e->sub_stream[0], e->deep_stream, e->rest_stream}) {
            if (!st) continue;
            hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, st, e->d_ctr.as<unsigned>());
            HIP_TRY(hipGetLastError());
        }
        if ((rc = e->h_out.reserve(4096))) return rc;
        HIP_TRY(hipMemcpyAsync(e->h_out.p, e->d_ctr.p, 256, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(e->d_ctr.p, e->h_out.p, 256, hipMemcpyHostToDevice, e->copy_stream));
        HIP_TRY(hipDeviceSynchronize());
        e->warmed = true;
    }
    if (nw == 0) return RCN_OK;
    // inputs of a batch (polish_view's layout)
    if ((rc = e->d_meta.reserve(meta_layout(nw, ns).end + kDmaCopyBytes)) || (rc = e->d_bases.reserve(nb + 16)) || (rc = e->d_quals.reserve(nb + 16))) return rc;
    if ((rc = e->h_stage.reserve(meta_layout(nw, ns).end + 2 * nb + 1024))) return rc;
    // results: first_pass_out_cap per window
    const uint64_t L = std::max<uint32_t>(1, h->window_length);
    const uint64_t out_bytes = nw * ((2 * L + 64 + 15) & ~uint64_t(15));
    if ((rc = e->h_out.reserve(5 * nw + 64 + out_bytes + 16))) return rc;
    // scratch arena: one slot per resident window at the capacities the hinted shape gets
    WinShape sh;
    sh.L = static_cast<int32_t>(L);
    const uint64_t deepest = h->max_window_bases ? h->max_window_bases : nb / nw;
    sh.sum_l = static_cast<int32_t>(std::min<uint64_t>(deepest > L ? deepest - L : 0, 1u << 30));
    sh.lmax = static_cast<int32_t>(h->max_layer_length ? h->max_layer_length : L + (3 * L + 9) / 10);
    sh.nsym = 5;
    Caps c = first_pass_caps(&sh, &sh + 1, !e->knobs.wide_only, e->caps_level, e->knobs.hrows_div);
    uint64_t slots = std::min<uint64_t>(nw, e->cfg.max_slots ? e->cfg.max_slots : static_cast<uint64_t>(e->n_cu) * 8u);
    {
        // windows of the small-window kernel's shape: its slots (a few tens of KB each), not poa_window_kernel2's
        Caps cs;
        if (small_caps(e, &sh, &sh + 1, cs)) { c = cs; slots = std::min<uint64_t>(nw, e->cfg.max_slots ? e->cfg.max_slots : static_cast<uint64_t>(e->n_cu) * cs.per_cu); }
    }
    { size_t fr = 0, tot = 0; HIP_TRY(hipMemGetInfo(&fr, &tot)); e->free_mem = fr + e->d_scratch.cap; }
    const uint64_t want = std::min<uint64_t>(slots * c.slot_bytes, scratch_budget(e));
    if ((rc = e->d_scratch.reserve(want))) return rc;
    return RCN_OK;
}

int rcn_engine_result(rcn_engine* e, rcn_result* out) {
    if (!e || !out) return RCN_E_ARG;
    if (!e->ran) return RCN_E_STATE;
    out->n_windows = e->n_windows;
    out->cons_off = e->cons_off.data(); out->cons = e->cons.data();
    out->polished = e->polished.data(); out->chimeric = e->chimeric.data();
    return RCN_OK;
}

int rcn_engine_stats(rcn_engine* e, rcn_run_stats* out) {
    if (!e || !out) return RCN_E_ARG;
    if (e->stats_pending) {
        // the work counters live on the device (every work-group adds its share when it retires): read on demand, on the
        // engine's own non-blocking stream (a plain hipMemcpy runs on the null stream and waits for every blocking stream of
        // the process -- the CU-masked launch streams of the device's other engine among them)
        HIP_TRY(hipSetDevice(e->cfg.device));
        unsigned long long st[41] = {0};
        HIP_TRY(hipMemcpyAsync(st, e->d_ctr.as<uint8_t>() + kStatsOff, sizeof(st), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        e->stats.dp_cells = st[0]; e->stats.dp_pred_cells = st[1]; e->stats.dp_bytes = st[2];
        for (int k = 0; k < 8; ++k) e->stats.phase_clocks[k] = st[3 + k];
        e->stats.n_sink_ties = st[11];
        e->stats.dp_cells_full = st[12]; e->stats.dp_bytes_full = st[13]; e->stats.n_banded = st[14]; e->stats.n_band_redone = st[15];
        for (int k = 0; k < 8; ++k) e->stats.band_redo_why[k] = st[16 + k];
        e->stats.n_code_wave = st[24];
        e->stats.n_small = st[25];
        for (int k = 0; k < 9; ++k) e->stats.small_bail_why[k] = st[26 + k];
        for (int k = 0; k < 6; ++k) e->stats.small_work[k] = st[35 + k];
        e->stats_pending = false;
    }
    *out = e->stats;
    return RCN_OK;
}

// Self-check: a deterministic sample of the last run's windows again, every shortcut off, compared with what the run returned
// (include/racon_hip.h).  The pattern of collect()'s retry tiers: a pass over a list of device windows into the retry buffers.
int rcn_engine_verify(rcn_engine* e, double fraction, rcn_verify_report* out) {
    if (!e || !out || !(fraction > 0.0)) return RCN_E_ARG;
    if (!e->ran || e->n_windows == 0 || e->shapes.size() != e->n_windows || e->lpt.size() != e->n_windows) return RCN_E_STATE;
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t nw = e->n_windows;
    *out = rcn_verify_report{}; out->first_window = 0xffffffffu;
    HIP_TRY(hipSetDevice(e->cfg.device));
    // the run's own counters first: the passes below add to the device counters and to the launch statistics
    { rcn_run_stats tmp; int rc0 = rcn_engine_stats(e, &tmp); if (rc0) return rc0; }
    const rcn_run_stats saved_stats = e->stats;
    const Knobs saved_knobs = e->knobs;
    struct Restore { rcn_engine* e; const rcn_run_stats& st; const Knobs& k; ~Restore() { e->stats = st; e->knobs = k; e->stats_pending = false; } } restore{e, saved_stats, saved_knobs};
    Knobs& K = e->knobs;
    K.no_band = true; K.force_band_fail = false; K.band_scores = false;      // full rows, int16 scores in HBM
    K.force_slow_tb = true;                                                   // traceback over the score matrix, cell by cell
    K.force_tie = 3; K.force_exact = true;                                    // spoa's DFS order at tied sinks and in the consensus
    K.no_small = true; K.force_small = false; K.no_code_wave = true; K.plant_fault = 0;
    // the sample: by a hash of the window index (the same windows whoever asks), windows of fewer than three sequences
    // are copied through by every path (window.cpp:68-71) and prove nothing
    std::vector<uint32_t> sample;
    {
        const uint64_t thr = fraction >= 1.0 ? (1ull << 32) : static_cast<uint64_t>(fraction * 4294967296.0);
        uint32_t fallback = 0xffffffffu;
        for (uint32_t w = 0; w < nw; ++w) {
            if (e->h_win_seq_off[w + 1] - e->h_win_seq_off[w] < 3) continue;
            if (fallback == 0xffffffffu) fallback = w;
            uint32_t h = w * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            if (h < thr) sample.push_back(w);
        }
        if (sample.empty() && fallback != 0xffffffffu) sample.push_back(fallback);
    }
    if (sample.empty()) return RCN_OK;
    std::vector<uint32_t> item_of;
    if (e->lpt_layout) { item_of.resize(nw); for (uint32_t wi = 0; wi < nw; ++wi) item_of[e->lpt[wi]] = wi; }
    int rc;
    std::vector<uint32_t> todo = sample;
    for (int tier = 0; tier < 2 && !todo.empty(); ++tier) {
        const uint32_t nr = static_cast<uint32_t>(todo.size());
        int32_t n2 = 0, l2 = 1, nsym = 2;
        std::vector<uint32_t> ids(nr);
        std::vector<uint64_t> off2(nr + 1, 0);
        std::vector<WinShape> sh(nr);
        for (size_t k = 0; k < nr; ++k) {
            const uint32_t w = todo[k];
            const auto& sw = e->shapes[w];
            sh[k] = sw;
            n2 = std::max<int32_t>(n2, sw.L + sw.sum_l + 8); l2 = std::max(l2, sw.lmax); nsym = std::max(nsym, sw.nsym);
            ids[k] = e->lpt_layout ? item_of[w] : w;
            off2[k + 1] = off2[k] + ((static_cast<uint64_t>(sw.L) + sw.sum_l + 8 + 15) & ~uint64_t(15));
        }
        const Caps c2 = tier == 0 ? first_pass_caps(sh.begin(), sh.end(), true, std::max(e->caps_level, 1), e->knobs.hrows_div)
                                  : make_caps(n2, n2 + 8, std::max(1, nsym - 1), l2, false);
        if ((rc = e->d_out_cons.reserve(off2[nr] + 16)) || (rc = e->d_out_len.reserve(4ull * nr)) || (rc = e->d_out_flags.reserve(nr))) return rc;
        if ((rc = upload_vec(e->d_retry_off, off2.data(), 8ull * (nr + 1), e->stream))) return rc;
        if ((rc = upload_vec(e->d_win_ids, ids.data(), 4ull * nr, e->stream))) return rc;
        if ((rc = run_pass(e, c2, e->d_win_ids.as<uint32_t>(), nr, /*host_out=*/false))) return rc;
        std::vector<uint32_t> len2(nr);
        std::vector<uint8_t> fl2(nr), block(off2[nr] + 16);
        HIP_TRY(hipMemcpyAsync(len2.data(), e->d_out_len.p, 4ull * nr, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(fl2.data(), e->d_out_flags.p, nr, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(block.data(), e->d_out_cons.p, off2[nr], hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        std::vector<uint32_t> again;
        for (size_t k = 0; k < nr; ++k) {
            const uint32_t w = todo[k];
            if (fl2[k] & rcn::kFlagError) { fprintf(stderr, "[racon_hip] internal error on window %u (self-check)\n", w); return RCN_E_STATE; }
            if (fl2[k] & rcn::kFlagOverflow) { if (tier == 1) return RCN_E_CAPACITY; again.push_back(w); continue; }
            out->n_checked += 1;
            const uint64_t a = e->cons_off[w], len = e->cons_off[w + 1] - a;
            const bool same = len == len2[k] && (len == 0 || std::memcmp(e->cons.data() + a, block.data() + off2[k], len) == 0) &&
                              e->polished[w] == ((fl2[k] & rcn::kFlagPolished) ? 1 : 0) && e->chimeric[w] == ((fl2[k] & rcn::kFlagChimeric) ? 1 : 0);
            if (!same) { out->n_differ += 1; out->first_window = std::min(out->first_window, w); }
        }
        if (tier == 0) out->n_int32 = static_cast<uint32_t>(again.size());
        todo.swap(again);
    }
    out->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (saved_knobs.debug) fprintf(stderr, "[racon_hip] self-check: %u of %u windows re-polished on the exact paths in %.1f ms, %u differ\n", out->n_checked, nw, out->ms, out->n_differ);
    return RCN_OK;
}

int rcn_engine_set_trim(rcn_engine* e, int trim) {
    if (!e) return RCN_E_ARG;
    e->cfg.trim = trim ? 1 : 0;
    return RCN_OK;
}

// ---- incremental form (CUDABatchProcessor::addWindow & co.) ----------------
int rcn_engine_add_window(rcn_engine* e, const rcn_window_desc* w) {
    if (!e || !w || w->n_seqs == 0 || !w->seq || !w->seq_len || !w->begin || !w->end) return RCN_E_ARG;
    uint64_t add = 0;
    for (uint32_t i = 0; i < w->n_seqs; ++i) add += w->seq_len[i];
    const uint64_t limit = e->cfg.arena_bytes ? e->cfg.arena_bytes / 64 : (1ull << 30);
    if (!e->b_win_type.empty() && (e->b_bases.size() + add > limit || e->b_win_type.size() >= (1u << 20)))
        return RCN_BATCH_FULL;
    for (uint32_t i = 0; i < w->n_seqs; ++i) {
        const uint32_t len = w->seq_len[i];
        e->b_bases.insert(e->b_bases.end(), w->seq[i], w->seq[i] + len);
        const bool hq = w->qual && w->qual[i];
        if (hq) e->b_quals.insert(e->b_quals.end(), w->qual[i], w->qual[i] + len);
        else e->b_quals.insert(e->b_quals.end(), len, '!');
        e->b_has_qual.push_back(hq ? 1 : 0);
        e->b_begin.push_back(w->begin[i]); e->b_end.push_back(w->end[i]);
        e->b_seq_off.push_back(e->b_bases.size());
    }
    e->b_win_type.push_back(w->type);
    e->b_win_seq_off.push_back(static_cast<uint32_t>(e->b_has_qual.size()));
    return RCN_OK;
}

int rcn_engine_has_windows(rcn_engine* e) { return e && !e->b_win_type.empty(); }

int rcn_engine_generate_consensus(rcn_engine* e) {
    if (!e) return RCN_E_ARG;
    rcn_batch b{};
    b.n_windows = static_cast<uint32_t>(e->b_win_type.size());
    b.n_seqs = static_cast<uint32_t>(e->b_has_qual.size());
    b.win_seq_off = e->b_win_seq_off.data(); b.win_type = e->b_win_type.data();
    b.seq_off = e->b_seq_off.data(); b.seq_has_qual = e->b_has_qual.data();
    b.seq_begin = e->b_begin.data(); b.seq_end = e->b_end.data();
    static const uint8_t kEmpty = 0;
    b.bases = e->b_bases.empty() ? &kEmpty : e->b_bases.data();
    b.quals = e->b_quals.empty() ? &kEmpty : e->b_quals.data();
    if (b.n_windows == 0) {
        e->n_windows = 0; e->uploaded = true;
        return rcn_engine_run(e);
    }
    int rc = rcn_engine_upload(e, &b);
    if (rc) return rc;
    return rcn_engine_run(e);
}

int rcn_engine_reset(rcn_engine* e) {
    if (!e) return RCN_E_ARG;
    e->b_win_seq_off.assign(1, 0); e->b_seq_off.assign(1, 0);
    e->b_win_type.clear(); e->b_has_qual.clear(); e->b_bases.clear(); e->b_quals.clear();
    e->b_begin.clear(); e->b_end.clear();
    return RCN_OK;
}

}  // extern "C"
