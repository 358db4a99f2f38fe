#include "hip_engine.hpp"

#include <dlfcn.h>

#include <algorithm>

#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>

#include "fatal.hpp"
#include "host_util.hpp"

extern "C" char** environ;
#include "window.hpp"

namespace racon {

namespace {

// Entry points of include/racon_hip.h, bound once per process.
struct Abi {
    void* lib = nullptr;
    std::string error;
    decltype(&rcn_engine_create) create = nullptr;
    decltype(&rcn_engine_destroy) destroy = nullptr;
    decltype(&rcn_engine_upload) upload = nullptr;
    decltype(&rcn_engine_run) run = nullptr;
    decltype(&rcn_engine_polish) polish = nullptr;
    decltype(&rcn_engine_polish_refs) polish_refs = nullptr;
    decltype(&rcn_engine_reserve) reserve = nullptr;
    decltype(&rcn_engine_reserve_refs) reserve_refs = nullptr;
    decltype(&rcn_engine_reserve_run) reserve_run = nullptr;
    decltype(&rcn_device_free_memory) free_memory = nullptr;
    decltype(&rcn_engine_result) result = nullptr;
    decltype(&rcn_engine_stats) stats = nullptr;
    decltype(&rcn_engine_verify) verify = nullptr;
    decltype(&rcn_engine_forget) forget = nullptr;
    decltype(&rcn_engine_set_trim) set_trim = nullptr;
    decltype(&rcn_engine_build_windows) build_windows = nullptr;
    decltype(&rcn_engine_build_windows_from_cigars) build_windows_from_cigars = nullptr;
    decltype(&rcn_engine_build_windows_from_pairs) build_windows_from_pairs = nullptr;
    decltype(&rcn_device_count) device_count = nullptr;
    decltype(&rcn_strerror) strerror_ = nullptr;
};

std::string self_directory() {
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&self_directory), &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t s = p.rfind('/');
        if (s != std::string::npos) return p.substr(0, s);
    }
    return ".";
}

const Abi& abi() {
    static Abi a;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> candidates;
        if (const char* env = getenv("RACON_HIP_LIB")) candidates.emplace_back(env);
        const std::string here = self_directory();
        candidates.push_back(here + "/../csrc/libracon_hip.so");
        candidates.push_back(here + "/libracon_hip.so");
        candidates.emplace_back("libracon_hip.so");
        for (const auto& c : candidates) {
            a.lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
            a.error += std::string(a.error.empty() ? "" : "; ") + dlerror();
        }
        if (!a.lib) return;
#define RCN_BIND(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, name)); \
        if (!a.field) { a.error = std::string("missing symbol ") + name; dlclose(a.lib); a.lib = nullptr; return; }
        RCN_BIND(create, "rcn_engine_create") RCN_BIND(destroy, "rcn_engine_destroy") RCN_BIND(upload, "rcn_engine_upload")
        RCN_BIND(run, "rcn_engine_run") RCN_BIND(polish, "rcn_engine_polish") RCN_BIND(polish_refs, "rcn_engine_polish_refs") RCN_BIND(reserve, "rcn_engine_reserve") RCN_BIND(reserve_refs, "rcn_engine_reserve_refs") RCN_BIND(reserve_run, "rcn_engine_reserve_run") RCN_BIND(free_memory, "rcn_device_free_memory") RCN_BIND(result, "rcn_engine_result") RCN_BIND(stats, "rcn_engine_stats") RCN_BIND(verify, "rcn_engine_verify") RCN_BIND(forget, "rcn_engine_forget")
        RCN_BIND(build_windows, "rcn_engine_build_windows") RCN_BIND(build_windows_from_cigars, "rcn_engine_build_windows_from_cigars")
        RCN_BIND(build_windows_from_pairs, "rcn_engine_build_windows_from_pairs")
        RCN_BIND(set_trim, "rcn_engine_set_trim") RCN_BIND(device_count, "rcn_device_count") RCN_BIND(strerror_, "rcn_strerror")
#undef RCN_BIND
    });
    return a;
}

}  // namespace

void PackedBatch::clear() {
    win_seq_off.assign(1, 0); seq_off.assign(1, 0);
    win_type.clear(); seq_has_qual.clear(); seq_begin.clear(); seq_end.clear(); bases.clear(); quals.clear();
}

void PackedBatch::add(const Window& w) {
    for (size_t i = 0; i < w.sequences_.size(); ++i) {
        const char* s = w.sequences_[i].first; const uint32_t n = w.sequences_[i].second;
        const char* q = w.qualities_[i].first;
        bases.insert(bases.end(), s, s + n);
        if (q) quals.insert(quals.end(), q, q + n); else quals.insert(quals.end(), n, static_cast<uint8_t>('!'));
        seq_has_qual.push_back(q ? 1 : 0);
        seq_begin.push_back(w.positions_[i].first); seq_end.push_back(w.positions_[i].second);
        seq_off.push_back(bases.size());
    }
    win_type.push_back(w.type_ == WindowType::kTGS ? 1 : 0);
    win_seq_off.push_back(static_cast<uint32_t>(seq_has_qual.size()));
}

void WindowRefs::clear() {
    win_seq_off.assign(1, 0); win_type.clear(); seq.clear(); qual.clear(); seq_len.clear(); seq_begin.clear(); seq_end.clear(); bases = 0;
}

void WindowRefs::add(const Window& w) {
    for (size_t i = 0; i < w.sequences_.size(); ++i) {
        seq.push_back(reinterpret_cast<const uint8_t*>(w.sequences_[i].first));
        qual.push_back(reinterpret_cast<const uint8_t*>(w.qualities_[i].first));
        seq_len.push_back(w.sequences_[i].second);
        seq_begin.push_back(w.positions_[i].first); seq_end.push_back(w.positions_[i].second);
        bases += w.sequences_[i].second;
    }
    win_type.push_back(w.type_ == WindowType::kTGS ? 1 : 0);
    win_seq_off.push_back(static_cast<uint32_t>(seq.size()));
}

rcn_window_refs WindowRefs::view(uint32_t flags) const {
    rcn_window_refs r{};
    r.n_windows = n_windows(); r.n_seqs = static_cast<uint32_t>(seq.size());
    r.win_seq_off = win_seq_off.data(); r.win_type = win_type.data();
    r.seq = seq.data(); r.qual = qual.data(); r.seq_len = seq_len.data(); r.seq_begin = seq_begin.data(); r.seq_end = seq_end.data();
    r.flags = flags;
    return r;
}

rcn_batch PackedBatch::view() const {
    static const uint8_t kNone = 0;
    rcn_batch b{};
    b.n_windows = n_windows(); b.n_seqs = static_cast<uint32_t>(seq_has_qual.size());
    b.win_seq_off = win_seq_off.data(); b.win_type = win_type.empty() ? &kNone : win_type.data();
    b.seq_off = seq_off.data(); b.seq_has_qual = seq_has_qual.empty() ? &kNone : seq_has_qual.data();
    static const uint32_t kZero = 0;
    b.seq_begin = seq_begin.empty() ? &kZero : seq_begin.data(); b.seq_end = seq_end.empty() ? &kZero : seq_end.data();
    b.bases = bases.empty() ? &kNone : bases.data(); b.quals = quals.empty() ? &kNone : quals.data();
    return b;
}

int32_t HipEngine::DeviceCount() {
    const Abi& a = abi();
    return a.lib ? a.device_count() : 0;
}

uint64_t HipEngine::UsedMemory(int32_t device) {
    const Abi& a = abi();
    uint64_t fr = 0, tot = 0;
    if (!a.lib || a.free_memory(device, &fr, &tot) != RCN_OK) return 0;
    return tot - fr;
}

uint64_t HipEngine::FreeMemory(int32_t device) {
    const Abi& a = abi();
    uint64_t fr = 0, tot = 0;
    if (!a.lib || a.free_memory(device, &fr, &tot) != RCN_OK) return 0;
    return fr;
}

// Engines outlive the Polisher that used them: a handle given back in good order waits here for the next Create() with the same
// device, scores and experiment switches (a process that runs one Polisher after the other -- bench.py's product leg, the tests --
// otherwise frees a multi-GB arena and allocates it again each time, and the THIRD allocation was seen to wait 1.2 s for the driver to
// recycle what the first two had freed: profiles/r06/c_second_polisher_in_one_process.txt).  At most four per key; what is left at
// exit goes with the process.  RACON_HIP_NO_ENGINE_POOL=1: every Polisher creates and destroys its own.
namespace {
struct Pooled { std::string key; rcn_engine* handle; };
std::mutex g_pool_mutex;
std::vector<Pooled> g_pool;
std::string pool_key(int32_t device, int8_t match, int8_t mismatch, int8_t gap) {
    std::string k = std::to_string(device) + "/" + std::to_string(match) + "/" + std::to_string(mismatch) + "/" + std::to_string(gap);
    // (the engine reads its RCN_* switches once, when it is created: another set of switches is another engine)
    std::vector<std::string> sw;
    for (char** ev = environ; ev && *ev; ++ev) if (!strncmp(*ev, "RCN_", 4)) sw.emplace_back(*ev);
    std::sort(sw.begin(), sw.end());
    for (const auto& v : sw) k += "|" + v;
    return k;
}
}  // namespace

std::shared_ptr<HipEngine> HipEngine::Create(int32_t device, int8_t match, int8_t mismatch, int8_t gap, uint64_t arena_bytes) {
    const Abi& a = abi();
    if (!a.lib)
        fatal("[racon::HipEngine::Create] error: unable to load libracon_hip.so (" + a.error +
              "); the consensus stage has no CPU fallback!");
    std::shared_ptr<HipEngine> e(new HipEngine());
    e->pool_key_ = getenv("RACON_HIP_NO_ENGINE_POOL") ? std::string() : pool_key(device, match, mismatch, gap);
    if (!e->pool_key_.empty()) {
        std::vector<rcn_engine*> evict;
        {
            std::lock_guard<std::mutex> lock(g_pool_mutex);
            for (size_t i = 0; i < g_pool.size(); ++i)
                if (g_pool[i].key == e->pool_key_) { e->handle_ = g_pool[i].handle; g_pool.erase(g_pool.begin() + static_cast<long>(i)); a.forget(e->handle_); return e; }
            // No engine of this kind waits: the ones of OTHER kinds on this device (other scores, other switches) only hold HBM that the new
            // engine is about to ask for -- a test process that goes through a dozen configurations had 40 engines with multi-GB arenas pooled,
            // hipMalloc failed, and the runtime's attempt to trim scratch memory crashed inside ROCr (profiles/r06/m_gpu_suite_crash_backtrace.txt).
            const std::string dev = std::to_string(device) + "/";
            for (size_t i = 0; i < g_pool.size();) {
                if (g_pool[i].key.compare(0, dev.size(), dev) == 0) { evict.push_back(g_pool[i].handle); g_pool.erase(g_pool.begin() + static_cast<long>(i)); }
                else ++i;
            }
        }
        for (rcn_engine* h : evict) a.destroy(h);
    }
    rcn_engine_config cfg{};
    cfg.device = device; cfg.match = match; cfg.mismatch = mismatch; cfg.gap = gap; cfg.trim = 1; cfg.arena_bytes = arena_bytes;
    const int rc = a.create(&cfg, &e->handle_);
    if (rc != RCN_OK)
        fatal(std::string("[racon::HipEngine::Create] error: ") + a.strerror_(rc) + "!");
    return e;
}

HipEngine::~HipEngine() {
    if (!handle_) return;
    if (!pool_key_.empty() && last_rc_ == RCN_OK) {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        size_t same = 0;
        for (const auto& p : g_pool) same += p.key == pool_key_ ? 1 : 0;
        if (same < 4) { g_pool.push_back({pool_key_, handle_}); return; }
    }
    abi().destroy(handle_);
}

void HipEngine::consensus(const PackedBatch& batch, bool trim, std::vector<std::string>* consensus,
                          std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric) {
    const Abi& a = abi();
    const rcn_batch b = batch.view();
    int rc = a.set_trim(handle_, trim ? 1 : 0);
    if (rc == RCN_OK) rc = a.polish(handle_, &b);              // upload hidden behind the kernel
    fetch(rc, consensus, polished, chimeric, /*run=*/false);
}

void HipEngine::consensus(const WindowRefs& refs, bool queued, bool trim, std::vector<std::string>* consensus,
                          std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric) {
    const Abi& a = abi();
    const rcn_window_refs r = refs.view(queued ? RCN_REFS_QUEUED : 0u);
    int rc = a.set_trim(handle_, trim ? 1 : 0);
    if (rc == RCN_OK) rc = a.polish_refs(handle_, &r);
    fetch(rc, consensus, polished, chimeric, /*run=*/false);
}

void HipEngine::reserve(uint32_t n_windows, uint32_t n_seqs, uint64_t n_bases, uint32_t window_length, uint32_t max_layer_length,
                        uint64_t max_window_bases) {
    rcn_reserve_hint h{};
    h.n_windows = n_windows; h.n_seqs = n_seqs; h.n_bases = n_bases; h.window_length = window_length; h.max_layer_length = max_layer_length;
    h.max_window_bases = max_window_bases;
    const int rc = abi().reserve(handle_, &h);
    if (rc != RCN_OK) fatal(std::string("[racon::HipEngine::reserve] error: ") + abi().strerror_(rc) + "!");
}

void HipEngine::reserve(const WindowRefs& refs, bool queued) {
    const rcn_window_refs r = refs.view(queued ? RCN_REFS_QUEUED : 0u);
    const int rc = abi().reserve_refs(handle_, &r);
    if (rc != RCN_OK) fatal(std::string("[racon::HipEngine::reserve] error: ") + abi().strerror_(rc) + "!");
}

// ---- windows built on the device: build (what Polisher::initialize does on the host, reference src/polisher.cpp:388-461) and
//      run (Polisher::polish) are separate steps; consensus(...) is one after the other ----
void HipEngine::built(int rc) {
    last_rc_ = rc;
    if (rc == RCN_E_LAYER) fatal("[racon::Window::add_layer] error: layer begin and end positions are invalid!");
    if (rc != RCN_OK) fatal(std::string("[racon::HipEngine::build] error: ") + abi().strerror_(rc) + "!");
}

void HipEngine::build(const rcn_read_set& reads, const rcn_overlap_set& overlaps, uint32_t window_length, double quality_threshold, uint8_t window_type) {
    built(abi().build_windows(handle_, &reads, &overlaps, window_length, quality_threshold, window_type));
}

void HipEngine::build(const rcn_read_set& reads, const rcn_cigar_set& alignments, uint32_t window_length, double quality_threshold, uint8_t window_type) {
    built(abi().build_windows_from_cigars(handle_, &reads, &alignments, window_length, quality_threshold, window_type));
}

void HipEngine::build(const rcn_read_set& reads, const rcn_pair_set& pairs, uint32_t window_length, double quality_threshold, uint8_t window_type) {
    const bool timing = getenv("RACON_HIP_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    built(abi().build_windows_from_pairs(handle_, &reads, &pairs, window_length, quality_threshold, window_type));
    if (timing) fprintf(stderr, "[racon::HipEngine::build] timing: %lu overlaps aligned and cut into windows on the device in %.1f ms\n",
                        static_cast<unsigned long>(pairs.n_pairs), 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
}

void HipEngine::reserve_run() {
    const int rc = abi().reserve_run(handle_);
    last_rc_ = rc;
    if (rc != RCN_OK) fatal(std::string("[racon::HipEngine::reserve_run] error: ") + abi().strerror_(rc) + "!");
}

void HipEngine::run(bool trim, std::vector<std::string>* consensus, std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric) {
    const int rc = abi().set_trim(handle_, trim ? 1 : 0);
    fetch(rc, consensus, polished, chimeric, /*run=*/true);
}

void HipEngine::consensus(const rcn_read_set& reads, const rcn_overlap_set& overlaps, uint32_t window_length, double quality_threshold,
                          uint8_t window_type, bool trim, std::vector<std::string>* consensus,
                          std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric) {
    build(reads, overlaps, window_length, quality_threshold, window_type);
    run(trim, consensus, polished, chimeric);
}

void HipEngine::consensus(const rcn_read_set& reads, const rcn_cigar_set& alignments, uint32_t window_length, double quality_threshold,
                          uint8_t window_type, bool trim, std::vector<std::string>* consensus,
                          std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric) {
    build(reads, alignments, window_length, quality_threshold, window_type);
    run(trim, consensus, polished, chimeric);
}

void HipEngine::consensus(const rcn_read_set& reads, const rcn_pair_set& pairs, uint32_t window_length, double quality_threshold,
                          uint8_t window_type, bool trim, std::vector<std::string>* consensus,
                          std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric) {
    build(reads, pairs, window_length, quality_threshold, window_type);
    run(trim, consensus, polished, chimeric);
}

void HipEngine::fetch(int rc, std::vector<std::string>* consensus, std::vector<uint8_t>* polished, std::vector<uint8_t>* chimeric, bool run) {
    const Abi& a = abi();
    const auto f0 = std::chrono::steady_clock::now();
    if (rc == RCN_OK && run) rc = a.run(handle_);
    const double t_run = seconds_since(f0);
    rcn_result r{};
    if (rc == RCN_OK) rc = a.result(handle_, &r);
    last_rc_ = rc;
    if (rc != RCN_OK) fatal(std::string("[racon::HipEngine::consensus] error: ") + a.strerror_(rc) + "!");
    // (the device counters are a copy + a stream synchronisation away: only for the timing lines, not inside every polish())
    static const bool want_stats = getenv("RACON_HIP_TIMING") != nullptr;
    rcn_run_stats st{};
    if (want_stats && a.stats(handle_, &st) == RCN_OK) last_kernel_ms_ = st.kernel_ms;
    const double t_stats = seconds_since(f0);
    // RACON_HIP_VERIFY=<fraction> (product option, off by default): that share of every batch's windows is polished a second time
    // on the GPU with every shortcut rule of the kernels switched off (rcn_engine_verify) -- a window that comes out differently is
    // a bug in a rule, and fatal
    static const double verify_fraction = [] { const char* v = getenv("RACON_HIP_VERIFY"); const double f = v ? atof(v) : 0.0; return f > 1.0 ? 1.0 : f; }();
    if (verify_fraction > 0.0 && r.n_windows > 0) {
        rcn_verify_report rep{};
        const int vrc = a.verify(handle_, verify_fraction, &rep);
        if (vrc != RCN_OK) fatal(std::string("[racon::HipEngine::consensus] error: the self-check could not run: ") + a.strerror_(vrc) + "!");
        verified_windows_ += rep.n_checked;
        if (rep.n_differ) {
            const uint64_t gw = verify_ids_ ? verify_ids_(rep.first_window) : rep.first_window;
            fatal("[racon::HipEngine::consensus] error: self-check failed: " + std::to_string(rep.n_differ) + " of " + std::to_string(rep.n_checked) +
                  " re-polished windows differ from the exact paths, first: window " + std::to_string(gw) + (verify_ids_ ? "" : " of its batch") + "!");
        }
        static const bool say = getenv("RACON_HIP_TIMING") != nullptr;
        if (say) fprintf(stderr, "[racon_hip] self-check: %u windows re-polished on the exact paths in %.1f ms (%u through the int32 kernel), none differs\n", rep.n_checked, rep.ms, rep.n_int32);
    }
    last_result_ = r;
    consensus->resize(r.n_windows); polished->resize(r.n_windows); chimeric->resize(r.n_windows);
    // (a shard of cfg5 hands back 250 000 strings: on one thread that is 60-100 ms between two shards with the device idle)
    const uint32_t blocks = (r.n_windows + 4095) / 4096;
    parallel_for(blocks, r.n_windows >= 32768 ? 8 : 1, [&](uint64_t blk) {
        for (uint32_t w = static_cast<uint32_t>(blk) * 4096, e = std::min<uint32_t>(r.n_windows, w + 4096); w < e; ++w) {
            if (w >= fetch_first_ && w < fetch_last_)
                (*consensus)[w].assign(reinterpret_cast<const char*>(r.cons + r.cons_off[w]), r.cons_off[w + 1] - r.cons_off[w]);
            else (*consensus)[w].clear();
            (*polished)[w] = r.polished[w]; (*chimeric)[w] = r.chimeric[w];
        }
    });
    if (want_stats)
        fprintf(stderr, "[racon::HipEngine::run] timing: engine run %.1f ms (kernel %.1f), result + stats %.1f ms, %u strings %.1f ms\n", 1e3 * t_run, last_kernel_ms_,
                1e3 * (t_stats - t_run), r.n_windows, 1e3 * (seconds_since(f0) - t_stats));
}

}  // namespace racon
