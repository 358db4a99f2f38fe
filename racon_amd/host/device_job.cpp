// device_job.cpp -- racon::Polisher with its windows built in HBM (SURVEY.md 8(f) rows 1, 2, 4 as the product path).
//
// The reference cuts its windows at the end of Polisher::initialize (src/polisher.cpp:388-461) and, on its GPU path, aligns its
// overlaps there too (CUDAPolisher::find_overlap_breaking_points, src/cuda/cudapolisher.cpp:74-214); its multi-device code hands
// window ranges to per-device batches (src/cuda/cudapolisher.cpp:228-240).  Here the same in four pieces, each with one job:
//
//   plan_device_job   the window index space cut into shards balanced by overlap count, the TARGET range behind every shard, and
//                     the overlaps of every shard -- bucketed ONCE, in one pass over the overlaps (pure host code: testable through
//                     rcnh_polisher_device_plan without a device);
//   make_shard_input  what rcn_engine_build_windows* takes for one shard: its targets (and only those: the engine's windows are the
//                     shard's targets' windows, not the whole job's), the reads its overlaps point into, its overlaps' slices of the
//                     layout arrays, re-numbered; copies by all host threads (rcnh_polisher_shard_dims);
//   build_shard       mode dispatch (breaking points / CIGAR walk / pairwise alignment on the device) with the host aligner as the
//                     way out when the device aligner has no room;
//   run_shard         the consensus of the resident windows and its results, into the job's arrays.
//
// device_job() strings them together: phase 1 (end of initialize(): build + reserve, windows stay resident), phase 2 (polish() of
// what phase 1 built), phase 0 (everything inside polish(), shard after shard -- a job cut into more window ranges than devices).
// In phase 0 a device's shards are PIPELINED on the host side: while the engine builds and polishes shard j, a helper thread slices
// shard j + 1 out of the layout (the device side too with RACON_HIP_SHARD_PIPELINE=1: measured slower, see device_job()).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <future>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "fatal.hpp"
#include "hip_engine.hpp"
#include "host_util.hpp"
#include "nw_path.hpp"
#include "polisher.hpp"

namespace racon {

// ---------------------------------------------------------------- plan
Polisher::DevicePlan Polisher::plan_device_job(uint32_t n_shards) const {
    DevicePlan p;
    const uint64_t n_targets = layout_.n_targets, n_ovl = layout_.q_id.size();
    p.first_window.assign(n_targets + 1, 0);
    for (uint64_t t = 0; t < n_targets; ++t) {
        const uint64_t len = layout_.seq_off[t + 1] - layout_.seq_off[t];
        p.first_window[t + 1] = p.first_window[t] + (len + window_length_ - 1) / window_length_;
    }
    const uint64_t nw = p.first_window[n_targets];
    n_shards = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint32_t>(1, n_shards), std::max<uint64_t>(1, nw)));
    p.n_shards = n_shards;
    // windows [w_lo, w_hi] every overlap touches; the cost of a window = the overlaps over it (a difference array: O(overlaps + windows))
    std::vector<uint64_t> w_lo(n_ovl), w_hi(n_ovl);
    std::vector<int64_t> diff(nw + 2, 0);
    for (uint64_t k = 0; k < n_ovl; ++k) {
        const uint64_t tb = layout_.t_begin[k], te = std::max<uint64_t>(layout_.t_end[k], tb + 1), fw = p.first_window[layout_.t_id[k]];
        w_lo[k] = std::min<uint64_t>(fw + tb / window_length_, nw ? nw - 1 : 0);
        w_hi[k] = std::min<uint64_t>(fw + (te - 1) / window_length_, nw ? nw - 1 : 0);
        diff[w_lo[k]] += 1; diff[w_hi[k] + 1] -= 1;
    }
    p.cut.assign(n_shards + 1, nw);
    p.cut[0] = 0;
    {
        double total = 0; int64_t run = 0;
        std::vector<double> cost(nw);
        for (uint64_t w = 0; w < nw; ++w) { run += diff[w]; cost[w] = static_cast<double>(run) + 0.05; total += cost[w]; }
        double acc = 0; uint32_t sidx = 1;
        for (uint64_t w = 0; w < nw && sidx < n_shards; ++w) {
            acc += cost[w];
            if (acc >= total * sidx / n_shards) p.cut[sidx++] = w + 1;
        }
    }
    // the targets whose windows a shard's range touches (a target cut by a boundary belongs to both sides)
    p.target_lo.assign(n_shards, 0); p.target_hi.assign(n_shards, 0);
    for (uint32_t s = 0; s < n_shards; ++s) {
        const uint64_t wa = p.cut[s], wb = p.cut[s + 1];
        if (wa >= wb) continue;
        p.target_lo[s] = static_cast<uint64_t>(std::upper_bound(p.first_window.begin(), p.first_window.end(), wa) - p.first_window.begin()) - 1;
        p.target_hi[s] = static_cast<uint64_t>(std::upper_bound(p.first_window.begin(), p.first_window.end(), wb - 1) - p.first_window.begin());
    }
    // the overlaps of every shard, ascending (an overlap across a boundary goes to both sides): count, then fill
    auto shard_of = [&](uint64_t w) { return static_cast<uint32_t>(std::upper_bound(p.cut.begin() + 1, p.cut.end(), w) - (p.cut.begin() + 1)); };
    p.bucket_off.assign(n_shards + 1, 0);
    std::vector<uint32_t> s_lo(n_ovl), s_hi(n_ovl);
    for (uint64_t k = 0; k < n_ovl; ++k) {
        s_lo[k] = std::min(shard_of(w_lo[k]), n_shards - 1); s_hi[k] = std::min(shard_of(w_hi[k]), n_shards - 1);
        for (uint32_t s = s_lo[k]; s <= s_hi[k]; ++s) p.bucket_off[s + 1] += 1;
    }
    for (uint32_t s = 0; s < n_shards; ++s) p.bucket_off[s + 1] += p.bucket_off[s];
    p.bucket.resize(p.bucket_off[n_shards]);
    std::vector<uint64_t> fill(p.bucket_off.begin(), p.bucket_off.end() - 1);
    for (uint64_t k = 0; k < n_ovl; ++k) for (uint32_t s = s_lo[k]; s <= s_hi[k]; ++s) p.bucket[fill[s]++] = k;
    return p;
}

// ---------------------------------------------------------------- one shard's input
void Polisher::make_shard_input(const DevicePlan& plan, uint32_t sidx, ShardInput* out) const {
    ShardInput& in = *out;
    // (the vectors of a recycled ShardInput keep their capacity: a device's shards take turns over two of them, and only the first
    //  two page their buffers in -- fresh pages faulted in while the HIP runtime maps and unmaps device buffers on another thread wait
    //  for the process's memory-map lock: a 30 ms slice was seen to take 2.3 s next to a running shard)
    in.sidx = sidx; in.wa = plan.cut[sidx]; in.wb = plan.cut[sidx + 1];
    in.whole = false;
    const Layout& L = layout_;
    const uint64_t n_seq = L.seq_off.size() - 1;
    const uint64_t* sel = plan.bucket.data() + plan.bucket_off[sidx];
    const uint64_t n_sel = plan.bucket_off[sidx + 1] - plan.bucket_off[sidx];
    static const uint8_t kNoByte = 0; static const uint32_t kNoWord = 0;
    if (plan.n_shards == 1) {
        // the whole job: views straight into the layout
        in.whole = true; in.window_base = 0; in.n_local = plan.first_window[L.n_targets];
        in.reads.n_seqs = n_seq; in.reads.n_targets = L.n_targets; in.reads.seq_off = L.seq_off.data();
        in.reads.bases = L.bases.data(); in.reads.quals = L.quals.data(); in.reads.seq_has_qual = L.seq_has_qual.data();
        in.overlaps.n_overlaps = L.q_id.size(); in.overlaps.q_id = L.q_id.data(); in.overlaps.t_id = L.t_id.data(); in.overlaps.strand = L.strand.data();
        in.overlaps.bp_off = L.bp_off.data(); in.overlaps.bp_t = L.bp_t.data(); in.overlaps.bp_q = L.bp_q.data();
        in.p_q_start = L.q_start.data(); in.p_t_begin = L.t_begin.data(); in.p_t_end = L.t_end.data(); in.p_q_begin = L.q_begin.data(); in.p_q_end = L.q_end.data();
        in.p_cigar_off = L.cigar_off.data(); in.p_cigar = L.cigar.data();
        return;
    }
    const uint64_t t_a = plan.target_lo[sidx], t_b = plan.target_hi[sidx];
    in.window_base = plan.first_window[t_a];
    in.n_local = plan.first_window[t_b] - plan.first_window[t_a];
    // sequences, re-numbered: the shard's targets first (the engine numbers its windows over them), then every other sequence one of its
    // overlaps points into -- a read, or (fragment correction) a target of another shard --, in order of first use.  The reference's
    // multi-device path keeps the reads on the host and packs per batch (src/cuda/cudapolisher.cpp:254-276); here a shard's engine
    // gets its own eighth of them.
    constexpr uint32_t kUnused = 0xffffffffu;
    std::vector<uint32_t>& remap = in.remap; std::vector<uint32_t>& old_of = in.old_of;
    remap.assign(n_seq, kUnused); old_of.clear();
    old_of.reserve((t_b - t_a) + n_sel / 4 + 16);
    const uint32_t copy_threads = std::min<uint32_t>(num_threads_, 8);
    for (uint64_t t = t_a; t < t_b; ++t) { remap[t] = static_cast<uint32_t>(t - t_a); old_of.push_back(static_cast<uint32_t>(t)); }
    in.q_id.resize(n_sel); in.t_id.resize(n_sel); in.strand.resize(n_sel);
    in.q_start.resize(n_sel); in.t_begin.resize(n_sel); in.t_end.resize(n_sel); in.q_begin.resize(n_sel); in.q_end.resize(n_sel);
    in.bp_off.assign(n_sel + 1, 0); in.cigar_off.assign(n_sel + 1, 0);
    for (uint64_t j = 0; j < n_sel; ++j) {
        const uint64_t k = sel[j];
        const uint32_t q = L.q_id[k];
        if (remap[q] == kUnused) { remap[q] = static_cast<uint32_t>(old_of.size()); old_of.push_back(q); }
        in.q_id[j] = remap[q];
        in.t_id[j] = static_cast<uint32_t>(L.t_id[k] - t_a);
        in.bp_off[j + 1] = in.bp_off[j] + (L.bp_off[k + 1] - L.bp_off[k]);
        in.cigar_off[j + 1] = in.cigar_off[j] + (L.cigar_off[k + 1] - L.cigar_off[k]);
    }
    in.bp_t.resize(in.bp_off[n_sel]); in.bp_q.resize(in.bp_off[n_sel]); in.cigar.resize(in.cigar_off[n_sel] + 1);
    parallel_for((n_sel + 4095) / 4096, copy_threads, [&](uint64_t blk) {
        for (uint64_t j = blk * 4096, e = std::min<uint64_t>(n_sel, j + 4096); j < e; ++j) {
            const uint64_t k = sel[j];
            in.strand[j] = L.strand[k];
            in.q_start[j] = L.q_start[k]; in.t_begin[j] = L.t_begin[k]; in.t_end[j] = L.t_end[k]; in.q_begin[j] = L.q_begin[k]; in.q_end[j] = L.q_end[k];
            if (const uint64_t nb = L.bp_off[k + 1] - L.bp_off[k]) {
                std::memcpy(in.bp_t.data() + in.bp_off[j], L.bp_t.data() + L.bp_off[k], 4 * nb);
                std::memcpy(in.bp_q.data() + in.bp_off[j], L.bp_q.data() + L.bp_off[k], 4 * nb);
            }
            if (const uint64_t nc = L.cigar_off[k + 1] - L.cigar_off[k]) std::memcpy(in.cigar.data() + in.cigar_off[j], L.cigar.data() + L.cigar_off[k], nc);
        }
    });
    const uint64_t n_new = old_of.size();
    in.seq_off.assign(n_new + 1, 0); in.has_qual.resize(n_new);
    for (uint64_t i = 0; i < n_new; ++i) { in.seq_off[i + 1] = in.seq_off[i] + (L.seq_off[old_of[i] + 1] - L.seq_off[old_of[i]]); in.has_qual[i] = L.seq_has_qual[old_of[i]]; }
    in.bases.resize(in.seq_off[n_new] + 1); in.quals.resize(in.seq_off[n_new] + 1);
    parallel_for((n_new + 63) / 64, copy_threads, [&](uint64_t blk) {
        for (uint64_t i = blk * 64, e = std::min<uint64_t>(n_new, i + 64); i < e; ++i) {
            const uint64_t a = L.seq_off[old_of[i]], len = L.seq_off[old_of[i] + 1] - a;
            std::memcpy(in.bases.data() + in.seq_off[i], L.bases.data() + a, len);
            std::memcpy(in.quals.data() + in.seq_off[i], L.quals.data() + a, len);
        }
    });
    in.reads.n_seqs = n_new; in.reads.n_targets = t_b - t_a; in.reads.seq_off = in.seq_off.data();
    in.reads.bases = in.bases.data(); in.reads.quals = in.quals.data(); in.reads.seq_has_qual = in.has_qual.data();
    in.overlaps.n_overlaps = n_sel; in.overlaps.q_id = n_sel ? in.q_id.data() : &kNoWord; in.overlaps.t_id = n_sel ? in.t_id.data() : &kNoWord;
    in.overlaps.strand = n_sel ? in.strand.data() : &kNoByte; in.overlaps.bp_off = in.bp_off.data();
    in.overlaps.bp_t = in.bp_t.empty() ? &kNoWord : in.bp_t.data(); in.overlaps.bp_q = in.bp_q.empty() ? &kNoWord : in.bp_q.data();
    in.p_q_start = n_sel ? in.q_start.data() : &kNoWord; in.p_t_begin = n_sel ? in.t_begin.data() : &kNoWord; in.p_t_end = n_sel ? in.t_end.data() : &kNoWord;
    in.p_q_begin = n_sel ? in.q_begin.data() : &kNoWord; in.p_q_end = n_sel ? in.q_end.data() : &kNoWord;
    in.p_cigar_off = in.cigar_off.data(); in.p_cigar = in.cigar.data();
}

// ---------------------------------------------------------------- build (rcn_engine_build_windows*, by mode)
void Polisher::build_shard(HipEngine& engine, const ShardInput& in) {
    static const uint8_t kNoByte = 0; static const uint32_t kNoWord = 0;
    const rcn_read_set& sr = in.reads; const rcn_overlap_set& so = in.overlaps;
    if (device_align_) {
        rcn_pair_set ps{};
        ps.n_pairs = so.n_overlaps; ps.q_id = so.q_id; ps.t_id = so.t_id; ps.strand = so.strand;
        ps.q_begin = in.p_q_begin; ps.q_end = in.p_q_end; ps.t_begin = in.p_t_begin; ps.t_end = in.p_t_end;
        try {
            if (getenv("RACON_HIP_FORCE_ALIGN_FALLBACK")) throw FatalError("forced");      // (tests)
            engine.build(sr, ps, window_length_, quality_threshold_, layout_.window_type);
            return;
        } catch (const FatalError&) {
            // The device aligner holds one op byte per row + column of every overlap and a per-wave scratch sized for the longest
            // read: an input it has no room for (RCN_E_CAPACITY / RCN_E_NOMEM; also a read beyond its 3 Mbp limit) is aligned HERE
            // instead, by the host's edlib-equivalent (reference src/overlap.cpp:205-224) -- same paths, hence the same windows --
            // and goes on through the CIGAR path.
            if (!getenv("RACON_HIP_FORCE_ALIGN_FALLBACK") && engine.last_rc() != RCN_E_CAPACITY && engine.last_rc() != RCN_E_NOMEM) throw;
        }
        static const struct Comp { char t[256]; Comp() { for (int i = 0; i < 256; ++i) t[i] = static_cast<char>(i); t['A'] = 'T'; t['T'] = 'A'; t['C'] = 'G'; t['G'] = 'C'; } } comp;
        const uint64_t n = so.n_overlaps;
        std::vector<std::string> cg(n);
        std::vector<uint32_t> host_q_start(n);
        parallel_for(n, num_threads_, [&](uint64_t k) {
            const uint64_t qa = sr.seq_off[so.q_id[k]], ql = sr.seq_off[so.q_id[k] + 1] - qa, ta = sr.seq_off[so.t_id[k]];
            std::string q(reinterpret_cast<const char*>(sr.bases + qa + in.p_q_begin[k]), in.p_q_end[k] - in.p_q_begin[k]);
            if (so.strand[k]) { std::reverse(q.begin(), q.end()); for (char& ch_ : q) ch_ = comp.t[static_cast<unsigned char>(ch_)]; }
            cg[k] = nwpath::align_cigar(q.data(), static_cast<uint32_t>(q.size()), reinterpret_cast<const char*>(sr.bases + ta + in.p_t_begin[k]), in.p_t_end[k] - in.p_t_begin[k]);
            host_q_start[k] = so.strand[k] ? static_cast<uint32_t>(ql - in.p_q_end[k]) : in.p_q_begin[k];       // reference src/overlap.cpp:241-242
        });
        std::vector<uint8_t> host_cigar; std::vector<uint64_t> host_cigar_off(1, 0);
        for (const auto& s_ : cg) { host_cigar.insert(host_cigar.end(), s_.begin(), s_.end()); host_cigar_off.push_back(host_cigar.size()); }
        rcn_cigar_set a{};
        a.n_overlaps = so.n_overlaps; a.q_id = so.q_id; a.t_id = so.t_id; a.strand = so.strand;
        a.q_start = host_q_start.empty() ? &kNoWord : host_q_start.data(); a.t_begin = in.p_t_begin; a.t_end = in.p_t_end;
        a.cigar_off = host_cigar_off.data(); a.cigar = host_cigar.empty() ? &kNoByte : host_cigar.data();
        engine.build(sr, a, window_length_, quality_threshold_, layout_.window_type);
    } else if (device_cigars_) {
        rcn_cigar_set a{};
        a.n_overlaps = so.n_overlaps; a.q_id = so.q_id; a.t_id = so.t_id; a.strand = so.strand;
        a.q_start = in.p_q_start; a.t_begin = in.p_t_begin; a.t_end = in.p_t_end; a.cigar_off = in.p_cigar_off; a.cigar = in.p_cigar;
        engine.build(sr, a, window_length_, quality_threshold_, layout_.window_type);
    } else {
        engine.build(sr, so, window_length_, quality_threshold_, layout_.window_type);
    }
}

// ---------------------------------------------------------------- run (the consensus of the resident windows) + results
void Polisher::run_shard(bool views, HipEngine& engine, uint64_t window_base, uint64_t n_local, uint64_t wa, uint64_t wb,
                         std::vector<std::string>& cons, std::vector<uint8_t>& pol, std::vector<uint8_t>& chim) {
    // `views`: no string per window -- the caller concatenates the windows straight from the engine's result block (cons_views_), which
    // stays as it is until the engine's next run
    if (views) engine.set_fetch_range(0, 0); else engine.set_fetch_range(wa - window_base, wb - window_base);   // the strings of its own windows only
    engine.set_verify_ids([window_base](uint32_t w) { return window_base + w; });
    struct Reset { HipEngine& e; ~Reset() { e.set_fetch_range(0, ~uint64_t(0)); e.set_verify_ids(nullptr); } } reset{engine};   // (also when run() throws)
    std::vector<std::string> c; std::vector<uint8_t> pl, ch;
    engine.run(trim_, &c, &pl, &ch);
    if (c.size() != n_local) throw std::runtime_error("[racon::Polisher::polish] error: window count mismatch between host and device!");
    if (views) {
        const rcn_result& r = engine.last_result();
        for (uint64_t w = wa; w < wb; ++w) {
            const uint64_t l = w - window_base;
            cons_views_[w] = ConsensusView(reinterpret_cast<const char*>(r.cons + r.cons_off[l]), static_cast<size_t>(r.cons_off[l + 1] - r.cons_off[l]));
            pol[w] = pl[l]; chim[w] = ch[l];
        }
        return;
    }
    for (uint64_t w = wa; w < wb; ++w) { const uint64_t l = w - window_base; cons[w].swap(c[l]); pol[w] = pl[l]; chim[w] = ch[l]; }
}

// ---------------------------------------------------------------- the job
uint32_t Polisher::device_shards() const {
    const int32_t n_devices = n_devices_ > 0 ? n_devices_ : HipEngine::DeviceCount();
    uint32_t n_shards = static_cast<uint32_t>(std::max(1, n_devices));
    if (const char* sh = getenv("RACON_HIP_DEVICE_SHARDS")) n_shards = std::max(1, atoi(sh));
    return std::max<uint32_t>(n_shards, device_min_shards_);
}

void Polisher::device_job(int phase, std::vector<std::string>* cons_out, std::vector<uint8_t>* pol_out, std::vector<uint8_t>* chim_out) {
    const int32_t n_devices = n_devices_ > 0 ? n_devices_ : HipEngine::DeviceCount();
    const uint64_t nw = windows_.size();
    static std::vector<std::string> no_cons; static std::vector<uint8_t> no_flags;
    std::vector<std::string>& cons = cons_out ? *cons_out : no_cons;
    std::vector<uint8_t>& pol = pol_out ? *pol_out : no_flags;
    std::vector<uint8_t>& chim = chim_out ? *chim_out : no_flags;
    const bool timing = getenv("RACON_HIP_TIMING") != nullptr;
    const auto job_begin = std::chrono::steady_clock::now();
    // the plan: phase 2 polishes what phase 1 cut
    const uint32_t want_shards = static_cast<uint32_t>(std::min<uint64_t>(device_shards(), std::max<uint64_t>(1, nw)));
    if (!(phase == 2 && device_plan_.n_shards == want_shards && !device_plan_.cut.empty())) {
        device_plan_ = plan_device_job(want_shards);
        if (timing) fprintf(stderr, "[racon::Polisher::%s] timing: %u shard(s) planned over %lu windows and %lu overlaps (%lu shard entries) in %.1f ms\n", phase == 1 ? "initialize" : "polish",
                            device_plan_.n_shards, static_cast<unsigned long>(nw), static_cast<unsigned long>(layout_.q_id.size()),
                            static_cast<unsigned long>(device_plan_.bucket.size()), 1e3 * seconds_since(job_begin));
    }
    const DevicePlan& plan = device_plan_;
    if (plan.first_window.back() != nw) fatal("[racon::Polisher::polish] error: window count mismatch between the layout and the windows!");
    const uint32_t n_shards = plan.n_shards;
    std::mutex peak_mutex; uint64_t peak_used = 0;
    auto used_hbm = [&](int32_t device) { return HipEngine::UsedMemory(device % std::max(1, HipEngine::DeviceCount())); };
    std::vector<std::string> shard_errors(n_shards);
    const uint32_t lanes = std::min<uint32_t>(n_shards, static_cast<uint32_t>(n_devices));
    // one thread per device; the shards of one device run one after the other on it (phase 0: pipelined over its two engines)
    auto lane = [&](uint32_t l) {
        FatalThrowsScope scope;
        std::vector<uint32_t> mine;
        for (uint32_t s = l; s < n_shards; s += lanes) if (plan.cut[s] < plan.cut[s + 1]) mine.push_back(s);
        if (mine.empty()) return;
        const int32_t device = static_cast<int32_t>(l);
        std::shared_ptr<HipEngine> eng[2] = {engines_[static_cast<size_t>(l)], nullptr};
        // RACON_HIP_SHARD_PIPELINE=1 (experiments): shard j + 1 is also BUILT -- uploaded, aligned, cut -- on the device's second engine while
        // shard j's consensus runs.  Off: measured slower (cfg5 x 0.25 in four shards: polish() 6.7 s with, 6.1 s without;
        // profiles/r06/b_shard_pipeline_ab.txt) -- aligner and consensus kernel are both bound by instruction issue, side by side each
        // takes as much longer as the other runs, and two resident shards double the HBM in use.  The slicing on the host overlaps either way.
        if (phase == 0 && mine.size() > 1 && engines_.size() >= static_cast<size_t>(l) + static_cast<size_t>(n_devices) + 1 && getenv("RACON_HIP_SHARD_PIPELINE"))
            eng[1] = engines_[static_cast<size_t>(l) + static_cast<size_t>(n_devices)];
        try {
            if (phase == 2) {
                for (uint32_t s : mine)
                    run_shard(mine.size() == 1, *eng[0], plan.first_window[plan.target_lo[s]], plan.n_shards == 1 ? nw : plan.first_window[plan.target_hi[s]] - plan.first_window[plan.target_lo[s]],
                              plan.cut[s], plan.cut[s + 1], cons, pol, chim);
                return;
            }
            // prepare (host, helper thread) -> build (device, this thread) -> run (device, helper thread on the other engine)
            // (three inputs in turn: shard j + 1 is sliced while shard j is built and shard j - 1 may still be running)
            std::shared_ptr<ShardInput> pool[3] = {std::make_shared<ShardInput>(), std::make_shared<ShardInput>(), std::make_shared<ShardInput>()};
            auto prepare = [&](uint32_t s, std::shared_ptr<ShardInput> in) { FatalThrowsScope scope1; in->t_begin_s = seconds_since(job_begin); make_shard_input(plan, s, in.get()); in->t_sliced_s = seconds_since(job_begin); return in; };
            std::future<std::shared_ptr<ShardInput>> next = std::async(std::launch::async, prepare, mine[0], pool[0]);
            std::future<void> running;
            for (size_t j = 0; j < mine.size(); ++j) {
                std::shared_ptr<ShardInput> in = next.get();
                if (j + 1 < mine.size()) next = std::async(std::launch::async, prepare, mine[j + 1], pool[(j + 1) % 3]);
                const std::shared_ptr<HipEngine> engine = eng[1] ? eng[j & 1] : eng[0];
                if (!eng[1] && running.valid()) running.get();              // (one engine: its previous shard first)
                const double t0 = seconds_since(job_begin);
                build_shard(*engine, *in);
                const double t_built = seconds_since(job_begin);
                if (phase == 1) {                                          // the windows stay resident for polish()
                    engine->reserve_run();
                    if (timing) fprintf(stderr, "[racon::Polisher::initialize] timing: shard %u (windows %lu..%lu, %lu overlaps) sliced in %.1f ms, built on device %d in %.1f ms, run reserved after %.1f ms, %.2f GB of HBM in use\n",
                                        in->sidx, static_cast<unsigned long>(in->wa), static_cast<unsigned long>(in->wb), static_cast<unsigned long>(in->overlaps.n_overlaps),
                                        1e3 * (in->t_sliced_s - in->t_begin_s), device, 1e3 * (t_built - t0), 1e3 * (seconds_since(job_begin) - t0), used_hbm(device) / 1e9);
                    continue;
                }
                const uint64_t used_built = timing ? used_hbm(device) : 0;
                if (running.valid()) running.get();                         // the previous shard's consensus (on the other engine)
                running = std::async(std::launch::async, [&, in, engine, t0, t_built, used_built, device]() {
                    FatalThrowsScope scope2;
                    const double t_run = seconds_since(job_begin);
                    run_shard(false, *engine, in->window_base, in->n_local, in->wa, in->wb, cons, pol, chim);
                    if (timing) {
                        const uint64_t used = std::max(used_built, used_hbm(device));
                        fprintf(stderr, "[racon::Polisher::polish] timing: shard %u (windows %lu..%lu, %lu overlaps, %lu reads) on device %d: sliced in %.1f ms, built in %.1f ms (from %.1f), consensus + results in %.1f ms (from %.1f, kernel %.1f), %.2f GB of HBM in use\n",
                                in->sidx, static_cast<unsigned long>(in->wa), static_cast<unsigned long>(in->wb), static_cast<unsigned long>(in->overlaps.n_overlaps),
                                static_cast<unsigned long>(in->reads.n_seqs - in->reads.n_targets), device, 1e3 * (in->t_sliced_s - in->t_begin_s),
                                1e3 * (t_built - t0), 1e3 * t0, 1e3 * (seconds_since(job_begin) - t_run), 1e3 * t_run, engine->last_kernel_ms(), used / 1e9);
                        std::lock_guard<std::mutex> lock(peak_mutex); peak_used = std::max(peak_used, used);
                    }
                });
            }
            if (running.valid()) running.get();
        } catch (const std::exception& ex) { shard_errors[mine[0]] = ex.what(); }
    };
    {
        std::vector<std::thread> pool;
        for (uint32_t l = 0; l < lanes; ++l) pool.emplace_back(lane, l);
        for (auto& th : pool) th.join();
    }
    for (const auto& e : shard_errors) if (!e.empty()) fatal(e);
    if (peak_used) fprintf(stderr, "[racon::Polisher::polish] timing: %u shard(s), peak HBM in use %.2f GB\n", n_shards, peak_used / 1e9);
}

}  // namespace racon
