// engine.hip — host side of the MI355X window-consensus engine and its C ABI
// (include/racon_hip.h).  Plays the role of the reference's CUDABatchProcessor
// (reference src/cuda/cudabatch.cpp:24-280) — pack windows, run the device POA,
// hand back consensus strings + per-window status — but with the CPU path's
// exact semantics (reference src/window.cpp:65-149), unbounded window shapes and
// no CPU fallback: a window that exceeds the first-pass capacities is re-run on
// the GPU with worst-case capacities.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/racon_hip.h"
#include "poa_kernel2.hpp"

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

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap && p) return RCN_OK;
        if (p) { if (hipFree(p) != hipSuccess) return RCN_E_HIP; p = nullptr; cap = 0; }
        size_t want = std::max<size_t>(bytes, 256);
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
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        if (hipHostMalloc(&p, std::max<size_t>(bytes, 256), hipHostMallocDefault) != hipSuccess) { p = nullptr; return RCN_E_NOMEM; }
        cap = std::max<size_t>(bytes, 256); return RCN_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct WinShape { int32_t L, sum_l, lmax, nsym; };

}  // namespace

struct rcn_engine {
    rcn_engine_config cfg{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cu = 256;
    size_t free_mem = 0;

    // resident batch
    uint32_t n_windows = 0, n_seqs = 0;
    uint64_t n_bases = 0;
    DevBuf d_win_seq_off, d_win_type, d_seq_off, d_has_qual, d_begin, d_end, d_bases, d_quals, d_order, d_full;
    DevBuf d_lpt_ids, d_win_ids, d_win_flags, d_scratch, d_out_cons, d_out_len, d_out_flags, d_ctr;
    HostBuf h_raw;
    std::vector<WinShape> shapes;
    int32_t heavy_ns = 0;
    std::vector<uint32_t> h_win_seq_off;
    std::vector<uint32_t> lpt;          // work item -> window, deepest windows first (longest processing time first)
    bool uploaded = false, ran = false;

    // results
    std::vector<uint64_t> cons_off;
    std::vector<uint8_t> cons, polished, chimeric;
    rcn_run_stats stats{};
    rcn_build_stats bstats{};
    DevBuf d_build[24];                 // rcn_engine_build_windows: resident reads / overlaps / work arrays

    // incremental builder (addWindow form)
    std::vector<uint32_t> b_win_seq_off{0};
    std::vector<uint8_t> b_win_type, b_has_qual, b_bases, b_quals;
    std::vector<uint64_t> b_seq_off{0};
    std::vector<uint32_t> b_begin, b_end;
};

namespace {

int upload_vec(DevBuf& d, const void* src, size_t bytes, hipStream_t s) {
    int rc = d.reserve(bytes);
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, s));
    return RCN_OK;
}

struct Caps { int32_t ncap, ecap, ring, lmax, hstride; uint64_t slot_bytes; uint64_t out_stride; bool fast; };

// fast = poa_window_kernel2 (4 waves per window, int16 Z matrix); else poa_window_kernel (1 wave, int32 H)
Caps make_caps(int32_t ncap, int32_t ecap, int32_t ring, int32_t lmax, bool fast) {
    Caps c; c.ncap = ncap; c.ecap = ecap; c.ring = ring; c.lmax = lmax; c.fast = fast;
    // fast: row stride in int16 cells, a multiple of 512: every DP shape (64 or 256 lanes x 2..8 cells) then
    // covers whole rows only, so the row store needs no lane mask
    c.hstride = fast ? ((lmax + 1 + 511) / 512) * 512 : (lmax + 1 + 128 + 3) & ~3;
    rcn::Win tmp;
    c.slot_bytes = rcn::win_bind(tmp, nullptr, ncap, ecap, ring, lmax, c.hstride, fast ? 2 : 4);
    c.slot_bytes = (c.slot_bytes + 255) & ~uint64_t(255);
    c.out_stride = static_cast<uint64_t>(ncap);
    return c;
}

// one kernel pass over `ids` (or all windows when ids == nullptr)
int run_pass(rcn_engine* e, const Caps& c, const uint32_t* ids, uint32_t n_work, uint64_t out_stride, const uint32_t* d_ids = nullptr) {
    if (n_work == 0) return RCN_OK;
    uint64_t budget = e->cfg.arena_bytes ? e->cfg.arena_bytes : static_cast<uint64_t>(e->free_mem * 0.80);
    uint32_t slots = e->cfg.max_slots ? e->cfg.max_slots : static_cast<uint32_t>(e->n_cu) * 8u;   // 8 work-groups per CU (20 KiB LDS each)
    slots = std::min(slots, n_work);
    while (slots > 1 && static_cast<uint64_t>(slots) * c.slot_bytes > budget) slots = (slots + 1) / 2;
    if (static_cast<uint64_t>(slots) * c.slot_bytes > budget) return RCN_E_CAPACITY;
    int rc = e->d_scratch.reserve(static_cast<uint64_t>(slots) * c.slot_bytes);
    if (rc) return rc;
    if (ids) { rc = upload_vec(e->d_win_ids, ids, sizeof(uint32_t) * n_work, e->stream); if (rc) return rc; }
    HIP_TRY(hipMemsetAsync(e->d_ctr.p, 0, 4, e->stream));

    rcn::KParams P{};
    P.win_seq_off = e->d_win_seq_off.as<uint32_t>(); P.win_type = e->d_win_type.as<uint8_t>();
    P.seq_off = e->d_seq_off.as<uint64_t>(); P.seq_has_qual = e->d_has_qual.as<uint8_t>();
    P.seq_begin = e->d_begin.as<uint32_t>(); P.seq_end = e->d_end.as<uint32_t>();
    P.bases = e->d_bases.as<uint8_t>(); P.quals = e->d_quals.as<uint8_t>();
    P.order = e->d_order.as<uint32_t>(); P.seq_full = e->d_full.as<uint8_t>();
    P.win_ids = ids ? e->d_win_ids.as<uint32_t>() : d_ids; P.n_work = n_work;
    P.win_flags = getenv("RCN_NO_PTAB") ? nullptr : e->d_win_flags.as<uint8_t>();
    P.m = e->cfg.match; P.x = e->cfg.mismatch; P.g = e->cfg.gap; P.trim = e->cfg.trim;
    P.heavy_ns = e->heavy_ns; P.force_exact = getenv("RCN_FORCE_EXACT") ? 1 : 0;
    P.force_tie = getenv("RCN_FORCE_TIE") ? atoi(getenv("RCN_FORCE_TIE")) : 0;
    P.force_slow_tb = getenv("RCN_FORCE_SLOW_TB") ? 1 : 0;
    P.scratch = e->d_scratch.as<uint8_t>(); P.slot_bytes = c.slot_bytes;
    P.ncap = c.ncap; P.ecap = c.ecap; P.ring = c.ring; P.lmax = c.lmax; P.hstride = c.hstride;
    P.out_cons = e->d_out_cons.as<uint8_t>(); P.out_stride = out_stride;
    P.out_len = e->d_out_len.as<uint32_t>(); P.out_flags = e->d_out_flags.as<uint8_t>();
    P.next = e->d_ctr.as<unsigned int>();
    P.stats = reinterpret_cast<unsigned long long*>(e->d_ctr.as<uint8_t>() + 16);

    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    if (c.fast) hipLaunchKernelGGL(rcn::poa_window_kernel2, dim3(slots), dim3(rcn::kThreads2), rcn::kLdsBytes + rcn::kCtxBytes, e->stream, P);
    else hipLaunchKernelGGL(rcn::poa_window_kernel, dim3(slots), dim3(64), rcn::kLdsBytes + rcn::kCtxBytes, e->stream, P);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->stats.kernel_ms += ms; e->stats.n_launches += 1;
    return RCN_OK;
}

}  // namespace

// Host-side preparation shared by rcn_engine_upload and rcn_engine_build_windows: layer order (window.cpp:79-86: the
// reference's std::sort, unstable -- libstdc++'s introsort is what decides ties), full-span flags (window.cpp:88,93-94),
// shape statistics for the scratch capacities, deepest-first work order.  `bases` may be null (batch built on the
// device): every window then gets the batch-wide symbol count `nsym_all`.
int prepare_resident(rcn_engine* e, uint32_t nw, uint32_t ns, const uint32_t* win_seq_off, const uint64_t* seq_off,
                            const uint32_t* seq_begin, const uint32_t* seq_end, const uint8_t* bases, int32_t nsym_all, bool acgt_all = false) {
    std::vector<uint8_t> wflags(nw, 0);
    e->h_win_seq_off.assign(win_seq_off, win_seq_off + nw + 1);
    std::vector<uint32_t> order(ns);
    std::vector<uint8_t> full(ns, 0);
    e->shapes.resize(nw);
    std::vector<uint32_t> rank;
    for (uint32_t w = 0; w < nw; ++w) {
        const uint32_t s0 = win_seq_off[w], n = win_seq_off[w + 1] - s0;
        if (n == 0) return RCN_E_ARG;
        rank.resize(n);
        for (uint32_t i = 0; i < n; ++i) rank[i] = i;
        std::sort(rank.begin() + 1, rank.end(), [&](uint32_t lhs, uint32_t rhs) {
            return seq_begin[s0 + lhs] < seq_begin[s0 + rhs]; });
        const uint32_t L = static_cast<uint32_t>(seq_off[s0 + 1] - seq_off[s0]);
        if (L == 0) return RCN_E_ARG;                         // createWindow rejects empty backbones (window.cpp:19-23)
        const uint32_t offset = static_cast<uint32_t>(0.01 * L);
        bool present[256] = {false};
        WinShape sh{static_cast<int32_t>(L), 0, 0, 0};
        for (uint32_t i = 0; i < n; ++i) {
            order[s0 + i] = rank[i];
            const uint32_t si = s0 + i;
            const uint64_t a = seq_off[si], z = seq_off[si + 1];
            if (i > 0) {
                const uint32_t bg = seq_begin[si], en = seq_end[si];
                if (z == a || bg >= en || bg > L || en > L) return RCN_E_ARG;   // add_layer contract (window.cpp:45-58)
                full[si] = (bg < offset && en > L - offset) ? 1 : 0;
                sh.sum_l += static_cast<int32_t>(z - a);
                sh.lmax = std::max<int32_t>(sh.lmax, static_cast<int32_t>(z - a));
            }
            if (bases) for (uint64_t k = a; k < z; ++k) present[bases[k]] = true;
        }
        if (bases) {
            for (bool p : present) sh.nsym += p;
            wflags[w] = (sh.nsym == int(present['A']) + int(present['C']) + int(present['G']) + int(present['T'])) ? 1 : 0;
        } else { sh.nsym = nsym_all; wflags[w] = acgt_all ? 1 : 0; }
        e->shapes[w] = sh;
    }
    // Longest processing time first: a window's cost grows with (layers x bases), and a launch ends with its
    // slowest window; when there are more windows than resident slots the deep ones must not start last.
    e->lpt.resize(nw);
    for (uint32_t w = 0; w < nw; ++w) e->lpt[w] = w;
    std::stable_sort(e->lpt.begin(), e->lpt.end(), [&](uint32_t a, uint32_t c) {
        const uint64_t ca = static_cast<uint64_t>(win_seq_off[a + 1] - win_seq_off[a]) * static_cast<uint64_t>(e->shapes[a].sum_l + e->shapes[a].L);
        const uint64_t cc = static_cast<uint64_t>(win_seq_off[c + 1] - win_seq_off[c]) * static_cast<uint64_t>(e->shapes[c].sum_l + e->shapes[c].L);
        return ca > cc; });
    int rc;
    if ((rc = upload_vec(e->d_lpt_ids, e->lpt.data(), 4ull * nw, e->stream))) return rc;
    if ((rc = upload_vec(e->d_order, order.data(), 4ull * ns, e->stream))) return rc;
    if ((rc = upload_vec(e->d_full, full.data(), ns, e->stream))) return rc;
    if ((rc = upload_vec(e->d_win_flags, wflags.data(), nw, e->stream))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));                 // order / full / flags are stack-scoped
    return RCN_OK;
}

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
        default: return "unknown";
    }
}

int rcn_engine_create(const rcn_engine_config* cfg, rcn_engine** out) {
    if (!cfg || !out) return RCN_E_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RCN_E_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= n) return RCN_E_ARG;
    HIP_TRY(hipSetDevice(cfg->device));
    auto* e = new rcn_engine();
    e->cfg = *cfg;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    e->n_cu = prop.multiProcessorCount;
    size_t fr = 0, tot = 0;
    HIP_TRY(hipMemGetInfo(&fr, &tot));
    e->free_mem = fr;
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&e->ev0));
    HIP_TRY(hipEventCreate(&e->ev1));
    int rc = e->d_ctr.reserve(256);
    if (rc) { delete e; return rc; }
    *out = e;
    return RCN_OK;
}

void rcn_engine_destroy(rcn_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    for (DevBuf* d : {&e->d_win_seq_off, &e->d_win_type, &e->d_seq_off, &e->d_has_qual, &e->d_begin, &e->d_end,
                      &e->d_bases, &e->d_quals, &e->d_order, &e->d_full, &e->d_lpt_ids, &e->d_win_ids, &e->d_win_flags, &e->d_scratch,
                      &e->d_out_cons, &e->d_out_len, &e->d_out_flags, &e->d_ctr})
        d->release();
    for (DevBuf& d : e->d_build) d.release();
    e->h_raw.release();
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int rcn_engine_upload(rcn_engine* e, const rcn_batch* b) {
    if (!e || !b) return RCN_E_ARG;
    if (b->n_windows && (!b->win_seq_off || !b->win_type || !b->seq_off || !b->seq_has_qual || !b->seq_begin ||
                         !b->seq_end || !b->bases || !b->quals))
        return RCN_E_ARG;
    HIP_TRY(hipSetDevice(e->cfg.device));
    hipEvent_t t0, t1;
    HIP_TRY(hipEventCreate(&t0)); HIP_TRY(hipEventCreate(&t1));
    HIP_TRY(hipEventRecord(t0, e->stream));
    const uint32_t nw = b->n_windows, ns = b->n_seqs;
    e->n_windows = nw; e->n_seqs = ns; e->n_bases = ns ? b->seq_off[ns] : 0;
    e->uploaded = false; e->ran = false;
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
    HIP_TRY(hipEventRecord(t1, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
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

int rcn_engine_run(rcn_engine* e) {
    if (!e) return RCN_E_ARG;
    if (!e->uploaded) return RCN_E_STATE;
    HIP_TRY(hipSetDevice(e->cfg.device));
    const uint32_t nw = e->n_windows;
    const double h2d = e->stats.h2d_ms; const uint64_t bin = e->stats.bytes_in;
    e->stats = rcn_run_stats{}; e->stats.h2d_ms = h2d; e->stats.bytes_in = bin;
    e->cons_off.assign(nw + 1, 0); e->polished.assign(nw, 0); e->chimeric.assign(nw, 0); e->cons.clear();
    if (nw == 0) { e->cons.push_back(0); e->ran = true; return RCN_OK; }
    { size_t fr = 0, tot = 0; HIP_TRY(hipMemGetInfo(&fr, &tot)); e->free_mem = fr + e->d_scratch.cap; }

    // first-pass capacities: typical growth; overflowing windows are re-run below
    int32_t ncap = 0, lmax = 1, nsym = 2;
    for (const auto& s : e->shapes) {
        const int64_t worst = static_cast<int64_t>(s.L) + s.sum_l + 8;
        const int64_t est = static_cast<int64_t>(s.L) + s.sum_l / 4 + 256;
        ncap = std::max<int32_t>(ncap, static_cast<int32_t>(std::min(worst, est)));
        lmax = std::max(lmax, s.lmax); nsym = std::max(nsym, s.nsym);
    }
    const int32_t ring = std::max(1, nsym - 1);
    const bool fast = !getenv("RCN_WIDE_ONLY");
    {
        // windows in the top tail of the depth distribution decide when a launch ends (one wave per window is
        // latency bound): they get the 4-wave DP.  Threshold = a high percentile of sequences per window.
        std::vector<uint32_t> depth(nw);
        for (uint32_t w = 0; w < nw; ++w) depth[w] = e->h_win_seq_off[w + 1] - e->h_win_seq_off[w];
        std::vector<uint32_t> sorted = depth;
        const char* pe = getenv("RCN_HEAVY_PCT");
        const double pct = pe ? atof(pe) : 1.0;
        const size_t kth = std::min<size_t>(nw - 1, static_cast<size_t>(pct * nw));
        std::nth_element(sorted.begin(), sorted.begin() + kth, sorted.end());
        e->heavy_ns = pct >= 1.0 ? 0 : static_cast<int32_t>(std::max<uint32_t>(sorted[kth], 3));
        if (pct <= 0.0) e->heavy_ns = 1;
    }
    Caps c1 = make_caps(ncap, 2 * ncap, ring, lmax, fast);
    int rc;
    if ((rc = e->d_out_cons.reserve(static_cast<uint64_t>(nw) * c1.out_stride))) return rc;
    if ((rc = e->d_out_len.reserve(4ull * nw))) return rc;
    if ((rc = e->d_out_flags.reserve(nw))) return rc;
    HIP_TRY(hipMemsetAsync(e->d_ctr.p, 0, 256, e->stream));
    if ((rc = run_pass(e, c1, nullptr, nw, c1.out_stride, e->d_lpt_ids.as<uint32_t>()))) return rc;

    std::vector<uint32_t> out_len(nw);
    std::vector<uint8_t> flags(nw);
    hipEvent_t t0, t1;
    HIP_TRY(hipEventCreate(&t0)); HIP_TRY(hipEventCreate(&t1));
    HIP_TRY(hipEventRecord(t0, e->stream));
    HIP_TRY(hipMemcpyAsync(out_len.data(), e->d_out_len.p, 4ull * nw, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(flags.data(), e->d_out_flags.p, nw, hipMemcpyDeviceToHost, e->stream));
    const size_t raw_bytes = static_cast<uint64_t>(nw) * c1.out_stride;
    if ((rc = e->h_raw.reserve(raw_bytes))) return rc;
    uint8_t* raw = e->h_raw.as<uint8_t>();
    HIP_TRY(hipMemcpyAsync(raw, e->d_out_cons.p, raw_bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipEventRecord(t1, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, t0, t1)); e->stats.d2h_ms += ms;

    // outputs of the first pass are indexed by work item: back to window order
    {
        std::vector<uint32_t> ol(nw); std::vector<uint8_t> fl(nw);
        for (uint32_t wi = 0; wi < nw; ++wi) { ol[e->lpt[wi]] = out_len[wi]; fl[e->lpt[wi]] = flags[wi]; }
        out_len.swap(ol); flags.swap(fl);
    }
    std::vector<uint32_t> item_of(nw);
    for (uint32_t wi = 0; wi < nw; ++wi) item_of[e->lpt[wi]] = wi;
    // retry pass with worst-case capacities for windows that overflowed
    std::vector<uint32_t> retry;
    for (uint32_t w = 0; w < nw; ++w) {
        if (flags[w] & rcn::kFlagError) { fprintf(stderr, "[racon_hip] internal error on window %u\n", w); return RCN_E_STATE; }
        if (flags[w] & rcn::kFlagOverflow) retry.push_back(w);
    }
    std::vector<std::string> retry_cons(retry.size());
    if (!retry.empty()) {
        int32_t n2 = 0, l2 = 1;
        for (uint32_t w : retry) {
            const auto& s = e->shapes[w];
            n2 = std::max<int32_t>(n2, s.L + s.sum_l + 8); l2 = std::max(l2, s.lmax);
        }
        Caps c2 = make_caps(n2, n2 + 8, ring, l2, false);     // int32 kernel, worst-case capacities
        const uint32_t nr = static_cast<uint32_t>(retry.size());
        const uint64_t stride2 = c2.out_stride;
        // first-pass bytes are already on the host in `raw`; the retry pass indexes its outputs by work item
        if ((rc = e->d_out_cons.reserve(static_cast<uint64_t>(nr) * stride2))) return rc;
        if ((rc = run_pass(e, c2, retry.data(), nr, stride2))) return rc;
        std::vector<uint32_t> len2(nr);
        std::vector<uint8_t> fl2(nr);
        HIP_TRY(hipMemcpy(len2.data(), e->d_out_len.p, 4ull * nr, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(fl2.data(), e->d_out_flags.p, nr, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < retry.size(); ++k) {
            const uint32_t w = retry[k];
            if (fl2[k] & (rcn::kFlagOverflow | rcn::kFlagError)) return RCN_E_CAPACITY;
            retry_cons[k].resize(len2[k]);
            if (len2[k]) HIP_TRY(hipMemcpy(&retry_cons[k][0], e->d_out_cons.as<uint8_t>() + static_cast<uint64_t>(k) * stride2,
                                           len2[k], hipMemcpyDeviceToHost));
            out_len[w] = len2[k]; flags[w] = fl2[k];
        }
        e->stats.n_retried = static_cast<uint32_t>(retry.size());
    }
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);

    unsigned long long st[19] = {0};
    HIP_TRY(hipMemcpy(st, e->d_ctr.as<uint8_t>() + 16, sizeof(st), hipMemcpyDeviceToHost));
    e->stats.dp_cells = st[0]; e->stats.dp_pred_cells = st[1]; e->stats.dp_bytes = st[2];
    for (int k = 0; k < 8; ++k) e->stats.phase_clocks[k] = st[3 + k];
    e->stats.n_sink_ties = st[11];
#ifdef RCN_PROF_WIN
    { static unsigned long long wc[4096][8]; HIP_TRY(hipMemcpyFromSymbol(wc, HIP_SYMBOL(rcn::g_wclk), sizeof(wc)));
      std::vector<std::pair<unsigned long long, int>> tot;
      for (int w = 0; w < 4096 && w < (int)nw; ++w) { unsigned long long s = 0; for (int k = 0; k < 7; ++k) s += wc[w][k]; tot.push_back({s, w}); }
      std::sort(tot.rbegin(), tot.rend());
      unsigned long long all = 0; for (auto& p : tot) all += p.first;
      fprintf(stderr, "[racon_hip] per-window clocks: mean %.3g\n", (double)all / std::max<size_t>(1, tot.size()));
      for (int k = 0; k < 3 && k < (int)tot.size(); ++k) { int w = tot[k].second; fprintf(stderr, "  window %d (%u seqs): total %.3g | sub %.3g desc %.3g dp %.3g tb %.3g add %.3g merge %.3g cons %.3g | tiles %llu boxes %llu slow %llu\n", w,
          e->h_win_seq_off[w + 1] - e->h_win_seq_off[w], (double)tot[k].first, (double)wc[w][0], (double)wc[w][1], (double)wc[w][2], (double)wc[w][3], (double)wc[w][4], (double)wc[w][5], (double)wc[w][6], wc[w][7] >> 40, (wc[w][7] >> 20) & 0xfffff, wc[w][7] & 0xfffff); }
      { int w = tot[tot.size() / 2].second; fprintf(stderr, "  median window %d: total %.3g tb %.3g tiles %llu boxes %llu slow %llu\n", w, (double)tot[tot.size() / 2].first, (double)wc[w][3], wc[w][7] >> 40, (wc[w][7] >> 20) & 0xfffff, wc[w][7] & 0xfffff); } }
#endif
#ifdef RCN_PROF_DP
    { unsigned long long pr[8]; HIP_TRY(hipMemcpyFromSymbol(pr, HIP_SYMBOL(rcn::g_prof_out), sizeof(pr)));
      fprintf(stderr, "[racon_hip] dp prof (cumulative): "); for (int k = 0; k < 2; ++k) fprintf(stderr, "wave%d row %llu bar %llu | ", k, pr[2*k], pr[2*k+1]);
      { unsigned long long db[8]; HIP_TRY(hipMemcpyFromSymbol(db, HIP_SYMBOL(rcn::g_dbg), sizeof(db))); fprintf(stderr, "dbg: "); for (int k = 0; k < 8; ++k) fprintf(stderr, "%llu ", db[k]); fprintf(stderr, "\n"); }
      fprintf(stderr, "traceback: stage %llu walk %llu tiles %llu boxes %llu\n", pr[4], pr[5], pr[6], pr[7]); }
#endif
    if (getenv("RCN_DEBUG")) fprintf(stderr, "[racon_hip] traceback: stage clocks %llu walk clocks %llu tiles %llu steps %llu\n", st[12], st[13], st[14], st[15]);

    for (uint32_t w = 0; w < nw; ++w) e->cons_off[w + 1] = e->cons_off[w] + out_len[w];
    e->cons.resize(e->cons_off[nw] + 1);
    size_t rk = 0;
    for (uint32_t w = 0; w < nw; ++w) {
        uint8_t* dst = e->cons.data() + e->cons_off[w];
        if (rk < retry.size() && retry[rk] == w) { std::memcpy(dst, retry_cons[rk].data(), out_len[w]); ++rk; }
        else std::memcpy(dst, raw + static_cast<uint64_t>(item_of[w]) * c1.out_stride, out_len[w]);
        e->polished[w] = (flags[w] & rcn::kFlagPolished) ? 1 : 0;
        e->chimeric[w] = (flags[w] & rcn::kFlagChimeric) ? 1 : 0;
    }
    e->stats.bytes_out = e->cons_off[nw] + 5ull * nw;
    e->ran = true;
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
    *out = e->stats;
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
