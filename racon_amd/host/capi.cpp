// capi.cpp — C ABI of the host layer (include/racon_host.h).
#include "../../include/racon_host.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include "fatal.hpp"
#include "hip_engine.hpp"
#include "nw_path.hpp"
#include "polisher.hpp"
#include "sequence.hpp"

struct rcnh_polisher {
    std::unique_ptr<racon::Polisher> polisher;
    racon::PackedBatch batch;
    std::string fasta;
    racon::Polisher::DevicePlan plan;           // rcnh_polisher_device_plan / _shard_input
    racon::Polisher::ShardInput shard;
};

namespace {
thread_local std::string g_error;
template <class F>
int guarded(F fn) {
    racon::set_fatal_throws(true);
    try { fn(); return 0; }
    catch (const std::exception& e) { g_error = e.what(); return -1; }
}
void to_fasta(const std::vector<std::unique_ptr<racon::Sequence>>& seqs, std::string* out) {
    out->clear();
    for (const auto& s : seqs) { *out += ">"; *out += s->name(); *out += "\n"; *out += s->data(); *out += "\n"; }
}
}  // namespace

extern "C" {

const char* rcnh_last_error(void) { return g_error.c_str(); }

int rcnh_polisher_create(const char* sequences_path, const char* overlaps_path, const char* target_path,
                         const rcnh_params* q, rcnh_polisher** out) {
    if (!sequences_path || !overlaps_path || !target_path || !q || !out) { g_error = "invalid argument"; return -1; }
    return guarded([&] {
        auto p = racon::createPolisher(sequences_path, overlaps_path, target_path,
            static_cast<racon::PolisherType>(q->type), q->window_length, q->quality_threshold, q->error_threshold,
            q->trim != 0, q->match, q->mismatch, q->gap, q->num_threads, q->hip_batches);
        *out = new rcnh_polisher{std::move(p), {}, {}, {}, {}};
    });
}

int rcnh_polisher_initialize(rcnh_polisher* p) {
    if (!p) { g_error = "invalid argument"; return -1; }
    return guarded([&] { p->polisher->initialize(); });
}

int rcnh_polisher_windows(rcnh_polisher* p, rcn_batch* out) {
    if (!p || !out) { g_error = "invalid argument"; return -1; }
    return guarded([&] { p->polisher->pack_windows(&p->batch); *out = p->batch.view(); });
}

int rcnh_polisher_keep_layout(rcnh_polisher* p, int on) {
    if (!p) { g_error = "invalid argument"; return -1; }
    p->polisher->keep_layout(on != 0);
    return 0;
}

int rcnh_polisher_layout(rcnh_polisher* p, rcn_read_set* r, rcn_overlap_set* o, uint8_t* window_type, uint32_t* window_length, double* quality_threshold) {
    if (!p || !r || !o) { g_error = "invalid argument"; return -1; }
    const racon::Polisher::Layout& l = p->polisher->layout();
    if (l.seq_off.size() < 2) { g_error = "no layout recorded (rcnh_polisher_keep_layout before initialize)"; return -1; }
    r->n_seqs = l.seq_off.size() - 1; r->n_targets = l.n_targets; r->seq_off = l.seq_off.data();
    r->bases = l.bases.data(); r->quals = l.quals.data(); r->seq_has_qual = l.seq_has_qual.data();
    o->n_overlaps = l.q_id.size(); o->q_id = l.q_id.data(); o->t_id = l.t_id.data(); o->strand = l.strand.data();
    o->bp_off = l.bp_off.data(); o->bp_t = l.bp_t.data(); o->bp_q = l.bp_q.data();
    if (window_type) *window_type = l.window_type;
    if (window_length) *window_length = p->polisher->window_length();
    if (quality_threshold) *quality_threshold = p->polisher->quality_threshold();
    return 0;
}

int rcnh_polisher_alignments(rcnh_polisher* p, rcn_cigar_set* a) {
    if (!p || !a) { g_error = "invalid argument"; return -1; }
    const racon::Polisher::Layout& l = p->polisher->layout();
    if (l.seq_off.size() < 2) { g_error = "no layout recorded (rcnh_polisher_keep_layout before initialize)"; return -1; }
    a->n_overlaps = l.q_id.size(); a->q_id = l.q_id.data(); a->t_id = l.t_id.data(); a->strand = l.strand.data();
    a->q_start = l.q_start.data(); a->t_begin = l.t_begin.data(); a->t_end = l.t_end.data();
    a->cigar_off = l.cigar_off.data(); a->cigar = l.cigar.data();
    return 0;
}

int rcnh_polisher_pairs(rcnh_polisher* p, rcn_pair_set* a) {
    if (!p || !a) { g_error = "invalid argument"; return -1; }
    const racon::Polisher::Layout& l = p->polisher->layout();
    if (l.seq_off.size() < 2) { g_error = "no layout recorded (rcnh_polisher_keep_layout before initialize)"; return -1; }
    a->n_pairs = l.q_id.size(); a->q_id = l.q_id.data(); a->t_id = l.t_id.data(); a->strand = l.strand.data();
    a->q_begin = l.q_begin.data(); a->q_end = l.q_end.data(); a->t_begin = l.t_begin.data(); a->t_end = l.t_end.data();
    return 0;
}

int rcnh_polisher_assemble(rcnh_polisher* p, const rcn_result* r, int drop, const char** fasta, uint64_t* len) {
    if (!p || !r || !fasta || !len) { g_error = "invalid argument"; return -1; }
    return guarded([&] {
        if (r->n_windows != p->polisher->num_windows()) throw std::runtime_error("result/window count mismatch");
        std::vector<std::string> cons(r->n_windows);
        for (uint32_t w = 0; w < r->n_windows; ++w)
            cons[w].assign(reinterpret_cast<const char*>(r->cons + r->cons_off[w]), r->cons_off[w + 1] - r->cons_off[w]);
        std::vector<std::unique_ptr<racon::Sequence>> dst;
        p->polisher->assemble([&](uint64_t i) -> const std::string& { return cons[i]; },
                              [&](uint64_t i) { return r->polished[i] != 0; }, dst, drop != 0);
        to_fasta(dst, &p->fasta);
        *fasta = p->fasta.c_str(); *len = p->fasta.size();
    });
}

int rcnh_polisher_polish(rcnh_polisher* p, int drop, const char** fasta, uint64_t* len) {
    if (!p || !fasta || !len) { g_error = "invalid argument"; return -1; }
    return guarded([&] {
        std::vector<std::unique_ptr<racon::Sequence>> dst;
        p->polisher->polish(dst, drop != 0);
        to_fasta(dst, &p->fasta);
        *fasta = p->fasta.c_str(); *len = p->fasta.size();
    });
}

double rcnh_polisher_polish_seconds(rcnh_polisher* p) { return p ? p->polisher->polish_seconds() : 0.0; }
int rcnh_polisher_polish_plan(rcnh_polisher* p, uint32_t* chunks, uint32_t* engines_used) {
    if (!p) return -1;
    if (chunks) *chunks = p->polisher->polish_chunks();
    if (engines_used) *engines_used = p->polisher->polish_engines_used();
    return 0;
}
uint64_t rcnh_polisher_num_windows(rcnh_polisher* p) { return p ? p->polisher->num_windows() : 0; }

int rcnh_polisher_device_plan(rcnh_polisher* p, uint32_t n_shards, uint64_t* cut, uint64_t* target_lo, uint64_t* target_hi, uint64_t* n_overlaps) {
    if (!p || n_shards == 0) { g_error = "invalid argument"; return -1; }
    if (p->polisher->layout().seq_off.size() < 2) { g_error = "no layout recorded (rcnh_polisher_keep_layout before initialize)"; return -1; }
    const int rc = guarded([&] { p->plan = p->polisher->plan_device_job(n_shards); });
    if (rc) return rc;
    const uint32_t n = p->plan.n_shards;
    for (uint32_t s = 0; s < n; ++s) {
        if (cut) cut[s] = p->plan.cut[s];
        if (target_lo) target_lo[s] = p->plan.target_lo[s];
        if (target_hi) target_hi[s] = p->plan.target_hi[s];
        if (n_overlaps) n_overlaps[s] = p->plan.bucket_off[s + 1] - p->plan.bucket_off[s];
    }
    if (cut) cut[n] = p->plan.cut[n];
    return static_cast<int>(n);
}

int rcnh_polisher_shard_input(rcnh_polisher* p, uint32_t n_shards, uint32_t shard, rcnh_shard_dims* dims, rcn_read_set* reads,
                              rcn_overlap_set* overlaps, rcn_cigar_set* a, rcn_pair_set* pr) {
    if (!p || n_shards == 0) { g_error = "invalid argument"; return -1; }
    if (p->polisher->layout().seq_off.size() < 2) { g_error = "no layout recorded (rcnh_polisher_keep_layout before initialize)"; return -1; }
    return guarded([&] {
        if (p->plan.n_shards == 0 || p->plan.n_shards != std::min<uint64_t>(n_shards, std::max<uint64_t>(1, p->plan.first_window.empty() ? 1 : p->plan.first_window.back())))
            p->plan = p->polisher->plan_device_job(n_shards);
        if (shard >= p->plan.n_shards) throw std::runtime_error("shard index out of range");
        p->polisher->make_shard_input(p->plan, shard, &p->shard);
        const auto& in = p->shard;
        if (dims) { dims->window_first = in.wa; dims->window_last = in.wb; dims->window_base = in.window_base; dims->n_windows_local = in.n_local; }
        if (reads) *reads = in.reads;
        if (overlaps) *overlaps = in.overlaps;
        if (a) {
            a->n_overlaps = in.overlaps.n_overlaps; a->q_id = in.overlaps.q_id; a->t_id = in.overlaps.t_id; a->strand = in.overlaps.strand;
            a->q_start = in.p_q_start; a->t_begin = in.p_t_begin; a->t_end = in.p_t_end; a->cigar_off = in.p_cigar_off; a->cigar = in.p_cigar;
        }
        if (pr) {
            pr->n_pairs = in.overlaps.n_overlaps; pr->q_id = in.overlaps.q_id; pr->t_id = in.overlaps.t_id; pr->strand = in.overlaps.strand;
            pr->q_begin = in.p_q_begin; pr->q_end = in.p_q_end; pr->t_begin = in.p_t_begin; pr->t_end = in.p_t_end;
        }
    });
}

void rcnh_polisher_destroy(rcnh_polisher* p) { delete p; }

int rcnh_align_cigar(const char* q, uint32_t ql, const char* t, uint32_t tl, char** cigar) {
    if (!cigar || (!q && ql) || (!t && tl)) { g_error = "invalid argument"; return -1; }
    return guarded([&] {
        const std::string c = racon::nwpath::align_cigar(q, ql, t, tl);
        *cigar = static_cast<char*>(malloc(c.size() + 1));
        memcpy(*cigar, c.c_str(), c.size() + 1);
    });
}

uint64_t rcnh_edit_distance(const char* q, uint64_t ql, const char* t, uint64_t tl) { return racon::nwpath::edit_distance(q, ql, t, tl); }

void rcnh_free(void* p) { free(p); }

}  // extern "C"
