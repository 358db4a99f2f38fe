// emul_main.cpp — TEST HARNESS: runs the flat-array graph code of
// racon_amd/csrc/poa_core.hpp (the code the HIP kernel executes on one lane)
// on the CPU with a scalar DP, so it can be checked against the oracle without
// a GPU.  Built by tests/test_core_emulation.py into tests/emul/libemul.so.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/racon_hip.h"
#include "../../racon_amd/csrc/poa_core.hpp"

using namespace rcn;

static int g_ties = 0, g_aligns = 0;
static long long g_hist[8] = {0};   // pred distance: 1, 2, 3-4, 5-8, 9-16, 17-32, 33-64, >64
static long long g_rows = 0, g_row0 = 0;
static long long g_align_maxin[4] = {0};   // alignments whose widest row has <= 6, 7..8, > 8 in-edges; [3] = by window depth >= 40: 7+
static int g_cur_maxin = 0;

// scalar DP over `rank` (any valid topological order); returns the row of the best sink
// and how many sinks tie at the best score.
static void host_dp(Win& g, const int32_t* rank, const Arr<int32_t>& nr, int V, bool sub, const uint8_t* seq, int len,
                    int m, int x, int gp, int& best_row, int& best, int& tied) {
    const int64_t W = g.hstride;
    for (int r = 0; r < V; ++r) g.desc[r] = make_row_desc(g, nr, rank[r], sub);
    for (int j = 0; j <= len; ++j) g.H[j] = j * gp;
    bool have = false; best = 0; best_row = 0; tied = 0;
    for (int r = 0; r < V; ++r) {
        const RowDesc d = g.desc[r];
        int32_t* row = &g.H[(int64_t)(r + 1) * W];
        const int sym = d.meta & 255;
        for (int j = 0; j <= len; ++j) row[j] = kNeg;
        auto acc = [&](int p) {
            const int32_t* hp = &g.H[p * W];
            for (int j = 0; j <= len; ++j) {
                int dg = j > 0 ? hp[j - 1] + (sym == seq[j - 1] ? m : x) : kNeg;
                row[j] = std::max(row[j], std::max(dg, hp[j] + gp));
            }
        };
        auto dist = [&](int p) { if (p == 0) { ++g_row0; return; } int dd = (r + 1) - p; int b = dd <= 1 ? 0 : dd <= 2 ? 1 : dd <= 4 ? 2 : dd <= 8 ? 3 : dd <= 16 ? 4 : dd <= 32 ? 5 : dd <= 64 ? 6 : 7; ++g_hist[b]; };
        const int np = (d.meta >> 9) & 15;
        ++g_rows;
        for (int q = 0; q < np; ++q) { dist(d.p[q]); acc(d.p[q]); }
        int nin = np;
        for (int e = d.erest; e >= 0; e = g.e_nin[e]) {
            int t = g.e_tail[e];
            if (sub && !g.inc[t]) continue;
            acc(nr[t] + 1); ++nin;
        }
        g_cur_maxin = std::max(g_cur_maxin, nin);
        for (int j = 1; j <= len; ++j) row[j] = std::max(row[j], row[j - 1] + gp);
        if (d.meta & 256) {
            if (!have || best < row[len]) { have = true; best = row[len]; best_row = r + 1; tied = 1; }
            else if (best == row[len]) ++tied;
        }
    }
}

// The wave-parallel AddAlignment of the kernel, emulated: every phase is a loop whose
// iterations are independent (run here in REVERSE order to expose any hidden
// order dependence); the scans between phases are what the wave does with shuffles.
static int emul_parallel_add(Win& g, int plen, const uint8_t* seq, const uint8_t* qual, int len) {
    const uint32_t count = len >= 2 ? 1u : 0u;
    for (int k = 0; k < plen; ++k) if (g.path_pos[k] != -1) g.pos_t[g.path_pos[k]] = g.path_node[k];
    std::vector<int> kind(len), idx(len), eflag(len, 0), eidx(len, 0);
    for (int pos = len - 1; pos >= 0; --pos) kind[pos] = addp_classify(g, seq, pos);
    int nn = 0, anchor = -1;
    for (int pos = 0; pos < len; ++pos) {                 // scans: exclusive count of new nodes, inclusive max of anchors
        idx[pos] = nn; if (kind[pos]) ++nn;
        anchor = std::max(anchor, (int)g.pos_a[pos]); g.pos_a[pos] = anchor;
    }
    const int n_old = g.n_nodes;
    if (n_old + nn > g.ncap) { g.overflow = 1; return 0; }
    for (int pos = len - 1; pos >= 0; --pos) if (kind[pos]) {
        addp_create(g, seq, pos, kind[pos], n_old + idx[pos], count);
        g.new_id[idx[pos]] = n_old + idx[pos]; g.new_anchor[idx[pos]] = g.pos_a[pos];
    }
    g.n_nodes = n_old + nn;
    for (int pos = len - 1; pos >= 1; --pos) eflag[pos] = addp_edge_find(g, qual, pos);
    int ne = 0;
    for (int pos = 1; pos < len; ++pos) { eidx[pos] = ne; ne += eflag[pos]; }
    if (g.n_edges + ne > g.ecap) { g.overflow = 1; return nn; }
    for (int pos = len - 1; pos >= 1; --pos) if (eflag[pos]) addp_edge_create(g, qual, pos, g.n_edges + eidx[pos]);
    g.n_edges += ne;
    for (int pos = 0; pos < len; ++pos) g.cov[g.pos_curr[pos]] += count;
    return nn;
}

extern "C" int rcn_emul_consensus(const rcn_batch* b, int m, int x, int gp, int trim,
                                  uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint8_t* polished) {
    uint64_t out = 0;
    for (uint32_t w = 0; w < b->n_windows; ++w) {
        const uint32_t s0 = b->win_seq_off[w], ns = b->win_seq_off[w + 1] - s0;
        auto sp = [&](uint32_t i) { return b->bases + b->seq_off[s0 + i]; };
        auto qp = [&](uint32_t i) { return b->seq_has_qual[s0 + i] ? b->quals + b->seq_off[s0 + i] : nullptr; };
        auto sl = [&](uint32_t i) { return (int)(b->seq_off[s0 + i + 1] - b->seq_off[s0 + i]); };
        const int L = sl(0);
        cons_off[w] = out;
        if (ns < 3) {
            if (out + L > cons_cap) return -1;
            memcpy(cons + out, sp(0), L); out += L; polished[w] = 0; continue;
        }
        int tot = L, lmax = 0;
        for (uint32_t i = 1; i < ns; ++i) { tot += sl(i); lmax = std::max(lmax, sl(i)); }
        const int ncap = tot + 8, ecap = tot + 8, ring = 8, hstride = lmax + 1 + 4;
        Win g{};
        uint64_t bytes = win_bind(g, nullptr, ncap, ecap, ring, lmax, hstride);
        std::vector<uint8_t> mem(bytes + 64);
        win_bind(g, mem.data(), ncap, ecap, ring, lmax, hstride);
        // backbone
        const uint8_t* q0 = qp(0);
        for (int i = 0; i < L; ++i) { int v = add_node(g, sp(0)[i]); g.cov[v] = L >= 2 ? 1 : 0; if (i) add_edge(g, v - 1, v, pair_weight(q0, i)); g.rank_full[i] = i; g.n2r[i] = i; }
        std::vector<uint32_t> rank(ns);
        for (uint32_t i = 0; i < ns; ++i) rank[i] = i;
        std::sort(rank.begin() + 1, rank.end(), [&](uint32_t l, uint32_t r) { return b->seq_begin[s0 + l] < b->seq_begin[s0 + r]; });
        const uint32_t offset = (uint32_t)(0.01 * L);
        for (uint32_t j = 1; j < ns; ++j) {
            const uint32_t i = rank[j];
            const uint32_t bg = b->seq_begin[s0 + i], en = b->seq_end[s0 + i];
            const bool full = bg < offset && en > (uint32_t)L - offset;
            const int32_t* rk = g.rank_full.ptr(); int V = g.n_nodes;
            const Arr<int32_t>* nr = &g.n2r;
            if (!full) {
                graph_subgraph_mask(g, bg, en, g.stack.ptr());
                V = 0;
                for (int r = 0; r < g.n_nodes; ++r) { int v = g.rank_full[r]; if (g.inc[v]) { g.rank_sub[V] = v; g.n2r_x[v] = V; ++V; } }
                rk = g.rank_sub.ptr(); nr = &g.n2r_x;
            }
            int best_row = 0, best = 0, tied = 0;
            g_cur_maxin = 0;
            host_dp(g, rk, *nr, V, !full, sp(i), sl(i), m, x, gp, best_row, best, tied);
            ++g_aligns;
            ++g_align_maxin[g_cur_maxin <= 6 ? 0 : g_cur_maxin <= 8 ? 1 : 2];
            if (ns >= 40 && g_cur_maxin > 6) ++g_align_maxin[3];
            if (tied > 1) {       // spoa picks the first best sink in ITS rank order: run the exact DFS
                ++g_ties;
                int nx = graph_toposort(g, g.rank_x.ptr(), !full, g.stack.ptr());
                for (int r = 0; r < nx; ++r) {
                    int v = g.rank_x[r]; int row = (*nr)[v] + 1;
                    if ((g.desc[row - 1].meta & 256) && g.H[(int64_t)row * g.hstride + sl(i)] == best) { best_row = row; break; }
                }
            }
            int plen = nw_traceback(g, rk, *nr, !full, sp(i), sl(i), best_row, m, x, gp);
            const int n_old = g.n_nodes;
            int nn;
            if (getenv("RCN_EMUL_SERIAL_ADD")) nn = graph_add_alignment(g, plen, sp(i), qp(i), sl(i));
            else nn = emul_parallel_add(g, plen, sp(i), qp(i), sl(i));
            if (g.overflow) { fprintf(stderr, "emul overflow %d\n", g.overflow); return -2; }
            order_merge_serial(g, n_old, nn);
            // invariant of the in-edge records (PredRec): the first kInlinePreds tails of every in-list in list order, the
            // first edge beyond them, kept up to date by both AddAlignment forms
            for (int v = 0; v < g.n_nodes; ++v) {
                const PredRec pr = g.in6[v];
                int k = 0, rest = -1;
                for (int e = g.in_head[v]; e >= 0; e = g.e_nin[e]) {
                    if (k == kInlinePreds) { rest = e; break; }
                    if (pr.t[k] != g.e_tail[e]) { fprintf(stderr, "emul: in-edge record of node %d, slot %d: %d, list says %d\n", v, k, pr.t[k], g.e_tail[e]); return -4; }
                    ++k;
                }
                if (pr.k != k || pr.erest != rest) { fprintf(stderr, "emul: in-edge record of node %d: k %d erest %d, list says %d %d\n", v, pr.k, pr.erest, k, rest); return -4; }
                for (int q = k; q < kInlinePreds; ++q) if (pr.t[q] != -1) return -4;
            }
        }
        {
            int nr_ = graph_toposort(g, g.rank_x.ptr(), false, g.stack.ptr());
            if (nr_ != g.n_nodes) return -3;
        }
        for (int r = 0; r < g.n_nodes; ++r) g.n2r_x[g.rank_x[r]] = r;
        std::vector<int32_t> cn(g.n_nodes);
        int k = graph_consensus(g, g.rank_x.ptr(), g.n2r_x, cn.data());
        int bgn = 0, end = k - 1;
        if (b->win_type[w] == 1 && trim) {
            const uint32_t avg = (ns - 1) / 2;
            for (; bgn < k; ++bgn) if (consensus_coverage(g, cn[bgn]) >= avg) break;
            for (; end >= 0; --end) if (consensus_coverage(g, cn[end]) >= avg) break;
            if (bgn >= end) { bgn = 0; end = k - 1; }
        }
        if (out + (end - bgn + 1) > cons_cap) return -1;
        for (int t = bgn; t <= end; ++t) cons[out++] = g.code[cn[t]];
        polished[w] = 1;
    }
    cons_off[b->n_windows] = out;
    if (getenv("RCN_EMUL_VERBOSE")) { fprintf(stderr, "[emul] alignments %d, sink ties %d rows %lld row0 %lld hist", g_aligns, g_ties, g_rows, g_row0); for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", g_hist[i]); fprintf(stderr, " | alignments by widest row: <=6 in-edges %lld, 7-8 %lld, >8 %lld (7+ in windows of >= 40 sequences: %lld)\n", g_align_maxin[0], g_align_maxin[1], g_align_maxin[2], g_align_maxin[3]); }
    return 0;
}
