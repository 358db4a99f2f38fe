// poa_core.hpp — flat-array partial-order graph for one window, as it lives in
// HBM scratch on the MI355X, and the single-lane ("serial phase") operations on
// it.  Everything here is plain integer code over raw arrays, compiled for the
// device by hipcc (poa_kernel.hip) and — for CPU-side unit tests of the very
// same code — by g++ (tests/emul/emul_main.cpp).
//
// Semantics follow spoa 4.0.8 as used by racon (reference src/window.cpp:73-123):
//   graph_add_alignment  <- spoa::Graph::AddAlignment / AddSequence / AddEdge
//   graph_toposort       <- spoa::Graph::TopologicalSort   (exact DFS order)
//   graph_subgraph_mask  <- spoa::Graph::Subgraph / ExtractSubgraph (as a mask:
//                           the sub-graph is never materialised; node ids stay
//                           original so UpdateAlignment is the identity)
//   graph_consensus      <- spoa::Graph::GenerateConsensus / TraverseHeaviestBundle /
//                           BranchCompletion / Node::Coverage
//   nw_traceback         <- traceback half of spoa::SisdAlignmentEngine::Linear
// Node "code" is the raw symbol byte: spoa's coder only serves symbol equality.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define RCN_HD __host__ __device__ __forceinline__
#else
#define RCN_HD inline
#endif
// Device pointers into the HBM scratch are typed as address space 1 ("global"): the backend then
// emits global_load/global_store.  A generic pointer that lost its provenance (e.g. after a trip
// through LDS) would be accessed with FLAT instructions, which are slower and tick lgkmcnt as
// well as vmcnt (every LDS wait would then also drain the outstanding HBM stores).
#if defined(__HIP_DEVICE_COMPILE__)
#define RCN_G __attribute__((address_space(1)))
#else
#define RCN_G
#endif

namespace rcn {

constexpr int32_t kNeg = -(1 << 29);

// Row descriptor of the DP matrix (one per graph node in rank order).
constexpr int kInlinePreds = 6;
struct RowDesc {
    int32_t p[kInlinePreds];   // predecessor ROWS in in-edge order (p[0] = 0, the virtual start row, if none)
    int32_t erest;             // edge id from which in-edges beyond the inline ones must be walked, or -1
    int32_t meta;              // bits 0-7 symbol, bit 8 sink, bits 9-11 number of inline predecessors (1..6), bit 12 banded DP: special row (poa_band.hpp)
};

// Array inside the slot's scratch block: one shared base pointer + a 32-bit
// byte offset (keeps the per-wave scalar-register footprint small on the GPU).
template <class T>
struct Arr {
    RCN_G uint8_t* base; uint32_t off;
    RCN_HD RCN_G T& operator[](int64_t i) const { return reinterpret_cast<RCN_G T*>(base + off)[i]; }
    RCN_HD RCN_G T* ptr() const { return reinterpret_cast<RCN_G T*>(base + off); }
};

// The first kInlinePreds in-edges of a node by TAIL NODE ID, in in-edge (creation) order, next to each other: what the row
// descriptors and the Subgraph sweep need of a node's in-list comes with one 32-byte load instead of one dependent load
// pair per edge (e_tail, e_nin).  Maintained wherever an in-list changes (add_node / add_edge / addp_create /
// addp_edge_create and the backbone set-up of the kernels); node ids never change, so it never needs rebuilding.
struct PredRec {
    int32_t t[kInlinePreds];   // tails, -1 = none
    int32_t erest;             // first in-edge beyond the inline ones (the list goes on through e_nin), -1 = none
    int32_t k;                 // number of inline tails in use
};
static_assert(sizeof(PredRec) == 32, "one 32-byte record per node");
RCN_HD PredRec pred_rec_empty() { PredRec p; for (int q = 0; q < kInlinePreds; ++q) p.t[q] = -1; p.erest = -1; p.k = 0; return p; }

struct Win {
    // capacities
    int32_t ncap, ecap, ring;      // ring = max aligned-ring size - 1 per node slot count
    // nodes
    int32_t n_nodes, n_edges;
    Arr<uint8_t> code;             // [ncap]
    Arr<uint8_t> al_cnt;           // [ncap]
    Arr<uint8_t> mark;             // [ncap] toposort marks (bits0-1) | ignored (bit2)
    Arr<uint8_t> inc;              // [ncap] subgraph inclusion mask
    Arr<int32_t> in_head, in_tail;       // [ncap] edge ids, -1 = none
    Arr<PredRec> in6;                    // [ncap] the head of the in-list by tail node, inline
    Arr<int32_t> out_head, out_tail;     // [ncap]
    Arr<uint32_t> cov;             // [ncap] #sequences (len>=2) through node == |labels|
    Arr<int32_t> al_nodes;         // [ncap*ring]
    Arr<int32_t> rank_full;        // [ncap] rank -> node: a VALID topological order of the whole graph
                                   //        with aligned rings contiguous, maintained incrementally
    Arr<int32_t> rank_tmp;         // [ncap] merge output (swapped with rank_full)
    Arr<int32_t> rank_sub;         // [ncap] rank -> node of the current subgraph (rank_full filtered)
    Arr<int32_t> rank_x;           // [ncap] spoa's EXACT DFS order (computed lazily: sink ties, consensus)
    Arr<int32_t> n2r;              // [ncap] inverse of rank_full
    Arr<int32_t> n2r_x;            // [ncap] inverse of rank_sub / rank_x (whichever is in use)
    Arr<int32_t> new_id;           // [lmax+2] nodes created by the current layer, in sequence order
    Arr<int32_t> new_anchor;       // [lmax+2] old rank after whose ring block each of them is inserted (-1 = front)
    // per-sequence-position scratch of the wave-parallel AddAlignment
    Arr<int32_t> pos_t;            // [lmax+2] aligned node (-1 = none)
    Arr<int32_t> pos_curr;         // [lmax+2] node that carries the base (existing or new)
    Arr<int32_t> pos_a;            // [lmax+2] order anchor contributed by / inherited at this position
    Arr<int32_t> pred;             // [ncap] consensus predecessor
    Arr<int64_t> score;            // [ncap] consensus score
    // edges
    Arr<int32_t> e_tail, e_head, e_nin, e_nout;   // [ecap]
    Arr<int64_t> e_w;              // [ecap]
    // alignment path (reversed order while tracing back)
    Arr<int32_t> path_node, path_pos;    // [ncap + lmax + 2]
    // DP
    Arr<RowDesc> desc;             // [ncap]
    Arr<int32_t> stack;            // [ecap + ncap*(ring+1) + 64] DFS stack scratch
    Arr<int32_t> H;                // [(ncap+1) * hstride]
    int64_t  hcap;                 // ints available in H
    int32_t  hstride;
    int32_t  overflow;             // set when a capacity is exceeded
};

// edge e (tail -> head) has just been appended to head's in-list
RCN_HD void pred_rec_append(Win& g, int32_t head, int32_t tail, int32_t e) {
    RCN_G int32_t* r = reinterpret_cast<RCN_G int32_t*>(&g.in6[head]);
    const int32_t k = r[kInlinePreds + 1];
    if (k < kInlinePreds) { r[k] = tail; r[kInlinePreds + 1] = k + 1; }
    else if (r[kInlinePreds] < 0) r[kInlinePreds] = e;
}

RCN_HD int32_t add_node(Win& g, uint8_t c) {
    if (g.n_nodes >= g.ncap) { g.overflow = 1; return g.ncap - 1; }
    int32_t v = g.n_nodes++;
    g.code[v] = c; g.al_cnt[v] = 0;
    g.in_head[v] = g.in_tail[v] = g.out_head[v] = g.out_tail[v] = -1;
    g.in6[v] = pred_rec_empty();
    g.cov[v] = 0;
    return v;
}

RCN_HD void add_edge(Win& g, int32_t tail, int32_t head, int64_t w) {
    for (int32_t e = g.out_head[tail]; e >= 0; e = g.e_nout[e]) {
        if (g.e_head[e] == head) { g.e_w[e] += w; return; }
    }
    if (g.n_edges >= g.ecap) { g.overflow = 1; return; }
    int32_t e = g.n_edges++;
    g.e_tail[e] = tail; g.e_head[e] = head; g.e_w[e] = w; g.e_nin[e] = -1; g.e_nout[e] = -1;
    if (g.out_tail[tail] < 0) g.out_head[tail] = e; else g.e_nout[g.out_tail[tail]] = e;
    g.out_tail[tail] = e;
    if (g.in_tail[head] < 0) g.in_head[head] = e; else g.e_nin[g.in_tail[head]] = e;
    g.in_tail[head] = e;
    pred_rec_append(g, head, tail, e);
}

RCN_HD uint32_t base_weight(RCN_G const uint8_t* qual, int32_t i) {
    // spoa: weights.emplace_back(quality[i] - 33) on `char`; no quality -> 1
    return qual ? static_cast<uint32_t>(static_cast<int32_t>(static_cast<signed char>(qual[i])) - 33) : 1u;
}
RCN_HD int64_t pair_weight(RCN_G const uint8_t* qual, int32_t i) {   // weights[i-1] + weights[i] (uint32 arithmetic)
    return static_cast<int64_t>(static_cast<uint32_t>(base_weight(qual, i - 1) + base_weight(qual, i)));
}

// chain of new nodes for seq[b,e); returns first node or -1.  `count` = bump coverage.
RCN_HD int32_t add_sequence(Win& g, RCN_G const uint8_t* seq, RCN_G const uint8_t* qual, int32_t b, int32_t e, uint32_t count) {
    if (b == e) return -1;
    int32_t first = -1, prev = -1;
    for (int32_t i = b; i < e; ++i) {
        int32_t c = add_node(g, seq[i]);
        g.cov[c] = count;
        if (first < 0) first = c;
        if (prev >= 0) add_edge(g, prev, c, pair_weight(qual, i));
        prev = c;
        if (g.overflow) return first;
    }
    return first;
}

// Rank (in rank_full) of the last member of the aligned-ring block that holds v.
RCN_HD int32_t block_end_rank(const Win& g, int32_t v) {
    int32_t r = g.n2r[v];
    const int32_t na = g.al_cnt[v];
    for (int32_t a = 0; a < na; ++a) { const int32_t q = g.n2r[g.al_nodes[v * g.ring + a]]; r = q > r ? q : r; }
    return r;
}

// spoa::Graph::AddAlignment for a non-empty alignment given as the REVERSED
// traceback path (path_node/path_pos[0..plen) from end to start).  Also records,
// in sequence order, every node it creates and the rank (in the current
// rank_full) after whose ring block the node has to be inserted to keep
// rank_full a valid, ring-contiguous topological order (see order_merge).
// Returns the number of nodes created.
RCN_HD int32_t graph_add_alignment(Win& g, int32_t plen, RCN_G const uint8_t* seq, RCN_G const uint8_t* qual, int32_t len) {
    if (len == 0) return 0;
    const uint32_t count = len >= 2 ? 1u : 0u;     // a 1-base sequence creates no edge, hence no label
    // first / last valid sequence positions
    int32_t vfront = -1, vback = -1;
    for (int32_t k = plen - 1; k >= 0; --k) if (g.path_pos[k] != -1) { vfront = g.path_pos[k]; break; }
    for (int32_t k = 0; k < plen; ++k) if (g.path_pos[k] != -1) { vback = g.path_pos[k]; break; }
    if (vfront < 0) { g.overflow = 2; return 0; }
    int32_t nn = 0;
    int32_t anchor = -1;
    const int32_t pre0 = g.n_nodes;
    int32_t begin = add_sequence(g, seq, qual, 0, vfront, count);
    int32_t prev = begin < 0 ? -1 : g.n_nodes - 1;
    for (int32_t v = pre0; v < g.n_nodes; ++v) { g.new_id[nn] = v; g.new_anchor[nn] = -1; ++nn; }
    const int32_t suf0 = g.n_nodes;
    int32_t last = add_sequence(g, seq, qual, vback + 1, len, count);
    const int32_t suf1 = g.n_nodes;
    if (g.overflow) return nn;
    for (int32_t k = plen - 1; k >= 0; --k) {
        const int32_t pos = g.path_pos[k];
        if (pos == -1) continue;
        const int32_t t = g.path_node[k];
        const uint8_t c = seq[pos];
        int32_t curr = -1;
        bool created = false;
        if (t == -1) {
            curr = add_node(g, c); created = true;
        } else if (g.code[t] == c) {
            curr = t;
        } else {
            const int32_t na = g.al_cnt[t];
            for (int32_t a = 0; a < na; ++a) {
                int32_t u = g.al_nodes[t * g.ring + a];
                if (g.code[u] == c) { curr = u; break; }
            }
            if (curr < 0) {
                if (na >= g.ring) { g.overflow = 3; return nn; }
                anchor = block_end_rank(g, t);          // joins t's block: goes right behind it
                curr = add_node(g, c); created = true;
                if (g.overflow) return nn;
                for (int32_t a = 0; a < na; ++a) {
                    int32_t u = g.al_nodes[t * g.ring + a];
                    g.al_nodes[u * g.ring + g.al_cnt[u]++] = curr;
                    g.al_nodes[curr * g.ring + g.al_cnt[curr]++] = u;
                }
                g.al_nodes[t * g.ring + g.al_cnt[t]++] = curr;
                g.al_nodes[curr * g.ring + g.al_cnt[curr]++] = t;
            }
        }
        if (g.overflow) return nn;
        if (created) { g.new_id[nn] = curr; g.new_anchor[nn] = anchor; ++nn; }
        else anchor = block_end_rank(g, curr);
        g.cov[curr] += count;
        if (begin < 0) begin = curr;
        if (prev >= 0) add_edge(g, prev, curr, pair_weight(qual, pos));
        prev = curr;
        if (g.overflow) return nn;
    }
    if (last >= 0) add_edge(g, prev, last, pair_weight(qual, vback + 1));
    for (int32_t v = suf0; v < suf1; ++v) { g.new_id[nn] = v; g.new_anchor[nn] = anchor; ++nn; }
    return nn;
}

// ---- wave-parallel AddAlignment: per-position phase bodies -------------------
// A global (NW) alignment consumes every sequence position exactly once, so there
// is no unaligned prefix/suffix chain and every position is independent except
// for (a) the node-id / edge-id numbering (prefix sums over "creates a node" /
// "creates an edge" flags, done by the caller) and (b) the order anchors (prefix
// max).  Distinct positions touch distinct nodes, rings and adjacency-list tails.

// phase 2: resolve position `pos`.  Returns 0 = existing node (pos_curr set),
// 1 = new node, unaligned (insertion), 2 = new node that joins the ring of pos_t.
RCN_HD int32_t addp_classify(Win& g, RCN_G const uint8_t* seq, int32_t pos) {
    const int32_t t = g.pos_t[pos];
    const uint8_t c = seq[pos];
    if (t == -1) { g.pos_curr[pos] = -1; g.pos_a[pos] = -1; return 1; }
    if (g.code[t] == c) { g.pos_curr[pos] = t; g.pos_a[pos] = block_end_rank(g, t); return 0; }
    const int32_t na = g.al_cnt[t];
    for (int32_t a = 0; a < na; ++a) {
        const int32_t u = g.al_nodes[t * g.ring + a];
        if (g.code[u] == c) { g.pos_curr[pos] = u; g.pos_a[pos] = block_end_rank(g, u); return 0; }
    }
    g.pos_curr[pos] = -1; g.pos_a[pos] = block_end_rank(g, t);
    return 2;
}

// phase 4: materialise the new node `id` for position pos (kind 1 or 2).
RCN_HD void addp_create(Win& g, RCN_G const uint8_t* seq, int32_t pos, int32_t kind, int32_t id, uint32_t count) {
    g.code[id] = seq[pos]; g.al_cnt[id] = 0;
    g.in_head[id] = g.in_tail[id] = g.out_head[id] = g.out_tail[id] = -1;
    g.in6[id] = pred_rec_empty();
    g.cov[id] = 0;
    (void)count;
    if (kind == 2) {
        const int32_t t = g.pos_t[pos];
        const int32_t na = g.al_cnt[t];
        for (int32_t a = 0; a < na; ++a) {
            const int32_t u = g.al_nodes[t * g.ring + a];
            g.al_nodes[u * g.ring + g.al_cnt[u]] = id; g.al_cnt[u] = g.al_cnt[u] + 1;
            g.al_nodes[id * g.ring + a] = u;
        }
        g.al_nodes[t * g.ring + na] = id; g.al_cnt[t] = static_cast<uint8_t>(na + 1);
        g.al_nodes[id * g.ring + na] = t;
        g.al_cnt[id] = static_cast<uint8_t>(na + 1);
    }
    g.pos_curr[pos] = id;
}

// phase 5: edge pos_curr[pos-1] -> pos_curr[pos]: reinforce it if present (returns 0)
// or report that it must be created (returns 1).
RCN_HD int32_t addp_edge_find(Win& g, RCN_G const uint8_t* qual, int32_t pos) {
    const int32_t tail = g.pos_curr[pos - 1], head = g.pos_curr[pos];
    for (int32_t e = g.out_head[tail]; e >= 0; e = g.e_nout[e]) {
        if (g.e_head[e] == head) { g.e_w[e] += pair_weight(qual, pos); return 0; }
    }
    return 1;
}

// phase 7: create edge `e` for position pos and append it to both adjacency lists.
RCN_HD void addp_edge_create(Win& g, RCN_G const uint8_t* qual, int32_t pos, int32_t e) {
    const int32_t tail = g.pos_curr[pos - 1], head = g.pos_curr[pos];
    g.e_tail[e] = tail; g.e_head[e] = head; g.e_w[e] = pair_weight(qual, pos); g.e_nin[e] = -1; g.e_nout[e] = -1;
    if (g.out_tail[tail] < 0) g.out_head[tail] = e; else g.e_nout[g.out_tail[tail]] = e;
    g.out_tail[tail] = e;
    if (g.in_tail[head] < 0) g.in_head[head] = e; else g.e_nin[g.in_tail[head]] = e;
    g.in_tail[head] = e;
    pred_rec_append(g, head, tail, e);
}

// Serial reference of the order merge (the kernel does it wave-parallel):
// rank_tmp = rank_full with the nn new nodes inserted behind their anchors.
RCN_HD void order_merge_serial(Win& g, int32_t n_old, int32_t nn) {
    int32_t k = 0, o = 0;
    while (k < nn && g.new_anchor[k] < 0) g.rank_tmp[o++] = g.new_id[k++];
    for (int32_t r = 0; r < n_old; ++r) {
        g.rank_tmp[o++] = g.rank_full[r];
        while (k < nn && g.new_anchor[k] == r) g.rank_tmp[o++] = g.new_id[k++];
    }
    Arr<int32_t> t = g.rank_full; g.rank_full = g.rank_tmp; g.rank_tmp = t;
    for (int32_t r = 0; r < o; ++r) g.n2r[g.rank_full[r]] = r;
}

// Exact spoa DFS topological sort.  With `use_mask`, restricted to nodes with
// inc[v] != 0 (== TopologicalSort of the materialised Subgraph).  `stack` needs
// n_edges + n_nodes*(ring+1) + 1 ints.  Returns the number of ranked nodes.
RCN_HD int32_t graph_toposort(Win& g, RCN_G int32_t* rank, bool use_mask, RCN_G int32_t* stack) {
    const int32_t n = g.n_nodes;
    for (int32_t i = 0; i < n; ++i) g.mark[i] = 0;
    int32_t nr = 0;
    for (int32_t s = 0; s < n; ++s) {
        if (use_mask && !g.inc[s]) continue;
        if ((g.mark[s] & 3) != 0) continue;
        int32_t sp = 0;
        stack[sp++] = s;
        while (sp > 0) {
            const int32_t c = stack[sp - 1];
            bool valid = true;
            const uint8_t mc = g.mark[c];
            if ((mc & 3) != 2) {
                for (int32_t e = g.in_head[c]; e >= 0; e = g.e_nin[e]) {
                    const int32_t t = g.e_tail[e];
                    if (use_mask && !g.inc[t]) continue;
                    if ((g.mark[t] & 3) != 2) { stack[sp++] = t; valid = false; }
                }
                const bool ign = (mc & 4) != 0;
                const int32_t na = g.al_cnt[c];
                if (!ign) {
                    for (int32_t a = 0; a < na; ++a) {
                        const int32_t u = g.al_nodes[c * g.ring + a];
                        if (use_mask && !g.inc[u]) continue;
                        if ((g.mark[u] & 3) != 2) { stack[sp++] = u; g.mark[u] |= 4; valid = false; }
                    }
                }
                if (valid) {
                    g.mark[c] = (mc & 4) | 2;
                    if (!ign) {
                        rank[nr++] = c;
                        for (int32_t a = 0; a < na; ++a) {
                            const int32_t u = g.al_nodes[c * g.ring + a];
                            if (use_mask && !g.inc[u]) continue;
                            rank[nr++] = u;
                        }
                    }
                } else {
                    g.mark[c] = (mc & 4) | 1;
                }
            }
            if (valid) --sp;
        }
    }
    return nr;
}

// spoa::Graph::ExtractSubgraph(end_node, begin_node): backward reachability from
// node `end` over in-edges and aligned links, keeping ids >= begin.
RCN_HD void graph_subgraph_mask(Win& g, int32_t begin, int32_t end, RCN_G int32_t* stack) {
    for (int32_t i = 0; i < g.n_nodes; ++i) g.inc[i] = 0;
    int32_t sp = 0;
    stack[sp++] = end;
    while (sp > 0) {
        const int32_t c = stack[--sp];
        if (!g.inc[c] && c >= begin) {
            for (int32_t e = g.in_head[c]; e >= 0; e = g.e_nin[e]) stack[sp++] = g.e_tail[e];
            const int32_t na = g.al_cnt[c];
            for (int32_t a = 0; a < na; ++a) stack[sp++] = g.al_nodes[c * g.ring + a];
            g.inc[c] = 1;
        }
    }
}

// Row descriptor of rank r (node v) for the (sub)graph being aligned.
RCN_HD RowDesc make_row_desc(const Win& g, const Arr<int32_t>& nr, int32_t v, bool use_mask) {
    RowDesc d; d.erest = -1;
    for (int32_t q = 0; q < kInlinePreds; ++q) d.p[q] = -1;
    int32_t k = 0;
    for (int32_t e = g.in_head[v]; e >= 0; e = g.e_nin[e]) {
        const int32_t t = g.e_tail[e];
        if (use_mask && !g.inc[t]) continue;
        if (k == kInlinePreds) { d.erest = e; break; }
        d.p[k++] = nr[t] + 1;
    }
    if (k == 0) { d.p[0] = 0; k = 1; }
    bool sink = true;
    for (int32_t e = g.out_head[v]; e >= 0; e = g.e_nout[e]) {
        if (!use_mask || g.inc[g.e_head[e]]) { sink = false; break; }
    }
    d.meta = static_cast<int32_t>(g.code[v]) | (sink ? 256 : 0) | (k << 9);
    return d;
}

// Traceback of spoa's linear NW (priority: diagonal, vertical, horizontal;
// predecessors in in-edge order).  H rows are indexed rank+1, row 0 virtual.
// Writes the REVERSED path into path_node/path_pos; returns its length.
RCN_HD int32_t nw_traceback(Win& g, RCN_G const int32_t* rank, const Arr<int32_t>& nr, bool use_mask, RCN_G const uint8_t* seq, int32_t len,
                            int32_t best_row, int32_t m, int32_t x, int32_t gp) {
    const int64_t W = g.hstride;
    int32_t i = best_row, j = len, n = 0;
    while (!(i == 0 && j == 0)) {
        const int32_t hij = g.H[i * W + j];
        int32_t pi = 0, pj = 0; bool found = false;
        if (i != 0) {
            const RowDesc d = g.desc[i - 1];
            const int32_t np = (d.meta >> 9) & 7;
            for (int32_t pass = (j != 0 ? 0 : 1); pass < 2 && !found; ++pass) {     // 0: diagonal, 1: vertical
                const int32_t col = pass == 0 ? j - 1 : j;
                const int32_t add = pass == 0 ? (((d.meta & 255) == seq[j - 1]) ? m : x) : gp;
                for (int32_t q = 0; q < np && !found; ++q) {
                    if (hij == g.H[d.p[q] * W + col] + add) { pi = d.p[q]; pj = col; found = true; }
                }
                for (int32_t e = d.erest; e >= 0 && !found; e = g.e_nin[e]) {
                    const int32_t t = g.e_tail[e];
                    if (use_mask && !g.inc[t]) continue;
                    const int32_t p = nr[t] + 1;
                    if (hij == g.H[p * W + col] + add) { pi = p; pj = col; found = true; }
                }
            }
        }
        if (!found) {            // horizontal (the only move left; spoa asserts it)
            pi = i; pj = j - 1;
            if (j == 0) { g.overflow = 4; return n; }
        }
        g.path_node[n] = (i == pi) ? -1 : rank[i - 1];
        g.path_pos[n] = (j == pj) ? -1 : j - 1;
        ++n;
        i = pi; j = pj;
    }
    return n;
}

RCN_HD void consensus_relax(Win& g, int32_t it, bool skip) {
    int64_t s = g.score[it]; int32_t p = g.pred[it];
    for (int32_t e = g.in_head[it]; e >= 0; e = g.e_nin[e]) {
        const int32_t t = g.e_tail[e];
        if (skip && g.score[t] == -1) continue;
        const int64_t w = g.e_w[e];
        if (s < w || (s == w && g.score[p] <= g.score[t])) { s = w; p = t; }
    }
    if (p >= 0) s += g.score[p];
    g.score[it] = s; g.pred[it] = p;
}

// Heaviest bundle + branch completion over rank_full; writes the consensus node
// ids (in order) to out_nodes; returns the length.  `rank` must be spoa's exact
// order (graph_toposort) and `nr` its inverse.
RCN_HD int32_t graph_consensus(Win& g, RCN_G const int32_t* rank, const Arr<int32_t>& nr, RCN_G int32_t* out_nodes) {
    const int32_t n = g.n_nodes;
    for (int32_t i = 0; i < n; ++i) { g.pred[i] = -1; g.score[i] = -1; }
    int32_t mx = -1;
    for (int32_t r = 0; r < n; ++r) {
        const int32_t it = rank[r];
        consensus_relax(g, it, false);
        if (mx < 0 || g.score[mx] < g.score[it]) mx = it;
    }
    while (g.out_head[mx] >= 0) {
        const int32_t start = mx;
        for (int32_t e = g.out_head[start]; e >= 0; e = g.e_nout[e]) {
            for (int32_t f = g.in_head[g.e_head[e]]; f >= 0; f = g.e_nin[f]) {
                if (g.e_tail[f] != start) g.score[g.e_tail[f]] = -1;
            }
        }
        int32_t m2 = -1;
        for (int32_t r = nr[start] + 1; r < n; ++r) {
            const int32_t it = rank[r];
            g.score[it] = -1; g.pred[it] = -1;
            consensus_relax(g, it, true);
            if (m2 < 0 || g.score[m2] < g.score[it]) m2 = it;
        }
        mx = m2;
    }
    int32_t k = 0;
    while (g.pred[mx] >= 0) { out_nodes[k++] = mx; mx = g.pred[mx]; }
    out_nodes[k++] = mx;
    for (int32_t a = 0, b = k - 1; a < b; ++a, --b) { int32_t t = out_nodes[a]; out_nodes[a] = out_nodes[b]; out_nodes[b] = t; }
    return k;
}

RCN_HD uint32_t consensus_coverage(const Win& g, int32_t v) {
    uint32_t c = g.cov[v];
    const int32_t na = g.al_cnt[v];
    for (int32_t a = 0; a < na; ++a) c += g.cov[g.al_nodes[v * g.ring + a]];
    return c;
}

// Carve a Win out of one slot's scratch block.  Returns bytes used.
// `hcell` = bytes per score cell: 4 (int32 H, poa_window_kernel) or 2 (int16 Z, poa_window_kernel2).
// `hrows`: rows of the DP matrix H (the last and by far the largest array of a slot) when that is fewer than ncap + 1 -- the
// first pass sizes the graph arrays generously and the matrix by what a window of its depth is expected to reach; an
// alignment with more rows flags the window for the retry pass (poa_window_kernel2).  -1: ncap + 1.
RCN_HD uint64_t win_bind(Win& g, RCN_G uint8_t* base, int32_t ncap, int32_t ecap, int32_t ring, int32_t lmax, int32_t hstride, int32_t hcell = 4,
                         int32_t hrows = -1) {
    uint64_t off = 0;
    g.ncap = ncap; g.ecap = ecap; g.ring = ring; g.hstride = hstride;
    g.n_nodes = 0; g.n_edges = 0; g.overflow = 0;
#define RCN_TAKE(field, bytes) do { g.field.base = base; g.field.off = static_cast<uint32_t>(off); \
                                    off += (static_cast<uint64_t>(bytes) + 15) & ~uint64_t(15); } while (0)
    const uint64_t n = static_cast<uint64_t>(ncap), e = static_cast<uint64_t>(ecap);
    RCN_TAKE(code, n); RCN_TAKE(al_cnt, n); RCN_TAKE(mark, n); RCN_TAKE(inc, n);
    RCN_TAKE(in_head, 4 * n); RCN_TAKE(in_tail, 4 * n); RCN_TAKE(out_head, 4 * n); RCN_TAKE(out_tail, 4 * n);
    RCN_TAKE(in6, sizeof(PredRec) * n);
    RCN_TAKE(cov, 4 * n); RCN_TAKE(al_nodes, 4 * n * ring);
    RCN_TAKE(rank_full, 4 * n); RCN_TAKE(rank_tmp, 4 * n); RCN_TAKE(rank_sub, 4 * n); RCN_TAKE(rank_x, 4 * n);
    RCN_TAKE(n2r, 4 * n); RCN_TAKE(n2r_x, 4 * n); RCN_TAKE(pred, 4 * (n + 1));
    RCN_TAKE(new_id, 4 * (static_cast<uint64_t>(lmax) + 2)); RCN_TAKE(new_anchor, 4 * (static_cast<uint64_t>(lmax) + 2));
    RCN_TAKE(pos_t, 4 * (static_cast<uint64_t>(lmax) + 2)); RCN_TAKE(pos_curr, 4 * (static_cast<uint64_t>(lmax) + 2));
    RCN_TAKE(pos_a, 4 * (static_cast<uint64_t>(lmax) + 2));
    RCN_TAKE(score, 8 * n);
    RCN_TAKE(e_tail, 4 * e); RCN_TAKE(e_head, 4 * e); RCN_TAKE(e_nin, 4 * e); RCN_TAKE(e_nout, 4 * e);
    RCN_TAKE(e_w, 8 * e);
    RCN_TAKE(path_node, 4 * (n + lmax + 2)); RCN_TAKE(path_pos, 4 * (n + lmax + 2));
    RCN_TAKE(desc, sizeof(RowDesc) * n);
    RCN_TAKE(stack, 4 * (e + n * (ring + 1) + 64));
    const uint64_t hints = (hrows > 0 ? static_cast<uint64_t>(hrows) : n + 1) * static_cast<uint64_t>(hstride);
    g.hcap = static_cast<int64_t>(hints);
    RCN_TAKE(H, static_cast<uint64_t>(hcell) * hints + 1024);   // last array; +1 KiB: traceback tiles may read past the final row
#undef RCN_TAKE
    return off;
}

}  // namespace rcn
