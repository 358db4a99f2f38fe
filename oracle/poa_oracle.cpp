// poa_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code).
//
// A scalar restatement of the per-window consensus of lbcb-sci/racon:
//   racon::Window::generate_consensus            reference src/window.cpp:65-149
// over the parts of rvaser/spoa 4.0.8 it calls (pinned in the reference's
// CMakeLists.txt:60-65; the spoa sources are NOT in /root/reference — vendor/spoa
// is an empty submodule — so what follows restates spoa's published algorithm
// and is pinned through racon's own end-to-end goldens, test/racon_test.cpp:86-295;
// see tests/test_oracle_goldens.py):
//   spoa::Graph::AddAlignment / AddSequence / AddEdge / TopologicalSort /
//               Subgraph / UpdateAlignment / GenerateConsensus (+ heaviest bundle,
//               branch completion, Node::Coverage)
//   spoa::AlignmentEngine::Align  (kNW, linear gap; SISD semantics)
// call sites: reference src/window.cpp:73-77, 95-97, 100-107, 111-118, 123 and
// src/polisher.cpp:180-182.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library.  The product (racon_amd/csrc) never links it.
//
// Deliberately simple: std::vector graph, int32 full DP matrix, serial traceback.

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <immintrin.h>

#include "../include/racon_hip.h"

namespace {

using Alignment = std::vector<std::pair<int32_t, int32_t>>;

struct Graph {
    int32_t n_backbone = 0;       // nodes 0..n_backbone-1 are backbone nodes (sink-tie rule check only)
    struct Node {
        int32_t code;
        std::vector<int32_t> in, out;      // edge ids, creation order
        std::vector<int32_t> aligned;      // node ids, insertion order
    };
    struct Edge {
        int32_t tail, head;
        int64_t weight;
        std::vector<uint32_t> labels;
    };
    int32_t num_codes = 0;
    int32_t coder[256];
    uint8_t decoder[256];
    uint32_t num_sequences = 0;
    std::vector<Node> nodes;
    std::vector<Edge> edges;
    std::vector<int32_t> rank_to_node;

    Graph() { for (auto& c : coder) c = -1; std::memset(decoder, 0, sizeof(decoder)); }

    int32_t AddNode(int32_t code) {
        nodes.push_back(Node{code, {}, {}, {}});
        return static_cast<int32_t>(nodes.size()) - 1;
    }

    void AddEdge(int32_t tail, int32_t head, int64_t w) {
        for (int32_t e : nodes[tail].out) {
            if (edges[e].head == head) {
                edges[e].labels.push_back(num_sequences);
                edges[e].weight += w;
                return;
            }
        }
        int32_t id = static_cast<int32_t>(edges.size());
        edges.push_back(Edge{tail, head, w, {num_sequences}});
        nodes[tail].out.push_back(id);
        nodes[head].in.push_back(id);
    }

    // chain of new nodes for seq[b, e); returns first node id or -1
    int32_t AddSequence(const uint8_t* seq, const std::vector<uint32_t>& w, uint32_t b, uint32_t e) {
        if (b == e) return -1;
        int32_t first = -1, prev = -1;
        for (uint32_t i = b; i < e; ++i) {
            int32_t c = AddNode(coder[seq[i]]);
            if (first < 0) first = c;
            if (prev >= 0) AddEdge(prev, c, static_cast<int64_t>(static_cast<uint32_t>(w[i - 1] + w[i])));   // uint32 sum, as spoa
            prev = c;
        }
        return first;
    }

    void AddAlignment(const Alignment& al, const uint8_t* seq, uint32_t len,
                      const std::vector<uint32_t>& w) {
        if (len == 0) return;
        for (uint32_t i = 0; i < len; ++i) {
            if (coder[seq[i]] == -1) {
                coder[seq[i]] = num_codes;
                decoder[num_codes] = seq[i];
                ++num_codes;
            }
        }
        if (al.empty()) {
            AddSequence(seq, w, 0, len);
            ++num_sequences;
            TopologicalSort();
            return;
        }
        std::vector<uint32_t> valid;
        for (const auto& p : al) if (p.second != -1) valid.push_back(p.second);
        if (valid.empty()) { fprintf(stderr, "[oracle] invalid alignment\n"); exit(1); }

        int32_t begin = AddSequence(seq, w, 0, valid.front());
        int32_t prev = begin < 0 ? -1 : static_cast<int32_t>(nodes.size()) - 1;
        int32_t last = AddSequence(seq, w, valid.back() + 1, len);

        for (const auto& p : al) {
            if (p.second == -1) continue;
            int32_t code = coder[seq[p.second]];
            int32_t curr = -1;
            if (p.first == -1) {
                curr = AddNode(code);
            } else {
                int32_t t = p.first;
                if (nodes[t].code == code) {
                    curr = t;
                } else {
                    for (int32_t a : nodes[t].aligned) {
                        if (nodes[a].code == code) { curr = a; break; }
                    }
                    if (curr < 0) {
                        curr = AddNode(code);
                        // copy: nodes may have reallocated, and we mutate lists while iterating
                        std::vector<int32_t> ring = nodes[t].aligned;
                        for (int32_t a : ring) {
                            nodes[a].aligned.push_back(curr);
                            nodes[curr].aligned.push_back(a);
                        }
                        nodes[t].aligned.push_back(curr);
                        nodes[curr].aligned.push_back(t);
                    }
                }
            }
            if (begin < 0) begin = curr;
            if (prev >= 0) {
                AddEdge(prev, curr, static_cast<int64_t>(static_cast<uint32_t>(w[p.second - 1] + w[p.second])));
            }
            prev = curr;
        }
        if (last >= 0) {
            AddEdge(prev, last, static_cast<int64_t>(static_cast<uint32_t>(w[valid.back()] + w[valid.back() + 1])));
        }
        ++num_sequences;
        TopologicalSort();
    }

    void TopologicalSort() {
        const size_t n = nodes.size();
        rank_to_node.clear();
        std::vector<uint8_t> marks(n, 0), ignored(n, 0);
        std::vector<int32_t> st;
        for (size_t s = 0; s < n; ++s) {
            if (marks[s] != 0) continue;
            st.push_back(static_cast<int32_t>(s));
            while (!st.empty()) {
                int32_t c = st.back();
                bool valid = true;
                if (marks[c] != 2) {
                    for (int32_t e : nodes[c].in) {
                        int32_t t = edges[e].tail;
                        if (marks[t] != 2) { st.push_back(t); valid = false; }
                    }
                    if (!ignored[c]) {
                        for (int32_t a : nodes[c].aligned) {
                            if (marks[a] != 2) { st.push_back(a); ignored[a] = 1; valid = false; }
                        }
                    }
                    if (valid) {
                        marks[c] = 2;
                        if (!ignored[c]) {
                            rank_to_node.push_back(c);
                            for (int32_t a : nodes[c].aligned) rank_to_node.push_back(a);
                        }
                    } else {
                        marks[c] = 1;
                    }
                }
                if (valid) st.pop_back();
            }
        }
        if (rank_to_node.size() != n) { fprintf(stderr, "[oracle] graph is not a DAG\n"); exit(1); }
    }

    // Subgraph(begin, end, &mapping): backward reachability from node `end`
    // restricted to ids >= begin; fresh graph in ascending original id.
    Graph Subgraph(uint32_t begin, uint32_t end, std::vector<int32_t>* mapping) const {
        const size_t n = nodes.size();
        std::vector<uint8_t> inc(n, 0);
        std::vector<int32_t> st{static_cast<int32_t>(end)};
        while (!st.empty()) {
            int32_t c = st.back(); st.pop_back();
            if (!inc[c] && static_cast<uint32_t>(c) >= begin) {
                for (int32_t e : nodes[c].in) st.push_back(edges[e].tail);
                for (int32_t a : nodes[c].aligned) st.push_back(a);
                inc[c] = 1;
            }
        }
        Graph g;
        g.num_codes = num_codes;
        std::memcpy(g.coder, coder, sizeof(coder));
        std::memcpy(g.decoder, decoder, sizeof(decoder));
        std::vector<int32_t> g2s(n, -1);
        mapping->clear();
        for (size_t i = 0; i < n; ++i) {
            if (inc[i]) { g2s[i] = g.AddNode(nodes[i].code); mapping->push_back(static_cast<int32_t>(i)); }
        }
        for (size_t i = 0; i < n; ++i) {
            if (!inc[i]) continue;
            int32_t j = g2s[i];
            for (int32_t e : nodes[i].in) {
                int32_t t = g2s[edges[e].tail];
                if (t >= 0) g.AddEdge(t, j, edges[e].weight);
            }
            for (int32_t a : nodes[i].aligned) {
                if (g2s[a] >= 0) g.nodes[j].aligned.push_back(g2s[a]);
            }
        }
        g.TopologicalSort();
        return g;
    }

    uint32_t Coverage(int32_t v) const {
        std::vector<uint32_t> lab;
        for (int32_t e : nodes[v].in) lab.insert(lab.end(), edges[e].labels.begin(), edges[e].labels.end());
        for (int32_t e : nodes[v].out) lab.insert(lab.end(), edges[e].labels.begin(), edges[e].labels.end());
        std::sort(lab.begin(), lab.end());
        return static_cast<uint32_t>(std::unique(lab.begin(), lab.end()) - lab.begin());
    }

    std::string GenerateConsensus(std::vector<uint32_t>* coverages) const {
        const int32_t n = static_cast<int32_t>(nodes.size());
        std::vector<int32_t> pred(n, -1);
        std::vector<int64_t> sc(n, -1);
        int32_t mx = -1;
        auto relax = [&](int32_t it, bool skip) {
            for (int32_t e : nodes[it].in) {
                int32_t t = edges[e].tail;
                if (skip && sc[t] == -1) continue;
                int64_t w = edges[e].weight;
                if (sc[it] < w || (sc[it] == w && sc[pred[it]] <= sc[t])) {
                    sc[it] = w; pred[it] = t;
                }
            }
            if (pred[it] >= 0) sc[it] += sc[pred[it]];
        };
        for (int32_t it : rank_to_node) {
            relax(it, false);
            if (mx < 0 || sc[mx] < sc[it]) mx = it;
        }
        if (!nodes[mx].out.empty()) {
            std::vector<int32_t> n2r(n, 0);
            for (int32_t i = 0; i < n; ++i) n2r[rank_to_node[i]] = i;
            while (!nodes[mx].out.empty()) {
                int32_t start = mx;
                for (int32_t e : nodes[start].out) {
                    for (int32_t f : nodes[edges[e].head].in) {
                        if (edges[f].tail != start) sc[edges[f].tail] = -1;
                    }
                }
                int32_t m2 = -1;
                for (int32_t i = n2r[start] + 1; i < n; ++i) {
                    int32_t it = rank_to_node[i];
                    sc[it] = -1; pred[it] = -1;
                    relax(it, true);
                    if (m2 < 0 || sc[m2] < sc[it]) m2 = it;
                }
                mx = m2;
            }
        }
        std::vector<int32_t> cons;
        while (pred[mx] >= 0) { cons.push_back(mx); mx = pred[mx]; }
        cons.push_back(mx);
        std::reverse(cons.begin(), cons.end());
        std::string s;
        coverages->clear();
        for (int32_t v : cons) {
            s += static_cast<char>(decoder[nodes[v].code]);
            uint32_t c = Coverage(v);
            for (int32_t a : nodes[v].aligned) c += Coverage(a);
            coverages->push_back(c);
        }
        return s;
    }
};

struct Engine {
    int32_t m, x, g;
    std::vector<int32_t> H;              // (V+1) x (L+1)
    std::vector<int32_t> profile;        // num_codes x (L+1)
    uint64_t cells = 0;                  // Σ (V+1)(L+1)
    double   cells_x_pred = 0;           // Σ cells * (1 + E/V)

    // Statistics for tests/test_oracle_spec.py: the HIP kernel resolves "several sinks share the best score"
    // without spoa's DFS order where a rule proves the winner (racon_amd/csrc/poa_kernel2.hpp,
    // phase_sink_tie_rule): key (smallest id in the sink's aligned ring, sink id) when that ring holds a
    // backbone node, (inf, id) for a non-backbone sink without aligned nodes.  Here the rule is evaluated next
    // to the exact choice (first tied sink in rank_to_node order) and must agree whenever it applies.
    uint64_t tie_events = 0, tie_ruled = 0, tie_rule_agrees = 0;
    void check_sink_tie_rule(const Graph& g_, const std::vector<int32_t>& n2r, size_t W, uint32_t L, int32_t best, int32_t bi) {
        const int32_t V = static_cast<int32_t>(g_.nodes.size());
        std::vector<int32_t> tied;
        for (int32_t r = 0; r < V; ++r) {
            const int32_t v = g_.rank_to_node[r];
            if (g_.nodes[v].out.empty() && H[static_cast<size_t>(r + 1) * W + L] == best) tied.push_back(v);
        }
        if (tied.size() < 2) return;
        (void)n2r;
        ++tie_events;
        const int32_t exact = g_.rank_to_node[bi - 1];
        bool classified = true; int64_t bestkey = INT64_MAX; int32_t pick = -1;
        for (int32_t v : tied) {
            int32_t rm = v;
            for (int32_t a : g_.nodes[v].aligned) rm = std::min(rm, a);
            int64_t key;
            if (rm < g_.n_backbone) key = (static_cast<int64_t>(rm) << 32) | static_cast<uint32_t>(v);
            else if (g_.nodes[v].aligned.empty()) key = (static_cast<int64_t>(0x7ffffffe) << 32) | static_cast<uint32_t>(v);
            else { classified = false; break; }
            if (key < bestkey) { bestkey = key; pick = v; }
        }
        if (classified) { ++tie_ruled; if (pick == exact) ++tie_rule_agrees; }
    }

    // ---- band study (test infrastructure for the kernel's exact banded DP, racon_amd/csrc/poa_kernel2.hpp) ----
    // Re-runs the alignment just computed with a WINDOW of band_wb columns per row (offset: a non-decreasing step
    // function of the row's backbone coordinate, quantised to band_g columns), cells outside their row's window = -inf,
    // and evaluates the exactness certificate next to the ground truth (the full matrix H is still in memory):
    //   T = best banded end score; a cell is ALIVE iff H'[i][j] + m (L - j) >= T (no path through a dead cell reaches T).
    //   exact  : every alive cell has all its DP successors inside their rows' windows  =>  every path scoring >= T is
    //            made of alive cells with H' = H  =>  same end cell, same traceback decisions as the full matrix
    //   cheap  : what the kernel can afford per row (Z = H - j g is non-decreasing along a row): the last window cell of
    //            a row is dead, and for every edge p -> s whose window starts further right the cell left of s's window
    //            start is dead even at the threshold of p's window start
    int32_t band_wb = 0, band_g = 16;
    struct BandStats { uint64_t n = 0, banded = 0, exact_ok = 0, cheap_ok = 0, same_path = 0, bad = 0, width_max = 0, width_sum = 0, rows = 0, shifts = 0; } bs;
    void band_study(const uint8_t* seq, uint32_t L, const Graph& g_, const std::vector<int32_t>& n2r, int32_t bi, const Alignment& full_al);

    // ---- CPU baseline variant (bench.py's cpu_baseline leg; NOT the oracle of the parity tests) ----
    // Same recurrence, same tie-breaks, but evaluated the way spoa's SIMD engine does it: int16 row vectors
    // (AVX2, 16 cells per op) and a log-step prefix max for the horizontal gap.  Scores are kept as
    // Z[i][j] = H[i][j] - j*g (the horizontal move then adds 0); -|g|V <= Z <= (max(m,x,0)+|g|)W bounds them,
    // rows that do not fit int16 take the scalar path below.  tests/test_oracle_spec.py checks it against the
    // scalar oracle window by window.
    bool simd = false;
    std::vector<int16_t> Z16, prof16, mrow16;

    static inline __m256i shl1(__m256i x, __m256i prevblk) {          // element j <- x[j-1], x[-1] = prevblk[15]
        const __m256i pv = _mm256_permute2x128_si256(prevblk, x, 0x21);
        return _mm256_alignr_epi8(x, pv, 14);
    }
    static inline __m256i prefix_max16(__m256i x) {
        const __m256i NEGV = _mm256_set1_epi16(-32768);
        __m256i pv = _mm256_permute2x128_si256(NEGV, x, 0x20);
        x = _mm256_max_epi16(x, _mm256_alignr_epi8(x, pv, 14));
        pv = _mm256_permute2x128_si256(NEGV, x, 0x20);
        x = _mm256_max_epi16(x, _mm256_alignr_epi8(x, pv, 12));
        pv = _mm256_permute2x128_si256(NEGV, x, 0x20);
        x = _mm256_max_epi16(x, _mm256_alignr_epi8(x, pv, 8));
        pv = _mm256_permute2x128_si256(NEGV, x, 0x20);
        return _mm256_max_epi16(x, pv);
    }

    bool fits_z16(int32_t V, size_t Wp) const {
        const int32_t ag = g < 0 ? -g : g, smax = std::max(std::max(m, x), 0);
        return g < 0 && static_cast<int64_t>(ag) * (V + 2) <= 31000 && static_cast<int64_t>(smax + ag) * static_cast<int64_t>(Wp) <= 31000;
    }

    Alignment AlignZ16(const uint8_t* seq, uint32_t L, const Graph& g_) {
        const int32_t V = static_cast<int32_t>(g_.nodes.size());
        const size_t W = static_cast<size_t>(L) + 1, Wp = (W + 15) / 16 * 16, nb = Wp / 16;
        Z16.resize(static_cast<size_t>(V + 1) * Wp);
        prof16.resize(static_cast<size_t>(g_.num_codes) * Wp);
        mrow16.resize(Wp);
        for (int32_t c = 0; c < g_.num_codes; ++c) {
            int16_t* P = &prof16[c * Wp];
            P[0] = 0;
            for (size_t j = 1; j < Wp; ++j) P[j] = static_cast<int16_t>(((j <= L && g_.decoder[c] == seq[j - 1]) ? m : x) - g);
        }
        cells += static_cast<uint64_t>(V + 1) * W;
        cells_x_pred += static_cast<double>(V + 1) * W * (1.0 + static_cast<double>(g_.edges.size()) / V);
        std::vector<int32_t> n2r(V);
        for (int32_t r = 0; r < V; ++r) n2r[g_.rank_to_node[r]] = r;
        std::memset(Z16.data(), 0, Wp * sizeof(int16_t));                       // row 0 of Z
        const __m256i GV = _mm256_set1_epi16(static_cast<int16_t>(g)), NEG = _mm256_set1_epi16(-32000);
        bool have_best = false; int32_t best = 0, bi = 0;
        std::vector<int32_t> ps;
        for (int32_t r = 0; r < V; ++r) {
            const auto& node = g_.nodes[g_.rank_to_node[r]];
            int16_t* row = &Z16[static_cast<size_t>(r + 1) * Wp];
            const int16_t* P = &prof16[node.code * Wp];
            ps.clear();
            for (int32_t e : node.in) ps.push_back(n2r[g_.edges[e].tail] + 1);
            if (ps.empty()) ps.push_back(0);
            const int16_t* M = &Z16[static_cast<size_t>(ps[0]) * Wp];
            if (ps.size() > 1) {
                for (size_t b = 0; b < nb; ++b) {
                    __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(M) + b);
                    for (size_t k = 1; k < ps.size(); ++k)
                        v = _mm256_max_epi16(v, _mm256_loadu_si256(reinterpret_cast<const __m256i*>(&Z16[static_cast<size_t>(ps[k]) * Wp]) + b));
                    _mm256_storeu_si256(reinterpret_cast<__m256i*>(mrow16.data()) + b, v);
                }
                M = mrow16.data();
            }
            __m256i prevm = NEG, carry = _mm256_set1_epi16(-32768);
            for (size_t b = 0; b < nb; ++b) {
                const __m256i mv = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(M) + b);
                const __m256i D = shl1(mv, prevm);
                const __m256i pr = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(P) + b);
                __m256i acc = _mm256_max_epi16(_mm256_adds_epi16(D, pr), _mm256_adds_epi16(mv, GV));
                acc = _mm256_max_epi16(prefix_max16(acc), carry);
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(row) + b, acc);
                carry = _mm256_set1_epi16(static_cast<int16_t>(_mm256_extract_epi16(acc, 15)));
                prevm = mv;
            }
            if (node.out.empty()) {
                if (!have_best || best < row[L]) { have_best = true; best = row[L]; bi = r + 1; }
            }
        }
        Alignment al;
        int32_t i = bi, j = static_cast<int32_t>(L);
        while (!(i == 0 && j == 0)) {
            const int32_t zij = Z16[static_cast<size_t>(i) * Wp + j];
            int32_t pi = 0, pj = 0; bool found = false;
            if (i != 0) {
                const auto& node = g_.nodes[g_.rank_to_node[i - 1]];
                ps.clear();
                for (int32_t e : node.in) ps.push_back(n2r[g_.edges[e].tail] + 1);
                if (ps.empty()) ps.push_back(0);
                if (j != 0) {
                    const int32_t mc = prof16[node.code * Wp + j];
                    for (int32_t p : ps) if (zij == Z16[static_cast<size_t>(p) * Wp + j - 1] + mc) { pi = p; pj = j - 1; found = true; break; }
                }
                if (!found) for (int32_t p : ps) if (zij == Z16[static_cast<size_t>(p) * Wp + j] + g) { pi = p; pj = j; found = true; break; }
            }
            if (!found && j != 0 && zij == Z16[static_cast<size_t>(i) * Wp + j - 1]) { pi = i; pj = j - 1; found = true; }
            if (!found) { fprintf(stderr, "[oracle/simd] traceback stuck at (%d,%d)\n", i, j); exit(1); }
            al.emplace_back(i == pi ? -1 : g_.rank_to_node[i - 1], j == pj ? -1 : j - 1);
            i = pi; j = pj;
        }
        std::reverse(al.begin(), al.end());
        return al;
    }

    Alignment Align(const uint8_t* seq, uint32_t L, const Graph& g_) {
        const int32_t V = static_cast<int32_t>(g_.nodes.size());
        if (V == 0 || L == 0) return {};
        if (simd && fits_z16(V, (static_cast<size_t>(L) + 16) / 16 * 16)) return AlignZ16(seq, L, g_);
        const size_t W = static_cast<size_t>(L) + 1;
        H.resize(static_cast<size_t>(V + 1) * W);
        profile.resize(static_cast<size_t>(g_.num_codes) * W);
        for (int32_t c = 0; c < g_.num_codes; ++c) {
            int32_t* P = &profile[c * W];
            P[0] = 0;
            for (uint32_t j = 1; j <= L; ++j) P[j] = (g_.decoder[c] == seq[j - 1]) ? m : x;
        }
        cells += static_cast<uint64_t>(V + 1) * W;
        cells_x_pred += static_cast<double>(V + 1) * W *
            (1.0 + static_cast<double>(g_.edges.size()) / V);

        std::vector<int32_t> n2r(V);
        for (int32_t r = 0; r < V; ++r) n2r[g_.rank_to_node[r]] = r;
        for (uint32_t j = 0; j <= L; ++j) H[j] = static_cast<int32_t>(j) * g;

        bool have_best = false; int32_t best = 0, bi = 0;
        std::vector<int32_t> ps;
        for (int32_t r = 0; r < V; ++r) {
            const int32_t v = g_.rank_to_node[r];
            const auto& node = g_.nodes[v];
            int32_t* row = &H[static_cast<size_t>(r + 1) * W];
            const int32_t* P = &profile[node.code * W];
            ps.clear();
            for (int32_t e : node.in) ps.push_back(n2r[g_.edges[e].tail] + 1);
            if (ps.empty()) ps.push_back(0);
            {
                const int32_t* Hp = &H[static_cast<size_t>(ps[0]) * W];
                int32_t h0 = Hp[0];
                for (uint32_t j = 1; j <= L; ++j) row[j] = std::max(Hp[j - 1] + P[j], Hp[j] + g);
                for (size_t k = 1; k < ps.size(); ++k) {
                    Hp = &H[static_cast<size_t>(ps[k]) * W];
                    h0 = std::max(h0, Hp[0]);
                    for (uint32_t j = 1; j <= L; ++j)
                        row[j] = std::max(row[j], std::max(Hp[j - 1] + P[j], Hp[j] + g));
                }
                row[0] = (node.in.empty() ? 0 : h0) + g;
            }
            for (uint32_t j = 1; j <= L; ++j) row[j] = std::max(row[j - 1] + g, row[j]);
            if (node.out.empty()) {
                if (!have_best || best < row[L]) { have_best = true; best = row[L]; bi = r + 1; }
            }
        }

        check_sink_tie_rule(g_, n2r, W, L, best, bi);
        Alignment al;
        int32_t i = bi, j = static_cast<int32_t>(L);
        while (!(i == 0 && j == 0)) {
            const int32_t hij = H[static_cast<size_t>(i) * W + j];
            int32_t pi = 0, pj = 0; bool found = false;
            if (i != 0) {
                const int32_t v = g_.rank_to_node[i - 1];
                const auto& node = g_.nodes[v];
                if (j != 0) {
                    const int32_t mc = profile[node.code * W + j];
                    if (node.in.empty()) {
                        if (hij == H[j - 1] + mc) { pi = 0; pj = j - 1; found = true; }
                    } else {
                        for (int32_t e : node.in) {
                            int32_t p = n2r[g_.edges[e].tail] + 1;
                            if (hij == H[static_cast<size_t>(p) * W + j - 1] + mc) {
                                pi = p; pj = j - 1; found = true; break;
                            }
                        }
                    }
                }
                if (!found) {
                    if (node.in.empty()) {
                        if (hij == H[j] + g) { pi = 0; pj = j; found = true; }
                    } else {
                        for (int32_t e : node.in) {
                            int32_t p = n2r[g_.edges[e].tail] + 1;
                            if (hij == H[static_cast<size_t>(p) * W + j] + g) {
                                pi = p; pj = j; found = true; break;
                            }
                        }
                    }
                }
            }
            if (!found && j != 0 && hij == H[static_cast<size_t>(i) * W + j - 1] + g) {
                pi = i; pj = j - 1; found = true;
            }
            if (!found) { fprintf(stderr, "[oracle] traceback stuck at (%d,%d)\n", i, j); exit(1); }
            al.emplace_back(i == pi ? -1 : g_.rank_to_node[i - 1], j == pj ? -1 : j - 1);
            i = pi; j = pj;
        }
        std::reverse(al.begin(), al.end());
        if (band_wb > 0) band_study(seq, L, g_, n2r, bi, al);
        return al;
    }
};

void Engine::band_study(const uint8_t* seq, uint32_t L, const Graph& g_, const std::vector<int32_t>& n2r, int32_t bi, const Alignment& full_al) {
    (void)seq;
    const int32_t V = static_cast<int32_t>(g_.nodes.size());
    const int32_t W = static_cast<int32_t>(L) + 1, Wb = band_wb, G = band_g;
    ++bs.n;
    if (W <= Wb) return;                                   // the whole row fits: nothing to band
    ++bs.banded;
    constexpr int32_t NEG = -(1 << 28);
    // window offset per row (row 0 = virtual start row: offset 0)
    const int32_t nbb = std::max(1, g_.n_backbone);
    const int32_t offmax = ((W - Wb + G - 1) / G) * G;
    std::vector<int32_t> off(V + 1, 0);
    int32_t pi_row = -1, cur = 0;
    for (int32_t r = 0; r < V; ++r) {
        const int32_t v = g_.rank_to_node[r];
        if (v < g_.n_backbone) pi_row = std::max(pi_row, v);
        const int64_t center = static_cast<int64_t>(pi_row + 1) * L / nbb;
        int32_t o = static_cast<int32_t>(center) - Wb / 2;
        o = o < 0 ? 0 : (o / G) * G;
        o = std::min(o, offmax);
        if (o > cur) { cur = o; ++bs.shifts; }
        off[r + 1] = cur;
    }
    bs.rows += V;
    std::vector<int32_t> Hb(static_cast<size_t>(V + 1) * W, NEG);
    auto in_win = [&](int32_t row, int32_t j) { return j >= off[row] && j < off[row] + Wb && j < W; };
    for (int32_t j = 0; j < W && j < Wb; ++j) Hb[j] = j * g;
    std::vector<int32_t> ps;
    bool have_best = false; int32_t best = 0, bib = 0;
    for (int32_t r = 0; r < V; ++r) {
        const auto& node = g_.nodes[g_.rank_to_node[r]];
        int32_t* row = &Hb[static_cast<size_t>(r + 1) * W];
        const int32_t* P = &profile[node.code * W];
        ps.clear();
        for (int32_t e : node.in) ps.push_back(n2r[g_.edges[e].tail] + 1);
        if (ps.empty()) ps.push_back(0);
        const int32_t lo = off[r + 1], hi = std::min(W, lo + Wb);
        for (int32_t j = lo; j < hi; ++j) {
            int32_t v = NEG;
            for (int32_t p : ps) {
                const int32_t* Hp = &Hb[static_cast<size_t>(p) * W];
                if (j >= 1 && Hp[j - 1] > NEG) v = std::max(v, Hp[j - 1] + P[j]);
                if (Hp[j] > NEG) v = std::max(v, Hp[j] + g);
            }
            if (j == 0) { int32_t h0 = NEG; for (int32_t p : ps) h0 = std::max(h0, Hb[static_cast<size_t>(p) * W]); v = (node.in.empty() ? 0 : h0) + g; }
            if (j > lo && row[j - 1] > NEG) v = std::max(v, row[j - 1] + g);
            row[j] = v;
        }
        if (node.out.empty() && in_win(r + 1, static_cast<int32_t>(L)) && row[L] > NEG) {
            if (!have_best || best < row[L]) { have_best = true; best = row[L]; bib = r + 1; }
        }
    }
    if (!have_best) return;
    const int32_t T = best;
    auto alive = [&](int32_t row, int32_t j) { const int32_t h = Hb[static_cast<size_t>(row) * W + j]; return h > NEG && h + m * (static_cast<int32_t>(L) - j) >= T; };
    // successors of row 0: nodes without in-edges
    bool exact_ok = true, cheap_ok = true;
    uint64_t wmax = 0;
    auto check_edge = [&](int32_t prow, int32_t srow) {
        const int32_t lo = off[prow], hi = std::min(W, lo + Wb);
        for (int32_t j = lo; j < hi; ++j) {
            if (!alive(prow, j)) continue;
            if (!in_win(srow, j)) exact_ok = false;
            if (j + 1 < W && !in_win(srow, j + 1)) exact_ok = false;
        }
        if (off[srow] > off[prow]) {
            // cheap: Z of p at the column left of s's window start, against the threshold at p's window start
            const int32_t jc = std::min(off[srow] - 1, hi - 1);
            const int32_t h = Hb[static_cast<size_t>(prow) * W + jc];
            if (h > NEG) {
                const int64_t z = static_cast<int64_t>(h) - static_cast<int64_t>(jc) * g;
                const int64_t thr = static_cast<int64_t>(T) - static_cast<int64_t>(m) * L + static_cast<int64_t>(m - g) * lo;
                if (z >= thr) cheap_ok = false;
            }
        } else if (off[srow] < off[prow]) cheap_ok = false;
    };
    for (int32_t r = 0; r <= V; ++r) {
        const int32_t lo = off[r], hi = std::min(W, lo + Wb);
        int32_t a0 = -1, a1 = -1;
        for (int32_t j = lo; j < hi; ++j) if (alive(r, j)) { if (a0 < 0) a0 = j; a1 = j; }
        if (a0 >= 0) { wmax = std::max<uint64_t>(wmax, a1 - a0 + 1); bs.width_sum += a1 - a0 + 1; }
        // horizontal successor of the last window cell
        if (hi < W) {
            if (alive(r, hi - 1)) exact_ok = false;
            if (alive(r, hi - 1)) cheap_ok = false;
        }
        if (r >= 1) {
            const auto& node = g_.nodes[g_.rank_to_node[r - 1]];
            if (node.in.empty()) check_edge(0, r);
            for (int32_t e : node.in) check_edge(n2r[g_.edges[e].tail] + 1, r);
        }
    }
    bs.width_max = std::max(bs.width_max, wmax);
    // banded traceback (cells outside their window never match: they are -inf)
    Alignment al;
    {
        int32_t i = bib, j = static_cast<int32_t>(L);
        bool stuck = false;
        while (!(i == 0 && j == 0) && !stuck) {
            const int32_t hij = Hb[static_cast<size_t>(i) * W + j];
            int32_t pi = 0, pj = 0; bool found = false;
            if (i != 0) {
                const auto& node = g_.nodes[g_.rank_to_node[i - 1]];
                ps.clear();
                for (int32_t e : node.in) ps.push_back(n2r[g_.edges[e].tail] + 1);
                if (ps.empty()) ps.push_back(0);
                if (j != 0) {
                    const int32_t mc = profile[node.code * W + j];
                    for (int32_t p : ps) { const int32_t h = Hb[static_cast<size_t>(p) * W + j - 1]; if (h > NEG && hij == h + mc) { pi = p; pj = j - 1; found = true; break; } }
                }
                if (!found) for (int32_t p : ps) { const int32_t h = Hb[static_cast<size_t>(p) * W + j]; if (h > NEG && hij == h + g) { pi = p; pj = j; found = true; break; } }
            }
            if (!found && j != 0) { const int32_t h = Hb[static_cast<size_t>(i) * W + j - 1]; if (h > NEG && hij == h + g) { pi = i; pj = j - 1; found = true; } }
            if (!found) { stuck = true; break; }
            al.emplace_back(i == pi ? -1 : g_.rank_to_node[i - 1], j == pj ? -1 : j - 1);
            i = pi; j = pj;
        }
        std::reverse(al.begin(), al.end());
        if (stuck) al.clear();
    }
    const bool same = (bib == bi) && al == full_al;
    bs.exact_ok += exact_ok; bs.cheap_ok += cheap_ok; bs.same_path += same;
    if ((exact_ok || cheap_ok) && !same) ++bs.bad;          // a passing certificate with a different result = the theory is wrong
}

struct WindowOut {
    std::string consensus;
    bool polished = false;
    bool chimeric = false;
};

// racon::Window::generate_consensus — reference src/window.cpp:65-149
WindowOut window_consensus(const rcn_batch* b, uint32_t w, Engine& eng, bool trim) {
    WindowOut out;
    const uint32_t s0 = b->win_seq_off[w], s1 = b->win_seq_off[w + 1];
    const uint32_t nseq = s1 - s0;
    auto seq_ptr = [&](uint32_t i) { return b->bases + b->seq_off[s0 + i]; };
    auto seq_len = [&](uint32_t i) { return static_cast<uint32_t>(b->seq_off[s0 + i + 1] - b->seq_off[s0 + i]); };
    auto weights = [&](uint32_t i) {
        const uint32_t len = seq_len(i);
        std::vector<uint32_t> wv(len, 1u);               // no-quality overload: weight 1 per base
        if (b->seq_has_qual[s0 + i]) {
            const uint8_t* q = b->quals + b->seq_off[s0 + i];
            // char -> uint32_t(quality[i] - 33); racon passes `const char*`
            for (uint32_t k = 0; k < len; ++k)
                wv[k] = static_cast<uint32_t>(static_cast<int32_t>(static_cast<signed char>(q[k])) - 33);
        }
        return wv;
    };

    const uint32_t L = seq_len(0);
    if (nseq < 3) {                                        // window.cpp:68-71
        out.consensus.assign(reinterpret_cast<const char*>(seq_ptr(0)), L);
        return out;
    }

    Graph graph;
    graph.AddAlignment(Alignment(), seq_ptr(0), L, weights(0));   // window.cpp:73-77
    graph.n_backbone = static_cast<int32_t>(L);

    std::vector<uint32_t> rank(nseq);
    for (uint32_t i = 0; i < nseq; ++i) rank[i] = i;
    std::sort(rank.begin() + 1, rank.end(), [&](uint32_t lhs, uint32_t rhs) {   // window.cpp:85-86
        return b->seq_begin[s0 + lhs] < b->seq_begin[s0 + rhs]; });

    const uint32_t offset = static_cast<uint32_t>(0.01 * L);     // window.cpp:88
    for (uint32_t j = 1; j < nseq; ++j) {
        const uint32_t i = rank[j];
        const uint32_t bg = b->seq_begin[s0 + i], en = b->seq_end[s0 + i];
        Alignment al;
        if (bg < offset && en > L - offset) {
            al = eng.Align(seq_ptr(i), seq_len(i), graph);
        } else {
            std::vector<int32_t> mapping;
            Graph sub = graph.Subgraph(bg, en, &mapping);
            for (int32_t v : mapping) if (v < static_cast<int32_t>(L)) ++sub.n_backbone;   // ids keep their order
            al = eng.Align(seq_ptr(i), seq_len(i), sub);
            for (auto& p : al) if (p.first != -1) p.first = mapping[p.first];
        }
        graph.AddAlignment(al, seq_ptr(i), seq_len(i), weights(i));
    }

    std::vector<uint32_t> cov;
    out.consensus = graph.GenerateConsensus(&cov);
    out.polished = true;

    if (b->win_type[w] == 1 && trim) {                        // window.cpp:125-146
        const uint32_t avg = (nseq - 1) / 2;
        int32_t begin = 0, end = static_cast<int32_t>(out.consensus.size()) - 1;
        for (; begin < static_cast<int32_t>(out.consensus.size()); ++begin)
            if (cov[begin] >= avg) break;
        for (; end >= 0; --end)
            if (cov[end] >= avg) break;
        if (begin >= end) out.chimeric = true;
        else out.consensus = out.consensus.substr(begin, end - begin + 1);
    }
    return out;
}

struct OracleResult {
    std::vector<uint64_t> cons_off;
    std::vector<uint8_t> cons, polished, chimeric;
    std::vector<uint64_t> cells;
    std::vector<double> cells_x_pred;
};

}  // namespace

extern "C" {

// Runs the oracle over a batch with `nthreads` host threads (windows are
// independent, reference src/polisher.cpp:496-503).  *handle must be released
// with rcn_oracle_free.  cells / cells_x_pred (each [n_windows], may be NULL)
// receive Σ(V'+1)(l+1) and Σ(V'+1)(l+1)(1+E'/V') per window.
// {sink-tie events, events the kernel's rule classifies, of those: rule == exact choice}; reset on read
static std::atomic<uint64_t> g_tie_stats[3];
void rcn_oracle_tie_stats(uint64_t* out3) { for (int k = 0; k < 3; ++k) out3[k] = g_tie_stats[k].exchange(0); }

static bool g_simd_next = false;
static int g_band_wb = 0, g_band_g = 16;
static std::atomic<uint64_t> g_band_stats[10];
// Band study switch + counters (n alignments, banded, exact certificate ok, cheap certificate ok, same result as the full
// matrix, certificate ok but different result (must stay 0), max alive width, sum of alive widths, rows, window shifts).
void rcn_oracle_band_study(int wb, int g) { g_band_wb = wb; g_band_g = g > 0 ? g : 16; }
void rcn_oracle_band_stats(uint64_t* out10) { for (int k = 0; k < 10; ++k) out10[k] = g_band_stats[k].exchange(0); }
int rcn_oracle_consensus(const rcn_batch* b, int m, int x, int g, int trim, int nthreads,
                         rcn_result* out, void** handle, uint64_t* cells, double* cells_x_pred);
// the AVX2 int16 variant (CPU baseline of bench.py); same results as rcn_oracle_consensus
int rcn_oracle_consensus_simd(const rcn_batch* b, int m, int x, int g, int trim, int nthreads,
                              rcn_result* out, void** handle, uint64_t* cells, double* cells_x_pred) {
    g_simd_next = true;
    const int rc = rcn_oracle_consensus(b, m, x, g, trim, nthreads, out, handle, cells, cells_x_pred);
    g_simd_next = false;
    return rc;
}

int rcn_oracle_consensus(const rcn_batch* b, int m, int x, int g, int trim, int nthreads,
                         rcn_result* out, void** handle, uint64_t* cells, double* cells_x_pred) {
    if (!b || !out || !handle) return RCN_E_ARG;
    const uint32_t n = b->n_windows;
    std::vector<WindowOut> res(n);
    std::vector<uint64_t> cl(n); std::vector<double> cx(n);
    std::atomic<uint32_t> next{0};
    const bool use_simd = g_simd_next;
    auto worker = [&]() {
        Engine eng; eng.m = m; eng.x = x; eng.g = g; eng.simd = use_simd;
        eng.band_wb = use_simd ? 0 : g_band_wb; eng.band_g = g_band_g;
        for (;;) {
            uint32_t w = next.fetch_add(1);
            if (w >= n) break;
            eng.cells = 0; eng.cells_x_pred = 0;
            res[w] = window_consensus(b, w, eng, trim != 0);
            cl[w] = eng.cells; cx[w] = eng.cells_x_pred;
        }
        g_tie_stats[0] += eng.tie_events; g_tie_stats[1] += eng.tie_ruled; g_tie_stats[2] += eng.tie_rule_agrees;
        const uint64_t bv[10] = {eng.bs.n, eng.bs.banded, eng.bs.exact_ok, eng.bs.cheap_ok, eng.bs.same_path, eng.bs.bad, 0, eng.bs.width_sum, eng.bs.rows, eng.bs.shifts};
        for (int k = 0; k < 10; ++k) if (k != 6) g_band_stats[k] += bv[k];
        for (uint64_t cur = g_band_stats[6].load(); eng.bs.width_max > cur && !g_band_stats[6].compare_exchange_weak(cur, eng.bs.width_max);) {}
    };
    if (nthreads <= 1) worker();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(worker);
        for (auto& t : th) t.join();
    }
    auto* r = new OracleResult();
    r->cons_off.resize(n + 1, 0); r->polished.resize(n); r->chimeric.resize(n);
    for (uint32_t w = 0; w < n; ++w) {
        r->cons_off[w + 1] = r->cons_off[w] + res[w].consensus.size();
        r->polished[w] = res[w].polished; r->chimeric[w] = res[w].chimeric;
    }
    r->cons.resize(r->cons_off[n] + 1);
    for (uint32_t w = 0; w < n; ++w)
        std::memcpy(r->cons.data() + r->cons_off[w], res[w].consensus.data(), res[w].consensus.size());
    out->n_windows = n; out->cons_off = r->cons_off.data(); out->cons = r->cons.data();
    out->polished = r->polished.data(); out->chimeric = r->chimeric.data();
    if (cells) std::memcpy(cells, cl.data(), n * sizeof(uint64_t));
    if (cells_x_pred) std::memcpy(cells_x_pred, cx.data(), n * sizeof(double));
    *handle = r;
    return RCN_OK;
}

void rcn_oracle_free(void* handle) { delete static_cast<OracleResult*>(handle); }

// Exact global (NW) unit-cost edit distance, Myers 1999 bit-vector, block based.
// Stand-in for edlibAlign(..., edlibDefaultAlignConfig()) in the reference's test
// helper calculateEditDistance (reference test/racon_test.cpp:14-23).
uint64_t rcn_oracle_edit_distance(const uint8_t* q, uint64_t m, const uint8_t* t, uint64_t n) {
    if (m == 0) return n;
    if (n == 0) return m;
    const uint64_t nb = (m + 63) / 64;
    std::vector<uint64_t> peq(256 * nb, 0), Pv(nb, ~0ull), Mv(nb, 0);
    for (uint64_t i = 0; i < m; ++i) peq[q[i] * nb + i / 64] |= 1ull << (i % 64);
    uint64_t score = m;
    const uint64_t last_bit = 1ull << ((m - 1) % 64);
    for (uint64_t j = 0; j < n; ++j) {
        const uint64_t* eqc = &peq[t[j] * nb];
        int hin = 1;   // global: row 0 increases by 1 per column
        for (uint64_t k = 0; k < nb; ++k) {
            uint64_t Eq = eqc[k], pv = Pv[k], mv = Mv[k];
            uint64_t hinNeg = hin < 0 ? 1ull : 0ull;
            uint64_t Xv = Eq | mv;
            Eq |= hinNeg;
            uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
            uint64_t Ph = mv | ~(Xh | pv);
            uint64_t Mh = pv & Xh;
            int hout = 0;
            const uint64_t top = (k == nb - 1) ? last_bit : (1ull << 63);
            if (Ph & top) hout = 1; else if (Mh & top) hout = -1;
            Ph <<= 1; Mh <<= 1;
            if (hin < 0) Mh |= 1; else if (hin > 0) Ph |= 1;
            Pv[k] = Mh | ~(Xv | Ph);
            Mv[k] = Ph & Xv;
            hin = hout;
        }
        score += hin;
    }
    return score;
}

}  // extern "C"
