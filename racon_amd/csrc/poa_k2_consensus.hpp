// poa_k2_consensus.hpp -- phase: consensus (reference src/window.cpp:122-146): heaviest bundle, branch completion, coverage trim
// Part of the fast path of the MI355X window-consensus engine: included by poa_kernel2.hpp, in this order, into one
// translation unit (see its header for the design).
#pragma once

namespace rcn {

// ---- phase: consensus (window.cpp:122-146) ----
// Heaviest bundle without spoa's exact DFS order in the common case.  Scores and predecessor choices do
// not depend on WHICH valid topological order is used; the exact order only matters (a) to pick the first
// of several nodes that tie for the maximal score and (b) inside BranchCompletion (max node with
// out-edges).  Both are rare (~0.5% of windows): they take the exact serial path of poa_kernel.hpp.
//   pass A (256 threads): per rank r of rank_full, the winning in-edge by weight -> record {tail rank of
//           the best edge, weight, up to two more tails that tie on weight (then the tail SCORE decides,
//           later edge wins: the predicate of TraverseHeaviestBundle is a lexicographic max over
//           (weight, score[tail], edge order))}, stored in the row-descriptor array.
//   pass B (wave 0): 64 ranks at a time; scores of earlier chunks come from LDS, dependencies inside the
//           chunk are resolved with one v_readlane per rank.
constexpr int kCons2MaxNodes = kLdsBytes / 6;     // int32 score + uint16 predecessor rank per node in LDS
struct ConsRec { int32_t trA, w, trB, trC; };     // trB/trC: -1 none; trC == -2: more than three edges tie

// exact = 0: over rank_full (any valid order); exact = 1: over rank_x, spoa's own order (phase_toposort4)
__device__ __noinline__ void phase_cons2_edges(int exact) {
    exact = uint_(exact);
    const int t = threadIdx.x;
    const Ctx c = ctx_load<Block4>();
    Win g = ctx_win(c);
    RCN_G ConsRec* rec = reinterpret_cast<RCN_G ConsRec*>(g.desc.ptr());
    RCN_G const int32_t* rank = exact ? g.rank_x.ptr() : g.rank_full.ptr();
    RCN_G const int32_t* n2r = exact ? g.n2r_x.ptr() : g.n2r.ptr();
    const int n = g.n_nodes;
    for (int r = t; r < n; r += kThreads2) {
        const int v = rank[r];
        ConsRec o; o.trA = -1; o.w = 0; o.trB = -1; o.trC = -1;
        long long wmax = -1; int ntie = 0;
        for (int e = g.in_head[v]; e >= 0; e = g.e_nin[e]) {
            const long long w = g.e_w[e];
            const int tr = n2r[g.e_tail[e]];
            if (w > wmax) { wmax = w; ntie = 1; o.trA = tr; o.w = static_cast<int32_t>(w); o.trB = -1; o.trC = -1; }
            else if (w == wmax) { ++ntie; if (ntie == 2) o.trB = tr; else if (ntie == 3) o.trC = tr; else o.trC = -2; }
        }
        rec[r] = o;
    }
    Block4::sync();
}

// returns (through ctx->tb_n) the consensus length, 0 = take the exact path; path ranks in LDS (reversed)
__device__ __noinline__ void phase_cons2_bundle(int exact) {
    exact = uint_(exact);
    const int lane = threadIdx.x;
    const Ctx c = ctx_load<Wave0Of4>();
    Win g = ctx_win(c);
    RCN_G const ConsRec* rec = reinterpret_cast<RCN_G const ConsRec*>(g.desc.ptr());
    RCN_G const int32_t* rank = exact ? g.rank_x.ptr() : g.rank_full.ptr();
    RCN_G const int32_t* n2r = exact ? g.n2r_x.ptr() : g.n2r.ptr();
    const int n = g.n_nodes;
    int* sc = Wave0Of4::work();                                              // [n]
    uint16_t* pr = reinterpret_cast<uint16_t*>(Wave0Of4::work() + n);       // [n] rank of the chosen predecessor, 0xFFFF none
    int gmax = static_cast<int>(0x80000000u), gmax_rank = -1, gtie = 0;
#pragma unroll 1
    for (int base = 0; base < n; base += 64) {
        const int r = base + lane;
        ConsRec e; e.trA = -1; e.w = 0; e.trB = -1; e.trC = -1;
        if (r < n) e = rec[r];
        int trA = e.trA;
        const int tl = trA >= base ? trA - base : -1;
        int fin = -1;
        if (trA >= 0 && trA < base) fin = e.w + sc[trA];
        const unsigned long long amb = __ballot(e.trB >= 0 || e.trC == -2);
        const int cnt = min(64, n - base);
#pragma unroll 1
        for (int k = 0; k < cnt; ++k) {
            if ((amb >> k) & 1ull) {
                // several in-edges tie on weight: the tail with the larger score wins, later edge on equal scores
                const int a = __builtin_amdgcn_readlane(e.trA, k), b = __builtin_amdgcn_readlane(e.trB, k), cc = __builtin_amdgcn_readlane(e.trC, k);
                const int wk = __builtin_amdgcn_readlane(e.w, k);
                int bt = -1, bs = 0; bool have = false;
                auto consider = [&](int tr) {
                    const int s = tr >= base ? __builtin_amdgcn_readlane(fin, tr - base) : bcast0(sc[tr]);
                    if (!have || s >= bs) { bs = s; bt = tr; have = true; }
                };
                if (cc == -2) {
                    // more than three candidates: walk the node's in-edge list again (edge order)
                    const int v = rank[base + k];
                    for (int ed = g.in_head[v]; ed >= 0; ed = g.e_nin[ed]) {
                        if (static_cast<int32_t>(g.e_w[ed]) == wk) consider(bcast0(n2r[g.e_tail[ed]]));
                    }
                } else {
                    consider(a); consider(b); if (cc >= 0) consider(cc);
                }
                if (lane == k) { fin = wk + bs; trA = bt; }
            }
            const int sk = __builtin_amdgcn_readlane(fin, k);
            if (tl == k && !((amb >> lane) & 1ull)) fin = e.w + sk;
        }
        if (r < n) { sc[r] = fin; pr[r] = static_cast<uint16_t>(trA < 0 ? 0xFFFF : trA); }
        // running maximum: first strictly greater in rank order; any equality makes the order matter
        const int fm = r < n ? fin : static_cast<int>(0x80000000u);
        int cm = fm;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cm = max(cm, __shfl_xor(cm, d));
        const unsigned long long at = __ballot(fm == cm);
        if (cm > gmax) { gmax = cm; gmax_rank = base + __builtin_ctzll(at); gtie = __popcll(at) > 1; }
        else if (cm == gmax) gtie = 1;
        Wave0Of4::sync();
    }
    Ctx* o = Wave0Of4::ctx();
    int k = 0;
    int mxnode = rank[gmax_rank];
    if (exact) {
        // over spoa's own order the first maximum IS spoa's choice; BranchCompletion (spoa graph.cpp, restated in
        // graph_consensus of poa_core.hpp) runs on the LDS scores, serially: it only touches the ranks behind the
        // maximum, normally the last few of the graph
        int mx = gmax_rank;
        if (lane == 0) {
            while (g.out_head[rank[mx]] >= 0) {
                const int start = rank[mx];
                for (int e = g.out_head[start]; e >= 0; e = g.e_nout[e]) {
                    for (int f = g.in_head[g.e_head[e]]; f >= 0; f = g.e_nin[f]) {
                        const int tl = g.e_tail[f];
                        if (tl != start) sc[n2r[tl]] = -1;
                    }
                }
                int m2 = -1, m2s = 0;
                for (int r = mx + 1; r < n; ++r) {
                    const int it = rank[r];
                    int sv = -1, p = -1, ps = 0;
                    for (int f = g.in_head[it]; f >= 0; f = g.e_nin[f]) {
                        const int tr = n2r[g.e_tail[f]];
                        const int ts = sc[tr];
                        if (ts == -1) continue;
                        const int w = static_cast<int32_t>(g.e_w[f]);
                        if (sv < w || (sv == w && ps <= ts)) { sv = w; p = tr; ps = ts; }
                    }
                    if (p >= 0) sv += ps;
                    sc[r] = sv; pr[r] = static_cast<uint16_t>(p < 0 ? 0xFFFF : p);
                    if (m2 < 0 || m2s < sv) { m2 = r; m2s = sv; }
                }
                mx = m2;
            }
        }
        gmax_rank = bcast0(mx); gtie = 0;
        mxnode = rank[gmax_rank];
    }
    if (!gtie && g.out_head[mxnode] < 0) {
        // backtrack through the LDS predecessor ranks; the rank list overwrites the scores
        int cur = gmax_rank;
        for (;;) {
            const int nxt = bcast0(static_cast<int>(pr[cur]));
            if (lane == 0) sc[k] = cur;
            ++k;
            if (nxt == 0xFFFF) break;
            cur = nxt;
        }
    }
    if (lane == 0) { o->tb_n = k; o->tb_j = exact; }
    Wave0Of4::sync();
}

__device__ __noinline__ void phase_cons2_finish(uint8_t* out_in, uint64_t out_cap, uint32_t* out_len_in, uint8_t* out_flags_in, int ns, int tgs) {
    RCN_G uint8_t* out = uptr(out_in); RCN_G uint32_t* out_len = uptr(out_len_in); RCN_G uint8_t* out_flags = uptr(out_flags_in);
    out_cap = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(out_cap >> 32))) << 32) | __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(out_cap));
    ns = uint_(ns); tgs = uint_(tgs);
    const int lane = threadIdx.x;
    const Ctx c = ctx_load<Wave0Of4>();
    Win g = ctx_win(c);
    const int k = c.tb_n;
    const int* plist = Wave0Of4::work();          // reversed consensus path, as ranks of the order the bundle ran over
    RCN_G int32_t* cn = g.path_node.ptr();
    RCN_G const int32_t* rank = c.tb_j ? g.rank_x.ptr() : g.rank_full.ptr();
    for (int i = lane; i < k; i += 64) cn[i] = rank[plist[k - 1 - i]];
    Wave0Of4::sync();
    int bgn = 0, end = k - 1, flags = kFlagPolished;
    if (tgs && c.trim) {
        const uint32_t avg = static_cast<uint32_t>(ns - 1) / 2;
        // first / last consensus position whose coverage reaches the threshold (window.cpp:128-137)
        bgn = k;
        for (int b0 = 0; b0 < k && bgn == k; b0 += 64) {
            const int i = b0 + lane;
            const bool ok = i < k && consensus_coverage(g, cn[i]) >= avg;
            const unsigned long long mk = __ballot(ok);
            if (mk) bgn = b0 + __builtin_ctzll(mk);
        }
        end = -1;
        for (int b0 = 0; b0 < k && end == -1; b0 += 64) {
            const int i = k - 1 - (b0 + lane);
            const bool ok = i >= 0 && consensus_coverage(g, cn[i]) >= avg;
            const unsigned long long mk = __ballot(ok);
            if (mk) end = k - 1 - (b0 + __builtin_ctzll(mk));
        }
        if (bgn >= end) { bgn = 0; end = k - 1; flags |= kFlagChimeric; }
    }
    const int clen = end - bgn + 1;
    if (static_cast<uint64_t>(clen) > out_cap) { if (lane == 0) { *out_len = 0; *out_flags = kFlagOverflow; } return; }
    for (int i = lane; i < clen; i += 64) out[i] = g.code[cn[bgn + i]];
    if (lane == 0) { *out_len = clen; *out_flags = static_cast<uint8_t>(flags); }
    Wave0Of4::sync();
}

}  // namespace rcn
