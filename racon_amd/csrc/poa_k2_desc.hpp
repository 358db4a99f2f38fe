// poa_k2_desc.hpp -- phase: row descriptors (all 256 threads) + row 0 of Z
// Part of the fast path of the MI355X window-consensus engine: included by poa_kernel2.hpp, in this order, into one
// translation unit (see its header for the design).
#pragma once

namespace rcn {

__device__ __forceinline__ void block4_scan(int* xch, int wv, int lane, int cnt, int wmax, int& off, int& total, int& pmax, int& tmax);
__device__ __forceinline__ void band_row_offsets(const Ctx& c, Win& g, RCN_G const int32_t* rank);
__host__ __device__ constexpr int dp2_ring_rows_band(int np, bool tab);

// ---- phase: row descriptors (all 256 threads) + row 0 of Z ----
// zero_row0: also write row 0 of Z (all zeros) -- what every alignment needs BEFORE its DP.  The rebuild of the descriptors after a sink tie's
// closure sweep (poa_kernel2.hpp: the sweep borrows the descriptor array) comes AFTER the DP and must leave the matrix alone: with move codes in
// it (one byte per cell) the 2 * hstride bytes of the int16 row 0 are the code rows 0 AND 1, and a traceback that reached row 1 with anything but
// a diagonal move read a zeroed code there (one window in 1.34 M of cfg5 whole: found by tools/cfg5_full_check.py, read f8913, round 6).
__device__ __noinline__ void phase_desc2(bool zero_row0 = true) {
    const int t = threadIdx.x;
    const Ctx c = ctx_load<Block4>();
    Win g = ctx_win(c);
    RCN_G const int32_t* rank = c.sub ? g.rank_sub.ptr() : g.rank_full.ptr();
    const Arr<int32_t> nr = c.sub ? g.n2r_x : g.n2r;
    const int cfg_ = dp2_cfg(c.len, c.pad0 != 0);
    const bool tab_ = c.tie_pad[1] != 0;
    const int R = c.band ? dp2_window(c.band) : dp2_window(cfg_ & 255);
    // "medium" rows: like fast rows, but some predecessor is beyond the register window and still in the LDS ring
    const int RM = min(15, (c.band ? dp2_ring_rows_band(c.band, tab_) : dp2_ring_rows(max(cfg_ & 255, 1), max(cfg_ >> 8, 1), (cfg_ >> 8) == 1 && tab_)) - 2);
    // banded alignment (poa_band.hpp): the window offset of every row first; the descriptors below mark the rows where
    // the window moves or a predecessor outside the register window was written under another offset (meta bit 12)
    RCN_G const int32_t* roff = g.pred.ptr();
    if (c.band) band_row_offsets(c, g, rank);
    // Every row is a chain of dependent HBM loads: rank -> node -> its in-edge record (PredRec: the first six tails next to
    // each other, one 32-byte load instead of a load pair per edge) -> the tails' rows.  For a full-graph alignment U rows per
    // thread are walked in lock step, with static register indices only (a runtime index into the descriptors would send
    // them to scratch memory), so that their loads are in flight together: the phase is pure latency on cfg2 -- and a queue
    // at the CU's memory pipeline on cfg4 (eight windows per CU, all in graph phases half of the time), where a wave-wide
    // scattered load costs its 64 requests whether the result is used or not: tails are only loaded where there are tails, a
    // third to sixth one only in waves that have a row with that many.  Subgraph alignments (tails filtered by the mask) go row by row.
    constexpr int U = 2;
    const bool sub = c.sub != 0;
    auto finish = [&](RowDesc d, int r) {
        // "fast" rows: at most 4 predecessors, every one among the R rows right above (the DP keeps those in
        // registers; R = dp2_window(NP)).  meta bit 13 = fast, bits 16-19 / 20-23 / 24-27 / 28-31 = distance
        // (1..R) to predecessor 0 / 1 / 2 / 3, bit 15 = the single predecessor is the row right above.  Sink rows are never fast.
        const int np = (d.meta >> 9) & 7, i = r + 1;
        if (np <= 4 && d.erest < 0 && !(d.meta & 256)) {
            unsigned int bits = 0; bool ok = true, okm = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < np) {
                    const int dist = i - d.p[q];
                    ok = ok && d.p[q] != 0 && dist <= R;
                    okm = okm && d.p[q] != 0 && dist <= RM;
                    bits |= static_cast<unsigned int>(dist & 15) << (16 + 4 * q);
                }
            }
            if (ok) d.meta |= static_cast<int>(bits | (1u << 13) | ((np == 1 && i - d.p[0] == 1) ? (1u << 15) : 0u));   // bit 15 = chain row
            else if (okm) d.meta |= static_cast<int>(bits | (1u << 14));      // bit 14 = medium
        }
        if (c.band) {
            const int my = roff[r], before = r > 0 ? roff[r - 1] : 0;
            bool special = my != before;
            if (!(d.meta & (1 << 13))) {
#pragma unroll
                for (int q = 0; q < kInlinePreds; ++q)
                    if (q < np && d.p[q] > 0 && roff[d.p[q] - 1] != my) special = true;
            }
            if (special) d.meta |= 1 << 12;
        }
        g.desc[r] = d;
    };
    if (sub) {
        // the included ones of the (at most six) inline in-edge tails, in order; a node with more in-edges takes the list walk
        for (int r = t; r < c.V; r += kThreads2) {
            const int v = rank[r];
            const PredRec pr = g.in6[v];
            const int eo = g.out_head[v], code = g.code[v];
            if (pr.erest >= 0) { finish(make_row_desc(g, nr, v, true), r); continue; }
            int inq[kInlinePreds], rowq[kInlinePreds];
#pragma unroll
            for (int q = 0; q < kInlinePreds; ++q) { inq[q] = 0; rowq[q] = 0; }
            if (pr.k > 0) { inq[0] = g.inc[pr.t[0]]; rowq[0] = nr[pr.t[0]]; }
            if (pr.k > 1) { inq[1] = g.inc[pr.t[1]]; rowq[1] = nr[pr.t[1]]; }
            if (__ballot(pr.k > 2)) {
#pragma unroll
                for (int q = 2; q < kInlinePreds; ++q) if (q < pr.k) { inq[q] = g.inc[pr.t[q]]; rowq[q] = nr[pr.t[q]]; }
            }
            const int h0 = eo >= 0 ? g.e_head[eo] : v;
            int e1 = eo >= 0 ? g.e_nout[eo] : -1;
            RowDesc d; d.erest = -1;
#pragma unroll
            for (int q = 0; q < kInlinePreds; ++q) d.p[q] = -1;
            int k = 0;
#pragma unroll
            for (int q = 0; q < kInlinePreds; ++q) {
                const bool take = q < pr.k && inq[q] != 0;
#pragma unroll
                for (int j = 0; j <= q; ++j) d.p[j] = (take && j == k) ? rowq[q] + 1 : d.p[j];       // (static indices: no scratch)
                k += take ? 1 : 0;
            }
            if (k == 0) { d.p[0] = 0; k = 1; }
            bool sink = true;
            if (eo >= 0) {
                if (g.inc[h0]) sink = false;
                else for (; e1 >= 0; e1 = g.e_nout[e1]) if (g.inc[g.e_head[e1]]) { sink = false; break; }
            }
            d.meta = code | (sink ? 256 : 0) | (k << 9);
            finish(d, r);
        }
    } else {
        for (int r0 = t; r0 < c.V; r0 += kThreads2 * U) {
            int v[U], eo[U], code[U];
            PredRec pr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int r = r0 + u * kThreads2; v[u] = r < c.V ? rank[r] : 0; }
#pragma unroll
            for (int u = 0; u < U; ++u) { pr[u] = g.in6[v[u]]; eo[u] = g.out_head[v[u]]; code[u] = g.code[v[u]]; }
            int pq[U][kInlinePreds];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < kInlinePreds; ++q) pq[u][q] = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                pq[u][0] = nr[pr[u].k > 0 ? pr[u].t[0] : v[u]];
                if (pr[u].k > 1) pq[u][1] = nr[pr[u].t[1]];
            }
            bool more = false;
#pragma unroll
            for (int u = 0; u < U; ++u) more = more || pr[u].k > 2;
            if (__ballot(more)) {                       // (a third in-edge is rare: most waves skip these altogether)
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int q = 2; q < kInlinePreds; ++q) if (q < pr[u].k) pq[u][q] = nr[pr[u].t[q]];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * kThreads2;
                if (r >= c.V) continue;
                RowDesc d;
#pragma unroll
                for (int q = 0; q < kInlinePreds; ++q) d.p[q] = q < pr[u].k ? pq[u][q] + 1 : -1;
                int k = pr[u].k;
                if (k == 0) { d.p[0] = 0; k = 1; }
                d.erest = pr[u].erest;                          // more than six: the DP / traceback walk the list
                d.meta = code[u] | (eo[u] < 0 ? 256 : 0) | (k << 9);
                finish(d, r);
            }
        }
    }
    if (zero_row0) {
        RCN_G uint32_t* H = reinterpret_cast<RCN_G uint32_t*>(g.H.ptr());
        for (int j = t; j < (g.hstride >> 1); j += kThreads2) H[j] = 0u;
    }
    Block4::sync();
}

#ifdef RCN_PROF_DP
__device__ unsigned long long g_prof_out[8];
__device__ unsigned long long g_dbg[8];
#endif
#ifdef RCN_PROF_WIN
#ifndef RCN_PROF_DP
__device__ unsigned long long g_dbg[8];          // code traceback: clocks of a tile's load issue / wait / walk, tiles
#endif
__device__ unsigned long long g_wtb2[4096][8];   // ... boxes left because: tile edge, origin, columns used up, climbed 1-2 box heights, fell below the skew line, climbed more; cells walked
__device__ unsigned long long g_wtb[4096][8];    // per work item, code traceback: clocks of tile load issue / wait / walk, tiles, box decode / walk / emit, boxes
__device__ unsigned long long g_whelp[8];           // code waves: lap checks of wave 0, polls spent in them; waits of the code waves, polls spent in them
__device__ unsigned long long g_wtie[8];            // sink ties, all windows: events, clocks, past the rule, closure sweeps, full DFS
__device__ unsigned long long g_wlay[4][128][5];   // work items 0-3 of the first launch, per layer: clocks dp, tie, traceback; {tied, tie level, band}; {V, len}
__device__ unsigned long long g_wclk[4096][8];   // per work item: phase clocks     // per wave: cycles in row bodies, cycles in barriers
#endif

}  // namespace rcn
