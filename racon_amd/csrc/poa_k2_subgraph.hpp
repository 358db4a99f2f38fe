// poa_k2_subgraph.hpp -- phase: Subgraph mask + filtered order (reference src/window.cpp:99-103) as a sweep over ring blocks
// Part of the fast path of the MI355X window-consensus engine: included by poa_kernel2.hpp, in this order, into one
// translation unit (see its header for the design).
#pragma once

namespace rcn {

// ---- phase: Subgraph mask + filtered order (window.cpp:99-103), without the serial DFS ----
// spoa's ExtractSubgraph(end, begin) = nodes with id >= begin that are backward-reachable from `end` over
// in-edges and aligned-node links.  On a ring-contiguous topological order (rank_full) this is one
// descending sweep over RING BLOCKS: a block is taken when any of its members (with id >= begin) is
// pending, then all its members (id >= begin) are taken and all their in-edge tails become pending.
//   pass A (256 threads): per rank, {tail ranks (6 inline + overflow edge), block start / size, id >= begin}
//   pass B (wave 0): 64 ranks per step, block decisions on the scalar unit over 64-bit masks
//   pass C: inc[] per node, compaction into rank_sub / n2r_x
struct SubRec { int32_t tr[6]; int32_t erest; int32_t info; };   // info: bit0 id>=begin, bits 4-7 #inline tails,
                                                                  // bits 8-15 rank - (first rank of its block), bits 16-23 block size
static_assert(sizeof(SubRec) == sizeof(RowDesc), "SubRec lives in the row-descriptor array");

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int k) {
    return (static_cast<unsigned long long>(static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(v >> 32), k))) << 32) |
           static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(v), k));
}

__device__ __forceinline__ void block4_scan(int* xch, int wv, int lane, int cnt, int wmax, int& off, int& total, int& pmax, int& tmax);
constexpr int kSubMaxNodes = kLdsBytes - 64;      // phase_subgraph2: one pending byte per rank in LDS + the words of a prefix count
#ifdef RCN_PROF_WIN
__device__ unsigned long long g_wsub[8];         // Subgraph sweep, all windows: clocks of set-up, pass A, pass B, pass C, calls, chunks of pass B, ranks swept
#endif
// returns false (through ctx->tb_i = 0) when a node has more than six in-edges: the caller then takes the
// serial DFS of poa_kernel.hpp for this layer
__device__ __noinline__ void phase_subgraph2() {
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const Ctx c = ctx_load<Block4>();
    Win g = ctx_win(c);
    const int n = g.n_nodes;
    RCN_G SubRec* rec = reinterpret_cast<RCN_G SubRec*>(g.desc.ptr());
    uint8_t* pend = reinterpret_cast<uint8_t*>(Block4::work());          // [n] pending / finally: included, by rank
    Ctx* o = Block4::ctx();
    int top;
    {
        int r = g.n2r[c.end];
        const int na = g.al_cnt[c.end];
        for (int a = 0; a < na; ++a) r = max(r, g.n2r[g.al_nodes[c.end * g.ring + a]]);
        top = bcast0(r);
    }
    if (t == 0) o->tb_i = 1;
#ifdef RCN_PROF_WIN
    const long long ts0__ = clock64();
    long long nch__ = 0;
#endif
    Block4::sync();
#ifdef RCN_PROF_WIN
    const long long ts1__ = clock64();
#endif
    // ---- pass A ----
    for (int r = t; r < n; r += kThreads2) pend[r] = 0;
    bool wide = false;
    // (three dependent loads per rank: node, its in-edge record and ring members, their ranks -- the in-list itself is
    //  only walked by pass B, for the rare node with more than six in-edges)
    for (int r = t; r <= top; r += kThreads2) {
        const int v = g.rank_full[r];
        const PredRec pr = g.in6[v];
        const int na = g.al_cnt[v];
        SubRec e; e.erest = pr.erest;
        // (loads only where there is something to load: with eight windows per CU these phases queue at the CU's memory
        //  pipeline, a wave-wide scattered load is 64 requests whether its result is used or not)
#pragma unroll
        for (int q = 0; q < 6; ++q) e.tr[q] = -1;
        if (pr.k > 0) e.tr[0] = g.n2r[pr.t[0]];
        if (pr.k > 1) e.tr[1] = g.n2r[pr.t[1]];
        if (__ballot(pr.k > 2)) {                   // (a third in-edge is rare: most waves skip these altogether)
#pragma unroll
            for (int q = 2; q < 6; ++q) if (q < pr.k) e.tr[q] = g.n2r[pr.t[q]];
        }
        int rb = r;
        for (int a = 0; a < na; ++a) rb = min(rb, g.n2r[g.al_nodes[v * g.ring + a]]);
        e.info = (v >= c.begin ? 1 : 0) | (pr.k << 4) | ((r - rb) << 8) | ((na + 1) << 16);
        rec[r] = e;
    }
    if (wide) o->tb_i = 0;
    Block4::sync();
    if (bcast0(o->tb_i) == 0) return;
    if (t == 0) pend[g.n2r[c.end]] = 1;
    Block4::sync();
#ifdef RCN_PROF_WIN
    const long long ts2__ = clock64();
#endif
    // ---- pass B ----
    if (wv == 0) {
        int hi = top, minpend = g.n2r[c.end];
        while (hi >= 0 && minpend <= hi) {
            const int base = hi - 63;                                      // lane l <-> rank base + l
            const int r = base + lane;
            SubRec e; e.erest = -1; e.info = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) e.tr[q] = -1;
            if (r >= 0) e = rec[r];
            const int off = (e.info >> 8) & 255, bsz = (e.info >> 16) & 255;
            // lanes whose ring block starts below the chunk are left to the next chunk
            const bool mine = r >= 0 && r - off >= base && r - off >= 0;
            const unsigned long long minemask = __ballot(mine);
            const int lo_lane = __builtin_ctzll(minemask);                 // lowest lane processed here (a block start)
            const bool idok = mine && (e.info & 1);
            unsigned long long pendmask = __ballot(mine && pend[r >= 0 ? r : 0] != 0);
            // in-edge tails inside the processed part of the chunk, as lane bits
            unsigned long long own_t = 0ull;
#pragma unroll
            for (int q = 0; q < 6; ++q) { const int tl = e.tr[q] - base; if (e.tr[q] >= 0 && tl >= lo_lane) own_t |= 1ull << tl; }
            for (int ed = e.erest; ed >= 0; ed = g.e_nin[ed]) {              // more than six in-edges (rare)
                const int tl = g.n2r[g.e_tail[ed]] - base;
                if (tl >= lo_lane) own_t |= 1ull << tl;
            }
            if (!idok) own_t = 0ull;
            const unsigned long long own_b = idok ? (1ull << lane) : 0ull;
            // per block, at its first lane: members with id >= begin, union of their tail masks (blocks are short: the
            // loop goes as far as the longest block of the chunk)
            unsigned long long bmask = own_b, btmask = own_t;
            const int maxd = __ballot(mine && bsz >= 5) ? 8 : __ballot(mine && bsz >= 3) ? 4 : __ballot(mine && bsz >= 2) ? 2 : 1;
            for (int d = 1; d < maxd; ++d) {
                const unsigned long long mb = __shfl_down(own_b, d), mt = __shfl_down(own_t, d);
                if (d < bsz && lane + d < 64) { bmask |= mb; btmask |= mt; }
            }
            // The sweep proper, highest rank first: a block is included when one of its members is pending, and then its
            // members' tails are pending.  Only pending ranks are looked at (a block nobody points to is never visited),
            // and a RUN of chain links -- one-rank blocks whose only tail inside the chunk is the rank right below --
            // is taken in one step with mask arithmetic on the scalar unit: most of a graph is such runs, and a step that
            // has to fetch a lane's masks (v_readlane into the scalar unit and back) costs ~100 clocks.
            const unsigned long long linkmask = __ballot(idok && bsz == 1 && lane > lo_lane && own_t == (1ull << ((lane - 1) & 63)));
            const int bstart = lane - off;
            unsigned long long incmask = 0ull, done = lo_lane > 0 ? ((1ull << lo_lane) - 1ull) : 0ull;   // (lanes below the processed part)
            for (;;) {
                const unsigned long long cand = pendmask & ~done;
                if (!cand) break;
                const int p = 63 - __builtin_clzll(cand);
                const unsigned long long upto = p == 63 ? ~0ull : ((2ull << p) - 1ull);
                if ((linkmask >> p) & 1ull) {
                    const int z = 63 - __builtin_clzll(~linkmask & upto);          // first rank below p that is not a link (>= lo_lane)
                    const unsigned long long run = upto & ~((2ull << z) - 1ull);   // ranks z + 1 .. p
                    incmask |= run; pendmask |= run >> 1; done |= run;
                } else {
                    const int k = __builtin_amdgcn_readlane(bstart, p);
                    const unsigned long long bm = readlane64(bmask, k);
                    if (bm & pendmask) { incmask |= bm; pendmask |= readlane64(btmask, k); }
                    done |= bm | (1ull << p);
                }
            }
            const bool inc = (incmask >> lane) & 1ull;
            // tails below the processed part become pending
            int lowest = 0x7fffffff;
            if (inc) {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int tr = e.tr[q];
                    if (tr >= 0 && tr - base < lo_lane) { pend[tr] = 1; lowest = min(lowest, tr); }
                }
                for (int ed = e.erest; ed >= 0; ed = g.e_nin[ed]) {
                    const int tr = g.n2r[g.e_tail[ed]];
                    if (tr - base < lo_lane) { pend[tr] = 1; lowest = min(lowest, tr); }
                }
            }
            if (mine) pend[r] = inc ? 1 : 0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) lowest = min(lowest, __shfl_xor(lowest, d));
            Wave0Of4::sync();
            const int lo_eff = base + lo_lane;
            if (minpend >= lo_eff) minpend = 0x7fffffff;                   // it has just been processed
            minpend = min(minpend, lowest);
            hi = lo_eff - 1;
#ifdef RCN_PROF_WIN
            ++nch__;
#endif
        }
    }
    Block4::sync();
#ifdef RCN_PROF_WIN
    const long long ts3__ = clock64();
#endif
    // ---- pass C ----
    if (c.tb_j == 1) {
        // closure query (phase_sink_tie_*): only DFS marks, nothing of the current alignment is touched
        for (int r = t; r < n; r += kThreads2) g.mark[g.rank_full[r]] = pend[r] ? 2 : 0;
        Block4::sync();
        return;
    }
    {
        // inclusion flags by node and the subgraph's own order (rank_full filtered): 256 ranks per step, the positions are a
        // prefix count across the four waves
        int* xch = Block4::work() + kSubMaxNodes / 4;
        int nv = 0;
        for (int b0 = 0; b0 < n; b0 += kThreads2) {
            const int r = b0 + t;
            const int v = r < n ? g.rank_full[r] : 0;
            const bool in = r < n && pend[r] != 0;
            const unsigned long long mk = __ballot(in);
            int off, total, pmax, tmax;
            block4_scan(xch, wv, lane, __popcll(mk), 0, off, total, pmax, tmax);
            if (r < n) g.inc[v] = in ? 1 : 0;
            if (in) {
                const int pos = nv + off + __popcll(mk & ((1ull << lane) - 1ull));
                g.rank_sub[pos] = v; g.n2r_x[v] = pos;
            }
            nv += total;
        }
        if (t == 0) o->V = nv;
    }
    Block4::sync();
#ifdef RCN_PROF_WIN
    if (t == 0 && (c.wi & 15) == 0) {            // (every sixteenth window: the atomics must not become the measurement)
        const long long ts4__ = clock64();
        atomicAdd(&g_wsub[0], (unsigned long long)(ts1__ - ts0__)); atomicAdd(&g_wsub[1], (unsigned long long)(ts2__ - ts1__));
        atomicAdd(&g_wsub[2], (unsigned long long)(ts3__ - ts2__)); atomicAdd(&g_wsub[3], (unsigned long long)(ts4__ - ts3__));
        atomicAdd(&g_wsub[4], 1ull); atomicAdd(&g_wsub[5], (unsigned long long)nch__); atomicAdd(&g_wsub[6], (unsigned long long)(top + 1));
        atomicAdd(&g_wsub[7], (unsigned long long)n);
    }
#endif
}

}  // namespace rcn
