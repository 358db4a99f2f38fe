// poa_small.hpp -- the small-window instance of the MI355X window-consensus engine (gfx950): ONE WAVE per window, the whole
// partial-order graph in LDS.
//
// Same per-window algorithm as poa_kernel2.hpp (racon's Window::generate_consensus, reference src/window.cpp:65-149; nearly
// every layer of a short-read window takes the Subgraph branch, window.cpp:99-107), for the shape poa_window_kernel2 is
// worst at: -w 200 windows of 150-bp reads at 60x (BASELINE configs[3]) are ~140 alignments of ~86 bases against a graph of
// ~220 nodes.  There the four-wave kernel spends a window's time on dependent HBM round trips -- eight phases per layer,
// each a handful of hops through node / edge / order arrays that together are a few KB -- and three of its four waves idle
// through the DP (profiles/r03/f_winprof_cfg4.txt, n_cfg4_residency.txt: bound by memory requests per CU).  Here:
//
//   * the graph lives in LDS for the life of the window, 16-bit ids: node symbol, aligned ring (at most four others: A, C,
//     G, T and N), the incrementally maintained ring-contiguous topological order and its inverse, and per node the TAILS of its
//     in-edges in creation order (at most kSmIn; spoa's in-edge order is all the traceback's tie-break needs -- out-edge
//     lists are not kept at all: "does edge tail -> head exist" is a look at head's in-record, "is it a sink" a scatter of
//     marks from the in-records).  31 bytes per node + 14 per node of per-layer work area: a 200-base window takes 13 KB,
//     eleven windows share a CU;
//   * edge weights and node coverage -- only read by the consensus at the very end -- stay in HBM and are only ever
//     touched by fire-and-forget atomics (slot [node][in-edge number]);
//   * the DP is the Z-domain packed-int16 row of poa_k2_dp.hpp (two cells per VGPR, DPP prefix max) over at most 256
//     columns, predecessor rows from a register window of sixteen rows (no LDS ring, no score matrix at all): it leaves
//     one MOVE CODE byte per cell (poa_band.hpp: diagonal / vertical reproduces the cell, first predecessor in in-edge
//     order that attains the maximum) and the traceback walks the codes, 8 x 8 cells per gather, the next box's gather in
//     flight while the current one is walked;
//   * one wave: no work-group barrier anywhere, a phase boundary is an LDS fence.
//
// Anything outside this shape -- a node with a ninth in-edge, a predecessor more than sixteen rows back, a ring beyond five
// symbols, a sink tie the id rule does not decide, a graph that outgrows the LDS -- flags the window (kFlagOverflow) and the
// engine re-runs it with poa_window_kernel2: results are bit-identical either way, the flag only costs time.
//
// Integer max-plus DP on an irregular DAG: no MFMA.
#pragma once
#include "poa_kernel2.hpp"

namespace rcn {

constexpr int kSmIn = 8;            // in-edge tails kept per node (move codes name a predecessor with three bits)
constexpr int kSmRing = 4;          // aligned-ring members of a node besides itself (A, C, G, T and one more symbol: N)
constexpr int kSmLen = 255;         // longest layer (256 columns: two packed VGPRs per lane)
constexpr int kSmWin = 16;          // rows of the register window = farthest predecessor row
constexpr int kSmMinCap = 256, kSmMaxCap = 1024;     // node capacities the LDS layout is made for
constexpr int kSmPosBytes = 512;    // per-position area: int16 x 256
constexpr int kSmRow0 = 1 << 12;    // row descriptor: the only predecessor is the virtual start row
constexpr int kSmWide = 1 << 13;    // row descriptor: five to eight predecessors, their distances in the side table
constexpr int kSmWideRows = 16;     // rows of that kind per alignment (more: the window leaves the kernel)
constexpr int kSmChain = 1 << 14;   // row descriptor: one predecessor, the row right above (the DP reads it from registers)

// LDS layout (byte offsets from the work-group's dynamic LDS) for a graph of up to `ncap` nodes.
//   persistent for the window: code, alcnt, ink [ncap] u8; rank, n2r [ncap] u16; intail [ncap][kSmIn = 8] u16; ring [ncap][kSmRing = 4] u16
//   per layer:   inc, mark [ncap] u8; rsub, nsub [ncap] u16; desc [ncap] u32; post [256] i16; misc: 32 words (rows of the tied
//                sinks, wide-row counter, distances of the wide rows); lseq [2][256] u8: the bases of this layer and
//                (prefetched) of the next one; lqual [256]: this layer's qualities
//   (phases that run after the traceback re-use inc .. desc: see sm_add / sm_consensus)
struct SmLayout { uint32_t code, alcnt, ink, rank, n2r, intail, ring, inc, mark, rsub, nsub, desc, post, misc, lseq, lqual, end; };
__host__ __device__ inline SmLayout small_layout(int ncap) {
    const uint32_t n = static_cast<uint32_t>((ncap + 7) & ~7);
    SmLayout l;
    l.code = 0; l.alcnt = l.code + n; l.ink = l.alcnt + n; l.rank = l.ink + n; l.n2r = l.rank + 2 * n;
    l.intail = l.n2r + 2 * n; l.ring = l.intail + 2 * kSmIn * n;
    l.inc = l.ring + 2 * kSmRing * n; l.mark = l.inc + n; l.rsub = l.mark + n; l.nsub = l.rsub + 2 * n; l.desc = l.nsub + 2 * n;
    l.post = l.desc + 4 * n; l.misc = l.post + kSmPosBytes; l.lseq = l.misc + 128; l.lqual = l.lseq + 512; l.end = l.lqual + 256;
    return l;
}
// HBM slot of a resident window: weights [ncap][8] u32, coverage [ncap] u32, then the move codes (256 bytes per DP row)
__host__ __device__ inline uint64_t small_slot_codes(int ncap) { return (static_cast<uint64_t>(ncap) * (4 * kSmIn + 4) + 255) & ~uint64_t(255); }
__host__ __device__ inline uint64_t small_slot_bytes(int ncap) { return small_slot_codes(ncap) + 256ull * (static_cast<uint64_t>(ncap) + 2); }

#ifndef RCN_SMALL_TU
__global__ void poa_window_kernel_small(KParams P);        // defined in engine_small.hip
#else
template <class T> using sm_lds = __attribute__((address_space(3))) T*;
template <class T> __device__ __forceinline__ sm_lds<T> sm_at(uint32_t byte_addr) { return reinterpret_cast<sm_lds<T>>(byte_addr); }
// Phase boundary of the one-wave work-group: this wave's LDS traffic is done (LDS operations of a wave execute in order; the
// clobber keeps the compiler from moving LDS accesses across).  NOT a fence of the memory model: a workgroup- or agent-scope
// fence also waits for the wave's outstanding global stores and atomics (vmcnt), and at agent scope writes the L2 back
// (buffer_wbl2) -- per layer, that was most of the 64 % of their time the waves spent parked (profiles/r04/c_sq_cfg4_summary.txt).
__device__ __forceinline__ void sm_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// this wave's global stores / atomics have been performed at the L2 (all a later access of the SAME wave's CU needs: the
// L2 of its XCD is where its atomics execute and where its sc1 loads read)
__device__ __forceinline__ void sm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Why a window left the kernel (statistics, stats[25 + why]): 1 graph capacity, 2 ninth in-edge (kSmIn = 8) or a seventeenth wide row, 3 far predecessor, 4 ring,
// 5 int16 range, 6 layer too long, 7 sink tie, 8 consensus scratch, 9 internal inconsistency (a bug: tests assert it is zero)
enum : int { kSmCap = 1, kSmInDeg = 2, kSmFar = 3, kSmRingFull = 4, kSmRange = 5, kSmLong = 6, kSmTie = 7, kSmStack = 8, kSmBug = 9 };

struct SmPtrs {
    sm_lds<uint8_t> code, alcnt, ink, inc, mark;
    sm_lds<uint16_t> rank, n2r, intail, ring, rsub, nsub;
    sm_lds<uint32_t> desc;
    sm_lds<int16_t> post;
    sm_lds<int32_t> misc;       // [0..7] rows of the tied sinks, [8] wide rows of this alignment, [16..31] their distance words
    sm_lds<uint8_t> lseq, lqual;   // [2][256], [256]
};
__device__ __forceinline__ SmPtrs sm_ptrs(uint32_t base, const SmLayout& l) {
    SmPtrs p;
    p.code = sm_at<uint8_t>(base + l.code); p.alcnt = sm_at<uint8_t>(base + l.alcnt); p.ink = sm_at<uint8_t>(base + l.ink);
    p.inc = sm_at<uint8_t>(base + l.inc); p.mark = sm_at<uint8_t>(base + l.mark);
    p.rank = sm_at<uint16_t>(base + l.rank); p.n2r = sm_at<uint16_t>(base + l.n2r); p.intail = sm_at<uint16_t>(base + l.intail);
    p.ring = sm_at<uint16_t>(base + l.ring); p.rsub = sm_at<uint16_t>(base + l.rsub); p.nsub = sm_at<uint16_t>(base + l.nsub);
    p.desc = sm_at<uint32_t>(base + l.desc); p.post = sm_at<int16_t>(base + l.post); p.misc = sm_at<int32_t>(base + l.misc);
    p.lseq = sm_at<uint8_t>(base + l.lseq); p.lqual = sm_at<uint8_t>(base + l.lqual);
    return p;
}

// ---- Subgraph (window.cpp:99-103): the sweep of poa_k2_subgraph.hpp, one wave, records made on the fly from LDS ----
// spoa's ExtractSubgraph(end, begin) = nodes with id >= begin backward-reachable from `end` over in-edges and aligned
// links = one descending sweep over the RING BLOCKS of the ring-contiguous topological order (see phase_subgraph2).
// Fills inc[] (by node), rsub / nsub (the subgraph's own order and its inverse); returns its size.
__device__ __forceinline__ int sm_subgraph(const SmPtrs& g, int n, int begin, int end, int lane, unsigned int& n_chunks, unsigned int& n_intervals) {
    sm_lds<uint8_t> pend = reinterpret_cast<sm_lds<uint8_t>>(g.desc);             // [n] pending / included, by rank
    int top, lo, end_lo;
    {
        int r = g.n2r[end], rl = r;
        const int na = g.alcnt[end];
        for (int a = 0; a < na; ++a) { const int q = g.n2r[g.ring[end * kSmRing + a]]; r = max(r, q); rl = min(rl, q); }
        top = bcast0(r); end_lo = bcast0(rl);
        int b = g.n2r[begin];
        const int nb = g.alcnt[begin];
        for (int a = 0; a < nb; ++a) b = min(b, static_cast<int>(g.n2r[g.ring[begin * kSmRing + a]]));
        lo = bcast0(b);
    }
    // ---- the common case on clean reads: the mask IS the rank interval [lo, top] ----
    // (first rank of begin's ring block .. last rank of end's): exactly so when every ring block of the interval below end's
    // own has a member with a successor inside the interval -- then, top down, every block is an ancestor of `end` -- and no
    // node of the interval has an in-edge from a NON-backbone node ranked below it (such a tail has id >= begin and would be
    // included; backbone nodes before `begin` rank below the interval and are cut by the id rule).  Two passes of LDS traffic
    // instead of the sweep; tests/emul/emul_main.cpp (clean_interval) checks the rule against the DFS on every partial layer of
    // the CPU test sets: 89 % of the layers of BASELINE configs[3] take it, ~half of ONT-like ones.
    if (lo <= top) {
        for (int r = lo + lane; r <= top; r += 64) g.mark[r] = 0;
        sm_fence();
        int bad = 0;
        for (int r = lo + lane; r <= top; r += 64) {
            const int v = g.rank[r];
            const int k = g.ink[v];
#pragma unroll
            for (int q = 0; q < kSmIn; ++q) {
                if (q < k) {
                    const int t = g.intail[v * kSmIn + q];
                    const int tr = g.n2r[t];
                    if (tr >= lo) g.mark[tr] = 1;
                    else if (t >= begin) bad = 1;
                    else g.inc[t] = 0;                              // (a backbone node before `begin`: the descriptors ask)
                }
            }
        }
        sm_fence();
        if (!__ballot(bad != 0)) {
            for (int r = lo + lane; r <= top; r += 64) {
                const int v = g.rank[r];
                bool ok = r >= end_lo || g.mark[r] != 0;
                if (!ok) {
                    const int na = g.alcnt[v];
                    for (int a = 0; a < na; ++a) ok = ok || g.mark[g.n2r[g.ring[v * kSmRing + a]]] != 0;
                }
                if (!ok) bad = 1;
                g.inc[v] = 1; g.rsub[r - lo] = static_cast<uint16_t>(v); g.nsub[v] = static_cast<uint16_t>(r - lo);
            }
            sm_fence();
            if (!__ballot(bad != 0)) { ++n_intervals; return top - lo + 1; }
        }
    }
    for (int r = lane; r <= top; r += 64) pend[r] = 0;          // (nothing above `top` is looked at: tails and ring mates rank below their node's block end)
    sm_fence();
    if (lane == 0) pend[g.n2r[end]] = 1;
    sm_fence();
    int hi = top, minpend = bcast0(static_cast<int>(g.n2r[end]));
    while (hi >= 0 && minpend <= hi) {
        ++n_chunks;
        const int base = hi - 63;
        const int r = base + lane;
        const bool have = r >= 0;
        const int v = have ? g.rank[r] : 0;
        const int k = have ? g.ink[v] : 0;
        int tr[kSmIn];
#pragma unroll
        for (int q = 0; q < kSmIn; ++q) tr[q] = -1;
        // (a third in-edge is rare, a fifth rarer: whole chunks skip those loads)
        auto tails = [&](int q0, int q1) {
#pragma unroll
            for (int q = 0; q < kSmIn; ++q) if (q >= q0 && q < q1) { const int tn = g.intail[v * kSmIn + q]; tr[q] = q < k ? static_cast<int>(g.n2r[tn < n ? tn : 0]) : -1; }
        };
        tails(0, 2);
        if (__ballot(k > 2)) tails(2, 4);
        if (__ballot(k > 4)) tails(4, 8);
        const int na = have ? g.alcnt[v] : 0;
        int rb = r;
        if (__ballot(na > 0)) {                                      // (most chunks of a clean graph have no aligned ring at all)
#pragma unroll
            for (int a = 0; a < kSmRing; ++a) { const int u = g.ring[v * kSmRing + a]; if (a < na) rb = min(rb, static_cast<int>(g.n2r[u < n ? u : 0])); }
        }
        const int off = r - rb, bsz = na + 1;
        const bool mine = have && r - off >= base && r - off >= 0;       // lanes whose ring block starts below the chunk wait for the next chunk
        const unsigned long long minemask = __ballot(mine);
        const int lo_lane = __builtin_ctzll(minemask);
        const bool idok = mine && v >= begin;
        unsigned long long pendmask = __ballot(mine && pend[have ? r : 0] != 0);
        unsigned long long own_t = 0ull;
#pragma unroll
        for (int q = 0; q < kSmIn; ++q) { const int tl = tr[q] - base; if (tr[q] >= 0 && tl >= lo_lane) own_t |= 1ull << tl; }
        if (!idok) own_t = 0ull;
        const unsigned long long own_b = idok ? (1ull << lane) : 0ull;
        unsigned long long bmask = own_b, btmask = own_t;
        const int maxd = __ballot(mine && bsz >= 5) ? 8 : __ballot(mine && bsz >= 3) ? 4 : __ballot(mine && bsz >= 2) ? 2 : 1;      // (ring blocks of up to kSmRing + 1 = 5)
        for (int d = 1; d < maxd; ++d) {
            const unsigned long long mb = __shfl_down(own_b, d), mt = __shfl_down(own_t, d);
            if (d < bsz && lane + d < 64) { bmask |= mb; btmask |= mt; }
        }
        const unsigned long long linkmask = __ballot(idok && bsz == 1 && lane > lo_lane && own_t == (1ull << ((lane - 1) & 63)));
        const int bstart = lane - off;
        unsigned long long incmask = 0ull, done = lo_lane > 0 ? ((1ull << lo_lane) - 1ull) : 0ull;
        for (;;) {
            const unsigned long long cand = pendmask & ~done;
            if (!cand) break;
            const int p = 63 - __builtin_clzll(cand);
            const unsigned long long upto = p == 63 ? ~0ull : ((2ull << p) - 1ull);
            if ((linkmask >> p) & 1ull) {
                const int z = 63 - __builtin_clzll(~linkmask & upto);
                const unsigned long long run = upto & ~((2ull << z) - 1ull);
                incmask |= run; pendmask |= run >> 1; done |= run;
            } else {
                const int kk = __builtin_amdgcn_readlane(bstart, p);
                const unsigned long long bm = readlane64(bmask, kk);
                if (bm & pendmask) { incmask |= bm; pendmask |= readlane64(btmask, kk); }
                done |= bm | (1ull << p);
            }
        }
        const bool inc = (incmask >> lane) & 1ull;
        int lowest = 0x7fffffff;
        if (inc) {
#pragma unroll
            for (int q = 0; q < kSmIn; ++q) {
                const int t2 = tr[q];
                if (t2 >= 0 && t2 - base < lo_lane) { pend[t2] = 1; lowest = min(lowest, t2); }
            }
        }
        if (mine) pend[r] = inc ? 1 : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) lowest = min(lowest, __shfl_xor(lowest, d));
        sm_fence();
        const int lo_eff = base + lo_lane;
        if (minpend >= lo_eff) minpend = 0x7fffffff;
        minpend = min(minpend, lowest);
        hi = lo_eff - 1;
    }
    // inclusion flags by node and the subgraph's own order
    int nv = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int b0 = 0; b0 <= top; b0 += 64) {
        const int r = b0 + lane;
        const int v = r <= top ? g.rank[r] : 0;
        const bool in = r <= top && pend[r] != 0;
        const unsigned long long mk = __ballot(in);
        if (r <= top) g.inc[v] = in ? 1 : 0;
        if (in) { const int pos = nv + __popcll(mk & lt); g.rsub[pos] = static_cast<uint16_t>(v); g.nsub[v] = static_cast<uint16_t>(pos); }
        nv += __popcll(mk);
    }
    sm_fence();
    return nv;
}

// ---- row descriptors: one word per DP row ----
// bits 0-7 symbol, 8 sink (no successor inside the (sub)graph), 9-11 number of predecessors (1..4), 12 the only predecessor
// is the virtual start row, 16-31 four 4-bit distances to the predecessor rows in in-edge order (16 is stored as 0); a row
// with five to eight predecessors (bit 13): 16-19 its entry of the side table (eight distance nibbles), 20-23 their number.
// Also leaves mark[r] = row r has a successor.  Returns a bail reason or 0.
__device__ __forceinline__ int sm_desc(const SmPtrs& g, int V, bool partial, int lane) {
    const sm_lds<uint16_t> rk = partial ? g.rsub : g.rank;
    const sm_lds<uint16_t> nr = partial ? g.nsub : g.n2r;
    for (int r = lane; r < V; r += 64) g.mark[r] = 0;
    if (lane == 0) g.misc[8] = 0;
    sm_fence();
    int bad = 0;
    for (int r = lane; r < V; r += 64) {
        const int v = rk[r];
        const int k = g.ink[v];
        int np = 0; uint32_t dd = 0;
#pragma unroll
        for (int q = 0; q < kSmIn; ++q) {
            if (q < k) {
                const int t = g.intail[v * kSmIn + q];
                if (!partial || g.inc[t]) {
                    const int p = nr[t];
                    const int dist = r - p;
                    if (dist > kSmWin || dist < 1) bad = dist < 1 ? kSmBug : kSmFar;
                    dd |= static_cast<uint32_t>(dist & 15) << (4 * np);
                    ++np;
                    g.mark[p] = 1;
                }
            }
        }
        uint32_t meta = static_cast<uint32_t>(g.code[v]);
        if (np == 0) meta |= static_cast<uint32_t>(kSmRow0) | (1u << 9);
        else if (np <= 4) meta |= (static_cast<uint32_t>(np) << 9) | (dd << 16) | ((np == 1 && dd == 1u) ? static_cast<uint32_t>(kSmChain) : 0u);
        else {
            // five to eight predecessors (rare): the eight distance nibbles go to the side table
            const int idx = __hip_atomic_fetch_add(g.misc + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (idx >= kSmWideRows) bad = kSmInDeg;
            else { g.misc[16 + idx] = static_cast<int32_t>(dd); meta |= static_cast<uint32_t>(kSmWide) | (static_cast<uint32_t>(idx) << 16) | (static_cast<uint32_t>(np) << 20); }
        }
        g.desc[r] = meta;
    }
    sm_fence();
    for (int r = lane; r < V; r += 64) if (!g.mark[r]) g.desc[r] |= 256u;
    sm_fence();
    const unsigned long long bm = __ballot(bad != 0);
    if (bm) return __builtin_amdgcn_readlane(bad, __builtin_ctzll(bm));
    return 0;
}

// ---- NW DP (window.cpp:95-97,104-106), Z domain, packed int16, move codes out ----
struct SmDpOut { int best, best_row, tied; unsigned int pred_rows; };
// One step of the move-code assembly of the PREVIOUS row, issued between two steps of the current row's prefix scan: a DPP
// read needs two wait states after the VALU write of its source, and the s_nop the compiler would put there costs the wave
// an issue slot all the same (the kernel is issue-bound: profiles/r04/f_sq_cfg4_summary.txt).  bit 0 clear = a diagonal move
// reproduces the cell, bit 1 clear = a vertical one does (poa_band.hpp); the predecessor numbers of a row with several
// predecessors are added behind the scan (rare).
template <int NP, int STEP>
__device__ __forceinline__ void sm_code_step(const uint32_t (&accp)[NP], const uint32_t (&dpvp)[NP], const uint32_t (&uvp)[NP], uint32_t (&td)[NP], uint32_t (&tu)[NP],
                                             uint32_t& word, uint32_t ONE) {
    if constexpr (STEP == 0) {
#pragma unroll
        for (int q = 0; q < NP; ++q) td[q] = pk_sub(accp[q], dpvp[q]);
    } else if constexpr (STEP == 1) {
#pragma unroll
        for (int q = 0; q < NP; ++q) tu[q] = pk_sub(accp[q], uvp[q]);
    } else if constexpr (STEP == 2) {
#pragma unroll
        for (int q = 0; q < NP; ++q) td[q] = pk_minu(td[q], ONE);
    } else if constexpr (STEP == 3) {
#pragma unroll
        for (int q = 0; q < NP; ++q) tu[q] = pk_minu(tu[q], ONE);
    } else if constexpr (STEP == 4) {
        if constexpr (NP == 2) { td[0] = __builtin_amdgcn_perm(td[NP - 1], td[0], 0x06040200u); tu[0] = __builtin_amdgcn_perm(tu[NP - 1], tu[0], 0x06040200u); }
        else { td[0] = __builtin_amdgcn_perm(0u, td[0], 0x0c0c0200u); tu[0] = __builtin_amdgcn_perm(0u, tu[0], 0x0c0c0200u); }
    } else {
        word = (tu[0] << 1) | td[0];
    }
    // (a use right here: without it the steps are sunk out of the gaps into the block that consumes the word)
    if constexpr (STEP == 0 || STEP == 2) { for (int q = 0; q < NP; ++q) asm volatile("" : "+v"(td[q])); }
    else if constexpr (STEP == 1 || STEP == 3) { for (int q = 0; q < NP; ++q) asm volatile("" : "+v"(tu[q])); }
    else if constexpr (STEP == 4) asm volatile("" : "+v"(td[0]), "+v"(tu[0]));
    else asm volatile("" : "+v"(word));
}

template <int NP>
__device__ __forceinline__ SmDpOut sm_dp(const SmPtrs& g, int V, int len, sm_lds<uint8_t> seq, RCN_G uint8_t* cmat, int m, int x, int gp, int lane) {
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    constexpr int LPC = 2 * NP;                 // columns per lane
    constexpr uint32_t rowb = 128u * NP;        // bytes of a code row
    const int mg = m - gp, xg = x - gp;
    uint32_t MG = pack2(mg, mg), XM = pack2(xg - mg, xg - mg), ONE = 0x00010001u;
    const uint32_t GG = pack2(gp, gp), NEGP = pack2(kNeg16, kNeg16);
    asm volatile("; constants live in VGPRs" : "+v"(MG), "+v"(XM), "+v"(ONE));
    uint32_t sqx[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int j0 = lane * LPC + 2 * q, j1 = j0 + 1;
        const int l0 = seq[min(max(j0 - 1, 0), 255)], l1 = seq[min(max(j1 - 1, 0), 255)];
        const int s0 = (j0 >= 1 && j0 <= len) ? l0 : 0x100, s1 = (j1 >= 1 && j1 <= len) ? l1 : 0x100;
        sqx[q] = pack2(s0, s1);
    }
    u32x16 w0, w1;
#pragma unroll
    for (int k = 0; k < 16; ++k) { w0[k] = NEGP; w1[k] = NEGP; }
    int zsh = static_cast<int>(0x80000000u);
    uint32_t mpv = static_cast<uint32_t>(kNeg16) << 16;
    const int own_lane = (len / LPC) & 63, own_q = (len % LPC) >> 1, own_hi = len & 1;
    int best = 0, best_row = 0, have_best = 0, tied = 0;
    unsigned int pred_rows = 0;
    // the row just finished: what a chain row (its only predecessor is the row right above: most rows) reads without an
    // indexed register access, and what its move codes are made from one row later (sm_code_step).  Two sets, used in
    // turn by the two rows of a loop trip: no copies on the back-edge.
    struct RowRegs { uint32_t acc[NP], dpv[NP], uv[NP], aq[NP]; int npf; };
    RowRegs A, B;
#pragma unroll
    for (int q = 0; q < NP; ++q) { A.acc[q] = NEGP; A.dpv[q] = NEGP; A.uv[q] = NEGP; A.aq[q] = 0u; }
    A.npf = 1;
    uint32_t coff = static_cast<uint32_t>(LPC) * static_cast<uint32_t>(lane);       // this lane's codes of the row just finished (row 0: nothing is read there)
    auto store_codes = [&](const RowRegs& pv, uint32_t word) __attribute__((always_inline)) {
        if (__builtin_expect(pv.npf > 1, 0)) {
            // bits 2-4 / 5-7: the first predecessor in in-edge order that attains the maximum at the previous / at this column
            uint32_t bA, bAsh;
            if constexpr (NP == 2) {
                bA = __builtin_amdgcn_perm(pv.aq[NP - 1], pv.aq[0], 0x06040200u);
                const uint32_t bAl = __builtin_amdgcn_update_dpp(0u, bA, 0x138, 0xf, 0xf, true);
                bAsh = __builtin_amdgcn_alignbit(bA, bAl, 24);
            } else {
                bA = __builtin_amdgcn_perm(0u, pv.aq[0], 0x0c0c0200u);
                const uint32_t bAl = __builtin_amdgcn_update_dpp(0u, bA, 0x138, 0xf, 0xf, true);
                bAsh = ((bA << 8) | (bAl >> 8)) & 0xffffu;
            }
            word = (bA << 5) | (bAsh << 2) | word;
        }
        if constexpr (NP == 2) *reinterpret_cast<RCN_G uint32_t*>(cmat + coff) = word;
        else *reinterpret_cast<RCN_G uint16_t*>(cmat + coff) = static_cast<uint16_t>(word);
    };
    // one DP row: PV_ is the row above (its codes are assembled and stored here), CU_ receives this row.  A macro, expanded
    // twice per loop trip: as a lambda the register window it indexes (w0 / w1) ends up in scratch memory.
#define RCN_SM_GAP(PV_, k) do { __builtin_amdgcn_sched_barrier(0); sm_code_step<NP, (k)>(PV_.acc, PV_.dpv, PV_.uv, td, tu, wordp, ONE); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RCN_SM_ROW(I_, META_, PV_, CU_) do { \
        const int i__ = (I_); const uint32_t meta__ = (META_); \
        const uint32_t sy = meta__ & 255u, symsym = sy | (sy << 16); \
        uint32_t P[NP]; \
_Pragma("unroll") \
        for (int q = 0; q < NP; ++q) P[q] = pk_profile(sqx[q], symsym, ONE, XM, MG); \
        uint32_t M[NP]; \
_Pragma("unroll") \
        for (int q = 0; q < NP; ++q) CU_.aq[q] = 0u; \
        int npf = 1; \
        if (__builtin_expect((meta__ & kSmChain) != 0u, 1)) { \
_Pragma("unroll") \
            for (int q = 0; q < NP; ++q) M[q] = PV_.acc[q]; \
        } else { \
            uint32_t dd = meta__ >> 16; \
            npf = static_cast<int>((meta__ >> 9) & 7); \
            if (__builtin_expect((meta__ & kSmWide) != 0u, 0)) { \
                npf = static_cast<int>((meta__ >> 20) & 15); \
                dd = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(g.misc[16 + ((meta__ >> 16) & 15)])); \
            } \
            { \
                const int wi = (i__ - static_cast<int>(dd & 15)) & 15; \
                M[0] = w0[wi]; \
                if (NP > 1) M[NP - 1] = w1[wi]; \
                if (meta__ & kSmRow0) { \
_Pragma("unroll") \
                    for (int q = 0; q < NP; ++q) M[q] = 0u; \
                } \
            } \
_Pragma("unroll 1") \
            for (int e = 1; e < npf; ++e) { \
                const int wi = (i__ - static_cast<int>((dd >> (4 * e)) & 15)) & 15; \
                uint32_t zq[NP]; \
                zq[0] = w0[wi]; \
                if (NP > 1) zq[NP - 1] = w1[wi]; \
                const uint32_t Q = pack2(e, e); \
_Pragma("unroll") \
                for (int q = 0; q < NP; ++q) { \
                    const uint32_t gt = pk_minu(pk_sub(pk_max(M[q], zq[q]), M[q]), ONE); \
                    CU_.aq[q] = pk_mad(gt, pk_sub(Q, CU_.aq[q]), CU_.aq[q]); \
                    M[q] = pk_max(M[q], zq[q]); \
                } \
            } \
        } \
        CU_.npf = npf; \
        const uint32_t mprev = mpv = __builtin_amdgcn_update_dpp(mpv, M[NP - 1], 0x138, 0xf, 0xf, false); \
        uint32_t acc[NP]; \
_Pragma("unroll") \
        for (int q = 0; q < NP; ++q) { \
            const uint32_t D = __builtin_amdgcn_alignbit(M[q], q == 0 ? mprev : M[q - 1], 16); \
            CU_.dpv[q] = pk_add(D, P[q]); CU_.uv[q] = pk_add(M[q], GG); \
            acc[q] = pk_max(CU_.dpv[q], CU_.uv[q]); \
        } \
_Pragma("unroll") \
        for (int q = 0; q < NP; ++q) acc[q] = pk_chain_pair(acc[q]); \
_Pragma("unroll") \
        for (int q = 1; q < NP; ++q) acc[q] = pk_max_bhi(acc[q], acc[q - 1]); \
        int sc = static_cast<int>(acc[NP - 1]) >> 16; \
        uint32_t td[NP], tu[NP], wordp = 0u; \
        { \
            constexpr int I = static_cast<int>(0x80000000u); \
            RCN_SM_GAP(PV_, 0); sc = max(sc, dpp_or<0x111, 0xf>(I, sc)); \
            RCN_SM_GAP(PV_, 1); sc = max(sc, dpp_or<0x112, 0xf>(I, sc)); \
            RCN_SM_GAP(PV_, 2); sc = max(sc, dpp_or<0x114, 0xf>(I, sc)); \
            RCN_SM_GAP(PV_, 3); sc = max(sc, dpp_or<0x118, 0xf>(I, sc)); \
            RCN_SM_GAP(PV_, 4); sc = max(sc, dpp_or<0x142, 0xa>(I, sc)); \
            RCN_SM_GAP(PV_, 5); sc = max(sc, dpp_or<0x143, 0xc>(I, sc)); \
            __builtin_amdgcn_sched_barrier(0); \
        } \
        store_codes(PV_, wordp); \
        coff += rowb; \
        zsh = dpp_or<0x138, 0xf>(zsh, sc); \
        const int zex = max(zsh, kNeg16); \
_Pragma("unroll") \
        for (int q = 0; q < NP; ++q) { acc[q] = pk_max_blo(acc[q], static_cast<uint32_t>(zex)); CU_.acc[q] = acc[q]; } \
        { \
            const int wi = i__ & 15; \
            w0[wi] = acc[0]; \
            if (NP > 1) w1[wi] = acc[NP - 1]; \
        } \
        if (__builtin_expect((meta__ & 256u) != 0u, 0)) { \
            uint32_t fv = acc[0]; \
_Pragma("unroll") \
            for (int q = 1; q < NP; ++q) if (own_q == q) fv = acc[q]; \
            const int v16 = own_hi ? (static_cast<int>(fv) >> 16) : (static_cast<int>(fv << 16) >> 16); \
            const int val = __builtin_amdgcn_readlane(v16, own_lane); \
            if (!have_best || best < val) { have_best = 1; best = val; best_row = i__; tied = 1; } \
            else if (best == val) { \
                if (tied < 8 && lane == 0) g.misc[tied] = i__; \
                ++tied; \
            } \
        } \
    } while (0)
#pragma unroll 1
    for (int rbase = 0; rbase < V; rbase += 64) {
        uint32_t dmeta = (1u << 9) | static_cast<uint32_t>(kSmRow0);
        if (rbase + lane < V) dmeta = g.desc[rbase + lane];
        {
            int npl = rbase + lane < V ? static_cast<int>((dmeta & kSmWide) ? ((dmeta >> 20) & 15) : ((dmeta >> 9) & 7)) : 0;
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) npl += __shfl_xor(npl, sh);
            pred_rows += static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(npl));
        }
        const int rend = min(V, rbase + 64);
        uint32_t meta_next = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(dmeta), 0));
        int i = rbase + 1;
#pragma unroll 1
        for (; i + 1 <= rend; i += 2) {
            const uint32_t m0 = meta_next;
            const uint32_t m1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(dmeta), i & 63));
            meta_next = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(dmeta), (i + 1) & 63));
            RCN_SM_ROW(i, m0, A, B);
            RCN_SM_ROW(i + 1, m1, B, A);
        }
        if (i <= rend) {                          // an odd row at the end of the block
            RCN_SM_ROW(i, meta_next, A, B);
            A = B;
        }
    }
    {
        // the codes of the last row
        uint32_t td[NP], tu[NP], wordp = 0u;
        sm_code_step<NP, 0>(A.acc, A.dpv, A.uv, td, tu, wordp, ONE); sm_code_step<NP, 1>(A.acc, A.dpv, A.uv, td, tu, wordp, ONE);
        sm_code_step<NP, 2>(A.acc, A.dpv, A.uv, td, tu, wordp, ONE); sm_code_step<NP, 3>(A.acc, A.dpv, A.uv, td, tu, wordp, ONE);
        sm_code_step<NP, 4>(A.acc, A.dpv, A.uv, td, tu, wordp, ONE); sm_code_step<NP, 5>(A.acc, A.dpv, A.uv, td, tu, wordp, ONE);
        store_codes(A, wordp);
    }
#undef RCN_SM_ROW
#undef RCN_SM_GAP
    SmDpOut o; o.best = best; o.best_row = have_best ? best_row : 0; o.tied = tied; o.pred_rows = pred_rows;
    return o;
}

// ---- several sinks share the best score: spoa's first one in ITS rank order, by the id rule of phase_sink_tie_rule ----
// (poa_k2_sinktie.hpp has the proof).  Returns the row, or 0 when the rule does not decide (the window leaves the kernel).
__device__ __forceinline__ int sm_sink_tie(const SmPtrs& g, int tied, int best_row, bool partial, int L, int lane) {
    int row_out = 0;
    if (lane == 0 && tied <= 8) {
        const sm_lds<uint16_t> rk = partial ? g.rsub : g.rank;
        const sm_lds<uint16_t> nr = partial ? g.nsub : g.n2r;
        bool classified = true;
        long long bestkey = 0x7fffffffffffffffll; int pick = -1;
        for (int k = 0; k < tied && classified; ++k) {
            const int v = rk[(k == 0 ? best_row : g.misc[k]) - 1];
            const int na = g.alcnt[v];
            // (the ring as the (sub)graph has it: members outside a Subgraph start nothing and order nothing, see phase_sink_tie_rule)
            int rm = v, na_in = 0;
            for (int a = 0; a < na; ++a) {
                const int u = static_cast<int>(g.ring[v * kSmRing + a]);
                if (partial && !g.inc[u]) continue;
                ++na_in; rm = min(rm, u);
            }
            long long key;
            if (rm < L) key = (static_cast<long long>(rm) << 32) | static_cast<unsigned int>(v);
            else if (na_in == 0) key = (0x7ffffffell << 32) | (static_cast<unsigned int>(v) << 6);
            else {
                // a ring of non-backbone nodes none of which has a successor in the (sub)graph: the DFS start loop finds it at
                // its smallest id m and appends m, then m's aligned list in list order
                bool closed = true;
                for (int a = -1; a < na && closed; ++a) {
                    const int u = a < 0 ? v : static_cast<int>(g.ring[v * kSmRing + a]);
                    if (partial && !g.inc[u]) continue;
                    if (g.mark[nr[u]]) closed = false;
                }
                if (!closed) { classified = false; break; }
                int pos = 0;
                if (v != rm) {
                    if (partial && !g.inc[rm]) { classified = false; break; }
                    const int nm = g.alcnt[rm];
                    pos = -1;
                    for (int a = 0, q = 0; a < nm; ++a) {
                        const int u = g.ring[rm * kSmRing + a];
                        if (partial && !g.inc[u]) continue;
                        ++q;
                        if (u == v) { pos = q; break; }
                    }
                    if (pos < 0) { classified = false; break; }
                }
                key = (0x7ffffffell << 32) | (static_cast<unsigned int>(rm) << 6) | static_cast<unsigned int>(pos);
            }
            if (key < bestkey) { bestkey = key; pick = v; }
        }
        if (classified && pick >= 0) row_out = nr[pick] + 1;
    }
    return bcast0(row_out);
}

// ---- traceback over the move codes: post[pos] = DP row aligned to sequence position pos, or -1 ----
// spoa's priority (diagonal over the in-edges, vertical over the in-edges, horizontal) is a table lookup on the code byte.
// The 64 lanes decode the successor of every cell of an 8 x 8 box (a parallelogram: one row of skew per column, see
// poa_k2_traceback.hpp) gathered straight from the code matrix; the walk inside the box is one v_readlane per step; the
// gather of the box a purely diagonal walk reaches next is issued before the current box is decoded.  Returns 0, or
// kSmBug when the codes lead nowhere.
template <int NP>
__device__ __forceinline__ int sm_traceback(const SmPtrs& g, int best_row, int len, RCN_G const uint8_t* cmat, int lane, unsigned int& n_boxes, unsigned int& n_regather) {
    constexpr uint32_t rowb = 128u * NP;
    const int a = lane >> 3, b = lane & 7;
    auto gather = [&](int ci, int cj) -> int {
        const int ii = ci - a - b, jj = cj - b;
        const bool ok = ii >= 1 && jj >= 0;
        const uint32_t o = ok ? static_cast<uint32_t>(ii) * rowb + static_cast<uint32_t>(jj) : rowb;
        return cmat[o];
    };
    int i = best_row, j = len;
    // the boxes a purely diagonal walk visits are 8 rows and 8 columns apart: their gathers are kept four deep in flight (a
    // gather is an L2 round trip, a box's decode and walk a few hundred clocks); a walk that leaves the diagonal starts over
    int c0 = gather(i, j), c1 = gather(i - 8, j - 8), c2 = gather(i - 16, j - 16), c3 = gather(i - 24, j - 24);
    int guard = 0;
    while (!(i == 0 && j == 0)) {
        ++n_boxes;
        const int code = c0;
        const int ii = i - a - b, jj = j - b;
        const bool inside = ii >= 0 && jj >= 0;
        const uint32_t meta = (inside && ii >= 1) ? g.desc[ii - 1] : 0u;
        const bool row0 = ii == 0, jpos = jj > 0;
        const bool dg = !row0 && jpos && !(code & 1);
        const bool up = !row0 && !dg && !(code & 2);
        int mv = dg ? kMvDiag : up ? kMvUp : jpos ? kMvLeft : kMvInvalid;
        const int q = dg ? ((code >> 2) & 7) : up ? (code >> 5) : 0;
        const bool wide = (meta & kSmWide) != 0u;
        const uint32_t ddw = static_cast<uint32_t>(g.misc[16 + ((meta >> 16) & 15)]);      // (read by every lane: no exec-masked region)
        const uint32_t ddq = wide ? ddw : (meta >> 16);
        const int npq = wide ? static_cast<int>((meta >> 20) & 15) : static_cast<int>((meta >> 9) & 7);
        const int dist = ((static_cast<int>(ddq >> (4 * q)) - 1) & 15) + 1;
        int pi = (meta & kSmRow0) ? 0 : ii - dist;
        if (q >= npq) pi = -1;                                                 // a predecessor number the row does not have
        mv = (!inside || (mv != kMvLeft && pi < 0)) ? kMvInvalid : mv;
        const int ni = mv == kMvLeft ? ii : pi, nj = jj - (mv == kMvUp ? 0 : 1);
        const int nb = j - nj, na = i - ni - nb;
        const bool leaves = (ni == 0 && nj == 0) || na < 0 || na >= 8 || nb >= 8;
        const int nx = mv == kMvInvalid ? kNxInvalid : leaves ? kNxExit : na * 8 + nb;
        // the walk starts on the box's first row (lane b is cell b of it) and mostly stays there: one diagonal step per column,
        // predecessor one row up.  That leading run is taken in one go (a ballot instead of a v_readlane round trip per step)
        int idx = __builtin_ctzll(~__ballot(lane < 8 && nx == lane + 1) | 0x80ull), nxt = kNxInvalid;
        unsigned long long vis = (1ull << idx) - 1ull;
        for (;;) {
            nxt = __builtin_amdgcn_readlane(nx, idx);
            if (nxt >= 64) break;
            vis |= 1ull << idx;
            idx = nxt;
        }
        if (nxt != kNxInvalid) vis |= 1ull << idx;
        if (((vis >> lane) & 1ull) && mv != kMvUp && jj >= 1) g.post[jj - 1] = static_cast<int16_t>((mv == kMvDiag) ? ii : -1);
        int ti, tj;
        if (nxt == kNxInvalid) { if (idx == 0) return kSmBug; ti = __builtin_amdgcn_readlane(ii, idx); tj = __builtin_amdgcn_readlane(jj, idx); }
        else { ti = __builtin_amdgcn_readlane(ni, idx); tj = __builtin_amdgcn_readlane(nj, idx); }
        if (ti == i - 8 && tj == j - 8) { c0 = c1; c1 = c2; c2 = c3; c3 = gather(ti - 24, tj - 24); }
        else if (!(ti == 0 && tj == 0)) { ++n_regather; c0 = gather(ti, tj); c1 = gather(ti - 8, tj - 8); c2 = gather(ti - 16, tj - 16); c3 = gather(ti - 24, tj - 24); }
        i = ti; j = tj;
        if (++guard > 4 * (kSmMaxCap + kSmLen)) return kSmBug;
    }
    sm_fence();
    return 0;
}

// ---- AddAlignment (window.cpp:110-119) + order merge, one wave over the sequence positions ----
// Same per-position phases as phase_add / phase_merge (poa_kernel.hpp): a global alignment consumes every position exactly
// once, so positions are independent up to the node numbering (prefix count) and the order anchors (prefix max); distinct
// positions touch distinct nodes, rings and in-records.  Work arrays over the dead inc / mark / nsub / desc areas.
struct SmAddOut { int n, why; };
__device__ __forceinline__ SmAddOut sm_add(const SmPtrs& g, int n_old, int ncap, bool partial, int len, sm_lds<uint8_t> seq, sm_lds<uint8_t> qual, bool has_qual,
                                           RCN_G uint32_t* wgt, RCN_G uint32_t* cov, int lane) {
    const sm_lds<uint16_t> rk = partial ? g.rsub : g.rank;
    const sm_lds<uint16_t> curr = reinterpret_cast<sm_lds<uint16_t>>(g.desc);             // [len] node that carries the base
    const sm_lds<int16_t> anch = reinterpret_cast<sm_lds<int16_t>>(g.desc) + 256;         // [len] order anchor
    const sm_lds<int16_t> newa = reinterpret_cast<sm_lds<int16_t>>(g.nsub);               // [nn] anchors of the new nodes, in sequence order
    const sm_lds<uint8_t> kindv = g.inc;                                                   // [len] (inc | mark are contiguous: 2 ncap >= 256)
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t count = len >= 2 ? 1u : 0u;
    SmAddOut out; out.n = n_old; out.why = 0;
    int nn = 0, anchor = -1;
    for (int base = 0; base < len; base += 64) {
        const int pos = base + lane;
        const bool act = pos < len;
        const int row = act ? static_cast<int>(g.post[pos]) : -1;
        const int t = row <= 0 ? -1 : static_cast<int>(rk[row - 1]);
        const int ch = act ? seq[pos] : 0;
        int kind = 0, cur = -1, a = -1;
        if (act) {
            if (t < 0) kind = 1;
            else {
                a = g.n2r[t];
                const int na = g.alcnt[t];
                int found = g.code[t] == ch ? t : -1;
                for (int a2 = 0; a2 < na; ++a2) {
                    const int u = g.ring[t * kSmRing + a2];
                    a = max(a, static_cast<int>(g.n2r[u]));
                    if (found < 0 && g.code[u] == ch) found = u;
                }
                cur = found; kind = found >= 0 ? 0 : 2;
            }
        }
        const unsigned long long mk = __ballot(kind != 0);
        const int idx = nn + __popcll(mk & lt);
        a = max(wave_incl_scan_max(a), anchor);
        if (act) {
            curr[pos] = static_cast<uint16_t>(kind ? n_old + idx : cur);
            anch[pos] = static_cast<int16_t>(a);
            kindv[pos] = static_cast<uint8_t>(kind);
            g.post[pos] = static_cast<int16_t>(t);                 // (kind 2 joins the ring of t)
            if (kind) newa[idx] = static_cast<int16_t>(a);
        }
        nn += __popcll(mk);
        anchor = __builtin_amdgcn_readlane(a, 63);
    }
    if (n_old + nn > ncap) { out.why = kSmCap; return out; }
    sm_fence();
    int bad = 0;
    for (int pos = lane; pos < len; pos += 64) {
        const int kind = kindv[pos];
        if (kind) {
            const int id = curr[pos];
            g.code[id] = seq[pos]; g.ink[id] = 0;
            int nal = 0;
            if (kind == 2) {
                const int t = g.post[pos];
                const int na = g.alcnt[t];
                if (na >= kSmRing) bad = kSmRingFull;
                else {
                    for (int a2 = 0; a2 < na; ++a2) {
                        const int u = g.ring[t * kSmRing + a2];
                        const int nu = g.alcnt[u];
                        g.ring[u * kSmRing + nu] = static_cast<uint16_t>(id); g.alcnt[u] = static_cast<uint8_t>(nu + 1);
                        g.ring[id * kSmRing + a2] = static_cast<uint16_t>(u);
                    }
                    g.ring[t * kSmRing + na] = static_cast<uint16_t>(id); g.alcnt[t] = static_cast<uint8_t>(na + 1);
                    g.ring[id * kSmRing + na] = static_cast<uint16_t>(t);
                    nal = na + 1;
                }
            }
            g.alcnt[id] = static_cast<uint8_t>(nal);
        }
    }
    if (__ballot(bad != 0)) { out.why = kSmRingFull; return out; }
    sm_fence();
    // edges pos-1 -> pos: reinforce an existing one (a look at the head's in-record) or append it
    for (int pos = lane; pos < len; pos += 64) {
        if (pos >= 1) {
            const int tail = curr[pos - 1], head = curr[pos];
            // weights[pos - 1] + weights[pos] (uint32 arithmetic on (char)quality - 33; no quality: 1 + 1), as pair_weight of poa_core.hpp
            const uint32_t w = has_qual ? static_cast<uint32_t>(static_cast<int32_t>(static_cast<signed char>(qual[pos - 1])) - 33) +
                                          static_cast<uint32_t>(static_cast<int32_t>(static_cast<signed char>(qual[pos])) - 33) : 2u;
            const int k = g.ink[head];
            int slot = -1;
#pragma unroll
            for (int q = 0; q < kSmIn; ++q) if (q < k && slot < 0 && g.intail[head * kSmIn + q] == tail) slot = q;
            if (slot < 0) {
                if (k >= kSmIn) bad = kSmInDeg;
                else { g.intail[head * kSmIn + k] = static_cast<uint16_t>(tail); g.ink[head] = static_cast<uint8_t>(k + 1); slot = k; }
            }
            if (slot >= 0) __hip_atomic_fetch_add(&wgt[head * kSmIn + slot], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (count) __hip_atomic_fetch_add(&cov[curr[pos]], count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (__ballot(bad != 0)) { out.why = kSmInDeg; return out; }
    sm_fence();
    // order merge, in place: new node k (in sequence order, anchors ascending) goes behind rank newa[k]; an old rank r moves
    // up by the number of new nodes anchored below it.  Descending chunks: a chunk's targets lie at or above its own ranks.
    if (nn > 0) {
        const int a0 = bcast0(static_cast<int>(newa[0]));
        for (int base = ((n_old - 1) >> 6) << 6; base >= 0 && base + 63 > a0; base -= 64) {
            const int r = base + lane;
            const int v = r < n_old ? static_cast<int>(g.rank[r]) : 0;
            int cnt = 0;
            for (int k = 0; k < nn; ++k) cnt += static_cast<int>(newa[k]) < r ? 1 : 0;
            sm_fence();
            if (r < n_old && cnt > 0) { g.rank[r + cnt] = static_cast<uint16_t>(v); g.n2r[v] = static_cast<uint16_t>(r + cnt); }
            sm_fence();
        }
        for (int k = lane; k < nn; k += 64) {
            const int p = static_cast<int>(newa[k]) + 1 + k;
            g.rank[p] = static_cast<uint16_t>(n_old + k); g.n2r[n_old + k] = static_cast<uint16_t>(p);
        }
        sm_fence();
    }
    out.n = n_old + nn;
    return out;
}

// ---- consensus (window.cpp:122-146): heaviest bundle, branch completion, coverage trim ----
// The organisation of phase_cons2_* (poa_k2_consensus.hpp): over the maintained order while the maximum is unique and a
// sink; otherwise spoa's own DFS order first (lane 0, over LDS) and the same bundle over it, BranchCompletion serially.
struct SmConsOut { int len, flags, why; };
__device__ __forceinline__ SmConsOut sm_consensus(const SmPtrs& g, int n, int ncap, RCN_G const uint32_t* wgt, RCN_G const uint32_t* cov,
                                                  RCN_G uint8_t* out, uint64_t out_cap, int ns, bool tgs, bool trim, bool force_exact, int lane) {
    SmConsOut res; res.len = 0; res.flags = kFlagPolished; res.why = 0;
    const sm_lds<int32_t> sc = reinterpret_cast<sm_lds<int32_t>>(g.desc);        // [n] bundle scores, later the reversed path
    const sm_lds<uint16_t> pr = g.rsub;                                            // [n] rank of the chosen predecessor
    sm_lds<uint16_t> R = g.rank, N = g.n2r;                                        // the order the bundle runs over
    sm_lds<uint8_t> nons = g.mark;                                                 // [n] by rank: the node has an out-edge
    int k = 0;
    for (int pass = force_exact ? 1 : 0; pass < 2 && k == 0; ++pass) {
        if (pass == 1) {
            // spoa's exact DFS order (graph_toposort of poa_core.hpp, over the LDS graph): rank_x in nsub, its inverse over
            // inc | mark, the DFS marks in post, the stack in desc (bounded: a deeper stack leaves the kernel)
            const sm_lds<uint16_t> rx = g.nsub;
            const sm_lds<uint16_t> nx = reinterpret_cast<sm_lds<uint16_t>>(g.inc);
            const sm_lds<uint8_t> mk = reinterpret_cast<sm_lds<uint8_t>>(g.post);
            const sm_lds<uint16_t> stack = reinterpret_cast<sm_lds<uint16_t>>(g.desc);
            const int scap = 2 * ncap - 8;
            if (n > kSmPosBytes) { res.why = kSmStack; return res; }
            for (int v = lane; v < n; v += 64) mk[v] = 0;
            sm_fence();
            int bad = 0;
            if (lane == 0) {
                int nrk = 0;
                for (int s = 0; s < n && !bad; ++s) {
                    if ((mk[s] & 3) != 0) continue;
                    int sp = 0;
                    stack[sp++] = static_cast<uint16_t>(s);
                    while (sp > 0 && !bad) {
                        const int c = stack[sp - 1];
                        bool valid = true;
                        const int mc = mk[c];
                        if ((mc & 3) != 2) {
                            if (sp + kSmIn + kSmRing >= scap) { bad = 1; break; }
                            const int kk = g.ink[c];
                            for (int q = 0; q < kk; ++q) {
                                const int t = g.intail[c * kSmIn + q];
                                if ((mk[t] & 3) != 2) { stack[sp++] = static_cast<uint16_t>(t); valid = false; }
                            }
                            const bool ign = (mc & 4) != 0;
                            const int na = g.alcnt[c];
                            if (!ign) {
                                for (int a = 0; a < na; ++a) {
                                    const int u = g.ring[c * kSmRing + a];
                                    if ((mk[u] & 3) != 2) { stack[sp++] = static_cast<uint16_t>(u); mk[u] = static_cast<uint8_t>(mk[u] | 4); valid = false; }
                                }
                            }
                            if (valid) {
                                mk[c] = static_cast<uint8_t>((mc & 4) | 2);
                                if (!ign) {
                                    rx[nrk++] = static_cast<uint16_t>(c);
                                    for (int a = 0; a < na; ++a) rx[nrk++] = g.ring[c * kSmRing + a];
                                }
                            } else {
                                mk[c] = static_cast<uint8_t>((mc & 4) | 1);
                            }
                        }
                        if (valid) --sp;
                    }
                }
                if (!bad && nrk != n) bad = 2;
            }
            bad = bcast0(bad);
            if (bad) { res.why = bad == 1 ? kSmStack : kSmBug; return res; }
            sm_fence();
            for (int r = lane; r < n; r += 64) nx[rx[r]] = static_cast<uint16_t>(r);
            R = rx; N = nx; nons = reinterpret_cast<sm_lds<uint8_t>>(g.post);
            sm_fence();
        }
        // which nodes have an out-edge (by rank of the order in use)
        for (int r = lane; r < n; r += 64) nons[r] = 0;
        sm_fence();
        for (int v = lane; v < n; v += 64) {
            const int kk = g.ink[v];
#pragma unroll
            for (int q = 0; q < kSmIn; ++q) if (q < kk) nons[N[g.intail[v * kSmIn + q]]] = 1;
        }
        sm_fence();
        int gmax = static_cast<int>(0x80000000u), gmax_rank = -1, gtie = 0;
#pragma unroll 1
        for (int base = 0; base < n; base += 64) {
            const int r = base + lane;
            const int v = r < n ? static_cast<int>(R[r]) : 0;
            const int kk = r < n ? static_cast<int>(g.ink[v]) : 0;
            // the winning in-edge by weight; up to two more tails that tie on weight (then the tail SCORE decides, later edge
            // wins: TraverseHeaviestBundle's predicate is a lexicographic max over (weight, score[tail], edge order))
            int wq[kSmIn], tq[kSmIn];
#pragma unroll
            for (int q = 0; q < kSmIn; ++q) {
                wq[q] = q < kk ? static_cast<int>(__hip_atomic_load(&wgt[v * kSmIn + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : -1;
                tq[q] = q < kk ? static_cast<int>(N[g.intail[v * kSmIn + q]]) : -1;
            }
            int trA = -1, wA = 0, trB = -1, trC = -1, ntie = 0;
            {
                int wmax = -1;
#pragma unroll
                for (int q = 0; q < kSmIn; ++q) {
                    if (q < kk) {
                        if (wq[q] > wmax) { wmax = wq[q]; ntie = 1; trA = tq[q]; wA = wq[q]; trB = -1; trC = -1; }
                        else if (wq[q] == wmax) { ++ntie; if (ntie == 2) trB = tq[q]; else if (ntie == 3) trC = tq[q]; else trC = -2; }
                    }
                }
            }
            const int tl = trA >= base ? trA - base : -1;
            int fin = -1;
            if (trA >= 0 && trA < base) fin = wA + sc[trA];
            const unsigned long long amb = __ballot(trB >= 0 || trC == -2);
            const int cnt = min(64, n - base);
#pragma unroll 1
            for (int kq = 0; kq < cnt; ++kq) {
                if ((amb >> kq) & 1ull) {
                    const int wk = __builtin_amdgcn_readlane(wA, kq);
                    int bt = -1, bs = 0; bool have = false;
                    auto consider = [&](int tr) {
                        const int s = tr >= base ? __builtin_amdgcn_readlane(fin, tr - base) : bcast0(sc[tr]);
                        if (!have || s >= bs) { bs = s; bt = tr; have = true; }
                    };
#pragma unroll
                    for (int q = 0; q < kSmIn; ++q) {
                        const int wv = __builtin_amdgcn_readlane(wq[q], kq), tv = __builtin_amdgcn_readlane(tq[q], kq);
                        if (tv >= 0 && wv == wk) consider(tv);
                    }
                    if (lane == kq) { fin = wk + bs; trA = bt; }
                }
                const int sk = __builtin_amdgcn_readlane(fin, kq);
                if (tl == kq && !((amb >> lane) & 1ull)) fin = wA + sk;
            }
            if (r < n) { sc[r] = fin; pr[r] = static_cast<uint16_t>(trA < 0 ? 0xFFFF : trA); }
            const int fm = r < n ? fin : static_cast<int>(0x80000000u);
            int cm = fm;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) cm = max(cm, __shfl_xor(cm, d));
            const unsigned long long at = __ballot(fm == cm);
            if (cm > gmax) { gmax = cm; gmax_rank = base + __builtin_ctzll(at); gtie = __popcll(at) > 1; }
            else if (cm == gmax) gtie = 1;
            sm_fence();
        }
        if (pass == 1) {
            // over spoa's own order the first maximum IS spoa's choice; BranchCompletion on the LDS scores, serially
            int mx = gmax_rank;
            if (lane == 0) {
                while (nons[mx]) {
                    const int start = R[mx];
                    for (int h = 0; h < n; ++h) {
                        const int kk = g.ink[h];
                        bool is_out = false;
                        for (int q = 0; q < kk; ++q) is_out = is_out || g.intail[h * kSmIn + q] == start;
                        if (!is_out) continue;
                        for (int q = 0; q < kk; ++q) { const int tl2 = g.intail[h * kSmIn + q]; if (tl2 != start) sc[N[tl2]] = -1; }
                    }
                    int m2 = -1, m2s = 0;
                    for (int r = mx + 1; r < n; ++r) {
                        const int it = R[r];
                        int sv = -1, p = -1, ps = 0;
                        const int kk = g.ink[it];
                        for (int q = 0; q < kk; ++q) {
                            const int tr = N[g.intail[it * kSmIn + q]];
                            const int ts = sc[tr];
                            if (ts == -1) continue;
                            const int w = static_cast<int>(__hip_atomic_load(&wgt[it * kSmIn + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
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
            sm_fence();
        }
        if (!gtie && !nons[gmax_rank]) {
            int cur = gmax_rank;
            for (;;) {
                const int nxt = bcast0(static_cast<int>(pr[cur]));
                if (lane == 0) sc[k] = cur;
                ++k;
                if (nxt == 0xFFFF || k > n) break;
                cur = nxt;
            }
            sm_fence();
            if (k > n) { res.why = kSmBug; return res; }
        }
    }
    if (k == 0) { res.why = kSmBug; return res; }
    // consensus node k-1-i of the reversed list is position i; trim (window.cpp:125-146)
    auto node_at = [&](int i) -> int { return R[sc[k - 1 - i]]; };
    auto coverage = [&](int v) -> uint32_t {
        uint32_t c = __hip_atomic_load(&cov[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int na = g.alcnt[v];
        for (int a = 0; a < na; ++a) c += __hip_atomic_load(&cov[g.ring[v * kSmRing + a]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return c;
    };
    int bgn = 0, end = k - 1;
    if (tgs && trim) {
        const uint32_t avg = static_cast<uint32_t>(ns - 1) / 2;
        bgn = k;
        for (int b0 = 0; b0 < k && bgn == k; b0 += 64) {
            const int i = b0 + lane;
            const bool ok = i < k && coverage(node_at(i)) >= avg;
            const unsigned long long mk = __ballot(ok);
            if (mk) bgn = b0 + __builtin_ctzll(mk);
        }
        end = -1;
        for (int b0 = 0; b0 < k && end == -1; b0 += 64) {
            const int i = k - 1 - (b0 + lane);
            const bool ok = i >= 0 && coverage(node_at(i)) >= avg;
            const unsigned long long mk = __ballot(ok);
            if (mk) end = k - 1 - (b0 + __builtin_ctzll(mk));
        }
        if (bgn >= end) { bgn = 0; end = k - 1; res.flags |= kFlagChimeric; }
    }
    const int clen = end - bgn + 1;
    if (static_cast<uint64_t>(clen) > out_cap) { res.why = kSmCap; return res; }
    for (int i = lane; i < clen; i += 64) out[i] = g.code[node_at(bgn + i)];
    res.len = clen;
    return res;
}

// ---- the kernel ----
__global__ __launch_bounds__(64, 3) void poa_window_kernel_small(KParams P) {
    extern __shared__ int4 lds_dyn[];
    const int lane = threadIdx.x;
    const uint32_t lbase = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_dyn));
    const int ncap = P.ncap;
    const SmLayout lay = small_layout(ncap);
    const SmPtrs g = sm_ptrs(lbase, lay);
    RCN_G uint8_t* slot = gcast(P.scratch + static_cast<uint64_t>(blockIdx.x) * P.slot_bytes);
    RCN_G uint32_t* wgt = reinterpret_cast<RCN_G uint32_t*>(slot);                      // [ncap][kSmIn]
    RCN_G uint32_t* cov = wgt + static_cast<uint32_t>(ncap) * kSmIn;                    // [ncap]
    RCN_G uint8_t* cmat = slot + small_slot_codes(ncap);                                 // move codes, rows of 128 or 256 bytes
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // sub, desc, dp, traceback, add (+ merge), -, consensus, other
    unsigned long long st_cells = 0, st_pred = 0, st_bytes = 0, st_ties = 0;
    unsigned int st_done = 0, st_boxes = 0, st_regather = 0, st_chunks = 0, st_rows = 0, st_aligns = 0, st_intervals = 0;
    long long tck = clock64();
#define RCN_PHASE_S(k) do { const long long now__ = clock64(); ph[k] += now__ - tck; tck = now__; } while (0)
    for (;;) {
        RCN_PHASE_S(7);
        unsigned int wi = 0;
        if (lane == 0) wi = atomicAdd(P.next, 1u);
        wi = bcast0(wi);
        if (wi >= P.n_work) break;
        const uint32_t w = P.win_ids ? P.win_ids[wi] : P.work_base + wi;
        const uint32_t s0 = P.win_seq_off[w];
        const int ns = static_cast<int>(P.win_seq_off[w + 1] - s0);
        RCN_G const uint8_t* bb = gcast(P.bases + P.seq_off[s0]);
        const int L = static_cast<int>(P.seq_off[s0 + 1] - P.seq_off[s0]);
        const uint32_t oi = P.out_base + wi;
        RCN_G uint8_t* out = gcast(P.out_cons + P.out_off[oi]);
        const uint64_t out_cap = P.out_off[oi + 1] - P.out_off[oi];
        RCN_G uint32_t* out_len = gcast(P.out_len + oi);
        RCN_G uint8_t* out_flags = gcast(P.out_flags + oi);
        if (ns < 3) {                                              // window.cpp:68-71
            for (int i = lane; i < L; i += 64) out[i] = bb[i];
            if (lane == 0) { *out_len = L; *out_flags = 0; }
            continue;
        }
        int why = 0;
        if (L > ncap || L < 1) why = kSmCap;
        int n = L;
        if (!why) {
            // ---- backbone -> graph (window.cpp:73-77); every weight / coverage word of the slot starts at its final-or-zero value ----
            RCN_G const uint8_t* q0 = P.seq_has_qual[s0] ? gcast(P.quals + P.seq_off[s0]) : nullptr;
            static_assert(kSmIn == 8, "two 16-byte stores per node");
            for (int i = lane; i < ncap; i += 64) {
                uint4 w4 = make_uint4(0u, 0u, 0u, 0u);
                reinterpret_cast<RCN_G uint4*>(wgt)[2 * i + 1] = w4;
                if (i >= 1 && i < L) w4.x = static_cast<uint32_t>(pair_weight(q0, i));
                reinterpret_cast<RCN_G uint4*>(wgt)[2 * i] = w4;
                cov[i] = (i < L && L >= 2) ? 1u : 0u;
            }
            for (int i = lane; i < L; i += 64) {
                g.code[i] = bb[i]; g.alcnt[i] = 0; g.ink[i] = i > 0 ? 1 : 0;
                g.intail[i * kSmIn] = static_cast<uint16_t>(i > 0 ? i - 1 : 0);
                g.rank[i] = static_cast<uint16_t>(i); g.n2r[i] = static_cast<uint16_t>(i);
            }
            sm_fence(); sm_drain();                     // the plain stores above are ordered before every later atomic
        }
        // Layer metadata and bytes run one layer ahead of their use: the scalar loads of layer jl + 1 are issued at the top of
        // layer jl, its bases / qualities go into registers behind the Subgraph sweep and into the other half of lseq / lqual
        // after the traceback -- no phase of a layer waits for HBM on its own behalf.
        struct LayerMeta { uint32_t si; int len; bool partial, has_qual; int begin, end; uint64_t off; };
        auto load_meta = [&](int jl_) {
            LayerMeta m_;
            m_.si = s0 + P.order[s0 + jl_];
            m_.off = P.seq_off[m_.si];
            m_.len = static_cast<int>(P.seq_off[m_.si + 1] - m_.off);
            m_.partial = P.seq_full[m_.si] == 0; m_.has_qual = P.seq_has_qual[m_.si] != 0;
            m_.begin = static_cast<int>(P.seq_begin[m_.si]); m_.end = static_cast<int>(P.seq_end[m_.si]);
            return m_;
        };
        uint32_t pre_s = 0, pre_q = 0;                  // bytes lane, lane + 64, lane + 128, lane + 192 of the prefetched layer
        auto load_bytes = [&](const LayerMeta& m_) {
            RCN_G const uint8_t* sp = gcast(P.bases + m_.off);
            RCN_G const uint8_t* qp = gcast(P.quals + m_.off);
            pre_s = 0; pre_q = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pos = lane + 64 * k;
                const bool act = pos < m_.len && pos <= kSmLen;
                const uint32_t sb = act ? sp[pos] : 0u, qb = (act && m_.has_qual) ? qp[pos] : 0u;
                pre_s |= sb << (8 * k); pre_q |= qb << (8 * k);
            }
        };
        auto store_seq = [&](int half) {
#pragma unroll
            for (int k = 0; k < 4; ++k) g.lseq[half * 256 + lane + 64 * k] = static_cast<uint8_t>(pre_s >> (8 * k));
        };
        auto store_qual = [&]() {
#pragma unroll
            for (int k = 0; k < 4; ++k) g.lqual[lane + 64 * k] = static_cast<uint8_t>(pre_q >> (8 * k));
        };
        LayerMeta nxt = load_meta(1);
        if (!why) { load_bytes(nxt); store_seq(1); store_qual(); sm_fence(); }
        for (int jl = 1; jl < ns && !why; ++jl) {
            const LayerMeta cur = nxt;
            if (jl + 1 < ns) nxt = load_meta(jl + 1);
            const int len = cur.len;
            const bool partial = cur.partial;
            const sm_lds<uint8_t> seq = g.lseq + (jl & 1) * 256, qual = g.lqual;
            if (len > kSmLen || len < 1) { why = kSmLong; break; }
            int V = n;
            if (partial) {
                const int begin = cur.begin, end = cur.end;
                if (end >= n || begin > end) { why = kSmBug; break; }
                V = sm_subgraph(g, n, begin, end, lane, st_chunks, st_intervals);
            }
            if (jl + 1 < ns) load_bytes(nxt);           // (in flight through the descriptors, the DP and the traceback)
            RCN_PHASE_S(0);
            const int np_regs = len + 1 <= 128 ? 1 : 2;
            {
                const int ag = P.g < 0 ? -P.g : P.g, smax = max(max(P.m, P.x), 0);
                if (P.g >= 0 || static_cast<long long>(ag) * (V + 2) > kZLimit || static_cast<long long>(smax + ag) * (128 * np_regs) > kZLimit) { why = kSmRange; break; }
            }
            if (V < 1) { why = kSmBug; break; }
            why = sm_desc(g, V, partial, lane);
            RCN_PHASE_S(1);
            if (why) break;
            SmDpOut d = np_regs == 1 ? sm_dp<1>(g, V, len, seq, cmat, P.m, P.x, P.g, lane) : sm_dp<2>(g, V, len, seq, cmat, P.m, P.x, P.g, lane);
            {
                const unsigned long long W = static_cast<unsigned long long>(len) + 1ull;
                st_cells += static_cast<unsigned long long>(V + 1) * W;
                st_pred += static_cast<unsigned long long>(d.pred_rows) * W;
                const int amax = max(max(abs(P.m), abs(P.x)), abs(P.g));
                const unsigned long long sbytes = (static_cast<long long>(amax) * (V + static_cast<long long>(W)) < 32767) ? 2ull : 4ull;
                st_bytes += sbytes * (static_cast<unsigned long long>(V + 1) + d.pred_rows) * W;
            }
            RCN_PHASE_S(2);
            if (d.best_row == 0) { why = kSmBug; break; }
            int best_row = d.best_row;
            if (d.tied > 1) {
                sm_fence();
                best_row = sm_sink_tie(g, d.tied, d.best_row, partial, L, lane);
                if (best_row == 0) { why = kSmTie; break; }
                ++st_ties;
            }
            sm_drain();                                 // the code rows are in memory before the traceback gathers them
            why = np_regs == 1 ? sm_traceback<1>(g, best_row, len, cmat, lane, st_boxes, st_regather) : sm_traceback<2>(g, best_row, len, cmat, lane, st_boxes, st_regather);
            st_rows += static_cast<unsigned int>(V); ++st_aligns;
            if (jl + 1 < ns) store_seq((jl + 1) & 1);
            RCN_PHASE_S(3);
            if (why) break;
            const SmAddOut ao = sm_add(g, n, ncap, partial, len, seq, qual, cur.has_qual, wgt, cov, lane);
            if (jl + 1 < ns) { sm_fence(); store_qual(); }        // (the one quality buffer is this layer's until its edges are in)
            RCN_PHASE_S(4);
            why = ao.why;
            n = ao.n;
        }
        if (!why) {
            sm_drain();                                 // every weight / coverage atomic of this window has been performed
            const uint64_t wbases = P.seq_off[s0 + ns] - P.seq_off[s0];
            if (wbases * 444ull >= 0x7fffffffull) why = kSmRange;
            else {
                const SmConsOut co = sm_consensus(g, n, ncap, wgt, cov, out, out_cap, ns, P.win_type[w] == 1, P.trim != 0, P.force_exact != 0, lane);
                why = co.why;
                if (!why && lane == 0) { *out_len = static_cast<uint32_t>(co.len); *out_flags = static_cast<uint8_t>(co.flags); }
            }
            RCN_PHASE_S(6);
        }
        if (why) {
            sm_drain();                                 // (a bail inside sm_add leaves weight / coverage atomics in flight: the next window zero-fills the same slot words with plain stores)
            // (rare: straight to the counter -- an array indexed by `why` would live in scratch memory)
            if (lane == 0) { *out_len = 0; *out_flags = kFlagOverflow; atomicAdd(&P.stats[25 + (why < 10 ? why : 9)], 1ull); }
        } else ++st_done;
    }
    if (lane == 0) {
        atomicAdd(&P.stats[0], st_cells); atomicAdd(&P.stats[1], st_pred); atomicAdd(&P.stats[2], st_bytes);
        for (int k = 0; k < 8; ++k) atomicAdd(&P.stats[3 + k], ph[k]);
        atomicAdd(&P.stats[11], st_ties);
        atomicAdd(&P.stats[12], st_cells); atomicAdd(&P.stats[13], st_bytes);
        atomicAdd(&P.stats[25], static_cast<unsigned long long>(st_done));
        // work counters: alignments, DP rows, Subgraph sweep chunks, traceback boxes, boxes whose gather had to start over
        atomicAdd(&P.stats[35], static_cast<unsigned long long>(st_aligns)); atomicAdd(&P.stats[36], static_cast<unsigned long long>(st_rows));
        atomicAdd(&P.stats[37], static_cast<unsigned long long>(st_chunks)); atomicAdd(&P.stats[38], static_cast<unsigned long long>(st_boxes));
        atomicAdd(&P.stats[39], static_cast<unsigned long long>(st_regather));
        atomicAdd(&P.stats[40], static_cast<unsigned long long>(st_intervals));
    }
}

#endif  // RCN_SMALL_TU

}  // namespace rcn
