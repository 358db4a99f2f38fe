// poa_kernel2.hpp — the fast path of the MI355X window-consensus engine (gfx950).
//
// Same per-window algorithm as poa_kernel.hpp (racon's Window::generate_consensus, reference
// src/window.cpp:65-149), re-shaped around what the first kernel's measurements showed: with
// ~2000 windows per launch one wave per window leaves the chip idle and the launch time IS the
// latency of the slowest window.  Here a window is owned by a WORK-GROUP OF FOUR WAVES:
//
//   * sequence-to-graph NW (window.cpp:95-97,104-106): the four waves form a pipeline over
//     column blocks.  Wave w owns columns [w*128*NP, (w+1)*128*NP) and lags one row behind
//     wave w-1; the only things that cross a block border are one score per row (horizontal
//     carry) and one per predecessor row (diagonal carry), both read straight out of the
//     neighbour wave's LDS ring.  One `s_barrier` per row step keeps the skew.
//   * scores are int16, two cells per VGPR (v_pk_max_i16 / v_pk_add_i16 / v_pk_mad_i16), in the
//     "Z domain": Z[i][j] = H[i][j] - j*g.  The horizontal gap move then adds 0 (a plain prefix
//     max: in-register, then six v_max_i32_dpp steps across the wave), the vertical move adds
//     g and the diagonal move adds s(i,j) - g.  Because s(i,j) does not depend on the
//     predecessor, max over predecessors commutes with the one-column shift: the predecessor
//     rows are max-combined FIRST (one v_pk_max per extra in-edge) and shifted ONCE per row.
//     Z is bounded by -|g|*V <= Z <= (max(m,x,0)+|g|)*W, so int16 holds for every racon
//     parameterisation at w=500/1000 (checked per alignment; anything else goes to the int32
//     kernel).  Row 0 is identically zero in this domain.
//   * the score matrix is written to HBM once (2 B/cell, coalesced 256*NP B per wave and row)
//     for the traceback; predecessor rows are served from registers (row i-1) or the LDS ring.
//   * traceback (spoa priority diag > vertical > horizontal, predecessors in in-edge order):
//     all four waves stage a 64-row x 128-column int16 tile with global_load_lds, wave 0 walks it.
//   * graph phases (Subgraph mask, AddAlignment, order merge, consensus) are the single-wave
//     templates of poa_kernel.hpp run by wave 0; row descriptors are built by all 256 threads.
//
// Integer max-plus DP on an irregular DAG: no MFMA.
#pragma once
#include "poa_kernel.hpp"

namespace rcn {

constexpr int kWaves2 = 4;
constexpr int kThreads2 = 64 * kWaves2;
constexpr int kNeg16 = -32000;
constexpr int kZLimit = 31000;          // |Z| bound accepted for the int16 path

// ---- execution policies (see OneWaveBlock in poa_kernel.hpp) ----
__device__ __forceinline__ int* lds_words2() { extern __shared__ int4 lds_dyn[]; return reinterpret_cast<int*>(lds_dyn); }
__device__ __forceinline__ Ctx* ctx_lds2() { return reinterpret_cast<Ctx*>(lds_words2() + kLdsBytes / 4); }
struct Wave0Of4 {            // wave 0 of the 4-wave work-group, the other waves wait at the next Block4::sync()
    static constexpr int NT = 64;
    static __device__ __forceinline__ int tid() { return threadIdx.x; }
    static __device__ __forceinline__ void sync() { __threadfence_block(); }
    static __device__ __forceinline__ Ctx* ctx() { return ctx_lds2(); }
    static __device__ __forceinline__ int* work() { return lds_words2(); }
};
struct Block4 {
    static constexpr int NT = kThreads2;
    static __device__ __forceinline__ int tid() { return threadIdx.x; }
    static __device__ __forceinline__ void sync() { __threadfence_block(); __syncthreads(); }
    static __device__ __forceinline__ Ctx* ctx() { return ctx_lds2(); }
    static __device__ __forceinline__ int* work() { return lds_words2(); }
};

// one LDS word, read now (polling loops); invisible to the compiler's memory model on purpose
__device__ __forceinline__ uint32_t lds_poll(const uint32_t* p) {
    uint32_t v;
    const uint32_t a = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));      // LDS addresses are the low 32 bits
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
    return v;
}
// LDS-only barrier: waits for this wave's LDS traffic, NOT for its outstanding HBM stores
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- packed int16 helpers (two cells per VGPR: low half = even column) ----
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, static_cast<s16x2>(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pack2(int lo, int hi) { return (static_cast<uint32_t>(lo) & 0xffffu) | (static_cast<uint32_t>(hi) << 16); }
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// (sym == seq ? m - g : x - g) for both halves: t = min(seq ^ sym, 1); t * (x - m) + (m - g).  The empty asm
// keeps the compiler from turning min(a ^ b, 1) back into compare + select chains (five instructions).
__device__ __forceinline__ uint32_t pk_profile(uint32_t sqx, uint32_t symsym, uint32_t one, uint32_t xm, uint32_t mg) {
    uint32_t t = sqx ^ symsym;
    asm("" : "+v"(t));
    const u16x2 mn = __builtin_elementwise_min(__builtin_bit_cast(u16x2, t), __builtin_bit_cast(u16x2, one));
    uint32_t u = __builtin_bit_cast(uint32_t, mn);
    asm("" : "+v"(u));
    const s16x2 r = __builtin_bit_cast(s16x2, u) * __builtin_bit_cast(s16x2, xm) + __builtin_bit_cast(s16x2, mg);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_minu(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_bit_cast(uint32_t, static_cast<s16x2>(__builtin_bit_cast(s16x2, a) * __builtin_bit_cast(s16x2, b) + __builtin_bit_cast(s16x2, c)));
}
// One of the 3 * NP independent instructions of the next row's substitution profile (xor, min, mad per register).
// dp2_rows pins two of them between consecutive steps of the DPP prefix scan (scheduling barriers on both sides): a DPP
// read needs two wait states after the VALU write of its source, and every s_nop the compiler would otherwise put
// there costs the wave a full issue slot.
template <int NP, int O>
__device__ __forceinline__ void dp2_gap_op(uint32_t (&pw)[NP], const uint32_t (&sqx)[NP], uint32_t symsym, uint32_t one, uint32_t xm, uint32_t mg) {
    if constexpr (O < 3 * NP) {
        constexpr int q = O % NP, st = O / NP;
        if constexpr (st == 0) pw[q] = sqx[q] ^ symsym;
        else if constexpr (st == 1) pw[q] = pk_minu(pw[q], one);
        else pw[q] = pk_mad(pw[q], xm, mg);
    }
}
// Half broadcasts written as vector shuffles: instruction selection folds them into the VOP3P op_sel / op_sel_hi
// source modifiers of v_pk_max_i16 (no v_perm_b32 in front; inline asm would cost a hazard s_nop per use on gfx950).
// {lo, max(hi, lo)}
__device__ __forceinline__ uint32_t pk_chain_pair(uint32_t a) {
    const s16x2 av = __builtin_bit_cast(s16x2, a);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(av, __builtin_shufflevector(av, av, 0, 0)));
}
// {max(a.lo, b.hi), max(a.hi, b.hi)}
__device__ __forceinline__ uint32_t pk_max_bhi(uint32_t a, uint32_t b) {
    const s16x2 bv = __builtin_bit_cast(s16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_shufflevector(bv, bv, 1, 1)));
}
// {max(a.lo, b.lo), max(a.hi, b.lo)}
__device__ __forceinline__ uint32_t pk_max_blo(uint32_t a, uint32_t b) {
    const s16x2 bv = __builtin_bit_cast(s16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_shufflevector(bv, bv, 0, 0)));
}
__device__ __forceinline__ int wave_incl_scan_max_id(int v) {      // INT_MIN is max's identity: each step fuses into one v_max_i32_dpp
    constexpr int I = static_cast<int>(0x80000000u);
    v = max(v, dpp_or<0x111, 0xf>(I, v));
    v = max(v, dpp_or<0x112, 0xf>(I, v));
    v = max(v, dpp_or<0x114, 0xf>(I, v));
    v = max(v, dpp_or<0x118, 0xf>(I, v));
    v = max(v, dpp_or<0x142, 0xa>(I, v));
    v = max(v, dpp_or<0x143, 0xc>(I, v));
    return v;
}

// DP shape for a layer of `len` bases: one wave over (len+1) <= 512 columns, else the 4-wave pipeline;
// NP = packed VGPRs per lane (2 NP columns).  Returns NP | (WV << 8), 0 = not supported (int32 kernel).
// `wide` (heavy windows, see KParams::heavy_ns): the pipeline also for short layers -- twice the instructions
// in total but about half the latency per row, which is what counts for the windows that finish last.
__host__ __device__ __forceinline__ int dp2_cfg(int len, bool wide) {
    const int W = len + 1;
    if (W <= 512 && !wide) return ((W + 127) / 128) | (1 << 8);
    const int n = (W + 511) / 512;
    return n <= 4 ? (n | (4 << 8)) : 0;
}
// rows of the register window for NP packed VGPRs per lane (16 VGPRs in all; a power of two)
__host__ __device__ constexpr int dp2_window(int np) { return np <= 1 ? 16 : np == 2 ? 8 : 4; }
// rows of the LDS ring behind it (K of dp2_rows<NP, WV>)
// (tab: the one-wave DP keeps a 4-symbol substitution-profile table behind the ring, dp2_rows<NP, 1, true>)
__host__ __device__ constexpr int dp2_ring_rows(int np, int wv, bool tab = false) {
    return (kLdsBytes - 64 - (wv > 1 ? 64 * 4 * 4 + 64 : 0) - (tab ? 4 * 4 * 64 * wv * np : 0)) / (4 * 64 * wv * np) - 1;
}

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

__device__ __forceinline__ void block4_scan(int* xch, int wv, int lane, int cnt, int wmax, int& off, int& total, int& pmax, int& tmax);
__device__ __forceinline__ void band_row_offsets(const Ctx& c, Win& g, RCN_G const int32_t* rank);
__host__ __device__ constexpr int dp2_ring_rows_band(int np, bool tab);

// ---- phase: row descriptors (all 256 threads) + row 0 of Z ----
__device__ __noinline__ void phase_desc2() {
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
    RCN_G uint32_t* H = reinterpret_cast<RCN_G uint32_t*>(g.H.ptr());
    for (int j = t; j < (g.hstride >> 1); j += kThreads2) H[j] = 0u;
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
__device__ unsigned long long g_wclk[4096][8];   // per work item: phase clocks     // per wave: cycles in row bodies, cycles in barriers
#endif
// ---- phase: NW sequence-to-graph DP ----
// WV = 1: wave 0 alone owns all columns (up to 128*NP); no barrier, no border traffic.  The default for
//         w=500 windows: with ~2000 windows per launch the chip is latency bound, and a row costs about the
//         same number of instructions whether a lane owns 2 or 8 columns.
// WV = 4: the four waves form a pipeline over column blocks of 128*NP (layers longer than 511 bases).
// TAB (WV = 1, windows whose bases are all A/C/G/T): the substitution profile of a row -- 3 VALU instructions per
// register, 12 of the ~58 of a row, and VALU instructions are what a row costs -- comes from a 4-symbol table built once
// per layer behind the LDS ring (slot = (code >> 1) & 3: A 0, C 1, T 2, G 3): one address add and one ds_read per row.
template <int NP, int WV, bool TAB = false>
__device__ __noinline__ void dp2_rows() {
    static_assert(!TAB || WV == 1, "profile table: one-wave DP only");
    constexpr int NTH = 64 * WV;
    const int t = threadIdx.x, lane = t & 63, wv = WV == 1 ? 0 : __builtin_amdgcn_readfirstlane(t >> 6);
    const Ctx c = ctx_load<Block4>();
    Win g = ctx_win(c);
    RCN_G const int32_t* nr = (c.sub ? g.n2r_x : g.n2r).ptr();
    RCN_G const RowDesc* desc = g.desc.ptr();
    RCN_G const int32_t* e_nin = g.e_nin.ptr();
    RCN_G const int32_t* e_tail = g.e_tail.ptr();
    RCN_G const uint8_t* inc = g.inc.ptr();
    RCN_G uint32_t* __restrict__ H = reinterpret_cast<RCN_G uint32_t*>(g.H.ptr());
    RCN_G const int16_t* H16 = reinterpret_cast<RCN_G const int16_t*>(g.H.ptr());
    RCN_G const uint8_t* seq = gcast(c.seq);
    const int V = c.V, len = c.len;
    const bool sub = c.sub != 0;
    const int hs = c.hstride;                   // row stride in int16 cells (multiple of 512)
    const int hs2 = hs >> 1;                    // ... in packed dwords
    // WV > 1: the waves run FREE of barriers.  Wave w hands the border cell Z[i][last column of w] to wave w + 1
    // through a 64-entry LDS mailbox, one tagged word {row : 16 | value : 16} per row (a single 32-bit store, so the
    // reader either sees the old word or the complete new one), and every 8 rows it publishes how far it is, so that
    // the wave to its left never laps the mailbox.  Wave 0 depends on nobody.
    constexpr int kMail = WV > 1 ? 64 * 4 * 4 + 64 : 0;     // bytes: mailboxes [4][64] + progress words
    constexpr int kTab = TAB ? 4 * 4 * NTH * NP : 0;       // bytes of the profile table [4 symbols][64 lanes][NP]
    constexpr int KT = (kLdsBytes - 64 - kMail - kTab) / (4 * NTH * NP);   // LDS row slots: K ring rows + 1 staging slot
    static_assert(KT - 1 == dp2_ring_rows(NP, WV, TAB), "phase_desc2 classifies rows with the same ring depth");
    constexpr int K = KT - 1;
    uint32_t* ring = reinterpret_cast<uint32_t*>(Block4::work());   // [KT][NTH][NP]
    int* farb = Block4::work() + (kLdsBytes - 64) / 4;   // [4] staged border cell of a far predecessor row, per wave
    // (not `volatile`: the backend brackets volatile accesses with s_waitcnt vmcnt(0), i.e. a wait for the previous
    //  H-row store in every row; the polls below are inline-asm LDS reads instead)
    uint32_t* mail = reinterpret_cast<uint32_t*>(Block4::work() + (kLdsBytes - 64 - kMail) / 4);   // [4][64]
    uint32_t* prog = mail + 4 * 64;                                                                  // [4] rows finished, per wave
    const int col0 = t * 2 * NP;                // first column of this thread
    const int bcol = wv * 128 * NP - 1;         // column left of this wave's block (wv > 0)

    const int mg = c.m - c.gp, xg = c.x - c.gp;
    uint32_t MG = pack2(mg, mg), XM = pack2(xg - mg, xg - mg), ONE = 0x00010001u;
    const uint32_t GG = pack2(c.gp, c.gp);
    asm volatile("; constants live in VGPRs" : "+v"(MG), "+v"(XM), "+v"(ONE));
    uint32_t sqx[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int j0 = col0 + 2 * q, j1 = j0 + 1;
        const int s0 = (j0 >= 1 && j0 <= len) ? seq[j0 - 1] : 0x100, s1 = (j1 >= 1 && j1 <= len) ? seq[j1 - 1] : 0x100;
        sqx[q] = pack2(s0, s1);
    }
    uint32_t tie_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&Block4::ctx()->tie_rows[0]));
    asm volatile("" : "+s"(tie_base));
    uint32_t* ptab = ring + KT * NTH * NP;      // [4][NTH][NP] (TAB)
    if (TAB) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const uint32_t code = sl == 0 ? 'A' : sl == 1 ? 'C' : sl == 2 ? 'T' : 'G', symsym = code | (code << 16);
#pragma unroll
            for (int q = 0; q < NP; ++q) ptab[(sl * NTH + t) * NP + q] = pk_profile(sqx[q], symsym, ONE, XM, MG);
        }
    }
    // Register window: the last R rows of Z for this lane's columns, row r at win[(r % R) * NP + q].  The
    // index is wave-uniform, so a read or write is s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off.
    constexpr int R = dp2_window(NP);
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    u32x16 win = {};
    uint32_t prev[NP] = {};                      // the row just finished (what a chain row reads)
    int zsh = static_cast<int>(0x80000000u);    // lane l: scan value of lane l - 1; lane 0: max's identity, never overwritten
    uint32_t mpv = static_cast<uint32_t>(kNeg16) << 16;   // same for the diagonal shift (WV = 1: -inf left of column 0)
    // cwin (WV = 4): lane (r % 64) holds Z[r][bcol], the border cell this wave received as horizontal carry
    // of row r = the diagonal carry of predecessor row r (wave 0 has no left neighbour: -inf)
    int cwin = wv == 0 ? kNeg16 : 0;
    const int t_own = len / (2 * NP), own_wave = t_own >> 6, own_lane = t_own & 63, own_q = (len % (2 * NP)) >> 1, own_hi = len & 1;
    int best = 0, best_row = 0, have_best = 0, tied = 0;
    unsigned int pred_rows = 0, not_chain = 0;  // chain rows (one predecessor each) are counted as V - not_chain at the end
    int slot = 1 % K;                           // ring slot of row i is i % K
    int dl_p0 = 0, dl_p1 = 0, dl_p2 = 0, dl_p3 = 0, dl_p4 = 0, dl_p5 = 0, dl_er = -1, dl_meta = 1 << 9;

    // WV = 4, skewed pipeline: wave wv starts wv steps late and finishes wv steps late; every wave executes
    // exactly V + WV - 1 barriers.
    int seen_next = 0;                          // progress of wave wv + 1 as last read
    if (WV > 1) {
        for (int k = t; k < 4 * 64 + 4; k += NTH) mail[k] = 0u;        // tag 0 never matches: rows start at 1
        Block4::sync();
    }
#ifdef RCN_PROF_DP
    long long prof_row__ = 0, prof_bar__ = 0, tr0__ = clock64();
#endif
#pragma unroll 1
    for (int rbase = 0; rbase < V; rbase += 64) {
        {
            // 64 row descriptors per coalesced load, one per lane; read back with v_readlane
            RowDesc d; d.erest = -1; d.meta = 1 << 9;
#pragma unroll
            for (int q = 0; q < kInlinePreds; ++q) d.p[q] = 0;
            if (rbase + lane < V) d = desc[rbase + lane];
            dl_p0 = d.p[0]; dl_p1 = d.p[1]; dl_p2 = d.p[2]; dl_p3 = d.p[3]; dl_p4 = d.p[4]; dl_p5 = d.p[5]; dl_er = d.erest; dl_meta = d.meta;
            // an (empty) asm that consumes and redefines the eight registers: the compiler has to place its
            // s_waitcnt for the load in front of it, i.e. outside the row loop (a wait inside the row loop
            // would also wait for every outstanding H-row store, every row)
            asm volatile("; row descriptors retired" : "+v"(dl_p0), "+v"(dl_p1), "+v"(dl_p2), "+v"(dl_p3), "+v"(dl_p4), "+v"(dl_p5), "+v"(dl_er), "+v"(dl_meta));
        }
        const int rend = min(V, rbase + 64);
        // software pipeline: the descriptor word and the substitution profile of row r + 1 are produced while
        // row r is in its scan (v_readlane -> SALU has ~20 cycles of latency; the profile fills DPP wait states)
        int meta_next = __builtin_amdgcn_readlane(dl_meta, 0);
        uint32_t Pn[NP];
        if (TAB) {
            const uint32_t* src = ptab + ((((meta_next & 255) >> 1) & 3) * NTH + t) * NP;
#pragma unroll
            for (int q = 0; q < NP; ++q) Pn[q] = src[q];
        } else {
            const uint32_t sy = meta_next & 255, symsym = sy | (sy << 16);
#pragma unroll
            for (int q = 0; q < NP; ++q) Pn[q] = pk_profile(sqx[q], symsym, ONE, XM, MG);
        }
#pragma unroll 1
        for (int r = rbase; r < rend; ++r) {
            const int k = r - rbase;
            const int i = r + 1;
            // horizontal carry into this block: Z[i][bcol], finished by wave wv-1 one step ago.  Issued first,
            // consumed last (wave 0 reads its own slot and ignores it).
            uint32_t cin_raw = 0;
            if (WV > 1 && wv > 0) cin_raw = mail[(wv - 1) * 64 + (i & 63)];
            const int meta = meta_next;
            meta_next = __builtin_amdgcn_readlane(dl_meta, (k + 1) & 63);
            uint32_t P[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) P[q] = Pn[q];

            uint32_t M[NP];
            int mleft = kNeg16;                 // max over predecessors of Z[p][bcol] (diagonal carry into lane 0)
            if (meta & (1 << 15)) {
                // ---- chain row (most rows): the only predecessor is the row just finished, still in registers ----
#pragma unroll
                for (int q = 0; q < NP; ++q) M[q] = prev[q];
                if (WV > 1) mleft = __builtin_amdgcn_readlane(cwin, (i - 1) & 63);
            } else if (meta & (1 << 13)) {
                ++not_chain;
                // ---- fast row: predecessors come from the register window, their border cells from cwin ----
                const unsigned int dd = static_cast<unsigned int>(meta) >> 16;
                const int npf = (meta >> 9) & 7;
                {
                    const int d = dd & 15;
                    const int wi = ((i - d) & (R - 1)) * NP;
#pragma unroll
                    for (int q = 0; q < NP; ++q) M[q] = win[wi + q];
                    if (WV > 1) mleft = __builtin_amdgcn_readlane(cwin, (i - d) & 63);
                }
#pragma unroll 1
                for (int e = 1; e < npf; ++e) {
                    const int d = (dd >> (4 * e)) & 15;
                    const int wi = ((i - d) & (R - 1)) * NP;
#pragma unroll
                    for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], win[wi + q]);
                    if (WV > 1) mleft = max(mleft, __builtin_amdgcn_readlane(cwin, (i - d) & 63));
                }
                pred_rows += npf;
#ifdef RCN_PROF_CNT
                if (lane == 0) atomicAdd(&g_dbg[0], 1ull);
#endif
            } else if (meta & (1 << 14)) {
                // ---- medium row: every predecessor from the LDS ring, all reads in flight together ----
                const unsigned int dd = static_cast<unsigned int>(meta) >> 16;
                const int npf = (meta >> 9) & 7;
                uint32_t hp[4][NP];
                int ml[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = (dd >> (4 * (e < npf ? e : 0))) & 15;      // unused slots repeat predecessor 0
                    int sp = slot - d; if (sp < 0) sp += K;
                    const uint32_t* src = ring + (sp * NTH + t) * NP;
#pragma unroll
                    for (int q = 0; q < NP; ++q) hp[e][q] = src[q];
                    ml[e] = WV > 1 ? __builtin_amdgcn_readlane(cwin, (i - d) & 63) : kNeg16;
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) M[q] = pk_max(pk_max(hp[0][q], hp[1][q]), pk_max(hp[2][q], hp[3][q]));
                if (WV > 1) mleft = max(max(ml[0], ml[1]), max(ml[2], ml[3]));
                pred_rows += npf;
                ++not_chain;
            } else {
                ++not_chain;
#ifdef RCN_PROF_CNT
                if (lane == 0) { atomicAdd(&g_dbg[1], 1ull); if (meta & 256) atomicAdd(&g_dbg[2], 1ull); if (((meta >> 9) & 7) > 4) atomicAdd(&g_dbg[3], 1ull); }
#endif
                // ---- general row: any number of predecessors, LDS ring or (rare) HBM ----
                const int p0 = __builtin_amdgcn_readlane(dl_p0, k);
                const int er = __builtin_amdgcn_readlane(dl_er, k);
                const int np = (meta >> 9) & 7;
                bool first = true;
                auto combine = [&](int p) {
                    uint32_t hp[NP];
                    int bl = 0;
                    if (p == 0) {
#pragma unroll
                        for (int q = 0; q < NP; ++q) hp[q] = 0u;
                    } else if (i - p < K - 1) {     // LDS ring
                        int sp = slot - (i - p); if (sp < 0) sp += K;
                        const uint32_t* src = ring + (sp * NTH + t) * NP;
#pragma unroll
                        for (int q = 0; q < NP; ++q) hp[q] = src[q];
                        if (WV > 1 && wv > 0) bl = __builtin_amdgcn_readlane(cwin, p & 63);      // i - p < K <= 63
                    } else {
#ifdef RCN_PROF_CNT
                        if (lane == 0) atomicAdd(&g_dbg[4], 1ull);
#endif
                        // rare: older than the ring -> HBM, staged through the spare LDS slot so that the common
                        // path never has a global load pending at the join (its s_waitcnt vmcnt would also wait
                        // for every outstanding H-row store, every row)
                        uint32_t* sdst = ring + (K * NTH + t) * NP;
#pragma unroll
                        for (int q = 0; q < NP; ++q) sdst[q] = H[p * hs2 + t * NP + q];
                        if (WV > 1 && wv > 0 && lane == 0) farb[wv] = H16[p * hs + bcol];
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int q = 0; q < NP; ++q) hp[q] = sdst[q];
                        if (WV > 1 && wv > 0) bl = farb[wv];
                    }
                    if (wv == 0) bl = kNeg16;
                    if (first) {
#pragma unroll
                        for (int q = 0; q < NP; ++q) M[q] = hp[q];
                        mleft = bl; first = false;
                    } else {
#pragma unroll
                        for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], hp[q]);
                        mleft = max(mleft, bl);
                    }
                    ++pred_rows;
                };
                combine(p0);
                if (np > 1) {
                    const int q1 = __builtin_amdgcn_readlane(dl_p1, k), q2 = __builtin_amdgcn_readlane(dl_p2, k);
                    const int q3 = __builtin_amdgcn_readlane(dl_p3, k), q4 = __builtin_amdgcn_readlane(dl_p4, k);
                    const int q5 = __builtin_amdgcn_readlane(dl_p5, k);
#pragma unroll 1
                    for (int q = 1; q < np; ++q) combine(q == 1 ? q1 : q == 2 ? q2 : q == 3 ? q3 : q == 4 ? q4 : q5);
                }
                for (int e = er; e >= 0; e = e_nin[e]) {
                    const int tl = e_tail[e];
                    if (sub && !inc[tl]) continue;
                    combine(nr[tl] + 1);
                }
                // retire the LDS reads here: if their s_waitcnt moved to the join below, every chain / fast row would
                // wait there too -- for the acknowledgement of the previous row's ring write (an LDS round trip per row)
#pragma unroll
                for (int q = 0; q < NP; ++q) asm volatile("" : "+v"(M[q]));
            }

            // diagonal sources = the combined predecessor row shifted right by one column
            uint32_t mprev;
            if (WV > 1) mprev = __builtin_amdgcn_update_dpp(static_cast<uint32_t>(mleft) << 16, M[NP - 1], 0x138, 0xf, 0xf, false);
            else mprev = mpv = __builtin_amdgcn_update_dpp(mpv, M[NP - 1], 0x138, 0xf, 0xf, false);   // lane 0 keeps -inf (loop carried, as zsh below)
            uint32_t acc[NP];
            if (TAB) {
                // everything that does not need the profile first: P comes from LDS and its wait (the compiler makes it an
                // lgkmcnt(0), which also covers the previous row's ring write) should find the LDS queue drained
                uint32_t D[NP], U[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) { D[q] = __builtin_amdgcn_alignbit(M[q], q == 0 ? mprev : M[q - 1], 16); U[q] = pk_add(M[q], GG); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NP; ++q) acc[q] = pk_max(pk_add(D[q], P[q]), U[q]);
            } else {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const uint32_t D = __builtin_amdgcn_alignbit(M[q], q == 0 ? mprev : M[q - 1], 16);
                acc[q] = pk_max(pk_add(D, P[q]), pk_add(M[q], GG));
            }
            }
            // horizontal move (+0 in the Z domain): in-lane chain, wave-wide prefix max of the lane tails
            // (pairs first: NP independent ops; then NP - 1 dependent carries between the registers)
#pragma unroll
            for (int q = 0; q < NP; ++q) acc[q] = pk_chain_pair(acc[q]);
#pragma unroll
            for (int q = 1; q < NP; ++q) acc[q] = pk_max_bhi(acc[q], acc[q - 1]);
            // wave-wide exclusive prefix max of the lane tails, the next row's profile in the DPP wait states
            int sc = static_cast<int>(acc[NP - 1]) >> 16;
            {
                constexpr int I = static_cast<int>(0x80000000u);     // max's identity: each step is one v_max_i32_dpp
                const uint32_t sy = meta_next & 255;
                const uint32_t symsym = sy | (sy << 16);
                uint32_t pw[NP];
                if (TAB) {
                    // the next row's profile: one LDS read, issued in front of the scan, retired by the next row
                    const uint32_t* src = ptab + (((sy >> 1) & 3) * NTH + t) * NP;
#pragma unroll
                    for (int q = 0; q < NP; ++q) pw[q] = src[q];
                    sc = max(sc, dpp_or<0x111, 0xf>(I, sc));
                    sc = max(sc, dpp_or<0x112, 0xf>(I, sc));
                    sc = max(sc, dpp_or<0x114, 0xf>(I, sc));
                    sc = max(sc, dpp_or<0x118, 0xf>(I, sc));
                    sc = max(sc, dpp_or<0x142, 0xa>(I, sc));
                    sc = max(sc, dpp_or<0x143, 0xc>(I, sc));
                } else {
#define RCN_GAP(o) do { __builtin_amdgcn_sched_barrier(0); dp2_gap_op<NP, (o)>(pw, sqx, symsym, ONE, XM, MG); \
                        dp2_gap_op<NP, (o) + 1>(pw, sqx, symsym, ONE, XM, MG); __builtin_amdgcn_sched_barrier(0); } while (0)
                RCN_GAP(0);  sc = max(sc, dpp_or<0x111, 0xf>(I, sc));
                RCN_GAP(2);  sc = max(sc, dpp_or<0x112, 0xf>(I, sc));
                RCN_GAP(4);  sc = max(sc, dpp_or<0x114, 0xf>(I, sc));
                RCN_GAP(6);  sc = max(sc, dpp_or<0x118, 0xf>(I, sc));
                RCN_GAP(8);  sc = max(sc, dpp_or<0x142, 0xa>(I, sc));
                RCN_GAP(10); sc = max(sc, dpp_or<0x143, 0xc>(I, sc));
                __builtin_amdgcn_sched_barrier(0);
#undef RCN_GAP
                // a use inside this block: without it the profile instructions are sunk out of the gaps into the
                // blocks that consume them
#pragma unroll
                for (int q = 0; q < NP; ++q) asm volatile("" :: "v"(pw[q]));
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) Pn[q] = pw[q];
            }
            // lane 0 has no source lane and keeps `old`: zsh is loop carried, so its lane 0 stays at the identity it
            // was initialised with and no constant has to be rebuilt per row
            zsh = dpp_or<0x138, 0xf>(zsh, sc);
            int zex = zsh;
            int cin = static_cast<int>(0x80000000u);
            if (WV > 1 && wv > 0) {
                asm volatile("; carry consumed here" : "+v"(cin_raw));
                while (__builtin_amdgcn_readfirstlane(cin_raw >> 16) != static_cast<uint32_t>(i & 0xffff)) {
                    __builtin_amdgcn_s_sleep(1);
#ifdef RCN_PROF_DP
                    ++prof_bar__;
#endif
                    cin_raw = lds_poll(mail + (wv - 1) * 64 + (i & 63));
                }
                cin = static_cast<int>(static_cast<int16_t>(cin_raw & 0xffffu));
            }
            zex = max(max(zex, cin), kNeg16);
#pragma unroll
            for (int q = 0; q < NP; ++q) acc[q] = pk_max_blo(acc[q], static_cast<uint32_t>(zex));

            {
                RCN_G uint32_t* dst = H + i * hs2 + t * NP;       // every lane is inside the row: hstride is a multiple of 512
#pragma unroll
                for (int q = 0; q < NP; ++q) dst[q] = acc[q];
            }
            uint32_t* rdst = ring + (slot * NTH + t) * NP;
#pragma unroll
            for (int q = 0; q < NP; ++q) { rdst[q] = acc[q]; win[(i & (R - 1)) * NP + q] = acc[q]; prev[q] = acc[q]; }
            if (WV > 1 && wv > 0) cwin = (lane == (i & 63)) ? cin : cwin;
            slot = (slot + 1 == K) ? 0 : slot + 1;

            if (__builtin_expect((meta & ((1 << 13) | 256)) == 256 && wv == own_wave, 0)) {      // sink rows are never "fast"
                uint32_t fv = acc[0];
#pragma unroll
                for (int q = 1; q < NP; ++q) if (own_q == q) fv = acc[q];
                const int v16 = own_hi ? (static_cast<int>(fv) >> 16) : (static_cast<int>(fv << 16) >> 16);
                const int val = __builtin_amdgcn_readlane(v16, own_lane);
                if (!have_best || best < val) { have_best = 1; best = val; best_row = i; tied = 1; }
                else if (best == val) {
                    // (explicit LDS address from a base computed once: the backend would otherwise re-derive the dynamic-LDS
                    //  base here with an s_load_dword, and a scalar load anywhere in the loop turns every LDS wait of the
                    //  loop into lgkmcnt(0) -- i.e. a wait for the row's ring write at the top of the next row)
                    if (tied < 8 && lane == 0) *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(tie_base + 4u * tied) = static_cast<uint32_t>(i);
                    ++tied;
                }
            }
            if (WV > 1) {
                if (wv < WV - 1) {
                    // never lap the mailbox of the wave to the right: it must have consumed row i - 64 before row i
                    // is posted (progress is published every 8 rows, so stay within 48)
                    while (i - seen_next > 48) {
                        seen_next = static_cast<int>(__builtin_amdgcn_readfirstlane(lds_poll(prog + wv + 1)));
                        if (i - seen_next > 48) __builtin_amdgcn_s_sleep(2);
                    }
                    if (lane == 63) mail[wv * 64 + (i & 63)] = (static_cast<uint32_t>(i) << 16) | (acc[NP - 1] >> 16);
                }
                if (wv > 0 && (i & 7) == 0 && lane == 0) prog[wv] = i;
#ifdef RCN_PROF_DP
                const long long tb1__ = clock64(); prof_row__ += tb1__ - tr0__; tr0__ = tb1__;
#endif
            } else {
                // one wave: LDS accesses of a wave execute in order, nothing to wait for
#ifdef RCN_PROF_DP
                const long long tb1__ = clock64(); prof_row__ += tb1__ - tr0__; tr0__ = tb1__;
#endif
            }
        }
    }
#ifdef RCN_PROF_DP
    if (lane == 0) { atomicAdd(&g_prof_out[wv * 2], (unsigned long long)prof_row__); atomicAdd(&g_prof_out[wv * 2 + 1], (unsigned long long)prof_bar__); }
#endif
    Ctx* o = Block4::ctx();
    if (wv == own_wave && lane == 0) { o->best = best; o->best_row = best_row; o->tied = tied; }
    if (t == 0) {
        const int W = len + 1;
        pred_rows += static_cast<unsigned int>(V) - not_chain;
        o->pred_rows = pred_rows;
        o->cells += static_cast<unsigned long long>(V + 1) * W;
        o->pred += static_cast<unsigned long long>(pred_rows) * W;
        // SURVEY 8(d) yardstick (same formula as poa_window_kernel): cells written once + predecessor rows
        // read once per in-edge, at 2 B/cell when the worst-case score bound fits int16, else 4 B/cell
        const int amax = max(max(abs(c.m), abs(c.x)), abs(c.gp));
        const unsigned long long sbytes = (static_cast<long long>(amax) * (V + W) < 32767) ? 2ull : 4ull;
        o->bytes += sbytes * (static_cast<unsigned long long>(V + 1) + pred_rows) * W;
        o->cells_full += static_cast<unsigned long long>(V + 1) * W;
        o->bytes_full += sbytes * (static_cast<unsigned long long>(V + 1) + pred_rows) * W;
    }
    if (WV > 1) Block4::sync(); else Wave0Of4::sync();
}

}  // namespace rcn
#include "poa_band.hpp"
namespace rcn {

// ---- phase: sink tie-break (rare) + traceback over int16 Z tiles ----
// Tile of the finished matrix staged in LDS for the walk: kTbRows consecutive DP rows x 64 columns.  The path climbs
// ~3.4 rows per column on a 30x graph, so rows, not columns, are what a tile runs out of: 112 x 64 instead of 64 x 128
// costs the same LDS and halves the number of stagings (each one is an HBM round trip plus two work-group barriers on
// the window's serial chain).  One 64-lane x 4 B global_load_lds moves 256 B = two rows of 64 cells, which land
// contiguously: row pair p at p * kTile2Pair cells, its odd row 64 cells further.
constexpr int kTbRows = 112;
constexpr int kTile2Cols = 64;         // int16 cells per tile row
constexpr int kTile2Pair = 136;        // LDS stride of a row PAIR in cells (272 B: 4 banks of skew per pair)
__device__ __forceinline__ int tile_at(int trow, int tcol) { return (trow >> 1) * kTile2Pair + (trow & 1) * kTile2Cols + tcol; }
static_assert((kTbRows / 2) * kTile2Pair * 2 + kTbRows * 32 + 68 + 64 * 4 <= kLdsBytes, "tile + row descriptors + sequence slice + the tile's pos_t must fit");

__device__ __forceinline__ void traceback2_slow_step(Win& g, RCN_G const int32_t* nr, bool sub, RCN_G const uint8_t* seq,
                                                     int m, int x, int gp, int& i, int& j, int& n) {
    const int64_t hs = g.hstride;
    RCN_G const int16_t* H = reinterpret_cast<RCN_G const int16_t*>(g.H.ptr());
    const int hij = H[i * hs + j];
    int pi = 0, pj = 0; bool found = false;
    if (i != 0) {
        const RowDesc d = g.desc[i - 1];
        const int np = (d.meta >> 9) & 7;
        for (int pass = (j != 0 ? 0 : 1); pass < 2 && !found; ++pass) {
            const int col = pass == 0 ? j - 1 : j;
            const int add = pass == 0 ? ((((d.meta & 255) == seq[j - 1]) ? m : x) - gp) : gp;
            for (int q = 0; q < np && !found; ++q) {
                if (hij == H[d.p[q] * hs + col] + add) { pi = d.p[q]; pj = col; found = true; }
            }
            for (int e = d.erest; e >= 0 && !found; e = g.e_nin[e]) {
                const int tl = g.e_tail[e];
                if (sub && !g.inc[tl]) continue;
                const int p = nr[tl] + 1;
                if (hij == H[p * hs + col] + add) { pi = p; pj = col; found = true; }
            }
        }
    }
    if (!found) {
        if (j == 0) { g.overflow = 4; i = 0; j = 0; return; }
        pi = i; pj = j - 1;
    }
    g.path_node[n] = (i == pi) ? -1 : i;
    g.path_pos[n] = (j == pj) ? -1 : j - 1;
    ++n; i = pi; j = pj;
}

// ---- phase: AddAlignment over 256 threads (window.cpp:110-119) ----
// Same per-position phases as phase_add<> of poa_kernel.hpp (a global alignment consumes every sequence
// position exactly once, so positions are independent up to the node / edge numbering and the order
// anchors), but 256 positions per step: the prefix count / prefix max across the four waves goes through
// eight LDS words.  Four times fewer dependent HBM round trips on the critical path.
__device__ __forceinline__ void block4_scan(int* xch, int wv, int lane, int cnt, int wmax, int& off, int& total, int& pmax, int& tmax) {
    if (lane == 0) { xch[wv] = cnt; xch[4 + wv] = wmax; }
    Block4::sync();
    off = 0; total = 0; pmax = -1; tmax = -1;
#pragma unroll
    for (int w = 0; w < kWaves2; ++w) {
        const int cw = xch[w], mw = xch[4 + w];
        if (w < wv) { off += cw; pmax = max(pmax, mw); }
        total += cw; tmax = max(tmax, mw);
    }
    Block4::sync();
}

__device__ __noinline__ void phase_add4() {
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const Ctx c = ctx_load<Block4>();
    Win g = ctx_win(c);
    RCN_G const int32_t* rank = c.sub ? g.rank_sub.ptr() : g.rank_full.ptr();
    RCN_G const uint8_t* seq = gcast(c.seq); RCN_G const uint8_t* qual = gcast(c.qual);
    const int len = c.len, n_old = g.n_nodes, ring = g.ring;
    const uint32_t count = len >= 2 ? 1u : 0u;
    int* xch = Block4::work();
    constexpr int U = 2;                        // positions per thread walked in lock step (loads in flight together)
    constexpr int RM = 4;                       // aligned-ring members looked at in lock step (more: generic loop)
    RCN_G int32_t* kindv = g.path_pos.ptr();
    RCN_G int32_t* idxv = g.path_node.ptr();
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nn = 0, anchor = -1;
    // classify positions (existing node / new node / new node joining a ring); number the new nodes (prefix count)
    // and propagate order anchors (prefix max).  The anchor of a position on an existing node is the last rank of
    // that node's ring block, the same for every member of the ring.
    for (int base = 0; base < len; base += U * kThreads2) {
        int pos[U], tt[U], ch[U], ct[U], na[U], ra[U], mem[U][RM], mc[U][RM], mr[U][RM], kind[U], curr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pos[u] = base + u * kThreads2 + t;
            const int row = pos[u] < len ? g.pos_t[pos[u]] : 0;           // the traceback left DP rows (-1 / 0 = none)
            ch[u] = pos[u] < len ? seq[pos[u]] : 0;
            tt[u] = row <= 0 ? -1 : row;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) tt[u] = tt[u] < 0 ? -1 : rank[tt[u] - 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = tt[u] < 0 ? 0 : tt[u];
            ct[u] = g.code[v]; na[u] = tt[u] < 0 ? 0 : g.al_cnt[v]; ra[u] = tt[u] < 0 ? -1 : g.n2r[v];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int a2 = 0; a2 < RM; ++a2) mem[u][a2] = a2 < na[u] ? g.al_nodes[tt[u] * ring + a2] : -1;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int a2 = 0; a2 < RM; ++a2) { const int m = mem[u][a2] < 0 ? 0 : mem[u][a2]; mc[u][a2] = g.code[m]; mr[u][a2] = mem[u][a2] < 0 ? -1 : g.n2r[m]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kind[u] = 0; curr[u] = -1;
            if (pos[u] < len) {
                if (tt[u] < 0) { kind[u] = 1; ra[u] = -1; }
                else {
                    int found = ct[u] == ch[u] ? tt[u] : -1;
#pragma unroll
                    for (int a2 = 0; a2 < RM; ++a2) {
                        if (a2 < na[u]) { ra[u] = max(ra[u], mr[u][a2]); if (found < 0 && mc[u][a2] == ch[u]) found = mem[u][a2]; }
                    }
                    for (int a2 = RM; a2 < na[u]; ++a2) {                 // rings beyond four members (IUPAC-rich input)
                        const int m = g.al_nodes[tt[u] * ring + a2];
                        ra[u] = max(ra[u], g.n2r[m]);
                        if (found < 0 && g.code[m] == ch[u]) found = m;
                    }
                    curr[u] = found; kind[u] = found >= 0 ? 0 : 2;
                }
                g.pos_t[pos[u]] = tt[u]; g.pos_curr[pos[u]] = curr[u];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int a = pos[u] < len ? ra[u] : -1;
            const unsigned long long mk = __ballot(kind[u] != 0);
            const int la = wave_incl_scan_max(a);
            int off, total, pmax, tmax;
            block4_scan(xch, wv, lane, __popcll(mk), __builtin_amdgcn_readlane(la, 63), off, total, pmax, tmax);
            if (pos[u] < len) { kindv[pos[u]] = kind[u]; idxv[pos[u]] = nn + off + __popcll(mk & lt); g.pos_a[pos[u]] = max(max(la, pmax), anchor); }
            nn += total; anchor = max(anchor, tmax);
        }
    }
    int overflow = g.overflow;
    if (n_old + nn > g.ncap) overflow = 1;
    Block4::sync();
    int n_edges = g.n_edges;
    if (!overflow) {
        for (int pos = t; pos < len; pos += kThreads2) {
            const int kind = kindv[pos];
            if (kind) {
                const int idx = idxv[pos];
                addp_create(g, seq, pos, kind, n_old + idx, count);
                g.new_id[idx] = n_old + idx; g.new_anchor[idx] = g.pos_a[pos];
            }
        }
        g.n_nodes = n_old + nn;
        Block4::sync();
        // edges pos-1 -> pos: reinforce an existing one or create it; the out-lists of U positions are walked in lock step
        for (int base = 0; base < len; base += U * kThreads2) {
            int pos[U], tail[U], head[U], e[U], f[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                pos[u] = base + u * kThreads2 + t;
                const bool act = pos[u] >= 1 && pos[u] < len;
                tail[u] = act ? g.pos_curr[pos[u] - 1] : -1; head[u] = act ? g.pos_curr[pos[u]] : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { e[u] = tail[u] >= 0 ? g.out_head[tail[u]] : -1; f[u] = tail[u] >= 0 ? 1 : 0; }
            for (;;) {
                bool any = false;
                int eh[U], en[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { eh[u] = e[u] >= 0 ? g.e_head[e[u]] : -2; en[u] = e[u] >= 0 ? g.e_nout[e[u]] : -1; }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (e[u] >= 0) {
                        if (eh[u] == head[u]) { g.e_w[e[u]] += pair_weight(qual, pos[u]); f[u] = 0; e[u] = -1; }
                        else e[u] = en[u];
                    }
                    any = any || e[u] >= 0;
                }
                if (!any) break;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned long long mk = __ballot(f[u] != 0);
                int off, total, pmax, tmax;
                block4_scan(xch, wv, lane, __popcll(mk), 0, off, total, pmax, tmax);
                const int ne = n_edges + off + __popcll(mk & lt);
                if (f[u] && ne < g.ecap) addp_edge_create(g, qual, pos[u], ne);
                n_edges += total;
            }
        }
        if (n_edges > g.ecap) { overflow = 1; n_edges = g.ecap; }
        for (int pos = t; pos < len; pos += kThreads2) g.cov[g.pos_curr[pos]] += count;
    }
    if (t == 0) {
        Ctx* o = Block4::ctx();
        o->n_old = n_old; o->nn = nn; o->n_nodes = overflow ? n_old : n_old + nn; o->n_edges = n_edges; o->overflow = overflow;
    }
    Block4::sync();
}

// ---- phase: order merge over 256 threads: insert the nn new nodes behind their anchors ----
__device__ __noinline__ void phase_merge4() {
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const Ctx c = ctx_load<Block4>();
    Win g = ctx_win(c);
    const int n_old = c.n_old, nn = c.nn;
    int* xch = Block4::work();
    RCN_G int32_t* delta = g.pred.ptr();                // [n_old + 1] scratch (pred is consensus-only)
    for (int r = t; r <= n_old; r += kThreads2) delta[r] = 0;
    Block4::sync();
    for (int k = t; k < nn; k += kThreads2) {
        const int a = g.new_anchor[k] + 1;
        atomicAdd((int*)&delta[a], 1);
        const int v = g.new_id[k];
        g.rank_tmp[a + k] = v; g.n2r[v] = a + k;
    }
    Block4::sync();
    int carry = 0;
    for (int base = 0; base < n_old; base += kThreads2) {
        const int r = base + t;
        int sc = r < n_old ? delta[r] : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(sc, d); if (lane >= d) sc += u; }
        int off, total, pmax, tmax;
        block4_scan(xch, wv, lane, __shfl(sc, 63), 0, off, total, pmax, tmax);
        if (r < n_old) { const int v = g.rank_full[r]; const int pos = r + carry + off + sc; g.rank_tmp[pos] = v; g.n2r[v] = pos; }
        carry += total;
    }
    if (t == 0) Block4::ctx()->swapped = c.swapped ^ 1;
    Block4::sync();
}

// ---- phase: several sinks share the best score (window.cpp:95-97 -> spoa's end cell) ----
// spoa takes the first of them in ITS rank order, the exact DFS post-order of Graph::TopologicalSort whose
// start nodes go in id order.  Three levels, cheapest first:
//  (1) rule: the DFS runs the backbone ids 0..L-1 first, so everything in the "backbone closure" (ancestors of
//      backbone nodes and their aligned rings) is appended before anything else, ring of backbone node b_p at
//      start p as (b_p, aligned list of b_p = ascending id).  A sink without aligned nodes and id >= L is in
//      nobody's closure: it is appended exactly when the start loop reaches its own id.  Hence the key
//      (p, id) for sinks whose ring holds a backbone node, (inf, id) for lone non-backbone sinks.
//  (2) the tied sinks include rings of non-backbone nodes: mark the backbone closure (= Subgraph(0, L-1), the
//      parallel sweep) as done and run the exact DFS only over the few nodes outside it.
//  (3) otherwise the full exact DFS.
// Result in ctx->best_row.  ctx->tb_n: 0 = done, 1 = level 2 wanted, 2 = level 3 wanted.
__device__ __noinline__ void phase_sink_tie_rule() {
    const Ctx c = ctx_load<Wave0Of4>();
    Win g = ctx_win(c);
    Ctx* o = Wave0Of4::ctx();
    if (threadIdx.x != 0) return;
    const bool sub = c.sub != 0;
    RCN_G const int32_t* rank = sub ? g.rank_sub.ptr() : g.rank_full.ptr();
    RCN_G const int32_t* nr = (sub ? g.n2r_x : g.n2r).ptr();
    int status = 2;
    o->tie_why = 1;                       // more than 8 tied
    if (c.tied <= 8) {
        bool classified = true;
        long long bestkey = 0x7fffffffffffffffll; int pick = -1;
        for (int k = 0; k < c.tied; ++k) {
            const int v = rank[(k == 0 ? c.best_row : o->tie_rows[k]) - 1];
            const int na = g.al_cnt[v];
            int rm = v;
            for (int a = 0; a < na; ++a) rm = min(rm, g.al_nodes[v * g.ring + a]);
            long long key;
            if (rm < c.bblen) key = (static_cast<long long>(rm) << 32) | static_cast<unsigned int>(v);
            else if (na == 0) key = (0x7ffffffell << 32) | static_cast<unsigned int>(v);
            else { classified = false; break; }
            if (key < bestkey) { bestkey = key; pick = v; }
        }
        if (classified) { o->best_row = nr[pick] + 1; status = 0; }
        else if (g.n_nodes <= kSubMaxNodes) status = 1;
        else o->tie_why = 2;
    }
    o->tb_n = status;
}

// level 2a (t == 0): p(v) = the backbone start whose DFS appends tied sink v = the smallest backbone id that is
// forward-reachable from v over out-edges and aligned links (search stops at backbone nodes: a small bubble).
// Leaves in ctx: tb_i = p* (smallest p, 0x7fffffff = none is in the backbone closure), tb_n = 0 when a single
// sink has p* (best_row set), 3 when a local DFS has to decide, 2 for the full DFS.
__device__ __noinline__ void phase_sink_tie_starts() {
    const Ctx c = ctx_load<Wave0Of4>();
    Win g = ctx_win(c);
    Ctx* o = Wave0Of4::ctx();
    if (threadIdx.x != 0) return;
    const bool sub = c.sub != 0;
    RCN_G const int32_t* rank = sub ? g.rank_sub.ptr() : g.rank_full.ptr();
    RCN_G const int32_t* nr = (sub ? g.n2r_x : g.n2r).ptr();
    RCN_G int32_t* stack = g.stack.ptr();          // [0, 256): visited list, [256, ...): work stack
    constexpr int kInf = 0x7fffffff;
    int ps[8], vs[8];
    bool ok = true;
    for (int k = 0; k < c.tied && ok; ++k) {
        const int v = rank[(k == 0 ? c.best_row : o->tie_rows[k]) - 1];
        vs[k] = v;
        int nvis = 0, sp = 256, pmin = kInf;
        stack[sp++] = v;
        while (sp > 256 && ok) {
            const int x = stack[--sp];
            bool seen = false;
            for (int q = 0; q < nvis; ++q) seen = seen || stack[q] == x;
            if (seen) continue;
            if (nvis == 256) { ok = false; break; }
            stack[nvis++] = x;
            if (x < c.bblen) { pmin = min(pmin, x); continue; }            // a backbone node: later ones only give larger p
            for (int e = g.out_head[x]; e >= 0; e = g.e_nout[e]) { const int h = g.e_head[e]; if (!sub || g.inc[h]) stack[sp++] = h; }
            const int na = g.al_cnt[x];
            for (int a = 0; a < na; ++a) { const int u = g.al_nodes[x * g.ring + a]; if (!sub || g.inc[u]) stack[sp++] = u; }
        }
        ps[k] = pmin;
    }
    int status = 2;
    o->tie_why = 3;                       // bubble too large
    if (ok) {
        int pstar = kInf, cnt = 0, who = -1;
        for (int k = 0; k < c.tied; ++k) pstar = min(pstar, ps[k]);
        for (int k = 0; k < c.tied; ++k) if (ps[k] == pstar) { ++cnt; who = vs[k]; }
        o->tb_i = pstar;
        if (cnt == 1) { o->best_row = nr[who] + 1; status = 0; }
        else status = 3;
        // the local DFS only has to look at the sinks that share p*
        int m = 0;
        for (int k = 0; k < c.tied; ++k) if (ps[k] == pstar) stack[512 + m++] = vs[k];
        stack[511] = m;
    }
    o->tb_n = status;
}

// level 2b, after the closure sweep has preset the DFS marks: spoa's DFS (same code as graph_toposort) from the
// single start b_p* -- or, when no tied sink is in the backbone closure, from the ids >= L in order -- until
// one of the candidates is appended.
__device__ __noinline__ void phase_sink_tie_local() {
    const Ctx c = ctx_load<Wave0Of4>();
    Win g = ctx_win(c);
    Ctx* o = Wave0Of4::ctx();
    if (threadIdx.x != 0) return;
    const bool sub = c.sub != 0;
    RCN_G const int32_t* nr = (sub ? g.n2r_x : g.n2r).ptr();
    RCN_G int32_t* stack = g.stack.ptr();
    const int ncand = stack[511];
    int cand[8];
    for (int k = 0; k < ncand; ++k) cand[k] = stack[512 + k];
    const int n = g.n_nodes, pstar = c.tb_i;
    const int s_lo = pstar == 0x7fffffff ? c.bblen : pstar, s_hi = pstar == 0x7fffffff ? n : pstar + 1;
    int winner = -1;
    for (int s = s_lo; s < s_hi && winner < 0; ++s) {
        if (sub && !g.inc[s]) continue;
        if ((g.mark[s] & 3) != 0) continue;
        int sp = 1024;
        stack[sp++] = s;
        while (sp > 1024 && winner < 0) {
            const int cur = stack[sp - 1];
            bool valid = true;
            const uint8_t mc = g.mark[cur];
            if ((mc & 3) != 2) {
                for (int e = g.in_head[cur]; e >= 0; e = g.e_nin[e]) {
                    const int tl = g.e_tail[e];
                    if (sub && !g.inc[tl]) continue;
                    if ((g.mark[tl] & 3) != 2) { stack[sp++] = tl; valid = false; }
                }
                const bool ign = (mc & 4) != 0;
                const int na = g.al_cnt[cur];
                if (!ign) {
                    for (int a = 0; a < na; ++a) {
                        const int u = g.al_nodes[cur * g.ring + a];
                        if (sub && !g.inc[u]) continue;
                        if ((g.mark[u] & 3) != 2) { stack[sp++] = u; g.mark[u] |= 4; valid = false; }
                    }
                }
                if (valid) {
                    g.mark[cur] = (mc & 4) | 2;
                    if (!ign) {
                        // appended now: cur, then its aligned nodes in list order
                        for (int k = 0; k < ncand && winner < 0; ++k) if (cand[k] == cur) winner = cur;
                        for (int a = 0; a < na && winner < 0; ++a) {
                            const int u = g.al_nodes[cur * g.ring + a];
                            if (sub && !g.inc[u]) continue;
                            for (int k = 0; k < ncand && winner < 0; ++k) if (cand[k] == u) winner = u;
                        }
                    }
                } else {
                    g.mark[cur] = (mc & 4) | 1;
                }
            }
            if (valid) --sp;
        }
    }
    if (winner >= 0) { o->best_row = nr[winner] + 1; o->tb_n = 0; } else { o->tb_n = 2; o->tie_why = 4; }
}

// level 3
__device__ __noinline__ void phase_sink_tie_full() {
    const Ctx c = ctx_load<Wave0Of4>();
    Win g = ctx_win(c);
    Ctx* o = Wave0Of4::ctx();
    if (threadIdx.x != 0) return;
    const bool sub = c.sub != 0;
    RCN_G const int32_t* nr = (sub ? g.n2r_x : g.n2r).ptr();
    RCN_G const int16_t* H = reinterpret_cast<RCN_G const int16_t*>(g.H.ptr());
    const int64_t hs = g.hstride;
    const int nx = graph_toposort(g, g.rank_x.ptr(), sub, g.stack.ptr());
    for (int r = 0; r < nx; ++r) {
        const int row = nr[g.rank_x[r]] + 1;
        const int zend = c.coded ? g.path_node[row] : H[static_cast<int64_t>(row) * hs + c.len];    // coded: the DP kept the sinks' end scores
        if ((g.desc[row - 1].meta & 256) && zend == c.best) { o->best_row = row; break; }
    }
    o->ties += 1;
#ifdef RCN_PROF_WIN
    printf("[tie3] why %d tied %d sub %d n %d V %d pstar %d\n", o->tie_why, c.tied, c.sub, g.n_nodes, c.V, c.tb_i);
#endif
}

// ---- phase: traceback, box walker ----
// Same decisions as phase_traceback2 (spoa priority diag > vertical > horizontal, predecessors in in-edge
// order) but organised around what a single wave is good at: the 64 lanes evaluate, in parallel, the move
// of every cell of an 8-row x 8-column box below/left of the current cell from the staged int16 Z tile
// (two LDS round trips per box), and the walk inside the box then costs one v_readlane per step instead of
// LDS round trips and ballots.  Output: pos_t[pos] = DP row aligned to sequence position pos, or -1.
constexpr int kMvDiag = 0, kMvUp = 1, kMvLeft = 2, kMvInvalid = 3;
#ifndef RCN_BOX_ROWS
#define RCN_BOX_ROWS 8
#define RCN_BOX_COLS 8
#endif
constexpr int kBoxRows = RCN_BOX_ROWS, kBoxCols = RCN_BOX_COLS;   // <= 64 cells; the path drops ~1.7 rows per column on a 30x graph
// The box is a parallelogram: its column b (b columns left of the anchor) holds the kBoxRows rows from kBoxSkew * b rows above
// the anchor's row upwards.  A diagonal step leaves its row for a predecessor, at least one row up, so with skew 1 only the
// rows the path climbs BEYOND one per column count against the box's height: measured (profiles/r02, exit statistics of the
// profiling build) a straight 9 x 7 box was left after 5 steps, in 61 % of the cases because the path had climbed 9 rows;
// the skewed one is left because its columns are used up (69 %), and 8 x 8 holds one column more (boxes per alignment 114 -> ~90).
#ifndef RCN_BOX_SKEW
#define RCN_BOX_SKEW 1
#endif
constexpr int kBoxSkew = RCN_BOX_SKEW;
constexpr int kNxExit = 64, kNxInvalid = 65;

__device__ __noinline__ void phase_traceback3() {
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const Ctx c = ctx_load<Block4>();
    Win g = ctx_win(c);
    RCN_G const int32_t* nr = (c.sub ? g.n2r_x : g.n2r).ptr();
    const bool sub = c.sub != 0;
    RCN_G const uint8_t* __restrict__ seq = gcast(c.seq);
    const int len = c.len, m = c.m, x = c.x, gp = c.gp;
    const int64_t hs = g.hstride;
    RCN_G const int16_t* __restrict__ H = reinterpret_cast<RCN_G const int16_t*>(g.H.ptr());
    Ctx* o = Block4::ctx();
    if (t == 0) {
        int best_row = c.best_row;
        o->tb_i = best_row; o->tb_j = len; o->tb_n = 0;
    }
    Block4::sync();

    int16_t* tile = reinterpret_cast<int16_t*>(Block4::work());                       // [kTbRows / 2][kTile2Pair]
    int* tdesc = Block4::work() + (kTbRows / 2) * kTile2Pair / 2;                      // kTbRows x RowDesc (8 ints each)
    uint8_t* tseq = reinterpret_cast<uint8_t*>(tdesc + kTbRows * (sizeof(RowDesc) / 4));   // seq[c0 - 1 + k], k in [0, 64]
    int* tpos = reinterpret_cast<int*>(tseq + 68);                                          // pos_t of the tile's 64 columns, flushed once per tile
    RCN_G int32_t* __restrict__ prow = g.pos_t.ptr();
    int i = bcast0(o->tb_i), j = bcast0(o->tb_j);
    int overflow = g.overflow;
    while (!(i == 0 && j == 0)) {
        // ---- stage the tile: rows [i - kTbRows + 1, i] (tile row r holds matrix row i - r), cols [c0, c0 + 63] ----
#ifdef RCN_PROF_DP
        const long long tp0__ = clock64();
#endif
        const int ti0 = i, j_stage = j;
#ifdef RCN_PROF_WIN
        if (t == 0) o->dbg_tiles += 1;
#endif
        int c0 = (j - 56) & ~7; if (c0 < 0) c0 = 0;
        const int rmin = ti0 - (kTbRows - 1) > 0 ? ti0 - (kTbRows - 1) : 0;
        {
            typedef __attribute__((address_space(3))) void* lds_ptr;
            constexpr int kPairsPerWave = kTbRows / 2 / kWaves2;       // 14
            static_assert(kPairsPerWave * kWaves2 * 2 == kTbRows, "rows split evenly over the waves, two per load");
#pragma unroll
            for (int kk = 0; kk < kPairsPerWave; ++kk) {
                const int pr = kPairsPerWave * wv + kk;           // row pair: tile rows 2 pr (lanes 0-31) and 2 pr + 1 (lanes 32-63)
                int r = ti0 - (2 * pr + (lane >> 5)); if (r < 0) r = 0;
                RCN_G const int16_t* src = H + r * hs + c0 + (lane & 31) * 2;
                __builtin_amdgcn_global_load_lds(src, (lds_ptr)(tile + pr * kTile2Pair), 4, 0, 0);
            }
            if (wv == 1 || wv == 2) {
                const int k = (wv - 1) * 64 + lane;               // tile row whose descriptor this lane stages
                if (k < kTbRows) {
                    const int r = ti0 - k;
                    int4 d0 = make_int4(0, -1, -1, -1), d1 = make_int4(-1, -1, -1, 1 << 9);
                    if (r >= 1) { RCN_G const int4* dsrc = reinterpret_cast<RCN_G const int4*>(g.desc.ptr() + (r - 1)); d0 = dsrc[0]; d1 = dsrc[1]; }
                    int4* ddst = reinterpret_cast<int4*>(tdesc + k * 8);
                    ddst[0] = d0; ddst[1] = d1;
                }
            } else if (wv == 3) {
                const int sc = c0 - 1 + lane;
                tseq[lane] = (sc >= 0 && sc < len) ? seq[sc] : 0;
                if (lane == 0) { const int s2 = c0 - 1 + 64; tseq[64] = (s2 >= 0 && s2 < len) ? seq[s2] : 0; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef RCN_PROF_DP
        const long long tp1__ = clock64();
        int nbox__ = 0;
#endif
        if (wv == 0) {
            // box = kBoxRows x kBoxCols cells below/left of the current cell, one per lane: (i - a, j - b)
            const int a = lane / kBoxCols, b = lane % kBoxCols;
            for (;;) {
                if (c.tie_pad[2]) break;                 // test switch (KParams::force_slow_tb): no box walk at all
                if (i == 0 && j == 0) break;
#ifdef RCN_PROF_DP
                ++nbox__;
#endif
#ifdef RCN_PROF_WIN
                if (lane == 0) o->dbg_boxes += 1;
#endif
                // ---- move of every cell of the box anchored at (i, j): all LDS reads first, compares after ----
                const int ii = i - a - kBoxSkew * b, jj = j - b;
                const bool inside = a < kBoxRows && ii >= rmin && ii >= 0 && jj >= c0 && jj >= 0 && !(jj > 0 && jj - 1 < c0);
                const int trow = inside ? ti0 - ii : 0, tcol = inside ? jj - c0 : 1;
                const int* dr = tdesc + trow * 8;
                const int4 pa = *reinterpret_cast<const int4*>(dr);
                const int4 pb = *reinterpret_cast<const int4*>(dr + 4);
                const int hij = tile[tile_at(trow, tcol)];
                const int symc = tseq[tcol];                                    // seq[jj - 1]
                const int meta = pb.w, erest = pb.z;
                const int np = (meta >> 9) & 7;
                const int pq[6] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y};
                bool ok = inside && erest < 0;
                const int mc = (((meta & 255) == symc) ? m : x) - gp;
                // first match in spoa's order (diagonal over the in-edges, then vertical over the in-edges, then
                // horizontal), branch-free per cell: later q first, earlier q overrides.  In-edges are looked
                // at in pairs, the later pairs only if some cell of the box has that many.
                int dlD = 0, dlU = 0, fD = 0, fU = 0;
                const int colok = jj > 0;
                const int npb = inside ? np : 0;
                auto look = [&](int q) {
                    const int useq = q < npb;
                    if (useq && pq[q] < rmin) ok = false;
                    const int16_t* zp = tile + tile_at((useq && pq[q] >= rmin) ? ti0 - pq[q] : 0, tcol);
                    const int hdq = zp[tcol > 0 ? -1 : 0], huq = zp[0];
                    const int isd = useq & colok & (hij == hdq + mc);
                    const int isu = useq & (hij == huq + gp);
                    dlD = isd ? ii - pq[q] : dlD; fD |= isd;
                    dlU = isu ? ii - pq[q] : dlU; fU |= isu;
                };
                if (__ballot(npb > 4)) { look(5); look(4); }
                if (__ballot(npb > 2)) { look(3); look(2); }
                look(1); look(0);
                int mv = fD ? kMvDiag : (fU ? kMvUp : (colok ? kMvLeft : kMvInvalid));
                int dl = fD ? dlD : dlU;
                if (ii == 0) { mv = colok ? kMvLeft : kMvInvalid; dl = 0; }
                if (!ok) mv = kMvInvalid;
                // successor of this cell: a lane of the box, or one of the exits
                const int ni = ii - (mv == kMvLeft ? 0 : dl), nj = jj - (mv == kMvUp ? 0 : 1);
                const int nb = j - nj, na = i - ni - kBoxSkew * nb;
                int nx;
                if (mv == kMvInvalid) nx = kNxInvalid;
                else if (ni == 0 && nj == 0) nx = kNxExit;
                else if (na < 0 || na >= kBoxRows || nb >= kBoxCols) nx = kNxExit;
                else nx = na * kBoxCols + nb;
                // ---- walk: one v_readlane per step ----
                int idx = 0, nxt;
                unsigned long long vis = 0ull;
#define RCN_WALK_STEP { nxt = __builtin_amdgcn_readlane(nx, idx); if (nxt >= 64) goto walk3_done; asm("s_bitset1_b64 %0, %1" : "+s"(vis) : "s"(idx)); idx = nxt; }
                for (;;) { RCN_WALK_STEP RCN_WALK_STEP RCN_WALK_STEP RCN_WALK_STEP }      // four steps per back-edge; see phase_traceback_code
#undef RCN_WALK_STEP
            walk3_done:
                if (nxt != kNxInvalid) vis |= 1ull << idx;
                // emit the sequence positions consumed inside the box
                if (((vis >> lane) & 1ull) && mv != kMvUp) tpos[jj - c0] = (mv == kMvDiag) ? ii : -1;     // (LDS, not HBM: see phase_traceback_code)
                bool stuck = false;
                if (nxt == kNxInvalid) { stuck = idx == 0; i = __builtin_amdgcn_readlane(ii, idx); j = __builtin_amdgcn_readlane(jj, idx); }
                else { i = __builtin_amdgcn_readlane(ni, idx); j = __builtin_amdgcn_readlane(nj, idx); }
#ifdef RCN_PROF_TB
                if (lane == 0) { atomicAdd(&g_dbg[6], (unsigned long long)__popcll(vis)); if (stuck) atomicAdd(&g_dbg[1], 1ull);
                    if (nxt == kNxInvalid && !stuck) atomicAdd(&g_dbg[2], 1ull); }
#endif
#ifdef RCN_PROF_TB
                { const int lna = __builtin_amdgcn_readlane(na, idx), lnb = __builtin_amdgcn_readlane(nb, idx), ldl = __builtin_amdgcn_readlane(dl, idx);
                  if (lane == 0 && nxt == kNxExit) { if (lnb >= kBoxCols) atomicAdd(&g_dbg[3], 1ull); else if (lna >= kBoxRows) atomicAdd(&g_dbg[4], 1ull); else atomicAdd(&g_dbg[5], 1ull);
                                                     if (ldl >= 8) atomicAdd(&g_dbg[7], 1ull); } }
#endif
                if (stuck) break;
            }
            { const int jc = j + 1 + lane; if (jc <= j_stage) prow[jc - 1] = tpos[jc - c0]; }     // the columns consumed on this tile
            if (!(i == 0 && j == 0) && i == ti0 && j == j_stage) {
                // no progress on a freshly anchored tile (predecessor beyond the tile's rows or > 6 in-edges): one
                // step against HBM
#ifdef RCN_PROF_WIN
                if (lane == 0) o->dbg_slow += 1;
#endif
                g.overflow = overflow;
                int pi = i, pj = j, n_dummy = 0;
                if (lane == 0) {
                    traceback2_slow_step(g, nr, sub, seq, m, x, gp, pi, pj, n_dummy);
                    if (pj != j) prow[j - 1] = (pi != i) ? i : -1;
                }
                i = bcast0(pi); j = bcast0(pj); overflow = bcast0(g.overflow);
            }
            if (lane == 0) { o->tb_i = i; o->tb_j = j; o->overflow = overflow; }
#ifdef RCN_PROF_DP
            if (lane == 0) { const long long tp2__ = clock64(); atomicAdd(&g_prof_out[4], (unsigned long long)(tp1__ - tp0__)); atomicAdd(&g_prof_out[5], (unsigned long long)(tp2__ - tp1__));
                             atomicAdd(&g_prof_out[6], 1ull); atomicAdd(&g_prof_out[7], (unsigned long long)nbox__); }
#endif
        }
        Block4::sync();
        i = bcast0(o->tb_i); j = bcast0(o->tb_j);
        if (bcast0(o->overflow)) break;
        Block4::sync();                                  // everyone has read the walk state before the next tile overwrites LDS
    }
    if (t == 0) { o->plen = -1; }
    Block4::sync();
}

// ---- phase: traceback over move codes (the banded DP with CODE, poa_band.hpp) ----
// The DP left one byte per cell: whether a diagonal / a vertical move reproduces the cell and which predecessor (first in
// in-edge order) it comes from.  spoa's priority (diagonal over the in-edges, then vertical over the in-edges, then
// horizontal) is then a table lookup: no score is read, nothing is compared.  Same organisation as phase_traceback3: the
// four waves stage a tile (112 rows x 64 columns, now 64 BYTES per row: four rows per global_load_lds), the 64 lanes of wave 0
// decode the successor of every cell of an 8 x 8 box at once and the walk inside the box is one v_readlane per step.
constexpr int kTileCQuad = 272;        // LDS stride of FOUR tile rows in bytes (4 x 64 + 16: skews the banks)
__device__ __forceinline__ int tilec_at(int trow, int tcol) { return (trow >> 2) * kTileCQuad + (trow & 3) * 64 + tcol; }
static_assert((kTbRows / 4) * kTileCQuad + kTbRows * 32 + 64 * 4 <= kLdsBytes, "code tile + row descriptors + the tile's pos_t must fit");

__device__ __noinline__ void phase_traceback_code() {
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const Ctx c = ctx_load<Block4>();
    Win g = ctx_win(c);
    const int len = c.len;
    const int64_t hs = g.hstride;                    // row stride of the code matrix in bytes
    RCN_G const uint8_t* __restrict__ C = reinterpret_cast<RCN_G const uint8_t*>(g.H.ptr());
    Ctx* o = Block4::ctx();
    if (t == 0) { o->tb_i = c.best_row; o->tb_j = len; o->tb_n = 0; }
    Block4::sync();

    uint8_t* tile = reinterpret_cast<uint8_t*>(Block4::work());                        // [kTbRows / 4][kTileCQuad]
    int* tdesc = Block4::work() + (kTbRows / 4) * kTileCQuad / 4;                      // kTbRows x RowDesc (8 ints each)
    int* tpos = tdesc + kTbRows * 8;                                                   // pos_t of the tile's 64 columns (see the flush below)
    RCN_G int32_t* __restrict__ prow = g.pos_t.ptr();
    RCN_G const int32_t* nr = (c.sub ? g.n2r_x : g.n2r).ptr();
    RCN_G const int32_t* e_nin = g.e_nin.ptr();
    RCN_G const int32_t* e_tail = g.e_tail.ptr();
    RCN_G const uint8_t* inc = g.inc.ptr();
    const bool sub = c.sub != 0;
    int i = bcast0(o->tb_i), j = bcast0(o->tb_j);
    int overflow = g.overflow;
#ifdef RCN_PROF_WIN
    long long ac__[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long ex__[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    while (!(i == 0 && j == 0)) {
#ifdef RCN_PROF_WIN
        const long long tq0__ = clock64();
#endif
        const int ti0 = i, j_stage = j;
#ifdef RCN_PROF_WIN
        if (t == 0) o->dbg_tiles += 1;
#endif
        int c0 = (j - 56) & ~7; if (c0 < 0) c0 = 0;
        const int rmin = ti0 - (kTbRows - 1) > 0 ? ti0 - (kTbRows - 1) : 0;
        {
            typedef __attribute__((address_space(3))) void* lds_ptr;
            constexpr int kQuadsPerWave = kTbRows / 4 / kWaves2;       // 7
            static_assert(kQuadsPerWave * kWaves2 * 4 == kTbRows, "rows split evenly over the waves, four per load");
#pragma unroll
            for (int kk = 0; kk < kQuadsPerWave; ++kk) {
                const int qd = kQuadsPerWave * wv + kk;            // tile rows 4 qd .. 4 qd + 3, sixteen lanes each
                int r = ti0 - (4 * qd + (lane >> 4)); if (r < 1) r = 1;
                RCN_G const uint8_t* src = C + r * hs + c0 + (lane & 15) * 4;
                __builtin_amdgcn_global_load_lds(src, (lds_ptr)(tile + qd * kTileCQuad), 4, 0, 0);
            }
            if (wv == 1 || wv == 2) {
                const int k = (wv - 1) * 64 + lane;               // tile row whose descriptor this lane stages
                if (k < kTbRows) {
                    const int r = ti0 - k;
                    int4 d0 = make_int4(0, -1, -1, -1), d1 = make_int4(-1, -1, -1, 1 << 9);
                    if (r >= 1) { RCN_G const int4* dsrc = reinterpret_cast<RCN_G const int4*>(g.desc.ptr() + (r - 1)); d0 = dsrc[0]; d1 = dsrc[1]; }
                    int4* ddst = reinterpret_cast<int4*>(tdesc + k * 8);
                    ddst[0] = d0; ddst[1] = d1;
                }
            }
        }
#ifdef RCN_PROF_WIN
        const long long tq1__ = clock64();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef RCN_PROF_WIN
        const long long tq2__ = clock64();
        long long bx0__ = 0, bx1__ = 0, bx2__ = 0, bxn__ = 0;
#endif
        if (wv == 0) {
            const int a = lane / kBoxCols, b = lane % kBoxCols;
            for (;;) {
                if (i == 0 && j == 0) break;
#ifdef RCN_PROF_WIN
                if (lane == 0) o->dbg_boxes += 1;
                const long long tb0__ = clock64();
#endif
                const int ii = i - a - kBoxSkew * b, jj = j - b;
                const bool inside = a < kBoxRows && ii >= rmin && ii >= 0 && jj >= c0 && jj >= 0;
                const int trow = inside ? ti0 - ii : 0, tcol = inside ? jj - c0 : 0;
                const int* dr = tdesc + trow * 8;
                const int4 pa = *reinterpret_cast<const int4*>(dr);
                const int2 pb = *reinterpret_cast<const int2*>(dr + 4);
                const int code = tile[tilec_at(trow, tcol)];
                // decode without branches (selects only: the lanes disagree on every one of these conditions, and as
                // branches each of them is an exec-mask region of its own -- the box used to spend more instructions on
                // entering and leaving those than on the decision)
                const bool row0 = ii == 0, jpos = jj > 0;
                const bool dg = !row0 && jpos && !(code & 1);
                const bool up = !row0 && !dg && !(code & 2);
                int mv = dg ? kMvDiag : up ? kMvUp : jpos ? kMvLeft : kMvInvalid;
                const int q = dg ? ((code >> 2) & 7) : up ? (code >> 5) : 0;
                // (the six predecessor rows are in registers before the choice: left to itself the compiler turns the
                //  select back into six conditional LDS loads, each in an exec-mask region)
                int p0_ = pa.x, p1_ = pa.y, p2_ = pa.z, p3_ = pa.w, p4_ = pb.x, p5_ = pb.y;
                asm volatile("" : "+v"(p0_), "+v"(p1_), "+v"(p2_), "+v"(p3_), "+v"(p4_), "+v"(p5_));
                const bool q1 = (q & 1) != 0, q2 = (q & 2) != 0, q4 = (q & 4) != 0;
                const int a01 = q1 ? p1_ : p0_, a23 = q1 ? p3_ : p2_, a45 = q1 ? p5_ : p4_;
                const int a03 = q2 ? a23 : a01;
                int pi = q4 ? a45 : a03;
                const bool far = q > 5 && inside && mv != kMvLeft;
                if (__builtin_expect(__ballot(far) != 0ull, 0)) {
                    if (far) {
                        // seventh / eighth in-edge (rare): not in the descriptor, the q - 6 th included tail of the rest of the list
                        pi = -1;
                        int left = q - 6;
                        for (int e = dr[6]; e >= 0; e = e_nin[e]) {
                            const int tl = e_tail[e];
                            if (sub && !inc[tl]) continue;
                            if (left == 0) { pi = nr[tl] + 1; break; }
                            --left;
                        }
                    }
                }
                mv = (!inside || (mv != kMvLeft && pi < 0)) ? kMvInvalid : mv;
                const int ni = mv == kMvLeft ? ii : pi, nj = jj - (mv == kMvUp ? 0 : 1);
                const int nb = j - nj, na = i - ni - kBoxSkew * nb;
                const bool leaves = (ni == 0 && nj == 0) || na < 0 || na >= kBoxRows || nb >= kBoxCols;
                const int nx = mv == kMvInvalid ? kNxInvalid : leaves ? kNxExit : na * kBoxCols + nb;
                // the walk inside the box: one readlane per step, four steps per loop iteration (a taken branch costs as
                // much as eight instructions, the early exits in between are not taken)
#ifdef RCN_PROF_WIN
                const long long tb1__ = clock64() + (nx & 0);
#endif
                int idx = 0, nxt = kNxInvalid;
                unsigned long long vis = 0ull;
                // (one compare per step: both ways out of the box are >= 64; the bit of the cell the walk stops on is set
                //  afterwards, unless its move is invalid.  s_bitset1_b64 takes the lane number, no 64-bit shift and or.)
#define RCN_WALK_STEP { nxt = __builtin_amdgcn_readlane(nx, idx); if (nxt >= 64) goto walk_done; asm("s_bitset1_b64 %0, %1" : "+s"(vis) : "s"(idx)); idx = nxt; }
                for (;;) { RCN_WALK_STEP RCN_WALK_STEP RCN_WALK_STEP RCN_WALK_STEP }
#undef RCN_WALK_STEP
            walk_done:
                if (nxt != kNxInvalid) vis |= 1ull << idx;
#ifdef RCN_PROF_WIN
                const long long tb2__ = clock64() + (nxt & 0);
#endif
                // (into LDS: a store to HBM here would be waited for by the next box -- the compiler puts an s_waitcnt vmcnt(0)
                //  at the join after the rare seventh-in-edge loads -- and a write round trip is most of what a box then costs)
                if (((vis >> lane) & 1ull) && mv != kMvUp) tpos[tcol] = (mv == kMvDiag) ? ii : -1;
                bool stuck = false;
                if (nxt == kNxInvalid) { stuck = idx == 0; i = __builtin_amdgcn_readlane(ii, idx); j = __builtin_amdgcn_readlane(jj, idx); }
                else { i = __builtin_amdgcn_readlane(ni, idx); j = __builtin_amdgcn_readlane(nj, idx); }
#ifdef RCN_PROF_WIN
                { const long long tb3__ = clock64() + (i & 0); bx0__ += tb1__ - tb0__; bx1__ += tb2__ - tb1__; bx2__ += tb3__ - tb2__; bxn__ += 1;
                  // why the walk left the box, and how many steps it made inside
                  const int xa__ = __builtin_amdgcn_readlane(na, idx), xb__ = __builtin_amdgcn_readlane(nb, idx);
                  const int why__ = nxt == kNxInvalid ? 0 : (i == 0 && j == 0) ? 1 : xb__ >= kBoxCols ? 2 : xa__ < 0 ? 4 : xa__ < 2 * kBoxRows ? 3 : 5;
                  ex__[why__] += 1; ex__[6] += __popcll(vis); }
#endif
                if (stuck) break;
            }
            // the columns the walk consumed on this tile: (j, j_stage], at most 64 (j >= c0 - 1), one store
            { const int jc = j + 1 + lane; if (jc <= j_stage) prow[jc - 1] = tpos[jc - c0]; }
            // a freshly anchored tile always holds the current cell: no progress means a corrupt code matrix
            if (!(i == 0 && j == 0) && i == ti0 && j == j_stage) overflow = 4;
#ifdef RCN_PROF_WIN
            { const long long tq3__ = clock64(); ac__[0] += tq1__ - tq0__; ac__[1] += tq2__ - tq1__; ac__[2] += tq3__ - tq2__; ac__[3] += 1;
              ac__[4] += bx0__; ac__[5] += bx1__; ac__[6] += bx2__; ac__[7] += bxn__; }
#endif
            if (lane == 0) { o->tb_i = i; o->tb_j = j; o->overflow = overflow; }
        }
        Block4::sync();
        i = bcast0(o->tb_i); j = bcast0(o->tb_j);
        if (bcast0(o->overflow)) break;
        Block4::sync();                                  // everyone has read the walk state before the next tile overwrites LDS
    }
    if (t == 0) { o->plen = -1; }
#ifdef RCN_PROF_WIN
    if (t == 0 && c.wi < 4096) for (int k = 0; k < 8; ++k) { g_wtb[c.wi][k] += (unsigned long long)ac__[k]; g_wtb2[c.wi][k] += (unsigned long long)ex__[k]; }
#endif
    Block4::sync();
}

// ---- phase: spoa's exact DFS topological order, in parallel ----
// spoa::Graph::TopologicalSort starts a DFS over in-edges and aligned rings from every not yet visited node in id
// order.  When the DFS from start s ends its stack is empty and every node it touched is finished, so the DFS from
// s only depends on WHICH nodes earlier starts finished, not on how: the finished set is the union of the backward
// closures (in-edges + ring links) of the earlier starts.  Hence with
//     key(X) = smallest node id in the FORWARD closure of X (out-edges + ring links; X itself included)
// node X is appended by the DFS that starts at key(X) (a node is a start iff key(X) == X), spoa's order is "by key,
// then by the post-order of that one DFS", and the DFS of different starts are independent of each other given the
// keys: a node with a smaller key is finished, a node with a larger key is never reached.  Backbone ids are the
// smallest ids and form a chain, so key(X) is the first backbone node X can reach (itself for a backbone node) and
// a typical DFS covers a backbone node plus the few insertion / mismatch nodes in front of it.
//   1. keys: descending sweep over the ring-contiguous order rank_full, 256 ranks at a time; dependencies inside a
//      chunk (non-backbone paths) by fixed-point iteration on the LDS copy of the keys;
//   2. nodes per key -> exclusive scan -> first exact rank of every start;
//   3. one thread per start runs spoa's DFS restricted to its own key (graph_toposort's loop with "finished" =
//      smaller key or local mark), writing its slice of rank_x.
// Sets ctx->tb_i = 1 on success (0: a per-thread stack overflowed or a count did not add up -> serial path).
__device__ __noinline__ void phase_toposort4() {
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const Ctx c = ctx_load<Block4>();
    Win g = ctx_win(c);
    const int n = g.n_nodes, L = c.bblen, ring = g.ring;
    uint16_t* key = reinterpret_cast<uint16_t*>(Block4::work());                 // [n]
    uint16_t* cnt = key + ((n + 2) & ~1);                                         // [n + 1] nodes per key, then first rank per key
    int* flag = Block4::work() + kLdsBytes / 4 - 8;                               // [0] changed, [1] error, [2..5] wave sums
    RCN_G const int32_t* rank_full = g.rank_full.ptr();
    RCN_G const int32_t* out_head = g.out_head.ptr();
    RCN_G const int32_t* e_nout = g.e_nout.ptr();
    RCN_G const int32_t* e_head = g.e_head.ptr();
    RCN_G const int32_t* in_head = g.in_head.ptr();
    RCN_G const int32_t* e_nin = g.e_nin.ptr();
    RCN_G const int32_t* e_tail = g.e_tail.ptr();
    RCN_G const uint8_t* al_cnt = g.al_cnt.ptr();
    RCN_G const int32_t* al_nodes = g.al_nodes.ptr();
    RCN_G uint8_t* mark = g.mark.ptr();
    RCN_G int32_t* rank_x = g.rank_x.ptr();
    for (int X = t; X < n; X += kThreads2) { key[X] = static_cast<uint16_t>(X); mark[X] = 0; }
    for (int X = t; X <= n; X += kThreads2) cnt[X] = 0;
    if (t == 0) { flag[0] = 0; flag[1] = 0; }
    Block4::sync();
    // ---- 1. keys ----
#pragma unroll 1
    for (int hi = n; hi > 0; hi -= kThreads2) {
        const int r = hi - 1 - t;
        const int X = r >= 0 ? rank_full[r] : -1;
        const bool act = X >= L;                          // a backbone node is its own key
        const int na = act ? al_cnt[X] : 0;
#pragma unroll 1
        for (;;) {
            if (act) {
                const int cur = key[X];
                int nb = cur;
                // the whole ring at once (its members may straddle a chunk border): ids and out-neighbours of every member
                for (int a = -1; a < na; ++a) {
                    const int M = a < 0 ? X : al_nodes[X * ring + a];
                    nb = min(nb, M);
                    for (int e = out_head[M]; e >= 0; e = e_nout[e]) nb = min(nb, static_cast<int>(key[e_head[e]]));
                }
                if (nb < cur) { key[X] = static_cast<uint16_t>(nb); flag[0] = 1; }
            }
            Block4::sync();
            const int ch = flag[0];
            Block4::sync();
            if (!ch) break;
            if (t == 0) flag[0] = 0;
            Block4::sync();
        }
    }
    // ---- 2. nodes per key, first rank per key ----
    {
        unsigned int* cnt32 = reinterpret_cast<unsigned int*>(cnt);
        for (int X = t; X < n; X += kThreads2) { const int k = key[X]; atomicAdd(&cnt32[k >> 1], 1u << (16 * (k & 1))); }
        Block4::sync();
        const int seg = (n + kThreads2 - 1) / kThreads2, lo = min(n, t * seg), hi2 = min(n, lo + seg);
        int sum = 0;
        for (int i = lo; i < hi2; ++i) sum += cnt[i];
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
        if (lane == 63) flag[2 + wv] = incl;
        Block4::sync();
        int run = incl - sum;
        for (int w = 0; w < kWaves2; ++w) if (w < wv) run += flag[2 + w];
        for (int i = lo; i < hi2; ++i) { const int cc = cnt[i]; cnt[i] = static_cast<uint16_t>(run); run += cc; }
        if (t == kThreads2 - 1) cnt[n] = static_cast<uint16_t>(run);
        Block4::sync();
    }
    // ---- 3. one DFS per start ----
    {
        const int64_t per = g.hcap / (2 * kThreads2);          // ints per thread of the (finished) int16 score matrix
        const int cap = static_cast<int>(per < (1 << 20) ? per : (1 << 20));
        RCN_G int32_t* stk = reinterpret_cast<RCN_G int32_t*>(g.H.ptr()) + static_cast<int64_t>(t) * cap;
        int err = 0;
#pragma unroll 1
        for (int s = t; s < n; s += kThreads2) {
            if (key[s] != s) continue;
            int out = cnt[s];
            const int out_end = cnt[s + 1];
            int sp = 0;
            stk[sp++] = s;
            while (sp > 0) {
                const int cu = stk[sp - 1];
                bool valid = true;
                const int mc = mark[cu];
                if ((mc & 3) != 2) {
                    for (int e = in_head[cu]; e >= 0; e = e_nin[e]) {
                        const int tl = e_tail[e];
                        if (key[tl] != s) continue;                       // finished by an earlier start
                        if ((mark[tl] & 3) != 2) { if (sp < cap) stk[sp++] = tl; else err = 1; valid = false; }
                    }
                    const bool ign = (mc & 4) != 0;
                    const int na = al_cnt[cu];
                    if (!ign) {
                        for (int a = 0; a < na; ++a) {
                            const int u = al_nodes[cu * ring + a];
                            const int mu = mark[u];
                            if ((mu & 3) != 2) { if (sp < cap) stk[sp++] = u; else err = 1; mark[u] = static_cast<uint8_t>(mu | 4); valid = false; }
                        }
                    }
                    if (err) break;
                    if (valid) {
                        mark[cu] = static_cast<uint8_t>((mc & 4) | 2);
                        if (!ign) {
                            if (out + 1 + na > out_end) { err = 1; break; }
                            rank_x[out++] = cu;
                            for (int a = 0; a < na; ++a) rank_x[out++] = al_nodes[cu * ring + a];
                        }
                    } else {
                        mark[cu] = static_cast<uint8_t>((mc & 4) | 1);
                    }
                }
                if (valid) --sp;
            }
            if (out != out_end) err = 1;
            if (err) break;
        }
        if (err) flag[1] = 1;
        Block4::sync();
        const int bad = flag[1];
        for (int r = t; r < n; r += kThreads2) g.n2r_x[rank_x[r]] = r;
        if (t == 0) Block4::ctx()->tb_i = bad ? 0 : 1;
        Block4::sync();
    }
}

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

__global__ __launch_bounds__(kThreads2, 8) void poa_window_kernel2(KParams P) {
    const int t = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // sub, desc, dp, traceback, add, merge, consensus, other
    long long tck = clock64();
#define RCN_PHASE2(k) do { long long now__ = clock64(); ph[k] += now__ - tck; tck = now__; } while (0)
    Ctx* ctx = ctx_lds2();
    if (t == 0) {
        ctx->scratch = P.scratch + static_cast<uint64_t>(blockIdx.x) * P.slot_bytes;
        ctx->ncap = P.ncap; ctx->ecap = P.ecap; ctx->ring = P.ring; ctx->lmax = P.lmax; ctx->hstride = P.hstride; ctx->hrows = P.hrows;
        ctx->m = P.m; ctx->x = P.x; ctx->gp = P.g; ctx->trim = P.trim;
        ctx->cells = 0; ctx->pred = 0; ctx->bytes = 0; ctx->ties = 0;
        ctx->cells_full = 0; ctx->bytes_full = 0; ctx->n_banded = 0; ctx->n_band_fail = 0; ctx->band = 0; ctx->band_fail = 0;
        ctx->band_why = 0; for (int k = 0; k < 8; ++k) ctx->band_whyn[k] = 0;
        ctx->dbg_tiles = 0; ctx->dbg_boxes = 0; ctx->dbg_slow = 0;
    }
    for (;;) {
        RCN_PHASE2(7);
        Block4::sync();                                  // previous work item fully retired (and ctx->wi read by all)
        if (t == 0) ctx->wi = static_cast<int32_t>(atomicAdd(P.next, 1u));
        Block4::sync();
        const unsigned int wi = static_cast<unsigned int>(bcast0(ctx->wi));
        if (wi >= P.n_work) break;
#ifdef RCN_PROF_WIN
        unsigned long long ph0__[8];
        for (int k = 0; k < 8; ++k) ph0__[k] = ph[k];
#endif
        const uint32_t w = P.win_ids ? P.win_ids[wi] : P.work_base + wi;
        const uint32_t s0 = P.win_seq_off[w];
        const int ns = static_cast<int>(P.win_seq_off[w + 1] - s0);
        const uint8_t* bb = P.bases + P.seq_off[s0];
        const int L = static_cast<int>(P.seq_off[s0 + 1] - P.seq_off[s0]);
        const uint32_t oi = P.out_base + wi;                                       // outputs are indexed by work item
        uint8_t* out = P.out_cons + P.out_off[oi];
        const uint64_t out_cap = P.out_off[oi + 1] - P.out_off[oi];

        const bool heavy = P.heavy_ns > 0 && ns >= P.heavy_ns;
        if (ns < 3) {                                          // window.cpp:68-71
            for (int i = t; i < L; i += kThreads2) out[i] = bb[i];
            if (t == 0) { P.out_len[oi] = L; P.out_flags[oi] = 0; }
            continue;
        }
        // ---- backbone -> graph (window.cpp:73-77) ----
        {
            Win g;
            win_bind(g, gcast(P.scratch + static_cast<uint64_t>(blockIdx.x) * P.slot_bytes), P.ncap, P.ecap, P.ring, P.lmax, P.hstride, 4, P.hrows);
            RCN_G const uint8_t* q0 = P.seq_has_qual[s0] ? gcast(P.quals + P.seq_off[s0]) : nullptr;
            for (int i = t; i < L; i += kThreads2) {
                g.code[i] = bb[i]; g.al_cnt[i] = 0;
                g.in_head[i] = g.in_tail[i] = (i > 0) ? i - 1 : -1;
                { PredRec pr = pred_rec_empty(); if (i > 0) { pr.t[0] = i - 1; pr.k = 1; } g.in6[i] = pr; }
                g.out_head[i] = g.out_tail[i] = (i < L - 1) ? i : -1;
                g.cov[i] = L >= 2 ? 1u : 0u;
                g.rank_full[i] = i; g.n2r[i] = i;
                if (i < L - 1) {
                    g.e_tail[i] = i; g.e_head[i] = i + 1; g.e_nin[i] = -1; g.e_nout[i] = -1;
                    g.e_w[i] = pair_weight(q0, i + 1);
                }
            }
            if (t == 0) { ctx->n_nodes = L; ctx->n_edges = L - 1; ctx->overflow = (L > P.ncap) ? 1 : 0; ctx->swapped = 0; ctx->pad0 = heavy; ctx->bblen = L; ctx->tie_pad[1] = P.win_flags ? (P.win_flags[w] & 1) : 0; ctx->tie_pad[2] = P.force_slow_tb; ctx->tie_pad[0] = P.band; }
        }
        Block4::sync();

        int overflow = bcast0(ctx->overflow);
        for (int jl = 1; jl < ns && !overflow; ++jl) {
            const uint32_t si = s0 + P.order[s0 + jl];
            const int len = static_cast<int>(P.seq_off[si + 1] - P.seq_off[si]);
            const bool partial = P.seq_full[si] == 0;
            if (t == 0) {
                ctx->seq = P.bases + P.seq_off[si];
                ctx->qual = P.seq_has_qual[si] ? P.quals + P.seq_off[si] : nullptr;
                ctx->len = len;
                ctx->sub = partial;
                ctx->begin = static_cast<int32_t>(P.seq_begin[si]); ctx->end = static_cast<int32_t>(P.seq_end[si]);
                ctx->V = ctx->n_nodes;
                ctx->tb_j = 0;                 // phase_subgraph2: 0 = normal, 1 = closure query (marks only)
                ctx->band = (P.band && !heavy) ? band_np(len) : 0; ctx->band_fail = 0; ctx->coded = 0;
            }
            Block4::sync();
            if (partial) {
                bool done = false;
                if (bcast0(ctx->n_nodes) <= kSubMaxNodes) { phase_subgraph2(); done = bcast0(ctx->tb_i) != 0; }
                if (!done) { if (wv == 0) phase_subgraph<Wave0Of4>(); Block4::sync(); }
            }
            RCN_PHASE2(0);
            // int16 (Z domain) validity of this alignment; otherwise the window goes to the int32 kernel
            const int V = bcast0(ctx->V);
            const int cfg = dp2_cfg(len, heavy), np_regs = cfg & 255, nwv = cfg >> 8;
            if (V + 1 > P.hrows) { overflow = 1; break; }          // more rows than this slot's matrix holds: the retry pass takes the window
            {
                const int ag = P.g < 0 ? -P.g : P.g, smax = max(max(P.m, P.x), 0);
                if (P.g >= 0 || cfg == 0 || static_cast<long long>(ag) * (V + 2) > kZLimit ||
                    static_cast<long long>(smax + ag) * (128 * np_regs * nwv) > kZLimit) { overflow = 5; break; }
            }
            phase_desc2();
            RCN_PHASE2(1);
            bool dp_done = false;
            const int bnp = bcast0(ctx->band);
            if (bnp) {
                // exact banded DP (poa_band.hpp); an alignment whose certificate fails is redone on full rows right below
#ifdef RCN_ABLATE
                if (wv == 0) {
                    if (bcast0(ctx->tie_pad[1]) != 0) dp2_rows_band<2, true, RCN_ABLATE>(); else dp2_rows_band<2, false, RCN_ABLATE>();
                }
                Block4::sync();
#endif
                const bool coded = P.band != 3;
                if (wv == 0) {
                    const bool tab = bcast0(ctx->tie_pad[1]) != 0;
                    if (coded) { if (tab) dp2_rows_band<2, true, 0, true>(); else dp2_rows_band<2, false, 0, true>(); }
                    else { if (tab) dp2_rows_band<2, true>(); else dp2_rows_band<2, false>(); }
                }
                Block4::sync();
                dp_done = bcast0(ctx->band_fail) == 0;
                if (t == 0) ctx->coded = (dp_done && coded) ? 1 : 0;
                Block4::sync();
                if (!dp_done) {
                    Block4::sync();
                    if (t == 0) ctx->band = 0;
                    Block4::sync();
                    phase_desc2();
                }
            }
            if (dp_done) {
            } else if (nwv == 1) {
                if (wv == 0) {
                    const bool tab = bcast0(ctx->tie_pad[1]) != 0;
                    switch (np_regs) {
                        case 1: if (tab) dp2_rows<1, 1, true>(); else dp2_rows<1, 1>(); break;
                        case 2: if (tab) dp2_rows<2, 1, true>(); else dp2_rows<2, 1>(); break;
                        case 3: if (tab) dp2_rows<3, 1, true>(); else dp2_rows<3, 1>(); break;
                        default: if (tab) dp2_rows<4, 1, true>(); else dp2_rows<4, 1>(); break;
                    }
                }
                Block4::sync();
            } else {
                switch (np_regs) {
                    case 1: dp2_rows<1, 4>(); break;
                    case 2: dp2_rows<2, 4>(); break;
                    case 3: dp2_rows<3, 4>(); break;
                    default: dp2_rows<4, 4>(); break;
                }
            }
            RCN_PHASE2(2);
            if (bcast0(ctx->tied) > 1) {
                if (wv == 0) phase_sink_tie_rule();
                Block4::sync();
                int st = bcast0(ctx->tb_n);
                if (P.force_tie) {
                    // test switches: the cheaper levels' answers are discarded, a later level must reproduce them
                    // (more than eight tied sinks / graphs beyond the LDS sweep stay on the full DFS either way)
                    if (P.force_tie >= 3) { st = 2; if (t == 0) ctx->tie_why = 6; }
                    else if (st == 0 && bcast0(ctx->n_nodes) <= kSubMaxNodes) st = 1;
                    Block4::sync();
                }
                if (st == 1) {
                    if (wv == 0) phase_sink_tie_starts();
                    Block4::sync();
                    st = bcast0(ctx->tb_n);
                    if (st == 3) {
                        // marks of what spoa's DFS has finished before the deciding start: the closure of backbone
                        // node p* - 1 (of L - 1 when no candidate is in the backbone closure) = a Subgraph sweep in
                        // mark-only mode.  It borrows the descriptor array: rebuilt afterwards.
                        const int pstar = bcast0(ctx->tb_i), sv_end = bcast0(ctx->end), sv_begin = bcast0(ctx->begin), sv_j = bcast0(ctx->tb_j);
                        const int first = partial ? sv_begin : 0;
                        const int upto = pstar == 0x7fffffff ? bcast0(ctx->bblen) - 1 : pstar - 1;
                        bool swept = true;
                        Block4::sync();
                        if (upto >= first) {
                            if (t == 0) { ctx->begin = first; ctx->end = upto; ctx->tb_j = 1; }
                            Block4::sync();
                            phase_subgraph2();
                            swept = bcast0(ctx->tb_i) != 0;
                            Block4::sync();
                            if (t == 0) { ctx->begin = sv_begin; ctx->end = sv_end; ctx->tb_j = sv_j; ctx->tb_i = pstar; }
                            Block4::sync();
                            phase_desc2();
                        } else {
                            Win g;
                            win_bind(g, gcast(P.scratch + static_cast<uint64_t>(blockIdx.x) * P.slot_bytes), P.ncap, P.ecap, P.ring, P.lmax, P.hstride, 4, P.hrows);
                            const int nn_ = bcast0(ctx->n_nodes);
                            for (int v = t; v < nn_; v += kThreads2) g.mark[v] = 0;
                            Block4::sync();
                        }
                        if (swept) { if (wv == 0) phase_sink_tie_local(); Block4::sync(); st = bcast0(ctx->tb_n); }
                        else { st = 2; if (t == 0) ctx->tie_why = 5; }
                    }
                }
                if (st == 2) { if (wv == 0) phase_sink_tie_full(); Block4::sync(); }
            }
            if (bcast0(ctx->coded)) phase_traceback_code(); else phase_traceback3();
            RCN_PHASE2(3);
            overflow = bcast0(ctx->overflow);
            if (!overflow) {
                phase_add4();
                RCN_PHASE2(4);
                overflow = bcast0(ctx->overflow);
                if (!overflow) phase_merge4();
                RCN_PHASE2(5);
            }
        }
        if (overflow) {
            if (t == 0) { P.out_len[oi] = 0; P.out_flags[oi] = (overflow == 1 || overflow == 3 || overflow == 5) ? kFlagOverflow : kFlagError; }
            continue;
        }
        {
            // 32-bit bundle scores: every edge weight is a sum of (quality - 33) pairs, at most 444 per base pair
            const uint64_t wbases = P.seq_off[s0 + ns] - P.seq_off[s0];
            const bool fast_cons = bcast0(ctx->n_nodes) <= kCons2MaxNodes && wbases * 444ull < 0x7fffffffull;
            int k = 0;
            if (fast_cons) {
                phase_cons2_edges(0);
                if (wv == 0) phase_cons2_bundle(0);
                Block4::sync();
                k = bcast0(ctx->tb_n);
                if (P.force_exact) k = 0;
                if (k == 0) {
                    // several nodes tie for the best score, or the best node is not a sink (BranchCompletion): both
                    // depend on spoa's own rank order -> the same bundle over that order
                    phase_toposort4();
                    if (bcast0(ctx->tb_i)) {
#ifdef RCN_VERIFY_TOPO
                        if (t == 0) {
                            Win g = ctx_win(*ctx);
                            const int nx = graph_toposort(g, g.rank_tmp.ptr(), false, g.stack.ptr());
                            int bad = nx != g.n_nodes;
                            for (int r = 0; r < g.n_nodes && !bad; ++r) bad = g.rank_tmp[r] != g.rank_x[r];
                            if (bad) printf("[toposort4] MISMATCH wi %d n %d\n", ctx->wi, g.n_nodes);
                        }
                        Block4::sync();
#endif
                        phase_cons2_edges(1);
                        if (wv == 0) phase_cons2_bundle(1);
                        Block4::sync();
                        k = bcast0(ctx->tb_n);
                    }
                }
            }
            if (wv == 0) {
                if (k > 0) phase_cons2_finish(out, out_cap, &P.out_len[oi], &P.out_flags[oi], ns, P.win_type[w] == 1);
                else phase_consensus<Wave0Of4>(out, out_cap, &P.out_len[oi], &P.out_flags[oi], ns, P.win_type[w] == 1);
            }
        }
        RCN_PHASE2(6);
#ifdef RCN_PROF_WIN
        if (t == 0 && wi < 4096) { for (int k = 0; k < 7; ++k) g_wclk[wi][k] = ph[k] - ph0__[k];
            g_wclk[wi][7] = (static_cast<unsigned long long>(ctx->dbg_tiles) << 40) | (static_cast<unsigned long long>(ctx->dbg_boxes) << 20) | static_cast<unsigned long long>(ctx->dbg_slow);
            ctx->dbg_tiles = 0; ctx->dbg_boxes = 0; ctx->dbg_slow = 0; }
#endif
    }
    if (t == 0) {
        atomicAdd(&P.stats[0], ctx->cells); atomicAdd(&P.stats[1], ctx->pred); atomicAdd(&P.stats[2], ctx->bytes);
        for (int k = 0; k < 8; ++k) atomicAdd(&P.stats[3 + k], ph[k]);
        atomicAdd(&P.stats[11], ctx->ties);
        atomicAdd(&P.stats[12], ctx->cells_full); atomicAdd(&P.stats[13], ctx->bytes_full);
        atomicAdd(&P.stats[14], static_cast<unsigned long long>(ctx->n_banded)); atomicAdd(&P.stats[15], static_cast<unsigned long long>(ctx->n_band_fail));
        for (int k = 0; k < 8; ++k) if (ctx->band_whyn[k]) atomicAdd(&P.stats[16 + k], static_cast<unsigned long long>(ctx->band_whyn[k]));
    }
}

}  // namespace rcn
