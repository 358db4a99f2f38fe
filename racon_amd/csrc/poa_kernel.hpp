// poa_kernel.hpp — the MI355X (gfx950) window-consensus kernel.
//
// One 64-lane wavefront owns one window (one partial-order graph) at a time and
// loops over a device-side work queue (persistent slots).  Per window it runs
// racon's Window::generate_consensus (reference src/window.cpp:65-149):
//
//   backbone -> graph                         (wave-parallel)
//   for every layer, in the host-sorted order of window.cpp:79-86:
//       [subgraph mask + exact DFS order]     (lane 0;  window.cpp:99-103)
//       row descriptors                       (wave-parallel, 64 rows at a time)
//       NW sequence-to-graph DP               (wave-parallel: each lane owns CT
//                                              adjacent columns of a row; the
//                                              horizontal gap is a wave-wide
//                                              prefix-max; window.cpp:95-97,104-106)
//       traceback                             (lane 0)
//       AddAlignment + exact DFS toposort     (lane 0;  window.cpp:110-119)
//   heaviest-bundle consensus, coverage, trim (lane 0;  window.cpp:122-146)
//
// Integer DP on an irregular DAG: no MFMA.  Scores are int32 in HBM scratch;
// the previous row is kept in registers (the common predecessor).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "poa_core.hpp"

namespace rcn {

struct KParams {
    // resident batch (rcn_batch, device copies)
    const uint32_t* win_seq_off; const uint8_t* win_type; const uint64_t* seq_off;
    const uint8_t* seq_has_qual; const uint32_t* seq_begin; const uint32_t* seq_end;
    const uint8_t* bases; const uint8_t* quals;
    const uint32_t* order;        // [n_seqs] processing order inside each window (std::sort on host)
    const uint8_t*  seq_full;     // [n_seqs] 1 = full-span layer (window.cpp:93-94), else Subgraph
    const uint32_t* win_ids;      // [n_work] indirection (retry pass) or nullptr
    uint32_t n_work;
    int32_t m, x, g, trim;
    // per-slot scratch
    uint8_t* scratch; uint64_t slot_bytes; int32_t ncap, ecap, ring, lmax, hstride;
    // outputs
    uint8_t* out_cons; uint64_t out_stride; uint32_t* out_len; uint8_t* out_flags;
    // queue + counters
    unsigned int* next; unsigned long long* stats;   // stats[0]=cells, [1]=pred cells, [2]=algorithmic DP bytes, [3..10]=phase clocks, [11]=sink ties
};

enum : uint8_t { kFlagPolished = 1, kFlagChimeric = 2, kFlagOverflow = 4, kFlagError = 8 };

__device__ __forceinline__ int bcast0(int v) { return __builtin_amdgcn_readfirstlane(v); }

// DPP cross-lane primitives (gfx9-family encodings, valid on gfx950):
//   row_shr:n = 0x110+n (shift inside a row of 16 lanes), row_bcast15 = 0x142,
//   row_bcast31 = 0x143, wave_shr:1 = 0x138 (whole-wave shift by one lane).
// Lanes without a valid source keep `old` (bound_ctrl = false).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
// value of lane-1 (lane 0 gets `fill`)
__device__ __forceinline__ int wave_shr1(int v, int fill) { return dpp_or<0x138, 0xf>(fill, v); }

__device__ __forceinline__ int wave_incl_scan_max(int v) {
    v = max(v, dpp_or<0x111, 0xf>(kNeg, v));
    v = max(v, dpp_or<0x112, 0xf>(kNeg, v));
    v = max(v, dpp_or<0x114, 0xf>(kNeg, v));
    v = max(v, dpp_or<0x118, 0xf>(kNeg, v));
    v = max(v, dpp_or<0x142, 0xa>(kNeg, v));
    v = max(v, dpp_or<0x143, 0xc>(kNeg, v));
    return v;
}
__device__ __forceinline__ int wave_excl_scan_max(int z, int /*lane*/) {
    return wave_shr1(wave_incl_scan_max(z), kNeg);
}

// One column tile [t0, t0 + 64*CT) of the DP matrix, all rows.
constexpr int kMaxCT = 10;   // widest column tile: 64 * 12 = 768 columns per pass
constexpr int kLdsBytes = 20480;               // per wave (one wave per workgroup): 8 waves / CU
constexpr int kRingInts = kLdsBytes / 4;
struct DpState { int best, best_row, have_best, tied; unsigned int pred_rows; };

template <int CT, bool FIRST>      // FIRST: the tile starts at column 0 (no global loads in the row loop)
__device__ __forceinline__ DpState dp_tile(const Win& g, const Arr<int32_t>& nr, int V, bool sub, const uint8_t* __restrict__ seq, int len,
                                        int t0, bool last_tile, int m, int x, int gp, DpState st, int2* __restrict__ ring) {
    // LDS ring of the last K score rows (predecessors are almost always < 16 rows back in the
    // incrementally maintained order).  Layout: slot s, column pair q of lane l at
    // ring[(s * (CT/2) + q) * 64 + l]: every ds_read/write_b64 is stride-8B, conflict free.
    constexpr int K = kRingInts / (64 * CT) - 1;                // + one spare slot (index K) for far rows
    int slot = 0;                                               // ring slot of row i (row 0 is analytic)
    int best = st.best, best_row = st.best_row, have_best = st.have_best, tied = st.tied;
    unsigned int pred_rows = 0;
    const int lane = threadIdx.x;
    const int j0 = t0 + lane * CT;
    const int64_t hs = g.hstride;
    int32_t* __restrict__ H = g.H.ptr();

    uint8_t sq[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) { const int j = j0 + c; sq[c] = (j >= 1 && j <= len) ? seq[j - 1] : 0; }

    int last[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) last[c] = (j0 + c) * gp;      // row 0
    int last_row = 0;
    const int own_lane = (len - t0) / CT;                       // lane owning column `len` (last tile only)

    // Rows are processed in chunks of 64: one coalesced load brings 64 row descriptors (one per
    // lane), then the inner loop reads them with v_readlane.  Keeping the load (and its
    // s_waitcnt) out of the inner loop matters: the only VMEM traffic of the inner loop is the
    // H-row stores, and nothing there ever waits for them.
#pragma unroll 1
    for (int rbase = 0; rbase < V; rbase += 64) {
    RowDesc dl; dl.erest = -1; dl.meta = 1 << 9;
#pragma unroll
    for (int q = 0; q < kInlinePreds; ++q) dl.p[q] = 0;
    if (rbase + lane < V) dl = g.desc[rbase + lane];
    const int rend = min(V, rbase + 64);
#pragma unroll 1
    for (int r = rbase; r < rend; ++r) {
        const int k = r - rbase;
        const int p0 = __builtin_amdgcn_readlane(dl.p[0], k);
        const int er = __builtin_amdgcn_readlane(dl.erest, k);
        const int meta = __builtin_amdgcn_readlane(dl.meta, k);
        const int np = (meta >> 9) & 15;
        const uint8_t sym = meta & 255;
        const int i = r + 1;

        int acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = kNeg;

        auto accumulate = [&](int p) {
            int hp[CT];
            if (p == last_row) {
#pragma unroll
                for (int c = 0; c < CT; ++c) hp[c] = last[c];
            } else if (p == 0) {
#pragma unroll
                for (int c = 0; c < CT; ++c) hp[c] = (j0 + c) * gp;
            } else {
                int sp;
                if (i - p < K) {
                    sp = slot - (i - p); if (sp < 0) sp += K;
                } else {
                    // rare (<0.1%): predecessor older than the ring -> stage its row through the spare
                    // slot, so that the common path never has a global load pending at the join
                    const int2* gsrc = reinterpret_cast<const int2*>(H + p * hs + j0);
                    int2* sdst = ring + K * (CT / 2) * 64 + lane;
#pragma unroll
                    for (int c = 0; c < CT; c += 2) sdst[(c >> 1) * 64] = gsrc[c >> 1];
                    sp = K;
                }
                const int2* src = ring + sp * (CT / 2) * 64 + lane;
#pragma unroll
                for (int c = 0; c < CT; c += 2) { const int2 v = src[(c >> 1) * 64]; hp[c] = v.x; hp[c + 1] = v.y; }
            }
            int left = wave_shr1(hp[CT - 1], kNeg);
            if (!FIRST) { if (lane == 0) left = H[p * hs + t0 - 1]; }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int dg = (c == 0 ? left : hp[c - 1]) + (sq[c] == sym ? m : x);
                const int up = hp[c] + gp;
                acc[c] = max(acc[c], max(dg, up));
            }
            ++pred_rows;
        };
        accumulate(p0);
        if (np > 1) {
            accumulate(__builtin_amdgcn_readlane(dl.p[1], k));
            if (np > 2) {
                accumulate(__builtin_amdgcn_readlane(dl.p[2], k));
                if (np > 3) accumulate(__builtin_amdgcn_readlane(dl.p[3], k));
                if (np > 4) accumulate(__builtin_amdgcn_readlane(dl.p[4], k));
                if (np > 5) accumulate(__builtin_amdgcn_readlane(dl.p[5], k));
            }
        }
        for (int e = er; e >= 0; e = g.e_nin[e]) {
            const int t = g.e_tail[e];
            if (sub && !g.inc[t]) continue;
            accumulate(nr[t] + 1);
        }
        // horizontal gap: in-lane pass, then wave-wide prefix max of the transformed lane tails
#pragma unroll
        for (int c = 1; c < CT; ++c) acc[c] = max(acc[c], acc[c - 1] + gp);
        int z = acc[CT - 1] - (j0 + CT - 1) * gp;
        int zex = wave_excl_scan_max(z, lane);
        if (!FIRST) zex = max(zex, H[i * hs + t0 - 1] - (t0 - 1) * gp);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = max(acc[c], zex + (j0 + c) * gp);

        int2* dst = reinterpret_cast<int2*>(H + i * hs + j0);
        int2* rdst = ring + slot * (CT / 2) * 64 + lane;
#pragma unroll
        for (int c = 0; c < CT; c += 2) { const int2 v = make_int2(acc[c], acc[c + 1]); dst[c >> 1] = v; rdst[(c >> 1) * 64] = v; }
        slot = (slot + 1 == K) ? 0 : slot + 1;
#pragma unroll
        for (int c = 0; c < CT; ++c) last[c] = acc[c];
        last_row = i;

        if (last_tile && (meta & 256)) {
            int cand = kNeg;
#pragma unroll
            for (int c = 0; c < CT; ++c) if (j0 + c == len) cand = acc[c];
            const int val = __builtin_amdgcn_readlane(cand, own_lane);
            if (!have_best || best < val) { have_best = 1; best = val; best_row = i; tied = 1; }
            else if (best == val) ++tied;
        }
    }
    }
    DpState o; o.best = best; o.best_row = best_row; o.have_best = have_best; o.tied = tied;
    o.pred_rows = (t0 == 0) ? pred_rows : st.pred_rows;
    return o;
}

__device__ __forceinline__ void wave_sync() { __threadfence_block(); __syncthreads(); }

// Traceback of spoa's linear NW (same decisions as rcn::nw_traceback) with the
// score matrix read through LDS tiles: the whole wave stages a 64-row x 64-col
// tile of H, the 64 row descriptors and the 64 sequence symbols around the
// current cell; lane 0 then walks inside the tile with LDS reads only (the fast
// loop contains no global load, so nothing in it waits on memory) and leaves it
// when a needed cell falls outside.  If a freshly anchored tile still cannot
// serve the step (predecessor > 60 rows back, or > 2 in-edges), that single
// step is done against HBM.  Emits (row | -1, pos | -1) in reverse order; rows
// are mapped to node ids afterwards, in parallel.
constexpr int kTileStride = 68;   // ints per tile row: 64 + 4 pad (conflict-free ds_write_b128)

__device__ __forceinline__ void traceback_slow_step(Win& g, const Arr<int32_t>& nr, bool sub, const uint8_t* seq,
                                                    int m, int x, int gp, int& i, int& j, int& n) {
    const int64_t hs = g.hstride;
    const int32_t* H = g.H.ptr();
    const int hij = H[i * hs + j];
    int pi = 0, pj = 0; bool found = false;
    if (i != 0) {
        const RowDesc d = g.desc[i - 1];
        const int np = (d.meta >> 9) & 15;
        for (int pass = (j != 0 ? 0 : 1); pass < 2 && !found; ++pass) {
            const int col = pass == 0 ? j - 1 : j;
            const int add = pass == 0 ? (((d.meta & 255) == seq[j - 1]) ? m : x) : gp;
            for (int q = 0; q < np && !found; ++q) {
                if (hij == H[d.p[q] * hs + col] + add) { pi = d.p[q]; pj = col; found = true; }
            }
            for (int e = d.erest; e >= 0 && !found; e = g.e_nin[e]) {
                const int t = g.e_tail[e];
                if (sub && !g.inc[t]) continue;
                const int p = nr[t] + 1;
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

__device__ __forceinline__ int traceback_tiled(Win& g, const Arr<int32_t>& nr, bool sub, const uint8_t* __restrict__ seq,
                                               int len, int best_row, int m, int x, int gp, int* __restrict__ tile, unsigned long long* dbg) {
    const int lane = threadIdx.x;
    const int64_t hs = g.hstride;
    const int32_t* __restrict__ H = g.H.ptr();
    RowDesc* tdesc = reinterpret_cast<RowDesc*>(tile + 64 * kTileStride);
    uint8_t* tseq = reinterpret_cast<uint8_t*>(tile + 64 * kTileStride + 64 * (sizeof(RowDesc) / 4));     // seq[c0 - 1 + k], k = 0..63
    int32_t* __restrict__ pnode = g.path_node.ptr();
    int32_t* __restrict__ ppos = g.path_pos.ptr();
    int i = best_row, j = len, n = 0;
    while (!(i == 0 && j == 0)) {
        // ---- stage the tile: rows [i-63, i], cols [c0, c0+63] ----
        const int ti0 = i;
        int c0 = (j - 60) & ~3; if (c0 < 0) c0 = 0;
        const int rmin = ti0 - 63 > 0 ? ti0 - 63 : 0;
        {
            const int r = ti0 - lane;
            if (r >= 0) {
                const int4* src = reinterpret_cast<const int4*>(H + r * hs + c0);
                int4* dst = reinterpret_cast<int4*>(tile + lane * kTileStride);
#pragma unroll 1
                for (int q = 0; q < 16; q += 4) { const int4 a = src[q], b = src[q + 1], c = src[q + 2], d = src[q + 3]; dst[q] = a; dst[q + 1] = b; dst[q + 2] = c; dst[q + 3] = d; }
                if (r >= 1) tdesc[lane] = g.desc[r - 1];
            }
            const int sc = c0 - 1 + lane;
            tseq[lane] = (sc >= 0 && sc < len) ? seq[sc] : 0;
        }
        __syncthreads();
        int steps = 0;
        if (lane == 0) {
            int hij = tile[(ti0 - i) * kTileStride + (j - c0)];
            for (;;) {
                if (i == 0 && j == 0) break;
                int pi, pj, hnext;
                if (i == 0) {                       // only horizontal moves are left on the virtual row
                    pi = 0; pj = j - 1; hnext = hij - gp;
                    if (j - 1 < c0) break;
                } else {
                    const RowDesc d = tdesc[ti0 - i];
                    const int np = (d.meta >> 9) & 15;
                    // everything this step may read must be inside the tile
                    int pmin = d.p[0];
#pragma unroll
                    for (int q = 1; q < kInlinePreds; ++q) if (q < np) pmin = min(pmin, d.p[q]);
                    if (pmin < rmin || d.erest >= 0 || (j > 0 && j - 1 < c0)) break;
                    const int mc = ((d.meta & 255) == tseq[j - c0]) ? m : x;      // seq[j-1]
                    const int* col = tile + ti0 * kTileStride + (j - c0);         // col[-p * stride] = H[p][j]
                    bool found = false;
                    pi = i; pj = j - 1; hnext = hij - gp;                         // horizontal unless a predecessor matches
                    if (j > 0) {
#pragma unroll
                        for (int q = 0; q < kInlinePreds; ++q) {
                            if (q < np && !found) { const int h = col[-d.p[q] * kTileStride - 1]; if (hij == h + mc) { pi = d.p[q]; pj = j - 1; hnext = h; found = true; } }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < kInlinePreds; ++q) {
                        if (q < np && !found) { const int h = col[-d.p[q] * kTileStride]; if (hij == h + gp) { pi = d.p[q]; pj = j; hnext = h; found = true; } }
                    }
                }
                pnode[n] = (i == pi) ? -1 : i;          // ROW index (mapped to the node id later)
                ppos[n] = (j == pj) ? -1 : j - 1;
                ++n; ++steps;
                i = pi; j = pj; hij = hnext;
            }
            dbg[0] += 1; dbg[1] += steps;
            if (steps == 0 && !(i == 0 && j == 0)) { traceback_slow_step(g, nr, sub, seq, m, x, gp, i, j, n); dbg[2] += 1; }
        }
        i = bcast0(i); j = bcast0(j); n = bcast0(n);
        __syncthreads();
    }
    g.overflow = bcast0(g.overflow);
    return n;
}

__global__ __launch_bounds__(64, 2) void poa_window_kernel(KParams P) {
    const int lane = threadIdx.x;
    extern __shared__ int4 lds[];                    // kLdsBytes per wave: DP row ring / traceback tile
    Win g;
    win_bind(g, P.scratch + static_cast<uint64_t>(blockIdx.x) * P.slot_bytes, P.ncap, P.ecap, P.ring, P.lmax, P.hstride);
    unsigned long long st_cells = 0, st_pred = 0, st_bytes = 0, st_ties = 0;
    unsigned long long dbg[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // sub, desc, dp, traceback, add, toposort, consensus, other
    long long tck = clock64();
#define RCN_PHASE(k) do { long long now__ = clock64(); ph[k] += now__ - tck; tck = now__; } while (0)

    for (;;) {
        RCN_PHASE(7);
        unsigned int wi = 0;
        if (lane == 0) wi = atomicAdd(P.next, 1u);
        wi = bcast0(wi);
        if (wi >= P.n_work) break;
        const uint32_t w = P.win_ids ? P.win_ids[wi] : wi;
        const uint32_t s0 = P.win_seq_off[w];
        const int ns = static_cast<int>(P.win_seq_off[w + 1] - s0);
        const uint8_t* bb = P.bases + P.seq_off[s0];
        const int L = static_cast<int>(P.seq_off[s0 + 1] - P.seq_off[s0]);
        uint8_t* out = P.out_cons + static_cast<uint64_t>(wi) * P.out_stride;   // outputs are indexed by work item

        if (ns < 3) {                                          // window.cpp:68-71
            for (int i = lane; i < L; i += 64) out[i] = bb[i];
            if (lane == 0) { P.out_len[wi] = L; P.out_flags[wi] = 0; }
            continue;
        }
        // ---- backbone -> graph (window.cpp:73-77) ----
        g.n_nodes = L; g.n_edges = L - 1; g.overflow = 0;
        {
            const uint8_t* q0 = P.seq_has_qual[s0] ? P.quals + P.seq_off[s0] : nullptr;
            for (int i = lane; i < L; i += 64) {
                g.code[i] = bb[i]; g.al_cnt[i] = 0;
                g.in_head[i] = g.in_tail[i] = (i > 0) ? i - 1 : -1;
                g.out_head[i] = g.out_tail[i] = (i < L - 1) ? i : -1;
                g.cov[i] = L >= 2 ? 1u : 0u;
                g.rank_full[i] = i; g.n2r[i] = i;
                if (i < L - 1) {
                    g.e_tail[i] = i; g.e_head[i] = i + 1; g.e_nin[i] = -1; g.e_nout[i] = -1;
                    g.e_w[i] = pair_weight(q0, i + 1);
                }
            }
        }
        wave_sync();

        for (int jl = 1; jl < ns && !g.overflow; ++jl) {
            const uint32_t si = s0 + P.order[s0 + jl];
            const uint8_t* seq = P.bases + P.seq_off[si];
            const uint8_t* qual = P.seq_has_qual[si] ? P.quals + P.seq_off[si] : nullptr;
            const int len = static_cast<int>(P.seq_off[si + 1] - P.seq_off[si]);
            const bool sub = P.seq_full[si] == 0;
            int V = g.n_nodes;
            const int32_t* rank = g.rank_full.ptr();
            Arr<int32_t> nr = g.n2r;
            if (sub) {
                // Subgraph (window.cpp:99-103): reachability mask on lane 0, then the sub order is
                // rank_full filtered by the mask (any valid order restricted to a subset stays valid)
                if (lane == 0) graph_subgraph_mask(g, static_cast<int32_t>(P.seq_begin[si]), static_cast<int32_t>(P.seq_end[si]), g.stack.ptr());
                wave_sync();
                int nv = 0;
                for (int base = 0; base < g.n_nodes; base += 64) {
                    const int r = base + lane;
                    const int v = r < g.n_nodes ? g.rank_full[r] : -1;
                    const bool in = v >= 0 && g.inc[v] != 0;
                    const unsigned long long mk = __ballot(in);
                    if (in) {
                        const int pos = nv + __popcll(mk & ((1ull << lane) - 1ull));
                        g.rank_sub[pos] = v; g.n2r_x[v] = pos;
                    }
                    nv += __popcll(mk);
                }
                V = nv;
                rank = g.rank_sub.ptr(); nr = g.n2r_x;
                wave_sync();
            }
            RCN_PHASE(0);
            // ---- row descriptors ----
            for (int r = lane; r < V; r += 64) g.desc[r] = make_row_desc(g, nr, rank[r], sub);
            for (int j = lane; j < g.hstride; j += 64) g.H[j] = j * P.g;
            wave_sync();
            RCN_PHASE(1);
            // ---- DP ----
            DpState ds; ds.best = 0; ds.best_row = 0; ds.have_best = 0; ds.tied = 0; ds.pred_rows = 0;
            const int W = len + 1;
            for (int t0 = 0; t0 < W;) {
                const int need = (W - t0 + 63) / 64;
                int ct = (need + 1) & ~1;
                if (ct > kMaxCT) ct = kMaxCT;
                const bool lastt = t0 + 64 * ct >= W;
                switch (ct) {
#define RCN_CASE(C) case C: ds = (t0 == 0) ? dp_tile<C, true>(g, nr, V, sub, seq, len, 0, lastt, P.m, P.x, P.g, ds, reinterpret_cast<int2*>(lds)) \
                                           : dp_tile<C, false>(g, nr, V, sub, seq, len, t0, lastt, P.m, P.x, P.g, ds, reinterpret_cast<int2*>(lds)); break;
                    RCN_CASE(2) RCN_CASE(4) RCN_CASE(6) RCN_CASE(8) RCN_CASE(10)
#undef RCN_CASE
                }
                t0 += 64 * ct;
                wave_sync();
            }
            RCN_PHASE(2);
            st_cells += static_cast<unsigned long long>(V + 1) * W;
            st_pred += static_cast<unsigned long long>(ds.pred_rows) * W;
            {   // SURVEY 8(d) yardstick: every cell written once + every predecessor row read once per
                // in-edge, at 2 B/cell when the worst-case score bound fits int16, else 4 B/cell
                const int amax = max(max(abs(P.m), abs(P.x)), abs(P.g));
                const unsigned long long sbytes = (static_cast<long long>(amax) * (V + W) < 32767) ? 2ull : 4ull;
                st_bytes += sbytes * (static_cast<unsigned long long>(V + 1) + ds.pred_rows) * W;
            }
            // ---- traceback (serial, lane 0) ----
            const int n_old = g.n_nodes;
            int nn = 0, plen = 0;
            int best_row = ds.best_row;
            if (ds.tied > 1) {
                // several sinks share the best score: spoa takes the first one in ITS rank order
                // (exact DFS order), so compute that order now (rare: ~2% of alignments)
                if (lane == 0) {
                    const int nx = graph_toposort(g, g.rank_x.ptr(), sub, g.stack.ptr());
                    for (int r = 0; r < nx; ++r) {
                        const int row = nr[g.rank_x[r]] + 1;
                        if ((g.desc[row - 1].meta & 256) && g.H[static_cast<int64_t>(row) * g.hstride + len] == ds.best) { best_row = row; break; }
                    }
                    ++st_ties;
                }
                best_row = bcast0(best_row);
            }
            plen = traceback_tiled(g, nr, sub, seq, len, best_row, P.m, P.x, P.g, reinterpret_cast<int*>(lds), dbg);
            plen = bcast0(plen); g.overflow = bcast0(g.overflow);
            wave_sync();
            RCN_PHASE(3);
            // ---- AddAlignment, wave-parallel over sequence positions (window.cpp:110-119) ----
            if (!g.overflow) {
                const uint32_t count = len >= 2 ? 1u : 0u;
                for (int k = lane; k < plen; k += 64) {
                    const int pp = g.path_pos[k];
                    if (pp != -1) { const int row = g.path_node[k]; g.pos_t[pp] = row == -1 ? -1 : rank[row - 1]; }
                }
                wave_sync();
                // classify positions; number the new nodes (prefix count) and propagate order anchors (prefix max)
                int32_t* kindv = g.path_pos.ptr();           // path arrays are free from here on
                int32_t* idxv = g.path_node.ptr();
                int anchor = -1;
                const unsigned long long lt = (1ull << lane) - 1ull;
                for (int base = 0; base < len; base += 64) {
                    const int pos = base + lane;
                    int kind = 0, a = -1;
                    if (pos < len) { kind = addp_classify(g, seq, pos); a = g.pos_a[pos]; }
                    const unsigned long long mk = __ballot(kind != 0);
                    const int idx = nn + __popcll(mk & lt);
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(a, d); if (lane >= d) a = max(a, t); }
                    a = max(a, anchor);
                    if (pos < len) { kindv[pos] = kind; idxv[pos] = idx; g.pos_a[pos] = a; }
                    nn += __popcll(mk);
                    anchor = __shfl(a, 63);
                }
                if (n_old + nn > g.ncap) g.overflow = 1;
                wave_sync();
                if (!g.overflow) {
                    for (int pos = lane; pos < len; pos += 64) {
                        const int kind = kindv[pos];
                        if (kind) {
                            const int idx = idxv[pos];
                            addp_create(g, seq, pos, kind, n_old + idx, count);
                            g.new_id[idx] = n_old + idx; g.new_anchor[idx] = g.pos_a[pos];
                        }
                    }
                    g.n_nodes = n_old + nn;
                    wave_sync();
                    int ne = 0, ovf = 0;
                    for (int base = 0; base < len; base += 64) {
                        const int pos = base + lane;
                        int f = 0;
                        if (pos >= 1 && pos < len) f = addp_edge_find(g, qual, pos);
                        const unsigned long long mk = __ballot(f != 0);
                        const int e = g.n_edges + ne + __popcll(mk & lt);
                        if (f) { if (e < g.ecap) addp_edge_create(g, qual, pos, e); else ovf = 1; }
                        ne += __popcll(mk);
                    }
                    g.n_edges += ne;
                    if (__ballot(ovf != 0)) g.overflow = 1;
                    for (int pos = lane; pos < len; pos += 64) g.cov[g.pos_curr[pos]] += count;
                }
            }
            wave_sync();
            RCN_PHASE(4);
            // ---- order merge: insert the nn new nodes behind their anchors (wave-parallel) ----
            if (!g.overflow) {
                int32_t* delta = g.pred.ptr();                      // [n_old + 1] scratch (pred is consensus-only)
                for (int r = lane; r <= n_old; r += 64) delta[r] = 0;
                wave_sync();
                for (int k = lane; k < nn; k += 64) {
                    const int a = g.new_anchor[k] + 1;
                    atomicAdd(&delta[a], 1);
                    const int v = g.new_id[k];
                    g.rank_tmp[a + k] = v; g.n2r[v] = a + k;
                }
                wave_sync();
                int carry = 0;
                for (int base = 0; base < n_old; base += 64) {
                    const int r = base + lane;
                    int sc = r < n_old ? delta[r] : 0;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(sc, d); if (lane >= d) sc += t; }
                    if (r < n_old) { const int v = g.rank_full[r]; const int pos = r + carry + sc; g.rank_tmp[pos] = v; g.n2r[v] = pos; }
                    carry += __shfl(sc, 63);
                }
                { const Arr<int32_t> t = g.rank_full; g.rank_full = g.rank_tmp; g.rank_tmp = t; }
                wave_sync();
            }
            RCN_PHASE(5);
        }

        if (g.overflow) {
            if (lane == 0) { P.out_len[wi] = 0; P.out_flags[wi] = (g.overflow == 1 || g.overflow == 3) ? kFlagOverflow : kFlagError; }
            continue;
        }
        // ---- consensus (window.cpp:122-146): needs spoa's exact rank order once ----
        int nx = 0;
        if (lane == 0) nx = graph_toposort(g, g.rank_x.ptr(), false, g.stack.ptr());
        nx = bcast0(nx);
        wave_sync();
        if (nx != g.n_nodes) {
            if (lane == 0) { P.out_len[wi] = 0; P.out_flags[wi] = kFlagError; }
            continue;
        }
        for (int r = lane; r < g.n_nodes; r += 64) g.n2r_x[g.rank_x[r]] = r;
        wave_sync();
        int clen = 0, cb = 0, flags = kFlagPolished;
        if (lane == 0) {
            int32_t* cn = g.path_node.ptr();
            const int k = graph_consensus(g, g.rank_x.ptr(), g.n2r_x, cn);
            int bgn = 0, end = k - 1;
            if (P.win_type[w] == 1 && P.trim) {
                const uint32_t avg = static_cast<uint32_t>(ns - 1) / 2;
                for (; bgn < k; ++bgn) if (consensus_coverage(g, cn[bgn]) >= avg) break;
                for (; end >= 0; --end) if (consensus_coverage(g, cn[end]) >= avg) break;
                if (bgn >= end) { bgn = 0; end = k - 1; flags |= kFlagChimeric; }
            }
            cb = bgn; clen = end - bgn + 1;
        }
        clen = bcast0(clen); cb = bcast0(cb); flags = bcast0(flags);
        wave_sync();
        if (static_cast<uint64_t>(clen) > P.out_stride) {
            if (lane == 0) { P.out_len[wi] = 0; P.out_flags[wi] = kFlagOverflow; }
            continue;
        }
        for (int t = lane; t < clen; t += 64) out[t] = g.code[g.path_node[cb + t]];
        if (lane == 0) { P.out_len[wi] = clen; P.out_flags[wi] = static_cast<uint8_t>(flags); }
        wave_sync();
        RCN_PHASE(6);
    }
    if (lane == 0) { atomicAdd(&P.stats[0], st_cells); atomicAdd(&P.stats[1], st_pred); atomicAdd(&P.stats[2], st_bytes);
                     for (int k = 0; k < 8; ++k) atomicAdd(&P.stats[3 + k], ph[k]);
                     atomicAdd(&P.stats[11], st_ties);
                     for (int k = 0; k < 7; ++k) atomicAdd(&P.stats[12 + k], dbg[k]); }
}

}  // namespace rcn
