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
    const uint8_t* win_flags;     // [n_windows] bit 0: every base of the window is A, C, G or T (poa_window_kernel2: profile table); nullptr = unknown
    const uint32_t* win_ids;      // [n_work] indirection (work item -> window: deepest-first order, retry pass) or nullptr
    uint32_t n_work;
    uint32_t work_base;           // without win_ids: work item wi is window work_base + wi (streamed sub-launches)
    int32_t force_exact;          // poa_window_kernel2: 1 = every window takes the exact-order consensus path (tests; env RCN_FORCE_EXACT)
    int32_t heavy_ns;             // poa_window_kernel2: windows with at least this many sequences use the 4-wave DP (0 = none)
    int32_t force_tie;            // poa_window_kernel2, tests (env RCN_FORCE_TIE): 2 = every sink tie skips the id / backbone-position
                                  // rule (levels 2a/2b decide), 3 = every sink tie takes the full DFS (phase_sink_tie_full)
    int32_t band;                 // poa_window_kernel2: 1 = exact banded DP where it applies, leaving move codes (default), 0 = never
                                  // (env RCN_NO_BAND), 2 = banded pass runs but every certificate is treated as failed (tests the redo
                                  // path; RCN_FORCE_BAND_FAIL), 3 = banded DP that stores the scores (RCN_BAND_SCORES)
    int32_t force_slow_tb;        // poa_window_kernel2, tests (env RCN_FORCE_SLOW_TB): every traceback step is the one-cell step
                                  // against HBM (traceback2_slow_step) instead of the box walk over the staged tile
    int32_t m, x, g, trim;
    // per-slot scratch
    uint8_t* scratch; uint64_t slot_bytes; int32_t ncap, ecap, ring, lmax, hstride;
    int32_t hrows;                // rows of a slot's DP matrix (win_bind); poa_window_kernel2 flags a window whose alignment needs more
    int32_t lds_extra;            // poa_window_kernel2: bytes of dynamic LDS behind the work area + context (a launch with fewer than eight
                                  // work-groups per CU asks for more LDS to get there; with >= kHelpLdsBytes of it the banded DP runs with its
                                  // code wave, poa_band.hpp)
    int32_t no_help;              // tests / A-B (env RCN_NO_CODE_WAVE): never the code wave
    // outputs
    // outputs, indexed by out_base + work item: consensus bytes at out_cons + out_off[k], capacity out_off[k + 1] - out_off[k]
    // (a consensus that does not fit is flagged kFlagOverflow and redone by the retry pass)
    uint8_t* out_cons; const uint64_t* out_off; uint32_t out_base; uint32_t* out_len; uint8_t* out_flags;
    // queue + counters
    unsigned int* next; unsigned long long* stats;   // stats[0]=cells, [1]=pred cells, [2]=algorithmic DP bytes, [3..10]=phase clocks, [11]=sink ties,
                                                      // [12]=cells of the full matrices, [13]=bytes of the full matrices, [14]=banded alignments, [15]=band redos
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
constexpr int kMaxCT = 12;   // widest column tile: 64 * 12 = 768 columns per pass
constexpr int kLdsBytes = 19968;               // work area per wave (+512 B context = 20 KiB: 8 waves / CU)
constexpr int kRingInts = kLdsBytes / 4;
struct DpState { int best, best_row, have_best, tied; unsigned int pred_rows; };

// wave-uniform, global-address-space copy of a pointer that arrived in VGPRs (function arguments
// do) or came back from LDS
template <class T>
__device__ __forceinline__ RCN_G T* uptr(T* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return (RCN_G T*)((static_cast<uint64_t>(hi) << 32) | lo);
}
template <class T>
__device__ __forceinline__ RCN_G T* gcast(T* p) { return (RCN_G T*)p; }
__device__ __forceinline__ int uint_(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Everything dp_tile touches in HBM, passed by value (a reference to the caller's Win would
// force that struct through scratch memory).
struct DpMem {
    int32_t* H; const RowDesc* desc; const int32_t* e_nin; const int32_t* e_tail; const uint8_t* inc; const int32_t* nr;
    int32_t hstride;
};
struct DpMemG {
    RCN_G int32_t* H; RCN_G const RowDesc* desc; RCN_G const int32_t* e_nin; RCN_G const int32_t* e_tail;
    RCN_G const uint8_t* inc; RCN_G const int32_t* nr; int32_t hstride;
};

template <int CT, bool FIRST>      // FIRST: the tile starts at column 0 (no global loads in the row loop)
__device__ __noinline__ DpState dp_tile(DpMem mem_in, int V, bool sub, const uint8_t* seq_in, int len,
                                        int t0, bool last_tile, int m, int x, int gp, DpState st, int2* __restrict__ ring) {
    DpMemG mem;
    mem.H = uptr(mem_in.H); mem.desc = uptr(mem_in.desc); mem.e_nin = uptr(mem_in.e_nin); mem.e_tail = uptr(mem_in.e_tail);
    mem.inc = uptr(mem_in.inc); mem.nr = uptr(mem_in.nr); mem.hstride = uint_(mem_in.hstride);
    RCN_G const uint8_t* seq = uptr(seq_in);
    V = uint_(V); sub = uint_(sub) != 0; len = uint_(len); t0 = uint_(t0); last_tile = uint_(last_tile) != 0;
    m = uint_(m); x = uint_(x); gp = uint_(gp);
    st.best = uint_(st.best); st.best_row = uint_(st.best_row); st.have_best = uint_(st.have_best); st.tied = uint_(st.tied);
    st.pred_rows = uint_(st.pred_rows);
    // LDS ring of the last K score rows (predecessors are almost always < 16 rows back in the
    // incrementally maintained order).  Layout: slot s, column pair q of lane l at
    // ring[(s * (CT/2) + q) * 64 + l]: every ds_read/write_b64 is stride-8B, conflict free.
    constexpr int K = kRingInts / (64 * CT) - 1;                // + one spare slot (index K) for far rows
    int slot = 0;                                               // ring slot of row i (row 0 is analytic)
    int best = st.best, best_row = st.best_row, have_best = st.have_best, tied = st.tied;
    unsigned int pred_rows = 0;
    const int lane = threadIdx.x;
    const int j0 = t0 + lane * CT;
    const int64_t hs = mem.hstride;
    RCN_G int32_t* __restrict__ H = mem.H;

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
    if (rbase + lane < V) dl = mem.desc[rbase + lane];
    // retire the descriptor load HERE: otherwise the compiler parks its s_waitcnt vmcnt(0) at the first
    // v_readlane inside the row loop, where it also waits for every outstanding H-row store, every row
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int rend = min(V, rbase + 64);
#pragma unroll 1
    for (int r = rbase; r < rend; ++r) {
        const int k = r - rbase;
        const int p0 = __builtin_amdgcn_readlane(dl.p[0], k);
        const int er = __builtin_amdgcn_readlane(dl.erest, k);
        const int meta = __builtin_amdgcn_readlane(dl.meta, k);
        const int np = (meta >> 9) & 7;
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
                    RCN_G const int2* gsrc = reinterpret_cast<RCN_G const int2*>(H + p * hs + j0);
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
            // further inline predecessors: one call site in a uniform loop (keeps the code and the
            // register footprint of the row loop small)
            const int q1 = __builtin_amdgcn_readlane(dl.p[1], k), q2 = __builtin_amdgcn_readlane(dl.p[2], k);
            const int q3 = __builtin_amdgcn_readlane(dl.p[3], k), q4 = __builtin_amdgcn_readlane(dl.p[4], k);
            const int q5 = __builtin_amdgcn_readlane(dl.p[5], k);
#pragma unroll 1
            for (int q = 1; q < np; ++q) accumulate(q == 1 ? q1 : q == 2 ? q2 : q == 3 ? q3 : q == 4 ? q4 : q5);
        }
        for (int e = er; e >= 0; e = mem.e_nin[e]) {
            const int t = mem.e_tail[e];
            if (sub && !mem.inc[t]) continue;
            accumulate(mem.nr[t] + 1);
        }
        // horizontal gap: in-lane pass, then wave-wide prefix max of the transformed lane tails
#pragma unroll
        for (int c = 1; c < CT; ++c) acc[c] = max(acc[c], acc[c - 1] + gp);
        int z = acc[CT - 1] - (j0 + CT - 1) * gp;
        int zex = wave_excl_scan_max(z, lane);
        if (!FIRST) zex = max(zex, H[i * hs + t0 - 1] - (t0 - 1) * gp);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = max(acc[c], zex + (j0 + c) * gp);

        RCN_G int2* dst = reinterpret_cast<RCN_G int2*>(H + i * hs + j0);
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
constexpr int kTileStride = 64;   // ints per tile row (rows are filled by direct global->LDS loads, 1 KiB = 4 rows each)

__device__ __forceinline__ void traceback_slow_step(Win& g, const Arr<int32_t>& nr, bool sub, RCN_G const uint8_t* seq,
                                                    int m, int x, int gp, int& i, int& j, int& n) {
    const int64_t hs = g.hstride;
    RCN_G const int32_t* H = g.H.ptr();
    const int hij = H[i * hs + j];
    int pi = 0, pj = 0; bool found = false;
    if (i != 0) {
        const RowDesc d = g.desc[i - 1];
        const int np = (d.meta >> 9) & 7;
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

__device__ __forceinline__ int traceback_tiled(Win& g, const Arr<int32_t>& nr, bool sub, RCN_G const uint8_t* __restrict__ seq,
                                               int len, int best_row, int m, int x, int gp, int* __restrict__ tile) {
    const int lane = threadIdx.x;
    const int64_t hs = g.hstride;
    RCN_G const int32_t* __restrict__ H = g.H.ptr();
    int* tdesc = tile + 64 * kTileStride;                                   // 64 x RowDesc (8 ints each)
    uint8_t* tseq = reinterpret_cast<uint8_t*>(tile + 64 * kTileStride + 64 * (sizeof(RowDesc) / 4));   // seq[c0 - 1 + k]
    RCN_G int32_t* __restrict__ pnode = g.path_node.ptr();
    RCN_G int32_t* __restrict__ ppos = g.path_pos.ptr();
    int i = best_row, j = len, n = 0;               // wave-uniform walk state
    while (!(i == 0 && j == 0)) {
        // ---- stage the tile: rows [i-63, i], cols [c0, c0+63]; all 16 row loads of a lane in flight ----
        const int ti0 = i;
        int c0 = (j - 60) & ~3; if (c0 < 0) c0 = 0;
        const int rmin = ti0 - 63 > 0 ? ti0 - 63 : 0;
        {
            // H tile: 16 direct global->LDS loads (global_load_lds_dwordx4: lane l deposits its 16 B at
            // LDS base + 16*l, no VGPR round trip), all in flight together; instruction k fills tile
            // rows 4k..4k+3 (tile row t holds matrix row ti0 - t).
            typedef __attribute__((address_space(3))) void* lds_ptr;
            const int sub_row = lane >> 4, chunk = lane & 15;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                int r = ti0 - (4 * k + sub_row); if (r < 0) r = 0;
                RCN_G const int32_t* src = H + r * hs + c0 + chunk * 4;
                __builtin_amdgcn_global_load_lds(src, (lds_ptr)(tile + k * 256), 16, 0, 0);
            }
            const int r = ti0 - lane;
            int4 d0 = make_int4(0, -1, -1, -1), d1 = make_int4(-1, -1, -1, 1 << 9);
            if (r >= 1) { RCN_G const int4* dsrc = reinterpret_cast<RCN_G const int4*>(g.desc.ptr() + (r - 1)); d0 = dsrc[0]; d1 = dsrc[1]; }
            int4* ddst = reinterpret_cast<int4*>(tdesc + lane * 8);
            ddst[0] = d0; ddst[1] = d1;
            const int sc = c0 - 1 + lane;
            tseq[lane] = (sc >= 0 && sc < len) ? seq[sc] : 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- walk inside the tile: wave-uniform control flow; lane q looks at predecessor q ----
        int steps = 0, bn = 0, bp = 0, n0 = n;      // up to 64 path entries buffered one per lane
        int hij = tile[j - c0];                      // row ti0 is tile row 0
        for (;;) {
            if (i == 0 && j == 0) break;
            int pi, pj, hnext;
            if (i == 0) {                            // only horizontal moves are left on the virtual row
                if (j - 1 < c0) break;
                pi = 0; pj = j - 1; hnext = hij - gp;
            } else {
                const int* dr = tdesc + (ti0 - i) * 8;
                const int pq = dr[lane < kInlinePreds ? lane : 0];
                const int erest = dr[6], meta = dr[7];
                const int symc = tseq[j - c0];                               // seq[j-1]
                const int np = (meta >> 9) & 7;
                const bool valid = lane < np;
                if (__ballot(valid && pq < rmin) != 0ull || erest >= 0 || (j > 0 && j - 1 < c0)) break;   // leaves the tile
                const int mc = ((meta & 255) == symc) ? m : x;
                const int* cp = tile + (ti0 - (valid ? pq : ti0)) * kTileStride + (j - c0);
                const int hd = cp[j > 0 ? -1 : 0], hu = cp[0];
                const unsigned long long dmask = __ballot(valid && j > 0 && hij == hd + mc);
                const unsigned long long umask = __ballot(valid && hij == hu + gp);
                if (dmask) { const int q = __builtin_ctzll(dmask); pi = __builtin_amdgcn_readlane(pq, q); pj = j - 1; hnext = __builtin_amdgcn_readlane(hd, q); }
                else if (umask) { const int q = __builtin_ctzll(umask); pi = __builtin_amdgcn_readlane(pq, q); pj = j; hnext = __builtin_amdgcn_readlane(hu, q); }
                else { if (j == 0) { g.overflow = 4; i = 0; j = 0; break; } pi = i; pj = j - 1; hnext = hij - gp; }
            }
            if (lane == steps) { bn = (i == pi) ? -1 : i; bp = (j == pj) ? -1 : j - 1; }   // ROW index (node id later)
            ++steps;
            i = pi; j = pj; hij = hnext;
            if (steps == 64) { pnode[n0 + lane] = bn; ppos[n0 + lane] = bp; n0 += 64; steps = 0; }
        }
        if (lane < steps) { pnode[n0 + lane] = bn; ppos[n0 + lane] = bp; }
        const bool progressed = (n0 + steps) != n;
        n = n0 + steps;
        if (!progressed && !(i == 0 && j == 0)) {
            if (lane == 0) traceback_slow_step(g, nr, sub, seq, m, x, gp, i, j, n);
            i = bcast0(i); j = bcast0(j); n = bcast0(n); g.overflow = bcast0(g.overflow);
        }
        __syncthreads();
    }
    return n;
}

// ---------------------------------------------------------------------------
// Per-slot context.  It lives in LDS behind the kLdsBytes work area, so that each
// phase below can be its own NON-inlined device function with its own register
// budget: a phase re-derives the scratch layout (win_bind is pure arithmetic on a
// handful of uniform values) instead of inheriting ~50 live pointers from one
// giant kernel body.  All fields are wave-uniform.
// ---------------------------------------------------------------------------
struct Ctx {
    // window constants
    uint8_t* scratch; int32_t ncap, ecap, ring, lmax, hstride;
    int32_t m, x, gp, trim, pad0;
    // window state
    int32_t n_nodes, n_edges, overflow, swapped;
    // layer
    const uint8_t* seq; const uint8_t* qual;
    int32_t len, sub, begin, end;
    int32_t V, best, best_row, tied;
    int32_t plen, nn, n_old; uint32_t pred_rows;
    // statistics
    unsigned long long cells, pred, bytes, ties;
    // work-group shared scalars of poa_window_kernel2 (work item, traceback walk state)
    int32_t wi, tb_i, tb_j, tb_n;
    int32_t dbg_tiles, dbg_boxes, dbg_slow, bblen;
    int32_t tie_rows[8];          // rows of the sinks that share the best score (first 8)
    int32_t tie_why, tie_pad[3];   // tie_pad[0]: KParams::band, [1]: window is ACGT-only, [2]: KParams::force_slow_tb
    // exact banded DP (poa_band.hpp): NP of the window for the current alignment (0 = full rows), certificate verdict, counters
    int32_t band, band_fail, coded;                // coded: the finished alignment left move codes, not scores (phase_traceback_code)
    unsigned long long cells_full, bytes_full;     // the full-matrix figures next to the evaluated ones (cells / bytes)
    unsigned int n_banded, n_band_fail;
    unsigned int band_why, band_whyn[8];            // reasons of the redos (bit k of dp2_rows_band's `why`), counted
    int32_t hrows;                                  // KParams::hrows
    unsigned int n_help;                            // banded alignments done with the code waves
    int32_t big;                                    // the work-group owns the large LDS ring + mailbox of the code wave (KParams::lds_extra)
};
static_assert(sizeof(Ctx) % 4 == 0 && sizeof(Ctx) <= 512, "Ctx must fit its LDS slot");
constexpr int kCtxBytes = 512;

__device__ __forceinline__ int* lds_words() { extern __shared__ int4 lds_dyn[]; return reinterpret_cast<int*>(lds_dyn); }
__device__ __forceinline__ Ctx* ctx_lds() { return reinterpret_cast<Ctx*>(lds_words() + kLdsBytes / 4); }

// Execution policies of the shared phases.  A phase is written for NT cooperating threads with
// ids tid() in [0, NT); sync() orders their global/LDS traffic between sub-steps.
//   OneWaveBlock : the 64-thread workgroup of poa_window_kernel (one wave per window)
//   (poa_kernel2.hpp adds: one wave of a 4-wave workgroup, and the whole 4-wave workgroup)
struct OneWaveBlock {
    static constexpr int NT = 64;
    static __device__ __forceinline__ int tid() { return threadIdx.x; }
    static __device__ __forceinline__ void sync() { __threadfence_block(); __syncthreads(); }
    static __device__ __forceinline__ Ctx* ctx() { return ctx_lds(); }
    static __device__ __forceinline__ int* work() { return lds_words(); }
};

// Loads the context from LDS and makes every dword provably wave-uniform (SGPR).
template <class S>
__device__ __forceinline__ Ctx ctx_load() {
    Ctx c;
    const int* src = reinterpret_cast<const int*>(S::ctx());
    int* dst = reinterpret_cast<int*>(&c);
#pragma unroll
    for (int k = 0; k < static_cast<int>(sizeof(Ctx) / 4); ++k) dst[k] = __builtin_amdgcn_readfirstlane(src[k]);
    return c;
}
__device__ __forceinline__ Win ctx_win(const Ctx& c) {
    Win g;
    win_bind(g, gcast(c.scratch), c.ncap, c.ecap, c.ring, c.lmax, c.hstride, 4, c.hrows);
    g.n_nodes = c.n_nodes; g.n_edges = c.n_edges; g.overflow = c.overflow;
    if (c.swapped) { const Arr<int32_t> t = g.rank_full; g.rank_full = g.rank_tmp; g.rank_tmp = t; }
    return g;
}

// ---- phase: Subgraph mask + filtered order (window.cpp:99-103) ----
template <class S>
__device__ __noinline__ void phase_subgraph() {
    static_assert(S::NT == 64, "single-wave phase");
    const int lane = S::tid();
    const Ctx c = ctx_load<S>();
    Win g = ctx_win(c);
    if (lane == 0) graph_subgraph_mask(g, c.begin, c.end, g.stack.ptr());
    S::sync();
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
    if (lane == 0) S::ctx()->V = nv;
    S::sync();
}

// ---- phase: row descriptors + row 0 ----
__device__ __noinline__ void phase_desc() {
    const int lane = threadIdx.x;
    const Ctx c = ctx_load<OneWaveBlock>();
    Win g = ctx_win(c);
    RCN_G const int32_t* rank = c.sub ? g.rank_sub.ptr() : g.rank_full.ptr();
    const Arr<int32_t> nr = c.sub ? g.n2r_x : g.n2r;
    for (int r = lane; r < c.V; r += 64) g.desc[r] = make_row_desc(g, nr, rank[r], c.sub != 0);
    for (int j = lane; j < g.hstride; j += 64) g.H[j] = j * c.gp;
    wave_sync();
}

// ---- phase: NW sequence-to-graph DP (window.cpp:95-97, 104-106) ----
__device__ __noinline__ void phase_dp() {
    const int lane = threadIdx.x;
    const Ctx c = ctx_load<OneWaveBlock>();
    Win g = ctx_win(c);
    const Arr<int32_t> nr = c.sub ? g.n2r_x : g.n2r;
    DpState ds; ds.best = 0; ds.best_row = 0; ds.have_best = 0; ds.tied = 0; ds.pred_rows = 0;
    const int W = c.len + 1;
    int2* ring = reinterpret_cast<int2*>(lds_words());
    DpMem mem; mem.H = (int32_t*)g.H.ptr(); mem.desc = (const RowDesc*)g.desc.ptr(); mem.e_nin = (const int32_t*)g.e_nin.ptr();
    mem.e_tail = (const int32_t*)g.e_tail.ptr(); mem.inc = (const uint8_t*)g.inc.ptr(); mem.nr = (const int32_t*)nr.ptr(); mem.hstride = g.hstride;
    for (int t0 = 0; t0 < W;) {
        const int need = (W - t0 + 63) / 64;
        int ct = (need + 1) & ~1;
        if (ct > kMaxCT) ct = kMaxCT;
        const bool lastt = t0 + 64 * ct >= W;
        switch (ct) {
#define RCN_CASE(C) case C: ds = (t0 == 0) ? dp_tile<C, true>(mem, c.V, c.sub != 0, c.seq, c.len, 0, lastt, c.m, c.x, c.gp, ds, ring) \
                                           : dp_tile<C, false>(mem, c.V, c.sub != 0, c.seq, c.len, t0, lastt, c.m, c.x, c.gp, ds, ring); break;
            RCN_CASE(2) RCN_CASE(4) RCN_CASE(6) RCN_CASE(8) RCN_CASE(10) RCN_CASE(12)
#undef RCN_CASE
        }
        t0 += 64 * ct;
        wave_sync();
    }
    if (lane == 0) {
        Ctx* o = ctx_lds();
        o->best = ds.best; o->best_row = ds.best_row; o->tied = ds.tied; o->pred_rows = ds.pred_rows;
        o->cells += static_cast<unsigned long long>(c.V + 1) * W;
        o->pred += static_cast<unsigned long long>(ds.pred_rows) * W;
        // SURVEY 8(d) yardstick: every cell written once + every predecessor row read once per
        // in-edge, at 2 B/cell when the worst-case score bound fits int16, else 4 B/cell
        const int amax = max(max(abs(c.m), abs(c.x)), abs(c.gp));
        const unsigned long long sbytes = (static_cast<long long>(amax) * (c.V + W) < 32767) ? 2ull : 4ull;
        o->bytes += sbytes * (static_cast<unsigned long long>(c.V + 1) + ds.pred_rows) * W;
        o->cells_full += static_cast<unsigned long long>(c.V + 1) * W;
        o->bytes_full += sbytes * (static_cast<unsigned long long>(c.V + 1) + ds.pred_rows) * W;
    }
    wave_sync();
}

// ---- phase: sink tie-break (rare) + traceback ----
__device__ __noinline__ void phase_traceback() {
    const int lane = threadIdx.x;
    const Ctx c = ctx_load<OneWaveBlock>();
    Win g = ctx_win(c);
    const Arr<int32_t> nr = c.sub ? g.n2r_x : g.n2r;
    int best_row = c.best_row;
    if (c.tied > 1) {
        // several sinks share the best score: spoa takes the first one in ITS rank order
        // (exact DFS order), so compute that order now (rare: ~2% of alignments)
        if (lane == 0) {
            const int nx = graph_toposort(g, g.rank_x.ptr(), c.sub != 0, g.stack.ptr());
            for (int r = 0; r < nx; ++r) {
                const int row = nr[g.rank_x[r]] + 1;
                if ((g.desc[row - 1].meta & 256) && g.H[static_cast<int64_t>(row) * g.hstride + c.len] == c.best) { best_row = row; break; }
            }
        }
        best_row = bcast0(best_row);
    }
    const int plen = traceback_tiled(g, nr, c.sub != 0, gcast(c.seq), c.len, best_row, c.m, c.x, c.gp, lds_words());
    if (lane == 0) {
        Ctx* o = ctx_lds();
        o->plen = plen; o->overflow = g.overflow;
        if (c.tied > 1) o->ties += 1;
    }
    wave_sync();
}

// ---- phase: AddAlignment, wave-parallel over sequence positions (window.cpp:110-119) ----
template <class S>
__device__ __noinline__ void phase_add() {
    static_assert(S::NT == 64, "single-wave phase");
    const int lane = S::tid();
    const Ctx c = ctx_load<S>();
    Win g = ctx_win(c);
    RCN_G const int32_t* rank = c.sub ? g.rank_sub.ptr() : g.rank_full.ptr();
    RCN_G const uint8_t* seq = gcast(c.seq); RCN_G const uint8_t* qual = gcast(c.qual);
    const int len = c.len, plen = c.plen, n_old = g.n_nodes;
    const uint32_t count = len >= 2 ? 1u : 0u;
    int nn = 0;
    if (plen >= 0) {
        for (int k = lane; k < plen; k += 64) {
            const int pp = g.path_pos[k];
            if (pp != -1) { const int row = g.path_node[k]; g.pos_t[pp] = row == -1 ? -1 : rank[row - 1]; }
        }
    } else {
        // poa_window_kernel2's traceback leaves, per sequence position, the DP row it is aligned to (-1 = none)
        for (int pos = lane; pos < len; pos += 64) { const int row = g.pos_t[pos]; g.pos_t[pos] = row <= 0 ? -1 : rank[row - 1]; }
    }
    S::sync();
    // classify positions; number the new nodes (prefix count) and propagate order anchors (prefix max)
    RCN_G int32_t* kindv = g.path_pos.ptr();     // path arrays are free from here on
    RCN_G int32_t* idxv = g.path_node.ptr();
    int anchor = -1;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = 0; base < len; base += 64) {
        const int pos = base + lane;
        int kind = 0, a = -1;
        if (pos < len) { kind = addp_classify(g, seq, pos); a = g.pos_a[pos]; }
        const unsigned long long mk = __ballot(kind != 0);
        const int idx = nn + __popcll(mk & lt);
        a = max(wave_incl_scan_max(a), anchor);
        if (pos < len) { kindv[pos] = kind; idxv[pos] = idx; g.pos_a[pos] = a; }
        nn += __popcll(mk);
        anchor = __builtin_amdgcn_readlane(a, 63);
    }
    if (n_old + nn > g.ncap) g.overflow = 1;
    S::sync();
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
        S::sync();
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
    if (lane == 0) {
        Ctx* o = S::ctx();
        o->n_old = n_old; o->nn = nn; o->n_nodes = g.n_nodes; o->n_edges = g.n_edges; o->overflow = g.overflow;
    }
    S::sync();
}

// ---- phase: order merge: insert the nn new nodes behind their anchors ----
template <class S>
__device__ __noinline__ void phase_merge() {
    static_assert(S::NT == 64, "single-wave phase");
    const int lane = S::tid();
    const Ctx c = ctx_load<S>();
    Win g = ctx_win(c);
    const int n_old = c.n_old, nn = c.nn;
    RCN_G int32_t* delta = g.pred.ptr();                // [n_old + 1] scratch (pred is consensus-only)
    for (int r = lane; r <= n_old; r += 64) delta[r] = 0;
    S::sync();
    for (int k = lane; k < nn; k += 64) {
        const int a = g.new_anchor[k] + 1;
        atomicAdd((int*)&delta[a], 1);
        const int v = g.new_id[k];
        g.rank_tmp[a + k] = v; g.n2r[v] = a + k;
    }
    S::sync();
    int carry = 0;
    for (int base = 0; base < n_old; base += 64) {
        const int r = base + lane;
        int sc = r < n_old ? delta[r] : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(sc, d); if (lane >= d) sc += t; }
        if (r < n_old) { const int v = g.rank_full[r]; const int pos = r + carry + sc; g.rank_tmp[pos] = v; g.n2r[v] = pos; }
        carry += __shfl(sc, 63);
    }
    if (lane == 0) S::ctx()->swapped = c.swapped ^ 1;
    S::sync();
}

// ---- phase: consensus + coverage + trim (window.cpp:122-146); returns via out arrays ----
template <class S>
__device__ __noinline__ void phase_consensus(uint8_t* out_in, uint64_t out_cap, uint32_t* out_len_in, uint8_t* out_flags_in, int ns, int tgs) {
    RCN_G uint8_t* out = uptr(out_in); RCN_G uint32_t* out_len = uptr(out_len_in); RCN_G uint8_t* out_flags = uptr(out_flags_in);
    out_cap = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(out_cap >> 32))) << 32) | __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(out_cap));
    ns = uint_(ns); tgs = uint_(tgs);
    const int lane = S::tid();
    const Ctx c = ctx_load<S>();
    Win g = ctx_win(c);
    int nx = 0;
    if (lane == 0) nx = graph_toposort(g, g.rank_x.ptr(), false, g.stack.ptr());   // spoa's exact rank order, once
    nx = bcast0(nx);
    S::sync();
    if (nx != g.n_nodes) { if (lane == 0) { *out_len = 0; *out_flags = kFlagError; } return; }
    for (int r = lane; r < g.n_nodes; r += 64) g.n2r_x[g.rank_x[r]] = r;
    S::sync();
    int clen = 0, cb = 0, flags = kFlagPolished;
    if (lane == 0) {
        RCN_G int32_t* cn = g.path_node.ptr();
        const int k = graph_consensus(g, g.rank_x.ptr(), g.n2r_x, cn);
        int bgn = 0, end = k - 1;
        if (tgs && c.trim) {
            const uint32_t avg = static_cast<uint32_t>(ns - 1) / 2;
            for (; bgn < k; ++bgn) if (consensus_coverage(g, cn[bgn]) >= avg) break;
            for (; end >= 0; --end) if (consensus_coverage(g, cn[end]) >= avg) break;
            if (bgn >= end) { bgn = 0; end = k - 1; flags |= kFlagChimeric; }
        }
        cb = bgn; clen = end - bgn + 1;
    }
    clen = bcast0(clen); cb = bcast0(cb); flags = bcast0(flags);
    S::sync();
    if (static_cast<uint64_t>(clen) > out_cap) { if (lane == 0) { *out_len = 0; *out_flags = kFlagOverflow; } return; }
    for (int t = lane; t < clen; t += 64) out[t] = g.code[g.path_node[cb + t]];
    if (lane == 0) { *out_len = clen; *out_flags = static_cast<uint8_t>(flags); }
    S::sync();
}

#ifndef RCN_DEEP_TU
__global__ __launch_bounds__(64, 2) void poa_window_kernel(KParams P) {
    const int lane = threadIdx.x;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // sub, desc, dp, traceback, add, merge, consensus, other
    long long tck = clock64();
#define RCN_PHASE(k) do { long long now__ = clock64(); ph[k] += now__ - tck; tck = now__; } while (0)
    Ctx* ctx = ctx_lds();
    if (lane == 0) {
        ctx->scratch = P.scratch + static_cast<uint64_t>(blockIdx.x) * P.slot_bytes;
        ctx->ncap = P.ncap; ctx->ecap = P.ecap; ctx->ring = P.ring; ctx->lmax = P.lmax; ctx->hstride = P.hstride; ctx->hrows = P.hrows;
        ctx->m = P.m; ctx->x = P.x; ctx->gp = P.g; ctx->trim = P.trim;
        ctx->cells = 0; ctx->pred = 0; ctx->bytes = 0; ctx->ties = 0; ctx->cells_full = 0; ctx->bytes_full = 0;
    }
    wave_sync();

    for (;;) {
        RCN_PHASE(7);
        unsigned int wi = 0;
        if (lane == 0) wi = atomicAdd(P.next, 1u);
        wi = bcast0(wi);
        if (wi >= P.n_work) break;
        const uint32_t w = P.win_ids ? P.win_ids[wi] : P.work_base + wi;
        const uint32_t s0 = P.win_seq_off[w];
        const int ns = static_cast<int>(P.win_seq_off[w + 1] - s0);
        const uint8_t* bb = P.bases + P.seq_off[s0];
        const int L = static_cast<int>(P.seq_off[s0 + 1] - P.seq_off[s0]);
        const uint32_t oi = P.out_base + wi;                                       // outputs are indexed by work item
        uint8_t* out = P.out_cons + P.out_off[oi];
        const uint64_t out_cap = P.out_off[oi + 1] - P.out_off[oi];

        if (ns < 3) {                                          // window.cpp:68-71
            for (int i = lane; i < L; i += 64) out[i] = bb[i];
            if (lane == 0) { P.out_len[oi] = L; P.out_flags[oi] = 0; }
            continue;
        }
        // ---- backbone -> graph (window.cpp:73-77) ----
        {
            Win g;
            win_bind(g, gcast(P.scratch + static_cast<uint64_t>(blockIdx.x) * P.slot_bytes), P.ncap, P.ecap, P.ring, P.lmax, P.hstride, 4, P.hrows);
            RCN_G const uint8_t* q0 = P.seq_has_qual[s0] ? gcast(P.quals + P.seq_off[s0]) : nullptr;
            for (int i = lane; i < L; i += 64) {
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
            if (lane == 0) { ctx->n_nodes = L; ctx->n_edges = L - 1; ctx->overflow = 0; ctx->swapped = 0; }
        }
        wave_sync();

        int overflow = 0;
        for (int jl = 1; jl < ns && !overflow; ++jl) {
            const uint32_t si = s0 + P.order[s0 + jl];
            if (lane == 0) {
                ctx->seq = P.bases + P.seq_off[si];
                ctx->qual = P.seq_has_qual[si] ? P.quals + P.seq_off[si] : nullptr;
                ctx->len = static_cast<int>(P.seq_off[si + 1] - P.seq_off[si]);
                ctx->sub = P.seq_full[si] == 0;
                ctx->begin = static_cast<int32_t>(P.seq_begin[si]); ctx->end = static_cast<int32_t>(P.seq_end[si]);
                ctx->V = ctx->n_nodes;
            }
            wave_sync();
            if (P.seq_full[si] == 0) phase_subgraph<OneWaveBlock>();
            RCN_PHASE(0);
            phase_desc();
            RCN_PHASE(1);
            phase_dp();
            RCN_PHASE(2);
            phase_traceback();
            RCN_PHASE(3);
            overflow = bcast0(ctx->overflow);
            if (!overflow) {
                phase_add<OneWaveBlock>();
                RCN_PHASE(4);
                overflow = bcast0(ctx->overflow);
                if (!overflow) phase_merge<OneWaveBlock>();
                RCN_PHASE(5);
            }
        }
        if (overflow) {
            if (lane == 0) { P.out_len[oi] = 0; P.out_flags[oi] = (overflow == 1 || overflow == 3) ? kFlagOverflow : kFlagError; }
            continue;
        }
        phase_consensus<OneWaveBlock>(out, out_cap, &P.out_len[oi], &P.out_flags[oi], ns, P.win_type[w] == 1);
        RCN_PHASE(6);
    }
    if (lane == 0) {
        atomicAdd(&P.stats[0], ctx->cells); atomicAdd(&P.stats[1], ctx->pred); atomicAdd(&P.stats[2], ctx->bytes);
        for (int k = 0; k < 8; ++k) atomicAdd(&P.stats[3 + k], ph[k]);
        atomicAdd(&P.stats[11], ctx->ties);
        atomicAdd(&P.stats[12], ctx->cells_full); atomicAdd(&P.stats[13], ctx->bytes_full);
    }
}
#endif  // RCN_DEEP_TU

}  // namespace rcn
