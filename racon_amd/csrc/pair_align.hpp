// pair_align.hpp — exact global pairwise alignment WITH PATH on the device (included from engine.hip).
//
// What it replaces: Overlap::find_breaking_points' call of
//   edlibAlign(q, ql, t, tl, edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_PATH, NULL, 0)) + edlibAlignmentToCigar
// (reference src/overlap.cpp:205-224; shape model of a batched device aligner: src/cuda/cudaaligner.cpp:51-102), i.e. what
// racon_amd/host/nw_path.cpp computes on the host.  edlib 1.2.7 is not vendored in the reference; WHICH co-optimal path it
// returns is specified in SURVEY.md Appendix B (validated there against the PAF / MHAP goldens) and restated here:
//   obtain(q, t, best):  |q| == 0 -> |t| x D;  |t| == 0 -> |q| x I;
//     blocks = ceil(|q| / 64); if 20 * blocks * |t| + 8 * |t| < 2^20: plain traceback from the bottom-right cell, at each
//       cell preferring up (I), then left (D), else diagonal (M);
//     else Hirschberg on the target axis: lw = |t| / 2, left[h] = ED(q[:h], t[:lw]), right[k] = ED(q[|q|-k:], t[lw:]),
//       the smallest h in 1..|q|-1 with left[h] + right[|q|-h] == best, else h = 0, else h = |q|; recurse on both sides.
//
// How: Myers' bit-vector recurrence, one 64-row word per lane, the 64 lanes of a wave on an anti-diagonal of
// (word, column) cells: lane k works on column s - k at step s, its horizontal carry-out is the carry-in of lane k + 1
// at the next step (one DPP shift), so a step advances 64 words x 64 rows = 4096 cells with ~45 instructions and no
// serial carry chain.  Queries longer than 4096 rows take several passes (the carries of a pass's last word go through
// a byte per column in HBM).  The query is held as bit planes of a dense symbol code (3 planes: up to 7 distinct query
// symbols; 8 planes of the raw byte otherwise), Eq = "all planes agree with the column's symbol": no per-symbol table,
// any byte alphabet.  A TEAM OF TWO WAVES (one work-group) owns an overlap from start to end: a launch that holds all its
// overlaps at once ends with its largest one, and that overlap is a serial chain of passes -- but the two passes of a
// split (left half forwards, right half backwards) do not depend on each other, nor do two leaves: wave 0 takes the one,
// wave 1 the other, and the chain is half as long (profiles/r04: 3000 cfg2 overlaps 17.0 ms with one wave each).  The
// Hirschberg recursion is an explicit stack in LDS that both waves read (uniform control flow, work-group barriers),
// the two last-column vectors of a split live in the team's HBM scratch, a leaf stores the vertical / horizontal delta
// words (Pv, Ph) of its cells (at most ~0.8 MiB + skew padding, the same bound edlib's own traceback state has) and the
// walk tests one bit per move.  The path is written as one op byte per (row + column) position of the move's start
// cell -- positions are unique along a monotone path, sub-problems own disjoint ranges, so pieces land in place in any
// order -- and either run-length encoded on the host (tests, rcn_engine_alignment_cigars) or walked on the device into
// breaking points (k_ops_breaking_points) without leaving HBM.
#pragma once
#include "pair_cell.hpp"

namespace rcn {

#ifdef RCN_PROF_PAIR
// profiling build (make pairprof): wave clocks by phase, summed over all waves since the library was loaded
//  0 steady blocks  1 ramp / tail / snapshot blocks  2 row load  3 emit (scores of a pass)  4 leaf walk  5 cut  6 symbol set-up
//  7 whole overlap (per wave)  8 barriers  9 steady steps  10 general steps  11 passes  12 leaves  13 cuts  14 sum of words per steady step (lane use)
__device__ unsigned long long g_pairprof[16];
#define RCN_PP_T(var) const long long var = clock64()
#define RCN_PP_ADD(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_pairprof[k], static_cast<unsigned long long>(v)); } while (0)
#else
#define RCN_PP_T(var) do {} while (0)
#define RCN_PP_ADD(k, v) do {} while (0)
#endif

struct PairParams {
    const uint8_t* bases;            // resident read set (forward strand)
    const uint64_t* q_pos;           // [n] byte offset of the first base of the query segment (forward storage)
    const uint64_t* t_pos;           // [n] ... of the target segment
    const uint32_t* q_len;           // [n] rows
    const uint32_t* t_len;           // [n] columns
    const uint8_t* q_rc;             // [n] 1: the rows are the reverse complement of the stored segment
    const uint32_t* order;           // [n] work order (largest problems first)
    uint32_t n_pairs;
    unsigned int* next;              // work queue counter
    uint8_t* ops;                    // op bytes ('M' 'I' 'D', 0 = no move starts here), zeroed
    const uint64_t* ops_off;         // [n + 1]
    int32_t* dist;                   // [n] edit distance
    uint8_t* scratch; uint64_t slot_bytes;
    uint32_t m_cap, n_cap;           // largest rows / columns of the batch (scratch layout)
    uint64_t leaf_bytes;             // leaf store per wave
    uint32_t arena_ints;             // inherited column vectors of a team (pair_arena_ints)
    uint32_t* err;                   // [0] != 0: internal error (no optimal split found / stack overflow)
};

constexpr int kPairRing = 198;       // entries of a wave's symbol ring: 128 columns, the mirror of the first 64, the read-ahead of a block's last trip
constexpr int kPairStack = 64;       // Hirschberg tasks pending per overlap (depth ~ log2 columns, two pushes per level)

struct PairView {                    // a sequence read forwards, or backwards and complemented
    const uint8_t* p; bool rc;
    int64_t n;
};
__device__ __forceinline__ uint32_t pair_comp(uint32_t c) {    // Sequence::create_reverse_complement (reference src/sequence.cpp:49-84)
    uint32_t r = c;                                              // (selects, not a chain of branches)
    r = c == 'A' ? 'T' : r; r = c == 'T' ? 'A' : r; r = c == 'C' ? 'G' : r; r = c == 'G' ? 'C' : r;
    return r;
}
__device__ __forceinline__ uint32_t pv_at(const PairView& v, int64_t i) {
    return v.rc ? pair_comp(v.p[v.n - 1 - i]) : static_cast<uint32_t>(v.p[i]);
}

// leaf rule of the reference aligner (see the header)
__host__ __device__ __forceinline__ bool pair_is_leaf(int64_t m, int64_t n) {
    const int64_t blocks = (m + 63) / 64;
    return (2 * 8 + 4) * blocks * n + 2 * 4 * n < 1024 * 1024;
}
// bytes of the (Pv, Ph) store of the largest leaf a query of at most m_cap rows can produce: 16 B per (word, column)
// cell plus the skew padding of every 64-word pass
__host__ __device__ __forceinline__ uint64_t pair_leaf_bytes(uint64_t m_cap) {
    const uint64_t nb = (m_cap + 63) / 64;
    return 16ull * (52429ull + 64ull * (nb + 64)) + 4096;
}

// What one lane of the wave stored, another lane of the SAME wave loads next: the stores must have been performed, nothing
// else -- a work-group-scope fence (its waves share the CU's vector L1, which takes write hits: s_waitcnt only).  The
// agent-scope __threadfence() that stood here wrote the L2 back and invalidated the caches (buffer_wbl2 sc1 + buffer_inv sc1)
// after every 64-word pass of every sub-problem.
__device__ __forceinline__ void pair_wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__device__ __forceinline__ int pair_shr1(int fill, int v) {      // lane l <- lane l - 1, lane 0 <- fill
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t pair_shr1_zero(uint32_t v) {  // ... lane 0 <- 0 (no move for the fill: the DPP's own zero fill)
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x138, 0xf, 0xf, true));
}
// carry byte of a column as the passes hand it on through HBM (bit 0 = +1, bit 1 = -1) <-> the cell's form (pair_cell.hpp)
__device__ __forceinline__ PairCarry pair_carry_of(int h) { return PairCarry{static_cast<uint32_t>((~h) & 1) << 31, static_cast<uint32_t>(h & 2) << 30}; }
__device__ __forceinline__ int pair_carry_byte(PairCarry c) { return static_cast<int>((~c.np) >> 31) | (static_cast<int>(c.mn >> 31) << 1); }

// One pass of the recurrence: words [w0, w0 + nwp) of the rows, all n columns.
//   rows: logical row r of this sub-problem = Q[q0 + r], or Q[q0 + m - 1 - r] when flipped (the backward half of a split)
//   columns likewise.  hin_buf / hout_buf: carries of the word above / for the word below (one byte per column:
//   bit 0 = +1, bit 1 = -1); hin_buf == nullptr: the top boundary (+1 per column).  store != nullptr: leaf, (Pv, ~Ph)
//   of the cell of lane l at step s goes to store[s * nwp + l].
// HIN / HB: the pass has a word above / below it (a query of more than 4096 rows); the common single pass pays for neither the
// per-step injection at lane 0 nor the collection of the last word's carries.
template <int NPL, bool STORE, bool HIN, bool HB>
__device__ __forceinline__ void pair_pass_impl(const PairView& Q, int64_t q0, int m, bool qflip, const PairView& T, int64_t t0, int n, bool tflip,
                                               int w0, int nwp, const uint8_t* codes, const uint8_t* hin_buf, uint8_t* hout_buf,
                                               ulonglong2* store, unsigned long long& Pv_out, unsigned long long& Mv_out,
                                               int nsnap_, int sc0_, int sc1_, int sc2_, ulonglong2* snapbuf, uint32_t* ring) {
    const int lane = threadIdx.x & 63;
    // (all of these are the same in every lane; said so, the loop counters and range tests stay in scalar registers)
    m = __builtin_amdgcn_readfirstlane(m); n = __builtin_amdgcn_readfirstlane(n); w0 = __builtin_amdgcn_readfirstlane(w0); nwp = __builtin_amdgcn_readfirstlane(nwp);
    const int nsnap = __builtin_amdgcn_readfirstlane(nsnap_), sc0 = __builtin_amdgcn_readfirstlane(sc0_), sc1 = __builtin_amdgcn_readfirstlane(sc1_), sc2 = __builtin_amdgcn_readfirstlane(sc2_);
    // The column symbols on their way down the lanes.  Lane l works on column s - l at step s; the code used to travel with it (one
    // v_readlane + v_mov + DPP shift per step, then one bit-field extract per plane for the cell's masks).  Up to four planes the
    // masks now stand ready-made in a ring in LDS -- this wave's, 128 columns + a mirror of the first 64 so that a block's 64 steps
    // read ring[((s0 - l) & 127) + k] without wrapping -- written once per block of 64 columns, read by every lane once per step:
    // no vector instruction per step at all (the LDS port is otherwise idle: 8 or 16 bytes per lane and step are ~15-25 % of it).
    constexpr bool RING = NPL <= 4;
    constexpr int EW = NPL <= 2 ? 2 : 4;             // dwords per ring entry
    RCN_PP_T(pp0__);
#ifdef RCN_PROF_PAIR
    long long pp_steady__ = 0, pp_nsteady__ = 0;
#endif
    PairLane<NPL> L;
#pragma unroll
    for (int k = 0; k < NPL; ++k) { L.pl[k] = 0u; L.ph[k] = 0u; }
    L.vl = 0u; L.vh = 0u; L.Pvl = ~0u; L.Pvh = ~0u; L.Mvl = 0u; L.Mvh = 0u;
    if (lane < nwp) {
        // this lane's 64 rows are 64 consecutive bytes of the stored read, up or down; sixteen loads in flight at a time
        // (one load, one wait and a branchy complement per row was 64 memory round trips before the first step of a pass),
        // the code -- complement included -- is one lookup in the pair's table
        const int64_t r0 = static_cast<int64_t>(w0 + lane) * 64;
        const int nrow = static_cast<int>(min(static_cast<int64_t>(64), static_cast<int64_t>(m) - r0));
        const int64_t l0 = qflip ? q0 + m - 1 - r0 : q0 + r0;                  // logical position of row r0; row r0 + r: l0 -/+ r
        const int64_t i0 = Q.rc ? Q.n - 1 - l0 : l0;
        const int64_t step = (qflip != Q.rc) ? -1 : 1;
        unsigned long long plane[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) plane[k] = 0ull;
#pragma unroll 1
        for (int rb = 0; rb < 64; rb += 16) {
            uint32_t raw[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) raw[u] = Q.p[i0 + step * min(rb + u, nrow - 1)];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t code = codes[raw[u]];
#pragma unroll
                for (int k = 0; k < NPL; ++k) plane[k] |= static_cast<unsigned long long>((code >> k) & 1u) << (rb + u);
            }
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) { L.pl[k] = static_cast<uint32_t>(plane[k]); L.ph[k] = static_cast<uint32_t>(plane[k] >> 32); }
        const unsigned long long valid = nrow >= 64 ? ~0ull : ((1ull << nrow) - 1ull);     // (rows past the end repeat the last one: masked here)
        L.vl = static_cast<uint32_t>(valid); L.vh = static_cast<uint32_t>(valid >> 32);
    }
    RCN_PP_T(pp1__);
    // the target is stored forwards (PairView::rc is the query's): column `col` of this pass
    auto tcol = [&](int col) -> uint32_t { return T.p[tflip ? t0 + n - 1 - col : t0 + col]; };
    uint32_t traw = lane < n ? tcol(lane) : 0u;                                  // raw symbol / carry-in of the block's columns, fetched a block ahead
    int hraw = (HIN && lane < n) ? hin_buf[lane] : 1;
    int tbuf = 0, hbuf = 1;          // columns s0 .. s0 + 63: symbol code / carry-in byte of the top word, one per lane
    int tc = 0;                      // this lane's column symbol
    PairCarry hc{0x80000000u, 0u};   // what the cell of this lane handed down at the previous step
    const int steps = n + nwp - 1;
    for (int s0 = 0; s0 < steps; s0 += 64) {
        {   // the 64 columns that enter at lane 0 during this block; the next block's are requested now and looked at then
            const int col = s0 + lane;
            traw = col < n ? tcol(col) : 0u;
            hraw = (HIN && col < n) ? hin_buf[col] : 1;
            tbuf = col < n ? static_cast<int>(codes[256 + traw]) : 7;
            hbuf = hraw;
            if (RING) {
                const PairSym<NPL> y = pair_sym_of<NPL>(tbuf);
                uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int k = 0; k < NPL && k < 4; ++k) w[k] = y.mk[k];
                const int slot = (s0 >> 6) & 1;
                uint32_t* e = ring + (slot * 64 + lane) * EW;
                if (EW == 2) { *reinterpret_cast<uint2*>(e) = make_uint2(w[0], w[1]); if (slot == 0) *reinterpret_cast<uint2*>(e + 128 * EW) = make_uint2(w[0], w[1]); }
                else { *reinterpret_cast<uint4*>(e) = make_uint4(w[0], w[1], w[2], w[3]); if (slot == 0) *reinterpret_cast<uint4*>(e + 128 * EW) = make_uint4(w[0], w[1], w[2], w[3]); }
            }
        }
        // a column whose vector a later sub-problem inherits (pair_align_one): its words pass it during [c - 1, c - 1 + nwp - 1]
        bool snap_here = false;
        if (nsnap > 0) snap_here |= sc0 - 1 <= s0 + 63 && sc0 + nwp - 2 >= s0;
        if (nsnap > 1) snap_here |= sc1 - 1 <= s0 + 63 && sc1 + nwp - 2 >= s0;
        if (nsnap > 2) snap_here |= sc2 - 1 <= s0 + 63 && sc2 + nwp - 2 >= s0;
        const uint32_t* rp = ring + ((s0 - lane) & 127) * EW;              // this lane's column at step s0; + one entry per step
        auto ring_at = [&](int k) -> PairSym<NPL> {
            PairSym<NPL> y;
            uint32_t w[4];
            if (EW == 2) { const uint2 v = *reinterpret_cast<const uint2*>(rp + k * EW); w[0] = v.x; w[1] = v.y; w[2] = 0u; w[3] = 0u; }
            else { const uint4 v = *reinterpret_cast<const uint4*>(rp + k * EW); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
#pragma unroll
            for (int q = 0; q < NPL; ++q) y.mk[q] = w[q & 3];
            return y;
        };
        if (s0 >= nwp - 1 && s0 + 63 < n && !snap_here) {
            // steady state: at every step of the block every word of the pass has a column in [0, n) -- no range tests, no
            // exec-mask regions (lanes past the pass's words compute on valid = 0; nobody reads them).  The carries out of the
            // last word: every lane shifts its own two bits per step into two registers (one v_alignbit each), and at the end
            // of the block the last word's 2 x 64 bits are handed out, one column per lane, and stored once.
            RCN_PP_T(ppb__);
            uint32_t accP[2] = {0u, 0u}, accN[2] = {0u, 0u};
            // two steps' masks per read (one ds_read2_b64 with two planes), two such pairs in flight: a pair is read again as soon as
            // its two steps are done and used two steps later -- no copies, the LDS latency under ~60 vector instructions
            PairSym<NPL> ya[2], yb[2];
#pragma unroll
            for (int q = 0; q < NPL; ++q) { ya[0].mk[q] = 0u; ya[1].mk[q] = 0u; yb[0].mk[q] = 0u; yb[1].mk[q] = 0u; }
            if (RING) { ya[0] = ring_at(0); ya[1] = ring_at(1); yb[0] = ring_at(2); yb[1] = ring_at(3); }
            auto step = [&](int k, const PairSym<NPL>& yk, uint32_t& aP, uint32_t& aN) {
                PairSym<NPL> y = yk;
                if (!RING) {
                    const int t_new = __builtin_amdgcn_readlane(tbuf, k);
                    tc = pair_shr1(t_new, tc);
                    y = pair_sym_of<NPL>(tc);
                }
                PairCarry cin;
                if (HIN) {
                    const PairCarry top = pair_carry_of(__builtin_amdgcn_readlane(hbuf, k));
                    cin.np = static_cast<uint32_t>(pair_shr1(static_cast<int>(top.np), static_cast<int>(hc.np)));
                    cin.mn = static_cast<uint32_t>(pair_shr1(static_cast<int>(top.mn), static_cast<int>(hc.mn)));
                } else {
                    cin.np = pair_shr1_zero(hc.np); cin.mn = pair_shr1_zero(hc.mn);
                }
                uint32_t nl, nh;
                hc = pair_cell<NPL>(L, y, cin, nl, nh);
                if (STORE) {
                    if (lane < nwp) {
                        ulonglong2 v;
                        v.x = (static_cast<unsigned long long>(L.Pvh) << 32) | L.Pvl; v.y = (static_cast<unsigned long long>(nh) << 32) | nl;
                        store[static_cast<int64_t>(s0 + k) * nwp + lane] = v;
                    }
                }
                if (HB) { aP = pc_alignbit(aP, hc.np, 31); aN = pc_alignbit(aN, hc.mn, 31); }
            };
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint32_t aP = 0u, aN = 0u;
#pragma unroll 1
                for (int kk = 0; kk < 32; kk += 4) {
                    const int k = 32 * half + kk;
                    step(k, ya[0], aP, aN); step(k + 1, ya[1], aP, aN);
                    if (RING) { ya[0] = ring_at(k + 4); ya[1] = ring_at(k + 5); }      // (the block's last trip reads four entries past it: the ring has them)
                    step(k + 2, yb[0], aP, aN); step(k + 3, yb[1], aP, aN);
                    if (RING) { yb[0] = ring_at(k + 6); yb[1] = ring_at(k + 7); }
                }
                accP[half] = aP; accN[half] = aN;
            }
            if (HB) {
                // the last word was at column s0 + k - (nwp - 1) at step k; step k = 32 half + kk sits at bit 31 - kk of its half
                const int src = nwp - 1;
                const uint32_t p0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(accP[0]), src));
                const uint32_t p1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(accP[1]), src));
                const uint32_t n0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(accN[0]), src));
                const uint32_t n1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(accN[1]), src));
                const uint32_t wp = lane < 32 ? p0 : p1, wn = lane < 32 ? n0 : n1;
                const int sh = 31 - (lane & 31);
                hout_buf[s0 + lane - (nwp - 1)] = static_cast<uint8_t>(((~wp >> sh) & 1u) | (((wn >> sh) & 1u) << 1));
            }
#ifdef RCN_PROF_PAIR
            pp_steady__ += clock64() - ppb__; pp_nsteady__ += 64;
#endif
            continue;
        }
        // An edge block: the ramp at the start of a pass (words still waiting for their first column), its tail (words that are
        // through), a block that holds a column whose vector a later sub-problem inherits.  Same straight-line step as the steady
        // state -- every lane computes -- and a lane whose column is outside [0, n) keeps its state by four selects; no exec-mask
        // region around the cell, no taken branch per step (as branches the edge blocks cost 3.2 x a steady step each, and with
        // three inherited columns per pass they were a quarter of all steps: profiles/r06/g_pair_phase_clocks.txt).  The carry
        // variants are decided at run time here (injection from hbuf, which holds the top boundary when no word is above).
        const int kend = min(64, steps - s0);
        const bool hb = hout_buf != nullptr;
        uint32_t accP[2] = {0u, 0u}, accN[2] = {0u, 0u};
        PairSym<NPL> yn;
#pragma unroll
        for (int q = 0; q < NPL; ++q) yn.mk[q] = 0u;
        if (RING) yn = ring_at(0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t aP = 0u, aN = 0u;
            const int kk_end = min(32, kend - 32 * half);
#pragma unroll 2
            for (int kk = 0; kk < kk_end; ++kk) {
                const int k = 32 * half + kk;
                PairSym<NPL> y = yn;
                if (RING) yn = ring_at(k + 1);
                else {
                    const int t_new = __builtin_amdgcn_readlane(tbuf, k);      // lane l takes over the column lane l - 1 had; lane 0 starts column s
                    tc = pair_shr1(t_new, tc);
                    y = pair_sym_of<NPL>(tc);
                }
                const PairCarry top = pair_carry_of(__builtin_amdgcn_readlane(hbuf, k));
                PairCarry cin;
                cin.np = static_cast<uint32_t>(pair_shr1(static_cast<int>(top.np), static_cast<int>(hc.np)));
                cin.mn = static_cast<uint32_t>(pair_shr1(static_cast<int>(top.mn), static_cast<int>(hc.mn)));
                const int j = s0 + k - lane;
                const bool active = lane < nwp && static_cast<unsigned>(j) < static_cast<unsigned>(n);
                PairLane<NPL> Lt = L;
                uint32_t nl, nh;
                hc = pair_cell<NPL>(Lt, y, cin, nl, nh);           // (a lane outside its columns hands down garbage: nobody inside takes it)
                L.Pvl = active ? Lt.Pvl : L.Pvl; L.Pvh = active ? Lt.Pvh : L.Pvh; L.Mvl = active ? Lt.Mvl : L.Mvl; L.Mvh = active ? Lt.Mvh : L.Mvh;
                if (hb) { aP = pc_alignbit(aP, hc.np, 31); aN = pc_alignbit(aN, hc.mn, 31); }
                if (STORE || snap_here) {
                    if (active) {
                        const unsigned long long pv64 = (static_cast<unsigned long long>(L.Pvh) << 32) | L.Pvl;
                        if (STORE) { ulonglong2 v; v.x = pv64; v.y = (static_cast<unsigned long long>(nh) << 32) | nl; store[static_cast<int64_t>(s0 + k) * nwp + lane] = v; }
                        if (snap_here) {
                            ulonglong2 v; v.x = pv64; v.y = (static_cast<unsigned long long>(L.Mvh) << 32) | L.Mvl;
                            if (nsnap > 0 && j == sc0 - 1) snapbuf[lane] = v;
                            if (nsnap > 1 && j == sc1 - 1) snapbuf[64 + lane] = v;
                            if (nsnap > 2 && j == sc2 - 1) snapbuf[128 + lane] = v;
                        }
                    }
                }
            }
            if (kk_end > 0 && kk_end < 32) { aP <<= (32 - kk_end); aN <<= (32 - kk_end); }      // step kk of a half sits at bit 31 - kk
            accP[half] = aP; accN[half] = aN;
        }
        if (hb) {
            const int src = nwp - 1;
            const uint32_t p0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(accP[0]), src));
            const uint32_t p1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(accP[1]), src));
            const uint32_t n0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(accN[0]), src));
            const uint32_t n1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(accN[1]), src));
            const uint32_t wp = lane < 32 ? p0 : p1, wn = lane < 32 ? n0 : n1;
            const int sh = 31 - (lane & 31);
            const int col = s0 + lane - (nwp - 1);                 // the last word's column at step `lane` of the block
            if (lane < kend && col >= 0 && col < n) hout_buf[col] = static_cast<uint8_t>(((~wp >> sh) & 1u) | (((wn >> sh) & 1u) << 1));
        }
    }
    Pv_out = (static_cast<unsigned long long>(L.Pvh) << 32) | L.Pvl; Mv_out = (static_cast<unsigned long long>(L.Mvh) << 32) | L.Mvl;
#ifdef RCN_PROF_PAIR
    { const long long pp2__ = clock64();
      RCN_PP_ADD(0, pp_steady__); RCN_PP_ADD(1, pp2__ - pp1__ - pp_steady__); RCN_PP_ADD(2, pp1__ - pp0__); RCN_PP_ADD(9, pp_nsteady__); RCN_PP_ADD(10, steps - pp_nsteady__);
      RCN_PP_ADD(11, 1); RCN_PP_ADD(14, pp_nsteady__ * nwp); }
#endif
}

template <int NPL, bool STORE>
__device__ __forceinline__ void pair_pass(const PairView& Q, int64_t q0, int m, bool qflip, const PairView& T, int64_t t0, int n, bool tflip,
                                          int w0, int nwp, const uint8_t* codes, const uint8_t* hin_buf, uint8_t* hout_buf,
                                          ulonglong2* store, unsigned long long& Pv_out, unsigned long long& Mv_out, uint32_t* ring,
                                          int nsnap = 0, int sc0 = 0, int sc1 = 0, int sc2 = 0, ulonglong2* snapbuf = nullptr) {
    if (hin_buf == nullptr && hout_buf == nullptr)
        pair_pass_impl<NPL, STORE, false, false>(Q, q0, m, qflip, T, t0, n, tflip, w0, nwp, codes, hin_buf, hout_buf, store, Pv_out, Mv_out, nsnap, sc0, sc1, sc2, snapbuf, ring);
    else if (hin_buf == nullptr)
        pair_pass_impl<NPL, STORE, false, true>(Q, q0, m, qflip, T, t0, n, tflip, w0, nwp, codes, hin_buf, hout_buf, store, Pv_out, Mv_out, nsnap, sc0, sc1, sc2, snapbuf, ring);
    else if (hout_buf == nullptr)
        pair_pass_impl<NPL, STORE, true, false>(Q, q0, m, qflip, T, t0, n, tflip, w0, nwp, codes, hin_buf, hout_buf, store, Pv_out, Mv_out, nsnap, sc0, sc1, sc2, snapbuf, ring);
    else
        pair_pass_impl<NPL, STORE, true, true>(Q, q0, m, qflip, T, t0, n, tflip, w0, nwp, codes, hin_buf, hout_buf, store, Pv_out, Mv_out, nsnap, sc0, sc1, sc2, snapbuf, ring);
}

// Last column of the sub-problem's matrix: out[i] = ED(rows[:i], all columns), i = 0 .. m.  Returns out[m].
// nsnap > 0: the vectors of the columns sc0 > sc1 > sc2 (counted in the pass's direction) as well, into snap_out + k * snap_stride.
constexpr int kPairSnap = 3;
template <int NPL>
__device__ __forceinline__ int pair_columns(const PairView& Q, int64_t q0, int m, bool qflip, const PairView& T, int64_t t0, int n, bool tflip,
                                            const uint8_t* codes, uint8_t* hbuf0, uint8_t* hbuf1, int32_t* out, uint32_t* ring,
                                            int nsnap = 0, int sc0 = 0, int sc1 = 0, int sc2 = 0, int32_t* snap_out = nullptr, int snap_stride = 0,
                                            ulonglong2* snapbuf = nullptr) {
    const int lane = threadIdx.x & 63;
    const int nb = (m + 63) / 64;
    int carry = n;                                   // score at the last row of the words done so far
    int scarry[kPairSnap] = {sc0, sc1, sc2};
    if (lane == 0) {
        out[0] = n;
        if (nsnap > 0) snap_out[0] = sc0;
        if (nsnap > 1) snap_out[snap_stride] = sc1;
        if (nsnap > 2) snap_out[2 * snap_stride] = sc2;
    }
    for (int w0 = 0, pass = 0; w0 < nb; w0 += 64, ++pass) {
        const int nwp = min(64, nb - w0);
        const uint8_t* hin = w0 == 0 ? nullptr : ((pass & 1) ? hbuf0 : hbuf1);
        uint8_t* hout = w0 + 64 < nb ? ((pass & 1) ? hbuf1 : hbuf0) : nullptr;
        unsigned long long Pv, Mv;
        pair_pass<NPL, false>(Q, q0, m, qflip, T, t0, n, tflip, w0, nwp, codes, hin, hout, nullptr, Pv, Mv, ring, nsnap, sc0, sc1, sc2, snapbuf);
        pair_wave_fence();                             // the carries of this pass are read (by other lanes) in the next one
        RCN_PP_T(ppe0__);
        // scores of this pass's rows: running sum of the vertical deltas down the column
        const int rows_here = lane < nwp ? min(64, m - (w0 + lane) * 64) : 0;
        const unsigned long long vmask = rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull);
        const int64_t r0 = static_cast<int64_t>(w0 + lane) * 64;
        auto emit = [&](unsigned long long pv, unsigned long long mv, int& cy, int32_t* dst) {
            int delta = __popcll(pv & vmask) - __popcll(mv & vmask);
            int incl = delta;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
            int acc = cy + incl - delta;
            for (int r = 0; r < rows_here; ++r) {
                acc += static_cast<int>((pv >> r) & 1ull) - static_cast<int>((mv >> r) & 1ull);
                dst[r0 + r + 1] = acc;
            }
            cy += __shfl(incl, 63);
        };
        emit(Pv, Mv, carry, out);
#pragma unroll
        for (int k = 0; k < kPairSnap; ++k) {
            if (k < nsnap) {
                ulonglong2 v; v.x = ~0ull; v.y = 0ull;
                if (lane < nwp) v = snapbuf[64 * k + lane];
                emit(v.x, v.y, scarry[k], snap_out + static_cast<int64_t>(k) * snap_stride);
            }
        }
#ifdef RCN_PROF_PAIR
        RCN_PP_ADD(3, clock64() - ppe0__);
#endif
    }
    pair_wave_fence();                                 // out[] is read by other lanes than the ones that wrote it
    return carry;
}

// plain traceback of a leaf (up, then left, else diagonal) over the stored delta words (Pv, ~Ph) of its forward passes
__device__ __forceinline__ void pair_leaf_walk(const ulonglong2* store, int nb, int m, int n, int64_t base, uint8_t* ops) {
    const int lane = threadIdx.x & 63;
    // every full pass has 64 words: pass p starts at p * (n + 63) * 64
    const int64_t pass_stride = static_cast<int64_t>(n + 63) * 64;
    RCN_PP_T(ppw0__);
    // The walk.  One move at a time it was ~70 instructions per move with one useful lane -- nearly half of all the kernel's
    // instructions on 10 kb reads -- and nine moves in ten are diagonal: lane l of the cache already holds the cell l columns to
    // the left, so every lane tests ITS cell of the current diagonal (bit b - t of its words), one ballot gives the length of the
    // run of diagonal moves (neither "up" nor "left": the same rule, cell by cell), the run's op bytes are stored by its lanes
    // at once, and only the cells that end a run are taken singly.  i, j and everything derived from them are wave-uniform.
    int i = m, j = n;                                 // current cell (rows 1..m, columns 1..n; 0 = boundary)
    int cw = -1, cj0 = -1;                            // cache: lane c holds cell (word cw, column cj0 - c)
    uint32_t pv_lo = 0, pv_hi = 0, ph_lo = 0, ph_hi = 0;        // (ph_*: the complement of the horizontal plus-deltas, as stored)
    while (i > 0 && j > 0) {
        const int w = (i - 1) >> 6, b = (i - 1) & 63;
        if (w != cw || j > cj0 || j <= cj0 - 64) {
            cw = w; cj0 = j;
            const int jc = j - lane;
            unsigned long long a = 0ull, h = 0ull;
            if (jc >= 1) {
                const int p = w >> 6, l = w & 63, nwp = min(64, nb - (p << 6));
                const ulonglong2 v = store[p * pass_stride + static_cast<int64_t>(jc - 1 + l) * nwp + l];
                a = v.x; h = v.y;
            }
            pv_lo = static_cast<uint32_t>(a); pv_hi = static_cast<uint32_t>(a >> 32);
            ph_lo = static_cast<uint32_t>(h); ph_hi = static_cast<uint32_t>(h >> 32);
        }
        const int c = cj0 - j;                        // lane of the current cell
        const int t = lane - c, bit = b - t;          // this lane's cell of the diagonal: (i - t, j - t), bit `bit` of word w
        const bool ok = t >= 0 && bit >= 0 && j - t >= 1;
        const uint32_t pw = (bit & 32) ? pv_hi : pv_lo, hw = (bit & 32) ? ph_hi : ph_lo;
        const bool up = ok && ((pw >> (bit & 31)) & 1u) != 0u, left = ok && ((hw >> (bit & 31)) & 1u) == 0u;     // (the store holds ~Ph: pair_cell.hpp)
        const unsigned long long diag = __ballot(ok && !up && !left) >> c;
        const int run = diag == ~0ull ? 64 : __builtin_ctzll(~diag);
        if (run > 0) {
            if (t >= 0 && t < run) ops[base + (i - 1 - t) + (j - 1 - t)] = 'M';
            i -= run; j -= run;
            continue;
        }
        const bool up_here = ((__ballot(up) >> c) & 1ull) != 0ull;
        uint8_t op; int64_t slot;
        if (up_here) { op = 'I'; slot = base + (i - 1) + j; --i; }
        else { op = 'D'; slot = base + i + (j - 1); --j; }
        if (lane == 0) ops[slot] = op;
    }
    for (int k = lane; k < i; k += 64) ops[base + k] = 'I';          // column 0: D[k][0] = k, only "up" is possible
    for (int k = lane; k < j; k += 64) ops[base + k] = 'D';          // row 0
#ifdef RCN_PROF_PAIR
    RCN_PP_ADD(4, clock64() - ppw0__); RCN_PP_ADD(12, 1);
#endif
}

// a leaf: forward passes with the (Pv, ~Ph) store, then the walk
template <int NPL>
__device__ __forceinline__ void pair_leaf(const PairView& Q, int64_t q0, int m, const PairView& T, int64_t t0, int n, const uint8_t* codes,
                                          uint8_t* hbuf0, uint8_t* hbuf1, ulonglong2* store, uint8_t* ops, uint32_t* ring) {
    const int nb = (m + 63) / 64;
    // forward passes with the (Pv, Ph) store; pass p starts at store + pass_off(p)
    {
        int64_t off = 0;
        for (int w0 = 0, pass = 0; w0 < nb; w0 += 64, ++pass) {
            const int nwp = min(64, nb - w0);
            const uint8_t* hin = w0 == 0 ? nullptr : ((pass & 1) ? hbuf0 : hbuf1);
            uint8_t* hout = w0 + 64 < nb ? ((pass & 1) ? hbuf1 : hbuf0) : nullptr;
            unsigned long long Pv, Mv;
            pair_pass<NPL, true>(Q, q0, m, false, T, t0, n, false, w0, nwp, codes, hin, hout, store + off, Pv, Mv, ring);
            pair_wave_fence();
            off += static_cast<int64_t>(n + nwp - 1) * nwp;
        }
    }
    pair_leaf_walk(store, nb, m, n, static_cast<int64_t>(q0) + t0, ops);
}

// TWO leaves in one wave.  Below the root the sub-problems have fewer words than a wave has lanes (a leaf of a 6 kb overlap: ~25; over a
// shard of cfg5 39 of 64 lanes held a word in the steady state, profiles/r06/g_pair_phase_clocks.txt), and the kernel is bound by vector
// instructions issued, whatever the lanes hold.  Two leaves of one overlap whose words fit a wave together (nbA + nbB <= 64) therefore
// run their forward pass side by side: leaf A in lanes [0, nbA), leaf B in lanes [nbA, nbA + nbB), each with its own rows, columns, symbol
// ring and store; the lane that starts B takes the top boundary instead of its neighbour's carry (two selects per step); a lane whose
// leaf has run out of columns keeps its state as in any edge block.  Then the two walks, one after the other.  (Two symbol planes only:
// the second ring takes the LDS the wider entries of three planes would.)
struct PairLeafJob { int q0, m, t0, n; };
__device__ __forceinline__ bool pair_leaf_pairs(int nbA, int nbB) { return nbA + nbB <= 64; }

template <int NPL>
__device__ __forceinline__ void pair_leaf_two(const PairView& Q, const PairView& T, const uint8_t* codes, PairLeafJob A, PairLeafJob B,
                                              ulonglong2* store, uint8_t* ops, uint32_t* ring) {
    static_assert(NPL == 2, "two rings of 8-byte entries share a wave's LDS");
    constexpr int EW = 2;
    const int lane = threadIdx.x & 63;
    A.q0 = __builtin_amdgcn_readfirstlane(A.q0); A.m = __builtin_amdgcn_readfirstlane(A.m); A.t0 = __builtin_amdgcn_readfirstlane(A.t0); A.n = __builtin_amdgcn_readfirstlane(A.n);
    B.q0 = __builtin_amdgcn_readfirstlane(B.q0); B.m = __builtin_amdgcn_readfirstlane(B.m); B.t0 = __builtin_amdgcn_readfirstlane(B.t0); B.n = __builtin_amdgcn_readfirstlane(B.n);
    const int nbA = (A.m + 63) / 64, nbB = (B.m + 63) / 64;
    ulonglong2* storeA = store;
    ulonglong2* storeB = store + static_cast<int64_t>(A.n + nbA - 1) * nbA;
    {
        RCN_PP_T(pp0__);
        const bool sb = lane >= nbA;                               // this lane's leaf
        const int ll = sb ? lane - nbA : lane;                     // its word
        const int nw_l = sb ? nbB : nbA, m_l = sb ? B.m : A.m, n_l = sb ? B.n : A.n;
        const int64_t q0_l = sb ? B.q0 : A.q0;
        ulonglong2* sp = (sb ? storeB : storeA) + ll;              // + step * nw_l
        PairLane<NPL> L;
#pragma unroll
        for (int k = 0; k < NPL; ++k) { L.pl[k] = 0u; L.ph[k] = 0u; }
        L.vl = 0u; L.vh = 0u; L.Pvl = ~0u; L.Pvh = ~0u; L.Mvl = 0u; L.Mvh = 0u;
        if (ll < nw_l) {
            const int64_t r0 = static_cast<int64_t>(ll) * 64;
            const int nrow = static_cast<int>(min(static_cast<int64_t>(64), static_cast<int64_t>(m_l) - r0));
            const int64_t l0 = q0_l + r0;
            const int64_t i0 = Q.rc ? Q.n - 1 - l0 : l0;
            const int64_t step = Q.rc ? -1 : 1;
            unsigned long long plane[NPL];
#pragma unroll
            for (int k = 0; k < NPL; ++k) plane[k] = 0ull;
#pragma unroll 1
            for (int rb = 0; rb < 64; rb += 16) {
                uint32_t raw[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) raw[u] = Q.p[i0 + step * min(rb + u, nrow - 1)];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t code = codes[raw[u]];
#pragma unroll
                    for (int k = 0; k < NPL; ++k) plane[k] |= static_cast<unsigned long long>((code >> k) & 1u) << (rb + u);
                }
            }
#pragma unroll
            for (int k = 0; k < NPL; ++k) { L.pl[k] = static_cast<uint32_t>(plane[k]); L.ph[k] = static_cast<uint32_t>(plane[k] >> 32); }
            const unsigned long long valid = nrow >= 64 ? ~0ull : ((1ull << nrow) - 1ull);
            L.vl = static_cast<uint32_t>(valid); L.vh = static_cast<uint32_t>(valid >> 32);
        }
        RCN_PP_T(pp1__);
#ifdef RCN_PROF_PAIR
        long long pp_steady__ = 0, pp_nsteady__ = 0;
#endif
        uint32_t* ringA = ring;
        uint32_t* ringB = ring + kPairRing * EW;
        const uint32_t first_b = lane == nbA ? 0u : ~0u;           // the lane that starts leaf B takes the top boundary {0, 0}
        PairCarry hc{0x80000000u, 0u};
        const int steps = max(A.n + nbA, B.n + nbB) - 1;
        const int nw_max = max(nbA, nbB), n_min = min(A.n, B.n);
        for (int s0 = 0; s0 < steps; s0 += 64) {
            {   // the 64 columns of each leaf that enter at its first lane during this block
                const int col = s0 + lane;
                const uint32_t ra = col < A.n ? T.p[A.t0 + col] : 0u, rb_ = col < B.n ? T.p[B.t0 + col] : 0u;
                const PairSym<NPL> ya = pair_sym_of<NPL>(col < A.n ? static_cast<int>(codes[256 + ra]) : 7);
                const PairSym<NPL> yb = pair_sym_of<NPL>(col < B.n ? static_cast<int>(codes[256 + rb_]) : 7);
                const int slot = (s0 >> 6) & 1;
                uint32_t* ea = ringA + (slot * 64 + lane) * EW;
                uint32_t* eb = ringB + (slot * 64 + lane) * EW;
                *reinterpret_cast<uint2*>(ea) = make_uint2(ya.mk[0], ya.mk[1]); *reinterpret_cast<uint2*>(eb) = make_uint2(yb.mk[0], yb.mk[1]);
                if (slot == 0) { *reinterpret_cast<uint2*>(ea + 128 * EW) = make_uint2(ya.mk[0], ya.mk[1]); *reinterpret_cast<uint2*>(eb + 128 * EW) = make_uint2(yb.mk[0], yb.mk[1]); }
            }
            const uint32_t* rp = (sb ? ringB : ringA) + ((s0 - ll) & 127) * EW;
            auto ring_at = [&](int k) -> PairSym<NPL> {
                PairSym<NPL> y;
                const uint2 v = *reinterpret_cast<const uint2*>(rp + k * EW);
                y.mk[0] = v.x; y.mk[1] = v.y;
                return y;
            };
            auto cell = [&](const PairSym<NPL>& y, PairLane<NPL>& Lw, uint32_t& nl, uint32_t& nh) {
                PairCarry cin;
                cin.np = pair_shr1_zero(hc.np) & first_b; cin.mn = pair_shr1_zero(hc.mn) & first_b;
                hc = pair_cell<NPL>(Lw, y, cin, nl, nh);
            };
            if (s0 >= nw_max - 1 && s0 + 63 < n_min) {
                RCN_PP_T(ppb__);
                PairSym<NPL> ya[2] = {ring_at(0), ring_at(1)}, yb[2] = {ring_at(2), ring_at(3)};
                auto step = [&](int k, const PairSym<NPL>& y) {
                    uint32_t nl, nh;
                    cell(y, L, nl, nh);
                    if (ll < nw_l) {
                        ulonglong2 v;
                        v.x = (static_cast<unsigned long long>(L.Pvh) << 32) | L.Pvl; v.y = (static_cast<unsigned long long>(nh) << 32) | nl;
                        sp[static_cast<int64_t>(s0 + k) * nw_l] = v;
                    }
                };
#pragma unroll 1
                for (int k = 0; k < 64; k += 4) {
                    step(k, ya[0]); step(k + 1, ya[1]);
                    ya[0] = ring_at(k + 4); ya[1] = ring_at(k + 5);
                    step(k + 2, yb[0]); step(k + 3, yb[1]);
                    yb[0] = ring_at(k + 6); yb[1] = ring_at(k + 7);
                }
#ifdef RCN_PROF_PAIR
                pp_steady__ += clock64() - ppb__; pp_nsteady__ += 64;
#endif
                continue;
            }
            const int kend = min(64, steps - s0);
            PairSym<NPL> yn = ring_at(0);
#pragma unroll 2
            for (int k = 0; k < kend; ++k) {
                const PairSym<NPL> y = yn;
                yn = ring_at(k + 1);
                const int j = s0 + k - ll;
                const bool active = ll < nw_l && static_cast<unsigned>(j) < static_cast<unsigned>(n_l);
                PairLane<NPL> Lt = L;
                uint32_t nl, nh;
                cell(y, Lt, nl, nh);
                L.Pvl = active ? Lt.Pvl : L.Pvl; L.Pvh = active ? Lt.Pvh : L.Pvh; L.Mvl = active ? Lt.Mvl : L.Mvl; L.Mvh = active ? Lt.Mvh : L.Mvh;
                if (active) {
                    ulonglong2 v;
                    v.x = (static_cast<unsigned long long>(L.Pvh) << 32) | L.Pvl; v.y = (static_cast<unsigned long long>(nh) << 32) | nl;
                    sp[static_cast<int64_t>(s0 + k) * nw_l] = v;
                }
            }
        }
#ifdef RCN_PROF_PAIR
        { const long long pp2__ = clock64();
          RCN_PP_ADD(0, pp_steady__); RCN_PP_ADD(1, pp2__ - pp1__ - pp_steady__); RCN_PP_ADD(2, pp1__ - pp0__); RCN_PP_ADD(9, pp_nsteady__); RCN_PP_ADD(10, steps - pp_nsteady__);
          RCN_PP_ADD(11, 1); RCN_PP_ADD(14, pp_nsteady__ * (nbA + nbB)); }
#endif
    }
    pair_wave_fence();
#pragma unroll 1
    for (int u = 0; u < 2; ++u) {
        const PairLeafJob& J = u ? B : A;
        pair_leaf_walk(u ? storeB : storeA, (J.m + 63) / 64, J.m, J.n, static_cast<int64_t>(J.q0) + J.t0, ops);
    }
}

// A sub-problem: rows [q0, q0 + m) x columns [t0, t0 + n), `best` = its distance (-1: the root).  lf / rt >= 0: its left / right
// column vector is INHERITED -- it already stands in the team's arena at that offset (lf_more / rt_more further vectors of the
// same pass follow at lf_stride / rt_stride: the next sub-problems down the same side).
struct PairTask { int q0, m, t0, n, best, lf, lf_more, lf_stride, rt, rt_more, rt_stride; };
// ints of a team's arena of inherited vectors: every split stores at most kPairSnap vectors of (rows + 1) ints per computed pass,
// the rows of one level of the recursion add up to the overlap's rows, and a 10 kb overlap is ~10 levels deep; a split that
// finds the arena full stores nothing and its sub-problems compute both their vectors
__host__ __device__ __forceinline__ uint64_t pair_arena_ints(uint64_t m_cap) { return 40ull * (m_cap + 64); }

// bytes of a team's scratch: two last-column vectors; two carry buffers, a store for two leaves and the snapshot words per wave; the arena
__host__ __device__ __forceinline__ uint64_t pair_slot_bytes(uint64_t m_cap, uint64_t n_cap) {       // (m_cap, n_cap: multiples of 16)
    return ((2 * 4 * (m_cap + 64) + 4 * (n_cap + 64) + 4 * pair_leaf_bytes(m_cap) + 2 * kPairSnap * 64 * 16 + 4 * pair_arena_ints(m_cap)) + 255) & ~uint64_t(255);
}

__device__ __forceinline__ PairTask pair_task_uniform(const PairTask& t) {
    PairTask u;
    u.q0 = __builtin_amdgcn_readfirstlane(t.q0); u.m = __builtin_amdgcn_readfirstlane(t.m); u.t0 = __builtin_amdgcn_readfirstlane(t.t0);
    u.n = __builtin_amdgcn_readfirstlane(t.n); u.best = __builtin_amdgcn_readfirstlane(t.best);
    u.lf = __builtin_amdgcn_readfirstlane(t.lf); u.lf_more = __builtin_amdgcn_readfirstlane(t.lf_more); u.lf_stride = __builtin_amdgcn_readfirstlane(t.lf_stride);
    u.rt = __builtin_amdgcn_readfirstlane(t.rt); u.rt_more = __builtin_amdgcn_readfirstlane(t.rt_more); u.rt_stride = __builtin_amdgcn_readfirstlane(t.rt_stride);
    return u;
}
__device__ __forceinline__ bool pair_task_splits(const PairTask& t) { return t.m > 0 && t.n > 0 && !pair_is_leaf(t.m, t.n); }

// The columns a split's computed passes leave behind for the sub-problems below it (see pair_align_one), and their ints in the arena.
struct PairSplitPlan { bool have_l, have_r; int nsf, nsb, fc[kPairSnap], bc[kPairSnap], need; };
__device__ __forceinline__ PairSplitPlan pair_split_plan(const PairTask& t) {
    PairSplitPlan p{t.lf >= 0, t.rt >= 0, 0, 0, {0, 0, 0}, {0, 0, 0}, 0};
    const int lw = t.n / 2, rw = t.n - lw;
    if (!p.have_l) {
        int c = lw;
#pragma unroll
        for (int k = 0; k < kPairSnap; ++k) { c = c / 2; if (c >= 1 && p.nsf == k) { p.fc[k] = c; p.nsf = k + 1; } }
    }
    if (!p.have_r) {
        int c = rw;
#pragma unroll
        for (int k = 0; k < kPairSnap; ++k) { const int cn = c - c / 2; if (cn >= 1 && cn < c && p.nsb == k) { p.bc[k] = cn; p.nsb = k + 1; } c = cn; }
    }
    p.need = (p.nsf + p.nsb) * (t.m + 1);
    return p;
}

struct PairKids { PairTask k[2]; };

template <int NPL>
__device__ __forceinline__ int pair_align_one(const PairParams& P, const PairView& Q, const PairView& T, const uint8_t* codes, PairTask* stack,
                                              PairKids* kids, int* nkids, uint8_t* slot, uint8_t* ops, uint32_t* ring) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    // scratch of this team: left[] / right[] and the arena are shared (one wave writes, both read after the barrier), the carry
    // buffers, the leaf store and the snapshot words are per wave
    int32_t* left = reinterpret_cast<int32_t*>(slot);
    int32_t* right = left + (P.m_cap + 64);
    uint8_t* after = reinterpret_cast<uint8_t*>(right + (P.m_cap + 64));
    uint8_t* hbuf0 = after + static_cast<uint64_t>(wv) * 2 * (P.n_cap + 64);
    uint8_t* hbuf1 = hbuf0 + (P.n_cap + 64);
    after += 4ull * (P.n_cap + 64);
    ulonglong2* store = reinterpret_cast<ulonglong2*>(after + static_cast<uint64_t>(wv) * 2 * P.leaf_bytes);       // (room for two leaves: pair_leaf_two)
    after += 4 * P.leaf_bytes;
    ulonglong2* snapbuf = reinterpret_cast<ulonglong2*>(after) + wv * (kPairSnap * 64);
    after += 2 * kPairSnap * 64 * 16;
    int32_t* arena = reinterpret_cast<int32_t*>(after);
    int atop = 0;                                      // ints of the arena in use (the same number in both waves)
    const int acap = static_cast<int>(P.arena_ints);
    int sp = 0, distance = -1;
    if (threadIdx.x == 0) stack[0] = PairTask{0, static_cast<int>(Q.n), 0, static_cast<int>(T.n), -1, -1, 0, 0, -1, 0, 0};
    sp = 1;

    // ---- a split ----
    // left[h] = ED(rows[:h], left half of the columns), right[k] = ED(last k rows, right half), each one forward / backward
    // pass over all the rows -- unless the vector is inherited: the left half of a LEFT child starts in the corner its parent's
    // forward pass started in, so its left vector is a column of that pass (the child's middle column, rows 0 .. h: a prefix),
    // and the right vector of a RIGHT child is a column of the parent's backward pass.  Every computed pass therefore also
    // leaves the columns its next kPairSnap descendants down that side will ask for, and below the root a split costs one
    // pass instead of two (edlib computes both every time; the values are the same numbers).
    // where the optimal path crosses the middle column (the smallest such row), and the two sub-problems; false: no such row
    auto cut = [&](const PairTask& t, const PairSplitPlan& pl, int off, const int32_t* Lv, const int32_t* Rv, int& best, PairTask& lc, PairTask& rc) -> bool {
        const int m = t.m, lw = t.n / 2, rw = t.n - lw;
        if (best < 0) {
            // the root: best = min over all rows of left + right (every path crosses the middle column somewhere)
            int mn = 0x7fffffff;
#pragma unroll 4
            for (int h = lane; h <= m; h += 64) mn = min(mn, Lv[h] + Rv[m - h]);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mn = min(mn, __shfl_xor(mn, d));
            best = mn;
        }
        int h = -1;
        for (int h0 = 1; h0 < m && h < 0; h0 += 256) {           // (four chunks of rows per trip: eight loads in flight, not two)
            int sum[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int hh = h0 + 64 * u + lane; const int hc_ = min(hh, m); sum[u] = Lv[hc_] + Rv[m - hc_]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int hh = h0 + 64 * u + lane;
                const unsigned long long mask = __ballot(hh < m && sum[u] == best);
                if (mask && h < 0) h = h0 + 64 * u + __builtin_ctzll(mask);
            }
        }
        if (h < 0 && lw + Rv[m] == best) h = 0;
        if (h < 0 && Lv[m] + rw == best) h = m;
        if (h < 0) return false;
        const int ls = h > 0 ? Lv[h] : lw, rs = h < m ? Rv[m - h] : rw;
        rc = PairTask{t.q0 + h, m - h, t.t0 + lw, rw, rs, -1, 0, 0, -1, 0, 0};
        if (pl.have_r) { if (t.rt_more > 0) { rc.rt = t.rt + t.rt_stride; rc.rt_more = t.rt_more - 1; rc.rt_stride = t.rt_stride; } }
        else if (pl.nsb > 0) { rc.rt = off + pl.nsf * (m + 1); rc.rt_more = pl.nsb - 1; rc.rt_stride = m + 1; }
        lc = PairTask{t.q0, h, t.t0, lw, ls, -1, 0, 0, -1, 0, 0};
        if (pl.have_l) { if (t.lf_more > 0) { lc.lf = t.lf + t.lf_stride; lc.lf_more = t.lf_more - 1; lc.lf_stride = t.lf_stride; } }
        else if (pl.nsf > 0) { lc.lf = off; lc.lf_more = pl.nsf - 1; lc.lf_stride = m + 1; }
        return true;
    };

    // A round: the wave decides what it has to do -- at most one pass over a sub-problem's rows (pair_columns) and at most one leaf
    // (pair_leaf) --, does it at ONE call site each (the pass is inlined in its four carry variants per symbol-plane count: every
    // further call site is another copy of all of them), and the round's kind decides what happens to the result.
    while (sp > 0) {
        __syncthreads();                               // the stack as the last round left it; left[] / right[] are no longer read
        const PairTask A = pair_task_uniform(stack[sp - 1]);
        const bool A_splits = pair_task_splits(A);
        const bool both = A_splits && A.lf < 0 && A.rt < 0;      // both vectors to compute (the root; a sub-problem whose ancestors' columns ran out): one pass per wave
        PairTask B = A;
        bool two = false, B_splits = false;
        if (!both && sp >= 2) {
            // a round of sub-problems that need one wave each: a leaf, an empty one, a split with an inherited vector
            B = pair_task_uniform(stack[sp - 2]);
            B_splits = pair_task_splits(B);
            two = !(B_splits && B.lf < 0 && B.rt < 0);
        }
        // partners: a leaf below the root runs side by side with another leaf of the stack when their words fit a wave together
        // (pair_leaf_two); candidates are the entries right below the round's own, offered to wave 0's leaf first
        bool pairA = false, pairB = false;
        PairTask C = A, D = A;
        if (NPL == 2 && !both) {
            auto leafy = [&](const PairTask& t) { return t.m > 0 && t.n > 0 && t.best >= 0 && !pair_task_splits(t); };
            auto words = [&](const PairTask& t) { return (t.m + 63) / 64; };
            int ci = sp - (two ? 2 : 1) - 1;
            if (ci >= 0 && leafy(A)) {
                const PairTask X = pair_task_uniform(stack[ci]);
                if (leafy(X) && pair_leaf_pairs(words(A), words(X))) { C = X; pairA = true; --ci; }
            }
            if (ci >= 0 && two && leafy(B)) {
                const PairTask X = pair_task_uniform(stack[ci]);
                if (leafy(X) && pair_leaf_pairs(words(B), words(X))) { D = X; pairB = true; --ci; }
            }
        }
        sp -= (two ? 2 : 1) + (pairA ? 1 : 0) + (pairB ? 1 : 0);
        PairSplitPlan pa = pair_split_plan(A), pb = pair_split_plan(B);
        if (!A_splits || atop + pa.need > acap) { pa.nsf = 0; pa.nsb = 0; pa.need = 0; }
        const int offa = atop;
        atop += pa.need;
        if (!two || !B_splits || atop + pb.need > acap) { pb.nsf = 0; pb.nsb = 0; pb.need = 0; }
        const int offb = atop;
        atop += pb.need;
        // this wave's sub-problem
        const bool mine_is_A = both || wv == 0;
        const bool active = both || wv == 0 || two;
        const PairTask& t = mine_is_A ? A : B;
        const PairSplitPlan& pl = mine_is_A ? pa : pb;
        const int off = mine_is_A ? offa : offb;
        const bool splits = mine_is_A ? A_splits : B_splits;
        int32_t* mine = wv == 0 ? left : right;                  // the vector this wave computes
        const int64_t base = static_cast<int64_t>(t.q0) + t.t0;
        const bool empty = t.m == 0 || t.n == 0;
        const bool leaf = active && !empty && !splits;
        // the pass: a split's forward / backward half (in a `both` round wave 0 goes forwards, wave 1 backwards; otherwise the side
        // that is not inherited), or -- a leaf at the root, whose distance no split has provided -- forwards over all the columns
        const bool do_pass = active && !empty && (splits || t.best < 0);
        if (do_pass) {
            const int lw = splits ? t.n / 2 : t.n, rw = t.n - lw;
            const bool forward = !splits || (both ? wv == 0 : !pl.have_l);
            const int ns = !splits ? 0 : forward ? pl.nsf : pl.nsb;
            const int* sc = forward ? pl.fc : pl.bc;
            const int d = pair_columns<NPL>(Q, t.q0, t.m, !forward, T, forward ? t.t0 : t.t0 + lw, forward ? lw : rw, !forward, codes, hbuf0, hbuf1, mine, ring,
                                            ns, sc[0], sc[1], sc[2], arena + off + (forward ? 0 : pl.nsf * (t.m + 1)), t.m + 1, snapbuf);
            if (!splits) distance = d;
        }
        const bool with_partner = mine_is_A ? pairA : pairB;
        if (NPL == 2 && leaf && with_partner) {
            if constexpr (NPL == 2) {
                const PairTask& u = mine_is_A ? C : D;
                pair_leaf_two<NPL>(Q, T, codes, PairLeafJob{t.q0, t.m, t.t0, t.n}, PairLeafJob{u.q0, u.m, u.t0, u.n}, store, ops, ring);
            }
        } else if (leaf) pair_leaf<NPL>(Q, t.q0, t.m, T, t.t0, t.n, codes, hbuf0, hbuf1, store, ops, ring);
        if (active && empty) {
            if (t.m == 0) { for (int k = lane; k < t.n; k += 64) ops[base + k] = 'D'; if (t.best < 0) distance = t.n; }
            else { for (int k = lane; k < t.m; k += 64) ops[base + k] = 'I'; if (t.best < 0) distance = t.m; }
        }
        if (both) {
            RCN_PP_T(ppb0__);
            __syncthreads();                           // (work-group fence + barrier: the other wave's vector is in HBM scratch)
#ifdef RCN_PROF_PAIR
            RCN_PP_ADD(8, clock64() - ppb0__);
#endif
            int best = A.best;
            PairTask lc, rc;
            RCN_PP_T(ppc0__);
            const bool ok = cut(A, pa, offa, left, right, best, lc, rc);
#ifdef RCN_PROF_PAIR
            RCN_PP_ADD(5, clock64() - ppc0__); RCN_PP_ADD(13, 1);
#endif
            if (A.best < 0) distance = best;
            if (!ok || sp + 2 > kPairStack) { if (threadIdx.x == 0) atomicAdd(P.err, 1u); break; }
            if (threadIdx.x == 0) { stack[sp] = rc; stack[sp + 1] = lc; }
            sp += 2;
            continue;
        }
        int nk = 0;
        PairTask k0 = A, k1 = A;
        if (active && splits) {
            const int32_t* Lv = pl.have_l ? arena + t.lf : mine;
            const int32_t* Rv = pl.have_r ? arena + t.rt : mine;
            int best = t.best;
            RCN_PP_T(ppc1__);
            nk = cut(t, pl, off, Lv, Rv, best, k1, k0) ? 2 : -1;
#ifdef RCN_PROF_PAIR
            RCN_PP_ADD(5, clock64() - ppc1__); RCN_PP_ADD(13, 1);
#endif
        }
        if (lane == 0) { kids[wv].k[0] = k0; kids[wv].k[1] = k1; nkids[wv] = nk; }
        RCN_PP_T(ppb1__);
        __syncthreads();
#ifdef RCN_PROF_PAIR
        RCN_PP_ADD(8, clock64() - ppb1__);
#endif
        const int n0 = __builtin_amdgcn_readfirstlane(nkids[0]), n1 = __builtin_amdgcn_readfirstlane(nkids[1]);
        if (n0 < 0 || n1 < 0 || sp + n0 + n1 > kPairStack) { if (threadIdx.x == 0) atomicAdd(P.err, 1u); break; }
        if (threadIdx.x == 0) {
            int q = sp;
            if (n1 == 2) { stack[q] = kids[1].k[0]; stack[q + 1] = kids[1].k[1]; q += 2; }
            if (n0 == 2) { stack[q] = kids[0].k[0]; stack[q + 1] = kids[0].k[1]; }
        }
        sp += n0 + n1;
    }
    return distance;
}

// One team (two waves) per overlap, persistent over the work queue.  The kernels of this file are compiled in a translation unit of
// their own (engine_pair.hip: the passes are inlined in four carry variants per symbol-plane count, two minutes of compile time that
// engine.hip's other kernels need not wait for); engine.hip sees the declarations and launches them.
constexpr int kPairThreads = 128;
#if !defined(RCN_PAIR_TU) && !defined(RCN_ONE_TU)
__global__ void k_pair_align(PairParams P);
#else
__global__ __launch_bounds__(kPairThreads, 4) void k_pair_align(PairParams P) {
    __shared__ uint8_t codes[512];          // [raw query byte] -> plane code of the (complemented) symbol; [256 + raw target byte] -> plane code
    __shared__ uint8_t present[256];        // raw bytes of the stored query segment
    __shared__ uint8_t lcode[256];          // logical symbol -> dense code
    __shared__ PairTask stack[kPairStack];
    __shared__ PairKids kids[2];
    __shared__ int nkids[2];
    __shared__ unsigned int s_work;
    __shared__ int s_nsym;
    __shared__ int s_foreign;               // the target segment holds a symbol the query does not
    __shared__ __attribute__((aligned(16))) uint32_t symring[2][kPairRing * 4];     // per wave: the column symbols' plane masks on their way down the lanes (pair_pass_impl)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t* slot = P.scratch + static_cast<uint64_t>(blockIdx.x) * P.slot_bytes;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_work = atomicAdd(P.next, 1u);
        __syncthreads();
        const unsigned int wi = __builtin_amdgcn_readfirstlane(s_work);
        if (wi >= P.n_pairs) break;
        RCN_PP_T(ppk0__);
        const uint32_t o = P.order[wi];
        PairView Q{P.bases + P.q_pos[o], P.q_rc[o] != 0, static_cast<int64_t>(P.q_len[o])};
        PairView T{P.bases + P.t_pos[o], false, static_cast<int64_t>(P.t_len[o])};
        uint8_t* ops = P.ops + P.ops_off[o];
        // dense codes of the query's symbols (in byte order of the symbols as aligned, i.e. complemented for a reverse strand);
        // 7 = "not in the query" for target symbols
        for (int k = tid; k < 256; k += kPairThreads) present[k] = 0;
        __syncthreads();
        for (int64_t i = tid; i < Q.n; i += 4 * kPairThreads) {
            uint32_t r4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int64_t iu = i + u * kPairThreads; r4[u] = Q.p[iu < Q.n ? iu : i]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) present[r4[u]] = 1;
        }
        __syncthreads();
        if (wv == 0) {
            // four symbols per lane, exclusive count across the wave
            int cnt = 0, pr[4];
            for (int k = 0; k < 4; ++k) { const uint32_t c = 4 * lane + k; pr[k] = present[Q.rc ? pair_comp(c) : c]; cnt += pr[k]; }
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
            if (lane == 63) s_nsym = incl;
            int code = incl - cnt;
            for (int k = 0; k < 4; ++k) { lcode[4 * lane + k] = pr[k] ? static_cast<uint8_t>(min(code, 7)) : 7; code += pr[k]; }
        }
        __syncthreads();
        const int nsym = __builtin_amdgcn_readfirstlane(s_nsym);
        // four symbols or fewer and none in the target that the query lacks (plain ACGT reads): two bit planes instead of three
        if (tid == 0) s_foreign = 0;
        __syncthreads();
        if (nsym <= 4) {
            int foreign = 0;
            for (int64_t i = tid; i < T.n; i += 4 * kPairThreads) {
                uint32_t r4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int64_t iu = i + u * kPairThreads; r4[u] = T.p[iu < T.n ? iu : i]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) foreign |= lcode[r4[u]] == 7 ? 1 : 0;
            }
            if (foreign) s_foreign = 1;
        }
        for (int k = tid; k < 256; k += kPairThreads) {
            const uint32_t cq = Q.rc ? pair_comp(static_cast<uint32_t>(k)) : static_cast<uint32_t>(k);
            codes[k] = nsym <= 7 ? lcode[cq] : static_cast<uint8_t>(cq);
            codes[256 + k] = nsym <= 7 ? lcode[k] : static_cast<uint8_t>(k);
        }
        __syncthreads();
        RCN_PP_T(ppk1__);
        int d;
        if (nsym <= 4 && __builtin_amdgcn_readfirstlane(s_foreign) == 0) d = pair_align_one<2>(P, Q, T, codes, stack, kids, nkids, slot, ops, symring[wv]);
        else if (nsym <= 7) d = pair_align_one<3>(P, Q, T, codes, stack, kids, nkids, slot, ops, symring[wv]);
        else d = pair_align_one<8>(P, Q, T, codes, stack, kids, nkids, slot, ops, symring[wv]);
        if (tid == 0) P.dist[o] = d;
#ifdef RCN_PROF_PAIR
        RCN_PP_ADD(6, ppk1__ - ppk0__); RCN_PP_ADD(7, clock64() - ppk0__);
#endif
    }
}

#endif  // RCN_PAIR_TU

// ---- breaking points from the op bytes (Overlap::find_breaking_points' walk, reference src/overlap.cpp:226-292) ----
// One wave per overlap, 64 path positions per step: two wave scans turn "consumes a target base" / "consumes a query
// base" into positions; the match columns of a step that fall into one window are a contiguous lane range, its first
// lane is the window's first candidate, its last lane the window's last.  Same slot convention as the CIGAR walk: one
// slot per window end the overlap touches, a window counts once the walk has passed its end.
struct OpsWalkParams {
    const uint8_t* ops; const uint64_t* ops_off;      // [n + 1]
    const uint32_t* q_start; const uint32_t* t_begin; const uint32_t* t_end;
    const uint64_t* bp_off; uint32_t* bp_t; uint32_t* bp_q;
    unsigned long long* first_key; unsigned long long* last_key;     // [slots]: ~0 / 0
    uint64_t n_overlaps; uint64_t W;
};

#if !defined(RCN_PAIR_TU) && !defined(RCN_ONE_TU)
__global__ void k_ops_breaking_points(OpsWalkParams C);
#else
__global__ __launch_bounds__(256) void k_ops_breaking_points(OpsWalkParams C) {
    const int lane = threadIdx.x & 63;
    const uint64_t o = static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (o >= C.n_overlaps) return;
    const uint64_t W = C.W;
    const uint64_t t_begin = C.t_begin[o], t_end = C.t_end[o];
    const uint64_t slot0 = C.bp_off[o] / 2, n_slots = (C.bp_off[o + 1] - C.bp_off[o]) / 2;
    const uint64_t wb = t_begin / W;
    auto end_of = [&](uint64_t k) -> uint64_t { return k + 1 < n_slots ? (wb + 1 + k) * W - 1 : t_end - 1; };
    const uint64_t a = C.ops_off[o], z = C.ops_off[o + 1];
    uint64_t t_run = t_begin, q_run = C.q_start[o];
    for (uint64_t p0 = a; p0 < z; p0 += 64) {
        const uint64_t p = p0 + lane;
        const uint32_t op = p < z ? C.ops[p] : 0;
        const bool isM = op == 'M';
        const unsigned int dt = (isM || op == 'D') ? 1u : 0u, dq = (isM || op == 'I') ? 1u : 0u;
        unsigned int st = dt, sq = dq;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int vt = __shfl_up(st, d), vq = __shfl_up(sq, d);
            if (lane >= d) { st += vt; sq += vq; }
        }
        const uint64_t ts = t_run + (st - dt), qs = q_run + (sq - dq);
        const uint64_t k = isM ? ts / W - wb : 0;
        unsigned long long rem = __ballot(isM);
        while (rem) {
            const int fl = __builtin_ctzll(rem);
            const uint64_t kk = __shfl(k, fl);
            const unsigned long long mk = __ballot(isM && k == kk);
            const int ll = 63 - __builtin_clzll(mk);
            const uint64_t ft = __shfl(ts, fl), fq = __shfl(qs, fl), lt = __shfl(ts, ll) + 1, lq = __shfl(qs, ll) + 1;
            if (lane == 0 && kk < n_slots) {          // positions grow along the walk: min keeps the first, max the last
                atomicMin(&C.first_key[slot0 + kk], (ft << 32) | fq);
                atomicMax(&C.last_key[slot0 + kk], (lt << 32) | lq);
            }
            rem &= ~mk;
        }
        t_run += __shfl(st, 63); q_run += __shfl(sq, 63);
    }
    __threadfence();
    for (uint64_t k = lane; k < n_slots; k += 64) {
        const unsigned long long f = C.first_key[slot0 + k], l = C.last_key[slot0 + k];
        if (l != 0 && f != ~0ull && end_of(k) + 1 <= t_run) {
            const uint64_t out = C.bp_off[o] + 2 * k;
            C.bp_t[out] = static_cast<uint32_t>(f >> 32); C.bp_q[out] = static_cast<uint32_t>(f);
            C.bp_t[out + 1] = static_cast<uint32_t>(l >> 32); C.bp_q[out + 1] = static_cast<uint32_t>(l);
        }
    }
}

#endif  // RCN_PAIR_TU

}  // namespace rcn
