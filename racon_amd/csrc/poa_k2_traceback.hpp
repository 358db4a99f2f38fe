// poa_k2_traceback.hpp -- phases: traceback over int16 scores and over move codes (box walker)
// Part of the fast path of the MI355X window-consensus engine: included by poa_kernel2.hpp, in this order, into one
// translation unit (see its header for the design).
#pragma once

namespace rcn {

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
                if (c.tie_pad[2] & 1) break;             // test switch (KParams::force_slow_tb bit 0): no box walk at all
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
                  // (static indices only: a run-time index sends the whole array to scratch memory, and every box then pays a
                  //  global-memory round trip inside the interval that is being measured)
#pragma unroll
                  for (int k__ = 0; k__ < 6; ++k__) ex__[k__] += (why__ == k__) ? 1 : 0;
                  ex__[6] += __popcll(vis); }
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

}  // namespace rcn
