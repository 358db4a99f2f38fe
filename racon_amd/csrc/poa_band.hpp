// poa_band.hpp — the EXACT banded variant of the one-wave NW DP of poa_kernel2.hpp (included from there).
//
// What it computes: the same alignment as dp2_rows<NP, 1> (spoa::AlignmentEngine::Align, kNW, linear gap; reference
// src/window.cpp:95-97,104-106), but every row only over a WINDOW of WB = 128 * NP columns that follows the
// row's backbone coordinate, two cells per VGPR, NP VGPRs per lane instead of ceil((len + 1) / 128): at w = 500 half
// the per-row instruction slots of the full row, half the matrix bytes.  Cells outside their row's window count as -inf.
//
// Why the result is still the full matrix's (the certificate; restated and measured on the CPU in
// the CPU checker's band study, tools/band_study.py):
//   Let T be the best end score the banded pass finds (some sink row, column len).  Call a computed cell ALIVE when
//   H'[i][j] + m (len - j) >= T: no path through a dead cell can reach T, because every remaining sequence base
//   adds at most m.  If every DP successor of every alive cell is a computed cell, then by induction along any full
//   path P with score >= T all of P's cells are alive and carry H' = H (a successor of an alive cell is computed, its
//   banded value is at least P's prefix score, hence alive again; and H' <= H always).  So T is the optimum, the tied
//   sinks are the true ones, and every equality the traceback tests comes out as on the full matrix: a matching
//   candidate lies on a co-optimal path (exact value), a non-matching one can only have become smaller.
//   The kernel verifies "successors of alive cells are computed" with what a row costs nothing to record:
//     (a) the LAST window cell of every row whose window ends left of column len must be dead (its horizontal and
//         diagonal successors lie outside);
//     (b) when the window moves right by delta columns, the delta leftmost cells of the rows that later rows can
//         still read (register window at the shift; LDS ring rows when they are read) must be dead;
//     (c) a source row (predecessor = the virtual row 0, alive at least in column 0) must have window offset 0;
//     (d) a predecessor older than the LDS ring is not supported.
//   Alive is tested as  Z - (m - g) j >= T - m len  with Z = H - j g the stored value: the left side is recorded as
//   a running maximum while T is still unknown.  Any violation -> ctx->band_fail = 1 and the caller redoes this one
//   alignment with the unbanded DP (measured on cfg2: < 0.2 % of the alignments).
//
// Matrix layout is unchanged (absolute columns, row stride hstride), so traceback / sink-tie code does not know about
// the band; cells it may look at outside a window are kept at -inf by guard stores: the cell left of a row's window,
// the columns a shift adds for the rows of the LDS ring, column len of sink rows whose window ends before it.
#pragma once

namespace rcn {

#ifndef RCN_BAND_G
#define RCN_BAND_G 32
#endif
constexpr int kBandG = RCN_BAND_G;    // window offsets are multiples of this many columns (64: half as many window shifts, a worse-centred window -- A/B in profiles/r03/r_band_granularity.txt and r06/z_)
constexpr int kBandSeq = 1088;        // LDS copy of the layer's bases (window shifts re-read their columns from it) + the cold window state
constexpr int kBandRing = 32;         // rows of the LDS ring of the banded DP (a power of two and a multiple of the octet: see dp2_rows_band_body)

// NP of the banded DP for a layer of `len` bases (0 = not banded).  The alive zone is about len / 3 wide
// (profiles/r02/band_study.txt), so a window serves layers up to ~2.5 x its width.
__host__ __device__ __forceinline__ int band_np(int len) {
    const int W = len + 1;
    if (W > 256 && W <= 640) return 2;
    return 0;
}
__host__ __device__ constexpr int dp2_band_slots(int np, bool tab) {       // row slots the work area has room for
    return (kLdsBytes - 64 - kBandSeq - (tab ? 4 * 4 * 64 * np : 0)) / (4 * 64 * np);
}
__host__ __device__ constexpr int dp2_ring_rows_band(int np, bool tab) { return np == 2 ? kBandRing : dp2_band_slots(np, tab) - 1; }

// ---- the code wave (HELP) ----
// A window that has a CU to itself (the deep launch of engine.hip's split: one work-group per CU, enforced by asking for
// the CU's whole LDS) is as slow as its DP wave issues instructions, and three SIMDs idle next to it.  With the large LDS
// at hand the DP wave (wave 0) leaves the move codes of the chain / fast rows -- 98 % of the rows; 12 of a chain row's 82
// instructions, 20-30 of a row with several predecessors (the running first-argmax included) -- to waves 1-3: the finished
// rows go into a ring of kHelpRows rows behind the context instead of the ~27 of the work area, wave 0 counts them in a
// mailbox word, and the code waves follow at their own pace, each taking every third row (re-deriving a row's inputs costs
// more instructions than wave 0 saves, but nothing in it waits for the previous row): they re-read the row and its
// predecessor rows from the ring, recombine them (same arithmetic, first-argmax included), and store the code word.
// Window shifts they replay from the row offsets; rows of the other classes (general, medium, sink: 2 %) keep their
// codes in wave 0.  Wave 0 checks once per 64 rows that no code wave is 128 rows behind (the ring holds 256).  Every wait
// is bounded: a wave that waits too long reports the alignment as failed and the caller redoes it on full rows, like a
// failed certificate.
constexpr int kHelpRows = 256;                                   // rows of the large ring (NP = 2: 512 bytes each)
constexpr int kHelpRingBytes = kHelpRows * 64 * 2 * 4;
constexpr int kHelpLdsBytes = kHelpRingBytes + 64 * 4 + 16;      // + progress words (one per lane of wave 0) + {rows done by waves 1-3, failure flag}
#ifndef RCN_HELP_SLEEP
#define RCN_HELP_SLEEP 1                                           // s_sleep argument between two polls of a code wave (experiment builds: -DRCN_HELP_SLEEP=n)
#endif
constexpr int kHelpSpin = 1 << 18;                               // polls (s_sleep RCN_HELP_SLEEP each) before a wait gives up
__device__ __forceinline__ uint32_t* help_ring() { return reinterpret_cast<uint32_t*>(lds_words2() + (kLdsBytes + kCtxBytes) / 4); }
__device__ __forceinline__ uint32_t* help_prog() { return help_ring() + kHelpRingBytes / 4; }        // [64] rows finished by wave 0, x 0x00010001; 0xffffffff = gave up
__device__ __forceinline__ uint32_t* help_done() { return help_prog() + 64; }                        // [0..2] rows waves 1-3 are through with, [3] one of them gave up
__device__ __forceinline__ void lds_store(uint32_t* p, uint32_t v) {
    const uint32_t a = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));
    asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v) : "memory");
}

__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) {        // saturating a - b per half (v_pk_sub_i16 clamp)
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, static_cast<s16x2>(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)));
}

// phase_desc2, banded alignments: window offset of every row (g.pred[r] = offset of DP row r + 1).  A row's coordinate
// is the last backbone node at or before it in the order being aligned (new nodes sit right behind their anchors, so
// this is the backbone position the row belongs to within a few columns); the window is centred on the column that
// position maps to, quantised to kBandG columns, never moving left.
__device__ __forceinline__ int band_offset_of(int pi, int bb0, float scale, int WB, int offmax) {
    const int rel = max(0, pi - bb0 + 1);
    int o = static_cast<int>(static_cast<float>(rel) * scale) - WB / 2;
    o = o < 0 ? 0 : (o & ~(kBandG - 1));
    return min(o, offmax);
}

__device__ __forceinline__ void band_row_offsets(const Ctx& c, Win& g, RCN_G const int32_t* rank) {
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    RCN_G int32_t* roff = g.pred.ptr();
    const int V = c.V, WB = 128 * c.band, W = c.len + 1;
    const int offmax = min(((W - WB + kBandG - 1) / kBandG) * kBandG, c.hstride - WB);
    const int bb0 = c.sub ? c.begin : 0, nbb = c.sub ? (c.end - c.begin + 1) : c.bblen;
    const float scale = static_cast<float>(c.len) / static_cast<float>(nbb > 0 ? nbb : 1);
    int* xch = Block4::work();
    int carry = -1;
    for (int base = 0; base < V; base += 4 * kThreads2) {
        int pv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = base + 4 * t + k;
            const int v = r < V ? rank[r] : 0x7fffffff;
            pv[k] = v < c.bblen ? v : -1;
        }
        pv[1] = max(pv[1], pv[0]); pv[2] = max(pv[2], pv[1]); pv[3] = max(pv[3], pv[2]);
        const int incl = wave_incl_scan_max(pv[3]);
        const int excl = wave_shr1(incl, -1);
        int off, total, pmax, tmax;
        block4_scan(xch, wv, lane, 0, __builtin_amdgcn_readlane(incl, 63), off, total, pmax, tmax);
        const int before = max(max(excl, pmax), carry);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = base + 4 * t + k;
            if (r < V) roff[r] = band_offset_of(max(before, pv[k]), bb0, scale, WB, offmax);
        }
        carry = max(carry, tmax);
    }
    Block4::sync();
}

#ifdef RCN_PROF_ROWS
// profiling build: clocks and counts of the banded DP's rows by class (0 chain, 1 / 2 / 3 = fast rows with one / two /
// three or four predecessors, 4 medium, 5 general, 6 rows where the window moved, 7 sink rows, 8 rows inside octets), spread over 256 copies
__device__ unsigned long long g_rowprof[256][20];
// sections of the rare rows (window shifts, general rows, their tail): clocks [k] and counts [20 + k] of section k, see engine.hip for the names
__device__ unsigned long long g_secprof[256][40];
#define RCN_SEC(k, dt) do { if (CODE && lane == 0) { atomicAdd(&g_secprof[blockIdx.x & 255][(k)], static_cast<unsigned long long>(dt)); atomicAdd(&g_secprof[blockIdx.x & 255][20 + (k)], 1ull); } } while (0)
#define RCN_LAP0() do { sec_last = clock64(); } while (0)
#define RCN_LAP(k, dep) do { const long long n__ = clock64() + (static_cast<int>(dep) & 0); RCN_SEC(k, n__ - sec_last); sec_last = clock64(); } while (0)
#else
#define RCN_LAP0() do {} while (0)
#define RCN_LAP(k, dep) do {} while (0)
#endif
// ---- the banded one-wave DP (wave 0 of the work-group) ----
// CODE: instead of the row of scores the wave stores, per cell, what the traceback would find out from the scores (one
// byte: bit 0 clear = a diagonal move reproduces the cell, bit 1 clear = a vertical one does, bits 2-4 / 5-7 = the first
// predecessor in in-edge order that attains the predecessor maximum at the previous / at this column) -- see
// phase_traceback_code.  A quarter of the bytes of the full int16 row, and the traceback neither re-reads scores nor
// compares them.  Rows with more than EIGHT in-edges (three bits name a predecessor) are left to the score-matrix path
// (band_fail); the seventh and eighth are not in the row descriptor, the traceback finds them on the in-edge list.
template <int NP, bool TAB, bool CODE = false, bool HELP = false>
__device__ __forceinline__ void dp2_rows_band_body() {
    static_assert(!CODE || NP == 2, "move codes: four cells per lane -> one dword per lane and row");
    static_assert(!HELP || CODE, "the code wave assembles move codes");
    constexpr int NTH = 64, WB = 128 * NP, LPC = 2 * NP;       // window columns, columns per lane
    const int t = threadIdx.x & 63, lane = t;
    const Ctx c = ctx_load<Block4>();
    Win g = ctx_win(c);
    RCN_G const RowDesc* desc = g.desc.ptr();
    RCN_G const int32_t* roff = g.pred.ptr();
    RCN_G const int32_t* nr = (c.sub ? g.n2r_x : g.n2r).ptr();
    RCN_G const int32_t* e_nin = g.e_nin.ptr();
    RCN_G const int32_t* e_tail = g.e_tail.ptr();
    RCN_G const uint8_t* inc = g.inc.ptr();
    const bool sub = c.sub != 0;
    RCN_G uint32_t* __restrict__ H = reinterpret_cast<RCN_G uint32_t*>(g.H.ptr());
    RCN_G int16_t* H16w = reinterpret_cast<RCN_G int16_t*>(g.H.ptr());
    RCN_G const uint8_t* seq = gcast(c.seq);
    const int V = c.V, len = c.len;
    const int hs = c.hstride, hs2 = hs >> 1;
    constexpr int kTab = TAB ? 4 * 4 * NTH * NP : 0;
    constexpr int KT = (kLdsBytes - 64 - kBandSeq - kTab) / (4 * NTH * NP);
    // HELP: the ring is the large one behind the context (phase_desc2 classifies rows for the small one: conservative)
    constexpr int K = HELP ? kHelpRows : kBandRing;
    static_assert(NP == 2 && kBandRing == dp2_ring_rows_band(NP, TAB) && KT == dp2_band_slots(NP, TAB), "phase_desc2 classifies rows with the same ring depth");
    static_assert(KT >= kBandRing && (K & (K - 1)) == 0 && K % 8 == 0, "ring: a power of two of rows, inside the work area");
    uint32_t* ring0 = reinterpret_cast<uint32_t*>(Block4::work());         // [KT][NTH][NP]
    uint32_t* ring = HELP ? help_ring() : ring0;
    uint32_t* ptab = ring0 + KT * NTH * NP;                                   // [4][NTH][NP] (TAB)
    uint8_t* lseq = reinterpret_cast<uint8_t*>(ptab + kTab / 4);             // [kBandSeq]
    for (int k = t; k < len; k += NTH) lseq[k] = seq[k];
    // this lane's slice of the profile table as an LDS byte address held in a VGPR: table slot -> address is one add
    typedef __attribute__((address_space(3))) const uint32_t* lds_cu32;
    uint32_t ptab_l = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ptab + t * NP));
    asm volatile("" : "+v"(ptab_l));

    const int mg = c.m - c.gp, xg = c.x - c.gp;
    uint32_t MG = pack2(mg, mg), XM = pack2(xg - mg, xg - mg), ONE = 0x00010001u;
    const uint32_t GG = pack2(c.gp, c.gp);
    const uint32_t NEGP = pack2(kNeg16, kNeg16);
    asm volatile("; constants live in VGPRs" : "+v"(MG), "+v"(XM), "+v"(ONE));
    uint32_t tie_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&Block4::ctx()->tie_rows[0]));
    asm volatile("" : "+s"(tie_base));
    uint32_t prog_l = HELP ? static_cast<uint32_t>(reinterpret_cast<uintptr_t>(help_prog() + t)) : 0u;   // this lane's progress word
    if (HELP) asm volatile("" : "+v"(prog_l));

    // ---- window state ----
    // State that only the rare rows touch (window shifts, general rows) lives in LDS, behind the layer's bases: kept in
    // registers it is loop-carried through the row loop, and the chain / fast rows pay for the copies that reconcile it.
    //   woff: first column of the window; rows >= s_row1 have offset woff, rows in [s_row2, s_row1) off1, [s_row3, s_row2) off2
    struct BandCold { int woff, s_row1, s_row2, s_row3, off1, off2; };
    static_assert(kBandSeq >= 1024 + static_cast<int>(sizeof(BandCold)), "cold state sits behind the bases (a layer has at most 640)");
    BandCold* cold = reinterpret_cast<BandCold*>(lseq + 1024);
    { BandCold z = {0, 0, 0, 0, 0, 0}; *cold = z; }
    auto cold_get = [&](const int* p_) { return __builtin_amdgcn_readfirstlane(*p_); };
    int bfail = 0;
#ifdef RCN_PROF_ROWS
    long long sec_last = clock64();
#endif
    uint32_t sqx[NP], thrv[NP];                 // bases / alive thresholds (m - g) * column of this lane's columns
    int own_lane = 0, own_q = 0, own_in = 0;    // where column len lives (own_in: inside the window)
    const int own_hi = len & 1;
    auto set_columns = [&]() {
        const int woff = cold_get(&cold->woff);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int j0 = woff + t * LPC + 2 * q, j1 = j0 + 1;
            // (loads at clamped indices + selects: no lane-dependent branch may sit inside the row loop, or the compiler
            //  structurizes the whole loop body into flag-guarded blocks and every row pays for their taken branches)
            int l0 = lseq[min(max(j0 - 1, 0), kBandSeq - 1)], l1 = lseq[min(max(j1 - 1, 0), kBandSeq - 1)];
            asm volatile("" : "+v"(l0), "+v"(l1));
            const int s0 = (j0 >= 1 && j0 <= len) ? l0 : 0x100, s1 = (j1 >= 1 && j1 <= len) ? l1 : 0x100;
            sqx[q] = pack2(s0, s1);
            thrv[q] = pack2(mg * j0, mg * j1);
        }
        RCN_LAP(3, sqx[0] + sqx[NP - 1]);
        if (TAB) {
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const uint32_t code = sl == 0 ? 'A' : sl == 1 ? 'C' : sl == 2 ? 'T' : 'G', symsym = code | (code << 16);
#pragma unroll
                for (int q = 0; q < NP; ++q) ptab[(sl * NTH + t) * NP + q] = pk_profile(sqx[q], symsym, ONE, XM, MG);
            }
        }
        RCN_LAP(4, 0);
        const int rel = len - woff;
        own_in = rel < WB;
        own_lane = (rel / LPC) & 63; own_q = (rel % LPC) >> 1;
    };
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);              // lseq written by this wave, read right below
    RCN_LAP0();
    set_columns();

    constexpr int R = dp2_window(NP);
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    u32x16 win;
#pragma unroll
    for (int k = 0; k < 16; ++k) win[k] = NEGP;
    uint32_t prev[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) prev[q] = NEGP;
    int zsh = static_cast<int>(0x80000000u);
    uint32_t mpv = static_cast<uint32_t>(kNeg16) << 16;
    // certificate accumulators: dropped cells (vector, Z - (m - g) j per half, saturating), last window cells (vector
    // of raw Z, reduced when the window moves), both as scalars in the H - m j form
    uint32_t emaxV = pack2(-32768, -32768), edgeR = NEGP;
    int edgeS = -(1 << 30);
    auto flush_edge = [&]() {
        const int woff = cold_get(&cold->woff);
        const int lastcol = woff + WB - 1;
        if (lastcol < len) {
            const int zr = static_cast<int>(__builtin_amdgcn_readlane(static_cast<int>(edgeR), 63)) >> 16;
            edgeS = max(edgeS, zr - mg * lastcol);
        }
        edgeR = NEGP;
    };

    int best = 0, best_row = 0, have_best = 0, tied = 0;
    unsigned int pred_rows = 0;             // predecessor rows combined (statistics: the algorithmic bytes of this alignment)
    int slot = 0;                               // ring slot of row i: (i - 1) & (K - 1) -- the eight rows of an octet never wrap
    RCN_G uint32_t* hrow = H + hs2;             // wave-uniform: row i of the matrix at the window's first column
    // CODE: one byte per cell, absolute columns, row stride hs BYTES (same base as the score matrix it replaces)
    RCN_G uint8_t* cbase = reinterpret_cast<RCN_G uint8_t*>(g.H.ptr());
    // this lane's dword of the current code row as a 32-bit offset in a VGPR: one vector add per row instead of a 64-bit
    // scalar add plus the lane term rebuilt for every store (the slot is ~3 MB: 32 bits are plenty)
    uint32_t coff = static_cast<uint32_t>(hs) + 4u * static_cast<uint32_t>(t);
    asm volatile("" : "+v"(coff));
    RCN_G int32_t* sinkz = g.path_node.ptr();   // CODE: end score (column len) of the sink rows, for phase_sink_tie_full
    int dl_p0 = 0, dl_p1 = 0, dl_p2 = 0, dl_p3 = 0, dl_p4 = 0, dl_p5 = 0, dl_er = -1, dl_meta = 1 << 9, dl_off = 0;
    unsigned int fast8 = 0u;                    // octets of the current descriptor block that consist of ordinary chain / fast rows

    // the window moves to new_off before row i is computed
    auto shift_to = [&](int i_in, int new_off) {
        int i = i_in, Vs = V;
        asm volatile("; window shift (rare): nothing of it is carried in the row loop" : "+s"(i), "+s"(Vs));
        RCN_LAP0();
        flush_edge();
        const int woff = cold_get(&cold->woff);
        const int delta = new_off - woff, dl = delta / LPC;
        {   // (b) dropped cells of the rows in the register window
            uint32_t ev = pack2(-32768, -32768);
#pragma unroll
            for (int k = 0; k < R * NP; ++k) ev = pk_max(ev, pk_subs(win[k], thrv[k % NP]));
            {
                uint32_t cand = pk_max(emaxV, ev);
                asm volatile("" : "+v"(cand));
                emaxV = lane < dl ? cand : emaxV;
            }
        }
        RCN_LAP(0, emaxV);
        {   // re-base the register window: lane l <- lane l + dl
            const int srcl = ((lane + dl) & 63) * 4;
            const bool keep = lane + dl < 64;
#pragma unroll
            for (int k = 0; k < R * NP; ++k) {
                const uint32_t v = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(srcl, static_cast<int>(win[k])));
                win[k] = keep ? v : NEGP;
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const uint32_t v = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(srcl, static_cast<int>(prev[q])));
                prev[q] = keep ? v : NEGP;
            }
        }
        RCN_LAP(1, prev[0]);
        // guards in HBM (score matrix only): the cell left of the window for every row still to come ...
        if (!CODE) for (int r = i + lane; r <= Vs; r += NTH) H16w[static_cast<int64_t>(r) * hs + new_off - 1] = static_cast<int16_t>(kNeg16);
        if (!CODE) {   // ... and the columns this shift adds, for the rows later rows can still name as predecessors
            const int r0 = max(1, i - (K + 1)), dw = delta >> 1, n = (i - r0) * dw;
            for (int idx = lane; idx < n; idx += NTH) {
                const int r = r0 + idx / dw, cw = idx % dw;
                H[static_cast<int64_t>(r) * hs2 + ((woff + WB) >> 1) + cw] = NEGP;
            }
        }
        {
            BandCold n_;
            n_.off2 = cold_get(&cold->off1); n_.s_row3 = cold_get(&cold->s_row2); n_.off1 = woff; n_.s_row2 = cold_get(&cold->s_row1); n_.s_row1 = i; n_.woff = new_off;
            *cold = n_;
        }
        RCN_LAP(2, 0);
        hrow = H + static_cast<int64_t>(i) * hs2 + (new_off >> 1);
        coff = static_cast<uint32_t>(i) * static_cast<uint32_t>(hs) + static_cast<uint32_t>(new_off) + 4u * static_cast<uint32_t>(t);
        set_columns();
        RCN_LAP(5, own_lane);
    };

#pragma unroll 1
    for (int rbase = 0; rbase < V && !bfail; rbase += 64) {
        if (HELP && rbase >= 128) {
            // the rows of this block overwrite ring rows rbase - 255 ...: the code waves must be through with block rbase - 128
            int spins = 0;
            for (;;) {
                const int d0 = static_cast<int>(__builtin_amdgcn_readfirstlane(lds_poll(help_done()))), d1 = static_cast<int>(__builtin_amdgcn_readfirstlane(lds_poll(help_done() + 1)));
                const int d2 = static_cast<int>(__builtin_amdgcn_readfirstlane(lds_poll(help_done() + 2)));
                if (min(min(d0, d1), d2) >= rbase - 128) break;
                if (++spins > kHelpSpin || __builtin_amdgcn_readfirstlane(lds_poll(help_done() + 3)) != 0u) { bfail |= 256; break; }
                __builtin_amdgcn_s_sleep(1);
            }
#ifdef RCN_PROF_WIN
            if (lane == 0) { atomicAdd(&g_whelp[0], 1ull); atomicAdd(&g_whelp[1], static_cast<unsigned long long>(spins)); }
#endif
            if (bfail) break;
        }
        {
            RowDesc d; d.erest = -1; d.meta = 1 << 9;
#pragma unroll
            for (int q = 0; q < kInlinePreds; ++q) d.p[q] = 0;
            int ro = 0;
            {
                // (loaded by every lane at a clamped index, then selected: a lane-dependent branch anywhere in the row
                //  loops makes the compiler turn ALL their control flow into flag-guarded blocks)
                const int rr = min(rbase + lane, V - 1);
                RowDesc dd_ = desc[rr]; int ro_ = roff[rr];
                asm volatile("" : "+v"(dd_.p[0]), "+v"(dd_.p[1]), "+v"(dd_.p[2]), "+v"(dd_.p[3]), "+v"(dd_.p[4]), "+v"(dd_.p[5]), "+v"(dd_.erest), "+v"(dd_.meta), "+v"(ro_));
                const bool in = rbase + lane < V;
#pragma unroll
                for (int q = 0; q < kInlinePreds; ++q) d.p[q] = in ? dd_.p[q] : d.p[q];
                d.erest = in ? dd_.erest : d.erest; d.meta = in ? dd_.meta : d.meta; ro = in ? ro_ : ro;
            }
            {
                // statistics off the row path: the predecessor counts of this block's chain / fast rows, summed once per
                // 64 rows (the other classes count theirs as they go)
                int npl = (rbase + lane < V && (d.meta & (1 << 13))) ? ((d.meta >> 9) & 7) : 0;
#pragma unroll
                for (int sh = 32; sh >= 1; sh >>= 1) npl += __shfl_xor(npl, sh);
                pred_rows += static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(npl));
            }
            {
                // chain / fast rows, this wave's copy of the descriptor word: predecessor DISTANCES (bits 16-31, four bits each) become
                // the predecessors' places in the register window -- row j lives in win[(j & (R - 1)) * NP ...] -- so that a row
                // reads a predecessor with the index it pulls out of the word (three scalar instructions less per predecessor).
                // HELP: a row with one predecessor names it twice: the octet rows below combine two predecessors without asking.
                const int il = rbase + lane + 1, npl_ = (d.meta >> 9) & 7;
                unsigned int f = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int de = static_cast<int>((static_cast<unsigned int>(d.meta) >> (16 + 4 * e)) & 15u);
                    const unsigned int fe = static_cast<unsigned int>(((il - de) & (R - 1)) * NP);
                    f |= (e < npl_ ? fe : 0u) << (4 * e);
                }
                if (HELP) f |= npl_ == 1 ? (f & 15u) << 4 : 0u;
                const int tm = static_cast<int>((static_cast<unsigned int>(d.meta) & 0xffffu) | (f << 16));
                d.meta = (d.meta & (1 << 13)) ? tm : d.meta;
            }
            // rows 8 g + 1 ... 8 g + 8 of this block are all ordinary chain / fast rows (bit 13 set, bit 12 clear): bit g
            fast8 = 0u;
            {
                const unsigned long long fm = __ballot(rbase + lane < V && (d.meta & ((1 << 13) | (1 << 12))) == (1 << 13));
#pragma unroll
                for (int g8 = 0; g8 < 8; ++g8) fast8 |= ((fm >> (8 * g8)) & 0xffull) == 0xffull ? (1u << g8) : 0u;
            }
            dl_p0 = d.p[0]; dl_p1 = d.p[1]; dl_p2 = d.p[2]; dl_p3 = d.p[3]; dl_p4 = d.p[4]; dl_p5 = d.p[5]; dl_er = d.erest; dl_meta = d.meta; dl_off = ro;
            asm volatile("; row descriptors retired" : "+v"(dl_p0), "+v"(dl_p1), "+v"(dl_p2), "+v"(dl_p3), "+v"(dl_p4), "+v"(dl_p5), "+v"(dl_er), "+v"(dl_meta), "+v"(dl_off));
        }
        const int rend = min(V, rbase + 64);
        int meta_next = __builtin_amdgcn_readlane(dl_meta, 0);
        uint32_t Pn[NP];
        auto profile_now = [&](int meta_) {
            if (TAB) {
                const lds_cu32 src = reinterpret_cast<lds_cu32>(ptab_l + (((meta_ & 255) >> 1) & 3) * (NTH * NP * 4));
#pragma unroll
                for (int q = 0; q < NP; ++q) Pn[q] = src[q];
            } else {
                const uint32_t sy = meta_ & 255, symsym = sy | (sy << 16);
#pragma unroll
                for (int q = 0; q < NP; ++q) Pn[q] = pk_profile(sqx[q], symsym, ONE, XM, MG);
            }
        };
        profile_now(meta_next);
#pragma unroll 1
        for (int i = rbase + 1; i <= rend; ++i) {
            const int k = (i - 1) & 63;
#ifndef RCN_NO_OCTETS
            if ((k & 7) == 0 && ((fast8 >> (k >> 3)) & 1u)) {
                // ---- an OCTET: eight ordinary chain / fast rows in one trip, straight-line, with STATIC register-window places.
                // A lone wave pays for every instruction it issues and ~70 clocks for a taken branch; the row-at-a-time loop
                // spends 4 instructions and a taken branch on its back-edge, 5 on the indexed window write, 3 on the class test,
                // 3 on the ring slot and 2 on row pointers -- per row.  Here row 8 g + 1 + o writes its result straight into
                // win[((o + 1) & (R - 1)) * NP ...], the ring row at a constant offset from the octet's first, and the rest once
                // per octet (84 % of the rows of cfg2 run through here).  Same arithmetic, same order: poa_band_row_octet.inc.
                static_assert(R == 8 && NP == 2, "octet = one turn of the register window");
#ifdef RCN_PROF_ROWS
                const long long oct_t0 = clock64();
#endif
#pragma unroll
                for (int q = 0; q < NP; ++q) win[q] = prev[q];                      // row i - 1 = 8 g: place 0
                uint32_t* const roct = ring + (slot * NTH + t) * NP;
#define RCN_OCT_ROW 0
#include "poa_band_row_octet.inc"
#undef RCN_OCT_ROW
#define RCN_OCT_ROW 1
#include "poa_band_row_octet.inc"
#undef RCN_OCT_ROW
#define RCN_OCT_ROW 2
#include "poa_band_row_octet.inc"
#undef RCN_OCT_ROW
#define RCN_OCT_ROW 3
#include "poa_band_row_octet.inc"
#undef RCN_OCT_ROW
#define RCN_OCT_ROW 4
#include "poa_band_row_octet.inc"
#undef RCN_OCT_ROW
#define RCN_OCT_ROW 5
#include "poa_band_row_octet.inc"
#undef RCN_OCT_ROW
#define RCN_OCT_ROW 6
#include "poa_band_row_octet.inc"
#undef RCN_OCT_ROW
#define RCN_OCT_ROW 7
#include "poa_band_row_octet.inc"
#undef RCN_OCT_ROW
#pragma unroll
                for (int q = 0; q < NP; ++q) prev[q] = win[q];                      // row 8 g + 8: place 0 again
#ifdef RCN_PROF_ROWS
                if (CODE && lane == 0) {
                    atomicAdd(&g_rowprof[blockIdx.x & 255][8], static_cast<unsigned long long>(clock64() - oct_t0));
                    atomicAdd(&g_rowprof[blockIdx.x & 255][10 + 8], 8ull);
                }
#endif
                slot = (slot + 8) & (K - 1);
                hrow += 8 * hs2;
                if (CODE && HELP) coff += 8u * static_cast<uint32_t>(hs);
                i += 7;
                continue;
            }
#endif
            const int meta = meta_next;
            // the row just finished enters the register window here, at ONE place: the copies of the row tail below then
            // only hand over `prev` (with the window written in each of them the compiler copies all sixteen registers
            // per row to reconcile the copies)
            {
                __builtin_amdgcn_sched_barrier(0);       // (the two indexed writes back to back: one register-index mode region, not two)
#pragma unroll
                for (int q = 0; q < NP; ++q) win[((i - 1) & (R - 1)) * NP + q] = prev[q];
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef RCN_PROF_ROWS
            const long long row_t0 = clock64();
            int row_cls = (meta & (1 << 15)) ? 0 : (meta & (1 << 13)) ? min(3, (meta >> 9) & 7) : ((meta & ((1 << 14) | (1 << 12))) == (1 << 14)) ? 4 : 5;
            if ((meta & ((1 << 13) | 256)) == 256) row_cls = 7;
#endif
            meta_next = __builtin_amdgcn_readlane(dl_meta, i & 63);
            uint32_t P[NP];

            uint32_t M[NP];
            uint32_t Aq[NP];                        // CODE: per cell, the first predecessor (in-edge order) that attains M
#pragma unroll
            for (int q = 0; q < NP; ++q) Aq[q] = 0u;
            // running "first argmax": predecessor number e replaces the holder where it is strictly greater
            auto arg_step = [&](const uint32_t (&zq)[NP], int e) {
                const uint32_t Q = pack2(e, e);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    // strictly greater <=> the maximum grows: the difference (mod 2^16, at most 63000) is non-zero
                    const uint32_t gt = pk_minu(pk_sub(pk_max(M[q], zq[q]), M[q]), ONE);
                    Aq[q] = pk_mad(gt, pk_sub(Q, Aq[q]), Aq[q]);
                }
            };
            // ONE test for the ordinary fast row (98 % of the rows: bit 13 set, bit 12 clear): a branch, taken or not, is what
            // a lone wave pays most for (a not-taken one ~15 clocks, a taken one ~70; plain scalar instructions next to
            // nothing: profiles/r03/r_row_sections.txt), so the special rows and the other classes share the first one and
            // sort themselves out behind it
            if (__builtin_expect((meta & ((1 << 13) | (1 << 12))) != (1 << 13), 0)) {
                if (meta & (1 << 12)) {
                    // special row: the window may move here (all on-chip state is re-based, the profile of this row redone)
                    const int new_off = __builtin_amdgcn_readlane(dl_off, k);
#ifdef RCN_PROF_ROWS
                    if (new_off != cold_get(&cold->woff)) row_cls = 6;
#endif
                    if (new_off != cold_get(&cold->woff)) { shift_to(i, new_off); profile_now(meta); }
                }
                if (!(meta & (1 << 13))) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) P[q] = Pn[q];
                    if ((meta & ((1 << 14) | (1 << 12))) == (1 << 14)) {
                        // ---- medium row whose predecessors all share this row's window: LDS ring, reads in flight together ----
                        const unsigned int dd = static_cast<unsigned int>(meta) >> 16;
                        const int npf = (meta >> 9) & 7;
                        uint32_t hp[4][NP];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int d = (dd >> (4 * (e < npf ? e : 0))) & 15;
                            int sp = slot - d; if (sp < 0) sp += K;
                            const uint32_t* src = ring + (sp * NTH + t) * NP;
#pragma unroll
                            for (int q = 0; q < NP; ++q) hp[e][q] = src[q];
                        }
                        if (CODE) {
#pragma unroll
                            for (int q = 0; q < NP; ++q) M[q] = hp[0][q];
#pragma unroll
                            for (int e = 1; e < 4; ++e) {           // unused slots repeat predecessor 0: never strictly greater
                                arg_step(hp[e], e);
#pragma unroll
                                for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], hp[e][q]);
                            }
                        } else {
#pragma unroll
                        for (int q = 0; q < NP; ++q) M[q] = pk_max(pk_max(hp[0][q], hp[1][q]), pk_max(hp[2][q], hp[3][q]));
                        }
                        pred_rows += npf;
                    } else {
                        // ---- general row: any number of predecessors from the LDS ring, each in the coordinates it was
                        //      written in (window offsets of the last two shifts are kept) ----
                        RCN_LAP0();
                        const int p0 = __builtin_amdgcn_readlane(dl_p0, k);
                        const int er = __builtin_amdgcn_readlane(dl_er, k);
                        // The common general row -- a sink, or a node with five or six in-edges: every predecessor is one of the six
                        // in the descriptor, a real row, still in the ring, written under this row's window offset (bit 12 clear: phase_desc2
                        // looked) -- takes its ring rows in ONE round trip, like a medium row: in situ the predecessor-by-predecessor
                        // loop below costs ~450 clocks per predecessor (a dependent LDS round trip and half a dozen scalar branches
                        // each: profiles/r06/z_rare_row_sections.txt), 1 % of the rows were 9-12 % of the DP's clocks.
                        bool wide_done = false;
                        if (!(meta & (1 << 12)) && er < 0 && ((meta >> 9) & 7) <= kInlinePreds) {
                            const int npw = (meta >> 9) & 7;
                            const int q1 = __builtin_amdgcn_readlane(dl_p1, k), q2 = __builtin_amdgcn_readlane(dl_p2, k);
                            const int q3 = __builtin_amdgcn_readlane(dl_p3, k), q4 = __builtin_amdgcn_readlane(dl_p4, k);
                            const int q5 = __builtin_amdgcn_readlane(dl_p5, k);
                            // unused slots repeat predecessor 0: never strictly greater, the first-argmax does not see them
                            const int pw[kInlinePreds] = {p0, npw > 1 ? q1 : p0, npw > 2 ? q2 : p0, npw > 3 ? q3 : p0, npw > 4 ? q4 : p0, npw > 5 ? q5 : p0};
                            const int pmin = min(min(min(pw[0], pw[1]), min(pw[2], pw[3])), min(pw[4], pw[5]));
                            if (pmin >= 1 && i - pmin < K - 1) {
                                auto ring_row = [&](int p_, uint32_t (&hq)[NP]) {
                                    int sp = slot - (i - p_); if (sp < 0) sp += K;
                                    const uint32_t* src = ring + (sp * NTH + t) * NP;
#pragma unroll
                                    for (int q = 0; q < NP; ++q) hq[q] = src[q];
                                };
                                uint32_t h0[NP], h1[NP];
                                ring_row(pw[0], h0); ring_row(pw[1], h1);
                                if (npw > 2) {
                                    uint32_t h2[NP], h3[NP], h4[NP], h5[NP];
                                    ring_row(pw[2], h2); ring_row(pw[3], h3); ring_row(pw[4], h4); ring_row(pw[5], h5);
#pragma unroll
                                    for (int q = 0; q < NP; ++q) M[q] = h0[q];
                                    if (CODE) arg_step(h1, 1);
#pragma unroll
                                    for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], h1[q]);
                                    if (CODE) arg_step(h2, 2);
#pragma unroll
                                    for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], h2[q]);
                                    if (CODE) arg_step(h3, 3);
#pragma unroll
                                    for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], h3[q]);
                                    if (CODE) arg_step(h4, 4);
#pragma unroll
                                    for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], h4[q]);
                                    if (CODE) arg_step(h5, 5);
#pragma unroll
                                    for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], h5[q]);
                                } else {
#pragma unroll
                                    for (int q = 0; q < NP; ++q) M[q] = h0[q];
                                    if (CODE) arg_step(h1, 1);
#pragma unroll
                                    for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], h1[q]);
                                }
                                pred_rows += npw;
                                wide_done = true;
                                RCN_LAP(14, M[0]);
                            }
                        }
                        if (!wide_done) {
                        const int woff = cold_get(&cold->woff), s_row1 = cold_get(&cold->s_row1), s_row2 = cold_get(&cold->s_row2), s_row3 = cold_get(&cold->s_row3);
                        const int off1 = cold_get(&cold->off1), off2 = cold_get(&cold->off2);
                        const int np = (meta >> 9) & 7;
                        RCN_LAP(6, woff + s_row1 + s_row2 + s_row3 + off1 + off2 + p0 + er);
                        bool first = true;
                        int nq = 0;                          // ordinal of the predecessor being combined (= its index in the descriptor)
                        auto combine = [&](int p) {
                            uint32_t hp[NP];
                            if (p == 0) {
                                if (woff > 0) bfail |= 1;                     // (c)
#pragma unroll
                                for (int q = 0; q < NP; ++q) hp[q] = 0u;
                            } else if (i - p < K - 1) {
                                int sp = slot - (i - p); if (sp < 0) sp += K;
                                int delta = 0;
                                if (p < s_row1) {
                                    if (p >= s_row2) delta = woff - off1;
                                    else if (p >= s_row3) delta = woff - off2;
                                    else { bfail |= 4; }
                                }
                                if (delta > 8 * kBandG) { bfail |= 16; delta = 0; }
                                const int dlp = delta / LPC;
                                if (dlp == 0) {
                                    const uint32_t* src = ring + (sp * NTH + t) * NP;
#pragma unroll
                                    for (int q = 0; q < NP; ++q) hp[q] = src[q];
                                } else {
                                    // (b) for a ring row: its dlp leftmost lanes fall off
                                    const uint32_t* old = ring + (sp * NTH + t) * NP;
                                    const uint32_t dthr = pack2(mg * delta, mg * delta);
                                    uint32_t ev = pack2(-32768, -32768);
#pragma unroll
                                    for (int q = 0; q < NP; ++q) {
                                        uint32_t ov = old[q];
                                        asm volatile("" : "+v"(ov));          // (loaded by every lane: no exec-masked region in the row loop)
                                        ev = pk_max(ev, pk_subs(ov, pk_sub(thrv[q], dthr)));
                                    }
                                    {
                                        uint32_t cand = pk_max(emaxV, ev);
                                        asm volatile("" : "+v"(cand));         // computed by every lane, then selected: no exec-masked region
                                        emaxV = lane < dlp ? cand : emaxV;
                                    }
                                    const uint32_t* src = ring + (sp * NTH + min(t + dlp, 63)) * NP;
                                    const bool keep = t + dlp < 64;
#pragma unroll
                                    for (int q = 0; q < NP; ++q) { uint32_t v = src[q]; asm volatile("" : "+v"(v)); hp[q] = keep ? v : NEGP; }
                                }
                            } else {
                                bfail |= 2;                                    // (d)
#pragma unroll
                                for (int q = 0; q < NP; ++q) hp[q] = NEGP;
                            }
                            if (first) {
#pragma unroll
                                for (int q = 0; q < NP; ++q) M[q] = hp[q];
                                first = false;
                            } else {
                                if (CODE) arg_step(hp, nq);
#pragma unroll
                                for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], hp[q]);
                            }
                            ++pred_rows; ++nq;
                        };
                        combine(p0);
                        RCN_LAP(7, M[0]);
                        if (np > 1) {
                            const int q1 = __builtin_amdgcn_readlane(dl_p1, k), q2 = __builtin_amdgcn_readlane(dl_p2, k);
                            const int q3 = __builtin_amdgcn_readlane(dl_p3, k), q4 = __builtin_amdgcn_readlane(dl_p4, k);
                            const int q5 = __builtin_amdgcn_readlane(dl_p5, k);
#pragma unroll 1
                            for (int q = 1; q < np; ++q) combine(q == 1 ? q1 : q == 2 ? q2 : q == 3 ? q3 : q == 4 ? q4 : q5);
                        }
                        RCN_LAP(8, M[0]);
                        for (int e = er; e >= 0; e = e_nin[e]) {          // more than six in-edges: the rest of the list
                            const int tl = e_tail[e];
                            if (sub && !inc[tl]) continue;
                            combine(nr[tl] + 1);
                        }
                        if (CODE && nq > 8) bfail |= 8;                   // move codes name predecessors 0..7 (three bits)
                        RCN_LAP(9, M[0]);
                        }
#pragma unroll
                        for (int q = 0; q < NP; ++q) asm volatile("" : "+v"(M[q]));
                    }

                    RCN_LAP0();
                    {
#define RCN_TAIL_MULTI 2
#define RCN_TAIL_SINK 1
#include "poa_band_row_tail.inc"
#undef RCN_TAIL_MULTI
#undef RCN_TAIL_SINK
                    }
                    continue;
                }
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) P[q] = Pn[q];
            {
                // ---- chain and fast rows (98 % of the rows): every predecessor is in the register window (always in current
                //      coordinates).  The first predecessor is one indexed register read whatever its distance (a chain row
                //      has distance 1), a second one follows in line, only a third / fourth loop; rows with one predecessor
                //      and rows with several each run through their own copy of the row tail ----
                const unsigned int dd = static_cast<unsigned int>(meta) >> 16;
                const int npf = (meta >> 9) & 7;
                {
                    const int wi = static_cast<int>(dd & 15);
#pragma unroll
                    for (int q = 0; q < NP; ++q) M[q] = win[wi + q];
                }
                if (__builtin_expect(npf > 1, 0)) {      // (38 % of the rows; kept off the fall-through path of the other 60 %)
                    {
                        const int wi = static_cast<int>((dd >> 4) & 15);
                        uint32_t zq[NP];
                        __builtin_amdgcn_sched_barrier(0);   // (both indexed reads in one register-index mode region)
#pragma unroll
                        for (int q = 0; q < NP; ++q) zq[q] = win[wi + q];
                        __builtin_amdgcn_sched_barrier(0);
                        if (CODE && !HELP) arg_step(zq, 1);
#pragma unroll
                        for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], zq[q]);
                    }
#pragma unroll 1
                    for (int e = 2; e < npf; ++e) {
                        const int wi = static_cast<int>((dd >> (4 * e)) & 15);
                        uint32_t zq[NP];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < NP; ++q) zq[q] = win[wi + q];
                        __builtin_amdgcn_sched_barrier(0);
                        if (CODE && !HELP) arg_step(zq, e);
#pragma unroll
                        for (int q = 0; q < NP; ++q) M[q] = pk_max(M[q], zq[q]);
                    }
                    {
#define RCN_TAIL_MULTI 1
#define RCN_TAIL_SINK 0
#include "poa_band_row_tail.inc"
#undef RCN_TAIL_MULTI
#undef RCN_TAIL_SINK
                    }
                    continue;
                }
                {
#define RCN_TAIL_MULTI 0
#define RCN_TAIL_SINK 0
#include "poa_band_row_tail.inc"
#undef RCN_TAIL_MULTI
#undef RCN_TAIL_SINK
                }
                continue;
            }
        }
    }
    if (HELP && bfail) lds_store(help_prog() + t, 0xffffffffu);       // left early: the code waves must not wait for the rest
    flush_edge();
    // ---- certificate: no recorded cell may be alive at T = the best end score found ----
    int ev = max(static_cast<int>(emaxV) >> 16, static_cast<int>(emaxV << 16) >> 16);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) ev = max(ev, __shfl_xor(ev, d));
    ev = __builtin_amdgcn_readfirstlane(ev);
    const int lim = best + len * c.gp - c.m * len;          // T - m len, with T = Z + len g at column len
    // why (statistics): 1 source row off the left edge, 2 predecessor older than the ring, 4 / 16 more than two window shifts
    // inside the ring, 8 more than six in-edges, 32 no end cell, 64 an alive last window cell, 128 an alive dropped cell
    // (256: the code wave fell too far behind -- not counted by reason)
    const int why = bfail | (!have_best ? 32 : 0) | ((have_best && edgeS >= lim) ? 64 : 0) | ((have_best && ev >= lim) ? 128 : 0);
    const int fail = (why || c.tie_pad[0] == 2) ? 1 : 0;
    Ctx* o = Block4::ctx();
    if (lane == 0) {
        o->best = best; o->best_row = best_row; o->tied = tied; o->band_fail = fail;
        if (!fail) {
            const int W = len + 1;
            o->pred_rows = pred_rows;
            const unsigned long long wcols = static_cast<unsigned long long>(W < WB ? W : WB);
            o->cells += static_cast<unsigned long long>(V + 1) * wcols;
            o->pred += static_cast<unsigned long long>(pred_rows) * wcols;
            o->cells_full += static_cast<unsigned long long>(V + 1) * W;
            const int amax = max(max(abs(c.m), abs(c.x)), abs(c.gp));
            const unsigned long long sbytes = (static_cast<long long>(amax) * (V + W) < 32767) ? 2ull : 4ull;
            o->bytes += sbytes * (static_cast<unsigned long long>(V + 1) + pred_rows) * wcols;
            o->bytes_full += sbytes * (static_cast<unsigned long long>(V + 1) + pred_rows) * W;
            o->n_banded += 1;
        } else {
            o->n_band_fail += 1;
            o->band_why |= why;
            for (int k = 0; k < 8; ++k) if (why & (1 << k)) o->band_whyn[k] += 1;
        }
    }
    Wave0Of4::sync();
}

// The instances as functions of their own.  Those with code waves (HELP) only exist in the kernel instance that has a CU to
// itself (poa_window_kernel2_deep, compiled in a translation unit of its own -- engine_deep.hip -- at two waves per SIMD:
// none of its functions is held to the 64 VGPRs that eight work-groups per CU leave a wave).
template <int NP, bool TAB, bool CODE = false>
__device__ __noinline__ void dp2_rows_band() { dp2_rows_band_body<NP, TAB, CODE, false>(); }
template <bool TAB>
__device__ __noinline__ void dp2_rows_band_help() { dp2_rows_band_body<2, TAB, true, true>(); }

// ---- a code wave (waves 1-3 of the work-group, next to dp2_rows_band<NP, TAB, true, true> on wave 0) ----
// For every chain / fast row, in row order and as soon as wave 0 has counted it: the row and its predecessor rows from the
// large ring (a predecessor written under an older window offset is read dl lanes further right, -inf beyond lane 63: what
// the re-based register window of wave 0 holds), the combined predecessor row M with its first-argmax, the diagonal and
// vertical candidates, and the code word -- the arithmetic of the row tail (poa_band_row_tail.inc), on values that are
// bit-identical to the ones wave 0 worked with.
template <int NP, bool TAB>
__device__ __noinline__ void dp2_band_codes(const int hw) {        // hw = 0, 1, 2: this wave takes the rows i with i % 3 == hw
    static_assert(NP == 2, "move codes: four cells per lane -> one dword per lane and row");
    constexpr int NTH = 64, LPC = 2 * NP, K = kHelpRows;
    constexpr int kTab = TAB ? 4 * 4 * NTH * NP : 0;
    constexpr int KT = (kLdsBytes - 64 - kBandSeq - kTab) / (4 * NTH * NP);
    const int t = threadIdx.x & 63;
    const Ctx c = ctx_load<Block4>();
    Win g = ctx_win(c);
    RCN_G const RowDesc* desc = g.desc.ptr();
    RCN_G const int32_t* roff = g.pred.ptr();
    RCN_G uint8_t* cbase = reinterpret_cast<RCN_G uint8_t*>(g.H.ptr());
    const int V = c.V, len = c.len, hs = c.hstride;
    const uint32_t* ring = help_ring();
    const uint8_t* lseq = reinterpret_cast<const uint8_t*>(reinterpret_cast<uint32_t*>(Block4::work()) + KT * NTH * NP + kTab / 4);
    const uint32_t* prog = help_prog() + t;
    const int mg = c.m - c.gp, xg = c.x - c.gp;
    const uint32_t MG = pack2(mg, mg), XM = pack2(xg - mg, xg - mg), ONE = 0x00010001u;
    const uint32_t GG = pack2(c.gp, c.gp), NEGP = pack2(kNeg16, kNeg16);
    uint32_t sqx[NP];
    auto set_columns = [&](int woff) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int j0 = woff + t * LPC + 2 * q, j1 = j0 + 1;
            const int l0 = lseq[min(max(j0 - 1, 0), kBandSeq - 1)], l1 = lseq[min(max(j1 - 1, 0), kBandSeq - 1)];
            const int s0 = (j0 >= 1 && j0 <= len) ? l0 : 0x100, s1 = (j1 >= 1 && j1 <= len) ? l1 : 0x100;
            sqx[q] = pack2(s0, s1);
        }
    };
    int have = 0;                               // rows wave 0 is known to have finished
    bool gone = false;                          // wave 0 left early, or this wave waited too long
    auto wait_for = [&](int i) {
        int spins = 0;
        while (have < i) {
            const uint32_t v = __builtin_amdgcn_readfirstlane(lds_poll(prog));
            if (v == 0xffffffffu) { gone = true; return; }
            have = static_cast<int>(v & 0xffffu);
            if (have >= i) break;
            if (++spins > kHelpSpin) { gone = true; lds_store(help_done() + 3, 1u); return; }
            __builtin_amdgcn_s_sleep(RCN_HELP_SLEEP);
        }
#ifdef RCN_PROF_WIN
        if (t == 0) { atomicAdd(&g_whelp[2], 1ull); atomicAdd(&g_whelp[3], static_cast<unsigned long long>(spins)); }
#endif
    };
    wait_for(1);                                // (the layer's bases are in LDS before wave 0 counts its first row)
    int woff = 0;                               // the window offset sqx[] was made for
    if (!gone) set_columns(0);
    // A row's window offset is roff[row - 1]: wave 0 moves its window exactly where that value changes (phase_desc2 marks
    // those rows).  64 offsets and descriptor words per block, one per lane; the previous block's offsets for predecessors
    // across the block edge.
    int dl_off_prev = 0;
    static_assert((K & (K - 1)) == 0, "ring slots by mask");
#pragma unroll 1
    for (int rbase = 0; rbase < V && !gone; rbase += 64) {
        int dl_meta = 1 << 9, dl_off = 0;
        if (rbase + t < V) { dl_meta = desc[rbase + t].meta; dl_off = roff[rbase + t]; }
        const int nrow = min(V, rbase + 64) - rbase;
        int k0 = hw - rbase % 3; if (k0 < 0) k0 += 3;                // rows i = rbase + k + 1 with (i - 1) % 3 == hw
#pragma unroll 1
        for (int k = k0; k < nrow; k += 3) {
            const int i = rbase + k + 1;
            const int meta = __builtin_amdgcn_readlane(dl_meta, k);
            if (!(meta & (1 << 13))) continue;                          // wave 0 keeps the codes of this row
            if (have < i) { wait_for(i); if (gone) break; }
#ifdef RCN_EXP_CODEWAVE_NOP
            continue;                                                   // (timing experiment: what wave 0 does when nobody holds it back; results are wrong)
#endif
            const int wi = __builtin_amdgcn_readlane(dl_off, k);
            if (wi != woff) { woff = wi; set_columns(woff); }
            const unsigned int dd = static_cast<unsigned int>(meta) >> 16;
            const int npf = (meta >> 9) & 7;
            const int slot = (i - 1) & (K - 1);
            uint32_t acc[NP], M[NP], Aq[NP];
            {
                const uint32_t* src = ring + (slot * NTH + t) * NP;
#pragma unroll
                for (int q = 0; q < NP; ++q) { acc[q] = src[q]; Aq[q] = 0u; }
            }
            // predecessor e: distance d rows up, read dlp lanes further right when it was written under an older offset
            auto pred_row = [&](int e, uint32_t (&zq)[NP]) {
                const int d = static_cast<int>((dd >> (4 * e)) & 15);
                const int kp = k - d;
                const int wp = kp >= 0 ? __builtin_amdgcn_readlane(dl_off, kp & 63) : __builtin_amdgcn_readlane(dl_off_prev, (kp + 64) & 63);
                const uint32_t* base = ring + (((slot - d) & (K - 1)) * NTH) * NP;
                if (__builtin_expect(wp == wi, 1)) {
                    const uint32_t* src = base + t * NP;
#pragma unroll
                    for (int q = 0; q < NP; ++q) zq[q] = src[q];
                } else {
                    const int dlp = (wi - wp) / LPC;
                    const bool keep = t + dlp < 64;
                    const uint32_t* src = base + min(t + dlp, 63) * NP;
#pragma unroll
                    for (int q = 0; q < NP; ++q) { const uint32_t v = src[q]; zq[q] = keep ? v : NEGP; }
                }
            };
            pred_row(0, M);
            for (int e = 1; e < npf; ++e) {
                uint32_t zq[NP];
                pred_row(e, zq);
                const uint32_t Q = pack2(e, e);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const uint32_t gt = pk_minu(pk_sub(pk_max(M[q], zq[q]), M[q]), ONE);
                    Aq[q] = pk_mad(gt, pk_sub(Q, Aq[q]), Aq[q]);
                    M[q] = pk_max(M[q], zq[q]);
                }
            }
            // (lane 0: -inf, the cell left of the window)
            const uint32_t mprev = __builtin_amdgcn_update_dpp(static_cast<uint32_t>(kNeg16) << 16, M[NP - 1], 0x138, 0xf, 0xf, false);
            const uint32_t sy = meta & 255, symsym = sy | (sy << 16);
            uint32_t nd[NP], nu[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const uint32_t D = __builtin_amdgcn_alignbit(M[q], q == 0 ? mprev : M[q - 1], 16);
                const uint32_t DPv = pk_add(D, pk_profile(sqx[q], symsym, ONE, XM, MG)), Uv = pk_add(M[q], GG);
                nd[q] = pk_minu(pk_sub(acc[q], DPv), ONE); nu[q] = pk_minu(pk_sub(acc[q], Uv), ONE);
            }
            const uint32_t bnd = __builtin_amdgcn_perm(nd[1], nd[0], 0x06040200u), bnu = __builtin_amdgcn_perm(nu[1], nu[0], 0x06040200u);
            uint32_t word = (bnu << 1) | bnd;
            if (npf > 1) {
                const uint32_t bA = __builtin_amdgcn_perm(Aq[1], Aq[0], 0x06040200u);
                const uint32_t bAl = __builtin_amdgcn_update_dpp(0u, bA, 0x138, 0xf, 0xf, true);
                const uint32_t bAsh = __builtin_amdgcn_alignbit(bA, bAl, 24);
                word = (bA << 5) | word;
                word = (bAsh << 2) | word;
            }
            __builtin_nontemporal_store(word, reinterpret_cast<RCN_G uint32_t*>(cbase + (static_cast<uint32_t>(i) * static_cast<uint32_t>(hs) + static_cast<uint32_t>(wi) + 4u * static_cast<uint32_t>(t))));
        }
        dl_off_prev = dl_off;
        const int rend = rbase + nrow;
        if (!gone) lds_store(help_done() + hw, static_cast<uint32_t>(rend));
    }
}

}  // namespace rcn
