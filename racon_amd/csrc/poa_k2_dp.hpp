// poa_k2_dp.hpp -- phase: sequence-to-graph NW DP on full rows (one wave, or the four-wave pipeline over column blocks)
// Part of the fast path of the MI355X window-consensus engine: included by poa_kernel2.hpp, in this order, into one
// translation unit (see its header for the design).
#pragma once

namespace rcn {

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
