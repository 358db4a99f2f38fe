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

// The phases, in the order a layer runs through them (one translation unit; every file opens namespace rcn itself):
#include "poa_k2_common.hpp"       // execution policies, packed int16 helpers, DP shapes
#include "poa_k2_subgraph.hpp"     // Subgraph mask + filtered order
#include "poa_k2_desc.hpp"         // row descriptors
#include "poa_k2_dp.hpp"           // NW DP on full rows
#include "poa_band.hpp"            // exact banded DP + move codes (poa_band_row_tail.inc)
#include "poa_k2_add.hpp"          // AddAlignment, order merge
#include "poa_k2_sinktie.hpp"      // sink tie-break
#include "poa_k2_traceback.hpp"    // traceback over scores / over move codes
#include "poa_k2_toposort.hpp"     // spoa's exact DFS order, in parallel
#include "poa_k2_consensus.hpp"    // heaviest bundle, branch completion, trim

namespace rcn {

// The body of the kernel.  DEEP: the instance for work-groups that have a CU and its LDS to themselves (the deep launch of
// engine.hip's split) -- the only one that runs the banded DP with code waves, so that those functions have one caller,
// launched at one work-group per CU, and are not held to the 64 VGPRs of eight work-groups per CU.
template <bool DEEP>
__device__ __forceinline__ void poa_window_body(const KParams& P) {
    const int t = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // sub, desc, dp, traceback, add, merge, consensus, other
    long long tck = clock64();
#define RCN_PHASE2(k) do { long long now__ = clock64(); ph[k] += now__ - tck; tck = now__; } while (0)
    Ctx* ctx = ctx_lds2();
    if (t == 0) {
        ctx->scratch = P.scratch + static_cast<uint64_t>(blockIdx.x) * P.slot_bytes;
        ctx->ncap = P.ncap; ctx->ecap = P.ecap; ctx->ring = P.ring; ctx->lmax = P.lmax; ctx->hstride = P.hstride; ctx->hrows = P.hrows;
        ctx->big = (DEEP && P.lds_extra >= kHelpLdsBytes && !P.no_help) ? 1 : 0;
        ctx->m = P.m; ctx->x = P.x; ctx->gp = P.g; ctx->trim = P.trim;
        ctx->cells = 0; ctx->pred = 0; ctx->bytes = 0; ctx->ties = 0;
        ctx->cells_full = 0; ctx->bytes_full = 0; ctx->n_banded = 0; ctx->n_band_fail = 0; ctx->band = 0; ctx->band_fail = 0;
        ctx->band_why = 0; for (int k = 0; k < 8; ++k) ctx->band_whyn[k] = 0;
        ctx->n_help = 0;
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
                const bool coded = P.band != 3;
                const bool help = DEEP && coded && bcast0(ctx->big) != 0;            // a CU's LDS to itself: waves 1-3 assemble the move codes
                if (help) {
                    if (t < 64) help_prog()[t] = 0u;
                    if (t < 4) help_done()[t] = 0u;
                    Block4::sync();
                }
                if (wv == 0) {
                    const bool tab = bcast0(ctx->tie_pad[1]) != 0;
                    if (DEEP && help) { if constexpr (DEEP) { if (tab) dp2_rows_band_help<true>(); else dp2_rows_band_help<false>(); } }
                    else if (coded) { if (tab) dp2_rows_band<2, true, true>(); else dp2_rows_band<2, false, true>(); }
                    else { if (tab) dp2_rows_band<2, true>(); else dp2_rows_band<2, false>(); }
                } else if (DEEP && help) {
                    if constexpr (DEEP) { if (bcast0(ctx->tie_pad[1]) != 0) dp2_band_codes<2, true>(wv - 1); else dp2_band_codes<2, false>(wv - 1); }
                }
                Block4::sync();
                if (help && t == 0 && help_done()[3] != 0u && ctx->band_fail == 0) { ctx->band_fail = 1; ctx->n_band_fail += 1; }   // (the code wave gave up)
                if (help) { if (t == 0 && ctx->band_fail == 0) ctx->n_help += 1; Block4::sync(); }
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
#ifdef RCN_PROF_WIN
            const unsigned long long dp0__ = ph[2];
#endif
            RCN_PHASE2(2);
#ifdef RCN_PROF_WIN
            const unsigned long long dpc__ = ph[2] - dp0__;
            const long long tl0__ = clock64();
            const int tl_tied__ = bcast0(ctx->tied), tl_bf__ = bcast0(ctx->band_fail) | (bcast0(ctx->coded) ? 0 : 2);
            int tl_lvl__ = 0, tl_route__ = 0;
#endif
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
#ifdef RCN_PROF_WIN
                    tl_route__ = 1;
#endif
                    if (wv == 0) phase_sink_tie_starts();
                    Block4::sync();
                    st = bcast0(ctx->tb_n);
#ifdef RCN_PROF_WIN
                    if (st == 3) tl_route__ = 2;
#endif
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
                            phase_desc2(false);                      // (after the DP: the matrix -- scores or move codes -- stays as it is)
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
#ifdef RCN_PROF_WIN
                tl_lvl__ = 1 + st;
                if (t == 0) {
                    // all windows: tie events, clocks, events that went past the rule / into the closure sweep / into the full DFS
                    atomicAdd(&g_wtie[0], 1ull); atomicAdd(&g_wtie[1], static_cast<unsigned long long>(clock64() - tl0__));
                    if (tl_route__ >= 1) atomicAdd(&g_wtie[2], 1ull);
                    if (tl_route__ >= 2) atomicAdd(&g_wtie[3], 1ull);
                    if (st == 2) atomicAdd(&g_wtie[4], 1ull);
                }
#endif
            }
#ifdef RCN_PROF_WIN
            const long long tl1__ = clock64();
#endif
            if (bcast0(ctx->coded)) phase_traceback_code(); else phase_traceback3();
#ifdef RCN_PROF_WIN
            if (t == 0 && wi < 4 && P.work_base == 0 && jl < 128) {
                g_wlay[wi][jl][0] = dpc__;
                g_wlay[wi][jl][1] = static_cast<unsigned long long>(tl1__ - tl0__); g_wlay[wi][jl][2] = static_cast<unsigned long long>(clock64() - tl1__);
                g_wlay[wi][jl][3] = (static_cast<unsigned long long>(tl_tied__) << 16) | ((tl_lvl__ + 16 * tl_route__) << 8) | tl_bf__;
                g_wlay[wi][jl][4] = (static_cast<unsigned long long>(bcast0(ctx->V)) << 32) | static_cast<unsigned int>(len);
            }
#endif
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
        if (ctx->n_help) atomicAdd(&P.stats[24], static_cast<unsigned long long>(ctx->n_help));
    }
}

// Two instances, two translation units (functions compiled for a kernel that runs eight work-groups per CU are held to 64
// VGPRs, and the compiler gives a function ONE register budget, its most generous caller's): engine.hip has the kernel
// for eight work-groups per CU, engine_deep.hip (RCN_DEEP_TU) the one for a work-group that has a CU to itself, two waves
// per SIMD at most.  RCN_ONE_TU (profiling builds, whose device-side counters must exist once): both here, same bounds.
#if defined(RCN_SMALL_TU)
// (engine_small.hip: the helpers of this file, none of its kernels)
#elif defined(RCN_DEEP_TU)
__global__ __launch_bounds__(kThreads2, 2) void poa_window_kernel2_deep(KParams P) { poa_window_body<true>(P); }
#else
__global__ __launch_bounds__(kThreads2, 8) void poa_window_kernel2(KParams P) { poa_window_body<false>(P); }
#ifdef RCN_ONE_TU
__global__ __launch_bounds__(kThreads2, 8) void poa_window_kernel2_deep(KParams P) { poa_window_body<true>(P); }
#else
__global__ void poa_window_kernel2_deep(KParams P);
#endif
#endif

}  // namespace rcn
