// poa_k2_sinktie.hpp -- phase: several sinks share the best score -- spoa's end cell without its full DFS
// Part of the fast path of the MI355X window-consensus engine: included by poa_kernel2.hpp, in this order, into one
// translation unit (see its header for the design).
#pragma once

namespace rcn {

// ---- phase: several sinks share the best score (window.cpp:95-97 -> spoa's end cell) ----
// spoa takes the first of them in ITS rank order, the exact DFS post-order of Graph::TopologicalSort whose
// start nodes go in id order.  Three levels, cheapest first:
//  (1) rule: the DFS runs the backbone ids 0..L-1 first, so everything in the "backbone closure" (ancestors of
//      backbone nodes and their aligned rings) is appended before anything else, ring of backbone node b_p at
//      start p as (b_p, aligned list of b_p = ascending id).  A sink without aligned nodes and id >= L is in
//      nobody's closure: it is appended exactly when the start loop reaches its own id.  Hence the key
//      (p, id) for sinks whose ring holds a backbone node, (inf, id) for lone non-backbone sinks, and
//      (inf, smallest id of the ring, position behind it) for rings of non-backbone nodes without out-edges.
//      Rings, out-edges and starts are those of the graph being aligned: for a Subgraph alignment only what `inc` marks.
//      Restated on the CPU and compared with the DFS at every tie of the CPU test sets: tests/emul/emul_main.cpp (rc -8).
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
            // the ring AS THE (SUB)GRAPH HAS IT: spoa's DFS over a subgraph neither starts at nor visits the nodes outside it, so a
            // ring member outside -- a backbone node b_p left of `begin` whose aligned alternatives were pulled in through
            // their own successors -- appends nothing and orders nothing.  (Taken over all members the key was (p, id) of a
            // start that never happens: one window in 666 600 of tools/fuzz_sweep.py, seed 5088, found this.)
            int rm = v, na_in = 0;
            for (int a = 0; a < na; ++a) {
                const int u = g.al_nodes[v * g.ring + a];
                if (sub && !g.inc[u]) continue;
                ++na_in; rm = min(rm, u);
            }
            long long key;
            if (rm < c.bblen) key = (static_cast<long long>(rm) << 32) | static_cast<unsigned int>(v);
            else if (v >= (1 << 25)) { classified = false; break; }
            else if (na_in == 0) key = (0x7ffffffell << 32) | (static_cast<unsigned int>(v) << 6);
            else {
                // a ring of non-backbone nodes none of which has an out-edge (alternative last bases past the backbone's
                // end: every later layer that stops short of them ties there): nothing is appended because of them and
                // nobody's DFS reaches them, so the start loop finds the ring at its smallest id m and appends m, then
                // m's aligned list in list order -> key (m, position).  (Measured on cfg2: 240 of 241 ties the rule
                // above left open, ~440 k clocks each through the closure sweep, eleven layers in a row of the window
                // that ended the launch.)
                bool closed = true;
                for (int a = -1; a < na && closed; ++a) {
                    const int u = a < 0 ? v : g.al_nodes[v * g.ring + a];
                    if (sub && !g.inc[u]) continue;
                    for (int e = g.out_head[u]; e >= 0 && closed; e = g.e_nout[e]) if (!sub || g.inc[g.e_head[e]]) closed = false;
                }
                if (!closed) { classified = false; break; }
                int pos = 0;
                if (v != rm) {
                    if (sub && !g.inc[rm]) { classified = false; break; }      // (a non-backbone ring is inside the subgraph as a whole)
                    const int nm = g.al_cnt[rm];
                    pos = -1;
                    for (int a = 0, q = 0; a < nm; ++a) {
                        const int u = g.al_nodes[rm * g.ring + a];
                        if (sub && !g.inc[u]) continue;
                        ++q;
                        if (u == v) { pos = q; break; }
                    }
                    if (pos < 0 || pos >= 64) { classified = false; break; }
                }
                key = (0x7ffffffell << 32) | (static_cast<unsigned int>(rm) << 6) | static_cast<unsigned int>(pos);
            }
            // (c.tie_pad[2] bit 1: RCN_PLANT_FAULT=1, the planted wrong rule for the tests of rcn_engine_verify -- the LAST key wins)
            if ((c.tie_pad[2] & 2) ? (pick < 0 || key > bestkey) : key < bestkey) { bestkey = key; pick = v; }
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
#ifdef RCN_PROF_TIE
        for (int k = 0; k < c.tied; ++k) printf("[tie-starts] V %d sink row %d node %d p %d\n", c.V, (k == 0 ? c.best_row : o->tie_rows[k]), vs[k], ps[k]);
        printf("[tie-starts] V %d pstar %d cnt %d -> status %d\n", c.V, pstar, cnt, status);
#endif
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
#ifdef RCN_PROF_TIE
    printf("[tie-local] V %d pstar %d ncand %d cands %d %d %d -> winner node %d row %d\n", c.V, pstar, ncand, ncand > 0 ? cand[0] : -1, ncand > 1 ? cand[1] : -1, ncand > 2 ? cand[2] : -1, winner, winner >= 0 ? nr[winner] + 1 : -1);
#endif
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
#ifdef RCN_PROF_TIE
        if ((g.desc[row - 1].meta & 256) && zend == c.best) printf("[tie-full] V %d tied sink row %d node %d at DFS rank %d\n", c.V, row, g.rank_x[r], r);
#endif
        if ((g.desc[row - 1].meta & 256) && zend == c.best) { o->best_row = row; break; }
    }
    o->ties += 1;
#ifdef RCN_PROF_WIN
    printf("[tie3] why %d tied %d sub %d n %d V %d pstar %d\n", o->tie_why, c.tied, c.sub, g.n_nodes, c.V, c.tb_i);
#endif
}

}  // namespace rcn
