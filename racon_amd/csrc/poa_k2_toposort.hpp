// poa_k2_toposort.hpp -- phase: spoa's exact DFS topological order, in parallel
// Part of the fast path of the MI355X window-consensus engine: included by poa_kernel2.hpp, in this order, into one
// translation unit (see its header for the design).
#pragma once

namespace rcn {

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

}  // namespace rcn
