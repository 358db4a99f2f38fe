// poa_k2_add.hpp -- traceback tile constants; phases: AddAlignment and order merge over 256 threads (reference src/window.cpp:110-119)
// Part of the fast path of the MI355X window-consensus engine: included by poa_kernel2.hpp, in this order, into one
// translation unit (see its header for the design).
#pragma once

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

}  // namespace rcn
