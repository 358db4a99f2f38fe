// pair_cell_main.cpp — TEST HARNESS (CPU): racon_amd/csrc/pair_cell.hpp, the bit-vector cell of the device pairwise aligner in the
// form it is issued on gfx950 (three-input bit operations, complemented plus-word, carries in bit 31), against the textbook 64-bit
// form of the same recurrence (the one racon_amd/host/nw_path.cpp and the previous kernel use), word by word and column by column
// over stacked words, and the column scores against a plain edit-distance DP.  Built and run by tests/test_pair_align_oracle.py.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../racon_amd/csrc/pair_cell.hpp"

using namespace rcn;

struct RefWord { uint64_t Pv = ~0ull, Mv = 0ull; };
// textbook cell: Eq of the word for this column, carry in / out as (+1, -1) bits
static void ref_cell(RefWord& w, uint64_t Eq, int hin_p, int hin_n, int& hout_p, int& hout_n, uint64_t& Ph0) {
    const uint64_t Xv = Eq | w.Mv;
    Eq |= static_cast<uint64_t>(hin_n);
    const uint64_t Xh = (((Eq & w.Pv) + w.Pv) ^ w.Pv) | Eq;
    uint64_t Ph = w.Mv | ~(Xh | w.Pv);
    uint64_t Mh = w.Pv & Xh;
    hout_p = static_cast<int>(Ph >> 63); hout_n = static_cast<int>(Mh >> 63);
    Ph0 = Ph;
    Ph = (Ph << 1) | static_cast<uint64_t>(hin_p); Mh = (Mh << 1) | static_cast<uint64_t>(hin_n);
    w.Pv = Mh | ~(Xv | Ph);
    w.Mv = Ph & Xv;
}

template <int NPL>
static int run(std::mt19937_64& rng, int m, int n, int nsym, double sim) {
    std::vector<int> q(m), t(n);
    for (int& c : q) c = static_cast<int>(rng() % nsym);
    for (int j = 0; j < n; ++j) t[j] = (j < m && (rng() % 1000) < sim * 1000) ? q[j] : static_cast<int>(rng() % (nsym + 1));   // (nsym: a symbol the query lacks)
    const int nb = (m + 63) / 64;
    std::vector<PairLane<NPL>> L(nb);
    std::vector<RefWord> R(nb);
    std::vector<uint64_t> valid(nb);
    for (int w = 0; w < nb; ++w) {
        PairLane<NPL>& l = L[w];
        for (int k = 0; k < NPL; ++k) l.pl[k] = l.ph[k] = 0;
        uint64_t v = 0;
        for (int r = 0; r < 64 && w * 64 + r < m; ++r) {
            v |= 1ull << r;
            for (int k = 0; k < NPL; ++k) if ((q[w * 64 + r] >> k) & 1) { if (r < 32) l.pl[k] |= 1u << r; else l.ph[k] |= 1u << (r - 32); }
        }
        valid[w] = v; l.vl = static_cast<uint32_t>(v); l.vh = static_cast<uint32_t>(v >> 32);
        l.Pvl = l.Pvh = ~0u; l.Mvl = l.Mvh = 0u;
    }
    // plain DP column by column for the scores
    std::vector<int> col(m + 1), nxt(m + 1);
    for (int i = 0; i <= m; ++i) col[i] = i;
    for (int j = 0; j < n; ++j) {
        const int code = t[j] < nsym ? t[j] : ((1 << NPL) - 1);          // a symbol outside the query's: the all-ones code no row has (nsym < 2^NPL)
        PairCarry c{0u, 0u};                                              // top boundary: +1
        int hp = 1, hn = 0;
        for (int w = 0; w < nb; ++w) {
            uint64_t Eq = 0;
            for (int r = 0; r < 64 && w * 64 + r < m; ++r) if (q[w * 64 + r] == t[j]) Eq |= 1ull << r;
            int op, on; uint64_t Ph0;
            ref_cell(R[w], Eq, hp, hn, op, on, Ph0);
            uint32_t nl, nh;
            const PairCarry o = pair_cell<NPL>(L[w], pair_sym_of<NPL>(code), c, nl, nh);
            const uint64_t Pv = (static_cast<uint64_t>(L[w].Pvh) << 32) | L[w].Pvl, Mv = (static_cast<uint64_t>(L[w].Mvh) << 32) | L[w].Mvl;
            const uint64_t nPh = (static_cast<uint64_t>(nh) << 32) | nl;
            // rows past the end of the query hold garbage in both forms; the recurrence only carries upwards, so the rows that exist agree
            const uint64_t vm = valid[w];
            if (((Pv ^ R[w].Pv) & vm) || ((Mv ^ R[w].Mv) & vm) || ((~nPh ^ Ph0) & vm)) { fprintf(stderr, "word %d column %d: state differs\n", w, j); return 1; }
            if (vm == ~0ull && ((static_cast<int>(~o.np >> 31) != op) || (static_cast<int>(o.mn >> 31) != on))) { fprintf(stderr, "word %d column %d: carry differs\n", w, j); return 1; }
            hp = op; hn = on; c = o;
        }
        nxt[0] = j + 1;
        for (int i = 1; i <= m; ++i) nxt[i] = std::min({col[i] + 1, nxt[i - 1] + 1, col[i - 1] + (q[i - 1] == t[j] ? 0 : 1)});
        col.swap(nxt);
        int acc = j + 1;
        for (int i = 1; i <= m; ++i) {
            const int w = (i - 1) >> 6, b = (i - 1) & 63;
            const uint64_t Pv = (static_cast<uint64_t>(L[w].Pvh) << 32) | L[w].Pvl, Mv = (static_cast<uint64_t>(L[w].Mvh) << 32) | L[w].Mvl;
            acc += static_cast<int>((Pv >> b) & 1) - static_cast<int>((Mv >> b) & 1);
            if (acc != col[i]) { fprintf(stderr, "column %d row %d: score %d, DP says %d\n", j, i, acc, col[i]); return 1; }
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    std::mt19937_64 rng(20260930);
    long long cells = 0;
    for (int r = 0; r < reps; ++r) {
        const int m = 1 + static_cast<int>(rng() % 400), n = 1 + static_cast<int>(rng() % 300);
        const double sim = (rng() % 100) / 100.0;
        if (run<2>(rng, m, n, 1 + static_cast<int>(rng() % 3), sim)) return 1;          // up to 3 query symbols + a foreign one in 2 planes
        if (run<3>(rng, m, n, 1 + static_cast<int>(rng() % 7), sim)) return 1;
        if (run<8>(rng, m, n, 1 + static_cast<int>(rng() % 200), sim)) return 1;
        cells += 3ll * m * n;
    }
    printf("pair_cell ok: %d problems x 3 plane counts, %lld cells\n", reps, cells);
    return 0;
}
