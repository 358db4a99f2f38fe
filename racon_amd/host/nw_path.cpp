#include "nw_path.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace racon {
namespace nwpath {

namespace {

struct View {                       // a string read forwards or backwards
    const uint8_t* p; int64_t n; bool rev;
    uint8_t operator[](int64_t i) const { return rev ? p[n - 1 - i] : p[i]; }
};

struct Scratch {
    std::vector<uint64_t> peq, pv, mv;
    std::vector<int32_t> left, right, band;
};

// Last column of the global edit-distance matrix: out[i] = ED(q[:i], t) for i = 0..m.
// Myers' bit-vector recurrence (64 rows per word), top boundary +1 per column.
void last_column(const View& q, const View& t, Scratch& s, std::vector<int32_t>& out) {
    const int64_t m = q.n, n = t.n;
    out.resize(static_cast<size_t>(m) + 1);
    if (m == 0) { out[0] = static_cast<int32_t>(n); return; }
    const int64_t nb = (m + 63) / 64;
    s.peq.assign(static_cast<size_t>(256 * nb), 0);
    for (int64_t i = 0; i < m; ++i) s.peq[static_cast<size_t>(q[i]) * nb + i / 64] |= 1ull << (i % 64);
    s.pv.assign(static_cast<size_t>(nb), ~0ull);
    s.mv.assign(static_cast<size_t>(nb), 0ull);
    for (int64_t j = 0; j < n; ++j) {
        const uint64_t* eqc = &s.peq[static_cast<size_t>(t[j]) * nb];
        int hin = 1;
        for (int64_t k = 0; k < nb; ++k) {
            uint64_t eq = eqc[k];
            const uint64_t pv = s.pv[k], mv = s.mv[k];
            const uint64_t xv = eq | mv;
            if (hin < 0) eq |= 1ull;
            const uint64_t xh = (((eq & pv) + pv) ^ pv) | eq;
            uint64_t ph = mv | ~(xh | pv);
            uint64_t mh = pv & xh;
            const int hout = static_cast<int>(ph >> 63) - static_cast<int>(mh >> 63);
            ph <<= 1; mh <<= 1;
            if (hin < 0) mh |= 1ull; else if (hin > 0) ph |= 1ull;
            s.pv[k] = mh | ~(xv | ph);
            s.mv[k] = ph & xv;
            hin = hout;
        }
    }
    int32_t d = static_cast<int32_t>(n);
    out[0] = d;
    for (int64_t i = 0; i < m; ++i) {
        const uint64_t bit = 1ull << (i % 64);
        if (s.pv[i / 64] & bit) ++d; else if (s.mv[i / 64] & bit) --d;
        out[static_cast<size_t>(i) + 1] = d;
    }
}

constexpr int32_t kInf = 1 << 29;

// Plain traceback over the region of the matrix optimal paths can visit: a cell
// (i, j) lies on some optimal path only if |j-i| + |(n-m)-(j-i)| <= best, and every
// comparison the walk makes involves such cells or values that can only be
// over-estimated outside it, so restricting the DP to that diagonal band leaves
// every decision unchanged.  Preference: up (I), left (D), diagonal (M).
void traceback_ops(const uint8_t* q, int64_t m, const uint8_t* t, int64_t n, int64_t best, Scratch& s, std::string& ops) {
    const int64_t delta = n - m;
    const int64_t slack = (best - (delta < 0 ? -delta : delta)) / 2;
    const int64_t dmin = std::min<int64_t>(0, delta) - slack, dmax = std::max<int64_t>(0, delta) + slack;
    const int64_t W = dmax - dmin + 1;                  // row i holds columns j = i + dmin + c, c in [0, W)
    s.band.assign(static_cast<size_t>((m + 1) * W), kInf);
    auto at = [&](int64_t i, int64_t j) -> int32_t {
        const int64_t c = j - i - dmin;
        return (c < 0 || c >= W || j < 0 || j > n) ? kInf : s.band[static_cast<size_t>(i * W + c)];
    };
    for (int64_t i = 0; i <= m; ++i) {
        int32_t* row = &s.band[static_cast<size_t>(i * W)];
        const int32_t* up = i ? row - W : nullptr;     // up[c+1] = (i-1, j), up[c] = (i-1, j-1)
        const int64_t j0 = std::max<int64_t>(0, i + dmin), j1 = std::min<int64_t>(n, i + dmax);
        for (int64_t j = j0; j <= j1; ++j) {
            const int64_t c = j - i - dmin;
            int32_t v;
            if (i == 0) v = static_cast<int32_t>(j);
            else if (j == 0) v = static_cast<int32_t>(i);
            else {
                v = up[c] + (q[i - 1] != t[j - 1]);
                if (c + 1 < W) v = std::min(v, up[c + 1] + 1);
                if (c > 0) v = std::min(v, row[c - 1] + 1);
            }
            row[c] = v;
        }
    }
    if (at(m, n) != best) throw std::runtime_error("[racon::nwpath] internal error: banded score mismatch");
    const size_t base = ops.size();
    int64_t i = m, j = n;
    while (i > 0 || j > 0) {
        const int32_t cur = at(i, j);
        if (i > 0 && at(i - 1, j) + 1 == cur) { ops.push_back('I'); --i; }
        else if (j > 0 && at(i, j - 1) + 1 == cur) { ops.push_back('D'); --j; }
        else { ops.push_back('M'); --i; --j; }
    }
    std::reverse(ops.begin() + static_cast<std::ptrdiff_t>(base), ops.end());
}

void obtain(const uint8_t* q, int64_t m, const uint8_t* t, int64_t n, int64_t best, Scratch& s, std::string& ops) {
    if (m == 0) { ops.append(static_cast<size_t>(n), 'D'); return; }
    if (n == 0) { ops.append(static_cast<size_t>(m), 'I'); return; }
    const int64_t blocks = (m + 63) / 64;
    if ((2 * 8 + 4) * blocks * n + 2 * 4 * n < 1024 * 1024) { traceback_ops(q, m, t, n, best, s, ops); return; }
    const int64_t lw = n / 2, rw = n - lw;
    std::vector<int32_t> left, right;                    // left[h] = ED(q[:h], t[:lw]); right[k] = ED(q[m-k:], t[lw:])
    last_column(View{q, m, false}, View{t, lw, false}, s, left);
    last_column(View{q, m, true}, View{t + lw, rw, true}, s, right);
    int64_t h = -1;
    for (int64_t c = 1; c <= m - 1; ++c) if (left[c] + right[m - c] == best) { h = c; break; }
    if (h < 0 && lw + right[m] == best) h = 0;
    if (h < 0 && left[m] + rw == best) h = m;
    if (h < 0) throw std::runtime_error("[racon::nwpath] internal error: no optimal split");
    const int64_t ls = h > 0 ? left[h] : lw, rs = h < m ? right[m - h] : rw;
    left = std::vector<int32_t>(); right = std::vector<int32_t>();
    obtain(q, h, t, lw, ls, s, ops);
    obtain(q + h, m - h, t + lw, rw, rs, s, ops);
}

}  // namespace

uint64_t edit_distance(const char* query, uint64_t m, const char* target, uint64_t n) {
    Scratch s; std::vector<int32_t> col;
    last_column(View{reinterpret_cast<const uint8_t*>(query), static_cast<int64_t>(m), false},
                View{reinterpret_cast<const uint8_t*>(target), static_cast<int64_t>(n), false}, s, col);
    return static_cast<uint64_t>(col[m]);
}

std::string align_cigar(const char* query, uint32_t m, const char* target, uint32_t n) {
    const uint8_t* q = reinterpret_cast<const uint8_t*>(query);
    const uint8_t* t = reinterpret_cast<const uint8_t*>(target);
    Scratch s;
    std::vector<int32_t> col;
    last_column(View{q, m, false}, View{t, n, false}, s, col);
    std::string ops;
    ops.reserve(static_cast<size_t>(m) + n);
    obtain(q, m, t, n, col[m], s, ops);
    std::string cigar;
    for (size_t a = 0; a < ops.size();) {
        size_t b = a;
        while (b < ops.size() && ops[b] == ops[a]) ++b;
        cigar += std::to_string(b - a);
        cigar += ops[a];
        a = b;
    }
    return cigar;
}

}  // namespace nwpath
}  // namespace racon
