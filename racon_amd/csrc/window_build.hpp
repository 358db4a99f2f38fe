// window_build.hpp — rcn_engine_build_windows: racon's windows built in HBM.
//
// What the reference does serially on the host at the end of Polisher::initialize —
//   * cut every target into backbone windows              (reference src/polisher.cpp:388-403, src/window.cpp:15-40)
//   * cut every overlap into layers at its breaking points (reference src/polisher.cpp:405-461), i.e. per pair of points
//     the length filter (:415), the mean-quality filter (:419-433), window rank / begin / end (:436-457) and
//     Window::add_layer (src/window.cpp:42-63), reading the reverse complement for strand 1 (src/sequence.cpp:49-84)
// — as four data-parallel steps over arrays that are uploaded once:
//   1. k_layer_filter : one wave per breaking-point pair: filters, window id, begin / end; layers per window (atomics)
//   2. stable radix sort of the kept pairs by window id (hipCUB): inside a window the layers keep overlap order, which
//      is the order the serial loop calls add_layer in (the engine's std::sort emulation depends on it)
//   3. exclusive scans -> win_seq_off, seq_off; k_seq_table writes one source descriptor per sequence of the batch
//   4. k_gather       : sixteen lanes per sequence copy (or reverse-complement) bases and qualities into the packed batch
// Everything is integer / byte work bounded by HBM traffic; the only floating point is the reference's own
// `q1 - q0 < 0.02 * w` and `mean quality < threshold` comparisons, evaluated in double exactly as there.
#pragma once

#include <hipcub/hipcub.hpp>

namespace rcn {

struct BuildParams {
    // reads (rcn_read_set, device copies)
    const uint64_t* seq_off; const uint8_t* bases; const uint8_t* quals; const uint8_t* has_qual;
    uint32_t n_targets;
    // overlaps (rcn_overlap_set, device copies)
    const uint32_t* q_id; const uint32_t* t_id; const uint8_t* strand; const uint64_t* bp_off;
    const uint32_t* bp_t; const uint32_t* bp_q;
    uint64_t n_overlaps, n_pairs;
    // windows
    const uint32_t* first_window;              // [n_targets + 1]
    uint32_t n_windows, W, window_type;
    double qthr;
    // per pair
    uint32_t* key; uint32_t* val; uint32_t* pair_begin; uint32_t* pair_end; uint32_t* pair_q0; uint32_t* pair_len; uint32_t* pair_ovl;
    uint32_t* win_cnt;                         // [n_windows + 1]: layers per window, then (+1 backbone) scanned into win_seq_off
    uint32_t* err;                             // [0] add_layer contract violations, [1..8] symbols seen (256 bits)
};

// largest i in [0, n) with a[i] <= x (a ascending, a[0] <= x)
template <class T>
__device__ __forceinline__ uint64_t last_le(const T* a, uint64_t n, T x) {
    uint64_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}

// which overlap a breaking-point pair belongs to: one thread per overlap writes its own pairs (a binary search over the
// offsets per pair is ~15 dependent HBM loads in front of everything else the filter does: it was most of its time)
__global__ __launch_bounds__(256) void k_pair_owner(BuildParams P) {
    const uint64_t o = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (o >= P.n_overlaps) return;
    for (uint64_t p = P.bp_off[o] / 2, e = P.bp_off[o + 1] / 2; p < e; ++p) P.pair_ovl[p] = static_cast<uint32_t>(o);
}

__global__ __launch_bounds__(256) void k_layer_filter(BuildParams P) {
    const int lane = threadIdx.x & 63;
    const uint64_t p = static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (p >= P.n_pairs) return;
    const uint64_t o = P.pair_ovl[p];
    const uint32_t t0 = P.bp_t[2 * p], t1 = P.bp_t[2 * p + 1], q0 = P.bp_q[2 * p], q1 = P.bp_q[2 * p + 1];
    const uint32_t qi = P.q_id[o], ti = P.t_id[o];
    const bool rev = P.strand[o] != 0;
    const uint32_t dl = q1 - q0;                                               // uint32 arithmetic, as the reference
    bool keep = !(static_cast<double>(dl) < 0.02 * static_cast<double>(P.W));  // polisher.cpp:415
    if (keep && P.has_qual[qi]) {                                              // polisher.cpp:419-433
        const uint64_t a = P.seq_off[qi], rlen = P.seq_off[qi + 1] - a;
        unsigned long long sum = 0;
        if (q1 > q0) for (uint32_t k = q0 + lane; k < q1; k += 64) sum += static_cast<uint32_t>(P.quals[a + (rev ? rlen - 1 - k : k)]) - 33u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        const double average_quality = static_cast<double>(sum) / static_cast<double>(dl);
        if (average_quality < P.qthr) keep = false;
    }
    uint32_t wid = P.n_windows, begin = 0, end = 0;
    if (keep) {
        const uint32_t rank = t0 / P.W, window_start = rank * P.W;
        const uint32_t n_win_t = P.first_window[ti + 1] - P.first_window[ti];
        begin = t0 - window_start; end = t1 - window_start - 1;                 // polisher.cpp:454-457
        if (dl == 0 || begin == end) keep = false;                             // window.cpp:45-47: silently ignored
        else {
            const uint64_t tlen = P.seq_off[ti + 1] - P.seq_off[ti];
            const uint64_t L = rank < n_win_t ? min(static_cast<uint64_t>(P.W), tlen - static_cast<uint64_t>(window_start)) : 0;
            if (rank >= n_win_t || begin >= end || begin > L || end > L) { keep = false; if (lane == 0) atomicAdd(&P.err[0], 1u); }   // window.cpp:49-58: fatal
            else wid = P.first_window[ti] + rank;
        }
    }
    if (lane == 0) {
        P.key[p] = keep ? wid : P.n_windows; P.val[p] = static_cast<uint32_t>(p);
        P.pair_begin[p] = begin; P.pair_end[p] = end; P.pair_q0[p] = q0; P.pair_len[p] = dl;
        if (keep) atomicAdd(&P.win_cnt[wid], 1u);
    }
}

__global__ void k_plus_one(uint32_t* cnt, uint32_t n) {      // layers -> sequences per window (backbone); slot n ends the scan
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n) cnt[w] += 1u; else if (w == n) cnt[w] = 0u;
}

struct SeqTable {
    const uint32_t* win_seq_off; const uint32_t* key_sorted; const uint32_t* val_sorted;
    uint64_t n_layers;
    uint64_t* seq_len;          // [n_seqs + 1] -> scanned into seq_off
    uint32_t* begin; uint32_t* end; uint8_t* has_qual; uint8_t* win_type;
    uint32_t* src_id; uint32_t* src_pos; uint8_t* src_flags;      // flags: 1 = reverse strand, 2 = dummy '!' quality
};

__global__ __launch_bounds__(256) void k_seq_table(BuildParams P, SeqTable T) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < P.n_windows) {                                   // backbone of window i (createWindow, window.cpp:29-37)
        const uint32_t w = static_cast<uint32_t>(i);
        const uint32_t ti = static_cast<uint32_t>(last_le<uint32_t>(P.first_window, static_cast<uint64_t>(P.n_targets) + 1, w));
        const uint32_t rank = w - P.first_window[ti];
        const uint64_t tlen = P.seq_off[ti + 1] - P.seq_off[ti];
        const uint32_t s = T.win_seq_off[w];
        T.seq_len[s] = min(static_cast<uint64_t>(P.W), tlen - static_cast<uint64_t>(rank) * P.W);
        T.begin[s] = 0; T.end[s] = 0; T.has_qual[s] = 1;     // a backbone always has a quality pointer: the target's or the dummy
        T.src_id[s] = ti; T.src_pos[s] = rank * P.W; T.src_flags[s] = P.has_qual[ti] ? 0 : 2;
        T.win_type[w] = static_cast<uint8_t>(P.window_type);
        if (w == 0) T.seq_len[T.win_seq_off[P.n_windows]] = 0;
    }
    if (i < T.n_layers) {                                    // layer number i of the window-sorted list
        const uint32_t w = T.key_sorted[i], p = T.val_sorted[i];
        const uint64_t s = i + w + 1;                         // = win_seq_off[w] + 1 + (i - layers before window w)
        const uint32_t o = P.pair_ovl[p], qi = P.q_id[o];
        T.seq_len[s] = P.pair_len[p];
        T.begin[s] = P.pair_begin[p]; T.end[s] = P.pair_end[p]; T.has_qual[s] = P.has_qual[qi] ? 1 : 0;
        T.src_id[s] = qi; T.src_pos[s] = P.pair_q0[p]; T.src_flags[s] = P.strand[o] ? 1 : 0;
    }
}

struct GatherParams {
    const uint64_t* read_off; const uint8_t* read_bases; const uint8_t* read_quals;
    const uint64_t* seq_off; const uint32_t* src_id; const uint32_t* src_pos; const uint8_t* src_flags; const uint8_t* has_qual;
    uint8_t* bases; uint8_t* quals; uint64_t n_seqs;
};

// Sixteen lanes per sequence, 512 consecutive bytes per step: two (unaligned) 16-byte accesses per lane and array, the tail
// byte by byte -- a layer of a w = 500 window is one step, a wave has four layers' loads in flight at once and a 256-thread
// work-group sixteen (one wave per sequence and a dword per lane moved 2.5-2.7 TB/s, half a wave and 16 bytes per lane
// 3.5: the 300 000 short copies of a batch are latency, not bandwidth).  Reverse strand: base k of the layer is the complement of base (rlen - 1 - (q0 + k))
// of the read, its quality the quality of that base (Sequence::create_reverse_complement).
__device__ __forceinline__ uint32_t comp_byte(uint32_t b) { return b == 'A' ? 'T' : b == 'T' ? 'A' : b == 'C' ? 'G' : b == 'G' ? 'C' : b; }
__device__ __forceinline__ uint32_t comp_dword_reversed(uint32_t v) {           // bytes reversed, each complemented
    v = __builtin_bswap32(v);
    return comp_byte(v & 255) | (comp_byte((v >> 8) & 255) << 8) | (comp_byte((v >> 16) & 255) << 16) | (comp_byte(v >> 24) << 24);
}

constexpr int kGatherLanes = 16;          // lanes per sequence: 32 bytes per lane and step (two 16-byte accesses per array)
__global__ __launch_bounds__(256) void k_gather(GatherParams G) {
    const int hl = threadIdx.x & (kGatherLanes - 1);
    const uint64_t s = static_cast<uint64_t>(blockIdx.x) * (256 / kGatherLanes) + threadIdx.x / kGatherLanes;
    if (s >= G.n_seqs) return;
    const uint64_t dst = G.seq_off[s], len = G.seq_off[s + 1] - dst;
    const uint32_t id = G.src_id[s], pos = G.src_pos[s], fl = G.src_flags[s];
    const uint64_t a = G.read_off[id], rlen = G.read_off[id + 1] - a;
    const bool rev = (fl & 1) != 0, real_q = !(fl & 2) && G.has_qual[s] != 0;
    const uint64_t len16 = len & ~15ull;
    const uint4 noq = make_uint4(0x21212121u, 0x21212121u, 0x21212121u, 0x21212121u);
    auto rc16 = [](uint4 v) { return make_uint4(comp_dword_reversed(v.w), comp_dword_reversed(v.z), comp_dword_reversed(v.y), comp_dword_reversed(v.x)); };
    auto rv16 = [](uint4 v) { return make_uint4(__builtin_bswap32(v.w), __builtin_bswap32(v.z), __builtin_bswap32(v.y), __builtin_bswap32(v.x)); };
    for (uint64_t k = 32ull * hl; k < len16; k += 32ull * kGatherLanes) {
        // forward: bytes src .. src + 15; reverse: the sixteen bytes ending at the mirrored position, reversed
        const bool two = k + 16 < len16;
        const uint64_t k1 = two ? k + 16 : k;
        const uint64_t s0 = a + (rev ? rlen - 16 - (pos + k) : pos + k), s1 = a + (rev ? rlen - 16 - (pos + k1) : pos + k1);
        uint4 b0 = *reinterpret_cast<const uint4*>(G.read_bases + s0), b1 = *reinterpret_cast<const uint4*>(G.read_bases + s1);
        uint4 q0 = noq, q1 = noq;
        if (real_q) { q0 = *reinterpret_cast<const uint4*>(G.read_quals + s0); q1 = *reinterpret_cast<const uint4*>(G.read_quals + s1); }
        if (rev) { b0 = rc16(b0); b1 = rc16(b1); q0 = rv16(q0); q1 = rv16(q1); }
        *reinterpret_cast<uint4*>(G.bases + dst + k) = b0;
        *reinterpret_cast<uint4*>(G.quals + dst + k) = q0;
        if (two) { *reinterpret_cast<uint4*>(G.bases + dst + k1) = b1; *reinterpret_cast<uint4*>(G.quals + dst + k1) = q1; }
    }
    for (uint64_t k = len16 + hl; k < len; k += kGatherLanes) {
        const uint64_t src = a + (rev ? rlen - 1 - (pos + k) : pos + k);
        uint32_t b8 = G.read_bases[src];
        if (rev) b8 = comp_byte(b8);
        G.bases[dst + k] = static_cast<uint8_t>(b8);
        G.quals[dst + k] = real_q ? G.read_quals[src] : static_cast<uint8_t>('!');
    }
}

// which byte values occur in the reads (256 bits): bounds the number of distinct symbols of any window, i.e. the size
// of an aligned ring (the engine sizes its per-node ring slots with it)
__global__ __launch_bounds__(256) void k_symbols(const uint8_t* bases, uint64_t n, uint32_t* bits) {
    __shared__ uint32_t pres[8];
    if (threadIdx.x < 8) pres[threadIdx.x] = 0;
    __syncthreads();
    uint32_t acgt = 0;                                       // symbols 64..95, where ACGT live; anything else through LDS
    auto one = [&](uint32_t b) { if ((b >> 5) == 2) acgt |= 1u << (b & 31); else atomicOr(&pres[b >> 5], 1u << (b & 31)); };
    // sixteen bytes per load (the buffer is 256-byte aligned and padded); a dword whose four bytes all lie in 64..95 -- all
    // of them, for A/C/G/T reads -- costs four shifts, anything else goes byte by byte
    const uint64_t n16 = n & ~15ull;
    const uint4* v = reinterpret_cast<const uint4*>(bases);
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n16 / 16; k += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint4 q = v[k];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (((w[j] >> 5) & 0x07070707u) == 0x02020202u)
                acgt |= (1u << (w[j] & 31)) | (1u << ((w[j] >> 8) & 31)) | (1u << ((w[j] >> 16) & 31)) | (1u << ((w[j] >> 24) & 31));
            else { one(w[j] & 255); one((w[j] >> 8) & 255); one((w[j] >> 16) & 255); one(w[j] >> 24); }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 16 && n16 + threadIdx.x < n) one(bases[n16 + threadIdx.x]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acgt |= __shfl_xor(acgt, d);
    if ((threadIdx.x & 63) == 0 && acgt) atomicOr(&pres[2], acgt);
    __syncthreads();
    if (threadIdx.x < 8 && pres[threadIdx.x]) atomicOr(&bits[threadIdx.x], pres[threadIdx.x]);
}

// Breaking points of one overlap from its CIGAR (reference src/overlap.cpp:226-292, restated per operation instead of per
// base): the first and the last + 1 (target, query) position of the match columns inside every window the overlap
// touches.  Slot k of the overlap belongs to the k-th window end in {multiples of W inside (t_begin, t_end)} + {t_end};
// a window without a match column keeps its zeroed slot, which the length filter of k_layer_filter then drops.
struct CigarParams {
    const uint64_t* cigar_off; const uint8_t* cigar;
    const uint32_t* q_start; const uint32_t* t_begin; const uint32_t* t_end;
    const uint64_t* bp_off;                       // [n_overlaps + 1] slot offsets, in points
    uint32_t* bp_t; uint32_t* bp_q;               // zero-initialised
    uint64_t n_overlaps; uint32_t W;
};

__global__ __launch_bounds__(256) void k_cigar_breaking_points(CigarParams C) {
    const uint64_t o = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (o >= C.n_overlaps) return;
    const uint64_t W = C.W;
    const int64_t t_begin = C.t_begin[o], t_end = C.t_end[o];
    const uint64_t out = C.bp_off[o], n_slots = (C.bp_off[o + 1] - out) / 2;
    // window end number k: the multiples of W strictly inside (t_begin, t_end), minus one; the last one is t_end - 1
    const int64_t k0 = t_begin / static_cast<int64_t>(W) + 1;
    auto end_of = [&](uint64_t k) -> int64_t { return k + 1 < n_slots ? (k0 + static_cast<int64_t>(k)) * static_cast<int64_t>(W) - 1 : t_end - 1; };
    uint64_t w = 0;
    bool open = false;
    uint32_t ft = 0, fq = 0, lt = 0, lq = 0;
    int64_t q = static_cast<int64_t>(C.q_start[o]) - 1, t = t_begin - 1;
    auto close_if_at_end = [&]() {
        if (w < n_slots && t == end_of(w)) {
            if (open) { C.bp_t[out + 2 * w] = ft; C.bp_q[out + 2 * w] = fq; C.bp_t[out + 2 * w + 1] = lt; C.bp_q[out + 2 * w + 1] = lq; }
            open = false; ++w;
        }
    };
    uint64_t count = 0; bool have = false;
    const uint64_t p_end = C.cigar_off[o + 1];
    // eight bytes of text per (unaligned) load: the walk is one dependent chain per thread, its time is the number of
    // memory round trips (the buffer has 16 bytes of slack behind the last CIGAR)
    for (uint64_t p8 = C.cigar_off[o]; p8 < p_end; p8 += 8) {
      const uint64_t chunk = *reinterpret_cast<const uint64_t*>(C.cigar + p8);
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        if (p8 + kb >= p_end) break;
        const uint8_t c = static_cast<uint8_t>(chunk >> (8 * kb));
        if (c >= '0' && c <= '9') { count = count * 10 + (c - '0'); have = true; continue; }
        uint64_t n = static_cast<uint32_t>(have ? count : 0);
        count = 0; have = false;
        if (c == 'M' || c == '=' || c == 'X') {
            while (n > 0) {
                // up to the next window end in one step (the reference walks base by base)
                uint64_t take = n;
                if (w < n_slots) { const int64_t room = end_of(w) - t; if (room > 0 && static_cast<uint64_t>(room) < take) take = static_cast<uint64_t>(room); }
                if (!open) { open = true; ft = static_cast<uint32_t>(t + 1); fq = static_cast<uint32_t>(q + 1); }
                q += static_cast<int64_t>(take); t += static_cast<int64_t>(take); n -= take;
                lt = static_cast<uint32_t>(t + 1); lq = static_cast<uint32_t>(q + 1);
                close_if_at_end();
            }
        } else if (c == 'I') {
            q += static_cast<int64_t>(n);
        } else if (c == 'D' || c == 'N') {
            while (n > 0) {
                uint64_t take = n;
                if (w < n_slots) { const int64_t room = end_of(w) - t; if (room > 0 && static_cast<uint64_t>(room) < take) take = static_cast<uint64_t>(room); }
                t += static_cast<int64_t>(take); n -= take;
                close_if_at_end();
            }
        }
      }
    }
}

// The same walk with one WAVE per overlap: 64 bytes of CIGAR text per step, no divergence.  Lanes holding an operation
// character get the number in front of it from one weighted prefix sum over the digits (which sit in the lanes before it,
// or in the carry of the previous step), prefix sums turn the operation lengths into target / query start positions (all
// sums in six DPP steps each: the first version walked the digits with ds_bpermute and scanned 64-bit values with
// __shfl_up, 1.09 ms for 24 000 overlaps), and every match run writes its
// "first match column" and "last + 1" of the windows it covers as packed (target << 32 | query) keys -- inside one overlap
// both coordinates grow together, so the first run that reaches a window holds its first column and the last one its last.  A window counts only once the walk has passed its end (the reference pushes a pair when it closes a window).
struct CigarWaveParams {
    CigarParams c;
    unsigned long long* first_key;                // [slots] initialised to ~0
    unsigned long long* last_key;                 // [slots] initialised to 0
};

// wave-wide inclusive sum in six DPP steps (lanes without a source add 0)
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false));
    return v;
}

__global__ __launch_bounds__(256) void k_cigar_breaking_points_wave(CigarWaveParams P) {
    const CigarParams& C = P.c;
    const int lane = threadIdx.x & 63;
    // 10^k mod 2^32, k = 0 .. 64: a number is the sum of its digits times these (the reference parses into uint32_t, n = n * 10 + d)
    __shared__ uint32_t p10[65];
    if (threadIdx.x < 65) { uint32_t v = 1; for (unsigned k = 0; k < threadIdx.x; ++k) v *= 10u; p10[threadIdx.x] = v; }
    __syncthreads();
    const uint64_t o = static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (o >= C.n_overlaps) return;
    // Positions are 32-bit, as in the reference (`uint32_t q_ptr, t_ptr`, src/overlap.cpp:239-242: a CIGAR whose numbers overflow them
    // wraps there exactly as here); only the slot and text offsets are 64-bit.  (Round 5: the walk's position arithmetic was 64-bit
    // throughout -- two 16-bit-half scans per sum, 64-bit compares and multiplies per run: most of the kernel's ~500 instructions per
    // 64-byte step.)
    const uint32_t W = static_cast<uint32_t>(C.W);
    const uint32_t t_begin = C.t_begin[o], t_end = C.t_end[o];
    const uint64_t slot0 = C.bp_off[o] / 2;
    const uint32_t n_slots = static_cast<uint32_t>((C.bp_off[o + 1] - C.bp_off[o]) / 2);
    const uint32_t wb = t_begin / W;                                  // window number of slot 0
    auto end_of = [&](uint32_t k) -> uint32_t { return k + 1 < n_slots ? (wb + 1 + k) * W - 1 : t_end - 1; };
    const uint64_t a = C.cigar_off[o], z = C.cigar_off[o + 1];
    uint32_t t_run = t_begin, q_run = C.q_start[o];                    // next target / query position to be consumed
    uint32_t carry = 0;                                               // value of the digits pending from the previous step
    int kdone = -1;                                                   // last window a match run of the previous steps reached
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t c_next = a + lane < z ? C.cigar[a + lane] : '0';           // (the next step's text is in flight while this one is worked on)
    for (uint64_t p0 = a; p0 < z; p0 += 64) {
        const uint32_t c = c_next;                                    // padding = digits that never meet an operation
        c_next = p0 + 64 + lane < z ? C.cigar[p0 + 64 + lane] : '0';
        const bool is_digit = c >= '0' && c <= '9';
        const unsigned long long opmask = __ballot(!is_digit);
        // Every digit weighs 10^(digits between it and the operation its number belongs to) -- trailing digits, whose
        // operation is in the next step: up to the end of this one -- and one wave-wide sum gives every operation its
        // number as a difference of two prefix sums, all mod 2^32.
        const unsigned long long above = lane < 63 ? (opmask >> (lane + 1)) : 0ull;
        const int nxt = above ? lane + 1 + __builtin_ctzll(above) : 64;
        const uint32_t wgt = is_digit ? (c - '0') * p10[nxt - 1 - lane] : 0u;
        const uint32_t S = wave_incl_add(wgt);
        const unsigned long long before = opmask & below;
        const int prev_op = before ? 63 - __builtin_clzll(before) : -1;
        const uint32_t Sprev = static_cast<uint32_t>(__shfl(static_cast<int>(S), prev_op < 0 ? 0 : prev_op));
        uint32_t n = 0;
        if (!is_digit) {
            n = S - (prev_op < 0 ? 0u : Sprev);
            if (prev_op < 0) n += carry * p10[lane];                   // the number started in the previous step(s)
        }
        {   // carry out: the digits behind the last operation of this step (everything, if there is none)
            const uint32_t S63 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(S), 63));
            if (opmask) carry = S63 - static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(S), 63 - __builtin_clzll(opmask)));
            else carry = carry * p10[64] + S63;
        }
        const bool isM = !is_digit && (c == 'M' || c == '=' || c == 'X');
        const uint32_t dt = (isM || (!is_digit && (c == 'D' || c == 'N'))) ? n : 0u;
        const uint32_t dq = (isM || (!is_digit && c == 'I')) ? n : 0u;
        const uint32_t st = wave_incl_add(dt), sq = wave_incl_add(dq);      // (mod 2^32, like the reference's pointers)
        const uint32_t ts = t_run + (st - dt), qs = q_run + (sq - dq);
        // This lane's match run covers the windows kf .. kl of the overlap.  Positions only grow along the walk, so a
        // window's first match column comes from the FIRST run that reaches it and its last one from the LAST: a run
        // writes "first" only for the windows no earlier run (of this step, or of the steps before: kdone) reached,
        // and "last" only for those the next run of this step does not reach as well -- plain stores, later steps
        // overwrite "last" (the slots belong to this overlap alone; one atomic pair per run and window was 34 M
        // 64-bit atomics for 24 000 overlaps, and their rate at the L2 was the kernel's time).
        const uint32_t te = ts + n;                                   // match columns ts .. te - 1
        int kf = 0, kl = -1;
        if (isM && n > 0 && te > ts) {                                // (te <= ts: the positions wrapped -- nothing sane to record)
            const uint32_t wf = ts / W, wl = (te - 1) / W;
            if (wf >= wb && wf - wb < n_slots && !(wf - wb == n_slots - 1 && ts >= t_end)) {
                kf = static_cast<int>(wf - wb);
                kl = static_cast<int>(min(wl - wb, n_slots - 1));
            }
        }
        const unsigned long long runs = __ballot(kl >= kf);
        const unsigned long long rb = runs & below, ra = lane < 63 ? (runs >> (lane + 1)) : 0ull;
        const int kl_prev = __shfl(kl, rb ? 63 - __builtin_clzll(rb) : 0), kf_next = __shfl(kf, ra ? lane + 1 + __builtin_ctzll(ra) : 0);
        if (kl >= kf) {
            const int reached = rb ? kl_prev : kdone;                 // windows up to here have their first match column
            const int lf = kf > reached + 1 ? kf : reached + 1;       // "first": windows lf .. kl
            const int ll = (ra && kf_next - 1 < kl) ? kf_next - 1 : kl;     // "last": windows kf .. ll
            for (int k = kf; k <= kl; ++k) {
                const uint32_t wstart = k == 0 ? t_begin : (wb + static_cast<uint32_t>(k)) * W, wend = end_of(static_cast<uint32_t>(k));
                const uint32_t f = ts > wstart ? ts : wstart, l = te < wend + 1 ? te : wend + 1;
                if (f < l) {
                    if (k >= lf) P.first_key[slot0 + k] = (static_cast<unsigned long long>(f) << 32) | (qs + (f - ts));
                    if (k <= ll) P.last_key[slot0 + k] = (static_cast<unsigned long long>(l) << 32) | (qs + (l - ts));
                }
            }
        }
        if (runs) kdone = __shfl(kl, 63 - __builtin_clzll(runs));
        t_run += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(st), 63)); q_run += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(sq), 63));
    }
    __threadfence();
    // a window whose end the walk has passed is closed: its pair (if a match column was seen) is what the reference pushed
    for (uint32_t k = lane; k < n_slots; k += 64) {
        const unsigned long long f = P.first_key[slot0 + k], l = P.last_key[slot0 + k];
        if (l != 0 && f != ~0ull && end_of(k) + 1 <= t_run) {
            const uint64_t out = C.bp_off[o] + 2ull * k;
            C.bp_t[out] = static_cast<uint32_t>(f >> 32); C.bp_q[out] = static_cast<uint32_t>(f);
            C.bp_t[out + 1] = static_cast<uint32_t>(l >> 32); C.bp_q[out + 1] = static_cast<uint32_t>(l);
        }
    }
}

// slots of rcn_engine::d_build
enum { kBReadOff, kBReadBases, kBReadQuals, kBReadHasQual, kBQid, kBTid, kBStrand, kBBpOff, kBBpT, kBBpQ, kBFirstWin, kBPairs, kBTemp, kBSeqSrc, kBMisc, kBSeqLen,
       kBCigarOff, kBCigar, kBQStart, kBTBegin, kBTEnd, kBKeyFirst, kBKeyLast, kBuildSlots };

// `C` != nullptr: the breaking points are computed on the device from the alignments (O then only carries n_overlaps,
// q_id, t_id, strand and a host vector of slot offsets in bp_off; its bp_t / bp_q are null).
// alignments that are already in HBM as op bytes (pair_align.hpp); the extents are host arrays
struct OpsSource {
    const uint8_t* d_ops; const uint64_t* d_ops_off;
    const uint32_t* q_start; const uint32_t* t_begin; const uint32_t* t_end;
};

// A large array out of PAGEABLE host memory (the caller's reads, qualities, CIGAR text): hipMemcpyAsync stages such a copy through the
// runtime's own bounce buffers at 1.5-3 GB/s (cfg3 whole: 4.5 GB, most of the 3.2 s "transformed data into windows (on the device)"
// took in round 5's first measurement).  Here host threads copy 32 MB chunks into two pinned slots and the DMA engine takes each
// slot as soon as it is full: the link rate, bounded by the host's memcpy.  Small arrays go the plain way.
inline int upload_staged(DevBuf& d, const void* src, size_t bytes, hipStream_t st) {
    int rc = d.reserve(bytes);
    if (rc) return rc;
    if (!bytes) return RCN_OK;
    constexpr size_t kChunk = 32u << 20;
    if (bytes < kChunk / 2) { HIP_TRY(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, st)); return RCN_OK; }
    // (one staging pair per device: the engines of one device take turns -- uploads of this size are rare and long --, the devices of a
    //  multi-GPU job upload side by side)
    static std::mutex mus[64];
    static HostBuf slot_bufs[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::mutex& mu = mus[dev & 63];
    HostBuf& slots = slot_bufs[dev & 63];
    std::lock_guard<std::mutex> lock(mu);
    if ((rc = slots.reserve(2 * kChunk))) { HIP_TRY(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, st)); return RCN_OK; }
    // (destroyed on every way out, an early return from the creation loop included; the staging pair itself stays for the life of the
    //  process -- 64 MB of pinned memory per device that uploaded this way, released with the process: a static destructor would run
    //  after the HIP runtime's own teardown)
    struct Events { hipEvent_t e[2] = {nullptr, nullptr}; ~Events() { for (auto& x : e) if (x) (void)hipEventDestroy(x); } } events;
    hipEvent_t* ev = events.e;
    for (int q = 0; q < 2; ++q) HIP_TRY(hipEventCreateWithFlags(&ev[q], hipEventDisableTiming));
    const unsigned threads = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    const uint8_t* s8 = static_cast<const uint8_t*>(src);
    int rc_out = RCN_OK;
    size_t k = 0;
    for (size_t off = 0; off < bytes; off += kChunk, ++k) {
        const size_t n = std::min(kChunk, bytes - off);
        uint8_t* slot = slots.as<uint8_t>() + (k & 1) * kChunk;
        if (k >= 2 && hipEventSynchronize(ev[k & 1]) != hipSuccess) { rc_out = RCN_E_HIP; break; }
        const size_t parts = (n + (4u << 20) - 1) / (4u << 20);
        host_parallel(parts, threads, [&](size_t q) { const size_t a = q * (4u << 20), z = std::min(n, a + (4u << 20)); std::memcpy(slot + a, s8 + off + a, z - a); });
        if (hipMemcpyAsync(static_cast<uint8_t*>(d.p) + off, slot, n, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(ev[k & 1], st) != hipSuccess) { rc_out = RCN_E_HIP; break; }
    }
    for (int q = 0; q < 2; ++q) if (k > static_cast<size_t>(q)) (void)hipEventSynchronize(ev[q]);          // the slots are free again when this returns
    return rc_out;
}

inline int upload_reads(rcn_engine* e, const rcn_read_set& R, hipStream_t st) {
    DevBuf* B = e->d_build;
    const uint64_t read_bytes = R.seq_off[R.n_seqs];
    int rc;
    if ((rc = upload_vec(B[kBReadOff], R.seq_off, 8 * (R.n_seqs + 1), st)) || (rc = upload_staged(B[kBReadBases], R.bases, read_bytes, st)) ||
        (rc = upload_staged(B[kBReadQuals], R.quals, read_bytes, st)) || (rc = upload_vec(B[kBReadHasQual], R.seq_has_qual, R.n_seqs, st))) return rc;
    return RCN_OK;
}

// `S` != nullptr: as with `C`, but the alignments are op bytes resident in HBM; `reads_resident`: upload_reads was done.
inline int build_windows(rcn_engine* e, const rcn_read_set& R, const rcn_overlap_set& O, uint32_t W, double qthr, uint8_t window_type,
                         const rcn_cigar_set* C = nullptr, const OpsSource* S = nullptr, bool reads_resident = false) {
    if (R.n_seqs == 0 || R.n_targets == 0 || R.n_targets > R.n_seqs || !R.seq_off || !R.bases || !R.quals || !R.seq_has_qual) return RCN_E_ARG;
    if (O.n_overlaps && (!O.q_id || !O.t_id || !O.strand || !O.bp_off || (!C && !S && (!O.bp_t || !O.bp_q)))) return RCN_E_ARG;
    if (R.n_seqs > 0xfffffffeull || O.n_overlaps > 0xfffffffeull) return RCN_E_ARG;
    HIP_TRY(hipSetDevice(e->cfg.device));
    e->uploaded = false; e->ran = false; e->queued = false;
    e->bstats = rcn_build_stats{};
    hipStream_t st = e->stream;
    hipEvent_t ev[4];
    for (auto& x : ev) HIP_TRY(hipEventCreate(&x));
    auto drop_events = [&]() { for (auto& x : ev) (void)hipEventDestroy(x); };

    // windows per target (polisher.cpp:392-401): ceil(len / W), none for an empty target
    std::vector<uint32_t> first_window(R.n_targets + 1, 0);
    for (uint64_t i = 0; i < R.n_targets; ++i) {
        const uint64_t len = R.seq_off[i + 1] - R.seq_off[i];
        const uint64_t k = (len + W - 1) / W;
        if (first_window[i] + k > 0xfffffff0ull) { drop_events(); return RCN_E_CAPACITY; }
        first_window[i + 1] = first_window[i] + static_cast<uint32_t>(k);
    }
    const uint32_t nw = first_window[R.n_targets];
    const uint64_t n_points = O.n_overlaps ? O.bp_off[O.n_overlaps] : 0;
    if (nw == 0 || (n_points & 1) || n_points / 2 > 0x7fffffffull) { drop_events(); return RCN_E_ARG; }
    const uint64_t n_pairs = n_points / 2;
    const uint64_t read_bytes = R.seq_off[R.n_seqs];

    HIP_TRY(hipEventRecord(ev[0], st));
    int rc;
    DevBuf* B = e->d_build;
    if ((!reads_resident && (rc = upload_reads(e, R, st))) ||
        (rc = upload_vec(B[kBQid], O.q_id, 4 * O.n_overlaps, st)) || (rc = upload_vec(B[kBTid], O.t_id, 4 * O.n_overlaps, st)) ||
        (rc = upload_vec(B[kBStrand], O.strand, O.n_overlaps, st)) || (rc = upload_vec(B[kBBpOff], O.bp_off ? O.bp_off : &n_points, 8 * (O.n_overlaps + 1), st)) ||
        (rc = upload_vec(B[kBFirstWin], first_window.data(), 4 * (R.n_targets + 1), st))) { drop_events(); return rc; }
    if (S) {
        if ((rc = B[kBBpT].reserve(4 * n_points + 16)) || (rc = B[kBBpQ].reserve(4 * n_points + 16)) ||
            (rc = upload_vec(B[kBQStart], S->q_start, 4 * O.n_overlaps, st)) || (rc = upload_vec(B[kBTBegin], S->t_begin, 4 * O.n_overlaps, st)) ||
            (rc = upload_vec(B[kBTEnd], S->t_end, 4 * O.n_overlaps, st))) { drop_events(); return rc; }
        HIP_TRY(hipMemsetAsync(B[kBBpT].p, 0, 4 * n_points + 16, st));
        HIP_TRY(hipMemsetAsync(B[kBBpQ].p, 0, 4 * n_points + 16, st));
    } else if (!C) {
        if ((rc = upload_vec(B[kBBpT], O.bp_t, 4 * n_points, st)) || (rc = upload_vec(B[kBBpQ], O.bp_q, 4 * n_points, st))) { drop_events(); return rc; }
    } else {
        const uint64_t cig_bytes = C->n_overlaps ? C->cigar_off[C->n_overlaps] : 0;
        if ((rc = B[kBBpT].reserve(4 * n_points + 16)) || (rc = B[kBBpQ].reserve(4 * n_points + 16)) ||
            (rc = upload_vec(B[kBCigarOff], C->cigar_off, 8 * (C->n_overlaps + 1), st)) || (rc = B[kBCigar].reserve(cig_bytes + 16)) || (rc = upload_staged(B[kBCigar], C->cigar, cig_bytes, st)) ||
            (rc = upload_vec(B[kBQStart], C->q_start, 4 * C->n_overlaps, st)) || (rc = upload_vec(B[kBTBegin], C->t_begin, 4 * C->n_overlaps, st)) ||
            (rc = upload_vec(B[kBTEnd], C->t_end, 4 * C->n_overlaps, st))) { drop_events(); return rc; }
        HIP_TRY(hipMemsetAsync(B[kBBpT].p, 0, 4 * n_points + 16, st));
        HIP_TRY(hipMemsetAsync(B[kBBpQ].p, 0, 4 * n_points + 16, st));
    }
    // per-pair work arrays: key, val, key_sorted, val_sorted, begin, end, q0, len, ovl
    const uint64_t np_al = (n_pairs + 63) & ~63ull;
    if ((rc = B[kBPairs].reserve(std::max<uint64_t>(9 * 4 * np_al, 256))) || (rc = B[kBMisc].reserve(4ull * (nw + 2) + 64)) ||
        (rc = e->d_win_seq_off.reserve(4ull * (nw + 1))) || (rc = e->d_win_type.reserve(nw))) { drop_events(); return rc; }
    uint32_t* pw = B[kBPairs].as<uint32_t>();
    uint32_t* d_err = B[kBMisc].as<uint32_t>();                       // [0] errors, [1..8] symbol bits, [16..] unused
    uint32_t* d_cnt = e->d_win_seq_off.as<uint32_t>();                 // counts, then scanned in place
    HIP_TRY(hipMemsetAsync(d_err, 0, 64, st));
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 4ull * (nw + 1), st));
    HIP_TRY(hipEventRecord(ev[1], st));

    BuildParams P{};
    P.seq_off = B[kBReadOff].as<uint64_t>(); P.bases = B[kBReadBases].as<uint8_t>(); P.quals = B[kBReadQuals].as<uint8_t>();
    P.has_qual = B[kBReadHasQual].as<uint8_t>(); P.n_targets = static_cast<uint32_t>(R.n_targets);
    P.q_id = B[kBQid].as<uint32_t>(); P.t_id = B[kBTid].as<uint32_t>(); P.strand = B[kBStrand].as<uint8_t>();
    P.bp_off = B[kBBpOff].as<uint64_t>(); P.bp_t = B[kBBpT].as<uint32_t>(); P.bp_q = B[kBBpQ].as<uint32_t>();
    P.n_overlaps = O.n_overlaps; P.n_pairs = n_pairs;
    P.first_window = B[kBFirstWin].as<uint32_t>(); P.n_windows = nw; P.W = W; P.window_type = window_type; P.qthr = qthr;
    P.key = pw; P.val = pw + np_al; uint32_t* key_sorted = pw + 2 * np_al; uint32_t* val_sorted = pw + 3 * np_al;
    P.pair_begin = pw + 4 * np_al; P.pair_end = pw + 5 * np_al; P.pair_q0 = pw + 6 * np_al; P.pair_len = pw + 7 * np_al; P.pair_ovl = pw + 8 * np_al;
    P.win_cnt = d_cnt; P.err = d_err;
    if (C && C->n_overlaps) {
        CigarParams K{};
        K.cigar_off = B[kBCigarOff].as<uint64_t>(); K.cigar = B[kBCigar].as<uint8_t>(); K.q_start = B[kBQStart].as<uint32_t>();
        K.t_begin = B[kBTBegin].as<uint32_t>(); K.t_end = B[kBTEnd].as<uint32_t>(); K.bp_off = P.bp_off;
        K.bp_t = B[kBBpT].as<uint32_t>(); K.bp_q = B[kBBpQ].as<uint32_t>(); K.n_overlaps = C->n_overlaps; K.W = W;
        if (e->knobs.cigar_serial) {             // (test switch) the per-thread restatement (kept: it is the reference's loop, operation by operation)
            hipLaunchKernelGGL(k_cigar_breaking_points, dim3(static_cast<uint32_t>((C->n_overlaps + 255) / 256)), dim3(256), 0, st, K);
        } else {
            const uint64_t n_slots_all = n_points / 2;
            if ((rc = B[kBKeyFirst].reserve(8 * n_slots_all + 16)) || (rc = B[kBKeyLast].reserve(8 * n_slots_all + 16))) { drop_events(); return rc; }
            HIP_TRY(hipMemsetAsync(B[kBKeyFirst].p, 0xff, 8 * n_slots_all + 16, st));
            HIP_TRY(hipMemsetAsync(B[kBKeyLast].p, 0, 8 * n_slots_all + 16, st));
            CigarWaveParams KW{K, B[kBKeyFirst].as<unsigned long long>(), B[kBKeyLast].as<unsigned long long>()};
            hipLaunchKernelGGL(k_cigar_breaking_points_wave, dim3(static_cast<uint32_t>((C->n_overlaps + 3) / 4)), dim3(256), 0, st, KW);
        }
    }
    if (S && O.n_overlaps) {
        const uint64_t n_slots_all = n_points / 2;
        if ((rc = B[kBKeyFirst].reserve(8 * n_slots_all + 16)) || (rc = B[kBKeyLast].reserve(8 * n_slots_all + 16))) { drop_events(); return rc; }
        HIP_TRY(hipMemsetAsync(B[kBKeyFirst].p, 0xff, 8 * n_slots_all + 16, st));
        HIP_TRY(hipMemsetAsync(B[kBKeyLast].p, 0, 8 * n_slots_all + 16, st));
        OpsWalkParams K{};
        K.ops = S->d_ops; K.ops_off = S->d_ops_off; K.q_start = B[kBQStart].as<uint32_t>(); K.t_begin = B[kBTBegin].as<uint32_t>();
        K.t_end = B[kBTEnd].as<uint32_t>(); K.bp_off = P.bp_off; K.bp_t = B[kBBpT].as<uint32_t>(); K.bp_q = B[kBBpQ].as<uint32_t>();
        K.first_key = B[kBKeyFirst].as<unsigned long long>(); K.last_key = B[kBKeyLast].as<unsigned long long>();
        K.n_overlaps = O.n_overlaps; K.W = W;
        hipLaunchKernelGGL(k_ops_breaking_points, dim3(static_cast<uint32_t>((O.n_overlaps + 3) / 4)), dim3(256), 0, st, K);
    }
    hipLaunchKernelGGL(k_symbols, dim3(static_cast<uint32_t>(std::min<uint64_t>(1024, (read_bytes + 1023) / 1024 + 1))), dim3(256), 0, st, P.bases, read_bytes, d_err + 1);
    if (n_pairs) {
        hipLaunchKernelGGL(k_pair_owner, dim3(static_cast<uint32_t>((O.n_overlaps + 255) / 256)), dim3(256), 0, st, P);
        hipLaunchKernelGGL(k_layer_filter, dim3(static_cast<uint32_t>((n_pairs + 3) / 4)), dim3(256), 0, st, P);
    }
    HIP_TRY(hipGetLastError());

    // stable sort of the pairs by window id (dropped pairs carry key nw and end up behind)
    int end_bit = 1; while ((1ull << end_bit) <= nw) ++end_bit;
    size_t temp_sort = 0, temp_scan32 = 0, temp_scan64 = 0;
    if (n_pairs) HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_sort, P.key, key_sorted, P.val, val_sorted, static_cast<int>(n_pairs), 0, end_bit, st));
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_scan32, d_cnt, d_cnt, static_cast<int>(nw + 1), st));
    if ((rc = B[kBTemp].reserve(std::max(temp_sort, temp_scan32) + 256))) { drop_events(); return rc; }
    if (n_pairs) HIP_TRY(hipcub::DeviceRadixSort::SortPairs(B[kBTemp].p, temp_sort, P.key, key_sorted, P.val, val_sorted, static_cast<int>(n_pairs), 0, end_bit, st));
    hipLaunchKernelGGL(k_plus_one, dim3((nw + 1 + 255) / 256), dim3(256), 0, st, d_cnt, nw);
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(B[kBTemp].p, temp_scan32, d_cnt, d_cnt, static_cast<int>(nw + 1), st));
    uint32_t ns = 0, h_err[16] = {0};
    HIP_TRY(hipMemcpyAsync(&ns, d_cnt + nw, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_err, d_err, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_err[0]) { drop_events(); return RCN_E_LAYER; }     // "[racon::Window::add_layer] error: layer begin and end positions are invalid!"
    const uint64_t n_layers = ns - nw;

    // sequence table + offsets
    if ((rc = B[kBSeqLen].reserve(8ull * (ns + 1))) || (rc = B[kBSeqSrc].reserve(9ull * ns + 64)) || (rc = e->d_seq_off.reserve(8ull * (ns + 1))) ||
        (rc = e->d_has_qual.reserve(ns)) || (rc = e->d_begin.reserve(4ull * ns)) || (rc = e->d_end.reserve(4ull * ns))) { drop_events(); return rc; }
    SeqTable T{};
    T.win_seq_off = d_cnt; T.key_sorted = key_sorted; T.val_sorted = val_sorted; T.n_layers = n_layers;
    T.seq_len = B[kBSeqLen].as<uint64_t>(); T.begin = e->d_begin.as<uint32_t>(); T.end = e->d_end.as<uint32_t>();
    T.has_qual = e->d_has_qual.as<uint8_t>(); T.win_type = e->d_win_type.as<uint8_t>();
    T.src_id = B[kBSeqSrc].as<uint32_t>(); T.src_pos = T.src_id + ns; T.src_flags = reinterpret_cast<uint8_t*>(T.src_pos + ns);
    const uint64_t n_tab = std::max<uint64_t>(nw, n_layers);
    hipLaunchKernelGGL(k_seq_table, dim3(static_cast<uint32_t>((n_tab + 255) / 256)), dim3(256), 0, st, P, T);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_scan64, T.seq_len, e->d_seq_off.as<uint64_t>(), static_cast<int>(ns + 1), st));
    if ((rc = B[kBTemp].reserve(temp_scan64 + 256))) { drop_events(); return rc; }
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(B[kBTemp].p, temp_scan64, T.seq_len, e->d_seq_off.as<uint64_t>(), static_cast<int>(ns + 1), st));
    uint64_t n_bases = 0;
    HIP_TRY(hipMemcpyAsync(&n_bases, e->d_seq_off.as<uint64_t>() + ns, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if ((rc = e->d_bases.reserve(n_bases + 16)) || (rc = e->d_quals.reserve(n_bases + 16))) { drop_events(); return rc; }

    // gather
    GatherParams G{};
    G.read_off = P.seq_off; G.read_bases = P.bases; G.read_quals = P.quals;
    G.seq_off = e->d_seq_off.as<uint64_t>(); G.src_id = T.src_id; G.src_pos = T.src_pos; G.src_flags = T.src_flags; G.has_qual = T.has_qual;
    G.bases = e->d_bases.as<uint8_t>(); G.quals = e->d_quals.as<uint8_t>(); G.n_seqs = ns;
    HIP_TRY(hipEventRecord(ev[2], st));
    hipLaunchKernelGGL(k_gather, dim3((ns + 256 / kGatherLanes - 1) / (256 / kGatherLanes)), dim3(256), 0, st, G);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev[3], st));

    // metadata back to the host for the layer order / capacities (17 bytes per sequence), then the usual preparation
    std::vector<uint32_t> h_wso(nw + 1), h_begin(ns), h_end(ns);
    std::vector<uint64_t> h_seq_off(ns + 1);
    HIP_TRY(hipMemcpyAsync(h_wso.data(), d_cnt, 4ull * (nw + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_seq_off.data(), e->d_seq_off.p, 8ull * (ns + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_begin.data(), e->d_begin.p, 4ull * ns, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_end.data(), e->d_end.p, 4ull * ns, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_err, d_err, 64, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // the reverse strand turns A/C/G/T into T/G/C/A: close the set under the complement before counting
    for (const auto& pr : {std::pair<int, int>('A', 'T'), std::pair<int, int>('C', 'G')}) {
        const bool has = ((h_err[1 + (pr.first >> 5)] >> (pr.first & 31)) & 1u) || ((h_err[1 + (pr.second >> 5)] >> (pr.second & 31)) & 1u);
        if (has && O.n_overlaps) { h_err[1 + (pr.first >> 5)] |= 1u << (pr.first & 31); h_err[1 + (pr.second >> 5)] |= 1u << (pr.second & 31); }
    }
    int32_t nsym = 0;
    for (int k = 1; k <= 8; ++k) nsym += __builtin_popcount(h_err[k]);
    e->n_windows = nw; e->n_seqs = ns; e->n_bases = n_bases;
    bool acgt_all = true;
    for (int k = 1; k <= 8; ++k) {
        uint32_t allowed = 0;
        for (int c : {'A', 'C', 'G', 'T'}) if ((c >> 5) == k - 1) allowed |= 1u << (c & 31);
        if (h_err[k] & ~allowed) acgt_all = false;
    }
    if ((rc = prepare_resident(e, nw, ns, h_wso.data(), h_seq_off.data(), h_begin.data(), h_end.data(), nullptr, std::max(nsym, 1), acgt_all))) { drop_events(); return rc; }

    float ms_h2d = 0, ms_k = 0, ms_g = 0;
    HIP_TRY(hipEventElapsedTime(&ms_h2d, ev[0], ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms_k, ev[1], ev[3]));
    HIP_TRY(hipEventElapsedTime(&ms_g, ev[2], ev[3]));
    drop_events();
    e->bstats.h2d_ms = ms_h2d; e->bstats.kernel_ms = ms_k; e->bstats.gather_ms = ms_g;
    e->bstats.n_pairs = n_pairs; e->bstats.n_layers = n_layers;
    e->bstats.gather_bytes = 4 * n_bases;                     // bases + qualities, each read once and written once
    e->stats = rcn_run_stats{};
    e->stats.h2d_ms = ms_h2d;
    e->stats.bytes_in = 2 * read_bytes + 17ull * n_points;
    e->uploaded = true;
    return RCN_OK;
}

// Breaking points from the alignments on the device, then the same construction.  One slot (two points) per window end an
// overlap touches: the multiples of W strictly inside (t_begin, t_end) and t_end itself (reference src/overlap.cpp:228-236).
inline int build_windows_from_cigars(rcn_engine* e, const rcn_read_set& R, const rcn_cigar_set& C, uint32_t W, double qthr, uint8_t window_type) {
    if (C.n_overlaps && (!C.q_id || !C.t_id || !C.strand || !C.q_start || !C.t_begin || !C.t_end || !C.cigar_off || !C.cigar)) return RCN_E_ARG;
    std::vector<uint64_t> bp_off(C.n_overlaps + 1, 0);
    for (uint64_t o = 0; o < C.n_overlaps; ++o) {
        const uint64_t tb = C.t_begin[o], te = C.t_end[o];
        if (te <= tb) return RCN_E_ARG;
        const uint64_t inside = (te - 1) / W - tb / W;            // multiples k*W with t_begin < k*W < t_end
        bp_off[o + 1] = bp_off[o] + 2 * (inside + 1);
    }
    rcn_overlap_set O{};
    O.n_overlaps = C.n_overlaps; O.q_id = C.q_id; O.t_id = C.t_id; O.strand = C.strand; O.bp_off = bp_off.data();
    return build_windows(e, R, O, W, qthr, window_type, &C);
}

// ---- exact pairwise alignment of overlaps without a CIGAR (pair_align.hpp) ----
enum { kAQPos, kATPos, kAQLen, kATLen, kAQRc, kAOrder, kAOps, kAOpsOff, kADist, kAScratch, kACtr, kAlignSlots };

// Aligns every pair; op bytes and distances stay resident (e->d_align).  `reads_resident`: upload_reads was done.
inline int align_pairs(rcn_engine* e, const rcn_read_set& R, const rcn_pair_set& S, bool reads_resident) {
    if (R.n_seqs == 0 || !R.seq_off || !R.bases || !R.quals || !R.seq_has_qual) return RCN_E_ARG;
    if (S.n_pairs && (!S.q_id || !S.t_id || !S.strand || !S.q_begin || !S.q_end || !S.t_begin || !S.t_end)) return RCN_E_ARG;
    if (S.n_pairs > 0xfffffffeull) return RCN_E_ARG;
    HIP_TRY(hipSetDevice(e->cfg.device));
    hipStream_t st = e->stream;
    const uint64_t n = S.n_pairs;
    e->astats = rcn_align_stats{};
    e->a_n_pairs = 0;
    EventPair tc, tk;
    if (tc.create() || tk.create()) return RCN_E_HIP;
    std::vector<uint64_t> q_pos(n), t_pos(n);
    std::vector<uint32_t> q_len(n), t_len(n), order(n);
    e->a_ops_off.assign(n + 1, 0);
    uint64_t m_cap = 1, n_cap = 1, cells = 0;
    for (uint64_t o = 0; o < n; ++o) {
        const uint64_t q = S.q_id[o], t = S.t_id[o];
        if (q >= R.n_seqs || t >= R.n_seqs || S.q_end[o] < S.q_begin[o] || S.t_end[o] < S.t_begin[o]) return RCN_E_ARG;
        const uint64_t ql = R.seq_off[q + 1] - R.seq_off[q], tl = R.seq_off[t + 1] - R.seq_off[t];
        if (S.q_end[o] > ql || S.t_end[o] > tl) return RCN_E_ARG;
        q_pos[o] = R.seq_off[q] + S.q_begin[o]; t_pos[o] = R.seq_off[t] + S.t_begin[o];
        q_len[o] = S.q_end[o] - S.q_begin[o]; t_len[o] = S.t_end[o] - S.t_begin[o];
        // (a one-column problem must be a leaf of the reference rule, or the target-axis split would not shrink it: 20 * ceil(rows / 64) + 8 < 2^20)
        if (q_len[o] > 3000000u || t_len[o] > 0x3fffffffu) return RCN_E_CAPACITY;
        e->a_ops_off[o + 1] = e->a_ops_off[o] + q_len[o] + t_len[o];
        m_cap = std::max<uint64_t>(m_cap, q_len[o]); n_cap = std::max<uint64_t>(n_cap, t_len[o]);
        cells += static_cast<uint64_t>(q_len[o]) * t_len[o];
        order[o] = static_cast<uint32_t>(o);
    }
    // largest problems first: a launch ends with its slowest overlap
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return static_cast<uint64_t>(q_len[a]) * t_len[a] > static_cast<uint64_t>(q_len[b]) * t_len[b]; });
    m_cap = (m_cap + 15) & ~uint64_t(15); n_cap = (n_cap + 15) & ~uint64_t(15);       // keeps the pieces of a slot 16-byte aligned
    const uint64_t ops_bytes = e->a_ops_off[n];
    // per-team scratch (two waves): two last-column vectors, two carry buffers and a leaf store per wave
    const uint64_t leaf_bytes = rcn::pair_leaf_bytes(m_cap);
    const uint64_t slot_bytes = rcn::pair_slot_bytes(m_cap, n_cap);
    DevBuf* A = e->d_align;
    HIP_TRY(hipEventRecord(tc.a, st));
    int rc;
    if (!reads_resident && (rc = upload_reads(e, R, st))) return rc;
    if ((rc = upload_vec(A[kAQPos], q_pos.data(), 8 * n, st)) || (rc = upload_vec(A[kATPos], t_pos.data(), 8 * n, st)) ||
        (rc = upload_vec(A[kAQLen], q_len.data(), 4 * n, st)) || (rc = upload_vec(A[kATLen], t_len.data(), 4 * n, st)) ||
        (rc = upload_vec(A[kAQRc], S.strand, n, st)) || (rc = upload_vec(A[kAOrder], order.data(), 4 * n, st)) ||
        (rc = upload_vec(A[kAOpsOff], e->a_ops_off.data(), 8 * (n + 1), st)) || (rc = A[kAOps].reserve(ops_bytes + 256)) ||
        (rc = A[kADist].reserve(4 * n + 16)) || (rc = A[kACtr].reserve(64))) return rc;
    HIP_TRY(hipMemsetAsync(A[kAOps].p, 0, ops_bytes + 256, st));
    HIP_TRY(hipMemsetAsync(A[kACtr].p, 0, 64, st));
    size_t fr = 0, tot = 0;
    HIP_TRY(hipMemGetInfo(&fr, &tot));
    // (80 % of what is free now, within the caller's arena: see scratch_budget)
    const uint64_t free_now = static_cast<uint64_t>((fr + A[kAScratch].cap) * 0.8);
    const uint64_t budget = e->cfg.arena_bytes ? std::min<uint64_t>(e->cfg.arena_bytes, free_now) : free_now;
    uint64_t slots = std::min<uint64_t>(std::max<uint64_t>(n, 1), static_cast<uint64_t>(e->n_cu) * 8);      // teams of two waves, 117 VGPRs: four waves per SIMD are resident
    while (slots > 1 && slots * slot_bytes > budget) slots = (slots + 1) / 2;
    if (slots * slot_bytes > budget) return RCN_E_CAPACITY;
    if ((rc = A[kAScratch].reserve(slots * slot_bytes))) return rc;
    HIP_TRY(hipEventRecord(tc.b, st));
    if (n) {
        rcn::PairParams P{};
        P.bases = e->d_build[kBReadBases].as<uint8_t>();
        P.q_pos = A[kAQPos].as<uint64_t>(); P.t_pos = A[kATPos].as<uint64_t>(); P.q_len = A[kAQLen].as<uint32_t>(); P.t_len = A[kATLen].as<uint32_t>();
        P.q_rc = A[kAQRc].as<uint8_t>(); P.order = A[kAOrder].as<uint32_t>(); P.n_pairs = static_cast<uint32_t>(n);
        P.next = A[kACtr].as<unsigned int>(); P.err = A[kACtr].as<uint32_t>() + 4;
        P.ops = A[kAOps].as<uint8_t>(); P.ops_off = A[kAOpsOff].as<uint64_t>(); P.dist = A[kADist].as<int32_t>();
        P.scratch = A[kAScratch].as<uint8_t>(); P.slot_bytes = slot_bytes;
        P.m_cap = static_cast<uint32_t>(m_cap); P.n_cap = static_cast<uint32_t>(n_cap); P.leaf_bytes = leaf_bytes;
        P.arena_ints = static_cast<uint32_t>(rcn::pair_arena_ints(m_cap));
        HIP_TRY(hipEventRecord(tk.a, st));
        hipLaunchKernelGGL(rcn::k_pair_align, dim3(static_cast<uint32_t>(slots)), dim3(rcn::kPairThreads), 0, st, P);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(tk.b, st));
    }
    uint32_t h_ctr[16] = {0};
    HIP_TRY(hipMemcpyAsync(h_ctr, A[kACtr].p, 64, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_ctr[4]) { fprintf(stderr, "[racon_hip] pairwise alignment: internal error on %u overlap(s)\n", h_ctr[4]); return RCN_E_STATE; }
#ifdef RCN_PROF_PAIR
    { unsigned long long pp[16]; HIP_TRY(hipMemcpyFromSymbol(pp, HIP_SYMBOL(rcn::g_pairprof), sizeof(pp)));
      const double tot = static_cast<double>(std::max(1ull, pp[7]));
      fprintf(stderr, "[racon_hip] pair aligner, wave clocks since load: steady blocks %.1f %% (%.1f clocks per step, %.1f of 64 lanes hold a word), ramps / tails / snapshot blocks %.1f %% (%.1f per step), "
                      "row load %.1f %%, scores of a pass %.1f %%, leaf walk %.1f %% (%.0f clocks per leaf), cut %.1f %% (%.0f each), symbol set-up %.1f %%, barriers %.1f %%; %llu passes, %llu leaves\n",
              100.0 * pp[0] / tot, static_cast<double>(pp[0]) / std::max(1ull, pp[9]), static_cast<double>(pp[14]) / std::max(1ull, pp[9]), 100.0 * pp[1] / tot, static_cast<double>(pp[1]) / std::max(1ull, pp[10]),
              100.0 * pp[2] / tot, 100.0 * pp[3] / tot, 100.0 * pp[4] / tot, static_cast<double>(pp[4]) / std::max(1ull, pp[12]), 100.0 * pp[5] / tot, static_cast<double>(pp[5]) / std::max(1ull, pp[13]),
              100.0 * pp[6] / tot, 100.0 * pp[8] / tot, pp[11], pp[12]); }
#endif
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, tc.a, tc.b)); e->astats.h2d_ms = ms;
    if (n) { HIP_TRY(hipEventElapsedTime(&ms, tk.a, tk.b)); e->astats.kernel_ms = ms; }
    e->astats.n_pairs = n; e->astats.cells = cells; e->astats.ops_bytes = ops_bytes; e->astats.slots = static_cast<uint32_t>(slots);
    e->a_n_pairs = n;
    return RCN_OK;
}

// Alignment + breaking points + window construction, all in HBM.
inline int build_windows_from_pairs(rcn_engine* e, const rcn_read_set& R, const rcn_pair_set& S, uint32_t W, double qthr, uint8_t window_type) {
    if (W == 0) return RCN_E_ARG;
    HIP_TRY(hipSetDevice(e->cfg.device));
    int rc;
    if (R.n_seqs == 0 || R.n_targets == 0 || R.n_targets > R.n_seqs || !R.seq_off || !R.bases || !R.quals || !R.seq_has_qual) return RCN_E_ARG;
    if ((rc = upload_reads(e, R, e->stream))) return rc;
    // an aligner that found no room (RCN_E_NOMEM / RCN_E_CAPACITY: the caller then aligns on the host and comes back through the
    // CIGAR path) must not leave its pair tables and op bytes -- one byte per row + column of every overlap -- behind
    auto drop_align = [&]() { for (DevBuf& d : e->d_align) d.release(); e->a_n_pairs = 0; e->a_ops_off.clear(); };
    const auto t_al0 = std::chrono::steady_clock::now();
    if ((rc = align_pairs(e, R, S, true))) { drop_align(); return rc; }
    if (e->knobs.debug) fprintf(stderr, "[racon_hip] pairs: %lu overlaps (%.3g matrix cells) aligned in %.1f ms (kernel %.1f ms = %.1f TCUPS, copies %.1f ms, %u teams)\n",
                                static_cast<unsigned long>(S.n_pairs), static_cast<double>(e->astats.cells), 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_al0).count(),
                                e->astats.kernel_ms, static_cast<double>(e->astats.cells) / (1e9 * std::max(1e-6, static_cast<double>(e->astats.kernel_ms))), e->astats.h2d_ms, e->astats.slots);
    e->d_align[kAScratch].release();            // the aligner's per-wave scratch (sized for the longest read) is done with
    std::vector<uint64_t> bp_off(S.n_pairs + 1, 0);
    std::vector<uint32_t> q_start(S.n_pairs);
    for (uint64_t o = 0; o < S.n_pairs; ++o) {
        const uint64_t tb = S.t_begin[o], te = S.t_end[o];
        if (te <= tb) { drop_align(); return RCN_E_ARG; }
        const uint64_t inside = (te - 1) / W - tb / W;
        bp_off[o + 1] = bp_off[o] + 2 * (inside + 1);
        const uint64_t ql = R.seq_off[S.q_id[o] + 1] - R.seq_off[S.q_id[o]];
        q_start[o] = S.strand[o] ? static_cast<uint32_t>(ql - S.q_end[o]) : S.q_begin[o];          // reference src/overlap.cpp:241-242
    }
    rcn_overlap_set O{};
    O.n_overlaps = S.n_pairs; O.q_id = S.q_id; O.t_id = S.t_id; O.strand = S.strand; O.bp_off = bp_off.data();
    OpsSource src{e->d_align[kAOps].as<uint8_t>(), e->d_align[kAOpsOff].as<uint64_t>(), q_start.data(), S.t_begin, S.t_end};
    const rcn_align_stats keep = e->astats;
    const auto t_bw0 = std::chrono::steady_clock::now();
    rc = build_windows(e, R, O, W, qthr, window_type, nullptr, &src, true);
    e->astats = keep;
    if (e->knobs.debug) fprintf(stderr, "[racon_hip] pairs: breaking points and windows in %.1f ms\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_bw0).count());
    // the paths have been walked into breaking points: their op bytes (one per row + column of every overlap) must not
    // stay allocated through the consensus run (rcn_engine_alignment_cigars then reports RCN_E_STATE)
    for (int k : {kAOps, kAOpsOff, kAQPos, kATPos, kAQLen, kATLen, kAQRc, kAOrder}) e->d_align[k].release();
    e->a_n_pairs = 0; e->a_ops_off.clear();
    return rc;
}

}  // namespace rcn
