// pair_cell.hpp — one cell of the bit-vector recurrence of pair_align.hpp, written for the gfx950 VALU (included by pair_align.hpp;
// compiled for the CPU as well by tests/emul/pair_cell_main.cpp, which checks it against the textbook form bit for bit).
//
// The recurrence is Myers' / Hyyro's: a 64-row word of vertical deltas (Pv, Mv) against one column symbol, horizontal carry in from
// the word above, horizontal carry out for the word below.  The kernel is bound by vector instruction issue (four waves per SIMD,
// no HBM-bound phase), so the cell is written in the operations the hardware has, not in 64-bit C:
//   * every three-input Boolean step is ONE v_bitop3_b32 per 32-bit half (the compiler found two of the seven on its own);
//   * the horizontal plus-word is kept COMPLEMENTED (nPh = ~Ph): the two uses after the shift take either polarity for free inside
//     a v_bitop3, the carry out is then "bit 31 of the high half" of nPh and of Mh as they stand -- no extraction --, the lane
//     above hands both registers over whole (two DPP shifts whose zero fill at lane 0 IS the top boundary: +1 in, -0 in), and
//     "(word << 1) | carry" is a v_alignbit_b32 per half with that register as the low operand;
//   * the 2 x 64 carry bits a multi-pass problem writes per block are collected by one v_alignbit per step and polarity.
// 27 vector instructions per cell with two symbol planes (ACGT reads) + 5 for the lane-to-lane shifts, against 47.5 per step of
// the 64-bit C form (profiles/r06/e_pair_cell_isa.txt).
#pragma once
#include <cstdint>

#if defined(__HIPCC__) || defined(__HIP__)
#define RCN_PC_HD __host__ __device__ __forceinline__
#else
#define RCN_PC_HD inline
#endif

namespace rcn {

// truth table of a three-input Boolean function for v_bitop3_b32: evaluate it on these three constants
constexpr unsigned kTA = 0xF0u, kTB = 0xCCu, kTC = 0xAAu;
#define RCN_TT(expr) static_cast<unsigned>((expr) & 0xFFu)

template <unsigned TT>
RCN_PC_HD uint32_t pc_bitop3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<uint32_t>(__builtin_amdgcn_bitop3_b32(static_cast<int>(a), static_cast<int>(b), static_cast<int>(c), TT));
#else
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) {
        if (!((TT >> i) & 1u)) continue;
        r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    }
    return r;
#endif
}
// ({hi, lo} >> sh) & 0xffffffff, sh in 0..31
RCN_PC_HD uint32_t pc_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> sh);
#endif
}

template <int NPL>
struct PairLane {                    // what a lane keeps of its word between the steps of a pass, in 32-bit halves
    uint32_t pl[NPL], ph[NPL];       // bit planes of the rows' symbol codes
    uint32_t vl, vh;                 // rows that exist
    uint32_t Pvl, Pvh, Mvl, Mvh;     // vertical deltas
};

// What the cell hands down / gets from above: bit 31 of `np` = NOT (plus carry), bit 31 of `mn` = minus carry; the other bits are
// whatever the words held.  The top boundary (+1 per column) is {0, 0}.
struct PairCarry { uint32_t np, mn; };

// The column's symbol code as the cell takes it: one 0 / ~0 mask per plane (bit k of the code).  The steady state of a pass reads
// them ready-made from LDS (pair_align.hpp), the ramps make them with one sign-extending bit-field extract per plane.
template <int NPL>
struct PairSym { uint32_t mk[NPL]; };
template <int NPL>
RCN_PC_HD PairSym<NPL> pair_sym_of(int tc) {
    PairSym<NPL> y;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < NPL; ++k) y.mk[k] = static_cast<uint32_t>((tc << (31 - k)) >> 31);
    return y;
}

// One cell: the lane's word against the column with the plane masks `y`, carry `cin` from the word above.
// Returns the carry for the word below; nph_l / nph_h = the COMPLEMENT of the word's horizontal plus-deltas before the shift (a
// leaf stores them for its traceback).
template <int NPL>
RCN_PC_HD PairCarry pair_cell(PairLane<NPL>& L, const PairSym<NPL>& y, PairCarry cin, uint32_t& nph_l, uint32_t& nph_h) {
    // Eq: rows whose code agrees with the column's in every plane
    uint32_t el = L.vl, eh = L.vh;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < NPL; ++k) {
        el = pc_bitop3<RCN_TT(kTA & ~(kTB ^ kTC))>(el, L.pl[k], y.mk[k]);
        eh = pc_bitop3<RCN_TT(kTA & ~(kTB ^ kTC))>(eh, L.ph[k], y.mk[k]);
    }
    const uint32_t hn = cin.mn >> 31;
    const uint32_t xvl = el | L.Mvl, xvh = eh | L.Mvh;                                      // Xv = Eq | Mv
    el |= hn;                                                                               // Eq |= carry-in minus
    const uint32_t tl = el & L.Pvl, th = eh & L.Pvh;
    const uint64_t pv = (static_cast<uint64_t>(L.Pvh) << 32) | L.Pvl;
    const uint64_t sum = ((static_cast<uint64_t>(th) << 32) | tl) + pv;
    const uint32_t sl = static_cast<uint32_t>(sum), sh = static_cast<uint32_t>(sum >> 32);
    const uint32_t xhl = pc_bitop3<RCN_TT((kTA ^ kTB) | kTC)>(sl, L.Pvl, el);               // Xh = ((Eq & Pv) + Pv) ^ Pv | Eq
    const uint32_t xhh = pc_bitop3<RCN_TT((kTA ^ kTB) | kTC)>(sh, L.Pvh, eh);
    nph_l = pc_bitop3<RCN_TT(~kTA & (kTB | kTC))>(L.Mvl, xhl, L.Pvl);                       // ~Ph = ~(Mv | ~(Xh | Pv))
    nph_h = pc_bitop3<RCN_TT(~kTA & (kTB | kTC))>(L.Mvh, xhh, L.Pvh);
    const uint32_t mhl = L.Pvl & xhl, mhh = L.Pvh & xhh;                                    // Mh = Pv & Xh
    const PairCarry out{nph_h, mhh};
    const uint32_t npl = pc_alignbit(nph_l, cin.np, 31), nphh = pc_alignbit(nph_h, nph_l, 31);    // ~((Ph << 1) | carry-in plus)
    const uint32_t msl = pc_alignbit(mhl, cin.mn, 31), msh = pc_alignbit(mhh, mhl, 31);           // (Mh << 1) | carry-in minus
    L.Pvl = pc_bitop3<RCN_TT(kTA | (~kTB & kTC))>(msl, xvl, npl);                           // Pv = Mh | ~(Xv | Ph)
    L.Pvh = pc_bitop3<RCN_TT(kTA | (~kTB & kTC))>(msh, xvh, nphh);
    L.Mvl = xvl & ~npl;                                                                     // Mv = Ph & Xv
    L.Mvh = xvh & ~nphh;
    return out;
}

}  // namespace rcn
