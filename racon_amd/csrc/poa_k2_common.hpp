// poa_k2_common.hpp -- execution policies of the four-wave work-group, packed int16 helpers, DP shapes
// Part of the fast path of the MI355X window-consensus engine: included by poa_kernel2.hpp, in this order, into one
// translation unit (see its header for the design).
#pragma once

namespace rcn {


constexpr int kWaves2 = 4;
constexpr int kThreads2 = 64 * kWaves2;
constexpr int kNeg16 = -32000;
constexpr int kZLimit = 31000;          // |Z| bound accepted for the int16 path

// ---- execution policies (see OneWaveBlock in poa_kernel.hpp) ----
__device__ __forceinline__ int* lds_words2() { extern __shared__ int4 lds_dyn[]; return reinterpret_cast<int*>(lds_dyn); }
__device__ __forceinline__ Ctx* ctx_lds2() { return reinterpret_cast<Ctx*>(lds_words2() + kLdsBytes / 4); }
struct Wave0Of4 {            // wave 0 of the 4-wave work-group, the other waves wait at the next Block4::sync()
    static constexpr int NT = 64;
    static __device__ __forceinline__ int tid() { return threadIdx.x; }
    static __device__ __forceinline__ void sync() { __threadfence_block(); }
    static __device__ __forceinline__ Ctx* ctx() { return ctx_lds2(); }
    static __device__ __forceinline__ int* work() { return lds_words2(); }
};
struct Block4 {
    static constexpr int NT = kThreads2;
    static __device__ __forceinline__ int tid() { return threadIdx.x; }
    static __device__ __forceinline__ void sync() { __threadfence_block(); __syncthreads(); }
    static __device__ __forceinline__ Ctx* ctx() { return ctx_lds2(); }
    static __device__ __forceinline__ int* work() { return lds_words2(); }
};

// one LDS word, read now (polling loops); invisible to the compiler's memory model on purpose
__device__ __forceinline__ uint32_t lds_poll(const uint32_t* p) {
    uint32_t v;
    const uint32_t a = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));      // LDS addresses are the low 32 bits
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
    return v;
}
// LDS-only barrier: waits for this wave's LDS traffic, NOT for its outstanding HBM stores
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- packed int16 helpers (two cells per VGPR: low half = even column) ----
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, static_cast<s16x2>(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pack2(int lo, int hi) { return (static_cast<uint32_t>(lo) & 0xffffu) | (static_cast<uint32_t>(hi) << 16); }
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// (sym == seq ? m - g : x - g) for both halves: t = min(seq ^ sym, 1); t * (x - m) + (m - g).  The empty asm
// keeps the compiler from turning min(a ^ b, 1) back into compare + select chains (five instructions).
__device__ __forceinline__ uint32_t pk_profile(uint32_t sqx, uint32_t symsym, uint32_t one, uint32_t xm, uint32_t mg) {
    uint32_t t = sqx ^ symsym;
    asm("" : "+v"(t));
    const u16x2 mn = __builtin_elementwise_min(__builtin_bit_cast(u16x2, t), __builtin_bit_cast(u16x2, one));
    uint32_t u = __builtin_bit_cast(uint32_t, mn);
    asm("" : "+v"(u));
    const s16x2 r = __builtin_bit_cast(s16x2, u) * __builtin_bit_cast(s16x2, xm) + __builtin_bit_cast(s16x2, mg);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_minu(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_bit_cast(uint32_t, static_cast<s16x2>(__builtin_bit_cast(s16x2, a) * __builtin_bit_cast(s16x2, b) + __builtin_bit_cast(s16x2, c)));
}
// One of the 3 * NP independent instructions of the next row's substitution profile (xor, min, mad per register).
// dp2_rows pins two of them between consecutive steps of the DPP prefix scan (scheduling barriers on both sides): a DPP
// read needs two wait states after the VALU write of its source, and every s_nop the compiler would otherwise put
// there costs the wave a full issue slot.
template <int NP, int O>
__device__ __forceinline__ void dp2_gap_op(uint32_t (&pw)[NP], const uint32_t (&sqx)[NP], uint32_t symsym, uint32_t one, uint32_t xm, uint32_t mg) {
    if constexpr (O < 3 * NP) {
        constexpr int q = O % NP, st = O / NP;
        if constexpr (st == 0) pw[q] = sqx[q] ^ symsym;
        else if constexpr (st == 1) pw[q] = pk_minu(pw[q], one);
        else pw[q] = pk_mad(pw[q], xm, mg);
    }
}
// Half broadcasts written as vector shuffles: instruction selection folds them into the VOP3P op_sel / op_sel_hi
// source modifiers of v_pk_max_i16 (no v_perm_b32 in front; inline asm would cost a hazard s_nop per use on gfx950).
// {lo, max(hi, lo)}
__device__ __forceinline__ uint32_t pk_chain_pair(uint32_t a) {
    const s16x2 av = __builtin_bit_cast(s16x2, a);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(av, __builtin_shufflevector(av, av, 0, 0)));
}
// {max(a.lo, b.hi), max(a.hi, b.hi)}
__device__ __forceinline__ uint32_t pk_max_bhi(uint32_t a, uint32_t b) {
    const s16x2 bv = __builtin_bit_cast(s16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_shufflevector(bv, bv, 1, 1)));
}
// {max(a.lo, b.lo), max(a.hi, b.lo)}
__device__ __forceinline__ uint32_t pk_max_blo(uint32_t a, uint32_t b) {
    const s16x2 bv = __builtin_bit_cast(s16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_shufflevector(bv, bv, 0, 0)));
}
__device__ __forceinline__ int wave_incl_scan_max_id(int v) {      // INT_MIN is max's identity: each step fuses into one v_max_i32_dpp
    constexpr int I = static_cast<int>(0x80000000u);
    v = max(v, dpp_or<0x111, 0xf>(I, v));
    v = max(v, dpp_or<0x112, 0xf>(I, v));
    v = max(v, dpp_or<0x114, 0xf>(I, v));
    v = max(v, dpp_or<0x118, 0xf>(I, v));
    v = max(v, dpp_or<0x142, 0xa>(I, v));
    v = max(v, dpp_or<0x143, 0xc>(I, v));
    return v;
}

// DP shape for a layer of `len` bases: one wave over (len+1) <= 512 columns, else the 4-wave pipeline;
// NP = packed VGPRs per lane (2 NP columns).  Returns NP | (WV << 8), 0 = not supported (int32 kernel).
// `wide` (heavy windows, see KParams::heavy_ns): the pipeline also for short layers -- twice the instructions
// in total but about half the latency per row, which is what counts for the windows that finish last.
__host__ __device__ __forceinline__ int dp2_cfg(int len, bool wide) {
    const int W = len + 1;
    if (W <= 512 && !wide) return ((W + 127) / 128) | (1 << 8);
    const int n = (W + 511) / 512;
    return n <= 4 ? (n | (4 << 8)) : 0;
}
// rows of the register window for NP packed VGPRs per lane (16 VGPRs in all; a power of two)
__host__ __device__ constexpr int dp2_window(int np) { return np <= 1 ? 16 : np == 2 ? 8 : 4; }
// rows of the LDS ring behind it (K of dp2_rows<NP, WV>)
// (tab: the one-wave DP keeps a 4-symbol substitution-profile table behind the ring, dp2_rows<NP, 1, true>)
__host__ __device__ constexpr int dp2_ring_rows(int np, int wv, bool tab = false) {
    return (kLdsBytes - 64 - (wv > 1 ? 64 * 4 * 4 + 64 : 0) - (tab ? 4 * 4 * 64 * wv * np : 0)) / (4 * 64 * wv * np) - 1;
}

}  // namespace rcn
