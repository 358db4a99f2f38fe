// What bounds a DP row of the banded one-wave DP (poa_band.hpp)?  One wave per work-group, one work-group per CU, a loop of
// N "rows" whose values feed the next iteration, clocks per iteration by s_memtime around the loop:
//   A  the chain row as it is: predecessor row -> diagonal shift across lanes -> candidates -> in-lane chain ->
//      six-step prefix maximum of the lane tails -> carry-in -> next row
//   B  "lazy carry": the row's pre-scan values are formed from the previous row's PRE-scan values and the previous carry
//      (max(a, z) + c = max(a + c, z + c)); only tail -> scan -> carry is on the row-to-row path
//   C  A without the scan (its other dependent operations only);  D  the scan alone
//   H  (round 6) the row as TWO HALF-ROW STREAMS in one wave: register 0 = columns [0, 128) two per lane, register 1 = columns
//      [128, 256) -- two independent six-step scans per row, coupled by one carried value (v_readlane of lane 63's running
//      maximum) and one diagonal cell; H1 = both halves of row i in one trip, H2 = skewed ((row i, half A) next to
//      (row i - 1, half B): nothing of one stream waits for the other), H3 = H2 + register window + LDS + scalar bookkeeping
//      (to be read against "A + all three")
//   U  "A + all three" with EIGHT rows per trip and static register-window places (what the octets of poa_band.hpp do)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/row_chain tools/probe/row_chain.hip && /tmp/row_chain
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, static_cast<s16x2>(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ uint32_t pk_chain_pair(uint32_t a) { const s16x2 av = __builtin_bit_cast(s16x2, a); return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(av, __builtin_shufflevector(av, av, 0, 0))); }
__device__ __forceinline__ uint32_t pk_max_bhi(uint32_t a, uint32_t b) { const s16x2 bv = __builtin_bit_cast(s16x2, b); return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_shufflevector(bv, bv, 1, 1))); }
__device__ __forceinline__ uint32_t pk_max_blo(uint32_t a, uint32_t b) { const s16x2 bv = __builtin_bit_cast(s16x2, b); return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_shufflevector(bv, bv, 0, 0))); }
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dpp_or(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROWMASK, 0xf, false); }
__device__ __forceinline__ int scan6(int sc) {
    constexpr int I = static_cast<int>(0x80000000u);
    sc = max(sc, dpp_or<0x111, 0xf>(I, sc)); sc = max(sc, dpp_or<0x112, 0xf>(I, sc)); sc = max(sc, dpp_or<0x114, 0xf>(I, sc));
    sc = max(sc, dpp_or<0x118, 0xf>(I, sc)); sc = max(sc, dpp_or<0x142, 0xa>(I, sc)); sc = max(sc, dpp_or<0x143, 0xc>(I, sc));
    return sc;
}
template <bool W, bool L, bool S>
__device__ __forceinline__ void rows_wls(int n, uint32_t seed, uint32_t& m0, uint32_t& m1, int& zsh) {
    const int t = threadIdx.x;
    uint32_t P0 = 0x00070007u ^ (t & 1 ? 0x8u : 0u), P1 = 0x0007ffffu, GG = 0xfffcfffcu;
    uint32_t mpv = 0x83000000u;
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    u32x16 win;
    for (int k = 0; k < 16; ++k) win[k] = seed + k;
    extern __shared__ uint32_t lds[];
    uint32_t* ring = lds; uint32_t* ptab = lds + 27 * 128;
    for (int k = t; k < 28 * 128; k += 64) lds[k] = k * seed;
    int slot = 1, meta_next = static_cast<int>(seed);
    int dl_meta = (t * 7919 + seed) & 0x0fff00ff;
    uint32_t Pn0 = P0, Pn1 = P1;
#pragma unroll 1
    for (int i = 1; i <= n; ++i) {
        const int meta = meta_next;
        if (W) {
            __builtin_amdgcn_sched_barrier(0);
            win[((i - 1) & 7) * 2] = m0; win[((i - 1) & 7) * 2 + 1] = m1;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (S) {
            meta_next = __builtin_amdgcn_readlane(dl_meta, i & 63);
            if (__builtin_expect((meta & (1 << 12)) != 0, 0)) { m0 ^= 1u; asm volatile("; rare a" : "+v"(m0)); }
            if (__builtin_expect((meta & (1 << 13)) != 0, 0)) { m1 ^= 1u; asm volatile("; rare b" : "+v"(m1)); }
            if (__builtin_expect((meta & (0xf << 9)) != 0, 0)) { m1 ^= 2u; asm volatile("; rare c" : "+v"(m1)); }
        }
        if (W) {
            const int wi = ((i - 1) & 7) * 2;
            m0 = win[wi]; m1 = win[wi + 1];
        }
        if (L) { P0 = Pn0; P1 = Pn1; }
        const uint32_t mprev = mpv = __builtin_amdgcn_update_dpp(mpv, m1, 0x138, 0xf, 0xf, false);
        const uint32_t D0 = __builtin_amdgcn_alignbit(m0, mprev, 16), D1 = __builtin_amdgcn_alignbit(m1, m0, 16);
        uint32_t a0 = pk_max(pk_add(D0, P0), pk_add(m0, GG)), a1 = pk_max(pk_add(D1, P1), pk_add(m1, GG));
        a0 = pk_chain_pair(a0); a1 = pk_chain_pair(a1);
        a1 = pk_max_bhi(a1, a0);
        int sc = static_cast<int>(a1) >> 16;
        if (L) { const uint32_t* src = ptab + ((meta_next >> 1) & 3) * 128 + t * 2; Pn0 = src[0]; Pn1 = src[1]; }
        sc = scan6(sc);
        zsh = dpp_or<0x138, 0xf>(zsh, sc);
        const int zex = max(zsh, -32000);
        m0 = pk_max_blo(a0, static_cast<uint32_t>(zex)); m1 = pk_max_blo(a1, static_cast<uint32_t>(zex));
        if (L) { uint32_t* dst = ring + (slot * 64 + t) * 2; dst[0] = m0; dst[1] = m1; }
        if (S || L) slot = (slot + 1 == 27) ? 0 : slot + 1;
        if (!L) P0 ^= 0x00010000u;
    }
}
// ---- half-row streams ----
// lane l: h0 = columns 2 l, 2 l + 1 (half A), h1 = columns 128 + 2 l, 129 + 2 l (half B)
template <bool SKEW, bool EXTRA>
__device__ __forceinline__ void rows_half(int n, uint32_t seed, uint32_t& m0, uint32_t& m1, int& zshA) {
    const int t = threadIdx.x;
    uint32_t P0 = 0x00070007u ^ (t & 1 ? 0x8u : 0u), P1 = 0x0007ffffu;
    const uint32_t GG = 0xfffcfffcu;
    uint32_t mpvA = 0x83000000u;
    int zshB = static_cast<int>(0x80000000u);
    int cA = -32000, cA_prev = -32000;           // half A's row maximum (carry into half B), of this row / of the row before
    uint32_t a63 = 0x83008300u;                  // half A's lane 63 of the predecessor row (diagonal into half B's first cell)
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    u32x16 win;
    for (int k = 0; k < 16; ++k) win[k] = seed + k;
    extern __shared__ uint32_t lds[];
    uint32_t* ring = lds; uint32_t* ptab = lds + 27 * 128;
    if (EXTRA) for (int k = t; k < 28 * 128; k += 64) lds[k] = k * seed;
    int slot = 1, meta_next = static_cast<int>(seed);
    int dl_meta = (t * 7919 + seed) & 0x0fff00ff;
    uint32_t Pn0 = P0, Pn1 = P1;
#pragma unroll 1
    for (int i = 1; i <= n; ++i) {
        const int meta = meta_next;
        if (EXTRA) {
            __builtin_amdgcn_sched_barrier(0);
            win[((i - 1) & 7) * 2] = m0; win[((i - 1) & 7) * 2 + 1] = m1;
            __builtin_amdgcn_sched_barrier(0);
            meta_next = __builtin_amdgcn_readlane(dl_meta, i & 63);
            if (__builtin_expect((meta & (1 << 12)) != 0, 0)) { m0 ^= 1u; asm volatile("; rare a" : "+v"(m0)); }
            if (__builtin_expect((meta & (1 << 13)) != 0, 0)) { m1 ^= 1u; asm volatile("; rare b" : "+v"(m1)); }
            if (__builtin_expect((meta & (0xf << 9)) != 0, 0)) { m1 ^= 2u; asm volatile("; rare c" : "+v"(m1)); }
            const int wi = ((i - 1) & 7) * 2;
            m0 = win[wi]; m1 = win[wi + 1];
            P0 = Pn0; P1 = Pn1;
        }
        // stream A: row i, columns [0, 128)
        const uint32_t mprevA = mpvA = __builtin_amdgcn_update_dpp(mpvA, m0, 0x138, 0xf, 0xf, false);
        const uint32_t D0 = __builtin_amdgcn_alignbit(m0, mprevA, 16);
        uint32_t a0 = pk_chain_pair(pk_max(pk_add(D0, P0), pk_add(m0, GG)));
        int s0 = static_cast<int>(a0) >> 16;
        // stream B: row i (H1) or row i - 1 (H2: m1 is a row behind, its inputs from stream A are a row old) -- same instructions
        const uint32_t rot = __builtin_amdgcn_update_dpp(a63, m1, 0x138, 0xf, 0xf, false);      // lane 0 keeps half A's last cell
        const uint32_t D1 = __builtin_amdgcn_alignbit(m1, rot, 16);
        uint32_t a1 = pk_chain_pair(pk_max(pk_add(D1, P1), pk_add(m1, GG)));
        int s1 = static_cast<int>(a1) >> 16;
        if (EXTRA) { const uint32_t* src = ptab + ((meta_next >> 1) & 3) * 128 + t * 2; Pn0 = src[0]; Pn1 = src[1]; }
        constexpr int I = static_cast<int>(0x80000000u);
        s0 = max(s0, dpp_or<0x111, 0xf>(I, s0)); s1 = max(s1, dpp_or<0x111, 0xf>(I, s1));
        s0 = max(s0, dpp_or<0x112, 0xf>(I, s0)); s1 = max(s1, dpp_or<0x112, 0xf>(I, s1));
        s0 = max(s0, dpp_or<0x114, 0xf>(I, s0)); s1 = max(s1, dpp_or<0x114, 0xf>(I, s1));
        s0 = max(s0, dpp_or<0x118, 0xf>(I, s0)); s1 = max(s1, dpp_or<0x118, 0xf>(I, s1));
        s0 = max(s0, dpp_or<0x142, 0xa>(I, s0)); s1 = max(s1, dpp_or<0x142, 0xa>(I, s1));
        s0 = max(s0, dpp_or<0x143, 0xc>(I, s0)); s1 = max(s1, dpp_or<0x143, 0xc>(I, s1));
        zshA = dpp_or<0x138, 0xf>(zshA, s0);
        zshB = dpp_or<0x138, 0xf>(zshB, s1);
        cA_prev = cA;
        cA = __builtin_amdgcn_readlane(s0, 63);
        const int zexA = max(zshA, -32000);
        const int zexB = max(max(zshB, SKEW ? cA_prev : cA), -32000);
        const uint32_t n0 = pk_max_blo(a0, static_cast<uint32_t>(zexA)), n1 = pk_max_blo(a1, static_cast<uint32_t>(zexB));
        // the diagonal into half B's first cell of the NEXT row B works on: half A's lane 63 of its predecessor row
        a63 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(SKEW ? m0 : n0), 63));
        m0 = n0; m1 = n1;
        if (EXTRA) { uint32_t* dst = ring + (slot * 64 + t) * 2; dst[0] = m0; dst[1] = m1; slot = (slot + 1 == 27) ? 0 : slot + 1; }
        else P0 ^= 0x00010000u;
    }
}
// ---- eight rows per trip, static register-window places ----
__device__ __forceinline__ void rows_oct(int n, uint32_t seed, uint32_t& m0, uint32_t& m1, int& zsh) {
    const int t = threadIdx.x;
    uint32_t P0, P1; const uint32_t GG = 0xfffcfffcu;
    uint32_t mpv = 0x83000000u;
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    u32x16 win;
    for (int k = 0; k < 16; ++k) win[k] = seed + k;
    extern __shared__ uint32_t lds[];
    uint32_t* ring = lds; uint32_t* ptab = lds + 27 * 128;
    for (int k = t; k < 28 * 128; k += 64) lds[k] = k * seed;
    int slot = 0, meta_next = static_cast<int>(seed);
    int dl_meta = (t * 7919 + seed) & 0x0fff00ff;
    uint32_t Pn0 = 0x00070007u, Pn1 = 0x0007ffffu;
    win[0] = m0; win[1] = m1;
#pragma unroll 1
    for (int i = 1; i <= n; i += 8) {
        uint32_t* const roct = ring + (slot * 64 + t) * 2;
#define OCT_ROW(o) { \
            const int meta = meta_next; \
            meta_next = __builtin_amdgcn_readlane(dl_meta, (i + (o)) & 63); \
            const int wi = (meta >> 16) & 14; \
            __builtin_amdgcn_sched_barrier(0); \
            m0 = win[wi]; m1 = win[wi + 1]; \
            __builtin_amdgcn_sched_barrier(0); \
            if (__builtin_expect((meta & (0xf << 9)) != 0, 0)) { m1 ^= 2u; asm volatile("; rare c" : "+v"(m1)); } \
            P0 = Pn0; P1 = Pn1; \
            const uint32_t mprev = mpv = __builtin_amdgcn_update_dpp(mpv, m1, 0x138, 0xf, 0xf, false); \
            const uint32_t D0 = __builtin_amdgcn_alignbit(m0, mprev, 16), D1 = __builtin_amdgcn_alignbit(m1, m0, 16); \
            uint32_t a0 = pk_max(pk_add(D0, P0), pk_add(m0, GG)), a1 = pk_max(pk_add(D1, P1), pk_add(m1, GG)); \
            a0 = pk_chain_pair(a0); a1 = pk_chain_pair(a1); \
            a1 = pk_max_bhi(a1, a0); \
            int sc = static_cast<int>(a1) >> 16; \
            { const uint32_t* src = ptab + ((meta_next >> 1) & 3) * 128 + t * 2; Pn0 = src[0]; Pn1 = src[1]; } \
            sc = scan6(sc); \
            zsh = dpp_or<0x138, 0xf>(zsh, sc); \
            const int zex = max(zsh, -32000); \
            m0 = pk_max_blo(a0, static_cast<uint32_t>(zex)); m1 = pk_max_blo(a1, static_cast<uint32_t>(zex)); \
            roct[(o) * 128] = m0; roct[(o) * 128 + 1] = m1; \
            win[(((o) + 1) & 7) * 2] = m0; win[(((o) + 1) & 7) * 2 + 1] = m1; }
        OCT_ROW(0) OCT_ROW(1) OCT_ROW(2) OCT_ROW(3) OCT_ROW(4) OCT_ROW(5) OCT_ROW(6) OCT_ROW(7)
#undef OCT_ROW
        slot = (slot + 8) & 15;
    }
}
__global__ void probe(unsigned long long* out, uint32_t* sink, int n, int mode, uint32_t seed) {
    const int t = threadIdx.x;
    uint32_t m0 = seed * (t + 1), m1 = seed * (t + 7);
    uint32_t P0 = 0x00070007u ^ (t & 1 ? 0x8u : 0u), P1 = 0x0007ffffu, GG = 0xfffcfffcu;
    int zsh = static_cast<int>(0x80000000u), cin = -32000;
    uint32_t mpv = 0x83000000u;
    uint32_t l0 = m0, l1 = m1;                                   // B: the previous row's pre-carry values
    const int tB = 3;
    const uint32_t pB0 = 0x00030003u, pB1 = 0x00030003u;
    const long long t0 = clock64();
    if (mode == 0) {
        for (int i = 0; i < n; ++i) {
            const uint32_t mprev = mpv = __builtin_amdgcn_update_dpp(mpv, m1, 0x138, 0xf, 0xf, false);
            const uint32_t D0 = __builtin_amdgcn_alignbit(m0, mprev, 16), D1 = __builtin_amdgcn_alignbit(m1, m0, 16);
            uint32_t a0 = pk_max(pk_add(D0, P0), pk_add(m0, GG)), a1 = pk_max(pk_add(D1, P1), pk_add(m1, GG));
            a0 = pk_chain_pair(a0); a1 = pk_chain_pair(a1);
            a1 = pk_max_bhi(a1, a0);
            int sc = static_cast<int>(a1) >> 16;
            sc = scan6(sc);
            zsh = dpp_or<0x138, 0xf>(zsh, sc);
            const int zex = max(zsh, -32000);
            m0 = pk_max_blo(a0, static_cast<uint32_t>(zex)); m1 = pk_max_blo(a1, static_cast<uint32_t>(zex));
            P0 ^= 0x00010000u;
        }
    } else if (mode == 1) {
        for (int i = 0; i < n; ++i) {
            // off the path: the previous row's final values are max(l, cin); the candidates from the pre-carry values
            const uint32_t lprev = __builtin_amdgcn_update_dpp(0x83000000u, 0x83008300u, 0x138, 0xf, 0xf, false);     // (first cell: no in-lane left neighbour)
            const uint32_t D0 = __builtin_amdgcn_alignbit(l0, lprev, 16), D1 = __builtin_amdgcn_alignbit(l1, l0, 16);
            uint32_t a0 = pk_max(pk_add(D0, P0), pk_add(l0, GG)), a1 = pk_max(pk_add(D1, P1), pk_add(l1, GG));
            a0 = pk_chain_pair(a0); a1 = pk_chain_pair(a1);
            a1 = pk_max_bhi(a1, a0);
            const int tA = static_cast<int>(a1) >> 16;
            // on the path: carry of the previous row -> this row's tail -> scan -> this row's carry
            const int tail = max(tA, cin + tB);
            const uint32_t cb = static_cast<uint32_t>(cin) & 0xffffu;
            const uint32_t cc = cb | (cb << 16);
            l0 = pk_max(a0, pk_add(pB0, cc)); l1 = pk_max(a1, pk_add(pB1, cc));
            const int sc = scan6(tail);
            zsh = dpp_or<0x138, 0xf>(zsh, sc);
            cin = max(zsh, -32000);
            P0 ^= 0x00010000u;
        }
        m0 = l0; m1 = l1 + cin;
    } else if (mode == 2) {
        for (int i = 0; i < n; ++i) {
            const uint32_t mprev = mpv = __builtin_amdgcn_update_dpp(mpv, m1, 0x138, 0xf, 0xf, false);
            const uint32_t D0 = __builtin_amdgcn_alignbit(m0, mprev, 16), D1 = __builtin_amdgcn_alignbit(m1, m0, 16);
            uint32_t a0 = pk_max(pk_add(D0, P0), pk_add(m0, GG)), a1 = pk_max(pk_add(D1, P1), pk_add(m1, GG));
            a0 = pk_chain_pair(a0); a1 = pk_chain_pair(a1);
            a1 = pk_max_bhi(a1, a0);
            const int sc = static_cast<int>(a1) >> 16;
            const int zex = max(sc, -32000);
            m0 = pk_max_blo(a0, static_cast<uint32_t>(zex)); m1 = pk_max_blo(a1, static_cast<uint32_t>(zex));
            P0 ^= 0x00010000u;
        }
    } else if (mode >= 10) {
        switch (mode) {
            case 10: rows_half<false, false>(n, seed, m0, m1, zsh); break;
            case 11: rows_half<true, false>(n, seed, m0, m1, zsh); break;
            case 12: rows_half<true, true>(n, seed, m0, m1, zsh); break;
            default: rows_oct(n, seed, m0, m1, zsh); break;
        }
    } else if (mode >= 4) {
        // A plus pieces of the real row: 5 = the register window (indexed write of the finished row, indexed read of the
        // predecessor), 6 = the LDS traffic (profile table read for the next row, ring write), 7 = the scalar bookkeeping
        // (descriptor word by v_readlane, three class tests with branches not taken, ring slot counter), 8 = all three
        switch (mode) {
            case 4: rows_wls<false, false, false>(n, seed, m0, m1, zsh); break;
            case 5: rows_wls<true, false, false>(n, seed, m0, m1, zsh); break;
            case 6: rows_wls<false, true, false>(n, seed, m0, m1, zsh); break;
            case 7: rows_wls<false, false, true>(n, seed, m0, m1, zsh); break;
            case 8: rows_wls<true, true, false>(n, seed, m0, m1, zsh); break;
            default: rows_wls<true, true, true>(n, seed, m0, m1, zsh); break;
        }
    } else {
        int sc = static_cast<int>(m0);
        for (int i = 0; i < n; ++i) { sc = scan6(sc); zsh = dpp_or<0x138, 0xf>(zsh, sc); sc = max(zsh, -32000) + i; }
        m0 = static_cast<uint32_t>(sc);
    }
    const long long t1 = clock64();
    sink[blockIdx.x * 64 + t] = m0 ^ m1 ^ static_cast<uint32_t>(zsh);
    if (t == 0) out[blockIdx.x * 16 + mode] = static_cast<unsigned long long>(t1 - t0);
}
int main() {
    unsigned long long* d; uint32_t* s;
    if (hipMalloc(&d, 64 * 16 * 8) != hipSuccess || hipMalloc(&s, 64 * 64 * 4) != hipSuccess) return 1;
    const int n = 200000;
    const char* names[14] = {"A chain row as it is", "B lazy carry", "C row without the scan", "D scan + carry alone", "A again (other loop form)", "A + register window", "A + LDS table read, ring write", "A + scalar bookkeeping", "A + window + LDS", "A + all three",
        "H1 half-row streams, one row per trip", "H2 half-row streams, skewed", "H3 = H2 + all three", "U  A + all three, 8 rows per trip"};
    for (int mode = 0; mode < 14; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 20480, 0, d, s, n, mode, 0x10001u);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        unsigned long long h[16];
        if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 3;
        printf("%-32s %7.1f clocks per row\n", names[mode], (double)h[mode] / n);
    }
    return 0;
}
