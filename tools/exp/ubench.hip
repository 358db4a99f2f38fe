// Single-wave instruction cost microbenchmarks on gfx950: cycles per instruction slot (s_memtime ticks),
// one 64-thread block per CU, only block 0 reports.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define BENCH(idx, n, body) { long long t0 = clock64(); for (int it = 0; it < 16; ++it) { asm volatile(body : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) :: "vcc", "m0", "memory"); } long long t1 = clock64(); if (threadIdx.x == 0 && blockIdx.x == 0) out[idx] = (double)(t1 - t0) / (16.0 * n); }
__global__ void k(double* out, int seed) {
    extern __shared__ int lds[];
    unsigned a = threadIdx.x + seed, b = threadIdx.x * 3 + seed, c = seed, d = 7;
    unsigned w0 = 1, w1 = 2, w2 = 3, w3 = 4;
    unsigned s0 = seed & 3, s1 = seed & 7;
    lds[threadIdx.x] = a;
    BENCH(0, 64, REP64("v_add_u32 %0, %0, %1\n"))                                   // dependent VALU chain
    BENCH(1, 64, REP16("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_xor_b32 %3, %3, %1\n v_add_u32 %6, 1, %6\n"))   // 4 independent chains
    BENCH(2, 64, REP64("v_pk_max_i16 %0, %0, %1\n s_nop 0\n"))                      // dependent packed + 1 wait state
    BENCH(3, 64, REP16("v_pk_max_i16 %0, %0, %1\n v_pk_add_u16 %2, %2, %1\n v_pk_max_i16 %3, %3, %1\n v_pk_add_u16 %6, %6, %1\n"))   // independent packed
    BENCH(4, 64, REP64("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"))   // scan step
    BENCH(5, 64, REP64("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n"))               // whole-wave shift (independent)
    BENCH(6, 64, REP64("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"))      // dependent wave shift
    BENCH(7, 64, REP64("s_set_gpr_idx_on %4, gpr_idx(SRC0)\n v_mov_b32 %0, %6\n s_set_gpr_idx_off\n")) // indexed read triple (count as 1)
    BENCH(8, 64, REP64("v_readlane_b32 %4, %0, %5\n"))                              // readlane with sgpr index
    BENCH(9, 64, REP64("v_readlane_b32 %4, %0, 3\n s_add_u32 %5, %5, %4\n"))        // readlane -> salu use (pair counts as 1)
    BENCH(10, 64, REP64("s_add_u32 %4, %4, %5\n"))                                  // dependent SALU
    BENCH(11, 64, REP64("s_add_u32 %4, %4, 1\n s_and_b32 %5, %5, 7\n"))             // 2 independent SALU (pair = 1)
    BENCH(12, 64, REP64("s_cmp_eq_u32 %4, 12345\n s_cbranch_scc1 1f\n s_nop 0\n1:\n"))   // not-taken branch... (taken skips nop)
    BENCH(13, 64, REP64("s_cmp_lg_u32 %4, 12345\n s_cbranch_scc1 1f\n s_nop 0\n1:\n"))   // taken short branch
    BENCH(14, 64, REP64("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n"))            // LDS round trip (b = address garbage bounded below)
    BENCH(15, 64, REP64("v_alignbit_b32 %0, %0, %1, 16\n"))
    BENCH(16, 64, REP64("v_pk_mad_i16 %0, %0, %1, %2\n s_nop 0\n"))
    BENCH(17, 64, REP64("s_nop 0\n"))
    BENCH(18, 64, REP64("v_cndmask_b32 %0, %0, %1, vcc\n"))
    BENCH(19, 64, REP64("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n"))                   // 2 independent movs = 1
    if (threadIdx.x == 0) out[63] = a + b + c + d + s0 + s1 + w0 + w1 + w2 + w3;
}
int main() {
    double* d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    for (int nb : {1, 256 * 8}) {
        hipLaunchKernelGGL(k, dim3(nb), dim3(64), 1024, 0, d, 0);
        hipDeviceSynchronize();
        double h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const char* names[] = {"dep v_add", "4 indep valu", "dep v_pk_max+nop", "indep v_pk", "dpp row_shr max+nop1", "wave_shr mov indep", "wave_shr dep+nop1",
                               "gpr_idx read triple", "readlane sidx", "readlane->salu pair", "dep salu", "2 indep salu", "branch not taken (cmp+br+nop)", "branch taken (cmp+br)",
                               "ds_read+wait", "alignbit dep", "pk_mad dep+nop", "s_nop 0", "cndmask dep", "2 indep mov"};
        printf("blocks=%d\n", nb);
        for (int i = 0; i < 20; ++i) printf("  %-32s %.2f\n", names[i], h[i]);
    }
    return 0;
}
