// Where do the waves of 256-thread work-groups land?  Prints (xcc, se, cu, simd) statistics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void k(unsigned* out, int spin) {
    extern __shared__ int lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main() {
    const int nb = 2048;
    unsigned* d; hipMalloc(&d, nb * 4 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 20480, 0, d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    int hist[4][4] = {};   // wave index in block x simd
    std::map<unsigned, int> w0_per_cu_simd;
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) {
        unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15;
        unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        hist[w][simd]++;
        if (w == 0) w0_per_cu_simd[(xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd]++;
        if (b < 4 || (b % 256 == 0 && b < 1100)) printf("block %d wave %d: xcc %u se %u sh %u cu %u simd %u waveslot %u\n", b, w, xcc, se, sh, cu, simd, hw & 15);
    }
    for (int w = 0; w < 4; ++w) printf("wave %d -> simd hist: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    std::map<int,int> occ; for (auto& kv : w0_per_cu_simd) occ[kv.second]++;
    for (auto& kv : occ) printf("(cu,simd) pairs hosting %d wave-0s: %d\n", kv.first, kv.second);
    return 0;
}
