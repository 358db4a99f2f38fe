// Which SIMD does wave k of a 4-wave work-group land on?  (8 work-groups per CU, 20.5 KiB LDS each, like poa_window_kernel2.)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/simd_map tools/probe/simd_map.hip && /tmp/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <array>
__global__ void probe(unsigned* out, int spin) {
    extern __shared__ int lds[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    long long t0 = clock64();
    while (clock64() - t0 < spin) { lds[threadIdx.x] = (int)t0; }      // stay resident so that all work-groups coexist
}
int main() {
    const int nb = 2048;
    unsigned* d; hipMalloc(&d, nb * 4 * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 20480 + 512, 0, d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int hist[4][4] = {{0}};          // [wave in group][simd]
    int same_cu_simd0[8] = {0};
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) { unsigned hw = h[(b * 4 + w) * 2]; hist[w][(hw >> 4) & 3]++; }
    for (int w = 0; w < 4; ++w) printf("wave %d of its group -> SIMD 0..3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    // per CU: how many wave-0s share a SIMD
    struct K { unsigned key; int simd; };
    std::vector<std::vector<int>> per;  // per (xcc, se, cu): simd histogram of wave 0
    int maxshare[9] = {0};
    std::vector<unsigned> keys; std::vector<std::array<int,4>> cnt;
    for (int b = 0; b < nb; ++b) { unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15; unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15);
        size_t k = 0; for (; k < keys.size(); ++k) if (keys[k] == key) break;
        if (k == keys.size()) { keys.push_back(key); cnt.push_back({0, 0, 0, 0}); }
        cnt[k][(hw >> 4) & 3]++; }
    for (auto& c : cnt) { int m = 0; for (int s = 0; s < 4; ++s) m = c[s] > m ? c[s] : m; maxshare[m > 8 ? 8 : m]++; }
    printf("%zu CUs seen; CUs by the largest number of wave-0s on one SIMD:", keys.size());
    for (int m = 0; m <= 8; ++m) printf(" %d:%d", m, maxshare[m]);
    printf("\nfirst 16 groups (wave 0): ");
    for (int b = 0; b < 16; ++b) { unsigned hw = h[b * 8]; printf("[xcc %u se %u cu %u simd %u] ", h[b * 8 + 1] & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3); }
    printf("\n");
    return 0;
}
