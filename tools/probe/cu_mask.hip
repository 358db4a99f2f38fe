// Where do the bits of hipExtStreamCreateWithCUMask land on an MI355X (8 XCDs x 32 CUs)?  Launches the placement probe on
// streams whose mask has the first N bits set and prints, per XCD, how many distinct CUs ran work-groups.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask tools/probe/cu_mask.hip && /tmp/cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
    extern __shared__ int lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
    long long t0 = clock64();
    while (clock64() - t0 < spin) { lds[threadIdx.x] = (int)t0; }
}
static void run(const char* what, hipStream_t s) {
    const int nb = 1024;
    unsigned* d; if (hipMalloc(&d, nb * 2 * sizeof(unsigned)) != hipSuccess) return;
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 20480 + 512, s, d, 200000);
    if (hipStreamSynchronize(s) != hipSuccess) { printf("%s: launch failed\n", what); return; }
    std::vector<unsigned> h(nb * 2);
    if (hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
    std::set<unsigned> cus[8];
    for (int b = 0; b < nb; ++b) { const unsigned hw = h[b * 2], x = h[b * 2 + 1] & 7; cus[x].insert(((hw >> 13) & 7) << 8 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 15)); }
    printf("%-28s distinct CUs per XCD:", what);
    int tot = 0;
    for (int x = 0; x < 8; ++x) { printf(" %2zu", cus[x].size()); tot += (int)cus[x].size(); }
    printf("  (total %d)\n", tot);
    (void)hipFree(d);
}
int main() {
    hipStream_t s0; if (hipStreamCreate(&s0) != hipSuccess) return 1;
    run("no mask", s0);
    for (int n : {32, 64, 96, 128}) {
        std::vector<uint32_t> mask(8, 0u);
        for (int b = 0; b < n; ++b) mask[b >> 5] |= 1u << (b & 31);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("mask of the first %d bits: stream creation failed\n", n); continue; }
        char what[64]; snprintf(what, sizeof what, "first %d bits", n);
        run(what, s);
    }
    for (int n : {96, 128}) {   // the complement of a prefix: bits n .. 255
        std::vector<uint32_t> mask(8, 0u);
        for (int b = n; b < 256; ++b) mask[b >> 5] |= 1u << (b & 31);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("mask of bits %d..255: stream creation failed\n", n); continue; }
        char what[64]; snprintf(what, sizeof what, "bits %d..255", n);
        run(what, s);
    }
    {   // every fourth bit: does a sparse mask spread over the XCDs?
        std::vector<uint32_t> mask(8, 0x11111111u);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) == hipSuccess) run("every fourth bit (64 set)", s);
    }
    return 0;
}
