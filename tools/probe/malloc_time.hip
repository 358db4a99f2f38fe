// How long do hipMalloc / hipFree / hipHostMalloc take on this box, by size and by history?  (The engine's arenas are tens
// of GB: profiles/r03 showed the same 23 GB allocation take 0.3 ms in one process and 1.2 s in the next.)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/malloc_time tools/probe/malloc_time.hip && /tmp/malloc_time
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    (void)hipFree(nullptr);
    for (int round = 0; round < 3; ++round) {
        std::vector<void*> blocks;
        printf("round %d: hipMalloc of 8 GB blocks, ms each:", round);
        for (int k = 0; k < 8; ++k) {
            void* p = nullptr; const double t = now();
            if (hipMalloc(&p, 8ull << 30) != hipSuccess) { printf(" fail"); break; }
            printf(" %.1f", now() - t); blocks.push_back(p);
        }
        printf("\n         hipFree, ms each:");
        for (void* p : blocks) { const double t = now(); (void)hipFree(p); printf(" %.1f", now() - t); }
        printf("\n");
    }
    for (size_t mb : {16, 64, 128, 256}) {
        void* p = nullptr; const double t = now();
        if (hipHostMalloc(&p, mb << 20, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc %zu MB failed\n", mb); continue; }
        const double t1 = now(); (void)hipHostFree(p);
        printf("hipHostMalloc %4zu MB: %.1f ms, hipHostFree %.1f ms\n", mb, t1 - t, now() - t1);
    }
    { void* p = nullptr; const double t = now(); (void)hipMalloc(&p, 24ull << 30); printf("one 24 GB block: %.1f ms", now() - t); const double t1 = now(); (void)hipFree(p); printf(", free %.1f ms\n", now() - t1); }
    return 0;
}
