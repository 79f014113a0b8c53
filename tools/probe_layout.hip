// probe_layout.hip -- how does the spacing between the four score columns (their relative base addresses) change the
// classify kernel's time?  One big allocation, column c at c * spacing; spacings from the command line (bytes).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I include -I xenomapper_amd/csrc tools/probe_layout.hip -o build/probe_layout
//   build/probe_layout 100000000 400000000 402653184 536870912 ...
#include "../xenomapper_amd/csrc/xm_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char **argv)
{
    const uint64_t N = strtoull(argv[1], 0, 10);
    std::vector<uint64_t> spacings;
    for (int i = 2; i < argc; ++i) spacings.push_back(strtoull(argv[i], 0, 10));
    uint64_t max_sp = *std::max_element(spacings.begin(), spacings.end());
    char *big; CK(hipMalloc(&big, 3 * max_sp + N * 4 + (4u << 20)));
    char *base = (char *)(((uintptr_t)big + (2u << 20) - 1) / (2u << 20) * (2u << 20));
    uint8_t *code, *bins4; uint64_t *bits; CK(hipMalloc(&code, N + 64)); CK(hipMalloc(&bins4, N / 2 + 4096)); CK(hipMalloc(&bits, N / 8 + 64));
    CK(hipMemset(bits, 0xAA, N / 8 + 64));
    uint32_t *gc; uint64_t *rep; CK(hipMalloc(&gc, 8 * (N / 2048 + 256) * 4)); CK(hipMalloc(&rep, (64 * 64 + 8) * 8)); CK(hipMemset(rep, 0, (64 * 64 + 8) * 8));
    std::vector<int32_t> h(N);
    uint64_t s = 88172645463325252ull;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("base %p\n", (void *)base);
    for (int rep_round = 0; rep_round < 2; ++rep_round)
    for (uint64_t sp : spacings) {
        int32_t *col[4];
        for (int c = 0; c < 4; ++c) {
            col[c] = (int32_t *)(base + c * sp);
            for (uint64_t i = 0; i < N; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int32_t)(61 + (s >> 40) % 240); }
            if (rep_round == 0 || true) CK(hipMemcpy(col[c], h.data(), N * 4, hipMemcpyHostToDevice));
        }
        std::vector<float> t;
        for (int r = 0; r < 12; ++r) {
            CK(hipEventRecord(e0));
            xm::launch_classify_i32(0, 1, N, col[0], col[1], col[2], col[3], bits, INT32_MIN, code, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("spacing %12llu B (%9.3f MiB, mod 2MiB %7llu, mod 4KiB %4llu): classify median %7.1f us min %7.1f us\n", (unsigned long long)sp, sp / 1048576.0,
               (unsigned long long)(sp % (2u << 20)), (unsigned long long)(sp % 4096), t[t.size() / 2] * 1e3, t[0] * 1e3);
        fflush(stdout);
    }
    return 0;
}
