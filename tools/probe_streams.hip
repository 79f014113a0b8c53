// probe_streams.hip -- on-box microbenchmark (not shipped): how fast does HBM deliver a fixed number of bytes when they
// come as K concurrent streams?  Every lane reads 16 bytes from each of K equally long arrays (one 1 KiB request per
// wave and array, as the classify kernels do), nothing is written.  Variants: workgroup size, and "tiled": the same
// bytes as ONE array in which each 256-record tile holds its K pieces back to back (an AoSoA layout).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_streams.hip -o /tmp/probe_streams && /tmp/probe_streams
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int K, int BLOCK, bool TILED>
__global__ void __launch_bounds__(BLOCK) read_k(const v4i *base, size_t stride_v4, size_t groups, unsigned *out)
{
    const size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= groups) return;
    v4i acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < K; ++k) {
        // SoA: array k starts at k * stride; tiled: tile (g / 64) holds K pieces of 64 x 16 bytes back to back
        const v4i *p = TILED ? base + (g / 64) * (64 * K) + k * 64 + (g % 64) : base + k * stride_v4 + g;
        acc ^= __builtin_nontemporal_load(p);
    }
    unsigned r = (unsigned)(acc.x ^ acc.y ^ acc.z ^ acc.w);
    r ^= (unsigned)__shfl_xor((int)r, 32, 64);
    if (r == 0x12345678u && (threadIdx.x & 63) == 0) out[g >> 6] = r;      // practically never
}

template <int K, int BLOCK, bool TILED>
static void run(const v4i *buf, size_t total_bytes, unsigned *out)
{
    const size_t groups = total_bytes / 16 / K;                  // 16-byte groups per stream
    const size_t stride = (groups + 4095 + 977) & ~(size_t)63;   // not a power of two apart
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(a));
        read_k<K, BLOCK, TILED><<<(unsigned)((groups + BLOCK - 1) / BLOCK), BLOCK>>>(buf, stride, groups, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        if (it >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)groups * 16 * K;
    printf("K=%2d block=%4d %-5s  median %.1f us  min %.1f us  ->  %.2f TB/s (median)\n", K, BLOCK, TILED ? "tiled" : "soa",
           ms[ms.size() / 2] * 1e3, ms[0] * 1e3, bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
}

// write-only ceiling: every lane stores 4 (W4 = false) or 16 bytes, consecutive lanes consecutive addresses
template <bool W4, bool NT>
__global__ void __launch_bounds__(256) write_k(unsigned *out, size_t words)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * (W4 ? 4 : 1);
    if (i >= words) return;
    if (W4) {
        v4i v = {(int)i, (int)i + 1, (int)i + 2, (int)i + 3};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4i *>(out + i)); else *reinterpret_cast<v4i *>(out + i) = v;
    } else {
        if (NT) __builtin_nontemporal_store((unsigned)i, out + i); else out[i] = (unsigned)i;
    }
}

template <bool W4, bool NT>
static void run_write(unsigned *buf, size_t bytes)
{
    const size_t words = bytes / 4, threads = W4 ? words / 4 : words;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(a));
        write_k<W4, NT><<<(unsigned)((threads + 255) / 256), 256>>>(buf, words);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        if (it >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("write %5zu MB  %2d B/lane %-5s  median %.1f us  min %.1f us  ->  %.2f TB/s (median)\n", bytes >> 20, W4 ? 16 : 4,
           NT ? "nt" : "plain", ms[ms.size() / 2] * 1e3, ms[0] * 1e3, (double)bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
}

// The same bytes from a PERSISTENT grid (MI355X_MICROARCH.md's "plain stores of the same shape" row: one dword per lane,
// 256 B per wave instruction, 8 waves per CU -> 6.0-6.2 TB/s): WAVES_PER_CU x 256 CUs waves, each wave writes 256-byte
// (dword per lane) or 1-KiB (16 B per lane) pieces; ROWS: pieces of one wave are `nwaves` apart (a grid-stride loop);
// otherwise every wave owns one contiguous chunk.  UNROLL stores are issued back to back.
template <bool W4, bool ROWS, int UNROLL>
__global__ void __launch_bounds__(256) write_loop(unsigned *out, size_t words, unsigned nwaves)
{
    const unsigned lane = threadIdx.x & 63u;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t piece_words = W4 ? 256 : 64;                      // words one wave instruction writes
    const size_t pieces = words / piece_words;
    const size_t per_wave = (pieces + nwaves - 1) / nwaves;
    for (size_t k0 = 0; k0 < per_wave; k0 += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t k = k0 + u;
            const size_t piece = ROWS ? k * nwaves + wave : wave * per_wave + k;
            if (k < per_wave && piece < pieces) {
                const size_t i = piece * piece_words + lane * (W4 ? 4 : 1);
                if (W4) { v4i v = {(int)i, (int)i + 1, (int)i + 2, (int)i + 3}; *reinterpret_cast<v4i *>(out + i) = v; }
                else out[i] = (unsigned)i;
            }
        }
    }
}

template <bool W4, bool ROWS, int UNROLL>
static void run_write_loop(unsigned *buf, size_t bytes, int waves_per_cu)
{
    const size_t words = bytes / 4;
    const unsigned nwaves = 256u * (unsigned)waves_per_cu;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(a));
        write_loop<W4, ROWS, UNROLL><<<nwaves / 4, 256>>>(buf, words, nwaves);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        if (it >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("write-loop %5zu MB  %2d B/lane %2d waves/CU %-6s unroll %d  median %.1f us  min %.1f us  ->  %.2f TB/s (median)\n", bytes >> 20,
           W4 ? 16 : 4, waves_per_cu, ROWS ? "rows" : "chunks", UNROLL, ms[ms.size() / 2] * 1e3, ms[0] * 1e3,
           (double)bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    const bool writes_only = argc > 1 && argv[1][0] == 'w';
    if (writes_only) {
        unsigned *wb;
        CK(hipMalloc(&wb, 2048ull << 20));
        CK(hipMemset(wb, 1, 2048ull << 20));
        for (size_t mb : {200, 2000}) {
            run_write<false, false>(wb, mb << 20);
            run_write<true, false>(wb, mb << 20);
            for (int wpc : {8, 16, 32}) {
                run_write_loop<false, true, 1>(wb, mb << 20, wpc);
                run_write_loop<false, true, 4>(wb, mb << 20, wpc);
                run_write_loop<false, false, 4>(wb, mb << 20, wpc);
                run_write_loop<true, true, 1>(wb, mb << 20, wpc);
                run_write_loop<true, true, 4>(wb, mb << 20, wpc);
                run_write_loop<true, false, 4>(wb, mb << 20, wpc);
            }
        }
        return 0;
    }

    const size_t total = 2400ull << 20;                          // 2.4 GiB read per launch, whatever K
    v4i *buf; unsigned *out;
    CK(hipMalloc(&buf, total + (64u << 20)));
    CK(hipMemset(buf, 1, total + (64u << 20)));
    CK(hipMalloc(&out, 64u << 20));
    run<1, 256, false>(buf, total, out);
    run<2, 256, false>(buf, total, out);
    run<4, 256, false>(buf, total, out);
    run<4, 512, false>(buf, total, out);
    run<6, 256, false>(buf, total, out);
    run<8, 256, false>(buf, total, out);
    run<8, 512, false>(buf, total, out);
    run<10, 256, false>(buf, total, out);
    run<10, 512, false>(buf, total, out);
    run<12, 256, false>(buf, total, out);
    run<4, 256, true>(buf, total, out);
    run<8, 256, true>(buf, total, out);
    run<10, 512, true>(buf, total, out);
    run<12, 256, true>(buf, total, out);
    for (size_t mb : {200, 400, 2000}) {
        run_write<false, false>((unsigned *)buf, mb << 20);
        run_write<false, true>((unsigned *)buf, mb << 20);
        run_write<true, false>((unsigned *)buf, mb << 20);
        run_write<true, true>((unsigned *)buf, mb << 20);
    }
    return 0;
}
