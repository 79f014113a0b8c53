// Pieces both device front ends use when they gather the six OUTPUTS on the device (xm_bamdev_fetch_bins, xm_strip_fetch_bins;
// SURVEY f-2's gather kernel): which file a bin prints from, the bin of a place in the packed unit list, the size scan (three small
// launches over a uint32 array), where each bin's text begins, and the copy of the finished stream into the host's page-locked,
// device-mapped buffer.  Every kernel lives in an anonymous namespace: each translation unit that includes this gets its own.
#pragma once
#include <chrono>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace {

constexpr uint32_t NO_RECORD = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t files_of_bin(uint32_t bin, uint32_t sink_mask)     // bit 0: file 1's line, bit 1: file 2's
{
    if (bin > 5u || ((sink_mask >> bin) & 1u) == 0u) return 0u;
    return bin == 4u ? 3u : (bin == 1u || bin == 3u) ? 2u : 1u;
}

constexpr uint32_t SCAN_ITEMS = 16, SCAN_TILE = 256 * SCAN_ITEMS;
__global__ void __launch_bounds__(256)
size_sum_kernel(const uint32_t *__restrict__ v, uint32_t n, uint32_t *__restrict__ part)
{
    __shared__ uint32_t ws[4];
    const uint32_t i0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k) s += (i0 + k < n) ? v[i0 + k] : 0u;
    for (int d = 32; d; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63u) == 0u) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0u) part[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ void __launch_bounds__(1024)
part_scan_kernel(uint32_t *__restrict__ part, uint32_t n_part, uint32_t *__restrict__ total)
{
    __shared__ uint32_t sh[1024];
    const uint32_t t = threadIdx.x, per = (n_part + 1023u) / 1024u;
    const uint32_t a = min(t * per, n_part), e = min(a + per, n_part);
    uint32_t s = 0;
    for (uint32_t k = a; k < e; ++k) s += part[k];
    sh[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t x = t >= d ? sh[t - d] : 0u;
        __syncthreads();
        sh[t] += x;
        __syncthreads();
    }
    uint32_t run = sh[t] - s;
    for (uint32_t k = a; k < e; ++k) { const uint32_t x = part[k]; part[k] = run; run += x; }
    if (t == 1023u) *total = sh[1023];
}

template <bool ALL>          // ALL: every item gets its place (an empty one that of the next); else empty items get NO_RECORD
__global__ void __launch_bounds__(256)
size_place_kernel(const uint32_t *__restrict__ v, uint32_t n, const uint32_t *__restrict__ part, uint32_t *__restrict__ place)
{
    __shared__ uint32_t ws[4];
    const uint32_t i0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x[SCAN_ITEMS], s = 0;
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k) { x[k] = (i0 + k < n) ? v[i0 + k] : 0u; s += x[k]; }
    uint32_t incl = s;                                                      // inclusive scan of the lanes' sums inside the wave
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += y; }
    if (lane == 63u) ws[wave] = incl;
    __syncthreads();
    uint32_t base = part[blockIdx.x] + incl - s;
    for (uint32_t w = 0; w < wave; ++w) base += ws[w];
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k)
        if (i0 + k < n) { place[i0 + k] = (ALL || x[k]) ? base : NO_RECORD; base += x[k]; }
}

__device__ __forceinline__ uint32_t bin_of_place(uint32_t p, const unsigned long long *__restrict__ off)
{
    uint32_t b = 0;
#pragma unroll
    for (uint32_t k = 1; k < 7u; ++k) b += (off[k] <= (unsigned long long)p) ? 1u : 0u;
    return b;
}

// where bin b's text begins (b = 0..6; [7] = the total), and whether the 32-bit places wrapped (the sizes summed again in 64 bits)
__global__ void __launch_bounds__(64)
bin_start_kernel(const uint32_t *__restrict__ uplace, const unsigned long long *__restrict__ off, uint32_t n_units,
                 uint32_t *__restrict__ total_and_wrapped, uint32_t *__restrict__ starts)
{
    const uint32_t k = threadIdx.x;
    const uint32_t total = total_and_wrapped[0];
    if (k < 8u) starts[k] = (k < 7u && off[k] < (unsigned long long)n_units) ? uplace[off[k]] : total;
}

// G3: the stream to the host's page-locked buffer (device-mapped), 16 bytes per lane and step, both sides 16-byte aligned.  Its own
// kernel instead of hipMemcpyAsync, whose shader blit takes whatever share of the chip it likes beside the next window's inflate
// launch: this one is a fixed, small number of ONE-WAVE workgroups (XM_BAMDEV_COPY_WG of them x 4), each streaming its contiguous
// share with four pieces per lane in flight.  One wave per workgroup because the inflate launch beside it is persistent and holds
// every wave slot its LDS allows -- 31 of a CU's 32: a single wave finds the free slot at once, a workgroup of four waits until four
// chains on one CU have run out of blocks, i.e. for the end of the launch it was meant to run beside.
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
template <int DEEP>
__global__ void __launch_bounds__(64)
out_copy_kernel_t(const v4u32 *__restrict__ src, v4u32 *__restrict__ dst, uint64_t n16)
{
    const uint64_t per = (n16 + gridDim.x - 1u) / gridDim.x;
    const uint64_t lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
    uint64_t k = lo + threadIdx.x;
    if (DEEP == 8)
        for (; k + 448u < hi; k += 512u) {
            v4u32 r[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = __builtin_nontemporal_load(src + k + 64u * q);
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[k + 64u * q] = r[q];
        }
    for (; k + 192u < hi; k += 256u) {
        const v4u32 a = __builtin_nontemporal_load(src + k), b = __builtin_nontemporal_load(src + k + 64u),
                    c = __builtin_nontemporal_load(src + k + 128u), d = __builtin_nontemporal_load(src + k + 192u);
        dst[k] = a; dst[k + 64u] = b; dst[k + 128u] = c; dst[k + 192u] = d;
    }
    for (; k < hi; k += 64u) dst[k] = __builtin_nontemporal_load(src + k);
}

static uint32_t out_copy_waves(uint32_t wg)                                     // XM_BAMDEV_COPY_WAVES (A/B): the waves themselves, not workgroups x 4
{
    static const long waves = [] { const char *v = getenv("XM_BAMDEV_COPY_WAVES"); return v && *v ? strtol(v, nullptr, 10) : 0L; }();
    return waves > 0 ? (uint32_t)(waves > 16384 ? 16384 : waves) : 4u * wg;
}

static void out_copy_launch(uint32_t grid, hipStream_t st, const v4u32 *src, v4u32 *dst, uint64_t n16)
{
    static const bool deep = [] { const char *v = getenv("XM_BAMDEV_COPY_DEEP"); return v && v[0] == '8'; }();   // A/B: eight pieces per lane in flight
    if (deep) out_copy_kernel_t<8><<<grid, 64, 0, st>>>(src, dst, n16);
    else out_copy_kernel_t<4><<<grid, 64, 0, st>>>(src, dst, n16);
}

// Workgroups (x 4) of out_copy_kernel.  The copy's stores wait in the same queues towards the fabric as the stores of the inflate launch
// beside it: the more of them are in flight, the slower that launch, whether it reads its input over the link or from HBM
// (profiles/r06_ab_copy_wg.txt, 4.5 GB of BAM -> 9.8 GB of text to /dev/null, one box, alternating): 1 x 4 waves 30.2-30.4 M pairs/s
// (the copy itself is the longest party), 2 x 4: 38.5-39.4, 3 x 4: 37.6-37.8, 4 x 4: 36.8-37.2, 8 x 4: 34.4, 16 x 4: 31.8-31.9 (the
// default until the printer was ordered in front of the inflate launch: then it was the best of the sweep, r06_ab_bam_bins.txt).
// Eight waves, one per XCD, keep the link at ~50 GB/s and the GPU's window (19.5 ms) and the copy's (18 ms) level.  The SAM path
// shares the setting and does not care (r06_ab_sam_copy_wg.txt).  XM_BAMDEV_COPY_WG overrides (0: hipMemcpyAsync instead); A/B only:
// XM_BAMDEV_COPY_WAVES (any number of waves) and XM_BAMDEV_COPY_DEEP=8 (eight pieces per lane in flight) -- 8 x 4 is on a flat
// optimum (r06_ab_copy_deep.txt: 10 x 4 and 6 x 8 the same, 8 x 8 and 6 x 4 3 - 8 % slower).
static uint32_t out_copy_workgroups()
{
    static const uint32_t wg = [] {
        const char *v = getenv("XM_BAMDEV_COPY_WG");
        const long n = v && *v ? strtol(v, nullptr, 10) : 2;
        return (uint32_t)(n < 0 ? 0 : n > 4096 ? 4096 : n);
    }();
    return wg;
}

// ---- the stream the output copy runs on ----
// The copy of window k has to run BESIDE the compute stream's work on window k + 1, and it does only if the two streams lie on
// different hardware queues.  The runtime deals streams onto a few queues (four per priority by default) by rules of its own, and which
// two share one depends on how many streams the process has made before: inside bench.py the copy stream of one slot came to lie on
// the queue of the other slot's compute stream, whose inflate launch then waited for the whole copy -- every other window 38 ms
// instead of 21, 31 M pairs/s instead of 40 (profiles/r06_ab_queue_collision.txt, tools/probe_queue_collision.py).  So the library
// does not guess: it makes candidates and TRIES each against the compute streams -- a kernel on the compute stream waits (bounded:
// 3 ms) for a word that a kernel on the candidate sets; on one queue the second cannot start before the first has ended, and the
// word is never seen.  One copy stream serves both slots (two copies never need to overlap).  With 0 - 9 streams made by the
// process beforehand the BAM path then runs at 39.0 - 40.6 M pairs/s throughout; a first candidate of the greatest priority (its
// queues are apart from the ordinary ones on this runtime; XM_COPY_STREAM_PRIORITY=1) passes the test but copies 8 % slower in some
// of those processes (34.8 - 38.9), and is not the default.  XM_COPY_STREAM_PROBE=0: the first candidate untested (A/B).
__global__ void __launch_bounds__(64)
probe_wait_kernel(volatile uint32_t *started_host, uint32_t *word, uint32_t *seen, unsigned long long max_ticks)
{
    if (threadIdx.x != 0u) return;
    *started_host = 1u;
    __threadfence_system();
    const unsigned long long t0 = wall_clock64();                              // 100 MHz
    uint32_t v = 0;
    while ((v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && wall_clock64() - t0 < max_ticks)
        __builtin_amdgcn_s_sleep(16);
    *seen = v;
}

__global__ void __launch_bounds__(64)
probe_set_kernel(uint32_t *word)
{
    if (threadIdx.x == 0u) __hip_atomic_store(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// true: work on `b` ran while a kernel on `a` was still running (or the test could not be made: then nothing is known against b)
static bool streams_run_side_by_side(hipStream_t a, hipStream_t b)
{
    uint32_t *d = nullptr, *h = nullptr;
    if (hipMalloc((void **)&d, 2 * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return true; }
    if (hipHostMalloc((void **)&h, 2 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); return true; }
    bool beside = true;
    h[0] = 0; h[1] = 0;
    if (hipMemsetAsync(d, 0, 2 * sizeof(uint32_t), a) == hipSuccess && hipStreamSynchronize(a) == hipSuccess) {
        probe_wait_kernel<<<1, 64, 0, a>>>(h, d, d + 1, 300000ull);            // at most 3 ms
        const auto t0 = std::chrono::steady_clock::now();
        bool running = false;
        while (!(running = *(volatile uint32_t *)h != 0u) &&
               std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.05) { }
        probe_set_kernel<<<1, 64, 0, b>>>(d);
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        uint32_t seen = 1;
        if (running && hipMemcpy(&seen, d + 1, sizeof seen, hipMemcpyDeviceToHost) == hipSuccess) beside = seen != 0u;
    }
    (void)hipGetLastError();
    (void)hipFree(d);
    (void)hipHostFree(h);
    return beside;
}

// (also what makes a compute stream that has to run beside others: the SAM front end's second slot)
static hipError_t create_copy_stream(hipStream_t *st, const hipStream_t *compute, int n_compute)
{
    static const bool probe = [] { const char *v = getenv("XM_COPY_STREAM_PROBE"); return !(v && v[0] == '0'); }();
    static const bool prio = [] { const char *v = getenv("XM_COPY_STREAM_PRIORITY"); return v && v[0] == '1'; }();
    constexpr int TRIES = 8;
    hipStream_t tried[TRIES] = {};
    int n = 0, pick = -1;
    hipError_t e = hipSuccess;
    for (; n < TRIES && pick < 0; ++n) {
        int least = 0, greatest = 0;
        if (n == 0 && prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            e = hipStreamCreateWithPriority(&tried[n], hipStreamNonBlocking, greatest);
        else
            e = hipStreamCreateWithFlags(&tried[n], hipStreamNonBlocking);
        if (e != hipSuccess) break;
        bool ok = true;
        for (int c = 0; probe && ok && c < n_compute; ++c) ok = streams_run_side_by_side(compute[c], tried[n]);
        if (ok) pick = n;
    }
    if (pick < 0 && n > 0 && tried[0]) pick = 0;                                // none runs beside: the first one, as before
    for (int k = 0; k < TRIES; ++k)
        if (tried[k] && k != pick) (void)hipStreamDestroy(tried[k]);
    if (pick < 0) return e != hipSuccess ? e : hipErrorUnknown;
    *st = tried[pick];
    return hipSuccess;
}

}  // namespace
