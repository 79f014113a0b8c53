// xm_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the xenograft read classifier.
//
// Three stages, all HBM-bound integer/byte work (no MFMA):
//   K1 classify   score columns -> one category byte per record     (16 B in, 1 B out / record)
//   K2 compact    category bytes -> category_counts + a stable split of unit indices by bin
//                 K2a histogram (LDS, 64 slots x 32 replicas) -> per-chunk bin counts + counts[64]
//                 K2b scan of the per-chunk counts (one workgroup)
//                 K2c scatter through an LDS-staged, bin-sorted tile -> coalesced index stores
//   K3 cigar      NM + packed CIGAR (CSR) -> synthesised AS column
//
// Reference semantics restated (file:line into /root/reference/xenomapper/xenomapper.py):
//   get_mapping_state :258-289, pair rules :423-448 / :521-550, unit rule :402-405,
//   get_cigarbased_AS_tag :228-256.  The oracle (oracle/) is the checker; nothing here calls it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xm_kernels.h"

namespace xm {

// ---------------------------------------------------------------------------------------------
// state function: branch-flattened select chain, evaluated per lane
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ uint32_t mapping_state(T a1, T x1, T a2, T x2, T m)
{
    const bool low1 = a1 <= m;                        // :275  (written as the reference writes it,
    const bool low2 = a2 <= m;                        //        so NaN behaves identically in f64)
    const bool prim = (a1 > m) && (low2 || a1 > a2);  // :277
    const bool sec  = (a2 > m) && (low1 || a2 > a1);  // :284
    const bool spec1 = (x1 == T(0)) || (a1 > x1);     // :278  `not XS1 or AS1 > XS1`
    const bool spec2 = (x2 == T(0)) || (a2 > x2);     // :285
    uint32_t s = 6u;                                  // :289  fall-through (NaN only)
    s = sec ? (spec2 ? 1u : 3u) : s;                  // :284-288
    s = (a1 == a2) ? 4u : s;                          // :282-283
    s = prim ? (spec1 ? 0u : 2u) : s;                 // :277-281
    s = (low1 && low2) ? 5u : s;                      // :275-276
    return s;
}

// output bin of a category code (state, or fwd*8+rev).  7 = not a unit.
__device__ __forceinline__ uint32_t bin_of_code(int mode, uint32_t c)
{
    const uint32_t f = (c >> 3) & 7u, r = c & 7u;
    const uint32_t lo = f < r ? f : r;
    const uint32_t hi = f < r ? r : f;
    uint32_t b;
    if (mode == XM_MODE_SE) {
        b = r > 6u ? 6u : r;
    } else if (mode == XM_MODE_PE_LIBERAL) {
        b = lo;                                                  // :423-448 == min()
    } else {
        b = (hi == 5u) ? 5u                                      // :521
          : ((hi == 4u) || (((f ^ r) & 1u) != 0u)) ? 4u          // :525-529
          : lo;                                                  // :535-550
    }
    if (mode != XM_MODE_SE) b = (hi > 5u) ? 6u : b;
    return (c == XM_NO_UNIT) ? 7u : b;
}

// ---------------------------------------------------------------------------------------------
// K1: classify
// ---------------------------------------------------------------------------------------------
template <typename T> struct Vec4;
template <> struct Vec4<int32_t> { typedef int4 type; };
template <> struct Vec4<double>  { typedef double4 type; };

template <typename T> __device__ __forceinline__ T absent_value();
template <> __device__ __forceinline__ int32_t absent_value<int32_t>() { return INT32_MIN; }
template <> __device__ __forceinline__ double  absent_value<double>()  { return -__builtin_huge_val(); }

template <typename T>
__device__ __forceinline__ void load4(const T *__restrict__ col, uint64_t r0, uint64_t n, T out[4])
{
    typedef typename Vec4<T>::type V;
    if (r0 + 4 <= n) {
        const V v = *reinterpret_cast<const V *>(col + r0);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = (r0 + j < n) ? col[r0 + j] : absent_value<T>();
    }
}

// One lane owns 4 consecutive records, so each column is read with one 16-byte (int32) load per
// lane, 1 KiB contiguous per wave instruction.  The forward mate's state of a lane's first record
// comes from lane-1 (shuffle); lane 0 takes it from the record just before the wave's tile, whose
// four scores are fetched with wave-uniform (scalar) loads.
template <typename T, bool PAIRED>
__global__ void __launch_bounds__(XM_BLOCK)
classify_kernel(const T *__restrict__ as1, const T *__restrict__ xs1,
                const T *__restrict__ as2, const T *__restrict__ xs2,
                const uint8_t *__restrict__ unit_bits8, T m,
                uint8_t *__restrict__ code, uint64_t n)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t n_groups = (n + 3) >> 2;
    const uint64_t n_wtiles = (n_groups + 63) >> 6;
    const uint64_t n_waves = (uint64_t)gridDim.x * (XM_BLOCK / 64);

    for (uint64_t wt = (uint64_t)blockIdx.x * (XM_BLOCK / 64) + wave_in_block; wt < n_wtiles; wt += n_waves) {
        const uint64_t g = wt * 64 + lane;
        const uint64_t r0 = g * 4;

        T a1[4], x1[4], a2[4], x2[4];
        load4(as1, r0, n, a1);
        load4(xs1, r0, n, x1);
        load4(as2, r0, n, a2);
        load4(xs2, r0, n, x2);
        uint32_t mb = 0;
        if (r0 < n) mb = (uint32_t)(unit_bits8[g >> 1] >> ((g & 1u) * 4u)) & 0xFu;
        if (r0 + 4 > n) mb &= (r0 < n) ? ((1u << (uint32_t)(n - r0)) - 1u) : 0u;

        uint32_t s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = mapping_state<T>(a1[j], x1[j], a2[j], x2[j], m);

        uint32_t c[4];
        if (PAIRED) {
            uint32_t prev = (uint32_t)__shfl_up((int)s[3], 1, 64);
            const uint64_t tile_first = wt * 256;              // wave-uniform
            if (tile_first > 0) {
                const uint64_t h = tile_first - 1;             // exists: h < n because wt < n_wtiles
                const uint32_t hs = mapping_state<T>(as1[h], xs1[h], as2[h], xs2[h], m);
                prev = (lane == 0) ? hs : prev;
            } else {
                mb &= (lane == 0) ? ~1u : ~0u;                 // record 0 has no predecessor (:402)
            }
            c[0] = (mb & 1u) ? ((prev << 3) | s[0]) : XM_NO_UNIT;
            c[1] = (mb & 2u) ? ((s[0] << 3) | s[1]) : XM_NO_UNIT;
            c[2] = (mb & 4u) ? ((s[1] << 3) | s[2]) : XM_NO_UNIT;
            c[3] = (mb & 8u) ? ((s[2] << 3) | s[3]) : XM_NO_UNIT;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = ((mb >> j) & 1u) ? s[j] : XM_NO_UNIT;
        }

        if (r0 + 4 <= n) {
            *reinterpret_cast<uint32_t *>(code + r0) = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r0 + j < n) code[r0 + j] = (uint8_t)c[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2 shared: load one thread's 16 category bytes of a 4096-record tile
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_codes16(const uint8_t *__restrict__ code, uint64_t base, uint64_t n,
                                             uint32_t w[4])
{
    if (base + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(code + base);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t i = base + 4 * k + j;
                const uint32_t c = (i < n) ? (uint32_t)code[i] : XM_NO_UNIT;
                acc |= c << (8 * j);
            }
            w[k] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2a: histogram.  Workgroup `chunk` owns tiles [chunk*tpc, (chunk+1)*tpc).  LDS holds 64 category
// slots x 32 replicas (replica = lane & 31, so a wave's 64 atomics spread over all 32 banks).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(XM_BLOCK)
hist_kernel(const uint8_t *__restrict__ code, uint64_t n, int mode, uint32_t tiles_per_chunk,
            uint32_t *__restrict__ chunk_counts, unsigned long long *__restrict__ counts)
{
    __shared__ uint32_t hist[64 * 32];
    __shared__ uint32_t binc[8];
    const uint32_t t = threadIdx.x;
    for (uint32_t k = t; k < 64 * 32; k += XM_BLOCK) hist[k] = 0;
    if (t < 8) binc[t] = 0;
    __syncthreads();

    const uint64_t n_tiles = (n + XM_TILE - 1) / XM_TILE;
    const uint64_t tile0 = (uint64_t)blockIdx.x * tiles_per_chunk;
    uint64_t tile1 = tile0 + tiles_per_chunk;
    if (tile1 > n_tiles) tile1 = n_tiles;
    const uint32_t rep = t & 31u;

    for (uint64_t tile = tile0; tile < tile1; ++tile) {
        uint32_t w[4];
        load_codes16(code, tile * XM_TILE + (uint64_t)t * 16, n, w);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t c = (w[k] >> (8 * j)) & 0xFFu;
                const uint32_t slot = (c == XM_NO_UNIT) ? 63u : (c & 63u);
                atomicAdd(&hist[slot * 32 + rep], 1u);
            }
        }
    }
    __syncthreads();

    // 4 threads per slot, 8 replicas each
    const uint32_t slot = t >> 2, q = t & 3u;
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += hist[slot * 32 + q * 8 + r];
    s += (uint32_t)__shfl_xor((int)s, 1, 64);
    s += (uint32_t)__shfl_xor((int)s, 2, 64);
    if (q == 0 && slot != 63u && s != 0u) {
        atomicAdd(&counts[slot], (unsigned long long)s);
        atomicAdd(&binc[bin_of_code(mode, slot)], s);
    }
    __syncthreads();
    if (t < 8) chunk_counts[(uint64_t)blockIdx.x * 8 + t] = binc[t];
}

// ---------------------------------------------------------------------------------------------
// K2b: exclusive scan of the per-chunk bin counts (one workgroup, wave b scans bin b).
// chunk_off[k][b] = start of chunk k's units of bin b inside idx_out; bin_offsets[0..7].
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
scan_kernel(const uint32_t *__restrict__ chunk_counts, uint32_t n_chunks,
            uint32_t *__restrict__ chunk_off, unsigned long long *__restrict__ bin_offsets)
{
    __shared__ uint32_t tot[8];
    const uint32_t b = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t per = (n_chunks + 63u) / 64u;
    uint32_t k0 = lane * per, k1 = k0 + per;
    if (k0 > n_chunks) k0 = n_chunks;
    if (k1 > n_chunks) k1 = n_chunks;
    uint32_t sum = 0;
    for (uint32_t k = k0; k < k1; ++k) sum += chunk_counts[(uint64_t)k * 8 + b];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        incl += (lane >= (uint32_t)d) ? up : 0u;
    }
    if (lane == 63) tot[b] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t bb = 0; bb < b; ++bb) base += tot[bb];
    uint32_t run = base + incl - sum;
    for (uint32_t k = k0; k < k1; ++k) {
        chunk_off[(uint64_t)k * 8 + b] = run;
        run += chunk_counts[(uint64_t)k * 8 + b];
    }
    if (lane == 0) {
        // bins 0..6 are real; slot 7 of the counts is unused (always 0) so bin_offsets[7] = #units
        bin_offsets[b] = base;
    }
}

// ---------------------------------------------------------------------------------------------
// K2c: scatter.  Per 4096-record tile: per-lane bin counts -> wave scan -> tile-local sorted
// position of every unit -> record indices staged in LDS in (bin, input order) -> written out as
// contiguous runs, one run per bin.  Eight 16-bit counters live in two 64-bit registers.
// ---------------------------------------------------------------------------------------------
struct Packed8 {
    uint64_t lo, hi;      // lo: bins 0..3, hi: bins 4..7 (7 = "not a unit"), 16 bits each
};

__device__ __forceinline__ void packed_add(Packed8 &p, uint32_t bin)
{
    const uint64_t one = 1ull << ((bin & 3u) * 16u);
    p.lo += (bin < 4u) ? one : 0ull;
    p.hi += (bin < 4u) ? 0ull : one;
}

__device__ __forceinline__ uint32_t packed_get(const Packed8 &p, uint32_t bin)
{
    const uint64_t w = (bin < 4u) ? p.lo : p.hi;
    return (uint32_t)(w >> ((bin & 3u) * 16u)) & 0xFFFFu;
}

__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int d)
{
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, 64);
    const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64);
    return ((uint64_t)hi << 32) | lo;
}

__global__ void __launch_bounds__(XM_BLOCK)
scatter_kernel(const uint8_t *__restrict__ code, uint64_t n, int mode, uint32_t tiles_per_chunk,
               const uint32_t *__restrict__ chunk_off, uint32_t *__restrict__ idx_out)
{
    __shared__ uint32_t stage[XM_TILE];
    __shared__ uint32_t wave_tot[XM_BLOCK / 64][8];
    __shared__ uint32_t wave_base[XM_BLOCK / 64][8];
    __shared__ uint32_t tile_start[8];      // start of bins 0..6 in the sorted tile; [7] = units in tile
    __shared__ uint32_t tile_cnt[8];
    __shared__ uint32_t run[8];

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t < 8) run[t] = chunk_off[(uint64_t)blockIdx.x * 8 + t];

    const uint64_t n_tiles = (n + XM_TILE - 1) / XM_TILE;
    const uint64_t tile0 = (uint64_t)blockIdx.x * tiles_per_chunk;
    uint64_t tile1 = tile0 + tiles_per_chunk;
    if (tile1 > n_tiles) tile1 = n_tiles;

    for (uint64_t tile = tile0; tile < tile1; ++tile) {
        const uint64_t base = tile * XM_TILE + (uint64_t)t * 16;
        uint32_t w[4];
        load_codes16(code, base, n, w);

        // A: per-lane counts
        uint32_t bins[16];
        Packed8 cnt = {0ull, 0ull};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            bins[j] = bin_of_code(mode, (w[j >> 2] >> (8 * (j & 3))) & 0xFFu);
            packed_add(cnt, bins[j]);
        }
        // B: inclusive scan over the wave
        Packed8 incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t ulo = shfl_up64(incl.lo, d), uhi = shfl_up64(incl.hi, d);
            if (lane >= (uint32_t)d) { incl.lo += ulo; incl.hi += uhi; }
        }
        // C: wave totals
        if (lane == 63) {
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) wave_tot[wave][b] = packed_get(incl, b);
        }
        __syncthreads();
        // D: tile layout (lanes 0..7 of wave 0, lane b owns bin b)
        if (t < 8) {
            uint32_t tot = 0, pre[XM_BLOCK / 64];
#pragma unroll
            for (int wv = 0; wv < XM_BLOCK / 64; ++wv) { pre[wv] = tot; tot += wave_tot[wv][t]; }
            const uint32_t real = (t < 7) ? tot : 0u;          // bin 7 = not-a-unit, takes no room
            uint32_t incl8 = real;
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl8, d, 64);
                incl8 += (t >= (uint32_t)d) ? up : 0u;
            }
            const uint32_t start = incl8 - real;               // for t == 7 this is the unit total
            tile_start[t] = start;
            tile_cnt[t] = real;
#pragma unroll
            for (int wv = 0; wv < XM_BLOCK / 64; ++wv) wave_base[wv][t] = start + pre[wv];
        }
        __syncthreads();
        // E: ranks -> LDS stage
        Packed8 pos;
        {
            uint64_t blo = 0, bhi = 0;
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) {
                blo |= (uint64_t)wave_base[wave][b] << (16 * b);
                bhi |= (uint64_t)(wave_base[wave][b + 4] & 0xFFFFu) << (16 * b);
            }
            pos.lo = blo + (incl.lo - cnt.lo);
            pos.hi = bhi + (incl.hi - cnt.hi);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t b = bins[j];
            const uint32_t p = packed_get(pos, b);
            if (b < 7u) stage[p] = (uint32_t)(base + j);
            packed_add(pos, b);
        }
        __syncthreads();
        // F: coalesced write-out, one contiguous run per bin
        const uint32_t total = tile_start[7];
        for (uint32_t e = t; e < total; e += XM_BLOCK) {
            uint32_t b = 0;
#pragma unroll
            for (uint32_t k = 1; k < 7; ++k) b += (e >= tile_start[k]) ? 1u : 0u;
            idx_out[run[b] + (e - tile_start[b])] = stage[e];
        }
        __syncthreads();
        // G: advance the chunk's running offsets
        if (t < 7) run[t] += tile_cnt[t];
    }
}

// ---------------------------------------------------------------------------------------------
// K3: CIGAR-derived AS (one lane per record, CSR ops)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(XM_BLOCK)
cigar_kernel(uint64_t n, const int32_t *__restrict__ nm, const uint32_t *__restrict__ cig_off,
             const uint32_t *__restrict__ cig_oplen, int32_t *__restrict__ as_out,
             uint32_t *__restrict__ range_flag)
{
    const uint64_t stride = (uint64_t)gridDim.x * XM_BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * XM_BLOCK + threadIdx.x; i < n; i += stride) {
        const int32_t nmv = nm[i];
        int32_t out = INT32_MIN;
        if (nmv != INT32_MIN) {
            long long s = -6ll * (long long)nmv;                           // :250, :255
            const uint32_t k1 = cig_off[i + 1];
            for (uint32_t k = cig_off[i]; k < k1; ++k) {
                const uint32_t v = cig_oplen[k];
                const uint32_t op = v & 15u;
                const long long len = (long long)(v >> 4);
                s -= (op == 1u || op == 2u) ? (5ll + 3ll * len) : 0ll;     // I, D: open + extend
                s -= (op == 4u) ? 2ll * len : 0ll;                         // S
            }
            if (s <= (long long)INT32_MIN || s > (long long)INT32_MAX) {
                if (range_flag) atomicOr(range_flag, 1u);
                s = s < 0 ? (long long)INT32_MIN + 1 : (long long)INT32_MAX;
            }
            out = (int32_t)s;
        }
        as_out[i] = out;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline uint32_t classify_grid(uint64_t n, uint32_t max_blocks)
{
    const uint64_t n_wtiles = (((n + 3) >> 2) + 63) >> 6;
    uint64_t blocks = (n_wtiles + (XM_BLOCK / 64) - 1) / (XM_BLOCK / 64);
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks == 0) blocks = 1;
    return (uint32_t)blocks;
}

template <typename T>
static void launch_classify_t(hipStream_t st, uint32_t max_blocks, int mode, uint64_t n,
                              const T *as1, const T *xs1, const T *as2, const T *xs2,
                              const uint64_t *unit_bits, T m, uint8_t *code)
{
    const uint32_t grid = classify_grid(n, max_blocks);
    const uint8_t *bits8 = reinterpret_cast<const uint8_t *>(unit_bits);
    if (mode == XM_MODE_SE)
        classify_kernel<T, false><<<grid, XM_BLOCK, 0, st>>>(as1, xs1, as2, xs2, bits8, m, code, n);
    else
        classify_kernel<T, true><<<grid, XM_BLOCK, 0, st>>>(as1, xs1, as2, xs2, bits8, m, code, n);
}

void launch_classify_i32(hipStream_t st, uint32_t max_blocks, int mode, uint64_t n,
                         const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                         const uint64_t *unit_bits, int32_t m, uint8_t *code)
{
    launch_classify_t<int32_t>(st, max_blocks, mode, n, as1, xs1, as2, xs2, unit_bits, m, code);
}

void launch_classify_f64(hipStream_t st, uint32_t max_blocks, int mode, uint64_t n,
                         const double *as1, const double *xs1, const double *as2, const double *xs2,
                         const uint64_t *unit_bits, double m, uint8_t *code)
{
    launch_classify_t<double>(st, max_blocks, mode, n, as1, xs1, as2, xs2, unit_bits, m, code);
}

ChunkPlan plan_chunks(uint64_t n, uint32_t max_chunks)
{
    ChunkPlan p;
    const uint64_t n_tiles = (n + XM_TILE - 1) / XM_TILE;
    uint64_t chunks = n_tiles < max_chunks ? n_tiles : max_chunks;
    if (chunks == 0) chunks = 1;
    p.tiles_per_chunk = (uint32_t)((n_tiles + chunks - 1) / chunks);
    if (p.tiles_per_chunk == 0) p.tiles_per_chunk = 1;
    p.n_chunks = (uint32_t)((n_tiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk);
    if (p.n_chunks == 0) p.n_chunks = 1;
    return p;
}

void launch_hist(hipStream_t st, const ChunkPlan &p, int mode, uint64_t n, const uint8_t *code,
                 uint32_t *chunk_counts, uint64_t *counts)
{
    hist_kernel<<<p.n_chunks, XM_BLOCK, 0, st>>>(code, n, mode, p.tiles_per_chunk, chunk_counts,
                                                  reinterpret_cast<unsigned long long *>(counts));
}

void launch_scan(hipStream_t st, const ChunkPlan &p, const uint32_t *chunk_counts, uint32_t *chunk_off,
                 uint64_t *bin_offsets)
{
    scan_kernel<<<1, 512, 0, st>>>(chunk_counts, p.n_chunks, chunk_off,
                                   reinterpret_cast<unsigned long long *>(bin_offsets));
}

void launch_scatter(hipStream_t st, const ChunkPlan &p, int mode, uint64_t n, const uint8_t *code,
                    const uint32_t *chunk_off, uint32_t *idx_out)
{
    scatter_kernel<<<p.n_chunks, XM_BLOCK, 0, st>>>(code, n, mode, p.tiles_per_chunk, chunk_off, idx_out);
}

void launch_cigar(hipStream_t st, uint32_t max_blocks, uint64_t n, const int32_t *nm, const uint32_t *cig_off,
                  const uint32_t *cig_oplen, int32_t *as_out, uint32_t *range_flag)
{
    uint64_t blocks = (n + XM_BLOCK - 1) / XM_BLOCK;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks == 0) blocks = 1;
    cigar_kernel<<<(uint32_t)blocks, XM_BLOCK, 0, st>>>(n, nm, cig_off, cig_oplen, as_out, range_flag);
}

}  // namespace xm
