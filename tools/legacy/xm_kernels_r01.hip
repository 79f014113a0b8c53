// FROZEN COPY of the round-1 kernels (git f617191), kept only so that tools/tune_kernels.hip can time old against new
// on the same box in one process.  Not built into any product library.  Compile with -Dxm=xm_r01.
// xm_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the xenograft read classifier.
//
// Three stages, all HBM-bound integer/byte work (no MFMA):
//   K1 classify   score columns -> one category byte per record     (16 B in, 1 B out / record)
//   K2 compact    category bytes -> category_counts + a stable split of unit indices by bin
//                 K2a histogram (LDS, 64 slots x 32 replicas) -> per-chunk bin counts + counts[64]
//                 K2b scan of the per-chunk counts (one workgroup)
//                 K2c scatter: per-wave DPP scans, wave-private LDS slab -> contiguous index runs
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
typedef int32_t v4i32 __attribute__((ext_vector_type(4)));
typedef int32_t v4i32_a4 __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte access
typedef double  v4f64 __attribute__((ext_vector_type(4)));
template <> struct Vec4<int32_t> { typedef v4i32 type; };
template <> struct Vec4<double>  { typedef v4f64 type; };

template <typename T> __device__ __forceinline__ T absent_value();
template <> __device__ __forceinline__ int32_t absent_value<int32_t>() { return INT32_MIN; }
template <> __device__ __forceinline__ double  absent_value<double>()  { return -__builtin_huge_val(); }

// FULL: the caller knows (workgroup-uniformly) that all 4 records exist, so no bounds test is compiled in
template <typename T, bool NT, bool FULL = false>
__device__ __forceinline__ void load4(const T *__restrict__ col, uint64_t r0, uint64_t n, T out[4])
{
    typedef typename Vec4<T>::type V;
    if (FULL || r0 + 4 <= n) {
        const V *p = reinterpret_cast<const V *>(col + r0);
        const V v = NT ? __builtin_nontemporal_load(p) : *p;
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = (r0 + j < n) ? col[r0 + j] : absent_value<T>();
    }
}

// Shared K1 epilogue: 4 states per lane -> forward mate's state (lane-1 / previous wave via LDS / halo) ->
// 4 category bytes -> one 4-byte store.
template <typename T, bool PAIRED, int BLOCK, bool FULL>
__device__ __forceinline__ void classify_finish(const T a1[4], const T x1[4], const T a2[4], const T x2[4], T m,
                                                uint32_t mb, uint32_t halo, uint32_t *last_state,
                                                uint8_t *__restrict__ code, uint64_t r0, uint64_t n)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = mapping_state<T>(a1[j], x1[j], a2[j], x2[j], m);

    uint32_t c[4];
    if (PAIRED) {
        uint32_t prev = (uint32_t)__shfl_up((int)s[3], 1, 64);
        if (lane == 63) last_state[wave] = s[3];
        __syncthreads();
        if (lane == 0) prev = (wave == 0) ? halo : last_state[wave - 1];
        c[0] = (mb & 1u) ? ((prev << 3) | s[0]) : XM_NO_UNIT;
        c[1] = (mb & 2u) ? ((s[0] << 3) | s[1]) : XM_NO_UNIT;
        c[2] = (mb & 4u) ? ((s[1] << 3) | s[2]) : XM_NO_UNIT;
        c[3] = (mb & 8u) ? ((s[2] << 3) | s[3]) : XM_NO_UNIT;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = ((mb >> j) & 1u) ? s[j] : XM_NO_UNIT;
    }

    if (FULL || r0 + 4 <= n) {
        *reinterpret_cast<uint32_t *>(code + r0) = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r0 + j < n) code[r0 + j] = (uint8_t)c[j];
    }
}

// One lane owns 4 consecutive records, so each column is read with one 16-byte (int32) load per
// lane, 1 KiB contiguous per wave instruction; a wave covers a 256-record tile, a workgroup BLOCK*4
// consecutive records, and the grid covers the whole input once (no grid-stride loop: on MI355X the
// one-tile-per-wave launch measured 20-25 % faster than a persistent 2048-workgroup loop, see
// profiles/r01_tune_classify.txt).  The forward mate's state of a lane's first record comes from
// lane-1 (shuffle); lane 0 takes it from the previous wave of the workgroup through LDS, and the
// first wave of a workgroup from the one record in front of the workgroup's range.
// NT: the score columns are read once and never again, so they are loaded non-temporally; the
// category bytes are stored with the default policy because K2 reads them next (100 MB at the
// 50 M-pair configuration, which fits the 256 MiB Infinity Cache).
template <typename T, bool PAIRED, bool NT, int BLOCK, bool FULL>
__device__ __forceinline__ void classify_body(const T *__restrict__ as1, const T *__restrict__ xs1,
                                              const T *__restrict__ as2, const T *__restrict__ xs2,
                                              const uint8_t *__restrict__ unit_bits8, T m,
                                              uint8_t *__restrict__ code, uint64_t n, uint32_t *last_state)
{
    const uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;       // group of 4 records
    const uint64_t r0 = g * 4;

    T a1[4], x1[4], a2[4], x2[4];
    load4<T, NT, FULL>(as1, r0, n, a1);
    load4<T, NT, FULL>(xs1, r0, n, x1);
    load4<T, NT, FULL>(as2, r0, n, a2);
    load4<T, NT, FULL>(xs2, r0, n, x2);
    uint32_t mb = 0;
    if (FULL || r0 < n) mb = (uint32_t)(unit_bits8[g >> 1] >> ((g & 1u) * 4u)) & 0xFu;
    if (!FULL && r0 + 4 > n) mb &= (r0 < n) ? ((1u << (uint32_t)(n - r0)) - 1u) : 0u;

    // the record in front of the workgroup's range (thread 0 only)
    uint32_t halo = 0;
    if (PAIRED && threadIdx.x == 0) {
        if (r0 > 0) {
            const uint64_t h = r0 - 1;                         // thread 0's r0 < n (grid sizing), so h < n
            halo = mapping_state<T>(as1[h], xs1[h], as2[h], xs2[h], m);
        } else {
            mb &= ~1u;                                         // record 0 has no predecessor (:402)
        }
    }
    classify_finish<T, PAIRED, BLOCK, FULL>(a1, x1, a2, x2, m, mb, halo, last_state, code, r0, n);
}

template <typename T, bool PAIRED, bool NT, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
classify_kernel(const T *__restrict__ as1, const T *__restrict__ xs1,
                const T *__restrict__ as2, const T *__restrict__ xs2,
                const uint8_t *__restrict__ unit_bits8, T m,
                uint8_t *__restrict__ code, uint64_t n)
{
    __shared__ uint32_t last_state[BLOCK / 64];
    // every workgroup but possibly the last covers BLOCK*4 existing records: no bounds tests on that path
    if (((uint64_t)blockIdx.x + 1) * (BLOCK * 4) <= n)
        classify_body<T, PAIRED, NT, BLOCK, true>(as1, xs1, as2, xs2, unit_bits8, m, code, n, last_state);
    else
        classify_body<T, PAIRED, NT, BLOCK, false>(as1, xs1, as2, xs2, unit_bits8, m, code, n, last_state);
}

// ---------------------------------------------------------------------------------------------
// K2 shared: load one thread's 16 category bytes of a 4096-record tile
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((noinline)) uint4 load_codes16_tail(const uint8_t *__restrict__ code, uint64_t base, uint64_t n)
{
    uint32_t w[4];
    for (int k = 0; k < 4; ++k) {
        uint32_t acc = 0;
        for (int j = 0; j < 4; ++j) {
            const uint64_t i = base + 4 * k + j;
            const uint32_t c = (i < n) ? (uint32_t)code[i] : XM_NO_UNIT;
            acc |= c << (8 * j);
        }
        w[k] = acc;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ void load_codes16(const uint8_t *__restrict__ code, uint64_t base, uint64_t n,
                                             uint32_t w[4])
{
    uint4 v;
    if (base + 16 <= n) v = *reinterpret_cast<const uint4 *>(code + base);
    else v = load_codes16_tail(code, base, n);          // last, partial tile only
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}

// ---------------------------------------------------------------------------------------------
// K2 geometry.  A wave owns XM_K consecutive wave tiles of XM_WTILE records (16 category bytes per
// lane, lane l holds records 16l..16l+15 of the tile, so lane order == input order); a workgroup
// (chunk) owns 4 consecutive wave spans = XM_CHUNK records.  All of a wave's category bytes stay in
// registers between counting and scattering, so the bytes are read from memory once per kernel.
// ---------------------------------------------------------------------------------------------

// wave64 inclusive prefix sum with DPP (row_shr 1,2,4,8, row_bcast15, row_bcast31): no LDS traffic
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

__device__ __forceinline__ uint32_t lane_value(uint32_t v, int lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// ---------------------------------------------------------------------------------------------
// K2a: histogram.  LDS holds 64 category slots x 32 replicas (replica = lane & 31, so the 64 atomics
// of a wave instruction spread over all 32 banks).  A byte position at which no lane of the wave
// holds a unit (every other position of interleaved paired input) is skipped wave-uniformly.
// Outputs: counts_rep[replica][64] (partial category_counts) and chunk_counts[bin][chunk] for the scan.
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(XM_BLOCK)
hist_kernel(const uint8_t *__restrict__ code, uint64_t n, int mode, uint32_t chunk_stride,
            uint32_t *__restrict__ chunk_counts, unsigned long long *__restrict__ counts_rep)
{
    __shared__ uint32_t hist[64 * 32];
    __shared__ uint32_t binc[8];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    for (uint32_t k = t; k < 64 * 32; k += XM_BLOCK) hist[k] = 0;
    if (t < 8) binc[t] = 0;
    __syncthreads();

    const uint64_t span0 = ((uint64_t)blockIdx.x * (XM_BLOCK / 64) + wave) * (uint64_t)(K * XM_WTILE);
    uint32_t w[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) load_codes16(code, span0 + (uint64_t)k * XM_WTILE + lane * 16u, n, w[k]);

    const uint32_t rep = t & 31u;
    // One category usually dominates (both mates primary-specific in a xenograft).  The wave takes the first unit of
    // its first lane as its guess `common`; units of that category are counted in a register and reach LDS with one
    // atomic per lane at the end, so the per-position atomics carry only the other lanes -- fewer of the lane pairs
    // (l, l + 32) that share a bank are both active, and the instruction mostly takes one pass instead of two.
    uint32_t common = w[0][0] & 0xFFu;
    if (common == XM_NO_UNIT) common = (w[0][0] >> 8) & 0xFFu;
    common = (uint32_t)__builtin_amdgcn_readfirstlane((int)common);
    uint32_t n_common = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t c = (w[k][j >> 2] >> (8 * (j & 3))) & 0xFFu;
            const bool unit = c != XM_NO_UNIT;
            if (__ballot(unit) == 0ull) continue;                     // wave-uniform
            const bool same = unit && c == common;
            n_common += same ? 1u : 0u;
            if (unit && !same) atomicAdd(&hist[(c & 63u) * 32 + rep], 1u);
        }
    }
    if (n_common) atomicAdd(&hist[(common & 63u) * 32 + rep], n_common);
    __syncthreads();

    // 4 threads per slot, 8 replicas each
    const uint32_t slot = t >> 2, q = t & 3u;
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += hist[slot * 32 + q * 8 + r];
    s += (uint32_t)__shfl_xor((int)s, 1, 64);
    s += (uint32_t)__shfl_xor((int)s, 2, 64);
    if (q == 0 && s != 0u) {
        // XM_COUNT_REPLICAS copies of counts[64] keep same-address atomics rare; K2b adds them up
        atomicAdd(&counts_rep[(blockIdx.x % XM_COUNT_REPLICAS) * 64u + slot], (unsigned long long)s);
        atomicAdd(&binc[bin_of_code(mode, slot)], s);
    }
    __syncthreads();
    if (t < 8) chunk_counts[(uint64_t)t * chunk_stride + blockIdx.x] = binc[t];
}

// ---------------------------------------------------------------------------------------------
// K2b: exclusive scan of the per-chunk bin counts.  Workgroup b (1024 threads) scans bin b: each of its
// 16 waves owns a contiguous range of chunks, sums it, and after one barrier rescans it with the
// carry of the ranges before.  chunk_off[b][k] = units of bin b in chunks < k; bin_totals[b] = units of
// bin b.  The workgroups also add up the replicas of category_counts (8 slots each) and zero them again.
// ---------------------------------------------------------------------------------------------
#define XM_SCAN_THREADS 1024
__global__ void __launch_bounds__(XM_SCAN_THREADS)
scan_kernel(const uint32_t *__restrict__ chunk_counts, uint32_t n_chunks, uint32_t chunk_stride,
            uint32_t *__restrict__ chunk_off, unsigned long long *__restrict__ bin_totals,
            unsigned long long *__restrict__ counts_rep, unsigned long long *__restrict__ counts)
{
    __shared__ unsigned long long wsum[XM_SCAN_THREADS / 64];
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) {   // category_counts slots 8b..8b+7: 8 lanes per slot, 8 replicas each
        const uint32_t slot = b * 8u + (threadIdx.x >> 3), part = threadIdx.x & 7u;
        unsigned long long acc = 0;
        for (uint32_t r = part; r < XM_COUNT_REPLICAS; r += 8u) {
            acc += counts_rep[r * 64u + slot];
            counts_rep[r * 64u + slot] = 0;               // consumed: leave the replicas zeroed for the next K2a
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (part == 0) counts[slot] = acc;
    }
    const uint32_t *src = chunk_counts + (uint64_t)b * chunk_stride;
    uint32_t *dst = chunk_off + (uint64_t)b * chunk_stride;
    const uint32_t n_waves = XM_SCAN_THREADS / 64;
    const uint32_t seg = (((n_chunks + n_waves - 1) / n_waves) + 63u) & ~63u;     // chunks per wave, multiple of 64
    const uint32_t k_begin = wave * seg;
    const uint32_t k_end = (k_begin + seg < n_chunks) ? k_begin + seg : n_chunks;

    // pass 1: sum of the wave's range (8 coalesced loads in flight per lane)
    unsigned long long sum = 0;
    for (uint32_t j0 = k_begin; j0 < k_end; j0 += 64u * 8u) {
        uint32_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t k = j0 + (uint32_t)u * 64u + lane;
            x[u] = (k < k_end) ? src[k] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += x[u];
    }
    sum += __shfl_xor(sum, 1, 64);  sum += __shfl_xor(sum, 2, 64);  sum += __shfl_xor(sum, 4, 64);
    sum += __shfl_xor(sum, 8, 64);  sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
    if (lane == 0) wsum[wave] = sum;
    __syncthreads();
    unsigned long long carry = 0, total = 0;
    for (uint32_t wv = 0; wv < n_waves; ++wv) {
        const unsigned long long v = wsum[wv];
        carry += (wv < wave) ? v : 0ull;
        total += v;
    }
    if (threadIdx.x == 0) bin_totals[b] = total;

    // pass 2: rescan the range (now cache-resident) with the carry
    for (uint32_t j0 = k_begin; j0 < k_end; j0 += 64u * 8u) {
        uint32_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t k = j0 + (uint32_t)u * 64u + lane;
            x[u] = (k < k_end) ? src[k] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t k = j0 + (uint32_t)u * 64u + lane;
            const uint32_t incl = wave_scan_incl(x[u]);
            if (k < k_end) dst[k] = (uint32_t)carry + incl - x[u];
            carry += lane_value(incl, 63);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2c: scatter.  Per wave tile: per-lane bin counters packed in one 64-bit register (bins 0..2 as
// 10-bit fields of the low word, 3..5 of the high word, everything else into a 2-bit sink at bit 62)
// -> DPP wave scan -> position of every unit in the tile's (bin, input order) sorted order -> record
// indices staged in the wave's private LDS slab -> each bin's run written out contiguously.
// One workgroup barrier per chunk (to order the four waves' bases); everything else is per wave.
// Units holding state 6 (NaN input only) take a slow path that writes straight to slot 6.
// ---------------------------------------------------------------------------------------------

// shift of bin b's counter inside the packed 64-bit word: byte LUT {0,10,20,32,42,52,62,62} read with
// v_perm_b32 (selector byte = b picks byte b of {hi,lo}); only bits [5:0] of the result are used.
__device__ __forceinline__ uint32_t bin_shift(uint32_t b)
{
    return __builtin_amdgcn_perm(0x3E3E342Au, 0x20140A00u, b);
}

// bin (0..5), 6 = unit holding a state 6, 7 = not a unit, from a category byte; MODE is a template constant
template <int MODE>
__device__ __forceinline__ uint32_t bin_of_byte(uint32_t c)
{
    const uint32_t r = c & 7u;
    if (MODE == XM_MODE_SE) return r;                                  // 0xFF -> 7, state 6 -> 6
    const uint32_t f = (c >> 3) & 7u;
    const uint32_t lo = f < r ? f : r, hi = f < r ? r : f;            // 0xFF -> lo = hi = 7
    uint32_t b = lo;
    if (MODE == XM_MODE_PE_CONSERVATIVE) {
        b = (((f ^ r) & 1u) != 0u || hi == 4u) ? 4u : b;              // :525-529
        b = (hi == 5u) ? 5u : b;                                       // :521
        b = (hi == 7u) ? 7u : b;
    }
    return (hi == 6u) ? 6u : b;
}

// number of 4-bit fields of w equal to 6
__device__ __forceinline__ uint32_t count_nibbles_eq6(uint32_t w)
{
    const uint32_t t = w ^ 0x66666666u;                                // zero nibble <=> field was 6
    const uint32_t z = ~(((t & 0x77777777u) + 0x77777777u) | t) & 0x88888888u;
    return (uint32_t)__builtin_popcount(z);
}

__device__ __forceinline__ uint64_t wave_scan_incl64(uint64_t v)
{
    // fields never carry across bit 32 (each word holds three 10-bit fields + spare bits), so the two
    // halves scan independently
    const uint32_t lo = wave_scan_incl((uint32_t)v), hi = wave_scan_incl((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

template <int MODE, int K, int ABL = 0>      // ABL: ablation switches for tools/tune_kernels.hip only (0 = product)
__global__ void __launch_bounds__(XM_BLOCK)
scatter_kernel(const uint8_t *__restrict__ code, uint64_t n, uint32_t chunk_stride,
               const uint32_t *__restrict__ chunk_off, const unsigned long long *__restrict__ bin_totals,
               unsigned long long *__restrict__ bin_offsets, uint32_t *__restrict__ idx_out)
{
    __shared__ uint4 tile_state[K][XM_BLOCK];            // per lane and tile: {bins lo, bins hi, counters lo, hi}
    __shared__ uint16_t stage[XM_BLOCK / 64][XM_WTILE];   // record offsets inside the wave tile (0..1023)
    __shared__ uint32_t wave_tot[XM_BLOCK / 64][8];

    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint64_t span0 = ((uint64_t)blockIdx.x * (XM_BLOCK / 64) + wave) * (uint64_t)(K * XM_WTILE);

    // lane b < 8: where bin b starts in idx_out (exclusive prefix of the bin totals) plus this chunk's
    // offset inside the bin; fetched first so that the latency hides behind phase 1
    uint32_t bin_start, lane_base;
    {
        const uint32_t tot = (lane < 8u) ? (uint32_t)bin_totals[lane] : 0u;
        const uint32_t off = (lane < 7u) ? chunk_off[(uint64_t)lane * chunk_stride + blockIdx.x] : 0u;
        bin_start = wave_scan_incl(tot) - tot;
        lane_base = bin_start + off;
    }
    if (blockIdx.x == 0 && t < 8) bin_offsets[t] = bin_start;       // lanes 0..7 of wave 0

    // ---- phase 1: bytes -> bins (4 bits each, 7 = not a unit) and per-lane counters, every tile of the span;
    //      parked in LDS so that phase 2 can be a rolled loop
    uint32_t lane_tot[7] = {0, 0, 0, 0, 0, 0, 0};
    uint32_t err_tiles = 0;          // bit k: this lane holds a state-6 unit in tile k
    {
        uint32_t w[4], wn[4] = {0, 0, 0, 0};
        load_codes16(code, span0 + lane * 16u, n, w);
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            if (k + 1 < K) load_codes16(code, span0 + (uint64_t)(k + 1) * XM_WTILE + lane * 16u, n, wn);   // prefetch
            uint64_t cnt = 0;
            uint32_t n0 = 0, n1 = 0;
            // strictly interleaved mates (the usual paired input): units only at odd positions in every lane of
            // the wave -> a branch-free pass over the 8 odd bytes; anything else takes the general pass
            const bool even_free = ((w[0] & w[1] & w[2] & w[3]) & 0x00FF00FFu) == 0x00FF00FFu;
            if (__ballot(!even_free) == 0ull) {
                n0 = n1 = 0x07070707u;
#pragma unroll
                for (int j = 1; j < 16; j += 2) {
                    const uint32_t b = bin_of_byte<MODE>((w[j >> 2] >> (8 * (j & 3))) & 0xFFu);
                    if (j < 8) n0 |= b << (4 * j); else n1 |= b << (4 * (j - 8));
                    cnt += 1ull << bin_shift(b);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t c = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                    if (__ballot(c != XM_NO_UNIT) == 0ull) {          // wave-uniform: nobody has a unit here
                        if (j < 8) n0 |= 7u << (4 * j); else n1 |= 7u << (4 * (j - 8));
                        continue;
                    }
                    const uint32_t b = bin_of_byte<MODE>(c);
                    if (j < 8) n0 |= b << (4 * j); else n1 |= b << (4 * (j - 8));
                    cnt += 1ull << bin_shift(b);
                }
            }
            tile_state[k][t] = make_uint4(n0, n1, (uint32_t)cnt, (uint32_t)(cnt >> 32));
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) {
                lane_tot[bb] += ((uint32_t)cnt >> (10 * bb)) & 0x3FFu;
                lane_tot[3 + bb] += ((uint32_t)(cnt >> 32) >> (10 * bb)) & 0x3FFu;
            }
            const uint32_t c6 = count_nibbles_eq6(n0) + count_nibbles_eq6(n1);
            lane_tot[6] += c6;
            err_tiles |= (c6 != 0u) ? (1u << k) : 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = wn[q];
        }
    }
    // wave totals of the span (two 16-bit fields per word: at most K*1024 per bin)
    {
        const uint32_t p01 = wave_scan_incl(lane_tot[0] | (lane_tot[1] << 16));
        const uint32_t p23 = wave_scan_incl(lane_tot[2] | (lane_tot[3] << 16));
        const uint32_t p45 = wave_scan_incl(lane_tot[4] | (lane_tot[5] << 16));
        const uint32_t p6 = wave_scan_incl(lane_tot[6]);
        if (lane == 63) {
            wave_tot[wave][0] = p01 & 0xFFFFu; wave_tot[wave][1] = p01 >> 16;
            wave_tot[wave][2] = p23 & 0xFFFFu; wave_tot[wave][3] = p23 >> 16;
            wave_tot[wave][4] = p45 & 0xFFFFu; wave_tot[wave][5] = p45 >> 16;
            wave_tot[wave][6] = p6; wave_tot[wave][7] = 0;
        }
    }
    __syncthreads();
    // where this wave's units of each bin start in idx_out (wave-uniform)
    uint32_t gbase[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) {
        uint32_t g = lane_value(lane_base, b);
        for (uint32_t wv = 0; wv < wave; ++wv) g += wave_tot[wv][b];
        gbase[b] = __builtin_amdgcn_readfirstlane(g);
    }

    // ---- phase 2: tile by tile, no workgroup synchronisation
    uint16_t *slab = stage[wave];
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const uint4 ts = tile_state[k][t];
        const uint32_t nib0 = ts.x, nib1 = ts.y;
        const uint64_t cnt = ((uint64_t)ts.w << 32) | ts.z;
        const uint64_t excl = wave_scan_incl64(cnt) - cnt;
        // tile totals per bin = lane 63's exclusive prefix + own count (added per field: a field may reach 1024)
        const uint32_t ex63 = lane_value((uint32_t)excl, 63), ey63 = lane_value((uint32_t)(excl >> 32), 63);
        const uint32_t cx63 = lane_value((uint32_t)cnt, 63), cy63 = lane_value((uint32_t)(cnt >> 32), 63);
        uint32_t tcnt[6], lstart[6];
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const uint32_t e = (b < 3) ? ex63 : ey63, c = (b < 3) ? cx63 : cy63;
            const int f = 10 * (b % 3);
            tcnt[b] = ((e >> f) & 0x3FFu) + ((c >> f) & 0x3FFu);
            lstart[b] = acc;
            acc += tcnt[b];
        }
        // running positions, packed like the counters
        uint64_t pos = excl + (((uint64_t)(lstart[3] | (lstart[4] << 10) | (lstart[5] << 20)) << 32)
                               | (uint64_t)(lstart[0] | (lstart[1] << 10) | (lstart[2] << 20)));
        const uint32_t tile0 = (uint32_t)(span0 + (uint64_t)k * XM_WTILE);
        const uint32_t rec0 = tile0 + lane * 16u;
        const bool even_free2 = ((nib0 & nib1) & 0x0F0F0F0Fu) == 0x07070707u;
        if (__ballot(!even_free2) == 0ull) {                          // interleaved mates: odd positions only
#pragma unroll
            for (int j = 1; j < 16; j += 2) {
                const uint32_t b = ((j < 8 ? nib0 : nib1) >> (4 * (j & 7))) & 7u;
                const uint32_t sh = bin_shift(b);
                const uint32_t p = (uint32_t)(pos >> sh) & 0x3FFu;
                if (ABL < 2) { if (b < 6u) slab[p] = (uint16_t)(lane * 16u + (uint32_t)j); }
                else asm volatile("" :: "v"(p));
                pos += 1ull << sh;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t b = ((j < 8 ? nib0 : nib1) >> (4 * (j & 7))) & 7u;
                const bool valid = b < 6u;
                if (__ballot(valid) == 0ull) continue;                // wave-uniform
                const uint32_t sh = bin_shift(b);
                const uint32_t p = (uint32_t)(pos >> sh) & 0x3FFu;
                if (ABL < 2) { if (valid) slab[p] = (uint16_t)(lane * 16u + (uint32_t)j); }
                else asm volatile("" :: "v"(p));
                pos += 1ull << sh;                                    // bins 6, 7 land in the sink
            }
        }
        // each bin's run, contiguous in LDS and contiguous in idx_out
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            // streamed out once and read by the host only: non-temporal, so the 200 MB of indices do not push
            // the category bytes (re-read by K2, rewritten by the next K1) out of the Infinity Cache
            if (ABL == 0) {
                for (uint32_t e = lane; e < tcnt[b]; e += 64u)
                    __builtin_nontemporal_store(tile0 + (uint32_t)slab[lstart[b] + e], idx_out + gbase[b] + e);
            } else if (ABL == 1) {
                for (uint32_t e = lane; e < tcnt[b]; e += 64u) asm volatile("" :: "v"((uint32_t)slab[lstart[b] + e]));
            } else if (ABL == 3) {
                for (uint32_t e = lane; e < tcnt[b]; e += 64u) idx_out[gbase[b] + e] = tile0 + (uint32_t)slab[lstart[b] + e];
            } else if (ABL == 5) {
                // whole 128-byte lines of the run stream out non-temporally; the ragged head and tail (which a
                // neighbouring run completes later) go through the cache so that the halves can merge there
                const uint32_t g0 = gbase[b], g1 = g0 + tcnt[b];
                const uint32_t a0 = (g0 + 31u) & ~31u, a1 = g1 & ~31u;
                for (uint32_t e = lane; e < tcnt[b]; e += 64u) {
                    const uint32_t g = g0 + e, v = tile0 + (uint32_t)slab[lstart[b] + e];
                    if (g >= a0 && g < a1) __builtin_nontemporal_store(v, idx_out + g);
                    else idx_out[g] = v;
                }
            } else if (ABL == 4) {
                for (uint32_t e = lane; e < tcnt[b]; e += 64u)
                    __hip_atomic_store(idx_out + gbase[b] + e, tile0 + (uint32_t)slab[lstart[b] + e], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            gbase[b] += tcnt[b];
        }
        // state-6 units (rare)
        if (__ballot((err_tiles >> k) & 1u) != 0ull) {
            const uint32_t c6 = count_nibbles_eq6(nib0) + count_nibbles_eq6(nib1);
            const uint32_t i6 = wave_scan_incl(c6);
            uint32_t run6 = gbase[6] + i6 - c6;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if ((((j < 8 ? nib0 : nib1) >> (4 * (j & 7))) & 7u) == 6u) idx_out[run6++] = rec0 + (uint32_t)j;
            gbase[6] += lane_value(i6, 63);
        }
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
// K1c: classify with AS synthesised from CIGAR + NM on the fly (the --cigar_scores path, ref :684-685):
// K3 fused into K1, so the AS columns are never written to or read from memory.  Per species a lane
// loads NM x4, XS x4, five CSR offsets, and the first XM_CIG_SPEC packed ops of each of its 4 records
// speculatively (all loads issued together; adjacent lanes read adjacent ops); records with more ops
// finish in a short loop.
// ---------------------------------------------------------------------------------------------
#define XM_CIG_SPEC 3

__device__ __forceinline__ long long cigar_term(uint32_t v)
{
    const uint32_t op = v & 15u;
    const long long len = (long long)(v >> 4);
    long long t = (op == 1u || op == 2u) ? (5ll + 3ll * len) : 0ll;       // I, D: open + extend (:252-255)
    t += (op == 4u) ? 2ll * len : 0ll;                                     // S
    return t;
}

// the same in 32 bits: a packed op carries len < 2^28, so one term is < 2^30 and three of them fit a uint32
__device__ __forceinline__ uint32_t cigar_term32(uint32_t v)
{
    const uint32_t op = v & 15u, len = v >> 4;
    uint32_t t = (op == 1u || op == 2u) ? (5u + 3u * len) : 0u;
    t += (op == 4u) ? 2u * len : 0u;
    return t;
}

__device__ __forceinline__ int32_t cigar_clamp(long long s, uint32_t *range_flag)
{
    if (s <= (long long)INT32_MIN || s > (long long)INT32_MAX) {
        if (range_flag) atomicOr(range_flag, 1u);
        s = s < 0 ? (long long)INT32_MIN + 1 : (long long)INT32_MAX;
    }
    return (int32_t)s;
}

__device__ __forceinline__ int32_t cigar_score_one(const int32_t *__restrict__ nm, const uint32_t *__restrict__ off,
                                                   const uint32_t *__restrict__ ops, uint64_t i, uint32_t *range_flag)
{
    const int32_t nmv = nm[i];
    if (nmv == INT32_MIN) return INT32_MIN;
    long long s = -6ll * (long long)nmv;
    const uint32_t k1 = off[i + 1];
    for (uint32_t k = off[i]; k < k1; ++k) s -= cigar_term(ops[k]);
    return cigar_clamp(s, range_flag);
}

struct __attribute__((packed, aligned(4))) Ops3 {
    uint32_t v[XM_CIG_SPEC];
};

// AS of the lane's 4 records of one species.  Hot path without divergent branches: the first XM_CIG_SPEC ops of
// every record come with ONE 12-byte load (reading into the next record's ops is harmless, the surplus is masked)
// whenever the whole wave stays XM_CIG_SPEC entries clear of the end of the op array (n_ops = cig_off[n]); longer
// CIGARs and out-of-range scores are handled after wave-uniform tests.  Returns true when a score left int32.
// Phase A of a species: NM x4 and the five CSR offsets of the lane's 4 records (independent loads).
template <bool FULL>
__device__ __forceinline__ void cigar_fetch_offsets(const int32_t *__restrict__ nm, const uint32_t *__restrict__ off,
                                                    uint64_t r0, uint64_t n, int32_t nmv[4], uint32_t o[5])
{
    load4<int32_t, true, FULL>(nm, r0, n, nmv);
    if (FULL || r0 + 4 <= n) {
        const v4i32 q = __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(off + r0));
        o[0] = (uint32_t)q.x; o[1] = (uint32_t)q.y; o[2] = (uint32_t)q.z; o[3] = (uint32_t)q.w;
        o[4] = off[r0 + 4];
    } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) o[j] = (r0 + j <= n) ? off[r0 + j] : 0u;
#pragma unroll
        for (int j = 1; j < 5; ++j) o[j] = (r0 + j <= n) ? o[j] : o[j - 1];
    }
}

// Phase B, part 1: the first XM_CIG_SPEC ops of every record with ONE 12-byte load (reading into the next record's
// ops is harmless, the surplus is masked later) whenever the whole wave stays XM_CIG_SPEC entries clear of the end of
// the op array (n_ops = cig_off[n]); a wave-uniform test, so the hot path has no divergent branch.
__device__ __forceinline__ void cigar_fetch_ops(const uint32_t *__restrict__ ops, uint32_t n_ops, const uint32_t o[5],
                                                uint32_t v[4][XM_CIG_SPEC])
{
    if (__ballot(o[3] + XM_CIG_SPEC > n_ops) == 0ull) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Ops3 t = *reinterpret_cast<const Ops3 *>(ops + o[j]);
#pragma unroll
            for (int q = 0; q < XM_CIG_SPEC; ++q) v[j][q] = t.v[q];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < XM_CIG_SPEC; ++q) v[j][q] = (o[j] + (uint32_t)q < o[j + 1]) ? ops[o[j] + q] : 0u;
    }
}

// Phase B, part 2: scores.  Longer CIGARs and out-of-range scores are handled after wave-uniform tests.
// Returns true when a score left int32.
__device__ __forceinline__ bool cigar_finish_scores(const uint32_t *__restrict__ ops, const int32_t nmv[4], const uint32_t o[5],
                                                    const uint32_t v[4][XM_CIG_SPEC], int32_t as_out[4])
{
    long long s[4];
    bool longer = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t k = o[j + 1] - o[j];
        uint32_t acc = 0;
#pragma unroll
        for (int q = 0; q < XM_CIG_SPEC; ++q) acc += ((uint32_t)q < k) ? cigar_term32(v[j][q]) : 0u;
        s[j] = -6ll * (long long)nmv[j] - (long long)acc;
        longer |= k > (uint32_t)XM_CIG_SPEC;
    }
    if (__ballot(longer) != 0ull) {                                    // some record of the wave has more ops
#pragma unroll
        for (int j = 0; j < 4; ++j)
            for (uint32_t k = o[j] + XM_CIG_SPEC; k < o[j + 1]; ++k) s[j] -= cigar_term(ops[k]);
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool present = nmv[j] != INT32_MIN;
        const bool out = present && (s[j] <= (long long)INT32_MIN || s[j] > (long long)INT32_MAX);
        bad |= out;
        const long long c = out ? (s[j] < 0 ? (long long)INT32_MIN + 1 : (long long)INT32_MAX) : s[j];
        as_out[j] = present ? (int32_t)c : INT32_MIN;
    }
    return bad;
}

// The usual case, op-parallel: the wave's records own one contiguous stretch of the op array, [o[0] of lane 0,
// o[4] of lane 63).  Every lane loads four consecutive ops of it with one 16-byte load (coalesced, instead of one
// 12-byte gather per record), turns them into penalty terms, and a wave prefix sum of the terms goes to LDS as table
// T; a record's penalty is then T[end] - T[begin].  Exact while a record's terms sum below 2^32: guaranteed by the
// wave-uniform guards (no op longer than 2^20, no record with more than 512 ops); a wave that trips one of them, or
// whose stretch has XM_CIG_WAVE_OPS ops or more, returns false and takes the per-record path above.
#define XM_CIG_WAVE_OPS 1024
__device__ __forceinline__ bool cigar_scores_by_prefix(const uint32_t *__restrict__ ops, uint32_t n_ops, const int32_t nmv[4],
                                                       const uint32_t o[5], uint32_t *T, int32_t as_out[4], bool &bad)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t base = __builtin_amdgcn_readfirstlane(o[0]);
    const uint32_t W = lane_value(o[4], 63) - base;
    bool odd = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) odd |= (o[j + 1] - o[j]) > 512u;
    if (W >= (uint32_t)XM_CIG_WAVE_OPS || __ballot(odd) != 0ull) return false;
    uint32_t carry = 0;
    const uint32_t chunks = W / 256u + 1u;                              // slot W (the grand total) is written too
    for (uint32_t c = 0; c < chunks; ++c) {
        const uint32_t s0 = c * 256u + 4u * lane;
        uint32_t v[4];
        if (base + (c + 1u) * 256u <= n_ops) {                          // wave-uniform: the chunk lies inside the array
            v[0] = v[1] = v[2] = v[3] = 0u;
            if (s0 < W) {                                               // lanes past the stretch fetch nothing
                const v4i32 q = *reinterpret_cast<const v4i32_a4 *>(ops + base + s0);
                v[0] = (uint32_t)q.x; v[1] = (uint32_t)q.y; v[2] = (uint32_t)q.z; v[3] = (uint32_t)q.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (base + s0 + (uint32_t)q < n_ops) ? ops[base + s0 + (uint32_t)q] : 0u;
        }
        uint32_t t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = s0 + (uint32_t)q < W;
            odd |= in && (v[q] >> 4) >= (1u << 20);
            t[q] = in ? cigar_term32(v[q]) : 0u;
        }
        const uint32_t p1 = t[0], p2 = p1 + t[1], p3 = p2 + t[2], tot = p3 + t[3];
        const uint32_t incl = wave_scan_incl(tot);
        const uint32_t ex = incl - tot + carry;
        *reinterpret_cast<uint4 *>(T + s0) = make_uint4(ex, ex + p1, ex + p2, ex + p3);
        carry += lane_value(incl, 63);
    }
    if (__ballot(odd) != 0ull) return false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t d = T[o[j + 1] - base] - T[o[j] - base];
        const long long sc = -6ll * (long long)nmv[j] - (long long)d;
        const bool present = nmv[j] != INT32_MIN;
        const bool out = present && (sc <= (long long)INT32_MIN || sc > (long long)INT32_MAX);
        bad |= out;
        const long long cl = out ? (sc < 0 ? (long long)INT32_MIN + 1 : (long long)INT32_MAX) : sc;
        as_out[j] = present ? (int32_t)cl : INT32_MIN;
    }
    return true;
}

template <bool PAIRED, int BLOCK, bool FULL>
__device__ __forceinline__ void classify_cigar_body(
    const int32_t *__restrict__ nm1, const uint32_t *__restrict__ off1, const uint32_t *__restrict__ ops1,
    const int32_t *__restrict__ xs1,
    const int32_t *__restrict__ nm2, const uint32_t *__restrict__ off2, const uint32_t *__restrict__ ops2,
    const int32_t *__restrict__ xs2,
    const uint8_t *__restrict__ unit_bits8, int32_t m, uint8_t *__restrict__ code, uint64_t n,
    uint32_t *__restrict__ range_flag, uint32_t *last_state, uint32_t *cig_T)
{
    const uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t r0 = g * 4;

    int32_t a1[4], x1[4], a2[4], x2[4];
    load4<int32_t, true, FULL>(xs1, r0, n, x1);
    load4<int32_t, true, FULL>(xs2, r0, n, x2);
    uint32_t mb = 0;
    if (FULL || r0 < n) mb = (uint32_t)(unit_bits8[g >> 1] >> ((g & 1u) * 4u)) & 0xFu;
    if (!FULL && r0 + 4 > n) mb &= (r0 < n) ? ((1u << (uint32_t)(n - r0)) - 1u) : 0u;
    bool bad = false;
    if (FULL || r0 < n) {
        // two dependent memory latencies in total: offsets of both species first, then the ops of both
        int32_t nmv1[4], nmv2[4];
        uint32_t o1[5], o2[5], v1[4][XM_CIG_SPEC], v2[4][XM_CIG_SPEC];
        const uint32_t n_ops1 = off1[n], n_ops2 = off2[n];
        cigar_fetch_offsets<FULL>(nm1, off1, r0, n, nmv1, o1);
        cigar_fetch_offsets<FULL>(nm2, off2, r0, n, nmv2, o2);
        // FULL: every lane of the wave is here, so the wave can work on its op stretch together
        const bool done1 = FULL && cigar_scores_by_prefix(ops1, n_ops1, nmv1, o1, cig_T, a1, bad);
        const bool done2 = FULL && cigar_scores_by_prefix(ops2, n_ops2, nmv2, o2, cig_T, a2, bad);
        if (!done1) {                                                   // wave-uniform when FULL
            cigar_fetch_ops(ops1, n_ops1, o1, v1);
            bad |= cigar_finish_scores(ops1, nmv1, o1, v1, a1);
        }
        if (!done2) {
            cigar_fetch_ops(ops2, n_ops2, o2, v2);
            bad |= cigar_finish_scores(ops2, nmv2, o2, v2, a2);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) a1[j] = a2[j] = INT32_MIN;
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63u) == 0u && range_flag) atomicOr(range_flag, 1u);

    uint32_t halo = 0;
    if (PAIRED && threadIdx.x == 0) {
        if (r0 > 0) {
            const uint64_t h = r0 - 1;
            halo = mapping_state<int32_t>(cigar_score_one(nm1, off1, ops1, h, range_flag), xs1[h],
                                          cigar_score_one(nm2, off2, ops2, h, range_flag), xs2[h], m);
        } else {
            mb &= ~1u;
        }
    }
    classify_finish<int32_t, PAIRED, BLOCK, FULL>(a1, x1, a2, x2, m, mb, halo, last_state, code, r0, n);
}

template <bool PAIRED, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
classify_cigar_kernel(const int32_t *__restrict__ nm1, const uint32_t *__restrict__ off1, const uint32_t *__restrict__ ops1,
                      const int32_t *__restrict__ xs1,
                      const int32_t *__restrict__ nm2, const uint32_t *__restrict__ off2, const uint32_t *__restrict__ ops2,
                      const int32_t *__restrict__ xs2,
                      const uint8_t *__restrict__ unit_bits8, int32_t m, uint8_t *__restrict__ code, uint64_t n,
                      uint32_t *__restrict__ range_flag)
{
    __shared__ uint32_t last_state[BLOCK / 64];
    __shared__ __attribute__((aligned(16))) uint32_t cig_table[BLOCK / 64][XM_CIG_WAVE_OPS + 4];
    uint32_t *cig_T = cig_table[threadIdx.x >> 6];
    if (((uint64_t)blockIdx.x + 1) * (BLOCK * 4) <= n)
        classify_cigar_body<PAIRED, BLOCK, true>(nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits8, m, code, n,
                                                 range_flag, last_state, cig_T);
    else
        classify_cigar_body<PAIRED, BLOCK, false>(nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits8, m, code, n,
                                                  range_flag, last_state, cig_T);
}

// ---------------------------------------------------------------------------------------------
// K4: mate-density correlation of a single-end mappability track (xenomappability, SURVEY 8f-4;
// Mappability.single_end_to_paired, /root/reference/xenomapper/mappability.py:94-124).
// out[i] = 1.0 where track[i] == 1, else sum_j track[i+j] * density[j] for j < min(m, n - i), accumulated left to
// right in binary64 with a separately rounded multiply and add (fp contraction switched off: never an FMA),
// which is what Python's `result += a * b` does -- bit-exact, not just close.  f64-VALU bound (2 flops per tap,
// m taps per output); the track window of a workgroup is staged in LDS (conflict-free ds_read_b64, one tap per
// step), the taps are wave-uniform loads.
// ---------------------------------------------------------------------------------------------
#define XM_CORR_T 256       // outputs per workgroup
#define XM_CORR_M 2048      // taps per pass (LDS: (256 + 2048) * 8 bytes)
__global__ void __launch_bounds__(XM_CORR_T)
mate_correlate_kernel(const double *__restrict__ track, uint64_t n, const double *__restrict__ density, uint32_t m,
                      double *__restrict__ out)
{
#pragma clang fp contract(off)          // hipcc contracts a * b + c into v_fma_f64 by default: one rounding instead of two
    __shared__ double tile[XM_CORR_T + XM_CORR_M];
    const uint32_t t = threadIdx.x;
    const uint64_t i0 = (uint64_t)blockIdx.x * XM_CORR_T, i = i0 + t;
    const uint64_t left = (i < n) ? n - i : 0;                         // taps that still have a track position
    double acc = 0.0;
    for (uint32_t j0 = 0; j0 < m; j0 += XM_CORR_M) {
        const uint32_t mm = (m - j0 < XM_CORR_M) ? m - j0 : XM_CORR_M;
        __syncthreads();
        for (uint32_t k = t; k < XM_CORR_T + mm; k += XM_CORR_T) {
            const uint64_t p = i0 + j0 + k;
            tile[k] = (p < n) ? track[p] : 0.0;
        }
        __syncthreads();
        const uint32_t lim = (left > j0) ? (uint32_t)((left - j0 < mm) ? left - j0 : mm) : 0u;
        if (__ballot(lim != mm) == 0ull) {                             // wave-uniform: nobody is near the track's end
            for (uint32_t j = 0; j < mm; ++j) {
                const double prod = tile[t + j] * density[j0 + j];      // rounded
                acc = acc + prod;                                       // rounded again (contraction is off above)
            }
        } else {
            for (uint32_t j = 0; j < mm; ++j) {
                const double prod = tile[t + j] * density[j0 + j];
                const double s = acc + prod;
                acc = (j < lim) ? s : acc;
            }
        }
    }
    if (i < n) out[i] = (track[i] == 1.0) ? 1.0 : acc;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <typename T>
static void launch_classify_t(hipStream_t st, int mode, uint64_t n,
                              const T *as1, const T *xs1, const T *as2, const T *xs2,
                              const uint64_t *unit_bits, T m, uint8_t *code)
{
    const uint64_t per_block = (uint64_t)XM_CLASSIFY_BLOCK * 4;
    const uint32_t grid = (uint32_t)((n + per_block - 1) / per_block);
    const uint8_t *bits8 = reinterpret_cast<const uint8_t *>(unit_bits);
    if (mode == XM_MODE_SE)
        classify_kernel<T, false, XM_CLASSIFY_NT, XM_CLASSIFY_BLOCK><<<grid, XM_CLASSIFY_BLOCK, 0, st>>>(as1, xs1, as2, xs2, bits8, m, code, n);
    else
        classify_kernel<T, true, XM_CLASSIFY_NT, XM_CLASSIFY_BLOCK><<<grid, XM_CLASSIFY_BLOCK, 0, st>>>(as1, xs1, as2, xs2, bits8, m, code, n);
}

void launch_classify_i32(hipStream_t st, int mode, uint64_t n,
                         const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                         const uint64_t *unit_bits, int32_t m, uint8_t *code)
{
    launch_classify_t<int32_t>(st, mode, n, as1, xs1, as2, xs2, unit_bits, m, code);
}

void launch_classify_f64(hipStream_t st, int mode, uint64_t n,
                         const double *as1, const double *xs1, const double *as2, const double *xs2,
                         const uint64_t *unit_bits, double m, uint8_t *code)
{
    launch_classify_t<double>(st, mode, n, as1, xs1, as2, xs2, unit_bits, m, code);
}

void launch_classify_cigar(hipStream_t st, int mode, uint64_t n,
                           const int32_t *nm1, const uint32_t *off1, const uint32_t *ops1, const int32_t *xs1,
                           const int32_t *nm2, const uint32_t *off2, const uint32_t *ops2, const int32_t *xs2,
                           const uint64_t *unit_bits, int32_t m, uint8_t *code, uint32_t *range_flag)
{
    const uint64_t per_block = (uint64_t)XM_CIGAR_BLOCK * 4;
    const uint32_t grid = (uint32_t)((n + per_block - 1) / per_block);
    const uint8_t *bits8 = reinterpret_cast<const uint8_t *>(unit_bits);
    if (mode == XM_MODE_SE)
        classify_cigar_kernel<false, XM_CIGAR_BLOCK><<<grid, XM_CIGAR_BLOCK, 0, st>>>(
            nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, bits8, m, code, n, range_flag);
    else
        classify_cigar_kernel<true, XM_CIGAR_BLOCK><<<grid, XM_CIGAR_BLOCK, 0, st>>>(
            nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, bits8, m, code, n, range_flag);
}

ChunkPlan plan_chunks(uint64_t n)
{
    ChunkPlan p;
    uint64_t chunks = (n + XM_CHUNK - 1) / XM_CHUNK;
    if (chunks == 0) chunks = 1;
    p.n_chunks = (uint32_t)chunks;
    p.chunk_stride = (p.n_chunks + 63u) & ~63u;
    return p;
}

void launch_hist(hipStream_t st, const ChunkPlan &p, int mode, uint64_t n, const uint8_t *code,
                 uint32_t *chunk_counts, uint64_t *counts_rep)
{
    hist_kernel<XM_K><<<p.n_chunks, XM_BLOCK, 0, st>>>(code, n, mode, p.chunk_stride, chunk_counts,
                                                        reinterpret_cast<unsigned long long *>(counts_rep));
}

void launch_scan(hipStream_t st, const ChunkPlan &p, const uint32_t *chunk_counts, uint32_t *chunk_off,
                 uint64_t *bin_totals, uint64_t *counts_rep, uint64_t *counts)
{
    scan_kernel<<<8, XM_SCAN_THREADS, 0, st>>>(chunk_counts, p.n_chunks, p.chunk_stride, chunk_off,
                                   reinterpret_cast<unsigned long long *>(bin_totals),
                                   reinterpret_cast<unsigned long long *>(counts_rep),
                                   reinterpret_cast<unsigned long long *>(counts));
}

void launch_scatter(hipStream_t st, const ChunkPlan &p, int mode, uint64_t n, const uint8_t *code,
                    const uint32_t *chunk_off, const uint64_t *bin_totals, uint64_t *bin_offsets, uint32_t *idx_out)
{
    const unsigned long long *bt = reinterpret_cast<const unsigned long long *>(bin_totals);
    unsigned long long *bo = reinterpret_cast<unsigned long long *>(bin_offsets);
    if (mode == XM_MODE_SE)
        scatter_kernel<XM_MODE_SE, XM_K><<<p.n_chunks, XM_BLOCK, 0, st>>>(code, n, p.chunk_stride, chunk_off, bt, bo, idx_out);
    else if (mode == XM_MODE_PE_LIBERAL)
        scatter_kernel<XM_MODE_PE_LIBERAL, XM_K><<<p.n_chunks, XM_BLOCK, 0, st>>>(code, n, p.chunk_stride, chunk_off, bt, bo, idx_out);
    else
        scatter_kernel<XM_MODE_PE_CONSERVATIVE, XM_K><<<p.n_chunks, XM_BLOCK, 0, st>>>(code, n, p.chunk_stride, chunk_off, bt, bo, idx_out);
}

void launch_mate_correlate(hipStream_t st, uint64_t n, const double *track, uint32_t m, const double *density, double *out)
{
    const uint32_t grid = (uint32_t)((n + XM_CORR_T - 1) / XM_CORR_T);
    mate_correlate_kernel<<<grid, XM_CORR_T, 0, st>>>(track, n, density, m, out);
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
