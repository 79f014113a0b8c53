// xm_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the xenograft read classifier.
//
// Three stages, all HBM-bound integer/byte work (no MFMA):
//   K1  classify  score columns -> one category per record          (16 B in, 1 B or 1/2 B out / record)
//                 fused form: K1 also counts its units per category, per bin and granule, per bin and part (LDS
//                 histogram, last wave flushes) and can emit the compact category stream (the output bin as a nibble)
//   K1p           the same from packed CIGAR columns: AS synthesised from NM + CIGAR ops in the kernel (--cigar_scores)
//   K1c           round 1's form of that on CSR columns (kept for CSR columns that are already on the device)
//   K2  compact   categories -> category_counts + a stable split of unit indices by bin
//                 K2a histogram (only when category bytes come from memory: wave-private LDS) -> per-granule bin counts
//                 K2b scan of the per-granule counts: one launch, carries from the part totals the counting side added up
//                 K2c scatter: ballot + mbcnt ranks, scalar per-bin bases -> dense index runs; straight to idx_out, or
//                     (single-end input) through a wave-private LDS slab and 16-byte stores
//   K3  cigar     NM + packed CIGAR ops (CSR) -> synthesised AS column (stand-alone xm_cigar_scores)
//
// Reference semantics restated (file:line into /root/reference/xenomapper/xenomapper.py):
//   get_mapping_state :258-289, pair rules :423-448 / :521-550, unit rule :402-405,
//   get_cigarbased_AS_tag :228-256.  The oracle (oracle/) is the checker; nothing here calls it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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

// The same for integer columns, where `a > b` is exactly `!(a <= b)` and one of <, ==, > always holds: 8 compares
// instead of 12 (the generic form keeps every comparison the reference makes, so that a NaN falls through as it does
// there).  State 6 cannot come out of this one.
template <>
__device__ __forceinline__ uint32_t mapping_state<int32_t>(int32_t a1, int32_t x1, int32_t a2, int32_t x2, int32_t m)
{
    const bool low1 = a1 <= m, low2 = a2 <= m;        // :275
    const bool gt = a1 > a2, lt = a2 > a1;
    const bool spec1 = (x1 == 0) || (a1 > x1);        // :278  `not XS1 or AS1 > XS1`
    const bool spec2 = (x2 == 0) || (a2 > x2);        // :285
    uint32_t s = spec2 ? 1u : 3u;                     // :284-288: what is left when the tests below all fail
    s = (gt || lt) ? s : 4u;                          // :282-283
    s = (!low1 && (low2 || gt)) ? (spec1 ? 0u : 2u) : s;   // :277-281
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

// The same for a workgroup whose records all exist: `wg` = the column at the workgroup's first record (uniform, lives in
// scalar registers), `byte_off` = this lane's offset from there -- one 32-bit VGPR shared by every column of the same
// element size instead of a 64-bit address per load (global_load ... v_off, s[base:base+1]).
template <typename T, bool NT>
__device__ __forceinline__ void load4_wg(const T *__restrict__ wg, uint32_t byte_off, T out[4])
{
    typedef typename Vec4<T>::type V;
    const V *p = reinterpret_cast<const V *>(reinterpret_cast<const char *>(wg) + byte_off);
    const V v = NT ? __builtin_nontemporal_load(p) : *p;
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}

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

// This wave's LDS operations so far have been performed (LDS executes a wave's operations in issue order), and the
// compiler moves no memory access across this point.
__device__ __forceinline__ void lds_settle()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// Counting (category_counts + the per-granule bin counts the stable split needs), shared by K1 (fused: the
// category bytes are still in registers) and K2a (stand-alone compaction of category bytes from memory).
//
// A *granule* is the stretch of XM_GRAN = 2048 records one counting group owns: a counting K1 workgroup or a K2a
// wave.  Per granule: bin counts -> gran_counts[bin][granule]; the 64 category slots ->
// one of XM_COUNT_REPLICAS global copies of counts[64] (K2b adds the copies up).
//
// LDS layout of one counting group: hist[64 slots][XM_HREP replicas] (replica = lane & 7: a wave instruction's
// atomics on one slot spread over 8 banks), then 16 words: [0] arrival counter, [8..15] bin counts.
// ---------------------------------------------------------------------------------------------
#define XM_HREP 8
#define XM_COUNT_LDS_WORDS (64 * XM_HREP + 16)

struct CountSink {
    uint32_t *gran_counts;               // [8][gran_stride]
    uint32_t *part_tot;                  // [XM_PART_REPLICAS][8][XM_PART_STRIDE]: units per bin and part, in replicas (all zero between calls)
    unsigned long long *counts_rep;      // [XM_COUNT_REPLICAS][64]
    uint8_t *bins4;                      // compact category stream (XM_GRAN / 2 bytes per granule) or null
    uint32_t gran_stride;
    int mode;                            // bin rule of the flush
    uint32_t blk0;                       // a launch over part of the input (chunked xm_classify_place*): its first workgroup's number
};

// one wave adds up the group's histogram (lane = category slot) and publishes it
__device__ __forceinline__ void count_flush(uint32_t *lds, uint32_t granule, const CountSink &sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *misc = lds + 64 * XM_HREP;
    const uint4 h0 = *reinterpret_cast<const uint4 *>(lds + lane * XM_HREP);
    const uint4 h1 = *reinterpret_cast<const uint4 *>(lds + lane * XM_HREP + 4);
    const uint32_t s = h0.x + h0.y + h0.z + h0.w + h1.x + h1.y + h1.z + h1.w;
    if (s != 0u) {
        atomicAdd(&sink.counts_rep[(granule % XM_COUNT_REPLICAS) * 64u + lane], (unsigned long long)s);
        atomicAdd(&misc[8u + bin_of_code(sink.mode, lane)], s);          // a counted slot is never 0xFF: bin <= 6
    }
    lds_settle();
    if (lane < 8u) {
        const uint32_t v = misc[8u + lane];
        sink.gran_counts[(uint64_t)lane * sink.gran_stride + granule] = v;
        // the first level of K2b's scan on the fly: units per bin and part (= XM_PART_GRAN granules).  One of
        // XM_PART_REPLICAS copies, 67 KB apart: the few (part, bin) cells that are hot at any moment would otherwise be
        // ~14 addresses taking 150 k atomics (that form cost K1 220 us; spread like this it costs nothing measurable)
        if (v != 0u) atomicAdd(&sink.part_tot[((granule & (XM_PART_REPLICAS - 1u)) * 8u + lane) * XM_PART_STRIDE + granule / XM_PART_GRAN], v);
    }
}

// K1 side: every wave adds its (up to 4 per lane) category bytes; the last wave to arrive flushes.  `lds` must have
// been zeroed by the workgroup before a barrier that precedes this call.  No barrier of its own.
template <int BLOCK>
__device__ __forceinline__ void count_units(const uint32_t c[4], uint32_t *lds, const CountSink &sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rep = lane & (XM_HREP - 1u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool unit = c[j] != XM_NO_UNIT;
        if (__ballot(unit) == 0ull) continue;                             // wave-uniform (interleaved mates: j = 0, 2)
        if (unit) atomicAdd(&lds[(c[j] & 63u) * XM_HREP + rep], 1u);
    }
    lds_settle();                                                         // this wave's counts are in
    uint32_t arrived = 0;
    if (lane == 0u) arrived = atomicAdd(&lds[64 * XM_HREP], 1u);
    arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
    if (arrived != (uint32_t)(BLOCK / 64 - 1)) return;
    lds_settle();
    count_flush(lds, blockIdx.x + sink.blk0, sink);
}

// Shared K1 epilogue: 4 states per lane -> forward mate's state (lane-1 / previous wave via LDS / halo) ->
// 4 category bytes -> one 4-byte store (-> counts, when fused).
// output bin of a category byte, rule fixed at compile time.  HAS6: state 6 can occur (binary64 input only).
// 0..5 bins, 6 = a unit holding a state 6, 7 = not a unit.
template <int MODE, bool HAS6>
__device__ __forceinline__ uint32_t unit_bin(uint32_t c)
{
    if (MODE == XM_MODE_SE) return c == XM_NO_UNIT ? 7u : (c & 7u);                 // the state itself (6 stays 6)
    const uint32_t f = (c >> 3) & 7u, r = c & 7u;
    const uint32_t lo = f < r ? f : r, hi = f < r ? r : f;
    uint32_t b = lo;
    if (MODE == XM_MODE_PE_CONSERVATIVE) {
        b = (((f ^ r) & 1u) != 0u || hi == 4u) ? 4u : b;                           // :525-529
        b = (hi == 5u) ? 5u : b;                                                    // :521
    }
    if (HAS6) b = (hi > 5u) ? 6u : b;
    return c == XM_NO_UNIT ? 7u : b;
}

// the same from the two states themselves (the classify kernels have them in registers: no unpacking of the code)
template <int MODE, bool HAS6>
__device__ __forceinline__ uint32_t unit_bin_of(uint32_t f, uint32_t r, bool unit)
{
    uint32_t b;
    if (MODE == XM_MODE_SE) {
        b = r;                                                                      // the state itself (6 stays 6)
    } else {
        const uint32_t lo = f < r ? f : r, hi = f < r ? r : f;
        b = lo;                                                                     // :423-448 == min()
        if (MODE == XM_MODE_PE_CONSERVATIVE) {
            b = (((f ^ r) & 1u) != 0u || hi == 4u) ? 4u : b;                       // :525-529
            b = (hi == 5u) ? 5u : b;                                                // :521
        }
        if (HAS6) b = (hi > 5u) ? 6u : b;
    }
    return unit ? b : 7u;
}

template <typename T, bool PAIRED, int BLOCK, bool FULL, bool COUNTS, int BINMODE>
__device__ __forceinline__ void classify_finish(const T a1[4], const T x1[4], const T a2[4], const T x2[4], T m,
                                                uint32_t mb, uint32_t halo, uint32_t *last_state,
                                                uint8_t *__restrict__ code, uint64_t r0, uint64_t n,
                                                uint32_t *count_lds, const CountSink &sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = mapping_state<T>(a1[j], x1[j], a2[j], x2[j], m);

    uint32_t c[4], fwd[4] = {0, 0, 0, 0};
    if (PAIRED) {
        // lane - 1's last state: a DPP wave shift (wave_shr:1), not a ds_bpermute
        uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s[3], 0x138, 0xf, 0xf, false);
        if (lane == 63) last_state[wave] = s[3];
        __syncthreads();
        if (lane == 0) prev = (wave == 0) ? halo : last_state[wave - 1];
        c[0] = (mb & 1u) ? ((prev << 3) | s[0]) : XM_NO_UNIT;
        c[1] = (mb & 2u) ? ((s[0] << 3) | s[1]) : XM_NO_UNIT;
        c[2] = (mb & 4u) ? ((s[1] << 3) | s[2]) : XM_NO_UNIT;
        c[3] = (mb & 8u) ? ((s[2] << 3) | s[3]) : XM_NO_UNIT;
        fwd[0] = prev; fwd[1] = s[0]; fwd[2] = s[1]; fwd[3] = s[2];
    } else {
        if (COUNTS) __syncthreads();                  // the zeroed count_lds is visible (paired: the barrier above)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = ((mb >> j) & 1u) ? s[j] : XM_NO_UNIT;
    }

    if (FULL || r0 + 4 <= n) {
        if (code != nullptr) *reinterpret_cast<uint32_t *>(code + r0) = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (code != nullptr && r0 + j < n) code[r0 + j] = (uint8_t)c[j];
    }
    if (BINMODE >= 0) {
        // the compact category stream (xm_classify_compact*_dev with bins4): the output bin of each of the lane's 4 records
        // as a nibble, record r in nibble r & 1 of byte r >> 1: one full 128-byte line per wave instruction.  (A
        // [lane][wave] transposition inside the granule, which would let K2c read 16 bytes per lane at once, made this
        // kernel 20 us slower: eight waves each writing a sliver of every line.)  The last workgroup writes its whole
        // block, so the buffer is XM_BINS4_BYTES(n) long.
        uint32_t nib = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) nib |= unit_bin_of<BINMODE, sizeof(T) == 8>(fwd[j], s[j], ((mb >> j) & 1u) != 0u) << (4 * j);
        reinterpret_cast<uint16_t *>(sink.bins4)[r0 >> 2] = (uint16_t)nib;
    }
    if (COUNTS) count_units<BLOCK>(c, count_lds, sink);
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
template <typename T, bool PAIRED, bool NT, int BLOCK, bool FULL, bool COUNTS, int BINMODE>
__device__ __forceinline__ void classify_body(const T *__restrict__ as1, const T *__restrict__ xs1,
                                              const T *__restrict__ as2, const T *__restrict__ xs2,
                                              const uint8_t *__restrict__ unit_bits8, T m,
                                              uint8_t *__restrict__ code, uint64_t n, uint32_t *last_state,
                                              uint32_t *count_lds, const CountSink &sink)
{
    const uint64_t g = (uint64_t)(blockIdx.x + sink.blk0) * BLOCK + threadIdx.x;       // group of 4 records
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
    classify_finish<T, PAIRED, BLOCK, FULL, COUNTS, BINMODE>(a1, x1, a2, x2, m, mb, halo, last_state, code, r0, n, count_lds, sink);
}

template <int BLOCK>
__device__ __forceinline__ void count_lds_clear(uint32_t *count_lds)
{
    for (uint32_t k = threadIdx.x; k < (uint32_t)XM_COUNT_LDS_WORDS; k += BLOCK) count_lds[k] = 0;
}

// COUNTS: the fused form (xm_classify_compact*): the workgroup also counts its units per category and per bin, so
// the category bytes are not read again for a histogram; a granule = this workgroup's BLOCK*4 records.
// BINMODE >= 0 (= the loop's mode): the kernel also writes the compact category stream bins4 and `code` may be null.
template <typename T, bool PAIRED, bool NT, int BLOCK, bool COUNTS, int BINMODE>
__global__ void __launch_bounds__(BLOCK)
classify_kernel(const T *__restrict__ as1, const T *__restrict__ xs1,
                const T *__restrict__ as2, const T *__restrict__ xs2,
                const uint8_t *__restrict__ unit_bits8, T m,
                uint8_t *__restrict__ code, uint64_t n, CountSink sink)
{
    __shared__ uint32_t last_state[BLOCK / 64];
    __shared__ __attribute__((aligned(16))) uint32_t count_lds[COUNTS ? XM_COUNT_LDS_WORDS : 4];
    if (COUNTS) count_lds_clear<BLOCK>(count_lds);
    // every workgroup but possibly the last covers BLOCK*4 existing records: no bounds tests on that path
    if (((uint64_t)blockIdx.x + sink.blk0 + 1) * (BLOCK * 4) <= n)
        classify_body<T, PAIRED, NT, BLOCK, true, COUNTS, BINMODE>(as1, xs1, as2, xs2, unit_bits8, m, code, n, last_state, count_lds, sink);
    else
        classify_body<T, PAIRED, NT, BLOCK, false, COUNTS, BINMODE>(as1, xs1, as2, xs2, unit_bits8, m, code, n, last_state, count_lds, sink);
}

// ---------------------------------------------------------------------------------------------
// K2 geometry.  Counting and splitting work granule by granule; a wave owns one granule and nothing is shared between
// waves, so K2a and K2c have no workgroup barrier.  Lane order == record order everywhere, which is what makes the
// split stable.  A granule is XM_GRAN = 2048 records: a wave of K2a / K2c, a workgroup of the counting K1.
// ---------------------------------------------------------------------------------------------

// 16 category bytes of one lane; past the end of the input: XM_NO_UNIT
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
    else v = load_codes16_tail(code, base, n);          // last, partial granule only
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}

__device__ __attribute__((noinline)) uint32_t load_codes4_tail(const uint8_t *__restrict__ code, uint64_t base, uint64_t n)
{
    uint32_t acc = 0;
    for (int j = 0; j < 4; ++j) {
        const uint64_t i = base + j;
        const uint32_t c = (i < n) ? (uint32_t)code[i] : XM_NO_UNIT;
        acc |= c << (8 * j);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// K2a: histogram of category bytes in memory (stand-alone xm_compact*; the fused path counts in K1).  One wave =
// one granule of XM_GRAN records (32 bytes per lane), wave-private LDS histogram.  A byte position at which no
// lane of the wave holds a unit (every other position of interleaved paired input) is skipped wave-uniformly; units
// of the wave's presumably dominant category (the first unit of its first lane) are counted in a register and reach
// LDS with one atomic per lane.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(XM_BLOCK)
hist_kernel(const uint8_t *__restrict__ code, uint64_t n, uint32_t n_gran, CountSink sink)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[XM_BLOCK / 64][XM_COUNT_LDS_WORDS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t *lds = lds_all[wave];
    const uint32_t g = blockIdx.x * (XM_BLOCK / 64) + wave;
    if (g >= n_gran) return;                                              // wave-uniform; no barrier in this kernel
    for (uint32_t k = lane; k < (uint32_t)XM_COUNT_LDS_WORDS; k += 64u) lds[k] = 0;

    constexpr int TILES = XM_GRAN / 1024;
    const uint64_t rec0 = (uint64_t)g * XM_GRAN;
    uint32_t w[TILES][4];
#pragma unroll
    for (int k = 0; k < TILES; ++k) load_codes16(code, rec0 + (uint64_t)k * 1024u + lane * 16u, n, w[k]);
    lds_settle();

    const uint32_t rep = lane & (XM_HREP - 1u);
    uint32_t common = w[0][0] & 0xFFu;
    if (common == XM_NO_UNIT) common = (w[0][0] >> 8) & 0xFFu;
    common = (uint32_t)__builtin_amdgcn_readfirstlane((int)common);
    uint32_t n_common = 0;
#pragma unroll
    for (int k = 0; k < TILES; ++k) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t c = (w[k][j >> 2] >> (8 * (j & 3))) & 0xFFu;
            const bool unit = c != XM_NO_UNIT;
            if (__ballot(unit) == 0ull) continue;                         // wave-uniform
            const bool same = unit && c == common;
            n_common += same ? 1u : 0u;
            if (unit && !same) atomicAdd(&lds[(c & 63u) * XM_HREP + rep], 1u);
        }
    }
    if (n_common) atomicAdd(&lds[(common & 63u) * XM_HREP + rep], n_common);
    lds_settle();
    count_flush(lds, g, sink);
}

// ---------------------------------------------------------------------------------------------
// K2b: exclusive scan of the per-granule bin counts, two levels, ONE launch.  A part = XM_PART_GRAN = 1024 consecutive
// granules.  The first level -- units per bin and part -- is added up by the counting side itself while it flushes a
// granule (count_flush: replicated atomics); workgroup (part p, bin b) takes its carry from the part totals in front of p
// and scans its own 1024 counts once -- nobody passes over all counts, nothing depends on another workgroup of the
// launch.  gran_off[b][g] = units of bin b in granules < g; bin_totals[b] = units of bin b.  Workgroups of part 0 also
// add up the replicas of category_counts (8 slots per bin row) and zero them again; the part totals are zeroed by K2c
// (every workgroup of this launch may still read them).
// (Round 2 had a second kernel for the part sums: two launches + the gap between them, 10-12 us per 100 M records.)
// ---------------------------------------------------------------------------------------------
#define XM_SCAN_THREADS 1024
__device__ __forceinline__ unsigned long long block_sum_1024(unsigned long long v, unsigned long long *wsum)
{
    v += __shfl_xor(v, 1, 64);  v += __shfl_xor(v, 2, 64);  v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);  v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if ((threadIdx.x & 63u) == 0u) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long total = 0;
#pragma unroll
    for (int w = 0; w < XM_SCAN_THREADS / 64; ++w) total += wsum[w];
    return total;
}

__global__ void __launch_bounds__(XM_SCAN_THREADS)
scan_kernel(const uint32_t *__restrict__ gran_counts, uint32_t n_gran, uint32_t gran_stride,
            const uint32_t *__restrict__ part_tot, uint32_t *__restrict__ gran_off,
            unsigned long long *__restrict__ bin_totals,
            unsigned long long *__restrict__ counts_rep, unsigned long long *__restrict__ counts, uint32_t part0, uint32_t n_parts)
{
    // part0 / n_parts: a launch over the parts [part0, part0 + gridDim.x) of n_parts (chunked xm_classify_place*: the part
    // totals of the chunks in front are still in place, so the carry below is the global one); the workgroups of the LAST
    // part close the call: bin totals, and the category_counts replicas every counting launch of the call has added to
    __shared__ unsigned long long wsum[XM_SCAN_THREADS / 64];
    __shared__ uint32_t wtot[XM_SCAN_THREADS / 64];
    const uint32_t p = blockIdx.x + part0, b = blockIdx.y;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (p + 1u == n_parts && t < 64u) {   // category_counts slots 8b..8b+7: 8 lanes per slot, 8 replicas each
        const uint32_t slot = b * 8u + (t >> 3), part = t & 7u;
        unsigned long long acc = 0;
        for (uint32_t r = part; r < XM_COUNT_REPLICAS; r += 8u) {
            acc += counts_rep[r * 64u + slot];
            counts_rep[r * 64u + slot] = 0;               // consumed: leave the replicas zeroed for the next count
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (part == 0) counts[slot] = acc;
    }
    // this part's granules: one per thread (loaded first: the carry loop below hides the latency)
    const uint32_t g = p * XM_PART_GRAN + t;
    const uint32_t x = (g < n_gran) ? gran_counts[(uint64_t)b * gran_stride + g] : 0u;
    // carry: units of bin b in front of this part = the p part totals before it, each in XM_PART_REPLICAS pieces;
    // consecutive threads walk consecutive parts of one replica row
    unsigned long long part_sum = 0;
    for (uint32_t e = t; e < p * XM_PART_REPLICAS; e += XM_SCAN_THREADS) {
        const uint32_t r = e / p, q = e - r * p;
        part_sum += part_tot[(r * 8u + b) * XM_PART_STRIDE + q];
    }
    const unsigned long long carry = block_sum_1024(part_sum, wsum);

    const uint32_t incl = wave_scan_incl(x);
    if (lane == 63u) wtot[wave] = incl;
    __syncthreads();
    uint32_t before_wave = 0, part_total = 0;
    for (uint32_t w = 0; w < XM_SCAN_THREADS / 64; ++w) {
        const uint32_t v = wtot[w];
        before_wave += (w < wave) ? v : 0u;
        part_total += v;
    }
    if (g < n_gran) gran_off[(uint64_t)b * gran_stride + g] = (uint32_t)carry + before_wave + incl - x;
    if (p + 1u == n_parts && t == 0u) bin_totals[b] = carry + part_total;
}

// ---------------------------------------------------------------------------------------------
// K2c: scatter.  One wave = one granule, taken 256 records (one dword of category bytes per lane) at a time.
// A unit's place in its bin is   (bin start) + (bin's units in earlier granules: K2b) + (rank inside the granule),
// and the rank comes from ballots: for every bin that occurs in the 256 records, the lanes holding it form a 64-bit
// mask per byte position, v_mbcnt counts the mask bits below the lane, and the running per-bin base lives in scalar
// registers.  Lanes of one bin therefore write one dense run of idx_out per store instruction -- no barrier, and in
// the direct form no LDS staging.  The byte -> bin rule (mode dependent) is a 64-entry wave-private LDS table, read
// conflict-free.
// SLOTS: which of a lane's 4 positions can hold units -- 0b1010 for strictly interleaved mates (a wave-uniform test per
// 256 records); 0b0011 = the lane's first and second unit, wherever they sit (paired input with unpaired reads: two
// units per lane at most); else 0b1111.  WIDE: byte offsets into idx_out need more than 32 bits.
// ---------------------------------------------------------------------------------------------
// XM_SCATTER_GUARD: no index is stored past the number of units (what K2b reported), whatever the category bytes hold
#ifndef XM_SCATTER_GUARD
#define XM_SCATTER_GUARD 1
#endif
__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask, uint32_t add)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, add));
}

template <bool WIDE>
__device__ __forceinline__ void store_index(uint32_t *__restrict__ idx_out, uint32_t pos, uint32_t rec)
{
    if (WIDE) idx_out[pos] = rec;
    else *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(idx_out) + (pos << 2)) = rec;   // SGPR base + 32-bit offset
}

// STAGED: `base` are places in the wave's LDS slab (one region per bin), `rec0` the lane's first record counted from the
// granule's first, `n_units` the slab's capacity: the units are only sorted locally here; the wave copies the regions out
// afterwards (scatter_copy_out).
// rec[j]: the record the lane's position j stands for (rec0 + j when the positions are the lane's four records; the lane's
// first / second unit in the two-units-per-lane form, SLOTS 0x3)
// LISTS (the six-list output contract, xm_classify_place*: SURVEY 8b (4), the reference's six sinks :332-350, :423-448,
// :521-550): bin b's units go to their own caller-allocated list; base[b] then counts from the start of list b and a lane
// takes its list's address from `lptr`, a wave-private LDS table of the seven list pointers (null: not listed).
template <int SLOTS, bool WIDE, bool STAGED, bool LISTS = false>
__device__ __forceinline__ void scatter_256(const uint32_t bin[4], const uint32_t rec[4], uint32_t base[7],
                                            uint32_t *__restrict__ idx_out, uint32_t n_units, uint16_t *slab,
                                            const uint64_t *lptr = nullptr)
{
    uint32_t pos[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 7; ++b) {
        uint64_t m[4], any = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[j] = ((SLOTS >> j) & 1) ? __ballot(bin[j] == (uint32_t)b) : 0ull;
            any |= m[j];
        }
        if (any == 0ull) continue;              // wave-uniform: bin b does not occur here (inside the pipeline 1-2 % faster)
        uint32_t t = base[b];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((SLOTS >> j) & 1) t = mbcnt64(m[j], t);                    // + units of bin b in lower lanes
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((SLOTS >> j) & 1) pos[j] = (bin[j] == (uint32_t)b) ? t : pos[j];
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((SLOTS >> j) & 1) total += (uint32_t)__builtin_popcountll(m[j]);
        base[b] += total;
    }
    // units of the same bin earlier in this lane
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < j; ++i)
            if (((SLOTS >> j) & 1) && ((SLOTS >> i) & 1)) pos[j] += (bin[i] == bin[j]) ? 1u : 0u;
    // (One 8- or 16-byte store per lane whose units share a bin -- they go to consecutive places -- measured slower than
    // these dword stores in every workload, 0.0587 against 0.0547 ms at 50 M interleaved pairs: profiles/r03_ab_scatter.txt.)
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (((SLOTS >> j) & 1) && bin[j] < 7u && (!XM_SCATTER_GUARD || pos[j] < n_units)) {
            if (STAGED) slab[pos[j]] = (uint16_t)rec[j];
            else if (LISTS) {
                uint32_t *list = reinterpret_cast<uint32_t *>(lptr[bin[j]]);
                if (list != nullptr) list[pos[j]] = rec[j];
            }
            else store_index<WIDE>(idx_out, pos[j], rec[j]);
        }
}

// XM_SCATTER_STAGED (0 in a tuning build: every unit's index goes to idx_out with its own dword store): where a granule
// holds XM_STAGE_MIN_UNITS units or more, they are first sorted by bin inside a wave-private LDS slab (16-bit record numbers), then every bin's run is copied
// to its place in idx_out with 16-byte stores from 16-byte-aligned places.  Why: dword stores top out at 4.2 TB/s on
// this chip however dense they are (a 200 MB fill takes 49.8 us; K2c with one dword store per unit took 53.8 us for
// its 200 MB), 16-byte stores reach 6.4 TB/s (tools/probe_streams.hip, profiles/r03_ab_scatter.txt).
#ifndef XM_SCATTER_STAGED
#define XM_SCATTER_STAGED 1
#endif
#define XM_SLAB_U16 (XM_GRAN + 64)     // region of bin b: room for its units rounded up to 4, + 4 for the alignment shift
#ifndef XM_STAGE_MIN_UNITS
#define XM_STAGE_MIN_UNITS 1536        // units in a granule from which its wave stages them
#endif

template <bool WIDE>
__device__ __forceinline__ void store_index4(uint32_t *__restrict__ idx_out, uint32_t pos, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3)
{
    typedef uint32_t v4u32_a4 __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte aligned when idx_out is; correct anyway
    v4u32_a4 v; v.x = r0; v.y = r1; v.z = r2; v.w = r3;
    if (WIDE) *reinterpret_cast<v4u32_a4 *>(idx_out + pos) = v;                           // pos is a multiple of 4
    else *reinterpret_cast<v4u32_a4 *>(reinterpret_cast<char *>(idx_out) + (pos << 2)) = v;
}

// One bin's run of the granule: N record numbers at slab[L ..], to idx_out[G ..].  (L + the head length) is a multiple
// of 4 by construction, so the body moves aligned 8-byte LDS reads into aligned 16-byte stores; the up to 3 + 3 places
// around it and runs shorter than a wave take dword stores.
template <bool WIDE>
__device__ __forceinline__ void scatter_copy_out(const uint16_t *slab, uint32_t L, uint32_t G, uint32_t N, uint32_t rec_g,
                                                 uint32_t *__restrict__ idx_out)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (N < 64u) {
        if (lane < N) store_index<WIDE>(idx_out, G + lane, rec_g + slab[L + lane]);
        return;
    }
    const uint32_t h = (0u - G) & 3u;                  // places up to the first 16-byte-aligned one
    const uint32_t body = (N - h) >> 2;                // whole groups of four
    const uint32_t t = (N - h) & 3u;                   // places after the last whole group
    {
        const bool head = lane < h, tail = lane >= 4u && lane - 4u < t;
        const uint32_t e = head ? lane : h + 4u * body + (lane - 4u);
        if (head || tail) store_index<WIDE>(idx_out, G + e, rec_g + slab[L + e]);
    }
    for (uint32_t c = lane; c < body; c += 64u) {
        const uint32_t e = h + 4u * c;
        const uint2 w = *reinterpret_cast<const uint2 *>(slab + L + e);
        store_index4<WIDE>(idx_out, G + e, rec_g + (w.x & 0xFFFFu), rec_g + (w.x >> 16), rec_g + (w.y & 0xFFFFu), rec_g + (w.y >> 16));
    }
}

// NIB: the categories come as the compact stream a counting K1 wrote (bins4: the bin itself, a nibble per record, 16 bits
// per lane and 256 records, no table); otherwise as category bytes (one dword per lane and 256 records, the
// byte -> bin rule of the mode in a 64-entry wave-private LDS table).
// the wave's granule, 256 records at a time: ranks by ballots, indices straight to idx_out or (STAGED) into the slab
template <int NSUB, bool WIDE, bool NIB, bool STAGED, bool LISTS>
__device__ __forceinline__ void scatter_granule(const uint32_t w[NSUB], const uint8_t *lut, uint32_t rec_g, uint32_t base[7],
                                                uint32_t *__restrict__ idx_out, uint32_t n_units, uint16_t *slab, const uint64_t *lptr)
{
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const uint32_t rec0 = (STAGED ? 0u : rec_g) + (uint32_t)s * 256u + lane * 4u;
        const uint32_t limit = STAGED ? (uint32_t)XM_SLAB_U16 : n_units;
        uint32_t bin[4];
        bool even_free;
        if (NIB) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bin[j] = (w[s] >> (4 * j)) & 7u;
            even_free = (w[s] & 0x0707u) == 0x0707u;
        } else {
            even_free = (w[s] & 0x00FF00FFu) == 0x00FF00FFu;
        }
        const uint32_t rec[4] = {rec0, rec0 + 1u, rec0 + 2u, rec0 + 3u};
        if (__ballot(!even_free) == 0ull) {                               // strictly interleaved mates: positions 1 and 3 only
            if (!NIB) { bin[0] = bin[2] = 7u; bin[1] = lut[(w[s] >> 8) & 63u]; bin[3] = lut[(w[s] >> 24) & 63u]; }
            scatter_256<0xA, WIDE, STAGED, LISTS && !STAGED>(bin, rec, base, idx_out, limit, slab, lptr);
            continue;
        }
        if (!NIB) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bin[j] = lut[(w[s] >> (8 * j)) & 63u];
        }
        // Mates that are not strictly interleaved (unpaired reads in between flip the parity): a lane still holds two
        // units at most unless three records in a row share a name.  Then the lane's first and second unit take the
        // place of its four records -- half the ballots and rank chains of the general form below.
        // y: bit 4 j set where position j holds a unit (from the nibbles themselves, or from the four bins)
        uint32_t y;
        if (NIB) {
            const uint32_t x = w[s] ^ 0x7777u;                             // nibble != 0  <=>  bin != 7
            y = (x | (x >> 1) | (x >> 2)) & 0x1111u;
        } else {
            y = (bin[0] < 7u ? 1u : 0u) | (bin[1] < 7u ? 0x10u : 0u) | (bin[2] < 7u ? 0x100u : 0u) | (bin[3] < 7u ? 0x1000u : 0u);
        }
        if (!STAGED && __ballot(__builtin_popcount(y) > 2) == 0ull) {
            const uint32_t rest = y & (y - 1u);                            // without the lane's first unit
            const uint32_t c0 = y ? (uint32_t)__builtin_ctz(y) : 0u, c1 = rest ? (uint32_t)__builtin_ctz(rest) : 0u;   // 4 j0, 4 j1
            const uint32_t j0 = c0 >> 2, j1 = c1 >> 2;
            uint32_t vb[4] = {7u, 7u, 7u, 7u};
            if (NIB) {
                vb[0] = y ? (w[s] >> c0) & 7u : 7u;
                vb[1] = rest ? (w[s] >> c1) & 7u : 7u;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {                              // bin[j0], bin[j1] without indexing registers dynamically
                    vb[0] = (y && j0 == (uint32_t)j) ? bin[j] : vb[0];
                    vb[1] = (rest && j1 == (uint32_t)j) ? bin[j] : vb[1];
                }
            }
            const uint32_t vrec[4] = {rec0 + j0, rec0 + j1, 0u, 0u};
            scatter_256<0x3, WIDE, STAGED, LISTS>(vb, vrec, base, idx_out, limit, slab, lptr);
        } else {
            scatter_256<0xF, WIDE, STAGED, LISTS && !STAGED>(bin, rec, base, idx_out, limit, slab, lptr);
        }
    }
}

// STAGE: this launch may stage (the single-end loop, where every record can be a unit; the paired loops, whose granules
// hold half as many units at most in ordinary input, use the instantiation without the slab and its bookkeeping, which
// costs them 2 us per 50 M pairs)
// LISTS: the six-list output contract -- bin b's units go to lo.p[b] (positions count from the start of that list, so the
// totals of the bins in front are not needed), bin_offsets receives the list lengths ([7] = all units).
// Launch shape: one granule per wave (MULTI = false, what every input up to 2^20 granules runs: the body below is the
// round-4 kernel, whose loads of the granule's counts and of its categories are in flight together and which has no loop
// state to carry).  MULTI: a wave owns a RUN of `gran_per_wave` consecutive granules (the per-bin places carry over from
// granule to granule in scalar registers: the scatter leaves base[b] advanced by the granule's units of bin b, so the scan's
// offsets are read for the wave's first granule only).  Round 5 tried long-lived waves for every input (a dense dword fill
// gains 4.1 -> 6.0 TB/s from exactly that, tools/probe_streams.hip w) and measured the opposite for this kernel --
// XM_SCATTER_WAVES in xm_kernels.h has the numbers -- and compiling the one-granule case through the loop cost the staged
// single-end form 13 % (133.7 against 118.6 us per 100 M reads, profiles/r05_se_kernel_stats.csv): hence the template flag.
template <int NSUB, bool WIDE, bool NIB, bool STAGE, bool LISTS, bool MULTI>
__global__ void __launch_bounds__(XM_BLOCK)
scatter_kernel(const uint8_t *__restrict__ code, uint64_t n, int mode, uint32_t n_gran, uint32_t gran_stride,
               const uint32_t *__restrict__ gran_counts, const uint32_t *__restrict__ gran_off,
               const unsigned long long *__restrict__ bin_totals,
               unsigned long long *__restrict__ bin_offsets, uint32_t *__restrict__ idx_out, uint32_t *__restrict__ part_tot,
               const ListOut lo, uint32_t gran_per_wave, uint32_t g_first, uint32_t last)
{
    // g_first / last: a launch over the granules [g_first, n_gran) of a chunked call (lists only: their places need the
    // units in front, not the bin totals); the LAST launch of a call publishes the totals and zeroes the part totals
    constexpr bool CAN_STAGE = STAGE && XM_SCATTER_STAGED != 0 && NSUB * 256 == XM_GRAN;
    __shared__ uint8_t lut_all[NIB ? 1 : XM_BLOCK / 64][64];
    __shared__ uint64_t lptr_all[LISTS ? XM_BLOCK / 64 : 1][8];
    __shared__ __attribute__((aligned(16))) uint16_t slab_all[CAN_STAGE ? XM_BLOCK / 64 : 1][CAN_STAGE ? XM_SLAB_U16 : 4];
    const uint32_t lane = threadIdx.x & 63u;
    if (last) {   // K2b has consumed the part totals: leave them zeroed for the next count (n_parts cells in each of the 8 x replicas rows)
        const uint32_t n_parts = (n_gran + XM_PART_GRAN - 1u) / XM_PART_GRAN, cells = 8u * XM_PART_REPLICAS * n_parts;
        for (uint32_t i = blockIdx.x * XM_BLOCK + threadIdx.x; i < cells; i += gridDim.x * XM_BLOCK)
            part_tot[(i / n_parts) * XM_PART_STRIDE + i % n_parts] = 0u;
    }
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g0 = g_first + (blockIdx.x * (XM_BLOCK / 64) + wave) * (MULTI ? gran_per_wave : 1u);
    if (g0 >= n_gran) return;                                             // wave-uniform; no barrier in this kernel
    uint8_t *lut = lut_all[NIB ? 0 : wave];
    if (!NIB) lut[lane] = (uint8_t)bin_of_code(mode, lane == 63u ? XM_NO_UNIT : lane);
    uint64_t *lptr = lptr_all[LISTS ? wave : 0];
    if (LISTS && lane < 8u) lptr[lane] = lane < 7u ? (uint64_t)(uintptr_t)lo.p[lane] : 0ull;      // read back by this wave only

    // lane b < 8: where bin b starts in idx_out (exclusive prefix of the bin totals) plus what the granules before
    // the wave's first one hold of it
    uint32_t lane_base, n_units;
    {
        const uint32_t tot = (lane < 8u) ? (uint32_t)bin_totals[lane] : 0u;
        const uint32_t off = (lane < 7u) ? gran_off[(uint64_t)lane * gran_stride + g0] : 0u;
        const uint32_t bin_start = wave_scan_incl(tot) - tot;
        lane_base = LISTS ? off : bin_start + off;
        n_units = LISTS ? lo.cap : lane_value(bin_start, 7);              // slot 7 counts nothing: the total
        if (g0 == g_first && last && lane < 8u) bin_offsets[lane] = (LISTS && lane < 7u) ? tot : bin_start;
    }
    uint16_t *slab = slab_all[CAN_STAGE ? wave : 0];

    auto load_granule = [&](uint32_t g, uint32_t w[NSUB]) {
        const uint64_t rec_g = (uint64_t)g * (NSUB * 256u);
        if (NIB) {
            // 16 bits per lane and 256 records; whole granule blocks exist (records past the end read as 7)
            const uint16_t *nib = reinterpret_cast<const uint16_t *>(code) + (rec_g >> 2) + lane;
#pragma unroll
            for (int s = 0; s < NSUB; ++s) w[s] = nib[s * 64];
        } else if (rec_g + NSUB * 256u <= n) {
#pragma unroll
            for (int s = 0; s < NSUB; ++s) w[s] = *reinterpret_cast<const uint32_t *>(code + rec_g + s * 256u + lane * 4u);
        } else {
#pragma unroll
            for (int s = 0; s < NSUB; ++s) w[s] = load_codes4_tail(code, rec_g + s * 256u + lane * 4u, n);
        }
    };
    // Staging pays where a granule holds many units (single-end input: 2048 of them; 117 against 166 us per 100 M reads)
    // and costs where it holds few (strictly interleaved mates, 1024: 62 against 55 us): decided per launch (STAGE) and
    // then per granule from its unit count (what the counting side reported).
    // lane b < 7 lays out bin b's region of the slab: room for its units rounded up to 4, + 4, regions back to back;
    // the run starts (its place in idx_out) mod 4 words into its region, so that 16-byte-aligned places of idx_out
    // are 8-byte-aligned places of the slab; never past the number of units, never past the slab, whatever the counts hold
    auto slab_layout = [&](uint32_t cnt_g, uint32_t &run_start, uint32_t &run_len) -> bool {
        const uint32_t room = (lane < 7u) ? ((cnt_g + 3u) & ~3u) + 4u : 0u;
        const uint32_t incl_room = wave_scan_incl(room), incl_cnt = wave_scan_incl(cnt_g);
        run_start = incl_room - room + (lane_base & 3u);
        const uint32_t cnt = lane_base < n_units ? (cnt_g < n_units - lane_base ? cnt_g : n_units - lane_base) : 0u;
        run_len = run_start < (uint32_t)XM_SLAB_U16 ? (cnt < XM_SLAB_U16 - run_start ? cnt : XM_SLAB_U16 - run_start) : 0u;
        return lane_value(incl_cnt, 6) >= (uint32_t)XM_STAGE_MIN_UNITS;
    };
    auto copy_runs_out = [&](uint32_t run_start, uint32_t run_len, uint32_t rec_g) {
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            const uint32_t N = lane_value(run_len, b);
            uint32_t *__restrict__ dst = LISTS ? lo.p[b] : idx_out;
            if (N == 0u || dst == nullptr) continue;                      // wave-uniform
            scatter_copy_out<WIDE>(slab, lane_value(run_start, b), lane_value(lane_base, b), N, rec_g, dst);
        }
    };

    if (!MULTI) {
        const uint32_t g = g0;
        uint32_t run_start = 0, run_len = 0;
        bool staged = false;
        if (CAN_STAGE) {
            const uint32_t cnt_g = (lane < 7u) ? gran_counts[(uint64_t)lane * gran_stride + g] : 0u;
            staged = slab_layout(cnt_g, run_start, run_len);
        }
        uint32_t base[7];
#pragma unroll
        for (int b = 0; b < 7; ++b) base[b] = lane_value(staged ? run_start : lane_base, b);
        const uint32_t rec_g = (uint32_t)((uint64_t)g * (NSUB * 256u));
        uint32_t w[NSUB];
        load_granule(g, w);
        if (!NIB || LISTS) lds_settle();
        if (CAN_STAGE && staged) {
            scatter_granule<NSUB, WIDE, NIB, true, LISTS>(w, lut, rec_g, base, idx_out, n_units, slab, lptr);
            lds_settle();                                                 // the wave's slab is complete
            copy_runs_out(run_start, run_len, rec_g);
        } else {
            scatter_granule<NSUB, WIDE, NIB, false, LISTS>(w, lut, rec_g, base, idx_out, n_units, slab, lptr);
        }
        return;
    }

    const uint32_t g1 = g0 + gran_per_wave < n_gran ? g0 + gran_per_wave : n_gran;
    uint32_t base[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) base[b] = lane_value(lane_base, b);
    uint32_t w[NSUB], wn[NSUB];
    load_granule(g0, w);
    if (!NIB || LISTS) lds_settle();
    for (uint32_t g = g0; g < g1; ++g) {
        if (g + 1u < g1) load_granule(g + 1u, wn);                        // in flight while this granule is placed
        const uint32_t rec_g = (uint32_t)((uint64_t)g * (NSUB * 256u));
        bool staged = false;
        uint32_t cnt_g = 0;
        if (CAN_STAGE) {
            cnt_g = (lane < 7u) ? gran_counts[(uint64_t)lane * gran_stride + g] : 0u;
            uint32_t run_start, run_len;
            staged = slab_layout(cnt_g, run_start, run_len);
            if (staged) {
                uint32_t sbase[7];
#pragma unroll
                for (int b = 0; b < 7; ++b) sbase[b] = lane_value(run_start, b);
                scatter_granule<NSUB, WIDE, NIB, true, LISTS>(w, lut, rec_g, sbase, idx_out, n_units, slab, lptr);
                lds_settle();                                             // the wave's slab is complete
                copy_runs_out(run_start, run_len, rec_g);
                lds_settle();                                             // the copy-out has read the slab: the next granule may fill it
            }
        }
        if (!staged) {
            if (CAN_STAGE) {
#pragma unroll
                for (int b = 0; b < 7; ++b) base[b] = lane_value(lane_base, b);
            }
            scatter_granule<NSUB, WIDE, NIB, false, LISTS>(w, lut, rec_g, base, idx_out, n_units, slab, lptr);
        }
        if (CAN_STAGE) lane_base += cnt_g;                                // where the next granule's runs begin
#pragma unroll
        for (int s = 0; s < NSUB; ++s) w[s] = wn[s];
    }
}

// ---------------------------------------------------------------------------------------------
// K1r: classify + SEGMENTED bin lists in one launch (xm_classify_runs*).  The six flat lists of K2 need a prefix over all
// granules in front (K2b; a look-back inside one launch lost by 3-7x in round 4); segmented by
// granule they need none: the workgroup = granule sorts its units by output bin inside an LDS slab (16-bit record numbers
// counted from the granule's first record: ranks by ballot + mbcnt per wave, per-wave bin counts through LDS, one
// barrier) and copies the slab out as it stands with 16-byte stores, next to the granule's eight bin counts.  List b in
// input order = for every granule in order, its run of bin b.  2 bytes per unit written instead of 1/2 + 4, nothing read
// twice, one launch (+ a 1-workgroup launch that adds up the category_counts replicas).
// The reference code this replaces: the bodies of the three main loops, xenomapper.py:321-350, :398-452, :498-554.
// ---------------------------------------------------------------------------------------------
// LDS words: histogram [64][XM_HREP] | per-wave bin counts [8 waves][8] | slab XM_GRAN u16
#define XM_RUNS_WCNT_AT (64 * XM_HREP)
#define XM_RUNS_SLAB_AT (XM_RUNS_WCNT_AT + 64)
#define XM_RUNS_LDS_WORDS (XM_RUNS_SLAB_AT + XM_GRAN / 2)

struct RunsSink {
    uint16_t *runs16;                    // [n_gran][XM_GRAN]
    uint16_t *gran_counts16;             // [n_gran][8]
    unsigned long long *counts_rep;      // [XM_COUNT_REPLICAS][64] (the context's; all zero between calls)
};

// ranks of the lane's units inside the wave, per bin: pos[j] = units of bin[j] in lower lanes and lower positions of this
// lane; returns (lane b) the wave's units of bin b
template <int SLOTS, int NB>
__device__ __forceinline__ uint32_t runs_ranks(const uint32_t bin[4], uint32_t pos[4])
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t wc = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        uint64_t m[4], any = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[j] = ((SLOTS >> j) & 1) ? __ballot(bin[j] == (uint32_t)b) : 0ull;
            any |= m[j];
        }
        if (any == 0ull) continue;                                        // wave-uniform
        uint32_t t = 0, total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((SLOTS >> j) & 1) { t = mbcnt64(m[j], t); total += (uint32_t)__builtin_popcountll(m[j]); }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((SLOTS >> j) & 1) pos[j] = (bin[j] == (uint32_t)b) ? t : pos[j];
        wc = (lane == (uint32_t)b) ? total : wc;
    }
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < j; ++i)
            if (((SLOTS >> j) & 1) && ((SLOTS >> i) & 1)) pos[j] += (bin[i] == bin[j]) ? 1u : 0u;
    return wc;
}

template <typename T, bool PAIRED, bool NT, int BLOCK, bool FULL, int BINMODE>
__device__ __forceinline__ void classify_runs_body(const T *__restrict__ as1, const T *__restrict__ xs1,
                                                   const T *__restrict__ as2, const T *__restrict__ xs2,
                                                   const uint8_t *__restrict__ unit_bits8, T m, uint64_t n,
                                                   uint32_t *last_state, uint32_t *lds, const RunsSink &rs)
{
    static_assert(BLOCK == 512 && BLOCK * 4 == XM_GRAN, "eight waves, one granule");
    constexpr bool HAS6 = sizeof(T) == 8;
    constexpr int NB = HAS6 ? 7 : 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g = blockIdx.x;
    const uint64_t g4 = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t r0 = g4 * 4;
    T a1[4], x1[4], a2[4], x2[4];
    load4<T, NT, FULL>(as1, r0, n, a1);
    load4<T, NT, FULL>(xs1, r0, n, x1);
    load4<T, NT, FULL>(as2, r0, n, a2);
    load4<T, NT, FULL>(xs2, r0, n, x2);
    uint32_t mb = 0;
    if (FULL || r0 < n) mb = (uint32_t)(unit_bits8[g4 >> 1] >> ((g4 & 1u) * 4u)) & 0xFu;
    if (!FULL && r0 + 4 > n) mb &= (r0 < n) ? ((1u << (uint32_t)(n - r0)) - 1u) : 0u;
    uint32_t halo = 0;
    if (PAIRED && threadIdx.x == 0) {
        if (r0 > 0) {
            const uint64_t h = r0 - 1;
            halo = mapping_state<T>(as1[h], xs1[h], as2[h], xs2[h], m);
        } else {
            mb &= ~1u;                                                    // record 0 has no predecessor (:402)
        }
    }
    uint32_t s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = mapping_state<T>(a1[j], x1[j], a2[j], x2[j], m);
    uint32_t c[4], fwd[4] = {0, 0, 0, 0};
    if (PAIRED) {
        uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s[3], 0x138, 0xf, 0xf, false);
        if (lane == 63) last_state[wave] = s[3];
        __syncthreads();                                                  // also: the zeroed histogram is visible
        if (lane == 0) prev = (wave == 0) ? halo : last_state[wave - 1];
        c[0] = (mb & 1u) ? ((prev << 3) | s[0]) : XM_NO_UNIT;
        c[1] = (mb & 2u) ? ((s[0] << 3) | s[1]) : XM_NO_UNIT;
        c[2] = (mb & 4u) ? ((s[1] << 3) | s[2]) : XM_NO_UNIT;
        c[3] = (mb & 8u) ? ((s[2] << 3) | s[3]) : XM_NO_UNIT;
        fwd[0] = prev; fwd[1] = s[0]; fwd[2] = s[1]; fwd[3] = s[2];
    } else {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = ((mb >> j) & 1u) ? s[j] : XM_NO_UNIT;
    }
    uint32_t bin[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bin[j] = unit_bin_of<BINMODE, HAS6>(fwd[j], s[j], ((mb >> j) & 1u) != 0u);
    {   // category_counts: the workgroup's histogram (as count_units)
        const uint32_t rep = lane & (XM_HREP - 1u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool unit = c[j] != XM_NO_UNIT;
            if (__ballot(unit) == 0ull) continue;
            if (unit) atomicAdd(&lds[(c[j] & 63u) * XM_HREP + rep], 1u);
        }
    }
    uint32_t pos[4] = {0, 0, 0, 0};
    const bool inter = PAIRED && __ballot((mb & 5u) != 0u) == 0ull;     // strictly interleaved mates: positions 1 and 3 only
    const uint32_t wc = inter ? runs_ranks<0xA, NB>(bin, pos) : runs_ranks<0xF, NB>(bin, pos);
    uint32_t *wave_cnt = lds + XM_RUNS_WCNT_AT;
    uint16_t *slab = reinterpret_cast<uint16_t *>(lds + XM_RUNS_SLAB_AT);
    if (lane < 8u) wave_cnt[wave * 8u + lane] = wc;
    __syncthreads();
    uint32_t off, tot;                                                   // lane: bin lane & 7 -- units in the waves before / in all waves
    {
        const uint32_t v = wave_cnt[lane];                               // lane = 8 w + b
        off = (lane >> 3) < wave ? v : 0u;
        tot = v;
        off += (uint32_t)__shfl_xor((int)off, 8, 64);  tot += (uint32_t)__shfl_xor((int)tot, 8, 64);
        off += (uint32_t)__shfl_xor((int)off, 16, 64); tot += (uint32_t)__shfl_xor((int)tot, 16, 64);
        off += (uint32_t)__shfl_xor((int)off, 32, 64); tot += (uint32_t)__shfl_xor((int)tot, 32, 64);
    }
    const uint32_t t8 = lane < 8u ? tot : 0u;                            // lanes 0..7: the granule's units per bin
    const uint32_t incl8 = wave_scan_incl(t8);
    const uint32_t where = incl8 - t8 + off;                             // lane b < 8: this wave's run of bin b in the slab
    const uint32_t units = lane_value(incl8, 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t at = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) at = (bin[j] == (uint32_t)b) ? lane_value(where, b) : at;
        if (bin[j] < 7u) slab[at + pos[j]] = (uint16_t)(threadIdx.x * 4u + (uint32_t)j);
    }
    __syncthreads();                                                     // the slab and the histogram are complete
    // the slab as it stands: 16 bytes per thread (entries past the granule's units are not written)
    if (threadIdx.x < (uint32_t)(XM_GRAN / 8) && threadIdx.x * 8u < units)
        reinterpret_cast<uint4 *>(rs.runs16 + (uint64_t)g * XM_GRAN)[threadIdx.x] = reinterpret_cast<const uint4 *>(slab)[threadIdx.x];
    if (wave == 0u) {
        if (lane < 8u) rs.gran_counts16[(uint64_t)g * 8u + lane] = (uint16_t)tot;
        const uint4 h0 = *reinterpret_cast<const uint4 *>(lds + lane * XM_HREP);
        const uint4 h1 = *reinterpret_cast<const uint4 *>(lds + lane * XM_HREP + 4);
        const uint32_t sum = h0.x + h0.y + h0.z + h0.w + h1.x + h1.y + h1.z + h1.w;
        if (sum != 0u) atomicAdd(&rs.counts_rep[(g % XM_COUNT_REPLICAS) * 64u + lane], (unsigned long long)sum);
    }
}

template <typename T, bool PAIRED, bool NT, int BLOCK, int BINMODE>
__global__ void __launch_bounds__(BLOCK)
classify_runs_kernel(const T *__restrict__ as1, const T *__restrict__ xs1, const T *__restrict__ as2, const T *__restrict__ xs2,
                     const uint8_t *__restrict__ unit_bits8, T m, uint64_t n, RunsSink rs)
{
    __shared__ uint32_t last_state[BLOCK / 64];
    __shared__ __attribute__((aligned(16))) uint32_t lds[XM_RUNS_LDS_WORDS];
    for (uint32_t q = threadIdx.x; q < 64u * XM_HREP; q += BLOCK) lds[q] = 0;
    if (((uint64_t)blockIdx.x + 1) * (BLOCK * 4) <= n)
        classify_runs_body<T, PAIRED, NT, BLOCK, true, BINMODE>(as1, xs1, as2, xs2, unit_bits8, m, n, last_state, lds, rs);
    else
        classify_runs_body<T, PAIRED, NT, BLOCK, false, BINMODE>(as1, xs1, as2, xs2, unit_bits8, m, n, last_state, lds, rs);
}

// second, 1-workgroup launch: category_counts from the replicas (left zeroed), the eight list lengths from the counts
__global__ void __launch_bounds__(256)
runs_finish_kernel(unsigned long long *__restrict__ counts_rep, unsigned long long *__restrict__ counts,
                   unsigned long long *__restrict__ n_out, int mode)
{
    __shared__ unsigned long long part[4][64];
    __shared__ unsigned long long bins[8];
    const uint32_t slot = threadIdx.x & 63u, q = threadIdx.x >> 6;
    unsigned long long acc = 0;
#pragma unroll
    for (uint32_t r = 0; r < XM_COUNT_REPLICAS / 4u; ++r) {
        acc += counts_rep[(q * (XM_COUNT_REPLICAS / 4u) + r) * 64u + slot];
        counts_rep[(q * (XM_COUNT_REPLICAS / 4u) + r) * 64u + slot] = 0;
    }
    part[q][slot] = acc;
    if (threadIdx.x < 8u) bins[threadIdx.x] = 0;
    __syncthreads();
    if (q == 0u) {
        const unsigned long long v = part[0][slot] + part[1][slot] + part[2][slot] + part[3][slot];
        counts[slot] = v;
        if (v != 0ull) {
            atomicAdd(&bins[bin_of_code(mode, slot)], v);                 // a counted slot is never 0xFF: bin <= 6
            atomicAdd(&bins[7], v);
        }
    }
    __syncthreads();
    if (threadIdx.x < 8u) n_out[threadIdx.x] = bins[threadIdx.x];
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

// the same for an op of length < 2^24, branch- and compare-free: penalty = A[op] + B[op] * len with the two 9-entry tables
// packed into nibbles (A: I, D -> 5; B: I, D -> 3, S -> 2); op codes above 8 are not CIGAR operations and score nothing
__device__ __forceinline__ uint32_t cigar_term24(uint32_t v)
{
    uint32_t op = v & 15u;
    op = op < 8u ? op : 8u;
    const uint32_t a = __builtin_amdgcn_ubfe(0x00000550u, op * 4u, 4u);
    const uint32_t b = __builtin_amdgcn_ubfe(0x00020330u, op * 4u, 4u);     // offset 32 (op 8) reads as offset 0: nibble 0 = 0
    return __umul24(b, v >> 4) + a;
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
        uint32_t prev = 0u;                                              // records past the end own no ops: begin = end
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint32_t v = (r0 + j <= n) ? off[r0 + j] : prev;
            o[j] = v;
            prev = v;
        }
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
    // no fused counting here: the counting epilogue cost this kernel 36-45 us per 50 M pairs, more than the separate
    // histogram pass (30 us) it would save; K1p (packed columns) is the counting form of the --cigar_scores path
    const CountSink none = {nullptr, nullptr, nullptr, nullptr, 0u, 0, 0u};
    classify_finish<int32_t, PAIRED, BLOCK, FULL, false, -1>(a1, x1, a2, x2, m, mb, halo, last_state, code, r0, n, nullptr, none);
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
// K1p: classify from PACKED CIGAR columns (round 3; what the --cigar_scores path runs: xm_classify_compact_cigar*).
// Same arithmetic as K1c (get_cigarbased_AS_tag :228-256 fused into the state function), other column contract:
// per species and record NM (int32), XS (int32) and ONE BYTE -- the number of CIGAR ops of the record -- instead of
// a 4-byte CSR offset; per tile of 256 records (= one wave) one u32, the position in the op array where the tile's
// ops begin (cig_tile[n_tiles + 1]).  A record with 255 ops or more carries 255 and its ops are followed by one
// trailer word (n_ops << 4 | 15: op code 15 scores nothing), so that every begin can be found from the tile's end.
//
// What that buys on the device: the tile base is wave-uniform (scalar loads), so the op stretch of the wave is known
// before any per-record load comes back -- NM, XS, the counts and the first 512 ops of both species are all fetched
// at once: ONE memory latency per wave instead of K1c's two (offsets, then ops).  The records' positions inside the
// stretch are a DPP prefix sum of the counts.  Scores as in K1c: every lane turns 8 consecutive ops into penalty terms, a
// DPP prefix sum of the terms goes to a per-wave LDS table T, a record's penalty is T[end] - T[begin].  The kernel
// counts its units and writes the compact category stream exactly as the counting K1 does (workgroup = granule).
// Anything unusual -- an escape byte, a tile with 1024 ops or more, an op of length >= 2^20, counts that do not add
// up to the tile's stretch, the last (partial) workgroup -- is settled wave-uniformly by the bounds-checked per-record
// path (cigp_slow), 64-bit accumulation, no read past cig_tile[n_tiles] whatever the columns hold.
// ---------------------------------------------------------------------------------------------
#define XM_CIGP_CHUNK 512u          // ops per prefix pass: 8 per lane

struct TileOps {
    uint32_t tb, te;                // the tile's stretch [tb, te) of the op array
    uint32_t v[8];                  // ops tb + 8 lane .. + 7 (first chunk), 0 where past the stretch
    bool fast;                      // wave-uniform: stretch sane, shorter than XM_CIG_WAVE_OPS, 16-byte loads stay inside the array
    bool brief;                     // wave-uniform: fast and fewer than 256 ops -- four op slots per lane do (v[0..3] only)
};

// ops [base + s0, base + s0 + 8) of a stretch of W ops with two 16-byte loads, unconditionally (a conditional load would make
// the compiler wait for it at the join): lanes past the stretch read its first words instead, and whatever lies past the
// stretch is masked by the caller.  The over-read (< 8 words) stays inside the array because the caller has checked
// base + W + 8 <= n_ops (TileOps::fast).
__device__ __forceinline__ void cigp_load8(const uint32_t *__restrict__ ops, uint32_t base, uint32_t W, uint32_t s0, uint32_t v[8])
{
    const uint32_t *p = ops + base + ((s0 < W) ? s0 : 0u);
    // non-temporal like every other input of the kernel: read once.  (Measured: with plain loads the ops push the compact
    // category stream this kernel writes out of the cache before K2c reads it -- K2c 70 instead of 48 us per 50 M pairs.)
    const v4i32 a = __builtin_nontemporal_load(reinterpret_cast<const v4i32_a4 *>(p));
    const v4i32 b = __builtin_nontemporal_load(reinterpret_cast<const v4i32_a4 *>(p + 4));
    v[0] = (uint32_t)a.x; v[1] = (uint32_t)a.y; v[2] = (uint32_t)a.z; v[3] = (uint32_t)a.w;
    v[4] = (uint32_t)b.x; v[5] = (uint32_t)b.y; v[6] = (uint32_t)b.z; v[7] = (uint32_t)b.w;
}

// the same for a stretch of fewer than 256 ops: ops [base + s0, base + s0 + 4), one 16-byte load per lane
__device__ __forceinline__ void cigp_load4(const uint32_t *__restrict__ ops, uint32_t base, uint32_t W, uint32_t s0, uint32_t v[8])
{
    const uint32_t *p = ops + base + ((s0 < W) ? s0 : 0u);
    const v4i32 a = __builtin_nontemporal_load(reinterpret_cast<const v4i32_a4 *>(p));
    v[0] = (uint32_t)a.x; v[1] = (uint32_t)a.y; v[2] = (uint32_t)a.z; v[3] = (uint32_t)a.w;
}

// ... and its prefix pass over 256 op slots (4 per lane): half the term arithmetic of cigp_chunk.  Typical tiles hold
// ~200 ops (k = 0.78 per record), so this is the usual case.
__device__ __forceinline__ uint32_t cigp_chunk4(const uint32_t v[8], uint32_t s0, uint32_t *T, uint32_t &any)
{
    uint32_t p[4], run = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        p[q] = run;
        any |= v[q];
        run += cigar_term24(v[q]);
    }
    const uint32_t tincl = wave_scan_incl(run);
    const uint32_t ex = tincl - run;
    *reinterpret_cast<uint4 *>(T + s0) = make_uint4(ex + p[0], ex + p[1], ex + p[2], ex + p[3]);
    return lane_value(tincl, 63);
}

// One prefix pass over 512 op slots (8 per lane, slot s0 + q in v[q]): penalty terms, their running sums inside the lane, a
// DPP scan over the lanes, T[s0 + q] = carry + the terms of the slots below s0 + q.  Returns carry + the pass's total.
// No masking of the slots past the stretch (the loads run up to 7 words over it, lanes behind it re-read its first
// words): T[j] sums the slots below j, and only T[0 .. W] is ever read.  A long op out there can at worst send the wave
// to the careful path (`any` collects the bits of every word seen).
__device__ __forceinline__ uint32_t cigp_chunk(const uint32_t v[8], uint32_t s0, uint32_t carry, uint32_t *T, uint32_t &any)
{
    uint32_t p[8], run = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        p[q] = run;                                                      // exclusive inside the lane
        any |= v[q];
        run += cigar_term24(v[q]);
    }
    const uint32_t tincl = wave_scan_incl(run);
    const uint32_t ex = tincl - run + carry;
    *reinterpret_cast<uint4 *>(T + s0) = make_uint4(ex + p[0], ex + p[1], ex + p[2], ex + p[3]);
    *reinterpret_cast<uint4 *>(T + s0 + 4u) = make_uint4(ex + p[4], ex + p[5], ex + p[6], ex + p[7]);
    return carry + lane_value(tincl, 63);
}

// The usual case.  cw = the lane's four op counts (one byte each).  false: the wave must take cigp_slow.
__device__ __forceinline__ bool cigp_fast(const uint32_t *__restrict__ ops, const TileOps &o, uint32_t cw,
                                          const int32_t nmv[4], uint32_t *T, int32_t as_out[4], bool &bad)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t W = o.te - o.tb;
    const uint32_t c0 = cw & 255u, c1 = (cw >> 8) & 255u, c2 = (cw >> 16) & 255u, c3 = cw >> 24;
    const uint32_t lsum = c0 + c1 + c2 + c3;
    const uint32_t incl = wave_scan_incl(lsum);
    const bool esc = c0 == 255u || c1 == 255u || c2 == 255u || c3 == 255u;
    if (lane_value(incl, 63) != W || __ballot(esc) != 0ull) return false;
    // the stretch is shorter than XM_CIG_WAVE_OPS = 2 chunks: the first (fetched by the caller) always -- it holds slot W,
    // the grand total, whenever W < 512 --, the second only for a stretch of 512 ops or more
    uint32_t any = 0;
    uint32_t carry = o.brief ? cigp_chunk4(o.v, 4u * lane, T, any) : cigp_chunk(o.v, 8u * lane, 0u, T, any);
    if (W >= XM_CIGP_CHUNK) {
        uint32_t v[8];
        cigp_load8(ops, o.tb, W, XM_CIGP_CHUNK + 8u * lane, v);
        carry = cigp_chunk(v, XM_CIGP_CHUNK + 8u * lane, carry, T, any);
    }
    const bool odd = any >= (1u << 24);                                  // some op of length >= 2^20
    if (__ballot(odd) != 0ull) return false;
    const uint32_t b0 = incl - lsum, b1 = b0 + c0, b2 = b1 + c1, b3 = b2 + c2, b4 = b3 + c3;      // <= W < XM_CIG_WAVE_OPS
    const uint32_t t0 = T[b0], t1 = T[b1], t2 = T[b2], t3 = T[b3], t4 = T[b4];
    const uint32_t d[4] = {t1 - t0, t2 - t1, t3 - t2, t4 - t3};
    // 32-bit scores: exact and inside int32 while |NM| < 2^23 and the penalty < 2^30 (|6 NM| < 2^26).  A wave with a
    // record outside that (never, in practice) takes the careful path, which works in 64 bits and reports overflow.
    // (64-bit scores here: +2 us per 50 M pairs, profiles/r03_ab_cigp.txt.)
    bool big = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool present = nmv[j] != INT32_MIN;
        const int32_t nme = present ? nmv[j] : 0;
        big |= (uint32_t)(nme + (1 << 23)) >= (1u << 24) || d[j] >= (1u << 30);
        as_out[j] = present ? __mul24(nme, -6) - (int32_t)d[j] : INT32_MIN;
    }
    (void)bad;
    return __ballot(big) == 0ull;
}

// one record's score from ops [b, e), bounds-checked against the array
__device__ __forceinline__ int32_t cigp_score_range(const uint32_t *__restrict__ ops, uint32_t n_ops, int32_t nmv,
                                                    uint32_t b, uint32_t e, uint32_t *range_flag)
{
    if (nmv == INT32_MIN) return INT32_MIN;                               // :248-249
    long long s = -6ll * (long long)nmv;
    if (e > n_ops) e = n_ops;
    for (uint32_t k = b; k < e; ++k) s -= cigar_term(ops[k]);
    return cigar_clamp(s, range_flag);
}

// number of ops of an escaped record whose stretch (trailer included) ends at `end`
__device__ __forceinline__ uint32_t cigp_trailer(const uint32_t *__restrict__ ops, uint32_t n_ops, uint32_t end)
{
    return (end > 0u && end <= n_ops) ? (ops[end - 1u] >> 4) : 0u;
}

// The careful path of one wave and species (rare): begins from the counts -- forwards by prefix sum, or, when the tile
// holds escaped records, backwards from the tile's end, one record after the other -- then one loop per record.
// (inlined on purpose: as a called function it pins everything that lives across the call into the callee-saved VGPR
// blocks v40-47 / v56-63 / v72-79, which costs the kernel two waves per SIMD)
__device__ __forceinline__ v4i32 cigp_slow(const uint32_t *__restrict__ ops, uint32_t n_ops, uint32_t tb, uint32_t te,
                                                     uint32_t cw, v4i32 nmv, uint32_t *range_flag)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (te > n_ops) te = n_ops;
    if (tb > te) tb = te;
    const uint32_t c0 = cw & 255u, c1 = (cw >> 8) & 255u, c2 = (cw >> 16) & 255u, c3 = cw >> 24;
    uint32_t b0, b1, b2, b3, e0, e1, e2, e3;
    if (__ballot(c0 == 255u || c1 == 255u || c2 == 255u || c3 == 255u) == 0ull) {
        const uint32_t lsum = c0 + c1 + c2 + c3;
        b0 = tb + wave_scan_incl(lsum) - lsum;
        e0 = b1 = b0 + c0; e1 = b2 = b1 + c1; e2 = b3 = b2 + c2; e3 = b3 + c3;
    } else {
        b0 = b1 = b2 = b3 = e0 = e1 = e2 = e3 = tb;
        uint32_t end = te;
        for (int L = 63; L >= 0; --L) {
            const uint32_t w = lane_value(cw, L);
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                uint32_t k = (w >> (8 * j)) & 255u, e = end;
                if (k == 255u) { k = cigp_trailer(ops, n_ops, end); e = end > tb ? end - 1u : tb; }
                const uint32_t b = (e - tb >= k) ? e - k : tb;
                if (lane == (uint32_t)L) {
                    if (j == 0) { b0 = b; e0 = e; } else if (j == 1) { b1 = b; e1 = e; }
                    else if (j == 2) { b2 = b; e2 = e; } else { b3 = b; e3 = e; }
                }
                end = b;
            }
        }
    }
    v4i32 r;
    r.x = cigp_score_range(ops, n_ops, nmv.x, b0, e0, range_flag);
    r.y = cigp_score_range(ops, n_ops, nmv.y, b1, e1, range_flag);
    r.z = cigp_score_range(ops, n_ops, nmv.z, b2, e2, range_flag);
    r.w = cigp_score_range(ops, n_ops, nmv.w, b3, e3, range_flag);
    return r;
}

// AS of record h, the last record in front of a workgroup, whose ops end where the workgroup's first tile begins
__device__ __attribute__((noinline)) int32_t cigp_score_before(const int32_t *__restrict__ nm, const uint8_t *__restrict__ cnt,
                                                               const uint32_t *__restrict__ ops, uint32_t n_ops, uint64_t h,
                                                               uint32_t end, uint32_t *range_flag)
{
    if (end > n_ops) end = n_ops;
    uint32_t k = cnt[h], e = end;
    if (k == 255u) { k = cigp_trailer(ops, n_ops, end); e = end > 0u ? end - 1u : 0u; }
    return cigp_score_range(ops, n_ops, nm[h], e >= k ? e - k : 0u, e, range_flag);
}

// the lane's four count bytes; records past the end count nothing
template <bool FULL>
__device__ __forceinline__ uint32_t cigp_counts4(const uint8_t *__restrict__ cnt, uint64_t r0, uint64_t n)
{
    if (FULL || r0 + 4 <= n) return __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(cnt + r0));
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (r0 + j < n) w |= (uint32_t)cnt[r0 + j] << (8 * j);
    return w;
}

template <bool PAIRED, int BLOCK, bool FULL, bool COUNTS, int BINMODE>
__device__ __forceinline__ void classify_cigp_body(const CigCols s1, const CigCols s2, const uint8_t *__restrict__ unit_bits8,
                                                   int32_t m, uint8_t *__restrict__ code, uint64_t n, uint32_t n_tiles,
                                                   uint32_t *__restrict__ range_flag, uint32_t *last_state, uint32_t *cig_T,
                                                   uint32_t *count_lds, const CountSink &sink)
{
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t t = blockIdx.x * (BLOCK / 64) + wave;                 // this wave's tile
    const uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t r0 = g * 4;
    const bool have_tile = FULL || t < n_tiles;                          // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;

    // What the scores need is fetched at once: NM and the op counts of both species (no dependency at all) ...
    int32_t a1[4], x1[4], a2[4], x2[4], nmv1[4], nmv2[4];
    uint32_t cw1, cw2, mb = 0;
    const uint64_t wg0 = (uint64_t)blockIdx.x * (BLOCK * 4);              // the workgroup's first record (uniform)
    const uint32_t t16 = threadIdx.x * 16u, t4 = threadIdx.x * 4u;
    if (FULL) {
        load4_wg<int32_t, true>(s1.nm + wg0, t16, nmv1);
        load4_wg<int32_t, true>(s2.nm + wg0, t16, nmv2);
        cw1 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(s1.cnt + wg0 + t4));
        cw2 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(s2.cnt + wg0 + t4));
    } else {
        load4<int32_t, true, false>(s1.nm, r0, n, nmv1);
        load4<int32_t, true, false>(s2.nm, r0, n, nmv2);
        cw1 = cigp_counts4<false>(s1.cnt, r0, n);
        cw2 = cigp_counts4<false>(s2.cnt, r0, n);
    }
    // ... the tile bases of both species (wave-uniform: scalar loads, one wait for all of them) ...
    TileOps o1, o2;
    o1.tb = o1.te = o2.tb = o2.te = 0u;
    const uint32_t n_ops1 = s1.tile[n_tiles], n_ops2 = s2.tile[n_tiles];
    if (have_tile) { o1.tb = s1.tile[t]; o1.te = s1.tile[t + 1u]; o2.tb = s2.tile[t]; o2.te = s2.tile[t + 1u]; }
    o1.fast = FULL && o1.te >= o1.tb && o1.te - o1.tb < (uint32_t)XM_CIG_WAVE_OPS && (uint64_t)o1.te + 8u <= n_ops1;
    o2.fast = FULL && o2.te >= o2.tb && o2.te - o2.tb < (uint32_t)XM_CIG_WAVE_OPS && (uint64_t)o2.te + 8u <= n_ops2;
    // ... and the first 512 ops of both stretches.
#ifndef XM_CIGP_BRIEF
#define XM_CIGP_BRIEF 1            // 0 in a tuning build: always eight op slots per lane
#endif
    o1.brief = XM_CIGP_BRIEF && o1.fast && o1.te - o1.tb < 256u;
    o2.brief = XM_CIGP_BRIEF && o2.fast && o2.te - o2.tb < 256u;
    if (o1.brief) cigp_load4(s1.ops, o1.tb, o1.te - o1.tb, 4u * lane, o1.v);
    else if (o1.fast) cigp_load8(s1.ops, o1.tb, o1.te - o1.tb, 8u * lane, o1.v);
    if (o2.brief) cigp_load4(s2.ops, o2.tb, o2.te - o2.tb, 4u * lane, o2.v);
    else if (o2.fast) cigp_load8(s2.ops, o2.tb, o2.te - o2.tb, 8u * lane, o2.v);

    bool bad = false;
    if (!(o1.fast && cigp_fast(s1.ops, o1, cw1, nmv1, cig_T, a1, bad))) {
        v4i32 q; q.x = nmv1[0]; q.y = nmv1[1]; q.z = nmv1[2]; q.w = nmv1[3];
        q = cigp_slow(s1.ops, n_ops1, o1.tb, o1.te, cw1, q, range_flag);
        a1[0] = q.x; a1[1] = q.y; a1[2] = q.z; a1[3] = q.w;
    }
    // XS and the unit mask are only needed by the state function at the very end: fetched here, they arrive while
    // species 2 is worked on.  (Where the compiler puts them makes no measurable difference, nor does the occupancy:
    // 63 registers; held to six waves per SIMD the kernel is exactly as fast -- profiles/r03_ab_cigp.txt.)
    if (FULL) {
        load4_wg<int32_t, true>(s1.xs + wg0, t16, x1);
        load4_wg<int32_t, true>(s2.xs + wg0, t16, x2);
        mb = (uint32_t)((unit_bits8 + (wg0 >> 3))[threadIdx.x >> 1] >> ((threadIdx.x & 1u) * 4u)) & 0xFu;
    } else {
        load4<int32_t, true, false>(s1.xs, r0, n, x1);
        load4<int32_t, true, false>(s2.xs, r0, n, x2);
        if (r0 < n) mb = (uint32_t)(unit_bits8[g >> 1] >> ((g & 1u) * 4u)) & 0xFu;
        if (r0 + 4 > n) mb &= (r0 < n) ? ((1u << (uint32_t)(n - r0)) - 1u) : 0u;
    }
    if (!(o2.fast && cigp_fast(s2.ops, o2, cw2, nmv2, cig_T, a2, bad))) {
        v4i32 q; q.x = nmv2[0]; q.y = nmv2[1]; q.z = nmv2[2]; q.w = nmv2[3];
        q = cigp_slow(s2.ops, n_ops2, o2.tb, o2.te, cw2, q, range_flag);
        a2[0] = q.x; a2[1] = q.y; a2[2] = q.z; a2[3] = q.w;
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63u) == 0u && range_flag) atomicOr(range_flag, 1u);

    // the record in front of the workgroup: only when the workgroup's first record closes a unit (never with strictly
    // interleaved mates: workgroups begin at even records)
    uint32_t halo = 0;
    if (PAIRED && threadIdx.x == 0) {
        if (r0 == 0) {
            mb &= ~1u;                                                    // record 0 has no predecessor (:402)
        } else if (mb & 1u) {
            const uint64_t h = r0 - 1;
            halo = mapping_state<int32_t>(cigp_score_before(s1.nm, s1.cnt, s1.ops, n_ops1, h, o1.tb, range_flag), s1.xs[h],
                                          cigp_score_before(s2.nm, s2.cnt, s2.ops, n_ops2, h, o2.tb, range_flag), s2.xs[h], m);
        }
    }
    classify_finish<int32_t, PAIRED, BLOCK, FULL, COUNTS, BINMODE>(a1, x1, a2, x2, m, mb, halo, last_state, code, r0, n,
                                                                   count_lds, sink);
}

template <bool PAIRED, int BLOCK, bool COUNTS, int BINMODE>
__global__ void __launch_bounds__(BLOCK)
classify_cigp_kernel(const CigCols s1, const CigCols s2, const uint8_t *__restrict__ unit_bits8, int32_t m,
                     uint8_t *__restrict__ code, uint64_t n, uint32_t n_tiles, uint32_t *__restrict__ range_flag, CountSink sink)
{
    static_assert(!COUNTS || BLOCK * 4 == XM_GRAN, "the counting workgroup is one granule");
    __shared__ uint32_t last_state[BLOCK / 64];
    __shared__ __attribute__((aligned(16))) uint32_t cig_table[BLOCK / 64][XM_CIG_WAVE_OPS + 8];
    __shared__ __attribute__((aligned(16))) uint32_t count_lds[COUNTS ? XM_COUNT_LDS_WORDS : 4];
    uint32_t *cig_T = cig_table[threadIdx.x >> 6];
    if (COUNTS) count_lds_clear<BLOCK>(count_lds);
    if (((uint64_t)blockIdx.x + 1) * (BLOCK * 4) <= n)
        classify_cigp_body<PAIRED, BLOCK, true, COUNTS, BINMODE>(s1, s2, unit_bits8, m, code, n, n_tiles, range_flag, last_state,
                                                                 cig_T, count_lds, sink);
    else
        classify_cigp_body<PAIRED, BLOCK, false, COUNTS, BINMODE>(s1, s2, unit_bits8, m, code, n, n_tiles, range_flag, last_state,
                                                                  cig_T, count_lds, sink);
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
// Streaming probe: the classify kernel's memory shape without its arithmetic -- every lane reads 16 bytes of each of the
// four score columns (non-temporal) and writes 2 bytes (one workgroup = XM_GRAN records, one tile per wave, the grid covers
// the input once).  What it reaches is the box's own ceiling for this access pattern ("measure an on-box
// streaming-copy ceiling alongside" the 8 TB/s specification); bench.py reports it as roofline.copy_ceiling_GBps.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(XM_CLASSIFY_BLOCK)
stream_probe_kernel(const v4i32 *__restrict__ c0, const v4i32 *__restrict__ c1, const v4i32 *__restrict__ c2,
                    const v4i32 *__restrict__ c3, uint16_t *__restrict__ out, uint64_t n_groups)
{
    const uint64_t g = (uint64_t)blockIdx.x * XM_CLASSIFY_BLOCK + threadIdx.x;
    if (g >= n_groups) return;
    const v4i32 a = __builtin_nontemporal_load(c0 + g), b = __builtin_nontemporal_load(c1 + g);
    const v4i32 c = __builtin_nontemporal_load(c2 + g), d = __builtin_nontemporal_load(c3 + g);
    const v4i32 x = a ^ b ^ c ^ d;
    out[g] = (uint16_t)(x.x ^ x.y ^ x.z ^ x.w);
}

void launch_stream_probe(hipStream_t st, uint64_t n, const int32_t *c0, const int32_t *c1, const int32_t *c2, const int32_t *c3,
                         uint8_t *out)
{
    const uint64_t n_groups = n / 4;
    const uint32_t grid = (uint32_t)((n_groups + XM_CLASSIFY_BLOCK - 1) / XM_CLASSIFY_BLOCK);
    if (grid == 0) return;
    stream_probe_kernel<<<grid, XM_CLASSIFY_BLOCK, 0, st>>>(reinterpret_cast<const v4i32 *>(c0), reinterpret_cast<const v4i32 *>(c1),
                                                            reinterpret_cast<const v4i32 *>(c2), reinterpret_cast<const v4i32 *>(c3),
                                                            reinterpret_cast<uint16_t *>(out), n_groups);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static_assert(XM_CLASSIFY_BLOCK * 4 == XM_GRAN, "the counting classify workgroup is one granule");

GranPlan plan_granules(uint64_t n)
{
    GranPlan p;
    uint64_t g = (n + XM_GRAN - 1) / XM_GRAN;
    if (g == 0) g = 1;
    p.n_gran = (uint32_t)g;
    p.gran_stride = (p.n_gran + 63u) & ~63u;
    return p;
}

static CountSink make_sink(const CountPlan *cp, int mode)
{
    CountSink s;
    s.gran_counts = cp ? cp->gran_counts : nullptr;
    s.part_tot = cp ? cp->part_tot : nullptr;
    s.counts_rep = cp ? reinterpret_cast<unsigned long long *>(cp->counts_rep) : nullptr;
    s.bins4 = cp ? cp->bins4 : nullptr;
    s.gran_stride = cp ? cp->plan.gran_stride : 0u;
    s.mode = mode;
    s.blk0 = cp ? cp->gran0 : 0u;
    return s;
}

template <typename T>
static void launch_classify_t(hipStream_t st, int mode, uint64_t n,
                              const T *as1, const T *xs1, const T *as2, const T *xs2,
                              const uint64_t *unit_bits, T m, uint8_t *code, const CountPlan *cp)
{
    const uint64_t per_block = (uint64_t)XM_CLASSIFY_BLOCK * 4;
    static_assert(XM_CLASSIFY_BLOCK * 4 == XM_GRAN, "a counting workgroup is a granule: CountPlan's chunk is in both");
    const uint32_t all = (uint32_t)((n + per_block - 1) / per_block);
    const uint32_t grid = (cp && cp->gran1) ? cp->gran1 - cp->gran0 : all;         // a chunk of the input, or all of it
    const uint8_t *bits8 = reinterpret_cast<const uint8_t *>(unit_bits);
    const CountSink sink = make_sink(cp, mode);
    const bool paired = mode != XM_MODE_SE;
#define XM_LAUNCH_CLS(P, C, B) classify_kernel<T, P, XM_CLASSIFY_NT, XM_CLASSIFY_BLOCK, C, B><<<grid, XM_CLASSIFY_BLOCK, 0, st>>>(as1, xs1, as2, xs2, bits8, m, code, n, sink)
    if (cp && cp->bins4) {
        if (mode == XM_MODE_SE) XM_LAUNCH_CLS(false, true, XM_MODE_SE);
        else if (mode == XM_MODE_PE_LIBERAL) XM_LAUNCH_CLS(true, true, XM_MODE_PE_LIBERAL);
        else XM_LAUNCH_CLS(true, true, XM_MODE_PE_CONSERVATIVE);
    }
    else if (cp) { if (paired) XM_LAUNCH_CLS(true, true, -1); else XM_LAUNCH_CLS(false, true, -1); }
    else         { if (paired) XM_LAUNCH_CLS(true, false, -1); else XM_LAUNCH_CLS(false, false, -1); }
#undef XM_LAUNCH_CLS
}

void launch_classify_i32(hipStream_t st, int mode, uint64_t n,
                         const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                         const uint64_t *unit_bits, int32_t m, uint8_t *code, const CountPlan *cp)
{
    launch_classify_t<int32_t>(st, mode, n, as1, xs1, as2, xs2, unit_bits, m, code, cp);
}

void launch_classify_f64(hipStream_t st, int mode, uint64_t n,
                         const double *as1, const double *xs1, const double *as2, const double *xs2,
                         const uint64_t *unit_bits, double m, uint8_t *code, const CountPlan *cp)
{
    launch_classify_t<double>(st, mode, n, as1, xs1, as2, xs2, unit_bits, m, code, cp);
}

template <typename T>
static void launch_classify_runs_t(hipStream_t st, int mode, uint64_t n,
                                   const T *as1, const T *xs1, const T *as2, const T *xs2,
                                   const uint64_t *unit_bits, T m, const RunsOut &ro)
{
    const uint64_t per_block = (uint64_t)XM_CLASSIFY_BLOCK * 4;
    const uint32_t grid = (uint32_t)((n + per_block - 1) / per_block);
    const uint8_t *bits8 = reinterpret_cast<const uint8_t *>(unit_bits);
    RunsSink rs;
    rs.runs16 = ro.runs16;
    rs.gran_counts16 = ro.gran_counts16;
    rs.counts_rep = reinterpret_cast<unsigned long long *>(ro.counts_rep);
#define XM_LAUNCH_RUNS(P, B) classify_runs_kernel<T, P, XM_CLASSIFY_NT, XM_CLASSIFY_BLOCK, B><<<grid, XM_CLASSIFY_BLOCK, 0, st>>>(as1, xs1, as2, xs2, bits8, m, n, rs)
    if (mode == XM_MODE_SE) XM_LAUNCH_RUNS(false, XM_MODE_SE);
    else if (mode == XM_MODE_PE_LIBERAL) XM_LAUNCH_RUNS(true, XM_MODE_PE_LIBERAL);
    else XM_LAUNCH_RUNS(true, XM_MODE_PE_CONSERVATIVE);
#undef XM_LAUNCH_RUNS
}

void launch_classify_runs_i32(hipStream_t st, int mode, uint64_t n,
                              const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                              const uint64_t *unit_bits, int32_t m, const RunsOut &ro)
{
    launch_classify_runs_t<int32_t>(st, mode, n, as1, xs1, as2, xs2, unit_bits, m, ro);
}

void launch_classify_runs_f64(hipStream_t st, int mode, uint64_t n,
                              const double *as1, const double *xs1, const double *as2, const double *xs2,
                              const uint64_t *unit_bits, double m, const RunsOut &ro)
{
    launch_classify_runs_t<double>(st, mode, n, as1, xs1, as2, xs2, unit_bits, m, ro);
}

void launch_runs_finish(hipStream_t st, int mode, uint64_t *counts_rep, uint64_t *counts, uint64_t *n_out)
{
    runs_finish_kernel<<<1, 256, 0, st>>>(reinterpret_cast<unsigned long long *>(counts_rep),
                                          reinterpret_cast<unsigned long long *>(counts),
                                          reinterpret_cast<unsigned long long *>(n_out), mode);
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

void launch_classify_cigp(hipStream_t st, int mode, uint64_t n, const CigCols &s1, const CigCols &s2,
                          const uint64_t *unit_bits, int32_t m, uint8_t *code, uint32_t *range_flag, const CountPlan &cp)
{
    const uint64_t per_block = (uint64_t)XM_CIGP_BLOCK * 4;
    const uint32_t grid = (uint32_t)((n + per_block - 1) / per_block);
    const uint32_t n_tiles = (uint32_t)((n + 255) / 256);
    const uint8_t *bits8 = reinterpret_cast<const uint8_t *>(unit_bits);
    const CountSink sink = make_sink(&cp, mode);
#define XM_LAUNCH_CIGP(P, B) classify_cigp_kernel<P, XM_CIGP_BLOCK, true, B><<<grid, XM_CIGP_BLOCK, 0, st>>>(s1, s2, bits8, m, code, n, n_tiles, range_flag, sink)
    if (cp.bins4) {
        if (mode == XM_MODE_SE) XM_LAUNCH_CIGP(false, XM_MODE_SE);
        else if (mode == XM_MODE_PE_LIBERAL) XM_LAUNCH_CIGP(true, XM_MODE_PE_LIBERAL);
        else XM_LAUNCH_CIGP(true, XM_MODE_PE_CONSERVATIVE);
    } else {
        if (mode == XM_MODE_SE) XM_LAUNCH_CIGP(false, -1); else XM_LAUNCH_CIGP(true, -1);
    }
#undef XM_LAUNCH_CIGP
}

void launch_hist(hipStream_t st, int mode, uint64_t n, const uint8_t *code, const CountPlan &cp)
{
    const uint32_t grid = (cp.plan.n_gran + (XM_BLOCK / 64) - 1) / (XM_BLOCK / 64);
    hist_kernel<<<grid, XM_BLOCK, 0, st>>>(code, n, cp.plan.n_gran, make_sink(&cp, mode));
}

void launch_scan(hipStream_t st, const CountPlan &cp, uint32_t *gran_off, uint64_t *bin_totals, uint64_t *counts)
{
    const uint32_t n_parts = (cp.plan.n_gran + XM_PART_GRAN - 1) / XM_PART_GRAN;
    unsigned long long *bt = reinterpret_cast<unsigned long long *>(bin_totals);
    unsigned long long *rep = reinterpret_cast<unsigned long long *>(cp.counts_rep);
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(counts);
    // a chunk [gran0, gran1) begins at a part boundary; the scan sees the granules up to the chunk's end
    const uint32_t part0 = cp.gran1 ? cp.gran0 / XM_PART_GRAN : 0u;
    const uint32_t part1 = cp.gran1 ? (cp.gran1 + XM_PART_GRAN - 1) / XM_PART_GRAN : n_parts;
    scan_kernel<<<dim3(part1 - part0, 8), XM_SCAN_THREADS, 0, st>>>(cp.gran_counts, cp.gran1 ? cp.gran1 : cp.plan.n_gran, cp.plan.gran_stride,
                                                                   cp.part_tot, gran_off, bt, rep, cnt, part0, n_parts);
}

// XM_SCATTER_WAVES (xm_kernels.h) waves at most; the environment variable of the same name (read per call, like
// XM_PLACE_CHUNK_PARTS) lowers it so that tests reach the run-of-granules form without 2^31 records
static uint32_t scatter_waves()
{
    const char *e = getenv("XM_SCATTER_WAVES");
    const long v = e && *e ? strtol(e, nullptr, 10) : 0;
    return v >= 1 && v < (long)XM_SCATTER_WAVES ? (uint32_t)v : (uint32_t)XM_SCATTER_WAVES;
}

void launch_scatter(hipStream_t st, const GranPlan &p, int mode, uint64_t n, const uint8_t *code, bool code_is_bins4,
                    const uint32_t *gran_counts, const uint32_t *gran_off, const uint64_t *bin_totals, uint64_t *bin_offsets,
                    uint32_t *idx_out, uint32_t *part_tot, const ListOut *lists, uint32_t gran0, uint32_t gran1)
{
    const unsigned long long *bt = reinterpret_cast<const unsigned long long *>(bin_totals);
    unsigned long long *bo = reinterpret_cast<unsigned long long *>(bin_offsets);
    // [gran0, gran1): a chunk of a chunked call (lists only), gran1 == 0: the whole input
    const uint32_t g_first = gran1 ? gran0 : 0u, g_end = gran1 ? gran1 : p.n_gran, span = g_end - g_first;
    const uint32_t last = g_end == p.n_gran ? 1u : 0u;
    // one granule per wave; beyond scatter_waves() granules a wave takes a run of consecutive ones (scatter_kernel, MULTI)
    const uint32_t max_waves = scatter_waves();
    const uint32_t gran_per_wave = (span + max_waves - 1u) / max_waves;
    const bool multi = gran_per_wave > 1u;
    const uint32_t n_waves = multi ? (span + gran_per_wave - 1u) / gran_per_wave : span;
    const uint32_t grid = (n_waves + (XM_BLOCK / 64) - 1) / (XM_BLOCK / 64);
    const bool wide = n > (1ull << 30);                 // unit positions * 4 bytes may pass 2^32
    const bool stage = mode == XM_MODE_SE;
    const ListOut lo = lists ? *lists : ListOut{{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, 0u};
#define XM_LAUNCH_SCT(W, NIB, STG, L, M) scatter_kernel<XM_GRAN / 256, W, NIB, STG, L, M><<<grid, XM_BLOCK, 0, st>>>(code, n, mode, g_end, p.gran_stride, gran_counts, gran_off, bt, bo, idx_out, part_tot, lo, gran_per_wave, g_first, last)
#define XM_LAUNCH_SCT0(W, NIB, STG, L) do { if (multi) XM_LAUNCH_SCT(W, NIB, STG, L, true); else XM_LAUNCH_SCT(W, NIB, STG, L, false); } while (0)
#define XM_LAUNCH_SCT1(W, NIB, STG) do { if (lists) XM_LAUNCH_SCT0(W, NIB, STG, true); else XM_LAUNCH_SCT0(W, NIB, STG, false); } while (0)
#define XM_LAUNCH_SCT2(W, NIB) do { if (stage) XM_LAUNCH_SCT1(W, NIB, true); else XM_LAUNCH_SCT1(W, NIB, false); } while (0)
    if (code_is_bins4) { if (wide) XM_LAUNCH_SCT2(true, true); else XM_LAUNCH_SCT2(false, true); }
    else               { if (wide) XM_LAUNCH_SCT2(true, false); else XM_LAUNCH_SCT2(false, false); }
#undef XM_LAUNCH_SCT2
#undef XM_LAUNCH_SCT1
#undef XM_LAUNCH_SCT0
#undef XM_LAUNCH_SCT
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
