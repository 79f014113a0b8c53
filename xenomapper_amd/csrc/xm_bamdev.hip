// xm_bamdev.hip -- BAM records -> the classifier's columns on the GPU (include/xenomapper_bgzf.h, xm_bamdev_*).
//
// Per window:  compressed BGZF blocks (page-locked staging) --H2D--> inflate (xm_inflate.hip) + CRC-32.  The inflate launch also does
// the record work of the BLOCKS (xm_bgzf_inflate_walk_dev): the chain of lanes that has written a block follows the block_size chain
// of the alignment records through it and reads the classifier's fields out of every record (xm_bamrec.h parse_record: the fixed
// fields, the name, the optional fields with the tag_func plugins' rules on the TYPED fields -- xm_bam.cpp's TagScan, restating
// get_tag xenomapper.py:176-191 -- everything the text rules might read differently flagged), into per-block slots.  The chain is
// serial, so the parallelism is the blocks -- which only works when every block begins with a record, as htslib / samtools write
// them (bgzf_flush_try in front of every record).  Then, per file:
//   B1 walk_kernel   the pieces of the CARRIED tail (bytes of the previous window), a lane per piece: counts the records of each,
//                    reports where its chain leaves it;
//   B2 scan_kernel   one workgroup: checks that every segment's chain (pieces and blocks) lands on the next segment's first
//                    byte -- else `unaligned` and the caller takes the host decoder --, exclusive scan of the counts;
//   B3 walk_kernel   the pieces again, writing their part of the record table;  gather_kernel: the blocks' part of the table and
//                    of the field arrays, from the slots (a wave per block);
//   B4 parse_kernel  the carried tail's records, a lane per record (the same parse_record);
//   B5 pair_kernel   one lane per PAIR of records k of the two files: score columns, names compared (:106 across files,
//                    :402 between neighbours -> unit mask by ballot), first mismatch by atomicMin.
// Then the fused pass on the columns (xm_bamdev_classify), and with the bins on the device:
//   W1 want_kernel   marks the records a sink takes (a unit's lines come from one file: :423-448) and notes their sizes;
//   W2 size_sum / part_scan / size_place: where each goes;   W3 pack_kernel: the records next to each other.
// Those records, the table of where each went and the record table go back to the host (page-locked, on a copy stream of the
// slot's own), where the writer prints their SAM text (xmh_bam_print); the whole windows only on request (xm_bamdev_fetch_raw).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/xenomapper_bgzf.h"
#include "xm_bamrec.h"
#include "xm_fmtg.h"
#include "xm_gather.h"
#include "xm_pinned.h"

namespace {

constexpr uint64_t CARRY_SEG = 128;                         // records per segment of a carried tail
using xmrec::ABSENT;
using xmrec::R_EX_A_SHIFT; using xmrec::R_EX_X_SHIFT; using xmrec::R_WEIRD; using xmrec::R_BAD;
using xmrec::ld16; using xmrec::ld32;

// ---- B1 / B3: the record chain, one lane per segment --------------------------------------------------------------
// seg_start[n_seg + 1]: first byte of every segment, seg_start[n_seg] = n_raw.  A record belongs to the segment its
// block_size word begins in; a record that does not end inside the window ends the walk (it is the next window's).
// Since the blocks' records are found by the inflate launch, this kernel walks the carried tail's pieces only (a few hundred lanes).
constexpr uint32_t WALK_T = 64;
template <bool FILL>
__global__ void __launch_bounds__(WALK_T)
walk_kernel(const uint8_t *__restrict__ raw, uint32_t n_raw, const uint32_t *__restrict__ seg_start, uint32_t n_seg,
            uint32_t *__restrict__ cnt, uint32_t *__restrict__ exit_at, const uint32_t *__restrict__ base,
            uint32_t *__restrict__ rec_off, uint32_t rec_cap)
{
    const uint32_t s = blockIdx.x * WALK_T + threadIdx.x;
    if (s >= n_seg) return;
    const uint32_t lo = seg_start[s], hi = seg_start[s + 1];
    uint32_t p = lo, n = 0;
    const uint32_t b0 = FILL ? base[s] : 0u;
    while (p < hi) {
        if (n_raw - p < 4u) break;                                          // the size word itself is cut
        const uint32_t size = ld32(raw + p);
        if (size > n_raw - p - 4u) break;                                   // the record continues behind the window
        if (FILL && b0 + n < rec_cap) rec_off[b0 + n] = p;
        ++n;
        p += 4u + size;                                                     // >= p + 4: the walk always advances
    }
    if (!FILL) { cnt[s] = n; exit_at[s] = p; }
}

// Where the records of the block that is segment s (counted among the BLOCK segments: k = s - carried pieces) are noted by the
// inflate launch: a block of alignment records holds one per 36 bytes at most, so slot_of(start, k) = start / 36 + 2 k leaves every
// block isize / 36 + 1 entries of its own.
__host__ __device__ inline uint64_t slot_of(uint32_t start, uint32_t k) { return (uint64_t)start / 36u + 2ull * k; }

struct SlotArrays {
    const uint32_t *off, *name_off, *name_len;
    const int32_t *a, *x;
    const uint8_t *flag;
    const uint32_t *n_cigar, *cig_at;        // --cigar_scores runs only (else null)
};

// B3 + B4 for the blocks: the record table and the per-record fields from the slots (one wave per segment), behind the scan's bases
__global__ void __launch_bounds__(256)
gather_kernel(SlotArrays sa, const uint32_t *__restrict__ seg_start, uint32_t n_carry_seg, uint32_t n_seg,
              const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ base, uint32_t *__restrict__ rec_off,
              uint32_t *__restrict__ name_off, uint32_t *__restrict__ name_len, int32_t *__restrict__ a, int32_t *__restrict__ x,
              uint8_t *__restrict__ flag, uint32_t *__restrict__ n_cigar, uint32_t *__restrict__ cig_at, uint32_t rec_cap)
{
    const uint32_t s = n_carry_seg + blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (s >= n_seg) return;
    const uint32_t n = cnt[s], b0 = base[s];
    const uint64_t so = slot_of(seg_start[s], s - n_carry_seg);
    for (uint32_t i = lane; i < n; i += 64u) {
        const uint32_t at = b0 + i;
        if (at >= rec_cap) break;
        rec_off[at] = sa.off[so + i];
        name_off[at] = sa.name_off[so + i];
        name_len[at] = sa.name_len[so + i];
        a[at] = sa.a[so + i];
        x[at] = sa.x[so + i];
        flag[at] = sa.flag[so + i];
        if (sa.n_cigar != nullptr) { n_cigar[at] = sa.n_cigar[so + i]; cig_at[at] = sa.cig_at[so + i]; }
    }
}

// ---- B2: alignment check + exclusive scan of the per-segment counts (one workgroup) --------------------------------
// summary: [0] records, [1] where the chain stopped (first byte not covered by a complete record), [2] aligned (1 / 0)
__global__ void __launch_bounds__(1024)
scan_kernel(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ exit_at, const uint32_t *__restrict__ seg_start,
            uint32_t n_seg, uint32_t *__restrict__ base, uint32_t *__restrict__ summary)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t bad;
    const uint32_t t = threadIdx.x;
    if (t == 0) bad = 0;
    __syncthreads();
    const uint32_t per = (n_seg + 1023u) / 1024u;
    const uint32_t a = min(t * per, n_seg), b = min(a + per, n_seg);
    uint32_t sum = 0, misaligned = 0;
    for (uint32_t s = a; s < b; ++s) {
        sum += cnt[s];
        // the chain of segment s must land exactly on the first byte of segment s + 1; the last one may stop anywhere
        // (an incomplete record at the end of the window) but must not have been cut short by a later segment's start
        if (s + 1 < n_seg && exit_at[s] != seg_start[s + 1]) misaligned = 1;
    }
    part[t] = sum;
    if (misaligned) atomicOr(&bad, 1u);
    __syncthreads();
    // Hillis-Steele over the 1024 partial sums
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t s = a; s < b; ++s) { base[s] = run; run += cnt[s]; }
    if (t == 1023u) {
        summary[0] = part[1023];
        summary[1] = n_seg ? exit_at[n_seg - 1] : 0u;
        summary[2] = bad ? 0u : 1u;
    }
}

// ---- B4: one lane per record ----------------------------------------------------------------------------------------
struct RecOut {
    uint32_t *name_off;       // first byte of QNAME in raw
    uint32_t *name_len;       // without the NUL
    int32_t *a, *x;           // AS, and XS or ZS by mode; ABSENT = no match
    uint8_t *flag;            // ex_a | ex_x << 2 | R_WEIRD | R_BAD
    uint32_t *n_cigar, *cig_at;   // --cigar_scores runs only (else null)
};

// n_dev: when not null, the number of records is read from there (at most n: the launch's upper bound)
__global__ void __launch_bounds__(256)
parse_kernel(const uint8_t *__restrict__ raw, const uint32_t *__restrict__ rec_off, uint32_t n, const uint32_t *__restrict__ n_dev,
             uint32_t tags /* xmrec::TAGS_* */, RecOut o)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (n_dev != nullptr) { const uint32_t m = *n_dev; n = m < n ? m : n; }
    if (i >= n) return;
    const xmrec::RecFields f = xmrec::parse_record(raw, rec_off[i], tags);
    o.name_off[i] = f.name_off;
    o.name_len[i] = f.name_len;
    o.a[i] = f.a;
    o.x[i] = f.x;
    o.flag[i] = (uint8_t)f.flag;
    if (o.n_cigar != nullptr) { o.n_cigar[i] = f.n_cigar; o.cig_at[i] = f.cig_at; }
}

// ---- the records a sink takes, packed for the way back ---------------------------------------------------------------------
// The writer prints SAM text only for records whose unit goes to a sink that was given, and a unit's lines come from ONE file
// (primary bins: file 1, secondary bins: file 2, unresolved: both; xenomapper.py:423-448) -- about half of every window.  With the
// bins known on the device (the compact category stream of the fused pass), only those records go back over PCIe: W1 marks
// them and notes their sizes, W2 scans the sizes (three small launches), W3 copies the records next to each other.
__global__ void __launch_bounds__(256)
want_kernel(const uint8_t *__restrict__ raw1, const uint8_t *__restrict__ raw2, const uint32_t *__restrict__ rec_off1,
            const uint32_t *__restrict__ rec_off2, const uint8_t *__restrict__ bins4, uint32_t n, int paired, uint32_t sink_mask,
            uint32_t *__restrict__ wsize1, uint32_t *__restrict__ wsize2)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t w = files_of_bin((bins4[i >> 1] >> ((i & 1u) * 4u)) & 15u, sink_mask);
    if (paired && i + 1u < n) w |= files_of_bin((bins4[(i + 1u) >> 1] >> (((i + 1u) & 1u) * 4u)) & 15u, sink_mask);   // a pair covers records i - 1 and i
    wsize1[i] = (w & 1u) ? 4u + ld32(raw1 + rec_off1[i]) : 0u;
    wsize2[i] = (w & 2u) ? 4u + ld32(raw2 + rec_off2[i]) : 0u;
}

// W3: a wave per record
__global__ void __launch_bounds__(256)
pack_kernel(const uint8_t *__restrict__ raw, const uint32_t *__restrict__ rec_off, const uint32_t *__restrict__ wsize,
            const uint32_t *__restrict__ place, uint32_t n, uint8_t *__restrict__ packed)
{
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= n) return;
    const uint32_t size = wsize[i];
    if (size == 0u) return;
    const uint8_t *src = raw + rec_off[i];
    uint8_t *dst = packed + place[i];
    for (uint32_t k = lane; k < size; k += 64u) dst[k] = src[k];
}

// ---- T: the SAM text of the records a sink takes, printed ON the device -------------------------------------------------------
// What `samtools view` prints for an alignment record (the reference reads BAM through it: getBamReadPairs / bam_lines,
// /root/reference/xenomapper/xenomapper.py:56-93), restated as csrc/xm_bam.cpp's format_record restates it for the host -- the two are
// compared byte for byte in tests/test_bam_gpu.py.  One lane prints one record, twice: first into a sink that only counts (T1; the
// size scan of W2 places the lines), then into the text (T2).  Floating-point fields of the specification's types (f, B:f) are
// printed as printf("%g") prints them, exactly (xm_fmtg.h, round 6); a binary64 field (type d, which the specification does not
// have and htslib accepts) raises a flag and the host printer takes THAT window, as it does for every window the device does not
// vouch for (a CIGAR in a CG:B,I field: `weird`).
struct RefTable {
    const uint8_t *names;        // the reference names back to back
    const uint32_t *at;          // n + 1 positions in `names`
    uint32_t n;
};

// Where the printer reads a record from: the inflated window in global memory, by unaligned loads of 1 to 16 bytes.  Positions count
// from the byte behind the record's block_size word.
typedef uint32_t v4u32_any __attribute__((ext_vector_type(4), aligned(1)));
struct GlobalRec {
    const uint8_t *r;
    __device__ __forceinline__ uint32_t u8(uint32_t p) const { return r[p]; }
    __device__ __forceinline__ uint32_t u16(uint32_t p) const { return ld16(r + p); }
    __device__ __forceinline__ uint32_t u32(uint32_t p) const { return ld32(r + p); }
    __device__ __forceinline__ v4u32_any u128(uint32_t p) const { return *reinterpret_cast<const v4u32_any *>(r + p); }
};
__device__ __forceinline__ uint32_t base_letter(uint32_t code)          // "=ACMGRSVTWYHKDBN"
{
    const unsigned long long lo = 0x565352474D43413Dull, hi = 0x4E42444B48595754ull;
    return (uint32_t)(((code < 8u ? lo : hi) >> (8u * (code & 7u))) & 0xFFu);
}

struct CountChars {
    uint32_t n = 0;
    __device__ __forceinline__ void ch(uint32_t) { ++n; }
    __device__ __forceinline__ void gbytes(const uint8_t *, uint32_t len) { n += len; }
    template <typename R> __device__ __forceinline__ void bytes(const R &, uint32_t, uint32_t len) { n += len; }
    template <typename R> __device__ __forceinline__ void seq(const R &, uint32_t, uint32_t len) { n += len; }
    template <typename R> __device__ __forceinline__ void qual(const R &, uint32_t, uint32_t len) { n += len; }
};

struct __attribute__((packed, aligned(1))) U32Store { uint32_t v; };
// A lane writes its line front to back, reading its record front to back: 64 lanes = 64 streams, every access a cache line of its
// own.  With 100 000 lanes in flight no line survives in a cache between two touches, so HBM traffic is (accesses x sector), not
// bytes: with four bytes per access G2 fetched 11.4 GB and wrote 5.0 GB per window for 0.55 GB of records and 0.9 GB of text (PMC,
// profiles/r06_bam_pmc.txt).  Hence the long fields -- bases, qualities, names: 3/4 of a line -- move SIXTEEN bytes per access
// (unaligned dwordx4 on both sides); the short ones are gathered four at a time.
struct WriteChars {
    uint8_t *o;
    unsigned long long acc = 0;
    uint32_t k = 0;                              // bytes waiting in acc (< 4 between calls)
    uint32_t wide_min = 16u;                     // A/B: 0xFFFFFFFF switches the 16-byte paths off
    __device__ __forceinline__ void word(uint32_t w, uint32_t n_bytes)          // n_bytes <= 4, the bytes above them zero
    {
        acc |= (unsigned long long)w << (8u * k);
        k += n_bytes;
        if (k >= 4u) {
            reinterpret_cast<U32Store *>(o)->v = (uint32_t)acc;
            o += 4;
            acc >>= 32;
            k -= 4u;
        }
    }
    __device__ __forceinline__ void wide(const v4u32_any &v)                    // 16 bytes at once, behind the bytes still waiting
    {
        for (uint32_t j = 0; j < k; ++j) o[j] = (uint8_t)(acc >> (8u * j));
        o += k;
        k = 0;
        acc = 0;
        *reinterpret_cast<v4u32_any *>(o) = v;
        o += 16;
    }
    __device__ __forceinline__ void ch(uint32_t c) { word(c & 0xFFu, 1u); }
    __device__ __forceinline__ void gbytes(const uint8_t *p, uint32_t len)
    {
        uint32_t j = 0;
        for (; j + 4u <= len; j += 4u) word(ld32(p + j), 4u);
        for (; j < len; ++j) word(p[j], 1u);
    }
    template <typename R> __device__ __forceinline__ void bytes(const R &rec, uint32_t p, uint32_t len)
    {
        uint32_t j = 0;
        for (; j + 64u <= len && wide_min == 16u; j += 64u) {                    // (64 bytes fetched at once: see qual())
            const v4u32_any a = rec.u128(p + j), b = rec.u128(p + j + 16u), c = rec.u128(p + j + 32u), d = rec.u128(p + j + 48u);
            wide(a); wide(b); wide(c); wide(d);
        }
        for (; j + 16u <= len && wide_min == 16u; j += 16u) wide(rec.u128(p + j));
        for (; j + 4u <= len; j += 4u) word(rec.u32(p + j), 4u);
        for (; j < len; ++j) word(rec.u8(p + j), 1u);
    }
    template <typename R> __device__ __forceinline__ void seq(const R &rec, uint32_t p, uint32_t len)
    {
        // two bases a byte, the first in the high nibble; 16 bytes of them are 32 letters
        uint32_t j = 0;
        auto letters32 = [&](const v4u32_any &in) {
            uint32_t w8[8];
#pragma unroll
            for (uint32_t d = 0; d < 4u; ++d) {
                const uint32_t x = in[d];
                w8[2u * d] = base_letter((x >> 4) & 15u) | (base_letter(x & 15u) << 8) | (base_letter((x >> 12) & 15u) << 16) |
                             (base_letter((x >> 8) & 15u) << 24);
                w8[2u * d + 1u] = base_letter((x >> 20) & 15u) | (base_letter((x >> 16) & 15u) << 8) | (base_letter(x >> 28) << 16) |
                                  (base_letter((x >> 24) & 15u) << 24);
            }
            v4u32_any a, b;
            a[0] = w8[0]; a[1] = w8[1]; a[2] = w8[2]; a[3] = w8[3];
            b[0] = w8[4]; b[1] = w8[5]; b[2] = w8[6]; b[3] = w8[7];
            wide(a);
            wide(b);
        };
        for (; j + 128u <= len && wide_min == 16u; j += 128u) {                  // (64 bytes fetched at once: see qual())
            const uint32_t at = p + (j >> 1);
            const v4u32_any i0 = rec.u128(at), i1 = rec.u128(at + 16u), i2 = rec.u128(at + 32u), i3 = rec.u128(at + 48u);
            letters32(i0); letters32(i1); letters32(i2); letters32(i3);
        }
        for (; j + 32u <= len && wide_min == 16u; j += 32u) letters32(rec.u128(p + (j >> 1)));
        for (; j + 4u <= len; j += 4u) {
            const uint32_t two = rec.u16(p + (j >> 1));
            uint32_t w = 0;
            for (uint32_t q = 0; q < 4u; ++q) w |= base_letter((two >> (8u * (q >> 1) + ((q & 1u) ? 0u : 4u))) & 15u) << (8u * q);
            word(w, 4u);
        }
        for (; j < len; ++j) word(base_letter((rec.u8(p + (j >> 1)) >> ((j & 1u) ? 0u : 4u)) & 15u), 1u);
    }
    template <typename R> __device__ __forceinline__ void qual(const R &rec, uint32_t p, uint32_t len)
    {
        uint32_t j = 0;
        auto plus33 = [&](v4u32_any q) {
#pragma unroll
            for (uint32_t d = 0; d < 4u; ++d) q[d] = ((q[d] & 0x7F7F7F7Fu) + 0x21212121u) ^ (q[d] & 0x80808080u);
            wide(q);
        };
        // 64 bytes fetched at once where the field has them: a lane is on cache lines of its own (the records of 64 lanes lie ~330
        // bytes apart), and with 16 bytes per trip a line had left the L2 before its next quarter was asked for -- every byte of a
        // record was fetched from HBM more than once (profiles/r06_bam_pmc.txt)
        for (; j + 64u <= len && wide_min == 16u; j += 64u) {
            const v4u32_any a = rec.u128(p + j), b = rec.u128(p + j + 16u), c = rec.u128(p + j + 32u), d = rec.u128(p + j + 48u);
            plus33(a); plus33(b); plus33(c); plus33(d);
        }
        for (; j + 16u <= len && wide_min == 16u; j += 16u) plus33(rec.u128(p + j));
        for (; j + 4u <= len; j += 4u) {
            const uint32_t q4 = rec.u32(p + j);
            word(((q4 & 0x7F7F7F7Fu) + 0x21212121u) ^ (q4 & 0x80808080u), 4u);          // + 33 in every byte, no carry across bytes
        }
        for (; j < len; ++j) word((rec.u8(p + j) + 33u) & 0xFFu, 1u);
    }
    __device__ __forceinline__ void finish() { for (uint32_t j = 0; j < k; ++j) o[j] = (uint8_t)(acc >> (8u * j)); }
};

template <typename S> __device__ __forceinline__ void put_u64(S &s, unsigned long long v)
{
    uint8_t d[20];
    uint32_t k = 0;
    do { d[k++] = (uint8_t)('0' + (uint32_t)(v % 10ull)); v /= 10ull; } while (v);
    while (k) s.ch(d[--k]);
}
template <typename S> __device__ __forceinline__ void put_u32(S &s, uint32_t v)
{
    uint8_t d[10];
    uint32_t k = 0;
    do { d[k++] = (uint8_t)('0' + v % 10u); v /= 10u; } while (v);
    while (k) s.ch(d[--k]);
}
template <typename S> __device__ __forceinline__ void put_i64(S &s, long long v)
{
    if (v < 0) { s.ch('-'); put_u64(s, (unsigned long long)(-(v + 1)) + 1ull); } else put_u64(s, (unsigned long long)v);
}
template <typename S> __device__ __forceinline__ void put_i32(S &s, int32_t v)
{
    if (v < 0) { s.ch('-'); put_u32(s, (uint32_t)(-(v + 1)) + 1u); } else put_u32(s, (uint32_t)v);
}

// one value of type t at position p of the record (parse_record has walked the fields): false = a type this printer leaves to the host
template <typename R, typename S> __device__ __forceinline__ bool put_scalar(S &s, const R &rec, uint32_t &p, uint32_t t)
{
    switch (t) {
    case 'A': s.ch(rec.u8(p)); p += 1u; return true;
    case 'c': put_i32(s, (int8_t)rec.u8(p)); p += 1u; return true;
    case 'C': put_u32(s, rec.u8(p)); p += 1u; return true;
    case 's': put_i32(s, (int16_t)rec.u16(p)); p += 2u; return true;
    case 'S': put_u32(s, rec.u16(p)); p += 2u; return true;
    case 'i': put_i32(s, (int32_t)rec.u32(p)); p += 4u; return true;
    case 'I': put_u32(s, rec.u32(p)); p += 4u; return true;
    case 'f': {                                  // printf("%g") of the value, exact (xm_fmtg.h)
        const xmfmt::Text16 t16 = xmfmt::fmt_g_f32(rec.u32(p));
        for (uint32_t k = 0; k < t16.n; ++k) s.ch(t16.at(k));
        p += 4u;
        return true;
    }
    default: return false;                       // d (binary64; not a type of the BAM specification, htslib accepts it): the host
    }
}

// bytes in front of the first NUL at or behind position p, `limit` at most: sixteen bytes per access (the exact SWAR zero-byte test:
// borrows only run upwards, so the LOWEST flagged byte of a word is a true zero)
template <typename R> __device__ __forceinline__ uint32_t span_to_nul(const R &rec, uint32_t p, uint32_t limit)
{
    for (uint32_t at = 0; at < limit; at += 16u) {
        const v4u32_any v = rec.u128(p + at);
#pragma unroll
        for (uint32_t d = 0; d < 4u; ++d) {
            const uint32_t z = (v[d] - 0x01010101u) & ~v[d] & 0x80808080u;
            if (z != 0u) {
                const uint32_t found = at + 4u * d + ((uint32_t)__builtin_ctz(z) >> 3);
                return found < limit ? found : limit;
            }
        }
    }
    return limit;
}

// the record of `size` bytes behind its block_size word -> its SAM line with the '\n'; false: leave the record to the host printer
template <typename R, typename S> __device__ bool sam_line(const R &rec, uint32_t size, const RefTable refs, S &s)
{
    // the 32 fixed bytes in two accesses (refID, pos, l_read_name, mapq, bin, n_cigar_op, flag | l_seq, next_refID, next_pos, tlen)
    const v4u32_any c0 = rec.u128(0), c1 = rec.u128(16);
    const int32_t ref_id = (int32_t)c0[0], pos = (int32_t)c0[1];
    const uint32_t l_read_name = c0[2] & 0xFFu, mapq = (c0[2] >> 8) & 0xFFu, n_cigar = c0[3] & 0xFFFFu, flag = c0[3] >> 16, l_seq = c1[0];
    const int32_t next_ref = (int32_t)c1[1], next_pos = (int32_t)c1[2], tlen = (int32_t)c1[3];
    uint32_t p = 32u;
    const uint32_t nl = span_to_nul(rec, p, l_read_name);
    s.bytes(rec, p, nl);
    p += l_read_name;
    s.ch('\t'); put_u32(s, flag);
    s.ch('\t');
    if (ref_id < 0 || (uint32_t)ref_id >= refs.n) s.ch('*');
    else s.gbytes(refs.names + refs.at[ref_id], refs.at[ref_id + 1] - refs.at[ref_id]);
    s.ch('\t'); put_i64(s, (long long)pos + 1);
    s.ch('\t'); put_u32(s, mapq);
    s.ch('\t');
    if (n_cigar == 0u) s.ch('*');
    for (uint32_t k = 0; k < n_cigar; ++k) {
        const uint32_t v = rec.u32(p + 4u * k);
        put_u32(s, v >> 4);
        const unsigned long long lo = 0x3D5048534E44494Dull, hi = 0x3F3F3F3F3F3F4258ull;               // "MIDNSHP=" "XB??????"
        s.ch((uint32_t)((((v & 8u) ? hi : lo) >> (8u * (v & 7u))) & 0xFFu));
    }
    p += 4u * n_cigar;
    s.ch('\t');
    if (next_ref < 0 || (uint32_t)next_ref >= refs.n) s.ch('*');
    else if (next_ref == ref_id) s.ch('=');
    else s.gbytes(refs.names + refs.at[next_ref], refs.at[next_ref + 1] - refs.at[next_ref]);
    s.ch('\t'); put_i64(s, (long long)next_pos + 1);
    s.ch('\t'); put_i32(s, tlen);
    s.ch('\t');
    if (l_seq == 0u) s.ch('*');
    s.seq(rec, p, l_seq);
    p += (l_seq + 1u) / 2u;
    s.ch('\t');
    if (l_seq == 0u || rec.u8(p) == 0xFFu) s.ch('*'); else s.qual(rec, p, l_seq);
    p += l_seq;
    while (p + 3u <= size) {
        const uint32_t head = rec.u32(p), type = (head >> 16) & 0xFFu;         // tag, tag, type (and a byte of the value) in one access
        s.ch('\t'); s.ch(head & 0xFFu); s.ch((head >> 8) & 0xFFu); s.ch(':');
        p += 3u;
        if (type == 'A') { s.ch('A'); s.ch(':'); put_scalar(s, rec, p, 'A'); }
        else if (type == 'c' || type == 'C' || type == 's' || type == 'S' || type == 'i' || type == 'I') { s.ch('i'); s.ch(':'); put_scalar(s, rec, p, type); }
        else if (type == 'f') { s.ch('f'); s.ch(':'); put_scalar(s, rec, p, 'f'); }
        else if (type == 'Z' || type == 'H') {
            s.ch(type); s.ch(':');
            const uint32_t l = span_to_nul(rec, p, size > p ? size - p : 0u);
            s.bytes(rec, p, l);
            p += l + 1u;
        } else if (type == 'B') {
            const uint32_t sub = rec.u8(p), cnt = rec.u32(p + 1);
            p += 5u;
            s.ch('B'); s.ch(':'); s.ch(sub);
            for (uint32_t k = 0; k < cnt; ++k) { s.ch(','); if (!put_scalar(s, rec, p, sub)) return false; }
        } else return false;                                                 // d: the host prints what printf("%g") would
    }
    s.ch('\n');
    return true;
}

// the record whose block_size word is at raw + off, read where it lies in the window
template <typename S> __device__ __forceinline__ bool sam_line_at(const uint8_t *__restrict__ raw, uint32_t off, const RefTable refs, S &s)
{
    const GlobalRec rec = {raw + off + 4u};
    return sam_line(rec, ld32(raw + off), refs, s);
}

// T1: wsize[i] (want_kernel: the record's bytes when a sink takes it, else 0) becomes the length of its line; state[13] != 0: a
// record for the host printer
__global__ void __launch_bounds__(256)
text_size_kernel(const uint8_t *__restrict__ raw, const uint32_t *__restrict__ rec_off, uint32_t n, const RefTable refs,
                 uint32_t *__restrict__ wsize, uint32_t *__restrict__ state)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || wsize[i] == 0u) return;
    CountChars c;
    if (!sam_line_at(raw, rec_off[i], refs, c)) atomicOr(&state[13], 1u);
    wsize[i] = c.n;
}

// T2: the lines themselves; place[i] / wsize[i] become the line table the writer gathers from (position; length without '\n')
__global__ void __launch_bounds__(256)
text_fill_kernel(const uint8_t *__restrict__ raw, const uint32_t *__restrict__ rec_off, uint32_t n, const RefTable refs,
                 uint32_t *__restrict__ wsize, uint32_t *__restrict__ place, uint8_t *__restrict__ text, uint32_t text_cap)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t len = wsize[i], at = place[i];
    if (len == 0u || at == NO_RECORD || at + len > text_cap) { wsize[i] = 0u; place[i] = 0u; return; }
    WriteChars w;
    w.o = text + at;
    (void)sam_line_at(raw, rec_off[i], refs, w);
    w.finish();
    wsize[i] = len - 1u;
}

// ---- G: the six outputs themselves, gathered ON the device (xm_bamdev_fetch_bins; SURVEY f-2's gather kernel) ----
// With the bins' unit lists (d_idx: unit indices sorted by bin, input order inside a bin; d_off_counts[0..7]: where each bin begins)
// and the wanted records' line lengths (T1) on the device, the text of every output file is a pure function of device data:
// bin b's text = for its units in order, the unit's lines (xenomapper.py:332-350, :423-448, :521-550: primary bins print file 1's
// line(s), secondary bins file 2's, `unresolved` file 1's then file 2's; a paired unit is records i - 1 and i).  G1 sizes every
// unit, the size scan places it -- the units are in bin order, so the scan IS the layout of the six texts back to back --, G2
// prints every line where it belongs (the same sam_line as T2), G3 moves the stream to the host's page-locked buffer in aligned
// 16-byte pieces.  The host then writes six contiguous byte ranges per window instead of gathering a million lines.
__global__ void __launch_bounds__(256)
unit_size_kernel(const uint32_t *__restrict__ idx, const unsigned long long *__restrict__ off, uint32_t n_units, uint32_t n_records, int paired,
                 uint32_t sink_mask, const uint32_t *__restrict__ ws1, const uint32_t *__restrict__ ws2, uint32_t *__restrict__ usize,
                 unsigned long long *__restrict__ total64)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    uint32_t s = 0;
    if (p < n_units) {
        const uint32_t i = idx[p], files = files_of_bin(bin_of_place(p, off), sink_mask);
        if (i < n_records && (!paired || i > 0u)) {
            if (files & 1u) s += ws1[i] + (paired ? ws1[i - 1u] : 0u);
            if (files & 2u) s += ws2[i] + (paired ? ws2[i - 1u] : 0u);
        }
        usize[p] = s;
    }
    // the places are 32-bit; whether all the text fits them (and the buffers) is decided on a 64-bit total
    unsigned long long t = s;
    for (int d = 32; d; d >>= 1) t += __shfl_xor(t, d, 64);
    if ((threadIdx.x & 63u) == 0u && t) atomicAdd(total64, t);
}

// G2: a lane per (unit, line of the unit): paired units have two lines per file, single-end units one.
// (Round 6 also built a form that stages a wave's 64 records and lines in LDS -- coalesced 16-byte copies in and out, the printing
// LDS to LDS a byte at a time: byte-exact, and slower: ~8 ms a window against 6.1, the byte-wise LDS round trips are a dependent
// chain of ~800 steps per lane, and its 53 KB of LDS per wave cannot start beside the inflate chains, which hold 158 of a CU's
// 160 KB.  profiles/r06_ab_fill_stage.txt.)
__global__ void __launch_bounds__(256)
line_fill_kernel(const uint8_t *__restrict__ raw1, const uint8_t *__restrict__ raw2, const uint32_t *__restrict__ rec_off1,
                 const uint32_t *__restrict__ rec_off2, const RefTable refs1, const RefTable refs2,
                 const uint32_t *__restrict__ idx, const unsigned long long *__restrict__ off, uint32_t n_units, uint32_t n_records, int paired,
                 uint32_t sink_mask, const uint32_t *__restrict__ ws1, const uint32_t *__restrict__ ws2,
                 const uint32_t *__restrict__ usize, const uint32_t *__restrict__ uplace, uint8_t *__restrict__ out, uint32_t out_cap,
                 uint32_t wide_min)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t p = paired ? g >> 1 : g, j = paired ? g & 1u : 0u;
    if (p >= n_units) return;
    const uint32_t size = usize[p];
    if (size == 0u) return;
    const uint32_t i = idx[p], files = files_of_bin(bin_of_place(p, off), sink_mask);
    if (i >= n_records || (paired && i == 0u)) return;
    const uint32_t r = paired ? i - 1u + j : i;
    uint32_t at = uplace[p];
    if (at > out_cap || size > out_cap - at) return;                              // (the total was checked before the launch)
    for (uint32_t f = 0; f < 2u; ++f) {
        if (((files >> f) & 1u) == 0u) continue;
        const uint32_t *ws = f ? ws2 : ws1;
        const uint32_t first = paired ? ws[i - 1u] : 0u, mine = ws[r];
        if (mine != 0u) {
            WriteChars w;
            w.o = out + at + (j ? first : 0u);
            w.wide_min = wide_min;
            (void)sam_line_at(f ? raw2 : raw1, (f ? rec_off2 : rec_off1)[r], f ? refs2 : refs1, w);
            w.finish();
        }
        at += first + ws[i];                                                       // behind file 1's lines of the unit: file 2's
    }
}

// ---- --cigar_scores: the records' CIGAR words as the packed CIGAR columns K1p reads (include/xenomapper_hip.h) ------------------
// BAM holds the operations as the kernel wants them (len << 4 | op); what is left to do is what xm_cigar_pack does on the host:
// a count byte per record (255 = "255 or more": a trailer word n_ops << 4 | 15 behind the operations), the operations back to
// back, and where every 256th record's operations begin.  C1 sizes, the size scan of W2, C2 copies.
__global__ void __launch_bounds__(256)
cig_size_kernel(const uint32_t *__restrict__ n_cigar, uint32_t n, uint8_t *__restrict__ cnt8, uint32_t *__restrict__ words)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = n_cigar[i];
    cnt8[i] = (uint8_t)(c < 255u ? c : 255u);
    words[i] = c + (c >= 255u ? 1u : 0u);
}

__global__ void __launch_bounds__(256)
cig_fill_kernel(const uint8_t *__restrict__ raw, const uint32_t *__restrict__ n_cigar, const uint32_t *__restrict__ cig_at,
                const uint32_t *__restrict__ place, const uint32_t *__restrict__ total, uint32_t n, uint32_t *__restrict__ tile,
                uint32_t *__restrict__ ops, uint32_t ops_cap)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0u) tile[(n + 255u) / 256u] = *total;
    if (i >= n) return;
    const uint32_t c = n_cigar[i], pos = place[i];
    if ((i & 255u) == 0u) tile[i >> 8] = pos;
    if (pos + c + (c >= 255u ? 1u : 0u) > ops_cap) return;                 // cannot happen: the words are a part of the window
    const uint8_t *src = raw + cig_at[i];
    for (uint32_t k = 0; k < c; ++k) ops[pos + k] = ld32(src + 4u * k);
    if (c >= 255u) ops[pos + c] = (c << 4) | 15u;
}

// ---- skip_repeated_reads (getReadPairs, xenomapper.py:110-117): a file is cut into runs of adjacent records with one name; pair k
// is the first record of run k of both files.  R1 marks the run starts, the size scan of W2 ranks them, R2 moves the fields of
// the run starts next to each other -- behind it the pair kernel and everything after it see a file of run starts only.
struct RecCols {
    uint32_t *rec_off, *name_off, *name_len;
    int32_t *a, *x;
    uint8_t *flag;
    uint32_t *n_cigar, *cig_at;                 // --cigar_scores runs only (else null)
};

__global__ void __launch_bounds__(256)
run_start_kernel(const uint8_t *__restrict__ raw, const uint32_t *__restrict__ name_off, const uint32_t *__restrict__ name_len,
                 const uint8_t *__restrict__ flag, uint32_t n, uint32_t *__restrict__ is_start, uint32_t *__restrict__ state)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    // a record the text rules might read differently may cut the runs differently, yielded or not (state[2] as the pair kernel's)
    const uint32_t g = flag[i], wb = ((g & R_WEIRD) ? 1u : 0u) | ((g & R_BAD) ? 2u : 0u);
    if (wb) atomicOr(&state[2], wb);
    bool st = i == 0u;
    if (!st) {
        const uint32_t l = name_len[i];
        st = l != name_len[i - 1u];
        if (!st) {
            const uint8_t *p = raw + name_off[i], *q = raw + name_off[i - 1u];
            uint32_t diff = 0;
            for (uint32_t k = 0; k < l; ++k) diff |= (uint32_t)(p[k] ^ q[k]);
            st = diff != 0u;
        }
    }
    is_start[i] = st ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
run_select_kernel(RecCols from, const uint32_t *__restrict__ place, uint32_t n, RecCols to)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = place[i];
    if (j == 0xFFFFFFFFu) return;
    to.rec_off[j] = from.rec_off[i];
    to.name_off[j] = from.name_off[i];
    to.name_len[j] = from.name_len[i];
    to.a[j] = from.a[i];
    to.x[j] = from.x[i];
    to.flag[j] = from.flag[i];
    if (from.n_cigar != nullptr) { to.n_cigar[j] = from.n_cigar[i]; to.cig_at[j] = from.cig_at[i]; }
}

// ---- B5: one lane per pair --------------------------------------------------------------------------------------------
struct FileRecs {
    const uint8_t *raw;
    const uint32_t *name_off, *name_len;
    const int32_t *a, *x;
    const uint8_t *flag;
};

// n bytes at p and at q equal?  Sixteen bytes per access (unaligned dwordx4; the last access may reach up to 15 bytes behind the
// names -- still inside the window, whose buffer ends 64 bytes behind its last record): a byte at a time, every byte of the three
// names a lane compares was a sector of its own to fetch, 4.3 GB per window (PMC, profiles/r06_bam_pmc.txt).
__device__ __forceinline__ bool same_name(const uint8_t *p, const uint8_t *q, uint32_t n)
{
    uint32_t diff = 0;
    for (uint32_t k = 0; k < n; k += 16u) {
        const v4u32_any a = *reinterpret_cast<const v4u32_any *>(p + k), b = *reinterpret_cast<const v4u32_any *>(q + k);
        const uint32_t rem = n - k;                                          // bytes of this piece that belong to the names (>= 1)
#pragma unroll
        for (uint32_t d = 0; d < 4u; ++d) {
            const uint32_t valid = rem > 4u * d ? (rem - 4u * d < 4u ? rem - 4u * d : 4u) : 0u;
            const uint32_t mask = valid >= 4u ? 0xFFFFFFFFu : ((1u << (8u * valid)) - 1u);
            diff |= (a[d] ^ b[d]) & mask;
        }
    }
    return diff == 0u;
}

// state: [0] first mismatch (atomicMin), [1] pairs with an exception, [2] weird | bad << 1, [3] first open run (atomicMin);
// runs (the skipping walk; else null): runs[f] = run starts of file f (the most it can pair), open bit f = the file goes on behind
// its window, so its last run may go on too and must not be yielded yet
__global__ void __launch_bounds__(256)
pair_kernel(FileRecs f1, FileRecs f2, uint32_t n_launch, int paired, const uint32_t *__restrict__ runs, uint32_t open,
            int32_t *__restrict__ as1, int32_t *__restrict__ xs1,
            int32_t *__restrict__ as2, int32_t *__restrict__ xs2, unsigned long long *__restrict__ unit_bits,
            uint8_t *__restrict__ lflag1, uint8_t *__restrict__ lflag2, uint32_t *__restrict__ state)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    uint32_t n = n_launch;
    if (runs != nullptr) {
        const uint32_t r1 = runs[0], r2 = runs[1];
        n = min(n, min(r1, r2));
        if (k < n && (((open & 1u) && k + 1u == r1) || ((open & 2u) && k + 1u == r2))) atomicMin(&state[3], k);
    }
    bool unit = false;
    if (k < n) {
        const uint32_t g1 = f1.flag[k], g2 = f2.flag[k];
        as1[k] = f1.a[k]; xs1[k] = f1.x[k];
        as2[k] = f2.a[k]; xs2[k] = f2.x[k];
        const uint32_t n1 = f1.name_len[k], n2 = f2.name_len[k];
        const uint8_t *p1 = f1.raw + f1.name_off[k], *p2 = f2.raw + f2.name_off[k];
        if (n1 != n2 || !same_name(p1, p2, n1)) atomicMin(&state[0], k);
        if (paired) {
            if (k > 0u) {
                const uint32_t n0 = f1.name_len[k - 1u];
                unit = n0 == n1 && same_name(f1.raw + f1.name_off[k - 1u], p1, n1);
            }
        } else {
            unit = true;
        }
        // per-pair line flags in the layout of xenomapper_strip.h: NORMAL | ex_a << 2 | ex_x << 5
        const uint32_t e1a = (g1 >> R_EX_A_SHIFT) & 3u, e1x = (g1 >> R_EX_X_SHIFT) & 3u;
        const uint32_t e2a = (g2 >> R_EX_A_SHIFT) & 3u, e2x = (g2 >> R_EX_X_SHIFT) & 3u;
        lflag1[k] = (uint8_t)(XMS_LINE_NORMAL | (e1a << XMS_LINE_EX_A_SHIFT) | (e1x << XMS_LINE_EX_X_SHIFT));
        lflag2[k] = (uint8_t)(XMS_LINE_NORMAL | (e2a << XMS_LINE_EX_A_SHIFT) | (e2x << XMS_LINE_EX_X_SHIFT));
        if (e1a | e1x | e2a | e2x) atomicAdd(&state[1], 1u);
        const uint32_t wb = (((g1 | g2) & R_WEIRD) ? 1u : 0u) | (((g1 | g2) & R_BAD) ? 2u : 0u);
        if (wb) atomicOr(&state[2], wb);
    }
    const unsigned long long m = __ballot(unit);
    if ((threadIdx.x & 63u) == 0u && (k & ~63u) < ((n_launch + 63u) & ~63u)) unit_bits[k >> 6] = m;
}

// ---------------------------------------------------------------------------------------------------------------------
struct PerFile {
    uint8_t *h_comp = nullptr, *d_comp = nullptr;           // staged compressed bytes (+ XMB_COMP_PAD): d_comp = a half of Slot::d_comp_all
    uint8_t *d_raw = nullptr, *h_raw = nullptr;             // the inflated window: d_raw = a half of Slot::d_raw_all
    uint32_t *h_seg = nullptr, *d_seg = nullptr, *d_cnt = nullptr, *d_exit = nullptr, *d_base = nullptr;
    // what the inflate launch notes per block (slot_of): record starts and the stripper's fields of every record
    // the records a sink takes, packed (xm_bamdev_fetch_wanted): bytes, and per record where it went (NO_RECORD: not taken)
    uint8_t *d_packed = nullptr, *h_packed = nullptr;
    uint32_t *d_wsize = nullptr, *d_place = nullptr, *h_place = nullptr, *d_part = nullptr;
    uint32_t *d_s_off = nullptr, *d_s_name_off = nullptr, *d_s_name_len = nullptr, *d_s_ncig = nullptr, *d_s_cig_at = nullptr;
    // --cigar_scores: per record how many CIGAR words and where, and the packed CIGAR columns made of them (ensure_cigar)
    uint32_t *d_ncig = nullptr, *d_cig_at = nullptr, *d_cig_tile = nullptr, *d_cig_ops = nullptr;
    uint8_t *d_cig_cnt = nullptr;
    uint64_t ops_cap = 0, cig_records = 0, cig_slots = 0;
    int32_t *d_s_a = nullptr, *d_s_x = nullptr;
    uint8_t *d_s_flag = nullptr;
    uint64_t slots_len = 0;
    uint32_t *d_rec_off = nullptr, *h_rec_off = nullptr;
    uint32_t *d_name_off = nullptr, *d_name_len = nullptr;
    int32_t *d_a = nullptr, *d_x = nullptr;
    uint8_t *d_rflag = nullptr, *d_lflag = nullptr, *h_lflag = nullptr;
    // the device printer: the file's reference names (xm_bamdev_set_refs: the same for both slots), the line lengths on the host
    uint8_t *d_ref_names = nullptr;
    uint32_t *d_ref_at = nullptr, n_refs = 0, *h_llen = nullptr;
    bool refs_set = false;
    uint64_t llen_records = 0;
    // the skipping walk: the fields of the run starts, next to each other (ensure_skip), and their record starts on the host
    uint32_t *d_r_rec_off = nullptr, *d_r_name_off = nullptr, *d_r_name_len = nullptr, *d_r_ncig = nullptr, *d_r_cig_at = nullptr, *h_pair_off = nullptr;
    int32_t *d_r_a = nullptr, *d_r_x = nullptr;
    uint8_t *d_r_flag = nullptr;
    uint64_t skip_records = 0, skip_cig_records = 0;
    // what the pair kernel and everything behind it read: the records of the window, or its run starts
    const uint32_t *v_rec_off = nullptr, *v_ncig = nullptr, *v_cig_at = nullptr;
    uint32_t *d_summary = nullptr, *h_summary = nullptr;
    uint64_t raw_len = 0;
    uint64_t table_len = 0;                 // entries of h_rec_off that are valid (the records of the slot's last window)
};

struct Slot {
    uint64_t comp_cap = 0, raw_cap = 0, block_cap = 0, record_cap = 0;
    PerFile pf[2];
    // both files' blocks are inflated by ONE launch (a window of one file is only a third of the blocks the chip can hold):
    // one compressed buffer and one output buffer, file 1's half behind file 0's; one block table, status and CRC array
    uint8_t *d_comp_all = nullptr, *d_raw_all = nullptr;
    uint64_t comp_stride = 0, raw_stride = 0;
    // the packed-record / text buffers of the two files are halves of ONE allocation each (pf[f].d_packed, pf[f].h_packed):
    // xm_bamdev_fetch_bins prints the six outputs into them as one stream
    uint8_t *d_packed_all = nullptr, *h_packed_all = nullptr;
    uint64_t packed_stride = 0;
    uint32_t *d_usize = nullptr, *d_uplace = nullptr, *d_upart = nullptr;      // per unit: bytes of its lines, where they go (fetch_bins)
    xm_bgzf_block *h_blocks = nullptr, *d_blocks = nullptr;
    xm_bgzf_walk *h_walk = nullptr, *d_walk = nullptr;      // per block: where its record chain starts and where its results go
    uint32_t *d_status = nullptr, *h_status = nullptr, *d_crc = nullptr, *h_crc = nullptr, *d_work = nullptr;
    int32_t *d_col[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t *d_bits = nullptr;
    uint32_t *d_state = nullptr, *h_state = nullptr;
    uint8_t *d_code = nullptr, *d_bins4 = nullptr, *h_code = nullptr;
    uint32_t *d_idx = nullptr, *h_idx = nullptr;
    uint64_t *d_off_counts = nullptr, *h_off_counts = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_wait = nullptr;            // blocking-sync event: the waiting thread sleeps instead of spinning on a core the
                                             // writer's printing threads need (16 of 16 CPUs print while the next window inflates)
    // the inflated window goes back to the host (the writer prints the records' text from it) on a stream of its own, behind
    // the inflate launch only: the record kernels, the fused pass and the NEXT window's upload and inflate (other slot) run
    // beside it.  ev_raw marks its end; whoever reads h_raw waits for it (xm_bamdev_raw_wait; a carried tail: the next run).
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_inflated = nullptr, ev_raw = nullptr;
    // compressed bytes sent ahead (xm_bamdev_upload, from the thread that reads the next window while this slot is idle)
    hipStream_t up_stream = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr};
    uint64_t up_len[2] = {0, 0};
    int last_score_mode = XMS_SCORE_AS_XS;   // of the last run: which columns xm_bamdev_classify reads
    bool raw_issued = false;                 // ev_raw has been recorded at least once (stays true: waiting for a past event costs nothing)
    bool fill_issued = false;                // ev_inflated was recorded behind a G2 launch (xm_bamdev_fetch_bins) at least once
    bool have_columns = false;
    bool classified = false;                 // the fused pass has run on the slot's columns (its compact category stream is in d_bins4)
};

}  // namespace

struct xm_bamdev {
    xm_ctx *ctx = nullptr;
    int device = 0;
    Slot slot[2];
    std::mutex error_lock;
    std::string last_error;
};

namespace {

int fail(xm_bamdev *b, hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    if (b) {
        std::lock_guard<std::mutex> hold(b->error_lock);
        b->last_error = buf;
    }
    (void)hipGetLastError();        // reported here: a later launch check on this thread must not find it again
    return e == hipErrorOutOfMemory ? XM_ERR_OOM : XM_ERR_HIP;
}

#define XMB_HIP(b, call)                                   \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return fail((b), e_, #call); \
    } while (0)
#define XMB_TRY(expr)                  \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != XM_OK) return rc_;  \
    } while (0)

template <typename T> void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }
template <typename T> void hfree(T *&p) { if (p) { (void)xmpin::host_free(p); p = nullptr; } }
template <typename T> int dalloc(xm_bamdev *b, T *&p, size_t count)
{
    dfree(p);
    XMB_HIP(b, hipMalloc((void **)&p, std::max<size_t>(count, 16) * sizeof(T)));
    return XM_OK;
}
template <typename T> int halloc(xm_bamdev *b, T *&p, size_t count)
{
    hfree(p);
    XMB_HIP(b, xmpin::host_malloc((void **)&p, std::max<size_t>(count, 16) * sizeof(T)));
    return XM_OK;
}

void free_slot(Slot &sl)
{
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        hfree(q.h_comp); q.d_comp = nullptr; q.d_raw = nullptr; hfree(q.h_raw);
        q.d_packed = nullptr; q.h_packed = nullptr; dfree(q.d_wsize); dfree(q.d_place); hfree(q.h_place); dfree(q.d_part);
        hfree(q.h_seg); dfree(q.d_seg); dfree(q.d_cnt); dfree(q.d_exit); dfree(q.d_base);
        dfree(q.d_s_off); dfree(q.d_s_name_off); dfree(q.d_s_name_len); dfree(q.d_s_a); dfree(q.d_s_x); dfree(q.d_s_flag);
        dfree(q.d_s_ncig); dfree(q.d_s_cig_at); dfree(q.d_ncig); dfree(q.d_cig_at); dfree(q.d_cig_tile); dfree(q.d_cig_ops); dfree(q.d_cig_cnt);
        q.slots_len = 0; q.ops_cap = q.cig_records = q.cig_slots = 0;
        dfree(q.d_rec_off); hfree(q.h_rec_off); dfree(q.d_name_off); dfree(q.d_name_len); dfree(q.d_a); dfree(q.d_x);
        dfree(q.d_rflag); dfree(q.d_lflag); hfree(q.h_lflag);
        dfree(q.d_r_rec_off); dfree(q.d_r_name_off); dfree(q.d_r_name_len); dfree(q.d_r_ncig); dfree(q.d_r_cig_at); hfree(q.h_pair_off);
        dfree(q.d_r_a); dfree(q.d_r_x); dfree(q.d_r_flag);
        dfree(q.d_ref_names); dfree(q.d_ref_at); hfree(q.h_llen);
        q.refs_set = false;
        q.skip_records = q.skip_cig_records = q.llen_records = 0; q.n_refs = 0;
    }
    dfree(sl.d_comp_all); dfree(sl.d_raw_all); dfree(sl.d_packed_all); hfree(sl.h_packed_all);
    dfree(sl.d_usize); dfree(sl.d_uplace); dfree(sl.d_upart);
    hfree(sl.h_blocks); dfree(sl.d_blocks); hfree(sl.h_walk); dfree(sl.d_walk); dfree(sl.d_status); hfree(sl.h_status); dfree(sl.d_crc); hfree(sl.h_crc);
    for (int c = 0; c < 4; ++c) dfree(sl.d_col[c]);
    dfree(sl.d_bits); dfree(sl.d_code); dfree(sl.d_bins4); dfree(sl.d_idx);
    hfree(sl.h_code); hfree(sl.h_idx);
    sl.comp_cap = sl.raw_cap = sl.block_cap = sl.record_cap = 0;
}

}  // namespace

// --cigar_scores: the arrays only such a run needs, made when the first one comes (and again when reserve() has grown the rest)
static int ensure_cigar(xm_bamdev *b, Slot &sl)
{
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        if (q.cig_slots < q.slots_len) {
            q.cig_slots = 0;
            XMB_TRY(dalloc(b, q.d_s_ncig, (size_t)q.slots_len)); XMB_TRY(dalloc(b, q.d_s_cig_at, (size_t)q.slots_len));
            q.cig_slots = q.slots_len;
        }
        if (q.cig_records < sl.record_cap) {
            q.cig_records = 0;
            const size_t n = (size_t)sl.record_cap + 64;
            XMB_TRY(dalloc(b, q.d_ncig, n)); XMB_TRY(dalloc(b, q.d_cig_at, n)); XMB_TRY(dalloc(b, q.d_cig_cnt, n));
            XMB_TRY(dalloc(b, q.d_cig_tile, n / 256 + 8));
            q.cig_records = sl.record_cap;
        }
        // the operations are words of the window; a record with 255 or more of them has one trailer word behind them
        const uint64_t need_ops = sl.raw_cap / 4u + sl.raw_cap / 1020u + 64u;
        if (q.ops_cap < need_ops) {
            q.ops_cap = 0;
            XMB_TRY(dalloc(b, q.d_cig_ops, (size_t)need_ops));
            q.ops_cap = need_ops;
        }
    }
    return XM_OK;
}

// The inflate launch reads the compressed blocks where the host staged them -- page-locked memory the device has mapped -- instead of
// a copy in HBM: every chain keeps one 128-byte piece of its block in flight, 8192 chains hide the link's latency, and the copy
// that is not made is a shader kernel that does not run beside the inflate launch (host <-> device copies are blit kernels on this
// pool): 20.4-20.7 against 18.5-19.8 M pairs/s end to end (profiles/r05_bam_device_text.txt).  XM_BAMDEV_ZEROCOPY=0: upload first.
static bool zero_copy_input()
{
    static const bool on = [] { const char *v = getenv("XM_BAMDEV_ZEROCOPY"); return !(v && v[0] == '0'); }();
    return on;
}

// skip_repeated_reads: the arrays only the skipping walk needs
static int ensure_skip(xm_bamdev *b, Slot &sl, bool cigar)
{
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        const size_t n = (size_t)sl.record_cap + 64;
        if (q.skip_records < sl.record_cap) {
            q.skip_records = 0;
            XMB_TRY(dalloc(b, q.d_r_rec_off, n)); XMB_TRY(dalloc(b, q.d_r_name_off, n)); XMB_TRY(dalloc(b, q.d_r_name_len, n));
            XMB_TRY(dalloc(b, q.d_r_a, n)); XMB_TRY(dalloc(b, q.d_r_x, n)); XMB_TRY(dalloc(b, q.d_r_flag, n));
            XMB_TRY(halloc(b, q.h_pair_off, n));
            q.skip_records = sl.record_cap;
        }
        if (cigar && q.skip_cig_records < sl.record_cap) {
            q.skip_cig_records = 0;
            XMB_TRY(dalloc(b, q.d_r_ncig, n)); XMB_TRY(dalloc(b, q.d_r_cig_at, n));
            q.skip_cig_records = sl.record_cap;
        }
    }
    return XM_OK;
}

// the packed CIGAR columns of one file's first n records, queued on the slot's stream (C1, the size scan, C2)
static int pack_cigar(Slot &sl, int f, uint32_t n)
{
    PerFile &q = sl.pf[f];
    if (q.cig_records < n || q.ops_cap == 0 || n == 0) return XM_ERR_INVALID_ARG;
    hipStream_t st = sl.stream;
    const uint32_t n_part = (n + SCAN_TILE - 1u) / SCAN_TILE;
    cig_size_kernel<<<(n + 255u) / 256u, 256, 0, st>>>(q.v_ncig, n, q.d_cig_cnt, q.d_wsize);
    size_sum_kernel<<<n_part, 256, 0, st>>>(q.d_wsize, n, q.d_part);
    part_scan_kernel<<<1, 1024, 0, st>>>(q.d_part, n_part, sl.d_state + 10 + f);
    size_place_kernel<true><<<n_part, 256, 0, st>>>(q.d_wsize, n, q.d_part, q.d_place);
    cig_fill_kernel<<<(n + 255u) / 256u, 256, 0, st>>>(q.d_raw, q.v_ncig, q.v_cig_at, q.d_place, sl.d_state + 10 + f, n, q.d_cig_tile, q.d_cig_ops,
                                                      (uint32_t)std::min<uint64_t>(q.ops_cap, 0xFFFFFFFFull));
    return XM_OK;
}

extern "C" {

int xm_bamdev_create(xm_ctx *ctx, int device_id, xm_bamdev **out)
{
    if (!ctx || !out) return XM_ERR_INVALID_ARG;
    xm_bamdev *b = new (std::nothrow) xm_bamdev();
    if (!b) return XM_ERR_OOM;
    b->ctx = ctx;
    b->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        Slot &sl = b->slot[k];
        e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking);
        for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreate(&sl.ev[i]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_wait, hipEventBlockingSync | hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_inflated, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_raw, hipEventBlockingSync | hipEventDisableTiming);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&sl.up_stream, hipStreamNonBlocking);
        for (int f = 0; f < 2 && e == hipSuccess; ++f) e = hipEventCreateWithFlags(&sl.ev_up[f], hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_state, 32 * sizeof(uint32_t));
        if (e == hipSuccess) e = xmpin::host_malloc((void **)&sl.h_state, 32 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_off_counts, 72 * sizeof(uint64_t));
        if (e == hipSuccess) e = xmpin::host_malloc((void **)&sl.h_off_counts, 72 * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_work, 16 * sizeof(uint32_t));
        for (int f = 0; f < 2 && e == hipSuccess; ++f) {
            PerFile &q = sl.pf[f];
            e = hipMalloc((void **)&q.d_summary, 16 * sizeof(uint32_t));
            if (e == hipSuccess) e = xmpin::host_malloc((void **)&q.h_summary, 16 * sizeof(uint32_t));
        }
    }
    if (e == hipSuccess) {
        // one copy stream for both slots, one that runs beside either slot's compute stream (xm_gather.h: tried, not assumed)
        const hipStream_t compute[2] = {b->slot[0].stream, b->slot[1].stream};
        e = create_copy_stream(&b->slot[0].copy_stream, compute, 2);
        b->slot[1].copy_stream = b->slot[0].copy_stream;
    }
    if (e != hipSuccess) {
        xm_bamdev_destroy(b);
        return e == hipErrorOutOfMemory ? XM_ERR_OOM : XM_ERR_HIP;
    }
    *out = b;
    return XM_OK;
}

int xm_bamdev_destroy(xm_bamdev *b)
{
    if (!b) return XM_ERR_INVALID_ARG;
    (void)hipSetDevice(b->device);
    for (int k = 0; k < 2; ++k) {
        Slot &sl = b->slot[k];
        if (sl.stream) {
            (void)hipStreamSynchronize(sl.stream);
            (void)xm_workspace_release(b->ctx, sl.stream);
        }
        if (sl.copy_stream) (void)hipStreamSynchronize(sl.copy_stream);
        if (sl.up_stream) (void)hipStreamSynchronize(sl.up_stream);
        free_slot(sl);
        dfree(sl.d_state); hfree(sl.h_state); dfree(sl.d_off_counts); hfree(sl.h_off_counts);
        dfree(sl.d_work);
        for (int f = 0; f < 2; ++f) { dfree(sl.pf[f].d_summary); hfree(sl.pf[f].h_summary); }
        for (int i = 0; i < 3; ++i)
            if (sl.ev[i]) (void)hipEventDestroy(sl.ev[i]);
        if (sl.ev_wait) (void)hipEventDestroy(sl.ev_wait);
        if (sl.ev_inflated) (void)hipEventDestroy(sl.ev_inflated);
        if (sl.ev_raw) (void)hipEventDestroy(sl.ev_raw);
        if (k == 1 && sl.copy_stream) (void)hipStreamDestroy(sl.copy_stream);       // shared by the slots: once, behind both
        for (int f = 0; f < 2; ++f)
            if (sl.ev_up[f]) (void)hipEventDestroy(sl.ev_up[f]);
        if (sl.up_stream) (void)hipStreamDestroy(sl.up_stream);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    delete b;
    return XM_OK;
}

int xm_bamdev_reserve(xm_bamdev *b, int slot, uint64_t comp_bytes, uint64_t raw_bytes, uint64_t max_blocks, uint64_t max_records)
{
    if (!b || slot < 0 || slot > 1 || raw_bytes >= 0xFFFF0000ull || comp_bytes >= 0xFFFF0000ull || max_blocks == 0 || max_records == 0 ||
        max_records >= 0xFFFFFF00ull || max_blocks >= 0x7FFFFF00ull)
        return XM_ERR_INVALID_ARG;
    XMB_HIP(b, hipSetDevice(b->device));
    Slot &sl = b->slot[slot];
    XMB_HIP(b, hipStreamSynchronize(sl.stream));
    if (sl.raw_issued) XMB_HIP(b, hipEventSynchronize(sl.ev_raw));            // THIS slot's last copy (the copy stream also carries the other slot's)
    XMB_HIP(b, hipStreamSynchronize(sl.up_stream));
    sl.have_columns = false;
    if (comp_bytes > sl.comp_cap) {
        sl.comp_cap = 0;
        sl.up_len[0] = sl.up_len[1] = 0;                                     // what was sent ahead went to the old buffer
        sl.comp_stride = (comp_bytes + XMB_COMP_PAD + 255u) & ~(uint64_t)255;
        XMB_TRY(dalloc(b, sl.d_comp_all, (size_t)(2 * sl.comp_stride)));
        XMB_HIP(b, hipMemset(sl.d_comp_all, 0, (size_t)(2 * sl.comp_stride)));
        for (int f = 0; f < 2; ++f) {
            XMB_TRY(halloc(b, sl.pf[f].h_comp, (size_t)comp_bytes + XMB_COMP_PAD));      // (+ pad: the inflate launch reads it in place)
            sl.pf[f].d_comp = sl.d_comp_all + f * sl.comp_stride;
        }
        sl.comp_cap = comp_bytes;
    }
    if (raw_bytes > sl.raw_cap) {
        sl.raw_cap = 0;
        sl.raw_stride = (raw_bytes + 64u + 255u) & ~(uint64_t)255;
        XMB_TRY(dalloc(b, sl.d_raw_all, (size_t)(2 * sl.raw_stride)));
        for (int f = 0; f < 2; ++f) {
            sl.pf[f].d_raw = sl.d_raw_all + f * sl.raw_stride;
            hfree(sl.pf[f].h_raw);                                           // made again, for the new capacity, by the next xm_bamdev_fetch_raw
        }
        sl.packed_stride = (raw_bytes + 64u + 255u) & ~(uint64_t)255;
        XMB_TRY(dalloc(b, sl.d_packed_all, (size_t)(2 * sl.packed_stride))); XMB_TRY(halloc(b, sl.h_packed_all, (size_t)(2 * sl.packed_stride)));
        for (int f = 0; f < 2; ++f) {
            sl.pf[f].d_packed = sl.d_packed_all + f * sl.packed_stride;
            sl.pf[f].h_packed = sl.h_packed_all + f * sl.packed_stride;
        }
        sl.raw_cap = raw_bytes;
    }
    if (max_blocks > sl.block_cap) {
        sl.block_cap = 0;
        const size_t nb = (size_t)max_blocks + 2;
        XMB_TRY(halloc(b, sl.h_blocks, 2 * nb)); XMB_TRY(dalloc(b, sl.d_blocks, 2 * nb));
        XMB_TRY(halloc(b, sl.h_walk, 2 * nb)); XMB_TRY(dalloc(b, sl.d_walk, 2 * nb));
        XMB_TRY(dalloc(b, sl.d_status, 2 * nb)); XMB_TRY(halloc(b, sl.h_status, 2 * nb));
        XMB_TRY(dalloc(b, sl.d_crc, 2 * nb)); XMB_TRY(halloc(b, sl.h_crc, 2 * nb));
        for (int f = 0; f < 2; ++f) {
            PerFile &q = sl.pf[f];
            // segments: the blocks plus the pieces of a carried tail (at most as many again)
            XMB_TRY(halloc(b, q.h_seg, 2 * nb + 4)); XMB_TRY(dalloc(b, q.d_seg, 2 * nb + 4));
            XMB_TRY(dalloc(b, q.d_cnt, 2 * nb + 4)); XMB_TRY(dalloc(b, q.d_exit, 2 * nb + 4)); XMB_TRY(dalloc(b, q.d_base, 2 * nb + 4));
        }
        sl.block_cap = max_blocks;
    }
    if (max_records > sl.record_cap) {
        sl.record_cap = 0;
        const size_t n = (size_t)max_records + 64;
        for (int f = 0; f < 2; ++f) {
            PerFile &q = sl.pf[f];
            XMB_TRY(dalloc(b, q.d_rec_off, n)); XMB_TRY(halloc(b, q.h_rec_off, n));
            XMB_TRY(dalloc(b, q.d_name_off, n)); XMB_TRY(dalloc(b, q.d_name_len, n));
            XMB_TRY(dalloc(b, q.d_a, n)); XMB_TRY(dalloc(b, q.d_x, n));
            XMB_TRY(dalloc(b, q.d_rflag, n)); XMB_TRY(dalloc(b, q.d_lflag, n)); XMB_TRY(halloc(b, q.h_lflag, n));
            XMB_TRY(dalloc(b, q.d_wsize, n)); XMB_TRY(dalloc(b, q.d_place, n)); XMB_TRY(halloc(b, q.h_place, n));
            XMB_TRY(dalloc(b, q.d_part, n / SCAN_TILE + 8));
        }
        for (int c = 0; c < 4; ++c) XMB_TRY(dalloc(b, sl.d_col[c], n));
        XMB_TRY(dalloc(b, sl.d_bits, n / 64 + 2));
        XMB_TRY(dalloc(b, sl.d_code, n));
        XMB_TRY(dalloc(b, sl.d_bins4, (size_t)XM_BINS4_BYTES(max_records) + 16));
        XMB_TRY(dalloc(b, sl.d_idx, n));
        XMB_TRY(dalloc(b, sl.d_usize, n)); XMB_TRY(dalloc(b, sl.d_uplace, n)); XMB_TRY(dalloc(b, sl.d_upart, n / SCAN_TILE + 8));
        XMB_TRY(halloc(b, sl.h_code, n));
        XMB_TRY(halloc(b, sl.h_idx, n));
        sl.record_cap = max_records;
    }
    // the slots the inflate launch notes records in: slot_of() of the largest window and block count
    const uint64_t need_slots = sl.raw_cap / 36u + 2u * sl.block_cap + 64u;
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        if (need_slots > q.slots_len) {
            q.slots_len = 0;
            XMB_TRY(dalloc(b, q.d_s_off, (size_t)need_slots)); XMB_TRY(dalloc(b, q.d_s_name_off, (size_t)need_slots));
            XMB_TRY(dalloc(b, q.d_s_name_len, (size_t)need_slots)); XMB_TRY(dalloc(b, q.d_s_a, (size_t)need_slots));
            XMB_TRY(dalloc(b, q.d_s_x, (size_t)need_slots)); XMB_TRY(dalloc(b, q.d_s_flag, (size_t)need_slots));
            q.slots_len = need_slots;
        }
    }
    return XM_OK;
}

uint8_t *xm_bamdev_staging(xm_bamdev *b, int slot, int file)
{
    if (!b || slot < 0 || slot > 1 || file < 0 || file > 1) return nullptr;
    return b->slot[slot].pf[file].h_comp;
}

int xm_bamdev_run(xm_bamdev *b, int slot, const xm_bamdev_input in[2], int score_mode, int paired, int skip_repeated, int keep_halo,
                  uint64_t max_records, xm_bamdev_block *out)
{
    if (!b || !in || !out || slot < 0 || slot > 1 || (score_mode != XMS_SCORE_AS_XS && score_mode != XMS_SCORE_AS_ZS && score_mode != XMS_SCORE_CIGAR) || max_records == 0)
        return XM_ERR_INVALID_ARG;
    Slot &sl = b->slot[slot];
    if (max_records > sl.record_cap) return XM_ERR_INVALID_ARG;
    const bool cigar = score_mode == XMS_SCORE_CIGAR;
    const uint32_t tags = cigar ? xmrec::TAGS_NM_XS : score_mode == XMS_SCORE_AS_ZS ? xmrec::TAGS_AS_ZS : xmrec::TAGS_AS_XS;
    sl.last_score_mode = score_mode;
    memset(out, 0, sizeof *out);
    out->mismatch_at = -1;
    sl.have_columns = false;
    sl.classified = false;
    XMB_HIP(b, hipSetDevice(b->device));
    if (cigar) XMB_TRY(ensure_cigar(b, sl));
    const bool skip = skip_repeated != 0;
    if (skip) XMB_TRY(ensure_skip(b, sl, cigar));
    hipStream_t st = sl.stream;
    static const bool profile = getenv("XM_BAMDEV_PROFILE") != nullptr;
    const bool zero_copy = zero_copy_input();
    // The block, walk and segment tables (24 + ~110 bytes a block, 4 bytes a segment) are read by the kernels where the host wrote
    // them -- page-locked, device-mapped -- instead of being copied up first: a copy of 0.1-2 MB is ONE workgroup of the runtime's
    // blit pulling 4 KB per round trip of the link, 0.3-2.8 ms each and four of them in front of every inflate launch
    // (profiles/r06_bam_timeline.txt); a chain's own read of its 134 bytes is one round trip per block.  XM_BAMDEV_ZEROCOPY_TABLES=0: copy.
    static const bool tables_in_place = [] { const char *v = getenv("XM_BAMDEV_ZEROCOPY_TABLES"); return !(v && v[0] == '0'); }();
    const xm_bgzf_block *blocks_at = tables_in_place ? sl.h_blocks : sl.d_blocks;
    const xm_bgzf_walk *walk_at = tables_in_place ? sl.h_walk : sl.d_walk;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    double t_staged = 0, t_issued = 0, t_sync1 = 0;
    uint64_t new_bytes[2] = {0, 0}, first_block[2] = {0, 0}, n_all = 0;
    if (in[0].n_blocks + in[1].n_blocks > 2 * sl.block_cap) return XM_ERR_INVALID_ARG;
    // ---- stage: carry, compressed bytes, block tables; inflate + CRC; the record chain --------------------------------
    // this slot's previous window has left d_raw (its copy to the host ended long ago: the caller has printed from it)
    if (sl.raw_issued) XMB_HIP(b, hipStreamWaitEvent(st, sl.ev_raw, 0));
    {
        // The OTHER slot's printer (G2 of the window in front) first: beside this window's inflate launch the two take 28 ms + 18 ms
        // where one after the other they take 6 + 17 (the decoder's window reads wait behind the printer's 6.6 GB of traffic: an
        // issue-bound launch turns latency-bound); profiles/r06_ab_serial_fill.txt.  XM_BAMDEV_SERIAL_FILL=0: side by side.
        static const bool serial = [] { const char *v = getenv("XM_BAMDEV_SERIAL_FILL"); return !(v && v[0] == '0'); }();
        Slot &other = b->slot[slot ^ 1];
        if (serial && other.fill_issued) XMB_HIP(b, hipStreamWaitEvent(st, other.ev_inflated, 0));
    }
    XMB_HIP(b, hipEventRecord(sl.ev[0], st));
    for (int f = 0; f < 2; ++f) {
        const xm_bamdev_input &x = in[f];
        PerFile &q = sl.pf[f];
        if (x.comp_len > sl.comp_cap || x.n_blocks > sl.block_cap || (x.n_blocks && (!x.blocks || !x.crc)) || x.carry_len > sl.raw_cap)
            return XM_ERR_INVALID_ARG;
        for (uint64_t k = 0; k < x.n_blocks; ++k) {
            const xm_bgzf_block &d = x.blocks[k];
            if (d.cdata_off + d.cdata_len > x.comp_len || d.isize > 65536u || d.out_off != new_bytes[f]) return XM_ERR_INVALID_ARG;
            new_bytes[f] += d.isize;
        }
        if (x.carry_len + new_bytes[f] > sl.raw_cap) return XM_ERR_INVALID_ARG;
        if (x.carry_len) {
            if (x.carry_slot < 0 || x.carry_slot > 1) return XM_ERR_INVALID_ARG;
            const PerFile &src = b->slot[x.carry_slot].pf[f];
            if (x.carry_off + x.carry_len > b->slot[x.carry_slot].raw_cap) return XM_ERR_INVALID_ARG;
            if (x.carry_slot == slot && x.carry_off < x.carry_len) return XM_ERR_INVALID_ARG;          // would overlap itself
            // (the other slot's stream finished its window before the caller could know what to carry; its copy of the window
            // to the host may still be on its way)
            XMB_HIP(b, hipMemcpyAsync(q.d_raw, src.d_raw + x.carry_off, (size_t)x.carry_len, hipMemcpyDeviceToDevice, st));
        }
        q.raw_len = x.carry_len + new_bytes[f];
        // segments: the carry (when there is one), then every block; the launch's block table holds both files
        uint32_t n_seg = 0;
        if (x.carry_len) {
            // The carried bytes are records of the previous window: ONE lane following their chain takes a microsecond a record
            // (a 13 MB tail = 45 k records = 45 ms, and the tail of the file with the shorter records grows window by window),
            // so the carry is cut into segments of CARRY_SEG records at boundaries the previous window's record table knows.
            q.h_seg[n_seg++] = 0u;
            const PerFile &src = b->slot[x.carry_slot].pf[f];
            const uint32_t *tab = src.h_rec_off;
            const uint64_t tn = src.table_len;
            const uint32_t *at = std::lower_bound(tab, tab + tn, (uint32_t)x.carry_off);
            if (tn && at != tab + tn && *at == x.carry_off) {
                const uint64_t j0 = (uint64_t)(at - tab);
                for (uint64_t j = j0 + CARRY_SEG; j < tn && n_seg + 2 < sl.block_cap; j += CARRY_SEG) q.h_seg[n_seg++] = tab[j] - (uint32_t)x.carry_off;
            }
        }
        if (x.skip && (x.carry_len || x.n_blocks == 0 || x.skip >= x.blocks[0].isize)) return XM_ERR_INVALID_ARG;
        q.h_summary[9] = n_seg;                                             // segments of the carried tail: walked by walk_kernel
        for (uint64_t k = 0; k < x.n_blocks; ++k) {
            xm_bgzf_block d = x.blocks[k];
            const uint64_t out_in_file = d.out_off + x.carry_len;
            // the block's records are found and read by the chain of lanes that inflates it (xm_bgzf_inflate_walk_dev)
            xm_bgzf_walk w;
            memset(&w, 0, sizeof w);
            if (d.isize) {
                const uint32_t kb = n_seg - q.h_summary[9];
                w.raw_base = (uint64_t)f * sl.raw_stride;
                w.start = (uint32_t)(out_in_file + (k == 0 ? x.skip : 0));
                w.end = (uint32_t)(out_in_file + d.isize);
                w.n_raw = (uint32_t)q.raw_len;
                const uint64_t so = slot_of(w.start, kb);
                w.slot_cap = (uint32_t)(slot_of(w.end, kb + 1u) - so);
                w.count = q.d_cnt + n_seg;
                w.exit_at = q.d_exit + n_seg;
                w.slots = q.d_s_off + so;
                w.name_off = q.d_s_name_off + so; w.name_len = q.d_s_name_len + so;
                w.a = q.d_s_a + so; w.x = q.d_s_x + so; w.flag = q.d_s_flag + so;
                w.tags = tags;
                if (cigar) { w.n_cigar = q.d_s_ncig + so; w.cig_at = q.d_s_cig_at + so; }
                q.h_seg[n_seg++] = w.start;
            }
            sl.h_walk[n_all + k] = w;
            if (zero_copy) d.cdata_off += (uint64_t)(reinterpret_cast<uintptr_t>(q.h_comp) - reinterpret_cast<uintptr_t>(sl.pf[0].h_comp));
            else d.cdata_off += (uint64_t)f * sl.comp_stride;
            d.out_off = out_in_file + (uint64_t)f * sl.raw_stride;
            sl.h_blocks[n_all + k] = d;
        }
        q.h_seg[n_seg] = (uint32_t)q.raw_len;
        q.h_summary[8] = n_seg;
        first_block[f] = n_all;
        n_all += x.n_blocks;
        // the part of the staged bytes that went up ahead of this call (xm_bamdev_upload) is waited for, the rest is sent now
        const uint64_t up = x.uploaded < x.comp_len ? x.uploaded : x.comp_len;
        if (up > sl.up_len[f]) return XM_ERR_INVALID_ARG;
        if (up && !zero_copy) XMB_HIP(b, hipStreamWaitEvent(st, sl.ev_up[f], 0));
        if (x.comp_len > up && !zero_copy)
            XMB_HIP(b, hipMemcpyAsync(q.d_comp + up, q.h_comp + up, (size_t)(x.comp_len - up), hipMemcpyHostToDevice, st));
        sl.up_len[f] = 0;
        if (!tables_in_place) XMB_HIP(b, hipMemcpyAsync(q.d_seg, q.h_seg, (size_t)(n_seg + 1) * 4, hipMemcpyHostToDevice, st));
    }
    t_staged = since();
    if (n_all) {
        if (!tables_in_place) {
            XMB_HIP(b, hipMemcpyAsync(sl.d_blocks, sl.h_blocks, (size_t)n_all * sizeof(xm_bgzf_block), hipMemcpyHostToDevice, st));
            XMB_HIP(b, hipMemcpyAsync(sl.d_walk, sl.h_walk, (size_t)n_all * sizeof(xm_bgzf_walk), hipMemcpyHostToDevice, st));
        }
        int rc = xm_bgzf_inflate_walk_dev(b->ctx, st, zero_copy ? sl.pf[0].h_comp : sl.d_comp_all, blocks_at, n_all, sl.d_raw_all, sl.d_status, sl.d_work, walk_at);
        if (rc == XM_OK) rc = xm_bgzf_crc32_dev(b->ctx, st, sl.d_raw_all, blocks_at, n_all, sl.d_crc);
        if (rc != XM_OK) return rc;
        XMB_HIP(b, hipMemcpyAsync(sl.h_status, sl.d_status, (size_t)n_all * 4, hipMemcpyDeviceToHost, st));
        XMB_HIP(b, hipMemcpyAsync(sl.h_crc, sl.d_crc, (size_t)n_all * 4, hipMemcpyDeviceToHost, st));
    }
    XMB_HIP(b, hipEventRecord(sl.ev[1], st));
    // (what of the inflated window goes back for the writer is asked for afterwards: xm_bamdev_fetch_wanted / _fetch_raw)
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        const uint32_t n_seg = q.h_summary[8];
        const uint32_t *seg_at = tables_in_place ? q.h_seg : q.d_seg;
        XMB_HIP(b, hipMemsetAsync(q.d_summary, 0, 8 * sizeof(uint32_t), st));
        if (n_seg) {
            // counts, exits, record starts and fields of the blocks' segments came with the inflate launch; the pieces of the
            // carried tail (bytes of the previous window) are walked here
            const uint32_t n_carry_seg = q.h_summary[9];
            const uint32_t rec_cap = (uint32_t)std::min<uint64_t>(sl.record_cap, 0xFFFFFFFFull);
            if (n_carry_seg)
                walk_kernel<false><<<(n_carry_seg + WALK_T - 1u) / WALK_T, WALK_T, 0, st>>>(q.d_raw, (uint32_t)q.raw_len, seg_at, n_carry_seg, q.d_cnt, q.d_exit,
                                                                                          nullptr, nullptr, 0u);
            scan_kernel<<<1, 1024, 0, st>>>(q.d_cnt, q.d_exit, seg_at, n_seg, q.d_base, q.d_summary);
            if (n_carry_seg)
                walk_kernel<true><<<(n_carry_seg + WALK_T - 1u) / WALK_T, WALK_T, 0, st>>>(q.d_raw, (uint32_t)q.raw_len, seg_at, n_carry_seg, nullptr, nullptr,
                                                                                         q.d_base, q.d_rec_off, rec_cap);
            if (n_seg > n_carry_seg) {
                const SlotArrays sa = {q.d_s_off, q.d_s_name_off, q.d_s_name_len, q.d_s_a, q.d_s_x, q.d_s_flag,
                                       cigar ? q.d_s_ncig : nullptr, cigar ? q.d_s_cig_at : nullptr};
                gather_kernel<<<(n_seg - n_carry_seg + 3u) / 4u, 256, 0, st>>>(sa, seg_at, n_carry_seg, n_seg, q.d_cnt, q.d_base, q.d_rec_off, q.d_name_off,
                                                                               q.d_name_len, q.d_a, q.d_x, q.d_rflag, q.d_ncig, q.d_cig_at, rec_cap);
            }
        }
        XMB_HIP(b, hipMemcpyAsync(q.h_summary, q.d_summary, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    t_issued = since();
    XMB_HIP(b, hipEventRecord(sl.ev_wait, st));
    XMB_HIP(b, hipEventSynchronize(sl.ev_wait));
    t_sync1 = since();
    if (hipGetLastError() != hipSuccess) return XM_ERR_HIP;
    // ---- what the device found ------------------------------------------------------------------------------------------
    uint64_t n_rec[2], stop[2];
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        for (uint64_t k = 0; k < in[f].n_blocks; ++k)
            if (sl.h_status[first_block[f] + k] != 0u || sl.h_crc[first_block[f] + k] != in[f].crc[k]) out->bad_block = 1;
        const uint32_t n_seg = q.h_summary[8];
        n_rec[f] = n_seg ? q.h_summary[0] : 0;
        stop[f] = n_seg ? q.h_summary[1] : 0;
        if (n_seg && q.h_summary[2] == 0u) out->unaligned = 1;
        if (n_rec[f] > sl.record_cap) n_rec[f] = sl.record_cap;            // the caller's windows are sized so that this cannot bind
        if (in[f].eof && stop[f] != q.raw_len && !out->unaligned) out->bad_block = 2;     // the file ends inside a record
    }
    out->raw_len1 = sl.pf[0].raw_len; out->raw_len2 = sl.pf[1].raw_len;
    out->n_rec1 = n_rec[0]; out->n_rec2 = n_rec[1];
    out->raw1 = sl.pf[0].h_raw; out->raw2 = sl.pf[1].h_raw;
    out->rec_off1 = skip ? sl.pf[0].h_pair_off : sl.pf[0].h_rec_off; out->rec_off2 = skip ? sl.pf[1].h_pair_off : sl.pf[1].h_rec_off;
    out->flags1 = sl.pf[0].h_lflag; out->flags2 = sl.pf[1].h_lflag;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, sl.ev[0], sl.ev[1]);
    out->ms_inflate = ms;
    sl.pf[0].table_len = sl.pf[1].table_len = 0;
    if (out->bad_block || out->unaligned) return XM_OK;
    uint64_t n = std::min(std::min(n_rec[0], n_rec[1]), max_records);
    const bool by_runs = skip && n > 0;                                    // the run kernels ran (a file without a record has no run either)
    // ---- strip + pair ------------------------------------------------------------------------------------------------------
    XMB_HIP(b, hipEventRecord(sl.ev[1], st));
    sl.h_state[0] = 0xFFFFFFFFu; sl.h_state[1] = 0; sl.h_state[2] = 0; sl.h_state[3] = 0xFFFFFFFFu;
    sl.h_state[4] = sl.h_state[5] = sl.h_state[6] = sl.h_state[7] = 0;
    XMB_HIP(b, hipMemcpyAsync(sl.d_state, sl.h_state, 8 * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    bool whole[2];
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        whole[f] = in[f].eof && stop[f] == q.raw_len;                      // no record of the file lies behind this window
        q.v_rec_off = skip ? q.d_r_rec_off : q.d_rec_off;
        q.v_ncig = skip ? q.d_r_ncig : q.d_ncig;
        q.v_cig_at = skip ? q.d_r_cig_at : q.d_cig_at;
    }
    if (n) {
        for (int f = 0; f < 2; ++f) {
            // the records of the carried tail (they come first): parsed here; the blocks' records came parsed with the inflate
            // launch.  Their number is base[first block segment], known on the device only: lanes behind it leave at once.
            PerFile &q = sl.pf[f];
            const uint32_t n_seg = q.h_summary[8], n_carry_seg = q.h_summary[9];
            if (n_carry_seg == 0u) continue;
            const RecOut ro = {q.d_name_off, q.d_name_len, q.d_a, q.d_x, q.d_rflag, cigar ? q.d_ncig : nullptr, cigar ? q.d_cig_at : nullptr};
            const uint64_t upper = std::min<uint64_t>(skip ? n_rec[f] : n, in[f].carry_len / 36u + 1u);     // (the skipping walk reads every name)
            parse_kernel<<<(uint32_t)((upper + 255) / 256), 256, 0, st>>>(q.d_raw, q.d_rec_off, (uint32_t)upper, n_carry_seg < n_seg ? q.d_base + n_carry_seg : nullptr,
                                                                         tags, ro);
        }
        for (int f = 0; f < 2 && skip; ++f) {
            // the run starts of the file's records, their fields next to each other; how many: d_state[4 + f]
            PerFile &q = sl.pf[f];
            const uint32_t nf = (uint32_t)n_rec[f], n_part = (nf + SCAN_TILE - 1u) / SCAN_TILE;
            run_start_kernel<<<(nf + 255u) / 256u, 256, 0, st>>>(q.d_raw, q.d_name_off, q.d_name_len, q.d_rflag, nf, q.d_wsize, sl.d_state);
            size_sum_kernel<<<n_part, 256, 0, st>>>(q.d_wsize, nf, q.d_part);
            part_scan_kernel<<<1, 1024, 0, st>>>(q.d_part, n_part, sl.d_state + 4 + f);
            size_place_kernel<false><<<n_part, 256, 0, st>>>(q.d_wsize, nf, q.d_part, q.d_place);
            const RecCols from = {q.d_rec_off, q.d_name_off, q.d_name_len, q.d_a, q.d_x, q.d_rflag, cigar ? q.d_ncig : nullptr, cigar ? q.d_cig_at : nullptr};
            const RecCols to = {q.d_r_rec_off, q.d_r_name_off, q.d_r_name_len, q.d_r_a, q.d_r_x, q.d_r_flag, q.d_r_ncig, q.d_r_cig_at};
            run_select_kernel<<<(nf + 255u) / 256u, 256, 0, st>>>(from, q.d_place, nf, to);
        }
        const PerFile &a = sl.pf[0], &c = sl.pf[1];
        const FileRecs f1 = skip ? FileRecs{a.d_raw, a.d_r_name_off, a.d_r_name_len, a.d_r_a, a.d_r_x, a.d_r_flag}
                                 : FileRecs{a.d_raw, a.d_name_off, a.d_name_len, a.d_a, a.d_x, a.d_rflag};
        const FileRecs f2 = skip ? FileRecs{c.d_raw, c.d_r_name_off, c.d_r_name_len, c.d_r_a, c.d_r_x, c.d_r_flag}
                                 : FileRecs{c.d_raw, c.d_name_off, c.d_name_len, c.d_a, c.d_x, c.d_rflag};
        pair_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, st>>>(f1, f2, (uint32_t)n, paired ? 1 : 0, skip ? sl.d_state + 4 : nullptr,
                                                                   (whole[0] ? 0u : 1u) | (whole[1] ? 0u : 2u), sl.d_col[0], sl.d_col[1], sl.d_col[2], sl.d_col[3],
                                                                   reinterpret_cast<unsigned long long *>(sl.d_bits), sl.pf[0].d_lflag, sl.pf[1].d_lflag, sl.d_state);
        for (int f = 0; f < 2; ++f) {
            PerFile &q = sl.pf[f];
            XMB_HIP(b, hipMemcpyAsync(q.h_lflag, q.d_lflag, (size_t)n, hipMemcpyDeviceToHost, st));
            if (skip)                                                      // where the run starts' records begin: entry n too, when there is one
                XMB_HIP(b, hipMemcpyAsync(q.h_pair_off, q.d_r_rec_off, (size_t)std::min<uint64_t>(n + 1, n_rec[f]) * 4, hipMemcpyDeviceToHost, st));
        }
    }
    for (int f = 0; f < 2; ++f)                                            // the whole record table: the writer prints from it, and the next
        if (n_rec[f])                                                      // window cuts its carried tail at boundaries it lists
            XMB_HIP(b, hipMemcpyAsync(sl.pf[f].h_rec_off, sl.pf[f].d_rec_off, (size_t)n_rec[f] * 4, hipMemcpyDeviceToHost, st));
    XMB_HIP(b, hipMemcpyAsync(sl.h_state, sl.d_state, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    XMB_HIP(b, hipEventRecord(sl.ev[2], st));
    XMB_HIP(b, hipEventRecord(sl.ev_wait, st));
    XMB_HIP(b, hipEventSynchronize(sl.ev_wait));
    if (hipGetLastError() != hipSuccess) return XM_ERR_HIP;
    (void)hipEventElapsedTime(&ms, sl.ev[1], sl.ev[2]);
    out->ms_kernels = ms;
    if (sl.h_state[2] & 2u) { out->bad_block = 3; return XM_OK; }          // a malformed record
    if (sl.h_state[2] & 1u) out->weird = 1;
    // what each file can pair: its records, or (the skipping walk) its runs
    uint64_t n_can[2] = {n_rec[0], n_rec[1]};
    if (by_runs) {
        n_can[0] = sl.h_state[4]; n_can[1] = sl.h_state[5];
        if (n_can[0] > n_rec[0] || n_can[1] > n_rec[1]) return XM_ERR_HIP;
        n = std::min(n, std::min(n_can[0], n_can[1]));
    }
    bool cut = false, open_run = false;
    if (sl.h_state[0] != 0xFFFFFFFFu && sl.h_state[0] < n) {
        out->mismatch_at = (int64_t)sl.h_state[0];
        n = sl.h_state[0];                                                  // records at and behind it are not reported
        cut = true;
    }
    if (by_runs && sl.h_state[3] < n) {                                        // a run that may go on in the next window comes first
        out->mismatch_at = -1;
        n = sl.h_state[3];
        cut = open_run = true;
    }
    out->n_exceptions = sl.h_state[1];
    if (cut) {                                                              // the device counted the pairs behind the stop too
        uint64_t e = 0;
        for (uint64_t k = 0; k < n; ++k) e += ((sl.pf[0].h_lflag[k] | sl.pf[1].h_lflag[k]) & (XMS_LINE_EX_A | XMS_LINE_EX_X)) ? 1u : 0u;
        out->n_exceptions = e;
    }
    // the walk's outcome, as xmh_parse reports it: a file that has no record left AND no byte left at its end ends the walk
    bool ended = false, starved = false;
    if (out->mismatch_at < 0) {
        for (int f = 0; f < 2 && !open_run; ++f) {
            const bool exhausted = n == n_can[f];                           // the window holds no further complete record (or run)
            if (exhausted && whole[f]) ended = true;
        }
        if (!ended && n < max_records) starved = true;                      // a window ran out before the other file did
    }
    if (profile)
        fprintf(stderr, "xm_bamdev_run: staged %.1f ms (carry memmove, tables, copies queued), issued %.1f, first sync %.1f, done %.1f; "
                        "device: inflate+crc+h2d %.1f ms, record kernels %.1f ms; %.1f + %.1f MB inflated\n",
                t_staged, t_issued, t_sync1, since(), out->ms_inflate, out->ms_kernels, new_bytes[0] / 1e6, new_bytes[1] / 1e6);
    out->ended = ended ? 1 : 0;
    out->starved = starved ? 1 : 0;
    out->n_records = n;
    sl.pf[0].table_len = n_rec[0];
    sl.pf[1].table_len = n_rec[1];
    for (int f = 0; f < 2; ++f) {
        const PerFile &q = sl.pf[f];
        const uint32_t *table = by_runs ? q.h_pair_off : q.h_rec_off;
        uint64_t c = n < n_can[f] ? table[n] : stop[f];                    // first byte behind the yielded records (and the ones skipped)
        if (keep_halo && n > 0 && !ended && out->mismatch_at < 0) c = table[n - 1];
        (f == 0 ? out->consumed1 : out->consumed2) = c;
    }
    sl.have_columns = true;
    return XM_OK;
}

int xm_bamdev_upload(xm_bamdev *b, int slot, int file, uint64_t bytes)
{
    if (!b || slot < 0 || slot > 1 || file < 0 || file > 1) return XM_ERR_INVALID_ARG;
    Slot &sl = b->slot[slot];
    if (bytes > sl.comp_cap) return XM_ERR_INVALID_ARG;
    sl.up_len[file] = 0;
    if (bytes == 0) return XM_OK;
    XMB_HIP(b, hipSetDevice(b->device));
    if (zero_copy_input()) { sl.up_len[file] = bytes; return XM_OK; }       // nothing to send: the inflate launch reads the staging buffer
    XMB_HIP(b, hipMemcpyAsync(sl.pf[file].d_comp, sl.pf[file].h_comp, (size_t)bytes, hipMemcpyHostToDevice, sl.up_stream));
    XMB_HIP(b, hipEventRecord(sl.ev_up[file], sl.up_stream));
    sl.up_len[file] = bytes;
    return XM_OK;
}

int xm_bamdev_fetch_raw(xm_bamdev *b, int slot)
{
    if (!b || slot < 0 || slot > 1) return XM_ERR_INVALID_ARG;
    Slot &sl = b->slot[slot];
    XMB_HIP(b, hipSetDevice(b->device));
    for (int f = 0; f < 2; ++f)                                             // the host copies of whole windows: only a slot that is asked has them
        if (!sl.pf[f].h_raw) XMB_TRY(halloc(b, sl.pf[f].h_raw, (size_t)sl.raw_cap + 64));
    XMB_HIP(b, hipEventRecord(sl.ev_inflated, sl.stream));
    XMB_HIP(b, hipStreamWaitEvent(sl.copy_stream, sl.ev_inflated, 0));
    for (int f = 0; f < 2; ++f)
        if (sl.pf[f].raw_len)
            XMB_HIP(b, hipMemcpyAsync(sl.pf[f].h_raw, sl.pf[f].d_raw, (size_t)sl.pf[f].raw_len, hipMemcpyDeviceToHost, sl.copy_stream));
    XMB_HIP(b, hipEventRecord(sl.ev_raw, sl.copy_stream));
    sl.raw_issued = true;
    return XM_OK;
}

const uint8_t *xm_bamdev_raw(xm_bamdev *b, int slot, int file)
{
    if (!b || slot < 0 || slot > 1 || file < 0 || file > 1) return nullptr;
    return b->slot[slot].pf[file].h_raw;
}

int xm_bamdev_fetch_wanted(xm_bamdev *b, int slot, uint64_t n_records, int paired, uint32_t sink_mask, xm_bamdev_text *out)
{
    if (!b || slot < 0 || slot > 1 || !out) return XM_ERR_INVALID_ARG;
    Slot &sl = b->slot[slot];
    if (n_records > sl.record_cap || n_records > 0xFFFFFFF0ull || !sl.have_columns || !sl.classified) return XM_ERR_INVALID_ARG;
    memset(out, 0, sizeof *out);
    out->raw1 = sl.pf[0].h_packed; out->raw2 = sl.pf[1].h_packed;
    out->off1 = sl.pf[0].h_place; out->off2 = sl.pf[1].h_place;
    XMB_HIP(b, hipSetDevice(b->device));
    hipStream_t st = sl.stream;
    const uint32_t n = (uint32_t)n_records;
    if (n) {
        const uint32_t n_part = (n + SCAN_TILE - 1u) / SCAN_TILE;
        want_kernel<<<(n + 255u) / 256u, 256, 0, st>>>(sl.pf[0].d_raw, sl.pf[1].d_raw, sl.pf[0].v_rec_off, sl.pf[1].v_rec_off, sl.d_bins4, n, paired ? 1 : 0,
                                                       sink_mask, sl.pf[0].d_wsize, sl.pf[1].d_wsize);
        for (int f = 0; f < 2; ++f) {
            PerFile &q = sl.pf[f];
            size_sum_kernel<<<n_part, 256, 0, st>>>(q.d_wsize, n, q.d_part);
            part_scan_kernel<<<1, 1024, 0, st>>>(q.d_part, n_part, sl.d_state + 8 + f);
            size_place_kernel<false><<<n_part, 256, 0, st>>>(q.d_wsize, n, q.d_part, q.d_place);
            pack_kernel<<<(n + 3u) / 4u, 256, 0, st>>>(q.d_raw, q.v_rec_off, q.d_wsize, q.d_place, n, q.d_packed);
        }
        XMB_HIP(b, hipMemcpyAsync(sl.h_state + 8, sl.d_state + 8, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        XMB_HIP(b, hipEventRecord(sl.ev_wait, st));
        XMB_HIP(b, hipEventSynchronize(sl.ev_wait));
        if (hipGetLastError() != hipSuccess) return XM_ERR_HIP;
        out->bytes1 = sl.h_state[8]; out->bytes2 = sl.h_state[9];
        if (out->bytes1 > sl.raw_cap || out->bytes2 > sl.raw_cap) return XM_ERR_HIP;
    }
    // the packed records and the table of where each went, on the copy stream (the kernels above have finished)
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        const uint64_t bytes = f == 0 ? out->bytes1 : out->bytes2;
        if (bytes) XMB_HIP(b, hipMemcpyAsync(q.h_packed, q.d_packed, (size_t)bytes, hipMemcpyDeviceToHost, sl.copy_stream));
        if (n) XMB_HIP(b, hipMemcpyAsync(q.h_place, q.d_place, (size_t)n * 4, hipMemcpyDeviceToHost, sl.copy_stream));
    }
    XMB_HIP(b, hipEventRecord(sl.ev_raw, sl.copy_stream));
    sl.raw_issued = true;
    return XM_OK;
}

int xm_bamdev_set_refs(xm_bamdev *b, int file, const uint8_t *names, const uint32_t *at, uint32_t n_refs)
{
    if (!b || file < 0 || file > 1 || (n_refs && (!names || !at))) return XM_ERR_INVALID_ARG;
    for (uint32_t k = 0; k < n_refs; ++k)
        if (at[k + 1] < at[k]) return XM_ERR_INVALID_ARG;
    XMB_HIP(b, hipSetDevice(b->device));
    XMB_HIP(b, hipDeviceSynchronize());
    for (int slot = 0; slot < 2; ++slot) {                                  // a copy per slot: the slots' streams share nothing
        PerFile &q = b->slot[slot].pf[file];
        q.n_refs = 0;
        q.refs_set = false;
        const size_t bytes = n_refs ? at[n_refs] : 0;
        XMB_TRY(dalloc(b, q.d_ref_names, bytes + 16)); XMB_TRY(dalloc(b, q.d_ref_at, (size_t)n_refs + 1));
        if (n_refs) {
            if (bytes) XMB_HIP(b, hipMemcpy(q.d_ref_names, names, bytes, hipMemcpyHostToDevice));
            XMB_HIP(b, hipMemcpy(q.d_ref_at, at, ((size_t)n_refs + 1) * 4, hipMemcpyHostToDevice));
        }
        q.n_refs = n_refs;
        q.refs_set = true;
    }
    return XM_OK;
}

int xm_bamdev_fetch_text(xm_bamdev *b, int slot, uint64_t n_records, int paired, uint32_t sink_mask, xm_bamdev_lines *out)
{
    if (!b || slot < 0 || slot > 1 || !out) return XM_ERR_INVALID_ARG;
    Slot &sl = b->slot[slot];
    if (n_records > sl.record_cap || n_records > 0xFFFFFFF0ull || !sl.have_columns || !sl.classified) return XM_ERR_INVALID_ARG;
    if (!sl.pf[0].refs_set || !sl.pf[1].refs_set) return XM_ERR_INVALID_ARG;          // RNAME / RNEXT need xm_bamdev_set_refs
    memset(out, 0, sizeof *out);
    XMB_HIP(b, hipSetDevice(b->device));
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        if (q.llen_records < sl.record_cap) {
            q.llen_records = 0;
            XMB_TRY(halloc(b, q.h_llen, (size_t)sl.record_cap + 64));
            q.llen_records = sl.record_cap;
        }
    }
    out->text1 = sl.pf[0].h_packed; out->text2 = sl.pf[1].h_packed;
    out->line_off1 = sl.pf[0].h_place; out->line_off2 = sl.pf[1].h_place;
    out->line_len1 = sl.pf[0].h_llen; out->line_len2 = sl.pf[1].h_llen;
    hipStream_t st = sl.stream;
    const uint32_t n = (uint32_t)n_records;
    if (n == 0) return XM_OK;
    // the line lengths are summed in 32 bits: a record's text is at most five times its bytes (xm_bam.cpp: a B:c element, 1 -> "-128,"),
    // so windows beyond 2^32 / 5 bytes are the host printer's
    if (sl.pf[0].raw_len > 0x30000000ull || sl.pf[1].raw_len > 0x30000000ull) { out->status = 2; return XM_OK; }
    const uint32_t n_part = (n + SCAN_TILE - 1u) / SCAN_TILE;
    const uint32_t text_cap = (uint32_t)std::min<uint64_t>(sl.raw_cap, 0xFFFFFFF0ull);      // the packed-record buffers hold the text
    XMB_HIP(b, hipMemsetAsync(sl.d_state + 13, 0, sizeof(uint32_t), st));
    want_kernel<<<(n + 255u) / 256u, 256, 0, st>>>(sl.pf[0].d_raw, sl.pf[1].d_raw, sl.pf[0].v_rec_off, sl.pf[1].v_rec_off, sl.d_bins4, n, paired ? 1 : 0,
                                                   sink_mask, sl.pf[0].d_wsize, sl.pf[1].d_wsize);
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        const RefTable refs = {q.d_ref_names, q.d_ref_at, q.n_refs};
        text_size_kernel<<<(n + 255u) / 256u, 256, 0, st>>>(q.d_raw, q.v_rec_off, n, refs, q.d_wsize, sl.d_state);
        size_sum_kernel<<<n_part, 256, 0, st>>>(q.d_wsize, n, q.d_part);
        part_scan_kernel<<<1, 1024, 0, st>>>(q.d_part, n_part, sl.d_state + 8 + f);
        size_place_kernel<false><<<n_part, 256, 0, st>>>(q.d_wsize, n, q.d_part, q.d_place);
    }
    XMB_HIP(b, hipMemcpyAsync(sl.h_state + 8, sl.d_state + 8, 6 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    XMB_HIP(b, hipEventRecord(sl.ev_wait, st));
    XMB_HIP(b, hipEventSynchronize(sl.ev_wait));
    if (hipGetLastError() != hipSuccess) return XM_ERR_HIP;
    out->bytes1 = sl.h_state[8]; out->bytes2 = sl.h_state[9];
    if (sl.h_state[13] != 0u) { out->status = 1; return XM_OK; }            // a field the host prints (binary64)
    if (out->bytes1 > text_cap || out->bytes2 > text_cap) { out->status = 2; return XM_OK; }     // more text than the buffers hold
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        const RefTable refs = {q.d_ref_names, q.d_ref_at, q.n_refs};
        text_fill_kernel<<<(n + 255u) / 256u, 256, 0, st>>>(q.d_raw, q.v_rec_off, n, refs, q.d_wsize, q.d_place, q.d_packed, text_cap);
    }
    // the text and its line table, on the copy stream behind the kernels
    XMB_HIP(b, hipEventRecord(sl.ev_inflated, st));
    XMB_HIP(b, hipStreamWaitEvent(sl.copy_stream, sl.ev_inflated, 0));
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        const uint64_t bytes = f == 0 ? out->bytes1 : out->bytes2;
        if (bytes) XMB_HIP(b, hipMemcpyAsync(q.h_packed, q.d_packed, (size_t)bytes, hipMemcpyDeviceToHost, sl.copy_stream));
        XMB_HIP(b, hipMemcpyAsync(q.h_place, q.d_place, (size_t)n * 4, hipMemcpyDeviceToHost, sl.copy_stream));
        XMB_HIP(b, hipMemcpyAsync(q.h_llen, q.d_wsize, (size_t)n * 4, hipMemcpyDeviceToHost, sl.copy_stream));
    }
    XMB_HIP(b, hipEventRecord(sl.ev_raw, sl.copy_stream));
    sl.raw_issued = true;
    return XM_OK;
}

int xm_bamdev_fetch_bins(xm_bamdev *b, int slot, uint64_t n_records, int paired, uint32_t sink_mask, xm_bamdev_bins *out)
{
    if (!b || slot < 0 || slot > 1 || !out) return XM_ERR_INVALID_ARG;
    Slot &sl = b->slot[slot];
    if (n_records > sl.record_cap || n_records > 0xFFFFFFF0ull || !sl.have_columns || !sl.classified) return XM_ERR_INVALID_ARG;
    if (!sl.pf[0].refs_set || !sl.pf[1].refs_set) return XM_ERR_INVALID_ARG;          // RNAME / RNEXT need xm_bamdev_set_refs
    memset(out, 0, sizeof *out);
    out->text = sl.h_packed_all;
    const uint32_t n = (uint32_t)n_records;
    const uint64_t units64 = sl.h_off_counts[7];                                    // of the slot's last xm_bamdev_classify
    if (n == 0 || units64 == 0) return XM_OK;
    if (units64 > n_records) return XM_ERR_INVALID_ARG;
    const uint32_t n_units = (uint32_t)units64;
    XMB_HIP(b, hipSetDevice(b->device));
    hipStream_t st = sl.stream;
    // line lengths and the places of the units are summed in 32 bits (as in xm_bamdev_fetch_text)
    if (sl.pf[0].raw_len > 0x30000000ull || sl.pf[1].raw_len > 0x30000000ull) { out->status = 2; return XM_OK; }
    const uint64_t out_cap = std::min<uint64_t>(2 * sl.packed_stride - 64u, 0xFFFFFFF0ull);
    const uint32_t n_part = (n_units + SCAN_TILE - 1u) / SCAN_TILE;
    const unsigned long long *d_off = reinterpret_cast<const unsigned long long *>(sl.d_off_counts);
    XMB_HIP(b, hipMemsetAsync(sl.d_state + 13, 0, 13 * sizeof(uint32_t), st));
    want_kernel<<<(n + 255u) / 256u, 256, 0, st>>>(sl.pf[0].d_raw, sl.pf[1].d_raw, sl.pf[0].v_rec_off, sl.pf[1].v_rec_off, sl.d_bins4, n, paired ? 1 : 0,
                                                   sink_mask, sl.pf[0].d_wsize, sl.pf[1].d_wsize);
    RefTable refs[2];
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        refs[f] = RefTable{q.d_ref_names, q.d_ref_at, q.n_refs};
        text_size_kernel<<<(n + 255u) / 256u, 256, 0, st>>>(q.d_raw, q.v_rec_off, n, refs[f], q.d_wsize, sl.d_state);
    }
    unit_size_kernel<<<(n_units + 255u) / 256u, 256, 0, st>>>(sl.d_idx, d_off, n_units, n, paired ? 1 : 0, sink_mask, sl.pf[0].d_wsize, sl.pf[1].d_wsize,
                                                             sl.d_usize, reinterpret_cast<unsigned long long *>(sl.d_state + 24));
    size_sum_kernel<<<n_part, 256, 0, st>>>(sl.d_usize, n_units, sl.d_upart);
    part_scan_kernel<<<1, 1024, 0, st>>>(sl.d_upart, n_part, sl.d_state + 14);
    size_place_kernel<true><<<n_part, 256, 0, st>>>(sl.d_usize, n_units, sl.d_upart, sl.d_uplace);
    // where each bin's text begins: the place of its first unit (state[16..23]; [14] the total, [13] the printer's flag)
    bin_start_kernel<<<1, 64, 0, st>>>(sl.d_uplace, d_off, n_units, sl.d_state + 14, sl.d_state + 16);
    XMB_HIP(b, hipMemcpyAsync(sl.h_state + 13, sl.d_state + 13, 13 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    XMB_HIP(b, hipEventRecord(sl.ev_wait, st));
    XMB_HIP(b, hipEventSynchronize(sl.ev_wait));
    if (hipGetLastError() != hipSuccess) return XM_ERR_HIP;
    if (sl.h_state[13] != 0u) { out->status = 1; return XM_OK; }            // a field the host prints (binary64)
    uint64_t total = 0;
    memcpy(&total, sl.h_state + 24, sizeof total);                          // summed in 64 bits: the 32-bit places hold only if this fits
    if (total > out_cap) { out->status = 2; return XM_OK; }                 // more text than the buffers hold
    for (int k = 0; k < 8; ++k) out->bin_off[k] = sl.h_state[16 + k];
    line_fill_kernel<<<((paired ? 2u : 1u) * n_units + 255u) / 256u, 256, 0, st>>>(
        sl.pf[0].d_raw, sl.pf[1].d_raw, sl.pf[0].v_rec_off, sl.pf[1].v_rec_off, refs[0], refs[1], sl.d_idx, d_off, n_units, n, paired ? 1 : 0, sink_mask,
        sl.pf[0].d_wsize, sl.pf[1].d_wsize, sl.d_usize, sl.d_uplace, sl.d_packed_all, (uint32_t)out_cap,
        []() -> uint32_t { static const bool off = [] { const char *v = getenv("XM_BAMDEV_FILL_WIDE"); return v && v[0] == '0'; }(); return off ? 0xFFFFFFFFu : 16u; }());
    // the stream goes to the host on the copy stream behind the kernels (beside the next window's inflate launch on the other slot)
    XMB_HIP(b, hipEventRecord(sl.ev_inflated, st));
    sl.fill_issued = true;
    XMB_HIP(b, hipStreamWaitEvent(sl.copy_stream, sl.ev_inflated, 0));
    if (total) {
        const uint32_t wg = out_copy_workgroups();
        if (wg == 0u) XMB_HIP(b, hipMemcpyAsync(sl.h_packed_all, sl.d_packed_all, (size_t)total, hipMemcpyDeviceToHost, sl.copy_stream));
        else {
            const uint64_t n16 = (total + 15u) / 16u;                       // (the buffers end 64 bytes behind out_cap)
            out_copy_launch((uint32_t)std::min<uint64_t>(out_copy_waves(wg), (n16 + 63u) / 64u), sl.copy_stream,
                            reinterpret_cast<const v4u32 *>(sl.d_packed_all), reinterpret_cast<v4u32 *>(sl.h_packed_all), n16);
        }
    }
    XMB_HIP(b, hipEventRecord(sl.ev_raw, sl.copy_stream));
    sl.raw_issued = true;
    if (hipGetLastError() != hipSuccess) return XM_ERR_HIP;
    return XM_OK;
}

int xm_bamdev_raw_wait(xm_bamdev *b, int slot)
{
    if (!b || slot < 0 || slot > 1) return XM_ERR_INVALID_ARG;
    Slot &sl = b->slot[slot];
    if (!sl.raw_issued) return XM_OK;
    XMB_HIP(b, hipSetDevice(b->device));
    XMB_HIP(b, hipEventSynchronize(sl.ev_raw));
    return XM_OK;
}

int xm_bamdev_classify(xm_bamdev *b, int slot, int mode, uint64_t n_records, int32_t min_score_floor,
                       const uint8_t **code, const uint32_t **idx, uint64_t bin_offsets[8], uint64_t counts[64])
{
    if (!b || slot < 0 || slot > 1 || !code || !idx || !bin_offsets || !counts) return XM_ERR_INVALID_ARG;
    Slot &sl = b->slot[slot];
    if (n_records > sl.record_cap || !sl.have_columns) return XM_ERR_INVALID_ARG;
    *code = sl.h_code;
    *idx = sl.h_idx;
    memset(bin_offsets, 0, 8 * sizeof(uint64_t));
    memset(counts, 0, 64 * sizeof(uint64_t));
    if (n_records == 0) { sl.classified = true; return XM_OK; }
    XMB_HIP(b, hipSetDevice(b->device));
    hipStream_t st = sl.stream;
    const bool cigar = sl.last_score_mode == XMS_SCORE_CIGAR;
    int rc;
    if (cigar) {
        // the records' CIGAR words as packed CIGAR columns (col 0 / 2 hold NM), then the kernel that makes AS of them
        for (int f = 0; f < 2; ++f) XMB_TRY(pack_cigar(sl, f, (uint32_t)n_records));
        XMB_HIP(b, hipMemsetAsync(sl.d_state + 12, 0, sizeof(uint32_t), st));
        const PerFile &a = sl.pf[0], &c = sl.pf[1];
        rc = xm_classify_compact_cigar_packed_dev(b->ctx, st, mode, n_records, sl.d_col[0], a.d_cig_cnt, a.d_cig_tile, a.d_cig_ops, sl.d_col[1],
                                                  sl.d_col[2], c.d_cig_cnt, c.d_cig_tile, c.d_cig_ops, sl.d_col[3], sl.d_bits, min_score_floor,
                                                  sl.d_code, sl.d_bins4, sl.d_state + 12, sl.d_idx, sl.d_off_counts, sl.d_off_counts + 8);
    } else {
        rc = xm_classify_compact_dev(b->ctx, st, mode, n_records, sl.d_col[0], sl.d_col[1], sl.d_col[2], sl.d_col[3], sl.d_bits,
                                     min_score_floor, sl.d_code, sl.d_bins4, sl.d_idx, sl.d_off_counts, sl.d_off_counts + 8);
    }
    if (rc != XM_OK) {
        std::lock_guard<std::mutex> hold(b->error_lock);
        b->last_error = xm_last_hip_error(b->ctx);
        return rc;
    }
    XMB_HIP(b, hipMemcpyAsync(sl.h_code, sl.d_code, n_records, hipMemcpyDeviceToHost, st));
    XMB_HIP(b, hipMemcpyAsync(sl.h_off_counts, sl.d_off_counts, 72 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    if (cigar) XMB_HIP(b, hipMemcpyAsync(sl.h_state + 12, sl.d_state + 12, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    XMB_HIP(b, hipStreamSynchronize(st));
    if (cigar && sl.h_state[12] != 0u) return XM_ERR_RANGE;               // a score left int32: the caller's text rules decide
    const uint64_t units = sl.h_off_counts[7];
    if (units > n_records) return XM_ERR_HIP;
    if (units) {
        XMB_HIP(b, hipMemcpyAsync(sl.h_idx, sl.d_idx, units * 4, hipMemcpyDeviceToHost, st));
        XMB_HIP(b, hipStreamSynchronize(st));
    }
    memcpy(bin_offsets, sl.h_off_counts, 8 * sizeof(uint64_t));
    memcpy(counts, sl.h_off_counts + 8, 64 * sizeof(uint64_t));
    sl.classified = true;
    return XM_OK;
}

int xm_bamdev_columns(xm_bamdev *b, int slot, uint64_t n_records, int32_t *as1, int32_t *xs1, int32_t *as2, int32_t *xs2,
                      uint64_t *unit_bits)
{
    if (!b || slot < 0 || slot > 1) return XM_ERR_INVALID_ARG;
    Slot &sl = b->slot[slot];
    if (n_records > sl.record_cap || !sl.have_columns) return XM_ERR_INVALID_ARG;
    if (n_records == 0) return XM_OK;
    XMB_HIP(b, hipSetDevice(b->device));
    int32_t *dst[4] = {as1, xs1, as2, xs2};
    for (int c = 0; c < 4; ++c)
        if (dst[c]) XMB_HIP(b, hipMemcpyAsync(dst[c], sl.d_col[c], n_records * 4, hipMemcpyDeviceToHost, sl.stream));
    if (unit_bits) XMB_HIP(b, hipMemcpyAsync(unit_bits, sl.d_bits, (n_records + 63) / 64 * 8, hipMemcpyDeviceToHost, sl.stream));
    XMB_HIP(b, hipStreamSynchronize(sl.stream));
    return XM_OK;
}

int xm_bamdev_cigar_columns(xm_bamdev *b, int slot, int file, uint64_t n_records, int32_t *nm, uint8_t *cig_cnt, uint32_t *cig_tile,
                            uint32_t *cig_ops, uint64_t ops_capacity, uint64_t *n_ops)
{
    if (!b || slot < 0 || slot > 1 || file < 0 || file > 1 || !n_ops) return XM_ERR_INVALID_ARG;
    Slot &sl = b->slot[slot];
    if (n_records > sl.record_cap || !sl.have_columns || sl.last_score_mode != XMS_SCORE_CIGAR) return XM_ERR_INVALID_ARG;
    *n_ops = 0;
    if (n_records == 0) { if (cig_tile) cig_tile[0] = 0; return XM_OK; }
    XMB_HIP(b, hipSetDevice(b->device));
    hipStream_t st = sl.stream;
    const PerFile &q = sl.pf[file];
    XMB_TRY(pack_cigar(sl, file, (uint32_t)n_records));
    XMB_HIP(b, hipMemcpyAsync(sl.h_state + 10 + file, sl.d_state + 10 + file, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (nm) XMB_HIP(b, hipMemcpyAsync(nm, sl.d_col[2 * file], n_records * 4, hipMemcpyDeviceToHost, st));
    if (cig_cnt) XMB_HIP(b, hipMemcpyAsync(cig_cnt, q.d_cig_cnt, n_records, hipMemcpyDeviceToHost, st));
    if (cig_tile) XMB_HIP(b, hipMemcpyAsync(cig_tile, q.d_cig_tile, (XM_CIG_TILES(n_records) + 1) * 4, hipMemcpyDeviceToHost, st));
    XMB_HIP(b, hipStreamSynchronize(st));
    *n_ops = sl.h_state[10 + file];
    if (*n_ops > q.ops_cap) return XM_ERR_HIP;
    if (cig_ops) {
        if (ops_capacity < *n_ops) return XM_ERR_INVALID_ARG;
        if (*n_ops) XMB_HIP(b, hipMemcpyAsync(cig_ops, q.d_cig_ops, *n_ops * 4, hipMemcpyDeviceToHost, st));
        XMB_HIP(b, hipStreamSynchronize(st));
    }
    return XM_OK;
}

const char *xm_bamdev_last_error(const xm_bamdev *b)
{
    static thread_local std::string mine;
    if (!b) return "";
    {
        std::lock_guard<std::mutex> hold(const_cast<xm_bamdev *>(b)->error_lock);
        mine = b->last_error;
    }
    return mine.c_str();
}

}  // extern "C"
