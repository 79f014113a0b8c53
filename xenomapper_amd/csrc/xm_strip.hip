// xm_strip.hip -- the SAM column stripper on the GPU (C ABI in include/xenomapper_strip.h), gfx950 only.
//
// Restates on the device what xm_sam.cpp does with host threads (and what the reference does in Python):
//   getReadPairs                    /root/reference/xenomapper/xenomapper.py:95-118  (readline / strip / split / names equal /
//                                   skip_repeated_reads)
//   get_tag (field search)          :186-190      get_tag_with_ZS_as_XS :204-206
//   get_cigarbased_AS_tag           :247-251      (first NM field, the CIGAR operations "MIDNSHP=X")
//   the unit rule                   :402-405      (name equals the previous record's name)
// All kernels are bound by reading the text from HBM, and the text arrives over PCIe at a hundredth of that rate -- which
// is what bounds the step:
//   S1 mark_kernel    16 bytes per lane: terminator bits of '\n' / '\r\n' / '\r' (Python's universal newlines) as one 16-bit
//                     mask per 16 bytes (SWAR, no branch), terminators per 64 KiB chunk, non-ASCII test
//   S2 chunk_scan     exclusive scan of the chunk counts (one workgroup per file)
//   S3 fill_kernel    reads the masks (1/8 of the text): line i ends at lend[i], line i + 1 starts at lnext[i]
//   S4 parse_kernel   one lane per LINE of each file: split by str.split()'s separators in aligned 8-byte words (words
//                     inside the long leading fields are skipped whole), tags (or NM + the CIGAR field), name -> a
//                     per-line record
//   S5 start_*        skip_repeated_reads only: a line starts a run when its name differs from the line in front; the run
//                     starts of each file are compacted (count, scan, fill) -- pair k is run k of both files
//   S6 pair_kernel    one lane per RECORD k: the two lines' records gathered into the score columns, names compared, unit
//                     bit (ballot -> one 64-bit word per wave), first stop (blank line / names differ / open run) by atomicMin
//   S7 cig_*          --cigar_scores only: operation counts scanned into positions, the CIGAR fields read again and written
//                     as the packed columns the classify kernel reads (xenomapper_hip.h)
//   S8 summary_kernel one lane: the walk's outcome (records, consumed bytes, ended / starved / mismatch) as xmh_parse reports it
// The score columns and the unit mask never leave the device: xm_strip_classify runs the fused classify pass on them.
#include "../../include/xenomapper_strip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "xm_gather.h"
#include "xm_pinned.h"

namespace {

constexpr int SB = 256;                               // lanes per workgroup
constexpr uint32_t CHUNK = 1u << 16;                  // bytes of text per workgroup in S1 / S3
constexpr uint32_t GROUPS = CHUNK / 16;               // 16-byte groups per chunk
constexpr uint32_t WAVES = SB / 64;                    // waves per workgroup
constexpr uint32_t ITER = GROUPS / SB;
constexpr int32_t ABSENT = INT32_MIN;

enum { ST_NTERM0 = 0, ST_NTERM1 = 1, ST_NONASCII = 2, ST_KSTOP = 3, ST_RUNS0 = 4, ST_RUNS1 = 5, ST_OPS0 = 6, ST_OPS1 = 7, ST_HASCR0 = 8, ST_HASCR1 = 9,
       ST_WORDS = 12 };
enum { SUM_N = 0, SUM_CONS1, SUM_CONS2, SUM_CL1, SUM_CL2, SUM_ENDED, SUM_STARVED, SUM_MISMATCH, SUM_NONASCII, SUM_L1, SUM_L2,
       SUM_OVERFLOW, SUM_WORDS = 16 };

struct FileView {                                     // one file's window, its line index and its per-line records, on the device
    const uint8_t *text;
    uint32_t len, usable, eof, n_chunks;
    uint32_t lead;                                    // the window's text begins here (0 unless xm_strip_set_lead said otherwise): bytes in front are no text
    uint16_t *mask16;
    uint32_t *chunk_cnt, *chunk_base;
    uint32_t *lend, *lnext;                           // cap_lines entries each
    // per-line records (S4), cap_lines entries
    uint32_t *r_name_off, *r_name_len, *r_norm;
    int32_t *r_a, *r_x, *r_nm;
    uint32_t *r_nops, *r_cig_off, *r_cig_len;
    uint8_t *r_flag;
    // skip_repeated_reads: run starts
    uint32_t *blk_cnt, *blk_base, *sel;
    // per-record outputs
    uint32_t *loff, *llen, *nlen;
    uint8_t *lflag;
    // packed CIGAR columns
    int32_t *nm;
    uint32_t *cpos;                                   // operations (+ trailer) per record, then their positions
    uint8_t *cig_cnt;
    uint32_t *cig_tile, *cig_ops;
    uint32_t ops_cap;
};

struct Job {
    FileView f[2];
    uint32_t *state;                                  // ST_WORDS
    uint64_t *summary;                                // SUM_WORDS
    uint32_t cap_lines;                               // max_records + 1
    uint32_t max_records;
    uint32_t xtag0;                                   // 'X' or 'Z'
    uint32_t cigar, skip, paired, keep_halo;
    int32_t *col[4];                                  // as1, xs1, as2, xs2
    uint64_t *unit_bits;
};

__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (uint32_t o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- S1: terminator masks, terminators per chunk, non-ASCII --------------------------------------------------------
// src: where the text is read -- the window's copy in HBM, or (zero-copy, the default since round 6) the host's page-locked staging
// buffer itself, read over the link 16 bytes per lane, 1 KiB per wave instruction; then `copy` is the HBM copy S3-S7 read and this
// kernel writes it beside its masks: the upload IS this kernel, and there is no copy engine launch per piece of a window.
struct MarkArgs {
    const uint8_t *src;
    uint8_t *copy;                                    // null: src is the HBM copy already
    uint16_t *mask16;
    uint32_t *chunk_cnt;
    uint32_t *state;
    uint32_t len, usable, file, chunk0, n_chunks;     // chunks [chunk0, chunk0 + n_chunks) of a window of `len` bytes
    uint32_t lead;                                    // bytes in front of the window's text: no terminators, no non-ASCII test there
};

__global__ void __launch_bounds__(SB) mark_kernel(const MarkArgs a)
{
    if (blockIdx.x >= a.n_chunks) return;
    const uint32_t chunk = a.chunk0 + blockIdx.x;
    const uint32_t n_groups = (a.len + 15u) / 16u;
    uint32_t cnt = 0, hi = 0, any_cr = 0;
    // a wave owns a contiguous quarter of the chunk (16 KiB, 1 KiB per step): S3 then needs no barrier -- its waves take their
    // own bases from the scan of these per-wave counts
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t it = 0; it < ITER; ++it) {
        const uint32_t g = chunk * GROUPS + wave * (GROUPS / WAVES) + it * 64u + lane;
        if (g >= n_groups) break;
        const uint32_t p0 = g * 16u;
        const uint4 v = *reinterpret_cast<const uint4 *>(a.src + p0);         // the buffer is readable past len (padding)
        if (a.copy != nullptr) *reinterpret_cast<uint4 *>(a.copy + p0) = v;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        // bytes equal to '\r' / '\n' as 16-bit masks, without a branch (nearly every wave holds a line end somewhere, so a
        // byte loop behind a test ran for all of them): exact zero-byte test of w ^ c, high bits gathered by a multiply
        uint32_t crm = 0, lfm = 0, him = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            const uint32_t xr = w[q] ^ 0x0D0D0D0Du, xn = w[q] ^ 0x0A0A0A0Au;
            const uint32_t zr = ~(((xr & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xr | 0x7F7F7F7Fu);      // 0x80 where the byte is '\r'
            const uint32_t zn = ~(((xn & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xn | 0x7F7F7F7Fu);
            crm |= ((((zr >> 7) * 0x01020408u) >> 24) & 0xFu) << (4u * q);
            lfm |= ((((zn >> 7) * 0x01020408u) >> 24) & 0xFu) << (4u * q);
            him |= (((((w[q] & 0x80808080u) >> 7) * 0x01020408u) >> 24) & 0xFu) << (4u * q);
        }
        // (a window begun behind a gap: the bytes in front of `lead` are whatever the buffer held)
        const uint32_t text = p0 >= a.lead ? 0xFFFFu : (p0 + 16u <= a.lead ? 0u : 0xFFFFu & ~((1u << (a.lead - p0)) - 1u));
        crm &= text; lfm &= text; him &= text;
        const uint32_t in_len = p0 + 16u <= a.len ? 0xFFFFu : (1u << (a.len - p0)) - 1u;           // p < len (p0 < len here)
        const uint32_t in_use = p0 + 16u <= a.usable ? 0xFFFFu : (p0 < a.usable ? (1u << (a.usable - p0)) - 1u : 0u);
        hi |= (him & in_len) ? 0x80u : 0u;
        any_cr |= crm & in_len;
        // '\r' always ends a line (alone or as the first half of "\r\n"); '\n' unless it is that second half
        const uint32_t prev_cr = ((lfm & 1u) && p0 > a.lead && a.src[p0 - 1u] == 13u) ? 1u : 0u;      // the byte in front of the group
        const uint32_t m = (crm | (lfm & ~((crm << 1) | prev_cr))) & in_use;
        a.mask16[g] = (uint16_t)m;
        cnt += (uint32_t)__popc(m);
    }
    cnt = wave_sum(cnt);
    hi = __any((hi & 0x80808080u) != 0) ? 1u : 0u;
    if (lane == 0u) a.chunk_cnt[chunk * WAVES + wave] = cnt;
    if (hi && lane == 0u) atomicOr(&a.state[ST_NONASCII], 1u);
    if (__any(any_cr != 0u) && lane == 0u) atomicOr(&a.state[ST_HASCR0 + a.file], 1u);     // S3 looks at the text only then
}

// ---- S2: exclusive scan of the chunk counts, one workgroup per file ------------------------------------------------
__global__ void __launch_bounds__(1024) chunk_scan_kernel(const Job job)
{
    const FileView &f = job.f[blockIdx.x];
    __shared__ uint32_t ws[16];
    const uint32_t n_cnt = f.n_chunks * WAVES;                     // one count per wave of S1
    const uint32_t per = (n_cnt + 1023u) / 1024u;
    const uint32_t b = min(threadIdx.x * per, n_cnt), e = min(b + per, n_cnt);
    uint32_t sum = 0;
    for (uint32_t c = b; c < e; ++c) sum += f.chunk_cnt[c];
    const uint32_t incl = wave_scan_incl(sum);
    if ((threadIdx.x & 63u) == 63u) ws[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (uint32_t i = 0; i < 16; ++i) {
        if (i < (threadIdx.x >> 6)) wbase += ws[i];
        total += ws[i];
    }
    uint32_t run = wbase + incl - sum;
    for (uint32_t c = b; c < e; ++c) {
        f.chunk_base[c] = run;
        run += f.chunk_cnt[c];
    }
    if (threadIdx.x == 0) {
        job.state[ST_NTERM0 + blockIdx.x] = total;
        if (blockIdx.x == 0) job.state[ST_KSTOP] = 0xFFFFFFFFu;
    }
}

// ---- S3: where every line ends and the next one starts ------------------------------------------------------------
__global__ void __launch_bounds__(SB) fill_kernel(const Job job)
{
    const FileView &f = job.f[blockIdx.y];
    if (blockIdx.x >= f.n_chunks) return;
    const uint32_t n_groups = (f.len + 15u) / 16u;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t run = f.chunk_base[blockIdx.x * WAVES + wave];            // lines in front of this wave's quarter of the chunk
    const bool has_cr = job.state[ST_HASCR0 + blockIdx.y] != 0u;       // without a CR in the window every terminator is one LF
    for (uint32_t it = 0; it < ITER; ++it) {
        const uint32_t g = blockIdx.x * GROUPS + wave * (GROUPS / WAVES) + it * 64u + lane;
        uint32_t m = g < n_groups ? (uint32_t)f.mask16[g] : 0u;
        const uint32_t c = (uint32_t)__popc(m), incl = wave_scan_incl(c);
        uint32_t j = run + incl - c;
        while (m) {
            const uint32_t p = g * 16u + (uint32_t)(__ffs((int)m) - 1);
            m &= m - 1u;
            if (j < job.cap_lines) {
                const bool crlf = has_cr && f.text[p] == 13u && p + 1u < f.len && f.text[p + 1u] == 10u;
                f.lend[j] = p;
                f.lnext[j] = p + (crlf ? 2u : 1u);
            }
            ++j;
        }
        run += (uint32_t)__shfl((int)incl, 63);
    }
}

// ---- lines of a window as the walk sees them ---------------------------------------------------------------------
struct Lines {
    uint32_t n_terms, count, complete_end;            // count: lines in the window (the unterminated last line of a file too)
};

__device__ __forceinline__ Lines lines_of(const FileView &f, uint32_t n_terms, uint32_t cap_lines)
{
    Lines L;
    L.n_terms = n_terms;
    if (n_terms >= cap_lines) {                       // more lines than the tables hold: the end of the window is never reached
        L.count = n_terms;
        L.complete_end = f.lead;
        return L;
    }
    const uint32_t last_next = n_terms ? f.lnext[n_terms - 1u] : f.lead;
    const bool extra = f.eof && last_next < f.len;
    L.count = n_terms + (extra ? 1u : 0u);
    L.complete_end = extra ? f.len : last_next;
    return L;
}

__device__ __forceinline__ void line_span(const FileView &f, const Lines &L, uint32_t i, uint32_t &start, uint32_t &len)
{
    start = i ? f.lnext[i - 1u] : f.lead;
    len = (i < L.n_terms ? f.lend[i] : f.len) - start;
}

// text[b, e): what follows the last ':' of a field as a plain integer in [-(2^31-1), 2^31-1] (xm_sam.cpp plain_int)
__device__ bool plain_int(const uint8_t *text, uint32_t b, uint32_t e, int32_t &out)
{
    uint32_t i = b;
    bool neg = false;
    if (i < e && (text[i] == '-' || text[i] == '+')) { neg = text[i] == '-'; ++i; }
    if (i >= e || e - i > 10u) return false;
    uint64_t v = 0;
    for (; i < e; ++i) {
        const uint32_t c = text[i];
        if (c < '0' || c > '9') return false;
        v = v * 10u + (uint64_t)(c - '0');
    }
    if (v > 2147483647ull) return false;
    out = neg ? -(int32_t)v : (int32_t)v;
    return true;
}

__device__ __forceinline__ int cigar_op(uint32_t c)
{
    switch (c) {
    case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4;
    case 'H': return 5; case 'P': return 6; case '=': return 7; case 'X': return 8; default: return -1;
    }
}

// re.findall(r'([0-9]+)([MIDNSHPX=])', cigar) over text[b, e): a run of ASCII digits directly followed by an operation
// letter.  out != nullptr: the operations as len << 4 | op; returns how many; big: a length of 2^28 or more was met (that
// operation is left out, as xm_sam.cpp does, and the line is flagged).
__device__ uint32_t cigar_ops(const uint8_t *text, uint32_t b, uint32_t e, uint32_t *out, bool &big)
{
    uint64_t run = 0;
    uint32_t digits = 0, n = 0;
    for (uint32_t p = b; p < e; ++p) {
        const uint32_t c = text[p];
        if (c >= '0' && c <= '9') {
            ++digits;
            if (run < (1ull << 40)) run = run * 10u + (uint64_t)(c - '0');
            continue;
        }
        const int op = cigar_op(c);
        if (op >= 0 && digits > 0) {
            if (run >= (1ull << 28)) big = true;
            else {
                if (out) out[n] = ((uint32_t)run << 4) | (uint32_t)op;
                ++n;
            }
        }
        run = 0;
        digits = 0;
    }
    return n;
}

#ifndef XMS_PARSE_STEP
#define XMS_PARSE_STEP 32
#endif
constexpr uint32_t STEP = XMS_PARSE_STEP;                 // bytes of a line per load step of S4 (the window is padded by as much)

__device__ __forceinline__ void load_step(const uint8_t *at, uint64_t (&ww)[STEP / 8u])
{
#pragma unroll
    for (uint32_t q = 0; q < STEP / 16u; ++q) {
        const uint4 v = *reinterpret_cast<const uint4 *>(at + 16u * q);
        ww[2u * q] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        ww[2u * q + 1u] = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
}

// ---- S4: one lane per line.  fields = line.split() (:103-104), the tag search of get_tag over fields[11:] (:186-190), in
// CIGAR mode the first NM field and the operations of fields[5] (:247-251) ------------------------------------------
__global__ void __launch_bounds__(SB) parse_kernel(const Job job)
{
    const int fi = (int)blockIdx.y;
    const FileView &f = job.f[fi];
    const Lines L0 = lines_of(job.f[0], job.state[ST_NTERM0], job.cap_lines);
    const Lines L1 = lines_of(job.f[1], job.state[ST_NTERM1], job.cap_lines);
    const Lines &L = fi ? L1 : L0;
    // the plain walk only ever looks at the lines both files have; the skipping walk needs every line of the window
    const uint32_t lim = job.skip ? min(L.count, job.cap_lines) : min(min(L0.count, L1.count), job.max_records);
    const uint8_t *text = f.text;
    const uint32_t xtag0 = job.xtag0;
    const bool cigar = job.cigar != 0;
    const uint32_t as_on = cigar ? 0u : 1u, nm_on = cigar ? 1u : 0u;
    for (uint32_t i = blockIdx.x * SB + threadIdx.x; i < lim; i += gridDim.x * SB) {
    uint32_t start, n;
    line_span(f, L, i, start, n);
    const uint32_t end = start + n;
    // A word at a time: the separators of 8 bytes as one 8-bit mask (SWAR range tests, the high bits gathered by a multiply),
    // field starts and ends by shifts of that mask, counts by popcount; the bytes 'A' 'S' 'X'/'Z' 'N' 'M' ':' as masks too, so
    // that "AS" / "XS" / "NM" inside an optional field are bit tests and the only loop left is over the fields that END in the
    // word.  History, per 2 x 128 MB: a byte at a time with branches 487 us (the CU's one scalar unit, shared by its four SIMDs,
    // drowned in exec-mask bookkeeping: 229 M scalar against 105 M vector instructions); the same with selects 349 us; words
    // for the mandatory fields and bytes for the optional ones 459 us at one word per load step (waiting for memory), 210 us at 32
    // bytes per step.
    uint32_t n_tok = 0, n_ends = 0, total = 0, name_off = start, name_len = 0, cig_b = 0, cig_e = 0;
    uint32_t normal = 1, in_tok = 0, prev_ws = 0;
    uint32_t o_ma = 0, o_mx = 0, o_mn = 0, o_colon = start;            // the field that is open: tag letters met so far, last ':' + 1
    uint32_t pA = 0, pX = 0, pN = 0;                                   // the last byte of the word in front was 'A' / 'X' ('Z') / 'N'
    uint32_t n_a = 0, n_x = 0, have_nm = 0, a_b = 0, a_e = 0, x_b = 0, x_e = 0, nm_b = 0, nm_e = 0;
    auto nth_set = [](uint32_t m, uint32_t nth) -> uint32_t {          // position of set bit number nth (< 6) of an 8-bit mask
#pragma unroll
        for (uint32_t j = 0; j < 5u; ++j) m = j < nth ? (m & (m - 1u)) : m;
        return (uint32_t)__ffs((int)m) - 1u;
    };
    auto field_ends = [&](uint32_t at, uint32_t ma, uint32_t mx, uint32_t mn, uint32_t colon) {   // field number n_ends ends in front of `at`
        if (n_ends == 0u) name_len = at - name_off;
        if (n_ends == 5u) cig_e = at;
        if (n_ends >= 11u) {                                            // fields[11:]: where get_tag looks
            if (ma) { if (n_a == 0u) { a_b = colon; a_e = at; } ++n_a; }
            if (mx) { if (n_x == 0u) { x_b = colon; x_e = at; } ++n_x; }
            if (mn && !have_nm) { have_nm = 1u; nm_b = colon; nm_e = at; }    // NM[0]: the first match, no duplicate rule
        }
        ++n_ends;
    };
    auto word = [&](uint32_t wa, uint64_t w) {
        const uint32_t lo = start > wa ? start - wa : 0u, hi = min(end - wa, 8u);
        if (lo >= hi) return;                                           // an empty line inside this word: no valid byte (the shifts below need one)
        const uint64_t H = 0x8080808080808080ull, O = 0x0101010101010101ull;
        // a whole word inside a mandatory field without a byte below 0x21 (the input is ASCII, or the result is thrown away):
        // nothing to learn from it but its length
        if (lo == 0u && hi == 8u && in_tok && n_ends < 11u && !((w - 0x21u * O) & ~w & H)) {
            total += 8u;
            return;
        }
        auto pack = [](uint64_t m) -> uint32_t {                        // the high bit of every byte -> one bit per byte
            const uint32_t l = ((((uint32_t)m >> 7) * 0x01020408u) >> 24) & 0xFu;
            const uint32_t h = ((((uint32_t)(m >> 32) >> 7) * 0x01020408u) >> 24) & 0xFu;
            return l | (h << 4);
        };
        auto eq = [&](uint32_t c) -> uint32_t {                         // bytes equal to c (exact zero-byte test of w ^ c)
            const uint64_t x = w ^ (c * O), m7 = 0x7F7F7F7F7F7F7F7Full;
            return pack(~(((x & m7) + m7) | x | m7));
        };
        // byte in [9, 13] or [28, 32]: Python's str.split() separators; byte == 9: the one '\t'.join() puts back
        const uint64_t t = w | H;                                       // every byte >= 0x80: the subtractions do not borrow
        const uint64_t ge9 = (t - 9u * O) & H, ge10 = (t - 10u * O) & H, ge14 = (t - 14u * O) & H, ge28 = (t - 28u * O) & H,
                       ge33 = (t - 33u * O) & H;
        const uint32_t vm = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        const uint32_t wsm = pack((ge9 & ~ge14) | (ge28 & ~ge33)) & vm, tabm = pack(ge9 & ~ge10) & vm;
        const uint32_t nonws = ~wsm & vm;
        const uint32_t pred = (nonws << 1) | in_tok;                    // bit b: the byte in front of b is part of a field
        const uint32_t starts = nonws & ~pred;
        uint32_t ends = wsm & pred;                                     // a field ends in front of these separators
        // '\t'.join(fields) == line  <=>  exactly one '\t' between fields, nothing in front or behind
        uint32_t bad = (wsm & ~tabm) | (wsm & ((wsm << 1) | prev_ws));
        if (n_tok == 0u) bad |= wsm & ((starts ? (1u << ((uint32_t)__ffs((int)starts) - 1u)) : 256u) - 1u);
        normal &= (uint32_t)(bad == 0u);
        const uint32_t n_st = (uint32_t)__popc(starts);
        if (n_tok == 0u && starts) name_off = wa + (uint32_t)__ffs((int)starts) - 1u;
        if (n_tok <= 5u && n_tok + n_st > 5u) cig_b = wa + nth_set(starts, 5u - n_tok);
        n_tok += n_st;
        total += (uint32_t)__popc(nonws);
        // the tag letters; a pair may straddle two words (never two fields: both bytes are part of one)
        uint32_t as_hit = 0, xs_hit = 0, nm_hit = 0, colons = 0;
        if (n_ends + (uint32_t)__popc(ends) + 1u > 11u) {               // an optional field touches this word
            const uint32_t eS = eq('S') & vm, eA = eq('A') & vm, eX = eq(xtag0) & vm;
            colons = eq(':') & vm;
            xs_hit = eS & ((eX << 1) | pX);
            pX = (eX >> 7) & 1u;
            if (as_on) { as_hit = eS & ((eA << 1) | pA); pA = (eA >> 7) & 1u; }
            if (nm_on) { const uint32_t eN = eq('N') & vm; nm_hit = (eq('M') & vm) & ((eN << 1) | pN); pN = (eN >> 7) & 1u; }
        } else {
            pA = pX = pN = 0u;
        }
        // the fields that end in this word, in order; `from`: where the part of the field inside this word begins
        uint32_t from = 0u, open = in_tok, rest = starts;
        while (ends) {
            const uint32_t e = (uint32_t)__ffs((int)ends) - 1u;
            ends &= ends - 1u;
            uint32_t ma = 0, mx = 0, mn = 0, colon = 0;
            if (!open) {                                                // it began in this word
                from = (uint32_t)__ffs((int)rest) - 1u;
                rest &= rest - 1u;
                colon = wa + from;
            } else {
                ma = o_ma; mx = o_mx; mn = o_mn; colon = o_colon;
            }
            const uint32_t span = ((1u << e) - 1u) & ~((1u << from) - 1u);
            ma |= (uint32_t)((as_hit & span) != 0u);
            mx |= (uint32_t)((xs_hit & span) != 0u);
            mn |= (uint32_t)((nm_hit & span) != 0u);
            const uint32_t c = colons & span;
            if (c) colon = wa + (32u - (uint32_t)__clz((int)c));         // behind the last ':' of the field
            field_ends(wa + e, ma, mx, mn, colon);
            open = 0u;
            from = e;
        }
        // what is left open at the end of the word
        const uint32_t last = 31u - (uint32_t)__clz((int)vm);
        in_tok = (nonws >> last) & 1u;
        prev_ws = (wsm >> last) & 1u;
        if (in_tok) {
            if (!open) {                                                // the field began in this word
                from = (uint32_t)__ffs((int)rest) - 1u;
                o_ma = o_mx = o_mn = 0u;
                o_colon = wa + from;
            }
            const uint32_t span = vm & ~((1u << from) - 1u);
            o_ma |= (uint32_t)((as_hit & span) != 0u);
            o_mx |= (uint32_t)((xs_hit & span) != 0u);
            o_mn |= (uint32_t)((nm_hit & span) != 0u);
            const uint32_t c = colons & span;
            if (c) o_colon = wa + (32u - (uint32_t)__clz((int)c));
        }
    };
    // STEP bytes per load step (16-byte loads, all in flight together): a word per step left the kernel waiting for memory --
    // every step's load touches 64 different cache lines, one per lane
    for (uint32_t base = start & ~(STEP - 1u); base < end; base += STEP) {
        uint64_t ww[STEP / 8u];
        load_step(text + base, ww);
#pragma unroll
        for (uint32_t j = 0; j < STEP / 8u; ++j) {
            const uint32_t wa = base + 8u * j;
            if (wa + 8u > start && wa < end) word(wa, ww[j]);
        }
    }
    if (in_tok) field_ends(end, o_ma, o_mx, o_mn, o_colon);             // the line ends: so does the field that is open
    normal &= prev_ws ^ 1u;
    int32_t a = ABSENT, x = ABSENT, nm = ABSENT;
    uint32_t ex_a = 0, ex_x = 0, n_ops = 0;
    if (n_a >= 1u && !plain_int(text, a_b, a_e, a)) { a = ABSENT; ex_a = 1u; }
    if (n_x >= 1u && !plain_int(text, x_b, x_e, x)) { x = ABSENT; ex_x = 1u; }
    if (n_a > 1u) ex_a = 2u;
    if (n_x > 1u) ex_x = 2u;
    if (cigar && have_nm) {
        if (!plain_int(text, nm_b, nm_e, nm)) { nm = ABSENT; ex_a = 1u; }
        if (n_tok < 6u) {
            ex_a = 3u;                                               // IndexError in the reference: fields[5] does not exist
        } else {
            bool big = false;
            n_ops = cigar_ops(text, cig_b, cig_e, nullptr, big);
            if (big && ex_a == 0u) ex_a = 4u;
        }
    }
    f.r_name_off[i] = name_off;
    f.r_name_len[i] = name_len;
    f.r_norm[i] = n_tok ? total + (n_tok - 1u) : 0u;
    f.r_a[i] = a;
    f.r_x[i] = x;
    f.r_flag[i] = (uint8_t)(((normal && n_tok > 0u) ? XMS_LINE_NORMAL : 0u) | (n_tok == 0u ? XMS_LINE_BLANK : 0u) | (ex_a << 2) |
                            (ex_x << 5));
    if (cigar) {
        f.r_nm[i] = nm;
        f.r_nops[i] = n_ops;
        f.r_cig_off[i] = cig_b;
        f.r_cig_len[i] = n_ops ? cig_e - cig_b : 0u;
    }
    }
}

// 32 bytes per round as five aligned 8-byte words per side, all ten loads in flight together, shifted into place and compared
// under a mask (a loop with an exit after every byte waits one memory latency per byte: 131 us for S6; eight byte loads per
// side at a time: 73 us).  Reads up to 15 bytes past either string: the text buffers are padded.
__device__ bool same_bytes(const uint8_t *a, const uint8_t *b, uint32_t n)
{
    for (uint32_t base = 0; base < n; base += 32u) {
        const uint32_t m = min(n - base, 32u);
        const uintptr_t pa = (uintptr_t)(a + base), pb = (uintptr_t)(b + base);
        const uint64_t *qa = reinterpret_cast<const uint64_t *>(pa & ~(uintptr_t)7), *qb = reinterpret_cast<const uint64_t *>(pb & ~(uintptr_t)7);
        const uint32_t sa = (uint32_t)(pa & 7u) * 8u, sb = (uint32_t)(pb & 7u) * 8u;
        uint64_t wa[5], wb[5];
#pragma unroll
        for (uint32_t j = 0; j < 5u; ++j) { wa[j] = qa[j]; wb[j] = qb[j]; }
        uint64_t diff = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint64_t va = sa ? (wa[j] >> sa) | (wa[j + 1u] << (64u - sa)) : wa[j];
            const uint64_t vb = sb ? (wb[j] >> sb) | (wb[j + 1u] << (64u - sb)) : wb[j];
            const uint32_t have = m > 8u * j ? min(m - 8u * j, 8u) : 0u;      // bytes of this word that belong to the strings
            const uint64_t mask = have >= 8u ? ~0ull : ((1ull << (8u * have)) - 1ull);
            diff |= (va ^ vb) & mask;
        }
        if (diff) return false;
    }
    return true;
}

// xm_sam.cpp same_name: equal lengths and bytes (two blank lines have the same, empty, name)
__device__ __forceinline__ bool same_name(const FileView &fa, uint32_t i, const FileView &fb, uint32_t j)
{
    const uint32_t n = fa.r_name_len[i];
    return n == fb.r_name_len[j] && same_bytes(fa.text + fa.r_name_off[i], fb.text + fb.r_name_off[j], n);
}

// ---- S5: skip_repeated_reads (:110-117).  A file is cut into runs of adjacent lines with one name (a blank line is a run
// of its own); pair k is the first line of run k of both files. ------------------------------------------------------
__device__ __forceinline__ bool is_run_start(const FileView &f, uint32_t i)
{
    if (i == 0u) return true;
    if ((f.r_flag[i] | f.r_flag[i - 1u]) & XMS_LINE_BLANK) return true;
    return !same_name(f, i, f, i - 1u);
}

__global__ void __launch_bounds__(SB) start_count_kernel(const Job job)
{
    const int fi = (int)blockIdx.y;
    const FileView &f = job.f[fi];
    const Lines L = lines_of(f, job.state[ST_NTERM0 + fi], job.cap_lines);
    const uint32_t lim = min(L.count, job.cap_lines);
    __shared__ uint32_t red[SB / 64];
    for (uint32_t blk = blockIdx.x; blk * SB < lim; blk += gridDim.x) {
        const uint32_t i = blk * SB + threadIdx.x;
        const bool st = i < lim && is_run_start(f, i);
        const uint32_t c = (uint32_t)__popcll(__ballot(st));
        if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int w = 0; w < SB / 64; ++w) t += red[w];
            f.blk_cnt[blk] = t;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(1024) start_scan_kernel(const Job job)
{
    const int fi = (int)blockIdx.x;
    const FileView &f = job.f[fi];
    const Lines L = lines_of(f, job.state[ST_NTERM0 + fi], job.cap_lines);
    const uint32_t n_blk = (min(L.count, job.cap_lines) + SB - 1u) / SB;
    __shared__ uint32_t ws[16];
    const uint32_t per = (n_blk + 1023u) / 1024u;
    const uint32_t b = min(threadIdx.x * per, n_blk), e = min(b + per, n_blk);
    uint32_t sum = 0;
    for (uint32_t c = b; c < e; ++c) sum += f.blk_cnt[c];
    const uint32_t incl = wave_scan_incl(sum);
    if ((threadIdx.x & 63u) == 63u) ws[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (uint32_t w = 0; w < 16; ++w) {
        if (w < (threadIdx.x >> 6)) wbase += ws[w];
        total += ws[w];
    }
    uint32_t run = wbase + incl - sum;
    for (uint32_t c = b; c < e; ++c) {
        f.blk_base[c] = run;
        run += f.blk_cnt[c];
    }
    if (threadIdx.x == 0) job.state[ST_RUNS0 + fi] = total;
}

__global__ void __launch_bounds__(SB) start_fill_kernel(const Job job)
{
    const int fi = (int)blockIdx.y;
    const FileView &f = job.f[fi];
    const Lines L = lines_of(f, job.state[ST_NTERM0 + fi], job.cap_lines);
    const uint32_t lim = min(L.count, job.cap_lines);
    __shared__ uint32_t ws[SB / 64];
    for (uint32_t blk = blockIdx.x; blk * SB < lim; blk += gridDim.x) {
        const uint32_t i = blk * SB + threadIdx.x;
        const bool st = i < lim && is_run_start(f, i);
        const uint64_t m = __ballot(st);
        const uint32_t lane = threadIdx.x & 63u;
        if (lane == 0u) ws[threadIdx.x >> 6] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t wb = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wb += ws[w];
        if (st) f.sel[f.blk_base[blk] + wb + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
    }
}

// what the walk may pair: records (plain walk: lines) of each file, and the most the block may hold
struct Walk {
    Lines L[2];
    uint32_t R[2];                                    // pairable entries per file: runs (skipping walk) or lines
    uint32_t lim;
    bool whole[2];
};

__device__ __forceinline__ Walk walk_of(const Job &job)
{
    Walk W;
    for (int f = 0; f < 2; ++f) {
        W.L[f] = lines_of(job.f[f], job.state[ST_NTERM0 + f], job.cap_lines);
        W.R[f] = job.skip ? job.state[ST_RUNS0 + f] : W.L[f].count;
        W.whole[f] = job.f[f].eof && W.L[f].n_terms < job.cap_lines && W.L[f].complete_end == job.f[f].len;
    }
    W.lim = min(min(W.R[0], W.R[1]), job.max_records);
    return W;
}

// ---- S6: one lane per record ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SB) pair_kernel(const Job job)
{
    const Walk W = walk_of(job);
    for (uint32_t k = blockIdx.x * SB + threadIdx.x; (k & ~63u) < W.lim; k += gridDim.x * SB) {      // wave-uniform bound
    bool unit = false;
    if (k < W.lim) {
        uint32_t i[2];
        for (int f = 0; f < 2; ++f) i[f] = job.skip ? job.f[f].sel[k] : k;
        const uint32_t fl0 = job.f[0].r_flag[i[0]], fl1 = job.f[1].r_flag[i[1]];
        const bool blank = ((fl0 | fl1) & XMS_LINE_BLANK) != 0u;
        const bool same = same_name(job.f[0], i[0], job.f[1], i[1]);
        // the skipping walk also stops at a run that reaches the end of a window which is not the end of its file (the run
        // may go on in the next window) -- after blank and mismatch, in that order (xm_sam.cpp)
        const bool open_run = job.skip && ((k + 1u == W.R[0] && !W.whole[0]) || (k + 1u == W.R[1] && !W.whole[1]));
        if (blank || !same || open_run) atomicMin(&job.state[ST_KSTOP], k);
        // the unit rule (:402-405): the name equals the name of the record in front (file 1's names, as the reference)
        if (!job.paired) unit = true;
        else if (k > 0u) unit = same_name(job.f[0], i[0], job.f[0], job.skip ? job.f[0].sel[k - 1u] : k - 1u);
        for (int f = 0; f < 2; ++f) {
            const FileView &v = job.f[f];
            uint32_t st, ln;
            line_span(v, W.L[f], i[f], st, ln);
            v.loff[k] = st;
            v.llen[k] = ln;
            v.nlen[k] = v.r_norm[i[f]];
            uint32_t fl = f ? fl1 : fl0;
            if (f == 0 && !blank && !same) fl |= XMS_LINE_MISMATCH;
            v.lflag[k] = (uint8_t)fl;
            job.col[2 * f][k] = v.r_a[i[f]];
            job.col[2 * f + 1][k] = v.r_x[i[f]];
            if (job.cigar) {
                const uint32_t n_ops = v.r_nops[i[f]];
                v.nm[k] = v.r_nm[i[f]];
                v.cpos[k] = n_ops + (n_ops >= 255u ? 1u : 0u);
            }
        }
    }
    const uint64_t word = __ballot(unit);
    if ((threadIdx.x & 63u) == 0u) job.unit_bits[k >> 6] = word;
    }
}

// ---- S7: the packed CIGAR columns of include/xenomapper_hip.h ------------------------------------------------------
__global__ void __launch_bounds__(1024) cig_scan_kernel(const Job job)
{
    const int fi = (int)blockIdx.x;
    const FileView &f = job.f[fi];
    const Walk W = walk_of(job);
    const uint32_t n = W.lim;
    __shared__ uint32_t ws[16];
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t b = min(threadIdx.x * per, n), e = min(b + per, n);
    uint32_t sum = 0;
    for (uint32_t c = b; c < e; ++c) sum += f.cpos[c];
    const uint32_t incl = wave_scan_incl(sum);
    if ((threadIdx.x & 63u) == 63u) ws[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (uint32_t w = 0; w < 16; ++w) {
        if (w < (threadIdx.x >> 6)) wbase += ws[w];
        total += ws[w];
    }
    uint32_t run = wbase + incl - sum;
    for (uint32_t c = b; c < e; ++c) {
        const uint32_t mine = f.cpos[c];
        f.cpos[c] = run;
        run += mine;
    }
    if (threadIdx.x == 0) {
        f.cpos[n] = total;
        job.state[ST_OPS0 + fi] = total;
    }
}

__global__ void __launch_bounds__(SB) cig_write_kernel(const Job job)
{
    const int fi = (int)blockIdx.y;
    const FileView &f = job.f[fi];
    const Walk W = walk_of(job);
    const uint32_t total = f.cpos[W.lim];
    if (total > f.ops_cap) return;                      // cannot happen (an operation takes two bytes of text); S8 reports it
    for (uint32_t k = blockIdx.x * SB + threadIdx.x; k < W.lim; k += gridDim.x * SB) {
    const uint32_t i = job.skip ? f.sel[k] : k;
    const uint32_t pos = f.cpos[k], n_ops = f.r_nops[i];
    if (n_ops) {
        bool big = false;
        const uint32_t b = f.r_cig_off[i];
        (void)cigar_ops(f.text, b, b + f.r_cig_len[i], f.cig_ops + pos, big);
        if (n_ops >= 255u) f.cig_ops[pos + n_ops] = (n_ops << 4) | 15u;     // the trailer: the record's begin from its end
    }
    f.cig_cnt[k] = (uint8_t)min(n_ops, 255u);
    if ((k & 255u) == 0u) f.cig_tile[k >> 8] = pos;
    if (k + 1u == W.lim) f.cig_tile[(W.lim + 255u) >> 8] = total;
    }
}

// ---- S8: the outcome of the walk, as xm_sam.cpp parse_common reports it ------------------------------------------
__global__ void summary_kernel(const Job job)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Walk W = walk_of(job);
    const uint32_t lim = W.lim;
    const uint32_t ks = min(job.state[ST_KSTOP], lim);
    uint32_t ended = 0, starved = 0, overflow = 0;
    int64_t mismatch = -1;
    if (job.skip && (W.L[0].n_terms >= job.cap_lines || W.L[1].n_terms >= job.cap_lines)) overflow = 1;   // more lines than the tables hold
    if (job.cigar && (job.state[ST_OPS0] > job.f[0].ops_cap || job.state[ST_OPS1] > job.f[1].ops_cap)) overflow = 1;
    if (ks < lim) {
        const uint32_t fl = job.f[0].lflag[ks] | job.f[1].lflag[ks];
        if (fl & XMS_LINE_BLANK) ended = 1;
        else if (job.f[0].lflag[ks] & XMS_LINE_MISMATCH) mismatch = (int64_t)ks;
        else starved = 1;                                            // an open run
    } else if (ks < job.max_records) {
        const bool end0 = job.skip ? (W.R[0] == lim && W.whole[0]) : (ks >= W.L[0].count && W.whole[0]);
        const bool end1 = job.skip ? (W.R[1] == lim && W.whole[1]) : (ks >= W.L[1].count && W.whole[1]);
        if (end0 || end1) ended = 1; else starved = 1;
    }
    uint64_t cons[2], cl[2];
    const bool halo = job.keep_halo && ks > 0u && !ended && mismatch < 0;
    for (int f = 0; f < 2; ++f) {
        // the line the next window starts with: the last yielded record's (halo) or the one the walk stopped at
        uint32_t i;
        if (halo) i = job.skip ? job.f[f].sel[ks - 1u] : ks - 1u;
        else if (job.skip) i = ks < W.R[f] ? job.f[f].sel[ks] : W.L[f].count;
        else i = ks;
        cons[f] = i < W.L[f].count ? (uint64_t)(i ? job.f[f].lnext[i - 1u] : job.f[f].lead) : (uint64_t)W.L[f].complete_end;
        cl[f] = min(i, W.L[f].count);
    }
    uint64_t *s = job.summary;
    s[SUM_N] = ks;
    s[SUM_CONS1] = cons[0];
    s[SUM_CONS2] = cons[1];
    s[SUM_CL1] = cl[0];
    s[SUM_CL2] = cl[1];
    s[SUM_ENDED] = ended;
    s[SUM_STARVED] = starved;
    s[SUM_MISMATCH] = (uint64_t)mismatch;
    s[SUM_NONASCII] = job.state[ST_NONASCII];
    s[SUM_L1] = W.L[0].count;
    s[SUM_L2] = W.L[1].count;
    s[SUM_OVERFLOW] = overflow;
}

// ---- G: the six outputs gathered on the device (xm_strip_fetch_bins; the BAM path's twin is in xm_bamdev.hip) -------------------
// With the window's text, the line table and the bins' unit lists all on the device, the text of every output file is a function of
// device data: for the bin's units in input order, the unit's lines (xenomapper.py:332-350, :423-448, :521-550) -- each line as the
// reference prints it, '\t'.join(fields) + '\n', which IS the input line without its terminator whenever the line already has
// single tabs between its fields (XMS_LINE_NORMAL: nearly every line of real SAM; a window with a wanted line that needs re-joining
// is left to the host writer, status 3).  G1 sizes every unit, the size scan places it (units are in bin order: the scan is the
// layout of the six texts back to back), G2 copies the lines, G3 (xm_gather.h) moves the stream to page-locked memory.
__global__ void __launch_bounds__(256)
sam_unit_size_kernel(const uint32_t *__restrict__ idx, const unsigned long long *__restrict__ off, uint32_t n_units, uint32_t n_records, int paired,
                     uint32_t sink_mask, const uint32_t *__restrict__ llen1, const uint32_t *__restrict__ llen2,
                     const uint8_t *__restrict__ lflag1, const uint8_t *__restrict__ lflag2, uint32_t *__restrict__ usize,
                     unsigned long long *__restrict__ total64, uint32_t *__restrict__ not_normal)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    uint32_t s = 0, odd = 0;
    if (p < n_units) {
        const uint32_t i = idx[p], files = files_of_bin(bin_of_place(p, off), sink_mask);
        if (i < n_records && (!paired || i > 0u)) {
            if (files & 1u) {
                s += llen1[i] + 1u; odd |= ~(uint32_t)lflag1[i] & XMS_LINE_NORMAL;
                if (paired) { s += llen1[i - 1u] + 1u; odd |= ~(uint32_t)lflag1[i - 1u] & XMS_LINE_NORMAL; }
            }
            if (files & 2u) {
                s += llen2[i] + 1u; odd |= ~(uint32_t)lflag2[i] & XMS_LINE_NORMAL;
                if (paired) { s += llen2[i - 1u] + 1u; odd |= ~(uint32_t)lflag2[i - 1u] & XMS_LINE_NORMAL; }
            }
        }
        usize[p] = s;
    }
    unsigned long long t = s;
    for (int d = 32; d; d >>= 1) t += __shfl_xor(t, d, 64);
    if ((threadIdx.x & 63u) == 0u && t) atomicAdd(total64, t);
    if (__any(odd != 0u) && (threadIdx.x & 63u) == 0u) atomicOr(not_normal, 1u);
}

struct __attribute__((packed, aligned(1))) U64Any { uint64_t v; };

// G2: a wave per (unit, line of the unit); the lanes copy the line eight bytes each per step (neither side is aligned), lane 0 adds '\n'
__global__ void __launch_bounds__(256)
sam_line_copy_kernel(const uint8_t *__restrict__ text1, const uint8_t *__restrict__ text2, const uint32_t *__restrict__ loff1,
                     const uint32_t *__restrict__ loff2, const uint32_t *__restrict__ llen1, const uint32_t *__restrict__ llen2,
                     const uint32_t *__restrict__ idx, const unsigned long long *__restrict__ off, uint32_t n_units, uint32_t n_records, int paired,
                     uint32_t sink_mask, const uint32_t *__restrict__ usize, const uint32_t *__restrict__ uplace,
                     uint8_t *__restrict__ out, uint32_t out_cap)
{
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t p = paired ? w >> 1 : w, j = paired ? w & 1u : 0u;
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
        const uint32_t *ll = f ? llen2 : llen1;
        const uint32_t first = paired ? ll[i - 1u] + 1u : 0u, len = ll[r];
        const uint8_t *src = (f ? text2 : text1) + (f ? loff2 : loff1)[r];
        uint8_t *dst = out + at + (j ? first : 0u);
        uint32_t k = lane * 8u;
        for (; k + 8u <= len; k += 512u) reinterpret_cast<U64Any *>(dst + k)->v = reinterpret_cast<const U64Any *>(src + k)->v;
        if (k < len) for (uint32_t q = k; q < len && q < k + 8u; ++q) dst[q] = src[q];
        if (lane == 0u) dst[len] = (uint8_t)'\n';
        at += first + ll[i] + 1u;                                                    // behind file 1's lines of the unit: file 2's
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct PerFile {
    char *h_text = nullptr;                            // page-locked staging (+ 64 bytes: S1 reads whole 16-byte words)
    uint64_t lead = 0;                                 // bytes in front of this window's text (xm_strip_set_lead; 0 unless said)
    uint8_t *d_text = nullptr;
    uint16_t *d_mask = nullptr;
    uint32_t *d_chunk_cnt = nullptr, *d_chunk_base = nullptr;
    uint32_t *d_lend = nullptr, *d_lnext = nullptr;
    uint32_t *d_r_name_off = nullptr, *d_r_name_len = nullptr, *d_r_norm = nullptr, *d_r_nops = nullptr, *d_r_cig_off = nullptr,
             *d_r_cig_len = nullptr;
    int32_t *d_r_a = nullptr, *d_r_x = nullptr, *d_r_nm = nullptr;
    uint8_t *d_r_flag = nullptr;
    uint32_t *d_blk_cnt = nullptr, *d_blk_base = nullptr, *d_sel = nullptr;
    uint32_t *d_loff = nullptr, *d_llen = nullptr, *d_nlen = nullptr;
    uint8_t *d_lflag = nullptr;
    uint32_t *h_loff = nullptr, *h_llen = nullptr, *h_nlen = nullptr;
    uint8_t *h_lflag = nullptr;
    int32_t *d_nm = nullptr;
    uint32_t *d_cpos = nullptr, *d_cig_tile = nullptr, *d_cig_ops = nullptr;
    uint8_t *d_cig_cnt = nullptr;
    uint64_t ops_cap = 0;
};

struct Slot {
    uint64_t window_cap = 0, record_cap = 0;
    bool cigar_ready = false;                          // the CIGAR-only arrays exist for the current capacities
    PerFile pf[2];
    int32_t *d_col[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t *d_bits = nullptr;
    // classify outputs
    uint8_t *d_code = nullptr, *d_bins4 = nullptr, *h_code = nullptr;
    uint32_t *d_idx = nullptr, *h_idx = nullptr, *d_range = nullptr;
    uint64_t *d_off_counts = nullptr, *h_off_counts = nullptr;     // 8 + 64 words (+ the range flag's word on the host side)
    uint32_t *d_state = nullptr;
    uint64_t *d_summary = nullptr, *h_summary = nullptr;
    // xm_strip_fetch_bins: per unit the bytes of its lines and where they go, the six outputs as one stream on the device and in
    // page-locked host memory, a few words of state; its copy to the host runs on a stream of its own
    uint32_t *d_usize = nullptr, *d_uplace = nullptr, *d_upart = nullptr, *d_gstate = nullptr, *h_gstate = nullptr;
    uint8_t *d_out = nullptr, *h_out = nullptr;
    uint64_t out_cap = 0, out_records = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_filled = nullptr, ev_out = nullptr;
    bool out_issued = false, classified = false;
    uint32_t held[2] = {0, 0};                         // chunks [0, held) of a window begun with xm_strip_begin_behind wait for the run (S1 needs the lead)
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    uint64_t uploaded[2] = {0, 0};                     // bytes of the staged windows already on their way (xm_strip_upload)
    uint32_t marked[2] = {0, 0};                       // zero-copy: chunks of the window S1 has been launched for already
    bool state_cleared = false;                        // the window's state words were zeroed (before its first S1 launch)
    bool upload_timed = false;
    int last_score_mode = -1;                          // of the block the slot holds
};

}  // namespace

struct xm_strip {
    xm_ctx *ctx = nullptr;
    int device = 0;
    Slot slot[XMS_SLOTS];
    std::mutex error_lock;                             // the two slots are driven by two threads
    std::string last_error;
};

namespace {

int fail(xm_strip *s, hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    if (s) {
        std::lock_guard<std::mutex> hold(s->error_lock);
        s->last_error = buf;
    }
    (void)hipGetLastError();        // reported here: a later launch check on this thread must not find it again
    return e == hipErrorOutOfMemory ? XM_ERR_OOM : XM_ERR_HIP;
}

#define XMS_HIP(s, call)                                   \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return fail((s), e_, #call); \
    } while (0)

#define XMS_TRY(expr)                  \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != XM_OK) return rc_;  \
    } while (0)

template <typename T> void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }
template <typename T> void hfree(T *&p) { if (p) { (void)xmpin::host_free(p); p = nullptr; } }

template <typename T> int dalloc(xm_strip *s, T *&p, size_t count)
{
    dfree(p);
    XMS_HIP(s, hipMalloc((void **)&p, std::max<size_t>(count, 16) * sizeof(T)));
    return XM_OK;
}
template <typename T> int halloc(xm_strip *s, T *&p, size_t count)
{
    hfree(p);
    XMS_HIP(s, xmpin::host_malloc((void **)&p, std::max<size_t>(count, 16) * sizeof(T)));
    return XM_OK;
}

void free_slot(Slot &sl)
{
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        hfree(q.h_text); dfree(q.d_text); dfree(q.d_mask); dfree(q.d_chunk_cnt); dfree(q.d_chunk_base);
        dfree(q.d_lend); dfree(q.d_lnext);
        dfree(q.d_r_name_off); dfree(q.d_r_name_len); dfree(q.d_r_norm); dfree(q.d_r_nops); dfree(q.d_r_cig_off); dfree(q.d_r_cig_len);
        dfree(q.d_r_a); dfree(q.d_r_x); dfree(q.d_r_nm); dfree(q.d_r_flag);
        dfree(q.d_blk_cnt); dfree(q.d_blk_base); dfree(q.d_sel);
        dfree(q.d_loff); dfree(q.d_llen); dfree(q.d_nlen); dfree(q.d_lflag);
        hfree(q.h_loff); hfree(q.h_llen); hfree(q.h_nlen); hfree(q.h_lflag);
        dfree(q.d_nm); dfree(q.d_cpos); dfree(q.d_cig_tile); dfree(q.d_cig_ops); dfree(q.d_cig_cnt);
        q.ops_cap = 0;
    }
    for (int c = 0; c < 4; ++c) dfree(sl.d_col[c]);
    dfree(sl.d_bits); dfree(sl.d_code); dfree(sl.d_bins4); dfree(sl.d_idx);
    hfree(sl.h_code); hfree(sl.h_idx);
    dfree(sl.d_usize); dfree(sl.d_uplace); dfree(sl.d_upart); dfree(sl.d_out); hfree(sl.h_out);
    sl.out_cap = sl.out_records = 0;
    sl.window_cap = sl.record_cap = 0;
    sl.cigar_ready = false;
}

int grow_window(xm_strip *s, Slot &sl, uint64_t bytes)
{
    if (bytes <= sl.window_cap) return XM_OK;
    const uint64_t cap = (bytes + CHUNK - 1) / CHUNK * CHUNK;
    const size_t n_chunks = (size_t)(cap / CHUNK);
    sl.window_cap = 0;
    sl.cigar_ready = false;
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        XMS_TRY(halloc(s, q.h_text, (size_t)cap + 64));
        q.lead = 0;
        XMS_TRY(dalloc(s, q.d_text, (size_t)cap + 256));              // S1 / S4 read whole 16-byte words / load steps
        XMS_TRY(dalloc(s, q.d_mask, (size_t)cap / 16 + 16));
        XMS_TRY(dalloc(s, q.d_chunk_cnt, n_chunks * WAVES));
        XMS_TRY(dalloc(s, q.d_chunk_base, n_chunks * WAVES));
    }
    sl.window_cap = cap;
    return XM_OK;
}

int grow_records(xm_strip *s, Slot &sl, uint64_t records)
{
    if (records <= sl.record_cap) return XM_OK;
    const size_t n = (size_t)records + 64, lines = (size_t)records + 1;
    sl.record_cap = 0;
    sl.cigar_ready = false;
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        XMS_TRY(dalloc(s, q.d_lend, lines)); XMS_TRY(dalloc(s, q.d_lnext, lines));
        XMS_TRY(dalloc(s, q.d_r_name_off, lines)); XMS_TRY(dalloc(s, q.d_r_name_len, lines)); XMS_TRY(dalloc(s, q.d_r_norm, lines));
        XMS_TRY(dalloc(s, q.d_r_a, lines)); XMS_TRY(dalloc(s, q.d_r_x, lines)); XMS_TRY(dalloc(s, q.d_r_flag, lines + 64));
        XMS_TRY(dalloc(s, q.d_blk_cnt, lines / SB + 2)); XMS_TRY(dalloc(s, q.d_blk_base, lines / SB + 2)); XMS_TRY(dalloc(s, q.d_sel, lines));
        XMS_TRY(dalloc(s, q.d_loff, n)); XMS_TRY(dalloc(s, q.d_llen, n)); XMS_TRY(dalloc(s, q.d_nlen, n)); XMS_TRY(dalloc(s, q.d_lflag, n));
        XMS_TRY(halloc(s, q.h_loff, n)); XMS_TRY(halloc(s, q.h_llen, n)); XMS_TRY(halloc(s, q.h_nlen, n)); XMS_TRY(halloc(s, q.h_lflag, n));
    }
    for (int c = 0; c < 4; ++c) XMS_TRY(dalloc(s, sl.d_col[c], n));
    XMS_TRY(dalloc(s, sl.d_bits, n / 64 + 2));
    XMS_TRY(dalloc(s, sl.d_code, n));
    XMS_TRY(dalloc(s, sl.d_bins4, (size_t)XM_BINS4_BYTES(records) + 16));
    XMS_TRY(dalloc(s, sl.d_idx, n));
    XMS_TRY(halloc(s, sl.h_code, n));
    XMS_TRY(halloc(s, sl.h_idx, n));
    sl.record_cap = records;
    return XM_OK;
}

// the arrays only --cigar_scores needs, sized for the slot's current capacities (allocated on the first CIGAR run)
int ensure_cigar(xm_strip *s, Slot &sl)
{
    if (sl.cigar_ready) return XM_OK;
    const size_t n = (size_t)sl.record_cap + 64, lines = (size_t)sl.record_cap + 1;
    // an operation is at least two bytes of text ("1M"); one trailer word per record at most
    const uint64_t ops_cap = std::min<uint64_t>(sl.window_cap / 2 + sl.record_cap + 16, 0xFFFFFFF0ull);
    for (int f = 0; f < 2; ++f) {
        PerFile &q = sl.pf[f];
        XMS_TRY(dalloc(s, q.d_r_nm, lines)); XMS_TRY(dalloc(s, q.d_r_nops, lines)); XMS_TRY(dalloc(s, q.d_r_cig_off, lines));
        XMS_TRY(dalloc(s, q.d_r_cig_len, lines));
        XMS_TRY(dalloc(s, q.d_nm, n)); XMS_TRY(dalloc(s, q.d_cpos, n)); XMS_TRY(dalloc(s, q.d_cig_cnt, n));
        XMS_TRY(dalloc(s, q.d_cig_tile, n / 256 + 4)); XMS_TRY(dalloc(s, q.d_cig_ops, (size_t)ops_cap + 16));
        q.ops_cap = ops_cap;
    }
    sl.cigar_ready = true;
    return XM_OK;
}

// The text crosses the link inside S1 (mark_kernel reads the page-locked staging buffer and writes the HBM copy) instead of by a copy
// of its own per piece.  XM_STRIP_ZEROCOPY=0: hipMemcpyAsync per piece, S1 on the copy (rounds 4 and 5).
bool zero_copy_text()
{
    static const bool on = [] { const char *v = getenv("XM_STRIP_ZEROCOPY"); return !(v && v[0] == '0'); }();
    return on;
}

// S1 for chunks [marked, upto_chunk) of one file's window, with `len` / `usable` as the kernel is to see them; last: upto_chunk is
// the window's chunk count (else the chunk holding the last staged byte is left for later)
int mark_range(xm_strip *s, Slot &sl, int file, uint64_t len, uint64_t usable, uint32_t chunk0, uint32_t upto_chunk);

int mark_chunks(xm_strip *s, Slot &sl, int file, uint64_t len, uint64_t usable, uint32_t upto_chunk, bool last)
{
    if (!last && upto_chunk > 0) --upto_chunk;
    if (upto_chunk <= sl.marked[file]) return XM_OK;
    XMS_TRY(mark_range(s, sl, file, len, usable, sl.marked[file], upto_chunk));
    sl.marked[file] = upto_chunk;
    return XM_OK;
}

int mark_range(xm_strip *s, Slot &sl, int file, uint64_t len, uint64_t usable, uint32_t chunk0, uint32_t upto_chunk)
{
    if (upto_chunk <= chunk0) return XM_OK;
    if (!sl.state_cleared) {
        XMS_HIP(s, hipMemsetAsync(sl.d_state, 0, ST_WORDS * sizeof(uint32_t), sl.stream));
        sl.state_cleared = true;
    }
    PerFile &q = sl.pf[file];
    MarkArgs a;
    a.src = zero_copy_text() ? reinterpret_cast<const uint8_t *>(q.h_text) : q.d_text;
    a.copy = zero_copy_text() ? q.d_text : nullptr;
    a.mask16 = q.d_mask; a.chunk_cnt = q.d_chunk_cnt; a.state = sl.d_state;
    a.len = (uint32_t)len; a.usable = (uint32_t)usable; a.file = (uint32_t)file;
    a.chunk0 = chunk0; a.n_chunks = upto_chunk - chunk0;
    a.lead = (uint32_t)q.lead;
    mark_kernel<<<a.n_chunks, SB, 0, sl.stream>>>(a);
    return XM_OK;
}

}  // namespace

extern "C" {

int xms_abi_version(void) { return XMS_ABI_VERSION; }

int xm_strip_create(xm_ctx *ctx, int device_id, xm_strip **out)
{
    if (!ctx || !out) return XM_ERR_INVALID_ARG;
    xm_strip *s = new (std::nothrow) xm_strip();
    if (!s) return XM_ERR_OOM;
    s->ctx = ctx;
    s->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    for (int k = 0; k < XMS_SLOTS && e == hipSuccess; ++k) {
        Slot &sl = s->slot[k];
        // (the slots' streams work side by side since a window is read -- and crosses the link, S1 -- while the one in front is
        // stripped: each on a hardware queue of its own, tried like the copy stream's)
        hipStream_t before[XMS_SLOTS];
        for (int j = 0; j < k; ++j) before[j] = s->slot[j].stream;
        e = create_copy_stream(&sl.stream, before, k);
        for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreate(&sl.ev[i]);
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_state, ST_WORDS * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_range, 4 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_summary, SUM_WORDS * sizeof(uint64_t));
        if (e == hipSuccess) e = xmpin::host_malloc((void **)&sl.h_summary, SUM_WORDS * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_off_counts, 72 * sizeof(uint64_t));
        if (e == hipSuccess) e = xmpin::host_malloc((void **)&sl.h_off_counts, 74 * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_gstate, 16 * sizeof(uint32_t));
        if (e == hipSuccess) e = xmpin::host_malloc((void **)&sl.h_gstate, 16 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_filled, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_out, hipEventBlockingSync | hipEventDisableTiming);
    }
    if (e == hipSuccess) {
        // one copy stream for all slots, one that runs beside every slot's compute stream (xm_gather.h: tried, not assumed)
        hipStream_t compute[XMS_SLOTS];
        for (int k = 0; k < XMS_SLOTS; ++k) compute[k] = s->slot[k].stream;
        e = create_copy_stream(&s->slot[0].copy_stream, compute, XMS_SLOTS);
        for (int k = 1; k < XMS_SLOTS; ++k) s->slot[k].copy_stream = s->slot[0].copy_stream;
    }
    if (e != hipSuccess) {
        xm_strip_destroy(s);
        return e == hipErrorOutOfMemory ? XM_ERR_OOM : XM_ERR_HIP;
    }
    *out = s;
    return XM_OK;
}

int xm_strip_destroy(xm_strip *s)
{
    if (!s) return XM_ERR_INVALID_ARG;
    (void)hipSetDevice(s->device);
    for (int k = 0; k < XMS_SLOTS; ++k) {
        Slot &sl = s->slot[k];
        if (sl.stream) {
            (void)hipStreamSynchronize(sl.stream);
            (void)xm_workspace_release(s->ctx, sl.stream);             // the context must not keep the handle of a dead stream
        }
        if (sl.copy_stream) (void)hipStreamSynchronize(sl.copy_stream);
        free_slot(sl);
        dfree(sl.d_state); dfree(sl.d_range); dfree(sl.d_summary); dfree(sl.d_off_counts);
        hfree(sl.h_summary); hfree(sl.h_off_counts);
        dfree(sl.d_gstate); hfree(sl.h_gstate);
        if (sl.ev_filled) (void)hipEventDestroy(sl.ev_filled);
        if (sl.ev_out) (void)hipEventDestroy(sl.ev_out);
        if (k == XMS_SLOTS - 1 && sl.copy_stream) (void)hipStreamDestroy(sl.copy_stream);    // shared by the slots: once, behind all
        for (int i = 0; i < 3; ++i)
            if (sl.ev[i]) (void)hipEventDestroy(sl.ev[i]);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    delete s;
    return XM_OK;
}

int xm_strip_reserve(xm_strip *s, int slot, uint64_t window_bytes, uint64_t max_records)
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || window_bytes > XMS_MAX_WINDOW || max_records == 0 ||
        max_records >= 0xFFFFFF00ull)
        return XM_ERR_INVALID_ARG;
    XMS_HIP(s, hipSetDevice(s->device));
    Slot &sl = s->slot[slot];
    XMS_HIP(s, hipStreamSynchronize(sl.stream));
    if (sl.out_issued) XMS_HIP(s, hipEventSynchronize(sl.ev_out));            // THIS slot's last copy (the copy stream also carries the other slots')
    // a window begins here: whatever an abandoned one (a read that failed half way, a run that was never issued) had
    // sent is forgotten, or the next upload from offset 0 would be refused for ever
    sl.uploaded[0] = sl.uploaded[1] = 0;
    sl.marked[0] = sl.marked[1] = 0;
    sl.pf[0].lead = sl.pf[1].lead = 0;
    sl.held[0] = sl.held[1] = 0;
    sl.state_cleared = false;
    sl.upload_timed = false;
    // a failed growth leaves the capacity at 0 and the pointers freed or null: the next reserve allocates afresh
    XMS_TRY(grow_window(s, sl, std::max<uint64_t>(window_bytes, 1)));
    XMS_TRY(grow_records(s, sl, max_records));
    return XM_OK;
}

char *xm_strip_staging(xm_strip *s, int slot, int file)
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || file < 0 || file > 1) return nullptr;
    return s->slot[slot].pf[file].h_text;
}

int xm_strip_begin_behind(xm_strip *s, int slot, int file, uint64_t room)
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || file < 0 || file > 1 || room % CHUNK != 0) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    if (room > sl.window_cap) return XM_ERR_INVALID_ARG;
    sl.pf[file].lead = 0;
    sl.uploaded[file] = room;                               // uploads go on from here; [0, room) arrives with xm_strip_set_lead
    // S1 of the chunk at `room` looks at the byte in front of it (a "\r\n" cut there): that chunk waits for the run as well
    sl.held[file] = (uint32_t)(room / CHUNK) + 1u;
    sl.marked[file] = sl.held[file];
    return XM_OK;
}

int xm_strip_set_lead(xm_strip *s, int slot, int file, uint64_t lead)
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || file < 0 || file > 1) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    if (lead > sl.window_cap || (sl.held[file] != 0 && lead > (uint64_t)(sl.held[file] - 1u) * CHUNK)) return XM_ERR_INVALID_ARG;
    sl.pf[file].lead = lead;
    return XM_OK;
}

int xm_strip_upload(xm_strip *s, int slot, int file, uint64_t offset, uint64_t bytes)
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || file < 0 || file > 1) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    if (offset == 0) { sl.uploaded[file] = 0; sl.held[file] = 0; sl.pf[file].lead = 0; }    // a new window of this file (the previous one may have been abandoned)
    if (offset != sl.uploaded[file] || offset + bytes > sl.window_cap) return XM_ERR_INVALID_ARG;
    if (bytes == 0) return XM_OK;
    XMS_HIP(s, hipSetDevice(s->device));
    if (!sl.upload_timed) {
        XMS_HIP(s, hipEventRecord(sl.ev[0], sl.stream));
        sl.upload_timed = true;
    }
    PerFile &q = sl.pf[file];
    if (offset == 0) sl.marked[file] = 0;
    if (zero_copy_text()) {
        // S1 on the chunks that are complete and not the last one staged so far (the window's last chunk needs its true length
        // and whether the window ends its file: xm_strip_run marks what is left)
        XMS_TRY(mark_chunks(s, sl, file, offset + bytes, offset + bytes, (uint32_t)((offset + bytes) / CHUNK), false));
    } else {
        XMS_HIP(s, hipMemcpyAsync(q.d_text + offset, q.h_text + offset, (size_t)bytes, hipMemcpyHostToDevice, sl.stream));
    }
    sl.uploaded[file] = offset + bytes;
    return XM_OK;
}

int xm_strip_run(xm_strip *s, int slot, uint64_t len1, int eof1, uint64_t len2, int eof2,
                 int score_mode, int paired, int skip_repeated, int keep_halo, uint64_t max_records, xm_strip_block *out)
{
    if (!s || !out || slot < 0 || slot >= XMS_SLOTS || score_mode < XMS_SCORE_AS_XS || score_mode > XMS_SCORE_CIGAR)
        return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    const uint64_t sent[2] = {sl.uploaded[0], sl.uploaded[1]};
    const bool timed = sl.upload_timed;
    sl.uploaded[0] = sl.uploaded[1] = 0;
    sl.upload_timed = false;
    sl.last_score_mode = -1;
    sl.classified = false;
    if (len1 > sl.window_cap || len2 > sl.window_cap || sl.pf[0].lead > len1 || sl.pf[1].lead > len2 || max_records == 0 || max_records > sl.record_cap || sent[0] > len1 ||
        sent[1] > len2)
        return XM_ERR_INVALID_ARG;
    XMS_HIP(s, hipSetDevice(s->device));
    const bool cigar = score_mode == XMS_SCORE_CIGAR;
    if (cigar) XMS_TRY(ensure_cigar(s, sl));
    const uint64_t len[2] = {len1, len2};
    const int eof[2] = {eof1, eof2};
    Job job;
    std::memset(&job, 0, sizeof job);
    uint32_t max_chunks = 1;
    for (int f = 0; f < 2; ++f) {
        FileView &v = job.f[f];
        PerFile &q = sl.pf[f];
        v.text = q.d_text;
        v.lead = (uint32_t)q.lead;
        v.len = (uint32_t)len[f];
        // a trailing '\r' may be the first half of a "\r\n" that continues in the next window
        v.usable = (!eof[f] && len[f] > 0 && q.h_text[len[f] - 1] == '\r') ? (uint32_t)len[f] - 1u : (uint32_t)len[f];
        v.eof = eof[f] ? 1u : 0u;
        v.n_chunks = (uint32_t)((len[f] + CHUNK - 1) / CHUNK);
        v.mask16 = q.d_mask; v.chunk_cnt = q.d_chunk_cnt; v.chunk_base = q.d_chunk_base;
        v.lend = q.d_lend; v.lnext = q.d_lnext;
        v.r_name_off = q.d_r_name_off; v.r_name_len = q.d_r_name_len; v.r_norm = q.d_r_norm;
        v.r_a = q.d_r_a; v.r_x = q.d_r_x; v.r_nm = q.d_r_nm; v.r_nops = q.d_r_nops; v.r_cig_off = q.d_r_cig_off; v.r_cig_len = q.d_r_cig_len;
        v.r_flag = q.d_r_flag;
        v.blk_cnt = q.d_blk_cnt; v.blk_base = q.d_blk_base; v.sel = q.d_sel;
        v.loff = q.d_loff; v.llen = q.d_llen; v.nlen = q.d_nlen; v.lflag = q.d_lflag;
        v.nm = q.d_nm; v.cpos = q.d_cpos; v.cig_cnt = q.d_cig_cnt; v.cig_tile = q.d_cig_tile; v.cig_ops = q.d_cig_ops;
        v.ops_cap = (uint32_t)q.ops_cap;
        max_chunks = std::max(max_chunks, v.n_chunks);
    }
    job.state = sl.d_state;
    job.summary = sl.d_summary;
    job.cap_lines = (uint32_t)max_records + 1u;
    job.max_records = (uint32_t)max_records;
    job.xtag0 = score_mode == XMS_SCORE_AS_ZS ? 'Z' : 'X';
    job.cigar = cigar ? 1u : 0u;
    job.skip = skip_repeated ? 1u : 0u;
    job.paired = paired ? 1u : 0u;
    job.keep_halo = keep_halo ? 1u : 0u;
    for (int c = 0; c < 4; ++c) job.col[c] = sl.d_col[c];
    job.unit_bits = sl.d_bits;

    hipStream_t st = sl.stream;
    if (!timed) XMS_HIP(s, hipEventRecord(sl.ev[0], st));
    if (!zero_copy_text()) {
        for (int f = 0; f < 2; ++f) {                  // what xm_strip_upload has not sent yet
            if (len[f] > sent[f])
                XMS_HIP(s, hipMemcpyAsync(sl.pf[f].d_text + sent[f], sl.pf[f].h_text + sent[f], (size_t)(len[f] - sent[f]),
                                          hipMemcpyHostToDevice, st));
            sl.marked[f] = 0;                          // S1 below, on the whole copy
        }
    } else {
        for (int f = 0; f < 2; ++f)
            if (sent[f] == 0) sl.marked[f] = 0;        // nothing of this window was announced: all of it is marked here
    }
    // S1 on what is left of each window (zero-copy: that is where the text crosses the link; the last chunk always is left) -- of
    // a window begun behind a gap also the chunks in front that waited for the lead
    for (int f = 0; f < 2; ++f) {
        int rc = XM_OK;
        if (sl.held[f] != 0) {
            if (!zero_copy_text()) rc = XM_ERR_INVALID_ARG;
            else {
                // the chunks that lie in front of the lead altogether hold no terminator: their masks and counts are zeroed, and
                // nothing of them crosses the link; S1 begins with the chunk the text begins in
                const uint32_t upto = std::min(sl.held[f], job.f[f].n_chunks);
                const uint32_t c_lead = std::min((uint32_t)(sl.pf[f].lead / CHUNK), upto);
                if (c_lead) {
                    if (!sl.state_cleared) {
                        XMS_HIP(s, hipMemsetAsync(sl.d_state, 0, ST_WORDS * sizeof(uint32_t), st));
                        sl.state_cleared = true;
                    }
                    XMS_HIP(s, hipMemsetAsync(sl.pf[f].d_mask, 0, (size_t)c_lead * GROUPS * sizeof(uint16_t), st));
                    XMS_HIP(s, hipMemsetAsync(sl.pf[f].d_chunk_cnt, 0, (size_t)c_lead * WAVES * sizeof(uint32_t), st));
                }
                rc = mark_range(s, sl, f, len[f], job.f[f].usable, c_lead, upto);
            }
            if (sl.marked[f] < sl.held[f]) sl.marked[f] = sl.held[f];
        }
        if (rc == XM_OK) rc = mark_chunks(s, sl, f, len[f], job.f[f].usable, job.f[f].n_chunks, true);
        if (rc != XM_OK) { sl.state_cleared = false; sl.marked[0] = sl.marked[1] = 0; sl.held[0] = sl.held[1] = 0; return rc; }
    }
    sl.held[0] = sl.held[1] = 0;
    sl.pf[0].lead = sl.pf[1].lead = 0;                                        // (the job has them: the next window says its own)
    if (!sl.state_cleared) XMS_HIP(s, hipMemsetAsync(sl.d_state, 0, ST_WORDS * sizeof(uint32_t), st));   // two empty windows
    sl.state_cleared = false;
    sl.marked[0] = sl.marked[1] = 0;
    XMS_HIP(s, hipEventRecord(sl.ev[1], st));
    chunk_scan_kernel<<<2, 1024, 0, st>>>(job);
    fill_kernel<<<dim3(max_chunks, 2), SB, 0, st>>>(job);
    // a line takes at least one byte of its window: no more lines than that, no more records than both files have lines
    const uint64_t line_bound = std::min<uint64_t>(max_records + 1, std::max(len1, len2) + 1);
    const uint64_t rec_bound = std::min<uint64_t>(max_records, std::min(len1, len2) + 1);
    const uint64_t parse_bound = skip_repeated ? line_bound : rec_bound;
    // the kernels stride over their lines / records: a grid the size of the bound would mostly be waves with nothing to do
    // (the bound is what a window could hold at two bytes a line; a 128 MB window of 2 x 150 bp reads holds 330 k lines)
    const uint32_t line_blocks = (uint32_t)std::min<uint64_t>((parse_bound + SB - 1) / SB, 2048),
                   rec_blocks = (uint32_t)std::min<uint64_t>((rec_bound + SB - 1) / SB, 2048);
    parse_kernel<<<dim3(line_blocks, 2), SB, 0, st>>>(job);
    if (skip_repeated) {
        start_count_kernel<<<dim3(line_blocks, 2), SB, 0, st>>>(job);
        start_scan_kernel<<<2, 1024, 0, st>>>(job);
        start_fill_kernel<<<dim3(line_blocks, 2), SB, 0, st>>>(job);
    }
    pair_kernel<<<rec_blocks, SB, 0, st>>>(job);
    if (cigar) {
        cig_scan_kernel<<<2, 1024, 0, st>>>(job);
        cig_write_kernel<<<dim3(rec_blocks, 2), SB, 0, st>>>(job);
    }
    summary_kernel<<<1, 64, 0, st>>>(job);
    XMS_HIP(s, hipGetLastError());
    XMS_HIP(s, hipEventRecord(sl.ev[2], st));
    XMS_HIP(s, hipMemcpyAsync(sl.h_summary, sl.d_summary, SUM_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    XMS_HIP(s, hipStreamSynchronize(st));

    const uint64_t *sum = sl.h_summary;
    const uint64_t n = sum[SUM_N];
    std::memset(out, 0, sizeof *out);
    out->n_records = n;
    out->consumed1 = sum[SUM_CONS1];
    out->consumed2 = sum[SUM_CONS2];
    out->consumed_lines1 = sum[SUM_CL1];
    out->consumed_lines2 = sum[SUM_CL2];
    out->ended = (int32_t)sum[SUM_ENDED];
    out->starved = (int32_t)sum[SUM_STARVED];
    out->mismatch_at = (int64_t)sum[SUM_MISMATCH];
    out->non_ascii = (int32_t)sum[SUM_NONASCII];
    out->overflow = (int32_t)sum[SUM_OVERFLOW];
    out->n_lines1 = sum[SUM_L1];
    out->n_lines2 = sum[SUM_L2];
    if (n) {
        for (int f = 0; f < 2; ++f) {
            PerFile &q = sl.pf[f];
            XMS_HIP(s, hipMemcpyAsync(q.h_loff, q.d_loff, n * 4, hipMemcpyDeviceToHost, st));
            XMS_HIP(s, hipMemcpyAsync(q.h_llen, q.d_llen, n * 4, hipMemcpyDeviceToHost, st));
            XMS_HIP(s, hipMemcpyAsync(q.h_nlen, q.d_nlen, n * 4, hipMemcpyDeviceToHost, st));
            XMS_HIP(s, hipMemcpyAsync(q.h_lflag, q.d_lflag, n, hipMemcpyDeviceToHost, st));
        }
        XMS_HIP(s, hipStreamSynchronize(st));
    }
    uint64_t n_exc = 0;
    for (int f = 0; f < 2; ++f)
        for (uint64_t k = 0; k < n; ++k) n_exc += (sl.pf[f].h_lflag[k] & (XMS_LINE_EX_A | XMS_LINE_EX_X)) ? 1 : 0;
    out->n_exceptions = n_exc;
    out->line_off1 = sl.pf[0].h_loff; out->line_off2 = sl.pf[1].h_loff;
    out->line_len1 = sl.pf[0].h_llen; out->line_len2 = sl.pf[1].h_llen;
    out->norm_len1 = sl.pf[0].h_nlen; out->norm_len2 = sl.pf[1].h_nlen;
    out->line_flags1 = sl.pf[0].h_lflag; out->line_flags2 = sl.pf[1].h_lflag;
    (void)hipEventElapsedTime(&out->ms_upload, sl.ev[0], sl.ev[1]);
    (void)hipEventElapsedTime(&out->ms_kernels, sl.ev[1], sl.ev[2]);
    sl.last_score_mode = score_mode;
    return XM_OK;
}

int xm_strip_classify(xm_strip *s, int slot, int mode, uint64_t n_records, int32_t min_score_floor,
                      const uint8_t **code, const uint32_t **idx, uint64_t bin_offsets[8], uint64_t counts[64])
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || !code || !idx || !bin_offsets || !counts) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    if (n_records > sl.record_cap || sl.last_score_mode < 0) return XM_ERR_INVALID_ARG;
    *code = sl.h_code;
    *idx = sl.h_idx;
    std::memset(bin_offsets, 0, 8 * sizeof(uint64_t));
    std::memset(counts, 0, 64 * sizeof(uint64_t));
    if (n_records == 0) return XM_OK;
    XMS_HIP(s, hipSetDevice(s->device));
    hipStream_t st = sl.stream;
    const bool cigar = sl.last_score_mode == XMS_SCORE_CIGAR;
    int rc;
    if (cigar) {
        XMS_HIP(s, hipMemsetAsync(sl.d_range, 0, sizeof(uint32_t), st));
        const PerFile &a = sl.pf[0], &b = sl.pf[1];
        rc = xm_classify_compact_cigar_packed_dev(s->ctx, st, mode, n_records, a.d_nm, a.d_cig_cnt, a.d_cig_tile, a.d_cig_ops, sl.d_col[1],
                                                  b.d_nm, b.d_cig_cnt, b.d_cig_tile, b.d_cig_ops, sl.d_col[3], sl.d_bits, min_score_floor,
                                                  sl.d_code, sl.d_bins4, sl.d_range, sl.d_idx, sl.d_off_counts, sl.d_off_counts + 8);
    } else {
        rc = xm_classify_compact_dev(s->ctx, st, mode, n_records, sl.d_col[0], sl.d_col[1], sl.d_col[2], sl.d_col[3], sl.d_bits,
                                     min_score_floor, sl.d_code, sl.d_bins4, sl.d_idx, sl.d_off_counts, sl.d_off_counts + 8);
    }
    if (rc != XM_OK) {
        std::lock_guard<std::mutex> hold(s->error_lock);
        s->last_error = xm_last_hip_error(s->ctx);
        return rc;
    }
    XMS_HIP(s, hipMemcpyAsync(sl.h_code, sl.d_code, n_records, hipMemcpyDeviceToHost, st));
    XMS_HIP(s, hipMemcpyAsync(sl.h_off_counts, sl.d_off_counts, 72 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    if (cigar) XMS_HIP(s, hipMemcpyAsync(sl.h_off_counts + 72, sl.d_range, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    XMS_HIP(s, hipStreamSynchronize(st));
    if (cigar && *reinterpret_cast<const uint32_t *>(sl.h_off_counts + 72) != 0u) return XM_ERR_RANGE;
    const uint64_t units = sl.h_off_counts[7];
    if (units > n_records) return XM_ERR_HIP;
    if (units) {
        XMS_HIP(s, hipMemcpyAsync(sl.h_idx, sl.d_idx, units * 4, hipMemcpyDeviceToHost, st));
        XMS_HIP(s, hipStreamSynchronize(st));
    }
    std::memcpy(bin_offsets, sl.h_off_counts, 8 * sizeof(uint64_t));
    std::memcpy(counts, sl.h_off_counts + 8, 64 * sizeof(uint64_t));
    sl.classified = true;
    return XM_OK;
}

int xm_strip_fetch_bins(xm_strip *s, int slot, uint64_t n_records, int paired, uint32_t sink_mask, xm_strip_bins *out)
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || !out) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    if (n_records > sl.record_cap || n_records > 0xFFFFFFF0ull || sl.last_score_mode < 0 || !sl.classified) return XM_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    const uint32_t n = (uint32_t)n_records;
    const uint64_t units64 = sl.h_off_counts[7];                                    // of the slot's last xm_strip_classify
    if (n == 0 || units64 == 0) return XM_OK;
    if (units64 > n_records) return XM_ERR_INVALID_ARG;
    const uint32_t n_units = (uint32_t)units64;
    XMS_HIP(s, hipSetDevice(s->device));
    // the stream of the six outputs: both windows' text at most (+ a '\n' per line that had none), unless units overlap
    const uint64_t want_cap = std::min<uint64_t>(2 * sl.window_cap + 4096, 0xFFFFFFF0ull);
    if (sl.out_cap < want_cap) {
        XMS_HIP(s, hipStreamSynchronize(sl.copy_stream));
        sl.out_cap = 0;
        XMS_TRY(dalloc(s, sl.d_out, (size_t)want_cap + 64)); XMS_TRY(halloc(s, sl.h_out, (size_t)want_cap + 64));
        sl.out_cap = want_cap;
    }
    if (sl.out_records < sl.record_cap) {
        sl.out_records = 0;
        const size_t nr = (size_t)sl.record_cap + 64;
        XMS_TRY(dalloc(s, sl.d_usize, nr)); XMS_TRY(dalloc(s, sl.d_uplace, nr)); XMS_TRY(dalloc(s, sl.d_upart, nr / SCAN_TILE + 8));
        sl.out_records = sl.record_cap;
    }
    out->text = sl.h_out;
    hipStream_t st = sl.stream;
    if (sl.out_issued) XMS_HIP(s, hipStreamWaitEvent(st, sl.ev_out, 0));            // the previous window's stream has left d_out
    const uint32_t n_part = (n_units + SCAN_TILE - 1u) / SCAN_TILE;
    const unsigned long long *d_off = reinterpret_cast<const unsigned long long *>(sl.d_off_counts);
    const PerFile &a = sl.pf[0], &b = sl.pf[1];
    XMS_HIP(s, hipMemsetAsync(sl.d_gstate, 0, 16 * sizeof(uint32_t), st));
    // gstate: [0] a wanted line that is not '\t'.join(fields) as it stands, [1] the scan's 32-bit total, [2..3] the 64-bit total,
    // [4..11] where each bin's text begins
    sam_unit_size_kernel<<<(n_units + 255u) / 256u, 256, 0, st>>>(sl.d_idx, d_off, n_units, n, paired ? 1 : 0, sink_mask, a.d_llen, b.d_llen,
                                                                 a.d_lflag, b.d_lflag, sl.d_usize,
                                                                 reinterpret_cast<unsigned long long *>(sl.d_gstate + 2), sl.d_gstate + 0);
    size_sum_kernel<<<n_part, 256, 0, st>>>(sl.d_usize, n_units, sl.d_upart);
    part_scan_kernel<<<1, 1024, 0, st>>>(sl.d_upart, n_part, sl.d_gstate + 1);
    size_place_kernel<true><<<n_part, 256, 0, st>>>(sl.d_usize, n_units, sl.d_upart, sl.d_uplace);
    bin_start_kernel<<<1, 64, 0, st>>>(sl.d_uplace, d_off, n_units, sl.d_gstate + 1, sl.d_gstate + 4);
    XMS_HIP(s, hipMemcpyAsync(sl.h_gstate, sl.d_gstate, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    XMS_HIP(s, hipStreamSynchronize(st));
    XMS_HIP(s, hipGetLastError());
    if (sl.h_gstate[0] != 0u) { out->status = 3; return XM_OK; }            // a line the writer has to re-join: the host's window
    uint64_t total = 0;
    std::memcpy(&total, sl.h_gstate + 2, sizeof total);
    if (total > sl.out_cap) { out->status = 2; return XM_OK; }              // more text than the buffers hold (overlapping units)
    for (int k = 0; k < 8; ++k) out->bin_off[k] = sl.h_gstate[4 + k];
    sam_line_copy_kernel<<<((paired ? 2u : 1u) * n_units + 3u) / 4u, 256, 0, st>>>(
        a.d_text, b.d_text, a.d_loff, b.d_loff, a.d_llen, b.d_llen, sl.d_idx, d_off, n_units, n, paired ? 1 : 0, sink_mask, sl.d_usize, sl.d_uplace,
        sl.d_out, (uint32_t)sl.out_cap);
    XMS_HIP(s, hipEventRecord(sl.ev_filled, st));
    XMS_HIP(s, hipStreamWaitEvent(sl.copy_stream, sl.ev_filled, 0));
    if (total) {
        const uint32_t wg = out_copy_workgroups();
        if (wg == 0u) XMS_HIP(s, hipMemcpyAsync(sl.h_out, sl.d_out, (size_t)total, hipMemcpyDeviceToHost, sl.copy_stream));
        else {
            const uint64_t n16 = (total + 15u) / 16u;
            out_copy_launch((uint32_t)std::min<uint64_t>(out_copy_waves(wg), (n16 + 63u) / 64u), sl.copy_stream,
                            reinterpret_cast<const v4u32 *>(sl.d_out), reinterpret_cast<v4u32 *>(sl.h_out), n16);
        }
    }
    XMS_HIP(s, hipEventRecord(sl.ev_out, sl.copy_stream));
    sl.out_issued = true;
    XMS_HIP(s, hipGetLastError());
    return XM_OK;
}

int xm_strip_out_wait(xm_strip *s, int slot)
{
    if (!s || slot < 0 || slot >= XMS_SLOTS) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    if (!sl.out_issued) return XM_OK;
    XMS_HIP(s, hipSetDevice(s->device));
    XMS_HIP(s, hipEventSynchronize(sl.ev_out));
    return XM_OK;
}

int xm_strip_columns(xm_strip *s, int slot, uint64_t n_records, int32_t *as1, int32_t *xs1, int32_t *as2, int32_t *xs2,
                     uint64_t *unit_bits)
{
    if (!s || slot < 0 || slot >= XMS_SLOTS) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    if (n_records > sl.record_cap) return XM_ERR_INVALID_ARG;
    if (n_records == 0) return XM_OK;
    XMS_HIP(s, hipSetDevice(s->device));
    int32_t *dst[4] = {as1, xs1, as2, xs2};
    for (int c = 0; c < 4; ++c)
        if (dst[c]) XMS_HIP(s, hipMemcpyAsync(dst[c], sl.d_col[c], n_records * 4, hipMemcpyDeviceToHost, sl.stream));
    if (unit_bits)
        XMS_HIP(s, hipMemcpyAsync(unit_bits, sl.d_bits, (n_records + 63) / 64 * 8, hipMemcpyDeviceToHost, sl.stream));
    XMS_HIP(s, hipStreamSynchronize(sl.stream));
    return XM_OK;
}

int xm_strip_cigar_columns(xm_strip *s, int slot, int file, uint64_t n_records, int32_t *nm, uint8_t *cig_cnt, uint32_t *cig_tile,
                           uint32_t *cig_ops, uint64_t ops_capacity, uint64_t *n_ops)
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || file < 0 || file > 1 || !n_ops) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    if (n_records > sl.record_cap || sl.last_score_mode != XMS_SCORE_CIGAR) return XM_ERR_INVALID_ARG;
    *n_ops = 0;
    if (n_records == 0) return XM_OK;
    XMS_HIP(s, hipSetDevice(s->device));
    const PerFile &q = sl.pf[file];
    const size_t tiles = (size_t)XM_CIG_TILES(n_records) + 1;
    uint32_t last = 0;
    XMS_HIP(s, hipMemcpy(&last, q.d_cig_tile + tiles - 1, 4, hipMemcpyDeviceToHost));
    *n_ops = last;
    if (nm) XMS_HIP(s, hipMemcpy(nm, q.d_nm, n_records * 4, hipMemcpyDeviceToHost));
    if (cig_cnt) XMS_HIP(s, hipMemcpy(cig_cnt, q.d_cig_cnt, n_records, hipMemcpyDeviceToHost));
    if (cig_tile) XMS_HIP(s, hipMemcpy(cig_tile, q.d_cig_tile, tiles * 4, hipMemcpyDeviceToHost));
    if (cig_ops) {
        if (ops_capacity < last) return XM_ERR_INVALID_ARG;
        if (last) XMS_HIP(s, hipMemcpy(cig_ops, q.d_cig_ops, (size_t)last * 4, hipMemcpyDeviceToHost));
    }
    return XM_OK;
}

int xm_strip_device_columns(xm_strip *s, int slot, void *ptrs[5])
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || !ptrs) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    for (int c = 0; c < 4; ++c) ptrs[c] = sl.d_col[c];
    ptrs[4] = sl.d_bits;
    return XM_OK;
}

const char *xm_strip_last_error(const xm_strip *s)
{
    // the other slot's thread may be assigning the text: copied under the lock into a buffer of the calling thread
    static thread_local std::string mine;
    if (!s) return "";
    {
        std::lock_guard<std::mutex> hold(const_cast<xm_strip *>(s)->error_lock);
        mine = s->last_error;
    }
    return mine.c_str();
}

}  // extern "C"
