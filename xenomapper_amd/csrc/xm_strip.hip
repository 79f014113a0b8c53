// xm_strip.hip -- the SAM column stripper on the GPU (C ABI in include/xenomapper_strip.h), gfx950 only.
//
// Restates on the device what xm_sam.cpp does with host threads (and what the reference does in Python):
//   getReadPairs without skipping   /root/reference/xenomapper/xenomapper.py:95-108  (readline / strip / split / names equal)
//   get_tag (field search)          :186-190      get_tag_with_ZS_as_XS :204-206
//   the unit rule                   :402-405      (name equals the previous record's name)
// Four kernels per pair of windows, all bound by reading the text from HBM (the text arrives over PCIe at a hundredth of
// that rate, which is what bounds the step):
//   S1 mark_kernel    16 bytes per lane: terminator bits of '\n' / '\r\n' / '\r' (Python's universal newlines) as one 16-bit
//                     mask per 16 bytes, terminators per 64 KiB chunk, non-ASCII test
//   S2 chunk_scan     exclusive scan of the chunk counts (one workgroup per file)
//   S3 fill_kernel    reads the masks (1/8 of the text): line k ends at lend[k], line k + 1 starts at lnext[k]
//   S4 strip_kernel   one lane per record k = line k of both files: split by str.split()'s separators in aligned 8-byte words
//                     (words inside the long leading fields are skipped whole), tags, names, unit bit (ballot -> one
//                     64-bit word per wave), first stop (blank line / names differ) by atomicMin
//   S5 summary_kernel one lane: the walk's outcome (records, consumed bytes, ended / starved / mismatch) as xmh_parse reports it
// The score columns and the unit mask never leave the device: xm_strip_classify runs the fused classify pass on them.
#include "../../include/xenomapper_strip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

namespace {

constexpr int SB = 256;                               // lanes per workgroup of S1 / S3 / S4
constexpr uint32_t CHUNK = 1u << 16;                  // bytes of text per workgroup in S1 / S3
constexpr uint32_t GROUPS = CHUNK / 16;               // 16-byte groups per chunk
constexpr uint32_t ITER = GROUPS / SB;
constexpr int32_t ABSENT = INT32_MIN;

enum { ST_NTERM0 = 0, ST_NTERM1 = 1, ST_NONASCII = 2, ST_KSTOP = 3, ST_WORDS = 8 };
enum { SUM_N = 0, SUM_CONS1, SUM_CONS2, SUM_CL1, SUM_CL2, SUM_ENDED, SUM_STARVED, SUM_MISMATCH, SUM_NONASCII, SUM_L1, SUM_L2,
       SUM_WORDS = 16 };

struct FileView {                                     // one file's window and its line index, on the device
    const uint8_t *text;
    uint32_t len, usable, eof, n_chunks;
    uint16_t *mask16;
    uint32_t *chunk_cnt, *chunk_base;
    uint32_t *lend, *lnext;                           // cap_lines entries each
};

struct Job {
    FileView f[2];
    uint32_t *state;                                  // ST_WORDS
    uint64_t *summary;                                // SUM_WORDS
    uint32_t cap_lines;                               // max_records + 1
    uint32_t max_records;
    uint32_t xtag0;                                   // 'X' or 'Z'
    uint32_t paired, keep_halo;
    int32_t *col[4];                                  // as1, xs1, as2, xs2
    uint64_t *unit_bits;
    uint32_t *loff[2], *llen[2], *nlen[2];
    uint8_t *lflag[2];
};

// Python's str.split() separators in the ASCII range: \t \n \v \f \r, FS GS RS US, space
__device__ __forceinline__ bool is_ws(uint32_t c) { return c == 32u || (c - 9u) <= 4u || (c - 28u) <= 3u; }

__device__ __forceinline__ uint32_t has_byte(uint32_t w, uint32_t c)
{
    const uint32_t x = w ^ (c * 0x01010101u);
    return (x - 0x01010101u) & ~x & 0x80808080u;
}

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
__global__ void __launch_bounds__(SB) mark_kernel(const Job job)
{
    const FileView &f = job.f[blockIdx.y];
    if (blockIdx.x >= f.n_chunks) return;
    __shared__ uint32_t red[SB / 64];
    const uint32_t n_groups = (f.len + 15u) / 16u;
    uint32_t cnt = 0, hi = 0;
    for (uint32_t it = 0; it < ITER; ++it) {
        const uint32_t g = blockIdx.x * GROUPS + it * SB + threadIdx.x;
        if (g >= n_groups) break;
        const uint32_t p0 = g * 16u;
        const uint4 v = *reinterpret_cast<const uint4 *>(f.text + p0);       // the buffer is readable past len (padding)
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        const bool full = p0 + 16u <= f.len;
        uint32_t any = 0;
        for (int q = 0; q < 4; ++q) any |= has_byte(w[q], 10u) | has_byte(w[q], 13u);
        if (full) hi |= v.x | v.y | v.z | v.w;
        uint32_t m = 0;
        if (any || !full) {
            uint32_t prev = p0 ? (uint32_t)f.text[p0 - 1] : 0u;
            for (uint32_t i = 0; i < 16; ++i) {
                const uint32_t c = (w[i >> 2] >> (8u * (i & 3u))) & 0xFFu, p = p0 + i;
                if (p < f.len) hi |= c;
                // '\r' always ends a line (alone or as the first half of "\r\n"); '\n' unless it is that second half
                const bool term = c == 13u || (c == 10u && prev != 13u);
                if (term && p < f.usable) m |= 1u << i;
                prev = c;
            }
        }
        f.mask16[g] = (uint16_t)m;
        cnt += (uint32_t)__popc(m);
    }
    cnt = wave_sum(cnt);
    hi = __any((hi & 0x80808080u) != 0) ? 1u : 0u;
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int i = 0; i < SB / 64; ++i) t += red[i];
        f.chunk_cnt[blockIdx.x] = t;
    }
    if (hi && (threadIdx.x & 63u) == 0) atomicOr(&job.state[ST_NONASCII], 1u);
}

// ---- S2: exclusive scan of the chunk counts, one workgroup per file ------------------------------------------------
__global__ void __launch_bounds__(1024) chunk_scan_kernel(const Job job)
{
    const FileView &f = job.f[blockIdx.x];
    __shared__ uint32_t ws[16];
    const uint32_t per = (f.n_chunks + 1023u) / 1024u;
    const uint32_t b = min(threadIdx.x * per, f.n_chunks), e = min(b + per, f.n_chunks);
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
    __shared__ uint32_t ws[SB / 64];
    const uint32_t n_groups = (f.len + 15u) / 16u;
    uint32_t run = f.chunk_base[blockIdx.x];
    for (uint32_t it = 0; it < ITER; ++it) {
        const uint32_t g = blockIdx.x * GROUPS + it * SB + threadIdx.x;
        uint32_t m = g < n_groups ? (uint32_t)f.mask16[g] : 0u;
        const uint32_t c = (uint32_t)__popc(m), incl = wave_scan_incl(c);
        if ((threadIdx.x & 63u) == 63u) ws[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t wb = 0, tot = 0;
        for (uint32_t i = 0; i < SB / 64; ++i) {
            if (i < (threadIdx.x >> 6)) wb += ws[i];
            tot += ws[i];
        }
        uint32_t j = run + wb + incl - c;
        while (m) {
            const uint32_t p = g * 16u + (uint32_t)(__ffs((int)m) - 1);
            m &= m - 1u;
            if (j < job.cap_lines) {
                const bool crlf = f.text[p] == 13u && p + 1u < f.len && f.text[p + 1u] == 10u;
                f.lend[j] = p;
                f.lnext[j] = p + (crlf ? 2u : 1u);
            }
            ++j;
        }
        run += tot;
        __syncthreads();
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
    if (n_terms >= cap_lines) {                       // more lines than a block may hold: the end of the window is never reached
        L.count = n_terms;
        L.complete_end = 0;
        return L;
    }
    const uint32_t last_next = n_terms ? f.lnext[n_terms - 1u] : 0u;
    const bool extra = f.eof && last_next < f.len;
    L.count = n_terms + (extra ? 1u : 0u);
    L.complete_end = extra ? f.len : last_next;
    return L;
}

__device__ __forceinline__ void line_span(const FileView &f, const Lines &L, uint32_t k, uint32_t &start, uint32_t &len)
{
    start = k ? f.lnext[k - 1u] : 0u;
    len = (k < L.n_terms ? f.lend[k] : f.len) - start;
}

struct Rec {
    uint32_t name_off, name_len;                      // first field, window offsets
    uint32_t norm_len, n_tok;
    int32_t a, x;
    uint32_t ex_a, ex_x;                              // 0, 1 = not a plain int32, 2 = tag matched more than once
    bool normal;
};

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

// One line: fields = line.split() (xenomapper.py:103-104), the tag search of get_tag over fields[11:] (:186-190).
__device__ void parse_line(const uint8_t *text, uint32_t start, uint32_t n, uint32_t xtag0, Rec &r)
{
    const uint32_t end = start + n;
    uint32_t n_tok = 0, total = 0, name_off = start, name_len = 0;
    bool normal = true, in_tok = false, prev_ws = false, ma = false, mx = false;
    uint32_t k = 0, prevc = 0, last_colon = 0;
    uint32_t n_a = 0, n_x = 0, a_b = 0, a_e = 0, x_b = 0, x_e = 0;
    uint32_t p = start;
    while (p < end) {
        const uint32_t wa = p & ~7u;
        const uint64_t w = *reinterpret_cast<const uint64_t *>(text + wa);
        const uint32_t hi = min(end, wa + 8u);
        // a whole word inside one of the eleven mandatory fields without a byte below 0x21 (the input is ASCII, or the
        // result is thrown away): nothing to learn from it but its length
        if (p == wa && hi == wa + 8u && in_tok && k < 11u &&
            !((w - 0x2121212121212121ull) & ~w & 0x8080808080808080ull)) {
            total += 8u;
            p += 8u;
            continue;
        }
        for (; p < hi; ++p) {
            const uint32_t c = (uint32_t)(w >> (8u * (p - wa))) & 0xFFu;
            if (is_ws(c)) {
                if (in_tok) {
                    if (k == 0u) name_len = p - name_off;
                    if (ma && ++n_a == 1u) { a_b = last_colon; a_e = p; }
                    if (mx && ++n_x == 1u) { x_b = last_colon; x_e = p; }
                    in_tok = false;
                }
                // '\t'.join(fields) == line  <=>  exactly one '\t' between fields, nothing in front or behind
                if (c != 9u || n_tok == 0u || prev_ws) normal = false;
                prev_ws = true;
            } else {
                if (!in_tok) {
                    in_tok = true;
                    k = n_tok++;
                    if (k == 0u) name_off = p;
                    last_colon = p;
                    ma = mx = false;
                    prevc = 0u;
                }
                ++total;
                if (k >= 11u) {
                    ma = ma || (prevc == 'A' && c == 'S');
                    mx = mx || (prevc == xtag0 && c == 'S');
                    if (c == ':') last_colon = p + 1u;
                }
                prevc = c;
                prev_ws = false;
            }
        }
    }
    if (in_tok) {
        if (k == 0u) name_len = end - name_off;
        if (ma && ++n_a == 1u) { a_b = last_colon; a_e = end; }
        if (mx && ++n_x == 1u) { x_b = last_colon; x_e = end; }
    }
    if (prev_ws) normal = false;
    r.name_off = name_off;
    r.name_len = name_len;
    r.n_tok = n_tok;
    r.norm_len = n_tok ? total + (n_tok - 1u) : 0u;
    r.normal = normal && n_tok > 0u;
    r.a = r.x = ABSENT;
    r.ex_a = r.ex_x = 0u;
    if (n_a >= 1u && !plain_int(text, a_b, a_e, r.a)) { r.a = ABSENT; r.ex_a = 1u; }
    if (n_x >= 1u && !plain_int(text, x_b, x_e, r.x)) { r.x = ABSENT; r.ex_x = 1u; }
    if (n_a > 1u) r.ex_a = 2u;
    if (n_x > 1u) r.ex_x = 2u;
}

__device__ bool same_bytes(const uint8_t *a, const uint8_t *b, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i)
        if (a[i] != b[i]) return false;
    return true;
}

// ---- S4: one lane per record ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SB) strip_kernel(const Job job)
{
    const Lines L0 = lines_of(job.f[0], job.state[ST_NTERM0], job.cap_lines);
    const Lines L1 = lines_of(job.f[1], job.state[ST_NTERM1], job.cap_lines);
    const uint32_t lim = min(min(L0.count, L1.count), job.max_records);
    const uint32_t k = blockIdx.x * SB + threadIdx.x;
    if ((k & ~63u) >= lim) return;                     // the whole wave is past the end (wave-uniform)
    bool unit = false;
    if (k < lim) {
        Rec r[2];
        uint32_t st[2], ln[2];
        line_span(job.f[0], L0, k, st[0], ln[0]);
        line_span(job.f[1], L1, k, st[1], ln[1]);
        parse_line(job.f[0].text, st[0], ln[0], job.xtag0, r[0]);
        parse_line(job.f[1].text, st[1], ln[1], job.xtag0, r[1]);
        const bool blank0 = r[0].n_tok == 0u, blank1 = r[1].n_tok == 0u;
        const bool same = r[0].name_len == r[1].name_len &&
                          same_bytes(job.f[0].text + r[0].name_off, job.f[1].text + r[1].name_off, r[0].name_len);
        if (blank0 || blank1 || !same) atomicMin(&job.state[ST_KSTOP], k);
        // the unit rule (:402-405): the name equals the name of the record in front (file 1's names, as the reference)
        if (!job.paired) {
            unit = true;
        } else if (k > 0u) {
            uint32_t ps, pl;
            line_span(job.f[0], L0, k - 1u, ps, pl);
            const uint8_t *t = job.f[0].text;
            uint32_t q = ps;
            const uint32_t pe = ps + pl;
            while (q < pe && is_ws(t[q])) ++q;
            const uint32_t nb = q;
            while (q < pe && !is_ws(t[q])) ++q;
            unit = (q - nb) == r[0].name_len && same_bytes(t + nb, t + r[0].name_off, r[0].name_len);
        }
        job.col[0][k] = r[0].a;
        job.col[1][k] = r[0].x;
        job.col[2][k] = r[1].a;
        job.col[3][k] = r[1].x;
        for (int f = 0; f < 2; ++f) {
            job.loff[f][k] = st[f];
            job.llen[f][k] = ln[f];
            job.nlen[f][k] = r[f].norm_len;
            uint32_t fl = (r[f].normal ? XMS_LINE_NORMAL : 0u) | (r[f].n_tok == 0u ? XMS_LINE_BLANK : 0u) |
                          (r[f].ex_a << 2) | (r[f].ex_x << 4);
            if (f == 0 && !blank0 && !blank1 && !same) fl |= XMS_LINE_MISMATCH;
            job.lflag[f][k] = (uint8_t)fl;
        }
    }
    const uint64_t word = __ballot(unit);
    if ((threadIdx.x & 63u) == 0u) job.unit_bits[k >> 6] = word;
}

// ---- S5: the outcome of the walk, as xm_sam.cpp parse_common reports it ------------------------------------------
__global__ void summary_kernel(const Job job)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Lines L[2] = {lines_of(job.f[0], job.state[ST_NTERM0], job.cap_lines),
                        lines_of(job.f[1], job.state[ST_NTERM1], job.cap_lines)};
    const uint32_t lim = min(min(L[0].count, L[1].count), job.max_records);
    const uint32_t ks = min(job.state[ST_KSTOP], lim);
    uint32_t ended = 0, starved = 0;
    int64_t mismatch = -1;
    if (ks < lim) {
        if ((job.lflag[0][ks] | job.lflag[1][ks]) & XMS_LINE_BLANK) ended = 1; else mismatch = (int64_t)ks;
    } else if (ks < job.max_records) {
        const bool whole0 = job.f[0].eof && L[0].complete_end == job.f[0].len;
        const bool whole1 = job.f[1].eof && L[1].complete_end == job.f[1].len;
        if ((ks >= L[0].count && whole0) || (ks >= L[1].count && whole1)) ended = 1; else starved = 1;
    }
    uint64_t cons[2], cl[2];
    const bool halo = job.keep_halo && ks > 0u && !ended && mismatch < 0;
    for (int f = 0; f < 2; ++f) {
        const uint32_t i = halo ? ks - 1u : ks;
        cons[f] = i < L[f].count ? (uint64_t)(i ? job.f[f].lnext[i - 1u] : 0u) : (uint64_t)L[f].complete_end;
        cl[f] = min(i, L[f].count);
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
    s[SUM_L1] = L[0].count;
    s[SUM_L2] = L[1].count;
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct Slot {
    uint64_t window_cap = 0, record_cap = 0;
    char *h_text[2] = {nullptr, nullptr};              // page-locked staging
    uint8_t *d_text[2] = {nullptr, nullptr};
    uint16_t *d_mask[2] = {nullptr, nullptr};
    uint32_t *d_chunk_cnt[2] = {nullptr, nullptr}, *d_chunk_base[2] = {nullptr, nullptr};
    uint32_t *d_lend[2] = {nullptr, nullptr}, *d_lnext[2] = {nullptr, nullptr};
    int32_t *d_col[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t *d_bits = nullptr;
    uint32_t *d_loff[2] = {nullptr, nullptr}, *d_llen[2] = {nullptr, nullptr}, *d_nlen[2] = {nullptr, nullptr};
    uint8_t *d_lflag[2] = {nullptr, nullptr};
    uint32_t *h_loff[2] = {nullptr, nullptr}, *h_llen[2] = {nullptr, nullptr}, *h_nlen[2] = {nullptr, nullptr};
    uint8_t *h_lflag[2] = {nullptr, nullptr};
    // classify outputs
    uint8_t *d_code = nullptr, *d_bins4 = nullptr, *h_code = nullptr;
    uint32_t *d_idx = nullptr, *h_idx = nullptr;
    uint64_t *d_off_counts = nullptr, *h_off_counts = nullptr;     // 8 + 64 words
    uint32_t *d_state = nullptr;
    uint64_t *d_summary = nullptr, *h_summary = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    uint64_t uploaded[2] = {0, 0};                     // bytes of the staged windows already on their way (xm_strip_upload)
    bool upload_timed = false;
};

}  // namespace

struct xm_strip {
    xm_ctx *ctx = nullptr;
    int device = 0;
    Slot slot[XMS_SLOTS];
    std::string last_error;
};

namespace {

int fail(xm_strip *s, hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    if (s) s->last_error = buf;
    return e == hipErrorOutOfMemory ? XM_ERR_OOM : XM_ERR_HIP;
}

#define XMS_HIP(s, call)                                   \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return fail((s), e_, #call); \
    } while (0)

template <typename T> void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }
template <typename T> void hfree(T *&p) { if (p) { (void)hipHostFree(p); p = nullptr; } }

void free_slot(Slot &sl)
{
    for (int f = 0; f < 2; ++f) {
        hfree(sl.h_text[f]); dfree(sl.d_text[f]); dfree(sl.d_mask[f]); dfree(sl.d_chunk_cnt[f]); dfree(sl.d_chunk_base[f]);
        dfree(sl.d_lend[f]); dfree(sl.d_lnext[f]); dfree(sl.d_loff[f]); dfree(sl.d_llen[f]); dfree(sl.d_nlen[f]); dfree(sl.d_lflag[f]);
        hfree(sl.h_loff[f]); hfree(sl.h_llen[f]); hfree(sl.h_nlen[f]); hfree(sl.h_lflag[f]);
    }
    for (int c = 0; c < 4; ++c) dfree(sl.d_col[c]);
    dfree(sl.d_bits); dfree(sl.d_code); dfree(sl.d_bins4); dfree(sl.d_idx); dfree(sl.d_off_counts);
    hfree(sl.h_code); hfree(sl.h_idx); hfree(sl.h_off_counts);
    sl.window_cap = sl.record_cap = 0;
}

template <typename T> int dalloc(xm_strip *s, T *&p, size_t count)
{
    XMS_HIP(s, hipMalloc((void **)&p, std::max<size_t>(count, 16) * sizeof(T)));
    return XM_OK;
}
template <typename T> int halloc(xm_strip *s, T *&p, size_t count)
{
    XMS_HIP(s, hipHostMalloc((void **)&p, std::max<size_t>(count, 16) * sizeof(T), hipHostMallocDefault));
    return XM_OK;
}

#define XMS_TRY(expr)                  \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != XM_OK) return rc_;  \
    } while (0)

int grow_window(xm_strip *s, Slot &sl, uint64_t bytes)
{
    if (bytes <= sl.window_cap) return XM_OK;
    const uint64_t cap = (bytes + CHUNK - 1) / CHUNK * CHUNK;
    const size_t n_chunks = (size_t)(cap / CHUNK);
    for (int f = 0; f < 2; ++f) {
        hfree(sl.h_text[f]); dfree(sl.d_text[f]); dfree(sl.d_mask[f]); dfree(sl.d_chunk_cnt[f]); dfree(sl.d_chunk_base[f]);
    }
    sl.window_cap = 0;
    for (int f = 0; f < 2; ++f) {
        XMS_TRY(halloc(s, sl.h_text[f], (size_t)cap));
        XMS_TRY(dalloc(s, sl.d_text[f], (size_t)cap + 64));            // S1 / S4 read whole 16- / 8-byte words
        XMS_TRY(dalloc(s, sl.d_mask[f], (size_t)cap / 16 + 16));
        XMS_TRY(dalloc(s, sl.d_chunk_cnt[f], n_chunks));
        XMS_TRY(dalloc(s, sl.d_chunk_base[f], n_chunks));
    }
    sl.window_cap = cap;
    return XM_OK;
}

int grow_records(xm_strip *s, Slot &sl, uint64_t records)
{
    if (records <= sl.record_cap) return XM_OK;
    const size_t n = (size_t)records + 64, lines = (size_t)records + 1;
    for (int f = 0; f < 2; ++f) {
        dfree(sl.d_lend[f]); dfree(sl.d_lnext[f]); dfree(sl.d_loff[f]); dfree(sl.d_llen[f]); dfree(sl.d_nlen[f]); dfree(sl.d_lflag[f]);
        hfree(sl.h_loff[f]); hfree(sl.h_llen[f]); hfree(sl.h_nlen[f]); hfree(sl.h_lflag[f]);
    }
    for (int c = 0; c < 4; ++c) dfree(sl.d_col[c]);
    dfree(sl.d_bits); dfree(sl.d_code); dfree(sl.d_bins4); dfree(sl.d_idx);
    hfree(sl.h_code); hfree(sl.h_idx);
    sl.record_cap = 0;
    for (int f = 0; f < 2; ++f) {
        XMS_TRY(dalloc(s, sl.d_lend[f], lines)); XMS_TRY(dalloc(s, sl.d_lnext[f], lines));
        XMS_TRY(dalloc(s, sl.d_loff[f], n)); XMS_TRY(dalloc(s, sl.d_llen[f], n)); XMS_TRY(dalloc(s, sl.d_nlen[f], n));
        XMS_TRY(dalloc(s, sl.d_lflag[f], n));
        XMS_TRY(halloc(s, sl.h_loff[f], n)); XMS_TRY(halloc(s, sl.h_llen[f], n)); XMS_TRY(halloc(s, sl.h_nlen[f], n));
        XMS_TRY(halloc(s, sl.h_lflag[f], n));
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
        e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking);
        for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreate(&sl.ev[i]);
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_state, ST_WORDS * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_summary, SUM_WORDS * sizeof(uint64_t));
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_summary, SUM_WORDS * sizeof(uint64_t), hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_off_counts, 72 * sizeof(uint64_t));
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_off_counts, 72 * sizeof(uint64_t), hipHostMallocDefault);
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
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
        free_slot(sl);
        dfree(sl.d_state); dfree(sl.d_summary); dfree(sl.d_off_counts);
        hfree(sl.h_summary); hfree(sl.h_off_counts);
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
    XMS_TRY(grow_window(s, sl, std::max<uint64_t>(window_bytes, 1)));
    XMS_TRY(grow_records(s, sl, max_records));
    return XM_OK;
}

char *xm_strip_staging(xm_strip *s, int slot, int file)
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || file < 0 || file > 1) return nullptr;
    return s->slot[slot].h_text[file];
}

int xm_strip_upload(xm_strip *s, int slot, int file, uint64_t offset, uint64_t bytes)
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || file < 0 || file > 1) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    if (offset != sl.uploaded[file] || offset + bytes > sl.window_cap) return XM_ERR_INVALID_ARG;
    if (bytes == 0) return XM_OK;
    XMS_HIP(s, hipSetDevice(s->device));
    if (!sl.upload_timed) {
        XMS_HIP(s, hipEventRecord(sl.ev[0], sl.stream));
        sl.upload_timed = true;
    }
    XMS_HIP(s, hipMemcpyAsync(sl.d_text[file] + offset, sl.h_text[file] + offset, (size_t)bytes, hipMemcpyHostToDevice, sl.stream));
    sl.uploaded[file] = offset + bytes;
    return XM_OK;
}

int xm_strip_run(xm_strip *s, int slot, uint64_t len1, int eof1, uint64_t len2, int eof2,
                 int score_mode, int paired, int keep_halo, uint64_t max_records, xm_strip_block *out)
{
    if (!s || !out || slot < 0 || slot >= XMS_SLOTS || (score_mode != XMS_SCORE_AS_XS && score_mode != XMS_SCORE_AS_ZS))
        return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    const uint64_t sent[2] = {sl.uploaded[0], sl.uploaded[1]};
    const bool timed = sl.upload_timed;
    sl.uploaded[0] = sl.uploaded[1] = 0;
    sl.upload_timed = false;
    if (len1 > sl.window_cap || len2 > sl.window_cap || max_records == 0 || max_records > sl.record_cap || sent[0] > len1 ||
        sent[1] > len2)
        return XM_ERR_INVALID_ARG;
    XMS_HIP(s, hipSetDevice(s->device));
    const uint64_t len[2] = {len1, len2};
    const int eof[2] = {eof1, eof2};
    Job job;
    std::memset(&job, 0, sizeof job);
    uint32_t max_chunks = 1;
    for (int f = 0; f < 2; ++f) {
        FileView &v = job.f[f];
        v.text = sl.d_text[f];
        v.len = (uint32_t)len[f];
        // a trailing '\r' may be the first half of a "\r\n" that continues in the next window
        v.usable = (!eof[f] && len[f] > 0 && sl.h_text[f][len[f] - 1] == '\r') ? (uint32_t)len[f] - 1u : (uint32_t)len[f];
        v.eof = eof[f] ? 1u : 0u;
        v.n_chunks = (uint32_t)((len[f] + CHUNK - 1) / CHUNK);
        v.mask16 = sl.d_mask[f];
        v.chunk_cnt = sl.d_chunk_cnt[f];
        v.chunk_base = sl.d_chunk_base[f];
        v.lend = sl.d_lend[f];
        v.lnext = sl.d_lnext[f];
        max_chunks = std::max(max_chunks, v.n_chunks);
        job.loff[f] = sl.d_loff[f]; job.llen[f] = sl.d_llen[f]; job.nlen[f] = sl.d_nlen[f]; job.lflag[f] = sl.d_lflag[f];
    }
    job.state = sl.d_state;
    job.summary = sl.d_summary;
    job.cap_lines = (uint32_t)max_records + 1u;
    job.max_records = (uint32_t)max_records;
    job.xtag0 = score_mode == XMS_SCORE_AS_ZS ? 'Z' : 'X';
    job.paired = paired ? 1u : 0u;
    job.keep_halo = keep_halo ? 1u : 0u;
    for (int c = 0; c < 4; ++c) job.col[c] = sl.d_col[c];
    job.unit_bits = sl.d_bits;

    hipStream_t st = sl.stream;
    if (!timed) XMS_HIP(s, hipEventRecord(sl.ev[0], st));
    for (int f = 0; f < 2; ++f)                        // what xm_strip_upload has not sent yet
        if (len[f] > sent[f])
            XMS_HIP(s, hipMemcpyAsync(sl.d_text[f] + sent[f], sl.h_text[f] + sent[f], (size_t)(len[f] - sent[f]),
                                      hipMemcpyHostToDevice, st));
    XMS_HIP(s, hipEventRecord(sl.ev[1], st));
    XMS_HIP(s, hipMemsetAsync(sl.d_state, 0, ST_WORDS * sizeof(uint32_t), st));
    mark_kernel<<<dim3(max_chunks, 2), SB, 0, st>>>(job);
    chunk_scan_kernel<<<2, 1024, 0, st>>>(job);
    fill_kernel<<<dim3(max_chunks, 2), SB, 0, st>>>(job);
    // a line takes at least one byte of its window: no more records than that
    const uint64_t bound = std::min<uint64_t>(max_records, std::min(len1, len2) + 1);
    strip_kernel<<<(uint32_t)((bound + SB - 1) / SB), SB, 0, st>>>(job);
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
    out->n_lines1 = sum[SUM_L1];
    out->n_lines2 = sum[SUM_L2];
    if (n) {
        for (int f = 0; f < 2; ++f) {
            XMS_HIP(s, hipMemcpyAsync(sl.h_loff[f], sl.d_loff[f], n * 4, hipMemcpyDeviceToHost, st));
            XMS_HIP(s, hipMemcpyAsync(sl.h_llen[f], sl.d_llen[f], n * 4, hipMemcpyDeviceToHost, st));
            XMS_HIP(s, hipMemcpyAsync(sl.h_nlen[f], sl.d_nlen[f], n * 4, hipMemcpyDeviceToHost, st));
            XMS_HIP(s, hipMemcpyAsync(sl.h_lflag[f], sl.d_lflag[f], n, hipMemcpyDeviceToHost, st));
        }
        XMS_HIP(s, hipStreamSynchronize(st));
    }
    uint64_t n_exc = 0;
    for (int f = 0; f < 2; ++f)
        for (uint64_t k = 0; k < n; ++k) n_exc += (sl.h_lflag[f][k] & (XMS_LINE_EX_A | XMS_LINE_EX_X)) ? 1 : 0;
    out->n_exceptions = n_exc;
    out->line_off1 = sl.h_loff[0]; out->line_off2 = sl.h_loff[1];
    out->line_len1 = sl.h_llen[0]; out->line_len2 = sl.h_llen[1];
    out->norm_len1 = sl.h_nlen[0]; out->norm_len2 = sl.h_nlen[1];
    out->line_flags1 = sl.h_lflag[0]; out->line_flags2 = sl.h_lflag[1];
    (void)hipEventElapsedTime(&out->ms_upload, sl.ev[0], sl.ev[1]);
    (void)hipEventElapsedTime(&out->ms_kernels, sl.ev[1], sl.ev[2]);
    return XM_OK;
}

int xm_strip_classify(xm_strip *s, int slot, int mode, uint64_t n_records, int32_t min_score_floor,
                      const uint8_t **code, const uint32_t **idx, uint64_t bin_offsets[8], uint64_t counts[64])
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || !code || !idx || !bin_offsets || !counts) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    if (n_records > sl.record_cap) return XM_ERR_INVALID_ARG;
    *code = sl.h_code;
    *idx = sl.h_idx;
    std::memset(bin_offsets, 0, 8 * sizeof(uint64_t));
    std::memset(counts, 0, 64 * sizeof(uint64_t));
    if (n_records == 0) return XM_OK;
    XMS_HIP(s, hipSetDevice(s->device));
    hipStream_t st = sl.stream;
    const int rc = xm_classify_compact_dev(s->ctx, st, mode, n_records, sl.d_col[0], sl.d_col[1], sl.d_col[2], sl.d_col[3], sl.d_bits,
                                           min_score_floor, sl.d_code, sl.d_bins4, sl.d_idx, sl.d_off_counts, sl.d_off_counts + 8);
    if (rc != XM_OK) {
        s->last_error = xm_last_hip_error(s->ctx);
        return rc;
    }
    XMS_HIP(s, hipMemcpyAsync(sl.h_code, sl.d_code, n_records, hipMemcpyDeviceToHost, st));
    XMS_HIP(s, hipMemcpyAsync(sl.h_off_counts, sl.d_off_counts, 72 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    XMS_HIP(s, hipStreamSynchronize(st));
    const uint64_t units = sl.h_off_counts[7];
    if (units > n_records) return XM_ERR_HIP;
    if (units) {
        XMS_HIP(s, hipMemcpyAsync(sl.h_idx, sl.d_idx, units * 4, hipMemcpyDeviceToHost, st));
        XMS_HIP(s, hipStreamSynchronize(st));
    }
    std::memcpy(bin_offsets, sl.h_off_counts, 8 * sizeof(uint64_t));
    std::memcpy(counts, sl.h_off_counts + 8, 64 * sizeof(uint64_t));
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

int xm_strip_device_columns(xm_strip *s, int slot, void *ptrs[5])
{
    if (!s || slot < 0 || slot >= XMS_SLOTS || !ptrs) return XM_ERR_INVALID_ARG;
    Slot &sl = s->slot[slot];
    for (int c = 0; c < 4; ++c) ptrs[c] = sl.d_col[c];
    ptrs[4] = sl.d_bits;
    return XM_OK;
}

const char *xm_strip_last_error(const xm_strip *s) { return s ? s->last_error.c_str() : ""; }

}  // extern "C"
