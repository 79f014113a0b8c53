// xm_bam.cpp -- native BGZF/BAM -> SAM text decoder (C ABI in include/xenomapper_host.h, section "BAM input").
//
// The reference reads BAM by piping it through `samtools view` (xenomapper.py:48-93) and then treats the text
// exactly like SAM input.  This decoder produces that text -- the header block and one line per alignment in
// the layout `samtools view` prints -- so that everything downstream (column stripper, kernels, writer) is the
// SAM path unchanged and inherits its parity pins.  BGZF blocks are inflated in parallel (zlib, raw deflate).
#include "../../include/xenomapper_host.h"
#include "xm_pool.h"

#include <dlfcn.h>
#include <sys/mman.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

namespace {

struct BgzfBlock {
    uint64_t cdata_off;      // offset of the deflate stream in the file image
    uint32_t cdata_len;
    uint32_t isize;          // uncompressed size
    uint32_t crc;            // CRC-32 of the uncompressed bytes
};

inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// walk the gzip member headers: every BGZF block carries its own size in the 'BC' extra subfield
bool index_blocks(const uint8_t *d, uint64_t len, std::vector<BgzfBlock> &out, size_t max_blocks = (size_t)-1)
{
    uint64_t p = 0;
    while (p < len && out.size() < max_blocks) {
        if (p + 18 > len || d[p] != 0x1f || d[p + 1] != 0x8b || d[p + 2] != 8 || !(d[p + 3] & 4)) return false;
        const uint32_t xlen = le16(d + p + 10);
        if (p + 12 + xlen > len) return false;
        uint32_t bsize = 0;
        bool found = false;
        for (uint64_t q = p + 12; q + 4 <= p + 12 + xlen;) {
            const uint32_t slen = le16(d + q + 2);
            if (d[q] == 'B' && d[q + 1] == 'C' && slen == 2 && q + 6 <= len) { bsize = le16(d + q + 4); found = true; }
            q += 4 + slen;
        }
        if (!found) return false;
        const uint64_t total = (uint64_t)bsize + 1;
        if (total < 12 + xlen + 8 || p + total > len) return false;
        BgzfBlock b;
        b.cdata_off = p + 12 + xlen;
        b.cdata_len = (uint32_t)(total - 12 - xlen - 8);
        b.isize = le32(d + p + total - 4);
        b.crc = le32(d + p + total - 8);
        out.push_back(b);
        p += total;
    }
    return true;
}

// libdeflate (raw-deflate decoder and CRC-32 two to three times faster than zlib's) is used when the shared library
// is installed -- looked up at run time, its four entry points declared here because the image ships no header --
// and zlib otherwise.  XMH_NO_LIBDEFLATE=1 forces zlib.
struct LibDeflate {
    void *(*alloc)(void) = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;
    void (*release)(void *) = nullptr;
    bool ok = false;
    LibDeflate()
    {
        if (getenv("XMH_NO_LIBDEFLATE")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)(void))dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        crc = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        ok = alloc && decompress && crc && release;
    }
};
const LibDeflate &libdeflate()
{
    static const LibDeflate lib;
    return lib;
}

// one decoder per worker, reused from block to block; the BGZF trailer's CRC-32 is checked like htslib does
struct Inflater {
    z_stream zs;
    bool ready = false;
    void *fast = nullptr;
    Inflater() { memset(&zs, 0, sizeof zs); }
    ~Inflater()
    {
        if (ready) inflateEnd(&zs);
        if (fast) libdeflate().release(fast);
    }
    Inflater(const Inflater &) = delete;
    Inflater &operator=(const Inflater &) = delete;

    bool block(const uint8_t *src, uint32_t slen, uint8_t *dst, uint32_t dlen, uint32_t crc)
    {
        if (dlen == 0) return true;                                  // e.g. the end-of-file marker block
        const LibDeflate &ld = libdeflate();
        if (ld.ok) {
            if (!fast && !(fast = ld.alloc())) return false;
            size_t got = 0;
            if (ld.decompress(fast, src, slen, dst, dlen, &got) != 0 || got != dlen) return false;
            return ld.crc(0, dst, dlen) == crc;
        }
        if (!ready) {
            if (inflateInit2(&zs, -15) != Z_OK) return false;
            ready = true;
        } else if (inflateReset(&zs) != Z_OK) {
            return false;
        }
        zs.next_in = const_cast<Bytef *>(src);
        zs.avail_in = slen;
        zs.next_out = dst;
        zs.avail_out = dlen;
        const int rc = inflate(&zs, Z_FINISH);
        if (rc != Z_STREAM_END || zs.total_out != dlen) return false;
        return (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, dlen) == crc;
    }
};

// "00" .. "99": two digits per division (a record prints some fifteen numbers; this is a quarter of its time)
struct DigitPairs {
    char d[100][2];
    DigitPairs() { for (int i = 0; i < 100; ++i) { d[i][0] = (char)('0' + i / 10); d[i][1] = (char)('0' + i % 10); } }
};
const DigitPairs DIGITS;

inline char *put_uint(char *o, uint64_t v)
{
    if (v < 10) { *o++ = (char)('0' + v); return o; }
    if (v < 100) { memcpy(o, DIGITS.d[v], 2); return o + 2; }
    if (v < 1000) { *o++ = (char)('0' + v / 100); memcpy(o, DIGITS.d[v % 100], 2); return o + 2; }
    if (v < 10000) { memcpy(o, DIGITS.d[v / 100], 2); memcpy(o + 2, DIGITS.d[v % 100], 2); return o + 4; }
    char tmp[20];
    int n = 20;
    while (v >= 100) { n -= 2; memcpy(tmp + n, DIGITS.d[v % 100], 2); v /= 100; }
    if (v >= 10) { n -= 2; memcpy(tmp + n, DIGITS.d[v], 2); }
    else tmp[--n] = (char)('0' + v);
    memcpy(o, tmp + n, (size_t)(20 - n));
    return o + (20 - n);
}

inline char *put_int(char *o, int64_t v)
{
    if (v < 0) { *o++ = '-'; return put_uint(o, (uint64_t)(-(v + 1)) + 1); }
    return put_uint(o, (uint64_t)v);
}

inline char *put_real(char *o, double v) { return o + snprintf(o, 40, "%g", v); }

inline char *put_str(char *o, const std::string &s)
{
    memcpy(o, s.data(), s.size());
    return o + s.size();
}

struct SeqTable {                      // one packed byte -> its two bases
    char pair[256][2];
    SeqTable()
    {
        for (int b = 0; b < 256; ++b) { pair[b][0] = "=ACMGRSVTWYHKDBN"[b >> 4]; pair[b][1] = "=ACMGRSVTWYHKDBN"[b & 15]; }
    }
};
const SeqTable SEQ;

// SEQ, 16 packed bytes -> 32 bases at a time where the CPU has SSSE3 (both nibbles through one 16-entry shuffle table, interleaved);
// the table loop for the rest and for other CPUs.  (Printing is what the file path's main thread does for BAM input on the GPU
// front end: a third of a record's time went into this loop and the quality loop behind it.)
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("ssse3"))) static char *put_seq_ssse3(char *o, const uint8_t *packed, uint32_t l_seq)
{
    const __m128i lut = _mm_setr_epi8('=', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N');
    const __m128i low = _mm_set1_epi8(0x0F);
    uint32_t k = 0;
    for (; k + 32 <= l_seq; k += 32) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(packed + k / 2));
        const __m128i hi = _mm_shuffle_epi8(lut, _mm_and_si128(_mm_srli_epi16(v, 4), low));
        const __m128i lo = _mm_shuffle_epi8(lut, _mm_and_si128(v, low));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(o + k), _mm_unpacklo_epi8(hi, lo));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(o + k + 16), _mm_unpackhi_epi8(hi, lo));
    }
    for (; k + 1 < l_seq; k += 2) memcpy(o + k, SEQ.pair[packed[k / 2]], 2);
    if (l_seq & 1) o[l_seq - 1] = SEQ.pair[packed[l_seq / 2]][0];
    return o + l_seq;
}
static const bool HAVE_SSSE3 = __builtin_cpu_supports("ssse3");
#else
static const bool HAVE_SSSE3 = false;
#endif

inline char *put_seq(char *o, const uint8_t *packed, uint32_t l_seq)
{
#if defined(__x86_64__)
    if (HAVE_SSSE3) return put_seq_ssse3(o, packed, l_seq);
#endif
    for (uint32_t k = 0; k + 1 < l_seq; k += 2) { memcpy(o, SEQ.pair[packed[k / 2]], 2); o += 2; }
    if (l_seq & 1) *o++ = SEQ.pair[packed[l_seq / 2]][0];
    return o;
}

}  // namespace

// inflated bytes not yet consumed: [pos, end) of one reused, never zero-filled buffer (a std::vector would memset --
// and page-fault -- every 64 MB batch on one thread before the workers get to write it)
struct ByteStream {
    std::unique_ptr<uint8_t[]> buf;
    size_t cap = 0, pos = 0, end = 0;
    uint8_t *data() { return buf.get(); }
    const uint8_t *data() const { return buf.get(); }
    size_t size() const { return end; }
    void compact()                       // drop what has been consumed; offsets into the stream change
    {
        if (pos == 0) return;
        memmove(buf.get(), buf.get() + pos, end - pos);
        end -= pos;
        pos = 0;
    }
    void grow_to(size_t new_end)         // room for [0, new_end); offsets stay valid
    {
        if (new_end <= cap) return;
        const size_t nc = std::max(new_end, cap + cap / 2);
        std::unique_ptr<uint8_t[]> nb(new uint8_t[nc]);
        if (end) memcpy(nb.get(), buf.get(), end);
        buf.swap(nb);
        cap = nc;
    }
};

struct BamWorker {
    Inflater inf;
    std::unique_ptr<char[]> text;       // formatted lines of this worker's records (uninitialised storage, reused)
    size_t text_cap = 0, text_len = 0, n_rec = 0;
    std::vector<uint64_t> rec;          // stream offsets of this worker's records in the current batch
    std::vector<xmh_pre> pre;           // what the stripper would dig out of each line again (xmh_bam_read_pre)
    std::vector<uint32_t> ops;          // CIGAR operations the pre records point into
    bool ok = true;
};

struct xmh_bam {
    const uint8_t *data;
    uint64_t len;
    int n_threads;
    std::unique_ptr<xmh::Pool> pool;
    std::vector<BamWorker> workers;
    std::vector<BgzfBlock> blocks;
    size_t next_block = 0;              // first block not yet inflated
    ByteStream stream;
    double text_per_byte = 2.0;         // text bytes per binary byte seen so far (sizes the next batch)
    std::string header;                 // SAM header text
    std::vector<std::string> ref_names;
    bool header_done = false;
    std::vector<char> pending;          // formatted lines that did not fit the caller's buffer, from pending_pos
    size_t pending_pos = 0;
    std::vector<xmh_pre> pending_pre;   // their descriptions, from pending_pre_pos (ops_at counts into pending_ops)
    std::vector<uint32_t> pending_ops;
    size_t pending_pre_pos = 0;
    bool want_pre = false;              // the current read call asked for descriptions
    std::vector<uint8_t> ref_weird;     // reference names the text rules might split
    bool header_only = false;           // xmh_bam_open_header: only the blocks the header needed are indexed -- no reading

    // make at least `want` bytes available after stream.pos (or reach the end of the file): one parallel pass
    // over as many blocks as that takes
    bool fill(size_t want)
    {
        if (stream.size() - stream.pos >= want || next_block >= blocks.size()) return true;
        std::vector<uint64_t> off;
        uint64_t acc = stream.size();
        size_t nb = 0;
        const size_t least = (size_t)n_threads * 4;              // enough blocks to keep every worker busy
        while (next_block + nb < blocks.size() && (acc - stream.pos < want || nb < least)) {
            off.push_back(acc);
            acc += blocks[next_block + nb].isize;
            ++nb;
        }
        stream.grow_to((size_t)acc);
        stream.end = (size_t)acc;
        {                                                        // map the compressed bytes of these blocks in bulk
            const uintptr_t page = 4096, lo = (uintptr_t)(data + blocks[next_block].cdata_off) & ~(page - 1);
            const BgzfBlock &last = blocks[next_block + nb - 1];
            const uintptr_t hi = ((uintptr_t)(data + last.cdata_off + last.cdata_len) + page - 1) & ~(page - 1);
            (void)madvise((void *)lo, (size_t)(hi - lo), MADV_POPULATE_READ);
        }
        for (auto &w : workers) w.ok = true;
        const int nt = (int)std::min<size_t>((size_t)pool->size(), nb);
        pool->run(nt, [&](int t) {
            BamWorker &w = workers[(size_t)t];
            for (size_t i = (size_t)t; i < nb; i += (size_t)nt) {
                const BgzfBlock &b = blocks[next_block + i];
                if (!w.inf.block(data + b.cdata_off, b.cdata_len, stream.data() + off[i], b.isize, b.crc)) w.ok = false;
            }
        });
        for (auto &w : workers)
            if (!w.ok) return false;
        next_block += nb;
        return true;
    }
    size_t avail() const { return stream.size() - stream.pos; }
    const uint8_t *cur() const { return stream.data() + stream.pos; }
};

namespace {

// a byte the text rules (Python's str.split(), ASCII input only) might treat specially
inline bool odd_byte(uint8_t c) { return c <= 0x20 || c >= 0x7F; }
inline bool odd_bytes(const uint8_t *s, size_t n)
{
    uint8_t any = 0;
    for (size_t i = 0; i < n; ++i) any |= (uint8_t)(odd_byte(s[i]) ? 1 : 0);
    return any != 0;
}
inline bool has2(const uint8_t *s, size_t n, char c0, char c1)
{
    for (size_t i = 0; i + 1 < n; ++i)
        if (s[i] == (uint8_t)c0 && s[i + 1] == (uint8_t)c1) return true;
    return false;
}

bool read_header(xmh_bam *b)
{
    if (!b->fill(12) || b->avail() < 12 || memcmp(b->cur(), "BAM\1", 4) != 0) return false;
    const uint32_t l_text = le32(b->cur() + 4);
    if (!b->fill(12 + (size_t)l_text) || b->avail() < 12 + (size_t)l_text) return false;
    b->header.assign((const char *)b->cur() + 8, l_text);
    while (!b->header.empty() && b->header.back() == '\0') b->header.pop_back();
    const uint32_t n_ref = le32(b->cur() + 8 + l_text);
    b->stream.pos += 12 + l_text;
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (!b->fill(4) || b->avail() < 4) return false;
        const uint32_t l_name = le32(b->cur());
        if (!b->fill(8 + (size_t)l_name) || b->avail() < 8 + (size_t)l_name) return false;
        std::string name((const char *)b->cur() + 4, l_name);
        while (!name.empty() && name.back() == '\0') name.pop_back();
        b->ref_names.push_back(name);
        b->ref_weird.push_back((uint8_t)((name.empty() || odd_bytes((const uint8_t *)name.data(), name.size())) ? 1 : 0));
        b->stream.pos += 8 + l_name;
    }
    b->header_done = true;
    return true;
}

// bytes of one optional-field value of type t starting at r[p] (the tag and type bytes already passed), or 0 when the
// record ends inside it
inline uint64_t aux_value_bytes(const uint8_t *r, uint64_t p, uint32_t size, char t)
{
    auto elem = [](char e) -> uint64_t {
        switch (e) {
        case 'A': case 'c': case 'C': return 1;
        case 's': case 'S': return 2;
        case 'i': case 'I': case 'f': return 4;
        case 'd': return 8;
        default: return 0;
        }
    };
    if (t == 'Z' || t == 'H') {
        const size_t l = strnlen((const char *)r + p, (size_t)(size - p));
        return p + l < size ? l + 1 : 0;
    }
    if (t == 'B') {
        if (p + 5 > size) return 0;
        const uint64_t e = elem((char)r[p]), n = le32(r + p + 1);
        return (e && p + 5 + e * n <= size) ? 5 + e * n : 0;
    }
    const uint64_t e = elem(t);
    return (e && p + e <= size) ? e : 0;
}

// A CIGAR of more than 65 535 operations does not fit the 16-bit count of a BAM record: writers store the placeholder
// "<l_seq>S<ref_len>N" and the real operations in a CG:B:I field, and htslib moves them back when it reads the record
// (sam.c, bam_tag2cigar: mapped record, first operation a soft clip of the whole read, CG of type B,I with at least
// n_cigar and fewer than 2^29 values), so `samtools view` prints the real CIGAR and no CG field.  Returns the offset
// of the CG field's tag bytes (and its value count) when that applies, else 0.
inline uint64_t real_cigar_field(const uint8_t *r, uint32_t size, uint64_t aux0, uint32_t n_cigar, uint32_t first_op,
                                 uint32_t l_seq, int32_t ref_id, int32_t pos, uint32_t *count)
{
    if (n_cigar == 0 || ref_id < 0 || pos < 0 || (first_op & 15u) != 4u || (first_op >> 4) != l_seq) return 0;
    uint64_t p = aux0;
    while (p + 3 <= size) {
        const char t = (char)r[p + 2];
        const uint64_t len = aux_value_bytes(r, p + 3, size, t);
        if (!len) return 0;
        if (r[p] == 'C' && r[p + 1] == 'G') {                             // the first CG field decides, as bam_aux_get does
            if (t != 'B' || r[p + 3] != 'I') return 0;
            const uint32_t n = le32(r + p + 4);
            if (n < n_cigar || n >= (1u << 29)) return 0;
            *count = n;
            return p;
        }
        p += 3 + len;
    }
    return 0;
}

// the four tags of the tag_func plugins, matched as the text rules match them (header: xmh_bam_read_pre)
struct TagScan {
    uint32_t n[4] = {0, 0, 0, 0};             // matches of AS, XS, ZS, NM
    int32_t v[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
    uint8_t ex[4] = {0, 0, 0, 0};
    void hit(int k, bool is_int, int64_t val)
    {
        if (++n[k] != 1) return;                                  // only the first match carries a value
        if (is_int && val >= -2147483647ll && val <= 2147483647ll) v[k] = (int32_t)val;
        else ex[k] = XMH_EX_NONINT;
    }
    // a match whose value is not an integer-typed field: what the text rules make of the printed field [s, e) -- the text
    // behind its last ':' as a plain integer ("AS:H:0", "AS:A:7", a float that prints as "3"), else not vouched for
    void hit_text(int k, const char *s, const char *e)
    {
        if (++n[k] != 1) return;
        const char *b = s;
        for (const char *c = s; c < e; ++c)
            if (*c == ':') b = c + 1;
        bool neg = false;
        if (b < e && (*b == '-' || *b == '+')) { neg = *b == '-'; ++b; }
        uint64_t val = 0;
        bool ok = b < e && e - b <= 10;
        for (const char *c = b; ok && c < e; ++c) {
            if (*c < '0' || *c > '9') ok = false;
            else val = val * 10 + (uint64_t)(*c - '0');
        }
        if (ok && val <= 2147483647ull) v[k] = neg ? -(int32_t)val : (int32_t)val;
        else ex[k] = XMH_EX_NONINT;
    }
    void finish(xmh_pre &q) const
    {
        q.as = v[0]; q.xs = v[1]; q.zs = v[2]; q.nm = v[3];
        q.ex_as = n[0] > 1 ? XMH_EX_DUP : ex[0];
        q.ex_xs = n[1] > 1 ? XMH_EX_DUP : ex[1];
        q.ex_zs = n[2] > 1 ? XMH_EX_DUP : ex[2];
        q.ex_nm = ex[3];                                          // NM: the first match decides, no duplicate rule (:247-250)
    }
};
const char TAGS[4][2] = {{'A', 'S'}, {'X', 'S'}, {'Z', 'S'}, {'N', 'M'}};

// one alignment record (without its block_size word) -> one SAM line at o, the way `samtools view` prints it.
// Returns the end of the line, or nullptr for a malformed record.  The caller guarantees room for 5 * size + 128
// bytes plus the longest reference name twice.
// q / qops (may be null): the line's description for xmh_parse_pre and the array its CIGAR operations are appended to.
char *format_record(const xmh_bam *b, const uint8_t *r, uint32_t size, char *o, xmh_pre *q = nullptr, std::vector<uint32_t> *qops = nullptr)
{
    if (size < 32) return nullptr;
    char *const line0 = o;
    bool weird = false;
    TagScan scan;
    const int32_t ref_id = (int32_t)le32(r), pos = (int32_t)le32(r + 4);
    const uint32_t l_read_name = r[8], mapq = r[9], n_cigar = le16(r + 12), flag = le16(r + 14);
    const uint32_t l_seq = le32(r + 16);
    const int32_t next_ref = (int32_t)le32(r + 20), next_pos = (int32_t)le32(r + 24), tlen = (int32_t)le32(r + 28);
    uint64_t p = 32;
    const uint64_t need = p + l_read_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + l_seq;
    if (need > size || l_read_name == 0) return nullptr;
    const size_t nl = strnlen((const char *)r + p, l_read_name);
    memcpy(o, r + p, nl);
    if (q) {
        weird |= nl == 0 || nl > 0xFFFF || odd_bytes(r + p, nl);
        q->name_len = (uint16_t)nl;
        if (ref_id >= 0 && (size_t)ref_id < b->ref_weird.size()) weird |= b->ref_weird[(size_t)ref_id] != 0;
        if (next_ref >= 0 && (size_t)next_ref < b->ref_weird.size()) weird |= b->ref_weird[(size_t)next_ref] != 0;
    }
    o += nl;
    p += l_read_name;
    *o++ = '\t'; o = put_uint(o, flag);
    *o++ = '\t';
    if (ref_id < 0 || (size_t)ref_id >= b->ref_names.size()) *o++ = '*'; else o = put_str(o, b->ref_names[(size_t)ref_id]);
    *o++ = '\t'; o = put_int(o, (int64_t)pos + 1);
    *o++ = '\t'; o = put_uint(o, mapq);
    *o++ = '\t';
    uint32_t cg_count = 0;
    const uint64_t cg_at = n_cigar ? real_cigar_field(r, size, need, n_cigar, le32(r + p), l_seq, ref_id, pos, &cg_count) : 0;
    const uint8_t *ops = cg_at ? r + cg_at + 8 : r + p;
    const uint32_t n_ops = cg_at ? cg_count : n_cigar;
    if (n_ops == 0) *o++ = '*';
    if (q) { q->ops_at = (uint32_t)qops->size(); }
    for (uint32_t k = 0; k < n_ops; ++k) {
        const uint32_t v = le32(ops + 4ull * k);
        o = put_uint(o, v >> 4);
        *o++ = "MIDNSHP=XB??????"[v & 15];
        if (q && (v & 15u) <= 8u) qops->push_back(v);               // re.findall(r'([0-9]+)([MIDNSHPX=])'): other codes print letters it skips
    }
    if (q) q->n_ops = (uint32_t)qops->size() - q->ops_at;
    p += 4ull * n_cigar;
    *o++ = '\t';
    if (next_ref < 0 || (size_t)next_ref >= b->ref_names.size()) *o++ = '*';
    else if (next_ref == ref_id) *o++ = '=';
    else o = put_str(o, b->ref_names[(size_t)next_ref]);
    *o++ = '\t'; o = put_int(o, (int64_t)next_pos + 1);
    *o++ = '\t'; o = put_int(o, tlen);
    *o++ = '\t';
    if (l_seq == 0) *o++ = '*';
    o = put_seq(o, r + p, l_seq);
    p += ((uint64_t)l_seq + 1) / 2;
    *o++ = '\t';
    if (l_seq == 0 || r[p] == 0xFF) *o++ = '*';
    else {
        uint8_t hi = 0;
        uint32_t k = 0;
#if defined(__x86_64__)
        {   // 16 qualities at a time (SSE2 is part of x86-64)
            __m128i acc = _mm_setzero_si128();
            const __m128i add = _mm_set1_epi8(33);
            for (; k + 16 <= l_seq; k += 16) {
                const __m128i c = _mm_add_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(r + p + k)), add);
                _mm_storeu_si128(reinterpret_cast<__m128i *>(o + k), c);
                acc = _mm_or_si128(acc, c);
            }
            hi = (uint8_t)(_mm_movemask_epi8(acc) ? 0x80 : 0);
        }
#endif
        for (; k < l_seq; ++k) { const uint8_t c = (uint8_t)(r[p + k] + 33); o[k] = (char)c; hi |= c; }
        weird |= (hi & 0x80) != 0;                                   // a quality above 94: not ASCII any more
        o += l_seq;
    }
    p += l_seq;
    // optional fields
    while (p + 3 <= size) {
        if (cg_at && p == cg_at) {                                      // its operations were printed as the CIGAR
            p += 8 + 4ull * cg_count;
            continue;
        }
        *o++ = '\t';
        *o++ = (char)r[p]; *o++ = (char)r[p + 1]; *o++ = ':';
        const char type = (char)r[p + 2];
        const uint8_t t0 = r[p], t1 = r[p + 1];
        char *const field0 = o - 3;
        p += 3;
        int by_name = -1;                                               // a tag whose value the printed text decides
        if (q) {
            weird |= odd_byte(t0) || odd_byte(t1);
            int64_t ival = 0;
            bool is_int = true;
            switch (type) {
            case 'c': ival = p + 1 <= size ? (int8_t)r[p] : 0; break;
            case 'C': ival = p + 1 <= size ? r[p] : 0; break;
            case 's': ival = p + 2 <= size ? (int16_t)le16(r + p) : 0; break;
            case 'S': ival = p + 2 <= size ? le16(r + p) : 0; break;
            case 'i': ival = p + 4 <= size ? (int32_t)le32(r + p) : 0; break;
            case 'I': ival = p + 4 <= size ? (int64_t)le32(r + p) : 0; break;
            default: is_int = false;
            }
            if (type != 'Z' && type != 'H') {                           // only the tag itself can hold the two letters
                for (int k = 0; k < 4; ++k)
                    if (t0 == (uint8_t)TAGS[k][0] && t1 == (uint8_t)TAGS[k][1]) {
                        if (is_int) scan.hit(k, true, ival); else by_name = k;
                    }
                if (type == 'A' && p + 1 <= size) weird |= odd_byte(r[p]);
            }
        }
        auto scalar = [&](char t) -> bool {                          // appends one value of type t, advances p
            switch (t) {
            case 'A': if (p + 1 > size) return false; *o++ = (char)r[p]; p += 1; return true;
            case 'c': if (p + 1 > size) return false; o = put_int(o, (int8_t)r[p]); p += 1; return true;
            case 'C': if (p + 1 > size) return false; o = put_uint(o, r[p]); p += 1; return true;
            case 's': if (p + 2 > size) return false; o = put_int(o, (int16_t)le16(r + p)); p += 2; return true;
            case 'S': if (p + 2 > size) return false; o = put_uint(o, le16(r + p)); p += 2; return true;
            case 'i': if (p + 4 > size) return false; o = put_int(o, (int32_t)le32(r + p)); p += 4; return true;
            case 'I': if (p + 4 > size) return false; o = put_uint(o, le32(r + p)); p += 4; return true;
            case 'f': {
                if (p + 4 > size) return false;
                float f;
                const uint32_t u = le32(r + p);
                memcpy(&f, &u, 4);
                o = put_real(o, f);
                p += 4;
                return true;
            }
            case 'd': {
                if (p + 8 > size) return false;
                double dv;
                const uint64_t u = (uint64_t)le32(r + p) | ((uint64_t)le32(r + p + 4) << 32);
                memcpy(&dv, &u, 8);
                o = put_real(o, dv);
                p += 8;
                return true;
            }
            default: return false;
            }
        };
        if (type == 'A') { *o++ = 'A'; *o++ = ':'; if (!scalar('A')) return nullptr; }
        else if (type == 'c' || type == 'C' || type == 's' || type == 'S' || type == 'i' || type == 'I') {
            *o++ = 'i'; *o++ = ':';
            if (!scalar(type)) return nullptr;
        }
        else if (type == 'f') { *o++ = 'f'; *o++ = ':'; if (!scalar('f')) return nullptr; }
        else if (type == 'd') { *o++ = 'd'; *o++ = ':'; if (!scalar('d')) return nullptr; }
        else if (type == 'Z' || type == 'H') {
            *o++ = type; *o++ = ':';
            const size_t l = strnlen((const char *)r + p, (size_t)(size - p));
            if (p + l >= size) return nullptr;
            memcpy(o, r + p, l);
            o += l;
            if (q) {                                                    // the printed field "TG:Z:value" as the text rules see it
                weird |= odd_bytes(r + p, l);
                const size_t fl = (size_t)(o - field0);
                for (int k = 0; k < 4; ++k)
                    if (has2((const uint8_t *)field0, fl, TAGS[k][0], TAGS[k][1])) scan.hit_text(k, field0, o);
            }
            p += l + 1;
        } else if (type == 'B') {
            if (p + 5 > size) return nullptr;
            const char sub = (char)r[p];
            const uint32_t cnt = le32(r + p + 1);
            p += 5;
            *o++ = 'B'; *o++ = ':'; *o++ = sub;
            for (uint32_t k = 0; k < cnt; ++k) { *o++ = ','; if (!scalar(sub)) return nullptr; }
        } else return nullptr;
        if (by_name >= 0) scan.hit_text(by_name, field0, o);
    }
    if (q) {
        q->line_len = (uint32_t)(o - line0);
        q->flags = weird ? XMH_PRE_WEIRD : 0;
        q->pad_ = 0;
        scan.finish(*q);
    }
    *o++ = '\n';
    return p == size ? o : nullptr;
}

// where a read call puts the descriptions of the lines it hands out (null: none wanted)
struct PreOut {
    xmh_pre *pre;
    uint64_t pre_cap, n_pre;
    uint32_t *ops;
    uint64_t ops_cap, n_ops;
};

// hand out pending text in whole lines; returns bytes copied (0 when the first pending line does not fit)
uint64_t take_pending(xmh_bam *b, char *dst, uint64_t cap, PreOut *po)
{
    const size_t have = b->pending.size() - b->pending_pos;
    if (have == 0) return 0;
    size_t take = have;
    if (po) {
        // line by line: the text, the description and the operations must all fit
        take = 0;
        size_t k = b->pending_pre_pos;
        while (k < b->pending_pre.size()) {
            const xmh_pre &q = b->pending_pre[k];
            if (take + q.line_len + 1 > cap || po->n_pre + (k - b->pending_pre_pos) + 1 > po->pre_cap) break;
            const uint64_t ops_so_far = (uint64_t)q.ops_at + q.n_ops - b->pending_pre[b->pending_pre_pos].ops_at;
            if (po->n_ops + ops_so_far > po->ops_cap) break;
            take += (size_t)q.line_len + 1;
            ++k;
        }
        const size_t n_lines = k - b->pending_pre_pos;
        if (n_lines) {
            const uint32_t ops0 = b->pending_pre[b->pending_pre_pos].ops_at;
            const xmh_pre &last = b->pending_pre[k - 1];
            const uint32_t ops1 = last.ops_at + last.n_ops;
            for (size_t i = 0; i < n_lines; ++i) {
                xmh_pre q = b->pending_pre[b->pending_pre_pos + i];
                q.ops_at = q.ops_at - ops0 + (uint32_t)po->n_ops;
                po->pre[po->n_pre + i] = q;
            }
            if (ops1 > ops0) memcpy(po->ops + po->n_ops, b->pending_ops.data() + ops0, (size_t)(ops1 - ops0) * 4);
            po->n_pre += n_lines;
            po->n_ops += ops1 - ops0;
            b->pending_pre_pos = k;
        }
    } else if (take > cap) {
        const char *base = b->pending.data() + b->pending_pos;
        const void *nl = cap ? memrchr(base, '\n', (size_t)cap) : nullptr;
        take = nl ? (size_t)((const char *)nl - base) + 1 : 0;
    }
    memcpy(dst, b->pending.data() + b->pending_pos, take);
    b->pending_pos += take;
    if (b->pending_pos == b->pending.size()) {
        b->pending.clear(); b->pending_pos = 0;
        b->pending_pre.clear(); b->pending_ops.clear(); b->pending_pre_pos = 0;
    }
    return take;
}


// One batch: inflate the next blocks (about `budget` bytes), find the record boundaries and print the records.
// Worker t owns a contiguous run of blocks.  Record boundaries form a chain (each record's length word gives the
// next record), which one thread following it through 100 MB written by fifteen other cores pays ~100 ns a hop for;
// here every worker follows the chain only through the bytes it has just inflated itself, starting where its
// left-hand neighbour's last record ends (handed over through `chain`), and then prints the records it found.
// A record belongs to the worker its first byte lies with, even when it runs on into later ranges.
// On return: text of worker t in workers[t].text (length text_len), stream.pos at the first record not printed,
// *n_records = records printed, *n_workers = workers used.  false = corrupt input.
bool produce_batch(xmh_bam *b, size_t budget, size_t *n_records, int *n_workers)
{
    ByteStream &st = b->stream;
    st.compact();                                                  // leftover (a partial record) moves to the front
    *n_records = 0;
    *n_workers = 0;
    std::vector<uint64_t> off;
    uint64_t acc = st.end;
    size_t nb = 0;
    while (b->next_block + nb < b->blocks.size() && acc - st.end < budget) {
        off.push_back(acc);
        acc += b->blocks[b->next_block + nb].isize;
        ++nb;
    }
    if (nb == 0 && st.end == 0) return true;                       // nothing left at all
    const size_t old_end = st.end;
    st.grow_to((size_t)acc);
    st.end = (size_t)acc;
    if (nb) {                                                      // map the compressed bytes of these blocks in bulk
        const uintptr_t page = 4096, lo = (uintptr_t)(b->data + b->blocks[b->next_block].cdata_off) & ~(page - 1);
        const BgzfBlock &last = b->blocks[b->next_block + nb - 1];
        const uintptr_t hi = ((uintptr_t)(b->data + last.cdata_off + last.cdata_len) + page - 1) & ~(page - 1);
        (void)madvise((void *)lo, (size_t)(hi - lo), MADV_POPULATE_READ);
    }
    // contiguous block runs of about equal inflated size; worker 0 also owns the leftover bytes
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)b->pool->size(), nb));
    std::vector<size_t> first_block((size_t)nt + 1, nb);
    std::vector<uint64_t> range((size_t)nt + 1, acc);              // worker t scans stream offsets [range[t], range[t+1])
    first_block[0] = 0;
    range[0] = 0;
    {
        const uint64_t per = (acc - old_end + (uint64_t)nt - 1) / (uint64_t)nt;
        size_t i = 0;
        for (int t = 1; t < nt; ++t) {
            while (i < nb && off[i] - old_end < (uint64_t)t * per) ++i;
            first_block[(size_t)t] = i;
            range[(size_t)t] = i < nb ? off[i] : acc;
        }
    }
    for (auto &w : b->workers) { w.ok = true; w.text_len = 0; w.n_rec = 0; w.pre.clear(); w.ops.clear(); }
    static const bool profile = getenv("XMH_PROFILE") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    uint8_t *sp = st.data();
    b->pool->run(nt, [&](int t) {
        BamWorker &w = b->workers[(size_t)t];
        for (size_t i = first_block[(size_t)t]; i < first_block[(size_t)t + 1]; ++i) {
            const BgzfBlock &blk = b->blocks[b->next_block + i];
            if (!w.inf.block(b->data + blk.cdata_off, blk.cdata_len, sp + off[i], blk.isize, blk.crc)) w.ok = false;
        }
    });
    for (int t = 0; t < nt; ++t)
        if (!b->workers[(size_t)t].ok) return false;
    b->next_block += nb;
    const auto tp1 = std::chrono::steady_clock::now();

    size_t longest_ref = 0;
    for (auto &name : b->ref_names) longest_ref = std::max(longest_ref, name.size());
    const uint64_t unset = ~(uint64_t)0;
    std::vector<std::atomic<uint64_t>> chain((size_t)nt + 1);
    for (auto &c : chain) c.store(unset, std::memory_order_relaxed);
    chain[0].store(0, std::memory_order_release);
    const uint64_t end = st.end;
    b->pool->run(nt, [&](int t) {
        BamWorker &w = b->workers[(size_t)t];
        uint64_t p;
        while ((p = chain[(size_t)t].load(std::memory_order_acquire)) == unset) std::this_thread::yield();
        const uint64_t first = p, stop = range[(size_t)t + 1];
        w.rec.clear();
        while (p < stop && end - p >= 4) {
            const uint64_t next = p + 4 + (uint64_t)le32(sp + p);
            if (next > end) break;                                   // the record continues in a later batch
            w.rec.push_back(p);
            p = next;
        }
        chain[(size_t)t + 1].store(p, std::memory_order_release);
        if (w.rec.empty()) return;
        // text a record can grow to: the worst ratios are a B:c array element (1 byte -> "-128,"), a CIGAR operation
        // (4 -> 11) and a packed base pair (1 -> 2), so 5x; the fixed fields add < 128 and two reference names
        const size_t room = 5 * (size_t)(p - first) + w.rec.size() * (128 + 2 * longest_ref);
        if (w.text_cap < room) {
            w.text.reset(new char[room]);
            w.text_cap = room;
        }
        char *o = w.text.get();
        w.pre.clear();
        w.ops.clear();
        if (b->want_pre) {
            w.pre.resize(w.rec.size());
            size_t k = 0;
            for (uint64_t at : w.rec) {
                o = format_record(b, sp + at + 4, le32(sp + at), o, &w.pre[k++], &w.ops);
                if (!o) { w.ok = false; return; }
            }
        } else {
            for (uint64_t at : w.rec) {
                o = format_record(b, sp + at + 4, le32(sp + at), o);
                if (!o) { w.ok = false; return; }
            }
        }
        w.text_len = (size_t)(o - w.text.get());
        w.n_rec = w.rec.size();
    });
    size_t n = 0, text = 0;
    for (int t = 0; t < nt; ++t) {
        if (!b->workers[(size_t)t].ok) return false;
        n += b->workers[(size_t)t].n_rec;
        text += b->workers[(size_t)t].text_len;
    }
    const uint64_t done = chain[(size_t)nt].load(std::memory_order_acquire);
    if (profile) {
        const auto tp2 = std::chrono::steady_clock::now();
        static double acc_inf = 0, acc_print = 0, acc_bytes = 0, acc_text = 0;
        acc_inf += std::chrono::duration<double, std::milli>(tp1 - tp0).count();
        acc_print += std::chrono::duration<double, std::milli>(tp2 - tp1).count();
        acc_bytes += (double)(acc - old_end);
        acc_text += (double)text;
        fprintf(stderr, "produce_batch totals: inflate %.1f ms (%.1f MB)  chain+print %.1f ms (%.1f MB text)\n", acc_inf, acc_bytes / 1e6, acc_print, acc_text / 1e6);
    }
    if (n) b->text_per_byte = std::max(1.0, (double)text / (double)done);
    st.pos = (size_t)done;
    *n_records = n;
    *n_workers = nt;
    return true;
}

}  // namespace

extern "C" {

int xmh_bam_open(const uint8_t *data, uint64_t len, int n_threads, xmh_bam **out)
{
    if (!data || !out) return XMH_ERR_INVALID_ARG;
    *out = nullptr;
    xmh_bam *b = nullptr;
    try {
        b = new xmh_bam();
        b->data = data;
        b->len = len;
        if (n_threads <= 0) n_threads = xmh_default_threads();
        b->n_threads = std::max(1, std::min(n_threads, 64));
        b->pool.reset(new xmh::Pool(b->n_threads));
        b->workers = std::vector<BamWorker>((size_t)b->n_threads);
        if (!index_blocks(data, len, b->blocks) || !read_header(b)) { delete b; return XMH_ERR_BAD_BAM; }
        *out = b;
        return XMH_OK;
    } catch (...) {                                                   // bad_alloc, or no more threads
        delete b;
        return XMH_ERR_OOM;
    }
}

// The GPU BAM path (xm_bamdev) wants the header text, the reference names and where the records begin -- and a printer -- but
// walks the members itself, window by window: indexing every block of the file here (a page touched per 64 KB: 0.15 s of a
// 1.5 s run on 6 GB of BAM) is not needed.  Only as many blocks as the header takes are indexed; reading records from this
// handle is refused.
int xmh_bam_open_header(const uint8_t *data, uint64_t len, int n_threads, xmh_bam **out)
{
    if (!data || !out) return XMH_ERR_INVALID_ARG;
    *out = nullptr;
    xmh_bam *b = nullptr;
    try {
        for (size_t limit = 256; ; limit *= 16) {
            b = new xmh_bam();
            b->data = data;
            b->len = len;
            const int nt = n_threads <= 0 ? xmh_default_threads() : n_threads;
            b->n_threads = std::max(1, std::min(nt, 64));
            b->pool.reset(new xmh::Pool(b->n_threads));
            b->workers = std::vector<BamWorker>((size_t)b->n_threads);
            b->header_only = true;
            if (!index_blocks(data, len, b->blocks, limit)) { delete b; return XMH_ERR_BAD_BAM; }
            const bool all = b->blocks.size() < limit;
            if (read_header(b)) break;
            delete b;
            b = nullptr;
            if (all) return XMH_ERR_BAD_BAM;                          // the whole file was indexed and the header still fails
        }
        *out = b;
        return XMH_OK;
    } catch (...) {
        delete b;
        return XMH_ERR_OOM;
    }
}

int xmh_bam_close(xmh_bam *b)
{
    if (!b) return XMH_ERR_INVALID_ARG;
    delete b;
    return XMH_OK;
}

int xmh_bam_header(xmh_bam *b, const char **text, uint64_t *len)
{
    if (!b || !text || !len) return XMH_ERR_INVALID_ARG;
    *text = b->header.data();
    *len = b->header.size();
    return XMH_OK;
}

static int bam_read(xmh_bam *b, char *dst, uint64_t cap, uint64_t *written, int *eof, PreOut *po)
{
    if (!b || !dst || !written || !eof || b->header_only) return XMH_ERR_INVALID_ARG;
    if (po && !b->pending.empty() && b->pending_pre.empty()) return XMH_ERR_INVALID_ARG;   // text left over by a plain read
    try {
        static const bool profile = getenv("XMH_PROFILE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        *eof = 0;
        b->want_pre = po != nullptr;
        uint64_t w = take_pending(b, dst, cap, po);
        *written = w;
        // batches small enough for a worker's share to stay in its cache between inflating, scanning and printing
        const size_t batch_max = (size_t)b->pool->size() << 21;
        size_t budget = 0, batches = 0;
        while (b->pending.empty()) {
            if (b->next_block >= b->blocks.size() && b->avail() == 0) break;
            const uint64_t room = cap - w;
            if (w > 0 && room < (64u << 10)) break;                  // close enough to full
            if (po && w > 0 && (po->pre_cap - po->n_pre < 4096 || po->ops_cap - po->n_ops < 65536)) break;
            const size_t fit = (size_t)((double)room / (b->text_per_byte * 1.02));
            budget = std::max(budget, std::min(batch_max, std::max<size_t>((size_t)64 << 10, fit)));
            size_t n_rec = 0;
            int nt = 0;
            if (!produce_batch(b, budget, &n_rec, &nt)) return XMH_ERR_BAD_BAM;
            ++batches;
            if (n_rec == 0) {
                if (b->next_block >= b->blocks.size()) {
                    if (b->avail() == 0) break;                      // only empty blocks (the end-of-file marker) were left
                    return XMH_ERR_BAD_BAM;                          // the file ends inside a record
                }
                budget = b->avail() >= 4 ? std::max(2 * budget, (size_t)le32(b->cur()) + 8) : 2 * budget;
                continue;                                            // a record longer than the batch: take more blocks
            }
            budget = 0;
            // whole worker buffers while they fit (copied in parallel), the rest becomes pending
            std::vector<uint64_t> at((size_t)nt + 1, w), pat((size_t)nt + 1, po ? po->n_pre : 0), oat((size_t)nt + 1, po ? po->n_ops : 0);
            int whole = 0;
            for (; whole < nt; ++whole) {
                const BamWorker &wk = b->workers[(size_t)whole];
                if (at[(size_t)whole] + wk.text_len > cap) break;
                if (po && (pat[(size_t)whole] + wk.pre.size() > po->pre_cap || oat[(size_t)whole] + wk.ops.size() > po->ops_cap)) break;
                at[(size_t)whole + 1] = at[(size_t)whole] + wk.text_len;
                pat[(size_t)whole + 1] = pat[(size_t)whole] + wk.pre.size();
                oat[(size_t)whole + 1] = oat[(size_t)whole] + wk.ops.size();
            }
            b->pool->run(whole, [&](int t) {
                const BamWorker &wk = b->workers[(size_t)t];
                if (wk.text_len) memcpy(dst + at[(size_t)t], wk.text.get(), wk.text_len);
                if (po) {
                    const uint32_t shift = (uint32_t)oat[(size_t)t];
                    xmh_pre *out = po->pre + pat[(size_t)t];
                    for (size_t i = 0; i < wk.pre.size(); ++i) { out[i] = wk.pre[i]; out[i].ops_at += shift; }
                    if (!wk.ops.empty()) memcpy(po->ops + oat[(size_t)t], wk.ops.data(), wk.ops.size() * 4);
                }
            });
            w = at[(size_t)whole];
            if (po) { po->n_pre = pat[(size_t)whole]; po->n_ops = oat[(size_t)whole]; }
            for (int t = whole; t < nt; ++t) {
                const BamWorker &wk = b->workers[(size_t)t];
                if (wk.text_len) b->pending.insert(b->pending.end(), wk.text.get(), wk.text.get() + wk.text_len);
                if (po) {
                    const uint32_t shift = (uint32_t)b->pending_ops.size();
                    for (xmh_pre q : wk.pre) { q.ops_at += shift; b->pending_pre.push_back(q); }
                    b->pending_ops.insert(b->pending_ops.end(), wk.ops.begin(), wk.ops.end());
                }
            }
            w += take_pending(b, dst + w, cap - w, po);
        }
        *written = w;
        *eof = (b->pending.empty() && b->avail() == 0 && b->next_block >= b->blocks.size()) ? 1 : 0;
        if (profile)
            fprintf(stderr, "xmh_bam_read: %.1f MB text in %zu batches, %.1f ms\n", (double)w / 1e6, batches,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        return XMH_OK;
    } catch (const std::bad_alloc &) {
        return XMH_ERR_OOM;
    }
}

int xmh_bam_read(xmh_bam *b, char *dst, uint64_t cap, uint64_t *written, int *eof)
{
    return bam_read(b, dst, cap, written, eof, nullptr);
}

int xmh_bam_records_start(xmh_bam *b, uint64_t *inflated_offset)
{
    if (!b || !inflated_offset || !b->header_done) return XMH_ERR_INVALID_ARG;
    // read_header leaves stream.pos behind the reference list, in a stream that began with the file's first block; nothing
    // has been compacted yet when this is asked right after xmh_bam_open
    *inflated_offset = b->stream.pos;
    return XMH_OK;
}

int xmh_bam_walk(const uint8_t *raw, uint64_t len, uint64_t start, uint32_t *rec_off, uint64_t cap, uint64_t *n_records, uint64_t *stop)
{
    if (!n_records || !stop || start > len || len > 0xFFFFFFFFull || (len && !raw)) return XMH_ERR_INVALID_ARG;
    uint64_t p = start, n = 0;
    while (len - p >= 4) {
        const uint64_t size = le32(raw + p);
        if (size > len - p - 4) break;                               // the record continues behind the window
        if (rec_off && n < cap) rec_off[n] = (uint32_t)p;
        ++n;
        p += 4 + size;
    }
    *n_records = n;
    *stop = p;
    return XMH_OK;
}

int xmh_bam_print(xmh_bam *b, const uint8_t *raw, const uint32_t *rec_off, uint64_t n, const uint8_t *wanted, char *dst, uint64_t cap,
                  uint32_t *line_off, uint32_t *line_len, int sparse, uint64_t *written)
{
    if (!b || !written || (n && (!raw || !rec_off || !line_off || !line_len))) return XMH_ERR_INVALID_ARG;
    *written = 0;
    if (n == 0) return XMH_OK;
    try {
        size_t longest_ref = 0;
        for (auto &name : b->ref_names) longest_ref = std::max(longest_ref, name.size());
        const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)b->pool->size(), (n + 255) / 256));
        std::vector<uint64_t> first((size_t)nt + 1), room((size_t)nt + 1, 0);
        for (int t = 0; t <= nt; ++t) first[(size_t)t] = n * (uint64_t)t / (uint64_t)nt;
        // text a record can grow to: the worst ratios are a B:c array element (1 byte -> "-128,"), a CIGAR operation (4 -> 11)
        // and a packed base pair (1 -> 2), so 5x; the fixed fields add < 128 and two reference names
        b->pool->run(nt, [&](int t) {
            uint64_t bytes = 0, recs = 0;
            for (uint64_t i = first[(size_t)t]; i < first[(size_t)t + 1]; ++i)
                if (!wanted || wanted[i]) { bytes += le32(raw + rec_off[i]); ++recs; }
            room[(size_t)t + 1] = (5 * bytes + recs * (128 + 2 * longest_ref) + 63) & ~(uint64_t)63;
        });
        for (int t = 0; t < nt; ++t) { b->workers[(size_t)t].ok = true; b->workers[(size_t)t].text_len = 0; }
        if (sparse) {
            // every worker prints straight into its own stretch of dst, sized for the worst case: no second copy, the text is
            // not contiguous (the line table says where every line is); untouched pages of dst cost nothing
            for (int t = 0; t < nt; ++t) room[(size_t)t + 1] += room[(size_t)t];
            *written = room[(size_t)nt];
            if (room[(size_t)nt] > cap || room[(size_t)nt] > 0xFFFFFFFFull || !dst) return XMH_ERR_INVALID_ARG;
            b->pool->run(nt, [&](int t) {
                BamWorker &w = b->workers[(size_t)t];
                char *const base = dst + room[(size_t)t];
                char *o = base;
                for (uint64_t i = first[(size_t)t]; i < first[(size_t)t + 1]; ++i) {
                    if (wanted && !wanted[i]) { line_off[i] = 0; line_len[i] = 0; continue; }    // no sink takes this record
                    char *const line = o;
                    o = format_record(b, raw + rec_off[i] + 4, le32(raw + rec_off[i]), o);
                    if (!o) { w.ok = false; return; }
                    line_off[i] = (uint32_t)(line - dst);
                    line_len[i] = (uint32_t)(o - line - 1);                    // without the '\n'
                }
            });
            for (int t = 0; t < nt; ++t)
                if (!b->workers[(size_t)t].ok) return XMH_ERR_BAD_BAM;
            return XMH_OK;
        }
        b->pool->run(nt, [&](int t) {
            BamWorker &w = b->workers[(size_t)t];
            const uint64_t lo = first[(size_t)t], hi = first[(size_t)t + 1];
            if (lo == hi) return;
            const size_t need = (size_t)room[(size_t)t + 1];
            if (w.text_cap < need) { w.text.reset(new char[need]); w.text_cap = need; }
            char *o = w.text.get();
            for (uint64_t i = lo; i < hi; ++i) {
                char *const line = o;
                if (wanted && !wanted[i]) { line_off[i] = (uint32_t)(line - w.text.get()); line_len[i] = 0; continue; }
                o = format_record(b, raw + rec_off[i] + 4, le32(raw + rec_off[i]), o);
                if (!o) { w.ok = false; return; }
                line_off[i] = (uint32_t)(line - w.text.get());                 // local: rebased below
                line_len[i] = (uint32_t)(o - line - 1);
            }
            w.text_len = (size_t)(o - w.text.get());
        });
        std::vector<uint64_t> at((size_t)nt + 1, 0);
        for (int t = 0; t < nt; ++t) {
            if (!b->workers[(size_t)t].ok) return XMH_ERR_BAD_BAM;
            at[(size_t)t + 1] = at[(size_t)t] + b->workers[(size_t)t].text_len;
        }
        *written = at[(size_t)nt];
        if (at[(size_t)nt] > cap || at[(size_t)nt] > 0xFFFFFFFFull || !dst) return XMH_ERR_INVALID_ARG;     // *written says what it takes
        b->pool->run(nt, [&](int t) {
            const BamWorker &w = b->workers[(size_t)t];
            if (w.text_len) memcpy(dst + at[(size_t)t], w.text.get(), w.text_len);
            const uint32_t shift = (uint32_t)at[(size_t)t];
            for (uint64_t i = first[(size_t)t]; i < first[(size_t)t + 1]; ++i) line_off[i] += shift;
        });
        return XMH_OK;
    } catch (const std::bad_alloc &) {
        return XMH_ERR_OOM;
    }
}

int xmh_bam_read_pre(xmh_bam *b, char *dst, uint64_t cap, uint64_t *written, int *eof,
                     xmh_pre *pre, uint64_t pre_cap, uint64_t *n_pre, uint32_t *ops, uint64_t ops_cap, uint64_t *n_ops)
{
    if (!pre || !n_pre || !ops || !n_ops || pre_cap == 0) return XMH_ERR_INVALID_ARG;
    PreOut po = {pre, pre_cap, 0, ops, ops_cap, 0};
    const int rc = bam_read(b, dst, cap, written, eof, &po);
    *n_pre = po.n_pre;
    *n_ops = po.n_ops;
    return rc;
}

}  // extern "C"
