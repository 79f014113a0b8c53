// xm_bam.cpp -- native BGZF/BAM -> SAM text decoder (C ABI in include/xenomapper_host.h, section "BAM input").
//
// The reference reads BAM by piping it through `samtools view` (xenomapper.py:48-93) and then treats the text
// exactly like SAM input.  This decoder produces that text -- the header block and one line per alignment in
// the layout `samtools view` prints -- so that everything downstream (column stripper, kernels, writer) is the
// SAM path unchanged and inherits its parity pins.  BGZF blocks are inflated in parallel (zlib, raw deflate).
#include "../../include/xenomapper_host.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

struct BgzfBlock {
    uint64_t cdata_off;      // offset of the deflate stream in the file image
    uint32_t cdata_len;
    uint32_t isize;          // uncompressed size
};

inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// walk the gzip member headers: every BGZF block carries its own size in the 'BC' extra subfield
bool index_blocks(const uint8_t *d, uint64_t len, std::vector<BgzfBlock> &out)
{
    uint64_t p = 0;
    while (p < len) {
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
        out.push_back(b);
        p += total;
    }
    return true;
}

bool inflate_block(const uint8_t *src, uint32_t slen, uint8_t *dst, uint32_t dlen)
{
    if (dlen == 0) return true;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = slen;
    zs.next_out = dst;
    zs.avail_out = dlen;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = (rc == Z_STREAM_END) && zs.total_out == dlen;
    inflateEnd(&zs);
    return ok;
}

inline void put_int(std::string &s, long long v)
{
    char buf[24];
    const int n = snprintf(buf, sizeof buf, "%lld", v);
    s.append(buf, (size_t)n);
}

inline void put_real(std::string &s, double v)
{
    char buf[40];
    const int n = snprintf(buf, sizeof buf, "%g", v);
    s.append(buf, (size_t)n);
}

}  // namespace

struct xmh_bam {
    const uint8_t *data;
    uint64_t len;
    int n_threads;
    std::vector<BgzfBlock> blocks;
    size_t next_block = 0;              // first block not yet inflated
    std::vector<uint8_t> stream;        // inflated bytes not yet consumed, starting at stream_pos
    size_t stream_pos = 0;
    std::string header;                 // SAM header text
    std::vector<std::string> ref_names;
    bool header_done = false;
    std::string pending;                // formatted text that did not fit the caller's buffer

    bool fill(size_t want)              // make at least `want` bytes available after stream_pos (or reach the end)
    {
        while (stream.size() - stream_pos < want && next_block < blocks.size()) {
            if (stream_pos > (64u << 20)) {                      // drop what has been consumed
                stream.erase(stream.begin(), stream.begin() + (ptrdiff_t)stream_pos);
                stream_pos = 0;
            }
            const size_t batch = std::min<size_t>(blocks.size() - next_block, (size_t)std::max(64, n_threads * 16));
            std::vector<uint64_t> off(batch + 1);
            uint64_t acc = stream.size();
            for (size_t i = 0; i < batch; ++i) { off[i] = acc; acc += blocks[next_block + i].isize; }
            off[batch] = acc;
            stream.resize((size_t)acc);
            std::vector<char> ok((size_t)std::max(1, n_threads), 1);
            const int nt = (int)std::min<size_t>((size_t)n_threads, batch);
            std::vector<std::thread> pool;
            for (int t = 0; t < nt; ++t)
                pool.emplace_back([&, t]() {
                    for (size_t i = (size_t)t; i < batch; i += (size_t)nt) {
                        const BgzfBlock &b = blocks[next_block + i];
                        if (!inflate_block(data + b.cdata_off, b.cdata_len, stream.data() + off[i], b.isize)) ok[(size_t)t] = 0;
                    }
                });
            for (auto &th : pool) th.join();
            for (char o : ok)
                if (!o) return false;
            next_block += batch;
        }
        return true;
    }
    size_t avail() const { return stream.size() - stream_pos; }
    const uint8_t *cur() const { return stream.data() + stream_pos; }
};

namespace {

bool read_header(xmh_bam *b)
{
    if (!b->fill(12) || b->avail() < 12 || memcmp(b->cur(), "BAM\1", 4) != 0) return false;
    const uint32_t l_text = le32(b->cur() + 4);
    if (!b->fill(12 + (size_t)l_text) || b->avail() < 12 + (size_t)l_text) return false;
    b->header.assign((const char *)b->cur() + 8, l_text);
    while (!b->header.empty() && b->header.back() == '\0') b->header.pop_back();
    const uint32_t n_ref = le32(b->cur() + 8 + l_text);
    b->stream_pos += 12 + l_text;
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (!b->fill(4) || b->avail() < 4) return false;
        const uint32_t l_name = le32(b->cur());
        if (!b->fill(8 + (size_t)l_name) || b->avail() < 8 + (size_t)l_name) return false;
        std::string name((const char *)b->cur() + 4, l_name);
        while (!name.empty() && name.back() == '\0') name.pop_back();
        b->ref_names.push_back(name);
        b->stream_pos += 8 + l_name;
    }
    b->header_done = true;
    return true;
}

// one alignment record (without its block_size word) -> one SAM line, the way `samtools view` prints it
bool format_record(const xmh_bam *b, const uint8_t *r, uint32_t size, std::string &s)
{
    if (size < 32) return false;
    const int32_t ref_id = (int32_t)le32(r), pos = (int32_t)le32(r + 4);
    const uint32_t l_read_name = r[8], mapq = r[9], n_cigar = le16(r + 12), flag = le16(r + 14);
    const uint32_t l_seq = le32(r + 16);
    const int32_t next_ref = (int32_t)le32(r + 20), next_pos = (int32_t)le32(r + 24), tlen = (int32_t)le32(r + 28);
    uint64_t p = 32;
    const uint64_t need = p + l_read_name + 4ull * n_cigar + (l_seq + 1) / 2 + l_seq;
    if (need > size || l_read_name == 0) return false;
    s.append((const char *)r + p, strnlen((const char *)r + p, l_read_name));
    p += l_read_name;
    s.push_back('\t'); put_int(s, flag);
    s.push_back('\t');
    if (ref_id < 0 || (size_t)ref_id >= b->ref_names.size()) s.push_back('*'); else s += b->ref_names[(size_t)ref_id];
    s.push_back('\t'); put_int(s, (long long)pos + 1);
    s.push_back('\t'); put_int(s, mapq);
    s.push_back('\t');
    if (n_cigar == 0) s.push_back('*');
    for (uint32_t k = 0; k < n_cigar; ++k) {
        const uint32_t v = le32(r + p + 4ull * k);
        put_int(s, v >> 4);
        s.push_back("MIDNSHP=XB??????"[v & 15]);
    }
    p += 4ull * n_cigar;
    s.push_back('\t');
    if (next_ref < 0 || (size_t)next_ref >= b->ref_names.size()) s.push_back('*');
    else if (next_ref == ref_id) s.push_back('=');
    else s += b->ref_names[(size_t)next_ref];
    s.push_back('\t'); put_int(s, (long long)next_pos + 1);
    s.push_back('\t'); put_int(s, tlen);
    s.push_back('\t');
    if (l_seq == 0) s.push_back('*');
    for (uint32_t k = 0; k < l_seq; ++k) {
        const uint8_t byte = r[p + k / 2];
        s.push_back("=ACMGRSVTWYHKDBN"[(k & 1) ? (byte & 15) : (byte >> 4)]);
    }
    p += (l_seq + 1) / 2;
    s.push_back('\t');
    if (l_seq == 0 || r[p] == 0xFF) s.push_back('*');
    else for (uint32_t k = 0; k < l_seq; ++k) s.push_back((char)(r[p + k] + 33));
    p += l_seq;
    // optional fields
    while (p + 3 <= size) {
        s.push_back('\t');
        s.push_back((char)r[p]); s.push_back((char)r[p + 1]); s.push_back(':');
        const char type = (char)r[p + 2];
        p += 3;
        auto scalar = [&](char t, bool emit) -> bool {              // advances p; appends the value when emit
            switch (t) {
            case 'A': if (p + 1 > size) return false; if (emit) s.push_back((char)r[p]); p += 1; return true;
            case 'c': if (p + 1 > size) return false; if (emit) put_int(s, (int8_t)r[p]); p += 1; return true;
            case 'C': if (p + 1 > size) return false; if (emit) put_int(s, r[p]); p += 1; return true;
            case 's': if (p + 2 > size) return false; if (emit) put_int(s, (int16_t)le16(r + p)); p += 2; return true;
            case 'S': if (p + 2 > size) return false; if (emit) put_int(s, le16(r + p)); p += 2; return true;
            case 'i': if (p + 4 > size) return false; if (emit) put_int(s, (int32_t)le32(r + p)); p += 4; return true;
            case 'I': if (p + 4 > size) return false; if (emit) put_int(s, le32(r + p)); p += 4; return true;
            case 'f': {
                if (p + 4 > size) return false;
                float f;
                const uint32_t u = le32(r + p);
                memcpy(&f, &u, 4);
                if (emit) put_real(s, f);
                p += 4;
                return true;
            }
            case 'd': {
                if (p + 8 > size) return false;
                double dv;
                const uint64_t u = (uint64_t)le32(r + p) | ((uint64_t)le32(r + p + 4) << 32);
                memcpy(&dv, &u, 8);
                if (emit) put_real(s, dv);
                p += 8;
                return true;
            }
            default: return false;
            }
        };
        if (type == 'A') { s += "A:"; if (!scalar('A', true)) return false; }
        else if (type == 'c' || type == 'C' || type == 's' || type == 'S' || type == 'i' || type == 'I') { s += "i:"; if (!scalar(type, true)) return false; }
        else if (type == 'f') { s += "f:"; if (!scalar('f', true)) return false; }
        else if (type == 'd') { s += "d:"; if (!scalar('d', true)) return false; }
        else if (type == 'Z' || type == 'H') {
            s.push_back(type); s.push_back(':');
            const size_t l = strnlen((const char *)r + p, (size_t)(size - p));
            if (p + l >= size) return false;
            s.append((const char *)r + p, l);
            p += l + 1;
        } else if (type == 'B') {
            if (p + 5 > size) return false;
            const char sub = (char)r[p];
            const uint32_t cnt = le32(r + p + 1);
            p += 5;
            s += "B:"; s.push_back(sub);
            for (uint32_t k = 0; k < cnt; ++k) { s.push_back(','); if (!scalar(sub, true)) return false; }
        } else return false;
    }
    s.push_back('\n');
    return p == size;
}

}  // namespace

extern "C" {

int xmh_bam_open(const uint8_t *data, uint64_t len, int n_threads, xmh_bam **out)
{
    if (!data || !out) return XMH_ERR_INVALID_ARG;
    *out = nullptr;
    try {
        xmh_bam *b = new xmh_bam();
        b->data = data;
        b->len = len;
        if (n_threads <= 0) n_threads = xmh_default_threads();
        b->n_threads = std::max(1, std::min(n_threads, 64));
        if (!index_blocks(data, len, b->blocks) || !read_header(b)) { delete b; return XMH_ERR_BAD_BAM; }
        *out = b;
        return XMH_OK;
    } catch (const std::bad_alloc &) {
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

int xmh_bam_read(xmh_bam *b, char *dst, uint64_t cap, uint64_t *written, int *eof)
{
    if (!b || !dst || !written || !eof) return XMH_ERR_INVALID_ARG;
    try {
        uint64_t w = 0;
        *eof = 0;
        if (!b->pending.empty()) {
            if (b->pending.size() > cap) return XMH_ERR_INVALID_ARG;      // buffer smaller than one line
            memcpy(dst, b->pending.data(), b->pending.size());
            w = b->pending.size();
            b->pending.clear();
        }
        std::string line;
        while (true) {
            if (!b->fill(4)) return XMH_ERR_BAD_BAM;
            if (b->avail() < 4) { *eof = (b->avail() == 0); if (!*eof) return XMH_ERR_BAD_BAM; break; }
            const uint32_t size = le32(b->cur());
            if (!b->fill(4 + (size_t)size)) return XMH_ERR_BAD_BAM;
            if (b->avail() < 4 + (size_t)size) return XMH_ERR_BAD_BAM;
            line.clear();
            if (!format_record(b, b->cur() + 4, size, line)) return XMH_ERR_BAD_BAM;
            b->stream_pos += 4 + (size_t)size;
            if (w + line.size() > cap) { b->pending.swap(line); break; }
            memcpy(dst + w, line.data(), line.size());
            w += line.size();
        }
        *written = w;
        return XMH_OK;
    } catch (const std::bad_alloc &) {
        return XMH_ERR_OOM;
    }
}

}  // extern "C"
