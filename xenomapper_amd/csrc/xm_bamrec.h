// xm_bamrec.h -- what the GPU BAM front end reads out of ONE alignment record (device code shared by xm_bamdev.hip's parse_kernel
// and the epilogue of the inflate launch in xm_inflate.hip, which parses the records of a block while the block is still near the
// CU that wrote it): the plugins' rules on the typed optional fields, restating get_tag / get_tag_with_ZS_as_XS
// (/root/reference/xenomapper/xenomapper.py:176-206) as csrc/xm_bam.cpp's TagScan does on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xmrec {

constexpr int32_t ABSENT = INT32_MIN;
constexpr uint32_t R_EX_A_SHIFT = 0, R_EX_X_SHIFT = 2;     // per-record flag byte: ex_a (2 bits), ex_x (2 bits),
constexpr uint32_t R_WEIRD = 0x10u, R_BAD = 0x20u;         // weird, malformed

// little-endian fields at any byte offset: ONE access each (gfx950 reads unaligned words and dwords)
struct __attribute__((packed, aligned(1))) U32Unaligned { uint32_t v; };
struct __attribute__((packed, aligned(1))) U16Unaligned { uint16_t v; };
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return reinterpret_cast<const U16Unaligned *>(p)->v; }
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return reinterpret_cast<const U32Unaligned *>(p)->v; }

__device__ __forceinline__ bool odd_byte(uint32_t c) { return c <= 0x20u || c >= 0x7Fu; }

__device__ __forceinline__ uint32_t elem_bytes(uint32_t t)
{
    switch (t) {
    case 'A': case 'c': case 'C': return 1u;
    case 's': case 'S': return 2u;
    case 'i': case 'I': case 'f': return 4u;
    case 'd': return 8u;
    default: return 0u;
    }
}

struct RecFields {
    uint32_t name_off;        // first byte of QNAME in raw
    uint32_t name_len;        // without the NUL
    int32_t a, x;             // AS (or NM in --cigar_scores mode), and XS or ZS by mode; ABSENT = no match
    uint32_t flag;            // ex_a | ex_x << 2 | R_WEIRD | R_BAD
    uint32_t n_cigar, cig_at; // the CIGAR operations (BAM words, len << 4 | op) of the record: how many, where in raw
};

// Which tags are read: x0 = 'X' or 'Z' (which tag plays XS: get_tag / get_tag_with_ZS_as_XS, :176-206); tags >> 8 = the two letters
// of the first value ('A' | 'S' << 8: AS, with the duplicate rule; 'N' | 'M' << 8 with bit 16 set: NM as get_cigarbased_AS_tag
// reads it, :247-250 -- the FIRST field that holds the letters decides, later ones do not matter).
constexpr uint32_t TAGS_AS_XS = 'X' | ('A' << 8) | ('S' << 16);
constexpr uint32_t TAGS_AS_ZS = 'Z' | ('A' << 8) | ('S' << 16);
constexpr uint32_t TAGS_NM_XS = 'X' | ('N' << 8) | ('M' << 16) | (1u << 24);

// the record whose block_size word is at raw + off
__device__ __forceinline__ RecFields parse_record(const uint8_t *__restrict__ raw, uint32_t off, uint32_t tags)
{
    const uint32_t x0 = tags & 0xFFu, a0 = (tags >> 8) & 0xFFu, a1 = (tags >> 16) & 0xFFu;
    const bool a_first_only = (tags >> 24) != 0u;
    uint32_t n_cig = 0, cig_at = 0;
    const uint32_t size = ld32(raw + off);
    const uint8_t *r = raw + off + 4u;
    uint32_t flag = 0, nl = 0;
    uint32_t cnt_a = 0, cnt_x = 0, ex_a = 0, ex_x = 0;
    int32_t va = ABSENT, vx = ABSENT;
    bool ok = size >= 32u;
    if (ok) {
        const int32_t ref_id = (int32_t)ld32(r), pos = (int32_t)ld32(r + 4);
        const uint32_t l_read_name = r[8], n_cigar = ld16(r + 12), l_seq = ld32(r + 16);
        const uint64_t need = 32ull + l_read_name + 4ull * n_cigar + ((uint64_t)l_seq + 1u) / 2u + l_seq;
        ok = need <= size && l_read_name != 0u;
        if (ok) {
            bool weird = false;
            n_cig = n_cigar;
            cig_at = off + 36u + l_read_name;
            // QNAME up to its NUL
            while (nl < l_read_name && r[32u + nl] != 0u) { weird |= odd_byte(r[32u + nl]); ++nl; }
            weird |= nl == 0u;
            uint32_t p = 32u + l_read_name;
            // a CIGAR that lives in a CG:B:I field (htslib moves it back when it prints): possible only with this placeholder
            const bool cg_possible = n_cigar != 0u && ref_id >= 0 && pos >= 0 && (ld32(r + p) & 15u) == 4u && (ld32(r + p) >> 4) == l_seq;
            p += 4u * n_cigar + (l_seq + 1u) / 2u;
            // qualities: printed as byte + 33; above 93 the text is not ASCII any more
            if (l_seq != 0u && r[p] != 0xFFu) {
                uint32_t hi = 0;
                for (uint32_t k = 0; k < l_seq; ++k) hi |= (r[p + k] + 33u) & 0xFFu;
                weird |= (hi & 0x80u) != 0u;
            }
            p += l_seq;
            // optional fields
            while (ok && p + 3u <= size) {
                const uint32_t t0 = r[p], t1 = r[p + 1], type = r[p + 2];
                p += 3u;
                weird |= odd_byte(t0) || odd_byte(t1);
                weird |= cg_possible && t0 == 'C' && t1 == 'G';
                const bool is_a = t0 == a0 && t1 == a1, is_x = t0 == x0 && t1 == 'S';
                if (type == 'Z' || type == 'H') {
                    // the printed field "TG:Z:value": matched by its name or by the two letters anywhere in the value
                    bool in_a = is_a, in_x = is_x, end = false;
                    uint32_t prev = 0;
                    while (p < size) {
                        const uint32_t c = r[p++];
                        if (c == 0u) { end = true; break; }
                        weird |= odd_byte(c);
                        in_a |= prev == a0 && c == a1;
                        in_x |= prev == x0 && c == 'S';
                        prev = c;
                    }
                    ok = end;
                    // a string value is never vouched for here: the text rules decide (the count still matters: duplicates)
                    if (in_a) { if (cnt_a++ == 0u) ex_a = 1u; }
                    if (in_x) { if (cnt_x++ == 0u) ex_x = 1u; }
                    continue;
                }
                uint32_t bytes;
                if (type == 'B') {
                    if (p + 5u > size) { ok = false; break; }
                    const uint32_t e = elem_bytes(r[p]);
                    const uint64_t len = 5ull + (uint64_t)e * ld32(r + p + 1);
                    if (e == 0u || p + len > size) { ok = false; break; }
                    bytes = (uint32_t)len;
                } else {
                    bytes = elem_bytes(type);
                    if (bytes == 0u || p + bytes > size) { ok = false; break; }
                    if (type == 'A') weird |= odd_byte(r[p]);
                }
                if (is_a || is_x) {                                        // only the tag itself can hold the two letters
                    long long v = 0;
                    bool is_int = true;
                    switch (type) {
                    case 'c': v = (int8_t)r[p]; break;
                    case 'C': v = r[p]; break;
                    case 's': v = (int16_t)ld16(r + p); break;
                    case 'S': v = ld16(r + p); break;
                    case 'i': v = (int32_t)ld32(r + p); break;
                    case 'I': v = ld32(r + p); break;
                    default: is_int = false;                               // A, f, d, B by name: what its printed text says
                    }
                    const bool fits = is_int && v >= -2147483647ll && v <= 2147483647ll;
                    if (is_a) { if (cnt_a++ == 0u) { if (fits) va = (int32_t)v; else ex_a = 1u; } }
                    if (is_x) { if (cnt_x++ == 0u) { if (fits) vx = (int32_t)v; else ex_x = 1u; } }
                }
                p += bytes;
            }
            ok = ok && p == size;
            if (weird) flag |= R_WEIRD;
        }
    }
    if (!ok) flag |= R_BAD;
    if (cnt_a > 1u && !a_first_only) ex_a = 2u;
    if (cnt_x > 1u) ex_x = 2u;
    if (ex_a) va = ABSENT;
    if (ex_x) vx = ABSENT;
    RecFields f;
    f.name_off = off + 36u;
    f.name_len = nl;
    f.a = va;
    f.x = vx;
    f.flag = flag | (ex_a << R_EX_A_SHIFT) | (ex_x << R_EX_X_SHIFT);
    f.n_cigar = n_cig;
    f.cig_at = cig_at;
    return f;
}

}  // namespace xmrec
