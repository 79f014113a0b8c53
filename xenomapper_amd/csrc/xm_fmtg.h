// printf("%g") of a binary32 value, exact -- for the device SAM printer (xm_bamdev.hip: optional fields of type f and B:f).
//
// Why it exists: the reference reads BAM as the TEXT `samtools view` prints (/root/reference/xenomapper/xenomapper.py:56-64),
// and samtools prints a float field with "%g" of the value promoted to double (htslib sam_format_aux1; its kputd is written to
// equal %g).  "%g" is a fixed-precision conversion (6 significant digits, correctly rounded from the EXACT binary value, ties to
// even as glibc does it), not a shortest-round-trip one, so it needs exact arithmetic and no tables: a binary32 value is
// m * 2^e with m < 2^24 and -149 <= e <= 104, i.e. a binary fixed-point number of at most 128 integer and 149 fraction bits.
// Held in nine 32-bit words, its decimal digits come out nine at a time -- the integer part by dividing by 10^9, the fraction by
// multiplying with 10^9 -- until seven significant digits are known; everything behind them only says "not zero" (sticky).
// All word indices are compile-time constants (the loops are unrolled), so the number lives in registers on the device.
//
// Compiled for the device (hipcc) and for the host (tests/fmtg_host.cpp runs it against snprintf over millions of bit patterns,
// tests/test_fmtg_host.py); nothing in the product calls the host build.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define XM_FMTG_HD __host__ __device__ __forceinline__
#else
#define XM_FMTG_HD inline
#endif

namespace xmfmt {

struct Text16 {                      // up to 16 characters, byte k of (lo, hi) = character k
    uint64_t lo = 0, hi = 0;
    uint32_t n = 0;
    XM_FMTG_HD void put(uint32_t c)
    {
        if (n < 8u) lo |= (uint64_t)(c & 0xFFu) << (8u * n);
        else hi |= (uint64_t)(c & 0xFFu) << (8u * (n - 8u));
        ++n;
    }
    XM_FMTG_HD uint32_t at(uint32_t k) const { return (uint32_t)((k < 8u ? lo >> (8u * k) : hi >> (8u * (k - 8u))) & 0xFFu); }
};

XM_FMTG_HD uint32_t decimal_digits(uint32_t c)          // of 1 <= c < 10^9
{
    uint32_t d = 1;
    if (c >= 10u) d = 2;
    if (c >= 100u) d = 3;
    if (c >= 1000u) d = 4;
    if (c >= 10000u) d = 5;
    if (c >= 100000u) d = 6;
    if (c >= 1000000u) d = 7;
    if (c >= 10000000u) d = 8;
    if (c >= 100000000u) d = 9;
    return d;
}

// "%g" of the binary32 value with these bits (as glibc prints the value promoted to double: inf, -inf, nan, -nan, -0)
XM_FMTG_HD Text16 fmt_g_f32(uint32_t bits)
{
    Text16 t;
    const uint32_t ex = (bits >> 23) & 0xFFu, fr = bits & 0x7FFFFFu;
    if (bits >> 31) t.put('-');
    if (ex == 255u) {
        if (fr) { t.put('n'); t.put('a'); t.put('n'); } else { t.put('i'); t.put('n'); t.put('f'); }
        return t;
    }
    if (ex == 0u && fr == 0u) { t.put('0'); return t; }
    const uint32_t m = ex ? (fr | 0x800000u) : fr;
    const int32_t e = (int32_t)(ex ? ex : 1u) - 150;                    // value = m * 2^e
    // W[0..4]: fraction (bit 31 of W[4] = 1/2), W[5..8]: integer part (W[5] lowest); bit 160 is 2^0
    uint32_t W[9];
    {
        const uint32_t sh = (uint32_t)(160 + e), w = sh >> 5, b = sh & 31u;
        const uint32_t lo = m << b, hi = b ? m >> (32u - b) : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 9u; ++k) W[k] = (k == w) ? lo : (k == w + 1u) ? hi : 0u;
    }
    uint64_t acc = 0;             // the significant digits met so far
    uint32_t nd = 0;              // how many (0: none yet)
    int32_t X = 0;                // decimal exponent of the first one
    bool sticky = false;          // a non-zero digit behind those in acc
    // integer part: five chunks of nine digits, the lowest first out of the division -- so they are kept, then read from the top
    uint32_t ic[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        uint64_t rem = 0;
#pragma unroll
        for (int k = 8; k >= 5; --k) {
            const uint64_t cur = (rem << 32) | W[k];
            W[k] = (uint32_t)(cur / 1000000000ull);
            rem = cur % 1000000000ull;
        }
        ic[j] = (uint32_t)rem;
    }
#pragma unroll
    for (int j = 4; j >= 0; --j) {
        const uint32_t c = ic[j];
        if (nd >= 7u) sticky = sticky || c != 0u;
        else if (nd == 0u) {
            if (c != 0u) { nd = decimal_digits(c); X = 9 * j + (int32_t)nd - 1; acc = c; }
        } else { acc = acc * 1000000000ull + c; nd += 9u; }
    }
    // fraction: chunks of nine digits while fewer than seven significant digits are known (the first one of a non-zero binary32
    // value stands at 10^-45 or above: five chunks of zeros at most, then two more)
#pragma unroll 1
    for (int f = 0; f < 8 && nd < 7u && (W[0] | W[1] | W[2] | W[3] | W[4]) != 0u; ++f) {
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint64_t cur = (uint64_t)W[k] * 1000000000ull + carry;
            W[k] = (uint32_t)cur;
            carry = cur >> 32;
        }
        const uint32_t c = (uint32_t)carry;
        if (nd == 0u) {
            if (c != 0u) { nd = decimal_digits(c); X = -9 * f - 10 + (int32_t)nd; acc = c; }
        } else { acc = acc * 1000000000ull + c; nd += 9u; }
    }
    sticky = sticky || (W[0] | W[1] | W[2] | W[3] | W[4]) != 0u;
    while (nd < 7u) { acc *= 10ull; ++nd; }                             // an exact value with fewer digits: zeros follow
    uint64_t unit = 1;
    for (uint32_t k = 6u; k < nd; ++k) unit *= 10ull;                   // 10^(nd - 6), nd <= 15
    uint32_t D = (uint32_t)(acc / unit);
    const uint64_t rem = acc % unit, half = unit / 2ull;
    if (rem > half || (rem == half && (sticky || (D & 1u)))) ++D;      // to nearest, ties to even (the exact value decides)
    if (D == 1000000u) { D = 100000u; ++X; }
    uint32_t d[6];
#pragma unroll
    for (int k = 5; k >= 0; --k) { d[k] = D % 10u; D /= 10u; }
    uint32_t last = 5;                                                  // %g drops trailing zeros (d[0] is never zero)
#pragma unroll
    for (int k = 5; k >= 1; --k)
        if (last == (uint32_t)k && d[k] == 0u) last = (uint32_t)k - 1u;
    if (X < -4 || X >= 6) {
        t.put('0' + d[0]);
        if (last >= 1u) {
            t.put('.');
#pragma unroll
            for (uint32_t k = 1; k < 6u; ++k) if (k <= last) t.put('0' + d[k]);
        }
        t.put('e');
        const uint32_t ax = (uint32_t)(X < 0 ? -X : X);
        t.put(X < 0 ? '-' : '+');
        t.put('0' + ax / 10u);
        t.put('0' + ax % 10u);
    } else if (X >= 0) {
#pragma unroll
        for (uint32_t k = 0; k < 6u; ++k) {
            if (k == (uint32_t)X + 1u && k <= last) t.put('.');
            if (k <= last || k <= (uint32_t)X) t.put('0' + d[k]);
        }
    } else {
        t.put('0'); t.put('.');
        for (int32_t z = -1; z > X; --z) t.put('0');
#pragma unroll
        for (uint32_t k = 0; k < 6u; ++k) if (k <= last) t.put('0' + d[k]);
    }
    return t;
}

}  // namespace xmfmt
