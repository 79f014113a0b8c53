// xm_inflate_core.h -- raw-DEFLATE decoder of one BGZF block, written for a GROUP of GS lanes of one wavefront ("a chain").
//
// What it replaces: the reference reads BAM by piping it through `samtools view` (getBamReadPairs / bam_lines /
// get_bam_header, /root/reference/xenomapper/xenomapper.py:48-93), i.e. htslib's BGZF layer + zlib's inflate.  The format is
// RFC 1951 (DEFLATE) inside RFC 1952 members with the BGZF extra field (SAM specification section 4.1); nothing here is
// taken from zlib or htslib -- the decoder is restated from the RFC for the execution model below.
//
// Execution model.  DEFLATE is a serial chain per block (every symbol's position depends on the one before), so the
// parallelism is ACROSS blocks: a chain = GS lanes of one wave decodes one BGZF block at a time; a wave runs 64 / GS chains
// side by side (SIMT: every lane of a chain computes the chain's scalar state redundantly -- bit buffer, positions -- so no
// cross-lane traffic is needed for it), and the lanes of a chain split what is data-parallel: filling decode tables, copying
// matches (GS bytes per step), moving input and output between LDS and global memory in 16-byte pieces.  A chain never spans
// waves and a wave executes in lockstep, so lanes of a chain communicate through LDS in program order: no barrier anywhere.
// With GS = 64 (what ships: a chain is a whole wave) the token loop of a Huffman block is WIDE instead: lane i decodes the
// token that would begin at bit P + i, the scalar unit walks from token to token, and a batch of up to 64 output bytes is
// produced a lane per byte (huffman_block() below; the serial loop remains for the host build, GS = 1, and -DXMI_SERIAL_TOKENS).
// GS is 64 on the device and 1 on the host, nothing else (static_assert in Chain): widths in between shared a wave between chains
// and hung in round 5 (xm_inflate.hip has the record).
//
// Per chain, in LDS (ChainMem, ~5 KB): the two Huffman decoders (a root table indexed by the next ROOT bits -- entries
// placed at bit-reversed codes, as the bits arrive LSB first -- plus the canonical (first code, limit, base) triples and the
// sorted symbol list for the rare longer codes), an input ring refilled 128 bytes at a time with the next piece already in
// flight, and an OUTPUT RING of 1 KiB: every produced byte goes there first; complete 16-byte lines are flushed to global
// memory 128 bytes at a time (aligned 16-byte stores), and a match reads its source from the ring when it is recent and
// from global memory when it is older than the flush before last (whose stores have been waited for since) -- so no store
// is ever waited for right after it was issued.
//
// The same source compiles for the host with GS = 1 (tests/inflate_core_host.cpp: the decoder's LOGIC against zlib on this
// CPU-only build container); that build is test infrastructure and is not part of any library.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define XMI_HD __device__ __forceinline__
#define XMI_DEVICE 1
#else
#define XMI_HD inline
#define XMI_DEVICE 0
#endif

namespace xmi {

enum : int {
    OK = 0,
    ERR_BTYPE = 1,          // reserved block type
    ERR_STORED = 2,         // LEN / NLEN mismatch
    ERR_LENGTHS = 3,        // bad code-length sequence (repeat with no previous length, too many lengths)
    ERR_OVERSUB = 4,        // over-subscribed Huffman code
    ERR_NO_EOB = 5,         // no end-of-block code
    ERR_CODE = 6,           // bits that match no code
    ERR_DIST = 7,           // distance beyond the start of the block, or distance code 30 / 31
    ERR_OUT = 8,            // more output than the block's ISIZE
    ERR_IN = 9,             // ran past the end of the compressed data
    ERR_SHORT = 10,         // stream ended with fewer bytes than ISIZE
    ERR_LENSYM = 11,        // length symbol 286 / 287
    ERR_GUARD = 12          // more decoding steps than any valid block of this size can take (the loops' own exit conditions
                            // make this unreachable; it is the belt to their braces: a wave must always drain)
};

constexpr int LIT_ROOT = 10;            // bits of the literal/length root table
constexpr int DIST_ROOT = 8;
constexpr int CLC_ROOT = 7;             // the code-length code: lengths <= 7, the root table is complete
constexpr uint32_t ORING = 1024;        // output ring bytes (power of two)
constexpr uint32_t FLUSH = 128;         // bytes after which complete output lines are stored
constexpr uint32_t IRING = 256;         // input ring bytes
constexpr uint32_t ICHUNK = 128;        // input refill granule

struct alignas(16) ChainMem {
    uint8_t oring[ORING];
    uint8_t iring[IRING];
    uint16_t lit_root[1 << LIT_ROOT];   // (symbol << 4) | code length; 0: a longer code (or no code) starts with these bits
    uint16_t dist_root[1 << DIST_ROOT]; // also holds the code-length code's root table while the lengths are read
    uint16_t lit_sym[288];              // symbols in canonical order (by length, then value)
    uint16_t dist_sym[32];
    uint16_t lit_meta[3][16];           // per code length: first code, limit (first + count), index of its first symbol
    uint16_t dist_meta[3][16];
    uint32_t cnt[16];                   // symbols per length while a table is built, then the running index per length
    uint32_t bcast;                     // one word handed from lane 0 to the chain
    uint8_t cl[352];                    // code lengths: [0, 316) literal/length + distance, [320, 339) the code-length code
};

struct u128 { uint32_t w[4]; };

XMI_HD uint32_t bitrev32(uint32_t v)
{
#if XMI_DEVICE
    return __brev(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
#endif
}

XMI_HD void drain_stores()
{
#if XMI_DEVICE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// the lanes of a chain are about to read LDS another lane of the chain wrote (or the reverse): nothing to wait for -- a
// wave's LDS operations execute in issue order -- but the compiler must not move accesses across this point
XMI_HD void chain_sync()
{
#if XMI_DEVICE
#ifdef XMI_SYNC_ASM
    asm volatile("" ::: "memory");
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
}

XMI_HD u128 load16(const uint8_t *p)
{
    u128 v;
#if XMI_DEVICE
    const uint4 q = *reinterpret_cast<const uint4 *>(p);
    v.w[0] = q.x; v.w[1] = q.y; v.w[2] = q.z; v.w[3] = q.w;
#else
    memcpy(&v, p, 16);
#endif
    return v;
}

XMI_HD void store16(uint8_t *p, const u128 &v)
{
#if XMI_DEVICE
    *reinterpret_cast<uint4 *>(p) = make_uint4(v.w[0], v.w[1], v.w[2], v.w[3]);
#else
    memcpy(p, &v, 16);
#endif
}

XMI_HD void count_inc(uint32_t *p)
{
#if XMI_DEVICE
    atomicAdd(p, 1u);
#else
    ++*p;
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// One chain's decoder state.  Everything here is identical in all lanes of the chain except `gl` and `pend`.
// ---------------------------------------------------------------------------------------------------------------------
// XMI_TRACE builds (tools/inflate_gpu_check.hip --trace): every chain leaves the number of the stage it has reached in a
// host-visible word, so that a launch that does not come back can be read from outside
#ifdef XMI_TRACE
#define XMI_STAGE(code) do { if (trace && gl == 0u) __hip_atomic_store(trace, (uint32_t)(code), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
#elif defined(XMI_STAGE_BARRIER)
#define XMI_STAGE(code) asm volatile("" ::: "memory")
#else
#define XMI_STAGE(code) do { } while (0)
#endif

// XMI_STAT_*: token statistics of the host build (tools: how many matches reach behind the output ring); nothing on the device
#ifndef XMI_STAT_LITERAL
#define XMI_STAT_LITERAL() do { } while (0)
#define XMI_STAT_MATCH(len, dist, behind_ring) do { } while (0)
#endif
#ifndef XMI_STAT_BLOCK
#define XMI_STAT_BLOCK(type) do { } while (0)               // a DEFLATE block of this type begins
#endif
#ifndef XMI_STAT_LONG_CODE
#define XMI_STAT_LONG_CODE(root_bits) do { } while (0)      // a code beyond a root table (the wide loop hands such a token to the serial reader)
#endif

template <int GS>
struct Chain {
#if XMI_DEVICE
    static_assert(GS == 64, "device build: a chain is a whole wave (narrower chains are not supported: xm_inflate.hip)");
#else
    static_assert(GS == 1, "host build (tests): a chain of one lane");
#endif
    uint32_t *trace = nullptr;
    static constexpr int NP = (8 + GS - 1) / GS;          // 16-byte pieces of an input granule per lane
    ChainMem *m;
    uint32_t gl;                 // lane within the chain
    // input: positions relative to cbase (128-byte aligned, at or below the block's first byte)
    const uint8_t *cbase;
    uint32_t in_pos;             // next 4-byte word to take from the ring
    uint32_t loaded;             // bytes [.., loaded) are in the ring; [loaded, loaded + 128) is in flight in `pend`
    uint32_t cend;               // end of the block's compressed bytes
    u128 pend[NP];
    uint64_t bits;
    uint32_t nbits;
    // output: positions in "v-space" = offset from vbase, the 16-byte aligned address at or below the block's first byte
    uint8_t *vbase;
    uint32_t op, oend;           // next byte to produce, end of the block (start + ISIZE)
    uint32_t ostart;             // first byte of the block
    uint32_t flushed;            // bytes below this have been stored to global memory
    uint32_t gsafe;              // bytes below this were stored AND waited for: readable from global memory
    uint32_t budget;             // decoding steps left (every symbol, every block header costs one)
    int err;

    // ---- input ----
    XMI_HD void issue_load()
    {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const uint32_t piece = gl + (uint32_t)q * GS;
            if (piece < 8u) pend[q] = load16(cbase + loaded + piece * 16u);
        }
    }
    XMI_HD void land_load()
    {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const uint32_t piece = gl + (uint32_t)q * GS;
            if (piece < 8u) store16(m->iring + ((loaded + piece * 16u) & (IRING - 1u)), pend[q]);
        }
        loaded += ICHUNK;
        chain_sync();
    }
    // start reading at byte `byte_pos` (relative to cbase): two granules land now, the third is in flight
    XMI_HD void in_init(uint32_t byte_pos)
    {
        loaded = byte_pos & ~(ICHUNK - 1u);
        issue_load(); land_load();
        issue_load(); land_load();
        issue_load();
        in_pos = byte_pos & ~3u;
        bits = 0; nbits = 0;
        need32();
        drop(8u * (byte_pos & 3u));
    }
    XMI_HD void need32()          // at least 32 valid bits (the tail of the stream is followed by padding)
    {
        if (nbits < 32u) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(m->iring + (in_pos & (IRING - 1u)));
            bits |= (uint64_t)w << nbits;
            nbits += 32u;
            in_pos += 4u;
            if (in_pos > cend + 12u) err = err ? err : ERR_IN;
            if (loaded - in_pos < 64u) { land_load(); issue_load(); }
        }
    }
    XMI_HD uint32_t peek(uint32_t n) const { return (uint32_t)bits & ((1u << n) - 1u); }       // n <= 16
    XMI_HD void drop(uint32_t n) { bits >>= n; nbits -= n; }
    XMI_HD uint32_t get(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }

    // ---- output ----
    XMI_HD void put_byte(uint32_t pos, uint32_t b) { m->oring[pos & (ORING - 1u)] = (uint8_t)b; }
    // store ring bytes [lo, hi) to global memory: whole aligned 16-byte lines with one store, the rest byte by byte
    XMI_HD void store_range(uint32_t lo, uint32_t hi)
    {
        for (uint32_t c = (lo & ~15u) + 16u * gl; c < hi; c += 16u * GS) {
            if (c >= lo && c + 16u <= hi) {
                store16(vbase + c, load16(m->oring + (c & (ORING - 1u))));
            } else {
                const uint32_t a = c > lo ? c : lo, b = c + 16u < hi ? c + 16u : hi;
                for (uint32_t p = a; p < b; ++p) vbase[p] = m->oring[p & (ORING - 1u)];
            }
        }
    }
    XMI_HD void maybe_flush()
    {
        while (op - flushed >= FLUSH) {
            chain_sync();
            drain_stores();                     // the flush before this one: long complete
            gsafe = flushed;
            const uint32_t hi = op & ~15u;
            store_range(flushed, hi);
            flushed = hi;
        }
    }
    XMI_HD void final_flush()
    {
        chain_sync();
        store_range(flushed, op);
        flushed = op;
    }
    XMI_HD uint32_t byte_at(uint32_t pos) const       // a byte produced earlier
    {
#if defined(XMI_TIMING_NO_FAR_READS)
        // TIMING ONLY, the bytes are WRONG: every source byte from the ring, none read back from global memory -- what the far
        // reads cost the launch (tools/ab_inflate_far.sh; round 5 measured 4 ms of 30.6 for the serial token loop)
        return (uint32_t)m->oring[pos & (ORING - 1u)];
#elif defined(XMI_GLOBAL_WINDOW) && XMI_DEVICE
        uint32_t b;
        if (pos >= gsafe) {
            b = m->oring[pos & (ORING - 1u)];
        } else {
            b = vbase[pos];
            asm volatile("" : "+v"(b));        // keeps the two loads apart (merged, they become one flat_load)
        }
        return b;
#else
        return pos >= gsafe ? (uint32_t)m->oring[pos & (ORING - 1u)] : (uint32_t)vbase[pos];
#endif
    }

    // ---- Huffman tables ----
    // code lengths cl[0 .. n) -> root table of RB bits, canonical triples, sorted symbols.  false: over-subscribed.
    XMI_HD bool build(const uint8_t *cl, uint32_t n, uint16_t *root, uint32_t RB, uint16_t *sym, uint16_t (*meta)[16])
    {
        XMI_STAGE(40);
        chain_sync();
        for (uint32_t i = gl; i < 16u; i += GS) m->cnt[i] = 0u;
        chain_sync();
        for (uint32_t s = gl; s < n; s += GS) count_inc(&m->cnt[cl[s]]);
        XMI_STAGE(41);
        const u128 zero = {{0u, 0u, 0u, 0u}};
        for (uint32_t i = gl * 8u; i < (1u << RB); i += GS * 8u) store16(reinterpret_cast<uint8_t *>(root + i), zero);
        chain_sync();
        uint32_t code = 0, idx = 0;
        int32_t left = 1;
        uint32_t first[16], base[16];
        first[0] = base[0] = 0;
#pragma unroll
        for (uint32_t L = 1; L < 16u; ++L) {
            const uint32_t c = m->cnt[L];
            left = (left << 1) - (int32_t)c;
            if (left < 0) return false;
            first[L] = code; base[L] = idx;
            if (gl == 0u) { meta[0][L] = (uint16_t)code; meta[1][L] = (uint16_t)(code + c); meta[2][L] = (uint16_t)idx; }
            code = (code + c) << 1;
            idx += c;
        }
        chain_sync();
#if XMI_DEVICE && !defined(XMI_SERIAL_BUILD)
        if (GS == 64) {
            // A lane per symbol, a lane per root entry (the loop below places one symbol at a time, all lanes alike: 22 k
            // instructions for a literal / length table, 4.5 % of the launch).  Symbols in canonical order: among the symbols of
            // one length a symbol's place is the number of smaller ones -- the lanes in front of it in its chunk of 64 (a ballot's
            // bits below the lane) plus the chunks in front.  Root entries: every entry looks for the code that begins its index
            // (the canonical first-code / limit test per length; a prefix code matches at one length at most).
            uint32_t Lc[5];
#pragma unroll
            for (uint32_t c = 0; c < 5u; ++c) { const uint32_t sy = c * 64u + gl; Lc[c] = sy < n ? cl[sy] : 0u; }
            for (uint32_t q = 1; q <= 15u; ++q) {
                const uint32_t b = meta[2][q];
                uint32_t run = 0;
#pragma unroll
                for (uint32_t c = 0; c < 5u; ++c) {
                    if (c * 64u >= n) break;
                    const bool mine = Lc[c] == q;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(mine);
                    if (mine) sym[b + run + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = (uint16_t)(c * 64u + gl);
                    run += (uint32_t)__popcll(mask);
                }
            }
            chain_sync();
            for (uint32_t q = 1; q <= RB; ++q) {
                const uint32_t f = meta[0][q], lim = meta[1][q], b = meta[2][q];
                if (f == lim) continue;
                for (uint32_t i = gl; i < (1u << RB); i += 64u) {
                    const uint32_t c = (bitrev32(i) >> (32u - RB)) >> (RB - q);          // the first q bits of the index, first bit on top
                    if (c >= f && c < lim) root[i] = (uint16_t)((sym[b + c - f] << 4) | q);
                }
            }
            chain_sync();
            return true;
        }
#endif
        for (uint32_t i = gl; i < 16u; i += GS) m->cnt[i] = 0u;          // now: symbols of each length placed so far
        chain_sync();
        XMI_STAGE(42);
        for (uint32_t s = 0; s < n; ++s) {
            const uint32_t L = cl[s];
            if (L == 0u) continue;
            const uint32_t k = m->cnt[L];
            uint32_t f = 0, b = 0;
#pragma unroll
            for (uint32_t q = 1; q < 16u; ++q) { f = (L == q) ? first[q] : f; b = (L == q) ? base[q] : b; }
            chain_sync();
            if (gl == 0u) { m->cnt[L] = k + 1u; sym[b + k] = (uint16_t)s; }
            if (L <= RB) {
                const uint32_t r = bitrev32(f + k) >> (32u - L);           // the code as its bits arrive
                const uint16_t e = (uint16_t)((s << 4) | L);
                for (uint32_t j = gl; j < (1u << (RB - L)); j += GS) root[r + (j << L)] = e;
            }
            chain_sync();
        }
        return true;
    }
    // one symbol; at least 15 valid bits must be buffered (need32 before)
    XMI_HD uint32_t decode(const uint16_t *root, uint32_t RB, const uint16_t *sym, const uint16_t (*meta)[16])
    {
        const uint32_t e = root[(uint32_t)bits & ((1u << RB) - 1u)];
        const uint32_t L = e & 15u;
        if (L) { drop(L); return e >> 4; }
        XMI_STAT_LONG_CODE(RB);
        const uint32_t rev = bitrev32((uint32_t)bits) >> 17;               // the next 15 bits, first bit on top
        for (uint32_t q = RB + 1u; q <= 15u; ++q) {
            const uint32_t c = rev >> (15u - q);
            const uint32_t f = meta[0][q], lim = meta[1][q];
            if (c >= f && c < lim) { drop(q); return sym[meta[2][q] + c - f]; }
        }
        err = err ? err : ERR_CODE;
        return 256u;                                                        // ends the block
    }

    // ---- blocks ----
    XMI_HD void stored_block()
    {
        drop(nbits & 7u);                                                   // to the next byte boundary
        need32();
        const uint32_t len = get(16u);
        need32();
        const uint32_t nlen = get(16u);
        if ((len ^ 0xFFFFu) != nlen) { err = ERR_STORED; return; }
        const uint32_t bp = in_pos - (nbits >> 3);                          // next unread byte of the input
        if (bp + len > cend) { err = ERR_IN; return; }
        if (op + len > oend) { err = ERR_OUT; return; }
        for (uint32_t k0 = 0; k0 < len; k0 += GS) {
            const uint32_t k = k0 + gl;
            if (k < len) put_byte(op + gl, cbase[bp + k]);                  // op has advanced by k0 already
            op += (len - k0 < (uint32_t)GS) ? len - k0 : (uint32_t)GS;
            maybe_flush();
        }
        chain_sync();
        in_init(bp + len);
    }
    XMI_HD bool fixed_tables()
    {
        for (uint32_t s = gl; s < 288u; s += GS) m->cl[s] = (uint8_t)(s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : 8);
        for (uint32_t s = gl; s < 32u; s += GS) m->cl[288u + s] = 5;
        chain_sync();
        return build(m->cl, 288u, m->lit_root, LIT_ROOT, m->lit_sym, m->lit_meta) &&
               build(m->cl + 288, 30u, m->dist_root, DIST_ROOT, m->dist_sym, m->dist_meta);
    }
    XMI_HD bool dynamic_tables()
    {
        need32();
        const uint32_t hlit = get(5u) + 257u, hdist = get(5u) + 1u, hclen = get(4u) + 4u;
        if (hlit > 286u || hdist > 30u) { err = ERR_LENGTHS; return false; }
        for (uint32_t i = gl; i < 19u; i += GS) m->cl[320u + i] = 0;
        chain_sync();
        for (uint32_t i = 0; i < hclen; ++i) {
            need32();
            const uint32_t v = get(3u);
            // the order of RFC 1951 3.2.7: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
            const uint32_t at = i < 3u ? 16u + i : i == 3u ? 0u : (i & 1u) ? (19u - i) >> 1 : 6u + (i >> 1);
            if (gl == 0u) m->cl[320u + at] = (uint8_t)v;
        }
        chain_sync();
        if (!build(m->cl + 320, 19u, m->dist_root, CLC_ROOT, m->dist_sym, m->dist_meta)) { err = ERR_OVERSUB; return false; }
        XMI_STAGE(50);
        uint32_t i = 0, prev = 0;
        const uint32_t total = hlit + hdist;
        while (i < total) {
            if (budget-- == 0u) { err = ERR_GUARD; return false; }
            need32();
            const uint32_t s = decode(m->dist_root, CLC_ROOT, m->dist_sym, m->dist_meta);
            if (err) return false;
            if (s < 16u) {
                if (gl == 0u) m->cl[i] = (uint8_t)s;
                prev = s; ++i;
                continue;
            }
            uint32_t rep, val = 0;
            if (s == 16u) {
                if (i == 0u) { err = ERR_LENGTHS; return false; }
                rep = 3u + get(2u); val = prev;
            } else if (s == 17u) {
                rep = 3u + get(3u);
            } else if (s == 18u) {
                rep = 11u + get(7u);
            } else { err = ERR_LENGTHS; return false; }
            if (i + rep > total) { err = ERR_LENGTHS; return false; }
            for (uint32_t k = gl; k < rep; k += GS) m->cl[i + k] = (uint8_t)val;
            if (s != 16u) prev = 0;
            i += rep;
        }
        chain_sync();
        if (m->cl[256] == 0) { err = ERR_NO_EOB; return false; }
        if (!build(m->cl, hlit, m->lit_root, LIT_ROOT, m->lit_sym, m->lit_meta) ||
            !build(m->cl + hlit, hdist, m->dist_root, DIST_ROOT, m->dist_sym, m->dist_meta)) { err = ERR_OVERSUB; return false; }
        return true;
    }
    XMI_HD void copy_match(uint32_t len, uint32_t dist)
    {
        if (dist >= (uint32_t)GS) {
            for (uint32_t k0 = 0; k0 < len; k0 += GS) {
                const uint32_t k = k0 + gl;
                if (k < len) put_byte(op + k, byte_at(op + k - dist));
                chain_sync();                                               // the next step may read what this one wrote
            }
        } else {
            // the source period is shorter than a step: every byte of the match is one of the `dist` bytes in front of it
            const uint32_t inv = 65536u / dist + 1u;                        // k / dist == (k * inv) >> 16 for k < 1024
            for (uint32_t k0 = 0; k0 < len; k0 += GS) {
                const uint32_t k = k0 + gl;
                if (k < len) put_byte(op + k, byte_at(op - dist + (k - ((k * inv) >> 16) * dist)));
            }
            chain_sync();
        }
        op += len;
    }
    // one token through the bit reader: a literal, a match, or the end of the block (true: the block goes on)
    XMI_HD bool serial_token()
    {
        need32();
        const uint32_t s = decode(m->lit_root, LIT_ROOT, m->lit_sym, m->lit_meta);
        if (s < 256u) {
            if (op >= oend) { err = ERR_OUT; return false; }
            if (gl == 0u) put_byte(op, s);
            XMI_STAT_LITERAL();
            ++op;
            maybe_flush();
            return true;
        }
        if (s == 256u) return false;
        if (s > 285u) { err = ERR_LENSYM; return false; }
        uint32_t len;
        if (s < 265u) len = s - 254u;
        else if (s == 285u) len = 258u;
        else { const uint32_t eb = (s - 261u) >> 2; len = ((4u + ((s - 261u) & 3u)) << eb) + 3u + get(eb); }
        need32();
        const uint32_t d = decode(m->dist_root, DIST_ROOT, m->dist_sym, m->dist_meta);
        if (err) return false;
        if (d > 29u) { err = ERR_DIST; return false; }
        uint32_t dist;
        if (d < 4u) dist = d + 1u;
        else { const uint32_t eb = (d >> 1) - 1u; dist = ((2u + (d & 1u)) << eb) + 1u + get(eb); }
        if (dist > op - ostart) { err = ERR_DIST; return false; }
        if (len > oend - op) { err = ERR_OUT; return false; }
        chain_sync();                                                   // literals written by lane 0 are in the ring
        XMI_STAT_MATCH(len, dist, op - dist < gsafe);
        copy_match(len, dist);
        maybe_flush();
        return true;
    }
    XMI_HD void huffman_block_serial()
    {
        // (no step budget in THIS loop, unlike the others: every pass drops at least one bit of the stream -- a code that matches
        // nothing sets err -- and need32() sets err once the reading position is 12 bytes past the block's compressed bytes, so
        // the loop ends after at most 8 x (compressed bytes + 16) passes whatever the data holds; the counter cost 4 % of the launch)
        while (!err && serial_token()) { }
    }

#if XMI_DEVICE && !defined(XMI_SERIAL_TOKENS)
    // ---- the wide token loop (one chain = one whole wave): what ships; -DXMI_SERIAL_TOKENS builds the loop above instead ----------
    // The serial loop above computes every token in all 64 lanes alike: ~90 instructions a token, and instruction issue is what
    // the launch is bound by (a wave64 vector instruction holds its SIMD for four cycles, eight waves a SIMD; the CU's one scalar
    // unit is as full).  Here the lanes do different work, twice over -- 30.7 -> 20.0 ms per GB of BAM, profiles/
    // r05_ab_inflate_variants.txt.
    // DECODING: lane i decodes the WHOLE token that would begin at bit P + i of the stream -- literal, or length + extra bits +
    // distance + extra bits (at most 48 bits: each lane reads its own 64) -- with both root-table lookups, branch-free; which
    // lanes really begin a token is found by walking from lane 0 (bit P begins one) with the scalar unit: read the lane's packed
    // token (v_readlane), step on by its bit count.  A window of 64 bits holds about six tokens of BAM data.
    // PRODUCING: the walk does not copy.  It deals the tokens out over the lanes of a BATCH of up to 64 output bytes -- lane j
    // learns what byte j of the batch is made of: a literal, or "the byte `dist` in front of me" (a match longer than the batch's
    // rest goes on in the next batch: the same rule) -- and emit_batch() makes all of them at once: sources in front of the batch
    // come from the ring or from memory as ever, sources INSIDE the batch (overlapping matches, a match that copies a token of
    // the same batch) are chased lane to lane (ds_bpermute pointer jumping: a chain of references halves every round).
    // A lane whose bits need a code beyond the root tables (4 % of BAM's tokens) or hold no valid token is marked; when the walk
    // reaches one, long_token() decodes that one token with the canonical walk, all lanes alike, and the walk goes on; only what
    // is no token at all goes to the serial reader, which names the error.
    // Packed token: bits 0-5 its length in bits, 6-14 the bytes it makes (0: not a token the walk can deal out -- then bits 15..
    // say which: T_EOB, T_SLOW), 15-31 what they are made of: V_LIT | byte, or the distance.
    static constexpr uint32_t V_LIT = 0x10000u, T_EOB = 1u, T_SLOW = 2u;
    XMI_HD void seek_bits(uint32_t bitpos)            // the serial reader continues at this bit of the stream
    {
        in_pos = (bitpos >> 5) << 2;
        const uint32_t w = *reinterpret_cast<const uint32_t *>(m->iring + (in_pos & (IRING - 1u)));
        bits = (uint64_t)w;
        nbits = 32u;
        in_pos += 4u;
        drop(bitpos & 31u);
    }
    XMI_HD uint32_t window_tokens(uint32_t bitpos) const
    {
        const uint32_t b = bitpos + gl, w0 = (b >> 5) << 2, sh = b & 31u;
        const uint32_t a0 = *reinterpret_cast<const uint32_t *>(m->iring + (w0 & (IRING - 1u)));
        const uint32_t a1 = *reinterpret_cast<const uint32_t *>(m->iring + ((w0 + 4u) & (IRING - 1u)));
        const uint32_t a2 = *reinterpret_cast<const uint32_t *>(m->iring + ((w0 + 8u) & (IRING - 1u)));
        const uint32_t x0 = __builtin_amdgcn_alignbit(a1, a0, sh), x1 = __builtin_amdgcn_alignbit(a2, a1, sh);
        const uint32_t e = m->lit_root[x0 & ((1u << LIT_ROOT) - 1u)];
        const uint32_t L = e & 15u, sym = e >> 4;
        // as a length symbol: RFC 1951's table as arithmetic -- symbols 257 + ls, ls < 4: lengths 3 + ls; from ls = 4 on, four
        // symbols share a count of extra bits, (ls >> 2) - 1, and lengths ((4 | ls & 3) << extra) + 3 + the extra bits' value
        // (which is ls + 3 again for ls = 4 .. 7); 285 is 258 with no extra bits where the rule would say 259 and six
        const uint32_t ls = sym - 257u;
        const bool top = ls == 28u;
        const uint32_t leb0 = (ls >> 2) > 1u ? (ls >> 2) - 1u : 0u;
        const uint32_t leb = top ? 0u : leb0;
        const uint32_t lman = ls < 4u ? ls : (4u | (ls & 3u));
        const uint32_t len = (lman << leb0) + 3u - (top ? 1u : 0u) + __builtin_amdgcn_ubfe(x0, L, leb);
        const uint32_t o2 = L + leb;                                        // <= 20
        const uint32_t y = __builtin_amdgcn_alignbit(x1, x0, o2);
        const uint32_t de = m->dist_root[y & ((1u << DIST_ROOT) - 1u)];
        const uint32_t DL = de & 15u, d = de >> 4;
        // distance symbols the same way: d < 2: 1 + d; from 2 on, two symbols share (d >> 1) - 1 extra bits, ((2 | d & 1) << extra) + 1
        const uint32_t deb = (d >> 1) > 1u ? (d >> 1) - 1u : 0u;
        const uint32_t dman = d < 2u ? d : (2u | (d & 1u));
        const uint32_t dist = (dman << deb) + 1u + __builtin_amdgcn_ubfe(y, DL, deb);
        const bool is_lit = sym < 256u, is_eob = sym == 256u;
        const bool bad = L == 0u || (sym > 256u && (ls > 28u || DL == 0u || d > 29u));
        const uint32_t total = sym > 256u ? o2 + DL + deb : L;              // <= 48
        const uint32_t n = (bad || is_eob) ? 0u : is_lit ? 1u : len;
        const uint32_t v = bad ? T_SLOW : is_eob ? T_EOB : is_lit ? (V_LIT | sym) : dist;
        return total | (n << 6) | (v << 15);
    }
    // The token at one bit of the stream with the codes BEYOND the root tables decoded too (the canonical first-code / limit walk of
    // decode()), computed by all lanes alike, in the packed form of window_tokens -- for the lane the walk has reached and found
    // marked: 4 % of the tokens of BAM data (tools/inflate_token_stats.cpp: literal / length codes longer than 10 bits, distance
    // codes longer than 8), too many to hand each to the serial reader, which needs the open batch written first and a fresh
    // window behind it.  T_SLOW again: no valid token here (the serial reader then names the error).
    XMI_HD uint32_t long_token(uint32_t bitpos) const
    {
        const uint32_t w0 = (bitpos >> 5) << 2, sh = bitpos & 31u;
        const uint32_t a0 = *reinterpret_cast<const uint32_t *>(m->iring + (w0 & (IRING - 1u)));
        const uint32_t a1 = *reinterpret_cast<const uint32_t *>(m->iring + ((w0 + 4u) & (IRING - 1u)));
        const uint32_t a2 = *reinterpret_cast<const uint32_t *>(m->iring + ((w0 + 8u) & (IRING - 1u)));
        const uint32_t x0 = __builtin_amdgcn_alignbit(a1, a0, sh), x1 = __builtin_amdgcn_alignbit(a2, a1, sh);
        const uint32_t slow = T_SLOW << 15;
        const uint32_t e = m->lit_root[x0 & ((1u << LIT_ROOT) - 1u)];
        uint32_t L = e & 15u, sym = e >> 4;
        if (L == 0u) {
            const uint32_t rev = bitrev32(x0) >> 17;                        // the next 15 bits, first bit on top
            for (uint32_t q = LIT_ROOT + 1u; q <= 15u; ++q) {
                const uint32_t c = rev >> (15u - q), f = m->lit_meta[0][q], lim = m->lit_meta[1][q];
                if (c >= f && c < lim) { L = q; sym = m->lit_sym[m->lit_meta[2][q] + c - f]; break; }
            }
            if (L == 0u) return slow;
        }
        if (sym < 256u) return L | (1u << 6) | ((V_LIT | sym) << 15);
        if (sym == 256u) return L | (T_EOB << 15);
        const uint32_t ls = sym - 257u;
        if (ls > 28u) return slow;
        const bool top = ls == 28u;
        const uint32_t leb0 = (ls >> 2) > 1u ? (ls >> 2) - 1u : 0u;
        const uint32_t leb = top ? 0u : leb0;
        const uint32_t lman = ls < 4u ? ls : (4u | (ls & 3u));
        const uint32_t len = (lman << leb0) + 3u - (top ? 1u : 0u) + __builtin_amdgcn_ubfe(x0, L, leb);
        const uint32_t o2 = L + leb;                                        // <= 20
        const uint32_t y = __builtin_amdgcn_alignbit(x1, x0, o2);
        const uint32_t de = m->dist_root[y & ((1u << DIST_ROOT) - 1u)];
        uint32_t DL = de & 15u, d = de >> 4;
        if (DL == 0u) {
            const uint32_t rev = bitrev32(y) >> 17;
            for (uint32_t q = DIST_ROOT + 1u; q <= 15u; ++q) {
                const uint32_t c = rev >> (15u - q), f = m->dist_meta[0][q], lim = m->dist_meta[1][q];
                if (c >= f && c < lim) { DL = q; d = m->dist_sym[m->dist_meta[2][q] + c - f]; break; }
            }
            if (DL == 0u) return slow;
        }
        if (d > 29u) return slow;
        const uint32_t deb = (d >> 1) > 1u ? (d >> 1) - 1u : 0u;
        const uint32_t dman = d < 2u ? d : (2u | (d & 1u));
        const uint32_t dist = (dman << deb) + 1u + __builtin_amdgcn_ubfe(y, DL, deb);
        return (o2 + DL + deb) | (len << 6) | (dist << 15);
    }
    // bytes [op, op + T) from what the lanes of the batch hold
    XMI_HD void emit_batch(uint32_t lane_val, uint32_t T)
    {
        if (T > oend - op) { err = ERR_OUT; return; }
        const uint32_t pos = op + gl;
        const bool active = gl < T, is_lit = (lane_val & V_LIT) != 0u;
        const uint32_t dist = lane_val & 0xFFFFu;
        const bool copies = active && !is_lit;
        if (__builtin_amdgcn_ballot_w64(copies && dist > pos - ostart) != 0ull) { err = ERR_DIST; return; }
        const uint32_t src = pos - dist;
        const bool inside = copies && src >= op;
        uint32_t ref = inside ? src - op : gl;
        uint32_t byte = lane_val & 0xFFu;
        if (copies && !inside) byte = byte_at(src);
        for (;;) {                                                          // ref[j] = ref[ref[j]] until nothing moves
            const uint32_t r2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ref << 2), (int)ref);
            const bool moved = r2 != ref;
            ref = r2;
            if (__builtin_amdgcn_ballot_w64(moved) == 0ull) break;
        }
        byte = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ref << 2), (int)byte);
        if (active) put_byte(pos, byte);
        op += T;
        chain_sync();
        maybe_flush();
    }
    XMI_HD void huffman_block()
    {
        if (GS != 64) { huffman_block_serial(); return; }
        uint32_t P = in_pos * 8u - nbits;                                   // the stream's next bit
        uint32_t lane_val = 0, fill = 0;                                    // the open batch: what each byte is made of, bytes dealt out
        bool more = true;
        // Every pass of this loop ends the block (more = false, err) or moves on: it refills the ring (only while P is within 64
        // bytes of `loaded`, and P never passes cend + 13 bytes), or P advances -- a pass that refills nothing and raises no
        // error has P below `stop`, so at least one window is looked at, and its first token either is dealt out (its bit count
        // is >= 1), or is handed to emit_batch / the serial reader below, which consume it or set err.
        while (more && !err) {
            if ((P >> 3) > cend + 12u) { err = ERR_IN; break; }
            if (loaded - (P >> 3) < 64u) { land_load(); issue_load(); }
            // The scalar unit's part, window after window while the input ring holds what a window reads (bits below `stop`):
            // tokens that fit the open batch whole are dealt out; it ends at `stop` (t = 0) or at the first token that needs
            // more than that: t, the token at bit pp.  Every value that steers these loops comes out of v_readfirstlane /
            // v_readlane, so that the compiler keeps them, and the branches on them, on the scalar unit.
            // (a window at byte b reads the ring up to b + 20: b <= loaded - 64, the same margin as the refill above; b <= cend + 12)
            const uint32_t cut = cend + 13u < loaded - 63u ? cend + 13u : loaded - 63u;
            const uint32_t stop = __builtin_amdgcn_readfirstlane(cut * 8u);
            uint32_t pp = __builtin_amdgcn_readfirstlane(P), ff = __builtin_amdgcn_readfirstlane(fill), t = 0;
            while (pp < stop) {
                const uint32_t info = window_tokens(pp);
                uint32_t ss = 0;
                bool out = false;
                for (;;) {
                    t = __builtin_amdgcn_readlane(info, ss);
                    uint32_t n = (t >> 6) & 0x1FFu;
                    if (n == 0u) {
                        if ((t >> 15) == T_SLOW) {                          // a code beyond the root tables, most likely
                            t = __builtin_amdgcn_readfirstlane(long_token(pp + ss));
                            n = (t >> 6) & 0x1FFu;
                        }
                        if (n == 0u) { out = true; break; }                 // the end of the block, or no token at all
                    }
                    if (ff + n > 64u) {
                        // a token longer than the batch's rest: batch after batch, and on in this window
                        const uint32_t v = t >> 15;
                        bool failed = false;
                        do {
                            const uint32_t room = 64u - ff, take = n < room ? n : room;
                            lane_val = (gl - ff < take) ? v : lane_val;
                            ff += take;
                            n -= take;
                            if (ff == 64u) {
                                emit_batch(lane_val, 64u);
                                ff = 0;
                                failed = __builtin_amdgcn_readfirstlane((uint32_t)err) != 0u;
                            }
                        } while (n && !failed);
                        if (failed) { out = true; break; }
                    } else {
                        lane_val = (gl - ff < n) ? (t >> 15) : lane_val;
                        ff += n;
                    }
                    ss += t & 63u;
                    if (ss >= 64u) break;
                }
                pp += ss;
                if (out) break;
                t = 0;
            }
            P = pp; fill = ff;
            if (t == 0u || err) continue;                                   // the ring needs input (or the stream has run out)
            // the end of the block, or a token for the serial reader
            if (fill) { emit_batch(lane_val, fill); fill = 0; }
            if ((t >> 15) == T_EOB) { P += t & 63u; more = false; break; }
            if (err) break;
            chain_sync();
            seek_bits(P);
            more = serial_token();
            P = in_pos * 8u - nbits;
            continue;
        }
        if (fill && !err) emit_batch(lane_val, fill);
        chain_sync();
        seek_bits(P);
    }
#else
    XMI_HD void huffman_block() { huffman_block_serial(); }
#endif

    // the whole BGZF block: comp + coff .. + clen -> out + ooff .. + isize.  Returns the status.
    XMI_HD int run(ChainMem *mem, uint32_t lane_in_chain, const uint8_t *comp, uint64_t coff, uint32_t clen,
                   uint8_t *out, uint64_t ooff, uint32_t isize)
    {
        m = mem; gl = lane_in_chain; err = OK;
        const uint64_t cb = coff & ~(uint64_t)(ICHUNK - 1u);
        cbase = comp + cb;
        cend = (uint32_t)(coff - cb) + clen;
        // The integer round trip makes the window accesses flat_* operations on the device (the address space is lost).
        // XMI_GLOBAL_WINDOW (A/B): pointer arithmetic instead, global_load / global_store -- measured 3 % SLOWER on one box
        // (38.9 against 40.0 ms per GB, profiles/r05_ab_inflate_variants.txt), so the flat form stays.
        ostart = (uint32_t)(reinterpret_cast<uintptr_t>(out + ooff) & 15u);
#ifdef XMI_GLOBAL_WINDOW
        vbase = out + ooff - ostart;
#else
        vbase = reinterpret_cast<uint8_t *>(reinterpret_cast<uintptr_t>(out + ooff) & ~(uintptr_t)15);
#endif
        op = flushed = gsafe = ostart;
        oend = ostart + isize;
        XMI_STAGE(1);
        in_init((uint32_t)(coff - cb));
        XMI_STAGE(2);
        bool last = false;
        budget = 8u * clen + isize + 4096u;          // a symbol costs at least one bit or yields at least one byte
        while (!last && !err) {
            if (budget-- == 0u) { err = ERR_GUARD; break; }
            need32();
            last = get(1u) != 0u;
            const uint32_t type = get(2u);
            XMI_STAGE(10u + type);
            XMI_STAT_BLOCK(type);
            if (type == 0u) stored_block();
            else if (type == 3u) err = ERR_BTYPE;
            else {
                const bool have = type == 1u ? fixed_tables() : dynamic_tables();
                XMI_STAGE(20);
                if (have) huffman_block();
                else err = err ? err : ERR_OVERSUB;
                XMI_STAGE(21);
            }
        }
        if (!err && op != oend) err = ERR_SHORT;
        if (!err && in_pos - (nbits >> 3) > cend) err = ERR_IN;
        XMI_STAGE(30);
        final_flush();
        XMI_STAGE(31);
        return err;
    }
};

}  // namespace xmi
