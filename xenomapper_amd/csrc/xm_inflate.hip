// xm_inflate.hip -- BGZF blocks inflated on the GPU (include/xenomapper_bgzf.h): the launch around xm_inflate_core.h, the
// CRC-32 kernel, and the host-side walk over the member headers.
//
// Launch shape.  One workgroup = one wave = 64 / GS chains, each decoding one BGZF block at a time from a shared work
// counter (a block takes 0.5 ms of a chain's time alone, ~18 ms on a full chip, and blocks differ, so a static split would
// leave tails).  A chain's LDS footprint (xmi::ChainMem, ~5 KB) and the registers (95 VGPRs: 5 waves per SIMD) bound the
// residency: 2 chains per wave = 10 KB, 16 waves per CU, ~8 000 blocks in flight on the chip -- a 1 GB window of BAM
// (~16 000 blocks) is two rounds.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/xenomapper_bgzf.h"
#include "xm_inflate_core.h"
#include "xm_bamrec.h"

// Lanes per chain and waves per SIMD.  A chain's speed is its own serial instruction stream, so what counts is how many chains a CU
// holds and how little they get in each other's way.  5.1 KB of LDS per chain allows 31 of them per CU, whatever their width.
// Measured on 16 802 blocks = 1.04 GB of inflated BAM (profiles/r05_inflate_first.txt, r05_ab_inflate_variants.txt):
//   as the compiler allocates (97 VGPRs = 5 waves per SIMD):  64 lanes 22.8-25.9 GB/s (20 chains per CU, one per wave),
//       32 lanes 28.1-28.3 (31 chains, two per wave: they diverge, and a wave runs its chains' paths one after the other),
//       16 lanes 26.2, 8 lanes 23.5;
//   asked for 8 waves per SIMD (64 VGPRs, 16 dwords spilled outside the token loop):  64 lanes 31.0 GB/s -- 31 chains per CU again,
//       each alone in its wave.  That is the shipped shape, and since round 6 the ONLY one the device build accepts: the 8- and
//       4-lane builds without the trace hooks did not come back from a 4-block file in round 5 (the same source with -DXMI_TRACE
//       did, byte-exact; every loop has an exit the data cannot disable; the two ISA listings differ in control-flow layout only).
//       Several chains in one wave run under divergent exec masks and hand words from lane to lane through LDS in "program order"
//       -- an order the compiler is free to lay out per branch; a whole-wave chain has no such hand-off between diverged lanes (its
//       lanes take every branch together, the steering values come out of v_readfirstlane / v_readlane).  The cause was not
//       isolated (doing so means launching the hanging build on a shared GPU again), so the configuration is gone instead of
//       offered: xmi::Chain<GS> is instantiated with 64 lanes on the device and with 1 lane on the host (tests), nothing else.
constexpr int XM_INFLATE_GS = 64;
static_assert(XM_INFLATE_GS == 64, "a chain is a whole wave (xm_inflate.hip: narrower chains hung in round 5)");
#ifndef XM_INFLATE_WG_PER_CU
#define XM_INFLATE_WG_PER_CU 32     // upper bound on resident waves per CU (the LDS of the chains binds at 31-32)
#endif
#ifndef XM_INFLATE_WAVES_PER_EU
#define XM_INFLATE_WAVES_PER_EU 8   // 0: leave the register budget to the compiler
#endif

#ifdef XMI_TRACE
__device__ uint32_t *xm_inflate_trace = nullptr;       // host-visible words, one per chain (tools/inflate_gpu_check.hip --trace)
extern "C" int xm_bgzf_set_trace(uint32_t *host_visible_words)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(xm_inflate_trace), &host_visible_words, sizeof host_visible_words) == hipSuccess ? 0 : -3;
}
#endif

namespace {

// XM_INFLATE_WAVES_PER_EU: ask the compiler for that many waves per SIMD (8 = at most 64 VGPRs; what does not fit is spilled)
#if XM_INFLATE_WAVES_PER_EU > 0
#define XM_INFLATE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(XM_INFLATE_WAVES_PER_EU, XM_INFLATE_WAVES_PER_EU)))
#else
#define XM_INFLATE_OCCUPANCY
#endif

template <int GS>
__global__ void __launch_bounds__(64) XM_INFLATE_OCCUPANCY
inflate_kernel(const uint8_t *__restrict__ comp, const xm_bgzf_block *__restrict__ blocks, uint32_t n_blocks,
               uint8_t *__restrict__ out, uint32_t *__restrict__ status, uint32_t *__restrict__ work,
               const xm_bgzf_walk *__restrict__ walk)
{
    constexpr int G = 64 / GS;
    __shared__ xmi::ChainMem mem[G];
    const uint32_t lane = threadIdx.x, c = lane / GS, gl = lane % GS;
    xmi::Chain<GS> chain;
#ifdef XMI_TRACE
    chain.trace = xm_inflate_trace ? xm_inflate_trace + (blockIdx.x * G + c) : nullptr;
    chain.gl = gl;
#endif
    for (;;) {
#ifdef XMI_TRACE
        if (chain.trace && gl == 0u) __hip_atomic_store(chain.trace, 100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
        // the chain's next block: lane 0 takes a number, the others read it from LDS (every chain of every wave ends here
        // once the counter has passed n_blocks: the grid always drains)
        if (gl == 0u) mem[c].bcast = atomicAdd(work, 1u);
        xmi::chain_sync();
        const uint32_t b = mem[c].bcast;
        xmi::chain_sync();
        if (b >= n_blocks) break;
#ifdef XMI_TRACE
        if (chain.trace && gl == 0u) __hip_atomic_store(chain.trace, 101u + (b << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
        const xm_bgzf_block d = blocks[b];
        int rc = xmi::OK;
        if (d.isize != 0u) rc = chain.run(&mem[c], gl, comp, d.cdata_off, d.cdata_len, out, d.out_off, d.isize);
        if (gl == 0u) status[b] = (uint32_t)rc;
        // BAM windows (xm_bgzf_inflate_walk_dev): the chain follows the alignment records' block_size chain through the block it has
        // just written and reads the classifier's fields out of its records -- while the block is still near this CU.  Kernels of
        // their own doing the same over the whole window afterwards (a lane per block, a lane per record) cost 12 ms per window of
        // dependent, uncoalesced reads (profiles/r05_bam_record_kernels.txt); here it is a fraction of a percent of the launch.
        // Only bytes in front of the window's end are read, and only of records that begin in this block.
        if (walk != nullptr) {
            const xm_bgzf_walk w = walk[b];
            if (w.end > w.start) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the block's last stores have left
                const uint8_t *raw = out + w.raw_base;
                uint32_t p = w.start, n = 0;
                if (rc == xmi::OK) {
                    while (p < w.end) {                                     // every lane of the chain the same walk
                        if (w.end - p < 4u) break;                          // the size word is cut by the block's end (the window's: a tail)
                        const uint32_t size = xmrec::ld32(raw + p);
                        if (size > w.n_raw - p - 4u) break;                 // the record continues behind the window
                        if (n >= w.slot_cap) { p = 0xFFFFFFFFu; break; }    // more records than bytes / 36: not alignment records
                        if (gl == 0u) w.slots[n] = p;
                        ++n;
                        p += 4u + size;
                    }
                } else {
                    p = 0xFFFFFFFFu;
                }
                if (gl == 0u) { *w.count = n; *w.exit_at = p; }
                if (w.tags != 0u && p != 0xFFFFFFFFu) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the record starts are where the other lanes read them
                    for (uint32_t i = gl; i < n; i += GS) {
                        const xmrec::RecFields f = xmrec::parse_record(raw, w.slots[i], w.tags);
                        w.name_off[i] = f.name_off;
                        w.name_len[i] = f.name_len;
                        w.a[i] = f.a;
                        w.x[i] = f.x;
                        w.flag[i] = (uint8_t)f.flag;
                        if (w.n_cigar != nullptr) { w.n_cigar[i] = f.n_cigar; w.cig_at[i] = f.cig_at; }
                    }
                }
            }
        }
    }
}

// CRC-32 of one block per workgroup (crc32_kernel below); the pieces' CRCs are combined by multiplying with
// x^(8 * bytes behind the piece) modulo the polynomial (square-and-multiply over GF(2)).
__device__ __forceinline__ uint32_t gf2_mulmod(uint32_t a, uint32_t b)       // reflected representation, as the CRC itself
{
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        p ^= (b & 0x80000000u) ? a : 0u;                                    // bit 31 of b = x^0
        a = (a >> 1) ^ ((a & 1u) ? 0xEDB88320u : 0u);                       // a *= x
        b <<= 1;
    }
    return p;
}

typedef uint32_t crc_v4u32_any __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t crc_u32_any __attribute__((aligned(1)));

__global__ void __launch_bounds__(256)
crc32_kernel(const uint8_t *__restrict__ out, const xm_bgzf_block *__restrict__ blocks, uint32_t n_blocks, uint32_t *__restrict__ crc_out)
{
    // one workgroup per block, thread t takes the t-th of 256 equal pieces (whole 4-byte words) counted from the block's END -- the
    // first pieces of a block are the short or empty ones; four bytes per step through four tables ("slicing by 4": T[k][v] = CRC
    // of byte v followed by k zero bytes), the bytes fetched 16 per lane and load (a lane's piece is contiguous: a byte per load
    // made every load 64 cache lines for 64 bytes, and the texture path, not the arithmetic, set this kernel's time)
    __shared__ uint32_t T[4][256];
    __shared__ uint32_t wave_x[4];
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
        T[0][threadIdx.x] = c;
        uint32_t v = c;
        for (int j = 1; j < 4; ++j) {
            // one more zero byte behind: the register shifts by 8 through the byte table -- which is T[0], computable locally
            uint32_t lo = v & 0xFFu, w = lo;
            for (int k = 0; k < 8; ++k) w = (w >> 1) ^ ((w & 1u) ? 0xEDB88320u : 0u);
            v = w ^ (v >> 8);
            T[j][threadIdx.x] = v;
        }
    }
    __syncthreads();
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const xm_bgzf_block d = blocks[b];
    const uint32_t t = threadIdx.x;
    const uint32_t piece = ((d.isize + 255u) / 256u + 3u) & ~3u;              // 256 * piece >= isize
    const uint32_t behind = (255u - t) * piece;                               // bytes of the block behind this thread's piece
    const uint32_t hi = d.isize > behind ? d.isize - behind : 0u;
    const uint32_t lo = hi > piece ? hi - piece : 0u;
    const uint8_t *p = out + d.out_off;
    // the ordinary CRC-32 of the piece (an empty piece: 0); CRC(A || B) = CRC(A) * x^(8 |B|) + CRC(B) over GF(2), so the block's
    // CRC is the sum of every piece's CRC times x^(8 * bytes behind the piece)
    uint32_t c = 0;
    if (hi > lo) {
        c = 0xFFFFFFFFu;
        uint32_t i = lo;
        auto word = [&](uint32_t w) {
            c ^= w;
            c = T[3][c & 0xFFu] ^ T[2][(c >> 8) & 0xFFu] ^ T[1][(c >> 16) & 0xFFu] ^ T[0][c >> 24];
        };
        for (; i + 64u <= hi; i += 64u) {
            // 64 bytes fetched at once: the lanes' pieces are 256 bytes apart, so every lane is on a cache line of its own, and with
            // 16 bytes per trip the line had left the L2 again before its next quarter was asked for (3.0 GB fetched per GB)
            const crc_v4u32_any v0 = *reinterpret_cast<const crc_v4u32_any *>(p + i), v1 = *reinterpret_cast<const crc_v4u32_any *>(p + i + 16u),
                                v2 = *reinterpret_cast<const crc_v4u32_any *>(p + i + 32u), v3 = *reinterpret_cast<const crc_v4u32_any *>(p + i + 48u);
            word(v0.x); word(v0.y); word(v0.z); word(v0.w); word(v1.x); word(v1.y); word(v1.z); word(v1.w);
            word(v2.x); word(v2.y); word(v2.z); word(v2.w); word(v3.x); word(v3.y); word(v3.z); word(v3.w);
        }
        for (; i + 16u <= hi; i += 16u) {
            const crc_v4u32_any v = *reinterpret_cast<const crc_v4u32_any *>(p + i);
            word(v.x); word(v.y); word(v.z); word(v.w);
        }
        for (; i + 4u <= hi; i += 4u) word(*reinterpret_cast<const crc_u32_any *>(p + i));
        for (; i < hi; ++i) c = T[0][(c ^ p[i]) & 0xFFu] ^ (c >> 8);
        c ^= 0xFFFFFFFFu;
    }
    // x^(8 * behind) = q^(255 - t) with q = x^(8 * piece): q and its squares are the same for the whole workgroup (scalar work),
    // a thread multiplies the ones its exponent's bits select
    uint32_t q = 0x80000000u /* x^0 */;
    {
        uint32_t e = piece, s = 0x00800000u /* x^8 */;
        while (e) {
            if (e & 1u) q = gf2_mulmod(q, s);
            s = gf2_mulmod(s, s);
            e >>= 1;
        }
    }
    uint32_t pw = 0x80000000u;
    for (uint32_t k = 255u - t, bit = 0; bit < 8u; ++bit, k >>= 1) {
        if (k & 1u) pw = gf2_mulmod(pw, q);
        q = gf2_mulmod(q, q);
    }
    c = gf2_mulmod(c, pw);
    c ^= __shfl_xor(c, 1, 64);  c ^= __shfl_xor(c, 2, 64);  c ^= __shfl_xor(c, 4, 64);
    c ^= __shfl_xor(c, 8, 64);  c ^= __shfl_xor(c, 16, 64); c ^= __shfl_xor(c, 32, 64);
    if ((t & 63u) == 0u) wave_x[t >> 6] = c;
    __syncthreads();
    if (t == 0u) crc_out[b] = wave_x[0] ^ wave_x[1] ^ wave_x[2] ^ wave_x[3];
}

inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

}  // namespace

extern "C" {

static int index_members(const uint8_t *d, uint64_t len, uint64_t start, uint64_t max_out, xm_bgzf_block *blocks, uint32_t *crc,
                         uint64_t cap, uint64_t *n_blocks, uint64_t *next, uint64_t *out_bytes, bool prefix)
{
    if (!d || !blocks || !n_blocks || !next || !out_bytes || start > len) return XM_ERR_INVALID_ARG;
    uint64_t p = start, n = 0, acc = 0;
    while (p < len && n < cap && acc < max_out) {
        if (p + 18 > len) { if (prefix) break; return XM_ERR_INVALID_ARG; }           // (prefix: the buffer ends inside a header)
        if (d[p] != 0x1f || d[p + 1] != 0x8b || d[p + 2] != 8 || !(d[p + 3] & 4)) return XM_ERR_INVALID_ARG;
        const uint32_t xlen = le16(d + p + 10);
        if (p + 12 + xlen > len) { if (prefix) break; return XM_ERR_INVALID_ARG; }
        uint32_t bsize = 0;
        bool found = false;
        for (uint64_t q = p + 12; q + 4 <= p + 12 + xlen;) {
            const uint32_t slen = le16(d + q + 2);
            if (d[q] == 'B' && d[q + 1] == 'C' && slen == 2 && q + 6 <= len) { bsize = le16(d + q + 4); found = true; }
            q += 4 + slen;
        }
        if (!found) return XM_ERR_INVALID_ARG;
        const uint64_t total = (uint64_t)bsize + 1;
        if (total < 12 + xlen + 8) return XM_ERR_INVALID_ARG;
        if (p + total > len) { if (prefix) break; return XM_ERR_INVALID_ARG; }      // a member cut by the end of the buffer
        const uint32_t isize = le32(d + p + total - 4);
        if (isize > 65536u) return XM_ERR_INVALID_ARG;
        blocks[n].cdata_off = p + 12 + xlen;
        blocks[n].cdata_len = (uint32_t)(total - 12 - xlen - 8);
        blocks[n].isize = isize;
        blocks[n].out_off = acc;
        if (crc) crc[n] = le32(d + p + total - 8);
        acc += isize;
        ++n;
        p += total;
    }
    *n_blocks = n;
    *next = p;
    *out_bytes = acc;
    return XM_OK;
}

int xm_bgzf_index(const uint8_t *d, uint64_t len, uint64_t start, uint64_t max_out, xm_bgzf_block *blocks, uint32_t *crc,
                  uint64_t cap, uint64_t *n_blocks, uint64_t *next, uint64_t *out_bytes)
{
    return index_members(d, len, start, max_out, blocks, crc, cap, n_blocks, next, out_bytes, false);
}

int xm_bgzf_index_prefix(const uint8_t *d, uint64_t len, uint64_t start, uint64_t max_out, xm_bgzf_block *blocks, uint32_t *crc,
                         uint64_t cap, uint64_t *n_blocks, uint64_t *next, uint64_t *out_bytes)
{
    return index_members(d, len, start, max_out, blocks, crc, cap, n_blocks, next, out_bytes, true);
}

int xm_bgzf_inflate_dev(xm_ctx *ctx, void *stream, const uint8_t *comp, const xm_bgzf_block *blocks, uint64_t n_blocks,
                        uint8_t *out, uint32_t *status, uint32_t *work)
{
    return xm_bgzf_inflate_walk_dev(ctx, stream, comp, blocks, n_blocks, out, status, work, nullptr);
}

int xm_bgzf_inflate_walk_dev(xm_ctx *ctx, void *stream, const uint8_t *comp, const xm_bgzf_block *blocks, uint64_t n_blocks,
                             uint8_t *out, uint32_t *status, uint32_t *work, const xm_bgzf_walk *walk)
{
    if (!ctx || n_blocks > 0x7FFFFFFFull) return XM_ERR_INVALID_ARG;
    if (n_blocks == 0) return XM_OK;
    if (!comp || !blocks || !out || !status || !work || ((uintptr_t)comp & 15u) || ((uintptr_t)blocks & 7u)) return XM_ERR_INVALID_ARG;
    if (walk && ((uintptr_t)walk & 7u)) return XM_ERR_INVALID_ARG;
    int n_cu = 0;
    if (xm_ctx_device_info(ctx, &n_cu, nullptr, 0) != XM_OK || n_cu <= 0) return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(work, 0, sizeof(uint32_t), st) != hipSuccess) return XM_ERR_HIP;
    constexpr int G = 64 / XM_INFLATE_GS;
    const uint64_t waves_needed = (n_blocks + G - 1) / G;
    // resident waves per CU: what the chains' LDS allows (160 KB per CU), at most XM_INFLATE_WG_PER_CU
    constexpr uint64_t by_lds = (160u * 1024u) / (G * sizeof(xmi::ChainMem));
    uint64_t grid = (uint64_t)n_cu * (by_lds < XM_INFLATE_WG_PER_CU ? by_lds : XM_INFLATE_WG_PER_CU);
    if (grid > waves_needed) grid = waves_needed;
    inflate_kernel<XM_INFLATE_GS><<<(uint32_t)grid, 64, 0, st>>>(comp, blocks, (uint32_t)n_blocks, out, status, work, walk);
    return hipGetLastError() == hipSuccess ? XM_OK : XM_ERR_HIP;
}

int xm_bgzf_crc32_dev(xm_ctx *ctx, void *stream, const uint8_t *out, const xm_bgzf_block *blocks, uint64_t n_blocks, uint32_t *crc_out)
{
    if (!ctx || n_blocks > 0x7FFFFFFFull) return XM_ERR_INVALID_ARG;
    if (n_blocks == 0) return XM_OK;
    if (!out || !blocks || !crc_out) return XM_ERR_INVALID_ARG;
    crc32_kernel<<<(uint32_t)n_blocks, 256, 0, (hipStream_t)stream>>>(out, blocks, (uint32_t)n_blocks, crc_out);
    return hipGetLastError() == hipSuccess ? XM_OK : XM_ERR_HIP;
}

const char *xm_bgzf_strerror(uint32_t s)
{
    switch (s) {
    case xmi::OK: return "ok";
    case xmi::ERR_BTYPE: return "reserved DEFLATE block type";
    case xmi::ERR_STORED: return "stored block: LEN / NLEN mismatch";
    case xmi::ERR_LENGTHS: return "bad code-length sequence";
    case xmi::ERR_OVERSUB: return "over-subscribed Huffman code";
    case xmi::ERR_NO_EOB: return "no end-of-block code";
    case xmi::ERR_CODE: return "bits that match no Huffman code";
    case xmi::ERR_DIST: return "match distance beyond the start of the block";
    case xmi::ERR_OUT: return "more output than the block's ISIZE";
    case xmi::ERR_IN: return "DEFLATE stream runs past the block's compressed bytes";
    case xmi::ERR_SHORT: return "DEFLATE stream ends before ISIZE bytes";
    case xmi::ERR_LENSYM: return "invalid length symbol";
    default: return "unknown block status";
    }
}

}  // extern "C"
