// inflate_token_stats.cpp -- what the GPU decoder's tokens look like on a BAM file (host build of xm_inflate_core.h, one lane):
// literals, matches, match lengths, and how many matches reach behind the 1 KiB output ring (those read the window back from
// global memory on the device).   build: g++ -O2 -std=c++17 tools/inflate_token_stats.cpp -o build/inflate_token_stats -lz
//   usage: build/inflate_token_stats file.bam
#include <cstdint>
#include <cstdio>
static uint64_t g_lit, g_match, g_match_bytes, g_far, g_far_bytes, g_dist_le[16], g_len_le[10];
#define XMI_STAT_LITERAL() (++g_lit)
#define XMI_STAT_MATCH(len, dist, behind) do { ++g_match; g_match_bytes += (len); if (behind) { ++g_far; g_far_bytes += (len); } \
    for (int k_ = 0; k_ < 16; ++k_) if ((dist) <= (1u << k_)) { ++g_dist_le[k_]; break; } \
    for (int k_ = 0; k_ < 10; ++k_) if ((len) <= (1u << k_)) { ++g_len_le[k_]; break; } } while (0)
static uint64_t g_long[16], g_blocks[4];
#define XMI_STAT_BLOCK(type) (++g_blocks[(type) & 3])
#define XMI_STAT_LONG_CODE(root_bits) (++g_long[(root_bits) & 15])
#include "../xenomapper_amd/csrc/xm_inflate_core.h"

#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *fh = fopen(argv[1], "rb");
    if (!fh) return 2;
    std::vector<uint8_t> d;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, fh)) > 0) d.insert(d.end(), buf, buf + got);
    fclose(fh);
    d.resize(d.size() + 2048, 0);
    static xmi::ChainMem mem;
    uint64_t p = 0, blocks = 0, out_bytes = 0, comp_bytes = 0;
    const uint64_t len = d.size() - 2048;
    std::vector<uint8_t> out(65536 + 64);
    while (p + 18 <= len) {
        const uint32_t xlen = d[p + 10] | (d[p + 11] << 8), bsize = d[p + 16] | (d[p + 17] << 8);
        const uint64_t total = (uint64_t)bsize + 1;
        const uint32_t isize = d[p + total - 4] | (d[p + total - 3] << 8) | (d[p + total - 2] << 16) | ((uint32_t)d[p + total - 1] << 24);
        const uint32_t clen = (uint32_t)(total - 12 - xlen - 8);
        if (isize) {
            xmi::Chain<1> ch;
            const int rc = ch.run(&mem, 0u, d.data(), p + 12 + xlen, clen, out.data(), 16, isize);
            if (rc) { fprintf(stderr, "block %llu: status %d\n", (unsigned long long)blocks, rc); return 1; }
        }
        ++blocks; out_bytes += isize; comp_bytes += clen;
        p += total;
    }
    const double tok = (double)(g_lit + g_match);
    printf("%s: %llu blocks, %llu -> %llu bytes; tokens %.0f = %.1f per block, %.2f output bytes per token\n", argv[1],
           (unsigned long long)blocks, (unsigned long long)comp_bytes, (unsigned long long)out_bytes, tok, tok / blocks, out_bytes / tok);
    printf("literals %.1f %% of tokens (%.1f %% of bytes); matches %.1f %%, mean length %.1f\n", 100.0 * g_lit / tok, 100.0 * g_lit / out_bytes,
           100.0 * g_match / tok, (double)g_match_bytes / g_match);
    printf("matches that begin behind the output ring (source older than the flush before last): %.1f %% of matches = %.1f %% of tokens, %.1f %% of bytes\n",
           100.0 * g_far / g_match, 100.0 * g_far / tok, 100.0 * g_far_bytes / out_bytes);
    printf("distance <= 2^k, k = 0..15 (%% of matches):");
    for (int k = 0; k < 16; ++k) printf(" %.1f", 100.0 * g_dist_le[k] / g_match);
    printf("\nlength <= 2^k, k = 0..9 (%% of matches):");
    for (int k = 0; k < 10; ++k) printf(" %.1f", 100.0 * g_len_le[k] / g_match);
    printf("\ncodes beyond the root tables (the wide token loop hands these tokens to the serial reader): literal/length %.2f %% of tokens, "
           "distance %.2f %% of matches (code-length code: %llu)\n", 100.0 * g_long[xmi::LIT_ROOT] / tok, 100.0 * g_long[xmi::DIST_ROOT] / g_match,
           (unsigned long long)g_long[xmi::CLC_ROOT]);
    printf("DEFLATE blocks: %llu stored, %llu fixed, %llu dynamic = %.2f per BGZF block\n", (unsigned long long)g_blocks[0], (unsigned long long)g_blocks[1],
           (unsigned long long)g_blocks[2], (double)(g_blocks[0] + g_blocks[1] + g_blocks[2]) / (double)blocks);
    return 0;
}
