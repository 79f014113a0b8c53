// inflate_core_host.cpp -- TEST INFRASTRUCTURE (never linked into a library): the decoder logic of
// xenomapper_amd/csrc/xm_inflate_core.h compiled for the host as a chain of ONE lane, against zlib, on this CPU-only
// build container.  What it can check: Huffman table construction, the slow path for long codes, stored / fixed / dynamic
// blocks, the output ring with its flush / read-back rule, the input ring, error detection.  What it cannot: the
// cross-lane cooperation of GS > 1 lanes -- that is tests/test_inflate_gpu.py on the GPU.
//   usage: inflate_core_host <file.bam|file.gz-with-BGZF-blocks> ...   (every BGZF block against zlib)
//          inflate_core_host --fuzz N seed                            (random inputs deflated by zlib at several levels / strategies)
#include "../xenomapper_amd/csrc/xm_inflate_core.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

static int run_block(const std::vector<uint8_t> &comp, uint64_t coff, uint32_t clen, std::vector<uint8_t> &out, uint64_t ooff, uint32_t isize)
{
    static xmi::ChainMem mem;
    xmi::Chain<1> ch;
    return ch.run(&mem, 0u, comp.data(), coff, clen, out.data(), ooff, isize);
}

static bool check_stream(const std::vector<uint8_t> &raw, const std::vector<uint8_t> &deflated, unsigned shift, const char *what)
{
    // the stream at an arbitrary byte offset inside a padded buffer, the output at an arbitrary offset too
    std::vector<uint8_t> comp(shift + deflated.size() + 2048, 0xA5);
    memcpy(comp.data() + shift, deflated.data(), deflated.size());
    const uint64_t ooff = 16 + (shift * 7) % 16;
    std::vector<uint8_t> out(ooff + raw.size() + 64 + 16, 0xEE);
    // the decoder works on a 16-byte aligned view of the output; keep the vector's own alignment out of the picture
    const int rc = run_block(comp, shift, (uint32_t)deflated.size(), out, ooff, (uint32_t)raw.size());
    if (rc != 0) { fprintf(stderr, "%s: status %d (shift %u, %zu -> %zu bytes)\n", what, rc, shift, deflated.size(), raw.size()); return false; }
    if (!raw.empty() && memcmp(out.data() + ooff, raw.data(), raw.size()) != 0) {
        size_t at = 0;
        while (out[ooff + at] == raw[at]) ++at;
        fprintf(stderr, "%s: byte %zu of %zu differs (shift %u)\n", what, at, raw.size(), shift);
        return false;
    }
    for (uint64_t i = 0; i < ooff; ++i) if (out[i] != 0xEE) { fprintf(stderr, "%s: wrote in front of the block\n", what); return false; }
    for (size_t i = ooff + raw.size(); i < out.size(); ++i) if (out[i] != 0xEE) { fprintf(stderr, "%s: wrote behind the block (+%zu)\n", what, i - ooff - raw.size()); return false; }
    return true;
}

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &raw, int level, int strategy)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> out(deflateBound(&zs, raw.size()) + 64);
    zs.next_in = const_cast<Bytef *>(raw.data()); zs.avail_in = (uInt)raw.size();
    zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

static int fuzz(int n, unsigned seed)
{
    std::mt19937 rng(seed);
    int bad = 0;
    for (int it = 0; it < n; ++it) {
        const size_t len = (it % 7 == 0) ? rng() % 64 : (it % 5 == 0) ? 60000 + rng() % 5536 : rng() % 20000;
        std::vector<uint8_t> raw(len);
        const int kind = (int)(rng() % 6);
        for (size_t i = 0; i < len; ++i) {
            switch (kind) {
            case 0: raw[i] = (uint8_t)rng(); break;                                        // incompressible -> stored blocks
            case 1: raw[i] = (uint8_t)("ACGTN"[rng() % 5]); break;
            case 2: raw[i] = (uint8_t)(i % 37 < 30 ? 'F' : ',' + rng() % 40); break;      // long runs: distance 1 matches
            case 3: raw[i] = (uint8_t)((i / 300) % 2 ? "read_name_"[i % 10] : rng() % 7);  break;
            case 4: raw[i] = (uint8_t)(rng() % 3 == 0 ? rng() : 0); break;
            default: raw[i] = (uint8_t)(i & 0xFF); break;
            }
        }
        if (kind == 3 && len > 4000) memcpy(raw.data() + len - 2000, raw.data(), 2000);   // far matches (beyond the ring)
        static const int levels[] = {0, 1, 4, 6, 9};
        static const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
        const int level = levels[rng() % 5], strategy = strategies[rng() % 5];
        const std::vector<uint8_t> d = deflate_raw(raw, level, strategy);
        char what[96];
        snprintf(what, sizeof what, "fuzz %d (kind %d, level %d, strategy %d, %zu bytes)", it, kind, level, strategy, len);
        if (!check_stream(raw, d, (unsigned)(rng() % 300), what)) ++bad;
        // damaged streams must end with a status, never crash or write outside the block (ASan build checks the latter)
        if (d.size() > 8 && it % 3 == 0) {
            std::vector<uint8_t> dd = d;
            dd[rng() % dd.size()] ^= (uint8_t)(1u << (rng() % 8));
            std::vector<uint8_t> comp(dd.size() + 2048, 0);
            memcpy(comp.data() + 5, dd.data(), dd.size());
            std::vector<uint8_t> out(32 + raw.size() + 64, 0xEE);
            (void)run_block(comp, 5, (uint32_t)dd.size(), out, 16, (uint32_t)raw.size());
            for (size_t i = 0; i < 16; ++i) if (out[i] != 0xEE) { fprintf(stderr, "%s: damaged stream wrote in front\n", what); ++bad; break; }
            for (size_t i = 16 + raw.size() + 16; i < out.size(); ++i) if (out[i] != 0xEE) { fprintf(stderr, "%s: damaged stream wrote behind\n", what); ++bad; break; }
        }
    }
    printf("fuzz: %d streams, %d failures\n", n, bad);
    return bad;
}

static int check_file(const char *path)
{
    FILE *fh = fopen(path, "rb");
    if (!fh) { perror(path); return 1; }
    std::vector<uint8_t> d;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, fh)) > 0) d.insert(d.end(), buf, buf + got);
    fclose(fh);
    const size_t file_len = d.size();
    d.resize(file_len + 2048, 0);
    size_t p = 0, n_blocks = 0, total = 0;
    int bad = 0;
    while (p + 18 <= file_len) {
        if (d[p] != 0x1f || d[p + 1] != 0x8b) { fprintf(stderr, "%s: not a gzip member at %zu\n", path, p); return 1; }
        const uint32_t xlen = d[p + 10] | (d[p + 11] << 8);
        const uint32_t bsize = d[p + 16] | (d[p + 17] << 8);                   // BGZF writes the BC subfield first
        const size_t tot = (size_t)bsize + 1;
        const uint64_t coff = p + 12 + xlen;
        const uint32_t clen = (uint32_t)(tot - 12 - xlen - 8);
        const uint32_t isize = d[p + tot - 4] | (d[p + tot - 3] << 8) | (d[p + tot - 2] << 16) | ((uint32_t)d[p + tot - 1] << 24);
        const uint32_t crc = d[p + tot - 8] | (d[p + tot - 7] << 8) | (d[p + tot - 6] << 16) | ((uint32_t)d[p + tot - 5] << 24);
        std::vector<uint8_t> out(32 + isize + 32, 0xEE);
        const int rc = run_block(d, coff, clen, out, 16 + n_blocks % 16, isize);
        const uint32_t got_crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), out.data() + 16 + n_blocks % 16, isize);
        if (rc != 0 || got_crc != crc) { fprintf(stderr, "%s: block %zu at %zu: status %d crc %08x want %08x\n", path, n_blocks, p, rc, got_crc, crc); ++bad; }
        total += isize;
        ++n_blocks;
        p += tot;
    }
    printf("%s: %zu blocks, %zu bytes inflated, %d bad\n", path, n_blocks, total, bad);
    return bad;
}

int main(int argc, char **argv)
{
    int bad = 0;
    for (int a = 1; a < argc; ++a) {
        if (std::string(argv[a]) == "--fuzz" && a + 2 < argc) { bad += fuzz(atoi(argv[a + 1]), (unsigned)atoi(argv[a + 2])); a += 2; }
        else bad += check_file(argv[a]);
    }
    return bad ? 1 : 0;
}
