// inflate_gpu_check.hip -- stand-alone check of the GPU BGZF decoder (no Python, starts in a second): every block of the
// given BGZF files through xm_bgzf_inflate_dev + xm_bgzf_crc32_dev against zlib on the host; prints per-file verdicts and
// the first bad blocks.   build: hipcc -O3 --offload-arch=gfx950 -I include tools/inflate_gpu_check.hip -L xenomapper_amd
//                                  -lxenomapper_hip -lz -o build/inflate_gpu_check        (run under `timeout`)
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>
#include <dlfcn.h>

#include "xenomapper_bgzf.h"

typedef int (*set_trace_fn)(uint32_t *);                                                      // only in -DXMI_TRACE builds of the library
static uint32_t *g_trace = nullptr;
static const int TRACE_WORDS = 1 << 16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

static int check(xm_ctx *ctx, const char *path, int reps)
{
    FILE *fh = fopen(path, "rb");
    if (!fh) { perror(path); return 1; }
    std::vector<uint8_t> img;
    static uint8_t buf[1 << 20];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, fh)) > 0) img.insert(img.end(), buf, buf + got);
    fclose(fh);
    std::vector<xm_bgzf_block> blocks(img.size() / 28 + 16);
    std::vector<uint32_t> crc(blocks.size());
    uint64_t n = 0, next = 0, total = 0;
    if (xm_bgzf_index(img.data(), img.size(), 0, ~0ull >> 2, blocks.data(), crc.data(), blocks.size(), &n, &next, &total) != XM_OK || next != img.size()) {
        printf("%s: not a BGZF image\n", path);
        return 1;
    }
    uint8_t *d_comp, *d_out;
    xm_bgzf_block *d_blocks;
    uint32_t *d_status, *d_work, *d_crc;
    CK(hipMalloc(&d_comp, img.size() + XMB_COMP_PAD));
    CK(hipMemset(d_comp, 0, img.size() + XMB_COMP_PAD));
    CK(hipMemcpy(d_comp, img.data(), img.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, total + 256));
    CK(hipMemset(d_out, 0xEE, total + 256));
    CK(hipMalloc(&d_blocks, (n + 1) * sizeof(xm_bgzf_block)));
    CK(hipMemcpy(d_blocks, blocks.data(), n * sizeof(xm_bgzf_block), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_status, (n + 1) * 4)); CK(hipMemset(d_status, 0xFF, (n + 1) * 4));
    CK(hipMalloc(&d_crc, (n + 1) * 4));
    CK(hipMalloc(&d_work, 16));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    float best = 1e30f, best_crc = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        if (xm_bgzf_inflate_dev(ctx, nullptr, d_comp, d_blocks, n, d_out, d_status, d_work) != XM_OK) { printf("inflate launch failed\n"); return 2; }
        CK(hipEventRecord(e1));
        if (xm_bgzf_crc32_dev(ctx, nullptr, d_out, d_blocks, n, d_crc) != XM_OK) { printf("crc launch failed\n"); return 2; }
        CK(hipEventRecord(e2));
        if (g_trace) {                                                       // do not block: watch the launch from outside
            const auto t0 = std::chrono::steady_clock::now();
            while (hipEventQuery(e2) == hipErrorNotReady) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
                    printf("launch did not finish within 5 s; stages of the chains that started:\n");
                    for (int i = 0; i < TRACE_WORDS; ++i)
                        if (g_trace[i]) printf("  chain %d: stage %u (block %u)\n", i, g_trace[i] & 0xFFu, g_trace[i] >> 8);
                    fflush(stdout);
                    _exit(3);
                }
            }
        }
        CK(hipEventSynchronize(e2));
        float a, b;
        CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
        if (a < best) best = a;
        if (b < best_crc) best_crc = b;
    }
    std::vector<uint8_t> out(total + 256);
    std::vector<uint32_t> status(n + 1), gcrc(n + 1);
    CK(hipMemcpy(out.data(), d_out, total + 256, hipMemcpyDeviceToHost));
    CK(hipMemcpy(status.data(), d_status, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gcrc.data(), d_crc, n * 4, hipMemcpyDeviceToHost));
    // zlib on the host, block by block
    uint64_t bad = 0, bad_crc = 0, bad_bytes = 0;
    std::vector<uint8_t> ref(65536);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t b = 0; b < n; ++b) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        inflateInit2(&zs, -15);
        zs.next_in = img.data() + blocks[b].cdata_off; zs.avail_in = blocks[b].cdata_len;
        zs.next_out = ref.data(); zs.avail_out = 65536;
        inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        const bool same = zs.total_out == blocks[b].isize && memcmp(ref.data(), out.data() + blocks[b].out_off, blocks[b].isize) == 0;
        if (status[b] != 0) { if (bad < 5) printf("  block %llu: status %u (%s)\n", (unsigned long long)b, status[b], xm_bgzf_strerror(status[b])); ++bad; }
        else if (!same) {
            if (bad_bytes < 5) {
                uint32_t at = 0;
                while (at < blocks[b].isize && ref[at] == out[blocks[b].out_off + at]) ++at;
                printf("  block %llu: byte %u of %u differs (got %02x want %02x)\n", (unsigned long long)b, at, blocks[b].isize, out[blocks[b].out_off + at], ref[at]);
            }
            ++bad_bytes;
        }
        if (gcrc[b] != crc[b]) ++bad_crc;
    }
    const double zs_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    bool guard = true;
    for (uint64_t i = total; i < total + 256; ++i) guard &= out[i] == 0xEE;
    if (g_trace) memset(g_trace, 0, TRACE_WORDS * 4);
    printf("%s: %llu blocks, %.1f MB -> %.1f MB; GPU inflate %.3f ms = %.2f GB/s of output, crc %.3f ms; bad status %llu, bad bytes %llu, bad crc %llu, "
           "tail %s; zlib one core %.2f GB/s\n", path, (unsigned long long)n, img.size() / 1e6, total / 1e6, best, total / (best * 1e-3) / 1e9,
           best_crc, (unsigned long long)bad, (unsigned long long)bad_bytes, (unsigned long long)bad_crc, guard ? "clean" : "OVERWRITTEN", total / zs_s / 1e9);
    (void)hipFree(d_comp); (void)hipFree(d_out); (void)hipFree(d_blocks); (void)hipFree(d_status); (void)hipFree(d_crc); (void)hipFree(d_work);
    return (bad || bad_bytes || bad_crc || !guard) ? 1 : 0;
}

int main(int argc, char **argv)
{
    xm_ctx *ctx = nullptr;
    if (xm_ctx_create(0, &ctx) != XM_OK) { printf("no context: %s\n", xm_last_hip_error(nullptr)); return 2; }
    int rc = 0, reps = 3;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (int a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "--trace")) {
            set_trace_fn xm_bgzf_set_trace = (set_trace_fn)dlsym(RTLD_DEFAULT, "xm_bgzf_set_trace");
            if (!xm_bgzf_set_trace) { printf("this library was not built with -DXMI_TRACE\n"); return 2; }
            CK(hipHostMalloc((void **)&g_trace, TRACE_WORDS * 4, hipHostMallocMapped | hipHostMallocCoherent));
            memset(g_trace, 0, TRACE_WORDS * 4);
            if (xm_bgzf_set_trace(g_trace) != 0) { printf("xm_bgzf_set_trace failed\n"); return 2; }
            continue;
        }
        if (!strcmp(argv[a], "--reps") && a + 1 < argc) { reps = atoi(argv[++a]); continue; }
        rc |= check(ctx, argv[a], reps);
    }
    xm_ctx_destroy(ctx);
    return rc;
}
