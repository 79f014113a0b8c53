// xm_api.hip -- the C ABI declared in include/xenomapper_hip.h: context, workspace, the
// host-buffer convenience entry points, the device-resident entry points and per-kernel timing.
// No CPU fallback lives here: every entry point needs a context, and a context needs a gfx950 device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>

#include <string>
#include <vector>

#include <rccl/rccl.h>          // types only: the library is looked up at run time (xm_comm_init), not linked

#include "xm_kernels.h"
#include "xm_pinned.h"

struct TimedSpan {
    int kernel;
    hipEvent_t start, stop;
};

struct xm_ctx {
    int device;
    int n_cu;
    char name[128];
    uint32_t max_blocks;            // grid cap for the streaming kernels: 8 workgroups per CU
    // K2 workspace (fixed size, allocated once; one compaction in flight per context)
    uint32_t *d_gran_counts;        // [8 bins][granule]: units per bin and granule (fused K1, or K2a)
    uint32_t *d_gran_off;           // [8 bins][granule]: K2b's exclusive scan of the above
    uint64_t *d_counts_rep;         // XM_COUNT_REPLICAS x 64 partial category_counts (all zero between calls), then 8 bin totals
    uint32_t *d_part_tot;           // [8][XM_PART_STRIDE]: per-part bin totals (K2b's first level)
    // the stream the workspace was last used on: a call on another stream is ordered behind it (order_workspace);
    // ws_released: that stream's owner has let go of it (xm_workspace_release) -- ws_event marks the end of its work
    hipStream_t ws_stream;
    bool ws_used, ws_released;
    hipEvent_t ws_event;
    // scratch of the host-buffer entry points (grown on demand, never inside *_dev calls)
    void *d_scratch[8];
    size_t scratch_bytes[8];
    // timing
    bool timing;
    uint32_t timing_mask;           // bit k: kernel k is bracketed with events while timing is on
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> free_events;
    double acc_ms[XM_K_COUNT];
    uint64_t acc_launches[XM_K_COUNT];
    std::string last_error;
    // RCCL communicator of the count all-reduce (xm_comm_init); null until then
    ncclComm_t comm;
    int comm_ranks;
};

namespace {

std::string g_create_error;      // why the last xm_ctx_create() failed (read with xm_last_hip_error(NULL))

int fail_hip(xm_ctx *ctx, hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    if (ctx) ctx->last_error = buf;
    else g_create_error = buf;
    (void)hipGetLastError();        // reported here: the next check_launch() on this thread must not find it again
    return (e == hipErrorOutOfMemory) ? XM_ERR_OOM : XM_ERR_HIP;
}

int no_device(const char *what, hipError_t e)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s%s%s", what, e == hipSuccess ? "" : ": ", e == hipSuccess ? "" : hipGetErrorString(e));
    g_create_error = buf;
    return XM_ERR_NO_DEVICE;
}

#define XM_HIP(ctx, call)                                        \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return fail_hip((ctx), e_, #call); \
    } while (0)

int ensure_scratch(xm_ctx *ctx, int slot, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (ctx->scratch_bytes[slot] >= bytes) return XM_OK;
    if (ctx->d_scratch[slot]) {
        XM_HIP(ctx, hipFree(ctx->d_scratch[slot]));
        ctx->d_scratch[slot] = nullptr;
        ctx->scratch_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    XM_HIP(ctx, hipMalloc(&ctx->d_scratch[slot], want));
    ctx->scratch_bytes[slot] = want;
    return XM_OK;
}

hipEvent_t take_event(xm_ctx *ctx)
{
    if (!ctx->free_events.empty()) {
        hipEvent_t e = ctx->free_events.back();
        ctx->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct Span {
    xm_ctx *ctx;
    hipStream_t st;
    TimedSpan sp;
    bool on;
    Span(xm_ctx *c, hipStream_t s, int kernel) : ctx(c), st(s), on(c->timing && ((c->timing_mask >> kernel) & 1u))
    {
        if (!on) return;
        sp.kernel = kernel;
        sp.start = take_event(ctx);
        sp.stop = take_event(ctx);
        if (!sp.start || !sp.stop) { on = false; return; }
        (void)hipEventRecord(sp.start, st);
    }
    ~Span()
    {
        if (!on) return;
        (void)hipEventRecord(sp.stop, st);
        ctx->spans.push_back(sp);
    }
};

int check_launch(xm_ctx *ctx, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(ctx, e, what);
    return XM_OK;
}

const size_t COUNTS_REP_BYTES = (XM_COUNT_REPLICAS * 64 + 8) * sizeof(uint64_t);
const size_t PART_TOT_BYTES = (size_t)XM_PART_REPLICAS * 8 * XM_PART_STRIDE * sizeof(uint32_t);

// a launch failed between counting and K2b: the count replicas may be non-zero -- clear them, so that the next call
// starts from zero again
void reset_count_state(xm_ctx *ctx, hipStream_t st);

bool bad_mode(int mode) { return mode < XM_MODE_SE || mode > XM_MODE_PE_CONSERVATIVE; }

// the *_dev entry points launch on the calling thread's current device, which must be the context's
bool wrong_device(const xm_ctx *ctx)
{
    int cur = -1;
    return hipGetDevice(&cur) != hipSuccess || cur != ctx->device;
}

void reset_count_state(xm_ctx *ctx, hipStream_t st)
{
    (void)hipMemsetAsync(ctx->d_counts_rep, 0, COUNTS_REP_BYTES, st);
    (void)hipMemsetAsync(ctx->d_part_tot, 0, PART_TOT_BYTES, st);
}

// The compaction workspace (per-granule counts and offsets, the count replicas, the part totals) belongs to the context:
// calls that use it must not overlap.  Calls on one stream are ordered anyway; a call on ANOTHER stream than the previous
// one is put behind everything enqueued on that stream so far (an event recorded there now, waited for here): no per-call
// cost while one stream is used (an event behind every call measured +5-7 us per 0.35 ms step, profiles/r05_ab_runs_first.txt).
// The previous stream's handle is therefore kept -- and must still be alive at the next switch: whoever destroys a stream
// that was used with this context calls xm_workspace_release first (xm_strip_destroy does, for its slot streams), which
// records the event while the stream still exists and forgets the handle.  A capturing stream is left alone: the graph's
// own edges order its nodes, and events cannot be mixed into a capture from outside.
bool capturing(hipStream_t st)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

int order_workspace(xm_ctx *ctx, hipStream_t st)
{
    if (capturing(st)) return XM_OK;
    if (ctx->ws_released) {                                   // the last user is gone; its event is what there is to wait for
        XM_HIP(ctx, hipStreamWaitEvent(st, ctx->ws_event, 0));
        ctx->ws_released = false;
    } else if (ctx->ws_used && ctx->ws_stream != st && !capturing(ctx->ws_stream)) {
        XM_HIP(ctx, hipEventRecord(ctx->ws_event, ctx->ws_stream));
        XM_HIP(ctx, hipStreamWaitEvent(st, ctx->ws_event, 0));
    }
    ctx->ws_stream = st;
    ctx->ws_used = true;
    return XM_OK;
}

}  // namespace

extern "C" {

int xm_abi_version(void) { return XM_ABI_VERSION; }

const char *xm_strerror(int status)
{
    switch (status) {
    case XM_OK: return "ok";
    case XM_ERR_INVALID_ARG: return "invalid argument";
    case XM_ERR_NO_DEVICE: return "no usable gfx950 (MI355X) device";
    case XM_ERR_HIP: return "HIP runtime error";
    case XM_ERR_OOM: return "out of device memory";
    case XM_ERR_RANGE: return "CIGAR-derived score outside int32";
    case XM_ERR_RCCL: return "RCCL error";
    default: return "unknown status";
    }
}

const char *xm_last_hip_error(const xm_ctx *ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

int xm_ctx_create(int device_id, xm_ctx **out)
{
    if (!out) return XM_ERR_INVALID_ARG;
    *out = nullptr;
    int n_dev = 0;
    hipError_t he = hipGetDeviceCount(&n_dev);
    if (he != hipSuccess || n_dev <= 0) return no_device("hipGetDeviceCount found no device", he);
    if (device_id < 0 || device_id >= n_dev) return XM_ERR_INVALID_ARG;
    hipDeviceProp_t prop;
    if ((he = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return no_device("hipGetDeviceProperties", he);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {                           // code objects are gfx950 only
        char buf[200];
        snprintf(buf, sizeof buf, "device %d is %s, not gfx950", device_id, prop.gcnArchName);
        return no_device(buf, hipSuccess);
    }
    if ((he = hipSetDevice(device_id)) != hipSuccess) return no_device("hipSetDevice", he);

    xm_ctx *ctx = new xm_ctx();
    ctx->device = device_id;
    ctx->n_cu = prop.multiProcessorCount;
    snprintf(ctx->name, sizeof ctx->name, "%s (%s)", prop.name, prop.gcnArchName);
    ctx->max_blocks = (uint32_t)ctx->n_cu * 8u;
    ctx->d_gran_counts = nullptr;
    ctx->d_gran_off = nullptr;
    ctx->d_counts_rep = nullptr;
    ctx->d_part_tot = nullptr;
    ctx->ws_stream = nullptr;
    ctx->ws_used = false;
    ctx->ws_released = false;
    ctx->ws_event = nullptr;
    for (int i = 0; i < 8; ++i) { ctx->d_scratch[i] = nullptr; ctx->scratch_bytes[i] = 0; }
    ctx->timing = false;
    ctx->comm = nullptr;
    ctx->comm_ranks = 0;
    ctx->timing_mask = ~0u;
    for (int k = 0; k < XM_K_COUNT; ++k) { ctx->acc_ms[k] = 0.0; ctx->acc_launches[k] = 0; }
    // [8 bins][granule pitch] for the largest input: 2 x 67 MB of the 288 GB
    const size_t ws = ((size_t)XM_MAX_GRANULES + 64) * 8 * sizeof(uint32_t);
    hipError_t e = hipMalloc((void **)&ctx->d_gran_counts, ws);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_gran_off, ws);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_counts_rep, COUNTS_REP_BYTES);
    if (e == hipSuccess) e = hipMemset(ctx->d_counts_rep, 0, COUNTS_REP_BYTES);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_part_tot, PART_TOT_BYTES);
    if (e == hipSuccess) e = hipMemset(ctx->d_part_tot, 0, PART_TOT_BYTES);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ws_event, hipEventDisableTiming);
    if (e != hipSuccess) {
        const int rc = fail_hip(nullptr, e, "xm_ctx_create: workspace");
        if (ctx->d_gran_counts) (void)hipFree(ctx->d_gran_counts);
        if (ctx->d_gran_off) (void)hipFree(ctx->d_gran_off);
        if (ctx->d_counts_rep) (void)hipFree(ctx->d_counts_rep);
        if (ctx->d_part_tot) (void)hipFree(ctx->d_part_tot);
        if (ctx->ws_event) (void)hipEventDestroy(ctx->ws_event);
        delete ctx;
        return rc;
    }
    *out = ctx;
    return XM_OK;
}

int xm_ctx_destroy(xm_ctx *ctx)
{
    if (!ctx) return XM_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->comm) (void)xm_comm_destroy(ctx);
    for (auto &sp : ctx->spans) { (void)hipEventDestroy(sp.start); (void)hipEventDestroy(sp.stop); }
    for (auto &e : ctx->free_events) (void)hipEventDestroy(e);
    for (int i = 0; i < 8; ++i)
        if (ctx->d_scratch[i]) (void)hipFree(ctx->d_scratch[i]);
    (void)hipFree(ctx->d_gran_counts);
    (void)hipFree(ctx->d_gran_off);
    (void)hipFree(ctx->d_counts_rep);
    (void)hipFree(ctx->d_part_tot);
    (void)hipEventDestroy(ctx->ws_event);
    delete ctx;
    return XM_OK;
}

int xm_ctx_device_info(const xm_ctx *ctx, int *n_cu, char *name, size_t name_len)
{
    if (!ctx) return XM_ERR_INVALID_ARG;
    if (n_cu) *n_cu = ctx->n_cu;
    if (name && name_len) snprintf(name, name_len, "%s", ctx->name);
    return XM_OK;
}

/* ---- packed CIGAR columns (host-side conversion; needs no device) -------------------------------------------- */

int xm_cigar_pack(uint64_t n, const uint32_t *cig_off, const uint32_t *cig_oplen, uint8_t *cig_cnt, uint32_t *cig_tile,
                  uint32_t *ops_packed, uint64_t ops_capacity, uint64_t *n_ops_packed)
{
    if (n > XM_MAX_RECORDS || !cig_tile || !n_ops_packed || (n && (!cig_off || !cig_cnt))) return XM_ERR_INVALID_ARG;
    if (ops_packed && n && cig_off[n] && !cig_oplen) return XM_ERR_INVALID_ARG;
    if (n && cig_off[0] != 0u) return XM_ERR_INVALID_ARG;      // "n_ops_packed == cig_off[n]: reuse cig_oplen as it is" relies on it
    uint64_t pos = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if ((i & (XM_CIG_TILE - 1u)) == 0) {
            if (pos > 0xFFFFFFFFull) return XM_ERR_RANGE;
            cig_tile[i / XM_CIG_TILE] = (uint32_t)pos;
        }
        if (cig_off[i + 1] < cig_off[i]) return XM_ERR_INVALID_ARG;
        const uint32_t k = cig_off[i + 1] - cig_off[i];
        const bool esc = k >= 255u;
        if (esc && k >= (1u << 28)) return XM_ERR_RANGE;
        cig_cnt[i] = esc ? (uint8_t)255 : (uint8_t)k;
        if (ops_packed) {
            if (pos + k + (esc ? 1u : 0u) > ops_capacity) return XM_ERR_INVALID_ARG;
            if (k) memcpy(ops_packed + pos, cig_oplen + cig_off[i], (size_t)k * sizeof(uint32_t));
            if (esc) ops_packed[pos + k] = (k << 4) | 15u;
        }
        pos += (uint64_t)k + (esc ? 1u : 0u);
    }
    if (pos > 0xFFFFFFFFull) return XM_ERR_RANGE;
    cig_tile[XM_CIG_TILES(n)] = (uint32_t)pos;
    *n_ops_packed = pos;
    return XM_OK;
}

/* ---- device-resident entry points ------------------------------------------------------ */

int xm_classify_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                    const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                    const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (n == 0) return XM_OK;
    if (!as1 || !xs1 || !as2 || !xs2 || !unit_bits || !code_out) return XM_ERR_INVALID_ARG;
    if ((((uintptr_t)as1 | (uintptr_t)xs1 | (uintptr_t)as2 | (uintptr_t)xs2) & 15u) || ((uintptr_t)code_out & 3u))
        return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    {
        Span span(ctx, st, XM_K_CLASSIFY);
        xm::launch_classify_i32(st, mode, n, as1, xs1, as2, xs2, unit_bits, min_score_floor, code_out, nullptr);
    }
    return check_launch(ctx, "classify_kernel<int32>");
}

int xm_classify_f64_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                        const double *as1, const double *xs1, const double *as2, const double *xs2,
                        const uint64_t *unit_bits, double min_score, uint8_t *code_out)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (n == 0) return XM_OK;
    if (!as1 || !xs1 || !as2 || !xs2 || !unit_bits || !code_out) return XM_ERR_INVALID_ARG;
    if ((((uintptr_t)as1 | (uintptr_t)xs1 | (uintptr_t)as2 | (uintptr_t)xs2) & 31u) || ((uintptr_t)code_out & 3u))
        return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    {
        Span span(ctx, st, XM_K_CLASSIFY);
        xm::launch_classify_f64(st, mode, n, as1, xs1, as2, xs2, unit_bits, min_score, code_out, nullptr);
    }
    return check_launch(ctx, "classify_kernel<f64>");
}

int xm_classify_cigar_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                          const int32_t *nm1, const uint32_t *off1, const uint32_t *ops1, const int32_t *xs1,
                          const int32_t *nm2, const uint32_t *off2, const uint32_t *ops2, const int32_t *xs2,
                          const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint32_t *range_flag)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (n == 0) return XM_OK;
    if (!nm1 || !off1 || !ops1 || !xs1 || !nm2 || !off2 || !ops2 || !xs2 || !unit_bits || !code_out)
        return XM_ERR_INVALID_ARG;
    if ((((uintptr_t)nm1 | (uintptr_t)off1 | (uintptr_t)xs1 | (uintptr_t)nm2 | (uintptr_t)off2 | (uintptr_t)xs2) & 15u) ||
        ((uintptr_t)code_out & 3u))
        return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    {
        Span span(ctx, st, XM_K_CLASSIFY);
        xm::launch_classify_cigar(st, mode, n, nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits, min_score_floor,
                                  code_out, range_flag);
    }
    return check_launch(ctx, "classify_cigar_kernel");
}

int xm_cigar_scores_dev(xm_ctx *ctx, void *stream, uint64_t n, const int32_t *nm,
                        const uint32_t *cig_off, const uint32_t *cig_oplen, int32_t *as_out,
                        uint32_t *range_flag)
{
    if (!ctx || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (n == 0) return XM_OK;
    if (!nm || !cig_off || !cig_oplen || !as_out) return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    {
        Span span(ctx, st, XM_K_CIGAR);
        xm::launch_cigar(st, ctx->max_blocks, n, nm, cig_off, cig_oplen, as_out, range_flag);
    }
    return check_launch(ctx, "cigar_kernel");
}

/* K2b + K2c after the counting side (fused K1 or K2a) has filled gran_counts / counts_rep.  When a launch fails
 * with the replicas possibly non-zero, they are cleared again so that the next call starts from zero. */
static int compact_tail(xm_ctx *ctx, hipStream_t st, int mode, uint64_t n, const uint8_t *code, const xm::CountPlan &cp,
                        uint32_t *idx_out, uint64_t *bin_offsets, uint64_t *counts, const xm::ListOut *lists = nullptr)
{
    const bool from_bins4 = cp.bins4 != nullptr;          // a counting classify kernel left the compact stream: K2c reads that
    uint64_t *bin_totals = ctx->d_counts_rep + XM_COUNT_REPLICAS * 64;
    int rc;
    {
        Span span(ctx, st, XM_K_SCAN);
        xm::launch_scan(st, cp, ctx->d_gran_off, bin_totals, counts);
    }
    if ((rc = check_launch(ctx, "scan_kernel")) != XM_OK) {
        reset_count_state(ctx, st);
        return rc;
    }
    {
        Span span(ctx, st, XM_K_SCATTER);
        xm::launch_scatter(st, cp.plan, mode, n, from_bins4 ? cp.bins4 : code, from_bins4, cp.gran_counts, ctx->d_gran_off,
                           bin_totals, bin_offsets, idx_out, ctx->d_part_tot, lists, cp.gran0, cp.gran1);
    }
    if ((rc = check_launch(ctx, "scatter_kernel")) != XM_OK) reset_count_state(ctx, st);      // K2c zeroes the part totals
    return rc;
}

static xm::CountPlan count_plan(xm_ctx *ctx, uint64_t n)
{
    xm::CountPlan cp;
    cp.plan = xm::plan_granules(n);
    cp.gran_counts = ctx->d_gran_counts;
    cp.counts_rep = ctx->d_counts_rep;
    cp.part_tot = ctx->d_part_tot;
    cp.bins4 = nullptr;
    return cp;
}

static int empty_compact(xm_ctx *ctx, hipStream_t st, uint64_t *bin_offsets, uint64_t *counts)
{
    XM_HIP(ctx, hipMemsetAsync(counts, 0, 64 * sizeof(uint64_t), st));
    XM_HIP(ctx, hipMemsetAsync(bin_offsets, 0, 8 * sizeof(uint64_t), st));
    return XM_OK;
}

int xm_compact_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n, const uint8_t *code,
                   uint32_t *idx_out, uint64_t *bin_offsets, uint64_t *counts)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (!bin_offsets || !counts) return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return empty_compact(ctx, st, bin_offsets, counts);
    // d_counts_rep is all zero here: zeroed at context creation and by every K2b after it has summed it
    if (!code || !idx_out || ((uintptr_t)code & 15u)) return XM_ERR_INVALID_ARG;
    const xm::CountPlan cp = count_plan(ctx, n);
    int rc;
    if ((rc = order_workspace(ctx, st)) != XM_OK) return rc;
    {
        Span span(ctx, st, XM_K_HIST);
        xm::launch_hist(st, mode, n, code, cp);
    }
    if ((rc = check_launch(ctx, "hist_kernel")) != XM_OK) return rc;
    return compact_tail(ctx, st, mode, n, code, cp, idx_out, bin_offsets, counts);
}

/* ---- fused: classify + count in one kernel, then scan + scatter ---------------------------------------------- */

int xm_classify_compact_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                            const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                            const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint8_t *bins4,
                            uint32_t *idx_out, uint64_t *bin_offsets, uint64_t *counts)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (!bin_offsets || !counts) return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return empty_compact(ctx, st, bin_offsets, counts);
    if (!as1 || !xs1 || !as2 || !xs2 || !unit_bits || (!code_out && !bins4) || !idx_out) return XM_ERR_INVALID_ARG;
    if ((((uintptr_t)as1 | (uintptr_t)xs1 | (uintptr_t)as2 | (uintptr_t)xs2 | (uintptr_t)code_out | (uintptr_t)bins4) & 15u))
        return XM_ERR_INVALID_ARG;
    xm::CountPlan cp = count_plan(ctx, n);
    cp.bins4 = bins4;
    int rc;
    if ((rc = order_workspace(ctx, st)) != XM_OK) return rc;
    {
        Span span(ctx, st, XM_K_CLASSIFY);
        xm::launch_classify_i32(st, mode, n, as1, xs1, as2, xs2, unit_bits, min_score_floor, code_out, &cp);
    }
    if ((rc = check_launch(ctx, "classify_kernel<int32, counts>")) != XM_OK) return rc;
    return compact_tail(ctx, st, mode, n, code_out, cp, idx_out, bin_offsets, counts);
}

int xm_classify_compact_f64_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                                const double *as1, const double *xs1, const double *as2, const double *xs2,
                                const uint64_t *unit_bits, double min_score, uint8_t *code_out, uint8_t *bins4,
                                uint32_t *idx_out, uint64_t *bin_offsets, uint64_t *counts)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (!bin_offsets || !counts) return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return empty_compact(ctx, st, bin_offsets, counts);
    if (!as1 || !xs1 || !as2 || !xs2 || !unit_bits || (!code_out && !bins4) || !idx_out) return XM_ERR_INVALID_ARG;
    if ((((uintptr_t)as1 | (uintptr_t)xs1 | (uintptr_t)as2 | (uintptr_t)xs2) & 31u) || (((uintptr_t)code_out | (uintptr_t)bins4) & 15u))
        return XM_ERR_INVALID_ARG;
    xm::CountPlan cp = count_plan(ctx, n);
    cp.bins4 = bins4;
    int rc;
    if ((rc = order_workspace(ctx, st)) != XM_OK) return rc;
    {
        Span span(ctx, st, XM_K_CLASSIFY);
        xm::launch_classify_f64(st, mode, n, as1, xs1, as2, xs2, unit_bits, min_score, code_out, &cp);
    }
    if ((rc = check_launch(ctx, "classify_kernel<f64, counts>")) != XM_OK) return rc;
    return compact_tail(ctx, st, mode, n, code_out, cp, idx_out, bin_offsets, counts);
}

int xm_classify_compact_cigar_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                                  const int32_t *nm1, const uint32_t *off1, const uint32_t *ops1, const int32_t *xs1,
                                  const int32_t *nm2, const uint32_t *off2, const uint32_t *ops2, const int32_t *xs2,
                                  const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out,
                                  uint32_t *range_flag, uint32_t *idx_out, uint64_t *bin_offsets, uint64_t *counts)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (!bin_offsets || !counts) return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return empty_compact(ctx, st, bin_offsets, counts);
    if (!nm1 || !off1 || !ops1 || !xs1 || !nm2 || !off2 || !ops2 || !xs2 || !unit_bits || !code_out || !idx_out)
        return XM_ERR_INVALID_ARG;
    if ((((uintptr_t)nm1 | (uintptr_t)off1 | (uintptr_t)xs1 | (uintptr_t)nm2 | (uintptr_t)off2 | (uintptr_t)xs2 |
          (uintptr_t)code_out) & 15u))
        return XM_ERR_INVALID_ARG;
    // K1c (CSR columns) does not count (it measured slower with the counting epilogue than with a separate histogram
    // pass): classify, then the stand-alone compaction on the category bytes it wrote.  The counting form of this path
    // is xm_classify_compact_cigar_packed_dev.
    int rc;
    {
        Span span(ctx, st, XM_K_CLASSIFY);
        xm::launch_classify_cigar(st, mode, n, nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits, min_score_floor,
                                  code_out, range_flag);
    }
    if ((rc = check_launch(ctx, "classify_cigar_kernel")) != XM_OK) return rc;
    return xm_compact_dev(ctx, stream, mode, n, code_out, idx_out, bin_offsets, counts);
}

int xm_classify_compact_cigar_packed_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                                         const int32_t *nm1, const uint8_t *cnt1, const uint32_t *tile1, const uint32_t *ops1,
                                         const int32_t *xs1,
                                         const int32_t *nm2, const uint8_t *cnt2, const uint32_t *tile2, const uint32_t *ops2,
                                         const int32_t *xs2,
                                         const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint8_t *bins4,
                                         uint32_t *range_flag, uint32_t *idx_out, uint64_t *bin_offsets, uint64_t *counts)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (!bin_offsets || !counts) return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return empty_compact(ctx, st, bin_offsets, counts);
    if (!nm1 || !cnt1 || !tile1 || !ops1 || !xs1 || !nm2 || !cnt2 || !tile2 || !ops2 || !xs2 || !unit_bits ||
        (!code_out && !bins4) || !idx_out)
        return XM_ERR_INVALID_ARG;
    if ((((uintptr_t)nm1 | (uintptr_t)xs1 | (uintptr_t)nm2 | (uintptr_t)xs2 | (uintptr_t)code_out | (uintptr_t)bins4) & 15u) ||
        (((uintptr_t)cnt1 | (uintptr_t)cnt2 | (uintptr_t)tile1 | (uintptr_t)tile2 | (uintptr_t)ops1 | (uintptr_t)ops2) & 3u))
        return XM_ERR_INVALID_ARG;
    xm::CountPlan cp = count_plan(ctx, n);
    cp.bins4 = bins4;
    const xm::CigCols s1 = {nm1, xs1, cnt1, tile1, ops1}, s2 = {nm2, xs2, cnt2, tile2, ops2};
    int rc;
    if ((rc = order_workspace(ctx, st)) != XM_OK) return rc;
    {
        Span span(ctx, st, XM_K_CLASSIFY);
        xm::launch_classify_cigp(st, mode, n, s1, s2, unit_bits, min_score_floor, code_out, range_flag, cp);
    }
    if ((rc = check_launch(ctx, "classify_cigp_kernel")) != XM_OK) return rc;
    return compact_tail(ctx, st, mode, n, code_out, cp, idx_out, bin_offsets, counts);
}

/* ---- the six-list output contract: the same three launches, the scatter writes one list per bin -------- */

static int list_out(uint32_t *const idx_out[6], uint32_t *idx_state6, uint64_t list_capacity, xm::ListOut &lo)
{
    if (!idx_out) return XM_ERR_INVALID_ARG;
    for (int b = 0; b < 6; ++b) {
        if (!idx_out[b] || ((uintptr_t)idx_out[b] & 3u)) return XM_ERR_INVALID_ARG;
        lo.p[b] = idx_out[b];
    }
    if ((uintptr_t)idx_state6 & 3u) return XM_ERR_INVALID_ARG;
    lo.p[6] = idx_state6;
    lo.cap = list_capacity > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)list_capacity;
    return XM_OK;
}

/* Large inputs in chunks -- built, measured, and left OFF.  The six lists need, for every granule, the units of its bin in
 * FRONT of it (running totals, not the bin totals), so K1, K2b and K2c can go over the input a chunk at a time: K1(c) leaves
 * 1/2 byte per record of bins4 and the per-granule counts, K2b(c) takes its carry from the part totals of the chunks in
 * front (left in place until the last chunk's K2c), K2c(c) reads what K1(c) wrote while it may still be in the 256 MB
 * Infinity Cache; the last chunk's K2b adds up category_counts, its K2c publishes the list lengths.  The idea was that at
 * 800 M records K2c (0.64 ms, against 8 x 0.06-0.07 at 100 M) loses by fetching 400 MB of bins4 from HBM again.  Measured
 * on one box, same process, 400 M read pairs in one block (tools/ab_place_chunks.py, profiles/r05_ab_place_chunks.txt):
 *     one pass 3.059 ms | chunks of 16 parts 3.413 | 32 parts 3.147 | 64 parts 3.120 | 128 parts 3.053
 * K2c gains 0.04 ms at 32-64 parts (0.643 -> 0.600), K1 loses 0.08 (2.365 -> 2.44: every chunk launch has its own ramp and
 * tail) and K2b 0.03-0.07 (every chunk's workgroups walk the part totals in front).  So the one-pass order stays the
 * default; XM_PLACE_CHUNK_PARTS (environment, read per call) = parts of XM_PART_GRAN granules per chunk switches the
 * chunked order on (inputs of more than two chunks), for A/B runs and tests/test_gpu_parity.py. */
static uint32_t place_chunk_granules(void)
{
    const char *e = getenv("XM_PLACE_CHUNK_PARTS");
    const long v = e && *e ? strtol(e, nullptr, 10) : 0;                   // off unless asked for (32 parts = 2^26 records = 32 MB of bins4)
    return (uint32_t)(v < 0 ? 0 : v > 2048 ? 2048 : v) * (uint32_t)XM_PART_GRAN;
}

extern "C++" {
template <typename LaunchK1>
static int place_steps(xm_ctx *ctx, hipStream_t st, int mode, uint64_t n, xm::CountPlan &cp, uint8_t *code_out,
                       const xm::ListOut &lo, uint64_t *n_out, uint64_t *counts, const char *k1_name, LaunchK1 k1)
{
    const uint32_t chunk = place_chunk_granules(), n_gran = cp.plan.n_gran;
    const bool chunked = chunk != 0u && n_gran > 2u * chunk;
    int rc;
    for (uint32_t g0 = 0; g0 < n_gran; g0 += chunked ? chunk : n_gran) {
        if (chunked) {
            cp.gran0 = g0;
            cp.gran1 = n_gran - g0 > chunk ? g0 + chunk : n_gran;
        }
        {
            Span span(ctx, st, XM_K_CLASSIFY);
            k1(cp);
        }
        if ((rc = check_launch(ctx, k1_name)) != XM_OK) {
            if (g0) reset_count_state(ctx, st);                             // the chunks in front left their totals behind
            return rc;
        }
        if ((rc = compact_tail(ctx, st, mode, n, code_out, cp, nullptr, n_out, counts, &lo)) != XM_OK) return rc;
    }
    return XM_OK;
}
}  // extern "C++"

int xm_classify_place_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                          const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                          const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint8_t *bins4,
                          uint32_t *const idx_out[6], uint64_t list_capacity, uint64_t *n_out, uint64_t *counts)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (!n_out || !counts) return XM_ERR_INVALID_ARG;
    xm::ListOut lo;
    int rc;
    if ((rc = list_out(idx_out, nullptr, list_capacity, lo)) != XM_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return empty_compact(ctx, st, n_out, counts);
    if (!as1 || !xs1 || !as2 || !xs2 || !unit_bits || (!code_out && !bins4)) return XM_ERR_INVALID_ARG;
    if ((((uintptr_t)as1 | (uintptr_t)xs1 | (uintptr_t)as2 | (uintptr_t)xs2 | (uintptr_t)code_out | (uintptr_t)bins4) & 15u))
        return XM_ERR_INVALID_ARG;
    xm::CountPlan cp = count_plan(ctx, n);
    cp.bins4 = bins4;
    if ((rc = order_workspace(ctx, st)) != XM_OK) return rc;
    return place_steps(ctx, st, mode, n, cp, code_out, lo, n_out, counts, "classify_kernel<int32, counts>", [&](const xm::CountPlan &c) {
        xm::launch_classify_i32(st, mode, n, as1, xs1, as2, xs2, unit_bits, min_score_floor, code_out, &c);
    });
}

int xm_classify_place_f64_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                              const double *as1, const double *xs1, const double *as2, const double *xs2,
                              const uint64_t *unit_bits, double min_score, uint8_t *code_out, uint8_t *bins4,
                              uint32_t *const idx_out[6], uint32_t *idx_state6, uint64_t list_capacity,
                              uint64_t *n_out, uint64_t *counts)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (!n_out || !counts) return XM_ERR_INVALID_ARG;
    xm::ListOut lo;
    int rc;
    if ((rc = list_out(idx_out, idx_state6, list_capacity, lo)) != XM_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return empty_compact(ctx, st, n_out, counts);
    if (!as1 || !xs1 || !as2 || !xs2 || !unit_bits || (!code_out && !bins4)) return XM_ERR_INVALID_ARG;
    if ((((uintptr_t)as1 | (uintptr_t)xs1 | (uintptr_t)as2 | (uintptr_t)xs2) & 31u) || (((uintptr_t)code_out | (uintptr_t)bins4) & 15u))
        return XM_ERR_INVALID_ARG;
    xm::CountPlan cp = count_plan(ctx, n);
    cp.bins4 = bins4;
    if ((rc = order_workspace(ctx, st)) != XM_OK) return rc;
    return place_steps(ctx, st, mode, n, cp, code_out, lo, n_out, counts, "classify_kernel<f64, counts>", [&](const xm::CountPlan &c) {
        xm::launch_classify_f64(st, mode, n, as1, xs1, as2, xs2, unit_bits, min_score, code_out, &c);
    });
}

int xm_classify_place_cigar_packed_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                                       const int32_t *nm1, const uint8_t *cnt1, const uint32_t *tile1, const uint32_t *ops1,
                                       const int32_t *xs1,
                                       const int32_t *nm2, const uint8_t *cnt2, const uint32_t *tile2, const uint32_t *ops2,
                                       const int32_t *xs2,
                                       const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint8_t *bins4,
                                       uint32_t *range_flag, uint32_t *const idx_out[6], uint64_t list_capacity,
                                       uint64_t *n_out, uint64_t *counts)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (!n_out || !counts) return XM_ERR_INVALID_ARG;
    xm::ListOut lo;
    int rc;
    if ((rc = list_out(idx_out, nullptr, list_capacity, lo)) != XM_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return empty_compact(ctx, st, n_out, counts);
    if (!nm1 || !cnt1 || !tile1 || !ops1 || !xs1 || !nm2 || !cnt2 || !tile2 || !ops2 || !xs2 || !unit_bits || (!code_out && !bins4))
        return XM_ERR_INVALID_ARG;
    if ((((uintptr_t)nm1 | (uintptr_t)xs1 | (uintptr_t)nm2 | (uintptr_t)xs2 | (uintptr_t)code_out | (uintptr_t)bins4) & 15u) ||
        (((uintptr_t)cnt1 | (uintptr_t)cnt2 | (uintptr_t)tile1 | (uintptr_t)tile2 | (uintptr_t)ops1 | (uintptr_t)ops2) & 3u))
        return XM_ERR_INVALID_ARG;
    xm::CountPlan cp = count_plan(ctx, n);
    cp.bins4 = bins4;
    const xm::CigCols s1 = {nm1, xs1, cnt1, tile1, ops1}, s2 = {nm2, xs2, cnt2, tile2, ops2};
    if ((rc = order_workspace(ctx, st)) != XM_OK) return rc;
    {
        Span span(ctx, st, XM_K_CLASSIFY);
        xm::launch_classify_cigp(st, mode, n, s1, s2, unit_bits, min_score_floor, code_out, range_flag, cp);
    }
    if ((rc = check_launch(ctx, "classify_cigp_kernel")) != XM_OK) return rc;
    return compact_tail(ctx, st, mode, n, code_out, cp, nullptr, n_out, counts, &lo);
}

/* ---- segmented bin lists: one launch, no scan, no scatter ------------------------------------------------------- */

static int runs_common(xm_ctx *ctx, hipStream_t st, int mode, uint64_t n, size_t elem,
                       const void *as1, const void *xs1, const void *as2, const void *xs2, const uint64_t *unit_bits,
                       int32_t mi, double mf, uint16_t *runs16, uint16_t *gran_counts, uint64_t *n_out, uint64_t *counts)
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (!n_out || !counts) return XM_ERR_INVALID_ARG;
    if (n == 0) return empty_compact(ctx, st, n_out, counts);
    if (!as1 || !xs1 || !as2 || !xs2 || !unit_bits || !runs16 || !gran_counts) return XM_ERR_INVALID_ARG;
    const uintptr_t col_mask = elem == 8 ? 31u : 15u;
    if ((((uintptr_t)as1 | (uintptr_t)xs1 | (uintptr_t)as2 | (uintptr_t)xs2) & col_mask) ||
        (((uintptr_t)runs16 | (uintptr_t)gran_counts) & 15u))
        return XM_ERR_INVALID_ARG;
    const xm::RunsOut ro = {runs16, gran_counts, ctx->d_counts_rep};
    int rc;
    if ((rc = order_workspace(ctx, st)) != XM_OK) return rc;
    {
        Span span(ctx, st, XM_K_CLASSIFY);
        if (elem == 4)
            xm::launch_classify_runs_i32(st, mode, n, (const int32_t *)as1, (const int32_t *)xs1, (const int32_t *)as2,
                                         (const int32_t *)xs2, unit_bits, mi, ro);
        else
            xm::launch_classify_runs_f64(st, mode, n, (const double *)as1, (const double *)xs1, (const double *)as2,
                                         (const double *)xs2, unit_bits, mf, ro);
    }
    if ((rc = check_launch(ctx, "classify_runs_kernel")) != XM_OK) return rc;
    {
        Span span(ctx, st, XM_K_SCAN);
        xm::launch_runs_finish(st, mode, ctx->d_counts_rep, counts, n_out);
    }
    if ((rc = check_launch(ctx, "runs_finish_kernel")) != XM_OK) reset_count_state(ctx, st);
    return rc;
}

int xm_classify_runs_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                         const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                         const uint64_t *unit_bits, int32_t min_score_floor,
                         uint16_t *runs16, uint16_t *gran_counts, uint64_t *n_out, uint64_t *counts)
{
    return runs_common(ctx, (hipStream_t)stream, mode, n, 4, as1, xs1, as2, xs2, unit_bits, min_score_floor, 0.0,
                       runs16, gran_counts, n_out, counts);
}

int xm_classify_runs_f64_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n,
                             const double *as1, const double *xs1, const double *as2, const double *xs2,
                             const uint64_t *unit_bits, double min_score,
                             uint16_t *runs16, uint16_t *gran_counts, uint64_t *n_out, uint64_t *counts)
{
    return runs_common(ctx, (hipStream_t)stream, mode, n, 8, as1, xs1, as2, xs2, unit_bits, 0, min_score,
                       runs16, gran_counts, n_out, counts);
}

/* Host side of the contract (no device needed): list `bin` of the segmented form as a flat list. */
int xm_runs_expand(uint64_t n, const uint16_t *runs16, const uint16_t *gran_counts, int bin,
                   uint32_t *idx_out, uint64_t capacity, uint64_t *n_written)
{
    if (n > XM_MAX_RECORDS || bin < 0 || bin > 6 || !n_written || (n && (!runs16 || !gran_counts))) return XM_ERR_INVALID_ARG;
    const uint64_t n_gran = (n + XM_GRAN - 1) / XM_GRAN;
    uint64_t w = 0;
    for (uint64_t g = 0; g < n_gran; ++g) {
        const uint16_t *c = gran_counts + g * 8;
        uint32_t at = 0;
        for (int b = 0; b < bin; ++b) at += c[b];
        const uint32_t k = c[bin];
        if (at + k > XM_GRAN) return XM_ERR_INVALID_ARG;                  // not counts the kernel wrote
        const uint16_t *run = runs16 + g * XM_GRAN + at;
        for (uint32_t i = 0; i < k; ++i, ++w)
            if (idx_out && w < capacity) idx_out[w] = (uint32_t)(g * XM_GRAN) + run[i];
    }
    *n_written = w;
    return XM_OK;
}

int xm_workspace_release(xm_ctx *ctx, void *stream)
{
    if (!ctx) return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!ctx->ws_used || ctx->ws_released || ctx->ws_stream != st) return XM_OK;       // not the stream the workspace is waiting on
    XM_HIP(ctx, hipSetDevice(ctx->device));
    if (!capturing(st)) XM_HIP(ctx, hipEventRecord(ctx->ws_event, st));
    ctx->ws_released = true;
    ctx->ws_stream = nullptr;
    return XM_OK;
}

int xm_workspace_is_clean(xm_ctx *ctx, int *clean)
{
    if (!ctx || !clean) return XM_ERR_INVALID_ARG;
    XM_HIP(ctx, hipSetDevice(ctx->device));
    XM_HIP(ctx, hipDeviceSynchronize());
    std::vector<uint32_t> part(PART_TOT_BYTES / sizeof(uint32_t));
    std::vector<uint64_t> rep(XM_COUNT_REPLICAS * 64);
    XM_HIP(ctx, hipMemcpy(part.data(), ctx->d_part_tot, PART_TOT_BYTES, hipMemcpyDeviceToHost));
    XM_HIP(ctx, hipMemcpy(rep.data(), ctx->d_counts_rep, rep.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    uint64_t any = 0;
    for (uint32_t v : part) any |= v;
    for (uint64_t v : rep) any |= v;
    *clean = any == 0 ? 1 : 0;
    return XM_OK;
}

int xm_stream_probe_dev(xm_ctx *ctx, void *stream, uint64_t n,
                        const int32_t *c0, const int32_t *c1, const int32_t *c2, const int32_t *c3, uint8_t *out)
{
    if (!ctx || n > XM_MAX_RECORDS || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    if (n == 0) return XM_OK;
    if (!c0 || !c1 || !c2 || !c3 || !out) return XM_ERR_INVALID_ARG;
    if ((((uintptr_t)c0 | (uintptr_t)c1 | (uintptr_t)c2 | (uintptr_t)c3) & 15u) || ((uintptr_t)out & 1u)) return XM_ERR_INVALID_ARG;
    xm::launch_stream_probe((hipStream_t)stream, n, c0, c1, c2, c3, out);
    return check_launch(ctx, "stream_probe_kernel");
}

int xm_mate_correlate_dev(xm_ctx *ctx, void *stream, uint64_t n, const double *track, uint64_t m,
                          const double *density, double *out)
{
    if (!ctx || wrong_device(ctx) || m > 0xFFFFFFFFull || n > (1ull << 40)) return XM_ERR_INVALID_ARG;
    if (n == 0) return XM_OK;
    if (!track || !out || (m && !density)) return XM_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    {
        Span span(ctx, st, XM_K_CORRELATE);
        xm::launch_mate_correlate(st, n, track, (uint32_t)m, density, out);
    }
    return check_launch(ctx, "mate_correlate_kernel");
}

/* ---- host-buffer entry points ------------------------------------------------------------ */

static int fetch_compact_results(xm_ctx *ctx, uint64_t n, const uint8_t *d_code, const uint32_t *d_idx, const uint64_t *d_off,
                                 uint8_t *code_out, uint32_t *idx_out, uint64_t bin_offsets[8], uint64_t counts[64]);

// category_counts of a counting classify launch without a compaction: K2b alone adds the replicas up (and zeroes them)
static int counts_only_tail(xm_ctx *ctx, const xm::CountPlan &cp, uint64_t *d_counts)
{
    xm::launch_scan(nullptr, cp, ctx->d_gran_off, ctx->d_counts_rep + XM_COUNT_REPLICAS * 64, d_counts);
    const int rc = check_launch(ctx, "scan_kernel");
    if (rc != XM_OK) reset_count_state(ctx, nullptr);
    else (void)hipMemsetAsync(ctx->d_part_tot, 0, PART_TOT_BYTES, nullptr);         // no K2c here to zero the part totals
    return rc;
}

static int classify_host(xm_ctx *ctx, int mode, uint64_t n, size_t elem, const void *as1, const void *xs1,
                         const void *as2, const void *xs2, const uint64_t *unit_bits, int32_t mi, double mf,
                         uint8_t *code_out, uint64_t counts[64])
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS) return XM_ERR_INVALID_ARG;
    if (counts) memset(counts, 0, 64 * sizeof(uint64_t));
    if (n == 0) return XM_OK;
    if (!as1 || !xs1 || !as2 || !xs2 || !unit_bits || !code_out) return XM_ERR_INVALID_ARG;
    XM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t col_bytes = (size_t)n * elem;
    const size_t bits_bytes = (size_t)((n + 63) / 64) * 8;
    const void *src[4] = {as1, xs1, as2, xs2};
    int rc;
    for (int c = 0; c < 4; ++c) {
        if ((rc = ensure_scratch(ctx, c, col_bytes)) != XM_OK) return rc;
        XM_HIP(ctx, hipMemcpy(ctx->d_scratch[c], src[c], col_bytes, hipMemcpyHostToDevice));
    }
    if ((rc = ensure_scratch(ctx, 4, bits_bytes)) != XM_OK) return rc;
    XM_HIP(ctx, hipMemcpy(ctx->d_scratch[4], unit_bits, bits_bytes, hipMemcpyHostToDevice));
    if ((rc = ensure_scratch(ctx, 5, (size_t)n + 16)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 7, 72 * sizeof(uint64_t))) != XM_OK) return rc;
    uint8_t *d_code = (uint8_t *)ctx->d_scratch[5];
    uint64_t *d_counts = (uint64_t *)ctx->d_scratch[7] + 8;
    // category_counts come from the counting form of the kernel (+ K2b, which adds the replicas up)
    const xm::CountPlan cp = count_plan(ctx, n);
    if (counts && (rc = order_workspace(ctx, nullptr)) != XM_OK) return rc;
    {
        Span span(ctx, nullptr, XM_K_CLASSIFY);
        if (elem == 4)
            xm::launch_classify_i32(nullptr, mode, n, (const int32_t *)ctx->d_scratch[0], (const int32_t *)ctx->d_scratch[1],
                                    (const int32_t *)ctx->d_scratch[2], (const int32_t *)ctx->d_scratch[3],
                                    (const uint64_t *)ctx->d_scratch[4], mi, d_code, counts ? &cp : nullptr);
        else
            xm::launch_classify_f64(nullptr, mode, n, (const double *)ctx->d_scratch[0], (const double *)ctx->d_scratch[1],
                                    (const double *)ctx->d_scratch[2], (const double *)ctx->d_scratch[3],
                                    (const uint64_t *)ctx->d_scratch[4], mf, d_code, counts ? &cp : nullptr);
    }
    if ((rc = check_launch(ctx, "classify_kernel")) != XM_OK) return rc;
    if (counts) {
        if ((rc = counts_only_tail(ctx, cp, d_counts)) != XM_OK) return rc;
        XM_HIP(ctx, hipMemcpy(counts, d_counts, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    XM_HIP(ctx, hipMemcpy(code_out, d_code, (size_t)n, hipMemcpyDeviceToHost));
    return XM_OK;
}

int xm_classify(xm_ctx *ctx, int mode, uint64_t n,
                const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint64_t counts[64])
{
    return classify_host(ctx, mode, n, sizeof(int32_t), as1, xs1, as2, xs2, unit_bits, min_score_floor, 0.0,
                         code_out, counts);
}

int xm_classify_f64(xm_ctx *ctx, int mode, uint64_t n,
                    const double *as1, const double *xs1, const double *as2, const double *xs2,
                    const uint64_t *unit_bits, double min_score, uint8_t *code_out, uint64_t counts[64])
{
    return classify_host(ctx, mode, n, sizeof(double), as1, xs1, as2, xs2, unit_bits, 0, min_score, code_out, counts);
}

int xm_cigar_scores(xm_ctx *ctx, uint64_t n, const int32_t *nm, const uint32_t *cig_off,
                    const uint32_t *cig_oplen, int32_t *as_out)
{
    if (!ctx || n > XM_MAX_RECORDS) return XM_ERR_INVALID_ARG;
    if (n == 0) return XM_OK;
    if (!nm || !cig_off || !as_out) return XM_ERR_INVALID_ARG;
    XM_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n_ops = cig_off[n];
    if (n_ops && !cig_oplen) return XM_ERR_INVALID_ARG;
    int rc;
    if ((rc = ensure_scratch(ctx, 0, n * 4)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 1, (n + 1) * 4)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 2, n_ops * 4)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 3, n * 4)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 6, 16)) != XM_OK) return rc;
    XM_HIP(ctx, hipMemcpy(ctx->d_scratch[0], nm, n * 4, hipMemcpyHostToDevice));
    XM_HIP(ctx, hipMemcpy(ctx->d_scratch[1], cig_off, (n + 1) * 4, hipMemcpyHostToDevice));
    if (n_ops) XM_HIP(ctx, hipMemcpy(ctx->d_scratch[2], cig_oplen, n_ops * 4, hipMemcpyHostToDevice));
    XM_HIP(ctx, hipMemset(ctx->d_scratch[6], 0, 16));
    rc = xm_cigar_scores_dev(ctx, nullptr, n, (const int32_t *)ctx->d_scratch[0], (const uint32_t *)ctx->d_scratch[1],
                             (const uint32_t *)ctx->d_scratch[2], (int32_t *)ctx->d_scratch[3],
                             (uint32_t *)ctx->d_scratch[6]);
    if (rc != XM_OK) return rc;
    uint32_t flag = 0;
    XM_HIP(ctx, hipMemcpy(as_out, ctx->d_scratch[3], n * 4, hipMemcpyDeviceToHost));
    XM_HIP(ctx, hipMemcpy(&flag, ctx->d_scratch[6], 4, hipMemcpyDeviceToHost));
    return flag ? XM_ERR_RANGE : XM_OK;
}

// CSR columns of one species, packed on the host (no copy of the ops unless a record has 255 ops or more)
struct PackedSpecies {
    std::vector<uint8_t> cnt;
    std::vector<uint32_t> tile, ops_buf;
    const uint32_t *ops;
    uint64_t n_ops;
};

static int pack_species(uint64_t n, const uint32_t *off, const uint32_t *oplen, PackedSpecies &p)
{
    p.cnt.resize(n);
    p.tile.resize(XM_CIG_TILES(n) + 1);
    int rc = xm_cigar_pack(n, off, oplen, p.cnt.data(), p.tile.data(), nullptr, 0, &p.n_ops);
    if (rc != XM_OK) return rc;
    p.ops = oplen;
    if (p.n_ops != off[n]) {                      // escaped records: the op array gets their trailer words
        p.ops_buf.resize(p.n_ops);
        rc = xm_cigar_pack(n, off, oplen, p.cnt.data(), p.tile.data(), p.ops_buf.data(), p.n_ops, &p.n_ops);
        p.ops = p.ops_buf.data();
    }
    return rc;
}

static int classify_cigar_host(xm_ctx *ctx, int mode, uint64_t n,
                               const int32_t *nm1, const uint32_t *off1, const uint32_t *ops1, const int32_t *xs1,
                               const int32_t *nm2, const uint32_t *off2, const uint32_t *ops2, const int32_t *xs2,
                               const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint64_t counts[64],
                               bool compact, uint32_t *idx_out, uint64_t bin_offsets[8])
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || (compact && !bin_offsets)) return XM_ERR_INVALID_ARG;
    if (counts) memset(counts, 0, 64 * sizeof(uint64_t));
    if (compact) memset(bin_offsets, 0, 8 * sizeof(uint64_t));
    if (n == 0) return XM_OK;
    if (!nm1 || !off1 || !xs1 || !nm2 || !off2 || !xs2 || !unit_bits) return XM_ERR_INVALID_ARG;
    if (compact ? !idx_out : !code_out) return XM_ERR_INVALID_ARG;
    XM_HIP(ctx, hipSetDevice(ctx->device));
    if ((off1[n] && !ops1) || (off2[n] && !ops2)) return XM_ERR_INVALID_ARG;
    // CSR -> packed columns (a byte per record instead of a 4-byte offset goes over PCIe, too)
    PackedSpecies p1, p2;
    int rc;
    if ((rc = pack_species(n, off1, ops1, p1)) != XM_OK || (rc = pack_species(n, off2, ops2, p2)) != XM_OK) return rc;
    // one scratch slab: [nm1 | xs1 | nm2 | xs2 | cnt1 | cnt2 | tile1 | tile2 | bits | flag | ops1 | ops2], every piece 256-byte aligned
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t sz_col = up(n * 4), sz_cnt = up(n + 4), sz_tile = up((XM_CIG_TILES(n) + 1) * 4), sz_bits = up(((n + 63) / 64) * 8);
    const size_t total = 4 * sz_col + 2 * sz_cnt + 2 * sz_tile + sz_bits + 256 + up(p1.n_ops * 4 + 4) + up(p2.n_ops * 4 + 4);
    if ((rc = ensure_scratch(ctx, 0, total)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 5, code_out ? (size_t)n + 16 : (size_t)XM_BINS4_BYTES(n))) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 6, (size_t)n * 4)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 7, 72 * sizeof(uint64_t))) != XM_OK) return rc;
    uint8_t *base = (uint8_t *)ctx->d_scratch[0];
    size_t o = 0;
    auto put = [&](const void *src, size_t bytes, size_t reserve) -> void * {
        void *d = base + o;
        o += reserve;
        if (bytes && src && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return d;
    };
    void *d_nm1 = put(nm1, n * 4, sz_col), *d_xs1 = put(xs1, n * 4, sz_col);
    void *d_nm2 = put(nm2, n * 4, sz_col), *d_xs2 = put(xs2, n * 4, sz_col);
    void *d_cnt1 = put(p1.cnt.data(), n, sz_cnt), *d_cnt2 = put(p2.cnt.data(), n, sz_cnt);
    void *d_tile1 = put(p1.tile.data(), p1.tile.size() * 4, sz_tile), *d_tile2 = put(p2.tile.data(), p2.tile.size() * 4, sz_tile);
    void *d_bits = put(unit_bits, ((n + 63) / 64) * 8, sz_bits);
    void *d_flag = put(nullptr, 0, 256);
    void *d_ops1 = put(p1.ops, p1.n_ops * 4, up(p1.n_ops * 4 + 4)), *d_ops2 = put(p2.ops, p2.n_ops * 4, up(p2.n_ops * 4 + 4));
    if (!d_nm1 || !d_xs1 || !d_nm2 || !d_xs2 || !d_cnt1 || !d_cnt2 || !d_tile1 || !d_tile2 || !d_bits || !d_ops1 || !d_ops2)
        return fail_hip(ctx, hipGetLastError(), "hipMemcpy(cigar columns)");
    XM_HIP(ctx, hipMemset(d_flag, 0, 16));
    // the category bytes when the caller wants them, else the compact category stream (as classify_compact_host)
    uint8_t *d_code = code_out ? (uint8_t *)ctx->d_scratch[5] : nullptr;
    uint8_t *d_bins4 = code_out ? nullptr : (uint8_t *)ctx->d_scratch[5];
    uint32_t *d_idx = (uint32_t *)ctx->d_scratch[6];
    uint64_t *d_off = (uint64_t *)ctx->d_scratch[7];
    // the list split is cheap next to the copies, so the one fused entry point serves xm_classify_cigar as well
    rc = xm_classify_compact_cigar_packed_dev(ctx, nullptr, mode, n, (const int32_t *)d_nm1, (const uint8_t *)d_cnt1,
                                              (const uint32_t *)d_tile1, (const uint32_t *)d_ops1, (const int32_t *)d_xs1,
                                              (const int32_t *)d_nm2, (const uint8_t *)d_cnt2, (const uint32_t *)d_tile2,
                                              (const uint32_t *)d_ops2, (const int32_t *)d_xs2, (const uint64_t *)d_bits,
                                              min_score_floor, d_code, d_bins4, (uint32_t *)d_flag, d_idx, d_off, d_off + 8);
    if (rc != XM_OK) return rc;
    uint32_t flag = 0;
    XM_HIP(ctx, hipMemcpy(&flag, d_flag, 4, hipMemcpyDeviceToHost));
    if (flag) return XM_ERR_RANGE;
    if (compact) return fetch_compact_results(ctx, n, d_code, d_idx, d_off, code_out, idx_out, bin_offsets, counts);
    XM_HIP(ctx, hipMemcpy(code_out, d_code, (size_t)n, hipMemcpyDeviceToHost));
    if (counts) XM_HIP(ctx, hipMemcpy(counts, d_off + 8, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return XM_OK;
}

int xm_classify_cigar(xm_ctx *ctx, int mode, uint64_t n,
                      const int32_t *nm1, const uint32_t *off1, const uint32_t *ops1, const int32_t *xs1,
                      const int32_t *nm2, const uint32_t *off2, const uint32_t *ops2, const int32_t *xs2,
                      const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint64_t counts[64])
{
    return classify_cigar_host(ctx, mode, n, nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits, min_score_floor, code_out,
                               counts, false, nullptr, nullptr);
}

int xm_classify_compact_cigar(xm_ctx *ctx, int mode, uint64_t n,
                              const int32_t *nm1, const uint32_t *off1, const uint32_t *ops1, const int32_t *xs1,
                              const int32_t *nm2, const uint32_t *off2, const uint32_t *ops2, const int32_t *xs2,
                              const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out,
                              uint32_t *idx_out, uint64_t bin_offsets[8], uint64_t counts[64])
{
    return classify_cigar_host(ctx, mode, n, nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits, min_score_floor, code_out,
                               counts, true, idx_out, bin_offsets);
}

int xm_mate_correlate(xm_ctx *ctx, uint64_t n, const double *track, uint64_t m, const double *density, double *out)
{
    if (!ctx || m > 0xFFFFFFFFull) return XM_ERR_INVALID_ARG;
    if (n == 0) return XM_OK;
    if (!track || !out || (m && !density)) return XM_ERR_INVALID_ARG;
    XM_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure_scratch(ctx, 0, n * 8)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 1, (m ? m : 1) * 8)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 2, n * 8)) != XM_OK) return rc;
    XM_HIP(ctx, hipMemcpy(ctx->d_scratch[0], track, n * 8, hipMemcpyHostToDevice));
    if (m) XM_HIP(ctx, hipMemcpy(ctx->d_scratch[1], density, m * 8, hipMemcpyHostToDevice));
    rc = xm_mate_correlate_dev(ctx, nullptr, n, (const double *)ctx->d_scratch[0], m, (const double *)ctx->d_scratch[1],
                               (double *)ctx->d_scratch[2]);
    if (rc != XM_OK) return rc;
    XM_HIP(ctx, hipMemcpy(out, ctx->d_scratch[2], n * 8, hipMemcpyDeviceToHost));
    return XM_OK;
}

int xm_compact(xm_ctx *ctx, int mode, uint64_t n, const uint8_t *code, uint32_t *idx_out,
               uint64_t bin_offsets[8], uint64_t counts[64])
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || !bin_offsets) return XM_ERR_INVALID_ARG;
    memset(bin_offsets, 0, 8 * sizeof(uint64_t));
    if (counts) memset(counts, 0, 64 * sizeof(uint64_t));
    if (n == 0) return XM_OK;
    if (!code || !idx_out) return XM_ERR_INVALID_ARG;
    XM_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure_scratch(ctx, 5, (size_t)n + 16)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 6, (size_t)n * 4)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 7, 72 * sizeof(uint64_t))) != XM_OK) return rc;
    uint64_t *d_off = (uint64_t *)ctx->d_scratch[7];
    uint64_t *d_counts = d_off + 8;
    XM_HIP(ctx, hipMemcpy(ctx->d_scratch[5], code, (size_t)n, hipMemcpyHostToDevice));
    rc = xm_compact_dev(ctx, nullptr, mode, n, (const uint8_t *)ctx->d_scratch[5], (uint32_t *)ctx->d_scratch[6],
                        d_off, d_counts);
    if (rc != XM_OK) return rc;
    uint64_t host[72];
    XM_HIP(ctx, hipMemcpy(host, d_off, sizeof host, hipMemcpyDeviceToHost));
    memcpy(bin_offsets, host, 8 * sizeof(uint64_t));
    if (counts) memcpy(counts, host + 8, 64 * sizeof(uint64_t));
    if (host[7]) XM_HIP(ctx, hipMemcpy(idx_out, ctx->d_scratch[6], (size_t)host[7] * 4, hipMemcpyDeviceToHost));
    return XM_OK;
}

/* ---- host-buffer fused entry points: columns up, one fused pass, lists (and category bytes) down -------------- */

// results of a fused pass: bin offsets + counts first (they size the index copy), then the lists, then the bytes
static int fetch_compact_results(xm_ctx *ctx, uint64_t n, const uint8_t *d_code, const uint32_t *d_idx, const uint64_t *d_off,
                                 uint8_t *code_out, uint32_t *idx_out, uint64_t bin_offsets[8], uint64_t counts[64])
{
    uint64_t host[72];
    XM_HIP(ctx, hipMemcpy(host, d_off, sizeof host, hipMemcpyDeviceToHost));
    memcpy(bin_offsets, host, 8 * sizeof(uint64_t));
    if (counts) memcpy(counts, host + 8, 64 * sizeof(uint64_t));
    if (host[7]) XM_HIP(ctx, hipMemcpy(idx_out, d_idx, (size_t)host[7] * 4, hipMemcpyDeviceToHost));
    if (code_out) XM_HIP(ctx, hipMemcpy(code_out, d_code, (size_t)n, hipMemcpyDeviceToHost));
    return XM_OK;
}

static int classify_compact_host(xm_ctx *ctx, int mode, uint64_t n, size_t elem, const void *as1, const void *xs1,
                                 const void *as2, const void *xs2, const uint64_t *unit_bits, int32_t mi, double mf,
                                 uint8_t *code_out, uint32_t *idx_out, uint64_t bin_offsets[8], uint64_t counts[64])
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || !bin_offsets) return XM_ERR_INVALID_ARG;
    memset(bin_offsets, 0, 8 * sizeof(uint64_t));
    if (counts) memset(counts, 0, 64 * sizeof(uint64_t));
    if (n == 0) return XM_OK;
    if (!as1 || !xs1 || !as2 || !xs2 || !unit_bits || !idx_out) return XM_ERR_INVALID_ARG;
    XM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t col_bytes = (size_t)n * elem;
    const size_t bits_bytes = (size_t)((n + 63) / 64) * 8;
    const void *src[4] = {as1, xs1, as2, xs2};
    int rc;
    for (int c = 0; c < 4; ++c) {
        if ((rc = ensure_scratch(ctx, c, col_bytes)) != XM_OK) return rc;
        XM_HIP(ctx, hipMemcpy(ctx->d_scratch[c], src[c], col_bytes, hipMemcpyHostToDevice));
    }
    if ((rc = ensure_scratch(ctx, 4, bits_bytes)) != XM_OK) return rc;
    XM_HIP(ctx, hipMemcpy(ctx->d_scratch[4], unit_bits, bits_bytes, hipMemcpyHostToDevice));
    // scratch 5: the category bytes when the caller wants them, else the compact category stream (half the bytes, and
    // the faster scatter); both at once measured slower than either
    if ((rc = ensure_scratch(ctx, 5, code_out ? (size_t)n + 16 : (size_t)XM_BINS4_BYTES(n))) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 6, (size_t)n * 4)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 7, 72 * sizeof(uint64_t))) != XM_OK) return rc;
    uint8_t *d_code = code_out ? (uint8_t *)ctx->d_scratch[5] : nullptr;
    uint8_t *d_bins4 = code_out ? nullptr : (uint8_t *)ctx->d_scratch[5];
    uint32_t *d_idx = (uint32_t *)ctx->d_scratch[6];
    uint64_t *d_off = (uint64_t *)ctx->d_scratch[7];
    if (elem == 4)
        rc = xm_classify_compact_dev(ctx, nullptr, mode, n, (const int32_t *)ctx->d_scratch[0], (const int32_t *)ctx->d_scratch[1],
                                     (const int32_t *)ctx->d_scratch[2], (const int32_t *)ctx->d_scratch[3],
                                     (const uint64_t *)ctx->d_scratch[4], mi, d_code, d_bins4, d_idx, d_off, d_off + 8);
    else
        rc = xm_classify_compact_f64_dev(ctx, nullptr, mode, n, (const double *)ctx->d_scratch[0], (const double *)ctx->d_scratch[1],
                                         (const double *)ctx->d_scratch[2], (const double *)ctx->d_scratch[3],
                                         (const uint64_t *)ctx->d_scratch[4], mf, d_code, d_bins4, d_idx, d_off, d_off + 8);
    if (rc != XM_OK) return rc;
    return fetch_compact_results(ctx, n, d_code, d_idx, d_off, code_out, idx_out, bin_offsets, counts);
}

int xm_classify_compact(xm_ctx *ctx, int mode, uint64_t n,
                        const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                        const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out,
                        uint32_t *idx_out, uint64_t bin_offsets[8], uint64_t counts[64])
{
    return classify_compact_host(ctx, mode, n, sizeof(int32_t), as1, xs1, as2, xs2, unit_bits, min_score_floor, 0.0,
                                 code_out, idx_out, bin_offsets, counts);
}

int xm_classify_compact_f64(xm_ctx *ctx, int mode, uint64_t n,
                            const double *as1, const double *xs1, const double *as2, const double *xs2,
                            const uint64_t *unit_bits, double min_score, uint8_t *code_out,
                            uint32_t *idx_out, uint64_t bin_offsets[8], uint64_t counts[64])
{
    return classify_compact_host(ctx, mode, n, sizeof(double), as1, xs1, as2, xs2, unit_bits, 0, min_score,
                                 code_out, idx_out, bin_offsets, counts);
}


/* ---- host-buffer form of the six-list contract: columns up, the fused pass, six lists down ------------------------- */

static int classify_place_host(xm_ctx *ctx, int mode, uint64_t n, size_t elem, const void *as1, const void *xs1,
                               const void *as2, const void *xs2, const uint64_t *unit_bits, int32_t mi, double mf,
                               uint8_t *code_out, uint32_t *const idx_out[6], uint32_t *idx_state6, uint64_t list_capacity,
                               uint64_t n_out[8], uint64_t counts[64])
{
    if (!ctx || bad_mode(mode) || n > XM_MAX_RECORDS || !n_out || !idx_out) return XM_ERR_INVALID_ARG;
    memset(n_out, 0, 8 * sizeof(uint64_t));
    if (counts) memset(counts, 0, 64 * sizeof(uint64_t));
    for (int b = 0; b < 6; ++b)
        if (!idx_out[b]) return XM_ERR_INVALID_ARG;
    if (n == 0) return XM_OK;
    if (!as1 || !xs1 || !as2 || !xs2 || !unit_bits) return XM_ERR_INVALID_ARG;
    XM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t col_bytes = (size_t)n * elem;
    const size_t bits_bytes = (size_t)((n + 63) / 64) * 8;
    const void *src[4] = {as1, xs1, as2, xs2};
    int rc;
    for (int c = 0; c < 4; ++c) {
        if ((rc = ensure_scratch(ctx, c, col_bytes)) != XM_OK) return rc;
        XM_HIP(ctx, hipMemcpy(ctx->d_scratch[c], src[c], col_bytes, hipMemcpyHostToDevice));
    }
    if ((rc = ensure_scratch(ctx, 4, bits_bytes)) != XM_OK) return rc;
    XM_HIP(ctx, hipMemcpy(ctx->d_scratch[4], unit_bits, bits_bytes, hipMemcpyHostToDevice));
    if ((rc = ensure_scratch(ctx, 5, code_out ? (size_t)n + 16 : (size_t)XM_BINS4_BYTES(n))) != XM_OK) return rc;
    // seven device lists (the seventh only for binary64 columns), each with room for min(capacity, n) entries
    const uint64_t cap = list_capacity < n ? list_capacity : n;
    const size_t pitch = (((size_t)cap * 4) + 255) & ~(size_t)255;
    const int n_lists = (elem == 8 && idx_state6) ? 7 : 6;
    if ((rc = ensure_scratch(ctx, 6, pitch * n_lists + 256)) != XM_OK) return rc;
    if ((rc = ensure_scratch(ctx, 7, 72 * sizeof(uint64_t))) != XM_OK) return rc;
    uint8_t *d_code = code_out ? (uint8_t *)ctx->d_scratch[5] : nullptr;
    uint8_t *d_bins4 = code_out ? nullptr : (uint8_t *)ctx->d_scratch[5];
    uint32_t *d_list[7];
    for (int b = 0; b < 7; ++b) d_list[b] = b < n_lists ? (uint32_t *)((uint8_t *)ctx->d_scratch[6] + pitch * b) : nullptr;
    uint64_t *d_out = (uint64_t *)ctx->d_scratch[7];
    if (elem == 4)
        rc = xm_classify_place_dev(ctx, nullptr, mode, n, (const int32_t *)ctx->d_scratch[0], (const int32_t *)ctx->d_scratch[1],
                                   (const int32_t *)ctx->d_scratch[2], (const int32_t *)ctx->d_scratch[3],
                                   (const uint64_t *)ctx->d_scratch[4], mi, d_code, d_bins4, d_list, cap, d_out, d_out + 8);
    else
        rc = xm_classify_place_f64_dev(ctx, nullptr, mode, n, (const double *)ctx->d_scratch[0], (const double *)ctx->d_scratch[1],
                                       (const double *)ctx->d_scratch[2], (const double *)ctx->d_scratch[3],
                                       (const uint64_t *)ctx->d_scratch[4], mf, d_code, d_bins4, d_list, d_list[6], cap, d_out, d_out + 8);
    if (rc != XM_OK) return rc;
    uint64_t host[72];
    XM_HIP(ctx, hipMemcpy(host, d_out, sizeof host, hipMemcpyDeviceToHost));
    memcpy(n_out, host, 8 * sizeof(uint64_t));
    if (counts) memcpy(counts, host + 8, 64 * sizeof(uint64_t));
    for (int b = 0; b < n_lists; ++b) {
        const uint64_t k = host[b] < cap ? host[b] : cap;
        uint32_t *dst = b < 6 ? idx_out[b] : idx_state6;
        if (k) XM_HIP(ctx, hipMemcpy(dst, d_list[b], (size_t)k * 4, hipMemcpyDeviceToHost));
    }
    if (code_out) XM_HIP(ctx, hipMemcpy(code_out, d_code, (size_t)n, hipMemcpyDeviceToHost));
    return XM_OK;
}

int xm_classify_place(xm_ctx *ctx, int mode, uint64_t n,
                      const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                      const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out,
                      uint32_t *const idx_out[6], uint64_t list_capacity, uint64_t n_out[8], uint64_t counts[64])
{
    return classify_place_host(ctx, mode, n, sizeof(int32_t), as1, xs1, as2, xs2, unit_bits, min_score_floor, 0.0, code_out,
                               idx_out, nullptr, list_capacity, n_out, counts);
}

int xm_classify_place_f64(xm_ctx *ctx, int mode, uint64_t n,
                          const double *as1, const double *xs1, const double *as2, const double *xs2,
                          const uint64_t *unit_bits, double min_score, uint8_t *code_out,
                          uint32_t *const idx_out[6], uint32_t *idx_state6, uint64_t list_capacity,
                          uint64_t n_out[8], uint64_t counts[64])
{
    return classify_place_host(ctx, mode, n, sizeof(double), as1, xs1, as2, xs2, unit_bits, 0, min_score, code_out,
                               idx_out, idx_state6, list_capacity, n_out, counts);
}

/* ---- page-locking caller buffers ------------------------------------------------------------------------------- */

int xm_host_register(xm_ctx *ctx, void *ptr, size_t bytes)
{
    if (!ctx || !ptr || !bytes) return XM_ERR_INVALID_ARG;
    if (xmpin::is_registered(ptr)) return XM_ERR_INVALID_ARG;             // twice: the second unregister would leave the first's pages locked
    XM_HIP(ctx, hipSetDevice(ctx->device));
    XM_HIP(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    xmpin::note_registered(ptr, bytes);
    return XM_OK;
}

int xm_host_unregister(xm_ctx *ctx, void *ptr)
{
    if (!ctx || !ptr) return XM_ERR_INVALID_ARG;
    XM_HIP(ctx, hipSetDevice(ctx->device));
    XM_HIP(ctx, hipHostUnregister(ptr));
    xmpin::note_unregistered(ptr);
    return XM_OK;
}

int xm_pinned_bytes(uint64_t *allocated, uint64_t *registered, uint64_t *peak)
{
    xmpin::Book &b = xmpin::book();
    std::lock_guard<std::mutex> hold(b.lock);
    if (allocated) *allocated = b.allocated_bytes;
    if (registered) *registered = b.registered_bytes;
    if (peak) *peak = b.peak_bytes;
    return XM_OK;
}

/* ---- the one collective of the path: category_counts summed over the GPUs (RCCL) ------------------------------ */

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// One RCCL per process: a copy that is already mapped (PyTorch-ROCm brings its own librccl) is reused, otherwise the
// ROCm installation's is loaded.
RcclApi *rccl()
{
    static RcclApi api;
    if (api.handle || !api.error.empty()) return &api;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *nm : names)
        if (!api.handle) api.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    for (const char *nm : names)
        if (!api.handle) api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!api.handle) api.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!api.handle) {
        const char *why = dlerror();
        api.error = std::string("librccl not found: ") + (why ? why : "dlopen failed");
        return &api;
    }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
        api.error = "librccl lacks an expected entry point";
        api.handle = nullptr;
    }
    return &api;
}

int fail_rccl(xm_ctx *ctx, RcclApi *api, ncclResult_t r, const char *what)
{
    std::string msg = std::string(what) + ": " + (api->GetErrorString ? api->GetErrorString(r) : "RCCL error");
    if (ctx) ctx->last_error = msg;
    else g_create_error = msg;
    return XM_ERR_RCCL;
}

int no_rccl(xm_ctx *ctx, RcclApi *api)
{
    if (ctx) ctx->last_error = api->error;
    else g_create_error = api->error;
    return XM_ERR_RCCL;
}

}  // namespace

int xm_comm_unique_id(void *id_out)
{
    if (!id_out) return XM_ERR_INVALID_ARG;
    RcclApi *api = rccl();
    if (!api->handle) return no_rccl(nullptr, api);
    ncclUniqueId id;
    const ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail_rccl(nullptr, api, r, "ncclGetUniqueId");
    static_assert(sizeof id == XM_UNIQUE_ID_BYTES, "RCCL unique id size");
    memcpy(id_out, &id, sizeof id);
    return XM_OK;
}

int xm_comm_init(xm_ctx *ctx, int n_ranks, int rank, const void *unique_id)
{
    if (!ctx || !unique_id || n_ranks < 1 || rank < 0 || rank >= n_ranks || ctx->comm) return XM_ERR_INVALID_ARG;
    RcclApi *api = rccl();
    if (!api->handle) return no_rccl(ctx, api);
    XM_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    const ncclResult_t r = api->CommInitRank(&ctx->comm, n_ranks, id, rank);
    if (r != ncclSuccess) {
        ctx->comm = nullptr;
        return fail_rccl(ctx, api, r, "ncclCommInitRank");
    }
    ctx->comm_ranks = n_ranks;
    return XM_OK;
}

int xm_comm_destroy(xm_ctx *ctx)
{
    if (!ctx) return XM_ERR_INVALID_ARG;
    if (!ctx->comm) return XM_OK;
    RcclApi *api = rccl();
    (void)hipSetDevice(ctx->device);
    const ncclResult_t r = api->CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_ranks = 0;
    return r == ncclSuccess ? XM_OK : fail_rccl(ctx, api, r, "ncclCommDestroy");
}

int xm_comm_size(const xm_ctx *ctx) { return ctx ? ctx->comm_ranks : 0; }

int xm_allreduce_counts(xm_ctx *ctx, void *stream, uint64_t *counts)
{
    if (!ctx || !counts || !ctx->comm || wrong_device(ctx)) return XM_ERR_INVALID_ARG;
    RcclApi *api = rccl();
    const ncclResult_t r = api->AllReduce(counts, counts, 64, ncclUint64, ncclSum, ctx->comm, (hipStream_t)stream);
    return r == ncclSuccess ? XM_OK : fail_rccl(ctx, api, r, "ncclAllReduce(category_counts)");
}

/* ---- timing -------------------------------------------------------------------------------- */

int xm_timing_enable(xm_ctx *ctx, int on)
{
    if (!ctx) return XM_ERR_INVALID_ARG;
    ctx->timing = on != 0;
    return XM_OK;
}

int xm_timing_select(xm_ctx *ctx, uint32_t kernel_mask)
{
    if (!ctx) return XM_ERR_INVALID_ARG;
    ctx->timing_mask = kernel_mask;
    return XM_OK;
}

static int drain_spans(xm_ctx *ctx)
{
    for (auto &sp : ctx->spans) {
        XM_HIP(ctx, hipEventSynchronize(sp.stop));
        float ms = 0.f;
        XM_HIP(ctx, hipEventElapsedTime(&ms, sp.start, sp.stop));
        ctx->acc_ms[sp.kernel] += (double)ms;
        ctx->acc_launches[sp.kernel] += 1;
        ctx->free_events.push_back(sp.start);
        ctx->free_events.push_back(sp.stop);
    }
    ctx->spans.clear();
    return XM_OK;
}

int xm_timing_reset(xm_ctx *ctx)
{
    if (!ctx) return XM_ERR_INVALID_ARG;
    int rc = drain_spans(ctx);
    for (int k = 0; k < XM_K_COUNT; ++k) { ctx->acc_ms[k] = 0.0; ctx->acc_launches[k] = 0; }
    return rc;
}

int xm_timing_read(xm_ctx *ctx, double ms[XM_K_COUNT], uint64_t launches[XM_K_COUNT])
{
    if (!ctx || !ms || !launches) return XM_ERR_INVALID_ARG;
    int rc = drain_spans(ctx);
    for (int k = 0; k < XM_K_COUNT; ++k) { ms[k] = ctx->acc_ms[k]; launches[k] = ctx->acc_launches[k]; }
    return rc;
}

}  // extern "C"
