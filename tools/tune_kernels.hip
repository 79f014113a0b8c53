// tune_kernels.hip -- on-box A/B harness (not shipped): variants of the kernels interleaved in one process (guide rule
// 24); prints median/min per variant.  Before timing, the three forms of the pipeline (fused with category bytes,
// two calls with the stand-alone histogram, fused with the compact category stream) must agree bit for bit.
// (Round 2 also linked a frozen copy of the round-1 kernels for same-box old-vs-new runs: profiles/r02_tune_*.txt.)
//   tools/build_tune.sh && /tmp/tune [n_records] [rounds] [mode] [interleaved]     XM_TUNE_ONLY=substr XM_TUNE_TRACE=1
#include "../xenomapper_amd/csrc/xm_kernels.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// streaming ceiling with K1's footprint: 4 x 16 B loads + 4 B store per lane, no arithmetic
__global__ void __launch_bounds__(256) copy_like(const xm::v4i32 *a, const xm::v4i32 *b, const xm::v4i32 *c, const xm::v4i32 *d, uint32_t *out, uint64_t ngroups)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= ngroups) return;
    xm::v4i32 va = __builtin_nontemporal_load(a + g), vb = __builtin_nontemporal_load(b + g);
    xm::v4i32 vc = __builtin_nontemporal_load(c + g), vd = __builtin_nontemporal_load(d + g);
    out[g] = (uint32_t)(va.x ^ vb.y ^ vc.z ^ vd.w ^ va.w ^ vb.x ^ vc.y ^ vd.z ^ va.y ^ va.z ^ vb.z ^ vb.w ^ vc.x ^ vc.w ^ vd.x ^ vd.y);
}

static int32_t *A1, *X1, *A2, *X2; static uint64_t *BITS; static uint8_t *CODE, *CODE_OLD; static uint64_t N;
static uint32_t *GC, *GO, *IDX, *IDX_OLD, *CC_OLD, *CO_OLD;
static uint64_t *REP, *REP_OLD, *BINOFF, *BINOFF_OLD, *COUNTS, *COUNTS_OLD;
static int MODE = XM_MODE_PE_LIBERAL;

static uint32_t *PART;
static uint8_t *BINS4;
static xm::CountPlan cplan(uint32_t = 0) { xm::CountPlan cp; cp.plan = xm::plan_granules(N); cp.gran_counts = GC; cp.counts_rep = REP; cp.part_tot = PART; cp.bins4 = nullptr; return cp; }
// the compact-stream form of the fused step: K1 writes bins4 (and no category bytes), K2c reads bins4
static void nib_classify_counts() { auto cp = cplan(); cp.bins4 = BINS4; xm::launch_classify_i32(0, MODE, N, A1, X1, A2, X2, BITS, INT32_MIN, nullptr, &cp); }
static void nib_classify_counts_code() { auto cp = cplan(); cp.bins4 = BINS4; xm::launch_classify_i32(0, MODE, N, A1, X1, A2, X2, BITS, INT32_MIN, CODE, &cp); }
static void new_scan();
static void nib_scatter() { auto cp = cplan(); xm::launch_scatter(0, cp.plan, MODE, N, BINS4, true, GO, REP + 64 * 64, BINOFF, IDX); }
static void nib_fused() { for (int r = 0; r < 4; ++r) { nib_classify_counts(); new_scan(); nib_scatter(); } }
static void nib_fused_code() { for (int r = 0; r < 4; ++r) { nib_classify_counts_code(); new_scan(); nib_scatter(); } }


static void new_classify() { xm::launch_classify_i32(0, MODE, N, A1, X1, A2, X2, BITS, INT32_MIN, CODE, nullptr); }
static void new_classify_counts() { auto cp = cplan(); xm::launch_classify_i32(0, MODE, N, A1, X1, A2, X2, BITS, INT32_MIN, CODE, &cp); }
static void new_hist() { auto cp = cplan(); xm::launch_hist(0, MODE, N, CODE, cp); }
static void new_scan() { auto cp = cplan(); xm::launch_scan(0, cp, GO, REP + 64 * 64, COUNTS); }
template <bool DIRECT> static void scan_variant() { auto cp = cplan(); const uint32_t n_parts = (cp.plan.n_gran + XM_PART_GRAN - 1) / XM_PART_GRAN;
    if (!DIRECT) xm::part_sum_kernel<<<dim3(n_parts, 8), XM_SCAN_THREADS>>>(cp.gran_counts, cp.plan.n_gran, cp.plan.gran_stride, cp.part_tot);
    xm::scan_kernel<DIRECT><<<dim3(n_parts, 8), XM_SCAN_THREADS>>>(cp.gran_counts, cp.plan.n_gran, cp.plan.gran_stride, cp.part_tot, GO,
        (unsigned long long *)(REP + 64 * 64), (unsigned long long *)REP, (unsigned long long *)COUNTS); }
static void new_scatter();
template <bool DIRECT> static void scanv_fused() { for (int r = 0; r < 4; ++r) { new_classify_counts(); scan_variant<DIRECT>(); new_scatter(); } }
static void new_scatter() { auto cp = cplan(); xm::launch_scatter(0, cp.plan, MODE, N, CODE, false, GO, REP + 64 * 64, BINOFF, IDX); }
static void new_fused() { for (int r = 0; r < 4; ++r) { new_classify_counts(); new_scan(); new_scatter(); } }
static void new_unfused() { for (int r = 0; r < 4; ++r) { new_classify(); new_hist(); new_scan(); new_scatter(); } }
// two independent batches in flight on two streams (consecutive windows of a file): K2 of one under K1 of the other
struct Lane2 { hipStream_t st; uint8_t *code; uint32_t *gc, *go, *idx, *part; uint64_t *rep, *binoff, *counts; };
static Lane2 L2[2];
static void fused_on(const Lane2 &l) {
    xm::CountPlan cp; cp.plan = xm::plan_granules(N); cp.gran_counts = l.gc; cp.counts_rep = l.rep; cp.part_tot = l.part; cp.bins4 = nullptr;
    xm::launch_classify_i32(l.st, MODE, N, A1, X1, A2, X2, BITS, INT32_MIN, l.code, &cp);
    xm::launch_scan(l.st, cp, l.go, l.rep + 64 * 64, l.counts);
    xm::launch_scatter(l.st, cp.plan, MODE, N, l.code, false, l.go, l.rep + 64 * 64, l.binoff, l.idx);
}
static hipEvent_t EV_FORK, EV_JOIN[2];
static void new_fused_2stream() {        // 4 steps, alternating lanes; fork from / join to the null stream so that the timing events bracket it
    CK(hipEventRecord(EV_FORK, 0));
    for (int k = 0; k < 2; ++k) CK(hipStreamWaitEvent(L2[k].st, EV_FORK, 0));
    for (int r = 0; r < 4; ++r) fused_on(L2[r & 1]);
    for (int k = 0; k < 2; ++k) { CK(hipEventRecord(EV_JOIN[k], L2[k].st)); CK(hipStreamWaitEvent(0, EV_JOIN[k], 0)); }
}
// software pipeline over consecutive batches: all K1s on one stream, K2 of batch s on a second stream under K1 of batch
// s + 1; batch s + 2 reuses batch s's buffers, so its K1 waits for that K2
static hipStream_t ST_A, ST_B; static hipEvent_t EV_K1[8], EV_K2[8];
template <int STEPS> static void new_fused_pipelined() {
    CK(hipEventRecord(EV_FORK, 0));
    CK(hipStreamWaitEvent(ST_A, EV_FORK, 0)); CK(hipStreamWaitEvent(ST_B, EV_FORK, 0));
    for (int s = 0; s < STEPS; ++s) {
        const Lane2 &l = L2[s & 1];
        xm::CountPlan cp; cp.plan = xm::plan_granules(N); cp.gran_counts = l.gc; cp.counts_rep = l.rep; cp.part_tot = l.part;
        if (s >= 2) CK(hipStreamWaitEvent(ST_A, EV_K2[s - 2], 0));
        xm::launch_classify_i32(ST_A, MODE, N, A1, X1, A2, X2, BITS, INT32_MIN, l.code, &cp);
        CK(hipEventRecord(EV_K1[s], ST_A));
        CK(hipStreamWaitEvent(ST_B, EV_K1[s], 0));
        xm::launch_scan(ST_B, cp, l.go, l.rep + 64 * 64, l.counts);
        xm::launch_scatter(ST_B, cp.plan, MODE, N, l.code, false, l.go, l.rep + 64 * 64, l.binoff, l.idx);
        CK(hipEventRecord(EV_K2[s], ST_B));
    }
    CK(hipEventRecord(EV_JOIN[0], ST_A)); CK(hipEventRecord(EV_JOIN[1], ST_B));
    CK(hipStreamWaitEvent(0, EV_JOIN[0], 0)); CK(hipStreamWaitEvent(0, EV_JOIN[1], 0));
}
template <int STEPS> static void new_fused_serial() { for (int r = 0; r < STEPS; ++r) { new_classify_counts(); new_scan(); new_scatter(); } }
// read-only ceiling with K1's input footprint: 4 x 16 B loads per lane, one dword per wave out
__global__ void __launch_bounds__(256) read_like(const xm::v4i32 *a, const xm::v4i32 *b, const xm::v4i32 *c, const xm::v4i32 *d, uint32_t *out, uint64_t ngroups)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= ngroups) return;
    xm::v4i32 va = __builtin_nontemporal_load(a + g), vb = __builtin_nontemporal_load(b + g);
    xm::v4i32 vc = __builtin_nontemporal_load(c + g), vd = __builtin_nontemporal_load(d + g);
    uint32_t r = (uint32_t)(va.x ^ vb.y ^ vc.z ^ vd.w ^ va.w ^ vb.x ^ vc.y ^ vd.z ^ va.y ^ va.z ^ vb.z ^ vb.w ^ vc.x ^ vc.w ^ vd.x ^ vd.y);
    r ^= (uint32_t)__shfl_xor((int)r, 32, 64);
    if (r == 0x12345678u && (threadIdx.x & 63u) == 0u) out[g >> 6] = r;           // practically never: no write traffic
}
static void run_read() { read_like<<<(unsigned)((N / 4 + 255) / 256), 256>>>((const xm::v4i32 *)A1, (const xm::v4i32 *)X1, (const xm::v4i32 *)A2, (const xm::v4i32 *)X2, (uint32_t *)CODE, N / 4); }
// the same columns at skewed base addresses (copies inside one big buffer): do the four streams collide in the channels?
static int32_t *SK[4][4];        // [skew variant][column]
template <int V> static void skew_classify() { xm::launch_classify_i32(0, MODE, N, SK[V][0], SK[V][1], SK[V][2], SK[V][3], BITS, INT32_MIN, CODE, nullptr); }
template <int V> static void skew_read() { read_like<<<(unsigned)((N / 4 + 255) / 256), 256>>>((const xm::v4i32 *)SK[V][0], (const xm::v4i32 *)SK[V][1], (const xm::v4i32 *)SK[V][2], (const xm::v4i32 *)SK[V][3], (uint32_t *)CODE, N / 4); }
// the reference the other forms are compared with: classify without counting, stand-alone histogram, scan, scatter
static void ref_chain() {
    xm::CountPlan cp; cp.plan = xm::plan_granules(N); cp.gran_counts = CC_OLD; cp.counts_rep = REP_OLD; cp.part_tot = PART;
    xm::launch_classify_i32(0, MODE, N, A1, X1, A2, X2, BITS, INT32_MIN, CODE_OLD, nullptr);
    xm::launch_hist(0, MODE, N, CODE_OLD, cp);
    xm::launch_scan(0, cp, CO_OLD, REP_OLD + 64 * 64, COUNTS_OLD);
    xm::launch_scatter(0, cp.plan, MODE, N, CODE_OLD, false, CO_OLD, REP_OLD + 64 * 64, BINOFF_OLD, IDX_OLD);
}
static void run_copy() { copy_like<<<(unsigned)((N / 4 + 255) / 256), 256>>>((const xm::v4i32 *)A1, (const xm::v4i32 *)X1, (const xm::v4i32 *)A2, (const xm::v4i32 *)X2, (uint32_t *)CODE, N / 4); }


// ---- ablations of the scatter kernel (copies of the product body with switches; harness only) ----
// ABL 1: no stores; 6: bins that do not occur are skipped (wave-uniform branch); 3: stores go to a coalesced dummy position;
// 4: arithmetic byte -> bin (liberal rule) instead of the LDS table; 5: no guard compare; 8: XCD-contiguous granules
namespace abl {
using namespace xm;
template <int SLOTS, int ABL>
__device__ __forceinline__ void scatter_256(uint32_t w, const uint8_t *lut, uint32_t rec0, uint32_t base[7],
                                            uint32_t *__restrict__ idx_out, uint32_t n_units)
{
    uint32_t bin[4], pos[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!((SLOTS >> j) & 1)) { bin[j] = 7u; continue; }
        if (ABL == 4) {
            const uint32_t c = (w >> (8 * j)) & 0xFFu, f = (c >> 3) & 7u, r = c & 7u;
            bin[j] = f < r ? f : r;
        } else bin[j] = (uint32_t)lut[(w >> (8 * j)) & 63u];
    }
#pragma unroll
    for (int b = 0; b < 7; ++b) {
        uint64_t m[4], any = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { m[j] = ((SLOTS >> j) & 1) ? __ballot(bin[j] == (uint32_t)b) : 0ull; any |= m[j]; }
        if (ABL == 6 && any == 0ull) continue;
        uint32_t t = base[b];
#pragma unroll
        for (int j = 0; j < 4; ++j) if ((SLOTS >> j) & 1) t = mbcnt64(m[j], t);
#pragma unroll
        for (int j = 0; j < 4; ++j) if ((SLOTS >> j) & 1) pos[j] = (bin[j] == (uint32_t)b) ? t : pos[j];
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) if ((SLOTS >> j) & 1) total += (uint32_t)__builtin_popcountll(m[j]);
        base[b] += total;
    }
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < j; ++i)
            if (((SLOTS >> j) & 1) && ((SLOTS >> i) & 1)) pos[j] += (bin[i] == bin[j]) ? 1u : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!((SLOTS >> j) & 1)) continue;
        if (ABL == 1) { asm volatile("" :: "v"(pos[j])); continue; }
        if (ABL == 3) { if (bin[j] < 7u) store_index<false>(idx_out, (rec0 + (uint32_t)j) >> 1, pos[j]); continue; }
        if (ABL == 7) { if (bin[j] < 7u && pos[j] < n_units) __builtin_nontemporal_store(rec0 + (uint32_t)j, idx_out + pos[j]); continue; }
        if (bin[j] < 7u && (ABL == 5 || pos[j] < n_units)) store_index<false>(idx_out, pos[j], rec0 + (uint32_t)j);
    }
}

template <int NSUB, int ABL, int GPW = 1>
__global__ void __launch_bounds__(XM_BLOCK)
scatter_kernel(const uint8_t *__restrict__ code, uint64_t n, int mode, uint32_t n_gran, uint32_t gran_stride,
               const uint32_t *__restrict__ gran_off, const unsigned long long *__restrict__ bin_totals,
               unsigned long long *__restrict__ bin_offsets, uint32_t *__restrict__ idx_out)
{
    __shared__ uint8_t lut_all[XM_BLOCK / 64][64];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t wg = blockIdx.x;
    if (ABL == 8) {             // workgroups go round-robin over the 8 XCDs: give every XCD one contiguous stretch of granules
        const uint32_t per = (gridDim.x + 7u) / 8u;
        wg = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
        if (wg >= gridDim.x) return;
    }
    const uint32_t g_first = (wg * (XM_BLOCK / 64) + wave) * GPW;
    if (g_first >= n_gran) return;
    uint8_t *lut = lut_all[wave];
    lut[lane] = (uint8_t)bin_of_code(mode, lane == 63u ? XM_NO_UNIT : lane);
    const uint32_t tot = (lane < 8u) ? (uint32_t)bin_totals[lane] : 0u;
    const uint32_t bin_start = wave_scan_incl(tot) - tot;
    const uint32_t n_units = lane_value(bin_start, 7);
    if (g_first == 0u && lane < 8u) bin_offsets[lane] = bin_start;
    uint32_t off[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) off[i] = (lane < 7u && g_first + i < n_gran) ? gran_off[(uint64_t)lane * gran_stride + g_first + i] : 0u;
    uint32_t w[NSUB], wn[NSUB];
    auto load = [&](uint32_t g, uint32_t dst[NSUB]) {
        const uint64_t rec_g = (uint64_t)g * (NSUB * 256u);
#pragma unroll
        for (int s = 0; s < NSUB; ++s) dst[s] = (g < n_gran && rec_g + NSUB * 256u <= n) ? *reinterpret_cast<const uint32_t *>(code + rec_g + s * 256u + lane * 4u) : 0xFFFFFFFFu;
    };
    load(g_first, w);
    lds_settle();
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const uint32_t g = g_first + i;
        if (i + 1 < GPW) load(g + 1, wn);
        const uint32_t lane_base = bin_start + off[i];
        uint32_t base[7];
#pragma unroll
        for (int b = 0; b < 7; ++b) base[b] = lane_value(lane_base, b);
        const uint32_t rec_g = g * (NSUB * 256u);
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const uint32_t rec0 = rec_g + (uint32_t)s * 256u + lane * 4u;
            const bool even_free = (w[s] & 0x00FF00FFu) == 0x00FF00FFu;
            if (__ballot(!even_free) == 0ull) scatter_256<0xA, ABL>(w[s], lut, rec0, base, idx_out, n_units);
            else scatter_256<0xF, ABL>(w[s], lut, rec0, base, idx_out, n_units);
        }
#pragma unroll
        for (int s = 0; s < NSUB; ++s) w[s] = wn[s];
    }
}
}  // namespace abl

static uint32_t *IDX_SCRATCH;
template <int ABL> static void ablp_scatter() { auto cp = cplan(); const uint32_t grid = (cp.plan.n_gran + 3) / 4;
    abl::scatter_kernel<8, ABL, 1><<<grid, XM_BLOCK>>>(CODE, N, MODE, cp.plan.n_gran, cp.plan.gran_stride, GO, (const unsigned long long *)(REP + 64 * 64), (unsigned long long *)BINOFF, IDX); }
template <int ABL> static void ablp_fused() { for (int r = 0; r < 4; ++r) { new_classify_counts(); new_scan(); ablp_scatter<ABL>(); } }
template <int GPW> static void gpw_scatter() { auto cp = cplan(); const uint32_t grid = (cp.plan.n_gran + 4 * GPW - 1) / (4 * GPW);
    abl::scatter_kernel<8, 0, GPW><<<grid, XM_BLOCK>>>(CODE, N, MODE, cp.plan.n_gran, cp.plan.gran_stride, GO, (const unsigned long long *)(REP + 64 * 64), (unsigned long long *)BINOFF, IDX); }
template <int GPW> static void gpw_fused() { for (int r = 0; r < 4; ++r) { new_classify_counts(); new_scan(); gpw_scatter<GPW>(); } }
template <int ABL> static void abl_scatter() { auto cp = cplan(); const uint32_t grid = ABL == 8 ? ((cp.plan.n_gran + 3) / 4 + 7) / 8 * 8 : (cp.plan.n_gran + 3) / 4;
    abl::scatter_kernel<8, ABL><<<grid, XM_BLOCK>>>(CODE, N, MODE, cp.plan.n_gran, cp.plan.gran_stride, GO, (const unsigned long long *)(REP + 64 * 64), (unsigned long long *)BINOFF, IDX_SCRATCH); }
// plain streaming stores with the scatter's output footprint (4 B per unit), and reads with its input footprint
__global__ void __launch_bounds__(256) stream_write(uint32_t *out, uint64_t n) { const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = (uint32_t)i; }
__global__ void __launch_bounds__(256) stream_write16(uint4 *out, uint64_t n16) { const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; if (i < n16) out[i] = make_uint4((uint32_t)i, 1, 2, 3); }
__global__ void __launch_bounds__(256) stream_write16_nt(uint4 *out, uint64_t n16) { const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; xm::v4i32 v = {(int)i, 1, 2, 3}; if (i < n16) __builtin_nontemporal_store(v, (xm::v4i32 *)out + i); }
static void run_stream_write16() { const uint64_t n16 = N / 8; stream_write16<<<(unsigned)((n16 + 255) / 256), 256>>>((uint4 *)IDX_SCRATCH, n16); }
static void run_stream_write16_nt() { const uint64_t n16 = N / 8; stream_write16_nt<<<(unsigned)((n16 + 255) / 256), 256>>>((uint4 *)IDX_SCRATCH, n16); }
static void run_stream_write() { const uint64_t units = N / 2; stream_write<<<(unsigned)((units + 255) / 256), 256>>>(IDX_SCRATCH, units); }

template <typename T> static std::vector<T> fetch(const T *d, size_t n) { std::vector<T> h(n); CK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost)); return h; }

static bool compare_outputs(const char *what)
{
    CK(hipDeviceSynchronize());
    auto off_new = fetch(BINOFF, 8), off_old = fetch(BINOFF_OLD, 8);
    auto cnt_new = fetch(COUNTS, 64), cnt_old = fetch(COUNTS_OLD, 64);
    bool ok = off_new == off_old && cnt_new == cnt_old;
    auto code_new = fetch(CODE, N), code_old = fetch(CODE_OLD, N);
    ok &= code_new == code_old;
    auto idx_new = fetch(IDX, off_old[7]), idx_old = fetch(IDX_OLD, off_old[7]);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < idx_old.size(); ++i) if (idx_new[i] != idx_old[i]) { if (!bad) first = i; ++bad; }
    ok &= bad == 0;
    printf("verify %-28s %s  (units %llu; offsets %s, counts %s, codes %s, idx mismatches %zu first at %zu)\n", what, ok ? "OK" : "MISMATCH",
           (unsigned long long)off_old[7], off_new == off_old ? "ok" : "BAD", cnt_new == cnt_old ? "ok" : "BAD",
           code_new == code_old ? "ok" : "BAD", bad, first);
    return ok;
}

int main(int argc, char **argv)
{
    N = argc > 1 ? strtoull(argv[1], 0, 10) : 100000000ull;
    const int rounds = argc > 2 ? atoi(argv[2]) : 15;
    MODE = argc > 3 ? atoi(argv[3]) : XM_MODE_PE_LIBERAL;
    const bool interleaved = !(argc > 4 && atoi(argv[4]) == 0);
    CK(hipMalloc(&A1, N * 4 + 64)); CK(hipMalloc(&X1, N * 4 + 64)); CK(hipMalloc(&A2, N * 4 + 64)); CK(hipMalloc(&X2, N * 4 + 64));
    CK(hipMalloc(&BITS, N / 8 + 64)); CK(hipMalloc(&CODE, N + 64)); CK(hipMalloc(&CODE_OLD, N + 64));
    const size_t ws = 8 * (N / 1024 + 256) * 4;
    CK(hipMalloc(&GC, ws)); CK(hipMalloc(&GO, ws)); CK(hipMalloc(&CC_OLD, ws)); CK(hipMalloc(&CO_OLD, ws));
    CK(hipMalloc(&IDX, N * 4 + 64)); CK(hipMalloc(&IDX_OLD, N * 4 + 64)); CK(hipMalloc(&IDX_SCRATCH, N * 4 + 64));
    CK(hipMalloc(&BINOFF, 64)); CK(hipMalloc(&BINOFF_OLD, 64)); CK(hipMalloc(&COUNTS, 512)); CK(hipMalloc(&COUNTS_OLD, 512));
    CK(hipMalloc(&REP, (64 * 64 + 8) * 8)); CK(hipMalloc(&REP_OLD, (64 * 64 + 8) * 8));
    CK(hipMemset(REP, 0, (64 * 64 + 8) * 8)); CK(hipMemset(REP_OLD, 0, (64 * 64 + 8) * 8));
    CK(hipMalloc(&PART, 8 * XM_PART_STRIDE * 4)); CK(hipMalloc(&BINS4, (N + 2047) / 2048 * 1024 + 64));
    CK(hipEventCreate(&EV_FORK)); CK(hipEventCreate(&EV_JOIN[0])); CK(hipEventCreate(&EV_JOIN[1]));
    CK(hipStreamCreateWithFlags(&ST_A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&ST_B, hipStreamNonBlocking));
    for (int k = 0; k < 8; ++k) { CK(hipEventCreateWithFlags(&EV_K1[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&EV_K2[k], hipEventDisableTiming)); }
    for (int k = 0; k < 2; ++k) {
        Lane2 &l = L2[k];
        CK(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking));
        CK(hipMalloc(&l.code, N + 64)); CK(hipMalloc(&l.gc, ws)); CK(hipMalloc(&l.go, ws)); CK(hipMalloc(&l.idx, N * 4 + 64));
        CK(hipMalloc(&l.part, 8 * XM_PART_STRIDE * 4)); CK(hipMalloc(&l.rep, (64 * 64 + 8) * 8)); CK(hipMemset(l.rep, 0, (64 * 64 + 8) * 8));
        CK(hipMalloc(&l.binoff, 64)); CK(hipMalloc(&l.counts, 512));
    }
    {   // pair-structured scores: ~86 % primary, 9 % secondary, 3 % both, 2 % neither, mates share the origin
        std::vector<int32_t> a1(N), x1(N), a2(N), x2(N);
        uint64_t s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); };
        for (uint64_t p = 0; p < (N + 1) / 2; ++p) {
            uint32_t o = rnd() % 100; int origin = o < 86 ? 0 : o < 95 ? 1 : o < 98 ? 2 : 3;
            for (int mte = 0; mte < 2; ++mte) {
                uint64_t i = 2 * p + mte;
                if (i >= N) break;
                int32_t ha = (rnd() % 100 < 98) ? (int32_t)(300 - 2 * (rnd() % 40)) : INT32_MIN;
                int32_t hx = (ha != INT32_MIN && rnd() % 100 < 35) ? ((rnd() % 4 == 0) ? ha : (int32_t)(61 + rnd() % (ha - 60))) : INT32_MIN;
                int32_t oa = (rnd() % 100 < 15) ? (int32_t)(61 + rnd() % 160) : INT32_MIN;
                int32_t ox = (oa != INT32_MIN && rnd() % 100 < 40) ? (int32_t)(61 + rnd() % (oa - 60)) : INT32_MIN;
                if (origin == 0) { a1[i] = ha; x1[i] = hx; a2[i] = oa; x2[i] = ox; }
                else if (origin == 1) { a1[i] = oa; x1[i] = ox; a2[i] = ha; x2[i] = hx; }
                else if (origin == 2) { a1[i] = ha; x1[i] = hx; a2[i] = ha; x2[i] = INT32_MIN; }
                else { a1[i] = x1[i] = a2[i] = x2[i] = INT32_MIN; }
            }
        }
        CK(hipMemcpy(A1, a1.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(X1, x1.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(A2, a2.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(X2, x2.data(), N * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemset(BITS, (MODE == XM_MODE_SE || !interleaved) ? 0xFF : 0xAA, N / 8 + 64));
    {   // skewed copies of the columns: column c starts c * skew bytes past a 2 MiB-aligned slot
        const size_t skews[4] = {0, 4096, 65536 + 4096, (1u << 20) + 65536 + 4096 + 256};
        const size_t slot = ((N * 4 + (4u << 20)) + (2u << 20) - 1) / (2u << 20) * (2u << 20);
        for (int v = 0; v < 4; ++v) {
            char *big; CK(hipMalloc(&big, 4 * slot + (8u << 20)));
            char *aligned = (char *)(((uintptr_t)big + (2u << 20) - 1) / (2u << 20) * (2u << 20));
            const int32_t *src[4] = {A1, X1, A2, X2};
            for (int c = 0; c < 4; ++c) { SK[v][c] = (int32_t *)(aligned + c * slot + c * skews[v]); CK(hipMemcpy(SK[v][c], src[c], N * 4, hipMemcpyDeviceToDevice)); }
        }
        printf("column bases A1 %p X1 %p A2 %p X2 %p\n", (void *)A1, (void *)X1, (void *)A2, (void *)X2);
    }

    // correctness first: the two-call chain as the reference, then the fused forms against it
    ref_chain();
    new_classify_counts(); new_scan(); new_scatter();
    bool ok = compare_outputs("fused K1+counts/scan/scatter");
    CK(hipMemset(IDX, 0xEE, N * 4)); CK(hipMemset(COUNTS, 0xEE, 512)); CK(hipMemset(BINOFF, 0xEE, 64)); CK(hipMemset(CODE, 0xEE, N));
    new_classify(); new_hist(); new_scan(); new_scatter();
    ok &= compare_outputs("K1, hist/scan/scatter");
    CK(hipMemset(IDX, 0xEE, N * 4)); CK(hipMemset(COUNTS, 0xEE, 512)); CK(hipMemset(BINOFF, 0xEE, 64));
    nib_classify_counts(); new_scan(); nib_scatter();              // CODE still holds the previous (correct) bytes: only idx/offsets/counts tell
    ok &= compare_outputs("fused via bins4 (no category bytes)");
#ifndef XM_TUNE_NOVERIFY
    if (!ok) { printf("STOP: results differ\n"); return 1; }
#endif

    struct Cfg { const char *name; void (*fn)(); double bytes; void (*prep)(); };

    const double cls = 16.5 * (double)N, step = 19.0 * (double)N;
    std::vector<Cfg> cfgs = {
        {"copy_like NT", run_copy, cls}, {"read_like NT", run_read, 16.0 * N},
        {"read skew 0", skew_read<0>, 16.0 * N}, {"read skew 4K", skew_read<1>, 16.0 * N}, {"read skew 68K", skew_read<2>, 16.0 * N}, {"read skew 1M+", skew_read<3>, 16.0 * N},
        {"classify skew 0", skew_classify<0>, cls}, {"classify skew 4K", skew_classify<1>, cls}, {"classify skew 68K", skew_classify<2>, cls}, {"classify skew 1M+", skew_classify<3>, cls},
        {"r02 classify", new_classify, cls}, {"r02 classify+counts", new_classify_counts, cls}, {"r02 hist", new_hist, (double)N},
        {"r02 scan", new_scan, 0}, {"scan 1 launch", scan_variant<true>, 0}, {"scan 2 launches", scan_variant<false>, 0},
        {"fused x4 scan1", scanv_fused<true>, 4 * step}, {"fused x4 scan2", scanv_fused<false>, 4 * step}, {"r02 scatter", new_scatter, 3.0 * N},
        {"abl scatter product", abl_scatter<0>, 3.0 * N}, {"abl scatter no-store", abl_scatter<1>, 3.0 * N},
        {"abl scatter skip-empty", abl_scatter<6>, 3.0 * N}, {"abl scatter xcd-contig", abl_scatter<8>, 3.0 * N}, {"abl scatter coalesced-st", abl_scatter<3>, 3.0 * N},
        {"abl scatter arith-decode", abl_scatter<4>, 3.0 * N}, {"abl scatter no-guard", abl_scatter<5>, 3.0 * N},
        {"stream write 4B/unit", run_stream_write, 2.0 * N}, {"stream write 16B/lane", run_stream_write16, 2.0 * N},
        {"stream write 16B nt", run_stream_write16_nt, 2.0 * N},
 {"r02 unfused x4", new_unfused, 4 * step}, {"r02 fused x4", new_fused, 4 * step}, {"bins4 fused x4", nib_fused, 4 * step}, {"bins4+code fused x4", nib_fused_code, 4 * step},
        {"bins4 classify+counts", nib_classify_counts, cls}, {"bins4 scatter", nib_scatter, 2.5 * N}, {"r02 fused x4 2-stream", new_fused_2stream, 4 * step},
        {"fused x4 pipelined", new_fused_pipelined<4>, 4 * step}, {"fused x8 serial", new_fused_serial<8>, 8 * step},
        {"fused x8 pipelined", new_fused_pipelined<8>, 8 * step},
        {"fused x4 all-bins", ablp_fused<0>, 4 * step}, {"fused x4 skip-empty", ablp_fused<6>, 4 * step},
        {"fused x4 no-guard", ablp_fused<5>, 4 * step}, {"fused x4 nt-store", ablp_fused<7>, 4 * step},
        {"fused x4 arith", ablp_fused<4>, 4 * step}, {"fused x4 no-store", ablp_fused<1>, 4 * step},
    };
    if (const char *only = getenv("XM_TUNE_ONLY")) {          // comma-free substring filter: time only the matching variants
        std::vector<Cfg> keep;
        for (auto &c : cfgs) if (strstr(c.name, only)) keep.push_back(c);
        cfgs = keep;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> t(cfgs.size());
    for (int r = 0; r < rounds + 2; ++r)
        for (size_t k = 0; k < cfgs.size(); ++k) {
            if (getenv("XM_TUNE_TRACE")) { printf("run %s\n", cfgs[k].name); fflush(stdout); }
            if (cfgs[k].prep) cfgs[k].prep();
            CK(hipEventRecord(e0)); cfgs[k].fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) t[k].push_back(ms);
        }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    // the timed launches left the buffers in a consistent state again? (run the chains once more and compare)
    ref_chain();
    new_classify_counts(); new_scan(); new_scatter();
    ok = compare_outputs("after timing, fused");
    auto off = fetch(BINOFF, 8);
    printf("N=%llu records mode=%d interleaved=%d; bin_offsets:", (unsigned long long)N, MODE, (int)interleaved);
    for (int b = 0; b < 8; ++b) printf(" %llu", (unsigned long long)off[b]);
    printf("\n");
    for (size_t k = 0; k < cfgs.size(); ++k) {
        std::sort(t[k].begin(), t[k].end());
        float med = t[k][t[k].size() / 2], mn = t[k][0];
        printf("%-24s median %7.1f us  min %7.1f us   %6.0f GB/s\n", cfgs[k].name, med * 1e3, mn * 1e3,
               cfgs[k].bytes / (med * 1e-3) / 1e9);
    }
    return ok ? 0 : 1;
}
