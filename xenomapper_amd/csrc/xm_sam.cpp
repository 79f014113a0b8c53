// xm_sam.cpp -- host-side SAM column stripper and line writer (C ABI in include/xenomapper_host.h).
//
// Restates, on raw bytes and in parallel, the text half of the reference's hot loop:
//   getReadPairs                 /root/reference/xenomapper/xenomapper.py:95-118
//   get_tag (field search)       :186-190        get_tag_with_ZS_as_XS :204-206
//   get_cigarbased_AS_tag (NM field + CIGAR operations)   :247-251
//   '\t'.join(fields) + print    :332-350, :423-448, :521-550
// It never evaluates a score, a state or a bin: those are the GPU's (csrc/xm_kernels.hip).
// Anything it cannot reproduce bit-for-bit is reported as an exception for the Python host to resolve.
#include "../../include/xenomapper_host.h"
#include "xm_pool.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <sched.h>
#include <sys/mman.h>
#include <cerrno>
#include <unistd.h>
#include <string>
#include <thread>
#include <vector>

namespace {

using xmh::Pool;
using xmh::parallel_for;

const int32_t ABSENT = INT32_MIN;

// Python's str.split() separators in the ASCII range (str.isspace): \t \n \v \f \r FS GS RS US and space
inline bool is_ws(unsigned char c)
{
    return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31);
}

// first index >= i (i < n) of a separator in s[0..n), or n.  Every separator is below 0x21 and the input is ASCII
// (the index pass has rejected anything else), so a word without a byte below 0x21 is skipped whole.
inline uint32_t token_end(const char *s, uint32_t i, uint32_t n)
{
    const uint64_t ones = 0x0101010101010101ull, tops = 0x8080808080808080ull;
    while (i + 8 <= n) {
        uint64_t w;
        std::memcpy(&w, s + i, 8);
        if ((w - ones * 0x21u) & ~w & tops) break;
        i += 8;
    }
    while (i < n && !is_ws((unsigned char)s[i])) ++i;
    return i;
}

struct Line {
    uint64_t off;
    uint32_t len;
};

struct Rec {                  // what the parser learnt about one line
    uint32_t name_off, name_len;      // first token, relative to the line
    uint32_t norm_len;                // length of '\t'.join(fields)
    uint32_t n_tok;
    int32_t a, x, nm;                 // AS (or unused), XS/ZS, NM
    uint8_t ex_a, ex_x;               // exception kind of the AS / XS column
    uint8_t normal;                   // already '\t'.join(fields)
    uint32_t ops_begin, ops_count;    // CIGAR ops in the worker's local vector
    uint32_t worker;
};

// per-worker scratch, kept across calls (fresh multi-megabyte vectors would be re-mapped and page-faulted on every
// window) and padded to its own cache lines (neighbouring vector headers would false-share on every push_back)
struct alignas(128) WorkerSlot {
    std::vector<std::pair<uint64_t, uint64_t>> ends;   // (terminator offset, next line start)
    std::vector<uint32_t> ops;                         // packed CIGAR operations
    bool non_ascii = false;
};

struct FileParse {
    std::vector<Line> lines;
    std::vector<Rec> recs;
    std::vector<WorkerSlot> slots;            // one per worker
    uint64_t complete_end = 0;                // offset just past the last complete line
    bool non_ascii = false;
};

}  // namespace

// Worker count when the caller passes 0: the CPUs this process may actually use -- the affinity mask, cut down to a
// cgroup CPU quota when one is set (a container with 16 CPUs' worth of quota on a 256-thread host) -- at most 64.
extern "C" int xmh_default_threads(void)
{
    long n = (long)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<long>(n > 0 ? n : 1 << 20, (long)CPU_COUNT(&set));
    long quota = -1, period = -1;
    if (FILE *fh = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                         // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        if (std::fscanf(fh, "%31s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atol(q);
        std::fclose(fh);
    } else {                                                                             // cgroup v1
        if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (std::fscanf(fq, "%ld", &quota) != 1) quota = -1;
            std::fclose(fq);
        }
        if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(fp, "%ld", &period) != 1) period = -1;
            std::fclose(fp);
        }
    }
    if (quota > 0 && period > 0) n = std::min(n, (quota + period - 1) / period);
    return (int)std::max<long>(1, std::min<long>(n, 64));
}

namespace {

// Map the pages of a window before the workers touch them.  A window of a memory-mapped file is first touched here;
// left to demand faults, sixteen workers take ~65 k minor faults per 256 MB through one address-space lock and the
// index pass runs at a fraction of its speed.  MADV_POPULATE_READ (Linux >= 5.14) maps the range in bulk; older
// kernels return EINVAL and the faults happen as before.  Harmless on memory that is already resident.
void prefault(const char *buf, uint64_t len, Pool &pool)
{
    if (!buf || len < (1u << 20)) return;
    const uintptr_t page = 4096, lo = (uintptr_t)buf & ~(page - 1), hi = ((uintptr_t)buf + len + page - 1) & ~(page - 1);
    const uint64_t pages = (hi - lo) / page;
    parallel_for(pool, pages, [&](int, uint64_t b, uint64_t e) {
        (void)madvise((void *)(lo + b * page), (size_t)((e - b) * page), MADV_POPULATE_READ);
    }, 8);
}

// ---- line index: universal newlines ('\n', '\r\n', '\r'), as Python text mode reads them --------------
void index_lines(const char *buf, uint64_t len, bool eof, Pool &pool, FileParse &fp)
{
    // a trailing '\r' might be the first half of "\r\n" continuing in the next window
    uint64_t usable = len;
    if (!eof && len > 0 && buf[len - 1] == '\r') usable = len - 1;
    if (fp.slots.size() < (size_t)pool.size()) fp.slots.resize((size_t)pool.size());
    for (auto &sl : fp.slots) { sl.ends.clear(); sl.non_ascii = false; }
    parallel_for(pool, usable, [&](int t, uint64_t b, uint64_t e) {
        auto &v = fp.slots[(size_t)t].ends;
        uint64_t hi = 0;
        auto one = [&](uint64_t p) {
            const unsigned char c = (unsigned char)buf[p];
            if (c == '\n') {
                if (p > 0 && buf[p - 1] == '\r') return;             // second half of "\r\n"
                v.emplace_back(p, p + 1);
            } else if (c == '\r') {
                const bool crlf = (p + 1 < len) && buf[p + 1] == '\n';
                v.emplace_back(p, p + (crlf ? 2 : 1));
            }
        };
        // eight bytes at a time: a word is looked at byte by byte only when it holds a '\n' or '\r'
        const uint64_t ones = 0x0101010101010101ull, tops = 0x8080808080808080ull;
        uint64_t p = b;
        for (; p + 8 <= e; p += 8) {
            uint64_t w;
            std::memcpy(&w, buf + p, 8);
            hi |= w;
            const uint64_t xn = w ^ (ones * (uint64_t)'\n'), xr = w ^ (ones * (uint64_t)'\r');
            if ((((xn - ones) & ~xn) | ((xr - ones) & ~xr)) & tops)
                for (uint64_t q = p; q < p + 8; ++q) one(q);
        }
        for (; p < e; ++p) {
            hi |= (uint64_t)(unsigned char)buf[p];
            one(p);
        }
        fp.slots[(size_t)t].non_ascii = (hi & tops) != 0;
    });
    fp.non_ascii = false;
    for (auto &sl : fp.slots) fp.non_ascii |= sl.non_ascii;
    // line k starts where terminator k-1 ends: per-chunk prefix of the terminator counts, then a parallel fill
    const size_t n_slots = fp.slots.size();
    std::vector<uint64_t> base(n_slots + 1, 0);
    for (size_t t = 0; t < n_slots; ++t) base[t + 1] = base[t] + fp.slots[t].ends.size();
    fp.lines.resize((size_t)base.back());
    std::vector<uint64_t> first_start(n_slots, 0);                    // start of the first line ending in chunk t
    uint64_t start = 0;
    for (size_t t = 0; t < n_slots; ++t) {
        first_start[t] = start;
        if (!fp.slots[t].ends.empty()) start = fp.slots[t].ends.back().second;
    }
    pool.run((int)std::min<size_t>(n_slots, (size_t)pool.size()), [&](int w) {
        for (size_t t = (size_t)w; t < n_slots; t += (size_t)pool.size()) {
            uint64_t st = first_start[t];
            Line *dst = fp.lines.data() + base[t];
            for (auto &pr : fp.slots[t].ends) {
                *dst++ = Line{st, (uint32_t)(pr.first - st)};
                st = pr.second;
            }
        }
    });
    fp.complete_end = start;
    if (eof && start < len) {                                        // last line without a terminator
        fp.lines.push_back(Line{start, (uint32_t)(len - start)});
        fp.complete_end = len;
        for (uint64_t p = start; p < len; ++p) fp.non_ascii |= ((unsigned char)buf[p] & 0x80) != 0;
    }
}

// ---- one line ---------------------------------------------------------------------------------------------
inline bool contains2(const char *s, uint32_t n, char c0, char c1)
{
    for (uint32_t i = 0; i + 1 < n; ++i)
        if (s[i] == c0 && s[i + 1] == c1) return true;
    return false;
}

// text after the last ':' as a plain integer in [-(2^31-1), 2^31-1]; false -> let Python's float()/int() decide
inline bool plain_int(const char *s, uint32_t n, int32_t &out)
{
    uint32_t b = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (s[i] == ':') b = i + 1;
    const char *p = s + b;
    uint32_t m = n - b, i = 0;
    bool neg = false;
    if (m && (p[0] == '-' || p[0] == '+')) { neg = p[0] == '-'; i = 1; }
    if (i >= m || m - i > 10) return false;
    uint64_t v = 0;
    for (; i < m; ++i) {
        if (p[i] < '0' || p[i] > '9') return false;
        v = v * 10 + (uint64_t)(p[i] - '0');
    }
    if (v > 2147483647ull) return false;
    out = neg ? -(int32_t)v : (int32_t)v;
    return true;
}

inline int cigar_op(char c)
{
    switch (c) {
    case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4;
    case 'H': return 5; case 'P': return 6; case '=': return 7; case 'X': return 8; default: return -1;
    }
}

void parse_line(const char *s, uint32_t n, int score_mode, uint32_t worker, std::vector<uint32_t> &ops, Rec &r)
{
    r.name_off = r.name_len = 0;
    r.norm_len = 0;
    r.n_tok = 0;
    r.a = r.x = r.nm = ABSENT;
    r.ex_a = r.ex_x = 0;
    r.ops_begin = (uint32_t)ops.size();
    r.ops_count = 0;
    r.worker = worker;
    const char xtag0 = (score_mode == XMH_SCORE_AS_ZS) ? 'Z' : 'X';
    uint32_t n_a = 0, n_x = 0;
    bool have_nm = false;
    const char *cig = nullptr;
    uint32_t cig_len = 0;
    bool normal = true;
    uint32_t i = 0, total = 0;
    while (i < n) {
        if (is_ws((unsigned char)s[i])) {
            // separators must be exactly one '\t' between tokens, none leading / trailing
            if (s[i] != '\t' || r.n_tok == 0 || i + 1 >= n || is_ws((unsigned char)s[i + 1])) normal = false;
            ++i;
            continue;
        }
        uint32_t b = i;
        i = token_end(s, i, n);
        const char *tok = s + b;
        const uint32_t tl = i - b;
        const uint32_t k = r.n_tok++;
        total += tl;
        if (k == 0) { r.name_off = b; r.name_len = tl; }
        else if (k == 5) { cig = tok; cig_len = tl; }
        else if (k >= 11) {
            if (score_mode != XMH_SCORE_CIGAR && contains2(tok, tl, 'A', 'S')) {
                if (++n_a == 1 && !plain_int(tok, tl, r.a)) r.ex_a = XMH_EX_NONINT;
            }
            if (contains2(tok, tl, xtag0, 'S')) {
                if (++n_x == 1 && !plain_int(tok, tl, r.x)) r.ex_x = XMH_EX_NONINT;
            }
            if (score_mode == XMH_SCORE_CIGAR && !have_nm && contains2(tok, tl, 'N', 'M')) {
                have_nm = true;                                      // NM[0]: the first match, no duplicate check (:247-250)
                if (!plain_int(tok, tl, r.nm)) r.ex_a = XMH_EX_NONINT;
            }
        }
    }
    if (n_a > 1) r.ex_a = XMH_EX_DUP;
    if (n_x > 1) r.ex_x = XMH_EX_DUP;
    r.norm_len = r.n_tok ? total + (r.n_tok - 1) : 0;
    r.normal = (normal && r.n_tok > 0) ? 1 : 0;
    if (score_mode == XMH_SCORE_CIGAR && have_nm) {
        if (r.n_tok < 6) {
            r.ex_a = XMH_EX_SHORT;
        } else {
            // re.findall(r'([0-9]+)([MIDNSHPX=])', cigar): a run of ASCII digits directly followed by an op letter
            uint64_t run = 0;
            uint32_t digits = 0;
            bool big = false;
            for (uint32_t p = 0; p < cig_len; ++p) {
                const char c = cig[p];
                if (c >= '0' && c <= '9') {
                    ++digits;
                    if (run < (1ull << 40)) run = run * 10 + (uint64_t)(c - '0');
                    continue;
                }
                const int op = cigar_op(c);
                if (op >= 0 && digits > 0) {
                    if (run >= (1ull << 28)) big = true;
                    else ops.push_back(((uint32_t)run << 4) | (uint32_t)op);
                }
                run = 0;
                digits = 0;
            }
            if (big && r.ex_a == 0) r.ex_a = XMH_EX_BIGLEN;
            r.ops_count = (uint32_t)ops.size() - r.ops_begin;
        }
    }
}

void parse_file(const char *buf, int score_mode, Pool &pool, FileParse &fp)
{
    const uint64_t n = fp.lines.size();
    fp.recs.resize(n);
    if (fp.slots.size() < (size_t)pool.size()) fp.slots.resize((size_t)pool.size());
    for (auto &sl : fp.slots) sl.ops.clear();
    parallel_for(pool, n, [&](int t, uint64_t b, uint64_t e) {
        auto &ops = fp.slots[(size_t)t].ops;
        for (uint64_t i = b; i < e; ++i)
            parse_line(buf + fp.lines[i].off, fp.lines[i].len, score_mode, (uint32_t)t, ops, fp.recs[i]);
    });
}

inline bool same_name(const char *b1, const FileParse &f1, uint64_t i, const char *b2, const FileParse &f2, uint64_t j)
{
    const Rec &a = f1.recs[i], &b = f2.recs[j];
    return a.name_len == b.name_len &&
           memcmp(b1 + f1.lines[i].off + a.name_off, b2 + f2.lines[j].off + b.name_off, a.name_len) == 0;
}

}  // namespace

struct xmh_parser {
    std::unique_ptr<Pool> pool;
    int n_threads;
    FileParse f[2];
    std::vector<uint64_t> sel[2];                 // line index of every yielded record, per file
    // output storage
    std::vector<int32_t> as1, xs1, as2, xs2, nm1, nm2;
    std::vector<uint32_t> off1, off2, ops1, ops2, llen1, llen2, nlen1, nlen2, exc_record;
    std::vector<uint64_t> loff1, loff2, bits;
    std::vector<uint8_t> lflag1, lflag2, exc_col, exc_kind;
};

extern "C" {

int xmh_abi_version(void) { return XMH_ABI_VERSION; }

const char *xmh_strerror(int status)
{
    switch (status) {
    case XMH_OK: return "ok";
    case XMH_ERR_INVALID_ARG: return "invalid argument";
    case XMH_ERR_OOM: return "out of memory";
    case XMH_ERR_NON_ASCII: return "non-ASCII byte in SAM input";
    case XMH_ERR_BAD_BAM: return "not a valid BGZF/BAM file";
    default: return "unknown status";
    }
}

int xmh_parser_create(int n_threads, xmh_parser **out)
{
    if (!out) return XMH_ERR_INVALID_ARG;
    xmh_parser *p = new (std::nothrow) xmh_parser();
    if (!p) return XMH_ERR_OOM;
    if (n_threads <= 0) n_threads = xmh_default_threads();
    p->n_threads = std::max(1, std::min(n_threads, 64));
    try {
        p->pool.reset(new Pool(p->n_threads));
    } catch (...) {                                                   // bad_alloc, or no more threads (system_error)
        delete p;
        return XMH_ERR_OOM;
    }
    *out = p;
    return XMH_OK;
}

int xmh_parser_destroy(xmh_parser *p)
{
    if (!p) return XMH_ERR_INVALID_ARG;
    delete p;
    return XMH_OK;
}

// lines and records of one window from what the BAM decoder already knows about them (xmh_bam_read_pre): no byte of the
// text is looked at.  XMH_NEED_TEXT: a line the text rules might split differently -- the caller parses the text instead.
// XMH_ERR_INVALID_ARG: a description points outside the operation array it came with (n_ops words): these arrays cross
// the language boundary as raw pointers and are rebased when windows are merged, so they are checked, never trusted.
static int fill_from_pre(const xmh_pre *pre, uint64_t n_pre, const uint32_t *ops, uint64_t n_ops, uint64_t len, int score_mode,
                         Pool &pool, FileParse &fp)
{
    // the lines that lie in the window completely (every line of BAM text ends with '\n'): line starts = prefix sums of
    // the line lengths, in two parallel passes over the same slices
    std::vector<uint64_t> part((size_t)pool.size() + 1, 0);
    parallel_for(pool, n_pre, [&](int t, uint64_t b, uint64_t e) {
        uint64_t sum = 0;
        for (uint64_t i = b; i < e; ++i) sum += (uint64_t)pre[i].line_len + 1;
        part[(size_t)t + 1] = sum;
    });
    for (size_t t = 1; t < part.size(); ++t) part[t] += part[t - 1];
    fp.lines.resize((size_t)n_pre);
    parallel_for(pool, n_pre, [&](int t, uint64_t b, uint64_t e) {
        uint64_t at = part[(size_t)t];
        for (uint64_t i = b; i < e; ++i) {
            fp.lines[(size_t)i] = Line{at, pre[i].line_len};
            at += (uint64_t)pre[i].line_len + 1;
        }
    });
    uint64_t n = n_pre;                                  // the first line that does not end inside the window cuts the list
    {
        uint64_t lo = 0, hi = n_pre;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if (fp.lines[(size_t)mid].off + fp.lines[(size_t)mid].len + 1 <= len) lo = mid + 1; else hi = mid;
        }
        n = lo;
    }
    fp.lines.resize((size_t)n);
    const uint64_t off = n ? fp.lines[(size_t)n - 1].off + fp.lines[(size_t)n - 1].len + 1 : 0;
    fp.complete_end = off;
    fp.non_ascii = false;
    fp.recs.resize((size_t)n);
    if (fp.slots.size() < (size_t)pool.size()) fp.slots.resize((size_t)pool.size());
    std::vector<uint8_t> weird((size_t)pool.size() + 1, 0);
    const bool cigar = score_mode == XMH_SCORE_CIGAR;
    auto &all_ops = fp.slots[0].ops;                 // one shared operation array: the records point into it
    all_ops.clear();
    uint64_t ops_hi = 0;
    if (cigar && n) {
        const uint64_t lo = pre[0].ops_at, hi = (uint64_t)pre[n - 1].ops_at + pre[n - 1].n_ops;
        if (lo > hi || hi > n_ops || (hi > lo && !ops)) return XMH_ERR_INVALID_ARG;
        all_ops.assign(ops + lo, ops + hi);
        ops_hi = hi;
    }
    const uint32_t ops_base = (cigar && n) ? pre[0].ops_at : 0u;
    std::vector<uint8_t> outside((size_t)pool.size() + 1, 0);
    parallel_for(pool, n, [&](int t, uint64_t b, uint64_t e) {
        uint8_t w = 0, bad = 0;
        for (uint64_t i = b; i < e; ++i) {
            const xmh_pre &q = pre[i];
            Rec &r = fp.recs[(size_t)i];
            w |= q.flags & XMH_PRE_WEIRD;
            if (cigar) bad |= (uint8_t)(q.ops_at < ops_base || (uint64_t)q.ops_at + q.n_ops > ops_hi);
            r.name_off = 0;
            r.name_len = q.name_len;
            r.norm_len = q.line_len;                 // the decoder prints '\t'.join(fields)
            r.normal = 1;
            r.n_tok = 11;                            // never blank, never short
            r.a = r.x = r.nm = ABSENT;
            r.ex_a = r.ex_x = 0;
            r.ops_begin = r.ops_count = 0;
            r.worker = 0;
            if (cigar) {
                r.nm = q.nm;
                r.ex_a = q.ex_nm;
                r.x = q.xs; r.ex_x = q.ex_xs;
                if (q.nm != ABSENT || q.ex_nm) { r.ops_begin = q.ops_at - ops_base; r.ops_count = q.n_ops; }   // parse_line: ops only with NM
            } else {
                r.a = q.as; r.ex_a = q.ex_as;
                if (score_mode == XMH_SCORE_AS_ZS) { r.x = q.zs; r.ex_x = q.ex_zs; } else { r.x = q.xs; r.ex_x = q.ex_xs; }
            }
        }
        weird[(size_t)t] = w;
        outside[(size_t)t] = bad;
    });
    for (uint8_t o : outside)
        if (o) return XMH_ERR_INVALID_ARG;
    for (uint8_t w : weird)
        if (w) return XMH_NEED_TEXT;
    return XMH_OK;
}

static int parse_common(xmh_parser *p, const char *buf1, uint64_t len1, int eof1, const xmh_pre *pre1, uint64_t n_pre1, const uint32_t *pops1,
                        uint64_t n_pops1,
                        const char *buf2, uint64_t len2, int eof2, const xmh_pre *pre2, uint64_t n_pre2, const uint32_t *pops2,
                        uint64_t n_pops2, bool from_pre, int score_mode, int paired, int skip_repeated, int keep_halo, uint64_t max_records,
                        xmh_block *out)
{
    if (!p || !out || (!buf1 && len1) || (!buf2 && len2) || score_mode < 0 || score_mode > 2)
        return XMH_ERR_INVALID_ARG;
    try {
        static const bool profile = getenv("XMH_PROFILE") != nullptr;
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        const auto t0 = now();
        const char *buf[2] = {buf1, buf2};
        const uint64_t len[2] = {len1, len2};
        const int eof[2] = {eof1, eof2};
        if (!from_pre)
            for (int f = 0; f < 2; ++f) prefault(buf[f], len[f], *p->pool);
        const auto t0b = now();
        if (from_pre) {
            const int r1 = fill_from_pre(pre1, n_pre1, pops1, n_pops1, len1, score_mode, *p->pool, p->f[0]);
            const int r2 = r1 == XMH_OK ? fill_from_pre(pre2, n_pre2, pops2, n_pops2, len2, score_mode, *p->pool, p->f[1]) : r1;
            if (r2 != XMH_OK) return r2;                          // XMH_NEED_TEXT, or descriptions that do not fit their arrays
        } else {
            for (int f = 0; f < 2; ++f) {
                index_lines(buf[f], len[f], eof[f] != 0, *p->pool, p->f[f]);
                if (p->f[f].non_ascii) return XMH_ERR_NON_ASCII;
            }
        }
        const auto t1 = now();
        if (!from_pre)
            for (int f = 0; f < 2; ++f) parse_file(buf[f], score_mode, *p->pool, p->f[f]);
        const auto t2 = now();

        // ---- the lock-step walk (xenomapper.py:103-117) ------------------------------------------------
        const uint64_t L[2] = {p->f[0].lines.size(), p->f[1].lines.size()};
        // all lines of the file are in the window (then running out of lines is EOF, i.e. a blank readline)
        const bool whole[2] = {eof1 && p->f[0].complete_end == len1, eof2 && p->f[1].complete_end == len2};
        auto blank = [&](int f, uint64_t i) { return p->f[f].recs[i].n_tok == 0; };
        p->sel[0].clear();
        p->sel[1].clear();
        uint64_t i1 = 0, i2 = 0;
        int ended = 0, starved = 0;
        int64_t mismatch = -1;
        if (!skip_repeated) {
            // record k is line k of both files: the walk stops at the first k that is blank in either file or
            // whose names differ -- found in parallel
            const uint64_t lim = std::min<uint64_t>(std::min(L[0], L[1]), max_records);
            std::vector<uint64_t> stop((size_t)std::max(1, p->n_threads), lim);
            parallel_for(*p->pool, lim, [&](int t, uint64_t b, uint64_t e) {
                for (uint64_t k = b; k < e; ++k)
                    if (blank(0, k) || blank(1, k) || !same_name(buf1, p->f[0], k, buf2, p->f[1], k)) { stop[(size_t)t] = k; break; }
            });
            uint64_t k_stop = lim;
            for (uint64_t v : stop) k_stop = std::min(k_stop, v);
            p->sel[0].resize((size_t)k_stop);
            p->sel[1].resize((size_t)k_stop);
            parallel_for(*p->pool, k_stop, [&](int, uint64_t b, uint64_t e) {
                for (uint64_t k = b; k < e; ++k) p->sel[0][(size_t)k] = p->sel[1][(size_t)k] = k;
            });
            i1 = i2 = k_stop;
            if (k_stop < lim) {
                if (blank(0, k_stop) || blank(1, k_stop)) ended = 1; else mismatch = (int64_t)k_stop;
            } else if (k_stop < max_records) {
                const bool end1 = i1 >= L[0] && whole[0], end2 = i2 >= L[1] && whole[1];
                if (end1 || end2) ended = 1; else starved = 1;
            }
        }
        if (skip_repeated) {
            // Each file is cut into runs of adjacent lines with one name (a blank line is a run of its own: it stops the
            // skipping, :110-117); pair k is the first line of run k of both files.  Run starts are collected in
            // parallel; the walk then stops at the first k that is blank, mismatched, or whose run reaches the end of
            // a window that is not the end of the file (the run may continue in the next window) -- in that order.
            const char *bufs[2] = {buf1, buf2};
            for (int f = 0; f < 2; ++f) {
                const FileParse &fp = p->f[f];
                const char *bf = bufs[f];
                auto is_start = [&](uint64_t i) {
                    return i == 0 || blank(f, i) || blank(f, i - 1) || !same_name(bf, fp, i, bf, fp, i - 1);
                };
                std::vector<uint64_t> cnt((size_t)p->pool->size() + 1, 0);
                parallel_for(*p->pool, L[f], [&](int t, uint64_t b, uint64_t e) {
                    uint64_t c = 0;
                    for (uint64_t i = b; i < e; ++i) c += is_start(i) ? 1 : 0;
                    cnt[(size_t)t + 1] = c;
                });
                for (size_t t = 1; t < cnt.size(); ++t) cnt[t] += cnt[t - 1];
                p->sel[f].resize((size_t)cnt.back());
                parallel_for(*p->pool, L[f], [&](int t, uint64_t b, uint64_t e) {     // same slices as the counting pass
                    uint64_t w = cnt[(size_t)t];
                    for (uint64_t i = b; i < e; ++i)
                        if (is_start(i)) p->sel[f][(size_t)w++] = i;
                });
            }
            const uint64_t R[2] = {p->sel[0].size(), p->sel[1].size()};
            const uint64_t lim = std::min<uint64_t>(std::min(R[0], R[1]), max_records);
            const auto &s0 = p->sel[0], &s1 = p->sel[1];
            auto open_run = [&](uint64_t k) { return (k + 1 == R[0] && !whole[0]) || (k + 1 == R[1] && !whole[1]); };
            std::vector<uint64_t> stop((size_t)p->pool->size(), lim);
            parallel_for(*p->pool, lim, [&](int t, uint64_t b, uint64_t e) {
                for (uint64_t k = b; k < e; ++k)
                    if (blank(0, s0[k]) || blank(1, s1[k]) || !same_name(buf1, p->f[0], s0[k], buf2, p->f[1], s1[k]) || open_run(k)) {
                        stop[(size_t)t] = k;
                        break;
                    }
            });
            uint64_t k_stop = lim;
            for (uint64_t v : stop) k_stop = std::min(k_stop, v);
            if (k_stop < lim) {
                if (blank(0, s0[k_stop]) || blank(1, s1[k_stop])) ended = 1;
                else if (!same_name(buf1, p->f[0], s0[k_stop], buf2, p->f[1], s1[k_stop])) mismatch = (int64_t)k_stop;
                else starved = 1;
            } else if (k_stop < max_records) {
                const bool end1 = R[0] == lim && whole[0], end2 = R[1] == lim && whole[1];
                if (end1 || end2) ended = 1; else starved = 1;
            }
            i1 = k_stop < R[0] ? s0[k_stop] : L[0];
            i2 = k_stop < R[1] ? s1[k_stop] : L[1];
            p->sel[0].resize((size_t)k_stop);
            p->sel[1].resize((size_t)k_stop);
        }
        const uint64_t n = p->sel[0].size();

        // ---- consumed bytes: where the next window starts ----------------------------------------------
        auto start_of = [&](int f, uint64_t line) -> uint64_t {
            return line < p->f[f].lines.size() ? p->f[f].lines[line].off : p->f[f].complete_end;
        };
        if (keep_halo && n > 0 && !ended && mismatch < 0) {
            out->consumed1 = start_of(0, p->sel[0][n - 1]);
            out->consumed2 = start_of(1, p->sel[1][n - 1]);
            out->consumed_lines1 = p->sel[0][n - 1];
            out->consumed_lines2 = p->sel[1][n - 1];
        } else {
            out->consumed1 = start_of(0, i1);
            out->consumed2 = start_of(1, i2);
            out->consumed_lines1 = std::min<uint64_t>(i1, L[0]);
            out->consumed_lines2 = std::min<uint64_t>(i2, L[1]);
        }

        // ---- gather columns for the yielded records -------------------------------------------------
        const bool cigar = score_mode == XMH_SCORE_CIGAR;
        std::vector<int32_t> *cols[2][3] = {{&p->as1, &p->xs1, &p->nm1}, {&p->as2, &p->xs2, &p->nm2}};
        std::vector<uint32_t> *offs[2] = {&p->off1, &p->off2}, *opsv[2] = {&p->ops1, &p->ops2};
        std::vector<uint64_t> *loff[2] = {&p->loff1, &p->loff2};
        std::vector<uint32_t> *llen[2] = {&p->llen1, &p->llen2}, *nlen[2] = {&p->nlen1, &p->nlen2};
        std::vector<uint8_t> *lflag[2] = {&p->lflag1, &p->lflag2};
        for (int f = 0; f < 2; ++f) {
            cols[f][0]->resize(n + 4); cols[f][1]->resize(n + 4); cols[f][2]->resize(cigar ? n + 4 : 4);
            loff[f]->resize(n + 1); llen[f]->resize(n + 1); nlen[f]->resize(n + 1); lflag[f]->resize(n + 1);
            offs[f]->assign(cigar ? n + 1 : 1, 0);
            const FileParse &fp = p->f[f];
            const auto &sel = p->sel[f];
            parallel_for(*p->pool, n, [&](int, uint64_t b, uint64_t e) {
                for (uint64_t k = b; k < e; ++k) {
                    const Rec &r = fp.recs[sel[k]];
                    (*cols[f][0])[k] = r.a;
                    (*cols[f][1])[k] = r.x;
                    if (cigar) (*cols[f][2])[k] = r.nm;
                    (*loff[f])[k] = fp.lines[sel[k]].off;
                    (*llen[f])[k] = fp.lines[sel[k]].len;
                    (*nlen[f])[k] = r.norm_len;
                    (*lflag[f])[k] = r.normal ? XMH_LINE_NORMAL : 0;
                }
            });
            if (cigar) {
                uint32_t acc = 0;
                for (uint64_t k = 0; k < n; ++k) { (*offs[f])[k] = acc; acc += fp.recs[sel[k]].ops_count; }
                (*offs[f])[n] = acc;
                opsv[f]->resize((size_t)acc + 4);
                parallel_for(*p->pool, n, [&](int, uint64_t b, uint64_t e) {
                    for (uint64_t k = b; k < e; ++k) {
                        const Rec &r = fp.recs[sel[k]];
                        if (r.ops_count)
                            memcpy(opsv[f]->data() + (*offs[f])[k], fp.slots[r.worker].ops.data() + r.ops_begin,
                                   (size_t)r.ops_count * 4);
                    }
                });
            } else {
                opsv[f]->assign(4, 0);
            }
        }
        // ---- unit mask (xenomapper.py:402: name equals the previous record's name) -------------------
        p->bits.assign((n + 63) / 64 + 1, 0);
        parallel_for(*p->pool, (n + 63) / 64, [&](int, uint64_t wb, uint64_t we) {
            for (uint64_t w = wb; w < we; ++w) {
                uint64_t word = 0;
                const uint64_t k0 = w * 64, k1 = std::min<uint64_t>(n, k0 + 64);
                for (uint64_t k = k0; k < k1; ++k) {
                    const bool unit = !paired || (k > 0 && same_name(buf1, p->f[0], p->sel[0][k], buf1, p->f[0], p->sel[0][k - 1]));
                    word |= (uint64_t)unit << (k & 63);
                }
                p->bits[w] = word;
            }
        });
        // ---- exceptions ---------------------------------------------------------------------------------
        p->exc_record.clear(); p->exc_col.clear(); p->exc_kind.clear();
        {
            struct Exc { uint32_t rec; uint8_t col, kind; };
            std::vector<std::vector<Exc>> part((size_t)std::max(1, p->n_threads));
            parallel_for(*p->pool, n, [&](int t, uint64_t b, uint64_t e) {
                for (uint64_t k = b; k < e; ++k)
                    for (int f = 0; f < 2; ++f) {
                        const Rec &r = p->f[f].recs[p->sel[f][k]];
                        if (r.ex_a) part[(size_t)t].push_back(Exc{(uint32_t)k, (uint8_t)(2 * f), r.ex_a});
                        if (r.ex_x) part[(size_t)t].push_back(Exc{(uint32_t)k, (uint8_t)(2 * f + 1), r.ex_x});
                    }
            });
            for (auto &v : part)
                for (auto &x : v) { p->exc_record.push_back(x.rec); p->exc_col.push_back(x.col); p->exc_kind.push_back(x.kind); }
        }
        if (profile)
            fprintf(stderr, "xmh_parse: %.1f MB  prefault %.1f ms  index %.1f ms  parse %.1f ms  walk+gather %.1f ms\n",
                    (double)(len1 + len2) / 1e6, ms(t0, t0b), ms(t0b, t1), ms(t1, t2), ms(t2, now()));
        out->n_records = n;
        out->ended = ended;
        out->starved = starved;
        out->mismatch_at = mismatch;
        out->as1 = p->as1.data(); out->xs1 = p->xs1.data(); out->as2 = p->as2.data(); out->xs2 = p->xs2.data();
        out->nm1 = p->nm1.data(); out->nm2 = p->nm2.data();
        out->cig_off1 = p->off1.data(); out->cig_off2 = p->off2.data();
        out->cig_ops1 = p->ops1.data(); out->cig_ops2 = p->ops2.data();
        out->unit_bits = p->bits.data();
        out->line_off1 = p->loff1.data(); out->line_off2 = p->loff2.data();
        out->line_len1 = p->llen1.data(); out->line_len2 = p->llen2.data();
        out->norm_len1 = p->nlen1.data(); out->norm_len2 = p->nlen2.data();
        out->line_flags1 = p->lflag1.data(); out->line_flags2 = p->lflag2.data();
        out->n_exc = p->exc_record.size();
        out->exc_record = p->exc_record.data(); out->exc_col = p->exc_col.data(); out->exc_kind = p->exc_kind.data();
        return XMH_OK;
    } catch (const std::bad_alloc &) {
        return XMH_ERR_OOM;
    }
}

int xmh_parse(xmh_parser *p, const char *buf1, uint64_t len1, int eof1, const char *buf2, uint64_t len2, int eof2,
              int score_mode, int paired, int skip_repeated, int keep_halo, uint64_t max_records, xmh_block *out)
{
    return parse_common(p, buf1, len1, eof1, nullptr, 0, nullptr, 0, buf2, len2, eof2, nullptr, 0, nullptr, 0, false, score_mode, paired,
                        skip_repeated, keep_halo, max_records, out);
}

int xmh_parse_pre(xmh_parser *p, const char *buf1, uint64_t len1, int eof1, const xmh_pre *pre1, uint64_t n_pre1, const uint32_t *ops1,
                  uint64_t n_ops1,
                  const char *buf2, uint64_t len2, int eof2, const xmh_pre *pre2, uint64_t n_pre2, const uint32_t *ops2,
                  uint64_t n_ops2,
                  int score_mode, int paired, int skip_repeated, int keep_halo, uint64_t max_records, xmh_block *out)
{
    if ((n_pre1 && !pre1) || (n_pre2 && !pre2)) return XMH_ERR_INVALID_ARG;
    return parse_common(p, buf1, len1, eof1, pre1, n_pre1, ops1, n_ops1, buf2, len2, eof2, pre2, n_pre2, ops2, n_ops2, true, score_mode, paired,
                        skip_repeated, keep_halo, max_records, out);
}

// ---- blocks stripped elsewhere (on the GPU: include/xenomapper_strip.h) -----------------------------------------------
int xmh_copy(xmh_parser *p, void *dst, const void *src, uint64_t n)
{
    if (!p || (n && (!dst || !src))) return XMH_ERR_INVALID_ARG;
    prefault((const char *)src, n, *p->pool);
    parallel_for(*p->pool, n, [&](int, uint64_t b, uint64_t e) {
        memcpy((char *)dst + b, (const char *)src + b, (size_t)(e - b));
    });
    return XMH_OK;
}

int xmh_pread(xmh_parser *p, int fd, uint64_t offset, void *dst, uint64_t n)
{
    if (!p || fd < 0 || (n && !dst)) return XMH_ERR_INVALID_ARG;
    std::vector<int> bad((size_t)p->pool->size() + 1, 0);
    parallel_for(*p->pool, n, [&](int t, uint64_t b, uint64_t e) {
        while (b < e) {
            const ssize_t got = pread(fd, (char *)dst + b, (size_t)(e - b), (off_t)(offset + b));
            if (got <= 0) {
                if (got < 0 && errno == EINTR) continue;
                bad[(size_t)t] = 1;                                  // an error, or the file is shorter than the caller said
                return;
            }
            b += (uint64_t)got;
        }
    });
    for (int v : bad)
        if (v) return XMH_ERR_INVALID_ARG;
    return XMH_OK;
}

int xmh_adopt_lines(xmh_parser *p, uint64_t n_records,
                    const uint32_t *line_off1, const uint32_t *line_len1, const uint32_t *norm_len1, const uint8_t *line_flags1,
                    const uint32_t *line_off2, const uint32_t *line_len2, const uint32_t *norm_len2, const uint8_t *line_flags2)
{
    if (!p || (n_records && (!line_off1 || !line_len1 || !norm_len1 || !line_flags1 || !line_off2 || !line_len2 || !norm_len2 ||
                             !line_flags2)))
        return XMH_ERR_INVALID_ARG;
    try {
        const uint32_t *off[2] = {line_off1, line_off2}, *len[2] = {line_len1, line_len2}, *nrm[2] = {norm_len1, norm_len2};
        const uint8_t *flg[2] = {line_flags1, line_flags2};
        std::vector<uint64_t> *loff[2] = {&p->loff1, &p->loff2};
        std::vector<uint32_t> *llen[2] = {&p->llen1, &p->llen2}, *nlen[2] = {&p->nlen1, &p->nlen2};
        std::vector<uint8_t> *lflag[2] = {&p->lflag1, &p->lflag2};
        for (int f = 0; f < 2; ++f) {
            p->sel[f].resize((size_t)n_records);                       // xmh_emit checks unit indices against its size
            loff[f]->resize(n_records + 1); llen[f]->resize(n_records + 1); nlen[f]->resize(n_records + 1); lflag[f]->resize(n_records + 1);
            parallel_for(*p->pool, n_records, [&](int, uint64_t b, uint64_t e) {
                for (uint64_t k = b; k < e; ++k) {
                    (*loff[f])[k] = off[f][k];
                    (*llen[f])[k] = len[f][k];
                    (*nlen[f])[k] = nrm[f][k];
                    (*lflag[f])[k] = flg[f][k] & XMH_LINE_NORMAL;
                }
            });
        }
        return XMH_OK;
    } catch (const std::bad_alloc &) {
        return XMH_ERR_OOM;
    }
}

static inline char *put_line(char *dst, const char *src, uint32_t len, uint8_t flags)
{
    if (flags & XMH_LINE_NORMAL) {
        memcpy(dst, src, len);
        dst += len;
    } else {                                                          // '\t'.join(line.split())
        bool first = true;
        uint32_t i = 0;
        while (i < len) {
            if (is_ws((unsigned char)src[i])) { ++i; continue; }
            uint32_t b = i;
            while (i < len && !is_ws((unsigned char)src[i])) ++i;
            if (!first) *dst++ = '\t';
            memcpy(dst, src + b, i - b);
            dst += i - b;
            first = false;
        }
    }
    *dst++ = '\n';
    return dst;
}

int xmh_emit(xmh_parser *p, const char *buf1, const char *buf2, int paired, int bin,
             const uint32_t *idx, uint64_t n_idx, char *out, uint64_t out_cap, uint64_t *out_len)
{
    if (!p || !out_len || bin < 0 || bin > 5 || (n_idx && !idx)) return XMH_ERR_INVALID_ARG;
    const uint64_t n = p->sel[0].size();
    const bool use1 = (bin == 0 || bin == 2 || bin == 5 || bin == 4), use2 = (bin == 1 || bin == 3 || bin == 4);
    const uint32_t *nl1 = p->nlen1.data(), *nl2 = p->nlen2.data();
    auto unit_bytes = [&](uint32_t i) -> uint64_t {
        uint64_t b = 0;
        if (use1) b += (uint64_t)nl1[i] + 1 + (paired ? (uint64_t)nl1[i - 1] + 1 : 0);
        if (use2) b += (uint64_t)nl2[i] + 1 + (paired ? (uint64_t)nl2[i - 1] + 1 : 0);
        return b;
    };
    for (uint64_t u = 0; u < n_idx; ++u)
        if (idx[u] >= n || (paired && idx[u] == 0)) return XMH_ERR_INVALID_ARG;
    try {
        std::vector<uint64_t> start(n_idx + 1);
        uint64_t acc = 0;
        for (uint64_t u = 0; u < n_idx; ++u) { start[u] = acc; acc += unit_bytes(idx[u]); }
        start[n_idx] = acc;
        *out_len = acc;
        if (!out) return XMH_OK;
        if (out_cap < acc) return XMH_ERR_INVALID_ARG;
        const char *b[2] = {buf1, buf2};
        const uint64_t *lo[2] = {p->loff1.data(), p->loff2.data()};
        const uint32_t *ll[2] = {p->llen1.data(), p->llen2.data()};
        const uint8_t *lf[2] = {p->lflag1.data(), p->lflag2.data()};
        parallel_for(*p->pool, n_idx, [&](int, uint64_t ub, uint64_t ue) {
            for (uint64_t u = ub; u < ue; ++u) {
                char *d = out + start[u];
                const uint32_t i = idx[u];
                for (int f = 0; f < 2; ++f) {                          // file-1 lines, then file-2 lines (:439-444)
                    if (!(f == 0 ? use1 : use2)) continue;
                    if (paired) d = put_line(d, b[f] + lo[f][i - 1], ll[f][i - 1], lf[f][i - 1]);
                    d = put_line(d, b[f] + lo[f][i], ll[f][i], lf[f][i]);
                }
            }
        });
        return XMH_OK;
    } catch (const std::bad_alloc &) {
        return XMH_ERR_OOM;
    }
}

}  // extern "C"
