// Internal launcher interface between xm_api.hip (C ABI, context, timing) and xm_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/xenomapper_hip.h"

#define XM_BLOCK 256          // threads per K2 workgroup (4 wavefronts = 4 granules)
#define XM_GRAN 2048          // records per granule: what one counting group (a counting K1 workgroup, a K2a wave) and one K2c wave own
#define XM_MAX_GRANULES ((uint32_t)((0xFFFFF000ull + XM_GRAN - 1) / XM_GRAN))
#define XM_PART_GRAN 1024u     // K2b: granules per part (first scan level; one K2b workgroup scans one part of one bin)
#define XM_PART_STRIDE 2112u   // row pitch of part_tot: >= XM_MAX_GRANULES / XM_PART_GRAN (2048) + slack
#define XM_PART_REPLICAS 64u  // the part totals are added into one of 64 copies (rows 8 x XM_PART_STRIDE words apart); K2b sums them
#define XM_COUNT_REPLICAS 64u // category_counts is added into one of 64 copies; K2b sums them
#define XM_CLASSIFY_BLOCK 512  // classify workgroup (tuned on the box with tools/tune_kernels.hip)
#ifndef XM_CIGAR_BLOCK
#define XM_CIGAR_BLOCK 256     // classify_cigar workgroup (128: 0.650, 256: 0.626, 512: 0.649, 1024: 0.754 ms per 50 M pairs)
#endif
#define XM_CIGP_BLOCK 512      // classify from packed CIGAR columns: workgroup = one granule (it counts)
#define XM_CLASSIFY_NT true    // non-temporal loads of the score columns in classify

namespace xm {

struct GranPlan {
    uint32_t n_gran;            // granules of XM_GRAN records
    uint32_t gran_stride;       // row pitch of gran_counts / gran_off ([bin][granule] layout)
};

// where the counting side (fused K1, or K2a) leaves its results
struct CountPlan {
    GranPlan plan = {0, 0};
    uint32_t *gran_counts = nullptr;   // [8][gran_stride]
    uint32_t *part_tot = nullptr;      // [XM_PART_REPLICAS][8][XM_PART_STRIDE]: K2b's first level (units per bin and part), all zero between calls
    uint8_t *bins4 = nullptr;          // where a counting classify kernel writes the compact category stream, or null
    uint64_t *counts_rep = nullptr;    // [XM_COUNT_REPLICAS][64], all zero between calls
};

GranPlan plan_granules(uint64_t n);

// packed CIGAR columns of one species (include/xenomapper_hip.h, "packed CIGAR columns")
struct CigCols {
    const int32_t *nm;          // [n] NM value, XM_ABSENT = no NM field
    const int32_t *xs;          // [n]
    const uint8_t *cnt;         // [n] ops of the record; 255 = 255 or more, the record's ops are followed by a trailer word
    const uint32_t *tile;       // [ceil(n / 256) + 1] where the ops of each 256-record tile begin; last = length of ops
    const uint32_t *ops;        // BAM-style len << 4 | op
};

// ---- single-pass classify + place (xm_classify_place*) ---------------------------------------------------------
#define XM_PLACE_S 32u          // granules per block of the look-back (block sums carry an arrival count up to this)
#define XM_PLACE_WIN 24u        // blocks of look-back window
#define XM_PLACE_PROBES 6u      // prefix records inside the window (every 4th block publishes one)
#ifndef XM_PLACE_SPIN_LIMIT
#define XM_PLACE_SPIN_LIMIT (1u << 17)   // polls before a workgroup gives up (each is a memory round trip: >= 0.1 s)
#endif
#define XM_PLACE_BLOCKS(n_gran) (((uint64_t)(n_gran) + XM_PLACE_S - 1u) / XM_PLACE_S)
// Where the hand-off records live.  Everything a workgroup reads in one look-back comes from different rows, the rows far
// apart and not a power of two apart, so that the polls of the ~800 resident workgroups spread over the memory channels
// instead of queueing on the few lines a contiguous window would occupy (XM_PLACE_SPREAD=0: contiguous, for A/B runs).
#ifndef XM_PLACE_SPREAD
#define XM_PLACE_SPREAD 1
#endif
#define XM_PLACE_GD_ROWS 64u
#define XM_PLACE_GD_PITCH ((XM_MAX_GRANULES + 64u) / XM_PLACE_GD_ROWS + 9u)                       // entries of 4 words
#define XM_PLACE_BS_ROWS 32u
#define XM_PLACE_BS_PITCH (((uint32_t)XM_PLACE_BLOCKS(XM_MAX_GRANULES) + 64u) / XM_PLACE_BS_ROWS + 9u)   // entries of 4 words
#define XM_PLACE_BP_ROWS 8u
#define XM_PLACE_BP_PITCH (((uint32_t)XM_PLACE_BLOCKS(XM_MAX_GRANULES) / 4u + 64u) / XM_PLACE_BP_ROWS + 5u)  // entries of 8 words
#if XM_PLACE_SPREAD
#define XM_PLACE_GD_AT(g) ((uint64_t)(((g) % XM_PLACE_GD_ROWS) * (uint64_t)XM_PLACE_GD_PITCH + (g) / XM_PLACE_GD_ROWS) * 4u)
#define XM_PLACE_BS_AT(b) ((uint64_t)(((b) % XM_PLACE_BS_ROWS) * (uint64_t)XM_PLACE_BS_PITCH + (b) / XM_PLACE_BS_ROWS) * 4u)
#define XM_PLACE_BP_AT(q) ((uint64_t)(((q) % XM_PLACE_BP_ROWS) * (uint64_t)XM_PLACE_BP_PITCH + (q) / XM_PLACE_BP_ROWS) * 8u)
#else
#define XM_PLACE_GD_AT(g) ((uint64_t)(g) * 4u)
#define XM_PLACE_BS_AT(b) ((uint64_t)(b) * 4u)
#define XM_PLACE_BP_AT(q) ((uint64_t)(q) * 8u)
#endif
#define XM_PLACE_GD_WORDS ((uint64_t)XM_PLACE_GD_ROWS * XM_PLACE_GD_PITCH * 4u)
#define XM_PLACE_BS_WORDS ((uint64_t)XM_PLACE_BS_ROWS * XM_PLACE_BS_PITCH * 4u)
#define XM_PLACE_BP_WORDS ((uint64_t)XM_PLACE_BP_ROWS * XM_PLACE_BP_PITCH * 8u)
#define XM_PLACE_RING 8192u        // granules of the bins ring (1 KB each): > lag + twice the resident workgroups
#ifndef XM_PLACE_LAG
#define XM_PLACE_LAG 2560u         // workgroup i places granule i - lag: ~15 us of arrivals at 50 M pairs per 0.3 ms
#endif
#define XM_PLACE_DONE_WORDS 256u   // arrival counters of finished workgroups: workgroup g adds to word g % 256
#ifndef XM_PLACE_SLEEP
#define XM_PLACE_SLEEP 4           // s_sleep argument between two polls (x 64 clocks)
#endif

// workspace + outputs of the placing kernels (passed by value)
struct PlaceSink {
    uint32_t *list[7];                 // the six bin lists, [6] = units holding state 6 (binary64 only; may be null)
    uint32_t cap;                      // capacity of every list, in entries
    uint32_t n_gran;
    uint32_t lag;                      // min(XM_PLACE_LAG, n_gran); the grid is n_gran + lag workgroups
    uint16_t *ring;                    // [XM_PLACE_RING][XM_GRAN / 4]: the bins of the granules in flight, a nibble per record
    unsigned long long *gdesc;         // [granule][4]   {epoch, 2 x 16-bit counts}
    unsigned long long *bsum;          // [block][4]     {arrivals, 2 x 24-bit sums}, all zero between calls
    unsigned long long *bpre;          // [block / 4][8] {epoch, inclusive prefix}
    uint32_t *done1;                   // [XM_PLACE_DONE_WORDS] workgroups that have finished (word g % 256), all zero between calls
    uint32_t *ctl;                     // [0] epoch  [2] a workgroup gave up (sticky)  [3] high-water granule count
    unsigned long long *n_out;         // [8]: list lengths 0..5, [6] units holding state 6, [7] all units (~0: gave up)
    unsigned long long *counts;        // [64] category_counts
    unsigned long long *counts_rep;    // [XM_COUNT_REPLICAS][64], all zero between calls
    unsigned long long *trace;         // tuning builds (-DXM_PLACE_TRACE): [granule][8] timestamps (100 MHz) and poll data; else unused
};

void launch_classify_place_i32(hipStream_t st, int mode, uint64_t n,
                               const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                               const uint64_t *unit_bits, int32_t m, uint8_t *code, const PlaceSink &ps);
void launch_classify_place_f64(hipStream_t st, int mode, uint64_t n,
                               const double *as1, const double *xs1, const double *as2, const double *xs2,
                               const uint64_t *unit_bits, double m, uint8_t *code, const PlaceSink &ps);

// cp != nullptr: the fused form, the kernel also counts (granule = its workgroup)
void launch_classify_i32(hipStream_t st, int mode, uint64_t n,
                         const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                         const uint64_t *unit_bits, int32_t m, uint8_t *code, const CountPlan *cp);
void launch_classify_f64(hipStream_t st, int mode, uint64_t n,
                         const double *as1, const double *xs1, const double *as2, const double *xs2,
                         const uint64_t *unit_bits, double m, uint8_t *code, const CountPlan *cp);
void launch_classify_cigar(hipStream_t st, int mode, uint64_t n,
                           const int32_t *nm1, const uint32_t *off1, const uint32_t *ops1, const int32_t *xs1,
                           const int32_t *nm2, const uint32_t *off2, const uint32_t *ops2, const int32_t *xs2,
                           const uint64_t *unit_bits, int32_t m, uint8_t *code, uint32_t *range_flag);
// the counting form only (workgroup = granule): category bytes and/or the compact stream (cp.bins4), counts, gran_counts
void launch_classify_cigp(hipStream_t st, int mode, uint64_t n, const CigCols &s1, const CigCols &s2,
                          const uint64_t *unit_bits, int32_t m, uint8_t *code, uint32_t *range_flag, const CountPlan &cp);
void launch_hist(hipStream_t st, int mode, uint64_t n, const uint8_t *code, const CountPlan &cp);
void launch_scan(hipStream_t st, const CountPlan &cp, uint32_t *gran_off, uint64_t *bin_totals, uint64_t *counts);
// code_is_bins4: `code` is the compact category stream of a counting classify kernel, not category bytes
// also zeroes the part totals K2b has consumed
void launch_scatter(hipStream_t st, const GranPlan &p, int mode, uint64_t n, const uint8_t *code, bool code_is_bins4,
                    const uint32_t *gran_counts, const uint32_t *gran_off, const uint64_t *bin_totals, uint64_t *bin_offsets,
                    uint32_t *idx_out, uint32_t *part_tot);
void launch_mate_correlate(hipStream_t st, uint64_t n, const double *track, uint32_t m, const double *density, double *out);
void launch_cigar(hipStream_t st, uint32_t max_blocks, uint64_t n, const int32_t *nm, const uint32_t *cig_off,
                  const uint32_t *cig_oplen, int32_t *as_out, uint32_t *range_flag);

}  // namespace xm
