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
// K2c: at most this many waves per launch, each placing a run of consecutive granules.  Measured (profiles/r05_ab_k2c_waves.txt,
// 50 M interleaved pairs = 48.8 k granules): one granule per wave 54.6 us, 16 k waves 58, 8 k 62, 4 k 70, 2 k 102 -- fewer, longer-
// lived waves are SLOWER although a plain dword fill gains from exactly that shape (4.1 -> 6.0 TB/s, r05_probe_write_shapes.txt):
// a granule is a chain of dependent phases (load, ballots, stores), and what hides it is the number of granules in flight.
// So the default keeps one granule per wave up to 2^20 granules (2^31 records).
#ifndef XM_SCATTER_WAVES
#define XM_SCATTER_WAVES 1048576u
#endif
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
    // gran1 != 0: the launches cover the granules [gran0, gran1) only -- one chunk of a chunked xm_classify_place* call
    // (gran0 a multiple of XM_PART_GRAN; the chunks are launched in order, the last one ends at plan.n_gran)
    uint32_t gran0 = 0, gran1 = 0;
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

// the six-list output contract of the scatter (xm_classify_place*): one caller-allocated list per output bin
struct ListOut {
    uint32_t *p[7];             // lists of bins 0..5; [6] = units holding state 6 (binary64 columns only; may be null)
    uint32_t cap;               // capacity of every list, in entries
};

// segmented bin lists (xm_classify_runs*): where the runs kernel leaves its results
struct RunsOut {
    uint16_t *runs16;           // [n_gran][XM_GRAN]: granule-local record numbers of the granule's units, sorted by bin
    uint16_t *gran_counts16;    // [n_gran][8]: units per bin of every granule
    uint64_t *counts_rep;       // the context's [XM_COUNT_REPLICAS][64], all zero between calls
};
void launch_classify_runs_i32(hipStream_t st, int mode, uint64_t n,
                              const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                              const uint64_t *unit_bits, int32_t m, const RunsOut &ro);
void launch_classify_runs_f64(hipStream_t st, int mode, uint64_t n,
                              const double *as1, const double *xs1, const double *as2, const double *xs2,
                              const uint64_t *unit_bits, double m, const RunsOut &ro);
// adds up the category_counts replicas (left zeroed) and derives the eight list lengths
void launch_runs_finish(hipStream_t st, int mode, uint64_t *counts_rep, uint64_t *counts, uint64_t *n_out);

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
                    uint32_t *idx_out, uint32_t *part_tot, const ListOut *lists = nullptr, uint32_t gran0 = 0, uint32_t gran1 = 0);
// memory shape of the classify kernel without arithmetic (bench.py: the box's streaming ceiling); n a multiple of 4
void launch_stream_probe(hipStream_t st, uint64_t n, const int32_t *c0, const int32_t *c1, const int32_t *c2, const int32_t *c3,
                         uint8_t *out);
void launch_mate_correlate(hipStream_t st, uint64_t n, const double *track, uint32_t m, const double *density, double *out);
void launch_cigar(hipStream_t st, uint32_t max_blocks, uint64_t n, const int32_t *nm, const uint32_t *cig_off,
                  const uint32_t *cig_oplen, int32_t *as_out, uint32_t *range_flag);

}  // namespace xm
