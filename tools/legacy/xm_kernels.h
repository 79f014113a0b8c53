// Internal launcher interface between xm_api.hip (C ABI, context, timing) and xm_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/xenomapper_hip.h"  // tools/legacy -> repo/include

#define XM_BLOCK 256          // threads per workgroup (4 wavefronts)
#define XM_WTILE 1024         // K2: records per wave tile (16 category bytes per lane)
#ifndef XM_K
#define XM_K     2            // K2: wave tiles per wave (all held in registers)
#endif
#define XM_CHUNK (XM_K * XM_WTILE * (XM_BLOCK / 64))   // K2: records per workgroup = 32768
#define XM_MAX_CHUNKS ((uint32_t)((0xFFFFF000ull + XM_CHUNK - 1) / XM_CHUNK))
#define XM_COUNT_REPLICAS 64u // K2a adds category_counts into one of 64 copies; K2b sums them
#define XM_CLASSIFY_BLOCK 512  // classify workgroup (tuned on the box with tools/tune_kernels.hip)
#ifndef XM_CIGAR_BLOCK
#define XM_CIGAR_BLOCK 256     // classify_cigar workgroup (128: 0.650, 256: 0.626, 512: 0.649, 1024: 0.754 ms per 50 M pairs)
#endif
#define XM_CLASSIFY_NT true    // non-temporal loads of the score columns in classify

namespace xm {

struct ChunkPlan {
    uint32_t n_chunks;
    uint32_t chunk_stride;      // row pitch of chunk_counts / chunk_off ([bin][chunk] layout)
};

ChunkPlan plan_chunks(uint64_t n);

void launch_classify_i32(hipStream_t st, int mode, uint64_t n,
                         const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                         const uint64_t *unit_bits, int32_t m, uint8_t *code);
void launch_classify_f64(hipStream_t st, int mode, uint64_t n,
                         const double *as1, const double *xs1, const double *as2, const double *xs2,
                         const uint64_t *unit_bits, double m, uint8_t *code);
void launch_classify_cigar(hipStream_t st, int mode, uint64_t n,
                           const int32_t *nm1, const uint32_t *off1, const uint32_t *ops1, const int32_t *xs1,
                           const int32_t *nm2, const uint32_t *off2, const uint32_t *ops2, const int32_t *xs2,
                           const uint64_t *unit_bits, int32_t m, uint8_t *code, uint32_t *range_flag);
void launch_hist(hipStream_t st, const ChunkPlan &p, int mode, uint64_t n, const uint8_t *code,
                 uint32_t *chunk_counts, uint64_t *counts_rep);
void launch_scan(hipStream_t st, const ChunkPlan &p, const uint32_t *chunk_counts, uint32_t *chunk_off,
                 uint64_t *bin_totals, uint64_t *counts_rep, uint64_t *counts);
void launch_scatter(hipStream_t st, const ChunkPlan &p, int mode, uint64_t n, const uint8_t *code,
                    const uint32_t *chunk_off, const uint64_t *bin_totals, uint64_t *bin_offsets, uint32_t *idx_out);
void launch_mate_correlate(hipStream_t st, uint64_t n, const double *track, uint32_t m, const double *density, double *out);
void launch_cigar(hipStream_t st, uint32_t max_blocks, uint64_t n, const int32_t *nm, const uint32_t *cig_off,
                  const uint32_t *cig_oplen, int32_t *as_out, uint32_t *range_flag);

}  // namespace xm
