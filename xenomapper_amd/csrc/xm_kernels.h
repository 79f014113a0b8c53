// Internal launcher interface between xm_api.hip (C ABI, context, timing) and xm_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/xenomapper_hip.h"

#define XM_BLOCK 256          // threads per workgroup (4 wavefronts)
#define XM_TILE  4096         // records per K2 tile (16 category bytes per thread)
#define XM_MAX_CHUNKS 2048    // K2 workgroups: 8 per CU on 256 CUs, all resident

namespace xm {

struct ChunkPlan {
    uint32_t n_chunks;
    uint32_t tiles_per_chunk;
};

ChunkPlan plan_chunks(uint64_t n, uint32_t max_chunks);

void launch_classify_i32(hipStream_t st, uint32_t max_blocks, int mode, uint64_t n,
                         const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                         const uint64_t *unit_bits, int32_t m, uint8_t *code);
void launch_classify_f64(hipStream_t st, uint32_t max_blocks, int mode, uint64_t n,
                         const double *as1, const double *xs1, const double *as2, const double *xs2,
                         const uint64_t *unit_bits, double m, uint8_t *code);
void launch_hist(hipStream_t st, const ChunkPlan &p, int mode, uint64_t n, const uint8_t *code,
                 uint32_t *chunk_counts, uint64_t *counts);
void launch_scan(hipStream_t st, const ChunkPlan &p, const uint32_t *chunk_counts, uint32_t *chunk_off,
                 uint64_t *bin_offsets);
void launch_scatter(hipStream_t st, const ChunkPlan &p, int mode, uint64_t n, const uint8_t *code,
                    const uint32_t *chunk_off, uint32_t *idx_out);
void launch_cigar(hipStream_t st, uint32_t max_blocks, uint64_t n, const int32_t *nm, const uint32_t *cig_off,
                  const uint32_t *cig_oplen, int32_t *as_out, uint32_t *range_flag);

}  // namespace xm
