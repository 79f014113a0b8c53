/*
 * xenomapper_hip.h -- C ABI of the MI355X (gfx950) xenograft read classifier.
 *
 * The reference (genomematt/xenomapper v1.0.2) is pure Python and has no FFI; its operator
 * interface for this path is the trio of main loops plus their pluggable score extractor
 * (/root/reference/xenomapper/xenomapper.py:291-299, :354-362, :456-464, tag_func :176-256,
 * get_mapping_state :258).  This header is what a maintainer's ctypes binding for those
 * functions would bind (INTEGRATION.md shows the stub).  Each entry point names the
 * reference code it replaces.
 *
 * Conventions: extern "C", plain pointers and sizes, no exceptions across the boundary,
 * int status (0 = ok, negative = error, xm_strerror() gives text), caller-allocated
 * buffers, the library never frees caller memory.  A context belongs to one device; use one
 * context per thread.  There is no CPU fallback: without a usable gfx950 device
 * xm_ctx_create() fails with XM_ERR_NO_DEVICE and nothing else can be called.
 *
 * Data model (one "record" = one SAM line index i, present in both the primary- and the
 * secondary-species file; columns are structure-of-arrays, one element per record):
 *   as1, xs1, as2, xs2   scores of record i in species 1 / 2.  int32 columns use
 *                        XM_ABSENT (INT32_MIN) for "tag absent" = the reference's
 *                        float('-inf') (xenomapper.py:187-188); f64 columns use -inf itself.
 *   unit_bits            packed little-endian bit mask, bit (i & 63) of word (i >> 6):
 *                        record i closes a unit.  Paired modes: name[i] == name[i-1]
 *                        (xenomapper.py:402-405); single-end: record i was yielded by the
 *                        reader (xenomapper.py:321).  ceil(n/64) words.
 *   code                 one byte per record: XM_NO_UNIT (0xFF) or the unit's category
 *                        code -- state (single-end) or fwd*8 + rev (paired), states 0..5 =
 *                        primary_specific, secondary_specific, primary_multi,
 *                        secondary_multi, unresolved, unassigned (priority order of
 *                        xenomapper.py:364-367); state 6 = the reference's RuntimeError
 *                        fall-through (xenomapper.py:289; reachable with NaN only).
 *   counts[64]           category_counts (xenomapper.py:330, :420, :520) indexed by code.
 */
#ifndef XENOMAPPER_HIP_H
#define XENOMAPPER_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XM_ABI_VERSION 6

#define XM_ABSENT   INT32_MIN
#define XM_NO_UNIT  0xFFu

/* loop selection (which reference main loop the call stands for) */
#define XM_MODE_SE              0   /* main_single_end               xenomapper.py:291-352 */
#define XM_MODE_PE_LIBERAL      1   /* main_paired_end               xenomapper.py:354-454 */
#define XM_MODE_PE_CONSERVATIVE 2   /* conservative_main_paired_end  xenomapper.py:456-556 */

/* status codes */
#define XM_OK               0
#define XM_ERR_INVALID_ARG (-1)
#define XM_ERR_NO_DEVICE   (-2)   /* no gfx950 device / HIP runtime unusable            */
#define XM_ERR_HIP         (-3)   /* a HIP call failed; xm_last_hip_error() has details */
#define XM_ERR_OOM         (-4)
#define XM_ERR_RANGE       (-5)   /* a CIGAR-derived score does not fit the int32 column */
#define XM_ERR_RCCL        (-6)   /* librccl missing or an RCCL call failed; xm_last_hip_error() has details */

typedef struct xm_ctx xm_ctx;

/* ---- context ------------------------------------------------------------------------- */
int         xm_abi_version(void);
const char *xm_strerror(int status);
/* Text of the last failing HIP call on this context (empty string if none); with ctx == NULL,
 * why the last xm_ctx_create() failed. */
const char *xm_last_hip_error(const xm_ctx *ctx);
/* Binds to HIP device `device_id`; fails with XM_ERR_NO_DEVICE if it is not a gfx950 part. */
int xm_ctx_create(int device_id, xm_ctx **out);
int xm_ctx_destroy(xm_ctx *ctx);
/* Number of compute units and name of the bound device (diagnostics). */
int xm_ctx_device_info(const xm_ctx *ctx, int *n_cu, char *name, size_t name_len);

/* ---- host-buffer entry points (H2D copy, kernels, D2H copy; blocking) ----------------- */

/*
 * Replaces the score->state->pair-combination part of the three main loops
 * (xenomapper.py:323-330, :408-420, :508-520; get_mapping_state :258-289) for n_records
 * records at once.  min_score_floor = floor(min_score) clamped to int32 (-inf -> INT32_MIN,
 * +inf -> INT32_MAX).  code_out: n_records bytes.  counts: 64 words (may be NULL).
 */
int xm_classify(xm_ctx *ctx, int mode, uint64_t n_records,
                const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                const uint64_t *unit_bits, int32_t min_score_floor,
                uint8_t *code_out, uint64_t counts[64]);

/* Same in the reference's own arithmetic (binary64): for non-integral scores, scores outside
 * int32, or a NaN min_score.  Absent = -inf. */
int xm_classify_f64(xm_ctx *ctx, int mode, uint64_t n_records,
                    const double *as1, const double *xs1, const double *as2, const double *xs2,
                    const uint64_t *unit_bits, double min_score,
                    uint8_t *code_out, uint64_t counts[64]);

/*
 * Replaces get_cigarbased_AS_tag(tag='AS') (xenomapper.py:228-256) on pre-parsed columns:
 * nm[i] = NM value or XM_ABSENT when no field contains 'NM' (:247-249); CIGAR as CSR,
 * cig_off[n_records+1] into cig_oplen[], each op packed BAM-style len<<4 | op with ops
 * "MIDNSHP=X" = 0..8 (only I, D, S contribute, :252-255).  as_out[i] = XM_ABSENT or
 * -6*NM - 5*(#I+#D) - 3*(sumI+sumD) - 2*sumS.  XM_ERR_RANGE if a score leaves int32.
 */
int xm_cigar_scores(xm_ctx *ctx, uint64_t n_records, const int32_t *nm,
                    const uint32_t *cig_off, const uint32_t *cig_oplen, int32_t *as_out);

/*
 * The --cigar_scores path in one call (main loop + tag_func = get_cigarbased_AS_tag, xenomapper.py:684-685,
 * :228-256): AS of both species is synthesised from NM + CIGAR inside the classify kernel (never stored),
 * XS comes from the xs1/xs2 columns (the real XS tag when present, :245-246).  Same outputs as xm_classify.
 * XM_ERR_RANGE if a synthesised score leaves int32.
 */
int xm_classify_cigar(xm_ctx *ctx, int mode, uint64_t n_records,
                      const int32_t *nm1, const uint32_t *cig_off1, const uint32_t *cig_oplen1, const int32_t *xs1,
                      const int32_t *nm2, const uint32_t *cig_off2, const uint32_t *cig_oplen2, const int32_t *xs2,
                      const uint64_t *unit_bits, int32_t min_score_floor,
                      uint8_t *code_out, uint64_t counts[64]);

/*
 * Replaces the bin routing of the main loops (the if/elif chains xenomapper.py:332-350,
 * :423-448, :521-550) and category_counts: a stable split of the unit indices by output bin.
 * idx_out (capacity >= number of units, n_records always suffices) receives, bin after bin,
 * the record index of each unit in input order; bin b is idx_out[bin_offsets[b] ..
 * bin_offsets[b+1]) for b = 0..5; slot 6 collects units holding state 6; bin_offsets[7] =
 * number of units.  counts (may be NULL) = category_counts indexed by code.
 * Emission rule for the caller: bins 0,2,5 -> lines of file 1; 1,3 -> file 2; 4 -> file 1's
 * lines then file 2's; paired units cover records idx-1 and idx.
 */
int xm_compact(xm_ctx *ctx, int mode, uint64_t n_records, const uint8_t *code,
               uint32_t *idx_out, uint64_t bin_offsets[8], uint64_t counts[64]);

/*
 * One whole main loop per call (xenomapper.py:321-350, :398-452, :498-554): what xm_classify + xm_compact give, from
 * one fused pass -- the category bytes stay on the device between the two stages and category_counts come from the
 * kernels.  code_out may be NULL when the caller only needs the bins (the emission rule needs nothing else);
 * idx_out, bin_offsets and counts as for xm_compact.
 */
int xm_classify_compact(xm_ctx *ctx, int mode, uint64_t n_records,
                        const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                        const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out,
                        uint32_t *idx_out, uint64_t bin_offsets[8], uint64_t counts[64]);

int xm_classify_compact_f64(xm_ctx *ctx, int mode, uint64_t n_records,
                            const double *as1, const double *xs1, const double *as2, const double *xs2,
                            const uint64_t *unit_bits, double min_score, uint8_t *code_out,
                            uint32_t *idx_out, uint64_t bin_offsets[8], uint64_t counts[64]);

int xm_classify_compact_cigar(xm_ctx *ctx, int mode, uint64_t n_records,
                              const int32_t *nm1, const uint32_t *cig_off1, const uint32_t *cig_oplen1, const int32_t *xs1,
                              const int32_t *nm2, const uint32_t *cig_off2, const uint32_t *cig_oplen2, const int32_t *xs2,
                              const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out,
                              uint32_t *idx_out, uint64_t bin_offsets[8], uint64_t counts[64]);

/*
 * The same main loop with the reference's own output shape: six independent sinks (the loops print into six file objects,
 * xenomapper.py:332-350, :423-448, :521-550; SURVEY 8b (4): idx_out[6], n_out[6]).  idx_out: six host buffers of
 * list_capacity entries each, one per output bin in the priority order above; list b receives the record indices of the
 * units routed to bin b in input order, n_out[b] its length; n_out[6] = units holding state 6 (binary64 columns with NaN
 * only; listed in idx_state6 when given), n_out[7] = all units.  A list longer than list_capacity is truncated, n_out
 * reports the full length (the number of units never exceeds n_records).  code_out and counts may be NULL.
 */
int xm_classify_place(xm_ctx *ctx, int mode, uint64_t n_records,
                      const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                      const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out,
                      uint32_t *const idx_out[6], uint64_t list_capacity, uint64_t n_out[8], uint64_t counts[64]);

int xm_classify_place_f64(xm_ctx *ctx, int mode, uint64_t n_records,
                          const double *as1, const double *xs1, const double *as2, const double *xs2,
                          const uint64_t *unit_bits, double min_score, uint8_t *code_out,
                          uint32_t *const idx_out[6], uint32_t *idx_state6, uint64_t list_capacity,
                          uint64_t n_out[8], uint64_t counts[64]);

/*
 * xenomappability (the reference's experimental companion tool): replaces the inner loop of
 * Mappability.single_end_to_paired (/root/reference/xenomapper/mappability.py:94-124) for one chromosome:
 * out[i] = 1.0 where track[i] == 1, else sum_j track[i+j] * density[j], j < min(m, n - i), accumulated left to right
 * in binary64 with separately rounded multiply and add (bit-identical to the Python loop).
 */
int xm_mate_correlate(xm_ctx *ctx, uint64_t n, const double *track, uint64_t m, const double *density, double *out);

/*
 * Optional: page-lock caller memory that is handed to the host-buffer entry points again and again (the column
 * buffers of a parser, the index buffer of a writer).  From registered memory the copies run as direct DMA at the
 * PCIe rate; from ordinary pageable memory the HIP runtime stages them through its own pinned buffers (~40 GB/s
 * instead of ~55 GB/s on this platform).  Registering costs about as much as one copy of the buffer, so it only pays
 * for buffers that live across calls.  Unregister before freeing the memory.  Results never depend on it.
 */
int xm_host_register(xm_ctx *ctx, void *ptr, size_t bytes);
int xm_host_unregister(xm_ctx *ctx, void *ptr);
/*
 * The page-locked host memory this library holds right now, process-wide (ABI 6): `allocated` = what the front ends
 * (xm_strip_*, xm_bamdev_*: staging windows, inflated copies, line tables, text) got from hipHostMalloc and have not given
 * back, `registered` = caller memory locked through xm_host_register and not yet unregistered, `peak` = the largest sum of
 * the two so far.  Any pointer may be NULL.  xm_strip_destroy / xm_bamdev_destroy give everything of theirs back; a process
 * that keeps front ends alive between jobs (the Python package's process-wide ones: xenomapper_amd.xenomapper.
 * release_buffers()) reads here what that costs.
 */
int xm_pinned_bytes(uint64_t *allocated, uint64_t *registered, uint64_t *peak);

/* ---- device-resident entry points (asynchronous on `stream`) -------------------------- */
/*
 * All pointers are device memory of the context's device; `stream` is a hipStream_t passed
 * as void* (NULL = the default stream).  Nothing is synchronised or allocated: the calls
 * only enqueue work (graph-capturable).  Column base pointers must be 16-byte aligned and
 * code 4-byte aligned (any hipMalloc / torch allocation is).  n_records <= XM_MAX_RECORDS.  The calling
 * thread's current HIP device must be the context's device (XM_ERR_INVALID_ARG otherwise).
 */
#define XM_MAX_RECORDS 0xFFFFF000ull

int xm_classify_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                    const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                    const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out);

int xm_classify_f64_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                        const double *as1, const double *xs1, const double *as2, const double *xs2,
                        const uint64_t *unit_bits, double min_score, uint8_t *code_out);

int xm_classify_cigar_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                          const int32_t *nm1, const uint32_t *cig_off1, const uint32_t *cig_oplen1, const int32_t *xs1,
                          const int32_t *nm2, const uint32_t *cig_off2, const uint32_t *cig_oplen2, const int32_t *xs2,
                          const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out,
                          uint32_t *range_flag);

/* range_flag: one device uint32, set non-zero when a score left int32 (may be NULL). */
int xm_cigar_scores_dev(xm_ctx *ctx, void *stream, uint64_t n_records, const int32_t *nm,
                        const uint32_t *cig_off, const uint32_t *cig_oplen, int32_t *as_out,
                        uint32_t *range_flag);

int xm_mate_correlate_dev(xm_ctx *ctx, void *stream, uint64_t n, const double *track, uint64_t m,
                          const double *density, double *out);

/* bin_offsets: 8 device uint64; counts: 64 device uint64 (both overwritten).
 * The compaction workspace (per-granule counts and offsets, the count replicas, the part totals) belongs to the context:
 * ONE xm_compact_dev / xm_classify_compact*_dev / xm_classify_place*_dev call can be in flight per context at a time.
 * Calls on one stream are ordered anyway; a call issued on another stream than the previous one is put behind
 * everything enqueued on that stream so far by the library (an event recorded there at that moment: the previous stream
 * must still exist -- see xm_workspace_release); calls captured into graphs are ordered by the graph's own edges only.
 * For real concurrency use one context per stream. */
int xm_compact_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records, const uint8_t *code,
                   uint32_t *idx_out, uint64_t *bin_offsets, uint64_t *counts);

/*
 * One whole main loop (xenomapper.py:321-350, :398-452, :498-554) on device-resident columns: xm_classify_dev
 * followed by xm_compact_dev, fused -- the classify kernel counts its units per category and per output bin while
 * the categories are still in registers.  Outputs as xm_classify_dev + xm_compact_dev.  Same one-in-flight rule.
 * The per-record output comes in two forms, at least one of which must be given:
 *   code_out  n_records bytes, the category byte of xm_classify_dev (state, or fwd*8 + rev; 0xFF = closes no unit);
 *   bins4     XM_BINS4_BYTES(n_records) bytes, the compact category stream: the OUTPUT BIN of every record as a
 *             nibble (0..5 = the six bins in the priority order above, 6 = a unit holding state 6, 7 = closes no
 *             unit), record r in bits 4 (r & 1) .. 4 (r & 1) + 3 of byte r >> 1 -- one byte per read pair.  The
 *             buffer is written in whole 1024-byte blocks (2048 records), hence the rounded-up size.
 * With bins4 the scatter reads the compact stream (half the bytes, no byte -> bin table) and code_out is an optional
 * extra; without it the scatter reads code_out.  Both 16-byte aligned.
 */
#define XM_BINS4_BYTES(n_records) ((((uint64_t)(n_records) + 2047u) / 2048u) * 1024u)

int xm_classify_compact_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                            const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                            const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint8_t *bins4,
                            uint32_t *idx_out, uint64_t *bin_offsets, uint64_t *counts);

int xm_classify_compact_f64_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                                const double *as1, const double *xs1, const double *as2, const double *xs2,
                                const uint64_t *unit_bits, double min_score, uint8_t *code_out, uint8_t *bins4,
                                uint32_t *idx_out, uint64_t *bin_offsets, uint64_t *counts);

/* The --cigar_scores form on CSR columns: code_out required (this kernel does not count; the compaction reads its bytes).
 * The faster form is xm_classify_compact_cigar_packed_dev below. */
int xm_classify_compact_cigar_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                                  const int32_t *nm1, const uint32_t *cig_off1, const uint32_t *cig_oplen1, const int32_t *xs1,
                                  const int32_t *nm2, const uint32_t *cig_off2, const uint32_t *cig_oplen2, const int32_t *xs2,
                                  const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out,
                                  uint32_t *range_flag, uint32_t *idx_out, uint64_t *bin_offsets, uint64_t *counts);

/*
 * Packed CIGAR columns -- the layout the --cigar_scores kernel reads fastest (get_cigarbased_AS_tag, xenomapper.py:228-256;
 * the CIGAR scan of :251 as pre-parsed ops).  Per species:
 *   nm, xs        as above (int32 per record)
 *   cig_cnt       one BYTE per record: its number of CIGAR ops; 255 = "255 or more"
 *   cig_tile      XM_CIG_TILES(n_records) + 1 words: cig_tile[t] = where in cig_oplen the ops of records
 *                 [256 t, 256 t + 256) begin (records in order, ops of a record contiguous); the last entry = length of
 *                 cig_oplen in words
 *   cig_oplen     the packed ops (len << 4 | op); a record with cig_cnt 255 is followed by ONE trailer word
 *                 n_ops << 4 | 15 (op code 15 scores nothing), so its begin can be found from its end
 * i.e. 9 + 1/64 bytes per record and species next to the ops, against the 12 of the CSR form (4-byte offsets), and the
 * position of a tile's ops is known without reading a per-record column.  xm_cigar_pack() converts CSR columns (host
 * memory, no device needed): it fills cig_cnt[n_records] and cig_tile[XM_CIG_TILES(n_records) + 1], stores the length
 * of the packed op array in *n_ops_packed and, when ops_packed is not NULL (capacity in words: ops_capacity), writes the
 * packed op array.  When *n_ops_packed == cig_off[n_records] no record needed a trailer and the packed op array IS
 * cig_oplen: call with ops_packed = NULL first and skip the copy.  cig_off[0] must be 0 (XM_ERR_INVALID_ARG otherwise: the
 * tile positions count from the start of the packed array).  XM_ERR_RANGE: a record with 2^28 ops or more, or more
 * than 2^32 - 1 packed ops.
 */
#define XM_CIG_TILE 256u
#define XM_CIG_TILES(n_records) (((uint64_t)(n_records) + 255u) / 256u)
int xm_cigar_pack(uint64_t n_records, const uint32_t *cig_off, const uint32_t *cig_oplen,
                  uint8_t *cig_cnt, uint32_t *cig_tile, uint32_t *ops_packed, uint64_t ops_capacity,
                  uint64_t *n_ops_packed);

/*
 * One whole main loop with tag_func = get_cigarbased_AS_tag (xenomapper.py:684-685, :228-256, and the loop bodies
 * :321-350, :398-452, :498-554) on device-resident packed CIGAR columns: AS of both species is synthesised inside the
 * classify kernel, which also counts (category_counts, per-granule bin counts) and writes the per-record output in
 * either form (code_out and/or bins4, at least one -- as xm_classify_compact_dev); then scan + scatter.  range_flag:
 * one device uint32, set non-zero when a synthesised score left int32 (may be NULL).  nm/xs 16-byte aligned, the
 * other columns 4-byte aligned.  No read goes past cig_tile[last] words of cig_oplen whatever the columns hold.
 * The host-buffer entry points xm_classify_cigar / xm_classify_compact_cigar take CSR columns, pack them and run this.
 */
int xm_classify_compact_cigar_packed_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                                         const int32_t *nm1, const uint8_t *cig_cnt1, const uint32_t *cig_tile1,
                                         const uint32_t *cig_oplen1, const int32_t *xs1,
                                         const int32_t *nm2, const uint8_t *cig_cnt2, const uint32_t *cig_tile2,
                                         const uint32_t *cig_oplen2, const int32_t *xs2,
                                         const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint8_t *bins4,
                                         uint32_t *range_flag, uint32_t *idx_out, uint64_t *bin_offsets, uint64_t *counts);

/*
 * The fused main loop with the six-list output contract of SURVEY 8b (4) (see xm_classify_place above) on device-resident
 * columns: idx_out is a HOST array of six DEVICE pointers (4-byte aligned; 16-byte alignment lets the single-end form
 * store 16 bytes at a time), each list with room for list_capacity entries; n_out = 8 device uint64 (six list lengths,
 * units holding state 6, all units), counts = 64 device uint64.  code_out / bins4 as for xm_classify_compact_dev (at
 * least one).  Same three launches and the same one-in-flight rule as xm_classify_compact_dev: the classify kernel
 * counts per bin and granule, the scan gives every granule its place in each bin, and the scatter writes bin b's units
 * to list b -- a unit's place no longer depends on the totals of the bins in front of its own.
 * (A single-kernel form of this call -- classify and place in one pass behind a decoupled look-back -- was built and
 * measured in round 4 and lost, 1.0-1.7 ms against 0.35 ms per 50 M pairs: DESIGN.md, profiles/r04_place_*.txt.)
 */
int xm_classify_place_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                          const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                          const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint8_t *bins4,
                          uint32_t *const idx_out[6], uint64_t list_capacity, uint64_t *n_out, uint64_t *counts);

int xm_classify_place_f64_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                              const double *as1, const double *xs1, const double *as2, const double *xs2,
                              const uint64_t *unit_bits, double min_score, uint8_t *code_out, uint8_t *bins4,
                              uint32_t *const idx_out[6], uint32_t *idx_state6, uint64_t list_capacity,
                              uint64_t *n_out, uint64_t *counts);

int xm_classify_place_cigar_packed_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                                       const int32_t *nm1, const uint8_t *cig_cnt1, const uint32_t *cig_tile1,
                                       const uint32_t *cig_oplen1, const int32_t *xs1,
                                       const int32_t *nm2, const uint8_t *cig_cnt2, const uint32_t *cig_tile2,
                                       const uint32_t *cig_oplen2, const int32_t *xs2,
                                       const uint64_t *unit_bits, int32_t min_score_floor, uint8_t *code_out, uint8_t *bins4,
                                       uint32_t *range_flag, uint32_t *const idx_out[6], uint64_t list_capacity,
                                       uint64_t *n_out, uint64_t *counts);

/*
 * The fused main loop (xenomapper.py:321-350, :398-452, :498-554) with SEGMENTED bin lists: ONE launch (plus a
 * 1-workgroup launch for category_counts), no scan, no scatter, nothing written that is read again.  The records are taken
 * in granules of XM_RUNS_GRAN = 2048; the workgroup that classifies a granule also sorts the granule's units by output bin
 * (on chip) and writes
 *   runs16       XM_RUNS16_ENTRIES(n_records) uint16: entries [2048 g, 2048 g + units(g)) = the units of granule g as record
 *                numbers counted from the granule's first record (0..2047), sorted by bin, input order inside a bin;
 *                entries past units(g) are not written
 *   gran_counts  8 uint16 per granule: gran_counts[8 g + b] = units of granule g in bin b (b = 0..5 the six bins in the
 *                priority order above, 6 = units holding state 6 (binary64 columns with NaN only), 7 = always 0)
 *   n_out        8 device uint64: the six list lengths, units holding state 6, all units
 *   counts       64 device uint64: category_counts
 * so that list b in input order is, for g = 0, 1, ...: 2048 g + runs16[2048 g + s .. 2048 g + s + gran_counts[8 g + b]) with
 * s = the granule's counts of the bins in front of b.  A consumer that emits per bin walks the granules once per bin
 * (xm_runs_expand does exactly that into a flat list; the emission rule is xm_compact's).  Why this shape: a unit's place
 * in a FLAT list needs the totals of all granules in front of it -- a prefix over the whole input, i.e. the scan and the
 * second pass of xm_classify_compact_dev (a look-back inside one launch lost by 3-7x in round 4) -- while its
 * place inside its own granule's run needs nothing from outside the workgroup.  Same one-in-flight rule (the count
 * replicas belong to the context).  Columns 16-byte aligned (binary64: 32), runs16 and gran_counts 16-byte aligned.
 */
#define XM_RUNS_GRAN 2048u
#define XM_RUNS_GRANULES(n_records) (((uint64_t)(n_records) + XM_RUNS_GRAN - 1u) / XM_RUNS_GRAN)
#define XM_RUNS16_ENTRIES(n_records) (XM_RUNS_GRANULES(n_records) * XM_RUNS_GRAN)

int xm_classify_runs_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                         const int32_t *as1, const int32_t *xs1, const int32_t *as2, const int32_t *xs2,
                         const uint64_t *unit_bits, int32_t min_score_floor,
                         uint16_t *runs16, uint16_t *gran_counts, uint64_t *n_out, uint64_t *counts);

int xm_classify_runs_f64_dev(xm_ctx *ctx, void *stream, int mode, uint64_t n_records,
                             const double *as1, const double *xs1, const double *as2, const double *xs2,
                             const uint64_t *unit_bits, double min_score,
                             uint16_t *runs16, uint16_t *gran_counts, uint64_t *n_out, uint64_t *counts);

/* Host memory, no device: list `bin` (0..6) of the segmented form as a flat list of record indices in input order;
 * *n_written = its length (entries past `capacity` are counted, not stored; idx_out may be NULL to count only). */
int xm_runs_expand(uint64_t n_records, const uint16_t *runs16, const uint16_t *gran_counts, int bin,
                   uint32_t *idx_out, uint64_t capacity, uint64_t *n_written);

/*
 * Before a stream that xm_compact_dev / xm_classify_compact*_dev / xm_classify_place*_dev / xm_classify_runs*_dev were issued
 * on is DESTROYED while the context lives on: tell the context.  The library orders workspace calls across streams lazily
 * -- at the next call on another stream it records an event on the previous stream -- so it keeps that stream's handle;
 * this call records the event now (the stream still exists) and forgets the handle.  Cheap, never blocks, harmless for a
 * stream the context does not remember.  xm_strip_destroy does it for the stripper's own streams.
 */
int xm_workspace_release(xm_ctx *ctx, void *stream);

/*
 * Test aid: synchronises the device and reports (*clean = 1) whether the context's counting workspace -- the replicated
 * category_counts and the per-part bin totals, which every counting kernel adds into and every scan / scatter consumes and
 * zeroes -- is all zero, as it must be between calls whatever the sequence of calls (counts only, compaction, a smaller input
 * after a larger one, failed launches).
 */
int xm_workspace_is_clean(xm_ctx *ctx, int *clean);

/*
 * Measurement aid ("measure an on-box streaming-copy ceiling alongside" the 8 TB/s specification): the classify
 * kernel's memory shape without its arithmetic -- reads 16 bytes per record from four int32 columns (n_records rounded down
 * to a multiple of 4), writes XM_BINS4_BYTES(n_records) bytes at most to `out` (the compact stream's size).  The values
 * written mean nothing.  Timed by the caller (events on `stream`).
 */
int xm_stream_probe_dev(xm_ctx *ctx, void *stream, uint64_t n_records,
                        const int32_t *c0, const int32_t *c1, const int32_t *c2, const int32_t *c3, uint8_t *out);

/* ---- multi-GPU: the one collective of the path ------------------------------------------ */
/*
 * The path shards by read block, one process (or thread) and one context per GPU; the only exchange is the sum of
 * category_counts at the end of a run (the Counter the main loops return, xenomapper.py:420 / :520 / :330; printed
 * once by output_summary :558-566).  RCCL over xGMI, 64 x uint64 = 512 bytes, latency-bound.  Bin index lists are
 * never exchanged: shard order is input order.
 * Rank 0 calls xm_comm_unique_id() and hands the XM_UNIQUE_ID_BYTES bytes to every rank out of band (a file, a
 * socket, MPI, torch.distributed ...); every rank then calls xm_comm_init() (collective: returns when all n_ranks
 * have joined).  librccl is looked up at run time -- a copy already mapped into the process (PyTorch's) is reused --
 * so single-GPU users need none.
 */
#define XM_UNIQUE_ID_BYTES 128
int xm_comm_unique_id(void *id_out /* XM_UNIQUE_ID_BYTES */);
int xm_comm_init(xm_ctx *ctx, int n_ranks, int rank, const void *unique_id);
int xm_comm_destroy(xm_ctx *ctx);                 /* also done by xm_ctx_destroy */
int xm_comm_size(const xm_ctx *ctx);              /* ranks of the communicator, 0 = none */
/* In-place sum over all ranks of 64 device uint64 (category_counts), asynchronous on `stream`. */
int xm_allreduce_counts(xm_ctx *ctx, void *stream, uint64_t *counts_dev);

/* ---- per-kernel timing (HIP events on the launch stream) ------------------------------ */
#define XM_K_CLASSIFY 0
#define XM_K_HIST     1
#define XM_K_SCAN     2
#define XM_K_SCATTER  3
#define XM_K_CIGAR    4
#define XM_K_CORRELATE 5
#define XM_K_COUNT    6
/* When enabled, every *_dev call brackets each kernel it launches with hipEventRecord on the
 * launch stream.  xm_timing_read() synchronises the recorded events and adds their elapsed
 * times: ms[k] = total milliseconds, launches[k] = number of launches since the last reset. */
int xm_timing_enable(xm_ctx *ctx, int on);
/* Restrict the bracketing to the kernels whose bit (1 << XM_K_*) is set (default: all).  An event pair costs a
 * few microseconds of stream time per launch, so a throughput run times only the kernel it reports on. */
int xm_timing_select(xm_ctx *ctx, uint32_t kernel_mask);
int xm_timing_reset(xm_ctx *ctx);
int xm_timing_read(xm_ctx *ctx, double ms[XM_K_COUNT], uint64_t launches[XM_K_COUNT]);

#ifdef __cplusplus
}
#endif
#endif /* XENOMAPPER_HIP_H */
