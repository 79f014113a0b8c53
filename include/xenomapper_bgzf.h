/*
 * xenomapper_bgzf.h -- C ABI of the BGZF / BAM input path that runs ON the GPU (libxenomapper_hip.so, gfx950).
 *
 * The reference reads BAM by piping it through `samtools view` and then treats the text like SAM input
 * (get_bam_header, bam_lines, getBamReadPairs: /root/reference/xenomapper/xenomapper.py:48-93).  The native host decoder of
 * include/xenomapper_host.h (xmh_bam_*) does that on CPU threads and is decoder-bound (inflate + printing, 55 % of the
 * wall time).  This header moves the two data-parallel halves to the device:
 *   1. xm_bgzf_inflate_dev   the BGZF blocks of a window -- independent raw-DEFLATE streams of <= 64 KiB (SAM specification
 *                            4.1; RFC 1951 / 1952) -- inflated by the GPU, many blocks at a time;
 *   2. xm_bam_columns_dev    the alignment records of the inflated bytes -> the classifier's columns in HBM (AS / XS / ZS / NM
 *                            with the plugins' substring and duplicate rules, the packed CIGAR columns, names for the unit rule).
 * What the host keeps: walking the member headers (a few bytes per 64 KiB block: xm_bgzf_index, no device), and printing the
 * SAM text of the records a sink takes.
 *
 * Conventions as in xenomapper_hip.h: extern "C", plain pointers and sizes, int status, caller-allocated buffers, *_dev calls
 * only enqueue work on `stream`.
 */
#ifndef XENOMAPPER_BGZF_H
#define XENOMAPPER_BGZF_H

#include "xenomapper_hip.h"
#include "xenomapper_strip.h"     /* XMS_SCORE_*, XMS_LINE_* */

#ifdef __cplusplus
extern "C" {
#endif

#define XMB_ABI_VERSION 3   /* 2: xm_bamdev_fetch_bins; f and B:f fields printed on the device.  3: raw1 / raw2 exist from the
                               first xm_bamdev_fetch_raw on (xm_bamdev_raw) */

/* One BGZF block of the compressed image (24 bytes; the layout the kernels read). */
typedef struct {
    uint64_t cdata_off;     /* first byte of the block's raw-DEFLATE stream in the compressed image            */
    uint64_t out_off;       /* where its ISIZE inflated bytes go in the output buffer (the blocks back to back)  */
    uint32_t cdata_len;     /* bytes of the DEFLATE stream (member size - header - 8 trailer bytes)             */
    uint32_t isize;         /* inflated size from the member trailer (<= 65536)                                   */
} xm_bgzf_block;

/* per-block status written by xm_bgzf_inflate_dev: 0 = ok, else the decoder's error (xm_bgzf_strerror) */
#define XMB_OK 0

/*
 * Host only, no device: walk the gzip member headers of a BGZF image (every block carries its size in the 'BC' extra
 * subfield) starting at byte `start`, until `max_out` inflated bytes are scheduled, the image ends or `cap` blocks are
 * listed.  Fills blocks[] (out_off counts from *out_off_base = 0 of this call), crc[] (the trailer's CRC-32 of every block;
 * may be NULL), *n_blocks, *next (the byte where the next call continues) and *out_bytes (sum of the ISIZEs).
 * XM_ERR_INVALID_ARG: not a BGZF member where one must start, or a member that runs past the image.
 */
int xm_bgzf_index(const uint8_t *data, uint64_t len, uint64_t start, uint64_t max_out, xm_bgzf_block *blocks, uint32_t *crc,
                  uint64_t cap, uint64_t *n_blocks, uint64_t *next, uint64_t *out_bytes);
/* the same over a buffer that may end inside a member (bytes read ahead of a file): stops in front of the member that is cut
 * instead of refusing the buffer; *next = where that member begins */
int xm_bgzf_index_prefix(const uint8_t *image, uint64_t len, uint64_t start, uint64_t max_out, xm_bgzf_block *blocks, uint32_t *crc,
                         uint64_t cap, uint64_t *n_blocks, uint64_t *next, uint64_t *out_bytes);

/*
 * Inflate n_blocks BGZF blocks.  comp: the compressed image on the device, 16-byte aligned, with at least
 * XMB_COMP_PAD readable bytes behind the last block's data (the decoder reads ahead in 128-byte granules; what it reads
 * there does not matter).  blocks: n_blocks descriptors on the device (cdata_off / out_off relative to comp / out).
 * out: the output buffer on the device (out_off + isize of every block inside it).  status: n_blocks uint32 on the device.
 * work: 4 bytes of device scratch (the launch's block counter; zeroed by the call).
 * No input makes the decoder read outside [comp, last block + XMB_COMP_PAD) or write outside a block's own
 * [out_off, out_off + isize): a damaged stream ends with a non-zero status.
 */
#define XMB_COMP_PAD 1024u
int xm_bgzf_inflate_dev(xm_ctx *ctx, void *stream, const uint8_t *comp, const xm_bgzf_block *blocks, uint64_t n_blocks,
                        uint8_t *out, uint32_t *status, uint32_t *work);

/* The same launch for windows of BAM: after a chain has written block k it also follows the alignment records' block_size chain
 * through that block and reads what the classifier needs out of every record, while the block is still near the CU that wrote
 * it (`out + raw_base` = first byte of the file's window, positions count from there): slots[0 .. *count) = where the records
 * that begin in [start, end) begin; name_off / name_len / a / x / flag / n_cigar / cig_at [0 .. *count) = the record fields of
 * xm_bamdev's stripper (tags: which tags are read -- 'X' | 'A' << 8 | 'S' << 16: AS and XS; 'Z' | ...: AS and ZS;
 * 'X' | 'N' << 8 | 'M' << 16 | 1 << 24: NM, first match, and XS, the --cigar_scores plugin -- 0: record starts only); *exit_at = where the chain leaves the block (== end when the
 * next block begins with a record; 0xFFFFFFFF: the block failed, or it holds more than slot_cap records); a record that does not
 * end in front of n_raw (the window's end) ends the walk, and so does a size word cut by the block's end.  Entries with
 * end <= start are skipped.  walk[] (device memory) has n_blocks entries; all arrays are device memory of slot_cap entries. */
#define XM_BGZF_TAGS_AS_XS ((uint32_t)'X' | (uint32_t)'A' << 8 | (uint32_t)'S' << 16)             /* get_tag, xenomapper.py:176-191 */
#define XM_BGZF_TAGS_AS_ZS ((uint32_t)'Z' | (uint32_t)'A' << 8 | (uint32_t)'S' << 16)             /* get_tag_with_ZS_as_XS, :193-206 */
#define XM_BGZF_TAGS_NM_XS ((uint32_t)'X' | (uint32_t)'N' << 8 | (uint32_t)'M' << 16 | 1u << 24)  /* get_cigarbased_AS_tag, :228-256 */
typedef struct {
    uint64_t raw_base;
    uint32_t start, end;             /* first byte to look at (the block's first, or behind the BAM header), the block's end     */
    uint32_t n_raw;
    uint32_t slot_cap;
    uint32_t *count, *exit_at;
    uint32_t *slots;
    uint32_t *name_off, *name_len;
    int32_t  *a, *x;
    uint8_t  *flag;
    uint32_t *n_cigar, *cig_at;      /* the records' CIGAR operations: how many, where in the window (may both be null)          */
    uint32_t tags, reserved;
} xm_bgzf_walk;
int xm_bgzf_inflate_walk_dev(xm_ctx *ctx, void *stream, const uint8_t *comp, const xm_bgzf_block *blocks, uint64_t n_blocks,
                             uint8_t *out, uint32_t *status, uint32_t *work, const xm_bgzf_walk *walk);

/* CRC-32 (the gzip polynomial) of every block's inflated bytes, for the comparison with the member trailers:
 * crc_out[b] for b < n_blocks, on the device.  One wave per block. */
int xm_bgzf_crc32_dev(xm_ctx *ctx, void *stream, const uint8_t *out, const xm_bgzf_block *blocks, uint64_t n_blocks,
                      uint32_t *crc_out);

const char *xm_bgzf_strerror(uint32_t block_status);

/* ---- BAM records -> the classifier's columns, on the device ------------------------------------------------------------
 *
 * xm_bamdev: the device half of BAM input for the lock-step walk of two files (getBamReadPairs, xenomapper.py:66-93, with
 * getReadPairs' walk :95-118): per window and file the caller stages compressed BGZF blocks, the library inflates them
 * (xm_bgzf_inflate_dev + the CRC check), finds the alignment records, reads AS / XS / ZS of every record with the plugins'
 * rules applied to the typed fields (get_tag :176-191, get_tag_with_ZS_as_XS :193-206: a tag is matched by the field of
 * that name and by any Z / H field whose printed text contains the two letters; two matches = the duplicate error;
 * a match that is not an integer inside [-(2^31-1), 2^31-1] is flagged, never guessed), compares the names of the two
 * files (:106) and of adjacent records (:402) and leaves the score columns and the unit mask IN HBM for the classify
 * kernels.  What comes back to the host: the inflated bytes and the record table (the writer prints the SAM text of the
 * records from them: xmh_bam_print), and per-record exception flags.
 *
 * Scope: the walk without skip_repeated_reads, score modes XMS_SCORE_AS_XS / XMS_SCORE_AS_ZS.  BGZF blocks must begin
 * at a record boundary (what htslib / samtools write: bgzf_flush_try before every record); a file written otherwise is
 * reported (`unaligned`) and goes through the host decoder of xenomapper_host.h, as do --cigar_scores, the skipping
 * walk, and windows holding a record the text rules might split differently (`weird`: white space, control or non-ASCII
 * bytes in a name, tag or string value, qualities above 93, CIGARs kept in a CG field).
 */
typedef struct xm_bamdev xm_bamdev;

typedef struct {
    uint64_t n_records;              /* record pairs yielded by the walk in this window                                          */
    uint64_t consumed1, consumed2;   /* bytes of each window (carry included) the walk is finished with; with keep_halo the
                                        last yielded record is NOT consumed                                                     */
    uint64_t raw_len1, raw_len2;     /* inflated bytes in each window: carry + the blocks of this call                           */
    uint64_t n_rec1, n_rec2;         /* complete records found in each window                                                    */
    int32_t  ended, starved;         /* as xmh_block: a file ran out of records at its end / more input is needed                */
    int64_t  mismatch_at;            /* first pair whose names differ (AssertionError, :106) or -1                               */
    int32_t  bad_block;              /* != 0: a BGZF block failed (decoder status or CRC-32), a malformed record, or a file that
                                        ends inside a record: nothing else is meaningful                                         */
    int32_t  unaligned;              /* 1: a block does not begin with a record: no record table, use the host decoder           */
    int32_t  weird;                  /* 1: a record the text rules might read differently: use the host decoder for this window  */
    int32_t  reserved;
    uint64_t n_exceptions;           /* pairs whose flags carry XMS_LINE_EX_A / _EX_X on either record                           */
    const uint8_t  *raw1, *raw2;     /* page-locked host copies of the inflated windows -- allocated and filled only when asked
                                        for (xm_bamdev_fetch_raw; until the slot's first one they are NULL here: take the
                                        addresses from xm_bamdev_raw afterwards), complete once xm_bamdev_raw_wait(slot) has
                                        returned, valid until the slot runs again                                               */
    const uint32_t *rec_off1, *rec_off2;   /* start of every record (its block_size word) in raw*: n_rec* entries              */
    const uint8_t  *flags1, *flags2;       /* per yielded pair: XMS_LINE_NORMAL | exception bits (XMS_LINE_EX_A / _EX_X)        */
    float ms_inflate, ms_kernels;    /* device time of the inflate + CRC launches, and of the record kernels (HIP events)       */
} xm_bamdev_block;

/* what one file contributes to a window */
typedef struct {
    uint64_t comp_len;               /* compressed bytes staged in xm_bamdev_staging(slot, file)                                */
    const xm_bgzf_block *blocks;     /* host array: the blocks inside those bytes (cdata_off relative to the staging buffer,
                                        out_off from 0 = behind the carry), with their trailer CRCs                              */
    const uint32_t *crc;
    uint64_t n_blocks;
    int32_t  carry_slot;             /* bytes [carry_off, carry_off + carry_len) of this file's previous window (slot             */
    uint64_t carry_off, carry_len;   /*   carry_slot) are put in front of the new bytes; carry_len 0: none                       */
    int32_t  eof;                    /* no block of the file is left behind these                                                */
    uint64_t skip;                   /* the first `skip` inflated bytes of the new blocks are not alignment records (the BAM
                                        header: magic, text, reference list); less than the first block's ISIZE                   */
    uint64_t uploaded;               /* the first `uploaded` of the comp_len staged bytes went up with xm_bamdev_upload already   */
} xm_bamdev_input;

int xm_bamdev_create(xm_ctx *ctx, int device_id, xm_bamdev **out);
int xm_bamdev_destroy(xm_bamdev *b);                          /* while ctx is alive (its streams are handed back)                */
/* room per slot and file for comp_bytes of compressed input, raw_bytes of inflated window, max_blocks, max_records */
int xm_bamdev_reserve(xm_bamdev *b, int slot, uint64_t comp_bytes, uint64_t raw_bytes, uint64_t max_blocks, uint64_t max_records);
uint8_t *xm_bamdev_staging(xm_bamdev *b, int slot, int file);
/* Send the first `bytes` of a file's staging buffer to the device now, on a stream of the slot's own -- from any thread, while
 * the slot is idle (its last xm_bamdev_run has returned, the next has not begun): the next run, told so in `uploaded`, waits for
 * the copy instead of making it.  A later xm_bamdev_reserve that grows the buffers forgets what was sent. */
int xm_bamdev_upload(xm_bamdev *b, int slot, int file, uint64_t bytes);
/* (By default the inflate launch of xm_bamdev_run reads the staging buffers where they are -- page-locked, device-mapped host memory --
 * and nothing is uploaded at all: xm_bamdev_upload then only notes the count.  XM_BAMDEV_ZEROCOPY=0 in the environment restores the
 * copy into HBM, ahead of the run or inside it.) */
/* inflate, find the records, strip, pair.  score_mode: XMS_SCORE_AS_XS, XMS_SCORE_AS_ZS, or XMS_SCORE_CIGAR (get_cigarbased_AS_tag,
 * xenomapper.py:228-256: column "AS" then holds NM -- the FIRST optional field that holds the letters decides, :247-250 -- and
 * xm_bamdev_classify makes the packed CIGAR columns of the records' CIGAR words and runs xm_classify_compact_cigar_packed_dev).
 * skip_repeated: the skipping walk of getReadPairs (xenomapper.py:110-117; what the command line does for single-end input, :691):
 * each file is cut into runs of adjacent records with one name, pair k = the first record of run k of both files; a run that
 * reaches the end of a window whose file goes on is not yielded (starved: it may go on in the next window); rec_off1/2 then list
 * the yielded records only, and consumed1/2 lie behind the records skipped.
 * Blocking (the slot's own stream).  The inflated windows stay on the device: what the
 * writer needs of them is asked for afterwards, one of */
int xm_bamdev_run(xm_bamdev *b, int slot, const xm_bamdev_input in[2], int score_mode, int paired, int skip_repeated, int keep_halo,
                  uint64_t max_records, xm_bamdev_block *out);
/* (a) the whole windows into raw1 / raw2 (a window the host has to walk or print whole: unaligned, weird, exceptions).  The two
 * page-locked buffers (a third of what a slot would pin otherwise) are made by the slot's first call, for the capacity reserved;
 * XM_ERR_OOM when they cannot be.  xm_bamdev_raw: their addresses (NULL before). */
int xm_bamdev_fetch_raw(xm_bamdev *b, int slot);
const uint8_t *xm_bamdev_raw(xm_bamdev *b, int slot, int file);
/* (b) after xm_bamdev_classify: only the records a sink takes -- a unit's lines come from one file (primary bins: file 1, secondary
 * bins: file 2, unresolved: both; xenomapper.py:423-448), sink_mask bit b = sink b is given -- packed next to each other:
 * raw1 / raw2 = the packed records, off1 / off2[i] = where record i went (0xFFFFFFFF: no sink takes it), bytes1 / bytes2 */
typedef struct {
    const uint8_t  *raw1, *raw2;
    const uint32_t *off1, *off2;
    uint64_t bytes1, bytes2;
} xm_bamdev_text;
int xm_bamdev_fetch_wanted(xm_bamdev *b, int slot, uint64_t n_records, int paired, uint32_t sink_mask, xm_bamdev_text *out);
/* (c) after xm_bamdev_classify: the SAM TEXT of the records a sink takes, printed on the device -- the lines `samtools view` would
 * print (the reference reads BAM through it: getBamReadPairs / bam_lines, xenomapper.py:56-93), next to each other in text1 / text2,
 * line_off*[i] / line_len*[i] = where record i's line is and how long (without its '\n'; 0 / 0: no sink takes it).  Needs the
 * files' reference names (xm_bamdev_set_refs: the names back to back, at[n_refs + 1] positions; once per pair of files;
 * XM_ERR_INVALID_ARG without them).
 * Floating-point fields of the specification's types (f, B:f) are printed as printf("%g") prints them (exact: csrc/xm_fmtg.h).
 * status 0: the text is on its way (xm_bamdev_raw_wait); 1: a record of THIS window holds a binary64 field (type d: htslib
 * accepts it, the specification does not have it; its "%g" is the host printer's), 2: more text than the slot's buffers hold --
 * nothing was copied then: ask for (b) and print this window on the host; the next window may be printed here again. */
typedef struct {
    const uint8_t  *text1, *text2;
    const uint32_t *line_off1, *line_off2, *line_len1, *line_len2;
    uint64_t bytes1, bytes2;
    int32_t  status, reserved;
} xm_bamdev_lines;
int xm_bamdev_set_refs(xm_bamdev *b, int file, const uint8_t *names, const uint32_t *at, uint32_t n_refs);
int xm_bamdev_fetch_text(xm_bamdev *b, int slot, uint64_t n_records, int paired, uint32_t sink_mask, xm_bamdev_lines *out);
/* (d) after xm_bamdev_classify: the six OUTPUTS themselves, gathered on the device (XMB_ABI_VERSION 2) -- the text of every bin whose
 * sink is given (sink_mask bit b), i.e. for the bin's units in input order the lines the reference's loop prints for them
 * (xenomapper.py:332-350, :423-448, :521-550: primary bins file 1's line(s), secondary bins file 2's, `unresolved` file 1's
 * then file 2's; a paired unit is records i - 1 and i), as `samtools view` prints them (as (c)).  One page-locked stream:
 * bin b's text = text[bin_off[b] .. bin_off[b + 1]), b = 0..5 in the state order PS, SS, PM, SM, unresolved, unassigned
 * (bin_off[6] = bin_off[7] = all of it: the seventh list, state 6, cannot occur for int32 columns).  Needs the reference
 * names as (c).  The caller writes six contiguous byte ranges per window; nothing else of the window crosses the link.
 * status as (c): 0 on its way (xm_bamdev_raw_wait), 1 a binary64 field, 2 more text than the slot's buffers hold. */
typedef struct {
    const uint8_t *text;
    uint64_t bin_off[8];
    int32_t  status, reserved;
} xm_bamdev_bins;
int xm_bamdev_fetch_bins(xm_bamdev *b, int slot, uint64_t n_records, int paired, uint32_t sink_mask, xm_bamdev_bins *out);
/* the copies run on a stream of their own; this blocks until the one asked for last has arrived (any thread) */
int xm_bamdev_raw_wait(xm_bamdev *b, int slot);
/* the fused main loop on the slot's columns (as xm_strip_classify) */
int xm_bamdev_classify(xm_bamdev *b, int slot, int mode, uint64_t n_records, int32_t min_score_floor,
                       const uint8_t **code, const uint32_t **idx, uint64_t bin_offsets[8], uint64_t counts[64]);
/* the slot's score columns and unit mask copied to the host (records the caller must patch by the text rules) */
int xm_bamdev_columns(xm_bamdev *b, int slot, uint64_t n_records, int32_t *as1, int32_t *xs1, int32_t *as2, int32_t *xs2,
                      uint64_t *unit_bits);
/* After a XMS_SCORE_CIGAR run: the packed CIGAR columns (include/xenomapper_hip.h) the device made of one file's first n_records
 * records -- NM, and the records' own CIGAR words (BAM stores them as the kernel reads them: len << 4 | op) -- copied to the host;
 * arguments as xm_strip_cigar_columns.  For tests and for callers that want the columns themselves. */
int xm_bamdev_cigar_columns(xm_bamdev *b, int slot, int file, uint64_t n_records, int32_t *nm, uint8_t *cig_cnt, uint32_t *cig_tile,
                            uint32_t *cig_ops, uint64_t ops_capacity, uint64_t *n_ops);
const char *xm_bamdev_last_error(const xm_bamdev *b);

#ifdef __cplusplus
}
#endif
#endif /* XENOMAPPER_BGZF_H */
