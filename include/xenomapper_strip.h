/*
 * xenomapper_strip.h -- C ABI of the SAM column stripper that runs ON the GPU (libxenomapper_hip.so, gfx950).
 *
 * The reference reads its two SAM files line by line on one core -- getReadPairs
 * (/root/reference/xenomapper/xenomapper.py:95-118: readline, strip, split, compare the names) and the text half
 * of tag_func (get_tag :176-191, get_tag_with_ZS_as_XS :193-206: find the optional fields that contain "AS" /
 * "XS" / "ZS", int() of what follows the last ':').  include/xenomapper_host.h does that with host threads
 * (xmh_parse); this header is the same step with the text shipped to the device instead: the host only copies
 * the two windows of text into page-locked staging buffers, the kernels index the lines (universal newlines),
 * split them by Python's white space rules, find the tags, compare the names and leave the score columns and
 * the unit mask IN HBM, where the classify kernels (include/xenomapper_hip.h) read them -- no column ever
 * crosses PCIe.  What comes back is what the line writer (xmh_emit) needs: where every line starts and ends.
 *
 * Same exactness contract as xmh_parse: a value that is not a plain int32 integer or a tag that matches twice
 * is flagged per line (XMS_LINE_EX_*), never guessed; non-ASCII input is refused as a whole (non_ascii).  The
 * caller resolves flagged records with the reference-equivalent text functions (xm_strip_columns brings the
 * columns to the host for that) or hands the window to xmh_parse.
 *
 * Scope: SAM text, all three built-in plugins, the walk with and without --skip_repeated_reads (xenomapper.py:110-117).
 * Windows are shorter than 4 GiB - 64 KiB.  BAM input, windows beyond that and non-ASCII text stay with xenomapper_host.h.
 *
 * A stripper has two slots so that one window pair can be copied, uploaded and stripped (one host thread) while
 * the block before it is classified and written (another thread); calls on one slot must not overlap.
 */
#ifndef XENOMAPPER_STRIP_H
#define XENOMAPPER_STRIP_H

#include "xenomapper_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define XMS_ABI_VERSION 3   /* 2: xm_strip_fetch_bins / xm_strip_out_wait.  3: xm_strip_begin_behind / xm_strip_set_lead */
#define XMS_SLOTS 2
#define XMS_MAX_WINDOW 0xFFFF0000ull

/* score_mode (the values of XMH_SCORE_*) */
#define XMS_SCORE_AS_XS 0   /* get_tag                 xenomapper.py:176-191 */
#define XMS_SCORE_AS_ZS 1   /* get_tag_with_ZS_as_XS   xenomapper.py:193-206 */
#define XMS_SCORE_CIGAR 2   /* get_cigarbased_AS_tag   xenomapper.py:228-256: NM + the CIGAR operations, as the packed CIGAR columns of
                               xenomapper_hip.h, left on the device; XS by get_tag */

/* line flags; bit 0 is XMH_LINE_NORMAL of xenomapper_host.h */
#define XMS_LINE_NORMAL   0x01u  /* the line already equals '\t'.join(fields)                                     */
#define XMS_LINE_BLANK    0x02u  /* no field at all: ends the walk (xenomapper.py:105)                            */
#define XMS_LINE_EX_A     0x1Cu  /* bits 2-4: exception kind of the AS column (XMH_EX_*: 1 NONINT, 2 DUP; CIGAR mode, about NM and
                                    the CIGAR field: 1 NONINT, 3 SHORT, 4 BIGLEN), 0 = none                           */
#define XMS_LINE_EX_A_SHIFT 2
#define XMS_LINE_EX_X     0x60u  /* bits 5-6: same for the XS (or ZS) column: 0, 1, 2                                */
#define XMS_LINE_EX_X_SHIFT 5
#define XMS_LINE_MISMATCH 0x80u  /* file-1 flags only: the names of the two files differ (xenomapper.py:106)      */

typedef struct xm_strip xm_strip;

/* Result of xm_strip_run; the fields up to mismatch_at mean what they mean in xmh_block.  The arrays are page-locked
 * host memory owned by the stripper: n_records entries each, valid until the next xm_strip_run on the same slot. */
typedef struct {
    uint64_t n_records;
    uint64_t consumed1, consumed2;
    uint64_t consumed_lines1, consumed_lines2;
    int32_t  ended, starved;
    int64_t  mismatch_at;
    int32_t  non_ascii;        /* 1: a byte >= 0x80 in either window; nothing else of the result is meaningful */
    int32_t  overflow;         /* 1: the skipping walk met more lines than max_records + 1 in a window (or the packed CIGAR
                                  operations outgrew their array): nothing else is meaningful, strip the window with xmh_parse */
    uint64_t n_exceptions;     /* records among the n_records whose flags carry XMS_LINE_EX_A / _EX_X on either line */
    uint64_t n_lines1, n_lines2;                        /* lines found in each window (complete ones; the last one of a file too) */
    const uint32_t *line_off1, *line_off2;              /* first byte of the record's line in its window */
    const uint32_t *line_len1, *line_len2;              /* bytes without the terminator */
    const uint32_t *norm_len1, *norm_len2;              /* length of '\t'.join(fields) */
    const uint8_t  *line_flags1, *line_flags2;          /* XMS_LINE_* */
    float ms_upload, ms_kernels;                        /* device time from the first upload to the last, and of the kernels (HIP events) */
} xm_strip_block;

int xms_abi_version(void);

/* ctx: the classifier context the columns will be classified with (same device).  Allocates nothing large. */
int xm_strip_create(xm_ctx *ctx, int device_id, xm_strip **out);
int xm_strip_destroy(xm_strip *s);      /* while ctx is still alive: the slot streams are handed back to it (xm_workspace_release) */

/* Make the slot's buffers hold windows of window_bytes per file and max_records records (grows only; not to be called
 * while the slot is in use).  XM_ERR_INVALID_ARG beyond XMS_MAX_WINDOW. */
int xm_strip_reserve(xm_strip *s, int slot, uint64_t window_bytes, uint64_t max_records);

/* Page-locked staging buffer of a slot for file 0 / 1 (window_bytes long): the caller copies the window there. */
char *xm_strip_staging(xm_strip *s, int slot, int file);

/* Reading AHEAD (XMS_ABI_VERSION 3).  The next window begins somewhere in the last bytes of the current one -- its walk says where
 * -- so a caller that wants to read the next window's bytes while the current one is still being stripped reads the bytes BEHIND
 * the current window into the other slot's staging buffer, `room` bytes into it (a multiple of 64 KiB):
 *   xm_strip_begin_behind(s, slot, file, room);                     a new window of this file; its bytes [0, room) come later
 *   xm_strip_upload(s, slot, file, room + k, n) ...                  piece by piece as they are read, as always
 *   ... the current window's xm_strip_run returns: the next one starts `tail` bytes in front of what was read ...
 *   copy those `tail` <= room bytes to staging + room - tail;  xm_strip_set_lead(s, slot, file, room - tail);
 *   xm_strip_run(s, slot, room + bytes read, ...)
 * The window is then the whole buffer [0, len) of which the first `lead` bytes are no text: no line begins or ends there, every
 * offset the run reports (consumed*, line_off*) still counts from the buffer's first byte, so consumed* - lead bytes of text are
 * done.  A lead holds for one run.  Needs the default zero-copy mode (XM_STRIP_ZEROCOPY=0: XM_ERR_INVALID_ARG from the run). */
int xm_strip_begin_behind(xm_strip *s, int slot, int file, uint64_t room);
int xm_strip_set_lead(xm_strip *s, int slot, int file, uint64_t lead);

/* Start the upload of bytes [offset, offset + bytes) of a staged window (asynchronous, on the slot's stream): called piece by
 * piece, in order from offset 0, while the host is still copying the rest of the window into the staging buffer, it hides
 * the copy behind the PCIe transfer.  Optional: xm_strip_run uploads whatever has not been sent. */
int xm_strip_upload(xm_strip *s, int slot, int file, uint64_t offset, uint64_t bytes);

/*
 * Upload the two staged windows (len1 / len2 bytes; eof*: the window reaches the end of its file) and strip them.
 * paired: the unit mask is name[k] == name[k-1] (else every record is a unit); skip_repeated / keep_halo / max_records as
 * in xmh_parse.  Blocking (the slot's own stream).  The columns of the block stay on the device for xm_strip_classify.
 */
int xm_strip_run(xm_strip *s, int slot, uint64_t len1, int eof1, uint64_t len2, int eof2,
                 int score_mode, int paired, int skip_repeated, int keep_halo, uint64_t max_records, xm_strip_block *out);

/*
 * The fused main loop (xm_classify_compact_dev; after a XMS_SCORE_CIGAR run xm_classify_compact_cigar_packed_dev, with
 * XM_ERR_RANGE when a synthesised score leaves int32) on the first n_records records of the slot's block.  Results in
 * page-locked host arrays owned by the stripper, valid until the next call on the slot: code (n_records bytes), the six
 * index lists back to back (*idx) with bin_offsets[8], counts[64].
 */
int xm_strip_classify(xm_strip *s, int slot, int mode, uint64_t n_records, int32_t min_score_floor,
                      const uint8_t **code, const uint32_t **idx, uint64_t bin_offsets[8], uint64_t counts[64]);

/*
 * After xm_strip_classify: the six OUTPUTS themselves, gathered on the device (ABI 2) -- for every bin whose sink is given
 * (sink_mask bit b) the lines the reference's loop prints for the bin's units, in input order (xenomapper.py:332-350, :423-448,
 * :521-550: primary bins file 1's line(s), secondary bins file 2's, `unresolved` file 1's then file 2's; a paired unit is
 * records i - 1 and i), each as '\t'.join(fields) + '\n'.  One page-locked stream owned by the stripper, valid after
 * xm_strip_out_wait and until the next call on the slot: bin b's text = text[bin_off[b] .. bin_off[b + 1]), b = 0..5 in the state
 * order PS, SS, PM, SM, unresolved, unassigned.  The copy to the host runs on a stream of the slot's own, beside the next
 * window's upload and kernels.  status 0: on its way; 2: more text than the buffers hold (units that overlap); 3: a wanted
 * line is not '\t'.join(fields) as it stands (mixed white space) -- then nothing was copied: write this window with xmh_emit
 * from the line tables of xm_strip_run.
 */
typedef struct {
    const uint8_t *text;
    uint64_t bin_off[8];
    int32_t  status, reserved;
} xm_strip_bins;
int xm_strip_fetch_bins(xm_strip *s, int slot, uint64_t n_records, int paired, uint32_t sink_mask, xm_strip_bins *out);
int xm_strip_out_wait(xm_strip *s, int slot);      /* blocks until the stream asked for last has arrived (any thread) */

/* Bring the first n_records of the slot's score columns and ceil(n/64) words of the unit mask to the host (for records
 * the caller must patch by the text rules).  Any pointer may be NULL. */
int xm_strip_columns(xm_strip *s, int slot, uint64_t n_records, int32_t *as1, int32_t *xs1, int32_t *as2, int32_t *xs2,
                     uint64_t *unit_bits);

/* After a XMS_SCORE_CIGAR run: the packed CIGAR columns of one file for the first n_records records copied to the host (any
 * pointer may be NULL; cig_tile: XM_CIG_TILES(n_records) + 1 words; *n_ops = words of cig_ops in use).  For tests and for
 * callers that want the columns themselves. */
int xm_strip_cigar_columns(xm_strip *s, int slot, int file, uint64_t n_records, int32_t *nm, uint8_t *cig_cnt, uint32_t *cig_tile,
                           uint32_t *cig_ops, uint64_t ops_capacity, uint64_t *n_ops);

/* Device addresses of the slot's columns: as1, xs1, as2, xs2 (int32) and unit_bits (uint64), for callers that launch the
 * classify kernels themselves (xm_classify_*_dev). */
int xm_strip_device_columns(xm_strip *s, int slot, void *ptrs[5]);

/* Text of the last failing HIP call of this stripper (empty string if none). */
const char *xm_strip_last_error(const xm_strip *s);

#ifdef __cplusplus
}
#endif
#endif /* XENOMAPPER_STRIP_H */
