/*
 * xenomapper_host.h -- C ABI of the host-side SAM column stripper and line writer
 * (libxenomapper_host.so, plain C++17 + threads; no GPU code).
 *
 * These are the data formats either side of the classification path:
 *   xmh_parse  replaces the lock-step reader getReadPairs (xenomapper.py:95-118) and the *text* half of the
 *              tag_func plugins -- finding the AS / XS / ZS / NM fields by substring (xenomapper.py:186-190,
 *              :247) and the CIGAR operations (xenomapper.py:251) -- and hands the device structure-of-arrays
 *              columns instead of Python lists.  Arithmetic (scores, states, bins) stays on the GPU.
 *   xmh_emit   replaces the '\t'.join(fields) + print() of the main loops (xenomapper.py:332-350, :423-448,
 *              :521-550): gathers the lines of the units of one output bin, whitespace-normalised, in order.
 *
 * Exactness contract: anything this parser cannot reproduce bit-for-bit is *reported*, never guessed --
 * values that are not plain int32 integers (floats, "inf", digits with '_', 64-bit values ...), duplicate tag
 * matches and short lines come back as exceptions (record, column, kind) which the Python host resolves with
 * the reference-equivalent text functions; input containing non-ASCII bytes is refused (XMH_ERR_NON_ASCII) and
 * the host uses its Python reader for that input.
 */
#ifndef XENOMAPPER_HOST_H
#define XENOMAPPER_HOST_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XMH_ABI_VERSION 5

#define XMH_OK              0
#define XMH_ERR_INVALID_ARG (-1)
#define XMH_ERR_OOM         (-2)
#define XMH_ERR_NON_ASCII   (-3)   /* a byte >= 0x80 in the window: Python's str.split() rules would apply */
#define XMH_ERR_BAD_BAM     (-4)   /* not a BGZF/BAM image, or a truncated / corrupt one */
#define XMH_NEED_TEXT       1     /* xmh_parse_pre: a record of this window must be split by the text rules; call xmh_parse */

/* which optional fields feed the score columns (the three tag_func plugins) */
#define XMH_SCORE_AS_XS 0   /* get_tag                 xenomapper.py:176-191 */
#define XMH_SCORE_AS_ZS 1   /* get_tag_with_ZS_as_XS   xenomapper.py:193-206 */
#define XMH_SCORE_CIGAR 2   /* get_cigarbased_AS_tag   xenomapper.py:228-256 (NM + CIGAR -> CSR; XS by get_tag) */

/* exception kinds (see xmh_block.exc_*) */
#define XMH_EX_NONINT   1   /* value is not a plain integer in [-(2^31-1), 2^31-1]: re-parse in Python        */
#define XMH_EX_DUP      2   /* more than one optional field contains the tag: ValueError (xenomapper.py:189)   */
#define XMH_EX_SHORT    3   /* CIGAR mode, NM present but the line has fewer than 6 fields: IndexError         */
#define XMH_EX_BIGLEN   4   /* a CIGAR operation length >= 2^28: does not fit the packed op column             */

/* exception columns */
#define XMH_COL_AS1 0
#define XMH_COL_XS1 1
#define XMH_COL_AS2 2
#define XMH_COL_XS2 3

/* line flags */
#define XMH_LINE_NORMAL 1   /* the line already equals '\t'.join(fields) */

typedef struct xmh_parser xmh_parser;

/* Result of one xmh_parse call; every pointer addresses memory owned by the parser, valid until the next
 * xmh_parse / xmh_parser_destroy on it. */
typedef struct {
    uint64_t n_records;        /* records yielded by the lock-step walk in this window                      */
    uint64_t consumed1;        /* bytes of each window the walk is finished with (the next window starts      */
    uint64_t consumed2;        /*   here).  With keep_halo the last yielded record is NOT consumed.           */
    int32_t  ended;            /* 1: a blank line or EOF of either file ended the walk (xenomapper.py:105)    */
    int32_t  starved;          /* 1: window exhausted before max_records, more input needed (not ended)       */
    int64_t  mismatch_at;      /* first record whose names disagree (AssertionError, :106) or -1; records    */
                               /*   at and after it are not reported                                           */
    /* score columns, n_records each (XM_ABSENT = INT32_MIN when the tag is absent) */
    const int32_t *as1, *xs1, *as2, *xs2;
    /* CIGAR mode: nm (INT32_MIN = no NM field), CSR offsets (n_records + 1) and packed ops len<<4|op */
    const int32_t *nm1, *nm2;
    const uint32_t *cig_off1, *cig_off2, *cig_ops1, *cig_ops2;
    /* packed unit mask, ceil(n/64) words: paired -> name[i] == name[i-1]; else every record */
    const uint64_t *unit_bits;
    /* the line of every record in each window (offset of first byte, length without terminator, flags,
     * length after whitespace normalisation) */
    const uint64_t *line_off1, *line_off2;
    const uint32_t *line_len1, *line_len2, *norm_len1, *norm_len2;
    const uint8_t  *line_flags1, *line_flags2;
    /* exceptions, ordered by (record, column) */
    uint64_t n_exc;
    const uint32_t *exc_record;
    const uint8_t  *exc_col, *exc_kind;
    /* lines of each window in front of consumed1 / consumed2 (skipped repeats included) */
    uint64_t consumed_lines1, consumed_lines2;
} xmh_block;

int xmh_abi_version(void);
const char *xmh_strerror(int status);
/* workers used when n_threads <= 0: CPUs in the affinity mask, cut to the cgroup CPU quota if one is set, at most 64 */
int xmh_default_threads(void);
int xmh_parser_create(int n_threads, xmh_parser **out);
int xmh_parser_destroy(xmh_parser *p);

/*
 * Walk two SAM record windows in lock-step.  buf*: the bytes after the header (or after what earlier calls
 * consumed); eof*: the window reaches the end of the file.  paired: compute the unit mask from adjacent equal
 * names (else every record is a unit).  skip_repeated: after each yielded pair advance each file past further
 * lines carrying the same name (xenomapper.py:110-114).  keep_halo: leave the last yielded record unconsumed
 * so that the next window starts with it -- as record 0 it closes no unit (its unit was emitted with this
 * window), but it is the forward mate the next record is compared with.  At most max_records are yielded.
 */
int xmh_parse(xmh_parser *p, const char *buf1, uint64_t len1, int eof1, const char *buf2, uint64_t len2, int eof2,
              int score_mode, int paired, int skip_repeated, int keep_halo, uint64_t max_records, xmh_block *out);

/*
 * Text of one output bin for the block parsed last: for every unit index in idx[0..n_idx) (ascending, as
 * xm_compact returns them) the lines the reference prints -- bins 0, 2, 5: file-1 lines; 1, 3: file-2 lines;
 * 4: file-1 lines then file-2 lines; a paired unit covers records idx-1 and idx -- each as '\t'.join(fields)
 * + '\n'.  Two-call protocol: with out == NULL only *out_len is computed.  buf1/buf2 must be the windows given
 * to the xmh_parse call that produced the block.
 */
int xmh_emit(xmh_parser *p, const char *buf1, const char *buf2, int paired, int bin,
             const uint32_t *idx, uint64_t n_idx, char *out, uint64_t out_cap, uint64_t *out_len);

/* ---- blocks stripped on the GPU (include/xenomapper_strip.h) -------------------------------------------------
 * xmh_copy: memcpy by the parser's threads (the window of a mapped file into a page-locked staging buffer: the pages are
 * mapped in bulk first, then copied at the memory system's rate rather than one core's).
 * xmh_adopt_lines: make the line tables of a block that was stripped elsewhere the parser's current block, so that
 * xmh_emit writes its units (window offsets as 32-bit values; of the flags only XMH_LINE_NORMAL is looked at). */
int xmh_copy(xmh_parser *p, void *dst, const void *src, uint64_t n);
/* xmh_pread: bytes [offset, offset + n) of an open file read into dst by the parser's threads (pread(2) on slices: the
 * kernel copies from the page cache at the memory system's rate and no page of the file is ever mapped).
 * XMH_ERR_INVALID_ARG when the file ends before offset + n or a read fails. */
int xmh_pread(xmh_parser *p, int fd, uint64_t offset, void *dst, uint64_t n);
int xmh_adopt_lines(xmh_parser *p, uint64_t n_records,
                    const uint32_t *line_off1, const uint32_t *line_len1, const uint32_t *norm_len1, const uint8_t *line_flags1,
                    const uint32_t *line_off2, const uint32_t *line_len2, const uint32_t *norm_len2, const uint8_t *line_flags2);

/* ---- BAM input ------------------------------------------------------------------------
 * The reference reads BAM by piping it through `samtools view` (get_bam_header, bam_lines, getBamReadPairs,
 * xenomapper.py:48-93) and then handles the text as SAM.  xmh_bam_* is that pipe, natively: BGZF blocks are
 * inflated in parallel and every alignment is printed as the SAM line `samtools view` prints, so the SAM path
 * (xmh_parse / kernels / xmh_emit) applies unchanged.  `data` must stay valid while the reader is open. */
typedef struct xmh_bam xmh_bam;
int xmh_bam_open(const uint8_t *data, uint64_t len, int n_threads, xmh_bam **out);
/* header, reference names, xmh_bam_records_start and xmh_bam_print only (the GPU BAM path walks the blocks itself): the file's
 * blocks are not indexed, xmh_bam_read* on this handle return XMH_ERR_INVALID_ARG */
int xmh_bam_open_header(const uint8_t *data, uint64_t len, int n_threads, xmh_bam **out);
int xmh_bam_close(xmh_bam *b);
/* The header as `samtools view -H` prints it (text owned by the reader, not NUL-terminated). */
int xmh_bam_header(xmh_bam *b, const char **text, uint64_t *len);
/* Append the SAM lines of the next alignments to dst (whole lines only, at most cap bytes; cap must hold at
 * least one line).  *eof = 1 once every alignment has been returned. */
int xmh_bam_read(xmh_bam *b, char *dst, uint64_t cap, uint64_t *written, int *eof);

/*
 * BAM already holds what the stripper would have to dig out of the text again: AS / XS / ZS / NM as typed values and the
 * CIGAR as len << 4 | op words.  xmh_bam_read_pre is xmh_bam_read that also describes every line it
 * writes, in order -- the tag_func plugins' rules (xenomapper.py:186-190 substring match with the duplicate error, :247
 * first NM match, :251 the CIGAR operations "MIDNSHP=X") applied to the typed fields:
 *   a tag is matched by the field of that name and by any Z / H field whose printed text contains the two letters; an
 *   integer-typed match inside [-(2^31-1), 2^31-1] gives the value, any other match XMH_EX_NONINT (the host re-reads that
 *   record's text with the reference-equivalent plugin), two matches XMH_EX_DUP (NM: the first match counts).
 * XMH_PRE_WEIRD marks a line the text rules might split differently (white space, control or non-ASCII bytes in a name,
 * tag or string value): xmh_parse_pre then answers XMH_NEED_TEXT and the window goes through xmh_parse as before.
 * pre / ops: caller arrays of pre_cap / ops_cap entries; the call stops early (as when dst is full) rather than overrun them.
 */
#define XMH_PRE_WEIRD 1
typedef struct {
    uint32_t line_len;                      /* bytes of the line without its '\n' */
    uint16_t name_len;                      /* bytes of QNAME, the first field */
    uint8_t  flags;                         /* XMH_PRE_* */
    uint8_t  ex_as, ex_xs, ex_zs, ex_nm;    /* 0, XMH_EX_NONINT or XMH_EX_DUP */
    uint8_t  pad_;
    int32_t  as, xs, zs, nm;                /* INT32_MIN: no field matches the tag */
    uint32_t n_ops, ops_at;                 /* CIGAR operations with codes 0..8, in order: ops[ops_at .. ops_at + n_ops) */
} xmh_pre;
int xmh_bam_read_pre(xmh_bam *b, char *dst, uint64_t cap, uint64_t *written, int *eof,
                     xmh_pre *pre, uint64_t pre_cap, uint64_t *n_pre, uint32_t *ops, uint64_t ops_cap, uint64_t *n_ops);

/*
 * The host half of BAM input decoded ON THE GPU (include/xenomapper_bgzf.h, xm_bamdev_*): the device inflates the blocks and
 * strips the records; what the writer needs is the SAM text of the records, printed here from the inflated bytes.
 * xmh_bam_records_start: inflated offset of the first alignment record (behind the magic, the header text and the
 * reference list), right after xmh_bam_open.  xmh_bam_print: records rec_off[0 .. n) of `raw` (each offset is the record's
 * block_size word) printed as `samtools view` prints them, back to back into dst (n lines, each ending in '\n'), in
 * parallel; line_off / line_len (n entries each, caller arrays): where every line begins in dst and its length without
 * the terminator.  *written = bytes the text takes; XMH_ERR_INVALID_ARG with *written set when cap is too small (or the
 * text passes 4 GiB), XMH_ERR_BAD_BAM for a malformed record.  Reference names come from b's header.
 * sparse != 0: every thread prints straight into its own stretch of dst, sized for the worst case (5 x the record bytes):
 * one pass, no copy -- the lines are where line_off says, with gaps between the threads' stretches (what the writer needs;
 * a caller that wants the text itself takes sparse = 0); *written = the room that takes.
 * wanted (may be NULL): one byte per record, 0 = do not print it (line_len 0).  The main loops print a unit's lines of ONE
 * file only (primary bins file 1, secondary bins file 2, xenomapper.py:423-448) -- with the bins known before the text is
 * printed, half the records need no text at all.
 */
int xmh_bam_records_start(xmh_bam *b, uint64_t *inflated_offset);
/* The record chain of an inflated window followed on the host (files whose BGZF blocks do not begin with a record: the
 * device walks block by block and cannot): records whose block_size word begins at or behind `start` and that end inside
 * the window; rec_off may be NULL (count only); *stop = first byte not covered by a complete record. */
int xmh_bam_walk(const uint8_t *raw, uint64_t len, uint64_t start, uint32_t *rec_off, uint64_t cap, uint64_t *n_records, uint64_t *stop);
int xmh_bam_print(xmh_bam *b, const uint8_t *raw, const uint32_t *rec_off, uint64_t n, const uint8_t *wanted, char *dst, uint64_t cap,
                  uint32_t *line_off, uint32_t *line_len, int sparse, uint64_t *written);

/* xmh_parse on windows of text that xmh_bam_read_pre wrote, without tokenising it again: pre1 / pre2 describe the lines
 * from the first byte of buf1 / buf2 on (entries past the window are ignored), ops1 / ops2 are the arrays (n_ops1 / n_ops2
 * words) their ops_at refer to.  Same results as xmh_parse, or XMH_NEED_TEXT (nothing parsed) when a line is marked
 * XMH_PRE_WEIRD; XMH_ERR_INVALID_ARG when a description points outside its operation array (checked, since ABI 4: the
 * arrays are rebased whenever windows are merged and cross the language boundary as raw pointers). */
int xmh_parse_pre(xmh_parser *p, const char *buf1, uint64_t len1, int eof1, const xmh_pre *pre1, uint64_t n_pre1, const uint32_t *ops1,
                  uint64_t n_ops1,
                  const char *buf2, uint64_t len2, int eof2, const xmh_pre *pre2, uint64_t n_pre2, const uint32_t *ops2,
                  uint64_t n_ops2,
                  int score_mode, int paired, int skip_repeated, int keep_halo, uint64_t max_records, xmh_block *out);

#ifdef __cplusplus
}
#endif
#endif /* XENOMAPPER_HOST_H */
