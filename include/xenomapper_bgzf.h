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

#ifdef __cplusplus
extern "C" {
#endif

#define XMB_ABI_VERSION 1

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

/* CRC-32 (the gzip polynomial) of every block's inflated bytes, for the comparison with the member trailers:
 * crc_out[b] for b < n_blocks, on the device.  One wave per block. */
int xm_bgzf_crc32_dev(xm_ctx *ctx, void *stream, const uint8_t *out, const xm_bgzf_block *blocks, uint64_t n_blocks,
                      uint32_t *crc_out);

const char *xm_bgzf_strerror(uint32_t block_status);

#ifdef __cplusplus
}
#endif
#endif /* XENOMAPPER_BGZF_H */
