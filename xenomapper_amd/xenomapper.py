"""Drop-in host interface of the MI355X xenograft read classifier.

Same names, argument meaning and error behaviour as the reference module
``xenomapper/xenomapper.py`` (genomematt/xenomapper v1.0.2) for the classification path:

    get_tag / get_tag_with_ZS_as_XS / get_cigarbased_AS_tag   (:176-256)  the tag_func plugins
    get_mapping_state                                          (:258-289)
    main_single_end / main_paired_end / conservative_main_paired_end  (:291-556)
    getReadPairs, get_sam_header, add_pg_tag, process_headers, output_summary, main (CLI)

What differs is where the work happens.  The host only does text work: it splits lines,
finds tag fields, compares read names, and writes lines.  Scores travel as structure-of-arrays
columns through the C ABI (include/xenomapper_hip.h) to HIP kernels on the GPU, which evaluate
the CIGAR score, the state function, the pair rules, category_counts and the stable split of
units into the six output bins.  There is no CPU fallback for any of that: without the HIP
library and a gfx950 device every function below that classifies raises.

Records are buffered in blocks (``BLOCK_RECORDS``); memory stays bounded, and output appears
block by block instead of line by line.  Everything already decided before an input error is
written before the exception propagates, as with the reference.
"""
from __future__ import annotations

import argparse
import contextlib
import errno
import fcntl
import io
import mmap
import math
import os
import re
import stat
import sys
import threading
import time
from collections import Counter
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _ffi

__version__ = "1.0.2"        # the reference version this is a drop-in for; written into @PG VN (:128)

STATE_NAMES = ("primary_specific", "secondary_specific", "primary_multi", "secondary_multi",
               "unresolved", "unassigned")
BLOCK_RECORDS = 1 << 17

_NEG_INF = float("-inf")
_ABSENT = _ffi.ABSENT
_I32_MAX = 2**31 - 1

# --------------------------------------------------------------------------------------------
# device context
# --------------------------------------------------------------------------------------------
_context = None


def default_context():
    """The process-wide classifier context (device from XENOMAPPER_DEVICE, else LOCAL_RANK, else 0)."""
    global _context
    if _context is None:
        dev = int(os.environ.get("XENOMAPPER_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _context = _ffi.Context(dev)
    return _context


def set_context(ctx):
    global _context, _stripper
    _context = ctx
    _stripper = None


_stripper = None


def _forget_stripper(stripper):
    """A stripper whose step failed (its buffers could not be allocated, a read into them failed) is dropped, so that the
    next run starts from a fresh one instead of inheriting a half-staged window.  Not closed here: the block before the
    failed one may still be classified from its other slot by the main thread; it closes itself with its last reference."""
    global _stripper
    if _stripper is stripper:
        _stripper = None


_bamdev = None


def default_bamdev():
    """The process-wide GPU BAM front end (xm_bamdev) of the default context."""
    global _bamdev
    ctx = default_context()
    if _bamdev is None or _bamdev.ctx is not ctx or not _bamdev._h:
        _bamdev = _ffi.BamDev(ctx)
    return _bamdev


def default_stripper():
    """The process-wide GPU column stripper of the default context (its page-locked staging buffers are kept)."""
    global _stripper
    ctx = default_context()
    if _stripper is None or _stripper.ctx is not ctx:
        _stripper = _ffi.Stripper(ctx)
    return _stripper


def release_buffers():
    """Give back what the process-wide GPU front ends hold between jobs: default_stripper() and default_bamdev() keep the largest
    page-locked staging / text / table buffers and device buffers any run so far asked for (two slots x two files each: several
    GB after one large BAM pair), so that the next run does not pay for allocating and page-locking them again.  A process that
    classifies now and then calls this in between: both front ends are destroyed (their slot streams handed back to the context,
    xm_workspace_release), the host-side text buffers of the BAM path dropped; the next file run builds fresh ones.  Not while a
    file run is in progress in another thread.  -> _ffi.pinned_bytes() afterwards."""
    global _stripper, _bamdev
    if _bamdev is not None:
        _bamdev.close()
        _bamdev = None
    if _stripper is not None:
        _stripper.close()
        _stripper = None
    _BAM_TEXT_BUFFERS.clear()
    return _ffi.pinned_bytes()


# --------------------------------------------------------------------------------------------
# SAM input / headers (host text work)
# --------------------------------------------------------------------------------------------

def get_sam_header(samfile):
    """Header lines of a seekable SAM stream; leaves it at the first record (ref :36-46)."""
    header = []
    while True:
        mark = samfile.tell()
        text = samfile.readline().strip("\n")
        if text[0] != "@":            # IndexError on a header-only file, as the reference
            break
        header.append(text)
    samfile.seek(mark)
    return header


def _bam_reader(bamfile, header_only=False):
    """Native BGZF/BAM decoder over a binary file object (no samtools needed).  A regular file is memory-mapped
    (nothing is read up front, nothing stays resident); anything else (a pipe, BytesIO) is read whole.
    header_only: the handle that knows the header alone -- the file's blocks are not indexed (walking a multi-gigabyte
    file's block headers through a mapping, and taking the mapping down again, was 0.3 s of the command line's 1.8 on 4.5 GB)."""
    from . import _host
    data = None
    try:
        fd = bamfile.fileno()
        st = os.fstat(fd)
        if stat.S_ISREG(st.st_mode) and st.st_size > 0:
            # mapped through the handle's own descriptor: the file the caller opened (not whatever its name points
            # at by now), whatever kind of name the handle has, and its read position is left alone
            data = np.frombuffer(mmap.mmap(fd, 0, access=mmap.ACCESS_READ), dtype=np.uint8)
    except (AttributeError, OSError, ValueError, TypeError, io.UnsupportedOperation):
        data = None
    if data is None:
        data = np.frombuffer(bamfile.read(), dtype=np.uint8)
        bamfile.seek(0)
    return _host.BamReader(data, header_only=header_only)


def get_bam_header(bamfile):
    """Header lines of a BAM file, as `samtools view -H` prints them (ref :48-54)."""
    reader = _bam_reader(bamfile, header_only=True)
    try:
        return [line for line in reader.header().split("\n") if line]
    finally:
        reader.close()


def bam_lines(f):
    """SAM text lines of the alignments of a BAM file, as `samtools view` prints them (ref :56-64)."""
    reader = _bam_reader(f)
    try:
        buf = np.empty(8 << 20, dtype=np.uint8)
        while not reader.eof:
            n = reader.read_into(buf, 0)
            if n == 0 and not reader.eof:                         # the next line alone is longer than the buffer
                if buf.shape[0] >= BAM_LINE_LIMIT:
                    raise ValueError("a BAM record prints as a SAM line longer than %d bytes" % BAM_LINE_LIMIT)
                buf = np.empty(2 * buf.shape[0], dtype=np.uint8)
                continue
            for line in bytes(buf[:n]).decode("ascii").splitlines(True):
                yield line
    finally:
        reader.close()


def _lockstep(next1, next2, skip_repeated_reads):
    rec1, rec2 = next1(), next2()
    while rec1 and rec2:
        assert rec1[0] == rec2[0]
        yield rec1, rec2
        name1, name2 = rec1[0], rec2[0]
        if skip_repeated_reads:
            while rec1 and rec2 and rec1[0] == name1:
                rec1 = next1()
            while rec1 and rec2 and rec2[0] == name2:
                rec2 = next2()
        else:
            rec1, rec2 = next1(), next2()


class _ReadPairs(object):
    """What getReadPairs / getBamReadPairs return: the reference's generator, which also remembers its sources.
    Handed untouched to main_single_end / main_paired_end / conservative_main_paired_end with a built-in tag_func,
    two regular files go through the C++ stripper and writer (_run_files) instead of being split line by line in
    Python -- same outputs, same exceptions."""

    def __init__(self, make, sources, skip_repeated_reads, bam):
        self._make = make
        self._it = None
        self.sources = sources
        self.skip_repeated_reads = skip_repeated_reads
        self.bam = bam

    @property
    def started(self):
        return self._it is not None

    def __iter__(self):
        return self

    def __next__(self):
        if self._it is None:
            self._it = self._make()
        return next(self._it)

    def file_backed(self):
        """(path1, path2, start offsets) when both sources are distinct regular files positioned at a byte offset
        this module can trust, else None."""
        if self.started or os.environ.get("XENOMAPPER_PYTHON_READER") or self.sources[0] is self.sources[1]:
            return None
        paths, starts = [], []
        for f in self.sources:
            name = getattr(f, "name", None)
            if not isinstance(name, str) or not os.path.isfile(name):
                return None
            try:
                pos = f.tell()
            except (OSError, ValueError):
                return None
            if self.bam:
                if not isinstance(f, (io.BufferedReader, io.FileIO)) or pos != 0:
                    return None
            else:
                # a text-mode tell() is the byte offset only while the decoder holds no state (then it is < size);
                # universal newlines is the mode the stripper implements, so a '\r' seen so far rules the file out
                if not isinstance(f, io.TextIOWrapper) or f.newlines not in (None, "\n"):
                    return None
                if (f.encoding or "").lower().replace("-", "").replace("_", "") not in ("utf8", "ascii", "usascii", "latin1", "iso88591"):
                    return None
                if not 0 <= pos <= os.path.getsize(name):
                    return None
            paths.append(name)
            starts.append(pos)
        return paths[0], paths[1], starts


def getReadPairs(sam1, sam2, skip_repeated_reads=False):
    """Yield (fields1, fields2) for the same read from two SAM streams (ref :95-118).

    Fields are split on any whitespace; iteration ends at the first blank line or EOF of either
    stream; names must agree (AssertionError); with skip_repeated_reads each stream skips further
    lines that carry the name just yielded."""
    return _ReadPairs(lambda: _lockstep(lambda: sam1.readline().strip("\n").split(),
                                        lambda: sam2.readline().strip("\n").split(), skip_repeated_reads),
                      (sam1, sam2), skip_repeated_reads, False)


def _bam_lockstep(bamfile1, bamfile2, skip_repeated_reads):
    it1, it2 = bam_lines(bamfile1), bam_lines(bamfile2)

    class _Done(Exception):
        pass

    def nxt(it):
        def f():
            try:
                return next(it).strip("\n").split()
            except StopIteration:
                raise _Done()
        return f
    try:
        for pair in _lockstep(nxt(it1), nxt(it2), skip_repeated_reads):
            yield pair
    except _Done:
        return


def getBamReadPairs(bamfile1, bamfile2, skip_repeated_reads=False):
    """As getReadPairs for BAM input (ref :66-93); decoded natively instead of through samtools."""
    return _ReadPairs(lambda: _bam_lockstep(bamfile1, bamfile2, skip_repeated_reads), (bamfile1, bamfile2),
                      skip_repeated_reads, True)


def add_pg_tag(sam_header_list, comment=None):
    """Header plus an @PG line for this program, chained to a trailing @PG, and an optional @CO
    (ref :120-131)."""
    header = list(sam_header_list)
    if any(entry[0] != "@" for entry in header):
        raise ValueError("Incorrect SAM header format :\n{0}".format("\n".join(header)))
    chain = ""
    if header[-1][:3] == "@PG":
        ident = [tok for tok in header[-1].split() if tok[:2] == "ID"][0]
        chain = "PP" + ident[2:] + "\t"
    header.append("@PG\tID:Xenomapper\tPN:Xenomapper\t" + chain + "VN:{0}".format(__version__))
    if comment:
        header.append("@CO\t" + comment)
    return header


_HEADER_PLAN = (            # sink keyword, which input's header, @CO text   (ref :151-173)
    ("primary_specific", 0, "species specific reads"),
    ("secondary_specific", 1, "species specific reads"),
    ("primary_multi", 0, "species specific multimapping reads"),
    ("secondary_multi", 1, "species specific multimapping reads"),
    ("unassigned", 0, "reads that could not be assigned"),
    ("unresolved", 0, "reads that could not be resolved"),
)


def process_headers(file1, file2, primary_specific=sys.stdout, secondary_specific=None, primary_multi=None,
                    secondary_multi=None, unassigned=None, unresolved=None, bam=False):
    """Read both headers and write each open sink its header (ref :133-174).  primary_specific is
    written unconditionally; secondary bins get file2's header, the rest file1's."""
    reader = get_bam_header if bam else get_sam_header
    headers = (reader(file1), reader(file2))
    sinks = dict(primary_specific=primary_specific, secondary_specific=secondary_specific,
                 primary_multi=primary_multi, secondary_multi=secondary_multi, unassigned=unassigned,
                 unresolved=unresolved)
    for key, which, comment in _HEADER_PLAN:
        if key == "primary_specific" or sinks[key]:
            print("\n".join(add_pg_tag(headers[which], comment=comment)), file=sinks[key])


def output_summary(category_counts, outfile=sys.stderr):
    """Markdown table of the category counts (ref :558-566)."""
    print("-" * 80, file=outfile)
    print("Read Count Category Summary\n", file=outfile)
    print("|       {0:45s}|     {1:10s}  |".format("Category", "Count"), file=outfile)
    print("|:", "-" * 50, ":|:", "-" * 15, ":|", sep="", file=outfile)
    for category in sorted(category_counts):
        print("|  {0:50s}|{1:15d}  |".format(str(category), category_counts[category]), file=outfile)
    print(file=outfile)


# --------------------------------------------------------------------------------------------
# tag_func plugins
# --------------------------------------------------------------------------------------------

def _field_with(sam_line, tag):
    """The single optional field containing `tag` as a substring, or None (ref :186-190)."""
    found = None
    for opt in sam_line[11:]:
        if tag in opt:
            if found is not None:
                raise ValueError("SAM line has multiple values of {0}: {1}".format(tag, sam_line))
            found = opt
    return found


def get_tag(sam_line, tag="AS"):
    """Value of a numeric optional field as float, -inf when absent (ref :176-191)."""
    opt = _field_with(sam_line, tag)
    if opt is None:
        return _NEG_INF
    return float(opt.split(":")[-1])


def get_tag_with_ZS_as_XS(sam_line, tag="AS"):
    """get_tag, with requests for XS answered from ZS (HISAT; ref :193-206)."""
    return get_tag(sam_line, "ZS" if tag == "XS" else tag)


_CIGAR_OP = re.compile(r"([0-9]+)([MIDNSHPX=])")
_OP_CODE = {c: i for i, c in enumerate("MIDNSHP=X")}


def _cigar_columns(sam_line):
    """(NM or None, packed ops) of one line, the pre-parsed form the CIGAR kernel consumes.
    NM is the first field containing 'NM' (ref :247-250); ops are every <digits><op> match in the
    CIGAR column (ref :251), packed BAM-style len<<4|op."""
    nm = None
    for opt in sam_line[11:]:
        if "NM" in opt:
            nm = int(opt.split(":")[-1])
            break
    if nm is None:
        return None, ()
    if not -_I32_MAX <= nm <= _I32_MAX:
        raise OverflowError("NM value %d does not fit the int32 column" % nm)
    ops = []
    for length, op in _CIGAR_OP.findall(sam_line[5]):
        length = int(length)
        if length >= 1 << 28:
            raise OverflowError("CIGAR operation length %d does not fit the packed column" % length)
        ops.append((length << 4) | _OP_CODE[op])
    return nm, ops


def get_cigarbased_AS_tag(sam_line, tag="AS"):
    """AS rebuilt from CIGAR and NM: -6*NM -5*(#I+#D) -3*(sumI+sumD) -2*sumS; other tags as get_tag
    (ref :228-256).  The arithmetic runs in the CIGAR kernel (xm_cigar_scores)."""
    if tag != "AS":
        return get_tag(sam_line, tag)
    nm, ops = _cigar_columns(sam_line)
    if nm is None:
        return _NEG_INF
    off = np.array([0, len(ops)], dtype=np.uint32)
    out = default_context().cigar_scores(np.array([nm], dtype=np.int32), off, np.array(ops, dtype=np.uint32))
    return int(out[0])


def get_mapping_state(AS1, XS1, AS2, XS2, min_score=float("-inf")):
    """Name of the mapping state of one read (ref :258-289), evaluated by the classify kernel in
    the reference's own arithmetic (binary64)."""
    cols = [np.array([float(v)], dtype=np.float64) for v in (AS1, XS1, AS2, XS2)]
    code, _ = default_context().classify_f64(_ffi.MODE_SE, *cols, np.array([1], dtype=np.uint64), float(min_score))
    state = int(code[0])
    if state > 5:
        raise RuntimeError("Error in processing logic with values {0} ".format((AS1, XS1, AS2, XS2)))
    return STATE_NAMES[state]


# --------------------------------------------------------------------------------------------
# block engine behind the three main loops
# --------------------------------------------------------------------------------------------

def _floor_min_score(m):
    if m == _NEG_INF:
        return _ABSENT
    if m == -_NEG_INF:
        return _I32_MAX
    return int(max(_ABSENT, min(_I32_MAX, math.floor(m))))


def _to_int_column(values):
    """float scores -> int32 column, or None when a value is not representable (then binary64 is used)."""
    arr = np.asarray(values, dtype=np.float64)
    present = arr != _NEG_INF
    ok = np.isfinite(arr[present]).all() and (np.abs(arr[present]) <= _I32_MAX).all() \
        and (arr[present] == np.floor(arr[present])).all()
    if not ok:
        return None
    out = np.full(arr.shape[0], _ABSENT, dtype=np.int32)
    out[present] = arr[present].astype(np.int64).astype(np.int32)
    return out


class _Block(object):
    __slots__ = ("rec1", "rec2", "flags")

    def __init__(self):
        self.rec1, self.rec2, self.flags = [], [], []


def _score_block(block, needed, tag_func, cigar_mode):
    """Score columns of a block.  Returns (columns dict, n_scored, error): records are scored in
    index order, each as file-1 AS, file-1 XS, file-2 AS, file-2 XS (the order of ref :408-415);
    on an exception scoring stops and n_scored is the index of the failing record."""
    n = len(block.rec1)
    vals = [[_NEG_INF] * n for _ in range(4)]
    nm = [[None] * n, [None] * n]
    ops = [[()] * n, [()] * n]
    err = None
    scored = n
    for i in range(n):
        if not needed[i]:
            continue
        try:
            for f, rec in ((0, block.rec1[i]), (1, block.rec2[i])):
                if cigar_mode:
                    nm[f][i], ops[f][i] = _cigar_columns(rec)
                    vals[2 * f + 1][i] = get_tag(rec, "XS")
                else:
                    vals[2 * f][i] = tag_func(rec, tag="AS")
                    vals[2 * f + 1][i] = tag_func(rec, tag="XS")
        except Exception as exc:      # input error: classify what precedes it, then re-raise
            err = exc
            scored = i
            break
    return vals, nm, ops, scored, err


def _cigar_csr(nm, ops, n):
    counts = np.fromiter((len(o) for o in ops[:n]), dtype=np.uint32, count=n)
    off = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(counts, out=off[1:])
    flat = np.fromiter((v for o in ops[:n] for v in o), dtype=np.uint32, count=int(off[-1]))
    nm_col = np.fromiter((_ABSENT if v is None else v for v in nm[:n]), dtype=np.int32, count=n)
    return nm_col, off, flat


def _classify_block(ctx, mode, block, n, vals, nm, ops, cigar_mode, min_score):
    """-> (code u8[n], idx u32[units], bin_offsets u64[8], counts u64[64])"""
    flags = np.asarray(block.flags[:n], dtype=np.uint8)
    bits = np.packbits(np.concatenate([flags, np.zeros((-n) % 64, dtype=np.uint8)]), bitorder="little").view(np.uint64)
    integral = min_score == min_score
    if cigar_mode:
        csr = [_cigar_csr(nm[f], ops[f], n) for f in (0, 1)]
        xs = [_to_int_column(vals[c][:n]) for c in (1, 3)]
        if integral and xs[0] is not None and xs[1] is not None:
            # AS is synthesised inside the classify kernel (fused K3 + K1)
            return ctx.classify_compact_cigar(mode, csr[0][0], csr[0][1], csr[0][2], xs[0],
                                              csr[1][0], csr[1][1], csr[1][2], xs[1], bits, _floor_min_score(min_score))
        # a non-integral XS or a NaN threshold: CIGAR kernel, then the binary64 classify kernel
        fcols = []
        for f in (0, 1):
            a = ctx.cigar_scores(*csr[f])
            fcols.append(np.where(a == _ABSENT, _NEG_INF, a.astype(np.float64)))
            fcols.append(np.asarray(vals[2 * f + 1][:n], dtype=np.float64))
        return ctx.classify_compact(mode, *fcols, bits, float(min_score))
    int_cols = [_to_int_column(vals[c][:n]) for c in range(4)]
    if integral and all(col is not None for col in int_cols):
        return ctx.classify_compact(mode, *int_cols, bits, _floor_min_score(min_score))
    fcols = [np.asarray(vals[c][:n], dtype=np.float64) for c in range(4)]
    return ctx.classify_compact(mode, *fcols, bits, float(min_score))


def _add_counts(totals, key_order, paired, code, counts):
    """Add a block's category_counts to the run's Counter.  Keys keep the order in which the reference's Counter would
    have met them (first occurrence in the input): only categories not seen in an earlier block need looking for."""
    new = []
    for c in np.flatnonzero(counts).tolist():
        key = STATE_NAMES[c] if not paired else (STATE_NAMES[c >> 3], STATE_NAMES[c & 7])
        if key not in totals:
            new.append((int(np.argmax(code == c)), key))            # first record holding category c
        totals[key] += int(counts[c])
    key_order.extend(key for _, key in sorted(new))


def _lines_of(block, mode, b, i):
    """Records a unit contributes to bin b (ref :332-350, :423-448, :521-550)."""
    if mode == _ffi.MODE_SE:
        ones, twos = (block.rec1[i],), (block.rec2[i],)
    else:
        ones, twos = (block.rec1[i - 1], block.rec1[i]), (block.rec2[i - 1], block.rec2[i])
    if b in (0, 2, 5):
        return ones
    if b in (1, 3):
        return twos
    return ones + twos


def _emit_block(block, mode, code, idx, off, sinks, limit):
    """Write the units with index < limit.  Bin by bin from the compacted lists when the sinks are
    distinct objects; in unit order when two bins share a sink (so interleaving matches the reference)."""
    active = [s for s in sinks if s]
    if len(set(id(s) for s in active)) == len(active):
        for b in range(6):
            sink = sinks[b]
            if not sink:
                continue
            seg = idx[int(off[b]):int(off[b + 1])]
            if limit is not None:
                seg = seg[seg < limit]
            if len(seg) == 0:
                continue
            parts = []
            for i in seg.tolist():
                for rec in _lines_of(block, mode, b, i):
                    parts.append("\t".join(rec))
                    parts.append("\n")
            sink.write("".join(parts))
    else:
        units = np.flatnonzero(code != _ffi.NO_UNIT)
        bins = np.empty(code.shape[0], dtype=np.uint8)
        for b in range(6):
            bins[idx[int(off[b]):int(off[b + 1])]] = b
        for i in units.tolist():
            if limit is not None and i >= limit:
                break
            b = int(bins[i])
            if sinks[b]:
                for rec in _lines_of(block, mode, b, i):
                    print("\t".join(rec), file=sinks[b])


def _run(mode, readpairs, sinks, min_score, tag_func):
    if isinstance(readpairs, _ReadPairs) and tag_func in (get_tag, get_tag_with_ZS_as_XS, get_cigarbased_AS_tag):
        backed = readpairs.file_backed()
        if backed is not None:
            try:
                return _run_files(mode, backed[0], backed[1], sinks, min_score, tag_func, readpairs.skip_repeated_reads,
                                  bam=readpairs.bam, starts=None if readpairs.bam else backed[2])
            finally:
                for f in readpairs.sources:                       # the records have been consumed
                    f.seek(0, 2)
    ctx = default_context()
    paired = mode != _ffi.MODE_SE
    cigar_mode = tag_func is get_cigarbased_AS_tag
    totals = Counter()
    key_order = []

    def flush(block, pending_error=None):
        """Classify + emit one block; raises the pending input error (or the state error) afterwards."""
        n = len(block.rec1)
        if n:
            if paired:
                needed = [False] * n
                for i in range(1, n):
                    if block.flags[i]:
                        needed[i - 1] = needed[i] = True
            else:
                needed = [True] * n
            vals, nm, ops, scored, err = _score_block(block, needed, tag_func, cigar_mode)
            if err is not None:
                n = scored                      # units closing at index >= scored are not reached
                pending_error = err
            if n:
                code, idx, off, counts = _classify_block(ctx, mode, block, n, vals, nm, ops, cigar_mode, min_score)
                limit = None
                state_error = None
                if int(off[7]) != int(off[6]):      # a unit fell through every branch (ref :289)
                    first = int(idx[int(off[6]):int(off[7])].min())
                    limit = first
                    c = int(code[first])
                    j = first - 1 if (paired and (c >> 3) > 5) else first
                    state_error = RuntimeError("Error in processing logic with values {0} ".format(
                        tuple(vals[k][j] for k in range(4))))
                _emit_block(block, mode, code, idx, off, sinks, limit)
                if state_error is not None:
                    raise state_error
                _add_counts(totals, key_order, paired, code, counts)
        if pending_error is not None:
            raise pending_error

    block = _Block()
    prev_name = None
    iterator = iter(readpairs)
    while True:
        try:
            rec1, rec2 = next(iterator)
        except StopIteration:
            break
        except BaseException as exc:            # the reader failed (e.g. names disagree): drain, then raise
            flush(block, exc)
            raise
        if rec1[0] != rec2[0]:
            flush(block, AssertionError())
        block.rec1.append(rec1)
        block.rec2.append(rec2)
        if paired:
            # a record closes a unit when its name equals the previous record's (ref :402-405)
            block.flags.append(1 if (prev_name is not None and prev_name == rec1[0]) else 0)
            prev_name = rec1[0]
        else:
            block.flags.append(1)
        if len(block.rec1) >= BLOCK_RECORDS:
            flush(block)
            nxt = _Block()
            if paired:                          # the last record may be the forward mate of the next
                nxt.rec1.append(block.rec1[-1])
                nxt.rec2.append(block.rec2[-1])
                nxt.flags.append(0)
            block = nxt
    flush(block)
    ordered = Counter()
    for key in key_order:
        ordered[key] = totals[key]
    return ordered


# --------------------------------------------------------------------------------------------
# file-to-file fast path: C++ column stripper -> GPU -> C++ line writer
# --------------------------------------------------------------------------------------------
BAM_LINE_LIMIT = 1 << 32            # bam_lines gives up on a single SAM line longer than this (the stripper's own limit)
FILE_WINDOW_BYTES = int(os.environ.get("XENOMAPPER_WINDOW_MB", "128")) << 20     # bytes of each file parsed per block
FILE_MAX_RECORDS = 1 << 22
STAGE_PIECE = 16 << 20              # bytes copied into page-locked memory per upload of the GPU stripper
SAM_TAIL_ROOM = 16 << 20            # room in front of bytes read ahead for the tail of the window before them (at most a window)


def _record_start(raw):
    """Byte offset of the first record of a SAM file held in `raw` (uint8 array): the first line that does
    not start with '@' (what get_sam_header leaves the stream at, ref :36-46)."""
    pos, size = 0, raw.shape[0]
    while pos < size and raw[pos] == 0x40:
        window = raw[pos:pos + (1 << 16)]
        while True:
            hits = np.flatnonzero((window == 10) | (window == 13))
            if hits.shape[0] or pos + window.shape[0] >= size:
                break
            window = raw[pos:pos + 4 * window.shape[0]]
        if hits.shape[0] == 0:
            return size
        end = pos + int(hits[0])
        pos = end + 1
        if raw[end] == 13 and pos < size and raw[pos] == 10:
            pos += 1
    return pos


_ASCII_SUPERSETS = ("utf8", "ascii", "usascii", "latin1", "iso88591", "cp1252", "ansix3.41968")


def _write_bytes(sink, data):
    """Append ASCII bytes to a text sink: through its binary buffer when it has one and its encoding leaves ASCII
    alone, else as decoded text.  (Writing one bin's text with several positional writes, or from a writer thread,
    was tried and bought nothing: buffered writes to one file serialise on its inode lock, ~5 GB/s on tmpfs, and the
    host threads are already at the CPU quota.)"""
    if data is None or len(data) == 0:
        return
    raw = getattr(sink, "buffer", None)
    enc = (getattr(sink, "encoding", None) or "").lower().replace("-", "").replace("_", "")
    if raw is not None and enc in _ASCII_SUPERSETS:
        sink.flush()
        raw.write(memoryview(data))
    else:
        sink.write(bytes(data).decode("ascii"))


MMAP_EMIT_MIN_BYTES = 1 << 20        # below this one buffered write is cheaper than mapping the file
# bytes per background extension job (0: the whole extension in one).  Pieces let the writer go on as soon as the bytes it needs
# are there, but every fallocate call beside the threads that fill the pages slows those: 32 MB pieces 5.9 - 7.0 M pairs/s, 8 MB
# 5.5 - 6.0, one piece 7.7 - 7.9 (SAM text in, six files on tmpfs out, one box, profiles/r06_ab_ahead_piece.txt)
AHEAD_PIECE = (int(os.environ.get("XENOMAPPER_AHEAD_PIECE_MB", "0")) << 20) or (1 << 62)
MAP_AHEAD = int(os.environ.get("XENOMAPPER_MAP_AHEAD_MB", "1024")) << 20       # the writer's mapping of an output file reaches this far from where it was made (0: a mapping per call)
AHEAD_MOST = int(os.environ.get("XENOMAPPER_AHEAD_MOST_MB", "1024")) << 20     # an output file is never extended further than this past its content
# 1: extend the output files towards the size the run predicts from the fraction of the input it has read, instead of twice the last
# call's bytes ahead.  Measured on one (slow) box, alternating (profiles/r06_ab_ahead_predict.txt): SAM text in 6.0 - 7.2 against
# 5.3 - 6.1 M pairs/s, BAM in 10.5 - 12.4 against 13.6 - 15.9 -- the further the extension runs ahead, the more it takes from the
# threads filling the pages behind it.  Not the default.
AHEAD_PREDICT = os.environ.get("XENOMAPPER_AHEAD_PREDICT", "0") == "1"
# (how far ahead: the further, the colder the pages when the writer's threads reach them -- x 2 fills at 0.22 - 0.31 s per 3.4 GB,
# x 0.5 .. 1.5 at 0.14 - 0.18, with more waiting for the extension in exchange; x 1: profiles/r06_ab_ahead_factor.txt)
AHEAD_FACTOR = float(os.environ.get("XENOMAPPER_AHEAD", "1"))      # output files are kept this many calls' worth of bytes longer than their content


_EMIT_CLOCK = {}            # seconds inside _emit_into_file by step, since the run began (shown with the phases of the run)


_libc_fallocate = None
_libc_map = None


def _map_file(fd, length, offset):
    """mmap(2) of [offset, offset + length) of a file, shared and writable -> address.  The system call itself (not the mmap
    module, which keeps a duplicate of the descriptor per mapping): the writer holds one long mapping per output file."""
    global _libc_map
    import ctypes
    if _libc_map is None:
        lib = ctypes.CDLL(None, use_errno=True)
        lib.mmap.restype = ctypes.c_void_p
        lib.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64]
        lib.munmap.restype = ctypes.c_int
        lib.munmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        _libc_map = lib
    addr = _libc_map.mmap(None, length, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_SHARED, fd, offset)
    if addr is None or addr == ctypes.c_void_p(-1).value:
        e = ctypes.get_errno()
        raise OSError(e, os.strerror(e))
    return addr


def _unmap_file(addr, length):
    if _libc_map is not None and addr:
        _libc_map.munmap(addr, length)


def _fallocate(fd, offset, length):
    """fallocate(2) itself (mode 0), not glibc's posix_fallocate: on a file system that cannot preallocate the latter
    silently falls back to writing a zero into every block, which races with anything else that writes to the file (the
    ahead-extension runs beside the sink); the system call fails with EOPNOTSUPP instead and the caller extends sparsely."""
    global _libc_fallocate
    if _libc_fallocate is None:
        import ctypes
        try:
            fn = ctypes.CDLL(None, use_errno=True).fallocate
            fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64]
            fn.restype = ctypes.c_int
            _libc_fallocate = fn
        except (OSError, AttributeError):
            _libc_fallocate = False
    if _libc_fallocate is False:
        os.posix_fallocate(fd, offset, length)
        return
    import ctypes
    while _libc_fallocate(fd, 0, offset, length) != 0:
        e = ctypes.get_errno()
        if e != errno.EINTR:
            raise OSError(e, os.strerror(e))


class _AheadFile(object):
    """An output file that is kept LONGER than its content while a run is writing it: fallocate is one kernel thread
    instantiating pages (7 - 18 GB/s on tmpfs) and used to sit in front of every bin of every block; extended in the background,
    ahead of the writer, it is off the critical path.  The extension is a job in the helper pool towards a target the writer
    moves: XENOMAPPER_AHEAD (1) calls' worth of bytes past the content -- further ahead, the pages are cold again when the
    writer's threads fill them.  (Measured and not the default: the extension in pieces, XENOMAPPER_AHEAD_PIECE_MB, each a job
    of its own so that the writer waits for the piece it needs only; a target predicted from the fraction of the input read
    so far, XENOMAPPER_AHEAD_PREDICT, at most AHEAD_MOST bytes ahead.)  finish() cuts the file back to its content.  Until
    then the file ends in NUL bytes: a process killed between two blocks leaves them behind (the content in front of them is
    complete lines); only finish() -- reached on every exit of the run, exceptions included -- truncates."""

    def __init__(self, fd2, size):
        self.fd2, self.size, self.target = fd2, size, size
        self.busy = False                                 # a piece is queued or being allocated
        self.cv = threading.Condition()
        self.mapped = None                                # (address, file offset, length) of the writer's current mapping

    def window(self, pos, need, pool):
        """-> the address of file byte `pos`, of a mapping that holds [pos, pos + need).  One LONG mapping per file (MAP_AHEAD
        bytes from where it was made), kept from call to call: a mapping per bin and window meant a munmap per bin and window in
        the helper threads, and every mmap of the next bin waited for the one before to let go of the process's mapping lock --
        2.5 ms a call, a third of the writer's time (profiles/r06_ab_map_reuse.txt).  The mapping may reach beyond the file's
        end; only [pos, pos + need), which the caller has extended the file to, is touched."""
        m = self.mapped
        if m is None or pos < m[1] or pos + need > m[1] + m[2]:
            start = pos - pos % mmap.ALLOCATIONGRANULARITY
            length = max(pos + need - start, MAP_AHEAD)
            addr = _map_file(self.fd2, length, start)
            if m is not None:
                self._drop(m, pool)
            m = self.mapped = (addr, start, length)
        return m[0] + (pos - m[1])

    @staticmethod
    def _drop(m, pool):
        try:
            pool.submit(_unmap_file, m[0], m[2])           # (taking ~10^5 pages out of the page table costs as much as a third of their fill)
        except RuntimeError:
            _unmap_file(m[0], m[2])

    def unmap(self, pool=None):
        m, self.mapped = self.mapped, None
        if m is not None:
            if pool is not None:
                self._drop(m, pool)
            else:
                _unmap_file(m[0], m[2])

    def _piece(self, pool):
        with self.cv:
            start, n = self.size, min(AHEAD_PIECE, self.target - self.size)
        ok = n > 0
        if ok:
            try:
                _fallocate(self.fd2, start, n)             # EOPNOTSUPP, ENOSPC: no extension ahead; the one really needed will say so
            except OSError:
                ok = False
        with self.cv:
            if ok:
                self.size = max(self.size, start + n)
            else:
                self.target = self.size
            again = ok and self.size < self.target
            if again:
                try:
                    pool.submit(self._piece, pool)          # behind the other files' pieces and the unmapping jobs
                except RuntimeError:                        # the pool is shutting down
                    again = False
            self.busy = again
            self.cv.notify_all()

    def settle(self, upto=None):
        """Wait until the file is `upto` bytes long (None: stop extending, and wait for the piece in flight); -> the file's size.
        A size below `upto` = the background extension does not reach that far and is idle now: the caller extends the file itself."""
        with self.cv:
            if upto is None:
                self.target = self.size                   # no further pieces
            while self.busy and (upto is None or self.size < upto):
                self.cv.wait()
            return self.size

    def grew(self, size):
        """The caller extended the file itself (settle() had returned less than it needed)."""
        with self.cv:
            self.size = max(self.size, size)
            self.target = max(self.target, self.size)

    def cut(self, size):
        """The file was cut back to `size` (an emit that failed; no piece is in flight: settle() first)."""
        with self.cv:
            self.size = self.target = size

    def extend_later(self, pool, upto):
        with self.cv:
            self.target = max(self.target, upto)
            if not self.busy and self.target > self.size:
                self.busy = True
                try:
                    pool.submit(self._piece, pool)
                except RuntimeError:
                    self.busy = False

    def finish(self, sink, pool=None):
        self.settle()
        self.unmap(pool)
        try:
            sink.flush()
            os.ftruncate(self.fd2, sink.buffer.tell())
        finally:
            os.close(self.fd2)


def _emit_into_file(parser, paired, b, seg, sink, ahead=None, ahead_pool=None, ready=None, progress=None):
    """The text of one bin's units written by the writer's threads STRAIGHT into the output file: the file is extended,
    its new pages are mapped, and xmh_emit gathers the lines into them in parallel -- no intermediate buffer and no
    single write(2) stream (which tops out at ~5 GB/s on one file and made the file path write-bound).  Only for a
    regular file positioned at its end, with an ASCII-compatible encoding and at least MMAP_EMIT_MIN_BYTES to write;
    False = not handled (the caller writes through the sink as before).  XENOMAPPER_MMAP_EMIT=0 switches it off.
    ahead: {id(sink): _AheadFile} of the run -- with it the file stays extended past its content between calls (towards the size
    it will have if the rest of the input -- progress: the fraction read so far -- fills it at the same rate; without a
    fraction, twice the bytes of the last call), allocated by ahead_pool's threads while the next blocks are classified; the
    run cuts the files back when it ends (_AheadFile.finish).
    ready: the bin's text as it stands (uint8 array: the outputs of the window were gathered on the device, xm_bamdev_fetch_bins /
    xm_strip_fetch_bins) -- then the threads only copy it into the file's pages.  (Positional writes of the same ranges instead --
    one stream per file, or 8 MB chunks over 6 / 16 / 32 threads -- take 3.4 GB in 0.18 s when nothing else runs on the box
    (tools/probe_tmpfs_write.py) but 0.47-0.92 s inside a run, beside the readers and the link, where this route takes 0.28-0.39:
    profiles/r06_ab_sam_bins.txt.)"""
    if os.environ.get("XENOMAPPER_MMAP_EMIT") == "0":
        return False
    raw = getattr(sink, "buffer", None)
    enc = (getattr(sink, "encoding", None) or "").lower().replace("-", "").replace("_", "")
    if raw is None or enc not in _ASCII_SUPERSETS:
        return False
    t0 = time.perf_counter()
    if ready is not None:
        idx, need = None, int(ready.shape[0])
    else:
        idx, need = parser.emit_size(paired, b, seg)
    _EMIT_CLOCK["emit_size"] = _EMIT_CLOCK.get("emit_size", 0.0) + time.perf_counter() - t0
    state = ahead.get(id(sink)) if ahead is not None else None
    if need < MMAP_EMIT_MIN_BYTES:
        if state is not None:
            state.settle()          # the caller writes through the sink: not while the helper thread is allocating in the same file
        return False
    t0 = time.perf_counter()
    fd2, pos, size, opened = None, 0, 0, False
    try:
        sink.flush()
        pos = raw.tell()
        if state is None:
            fd = sink.fileno()
            st = os.fstat(fd)
            if not stat.S_ISREG(st.st_mode) or pos != st.st_size:
                return False
            # a mapping needs a descriptor opened for reading as well; sinks are usually write-only
            fd2 = os.open("/proc/self/fd/%d" % fd, os.O_RDWR)
            opened = True
            size = pos
        else:
            fd2, size = state.fd2, state.settle(pos + need)
        if pos + need > size:
            try:
                _fallocate(fd2, size, pos + need - size)           # extends the file AND allocates its pages in one go: the
            except OSError as e:                                   # threads then only copy (page by page faults cost 3x on tmpfs)
                # Only a file system that cannot preallocate may be extended sparsely instead.  Anything else -- no space
                # left, a quota, the file size limit -- must NOT be papered over: stores into a mapping whose pages cannot
                # be backed end in SIGBUS.  Give the range back and let the ordinary write raise the real OSError.
                if e.errno not in (errno.EOPNOTSUPP, errno.ENOSYS, errno.EINVAL):
                    raise
                os.ftruncate(fd2, pos + need)
            size = pos + need
        t_m = time.perf_counter()
        own = None                                     # (address, length) of a mapping made for this call alone
        if state is not None and MAP_AHEAD > 0 and ahead_pool is not None:
            dst = state.window(pos, need, ahead_pool)  # the file's long mapping (made now, if the last one ends before pos + need)
        else:
            start = pos - pos % mmap.ALLOCATIONGRANULARITY
            own = (_map_file(fd2, pos + need - start, start), pos + need - start)
            dst = own[0] + (pos - start)
        _EMIT_CLOCK["emit_map"] = _EMIT_CLOCK.get("emit_map", 0.0) + time.perf_counter() - t_m
    except (OSError, ValueError, AttributeError, io.UnsupportedOperation):
        if fd2 is not None:
            if state is not None:
                state.settle()                         # (no piece in flight while the file is cut)
            try:
                os.ftruncate(fd2, pos)                 # the file is as the sink left it; the caller writes the text
            except OSError:
                pass
            if state is not None:
                state.cut(pos)
            elif opened:
                os.close(fd2)
        return False
    t1 = time.perf_counter()
    _EMIT_CLOCK["emit_extend"] = _EMIT_CLOCK.get("emit_extend", 0.0) + t1 - t0
    try:
        if ready is not None:
            parser.copy(dst, ready, 0, need)
            wrote = need
        else:
            wrote = parser.emit_to(paired, b, idx, dst, need)
        assert wrote == need
        t0 = time.perf_counter()
        _EMIT_CLOCK["emit_fill"] = _EMIT_CLOCK.get("emit_fill", 0.0) + t0 - t1
    except BaseException:
        if own is not None:
            _unmap_file(*own)
        if state is not None:
            state.unmap()
            state.settle()
        os.ftruncate(fd2, pos)                         # nothing of this bin's text stays behind
        if state is not None:
            state.cut(pos)
        else:
            os.close(fd2)
        raise
    if own is not None:
        if ahead is not None and ahead_pool is not None:
            ahead_pool.submit(_unmap_file, *own)       # taking ~10^5 pages out of the page table costs as much as a third of the fill
        else:
            _unmap_file(*own)
    _EMIT_CLOCK["emit_unmap"] = _EMIT_CLOCK.get("emit_unmap", 0.0) + time.perf_counter() - t0
    raw.seek(pos + need)
    # (a descriptor in append mode writes at the END of the file whatever its position: it must never be longer than its content)
    if ahead is not None and ahead_pool is not None and (state is not None or not fcntl.fcntl(fd2, fcntl.F_GETFL) & os.O_APPEND
                                                           and not fcntl.fcntl(sink.fileno(), fcntl.F_GETFL) & os.O_APPEND):
        if state is None:
            state = ahead[id(sink)] = _AheadFile(fd2, size)
        state.grew(size)
        end = pos + need
        if progress is not None and 0.0 < progress < 1.0:
            upto = min(max(int(end / progress * 1.02), end + need // 2), end + AHEAD_MOST)
        elif progress is not None:
            upto = end                                 # the input has been read: nothing follows
        else:
            upto = end + int(AHEAD_FACTOR * need)
        state.extend_later(ahead_pool, upto)
    elif state is None:
        os.close(fd2)
    return True


def _fields_of(raw, block, f, k, pos):
    start = pos + int(block.line_off[f][k])
    return bytes(raw[start:start + int(block.line_len[f][k])]).decode("ascii").split()


def _resolve_exceptions(block, raws, pos, needed, tag_func, cigar_mode):
    """Values the C++ stripper would not vouch for (non-integers, duplicate tags, short lines ...) are
    re-evaluated with the text-level plugin on the original line, in record order.  Returns
    (patches {record: [AS1, XS1, AS2, XS2 as Python numbers]}, first_bad_record or None, error or None)."""
    patches, seen = {}, set()
    for k, _col, _kind in block.exc:
        if k in seen or not needed[k]:
            continue
        seen.add(k)
        try:
            vals = []
            for f in (0, 1):
                fields = _fields_of(raws[f], block, f, k, pos[f])
                if cigar_mode:
                    nm, ops = _cigar_columns(fields)              # may raise like the reference would
                    vals.append(("cigar", nm, ops))
                    vals.append(get_tag(fields, "XS"))
                else:
                    vals.append(tag_func(fields, tag="AS"))
                    vals.append(tag_func(fields, tag="XS"))
            patches[k] = vals
        except Exception as exc:
            return patches, k, exc
    return patches, None, None


LAST_FILE_PROFILE = {}      # wall seconds per phase of the last file-to-file run (window / parse overlap classify / emit / write)


class _PhaseClock(dict):
    """Wall time per phase of the file path (printed when XENOMAPPER_PROFILE is set)."""

    @contextlib.contextmanager
    def __call__(self, name):
        start = time.perf_counter()
        try:
            yield
        finally:
            self[name] = self.get(name, 0.0) + time.perf_counter() - start


def _quietly(fn, *args):
    """fn(*args) in a helper thread whose failure is not the run's: whoever needs what it was to prepare does it again and raises."""
    try:
        fn(*args)
    except Exception:                                                # noqa: BLE001
        pass


def _input_fraction(sources):
    """How far the run is through its input files (0 .. 1), or None when the sources do not say."""
    done = total = 0
    for src in sources:
        if isinstance(src, _SamSource):
            done, total = done + src.pos, total + int(src.raw.shape[0])
        elif isinstance(src, _GpuBamFile):
            done, total = done + src.cursor, total + int(src.data.shape[0])
        else:
            return None
    return min(max(done / total, 0.0), 1.0) if total else None


class _SamSource(object):
    """Record text of a SAM file: windows of a read-only memory map."""

    def __init__(self, path, start=None):
        self.path = path
        self.raw = np.memmap(path, dtype=np.uint8, mode="r") if os.path.getsize(path) else np.zeros(0, np.uint8)
        self.pos = _record_start(self.raw) if start is None else int(start)
        self.fd = None

    def fileno(self):
        """A descriptor for positional reads (the GPU stripper reads windows straight into page-locked memory)."""
        if self.fd is None:
            self.fd = os.open(self.path, os.O_RDONLY)
        return self.fd

    def window(self, want):
        n = min(want, self.raw.shape[0] - self.pos)
        return self.raw, self.pos, n, self.pos + n >= self.raw.shape[0]

    def advance(self, consumed, lines=0):
        self.pos += consumed

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None


class _BamSource(object):
    """Record text of a BAM file: SAM lines decoded into three buffers in rotation -- one holds the window whose
    lines the writer is still gathering, one the window being parsed, and a decoder thread fills the third with
    the text that follows.  Every buffer keeps HEAD bytes free in front of its decoded text: the unread tail of the
    previous window is copied there, so moving on to the next window costs no large copy."""

    HEAD = 1 << 20

    def __init__(self, path, n_threads=0):
        from . import _host
        self.path = path
        self.data = np.memmap(path, dtype=np.uint8, mode="r")
        self.reader = _host.BamReader(self.data, n_threads)
        self.bufs = [None, None, None]
        self.cur = 0
        self.start = self.end = self.HEAD
        self.advanced = False
        self.decoder = ThreadPoolExecutor(max_workers=1)
        self.ahead = None                     # Future of (buffer index, bytes decoded at [HEAD, HEAD + n), line descriptions)
        # What the decoder knows about the lines in [start, end), in order: chunks of (xmh_pre array, CIGAR operations) as
        # xmh_bam_read_pre returned them, the first `pre_skip` lines of the first chunk already consumed.  With them the
        # stripper does not tokenise the text again (xmh_parse_pre); XENOMAPPER_BAM_PRE=0 switches that off.
        self.pre_ok = os.environ.get("XENOMAPPER_BAM_PRE") != "0"
        self.pre = []
        self.pre_skip = 0

    def _buffer(self, k, room):
        if self.bufs[k] is None or self.bufs[k].shape[0] < self.HEAD + room:
            self.bufs[k] = np.empty(self.HEAD + room, dtype=np.uint8)
        return self.bufs[k]

    def _decode(self, buf, at):
        """Decode behind buf[:at] until the buffer is full or the file ends; returns (the new end, the line descriptions)."""
        chunks = []
        while not self.reader.eof:
            if self.pre_ok:
                got, pre, ops = self.reader.read_into_pre(buf, at)
                if got:
                    chunks.append((pre, ops))
            else:
                got = self.reader.read_into(buf, at)
            if got == 0:
                break                                              # the next line does not fit what is left
            at += got
        return at, chunks

    def _decode_ahead(self, k):
        end, chunks = self._decode(self.bufs[k], self.HEAD)
        return k, end - self.HEAD, chunks

    def _take_ahead(self):
        k, n, chunks = self.ahead.result()
        self.ahead = None
        self.pre.extend(chunks)                                    # the text decoded ahead always goes right behind the live text
        return self.bufs[k], n

    def pre_window(self):
        """(xmh_pre array, operation array) of the lines from the window's first byte on, or None without descriptions."""
        if not self.pre_ok:
            return None
        from . import _host
        if not self.pre:
            return np.zeros((0, _host.PRE_WORDS), dtype=np.uint32), np.zeros(0, dtype=np.uint32)
        if len(self.pre) == 1:
            return self.pre[0][0][self.pre_skip:], self.pre[0][1]      # a view: the operation offsets stay valid
        parts, ops, base = [], [], 0
        for k, (pre, op) in enumerate(self.pre):
            q = (pre[self.pre_skip:] if k == 0 else pre).copy()
            q[:, _host.PRE_OPS_AT] += np.uint32(base)
            parts.append(q)
            ops.append(op)
            base += op.shape[0]
        merged = (np.concatenate(parts), np.concatenate(ops))
        self.pre, self.pre_skip = [merged], 0
        return merged

    def window(self, want):
        buf = self.bufs[self.cur]
        if self.end - self.start < want and not (self.reader.eof and self.ahead is None):
            live = self.end - self.start
            if self.advanced:                                      # move on: tail of this window + the text decoded ahead
                nxt = (self.cur + 1) % 3
                if self.ahead is not None:
                    nb, n = self._take_ahead()
                else:
                    nb, n = self._buffer(nxt, max(FILE_WINDOW_BYTES, want)), 0
                if live <= self.HEAD and nb.shape[0] - self.HEAD >= want:
                    nb[self.HEAD - live:self.HEAD] = buf[self.start:self.end]
                    self.start, self.end = self.HEAD - live, self.HEAD + n
                else:                                              # a tail longer than the head room, or a larger window
                    grown = np.empty(self.HEAD + live + max(n, want) + (1 << 16), dtype=np.uint8)
                    grown[self.HEAD:self.HEAD + live] = buf[self.start:self.end]
                    grown[self.HEAD + live:self.HEAD + live + n] = nb[self.HEAD:self.HEAD + n]
                    self.start, self.end = self.HEAD, self.HEAD + live + n
                    nb = grown
                self.cur = nxt
                self.bufs[nxt] = buf = nb
                self.advanced = False
            else:                                                  # the same window again, larger (or the very first one)
                extra = None
                if self.ahead is not None:                         # text already decoded ahead belongs behind this window
                    extra = self._take_ahead()
                need = live + (extra[1] if extra else 0) + want + (1 << 16)
                if buf is None or self.start + need > buf.shape[0]:
                    grown = np.empty(self.HEAD + 2 * need, dtype=np.uint8)
                    if live:
                        grown[self.HEAD:self.HEAD + live] = buf[self.start:self.end]
                    self.bufs[self.cur] = buf = grown
                    self.start, self.end = self.HEAD, self.HEAD + live
                if extra:
                    buf[self.end:self.end + extra[1]] = extra[0][self.HEAD:self.HEAD + extra[1]]
                    self.end += extra[1]
            if self.end - self.start < want:
                self.end, chunks = self._decode(buf, self.end)
                self.pre.extend(chunks)
            if not self.reader.eof:                                # decode what follows while this window is parsed
                k = (self.cur + 1) % 3
                self._buffer(k, max(FILE_WINDOW_BYTES, want))
                self.ahead = self.decoder.submit(self._decode_ahead, k)
        n = min(want, self.end - self.start)
        return buf, self.start, n, self.reader.eof and self.ahead is None and n == self.end - self.start

    def advance(self, consumed, lines=0):
        self.start += consumed
        self.advanced = self.advanced or consumed > 0
        if self.pre_ok:
            # the descriptions of the consumed lines must add up to the consumed bytes: they do unless a string value
            # held a line break (the text rules then saw more lines than the decoder printed records) -- from there on
            # the text is parsed as text
            left, total = lines, 0
            while left and self.pre:
                pre = self.pre[0][0]
                take = min(left, pre.shape[0] - self.pre_skip)
                total += int(pre[self.pre_skip:self.pre_skip + take, 0].sum(dtype=np.int64)) + take
                left -= take
                self.pre_skip += take
                if self.pre_skip >= pre.shape[0]:
                    self.pre.pop(0)
                    self.pre_skip = 0
            if left or total != consumed:
                self.pre_ok, self.pre, self.pre_skip = False, [], 0

    def close(self):
        if self.ahead is not None:
            try:
                self.ahead.result()
            except Exception:
                pass
        self.decoder.shutdown(wait=True)
        self.reader.close()


# inflated bytes of each file per window of the GPU BAM path: both files' blocks go into ONE inflate launch, and ~16 000 blocks
# (two 512 MB windows) are what keeps the chip's ~8 000 decoder chains busy for two rounds (256 MB: 5.0, 512 MB: 5.6 M pairs/s on
# a 1.9 M-pair input, profiles/r05_bam_gpu_path.txt)
BAM_GPU_WINDOW_BYTES = int(os.environ.get("XENOMAPPER_BAM_WINDOW_MB", "512")) << 20
# the text the writer prints per (slot, file): kept from run to run -- a fresh 1.4 GB buffer per window and file would be
# page-faulted in again by every run (one run at a time uses them: the default context's BAM front end is one object)
_BAM_TEXT_BUFFERS = {}


class _CompressedBytesOutgrowStaging(Exception):
    """_GpuBamFile.stage(): the window's compressed bytes do not fit the slot's staging buffer (reserved for half the inflated
    bytes: BAM that hardly compresses) -- the caller reserves for the worst case and stages again."""


class _GpuBamFile(object):
    """One BAM file of the GPU BAM path (include/xenomapper_bgzf.h): a cursor over its BGZF blocks.  Per window the host only
    walks the member headers of the next blocks (xm_bgzf_index) and reads their compressed bytes into the slot's page-locked
    staging buffer; inflating, finding the records and stripping them is the device's (xm_bamdev_run).  The inflated tail a
    window did not consume is carried into the next one on the device."""

    def __init__(self, path, n_threads=0):
        from . import _host
        self.path = path
        self.data = np.memmap(path, dtype=np.uint8, mode="r")
        self.reader = _host.BamReader(self.data, n_threads, header_only=True)   # header, reference names, the printer's threads
        at = self.reader.records_start()
        # the block the first record lies in: members are walked from the start until their inflated sizes pass it
        blocks, _crc, nxt, _total = _ffi.bgzf_index(self.data, 0, at + 1)
        ends = blocks["out_off"] + blocks["isize"]
        j = int(np.searchsorted(ends, at, side="right"))              # first block whose end lies behind the offset
        if j < len(blocks):
            self.cursor = int(blocks["cdata_off"][j]) - self._member_header(blocks, j)
            self.skip = at - int(blocks["out_off"][j])
        else:                                                        # no record at all
            self.cursor, self.skip = nxt, 0
        self.ref_names = self._reference_names(blocks, at)           # for the device printer (None: could not be read)
        self.fd = os.open(path, os.O_RDONLY)
        self.carry = (0, 0, 0)                                       # (slot, offset, bytes) of the previous window's tail
        self.bytes_per_record = 0.0                                  # of the windows so far (0: not known yet)
        self.seen = [0, 0]                                           # bytes and records consumed so far
        self.by_lines = None                                         # record table of a window parsed by the text rules
        self.pending = None
        self.ahead = None                                            # (staging address, file offset, bytes, the slot's capacities, blocks, CRCs, indexed bytes) read ahead for the next window
        self.ahead_hits = self.ahead_misses = 0                      # windows whose blocks read_ahead had in place / that stage() had to index and read itself
        self.last_comp = 0                                           # compressed bytes of the last window staged

    def _reference_names(self, blocks, at):
        """The reference names of the BAM header (SAM specification 4.2: magic, l_text, text, n_ref, then l_name, name, l_ref per
        reference), from the inflated bytes in front of the first record."""
        import zlib
        try:
            head = bytearray()
            for blk in blocks:
                if len(head) >= at:
                    break
                c0, cl = int(blk["cdata_off"]), int(blk["cdata_len"])
                head += zlib.decompress(bytes(self.data[c0:c0 + cl]), -15)
            if len(head) < at or head[:4] != b"BAM\x01":
                return None
            l_text = int.from_bytes(head[4:8], "little")
            p = 8 + l_text
            n_ref = int.from_bytes(head[p:p + 4], "little")
            p += 4
            names = []
            for _ in range(n_ref):
                l_name = int.from_bytes(head[p:p + 4], "little")
                name = bytes(head[p + 4:p + 4 + l_name])
                names.append(name.split(b"\0", 1)[0])
                p += 4 + l_name + 4
            return names if p == at else None
        except Exception:                                            # noqa: BLE001 -- the host printer does not need them
            return None

    def _member_header(self, blocks, j):
        """Bytes between a member's first byte and its DEFLATE data: cdata_off of block j minus the end of block j - 1."""
        if j == 0:
            return int(blocks["cdata_off"][0])
        prev_end = int(blocks["cdata_off"][j - 1]) + int(blocks["cdata_len"][j - 1]) + 8
        return int(blocks["cdata_off"][j]) - prev_end

    @property
    def at_end(self):
        return self.cursor >= self.data.shape[0]

    def stage(self, dev, slot, file, parser, want_raw, max_blocks):
        """Index the next blocks (about want_raw inflated bytes, less what is carried) and read their compressed bytes into the
        slot's staging buffer -> the dict xm_bamdev_run takes.  The cursor moves only in commit()."""
        carry_slot, carry_off, carry_len = self.carry
        budget = max(want_raw - carry_len, 0)
        blocks = np.zeros(0, dtype=_ffi.BGZF_BLOCK)
        crc = np.zeros(0, dtype=np.uint32)
        nxt, comp_len, uploaded = self.cursor, 0, 0
        dst = dev._L.xm_bamdev_staging(dev._h, slot, file)
        ahead = self.ahead
        if (budget and not self.at_end and ahead is not None and ahead[0] == dst and ahead[1] == self.cursor
                and ahead[3] == dev.capacity(slot) and not self.skip):
            # The window's bytes are in the staging buffer (read_ahead read them from the cursor while the GPU had the window in
            # front -- into THIS allocation of the buffer: one that has grown since may sit at the same address and holds
            # nothing) and their members are indexed already, in that thread, from the buffer: walking the headers through the
            # file's mapping costs a page fault per block, 5 - 10 ms per file and window in the thread the GPU waits for.
            a_blocks, a_crc, a_end = ahead[4], ahead[5], ahead[6]
            ends = np.cumsum(a_blocks["isize"], dtype=np.uint64)
            k = min(int(np.searchsorted(ends, budget, side="left")) + 1, len(a_blocks), max_blocks)
            whole = k == len(a_blocks) and self.cursor + a_end >= self.data.shape[0]       # the file ends with the last block indexed
            if k and (int(ends[k - 1]) >= budget or whole):
                blocks, crc = a_blocks[:k].copy(), a_crc[:k].copy()
                comp_len = int(blocks["cdata_off"][-1]) + int(blocks["cdata_len"][-1])     # (<= what read_ahead read: it fits)
                nxt = self.cursor + comp_len + 8                       # behind the last member's CRC-32 and ISIZE
                self.last_comp = comp_len
                uploaded = min(ahead[2], comp_len)                     # read_ahead sent what it read to the device as well
                self.ahead = None
                self.pending = nxt
                self.ahead_hits += 1
                return {"comp_len": comp_len, "blocks": blocks, "crc": crc, "carry_slot": carry_slot, "carry_off": carry_off,
                        "carry_len": carry_len, "eof": nxt >= self.data.shape[0], "skip": 0, "uploaded": uploaded}
        if budget and not self.at_end:
            self.ahead_misses += 1
            blocks, crc, nxt, _total = _ffi.bgzf_index(self.data, self.cursor, budget + self.skip, max_blocks)
            if len(blocks):
                c0 = int(blocks["cdata_off"][0])
                comp_len = int(blocks["cdata_off"][-1]) + int(blocks["cdata_len"][-1]) - c0
                if comp_len > dev.capacity(slot)[0]:
                    self.ahead_misses -= 1
                    raise _CompressedBytesOutgrowStaging(comp_len)
                blocks = blocks.copy()
                blocks["cdata_off"] -= np.uint64(c0)
                parser.pread(self.fd, c0, dst, comp_len)
                self.last_comp = comp_len
        self.ahead = None
        self.pending = nxt
        return {"comp_len": comp_len, "blocks": blocks, "crc": crc, "carry_slot": carry_slot, "carry_off": carry_off,
                "carry_len": carry_len, "eof": nxt >= self.data.shape[0], "skip": self.skip if len(blocks) else 0,
                "uploaded": uploaded}

    def read_ahead(self, dev, slot, file, reader, grow=1.0):
        """The compressed bytes of the NEXT window (from `pending`, where the window just staged ends) read into the other
        slot's staging buffer -- by a thread of its own, while the GPU has the window just staged: stage() of the next window
        then finds them in place.  The amount is a guess (what the last window took); stage() reads whatever is missing."""
        self.ahead = None
        at = self.pending
        if at is None or at + 18 > self.data.shape[0] or not self.last_comp:
            return
        room = dev.capacity(slot)[0]
        # (grow: the next window is asked to be that much larger than the one just staged -- the run's first windows double)
        n = min(self.data.shape[0] - at, int(self.last_comp * grow) + self.last_comp // 16 + (1 << 20), room)
        if n <= 0:
            return
        dst = dev._L.xm_bamdev_staging(dev._h, slot, file)
        if not dst:
            return
        reader.pread(self.fd, at, dst, n)                            # from the member's first byte: the headers come along
        dev.upload(slot, file, n)                                    # and on to the device, beside the GPU's work on the current window
        # the members of what was read, indexed here and from the buffer (stage() takes as many as its window holds)
        view = _ffi._host_view(dst, n, np.uint8)
        blocks, crc, end, _total = _ffi.bgzf_index(view, 0, 1 << 62, n // 2048 + 1024, prefix=True)   # (a file of tiny members: stage() indexes itself)
        self.ahead = (dst, at, n, dev.capacity(slot), blocks, crc, end)

    def ran(self, slot, raw_len, rec_off=None, stop=None, consumed=0, records=0):
        """What advance() needs to know about the window that was just run (and how many records its consumed bytes held)."""
        self.last = (slot, raw_len, rec_off, stop)
        if records:
            self.seen[0] += consumed - self.carry[1] * 0
            self.seen[1] += records
            self.bytes_per_record = self.seen[0] / self.seen[1]

    def advance(self, consumed, lines=0):
        """The window was processed: `consumed` of its inflated bytes are done (or, for a window that went through the text
        rules, `lines` of its records), the rest is carried into the next window."""
        slot, raw_len, rec_off, stop = self.last
        if rec_off is not None:
            consumed = int(rec_off[lines]) if lines < rec_off.shape[0] else stop
        if self.pending is not None and self.pending != self.cursor:
            self.cursor, self.skip = self.pending, 0
        self.carry = (slot, int(consumed), raw_len - int(consumed))
        self.pending = None

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None
        self.reader.close()


def _run_files(mode, path1, path2, sinks, min_score, tag_func, skip_repeated, n_threads=0, bam=False, starts=None):
    """The three main loops on two SAM (or BAM) *files*: same results as _run(mode, getReadPairs(...)), with the
    text work done by the C++ stripper / writer.  Falls back to the Python reader when the input is not ASCII."""
    from . import _host
    ctx = default_context()
    paired = mode != _ffi.MODE_SE
    cigar_mode = tag_func is get_cigarbased_AS_tag
    score_mode = _host.SCORE_CIGAR if cigar_mode else (_host.SCORE_AS_ZS if tag_func is get_tag_with_ZS_as_XS
                                                       else _host.SCORE_AS_XS)
    if not n_threads:                                                # 0: the CPUs this process may use (cgroup-aware)
        n_threads = int(os.environ.get("XENOMAPPER_THREADS", "0"))
    prof = _PhaseClock()
    _EMIT_CLOCK.clear()
    t_all = time.perf_counter()
    # BAM on the GPU (include/xenomapper_bgzf.h): inflate + record chain + stripper on the device, either walk, the three plugins
    # (--cigar_scores: NM and the records' CIGAR words become the packed CIGAR columns on the device); XENOMAPPER_GPU_BAM=0 keeps
    # the host decoder
    bamdev = None
    if (bam and min_score == min_score and os.environ.get("XENOMAPPER_GPU_BAM", "1") != "0"):
        try:
            bamdev = default_bamdev()
        except MemoryError:
            bamdev = None
    with prof("open"):
        if bamdev is not None:
            sources = [_GpuBamFile(path, n_threads) for path in (path1, path2)]
        else:
            sources = [(_BamSource(path, n_threads) if bam else _SamSource(path, None if starts is None else starts[k]))
                       for k, path in enumerate((path1, path2))]
        # Two parsers alternate so that the next window is decoded (BAM) and parsed in a helper thread -- the C++
        # code runs without the GIL -- while the GPU classifies and the writer emits the current one.
        parsers = [_host.Parser(n_threads), _host.Parser(n_threads)]
        # pread(2) into the page-locked staging buffers by a SMALL pool of its own (XENOMAPPER_PREAD_THREADS): with the stripper
        # on the GPU the helper thread only copies, and sixteen readers beside the writer's sixteen gatherers oversubscribe the
        # cores that feed the PCIe link
        n_readers = int(os.environ.get("XENOMAPPER_PREAD_THREADS", "0"))
        reader_pool = _host.Parser(n_readers) if n_readers > 0 else None
        pool = ThreadPoolExecutor(max_workers=1)
        # output files extended ahead of the writer, their pages unmapped behind it (more helper threads than two measured no
        # better: profiles/r06_ab_sam_bins.txt)
        ahead, ahead_pool = {}, ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("XENOMAPPER_AHEAD_THREADS", "2"))))
    totals, key_order = Counter(), []
    window = FILE_WINDOW_BYTES
    active = [s for s in sinks if s]
    distinct = len(set(id(s) for s in active)) == len(active)
    # SAM text: the text itself goes to the GPU and is split there (include/xenomapper_strip.h) -- the host threads only copy
    # windows into page-locked memory and write the outputs.  XENOMAPPER_GPU_STRIP=0 keeps the host stripper.
    stripper = None
    if not bam and min_score == min_score and os.environ.get("XENOMAPPER_GPU_STRIP", "1") != "0":
        try:
            stripper = default_stripper()
        except MemoryError:                                          # no room for its buffers: the host threads strip
            stripper = None

    # SAM text: the outputs gathered on the device as well (every bin needs a sink of its own; XENOMAPPER_GPU_SAM_BINS=0: the host gathers)
    sam_bins_on_device = stripper is not None and distinct and os.environ.get("XENOMAPPER_GPU_SAM_BINS", "1") != "0"
    # ... and, with XENOMAPPER_SAM_READ_AHEAD=1, the bytes behind a window read (by a reader pool and a thread of their own) and
    # sent over the link while it is stripped, instead of when it has said where it stopped.  Built to take the windows' serial
    # dependency out of the run (a fifth of it, during which nothing is read), byte-exact, and measured: 13.6 - 18.7 against
    # 13.1 - 14.3 M pairs/s to /dev/null on a box whose reads are slow, 14.4 - 17.7 against 17.2 - 20.0 on two where they are not
    # (the reads, the link and the copy home then run all at once and each gets slower: profiles/r06_ab_sam_read_ahead.txt), no
    # difference to files.  Not the default.
    sam_spec, sam_slot_free, spec_reader, spec_pool = {}, {}, None, None
    sam_prev_end, sam_last_tail = [None, None], [0, 0]
    if (sam_bins_on_device and bamdev is None and os.environ.get("XENOMAPPER_SAM_READ_AHEAD", "0") == "1"
            and os.environ.get("XM_STRIP_ZEROCOPY", "1") != "0"):
        spec_reader = _host.Parser(int(os.environ.get("XENOMAPPER_SAM_READ_AHEAD_THREADS", "0")) or n_threads)
        spec_pool = ThreadPoolExecutor(max_workers=1)
    bam_text = _BAM_TEXT_BUFFERS                                     # (slot, file) -> [text bytes, line_off, line_len] of the GPU BAM path
    bam_windows = [0]                                                # windows run so far
    # the records' SAM text is printed on the device when the reference names could be read (XENOMAPPER_GPU_BAM_TEXT=0: by the
    # host threads, from the packed records; a window with floating-point fields is printed that way in any case)
    bam_text_on_device = [False]
    if bamdev is not None and os.environ.get("XENOMAPPER_GPU_BAM_TEXT", "1") != "0" and all(src.ref_names is not None for src in sources):
        for f, src in enumerate(sources):
            bamdev.set_refs(f, src.ref_names)
        bam_text_on_device[0] = True
    # ... and gathered there into the six outputs (xm_bamdev_fetch_bins: the host writes six byte ranges per window instead of
    # gathering the lines; needs every bin to have a sink of its own -- two bins sharing one are written unit by unit, _emit_shared).
    # XENOMAPPER_GPU_BAM_BINS=0: the device prints, the host gathers (round 5's form)
    bam_bins_on_device = [bam_text_on_device[0] and distinct and os.environ.get("XENOMAPPER_GPU_BAM_BINS", "1") != "0"]
    # the GPU BAM path reads the next window's compressed blocks while the GPU works on the current one (_GpuBamFile.read_ahead):
    # a reader of its own (8 threads: pread into page-locked memory peaks there, e2e.host_ceilings) and one thread that drives it
    bam_reader = _host.Parser(8) if bamdev is not None and os.environ.get("XENOMAPPER_BAM_READ_AHEAD", "1") != "0" else None
    bam_ahead_pool = ThreadPoolExecutor(max_workers=1) if bam_reader is not None else None
    bam_ahead = [None]                                               # the read-ahead in flight
    bam_comp_worst_case = [False]                                    # a window's compressed bytes outgrew the staging reserved for half the inflated size

    def print_records(which, f, raw_addr, rec_off_addr, n, sparse=False, wanted=None):
        """SAM text of records [0, n) of a window decoded on the GPU -> (text array, line_off, line_len)."""
        buf = bam_text.get((which, f))
        if buf is None or buf[1].shape[0] < n:
            text = buf[0] if buf is not None else np.empty(1 << 20, dtype=np.uint8)
            buf = bam_text[(which, f)] = [text, np.empty(n + n // 4 + 64, dtype=np.uint32), np.empty(n + n // 4 + 64, dtype=np.uint32)]
        while True:
            got = sources[f].reader.print_records(raw_addr, rec_off_addr, n, buf[0], buf[1], buf[2], sparse, wanted)
            if got >= 0:
                return buf[0], buf[1], buf[2], got
            buf[0] = np.empty(-got + (-got >> 3) + (1 << 16), dtype=np.uint8)

    def parse_next_bamdev(which, want):
        """One window of both BAM files on the GPU: stage compressed blocks, xm_bamdev_run (inflate, CRC, record chain, strip,
        pair), print the records' text for the writer.  A window the device does not vouch for -- blocks that do not begin
        with a record, a record the text rules might read differently -- is printed whole and stripped by the text rules."""
        from . import _host
        want = max(want, BAM_GPU_WINDOW_BYTES)
        # the first windows are small: nothing can be printed or written before the first window is through the GPU, so the
        # pipeline is filled with a quarter and a half window before the full ones (which use the chip best) follow
        grow = 1.0                                                   # the window behind this one over this one, for the read-ahead
        full = want
        if bam_windows[0] < 2 and want >= (64 << 20):
            want = want >> (2 - bam_windows[0])
            grow = 2.0 if os.environ.get("XENOMAPPER_BAM_AHEAD_GROW", "1") != "0" else 1.0
        bam_windows[0] += 1
        with prof("window"):
            if bam_ahead[0] is not None:                             # the read-ahead into this slot's staging buffers has ended
                try:
                    bam_ahead[0].result()
                except Exception:                                    # noqa: BLE001 -- a failed read or copy ahead: stage() does it all
                    for src in sources:
                        src.ahead = None
                bam_ahead[0] = None
            carried = max(src.carry[2] for src in sources)
            # (room for a tail of a sixteenth of the window: a tail a little longer than the last one is no reason to make every buffer again)
            carried = max(carried, full // 16)
            raw_cap = want + carried + (1 << 20)
            if want < full and max(src.data.shape[0] - src.cursor for src in sources) * 2 >= full:
                # a full window will follow in this slot: its buffers are made once, for that, not for the quarter and again
                # for the whole (page-locking is the first run's largest cost: 0.50 s of 0.80 on 4.5 GB of BAM)
                raw_cap = full + carried + (1 << 20)
            # DEFLATE never expands a block by more than a few bytes; BAM as the aligners and samtools write it is a third of its
            # inflated size, so the page-locked staging is reserved for half and grows (once per process) for input that is not
            comp_cap = raw_cap if bam_comp_worst_case[0] else raw_cap // 2 + (64 << 10)
            max_blocks = raw_cap // 65536 + raw_cap // 4096 + 64
            bamdev.reserve(which, comp_cap, raw_cap, max_blocks, min(FILE_MAX_RECORDS, raw_cap // 36 + 2))
            if bam_windows[0] == 1 and bam_ahead_pool is not None and not all(src.at_end for src in sources):
                # the run's first window: the OTHER slot's buffers are made now, by the helper thread, beside this window's work
                # (page-locking them is a quarter of a process's first run; the read-ahead behind this window then has where to read to)
                bam_ahead_pool.submit(_quietly, bamdev.reserve, which ^ 1, comp_cap, raw_cap, max_blocks,
                                      min(FILE_MAX_RECORDS, raw_cap // 36 + 2))
            # the two files hold the same reads at different bytes per record: each gets a window in proportion, so that the
            # windows hold about as many records and neither file drags a growing tail from window to window
            per_rec = [src.bytes_per_record for src in sources]
            scale = [p / max(per_rec) for p in per_rec] if min(per_rec) > 0 else [1.0, 1.0]
            try:
                inputs = [src.stage(bamdev, which, f, parsers[which], int(want * scale[f]) + src.carry[2], max_blocks)
                          for f, src in enumerate(sources)]
            except _CompressedBytesOutgrowStaging:
                bam_comp_worst_case[0] = True
                prof["bam_staging_regrown"] = prof.get("bam_staging_regrown", 0) + 1
                bamdev.reserve(which, raw_cap, raw_cap, max_blocks, min(FILE_MAX_RECORDS, raw_cap // 36 + 2))
                for src in sources:
                    src.ahead = None                                 # (read into the buffers that were just replaced)
                inputs = [src.stage(bamdev, which, f, parsers[which], int(want * scale[f]) + src.carry[2], max_blocks)
                          for f, src in enumerate(sources)]
            prof["bam_carry_bytes"] = prof.get("bam_carry_bytes", 0) + sum(int(x["carry_len"]) for x in inputs)
        if bam_reader is not None and not all(x["eof"] for x in inputs):
            def ahead_job(slot=which ^ 1, grow=grow):
                for f, src in enumerate(sources):
                    src.read_ahead(bamdev, slot, f, bam_reader, grow)
            bam_ahead[0] = bam_ahead_pool.submit(ahead_job)
        with prof("strip"):
            blk = bamdev.run(which, inputs, score_mode, paired, paired, min(FILE_MAX_RECORDS, raw_cap // 36 + 2), wait_raw=False,
                             skip_repeated=skip_repeated)
            prof["strip_upload_ms"] = prof.get("strip_upload_ms", 0.0) + blk.ms_inflate
            prof["strip_kernels_ms"] = prof.get("strip_kernels_ms", 0.0) + blk.ms_kernels
        if blk.bad_block:
            what = {1: "a BGZF block fails its decoder or its CRC-32", 2: "a file ends inside an alignment record",
                    3: "a malformed alignment record"}.get(blk.bad_block, "damaged")
            raise ValueError("corrupt BAM input: %s (%s, %s)" % (what, path1, path2))
        eofs = [bool(x["eof"]) for x in inputs]
        # which way each window went, for the profile (and the tests: tests/test_file_path.py, test_bam_gpu.py): "raw" = the whole
        # inflated windows came back for the host to walk or print, "device_text" / "host_text" = who printed the wanted records
        prof["bam_windows"] = prof.get("bam_windows", 0) + 1
        if blk.unaligned or blk.weird or (cigar_mode and blk.n_exceptions):
            prof["bam_windows_raw"] = prof.get("bam_windows_raw", 0) + 1
            # (--cigar_scores: an NM or XS value the kernels do not vouch for sends the window the same way, as on the SAM path)
            with prof("parse"):
                # the whole windows as text, then the text rules (exactly the host path's semantics for this window)
                bamdev.fetch_raw(which, blk)
                bamdev.raw_wait(which)                               # the inflated bytes must have arrived
                texts, tables = [], []
                for f in (0, 1):
                    # the record chain followed on the host (the inflated bytes are here already): every complete record
                    rec = np.empty(blk.raw_len[f] // 36 + 2, dtype=np.uint32)
                    first = inputs[f]["skip"] if not inputs[f]["carry_len"] else 0
                    n_rec, stop = _host.bam_walk(blk.raw_addr[f], blk.raw_len[f], first, rec)
                    rec = rec[:n_rec]
                    text, _loff, _llen, got = print_records(which, f, blk.raw_addr[f], rec.ctypes.data, n_rec)
                    texts.append((text, got))
                    tables.append((rec, stop))
                    sources[f].ran(which, blk.raw_len[f], rec, stop)
                whole = [eofs[f] and tables[f][1] == blk.raw_len[f] for f in (0, 1)]
                hb = parsers[which].parse(texts[0][0], 0, texts[0][1], whole[0], texts[1][0], 0, texts[1][1], whole[1],
                                          score_mode, paired, skip_repeated, paired, FILE_MAX_RECORDS)
            return hb, [texts[0][0], texts[1][0]], [0, 0], eofs
        for f in (0, 1):
            sources[f].ran(which, blk.raw_len[f], consumed=blk.consumed[f] - inputs[f]["skip"], records=max(blk.n - (1 if paired else 0), 0))
        texts = [None, None]
        if blk.n and not blk.n_exceptions:
            # the fused pass right behind the strip kernels, on the slot's stream, BEFORE the next window's inflate launch is
            # queued (issued later, from the main thread, it waits behind that launch: 25-30 ms a window)
            with prof("classify"):
                blk.classified = bamdev.classify(which, mode, blk.n, _floor_min_score(min_score))
                # the bins are on the device: only the records a sink takes come back, packed (half of a window)
                mask = sum(1 << b for b in range(6) if sinks[b])
                blk.lines = None
                blk.bins = None
                if bam_bins_on_device[0]:
                    # the six outputs themselves, gathered on the device: what comes back is what the sinks get
                    bins = bamdev.fetch_bins(which, blk.n, paired, mask)
                    if bins[0] == 0:
                        blk.bins = bins
                if blk.bins is not None:
                    prof["bam_windows_device_bins"] = prof.get("bam_windows_device_bins", 0) + 1
                elif bam_text_on_device[0]:
                    lines = bamdev.fetch_text(which, blk.n, paired, mask)
                    if lines[0] == 0:
                        blk.lines = lines
                    # (status 1: a binary64 field -- f and B:f are printed on the device since round 6 --, 2: more text than the
                    # slot's buffers hold: the host prints THIS window; the next one is offered to the device again)
                if blk.lines is None and blk.bins is None:
                    blk.packed = bamdev.fetch_wanted(which, blk.n, paired, mask)
                key = "bam_windows_device_text" if (blk.lines is not None or blk.bins is not None) else "bam_windows_host_text"
                prof[key] = prof.get(key, 0) + 1
        else:
            prof["bam_windows_raw"] = prof.get("bam_windows_raw", 0) + 1
            bamdev.fetch_raw(which, blk)                                  # values the text rules must decide, or nothing: the whole windows

        def finish():
            # the records' SAM text, printed by the host threads when the block is settled -- in the main thread, while the
            # helper thread has the GPU inflate and strip the NEXT window (printing here instead would put the two in a row)
            with prof("parse"):
                t_w = time.perf_counter()
                bamdev.raw_wait(which)                               # the copy of the window ran beside the kernels and the next window's inflate
                prof["bam_wait_raw"] = prof.get("bam_wait_raw", 0.0) + time.perf_counter() - t_w
                if getattr(blk, "bins", None) is not None:          # gathered on the device: the outputs themselves are here
                    return
                if getattr(blk, "lines", None) is not None:         # printed on the device: the text and its line table are here
                    _st, text, loff, llen = blk.lines
                    texts[0], texts[1] = text[0], text[1]
                    blk.set_text(list(loff), list(llen))
                    return
                loffs, llens = [], []
                # A unit's lines come from ONE file (primary bins: file 1, secondary bins: file 2, unresolved: both; :423-448),
                # and a bin without a sink prints nothing: with the bins known on the device, only those records came back
                # (packed; their table says where each went), and only those are printed
                wanted = [None, None]
                raw_addr, off_addr = blk.raw_addr, blk.rec_off_addr
                if blk.packed is not None:
                    raw_addr, places, _bytes = blk.packed
                    off_addr = [places[0].ctypes.data if blk.n else 0, places[1].ctypes.data if blk.n else 0]
                    wanted = [(places[f] != 0xFFFFFFFF).view(np.uint8) for f in (0, 1)]
                t_p = time.perf_counter()
                for f in (0, 1):
                    text, loff, llen, _got = print_records(which, f, raw_addr[f], off_addr[f], blk.n, sparse=True,
                                                           wanted=wanted[f])
                    texts[f] = text
                    loffs.append(loff); llens.append(llen)
                prof["bam_print"] = prof.get("bam_print", 0.0) + time.perf_counter() - t_p
                blk.set_text(loffs, llens)
        blk.finish = finish
        return blk, texts, [0, 0], eofs

    def read_behind(slot, ends, counts, room):
        """The bytes BEHIND the window that is being stripped -- [ends[f], ends[f] + counts[f]) of either file -- read into the other
        slot's staging buffers, `room` bytes into them, and sent on to the device piece by piece as always: the next window begins
        somewhere in the last bytes of this one (its walk will say where), so its tail is put in front of these bytes when it is
        known (parse_next; include/xenomapper_strip.h, "reading ahead").  Runs beside the strip kernels, the fused pass and the
        hand-over of the window in front, which used to be a fifth of a run during which nothing was read."""
        t_r = time.perf_counter()
        for f in (0, 1):
            base = stripper.staging_address(slot, f)
            stripper.begin_behind(slot, f, room)
            for at in range(0, counts[f], STAGE_PIECE):
                piece = min(STAGE_PIECE, counts[f] - at)
                spec_reader.pread(sources[f].fileno(), ends[f] + at, base + room + at, piece)
                stripper.upload(slot, f, room + at, piece)
        prof["sam_read_behind"] = prof.get("sam_read_behind", 0.0) + time.perf_counter() - t_r

    def drop_stripper():
        """The stripper is given up for the rest of the run (and of the process): not while a read into its buffers is under way."""
        nonlocal stripper
        for sp in sam_spec.values():
            try:
                sp["job"].result()
            except Exception:                                        # noqa: BLE001
                pass
        sam_spec.clear()
        _forget_stripper(stripper)
        stripper = None

    def parse_next(which, want):
        nonlocal stripper
        if bamdev is not None:
            return parse_next_bamdev(which, want)
        with prof("window"):
            wins = [src.window(want) for src in sources]
        lead = [0, 0]                                                # bytes in front of each window's text in its staging buffer
        ahead_of_us = sam_spec.pop(which, None)                      # bytes read behind the window in front, into this slot
        if ahead_of_us is not None:
            t_w = time.perf_counter()
            try:
                ahead_of_us["job"].result()
            except Exception:                                        # noqa: BLE001 -- a read that failed: this window is read again
                ahead_of_us = None
            prof["sam_wait_read"] = prof.get("sam_wait_read", 0.0) + time.perf_counter() - t_w
        for f in (0, 1):                                             # what the window in front left over of either file
            if sam_prev_end[f] is not None:
                sam_last_tail[f] = max(sam_prev_end[f] - wins[f][1], 0)
        if ahead_of_us is not None and stripper is not None:
            tails = [ahead_of_us["ends"][f] - wins[f][1] for f in (0, 1)]
            if all(0 <= t <= ahead_of_us["room"] for t in tails):
                # the window = its tail (read now: a few KB) + what was read ahead, right-aligned in the room in front of that
                sizes = [int(src.raw.shape[0]) for src in sources]
                for f in (0, 1):
                    lead[f] = ahead_of_us["room"] - tails[f]
                    n_f = tails[f] + ahead_of_us["counts"][f]
                    wins[f] = (wins[f][0], wins[f][1], n_f, wins[f][1] + n_f >= sizes[f])
                prof["sam_windows_read_ahead"] = prof.get("sam_windows_read_ahead", 0) + 1
            else:
                ahead_of_us = None                                   # the walk stopped further back than the room in front allows
        if stripper is not None and max(w[2] for w in wins) <= _ffi.STRIP_MAX_WINDOW:
            # no line of a record is shorter than two bytes with its terminator
            records = min(FILE_MAX_RECORDS, max(w[2] for w in wins) // 2 + 2)
            try:
                need = max(lead[f] + wins[f][2] for f in (0, 1))
                if spec_reader is not None and want <= _ffi.STRIP_MAX_WINDOW // 2:
                    # (room for a window read ahead behind a gap, from the start: the buffers of a slot are never made again
                    # while the main thread may still hold the window before -- see below)
                    need = max(need, min(SAM_TAIL_ROOM, max(want, 1 << 16)) + want)
                stripper.reserve(which, need, max(records, min(FILE_MAX_RECORDS, need // 2 + 2)))
            except MemoryError:                                      # page-locked or device memory ran out: the host threads strip
                drop_stripper()
        blk = None
        sam_slot_free[which] = False                                 # until this window is known to go to the writer as six ranges
        if stripper is not None and max(w[2] for w in wins) <= _ffi.STRIP_MAX_WINDOW:
            try:
                with prof("stage"):
                    # read(2) into page-locked memory, piece by piece: the upload of one piece hides the read of the next; no
                    # page of the inputs is mapped, and the writer gathers its lines from the staging buffer
                    for f, w in enumerate(wins):
                        base = stripper.staging_address(which, f)
                        if ahead_of_us is not None:                  # all but the tail is there (and on the device) already
                            if tails[f]:
                                (reader_pool or parsers[which]).pread(sources[f].fileno(), w[1], base + lead[f], tails[f])
                            stripper.set_lead(which, f, lead[f])
                            continue
                        for at in range(0, w[2], STAGE_PIECE):
                            piece = min(STAGE_PIECE, w[2] - at)
                            (reader_pool or parsers[which]).pread(sources[f].fileno(), w[1] + at, base + at, piece)
                            stripper.upload(which, f, at, piece)
                    # the bytes behind this window, into the other slot, while this one is stripped and handed over -- if the window
                    # that slot held went to the writer as six ranges (then nobody reads its text any more) and both files go on
                    other = which ^ 1
                    if (spec_reader is not None and sam_slot_free.get(other, True) and not wins[0][3] and not wins[1][3]
                            and want <= _ffi.STRIP_MAX_WINDOW // 2):
                        room = min(SAM_TAIL_ROOM, max(want, 1 << 16))
                        # (the main thread may still be counting the categories of the window that slot held, in the slot's
                        # page-locked tables: only into buffers that are large enough as they stand)
                        if stripper.fits(other, room + want, min(FILE_MAX_RECORDS, (room + want) // 2 + 2)):
                            ends = [w[1] + w[2] for w in wins]
                            # the two files hold the same reads at different bytes per record: the one whose windows leave the
                            # longer tails gets that much less, or its tail would grow from window to window
                            counts = [min(max(want - sam_last_tail[f], want // 2), int(sources[f].raw.shape[0]) - ends[f])
                                      for f in (0, 1)]
                            sam_spec[other] = {"ends": ends, "counts": counts, "room": room,
                                               "job": spec_pool.submit(read_behind, other, ends, counts, room)}
                sam_prev_end[0], sam_prev_end[1] = wins[0][1] + wins[0][2], wins[1][1] + wins[1][2]
                with prof("strip"):
                    blk = stripper.run(which, lead[0] + wins[0][2], wins[0][3], lead[1] + wins[1][2], wins[1][3], score_mode, paired,
                                       skip_repeated, paired, records)
                    if lead[0] or lead[1]:                           # (offsets count from the buffer's first byte: so much TEXT is done)
                        blk.consumed = (blk.consumed[0] - lead[0], blk.consumed[1] - lead[1])
                    prof["strip_upload_ms"] = prof.get("strip_upload_ms", 0.0) + blk.ms_upload
                    prof["strip_kernels_ms"] = prof.get("strip_kernels_ms", 0.0) + blk.ms_kernels
            except (MemoryError, OSError):
                # the --cigar_scores arrays did not fit (MemoryError from the run), or a read into the staging buffer failed
                # half way: this window and the rest of the run go through the host stripper, and the half-staged stripper
                # is not kept for later runs of the process
                drop_stripper()
                blk = None
        if blk is not None:
            if blk.non_ascii:
                raise _host.NonAsciiInput()
            if (sam_bins_on_device and blk.n and not blk.n_exceptions and not blk.overflow):
                # the fused pass right behind the strip kernels, then the six outputs gathered on the device and sent back as one
                # stream (xm_strip_fetch_bins): the main thread writes six byte ranges instead of gathering the lines.  A window
                # with a wanted line that needs re-joining, or with more text than the buffers hold, is written as before.
                with prof("classify"):
                    blk.classified = stripper.classify(which, mode, blk.n, _floor_min_score(min_score))
                    mask = sum(1 << b for b in range(6) if sinks[b])
                    bins = stripper.fetch_bins(which, blk.n, paired, mask)
                if bins[0] == 0:
                    blk.bins = bins
                    prof["sam_windows_device_bins"] = prof.get("sam_windows_device_bins", 0) + 1
                    sam_slot_free[which] = True                      # nobody reads this slot's staged text again

                    def finish(strip=stripper, slot=which):
                        t_w = time.perf_counter()
                        strip.out_wait(slot)
                        prof["sam_wait_out"] = prof.get("sam_wait_out", 0.0) + time.perf_counter() - t_w
                    blk.finish = finish
            prof["sam_windows"] = prof.get("sam_windows", 0) + 1
            staged = [stripper.staging(which, f) for f in (0, 1)]
            if blk.overflow or (cigar_mode and blk.n_exceptions):
                # more lines than the device tables hold, or a --cigar_scores block with a value the kernels do not vouch for:
                # this window goes through the host stripper (the text is in the staging buffers already)
                with prof("parse"):
                    blk = parsers[which].parse(staged[0], lead[0], wins[0][2], wins[0][3], staged[1], lead[1], wins[1][2], wins[1][3],
                                               score_mode, paired, skip_repeated, paired, FILE_MAX_RECORDS)
                return blk, staged, lead, [w[3] for w in wins]       # (the host stripper's offsets count from where it was told to begin)
            return blk, staged, [0, 0], [w[3] for w in wins]
        with prof("parse"):
            blk = None
            if bam and all(src.pre_ok for src in sources):
                # BAM holds AS / XS / ZS / NM as typed values and the CIGAR as operations: the decoder described every line
                # it printed, the stripper only walks the two files in lock-step (no tokenising of its own output)
                pw = [src.pre_window() for src in sources]
                blk = parsers[which].parse_pre(wins[0][0], wins[0][1], wins[0][2], wins[0][3], pw[0][0], pw[0][1],
                                               wins[1][0], wins[1][1], wins[1][2], wins[1][3], pw[1][0], pw[1][1],
                                               score_mode, paired, skip_repeated, paired, FILE_MAX_RECORDS)
            if blk is None:                                       # SAM input, or a line that needs the text rules
                blk = parsers[which].parse(wins[0][0], wins[0][1], wins[0][2], wins[0][3], wins[1][0], wins[1][1], wins[1][2],
                                           wins[1][3], score_mode, paired, skip_repeated, paired, FILE_MAX_RECORDS)
        return blk, [w[0] for w in wins], [w[1] for w in wins], [w[3] for w in wins]

    def settle(block, raws, pos, parser, pending):
        """Classify one parsed block and hand its units to the sinks.  Returns the input error to raise once the
        units in front of it have been written (one found while resolving the stripper's exceptions comes first)."""
        if getattr(block, "finish", None) is not None:
            block.finish()                                       # GPU BAM path: print the records' text now
        n = block.n
        on_device = isinstance(block, (_ffi.StrippedBlock, _ffi.BamDevBlock))
        exc = block.exc
        patches, bad, err = {}, None, None
        if exc:
            flags = np.unpackbits(block.unit_bits.view(np.uint8), bitorder="little")[:n].astype(bool)
            if paired:
                needed = flags.copy()
                needed[:-1] |= flags[1:]
            else:
                needed = np.ones(n, dtype=bool)
            patches, bad, err = _resolve_exceptions(block, raws, pos, needed, tag_func, cigar_mode)
        if err is not None:
            n, pending = bad, err                                # units closing at index >= bad are not reached
        ready = getattr(block, "bins", None) if (not exc and err is None) else None
        if on_device and block.n and ready is None:
            parser.adopt_lines(raws[0], pos[0], raws[1], pos[1], block.n, block.tables)
        if n:
            with prof("classify"):          # one fused pass: category bytes, counts and the six bin lists
                if on_device and not patches and n == block.n and getattr(block, "classified", None) is not None:
                    code, idx, off, counts = block.classified        # classified right behind the strip kernels (GPU BAM path)
                elif on_device and not patches:                  # the columns never left the device
                    code, idx, off, counts = block.stripper.classify(block.slot, mode, n, _floor_min_score(min_score))
                else:
                    code, idx, off, counts = _classify_parsed(ctx, mode, block, n, patches, cigar_mode, min_score)
            limit, state_error = None, None
            if int(off[7]) != int(off[6]):                       # a unit fell through every branch (ref :289)
                limit = int(idx[int(off[6]):int(off[7])].min())
                j = limit - 1 if (paired and (int(code[limit]) >> 3) > 5) else limit    # the mate that fell through, as _run()
                shown = []
                for f in (0, 1):                                 # the values the reference would print: the plugin's own
                    fields = _fields_of(raws[f], block, f, j, pos[f])
                    shown += [tag_func(fields, tag="AS"), tag_func(fields, tag="XS")]
                state_error = RuntimeError("Error in processing logic with values {0} ".format(tuple(shown)))
            if ready is not None and limit is not None:          # (state 6 needs a NaN: not on the int32 columns of this path)
                raise RuntimeError("a unit fell through every branch on the GPU BAM path")
            if ready is not None and limit is None:
                # the outputs were gathered on the device (xm_bamdev_fetch_bins / xm_strip_fetch_bins): six byte ranges, written
                # as they stand (into a regular file's pages by the writer's threads, else through the sink)
                _st, text, boff = ready
                for b in range(6):
                    if sinks[b] and boff[b + 1] > boff[b]:
                        piece = text[boff[b]:boff[b + 1]]
                        with prof("emit"):
                            done = _emit_into_file(parser, paired, b, None, sinks[b], ahead, ahead_pool, ready=piece, progress=run_progress[0])
                        if not done:
                            with prof("write"):
                                _write_bytes(sinks[b], piece)
            for b in (range(6) if (distinct and (ready is None or limit is not None)) else ()):
                if sinks[b]:
                    seg = idx[int(off[b]):int(off[b + 1])]
                    if limit is not None:
                        seg = seg[seg < limit]
                    with prof("emit"):
                        done = _emit_into_file(parser, paired, b, seg, sinks[b], ahead, ahead_pool, progress=run_progress[0])
                        text = None if done else parser.emit(paired, b, seg, reuse=True)
                    with prof("write"):
                        _write_bytes(sinks[b], text)
            if not distinct:
                _emit_shared(parser, paired, code, idx, off, sinks, limit)
            if state_error is not None:
                raise state_error
            _add_counts(totals, key_order, paired, code, counts)
        return pending

    which = 0
    future = None
    run_progress = [None]                                            # fraction of the input behind the block being settled (None: not known)
    try:
        while True:
            try:
                if future is not None:
                    parsed, future = future.result(), None
                else:
                    parsed = parse_next(which, window)
            except _host.NonAsciiInput:
                if bam:
                    raise ValueError("non-ASCII bytes in BAM text fields are not supported")
                return _finish_in_python(mode, path1, path2, [src.pos for src in sources], sinks, min_score, tag_func,
                                         skip_repeated, totals, key_order)
            block, raws, pos, eofs = parsed
            progressed = block.consumed[0] > 0 or block.consumed[1] > 0
            if block.starved and not progressed and not (eofs[0] and eofs[1]):
                window *= 2                                      # a line (or run of equal names) longer than the window
                continue
            pending = AssertionError() if block.mismatch_at >= 0 else None
            last = block.ended or pending is not None or (eofs[0] and eofs[1] and not progressed)
            if not last:
                for f in (0, 1):
                    sources[f].advance(block.consumed[f], block.consumed_lines[f])
                future = pool.submit(parse_next, which ^ 1, window)   # parse the next window while this one is classified
            if AHEAD_PREDICT:
                run_progress[0] = 1.0 if last else _input_fraction(sources)
            pending = settle(block, raws, pos, parsers[which], pending)
            if pending is not None:
                raise pending
            if last:
                break
            which ^= 1
    finally:
        if future is not None:
            try:
                future.result()
            except Exception:
                pass
        with prof("close"):
            trouble = None
            for sink in active:                                      # cut the output files back to their content
                state = ahead.pop(id(sink), None)
                if state is not None:
                    try:
                        state.finish(sink, ahead_pool)
                    except OSError as exc:
                        trouble = trouble or exc
            ahead_pool.shutdown(wait=True)
            pool.shutdown(wait=True)
            if spec_pool is not None:
                for sp in sam_spec.values():
                    try:
                        sp["job"].result()
                    except Exception:                                # noqa: BLE001
                        pass
                spec_pool.shutdown(wait=True)
                spec_reader.close()
            if bam_ahead_pool is not None:
                bam_ahead_pool.shutdown(wait=True)
                bam_reader.close()
            for prs in parsers + ([reader_pool] if reader_pool is not None else []):
                prs.close()
            for src in sources:
                if hasattr(src, "ahead_hits"):
                    prof["bam_ahead_hits"] = prof.get("bam_ahead_hits", 0) + src.ahead_hits
                    prof["bam_ahead_misses"] = prof.get("bam_ahead_misses", 0) + src.ahead_misses
                src.close()
        total = time.perf_counter() - t_all
        prof["other"] = total - sum(v for k, v in prof.items() if not k.endswith("_ms") and not k.startswith("bam_") and not k.startswith("sam_"))   # negative: helper-thread phases overlap the rest
        LAST_FILE_PROFILE.clear()
        LAST_FILE_PROFILE.update(prof, total=total)
        LAST_FILE_PROFILE.update(_EMIT_CLOCK)
        if os.environ.get("XENOMAPPER_PROFILE"):
            print("xenomapper file path: %.3f s  " % total + "  ".join("%s %.3f" % kv for kv in prof.items()), file=sys.stderr)
        if trouble is not None and sys.exc_info()[0] is None:
            raise trouble
    ordered = Counter()
    for key in key_order:
        ordered[key] = totals[key]
    return ordered


def _emit_shared(parser, paired, code, idx, off, sinks, limit):
    """Two bins share a sink: write unit by unit so that the interleaving matches the reference."""
    bins = np.full(code.shape[0], 7, dtype=np.uint8)
    for b in range(6):
        bins[idx[int(off[b]):int(off[b + 1])]] = b
    for i in np.flatnonzero(bins < 6).tolist():
        if limit is not None and i >= limit:
            break
        b = int(bins[i])
        if sinks[b]:
            _write_bytes(sinks[b], parser.emit(paired, b, np.array([i], dtype=np.uint32)))


def _classify_parsed(ctx, mode, block, n, patches, cigar_mode, min_score):
    """One fused pass over the first n records of a parsed block (patched values merged in):
    -> (code u8[n], idx u32[units], bin_offsets u64[8], counts u64[64])."""
    bits = block.unit_bits
    if n < block.n:                                                # truncated at an input error
        flags = np.unpackbits(bits.view(np.uint8), bitorder="little")[:n]
        bits = np.packbits(np.concatenate([flags, np.zeros((-n) % 64, dtype=np.uint8)]), bitorder="little").view(np.uint64)
    cols = [np.array(c[:n]) for c in block.cols]
    fvals = {}                                                     # (record, column) -> value that needs binary64
    csr = None
    if cigar_mode:
        csr = [[np.array(block.csr[f][0][:n]), np.array(block.csr[f][1][:n + 1]), block.csr[f][2]] for f in (0, 1)]
    for k, vals in patches.items():
        if k >= n:
            continue
        for c, v in enumerate(vals):
            if isinstance(v, tuple):                                # ("cigar", nm, ops) of a record the stripper flagged
                _, nm, ops = v
                f = c // 2
                want = list(ops)
                have = csr[f][2][int(csr[f][1][k]):int(csr[f][1][k + 1])].tolist()
                if want != have:                                    # cannot happen: the op scan is deterministic
                    raise RuntimeError("CIGAR operations of record %d differ between parsers" % k)
                csr[f][0][k] = _ABSENT if nm is None else nm
            elif v == _NEG_INF:
                cols[c][k] = _ABSENT
            elif v == v and abs(v) <= _I32_MAX and float(v).is_integer():
                cols[c][k] = int(v)
            else:
                fvals[(k, c)] = float(v)
    integral = (min_score == min_score) and not fvals
    if cigar_mode:
        if integral:
            return ctx.classify_compact_cigar(mode, csr[0][0], csr[0][1], csr[0][2], cols[1],
                                              csr[1][0], csr[1][1], csr[1][2], cols[3], bits, _floor_min_score(min_score))
        fcols = []
        for f in (0, 1):
            a = ctx.cigar_scores(csr[f][0], csr[f][1], csr[f][2])
            fcols.append(np.where(a == _ABSENT, _NEG_INF, a.astype(np.float64)))
            fcols.append(np.where(cols[2 * f + 1] == _ABSENT, _NEG_INF, cols[2 * f + 1].astype(np.float64)))
    else:
        if integral:
            return ctx.classify_compact(mode, *cols, bits, _floor_min_score(min_score))
        fcols = [np.where(c == _ABSENT, _NEG_INF, c.astype(np.float64)) for c in cols]
    for (k, c), v in fvals.items():
        fcols[c][k] = v
    return ctx.classify_compact(mode, *fcols, bits, float(min_score))


def _finish_in_python(mode, path1, path2, pos, sinks, min_score, tag_func, skip_repeated, totals, key_order):
    """Non-ASCII input: continue from byte offsets `pos` with the Python reader (str.split() semantics)."""
    with open(path1, "rt") as f1, open(path2, "rt") as f2:
        f1.seek(pos[0])
        f2.seek(pos[1])
        pairs = _lockstep(lambda: f1.readline().strip("\n").split(), lambda: f2.readline().strip("\n").split(),
                          skip_repeated)                          # a plain generator: _run must not come back here
        rest = _run(mode, pairs, sinks, min_score, tag_func)
    ordered = Counter()
    for key in key_order:
        ordered[key] = totals[key]
    for key, val in rest.items():
        ordered[key] += val
    return ordered


def classify_sam_files(primary_sam, secondary_sam, primary_specific=sys.stdout, secondary_specific=None,
                       primary_multi=None, secondary_multi=None, unassigned=None, unresolved=None, paired=False,
                       conservative=False, min_score=float("-inf"), tag_func=get_tag, skip_repeated_reads=None,
                       n_threads=0, bam=False):
    """File-level entry point: classify two SAM files (paths) whose headers the caller has already dealt with
    (process_headers).  Equivalent to main_*(getReadPairs(open(primary_sam), open(secondary_sam), ...)) after the
    header lines, but parses and writes through the C++ stripper.  tag_func must be one of the three built-in
    plugins.  skip_repeated_reads defaults to `not paired`, as the command line does (ref :691).  bam=True: the
    inputs are BAM files, decoded natively to the text `samtools view` would print."""
    if tag_func not in (get_tag, get_tag_with_ZS_as_XS, get_cigarbased_AS_tag):
        raise ValueError("classify_sam_files needs a built-in tag_func; use main_* for custom plugins")
    if skip_repeated_reads is None:
        skip_repeated_reads = not paired
    mode = _ffi.MODE_SE if not paired else (_ffi.MODE_PE_CONSERVATIVE if conservative else _ffi.MODE_PE_LIBERAL)
    return _run_files(mode, primary_sam, secondary_sam,
                      _sinks(primary_specific, secondary_specific, primary_multi, secondary_multi, unassigned, unresolved),
                      min_score, tag_func, skip_repeated_reads, n_threads, bam)


def _sinks(primary_specific, secondary_specific, primary_multi, secondary_multi, unassigned, unresolved):
    # indexed by state: 0 PS, 1 SS, 2 PM, 3 SM, 4 unresolved, 5 unassigned
    return [primary_specific, secondary_specific, primary_multi, secondary_multi, unresolved, unassigned]


def main_single_end(readpairs, primary_specific=sys.stdout, secondary_specific=None, primary_multi=None,
                    secondary_multi=None, unassigned=None, unresolved=None, min_score=float("-inf"),
                    tag_func=get_tag):
    """Classify single-end reads (ref :291-352).  Returns a Counter keyed by state name."""
    return _run(_ffi.MODE_SE, readpairs, _sinks(primary_specific, secondary_specific, primary_multi,
                                                secondary_multi, unassigned, unresolved), min_score, tag_func)


def main_paired_end(readpairs, primary_specific=sys.stdout, secondary_specific=None, primary_multi=None,
                    secondary_multi=None, unassigned=None, unresolved=None, min_score=float("-inf"),
                    tag_func=get_tag):
    """Liberal paired-end loop: a pair goes to the highest-priority state of its mates (ref :354-454).
    Returns a Counter keyed by (forward_state, reverse_state)."""
    return _run(_ffi.MODE_PE_LIBERAL, readpairs, _sinks(primary_specific, secondary_specific, primary_multi,
                                                        secondary_multi, unassigned, unresolved), min_score, tag_func)


def conservative_main_paired_end(readpairs, primary_specific=sys.stdout, secondary_specific=None,
                                 primary_multi=None, secondary_multi=None, unassigned=None, unresolved=None,
                                 min_score=float("-inf"), tag_func=get_tag):
    """Conservative paired-end loop: any unassigned mate -> unassigned; unresolved or species-discordant
    -> unresolved; else highest priority (ref :456-556)."""
    return _run(_ffi.MODE_PE_CONSERVATIVE, readpairs, _sinks(primary_specific, secondary_specific, primary_multi,
                                                             secondary_multi, unassigned, unresolved), min_score, tag_func)


# --------------------------------------------------------------------------------------------
# command line (same flags as ref :568-678; wiring as ref :681-743)
# --------------------------------------------------------------------------------------------

# Prose of the command line (`xenomapper --help`, and what the usage error prints): the reference's own wording, kept
# byte for byte (xenomapper.py:570-668) because scripts and users read it; argparse re-flows the argument help, the
# description and epilogue are printed raw.
_CLI_DESCRIPTION = (
    "A script for parsing pairs of sam files and returning sam files\n"
    "containing only reads where no better mapping exist in other files.\n"
    "Used for filtering reads where multiple species may contribute \n"
    "(eg human tissue xenografted into mouse, pathogen growing on plant).\n"
    "\n"
    "Files should contain an AS and XS score and better matches must have\n"
    "a higher alignment score (but can be negative).\n"
    "Reads must be in the same order in both species.\n"
    "\n"
    "In practice this is best acchieved by using Bowtie2 in --local mode.\n"
    "If the -p option is used you must also use --reorder.\n"
    "\n"
    "Limited support is provided for aligners that do not produce AS and XS\n"
    "score tags via the --cigar_score option.\n"
    "\n"
    "All input files must be seekable\n"
    "(ie not a FIFO, process substitution or pipe)'\n")
_CLI_EPILOG = (
    "To output bam files in a bash shell use process subtitution:\n"
    "    xenomapper --primary_specific >(samtools view -bS - > outfilename.bam) \n"
    "\n"
    "This program is distributed in the hope that it will be useful,\n"
    "but WITHOUT ANY WARRANTY; without even the implied warranty of\n"
    "MERCHANTABILITY or FITNESS FOR A PARTICULAR PURPOSE.\n\n"
    "\n")
_SAM_IN = "a %s format Bowtie2 mapping output file corresponding to the %s"
_SAM_OUT = "name for SAM format output file for %s"
_CLI_ARGUMENTS = (          # (flag, kind, help); kinds: rt / rb / wt file types, flag, float
    ("primary_sam", "rt", _SAM_IN % ("SAM", "primary species of interest")),
    ("secondary_sam", "rt", _SAM_IN % ("SAM", "secondary or contaminating species")),
    ("primary_bam", "rb", _SAM_IN % ("BAM", "primary species of interest")),
    ("secondary_bam", "rb", _SAM_IN % ("BAM", "secondary or contaminating species")),
    ("primary_specific", "wt", _SAM_OUT % "reads mapping to a specific location in the primary species"),
    ("secondary_specific", "wt", _SAM_OUT % "reads mapping to a specific location in the secondary species"),
    ("primary_multi", "wt", _SAM_OUT % "reads multi mapping in the primary species"),
    ("secondary_multi", "wt", _SAM_OUT % "reads multi mapping in the secondary species"),
    ("unassigned", "wt", _SAM_OUT % "unassigned (non-mapping) reads"),
    ("unresolved", "wt", _SAM_OUT % "unresolved (maps equally well in both species) reads"),
    ("paired", "flag", "the SAM files consist of paired reads with forward and reverse reads occuring once and interlaced"),
    ("conservative", "flag", "conservatively allocate paired end reads with discordant category allocations. Only pairs "
     "that are both specific, or specific and multi will be allocated as specific. Pairs that are discordant for "
     "species will be deemed unresolved.  Pairs where any read is unassigned will be deemed unassigned."),
    ("min_score", "float", "the minimum mapping score.  Reads with scores less than or equal to min_score will be "
     "considered unassigned. Values should be chosen based on the mapping program and read length"),
    ("cigar_scores", "flag", "Use the cigar line and the NM tag to calculate a score. For aligners that do not support "
     "the AS tag. No determination of multimapping state will be done.  Reads that are unique in one species and "
     "multimap in the other species may be misassigned as no score can be calculated in the multimapping species. "
     "Score is -6 * mismatches + -5 * indel open + -3 * indel extend + -2 * softclip. Treatment of multimappers "
     "will vary with aligner.  If multimappers are assigned a cigar line they will be treated as species specific, "
     "otherwise as unassigned."),
    ("use_zs", "flag", "Use the value of the ZS tag in place of XS for determining the mapping score of the next best "
     "alignment.  Used with HISAT as the XS:A tag is conventionally used for strand in spliced mappers."),
    ("version", "flag", "print version information and exit"),
)


def command_line_interface(*args, **kw):
    """The reference's command line (xenomapper.py:568-678): same flags, defaults, help and usage-error behaviour."""
    parser = argparse.ArgumentParser(prog="xenomapper", formatter_class=argparse.RawDescriptionHelpFormatter,
                                     description=_CLI_DESCRIPTION, epilog=_CLI_EPILOG)
    for flag, kind, text in _CLI_ARGUMENTS:
        if kind == "flag":
            parser.add_argument("--" + flag, action="store_true", help=text)
        elif kind == "float":
            parser.add_argument("--" + flag, type=float, default=float("-inf"), help=text)
        else:
            parser.add_argument("--" + flag, type=argparse.FileType(kind),
                                default=sys.stdout if flag == "primary_specific" else None, help=text)
    ns = parser.parse_args(*args, **kw)
    if ns.version:
        print(__version__)
        sys.exit()
    if (not ns.primary_sam or not ns.secondary_sam) and (not ns.primary_bam or not ns.secondary_bam):
        print("ERROR: You must provide --primary_sam and --secondary_sam\n or --primary_bam and --secondary_bam\n")
        parser.print_help()
        sys.exit(1)
    return ns


def main(argv=None):
    args = command_line_interface(argv) if argv is not None else command_line_interface()
    if args.cigar_scores:
        tag_func = get_cigarbased_AS_tag
    elif args.use_zs:
        tag_func = get_tag_with_ZS_as_XS
    else:
        tag_func = get_tag
    sinks = dict(primary_specific=args.primary_specific, secondary_specific=args.secondary_specific,
                 primary_multi=args.primary_multi, secondary_multi=args.secondary_multi,
                 unassigned=args.unassigned, unresolved=args.unresolved)
    skip_repeated = not args.paired
    # regular files reach the C++ stripper / writer through the readpairs objects (see _ReadPairs); anything else is
    # split line by line in Python
    if args.primary_sam:
        process_headers(args.primary_sam, args.secondary_sam, **sinks)
        readpairs = getReadPairs(args.primary_sam, args.secondary_sam, skip_repeated_reads=skip_repeated)
    else:
        process_headers(args.primary_bam, args.secondary_bam, bam=True, **sinks)
        readpairs = getBamReadPairs(args.primary_bam, args.secondary_bam, skip_repeated_reads=skip_repeated)
    if args.paired:
        loop = conservative_main_paired_end if args.conservative else main_paired_end
    else:
        loop = main_single_end
    category_counts = loop(readpairs, min_score=args.min_score, tag_func=tag_func, **sinks)
    output_summary(category_counts=category_counts, outfile=sys.stderr)
    for sink in sinks.values():
        if sink and sink not in (sys.stdout, sys.stderr):
            sink.flush()


if __name__ == "__main__":  # pragma: no cover
    main()
