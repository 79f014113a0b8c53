"""ctypes binding of include/xenomapper_host.h (libxenomapper_host.so): the multi-threaded C++ SAM column
stripper and line writer used by the file-to-file fast path of xenomapper_amd.xenomapper.main()."""
from __future__ import annotations

import ctypes
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XENOMAPPER_HOST_LIB") or os.path.join(PKG, "libxenomapper_host.so")   # override: sanitizer builds

SCORE_AS_XS, SCORE_AS_ZS, SCORE_CIGAR = 0, 1, 2
EX_NONINT, EX_DUP, EX_SHORT, EX_BIGLEN = 1, 2, 3, 4
ERR_NON_ASCII = -3

EXPORTED = ("xmh_abi_version", "xmh_strerror", "xmh_default_threads", "xmh_parser_create", "xmh_parser_destroy", "xmh_parse", "xmh_emit",
            "xmh_bam_open", "xmh_bam_open_header", "xmh_bam_close", "xmh_bam_header", "xmh_bam_read", "xmh_bam_read_pre", "xmh_parse_pre",
            "xmh_copy", "xmh_pread", "xmh_adopt_lines", "xmh_bam_records_start", "xmh_bam_print", "xmh_bam_walk")
NEED_TEXT = 1
# xmh_pre (include/xenomapper_host.h): what the BAM decoder knows about every line it prints
PRE_DTYPE = np.dtype([("line_len", np.uint32), ("name_len", np.uint16), ("flags", np.uint8), ("ex_as", np.uint8),
                      ("ex_xs", np.uint8), ("ex_zs", np.uint8), ("ex_nm", np.uint8), ("pad", np.uint8),
                      ("as", np.int32), ("xs", np.int32), ("zs", np.int32), ("nm", np.int32),
                      ("n_ops", np.uint32), ("ops_at", np.uint32)], align=True)
assert PRE_DTYPE.itemsize == 36
PRE_WORDS = 9            # the arrays travel as uint32[n, 9] (plain copies; numpy moves structured arrays field by field):
PRE_LINE_LEN, PRE_OPS_AT = 0, 8        # ... columns that are whole words


def pre_view(pre):
    """The named fields of a uint32[n, 9] description array (a view)."""
    return np.ascontiguousarray(pre).view(PRE_DTYPE).reshape(-1)

_P = ctypes.c_void_p


class _Block(ctypes.Structure):
    _fields_ = [("n_records", ctypes.c_uint64), ("consumed1", ctypes.c_uint64), ("consumed2", ctypes.c_uint64),
                ("ended", ctypes.c_int32), ("starved", ctypes.c_int32), ("mismatch_at", ctypes.c_int64),
                ("as1", _P), ("xs1", _P), ("as2", _P), ("xs2", _P), ("nm1", _P), ("nm2", _P),
                ("cig_off1", _P), ("cig_off2", _P), ("cig_ops1", _P), ("cig_ops2", _P), ("unit_bits", _P),
                ("line_off1", _P), ("line_off2", _P), ("line_len1", _P), ("line_len2", _P),
                ("norm_len1", _P), ("norm_len2", _P), ("line_flags1", _P), ("line_flags2", _P),
                ("n_exc", ctypes.c_uint64), ("exc_record", _P), ("exc_col", _P), ("exc_kind", _P),
                ("consumed_lines1", ctypes.c_uint64), ("consumed_lines2", ctypes.c_uint64)]


class NonAsciiInput(Exception):
    """The window holds bytes >= 0x80; Python's own str.split() must read this input."""


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `python -m xenomapper_amd.build`" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.xmh_abi_version.restype = ctypes.c_int
        L.xmh_strerror.argtypes = [ctypes.c_int]
        L.xmh_strerror.restype = ctypes.c_char_p
        L.xmh_default_threads.restype = ctypes.c_int
        L.xmh_parser_create.argtypes = [ctypes.c_int, ctypes.POINTER(_P)]
        L.xmh_parser_destroy.argtypes = [_P]
        L.xmh_parse.argtypes = [_P, _P, ctypes.c_uint64, ctypes.c_int, _P, ctypes.c_uint64, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64,
                                ctypes.POINTER(_Block)]
        L.xmh_emit.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_uint64, _P, ctypes.c_uint64,
                               ctypes.POINTER(ctypes.c_uint64)]
        L.xmh_bam_open.argtypes = [_P, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(_P)]
        L.xmh_bam_open_header.argtypes = [_P, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(_P)]
        L.xmh_bam_open_header.restype = ctypes.c_int
        L.xmh_bam_close.argtypes = [_P]
        L.xmh_bam_header.argtypes = [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_uint64)]
        L.xmh_bam_read.argtypes = [_P, _P, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
        L.xmh_bam_read_pre.argtypes = [_P, _P, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int),
                                       _P, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), _P, ctypes.c_uint64,
                                       ctypes.POINTER(ctypes.c_uint64)]
        L.xmh_parse_pre.argtypes = [_P, _P, ctypes.c_uint64, ctypes.c_int, _P, ctypes.c_uint64, _P, ctypes.c_uint64,
                                    _P, ctypes.c_uint64, ctypes.c_int, _P, ctypes.c_uint64, _P, ctypes.c_uint64,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(_Block)]
        L.xmh_bam_records_start.argtypes = [_P, ctypes.POINTER(ctypes.c_uint64)]
        L.xmh_bam_walk.argtypes = [_P, ctypes.c_uint64, ctypes.c_uint64, _P, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                                   ctypes.POINTER(ctypes.c_uint64)]
        L.xmh_bam_print.argtypes = [_P, _P, _P, ctypes.c_uint64, _P, _P, ctypes.c_uint64, _P, _P, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_uint64)]
        L.xmh_copy.argtypes = [_P, _P, _P, ctypes.c_uint64]
        L.xmh_pread.argtypes = [_P, ctypes.c_int, ctypes.c_uint64, _P, ctypes.c_uint64]
        L.xmh_adopt_lines.argtypes = [_P, ctypes.c_uint64] + [_P] * 8
        _lib = L
    return _lib


def _view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class Block(object):
    """NumPy views of one parsed window (valid until the parser parses again)."""

    def __init__(self, raw, cigar):
        n = self.n = int(raw.n_records)
        self.consumed = (int(raw.consumed1), int(raw.consumed2))
        self.consumed_lines = (int(raw.consumed_lines1), int(raw.consumed_lines2))
        self.ended, self.starved, self.mismatch_at = bool(raw.ended), bool(raw.starved), int(raw.mismatch_at)
        self.cols = [_view(p, n, np.int32) for p in (raw.as1, raw.xs1, raw.as2, raw.xs2)]
        self.unit_bits = _view(raw.unit_bits, (n + 63) // 64, np.uint64)
        if cigar:
            off = [_view(p, n + 1, np.uint32) for p in (raw.cig_off1, raw.cig_off2)]
            self.csr = [(_view(raw.nm1, n, np.int32), off[0], _view(raw.cig_ops1, int(off[0][-1]) if n else 0, np.uint32)),
                        (_view(raw.nm2, n, np.int32), off[1], _view(raw.cig_ops2, int(off[1][-1]) if n else 0, np.uint32))]
        else:
            self.csr = None
        self.line_off = [_view(raw.line_off1, n, np.uint64), _view(raw.line_off2, n, np.uint64)]
        self.line_len = [_view(raw.line_len1, n, np.uint32), _view(raw.line_len2, n, np.uint32)]
        k = int(raw.n_exc)
        self.exc = list(zip(_view(raw.exc_record, k, np.uint32).tolist(), _view(raw.exc_col, k, np.uint8).tolist(),
                            _view(raw.exc_kind, k, np.uint8).tolist()))


class Parser(object):
    def __init__(self, n_threads=0):
        self._L = lib()
        h = _P()
        rc = self._L.xmh_parser_create(int(n_threads), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError("xmh_parser_create: " + self._L.xmh_strerror(rc).decode())
        self._h = h
        self._win = None
        self._out = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.xmh_parser_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def parse(self, arr1, pos1, len1, eof1, arr2, pos2, len2, eof2, score_mode, paired, skip_repeated, keep_halo,
              max_records):
        """arr*: uint8 NumPy arrays over the whole files (memmap); the windows are [pos, pos+len)."""
        raw = _Block()
        p1 = arr1.ctypes.data + pos1 if len1 else None
        p2 = arr2.ctypes.data + pos2 if len2 else None
        rc = self._L.xmh_parse(self._h, p1, len1, int(eof1), p2, len2, int(eof2), score_mode, int(paired),
                               int(skip_repeated), int(keep_halo), int(max_records), ctypes.byref(raw))
        if rc == ERR_NON_ASCII:
            raise NonAsciiInput()
        if rc != 0:
            raise RuntimeError("xmh_parse: " + self._L.xmh_strerror(rc).decode())
        self._win = (p1, p2)
        self._keep = (arr1, arr2)            # the windows must outlive emit()
        return Block(raw, score_mode == SCORE_CIGAR)

    def parse_pre(self, arr1, pos1, len1, eof1, pre1, ops1, arr2, pos2, len2, eof2, pre2, ops2, score_mode, paired, skip_repeated,
                  keep_halo, max_records):
        """parse() on windows of text the BAM decoder wrote, from its line descriptions (pre*: PRE_DTYPE arrays for the lines
        from pos* on, ops*: the uint32 arrays their ops_at index) instead of tokenising the text again.  None: a line
        needs the text rules -- call parse() on the same windows.  pre*: uint32[n, 9] as read_into_pre returns them."""
        raw = _Block()
        p1 = arr1.ctypes.data + pos1 if len1 else None
        p2 = arr2.ctypes.data + pos2 if len2 else None
        rc = self._L.xmh_parse_pre(self._h, p1, len1, int(eof1), pre1.ctypes.data if pre1.shape[0] else None, pre1.shape[0],
                                   ops1.ctypes.data if ops1.shape[0] else None, ops1.shape[0],
                                   p2, len2, int(eof2), pre2.ctypes.data if pre2.shape[0] else None, pre2.shape[0],
                                   ops2.ctypes.data if ops2.shape[0] else None, ops2.shape[0],
                                   score_mode, int(paired), int(skip_repeated), int(keep_halo), int(max_records), ctypes.byref(raw))
        if rc == NEED_TEXT:
            return None
        if rc != 0:
            raise RuntimeError("xmh_parse_pre: " + self._L.xmh_strerror(rc).decode())
        self._win = (p1, p2)
        self._keep = (arr1, arr2, pre1, ops1, pre2, ops2)
        return Block(raw, score_mode == SCORE_CIGAR)

    def copy(self, dst_address, arr, pos, n):
        """arr[pos:pos + n] (uint8, e.g. a memory-mapped file) copied to dst_address by the parser's threads."""
        if n:
            rc = self._L.xmh_copy(self._h, _P(dst_address), _P(arr.ctypes.data + pos), int(n))
            if rc != 0:
                raise RuntimeError("xmh_copy: " + self._L.xmh_strerror(rc).decode())

    def pread(self, fd, offset, dst_address, n):
        """Bytes [offset, offset + n) of the open file `fd` read to dst_address by the parser's threads."""
        if n:
            rc = self._L.xmh_pread(self._h, int(fd), int(offset), _P(dst_address), int(n))
            if rc != 0:
                raise OSError("xmh_pread: could not read %d bytes at offset %d" % (n, offset))

    def adopt_lines(self, arr1, pos1, arr2, pos2, n, tables):
        """Make a block that was stripped on the GPU the parser's current block: tables = the eight host addresses
        (line_off, line_len, norm_len, line_flags of file 1, then of file 2) of xm_strip_block; the windows start at
        arr1[pos1] / arr2[pos2].  emit*() then write its units."""
        rc = self._L.xmh_adopt_lines(self._h, int(n), *[_P(t) for t in tables])
        if rc != 0:
            raise RuntimeError("xmh_adopt_lines: " + self._L.xmh_strerror(rc).decode())
        self._win = (arr1.ctypes.data + pos1, arr2.ctypes.data + pos2)
        self._keep = (arr1, arr2)

    def emit_size(self, paired, bin_index, idx):
        """Bytes xmh_emit would write for these units (first half of its two-call protocol).  -> (idx as uint32, bytes)"""
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        if idx.shape[0] == 0:
            return idx, 0
        need = ctypes.c_uint64()
        rc = self._L.xmh_emit(self._h, self._win[0], self._win[1], int(paired), bin_index, idx.ctypes.data_as(_P), idx.shape[0],
                              None, 0, ctypes.byref(need))
        if rc != 0:
            raise RuntimeError("xmh_emit: " + self._L.xmh_strerror(rc).decode())
        return idx, int(need.value)

    def emit_to(self, paired, bin_index, idx, address, capacity):
        """xmh_emit straight into caller memory (e.g. the mapped pages of an output file); idx: uint32, as emit_size
        returned it.  -> bytes written"""
        need = ctypes.c_uint64()
        rc = self._L.xmh_emit(self._h, self._win[0], self._win[1], int(paired), bin_index, idx.ctypes.data_as(_P), idx.shape[0],
                              _P(address), capacity, ctypes.byref(need))
        if rc != 0:
            raise RuntimeError("xmh_emit: " + self._L.xmh_strerror(rc).decode())
        return int(need.value)

    def emit(self, paired, bin_index, idx, reuse=False):
        """Text (bytes) of one output bin for the last parsed block; idx: ascending uint32 unit indices.
        reuse=True returns a view of a buffer the parser keeps (valid until the next emit): a fresh multi-megabyte
        array per call is mapped, page-faulted and unmapped every time."""
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        n = idx.shape[0]
        if n == 0:
            return b""
        need = ctypes.c_uint64()
        ip = idx.ctypes.data_as(_P)
        rc = self._L.xmh_emit(self._h, self._win[0], self._win[1], int(paired), bin_index, ip, n, None, 0,
                              ctypes.byref(need))
        if rc != 0:
            raise RuntimeError("xmh_emit: " + self._L.xmh_strerror(rc).decode())
        if reuse:
            if self._out is None or self._out.shape[0] < need.value:
                self._out = np.empty(max(int(need.value * 1.25), 1 << 20), dtype=np.uint8)
            out = self._out[:need.value]
        else:
            out = np.empty(need.value, dtype=np.uint8)
        rc = self._L.xmh_emit(self._h, self._win[0], self._win[1], int(paired), bin_index, ip, n,
                              out.ctypes.data_as(_P), need.value, ctypes.byref(need))
        if rc != 0:
            raise RuntimeError("xmh_emit: " + self._L.xmh_strerror(rc).decode())
        return out


def bam_walk(raw_address, length, start, rec_off):
    """The record chain of an inflated window on the host (xmh_bam_walk) -> (records, stop); rec_off: uint32 array to fill."""
    n, stop = ctypes.c_uint64(), ctypes.c_uint64()
    rc = lib().xmh_bam_walk(_P(raw_address), int(length), int(start), rec_off.ctypes.data_as(_P), rec_off.shape[0],
                            ctypes.byref(n), ctypes.byref(stop))
    if rc != 0:
        raise ValueError("xmh_bam_walk: " + lib().xmh_strerror(rc).decode())
    return int(n.value), int(stop.value)


class BamReader(object):
    """BAM file image -> SAM text, like `samtools view` (header via .header(), lines via .read_into())."""

    def __init__(self, data, n_threads=0, header_only=False):
        """header_only: header(), records_start() and print_records() only (the GPU BAM path); the file is not indexed."""
        self._L = lib()
        self._data = np.ascontiguousarray(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        h = _P()
        opener = self._L.xmh_bam_open_header if header_only else self._L.xmh_bam_open
        rc = opener(self._data.ctypes.data if self._data.shape[0] else None, self._data.shape[0], int(n_threads), ctypes.byref(h))
        if rc != 0:
            raise ValueError("xmh_bam_open: " + self._L.xmh_strerror(rc).decode())
        self._h = h
        self.eof = False

    def close(self):
        if getattr(self, "_h", None):
            self._L.xmh_bam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def header(self):
        text, n = _P(), ctypes.c_uint64()
        self._L.xmh_bam_header(self._h, ctypes.byref(text), ctypes.byref(n))
        return ctypes.string_at(text, n.value).decode("ascii") if n.value else ""

    def records_start(self):
        """Inflated offset of the first alignment record (right after opening: behind magic, header text, reference list)."""
        v = ctypes.c_uint64()
        rc = self._L.xmh_bam_records_start(self._h, ctypes.byref(v))
        if rc != 0:
            raise ValueError("xmh_bam_records_start: " + self._L.xmh_strerror(rc).decode())
        return int(v.value)

    def print_records(self, raw_address, rec_off_address, n, out, line_off, line_len, sparse=False, wanted=None):
        """SAM text of records [0, n) of an inflated window (host addresses of the bytes and of the uint32 record table) into
        the uint8 array `out`; fills the uint32 arrays line_off / line_len.  -> bytes written, or -needed when `out` is
        too small.  sparse: one pass, every thread into its own worst-case stretch of `out` (lines where line_off says).
        wanted: uint8 per record, 0 = not printed (no sink takes it)."""
        w = ctypes.c_uint64()
        rc = self._L.xmh_bam_print(self._h, _P(raw_address), _P(rec_off_address), int(n),
                                   wanted.ctypes.data_as(_P) if wanted is not None else None, out.ctypes.data_as(_P), out.shape[0],
                                   line_off.ctypes.data_as(_P), line_len.ctypes.data_as(_P), int(bool(sparse)), ctypes.byref(w))
        if rc == -1 and sparse and w.value > 0xFFFFFFFF:
            # the one-pass form reserves every thread's WORST case (five times the record bytes) and its line table is 32-bit: a
            # window whose estimate passes 4 GiB (a large carried tail, XENOMAPPER_BAM_WINDOW_MB >= ~800) is printed in two
            # passes instead, which needs the real size only
            return self.print_records(raw_address, rec_off_address, n, out, line_off, line_len, False, wanted)
        if rc == -1 and w.value > out.shape[0]:
            return -int(w.value)
        if rc != 0:
            raise ValueError("xmh_bam_print: " + self._L.xmh_strerror(rc).decode())
        return int(w.value)

    def read_into_pre(self, out, start):
        """read_into() that also returns what the decoder knows about the lines it wrote:
        -> (bytes written, uint32[n, 9] descriptions of the lines (pre_view() names the fields), uint32 array of their
        CIGAR operations)."""
        cap = out.shape[0] - start
        pre = np.empty((cap // 24 + 16, PRE_WORDS), dtype=np.uint32)   # no SAM line of 11 fields is shorter than 21 bytes + '\n'
        ops = np.empty(max(cap // 8, 1 << 16), dtype=np.uint32)
        w, eof, n_pre, n_ops = ctypes.c_uint64(), ctypes.c_int(), ctypes.c_uint64(), ctypes.c_uint64()
        rc = self._L.xmh_bam_read_pre(self._h, out.ctypes.data + start, cap, ctypes.byref(w), ctypes.byref(eof),
                                      pre.ctypes.data, pre.shape[0], ctypes.byref(n_pre), ops.ctypes.data, ops.shape[0],
                                      ctypes.byref(n_ops))
        if rc != 0:
            raise ValueError("xmh_bam_read_pre: " + self._L.xmh_strerror(rc).decode())
        self.eof = bool(eof.value)
        return int(w.value), pre[:n_pre.value], ops[:n_ops.value]

    def read_into(self, out, start):
        """Append whole SAM lines to the uint8 array `out` from offset `start`; returns bytes written."""
        w, eof = ctypes.c_uint64(), ctypes.c_int()
        rc = self._L.xmh_bam_read(self._h, out.ctypes.data + start, out.shape[0] - start, ctypes.byref(w), ctypes.byref(eof))
        if rc != 0:
            raise ValueError("xmh_bam_read: " + self._L.xmh_strerror(rc).decode())
        self.eof = bool(eof.value)
        return int(w.value)
