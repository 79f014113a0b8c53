"""xm_cigar_pack (CSR CIGAR columns -> packed CIGAR columns; host code of libxenomapper_hip.so, needs no GPU) against
the plain NumPy restatement of the layout in tests/helpers.py."""
import ctypes

import numpy as np
import pytest

from tests import helpers as H


def _csr(rng, n, long_at=()):
    k = rng.integers(0, 7, n).astype(np.int64)
    k[rng.random(n) < 0.3] = 0
    for at, length in long_at:
        if at < n:
            k[at] = length
    off = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(k, out=off[1:])
    ops = (rng.integers(1, 200, int(off[-1])).astype(np.uint32) << 4) | rng.integers(0, 9, int(off[-1])).astype(np.uint32)
    return off, ops


@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 257, 511, 512, 513, 1000, 5000])
def test_pack_matches_the_layout(n):
    from xenomapper_amd import _ffi
    rng = np.random.default_rng(n)
    off, ops = _csr(rng, n)
    cnt, tile, packed = _ffi.cigar_pack(off, ops)
    wcnt, wtile, wpacked = H.np_cigar_pack(off, ops)
    assert np.array_equal(cnt, wcnt) and np.array_equal(tile, wtile) and np.array_equal(packed, wpacked)
    assert tile.shape[0] == _ffi.cigar_tiles(n) + 1 and int(tile[-1]) == packed.shape[0] == int(off[-1])


@pytest.mark.parametrize("lengths", [(254,), (255,), (256,), (300, 255), (1000, 254, 255, 700)])
def test_pack_escapes_long_records(lengths):
    """Records with 255 ops or more: count byte 255, one trailer word n_ops << 4 | 15 after their ops; every later
    position moves by the trailers in front of it."""
    from xenomapper_amd import _ffi
    n = 1300
    rng = np.random.default_rng(len(lengths) * 1000 + lengths[0])
    spots = [0, 255, 256, 700, 1299]
    off, ops = _csr(rng, n, long_at=list(zip(spots, lengths)))
    cnt, tile, packed = _ffi.cigar_pack(off, ops)
    wcnt, wtile, wpacked = H.np_cigar_pack(off, ops)
    assert np.array_equal(cnt, wcnt) and np.array_equal(tile, wtile) and np.array_equal(packed, wpacked)
    n_esc = sum(1 for v in lengths if v >= 255)
    assert int((cnt == 255).sum()) == n_esc and packed.shape[0] == int(off[-1]) + n_esc
    # every escaped record is recoverable from its end: the word in front of the next record's begin
    begin = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.diff(off.astype(np.int64)) + (np.diff(off.astype(np.int64)) >= 255), out=begin[1:])
    for i in np.flatnonzero(cnt == 255):
        t = int(packed[begin[i + 1] - 1])
        assert t & 15 == 15 and t >> 4 == int(off[i + 1]) - int(off[i])


def test_pack_rejects_bad_columns():
    from xenomapper_amd import _ffi
    L = _ffi.lib()
    off = np.array([0, 3, 2], dtype=np.uint32)                   # decreasing offsets
    ops = np.zeros(4, dtype=np.uint32)
    cnt = np.zeros(2, dtype=np.uint8)
    tile = np.zeros(2, dtype=np.uint32)
    out = ctypes.c_uint64(0)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)              # noqa: E731
    assert L.xm_cigar_pack(2, p(off), p(ops), p(cnt), p(tile), None, 0, ctypes.byref(out)) == -1
    good = np.array([0, 300, 301], dtype=np.uint32)
    big = np.zeros(301, dtype=np.uint32)
    small = np.zeros(301, dtype=np.uint32)                        # one word short: the trailer does not fit
    assert L.xm_cigar_pack(2, p(good), p(big), p(cnt), p(tile), None, 0, ctypes.byref(out)) == 0 and out.value == 302
    assert L.xm_cigar_pack(2, p(good), p(big), p(cnt), p(tile), p(small), 301, ctypes.byref(out)) == -1
    assert L.xm_cigar_pack(2, p(good), p(big), None, p(tile), None, 0, ctypes.byref(out)) == -1


def test_cigar_pack_rejects_offsets_that_do_not_start_at_zero():
    """The packed op array may BE the CSR op array when no record needed a trailer (n_ops_packed == cig_off[n]); that
    equality only means "nothing was escaped" when the offsets start at 0, so anything else is refused."""
    import ctypes
    from xenomapper_amd import _ffi
    L = _ffi.lib()
    off = np.array([3, 4, 6], dtype=np.uint32)
    ops = np.zeros(8, dtype=np.uint32)
    cnt = np.zeros(2, dtype=np.uint8)
    tile = np.zeros(2, dtype=np.uint32)
    n_packed = ctypes.c_uint64(0)
    rc = L.xm_cigar_pack(2, off.ctypes.data_as(ctypes.c_void_p), ops.ctypes.data_as(ctypes.c_void_p),
                         cnt.ctypes.data_as(ctypes.c_void_p), tile.ctypes.data_as(ctypes.c_void_p), None, 0, ctypes.byref(n_packed))
    assert rc == -1
