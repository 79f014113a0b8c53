"""Shared test plumbing: golden vectors, the oracle (Python + C), SAM inputs of golden cases."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import xm_oracle as ORACLE          # noqa: E402  (tests may import the oracle)
from xenomapper_amd import synth                 # noqa: E402

NEG = float("-inf")
STATES = list(ORACLE.STATE_NAMES)
MODES = {"se": 0, "pe": 1, "pe_conservative": 2}


def golden(name):
    with open(os.path.join(GOLDEN, name), "rt") as fh:
        return json.load(fh)


def unnum(x):
    return float(x) if isinstance(x, str) else x


_C = None


def c_oracle():
    """ctypes handle on oracle/libxm_oracle.so (built on demand with oracle/Makefile)."""
    global _C
    if _C is not None:
        return _C
    so = os.path.join(REPO, "oracle", "libxm_oracle.so")
    src = os.path.join(REPO, "oracle", "xm_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        import fcntl
        with open(os.path.join(REPO, "oracle", ".build.lock"), "w") as lock:       # several ranks may get here at once
            fcntl.flock(lock, fcntl.LOCK_EX)
            if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
                subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "-s"])
    lib = ctypes.CDLL(so)
    i32, u64, f64 = ctypes.c_int32, ctypes.c_uint64, ctypes.c_double
    P = ctypes.c_void_p
    lib.xmo_state_i32.argtypes = [i32] * 5
    lib.xmo_state_f64.argtypes = [f64] * 5
    lib.xmo_bin.argtypes = [ctypes.c_int] * 3
    lib.xmo_classify_i32.argtypes = [ctypes.c_int, u64, P, P, P, P, P, i32, P, P]
    lib.xmo_classify_i32.restype = None
    lib.xmo_classify_f64.argtypes = [ctypes.c_int, u64, P, P, P, P, P, f64, P, P]
    lib.xmo_classify_f64.restype = None
    lib.xmo_compact.argtypes = [ctypes.c_int, u64, P, P, P]
    lib.xmo_compact.restype = None
    lib.xmo_cigar_scores.argtypes = [u64, P, P, P, P]
    lib.xmo_cigar_scores.restype = u64
    lib.xmo_mate_correlate.argtypes = [u64, P, u64, P, P]
    lib.xmo_mate_correlate.restype = None
    _C = lib
    return lib


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def c_classify(mode, as1, xs1, as2, xs2, unit_bits, m):
    """Run the C oracle's main-loop core.  Columns int32 or float64.  -> (code u8[n], counts u64[64])"""
    lib = c_oracle()
    n = as1.shape[0]
    code = np.empty(n, dtype=np.uint8)
    counts = np.zeros(64, dtype=np.uint64)
    cols = [np.ascontiguousarray(c) for c in (as1, xs1, as2, xs2)]
    bits = np.ascontiguousarray(unit_bits, dtype=np.uint64)
    if cols[0].dtype == np.int32:
        lib.xmo_classify_i32(mode, n, *[ptr(c) for c in cols], ptr(bits), int(m), ptr(code), ptr(counts))
    else:
        lib.xmo_classify_f64(mode, n, *[ptr(c) for c in cols], ptr(bits), float(m), ptr(code), ptr(counts))
    return code, counts


def c_compact(mode, code):
    lib = c_oracle()
    n = code.shape[0]
    idx = np.empty(max(n, 1), dtype=np.uint32)
    off = np.zeros(8, dtype=np.uint64)
    lib.xmo_compact(mode, n, ptr(code), ptr(idx), ptr(off))
    return idx[:int(off[7])], off


def c_cigar_scores(nm, cig_off, cig_oplen):
    lib = c_oracle()
    n = nm.shape[0]
    out = np.empty(n, dtype=np.int32)
    ops = cig_oplen if cig_oplen.shape[0] else np.zeros(1, dtype=np.uint32)
    bad = lib.xmo_cigar_scores(n, ptr(nm), ptr(cig_off), ptr(ops), ptr(out))
    return out, int(bad)


def case_texts(case):
    """The two SAM texts a G3 golden case was recorded on."""
    src = case["source"]
    if src["kind"] == "ref_data":
        texts = []
        for name in src["files"]:
            with open(os.path.join(GOLDEN, "ref_data", name), "rt") as fh:
                texts.append(fh.read())
        return texts
    if src["kind"] == "inline":
        return list(src["text"])
    import hashlib
    t1, t2, _ = synth.sam_text_pair(**src["args"])
    got = [hashlib.sha224(t.encode()).hexdigest() for t in (t1, t2)]
    assert got == src["sha224"], "synthetic generator drifted from the text the golden run saw"
    return [t1, t2]


def floor_min_score(m):
    """float min_score -> the int32 threshold of the integer path."""
    import math
    if m != m:
        raise ValueError("NaN min_score has no integer form")
    if m == NEG:
        return -2**31
    if m == -NEG:
        return 2**31 - 1
    return int(max(-2**31, min(2**31 - 1, math.floor(m))))


def c_mate_correlate(track, density):
    lib = c_oracle()
    track = np.ascontiguousarray(track, dtype=np.float64)
    density = np.ascontiguousarray(density, dtype=np.float64)
    out = np.empty(track.shape[0], dtype=np.float64)
    t = track if track.shape[0] else np.zeros(1)
    d = density if density.shape[0] else np.zeros(1)
    lib.xmo_mate_correlate(track.shape[0], ptr(t), density.shape[0], ptr(d), ptr(out if out.shape[0] else np.zeros(1)))
    return out


def np_cigar_pack(cig_off, cig_oplen):
    """Independent restatement (plain Python/NumPy) of the packed CIGAR column layout of include/xenomapper_hip.h: a
    byte of op count per record (255 = 255 or more, the record's ops are then followed by a trailer word
    n_ops << 4 | 15), the op position of every 256-record tile, the packed op array.  The checker for xm_cigar_pack."""
    off = np.asarray(cig_off, dtype=np.int64)
    ops = np.asarray(cig_oplen, dtype=np.uint32)
    n = off.shape[0] - 1
    k = np.diff(off)
    cnt = np.minimum(k, 255).astype(np.uint8)
    stretch = k + (k >= 255)                                   # ops + trailer
    begin = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(stretch, out=begin[1:])
    n_tiles = (n + 255) // 256
    tile = np.empty(n_tiles + 1, dtype=np.uint32)
    tile[:n_tiles] = begin[0:n:256]
    tile[n_tiles] = begin[n]
    if not (k >= 255).any():
        return cnt, tile, ops[:int(off[n])].copy()
    packed = np.empty(int(begin[n]), dtype=np.uint32)
    for i in range(n):                                          # small inputs only
        packed[begin[i]:begin[i] + k[i]] = ops[off[i]:off[i + 1]]
        if k[i] >= 255:
            packed[begin[i] + k[i]] = (int(k[i]) << 4) | 15
    return cnt, tile, packed
