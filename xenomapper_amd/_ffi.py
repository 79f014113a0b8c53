"""ctypes binding of the C ABI in include/xenomapper_hip.h.

There is no CPU fallback: if libxenomapper_hip.so is missing, or no gfx950 device is present,
every entry point raises.  `Context` mirrors the C calls one to one; NumPy arrays go through the
host-buffer entry points, torch device tensors through the *_dev entry points.
"""
from __future__ import annotations

import contextlib
import ctypes
import functools
import importlib.util
import os
import sys
import threading
import weakref

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
# XENOMAPPER_HIP_LIB: another build of the same library (tuning builds for A/B runs); never a different implementation
LIB_PATH = os.environ.get("XENOMAPPER_HIP_LIB") or os.path.join(PKG, "libxenomapper_hip.so")

MODE_SE, MODE_PE_LIBERAL, MODE_PE_CONSERVATIVE = 0, 1, 2
NO_UNIT = 0xFF
ABSENT = -2**31
MAX_RECORDS = 0xFFFFF000
KERNELS = ("classify", "hist", "scan", "scatter", "cigar", "correlate")

XM_OK = 0
_ERRORS = {-1: ValueError, -2: RuntimeError, -3: RuntimeError, -4: MemoryError, -5: OverflowError, -6: RuntimeError}
UNIQUE_ID_BYTES = 128


def bins4_bytes(n_records):
    """XM_BINS4_BYTES: size of the compact category stream for n_records (whole 1024-byte blocks of 2048 records)."""
    return ((int(n_records) + 2047) // 2048) * 1024


def unpack_bins4(bins4, n_records):
    """The output bin of every record (uint8: 0..5, 6 = state-6 unit, 7 = closes no unit) from a compact category
    stream as xm_classify_compact*_dev writes it (host NumPy array): record r = nibble r & 1 of byte r >> 1."""
    b = np.ascontiguousarray(bins4[:(int(n_records) + 1) // 2])
    out = np.empty(2 * b.shape[0], dtype=np.uint8)
    out[0::2] = b & 7
    out[1::2] = (b >> 4) & 7
    return out[:n_records]


class HipExtensionMissing(RuntimeError):
    pass


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 (soname libamdhip64.so.7)
    but asks for it as `libamdhip64.so`, so if this library pulled in /opt/rocm's copy first, a later
    `import torch` would bring up a second runtime and one of the two would see no device.  When torch
    is installed, load its copy first; both then resolve to the same runtime."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """The loaded shared library; raises HipExtensionMissing when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            "%s is missing: build it with `python -m xenomapper_amd.build` (there is no CPU fallback)" % LIB_PATH)
    _preload_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    P, I, U64, I32, F64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int32, ctypes.c_double
    sig = {
        "xm_abi_version": ([], I),
        "xm_strerror": ([I], ctypes.c_char_p),
        "xm_last_hip_error": ([P], ctypes.c_char_p),
        "xm_ctx_create": ([I, ctypes.POINTER(P)], I),
        "xm_ctx_destroy": ([P], I),
        "xm_ctx_device_info": ([P, ctypes.POINTER(I), ctypes.c_char_p, ctypes.c_size_t], I),
        "xm_classify": ([P, I, U64, P, P, P, P, P, I32, P, P], I),
        "xm_classify_f64": ([P, I, U64, P, P, P, P, P, F64, P, P], I),
        "xm_cigar_scores": ([P, U64, P, P, P, P], I),
        "xm_classify_cigar": ([P, I, U64, P, P, P, P, P, P, P, P, P, I32, P, P], I),
        "xm_classify_cigar_dev": ([P, P, I, U64, P, P, P, P, P, P, P, P, P, I32, P, P], I),
        "xm_compact": ([P, I, U64, P, P, P, P], I),
        "xm_classify_compact": ([P, I, U64, P, P, P, P, P, I32, P, P, P, P], I),
        "xm_classify_compact_f64": ([P, I, U64, P, P, P, P, P, F64, P, P, P, P], I),
        "xm_classify_compact_cigar": ([P, I, U64, P, P, P, P, P, P, P, P, P, I32, P, P, P, P], I),
        "xm_mate_correlate": ([P, U64, P, U64, P, P], I),
        "xm_mate_correlate_dev": ([P, P, U64, P, U64, P, P], I),
        "xm_classify_dev": ([P, P, I, U64, P, P, P, P, P, I32, P], I),
        "xm_classify_f64_dev": ([P, P, I, U64, P, P, P, P, P, F64, P], I),
        "xm_cigar_scores_dev": ([P, P, U64, P, P, P, P, P], I),
        "xm_compact_dev": ([P, P, I, U64, P, P, P, P], I),
        "xm_classify_compact_dev": ([P, P, I, U64, P, P, P, P, P, I32, P, P, P, P, P], I),
        "xm_classify_compact_f64_dev": ([P, P, I, U64, P, P, P, P, P, F64, P, P, P, P, P], I),
        "xm_classify_compact_cigar_dev": ([P, P, I, U64, P, P, P, P, P, P, P, P, P, I32, P, P, P, P, P], I),
        "xm_host_register": ([P, P, ctypes.c_size_t], I),
        "xm_host_unregister": ([P, P], I),
        "xm_pinned_bytes": ([P, P, P], I),
        "xm_cigar_pack": ([U64, P, P, P, P, P, U64, ctypes.POINTER(U64)], I),
        "xm_classify_compact_cigar_packed_dev": ([P, P, I, U64, P, P, P, P, P, P, P, P, P, P, P, I32, P, P, P, P, P, P], I),
        "xm_classify_place": ([P, I, U64, P, P, P, P, P, I32, P, P, U64, P, P], I),
        "xm_classify_place_f64": ([P, I, U64, P, P, P, P, P, F64, P, P, P, U64, P, P], I),
        "xm_classify_place_dev": ([P, P, I, U64, P, P, P, P, P, I32, P, P, P, U64, P, P], I),
        "xm_classify_place_f64_dev": ([P, P, I, U64, P, P, P, P, P, F64, P, P, P, P, U64, P, P], I),
        "xm_classify_place_cigar_packed_dev": ([P, P, I, U64, P, P, P, P, P, P, P, P, P, P, P, I32, P, P, P, P, U64, P, P], I),
        "xm_classify_runs_dev": ([P, P, I, U64, P, P, P, P, P, I32, P, P, P, P], I),
        "xm_classify_runs_f64_dev": ([P, P, I, U64, P, P, P, P, P, F64, P, P, P, P], I),
        "xm_runs_expand": ([U64, P, P, I, P, U64, ctypes.POINTER(U64)], I),
        "xm_stream_probe_dev": ([P, P, U64, P, P, P, P, P], I),
        "xm_workspace_is_clean": ([P, ctypes.POINTER(I)], I),
        "xm_workspace_release": ([P, P], I),
        "xm_comm_unique_id": ([P], I),
        "xm_comm_init": ([P, I, I, P], I),
        "xm_comm_destroy": ([P], I),
        "xm_comm_size": ([P], I),
        "xm_allreduce_counts": ([P, P, P], I),
        "xm_timing_enable": ([P, I], I),
        "xm_timing_select": ([P, ctypes.c_uint32], I),
        "xm_timing_reset": ([P], I),
        "xm_timing_read": ([P, P, P], I),
        # include/xenomapper_strip.h
        "xms_abi_version": ([], I),
        "xm_strip_create": ([P, I, ctypes.POINTER(P)], I),
        "xm_strip_destroy": ([P], I),
        "xm_strip_reserve": ([P, I, U64, U64], I),
        "xm_strip_staging": ([P, I, I], P),
        "xm_strip_begin_behind": ([P, I, I, U64], I),
        "xm_strip_set_lead": ([P, I, I, U64], I),
        "xm_strip_upload": ([P, I, I, U64, U64], I),
        "xm_strip_run": ([P, I, U64, I, U64, I, I, I, I, I, U64, P], I),
        "xm_strip_cigar_columns": ([P, I, I, U64, P, P, P, P, U64, ctypes.POINTER(U64)], I),
        "xm_strip_classify": ([P, I, I, U64, I32, ctypes.POINTER(P), ctypes.POINTER(P), P, P], I),
        "xm_strip_fetch_bins": ([P, I, U64, I, ctypes.c_uint32, P], I),
        "xm_strip_out_wait": ([P, I], I),
        "xm_strip_columns": ([P, I, U64, P, P, P, P, P], I),
        "xm_strip_device_columns": ([P, I, P], I),
        "xm_strip_last_error": ([P], ctypes.c_char_p),
        # include/xenomapper_bgzf.h
        "xm_bgzf_index": ([P, U64, U64, U64, P, P, U64, ctypes.POINTER(U64), ctypes.POINTER(U64), ctypes.POINTER(U64)], I),
        "xm_bgzf_index_prefix": ([P, U64, U64, U64, P, P, U64, ctypes.POINTER(U64), ctypes.POINTER(U64), ctypes.POINTER(U64)], I),
        "xm_bgzf_inflate_dev": ([P, P, P, P, U64, P, P, P], I),
        "xm_bgzf_inflate_walk_dev": ([P, P, P, P, U64, P, P, P, P], I),
        "xm_bgzf_crc32_dev": ([P, P, P, P, U64, P], I),
        "xm_bgzf_strerror": ([ctypes.c_uint32], ctypes.c_char_p),
        "xm_bamdev_create": ([P, I, ctypes.POINTER(P)], I),
        "xm_bamdev_destroy": ([P], I),
        "xm_bamdev_reserve": ([P, I, U64, U64, U64, U64], I),
        "xm_bamdev_staging": ([P, I, I], P),
        "xm_bamdev_run": ([P, I, P, I, I, I, I, U64, P], I),
        "xm_bamdev_raw_wait": ([P, I], I),
        "xm_bamdev_fetch_raw": ([P, I], I),
        "xm_bamdev_raw": ([P, I, I], P),
        "xm_bamdev_fetch_wanted": ([P, I, U64, I, ctypes.c_uint32, P], I),
        "xm_bamdev_upload": ([P, I, I, U64], I),
        "xm_bamdev_set_refs": ([P, I, P, P, ctypes.c_uint32], I),
        "xm_bamdev_fetch_text": ([P, I, U64, I, ctypes.c_uint32, P], I),
        "xm_bamdev_fetch_bins": ([P, I, U64, I, ctypes.c_uint32, P], I),
        "xm_bamdev_classify": ([P, I, I, U64, I32, ctypes.POINTER(P), ctypes.POINTER(P), P, P], I),
        "xm_bamdev_columns": ([P, I, U64, P, P, P, P, P], I),
        "xm_bamdev_cigar_columns": ([P, I, I, U64, P, P, P, P, U64, ctypes.POINTER(ctypes.c_uint64)], I),
        "xm_bamdev_last_error": ([P], ctypes.c_char_p),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res
    _lib = L
    return L


EXPORTED = ("xm_abi_version", "xm_strerror", "xm_last_hip_error", "xm_ctx_create", "xm_ctx_destroy",
            "xm_ctx_device_info", "xm_classify", "xm_classify_f64", "xm_cigar_scores", "xm_classify_cigar",
            "xm_compact", "xm_classify_compact", "xm_classify_compact_f64", "xm_classify_compact_cigar", "xm_mate_correlate", "xm_mate_correlate_dev", "xm_classify_dev", "xm_classify_f64_dev", "xm_classify_cigar_dev", "xm_cigar_scores_dev",
            "xm_compact_dev", "xm_classify_compact_dev", "xm_classify_compact_f64_dev", "xm_classify_compact_cigar_dev",
            "xm_cigar_pack", "xm_classify_compact_cigar_packed_dev", "xm_host_register", "xm_host_unregister", "xm_pinned_bytes",
            "xm_classify_place", "xm_classify_place_f64", "xm_classify_place_dev", "xm_classify_place_f64_dev",
            "xm_classify_place_cigar_packed_dev", "xm_classify_runs_dev", "xm_classify_runs_f64_dev", "xm_runs_expand",
            "xm_stream_probe_dev", "xm_workspace_is_clean", "xm_workspace_release",
            "xm_comm_unique_id", "xm_comm_init", "xm_comm_destroy", "xm_comm_size", "xm_allreduce_counts",
            "xm_timing_enable", "xm_timing_select", "xm_timing_reset", "xm_timing_read",
            "xms_abi_version", "xm_strip_create", "xm_strip_destroy", "xm_strip_reserve", "xm_strip_staging", "xm_strip_begin_behind", "xm_strip_set_lead", "xm_strip_upload", "xm_strip_run",
            "xm_strip_classify", "xm_strip_fetch_bins", "xm_strip_out_wait", "xm_strip_columns", "xm_strip_cigar_columns", "xm_strip_device_columns", "xm_strip_last_error",
            "xm_bgzf_index", "xm_bgzf_index_prefix", "xm_bgzf_inflate_dev", "xm_bgzf_inflate_walk_dev", "xm_bgzf_crc32_dev", "xm_bgzf_strerror",
            "xm_bamdev_create", "xm_bamdev_destroy", "xm_bamdev_reserve", "xm_bamdev_staging", "xm_bamdev_run", "xm_bamdev_raw_wait", "xm_bamdev_fetch_raw", "xm_bamdev_raw", "xm_bamdev_fetch_wanted", "xm_bamdev_fetch_text", "xm_bamdev_fetch_bins", "xm_bamdev_set_refs", "xm_bamdev_upload", "xm_bamdev_classify",
            "xm_bamdev_columns", "xm_bamdev_cigar_columns", "xm_bamdev_last_error")


def _np_ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def pinned_bytes():
    """xm_pinned_bytes: {"allocated": page-locked bytes the front ends hold, "registered": caller memory locked through
    host_register, "peak": the largest sum so far} -- process-wide."""
    a, r, pk = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    lib().xm_pinned_bytes(ctypes.byref(a), ctypes.byref(r), ctypes.byref(pk))
    return {"allocated": int(a.value), "registered": int(r.value), "peak": int(pk.value)}


def _as(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


RUNS_GRAN = 2048


def runs_granules(n_records):
    """XM_RUNS_GRANULES: granules of 2048 records of the segmented bin lists (xm_classify_runs*_dev)."""
    return (int(n_records) + RUNS_GRAN - 1) // RUNS_GRAN


def runs_expand(n_records, runs16, gran_counts, bin):
    """List `bin` (0..6) of the segmented form as a flat uint32 array of record indices in input order (xm_runs_expand;
    host arrays, no device)."""
    runs16 = _as(runs16, np.uint16)
    gran_counts = _as(gran_counts, np.uint16)
    n = int(n_records)
    assert runs16.shape[0] >= runs_granules(n) * RUNS_GRAN and gran_counts.shape[0] >= runs_granules(n) * 8
    L = lib()
    k = ctypes.c_uint64()
    rc = L.xm_runs_expand(n, _np_ptr(runs16), _np_ptr(gran_counts), int(bin), None, 0, ctypes.byref(k))
    if rc == XM_OK:
        out = np.empty(max(int(k.value), 1), dtype=np.uint32)
        rc = L.xm_runs_expand(n, _np_ptr(runs16), _np_ptr(gran_counts), int(bin), _np_ptr(out), out.shape[0], ctypes.byref(k))
    if rc != XM_OK:
        raise _ERRORS.get(rc, RuntimeError)("xm_runs_expand: %s" % L.xm_strerror(rc).decode())
    return out[:int(k.value)]


def cigar_tiles(n_records):
    """XM_CIG_TILES: 256-record tiles of the packed CIGAR columns."""
    return (int(n_records) + 255) // 256


def cigar_pack(cig_off, cig_oplen):
    """CSR CIGAR columns -> packed CIGAR columns (xm_cigar_pack; host only, no device): a byte of op count per record,
    the op position of every 256-record tile, and the op array (the input array itself unless a record has 255 ops or
    more and gets its trailer word).  -> (cig_cnt uint8[n], cig_tile uint32[tiles + 1], cig_oplen uint32[])"""
    off = _as(cig_off, np.uint32)
    ops = _as(cig_oplen, np.uint32)
    n = off.shape[0] - 1
    assert n >= 0 and ops.shape[0] >= int(off[-1])
    cnt = np.empty(max(n, 1), dtype=np.uint8)
    tile = np.empty(cigar_tiles(n) + 1, dtype=np.uint32)
    n_packed = ctypes.c_uint64(0)
    L = lib()
    rc = L.xm_cigar_pack(n, _np_ptr(off), _np_ptr(ops), _np_ptr(cnt), _np_ptr(tile), None, 0, ctypes.byref(n_packed))
    if rc == XM_OK and n_packed.value != int(off[-1]):
        packed = np.empty(n_packed.value, dtype=np.uint32)
        rc = L.xm_cigar_pack(n, _np_ptr(off), _np_ptr(ops), _np_ptr(cnt), _np_ptr(tile), _np_ptr(packed), packed.shape[0],
                             ctypes.byref(n_packed))
        ops = packed
    if rc != XM_OK:
        raise _ERRORS.get(rc, RuntimeError)("xm_cigar_pack: %s" % L.xm_strerror(rc).decode())
    return cnt[:n], tile, ops[:n_packed.value]


def comm_unique_id():
    """A fresh RCCL unique id (bytes) -- call on rank 0 and hand it to every rank."""
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    rc = lib().xm_comm_unique_id(buf)
    if rc != XM_OK:
        raise _ERRORS.get(rc, RuntimeError)("xm_comm_unique_id: %s [%s]" % (lib().xm_strerror(rc).decode(),
                                                                              lib().xm_last_hip_error(None).decode()))
    return buf.raw


def _one_call_per_context(method):
    """include/xenomapper_hip.h allows ONE call in flight per context (the compaction workspace -- granule counts, part totals,
    count replicas, the stream it was last used on -- belongs to the context and has no lock of its own in C).  The file path
    runs two threads on the process-wide context (the helper thread's fused pass behind the GPU front end of window k + 1 beside
    the main thread settling window k, which classifies again when the window had exception records), so every Python entry
    point that launches on the workspace takes the context's lock for the length of the C call."""
    @functools.wraps(method)
    def locked(self, *args, **kwargs):
        ctx = self if isinstance(self, Context) else self.ctx
        with ctx.call_lock:
            return method(self, *args, **kwargs)
    return locked


class Context(object):
    """One classifier context on one GPU (xm_ctx)."""

    def __init__(self, device=0):
        self._L = lib()
        h = ctypes.c_void_p()
        rc = self._L.xm_ctx_create(int(device), ctypes.byref(h))
        if rc != XM_OK:
            raise _ERRORS.get(rc, RuntimeError)(
                "xm_ctx_create(device=%d): %s [%s]" % (device, self._L.xm_strerror(rc).decode(),
                                                       self._L.xm_last_hip_error(None).decode()))
        self._h = h
        self.device = int(device)
        self._strippers = weakref.WeakSet()       # closed with the context: xm_strip_destroy hands its streams back to it
        self.call_lock = threading.RLock()        # one workspace call at a time (_one_call_per_context)
        self._registered = {}                     # address -> array page-locked through host_register

    def close(self):
        if getattr(self, "_h", None):
            for st in list(getattr(self, "_strippers", ())):
                st.close()
            for array in list(getattr(self, "_registered", {}).values()):
                try:
                    self.host_unregister(array)
                except Exception:                                    # noqa: BLE001 -- closing goes on
                    pass
            self._L.xm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc == XM_OK:
            return
        msg = "%s: %s" % (what, self._L.xm_strerror(rc).decode())
        detail = self._L.xm_last_hip_error(self._h).decode()
        if detail and rc in (-3, -4, -6):
            msg += " [" + detail + "]"
        raise _ERRORS.get(rc, RuntimeError)(msg)

    def host_register(self, array):
        """Page-lock a NumPy array that will be handed to the host-buffer calls repeatedly (direct DMA instead of the
        runtime's staging copies).  Call host_unregister(array) before the array is freed; the context keeps the array alive
        until then, and close() unregisters what is left -- pages of memory that NumPy has given back to the allocator must
        not stay locked under whoever gets the address next.  `with ctx.registered(a, b, ...)` does both ends."""
        rc = self._L.xm_host_register(self._h, _np_ptr(array), array.nbytes)
        self._check(rc, "xm_host_register")
        self._registered[array.ctypes.data] = array

    def host_unregister(self, array):
        rc = self._L.xm_host_unregister(self._h, _np_ptr(array))
        self._check(rc, "xm_host_unregister")
        self._registered.pop(array.ctypes.data, None)

    @contextlib.contextmanager
    def registered(self, *arrays):
        """Page-lock the arrays for the length of the block; whatever was locked when the block ends (or when locking one of
        them fails half way) is unlocked again."""
        done = []
        try:
            for a in arrays:
                self.host_register(a)
                done.append(a)
            yield
        finally:
            for a in done:
                self.host_unregister(a)

    def device_info(self):
        n_cu = ctypes.c_int()
        buf = ctypes.create_string_buffer(128)
        self._check(self._L.xm_ctx_device_info(self._h, ctypes.byref(n_cu), buf, 128), "xm_ctx_device_info")
        return {"n_cu": n_cu.value, "name": buf.value.decode()}

    # ---- host-buffer entry points (NumPy) --------------------------------------------------
    @_one_call_per_context
    def classify(self, mode, as1, xs1, as2, xs2, unit_bits, min_score_floor):
        cols = [_as(c, np.int32) for c in (as1, xs1, as2, xs2)]
        n = cols[0].shape[0]
        bits = _as(unit_bits, np.uint64)
        assert bits.shape[0] >= (n + 63) // 64 and all(c.shape[0] == n for c in cols)
        code = np.empty(n, dtype=np.uint8)
        counts = np.zeros(64, dtype=np.uint64)
        rc = self._L.xm_classify(self._h, mode, n, *[_np_ptr(c) for c in cols], _np_ptr(bits),
                                 int(min_score_floor), _np_ptr(code), _np_ptr(counts))
        self._check(rc, "xm_classify")
        return code, counts

    @_one_call_per_context
    def classify_f64(self, mode, as1, xs1, as2, xs2, unit_bits, min_score):
        cols = [_as(c, np.float64) for c in (as1, xs1, as2, xs2)]
        n = cols[0].shape[0]
        bits = _as(unit_bits, np.uint64)
        assert bits.shape[0] >= (n + 63) // 64 and all(c.shape[0] == n for c in cols)
        code = np.empty(n, dtype=np.uint8)
        counts = np.zeros(64, dtype=np.uint64)
        rc = self._L.xm_classify_f64(self._h, mode, n, *[_np_ptr(c) for c in cols], _np_ptr(bits),
                                     float(min_score), _np_ptr(code), _np_ptr(counts))
        self._check(rc, "xm_classify_f64")
        return code, counts

    def cigar_scores(self, nm, cig_off, cig_oplen):
        nm = _as(nm, np.int32)
        off = _as(cig_off, np.uint32)
        ops = _as(cig_oplen, np.uint32)
        n = nm.shape[0]
        assert off.shape[0] == n + 1 and ops.shape[0] >= (int(off[-1]) if n else 0)
        if ops.shape[0] == 0:
            ops = np.zeros(1, dtype=np.uint32)
        out = np.empty(n, dtype=np.int32)
        rc = self._L.xm_cigar_scores(self._h, n, _np_ptr(nm), _np_ptr(off), _np_ptr(ops), _np_ptr(out))
        self._check(rc, "xm_cigar_scores")
        return out

    @_one_call_per_context
    def classify_cigar(self, mode, nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits, min_score_floor):
        """The --cigar_scores path: AS from NM + CIGAR inside the classify kernel."""
        a = [_as(nm1, np.int32), _as(off1, np.uint32), _as(ops1, np.uint32), _as(xs1, np.int32),
             _as(nm2, np.int32), _as(off2, np.uint32), _as(ops2, np.uint32), _as(xs2, np.int32)]
        n = a[0].shape[0]
        for k in (2, 6):
            if a[k].shape[0] == 0:
                a[k] = np.zeros(1, dtype=np.uint32)
        assert a[1].shape[0] == n + 1 and a[5].shape[0] == n + 1
        bits = _as(unit_bits, np.uint64)
        code = np.empty(n, dtype=np.uint8)
        counts = np.zeros(64, dtype=np.uint64)
        rc = self._L.xm_classify_cigar(self._h, mode, n, *[_np_ptr(x) for x in a], _np_ptr(bits),
                                       int(min_score_floor), _np_ptr(code), _np_ptr(counts))
        self._check(rc, "xm_classify_cigar")
        return code, counts

    @_one_call_per_context
    def compact(self, mode, code):
        code = _as(code, np.uint8)
        n = code.shape[0]
        idx = np.empty(max(n, 1), dtype=np.uint32)
        off = np.zeros(8, dtype=np.uint64)
        counts = np.zeros(64, dtype=np.uint64)
        rc = self._L.xm_compact(self._h, mode, n, _np_ptr(code), _np_ptr(idx), _np_ptr(off), _np_ptr(counts))
        self._check(rc, "xm_compact")
        return idx[:int(off[7])], off, counts

    @_one_call_per_context
    def classify_compact(self, mode, as1, xs1, as2, xs2, unit_bits, min_score, want_code=True, idx_out=None):
        """One fused pass: classify + count + stable split.  Columns int32 (min_score = int floor) or float64
        (min_score = float).  -> (code or None, idx, bin_offsets[8], counts[64]).  idx_out: a uint32 array of at least
        n elements to receive the lists (e.g. one that lives across calls and has been host_register()ed)."""
        f64 = np.asarray(as1).dtype == np.float64
        cols = [_as(c, np.float64 if f64 else np.int32) for c in (as1, xs1, as2, xs2)]
        n = cols[0].shape[0]
        bits = _as(unit_bits, np.uint64)
        assert bits.shape[0] >= (n + 63) // 64 and all(c.shape[0] == n for c in cols)
        code = np.empty(n, dtype=np.uint8) if want_code else None
        if idx_out is not None:
            assert idx_out.dtype == np.uint32 and idx_out.flags.c_contiguous and idx_out.shape[0] >= max(n, 1)
            idx = idx_out
        else:
            idx = np.empty(max(n, 1), dtype=np.uint32)
        off = np.zeros(8, dtype=np.uint64)
        counts = np.zeros(64, dtype=np.uint64)
        fn = self._L.xm_classify_compact_f64 if f64 else self._L.xm_classify_compact
        rc = fn(self._h, mode, n, *[_np_ptr(c) for c in cols], _np_ptr(bits),
                float(min_score) if f64 else int(min_score), _np_ptr(code) if want_code else None,
                _np_ptr(idx), _np_ptr(off), _np_ptr(counts))
        self._check(rc, "xm_classify_compact")
        return code, idx[:int(off[7])], off, counts

    @_one_call_per_context
    def classify_compact_cigar(self, mode, nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits, min_score_floor,
                               want_code=True):
        a = [_as(nm1, np.int32), _as(off1, np.uint32), _as(ops1, np.uint32), _as(xs1, np.int32),
             _as(nm2, np.int32), _as(off2, np.uint32), _as(ops2, np.uint32), _as(xs2, np.int32)]
        n = a[0].shape[0]
        for k in (2, 6):
            if a[k].shape[0] == 0:
                a[k] = np.zeros(1, dtype=np.uint32)
        assert a[1].shape[0] == n + 1 and a[5].shape[0] == n + 1
        bits = _as(unit_bits, np.uint64)
        code = np.empty(n, dtype=np.uint8) if want_code else None
        idx = np.empty(max(n, 1), dtype=np.uint32)
        off = np.zeros(8, dtype=np.uint64)
        counts = np.zeros(64, dtype=np.uint64)
        rc = self._L.xm_classify_compact_cigar(self._h, mode, n, *[_np_ptr(x) for x in a], _np_ptr(bits),
                                               int(min_score_floor), _np_ptr(code) if want_code else None,
                                               _np_ptr(idx), _np_ptr(off), _np_ptr(counts))
        self._check(rc, "xm_classify_compact_cigar")
        return code, idx[:int(off[7])], off, counts

    @_one_call_per_context
    def classify_place(self, mode, as1, xs1, as2, xs2, unit_bits, min_score, want_code=True, capacity=None):
        """One main loop, six lists out (xm_classify_place / _f64: SURVEY 8b (4)).  -> (code or None, lists, n_out[8],
        counts[64]); lists = six uint32 arrays (seven for float64 columns: the last holds the units with state 6)."""
        f64 = np.asarray(as1).dtype == np.float64
        cols = [_as(c, np.float64 if f64 else np.int32) for c in (as1, xs1, as2, xs2)]
        n = cols[0].shape[0]
        bits = _as(unit_bits, np.uint64)
        assert bits.shape[0] >= (n + 63) // 64 and all(c.shape[0] == n for c in cols)
        cap = n if capacity is None else int(capacity)
        code = np.empty(n, dtype=np.uint8) if want_code else None
        lists = [np.empty(max(cap, 1), dtype=np.uint32) for _ in range(7 if f64 else 6)]
        arr = (ctypes.c_void_p * 6)(*[l.ctypes.data for l in lists[:6]])
        n_out = np.zeros(8, dtype=np.uint64)
        counts = np.zeros(64, dtype=np.uint64)
        if f64:
            rc = self._L.xm_classify_place_f64(self._h, mode, n, *[_np_ptr(c) for c in cols], _np_ptr(bits), float(min_score),
                                               _np_ptr(code) if want_code else None, arr, _np_ptr(lists[6]), cap,
                                               _np_ptr(n_out), _np_ptr(counts))
        else:
            rc = self._L.xm_classify_place(self._h, mode, n, *[_np_ptr(c) for c in cols], _np_ptr(bits), int(min_score),
                                           _np_ptr(code) if want_code else None, arr, cap, _np_ptr(n_out), _np_ptr(counts))
        self._check(rc, "xm_classify_place")
        return code, [l[:min(int(n_out[b]), cap)] for b, l in enumerate(lists)], n_out, counts

    def mate_correlate(self, track, density):
        """Paired-end mappability of one chromosome track (float64 in, float64 out)."""
        track = _as(track, np.float64)
        density = _as(density, np.float64)
        out = np.empty(track.shape[0], dtype=np.float64)
        d = density if density.shape[0] else np.zeros(1)
        t = track if track.shape[0] else np.zeros(1)
        rc = self._L.xm_mate_correlate(self._h, track.shape[0], _np_ptr(t), density.shape[0], _np_ptr(d),
                                       _np_ptr(out if out.shape[0] else np.zeros(1)))
        self._check(rc, "xm_mate_correlate")
        return out

    def mate_correlate_dev(self, track, density, out, stream=None):
        rc = self._L.xm_mate_correlate_dev(self._h, self._stream_handle(stream), track.numel(),
                                           ctypes.c_void_p(track.data_ptr()), density.numel(),
                                           ctypes.c_void_p(density.data_ptr()), ctypes.c_void_p(out.data_ptr()))
        self._check(rc, "xm_mate_correlate_dev")

    # ---- device-resident entry points (torch tensors on this context's GPU) --------------
    @staticmethod
    def _stream_handle(stream):
        if stream is None:
            import torch
            return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        return ctypes.c_void_p(getattr(stream, "cuda_stream", stream))

    @_one_call_per_context
    def classify_dev(self, mode, as1, xs1, as2, xs2, unit_bits, min_score, code_out, stream=None):
        """Columns int32 (min_score = int floor) or float64 (min_score = float).  Asynchronous."""
        n = as1.numel()
        st = self._stream_handle(stream)
        ptrs = [ctypes.c_void_p(t.data_ptr()) for t in (as1, xs1, as2, xs2, unit_bits)]
        if as1.element_size() == 4:
            rc = self._L.xm_classify_dev(self._h, st, mode, n, *ptrs, int(min_score),
                                         ctypes.c_void_p(code_out.data_ptr()))
        else:
            rc = self._L.xm_classify_f64_dev(self._h, st, mode, n, *ptrs, float(min_score),
                                             ctypes.c_void_p(code_out.data_ptr()))
        self._check(rc, "xm_classify_dev")

    @_one_call_per_context
    def classify_cigar_dev(self, mode, nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits, min_score_floor,
                           code_out, range_flag=None, stream=None):
        ptrs = [ctypes.c_void_p(t.data_ptr()) for t in (nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits)]
        rc = self._L.xm_classify_cigar_dev(
            self._h, self._stream_handle(stream), mode, nm1.numel(), *ptrs, int(min_score_floor),
            ctypes.c_void_p(code_out.data_ptr()),
            ctypes.c_void_p(range_flag.data_ptr()) if range_flag is not None else None)
        self._check(rc, "xm_classify_cigar_dev")

    def cigar_scores_dev(self, nm, cig_off, cig_oplen, as_out, range_flag=None, stream=None):
        rc = self._L.xm_cigar_scores_dev(
            self._h, self._stream_handle(stream), nm.numel(), ctypes.c_void_p(nm.data_ptr()),
            ctypes.c_void_p(cig_off.data_ptr()), ctypes.c_void_p(cig_oplen.data_ptr()),
            ctypes.c_void_p(as_out.data_ptr()),
            ctypes.c_void_p(range_flag.data_ptr()) if range_flag is not None else None)
        self._check(rc, "xm_cigar_scores_dev")

    @_one_call_per_context
    def compact_dev(self, mode, code, idx_out, bin_offsets, counts, stream=None):
        rc = self._L.xm_compact_dev(
            self._h, self._stream_handle(stream), mode, code.numel(), ctypes.c_void_p(code.data_ptr()),
            ctypes.c_void_p(idx_out.data_ptr()), ctypes.c_void_p(bin_offsets.data_ptr()),
            ctypes.c_void_p(counts.data_ptr()))
        self._check(rc, "xm_compact_dev")

    @_one_call_per_context
    def classify_compact_dev(self, mode, as1, xs1, as2, xs2, unit_bits, min_score, code_out, idx_out, bin_offsets,
                             counts, bins4=None, stream=None):
        """The fused main loop on device-resident columns (int32 or float64): classify + count in one kernel, then
        scan + scatter.  `code_out` (category bytes) and/or `bins4` (compact category stream, bins4_bytes(n) bytes):
        at least one; with bins4 the scatter reads the compact stream.  Asynchronous."""
        n = as1.numel()
        st = self._stream_handle(stream)
        ptrs = [ctypes.c_void_p(t.data_ptr()) for t in (as1, xs1, as2, xs2, unit_bits)]
        outs = [ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in (code_out, bins4, idx_out, bin_offsets, counts)]
        if as1.element_size() == 4:
            rc = self._L.xm_classify_compact_dev(self._h, st, mode, n, *ptrs, int(min_score), *outs)
        else:
            rc = self._L.xm_classify_compact_f64_dev(self._h, st, mode, n, *ptrs, float(min_score), *outs)
        self._check(rc, "xm_classify_compact_dev")

    @_one_call_per_context
    def classify_compact_cigar_dev(self, mode, nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits, min_score_floor,
                                   code_out, idx_out, bin_offsets, counts, range_flag=None, stream=None):
        ptrs = [ctypes.c_void_p(t.data_ptr()) for t in (nm1, off1, ops1, xs1, nm2, off2, ops2, xs2, unit_bits)]
        rc = self._L.xm_classify_compact_cigar_dev(
            self._h, self._stream_handle(stream), mode, nm1.numel(), *ptrs, int(min_score_floor),
            ctypes.c_void_p(code_out.data_ptr()),
            ctypes.c_void_p(range_flag.data_ptr()) if range_flag is not None else None,
            ctypes.c_void_p(idx_out.data_ptr()), ctypes.c_void_p(bin_offsets.data_ptr()),
            ctypes.c_void_p(counts.data_ptr()))
        self._check(rc, "xm_classify_compact_cigar_dev")

    @_one_call_per_context
    def classify_compact_cigar_packed_dev(self, mode, nm1, cnt1, tile1, ops1, xs1, nm2, cnt2, tile2, ops2, xs2, unit_bits,
                                          min_score_floor, code_out, idx_out, bin_offsets, counts, bins4=None,
                                          range_flag=None, stream=None):
        """One whole --cigar_scores loop on device-resident PACKED CIGAR columns (cigar_pack): the classify kernel
        synthesises AS, counts, and writes category bytes (code_out) and/or the compact stream (bins4)."""
        ptrs = [ctypes.c_void_p(t.data_ptr()) for t in (nm1, cnt1, tile1, ops1, xs1, nm2, cnt2, tile2, ops2, xs2, unit_bits)]
        opt = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None     # noqa: E731
        rc = self._L.xm_classify_compact_cigar_packed_dev(
            self._h, self._stream_handle(stream), mode, nm1.numel(), *ptrs, int(min_score_floor), opt(code_out), opt(bins4),
            opt(range_flag), opt(idx_out), opt(bin_offsets), opt(counts))
        self._check(rc, "xm_classify_compact_cigar_packed_dev")

    @_one_call_per_context
    def classify_place_dev(self, mode, as1, xs1, as2, xs2, unit_bits, min_score, lists, n_out, counts, code_out=None,
                           bins4=None, list_state6=None, capacity=None, stream=None):
        """The fused main loop with the six-list output contract: `lists` = six uint32 device tensors,
        one per output bin; n_out: 8 x int64 device tensor (six list lengths, state-6 units, all units); counts: 64.
        code_out and/or bins4 as for classify_compact_dev.  Columns int32 or float64.  Asynchronous."""
        n = as1.numel()
        st = self._stream_handle(stream)
        ptrs = [ctypes.c_void_p(t.data_ptr()) for t in (as1, xs1, as2, xs2, unit_bits)]
        assert len(lists) == 6
        cap = min(t.numel() for t in lists) if capacity is None else int(capacity)
        arr = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in lists])
        opt = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None     # noqa: E731
        if as1.element_size() == 4:
            rc = self._L.xm_classify_place_dev(self._h, st, mode, n, *ptrs, int(min_score), opt(code_out), opt(bins4), arr, cap,
                                               opt(n_out), opt(counts))
        else:
            rc = self._L.xm_classify_place_f64_dev(self._h, st, mode, n, *ptrs, float(min_score), opt(code_out), opt(bins4), arr,
                                                   opt(list_state6), cap, opt(n_out), opt(counts))
        self._check(rc, "xm_classify_place_dev")

    @_one_call_per_context
    def classify_place_cigar_packed_dev(self, mode, nm1, cnt1, tile1, ops1, xs1, nm2, cnt2, tile2, ops2, xs2, unit_bits,
                                        min_score_floor, lists, n_out, counts, code_out=None, bins4=None, range_flag=None,
                                        capacity=None, stream=None):
        """--cigar_scores loop on packed CIGAR columns with the six-list output contract."""
        ptrs = [ctypes.c_void_p(t.data_ptr()) for t in (nm1, cnt1, tile1, ops1, xs1, nm2, cnt2, tile2, ops2, xs2, unit_bits)]
        opt = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None     # noqa: E731
        cap = min(t.numel() for t in lists) if capacity is None else int(capacity)
        arr = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in lists])
        rc = self._L.xm_classify_place_cigar_packed_dev(
            self._h, self._stream_handle(stream), mode, nm1.numel(), *ptrs, int(min_score_floor), opt(code_out), opt(bins4),
            opt(range_flag), arr, cap, opt(n_out), opt(counts))
        self._check(rc, "xm_classify_place_cigar_packed_dev")

    @_one_call_per_context
    def classify_runs_dev(self, mode, as1, xs1, as2, xs2, unit_bits, min_score, runs16, gran_counts, n_out, counts, stream=None):
        """The fused main loop with SEGMENTED bin lists, one launch (xm_classify_runs*_dev): runs16 = int16 device tensor of
        runs_granules(n) * 2048 entries, gran_counts = int16 device tensor of runs_granules(n) * 8, n_out 8 x int64,
        counts 64 x int64.  Columns int32 or float64.  Asynchronous."""
        n = as1.numel()
        st = self._stream_handle(stream)
        ptrs = [ctypes.c_void_p(t.data_ptr()) for t in (as1, xs1, as2, xs2, unit_bits)]
        outs = [ctypes.c_void_p(t.data_ptr()) for t in (runs16, gran_counts, n_out, counts)]
        assert runs16.numel() >= runs_granules(n) * RUNS_GRAN and gran_counts.numel() >= runs_granules(n) * 8
        if as1.element_size() == 4:
            rc = self._L.xm_classify_runs_dev(self._h, st, mode, n, *ptrs, int(min_score), *outs)
        else:
            rc = self._L.xm_classify_runs_f64_dev(self._h, st, mode, n, *ptrs, float(min_score), *outs)
        self._check(rc, "xm_classify_runs_dev")

    def bgzf_inflate_dev(self, comp, blocks, out, status, work, stream=None, walk=None):
        """BGZF blocks inflated on the GPU (xm_bgzf_inflate_dev): comp = uint8 device tensor of the compressed image (+ BGZF_COMP_PAD
        bytes behind the last block), blocks = device tensor holding a BGZF_BLOCK array (as uint8 / int64 bytes), out = uint8
        device tensor, status = int32 tensor with one entry per block, work = int32 tensor with one entry.  walk: device tensor
        holding a BGZF_WALK array, one entry per block (xm_bgzf_inflate_walk_dev: the records of every block found and read by
        the chain that inflated it).  Asynchronous."""
        n = blocks.numel() * blocks.element_size() // 24
        if walk is not None:
            assert walk.numel() * walk.element_size() == n * BGZF_WALK.itemsize
        rc = self._L.xm_bgzf_inflate_walk_dev(self._h, self._stream_handle(stream), ctypes.c_void_p(comp.data_ptr()),
                                              ctypes.c_void_p(blocks.data_ptr()), n, ctypes.c_void_p(out.data_ptr()),
                                              ctypes.c_void_p(status.data_ptr()), ctypes.c_void_p(work.data_ptr()),
                                              ctypes.c_void_p(walk.data_ptr()) if walk is not None else None)
        self._check(rc, "xm_bgzf_inflate_walk_dev")

    def bgzf_crc32_dev(self, out, blocks, crc_out, stream=None):
        n = blocks.numel() * blocks.element_size() // 24
        rc = self._L.xm_bgzf_crc32_dev(self._h, self._stream_handle(stream), ctypes.c_void_p(out.data_ptr()),
                                       ctypes.c_void_p(blocks.data_ptr()), n, ctypes.c_void_p(crc_out.data_ptr()))
        self._check(rc, "xm_bgzf_crc32_dev")

    @_one_call_per_context
    def workspace_release(self, stream):
        """Call before destroying a stream the compaction calls were issued on while this context lives on."""
        self._check(self._L.xm_workspace_release(self._h, self._stream_handle(stream)), "xm_workspace_release")

    @_one_call_per_context
    def workspace_is_clean(self):
        """Synchronises; True when the counting workspace is in its between-calls state (all zero)."""
        flag = ctypes.c_int(0)
        self._check(self._L.xm_workspace_is_clean(self._h, ctypes.byref(flag)), "xm_workspace_is_clean")
        return bool(flag.value)

    def stream_probe_dev(self, c0, c1, c2, c3, out, stream=None):
        """The classify kernel's memory shape without arithmetic (the box's streaming ceiling for it).  Asynchronous."""
        rc = self._L.xm_stream_probe_dev(self._h, self._stream_handle(stream), c0.numel(), *[ctypes.c_void_p(t.data_ptr())
                                                                                       for t in (c0, c1, c2, c3, out)])
        self._check(rc, "xm_stream_probe_dev")

    # ---- the count all-reduce (RCCL inside the library) -----------------------------------
    def comm_init(self, n_ranks, rank, unique_id):
        """Join the communicator of the category_counts all-reduce; `unique_id` = comm_unique_id() of rank 0."""
        assert len(unique_id) == UNIQUE_ID_BYTES
        self._check(self._L.xm_comm_init(self._h, int(n_ranks), int(rank), ctypes.c_char_p(bytes(unique_id))),
                    "xm_comm_init")

    def comm_destroy(self):
        self._check(self._L.xm_comm_destroy(self._h), "xm_comm_destroy")

    def comm_size(self):
        return int(self._L.xm_comm_size(self._h))

    def allreduce_counts(self, counts, stream=None):
        """In-place sum over all ranks of a 64-element int64/uint64 device tensor.  Asynchronous."""
        assert counts.numel() == 64 and counts.element_size() == 8
        self._check(self._L.xm_allreduce_counts(self._h, self._stream_handle(stream), ctypes.c_void_p(counts.data_ptr())),
                    "xm_allreduce_counts")

    # ---- timing --------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._check(self._L.xm_timing_enable(self._h, 1 if on else 0), "xm_timing_enable")

    def timing_select(self, kernels=None):
        """Time only the named kernels (None = all)."""
        mask = 0xFFFFFFFF if kernels is None else sum(1 << KERNELS.index(k) for k in kernels)
        self._check(self._L.xm_timing_select(self._h, mask), "xm_timing_select")

    def timing_reset(self):
        self._check(self._L.xm_timing_reset(self._h), "xm_timing_reset")

    def timing_read(self):
        ms = (ctypes.c_double * len(KERNELS))()
        launches = (ctypes.c_uint64 * len(KERNELS))()
        self._check(self._L.xm_timing_read(self._h, ms, launches), "xm_timing_read")
        return {k: {"ms": ms[i], "launches": int(launches[i])} for i, k in enumerate(KERNELS)}


# ---- include/xenomapper_bgzf.h: BGZF blocks inflated on the GPU -----------------------------------------------------
BGZF_BLOCK = np.dtype([("cdata_off", np.uint64), ("out_off", np.uint64), ("cdata_len", np.uint32), ("isize", np.uint32)])
assert BGZF_BLOCK.itemsize == 24
BGZF_WALK = np.dtype([("raw_base", np.uint64), ("start", np.uint32), ("end", np.uint32), ("n_raw", np.uint32), ("slot_cap", np.uint32),
                      ("count", np.uint64), ("exit_at", np.uint64), ("slots", np.uint64), ("name_off", np.uint64), ("name_len", np.uint64),
                      ("a", np.uint64), ("x", np.uint64), ("flag", np.uint64), ("n_cigar", np.uint64), ("cig_at", np.uint64),
                      ("tags", np.uint32), ("reserved", np.uint32)])
assert BGZF_WALK.itemsize == 112                                    # xm_bgzf_walk (device addresses)
BGZF_TAGS_AS_XS = ord("X") | ord("A") << 8 | ord("S") << 16
BGZF_TAGS_AS_ZS = ord("Z") | ord("A") << 8 | ord("S") << 16
BGZF_TAGS_NM_XS = ord("X") | ord("N") << 8 | ord("M") << 16 | 1 << 24
BGZF_COMP_PAD = 1024


def bgzf_index(data, start=0, max_out=1 << 62, cap=None, prefix=False):
    """Walk the BGZF member headers of a file image (uint8 array) from byte `start` (xm_bgzf_index; host only):
    -> (blocks as a BGZF_BLOCK array, crc uint32 array, next byte, inflated bytes).  prefix: the array may end inside a member
    (bytes read ahead of a file): the walk stops in front of it (xm_bgzf_index_prefix)."""
    data = np.ascontiguousarray(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    n_max = int(cap) if cap is not None else max(16, (data.shape[0] - int(start)) // 28 + 16)    # an empty member is 28 bytes
    blocks = np.zeros(n_max, dtype=BGZF_BLOCK)
    crc = np.zeros(n_max, dtype=np.uint32)
    n, nxt, total = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    fn = lib().xm_bgzf_index_prefix if prefix else lib().xm_bgzf_index
    rc = fn(ctypes.c_void_p(data.ctypes.data), data.shape[0], int(start), int(max_out), _np_ptr(blocks), _np_ptr(crc),
            n_max, ctypes.byref(n), ctypes.byref(nxt), ctypes.byref(total))
    if rc != XM_OK:
        raise ValueError("xm_bgzf_index: not a BGZF image (or a truncated one) at byte %d" % int(start))
    return blocks[:n.value], crc[:n.value], int(nxt.value), int(total.value)


def bgzf_strerror(status):
    return lib().xm_bgzf_strerror(int(status)).decode()


# ---- include/xenomapper_strip.h: the SAM column stripper on the GPU --------------------------------------------
LINE_NORMAL, LINE_BLANK, LINE_EX_A, LINE_EX_X, LINE_MISMATCH = 0x01, 0x02, 0x1C, 0x60, 0x80
LINE_EX_A_SHIFT, LINE_EX_X_SHIFT = 2, 5
STRIP_SLOTS = 2
STRIP_MAX_WINDOW = 0xFFFF0000


class _BamDevBins(ctypes.Structure):          # xm_bamdev_bins and xm_strip_bins
    _fields_ = [("text", ctypes.c_void_p), ("bin_off", ctypes.c_uint64 * 8), ("status", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class _StripBlock(ctypes.Structure):
    _fields_ = [("n_records", ctypes.c_uint64), ("consumed1", ctypes.c_uint64), ("consumed2", ctypes.c_uint64),
                ("consumed_lines1", ctypes.c_uint64), ("consumed_lines2", ctypes.c_uint64),
                ("ended", ctypes.c_int32), ("starved", ctypes.c_int32), ("mismatch_at", ctypes.c_int64),
                ("non_ascii", ctypes.c_int32), ("overflow", ctypes.c_int32), ("n_exceptions", ctypes.c_uint64),
                ("n_lines1", ctypes.c_uint64), ("n_lines2", ctypes.c_uint64),
                ("line_off1", ctypes.c_void_p), ("line_off2", ctypes.c_void_p), ("line_len1", ctypes.c_void_p),
                ("line_len2", ctypes.c_void_p), ("norm_len1", ctypes.c_void_p), ("norm_len2", ctypes.c_void_p),
                ("line_flags1", ctypes.c_void_p), ("line_flags2", ctypes.c_void_p),
                ("ms_upload", ctypes.c_float), ("ms_kernels", ctypes.c_float)]


def _host_view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class StrippedBlock(object):
    """One pair of windows stripped on the GPU (xm_strip_block): the walk's outcome and the line tables as NumPy views of
    the stripper's page-locked arrays (valid until the slot is run again); the score columns stay on the device."""

    def __init__(self, stripper, slot, raw):
        self.stripper, self.slot = stripper, slot
        n = self.n = int(raw.n_records)
        self.consumed = (int(raw.consumed1), int(raw.consumed2))
        self.consumed_lines = (int(raw.consumed_lines1), int(raw.consumed_lines2))
        self.ended, self.starved, self.mismatch_at = bool(raw.ended), bool(raw.starved), int(raw.mismatch_at)
        self.non_ascii = bool(raw.non_ascii)
        self.overflow = bool(raw.overflow)
        self.n_exceptions = int(raw.n_exceptions)
        self.n_lines = (int(raw.n_lines1), int(raw.n_lines2))
        self.tables = (raw.line_off1, raw.line_len1, raw.norm_len1, raw.line_flags1,
                       raw.line_off2, raw.line_len2, raw.norm_len2, raw.line_flags2)
        self.line_off = [_host_view(raw.line_off1, n, np.uint32), _host_view(raw.line_off2, n, np.uint32)]
        self.line_len = [_host_view(raw.line_len1, n, np.uint32), _host_view(raw.line_len2, n, np.uint32)]
        self.norm_len = [_host_view(raw.norm_len1, n, np.uint32), _host_view(raw.norm_len2, n, np.uint32)]
        self.line_flags = [_host_view(raw.line_flags1, n, np.uint8), _host_view(raw.line_flags2, n, np.uint8)]
        self.ms_upload, self.ms_kernels = float(raw.ms_upload), float(raw.ms_kernels)
        self.csr = None
        self._host_cols = None

    def _download(self):
        if self._host_cols is None:
            self._host_cols = self.stripper.columns(self.slot, self.n)
        return self._host_cols

    @property
    def cols(self):
        """The four score columns on the host (downloaded on first use: only blocks with flagged records need them)."""
        return self._download()[:4]

    @property
    def unit_bits(self):
        return self._download()[4]

    @property
    def exc(self):
        """(record, column, kind) of every flagged value, ordered by (record, column) as xmh_block lists them."""
        out = []
        if self.n_exceptions:
            flags = self.line_flags
            hit = np.flatnonzero((flags[0] | flags[1]) & (LINE_EX_A | LINE_EX_X))
            for k in hit.tolist():
                for f in (0, 1):
                    v = int(flags[f][k])
                    if v & LINE_EX_A:
                        out.append((k, 2 * f, (v & LINE_EX_A) >> LINE_EX_A_SHIFT))
                    if v & LINE_EX_X:
                        out.append((k, 2 * f + 1, (v & LINE_EX_X) >> LINE_EX_X_SHIFT))
        return out


class Stripper(object):
    """xm_strip: SAM text in page-locked staging buffers -> score columns in HBM + line tables on the host."""

    def __init__(self, ctx):
        self._L = lib()
        self.ctx = ctx
        h = ctypes.c_void_p()
        rc = self._L.xm_strip_create(ctx._h, ctx.device, ctypes.byref(h))
        if rc != XM_OK:
            raise _ERRORS.get(rc, RuntimeError)("xm_strip_create: " + self._L.xm_strerror(rc).decode())
        self._h = h
        self._cap = [(0, 0)] * STRIP_SLOTS
        ctx._strippers.add(self)

    def close(self):
        if getattr(self, "_h", None):
            self._L.xm_strip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != XM_OK:
            detail = self._L.xm_strip_last_error(self._h).decode()
            raise _ERRORS.get(rc, RuntimeError)("%s: %s%s" % (what, self._L.xm_strerror(rc).decode(),
                                                               " [" + detail + "]" if detail and rc in (-3, -4) else ""))

    def reserve(self, slot, window_bytes, max_records):
        have = self._cap[slot]
        if window_bytes > have[0] or max_records > have[1]:
            want = (max(window_bytes, have[0]), max(max_records, have[1]))
            self._cap[slot] = (0, 0)             # a failed growth leaves the slot without buffers: never trust the old sizes
            self._check(self._L.xm_strip_reserve(self._h, slot, want[0], want[1]), "xm_strip_reserve")
            self._cap[slot] = want

    def fits(self, slot, window_bytes, max_records):
        """The slot's buffers hold that much already (reserve() would not touch them)."""
        return window_bytes <= self._cap[slot][0] and max_records <= self._cap[slot][1]

    def staging_address(self, slot, file):
        return self._L.xm_strip_staging(self._h, slot, file)

    def staging(self, slot, file):
        """The slot's page-locked text buffer of one file as a uint8 array."""
        return _host_view(self.staging_address(slot, file), self._cap[slot][0], np.uint8)

    def begin_behind(self, slot, file, room):
        """A new window of this file whose bytes [0, room) come later (xm_strip_begin_behind): uploads go on from `room`."""
        self._check(self._L.xm_strip_begin_behind(self._h, slot, file, int(room)), "xm_strip_begin_behind")

    def set_lead(self, slot, file, lead):
        """The first `lead` bytes of the window staged for the next run are no text (xm_strip_set_lead)."""
        self._check(self._L.xm_strip_set_lead(self._h, slot, file, int(lead)), "xm_strip_set_lead")

    def upload(self, slot, file, offset, n):
        """Start sending staged bytes [offset, offset + n) of one window to the device (in order, from 0)."""
        self._check(self._L.xm_strip_upload(self._h, slot, file, int(offset), int(n)), "xm_strip_upload")

    def run(self, slot, len1, eof1, len2, eof2, score_mode, paired, skip_repeated, keep_halo, max_records):
        raw = _StripBlock()
        rc = self._L.xm_strip_run(self._h, slot, int(len1), int(bool(eof1)), int(len2), int(bool(eof2)), int(score_mode),
                                  int(bool(paired)), int(bool(skip_repeated)), int(bool(keep_halo)), int(max_records),
                                  ctypes.byref(raw))
        self._check(rc, "xm_strip_run")
        return StrippedBlock(self, slot, raw)

    @_one_call_per_context
    def classify(self, slot, mode, n_records, min_score_floor):
        """The fused pass on the slot's device columns -> (code, idx, bin_offsets, counts); code / idx are views of
        page-locked arrays, valid until the next classify on the slot."""
        code, idx = ctypes.c_void_p(), ctypes.c_void_p()
        off = np.zeros(8, dtype=np.uint64)
        counts = np.zeros(64, dtype=np.uint64)
        rc = self._L.xm_strip_classify(self._h, slot, int(mode), int(n_records), int(min_score_floor), ctypes.byref(code),
                                       ctypes.byref(idx), _np_ptr(off), _np_ptr(counts))
        self._check(rc, "xm_strip_classify")
        return (_host_view(code.value, int(n_records), np.uint8), _host_view(idx.value, int(off[7]), np.uint32), off, counts)

    def fetch_bins(self, slot, n_records, paired, sink_mask):
        """After classify(): the six outputs themselves, gathered on the device (xm_strip_fetch_bins) -> (status, text, bin_off):
        bin b's text = text[bin_off[b]:bin_off[b + 1]] (a view of a page-locked buffer, valid after out_wait() and until the
        next fetch on the slot); status 2 / 3: this window is the host writer's (more text than the buffers hold / a wanted line
        that needs its white space re-joined)."""
        t = _BamDevBins()                                               # (xm_strip_bins has the same layout)
        rc = self._L.xm_strip_fetch_bins(self._h, int(slot), int(n_records), int(bool(paired)), int(sink_mask), ctypes.byref(t))
        self._check(rc, "xm_strip_fetch_bins")
        if t.status != 0:
            return int(t.status), None, None
        off = [int(v) for v in t.bin_off]
        text = _host_view(t.text, off[7], np.uint8) if off[7] else np.zeros(0, dtype=np.uint8)
        return 0, text, off

    def out_wait(self, slot):
        self._check(self._L.xm_strip_out_wait(self._h, int(slot)), "xm_strip_out_wait")

    def cigar_columns(self, slot, file, n_records):
        """After a CIGAR-mode run: (nm int32, cig_cnt uint8, cig_tile uint32, cig_ops uint32) of one file, on the host."""
        n = int(n_records)
        nm = np.empty(n, dtype=np.int32)
        cnt = np.empty(n, dtype=np.uint8)
        tile = np.zeros(cigar_tiles(n) + 1, dtype=np.uint32)
        n_ops = ctypes.c_uint64()
        self._check(self._L.xm_strip_cigar_columns(self._h, slot, file, n, _np_ptr(nm), _np_ptr(cnt), _np_ptr(tile), None, 0,
                                                   ctypes.byref(n_ops)), "xm_strip_cigar_columns")
        ops = np.zeros(max(int(n_ops.value), 1), dtype=np.uint32)
        self._check(self._L.xm_strip_cigar_columns(self._h, slot, file, n, None, None, None, _np_ptr(ops), ops.shape[0],
                                                   ctypes.byref(n_ops)), "xm_strip_cigar_columns")
        return nm, cnt, tile, ops[:int(n_ops.value)]

    def columns(self, slot, n_records):
        """-> [as1, xs1, as2, xs2 (int32), unit_bits (uint64)] copied to the host."""
        n = int(n_records)
        out = [np.empty(n, dtype=np.int32) for _ in range(4)] + [np.zeros((n + 63) // 64, dtype=np.uint64)]
        if n:
            self._check(self._L.xm_strip_columns(self._h, slot, n, *[_np_ptr(a) for a in out]), "xm_strip_columns")
        return out


# ---- include/xenomapper_bgzf.h: BAM records -> columns on the GPU (xm_bamdev_*) -----------------------------------------
class _BamDevInput(ctypes.Structure):
    _fields_ = [("comp_len", ctypes.c_uint64), ("blocks", ctypes.c_void_p), ("crc", ctypes.c_void_p), ("n_blocks", ctypes.c_uint64),
                ("carry_slot", ctypes.c_int32), ("carry_off", ctypes.c_uint64), ("carry_len", ctypes.c_uint64),
                ("eof", ctypes.c_int32), ("skip", ctypes.c_uint64), ("uploaded", ctypes.c_uint64)]


class _BamDevText(ctypes.Structure):
    _fields_ = [("raw1", ctypes.c_void_p), ("raw2", ctypes.c_void_p), ("off1", ctypes.c_void_p), ("off2", ctypes.c_void_p),
                ("bytes1", ctypes.c_uint64), ("bytes2", ctypes.c_uint64)]


class _BamDevLines(ctypes.Structure):
    _fields_ = [("text1", ctypes.c_void_p), ("text2", ctypes.c_void_p), ("line_off1", ctypes.c_void_p), ("line_off2", ctypes.c_void_p),
                ("line_len1", ctypes.c_void_p), ("line_len2", ctypes.c_void_p), ("bytes1", ctypes.c_uint64), ("bytes2", ctypes.c_uint64),
                ("status", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class _BamDevBlock(ctypes.Structure):
    _fields_ = [("n_records", ctypes.c_uint64), ("consumed1", ctypes.c_uint64), ("consumed2", ctypes.c_uint64),
                ("raw_len1", ctypes.c_uint64), ("raw_len2", ctypes.c_uint64), ("n_rec1", ctypes.c_uint64), ("n_rec2", ctypes.c_uint64),
                ("ended", ctypes.c_int32), ("starved", ctypes.c_int32), ("mismatch_at", ctypes.c_int64),
                ("bad_block", ctypes.c_int32), ("unaligned", ctypes.c_int32), ("weird", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("n_exceptions", ctypes.c_uint64),
                ("raw1", ctypes.c_void_p), ("raw2", ctypes.c_void_p), ("rec_off1", ctypes.c_void_p), ("rec_off2", ctypes.c_void_p),
                ("flags1", ctypes.c_void_p), ("flags2", ctypes.c_void_p), ("ms_inflate", ctypes.c_float), ("ms_kernels", ctypes.c_float)]


class BamDevBlock(object):
    """One pair of BAM windows inflated and stripped on the GPU (xm_bamdev_block).  The score columns stay on the device; the
    inflated bytes and the record tables are page-locked host memory of the slot (valid until the slot runs again).  The
    caller prints the records' SAM text (set_text) before handing the block to the writer; then it looks like a StrippedBlock."""

    def __init__(self, dev, slot, raw):
        self.stripper, self.slot = dev, slot                     # `stripper`: whoever classifies the slot's columns
        n = self.n = int(raw.n_records)
        self.consumed = (int(raw.consumed1), int(raw.consumed2))
        self.consumed_lines = (0, 0)
        self.raw_len = (int(raw.raw_len1), int(raw.raw_len2))
        self.n_rec = (int(raw.n_rec1), int(raw.n_rec2))
        self.ended, self.starved, self.mismatch_at = bool(raw.ended), bool(raw.starved), int(raw.mismatch_at)
        self.bad_block, self.unaligned, self.weird = int(raw.bad_block), bool(raw.unaligned), bool(raw.weird)
        self.n_exceptions = int(raw.n_exceptions)
        self.raw_addr = (raw.raw1, raw.raw2)
        self.rec_off_addr = (raw.rec_off1, raw.rec_off2)
        self.line_flags = [_host_view(raw.flags1, n, np.uint8), _host_view(raw.flags2, n, np.uint8)]
        self.ms_inflate, self.ms_kernels = float(raw.ms_inflate), float(raw.ms_kernels)
        self.non_ascii = self.overflow = False
        self.csr = None
        self._host_cols = None
        self.line_off = self.line_len = self.norm_len = self.tables = None
        self.packed = None                                       # fetch_wanted()'s result when only the wanted records came back
        self.finish = None                                       # set by the file path: prints the records' text when called
        self.classified = None                                   # (code, idx, bin_offsets, counts) when the fused pass has run already

    def set_text(self, line_off, line_len):
        """The line tables of the printed text (uint32 arrays per file): what the writer gathers from."""
        self.line_off, self.line_len, self.norm_len = line_off, line_len, line_len
        self.tables = (line_off[0].ctypes.data, line_len[0].ctypes.data, line_len[0].ctypes.data, self.line_flags[0].ctypes.data,
                       line_off[1].ctypes.data, line_len[1].ctypes.data, line_len[1].ctypes.data, self.line_flags[1].ctypes.data)

    def _download(self):
        if self._host_cols is None:
            self._host_cols = self.stripper.columns(self.slot, self.n)
        return self._host_cols

    @property
    def cols(self):
        return self._download()[:4]

    @property
    def unit_bits(self):
        return self._download()[4]

    exc = StrippedBlock.exc


class BamDev(object):
    """xm_bamdev: BGZF blocks of two BAM files -> score columns in HBM, inflated bytes + record tables on the host."""

    def __init__(self, ctx):
        self._L = lib()
        self.ctx = ctx
        h = ctypes.c_void_p()
        rc = self._L.xm_bamdev_create(ctx._h, ctx.device, ctypes.byref(h))
        if rc != XM_OK:
            raise _ERRORS.get(rc, RuntimeError)("xm_bamdev_create: " + self._L.xm_strerror(rc).decode())
        self._h = h
        self._cap = [(0, 0, 0, 0)] * 2
        ctx._strippers.add(self)

    def close(self):
        if getattr(self, "_h", None):
            self._L.xm_bamdev_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != XM_OK:
            detail = self._L.xm_bamdev_last_error(self._h).decode()
            raise _ERRORS.get(rc, RuntimeError)("%s: %s%s" % (what, self._L.xm_strerror(rc).decode(),
                                                               " [" + detail + "]" if detail and rc in (-3, -4) else ""))

    def reserve(self, slot, comp_bytes, raw_bytes, max_blocks, max_records):
        have = self._cap[slot]
        want = (max(comp_bytes, have[0]), max(raw_bytes, have[1]), max(max_blocks, have[2]), max(max_records, have[3]))
        if want != have:
            self._cap[slot] = (0, 0, 0, 0)
            self._check(self._L.xm_bamdev_reserve(self._h, slot, *want), "xm_bamdev_reserve")
            self._cap[slot] = want

    def capacity(self, slot):
        return self._cap[slot]

    def staging(self, slot, file):
        return _host_view(self._L.xm_bamdev_staging(self._h, slot, file), self._cap[slot][0], np.uint8)

    def run(self, slot, inputs, score_mode, paired, keep_halo, max_records, wait_raw=True, skip_repeated=False):
        """inputs: two dicts {comp_len, blocks (BGZF_BLOCK array), crc (uint32 array), carry_slot, carry_off, carry_len, eof, skip}.
        skip_repeated: the skipping walk of getReadPairs (pair k = the first record of run k of either file).
        wait_raw=False: the inflated windows stay on the device -- the file path then asks for what the writer needs
        (fetch_wanted after classify, or fetch_raw) and waits for it (raw_wait) in the thread that prints the records."""
        arr = (_BamDevInput * 2)()
        keep = []
        for f, x in enumerate(inputs):
            blocks = np.ascontiguousarray(x["blocks"])
            crc = np.ascontiguousarray(x["crc"], dtype=np.uint32)
            keep += [blocks, crc]
            arr[f] = _BamDevInput(int(x["comp_len"]), blocks.ctypes.data if blocks.shape[0] else None,
                                  crc.ctypes.data if crc.shape[0] else None, blocks.shape[0], int(x.get("carry_slot", 0)),
                                  int(x.get("carry_off", 0)), int(x.get("carry_len", 0)), int(bool(x["eof"])), int(x.get("skip", 0)),
                                  int(x.get("uploaded", 0)))
        raw = _BamDevBlock()
        rc = self._L.xm_bamdev_run(self._h, slot, ctypes.byref(arr), int(score_mode), int(bool(paired)), int(bool(skip_repeated)),
                                   int(bool(keep_halo)), int(max_records), ctypes.byref(raw))
        self._check(rc, "xm_bamdev_run")
        blk = BamDevBlock(self, slot, raw)
        if wait_raw:                                                 # the whole inflated windows on the host (tests; windows the host walks)
            self.fetch_raw(slot, blk)
            self.raw_wait(slot)
        return blk

    def fetch_raw(self, slot, block=None):
        """Ask for the whole inflated windows in raw1 / raw2 of the slot's block (complete after raw_wait).  The slot's two
        page-locked buffers for them are made by its first call (a run whose windows all take the device's way never pins
        them): `block.raw_addr` is set here, not by run()."""
        self._check(self._L.xm_bamdev_fetch_raw(self._h, int(slot)), "xm_bamdev_fetch_raw")
        addr = (self._L.xm_bamdev_raw(self._h, int(slot), 0), self._L.xm_bamdev_raw(self._h, int(slot), 1))
        if block is not None:
            block.raw_addr = addr
        return addr

    def fetch_wanted(self, slot, n_records, paired, sink_mask):
        """After classify(): only the records a sink takes, packed (xm_bamdev_fetch_wanted) -> ((address of file 1's packed
        records, of file 2's), (uint32 views: where record i went, 0xFFFFFFFF = no sink takes it), (bytes1, bytes2));
        complete after raw_wait."""
        t = _BamDevText()
        rc = self._L.xm_bamdev_fetch_wanted(self._h, int(slot), int(n_records), int(bool(paired)), int(sink_mask), ctypes.byref(t))
        self._check(rc, "xm_bamdev_fetch_wanted")
        n = int(n_records)
        return ((t.raw1, t.raw2), (_host_view(t.off1, n, np.uint32), _host_view(t.off2, n, np.uint32)), (int(t.bytes1), int(t.bytes2)))

    def set_refs(self, file, names):
        """The file's reference names (a list of bytes, in reference-id order): what the device printer writes in RNAME / RNEXT."""
        blob = np.frombuffer(b"".join(names) or b"\0", dtype=np.uint8)
        at = np.zeros(len(names) + 1, dtype=np.uint32)
        if names:
            at[1:] = np.cumsum([len(x) for x in names], dtype=np.uint64).astype(np.uint32)
        self._check(self._L.xm_bamdev_set_refs(self._h, int(file), _np_ptr(blob), _np_ptr(at), len(names)), "xm_bamdev_set_refs")

    def fetch_text(self, slot, n_records, paired, sink_mask):
        """After classify(): the SAM text of the records a sink takes, printed on the device (xm_bamdev_fetch_text) ->
        (status, (text1, text2 uint8 views), (line_off1, line_off2), (line_len1, line_len2)); status 0: on its way, complete after
        raw_wait; 1 / 2: the host has to print this window (a binary64 field / more text than the buffers hold)."""
        t = _BamDevLines()
        rc = self._L.xm_bamdev_fetch_text(self._h, int(slot), int(n_records), int(bool(paired)), int(sink_mask), ctypes.byref(t))
        self._check(rc, "xm_bamdev_fetch_text")
        n = int(n_records)
        if t.status != 0 or n == 0:
            return int(t.status), None, None, None
        return (0, (_host_view(t.text1, max(int(t.bytes1), 1), np.uint8), _host_view(t.text2, max(int(t.bytes2), 1), np.uint8)),
                (_host_view(t.line_off1, n, np.uint32), _host_view(t.line_off2, n, np.uint32)),
                (_host_view(t.line_len1, n, np.uint32), _host_view(t.line_len2, n, np.uint32)))

    def fetch_bins(self, slot, n_records, paired, sink_mask):
        """After classify(): the six outputs themselves, gathered on the device (xm_bamdev_fetch_bins) -> (status, text, bin_off):
        bin b's text = text[bin_off[b]:bin_off[b + 1]] (a view of a page-locked buffer, valid after raw_wait and until the next
        run on the slot); status 1 / 2: this window is the host's (a binary64 field / more text than the buffers hold)."""
        t = _BamDevBins()
        rc = self._L.xm_bamdev_fetch_bins(self._h, int(slot), int(n_records), int(bool(paired)), int(sink_mask), ctypes.byref(t))
        self._check(rc, "xm_bamdev_fetch_bins")
        if t.status != 0:
            return int(t.status), None, None
        off = [int(v) for v in t.bin_off]
        text = _host_view(t.text, off[7], np.uint8) if off[7] else np.zeros(0, dtype=np.uint8)
        return 0, text, off

    def upload(self, slot, file, nbytes):
        """The first nbytes of the slot's staging buffer go to the device now (xm_bamdev_upload); the next run is told `uploaded`."""
        self._check(self._L.xm_bamdev_upload(self._h, int(slot), int(file), int(nbytes)), "xm_bamdev_upload")

    def raw_wait(self, slot):
        self._check(self._L.xm_bamdev_raw_wait(self._h, int(slot)), "xm_bamdev_raw_wait")

    @_one_call_per_context
    def classify(self, slot, mode, n_records, min_score_floor):
        code, idx = ctypes.c_void_p(), ctypes.c_void_p()
        off = np.zeros(8, dtype=np.uint64)
        counts = np.zeros(64, dtype=np.uint64)
        rc = self._L.xm_bamdev_classify(self._h, slot, int(mode), int(n_records), int(min_score_floor), ctypes.byref(code),
                                        ctypes.byref(idx), _np_ptr(off), _np_ptr(counts))
        self._check(rc, "xm_bamdev_classify")
        return (_host_view(code.value, int(n_records), np.uint8), _host_view(idx.value, int(off[7]), np.uint32), off, counts)

    def columns(self, slot, n_records):
        n = int(n_records)
        out = [np.empty(n, dtype=np.int32) for _ in range(4)] + [np.zeros((n + 63) // 64, dtype=np.uint64)]
        if n:
            self._check(self._L.xm_bamdev_columns(self._h, slot, n, *[_np_ptr(a) for a in out]), "xm_bamdev_columns")
        return out

    def cigar_columns(self, slot, file, n_records):
        """After a SCORE_CIGAR run: (nm int32, cig_cnt uint8, cig_tile uint32, cig_ops uint32) of one file, made on the device."""
        n = int(n_records)
        nm = np.empty(n, dtype=np.int32)
        cnt = np.empty(n, dtype=np.uint8)
        tile = np.zeros(cigar_tiles(n) + 1, dtype=np.uint32)
        n_ops = ctypes.c_uint64()
        self._check(self._L.xm_bamdev_cigar_columns(self._h, slot, file, n, _np_ptr(nm), _np_ptr(cnt), _np_ptr(tile), None, 0,
                                                    ctypes.byref(n_ops)), "xm_bamdev_cigar_columns")
        ops = np.zeros(max(int(n_ops.value), 1), dtype=np.uint32)
        self._check(self._L.xm_bamdev_cigar_columns(self._h, slot, file, n, None, None, None, _np_ptr(ops), ops.shape[0],
                                                    ctypes.byref(n_ops)), "xm_bamdev_cigar_columns")
        return nm, cnt, tile, ops[:int(n_ops.value)]
