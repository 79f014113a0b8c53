"""Property test of the kernels through the C ABI: random sizes, unit masks, score values (ties, zeros, extremes,
absent; binary64 with NaN / +-inf / fractions), thresholds and CIGAR shapes against the C oracle -- category bytes,
category_counts and the stable split, bit for bit."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from tests import helpers as H

pytestmark = pytest.mark.gpu
ABSENT = -2**31
N_EX = int(os.environ.get("XM_FUZZ_EXAMPLES", "300"))


@pytest.fixture(scope="module")
def ctx():
    from xenomapper_amd import _ffi
    c = _ffi.Context(0)
    yield c
    c.close()


SIZES = st.one_of(st.integers(0, 70), st.integers(250, 260), st.integers(2040, 2056), st.integers(4090, 4100),
                  st.integers(8185, 8200), st.integers(0, 40000))


def columns(rng, n, kind):
    if kind == "narrow":
        vals = np.array([ABSENT, ABSENT, -2, -1, 0, 0, 1, 2, 3], dtype=np.int64)
    elif kind == "wide":
        vals = np.array([ABSENT, ABSENT + 1, -2**30, -1, 0, 1, 2**30, 2**31 - 2, 2**31 - 1], dtype=np.int64)
    else:
        vals = np.concatenate([[ABSENT] * 3, np.arange(-60, 301)]).astype(np.int64)
    return [vals[rng.integers(0, len(vals), n)].astype(np.int32) for _ in range(4)]


def mask(rng, n, kind):
    if kind == "interleaved":
        f = np.zeros(n, dtype=np.uint8)
        f[1::2] = 1
    elif kind == "all":
        f = np.ones(n, dtype=np.uint8)
    elif kind == "none":
        f = np.zeros(n, dtype=np.uint8)
    else:
        f = (rng.random(n) < rng.random()).astype(np.uint8)
    return f


@settings(max_examples=N_EX, deadline=None, suppress_health_check=list(HealthCheck))
@given(n=SIZES, seed=st.integers(0, 2**31), mode=st.sampled_from([0, 1, 2]),
       kind=st.sampled_from(["narrow", "wide", "scores"]), mkind=st.sampled_from(["interleaved", "all", "none", "random"]),
       m=st.sampled_from([ABSENT, -3, 0, 1, 2**31 - 1, 100]))
def test_int_path(ctx, n, seed, mode, kind, mkind, m):
    rng = np.random.default_rng(seed)
    cols = columns(rng, n, kind)
    bits = H.synth.pack_unit_bits(mask(rng, n, mkind)) if n else np.zeros(1, dtype=np.uint64)
    code, counts = ctx.classify(mode, *cols, bits, m)
    want, want_counts = H.c_classify(mode, *cols, bits, m)
    assert np.array_equal(code, want) and np.array_equal(counts, want_counts)
    idx, off, counts2 = ctx.compact(mode, code)
    want_idx, want_off = H.c_compact(mode, want)
    assert np.array_equal(off, want_off) and np.array_equal(idx, want_idx) and np.array_equal(counts2, want_counts)
    fcode, fidx, foff, fcounts = ctx.classify_compact(mode, *cols, bits, m)            # category bytes between K1 and K2c
    assert np.array_equal(fcode, want) and np.array_equal(fcounts, want_counts)
    assert np.array_equal(foff, want_off) and np.array_equal(fidx, want_idx)
    fcode, fidx, foff, fcounts = ctx.classify_compact(mode, *cols, bits, m, want_code=False)   # the compact stream instead
    assert fcode is None and np.array_equal(fcounts, want_counts)
    assert np.array_equal(foff, want_off) and np.array_equal(fidx, want_idx)


@settings(max_examples=N_EX, deadline=None, suppress_health_check=list(HealthCheck))
@given(n=SIZES, seed=st.integers(0, 2**31), mode=st.sampled_from([0, 1, 2]),
       m=st.sampled_from([float("-inf"), float("inf"), float("nan"), -0.5, 0.0, 2.75]))
def test_f64_path(ctx, n, seed, mode, m):
    rng = np.random.default_rng(seed)
    vals = np.array([float("-inf"), float("-inf"), float("inf"), float("nan"), -0.0, 0.0, 0.5, 1.0, 1.5, -2.25, 3e10, 7.0])
    cols = [vals[rng.integers(0, len(vals), n)] for _ in range(4)]
    bits = H.synth.pack_unit_bits(mask(rng, n, "random")) if n else np.zeros(1, dtype=np.uint64)
    code, counts = ctx.classify_f64(mode, *cols, bits, m)
    want, want_counts = H.c_classify(mode, *cols, bits, m)
    assert np.array_equal(code, want) and np.array_equal(counts, want_counts)
    idx, off, _ = ctx.compact(mode, code)
    want_idx, want_off = H.c_compact(mode, want)
    assert np.array_equal(off, want_off) and np.array_equal(idx, want_idx)
    fcode, fidx, foff, fcounts = ctx.classify_compact(mode, *cols, bits, m)            # category bytes between K1 and K2c
    assert np.array_equal(fcode, want) and np.array_equal(fcounts, want_counts)
    assert np.array_equal(foff, want_off) and np.array_equal(fidx, want_idx)
    fcode, fidx, foff, fcounts = ctx.classify_compact(mode, *cols, bits, m, want_code=False)   # the compact stream instead
    assert fcode is None and np.array_equal(fcounts, want_counts)
    assert np.array_equal(foff, want_off) and np.array_equal(fidx, want_idx)


@settings(max_examples=N_EX, deadline=None, suppress_health_check=list(HealthCheck))
@given(n=SIZES, seed=st.integers(0, 2**31), mode=st.sampled_from([0, 1, 2]), max_ops=st.sampled_from([0, 1, 3, 4, 12]),
       m=st.sampled_from([ABSENT, -40, -7, 0]))
def test_cigar_path(ctx, n, seed, mode, max_ops, m):
    rng = np.random.default_rng(seed)

    def cig():
        n_ops = rng.integers(0, max_ops + 1, n).astype(np.uint32)
        off = np.zeros(n + 1, dtype=np.uint32)
        np.cumsum(n_ops, out=off[1:])
        total = int(off[-1])
        ops = (rng.integers(0, 300, total).astype(np.uint32) << 4) | rng.integers(0, 16, total).astype(np.uint32)
        nm = np.where(rng.random(n) < 0.2, ABSENT, rng.integers(0, 9, n)).astype(np.int32)
        return {"nm": nm, "cig_off": off, "cig_oplen": ops}
    c1, c2 = cig(), cig()
    xs = [np.where(rng.random(n) < 0.7, ABSENT, -rng.integers(0, 60, n)).astype(np.int32) for _ in range(2)]
    bits = H.synth.pack_unit_bits(mask(rng, n, "random")) if n else np.zeros(1, dtype=np.uint64)
    a1, bad1 = H.c_cigar_scores(c1["nm"], c1["cig_off"], c1["cig_oplen"])
    a2, bad2 = H.c_cigar_scores(c2["nm"], c2["cig_off"], c2["cig_oplen"])
    assert bad1 == 0 and bad2 == 0
    got = ctx.cigar_scores(c1["nm"], c1["cig_off"], c1["cig_oplen"])
    assert np.array_equal(got, a1)
    code, counts = ctx.classify_cigar(mode, c1["nm"], c1["cig_off"], c1["cig_oplen"], xs[0],
                                      c2["nm"], c2["cig_off"], c2["cig_oplen"], xs[1], bits, m)
    want, want_counts = H.c_classify(mode, a1, xs[0], a2, xs[1], bits, m)
    assert np.array_equal(code, want) and np.array_equal(counts, want_counts)
    want_idx, want_off = H.c_compact(mode, want)
    fcode, fidx, foff, fcounts = ctx.classify_compact_cigar(mode, c1["nm"], c1["cig_off"], c1["cig_oplen"], xs[0],
                                                            c2["nm"], c2["cig_off"], c2["cig_oplen"], xs[1], bits, m)
    assert np.array_equal(fcode, want) and np.array_equal(fcounts, want_counts)
    assert np.array_equal(foff, want_off) and np.array_equal(fidx, want_idx)
