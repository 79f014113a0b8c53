"""GPU parity of the segmented bin lists (xm_classify_runs*_dev: one launch, per-granule runs sorted by bin): expanding
the runs must give exactly the oracle's stable split (oracle/xm_oracle.c: xmo_classify_* + xmo_compact, restating
xenomapper.py:321-350, :398-452, :498-554), at every size of the flat forms' parity test, at BASELINE.json's full
sizes, single-end, binary64 with NaN, and irregular unit masks.  Two expansions are checked against each other too: the
C ABI's xm_runs_expand and an independent NumPy one written from the header's description of the layout."""
import itertools

import numpy as np
import pytest

from tests import helpers as H
from tests.helpers import NEG

pytestmark = pytest.mark.gpu

ABSENT = -2**31
GRAN = 2048


@pytest.fixture(scope="module")
def ctx():
    from xenomapper_amd import _ffi
    c = _ffi.Context(0)
    yield c
    c.close()


def np_expand(n, runs16, gran_counts, b):
    """List b from the layout as include/xenomapper_hip.h describes it (NumPy, no library call)."""
    n_gran = (n + GRAN - 1) // GRAN
    c = gran_counts[:8 * n_gran].reshape(n_gran, 8).astype(np.int64)
    start = np.cumsum(c, axis=1) - c                              # where bin b begins inside the granule's slab
    k = c[:, b]
    total = int(k.sum())
    if total == 0:
        return np.zeros(0, dtype=np.uint32)
    g = np.repeat(np.arange(n_gran, dtype=np.int64), k)           # granule of every unit of the list
    first = np.cumsum(k) - k
    within = np.arange(total, dtype=np.int64) - np.repeat(first, k)
    at = g * GRAN + np.repeat(start[:, b], k) + within
    return (g * GRAN + runs16[at].astype(np.int64)).astype(np.uint32)


def run_runs(ctx, mode, cols, bits, m):
    """-> host (runs16, gran_counts, n_out, counts) of one xm_classify_runs*_dev call."""
    import torch
    from xenomapper_amd import _ffi
    dev = torch.device("cuda:0")
    n = cols[0].shape[0]
    n_gran = max(_ffi.runs_granules(n), 1)
    d = [torch.from_numpy(np.ascontiguousarray(c)).to(dev) for c in cols]
    dbits = torch.from_numpy(np.ascontiguousarray(bits).view(np.int64)).to(dev)
    runs = torch.full((n_gran * GRAN,), -1, dtype=torch.int16, device=dev)
    gcnt = torch.full((n_gran * 8,), -1, dtype=torch.int16, device=dev)
    n_out = torch.full((8,), -1, dtype=torch.int64, device=dev)
    counts = torch.full((64,), -1, dtype=torch.int64, device=dev)
    ctx.classify_runs_dev(mode, *d, dbits, m, runs, gcnt, n_out, counts)
    torch.cuda.synchronize()
    return (runs.cpu().numpy().view(np.uint16), gcnt.cpu().numpy().view(np.uint16),
            n_out.cpu().numpy().astype(np.uint64), counts.cpu().numpy().astype(np.uint64))


def check_runs(ctx, mode, cols, bits, m, want=None):
    from xenomapper_amd import _ffi
    n = cols[0].shape[0]
    if want is None:
        want_code, want_counts = H.c_classify(mode, *cols, bits, m)
        want_idx, want_off = H.c_compact(mode, want_code)
    else:
        want_counts, want_idx, want_off = want
    runs16, gcnt, n_out, counts = run_runs(ctx, mode, cols, bits, m)
    assert np.array_equal(counts, want_counts)
    for b in range(7):
        assert int(n_out[b]) == int(want_off[b + 1] - want_off[b]), (b, n_out, want_off)
    assert int(n_out[7]) == int(want_off[7])
    if n == 0:
        return
    n_gran = _ffi.runs_granules(n)
    g = gcnt[:8 * n_gran].reshape(n_gran, 8)
    assert (g[:, 7] == 0).all() and (g.sum(axis=1) <= GRAN).all()
    for b in range(7):
        want_list = want_idx[int(want_off[b]):int(want_off[b + 1])]
        got = np_expand(n, runs16, gcnt, b)
        assert np.array_equal(got, want_list), (mode, b)
        assert np.array_equal(_ffi.runs_expand(n, runs16, gcnt, b), want_list), (mode, b)
    assert ctx.workspace_is_clean()


def random_columns(rng, n, spread=8):
    vals = np.concatenate([[ABSENT, ABSENT], np.arange(-spread, spread + 1)]).astype(np.int64)
    return [vals[rng.integers(0, len(vals), n)].astype(np.int32) for _ in range(4)]


SIZES = [0, 1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 12289,
         100_003, 1_000_003]


@pytest.mark.parametrize("n", SIZES)
def test_runs_sizes_modes(ctx, n):
    rng = np.random.default_rng(n + 29)
    cols = random_columns(rng, n)
    fcols = [np.where(c == ABSENT, NEG, c.astype(np.float64)) for c in cols]
    for mode, m in itertools.product((0, 1, 2), (NEG, 0.5, -3.0)):
        flags = rng.random(n) < (0.55 if mode else 0.9)
        bits = H.synth.pack_unit_bits(flags) if n else np.zeros(1, dtype=np.uint64)
        check_runs(ctx, mode, cols, bits, H.floor_min_score(m))
        check_runs(ctx, mode, fcols, bits, m)                       # binary64 columns: same lists on integral input


def test_runs_every_record_a_unit_and_none(ctx):
    """Granules that are full (2048 units: three equal names in a row everywhere / single-end) and empty."""
    rng = np.random.default_rng(5)
    n = 3 * GRAN + 77
    cols = random_columns(rng, n)
    for mode in (0, 1, 2):
        for flags in (np.ones(n, dtype=bool), np.zeros(n, dtype=bool),
                      np.concatenate([np.ones(GRAN, dtype=bool), np.zeros(n - GRAN, dtype=bool)])):
            check_runs(ctx, mode, cols, H.synth.pack_unit_bits(flags), ABSENT)


def test_runs_one_bin_only(ctx):
    """All units in one bin (one long run per granule), for every bin a mode can produce."""
    n = 2 * GRAN + 5
    rows = {0: (5, ABSENT, 1, ABSENT), 1: (1, ABSENT, 5, ABSENT), 2: (5, 5, 1, ABSENT), 3: (1, ABSENT, 5, 5),
            4: (5, ABSENT, 5, ABSENT), 5: (ABSENT, ABSENT, ABSENT, ABSENT)}
    for b, row in rows.items():
        cols = [np.full(n, v, dtype=np.int32) for v in row]
        for mode in (0, 1, 2):
            flags = np.ones(n, dtype=bool) if mode == 0 else (np.arange(n) % 2 == 1)
            check_runs(ctx, mode, cols, H.synth.pack_unit_bits(flags), ABSENT)


def test_runs_nan_units_go_to_slot_6(ctx):
    nan = float("nan")
    rng = np.random.default_rng(11)
    n = GRAN + 300
    cols = [c.astype(np.float64) for c in random_columns(rng, n)]
    cols = [np.where(c == ABSENT, NEG, c) for c in cols]
    for c in cols:
        c[rng.random(n) < 0.05] = nan
    for mode in (0, 1, 2):
        flags = rng.random(n) < 0.7
        bits = H.synth.pack_unit_bits(flags)
        want_code, want_counts = H.c_classify(mode, *cols, bits, NEG)
        want_idx, want_off = H.c_compact(mode, want_code)
        assert int(want_off[7] - want_off[6]) > 0                   # the case is exercised
        check_runs(ctx, mode, cols, bits, NEG, want=(want_counts, want_idx, want_off))


def test_runs_full_size_paired_and_single_end(ctx):
    """BASELINE.json configs[1] / configs[4] (50 M pairs, liberal and conservative) and 100 M single-end reads through the
    one-launch form; exact against the C oracle plus size-independent properties of the layout."""
    import torch
    from xenomapper_amd import _ffi
    n_pairs = 50_000_000
    cols = H.synth.score_columns(n_pairs, seed=2002)
    n = 2 * n_pairs
    dev = torch.device("cuda:0")
    d = {k: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) for k, v in cols.items()}
    n_gran = _ffi.runs_granules(n)
    runs = torch.empty(n_gran * GRAN, dtype=torch.int16, device=dev)
    gcnt = torch.empty(n_gran * 8, dtype=torch.int16, device=dev)
    n_out = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    se_bits = torch.full(((n + 63) // 64,), -1, dtype=torch.int64, device=dev)       # every record yielded
    for mode in (_ffi.MODE_PE_LIBERAL, _ffi.MODE_PE_CONSERVATIVE, _ffi.MODE_SE):
        bits_d = se_bits if mode == _ffi.MODE_SE else d["unit_bits"]
        bits_h = np.full((n + 63) // 64, ~np.uint64(0), dtype=np.uint64) if mode == _ffi.MODE_SE else cols["unit_bits"]
        runs.fill_(-1), gcnt.fill_(-1)
        ctx.classify_runs_dev(mode, d["as1"], d["xs1"], d["as2"], d["xs2"], bits_d, ABSENT, runs, gcnt, n_out, counts)
        torch.cuda.synchronize()
        want_code, want_counts = H.c_classify(mode, cols["as1"], cols["xs1"], cols["as2"], cols["xs2"], bits_h, ABSENT)
        want_idx, want_off = H.c_compact(mode, want_code)
        h_runs, h_gcnt = runs.cpu().numpy().view(np.uint16), gcnt.cpu().numpy().view(np.uint16)
        h_n = n_out.cpu().numpy().astype(np.uint64)
        assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
        units = n if mode == _ffi.MODE_SE else n_pairs
        assert int(h_n[7]) == units == int(want_off[7])
        seen = 0
        for b in range(6):
            got = np_expand(n, h_runs, h_gcnt, b)
            assert int(h_n[b]) == got.shape[0]
            assert np.array_equal(got, want_idx[int(want_off[b]):int(want_off[b + 1])]), (mode, b)
            assert got.shape[0] < 2 or (np.diff(got.astype(np.int64)) > 0).all()      # input order inside a list
            seen += got.shape[0]
        assert seen == units                                                          # every unit in exactly one list
        assert np.array_equal(_ffi.runs_expand(n, h_runs, h_gcnt, 0), want_idx[:int(want_off[1])])
    assert ctx.workspace_is_clean()


def test_runs_then_flat_forms_share_the_workspace(ctx):
    """The count replicas are the context's: a runs call between two flat calls leaves them as it found them."""
    rng = np.random.default_rng(3)
    n = 50_000
    cols = random_columns(rng, n)
    bits = H.synth.pack_unit_bits(rng.random(n) < 0.5)
    want_code, want_counts = H.c_classify(1, *cols, bits, ABSENT)
    want_idx, want_off = H.c_compact(1, want_code)
    for _ in range(2):
        _, idx, off, counts = ctx.classify_compact(1, *cols, bits, ABSENT)
        assert np.array_equal(idx, want_idx) and np.array_equal(off, want_off) and np.array_equal(counts, want_counts)
        check_runs(ctx, 1, cols, bits, ABSENT, want=(want_counts, want_idx, want_off))
