"""GPU parity: the HIP path, called through the C ABI, against the oracle -- bit-exact.

Run on the MI355X box with `pytest -m gpu`.  Sizes cover empty / ragged tails / tile and chunk
boundaries / BASELINE.json's 50 M-pair configuration; inputs cover every (fwd, rev) tuple, the
XS == 0 quirk, min_score thresholds, NaN (f64 path) and irregular unit masks.
"""
import itertools

import numpy as np
import pytest

from tests import helpers as H
from tests.helpers import ORACLE, NEG

pytestmark = pytest.mark.gpu

ABSENT = -2**31


@pytest.fixture(scope="module")
def ctx():
    from xenomapper_amd import _ffi
    c = _ffi.Context(0)
    yield c
    c.close()


def random_columns(rng, n, spread=8):
    vals = np.concatenate([[ABSENT, ABSENT], np.arange(-spread, spread + 1)]).astype(np.int64)
    return [vals[rng.integers(0, len(vals), n)].astype(np.int32) for _ in range(4)]


def check_place(ctx, mode, cols, bits, m, want_code, want_counts, want_idx, want_off, want_code_out=True):
    """The single-pass entry point (xm_classify_place: six lists) against the oracle's split."""
    code, lists, n_out, counts = ctx.classify_place(mode, *cols, bits, m, want_code=want_code_out)
    if want_code_out:
        assert np.array_equal(code, want_code)
    assert np.array_equal(counts, want_counts)
    for b in range(len(lists)):
        want = want_idx[int(want_off[b]):int(want_off[b + 1])]
        assert int(n_out[b]) == want.shape[0], (b, n_out, want_off)
        assert np.array_equal(lists[b], want), b
    if len(lists) == 6:
        assert int(n_out[6]) == int(want_off[7] - want_off[6])
    assert int(n_out[7]) == int(want_off[7])


def check_all(ctx, mode, cols, bits, m_float):
    n = cols[0].shape[0]
    mi = H.floor_min_score(m_float)
    code, counts = ctx.classify(mode, *cols, bits, mi)
    want_code, want_counts = H.c_classify(mode, *cols, bits, mi)
    assert np.array_equal(code, want_code)
    assert np.array_equal(counts, want_counts)
    idx, off, counts2 = ctx.compact(mode, code)
    want_idx, want_off = H.c_compact(mode, want_code)
    assert np.array_equal(off, want_off)
    assert np.array_equal(idx, want_idx)
    assert np.array_equal(counts2, want_counts)
    # the fused pass (classify + count in one kernel, scan, scatter) gives the same four results
    fcode, fidx, foff, fcounts = ctx.classify_compact(mode, *cols, bits, mi)
    assert np.array_equal(fcode, want_code) and np.array_equal(fcounts, want_counts)
    assert np.array_equal(foff, want_off) and np.array_equal(fidx, want_idx)
    check_place(ctx, mode, cols, bits, mi, want_code, want_counts, want_idx, want_off)
    # the binary64 path must agree with the integer path on integral input
    fcols = [np.where(c == ABSENT, NEG, c.astype(np.float64)) for c in cols]
    check_place(ctx, mode, fcols, bits, m_float, want_code, want_counts, want_idx, want_off)
    codef, countsf = ctx.classify_f64(mode, *fcols, bits, m_float)
    assert np.array_equal(codef, want_code)
    assert np.array_equal(countsf, want_counts)
    fcode, fidx, foff, fcounts = ctx.classify_compact(mode, *fcols, bits, m_float, want_code=False)
    assert fcode is None and np.array_equal(fcounts, want_counts)
    assert np.array_equal(foff, want_off) and np.array_equal(fidx, want_idx)


SIZES = [0, 1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 12289,
         100_003, 1_000_003]


@pytest.mark.parametrize("n", SIZES)
def test_sizes_modes(ctx, n):
    rng = np.random.default_rng(n + 17)
    cols = random_columns(rng, n)
    for mode, m in itertools.product((0, 1, 2), (NEG, 0.5, -3.0)):
        flags = rng.random(n) < (0.55 if mode else 0.9)
        bits = H.synth.pack_unit_bits(flags) if n else np.zeros(1, dtype=np.uint64)
        check_all(ctx, mode, cols, bits, m)


def test_g1_lattice_on_gpu(ctx):
    g = H.golden("g1_mapping_state.json")
    vals = [H.unnum(v) for v in g["lattice_values"]]
    mins = [H.unnum(m) for m in g["lattice_min_scores"]]
    grid = np.array(list(itertools.product(vals, repeat=4)), dtype=np.float64)
    n = grid.shape[0]
    bits = H.synth.pack_unit_bits(np.ones(n, dtype=np.uint8))
    k = 0
    for m in mins:
        want = np.frombuffer(g["lattice_states"][k:k + n].encode(), dtype=np.uint8) - ord("0")
        k += n
        fcols = [np.ascontiguousarray(grid[:, j]) for j in range(4)]
        code, _ = ctx.classify_f64(0, *fcols, bits, m)
        assert np.array_equal(code, want)
        icols = [np.where(c == NEG, ABSENT, c).astype(np.int32) for c in fcols]
        code, _ = ctx.classify(0, *icols, bits, H.floor_min_score(m))
        assert np.array_equal(code, want)


def test_g1_rows_f64_on_gpu(ctx):
    g = H.golden("g1_mapping_state.json")
    for row in g["rows"]:
        v = [H.unnum(x) for x in row[:5]]
        cols = [np.array([x], dtype=np.float64) for x in v[:4]]
        code, _ = ctx.classify_f64(0, *cols, np.array([1], dtype=np.uint64), v[4])
        assert int(code[0]) == row[5], row


def test_nan_gives_state6(ctx):
    nan = float("nan")
    a1 = np.array([nan, 3.0, 1.0, 2.0], dtype=np.float64)
    x1 = np.array([1.0, nan, NEG, NEG], dtype=np.float64)
    a2 = np.array([2.0, 2.0, nan, 2.0], dtype=np.float64)
    x2 = np.array([3.0, nan, NEG, NEG], dtype=np.float64)
    bits = np.array([0b1111], dtype=np.uint64)
    for mode in (0, 1, 2):
        code, counts = ctx.classify_f64(mode, a1, x1, a2, x2, bits, NEG)
        want, want_counts = H.c_classify(mode, a1, x1, a2, x2, bits, NEG)
        assert np.array_equal(code, want) and np.array_equal(counts, want_counts)
        idx, off, _ = ctx.compact(mode, code)
        widx, woff = H.c_compact(mode, want)
        assert np.array_equal(off, woff) and np.array_equal(idx, widx)
    code, _ = ctx.classify_f64(0, a1, x1, a2, x2, bits, NEG)
    assert code.tolist() == [6, 2, 6, 4]
    # NaN min_score: nothing is `<= m` or `> m` (xenomapper.py:275-288) -> unresolved if equal else fall-through
    code, _ = ctx.classify_f64(0, a1, x1, a2, x2, bits, nan)
    assert code.tolist() == [6, 6, 6, 4]


def test_all_tuples_and_xs0(ctx):
    # one realisation per state incl. the XS == 0 quirk (AS=0,XS=0 is *specific*)
    real = {0: (0, 0, -5, ABSENT), 1: (-5, ABSENT, 0, 0), 2: (7, 7, 3, ABSENT), 3: (3, ABSENT, 7, 9),
            4: (4, 0, 4, 0), 5: (ABSENT, 1, ABSENT, 1)}
    recs = []
    for f, r in itertools.product(range(6), repeat=2):
        recs += [real[f], real[r]]
    cols = [np.array([r[j] for r in recs], dtype=np.int32) for j in range(4)]
    n = len(recs)
    flags = np.zeros(n, dtype=np.uint8)
    flags[1::2] = 1
    bits = H.synth.pack_unit_bits(flags)
    for mode in (1, 2):
        code, counts = ctx.classify(mode, *cols, bits, ABSENT)
        units = code[1::2]
        assert units.tolist() == [f * 8 + r for f, r in itertools.product(range(6), repeat=2)]
        assert int(counts.sum()) == 36 and set(counts[counts > 0].tolist()) == {1}
        idx, off, _ = ctx.compact(mode, code)
        for b in range(6):
            for i in idx[int(off[b]):int(off[b + 1])]:
                c = int(code[i])
                assert ORACLE.bin_of(mode, c >> 3, c & 7) == b


def test_irregular_masks(ctx):
    rng = np.random.default_rng(5)
    n = 70_001
    cols = random_columns(rng, n)
    for pattern in ("none", "all", "first_only", "last_only", "runs"):
        flags = np.zeros(n, dtype=np.uint8)
        if pattern == "all":
            flags[:] = 1
        elif pattern == "first_only":
            flags[0] = 1
        elif pattern == "last_only":
            flags[-1] = 1
        elif pattern == "runs":
            flags[(np.arange(n) // 37) % 3 == 1] = 1
        bits = H.synth.pack_unit_bits(flags)
        for mode in (0, 1, 2):
            check_all(ctx, mode, cols, bits, NEG)


def test_extreme_values(ctx):
    vals = np.array([ABSENT, ABSENT + 1, -1, 0, 1, 2**31 - 2, 2**31 - 1], dtype=np.int64)
    grid = np.array(list(itertools.product(vals, repeat=4)), dtype=np.int64).astype(np.int32)
    cols = [np.ascontiguousarray(grid[:, j]) for j in range(4)]
    n = grid.shape[0]
    bits = H.synth.pack_unit_bits(np.ones(n, dtype=np.uint8))
    for mi in (ABSENT, ABSENT + 1, -1, 0, 2**31 - 2, 2**31 - 1):
        for mode in (0, 1):
            code, counts = ctx.classify(mode, *cols, bits, mi)
            want, wc = H.c_classify(mode, *cols, bits, mi)
            assert np.array_equal(code, want) and np.array_equal(counts, wc)


def test_cigar_scores(ctx):
    for n, seed in ((0, 1), (1, 2), (257, 3), (100_003, 4)):
        cols = H.synth.cigar_columns(n, seed) if n else {"nm": np.zeros(0, np.int32),
                                                         "cig_off": np.zeros(1, np.uint32),
                                                         "cig_oplen": np.zeros(0, np.uint32)}
        got = ctx.cigar_scores(cols["nm"], cols["cig_off"], cols["cig_oplen"])
        want, bad = H.c_cigar_scores(cols["nm"], cols["cig_off"], cols["cig_oplen"])
        assert bad == 0 and np.array_equal(got, want)
    # every op code, long ops, many ops per record
    rng = np.random.default_rng(9)
    n = 5000
    n_ops = rng.integers(0, 40, n).astype(np.uint32)
    off = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(n_ops, out=off[1:])
    ops = ((rng.integers(0, 3000, int(off[-1])).astype(np.uint32) << 4) | rng.integers(0, 9, int(off[-1])).astype(np.uint32))
    nm = np.where(rng.random(n) < 0.1, ABSENT, rng.integers(-3, 50, n)).astype(np.int32)
    got = ctx.cigar_scores(nm, off, ops)
    want, bad = H.c_cigar_scores(nm, off, ops)
    assert bad == 0 and np.array_equal(got, want)
    # range error
    big = np.array([(2**28 - 1) << 4 | 1] * 8, dtype=np.uint32)
    with pytest.raises(OverflowError):
        ctx.cigar_scores(np.array([0], np.int32), np.array([0, 8], np.uint32), big)


def _oracle_cigar_classify(mode, c1, xs1, c2, xs2, bits, mi):
    a1, bad1 = H.c_cigar_scores(c1["nm"], c1["cig_off"], c1["cig_oplen"])
    a2, bad2 = H.c_cigar_scores(c2["nm"], c2["cig_off"], c2["cig_oplen"])
    assert bad1 == 0 and bad2 == 0
    return H.c_classify(mode, a1, xs1, a2, xs2, bits, mi)


def _random_cigar(rng, n, max_ops=12, op_hi=9):
    n_ops = rng.integers(0, max_ops + 1, n).astype(np.uint32)
    n_ops[rng.random(n) < 0.5] = 1
    off = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(n_ops, out=off[1:])
    total = int(off[-1])
    # op codes 0..8 = MIDNSHP=X; op_hi = 16 also draws the codes no CIGAR operation has (they score nothing)
    ops = (rng.integers(1, 60, total).astype(np.uint32) << 4) | rng.integers(0, op_hi, total).astype(np.uint32)
    nm = np.where(rng.random(n) < 0.15, ABSENT, rng.integers(0, 6, n)).astype(np.int32)
    return {"nm": nm, "cig_off": off, "cig_oplen": ops}


def _check_csr_dev(ctx, mode, c1, xs1, c2, xs2, bits, mi):
    """The CSR-column device entry points (K1c, round 1's fused kernel): xm_classify_cigar_dev and
    xm_classify_compact_cigar_dev against the oracle."""
    import torch
    n = c1["nm"].shape[0]
    dev = torch.device("cuda:0")
    want, want_counts = _oracle_cigar_classify(mode, c1, xs1, c2, xs2, bits, mi)
    want_idx, want_off = H.c_compact(mode, want)

    def up(a):
        if a.shape[0] == 0:
            a = np.zeros(4, dtype=a.dtype)
        return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else (a.view(np.int64) if a.dtype == np.uint64 else a)).to(dev)
    d = [up(x) for x in (c1["nm"], c1["cig_off"], c1["cig_oplen"], xs1, c2["nm"], c2["cig_off"], c2["cig_oplen"], xs2, bits)]
    code = torch.full((n + 16,), 0xAA, dtype=torch.uint8, device=dev)
    flag = torch.zeros(4, dtype=torch.int32, device=dev)
    ctx.classify_cigar_dev(mode, *d, mi, code, range_flag=flag)
    torch.cuda.synchronize()
    assert int(flag[0].item()) == 0 and np.array_equal(code[:n].cpu().numpy(), want)
    code.fill_(0xAA)
    idx = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
    off = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    ctx.classify_compact_cigar_dev(mode, *d, mi, code, idx, off, counts, range_flag=flag)
    torch.cuda.synchronize()
    assert np.array_equal(code[:n].cpu().numpy(), want)
    assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
    h_off = off.cpu().numpy().astype(np.uint64)
    assert np.array_equal(h_off, want_off)
    assert np.array_equal(idx[:int(h_off[7])].cpu().numpy().view(np.uint32), want_idx)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 255, 257, 1023, 1025, 2049, 4097, 100_003])
def test_classify_cigar_csr_dev(ctx, n):
    """K1c keeps serving CSR columns that are already on the device: ragged tails (its bounds-checked partial
    workgroup), all modes, a threshold."""
    rng = np.random.default_rng(500 + n)
    c1, c2 = _random_cigar(rng, n), _random_cigar(rng, n)
    xs = [np.where(rng.random(n) < 0.8, ABSENT, -rng.integers(0, 200, n)).astype(np.int32) for _ in range(2)]
    for mode, m in itertools.product((0, 1, 2), (NEG, -40.5)):
        flags = rng.random(n) < (0.55 if mode else 0.9)
        _check_csr_dev(ctx, mode, c1, xs[0], c2, xs[1], H.synth.pack_unit_bits(flags), H.floor_min_score(m))


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 257, 2047, 2048, 2049, 4097, 100_003])
def test_classify_cigar_fused(ctx, n):
    """K3 fused into K1 (the --cigar_scores path) against oracle CIGAR scores + oracle classify."""
    rng = np.random.default_rng(100 + n)
    c1, c2 = _random_cigar(rng, n), _random_cigar(rng, n)
    xs = [np.where(rng.random(n) < 0.8, ABSENT, -rng.integers(0, 200, n)).astype(np.int32) for _ in range(2)]
    for mode, m in itertools.product((0, 1, 2), (NEG, -40.5)):
        flags = rng.random(n) < (0.55 if mode else 0.9)
        bits = H.synth.pack_unit_bits(flags) if n else np.zeros(1, dtype=np.uint64)
        mi = H.floor_min_score(m)
        code, counts = ctx.classify_cigar(mode, c1["nm"], c1["cig_off"], c1["cig_oplen"], xs[0],
                                          c2["nm"], c2["cig_off"], c2["cig_oplen"], xs[1], bits, mi)
        want, want_counts = _oracle_cigar_classify(mode, c1, xs[0], c2, xs[1], bits, mi)
        assert np.array_equal(code, want) and np.array_equal(counts, want_counts)
        # the same through the fused pass (1024-record granules: the classify_cigar workgroup)
        want_idx, want_off = H.c_compact(mode, want)
        fcode, fidx, foff, fcounts = ctx.classify_compact_cigar(mode, c1["nm"], c1["cig_off"], c1["cig_oplen"], xs[0],
                                                                c2["nm"], c2["cig_off"], c2["cig_oplen"], xs[1], bits, mi)
        assert np.array_equal(fcode, want) and np.array_equal(fcounts, want_counts)
        assert np.array_equal(foff, want_off) and np.array_equal(fidx, want_idx)


# one realisation of every state (AS1, XS1, AS2, XS2); state 6 needs a NaN (binary64 path only)
_REAL = {0: (0, 0, -5, ABSENT), 1: (-5, ABSENT, 0, 0), 2: (7, 7, 3, ABSENT), 3: (3, ABSENT, 7, 9),
         4: (4, 0, 4, 0), 5: (ABSENT, 1, ABSENT, 1)}


def _columns_of_states(states):
    table = np.array([_REAL[k] for k in range(6)], dtype=np.int64)
    return [np.ascontiguousarray(table[states, j]).astype(np.int32) for j in range(4)]


_EDGE_N = 5 * 2048 + 77          # > one K2 workgroup (4 granules of 2048), ragged tail


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_split_edges_single_bin(ctx, mode):
    """Whole granules / 256-record groups in ONE bin, for every bin: the rank of a unit is then its index, every
    ballot mask is full, and every other bin is empty (the wave-uniform skips)."""
    n = _EDGE_N
    for b in range(6):
        cols = _columns_of_states(np.full(n, b))
        flags = np.ones(n, dtype=np.uint8)
        if mode:
            flags[0::2] = 0
        check_all(ctx, mode, cols, H.synth.pack_unit_bits(flags), NEG)
        if mode == 0:
            code, idx, off, counts = ctx.classify_compact(0, *cols, H.synth.pack_unit_bits(flags), ABSENT)
            assert int(off[b + 1] - off[b]) == n and np.array_equal(idx, np.arange(n, dtype=np.uint32))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("k", [1, 2, 127, 128, 255, 512, 1023])
def test_split_edges_two_bins(ctx, mode, k):
    """Every 1024 records: the first k in bin a, the rest in bin b, for every ordered pair (a, b) -- counts of
    k / 1024-k (512-unit groups in the paired modes) meeting at every possible lane of a 256-record group."""
    n = _EDGE_N
    pos = np.arange(n) % 1024
    for a, b in itertools.permutations(range(6), 2):
        states = np.where(pos < k, a, b)
        if mode:
            states[0::2] = states[1::2][: len(states[0::2])] if n % 2 == 0 else np.append(states[1::2], states[-1])[: len(states[0::2])]
        cols = _columns_of_states(states)
        flags = np.ones(n, dtype=np.uint8)
        if mode:
            flags[0::2] = 0
        check_all(ctx, mode, cols, H.synth.pack_unit_bits(flags), NEG)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("n", [2 * (1 << 21) + 5, 7 * (1 << 21) + 2049, 8 * (1 << 21)])
def test_scan_carry_between_parts_of_the_one_launch_scan(ctx, n, mode):
    """2 to 8 scan parts (a part = 1024 granules = 2^21 records): K2b's carry between parts -- the part totals the
    counting side adds up in replicas (count_flush), summed by every scan workgroup for the parts in front of its own.
    One bin has no unit at all in part 0 and one none after part 0, so a wrong carry shows up in gran_off / the index
    lists; the stand-alone and the fused forms are both compared, and each twice (the totals must be zero again)."""
    rng = np.random.default_rng(n % 1000 + mode)
    states = rng.integers(0, 6, n)
    part0 = np.arange(n) < (1 << 21)
    states[part0 & (states == 3)] = 0                              # bin 3: nothing in part 0
    states[~part0 & (states == 1)] = 2                             # bin 1: nothing after part 0
    if mode:
        states[0::2] = states[1::2][: len(states[0::2])] if n % 2 == 0 else np.append(states[1::2], states[-1])[: len(states[0::2])]
    cols = _columns_of_states(states)
    flags = np.ones(n, dtype=np.uint8)
    if mode:
        flags[0::2] = 0
    bits = H.synth.pack_unit_bits(flags)
    want_code, want_counts = H.c_classify(mode, *cols, bits, ABSENT)
    want_idx, want_off = H.c_compact(mode, want_code)
    first3 = want_idx[int(want_off[3])] if want_off[4] > want_off[3] else None
    assert first3 is None or first3 >= (1 << 21)
    code, counts = ctx.classify(mode, *cols, bits, ABSENT)
    assert np.array_equal(code, want_code) and np.array_equal(counts, want_counts)
    idx, off, counts2 = ctx.compact(mode, code)
    assert np.array_equal(off, want_off) and np.array_equal(idx, want_idx) and np.array_equal(counts2, want_counts)
    for want_bytes in (True, False, True):                         # category bytes / compact stream between K1 and K2c
        fcode, fidx, foff, fcounts = ctx.classify_compact(mode, *cols, bits, ABSENT, want_code=want_bytes)
        assert np.array_equal(fcounts, want_counts) and np.array_equal(foff, want_off) and np.array_equal(fidx, want_idx)
        assert fcode is None or np.array_equal(fcode, want_code)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("p_single,p_triple", [(0.01, 0.0), (0.3, 0.0), (0.05, 0.002), (0.0, 0.2)])
def test_paired_input_with_unpaired_reads(ctx, mode, p_single, p_triple):
    """Mates that are not strictly interleaved: reads without a mate flip the parity of everything behind them, so a
    lane of K2c holds its (at most two) units at any of its four records -- the two-units-per-lane form; a few runs of
    three equal names (three units in a lane can then occur) send single 256-record groups to the general form."""
    rng = np.random.default_rng(int(1000 * p_single + 7 * mode + 100000 * p_triple))
    n = 3 * 2048 * 7 + 1234
    sizes = rng.choice([1, 2, 3], size=n, p=[p_single, 1.0 - p_single - p_triple, p_triple])
    names = np.repeat(np.arange(n), sizes)[:n]
    flags = np.zeros(n, dtype=np.uint8)
    flags[1:] = names[1:] == names[:-1]
    for skew in (None, 0):
        states = rng.integers(0, 6, n) if skew is None else np.where(rng.random(n) < 0.9, skew, rng.integers(0, 6, n))
        check_all(ctx, mode, _columns_of_states(states), H.synth.pack_unit_bits(flags), NEG)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_scatter_staging_threshold_granule_by_granule(ctx, mode):
    """In the single-end loop K2c sorts a granule's units inside an LDS slab and copies the bins' runs out with wide
    stores when the granule holds XM_STAGE_MIN_UNITS (1536) units or more, and writes every index directly otherwise --
    decided per granule (the paired loops always write directly; they run the same inputs here).
    Granules with 0, 1, 1535, 1536, 1537, 2047 and 2048 units side by side (in the paired modes every record of a dense
    granule closes a unit: runs of equal names), random states, so that runs of every length -- shorter than a wave,
    with and without head / tail places around the 16-byte groups -- start at every alignment."""
    rng = np.random.default_rng(1536 + mode)
    per_gran = [2048, 0, 1535, 1, 1536, 2047, 1537, 2048, 1536, 700, 2048]
    n = 2048 * len(per_gran) + 300
    flags = np.zeros(n, dtype=np.uint8)
    for k, units in enumerate(per_gran):
        at = rng.choice(2048, units, replace=False) + 2048 * k
        flags[at] = 1
    flags[2048 * len(per_gran):] = 1                                # a dense ragged tail
    flags[0] = 0
    for skew in (None, 0, 3):                                       # flat, and two heavily skewed state distributions
        if skew is None:
            states = rng.integers(0, 6, n)
        else:
            states = np.where(rng.random(n) < 0.9, skew, rng.integers(0, 6, n))
        check_all(ctx, mode, _columns_of_states(states), H.synth.pack_unit_bits(flags), NEG)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_split_edges_with_state6_units(ctx, mode):
    """State-6 units (NaN scores, binary64 path) inside otherwise single-bin granules: they go to slot 6 and leave
    the ranks of the units around them intact."""
    n = _EDGE_N
    rng = np.random.default_rng(66)
    for b in (0, 3, 5):
        cols = [c.astype(np.float64) for c in _columns_of_states(np.full(n, b))]
        cols = [np.where(c == ABSENT, NEG, c) for c in cols]
        hit = rng.choice(n, 40, replace=False)
        hit = np.concatenate([hit, [0, 1, 255, 256, 1023, 1024, 2047, 2048, n - 1]])
        cols[0][hit] = float("nan")
        cols[2][hit] = float("nan")
        flags = np.ones(n, dtype=np.uint8)
        if mode:
            flags[0::2] = 0
        bits = H.synth.pack_unit_bits(flags)
        want, want_counts = H.c_classify(mode, *cols, bits, NEG)
        want_idx, want_off = H.c_compact(mode, want)
        assert int(want_off[7] - want_off[6]) > 0
        code, counts = ctx.classify_f64(mode, *cols, bits, NEG)
        assert np.array_equal(code, want) and np.array_equal(counts, want_counts)
        idx, off, _ = ctx.compact(mode, code)
        assert np.array_equal(off, want_off) and np.array_equal(idx, want_idx)
        fcode, fidx, foff, fcounts = ctx.classify_compact(mode, *cols, bits, NEG)
        assert np.array_equal(fcode, want) and np.array_equal(fcounts, want_counts)
        assert np.array_equal(foff, want_off) and np.array_equal(fidx, want_idx)


@pytest.mark.parametrize("flavour", ["huge_op", "many_ops", "long_stretch", "huge_nm"])
def test_classify_cigar_guards_of_the_op_parallel_path(ctx, flavour):
    """Full workgroups (so the op-parallel path runs) with the inputs its wave-uniform guards are there for: an
    operation longer than 2^20, a record with more than 512 operations, a wave whose records own 1024 operations
    or more, and an NM that takes the score out of int32 (reported, not wrapped)."""
    n = 8192
    rng = np.random.default_rng(7)
    c1, c2 = _random_cigar(rng, n), _random_cigar(rng, n)
    xs = [np.where(rng.random(n) < 0.8, ABSENT, -rng.integers(0, 200, n)).astype(np.int32) for _ in range(2)]

    def splice(c, at, new_ops):
        off, ops = c["cig_off"].astype(np.int64), c["cig_oplen"]
        a, b = int(off[at]), int(off[at + 1])
        c["cig_oplen"] = np.concatenate([ops[:a], np.asarray(new_ops, dtype=np.uint32), ops[b:]])
        off[at + 1:] += len(new_ops) - (b - a)
        c["cig_off"] = off.astype(np.uint32)
        c["nm"][at] = 1
    if flavour == "huge_op":
        for at in (5, 700, 4099, 8191):
            splice(c1, at, [(10 << 4) | 0, ((2**27) << 4) | 4, ((2**20) << 4) | 2])   # 134217728S, 1048576D
            splice(c2, at + (0 if at == 8191 else 1), [((2**26) << 4) | 1])
    elif flavour == "many_ops":
        splice(c1, 1000, [(3 << 4) | (k % 9) for k in range(600)])
        splice(c2, 6000, [(2 << 4) | 1] * 513)
    elif flavour == "long_stretch":
        for at in range(2048, 2304):                                   # one wave's 256 records, 5 ops each
            splice(c1, at, [(7 << 4) | 4, (20 << 4) | 0, (1 << 4) | 1, (30 << 4) | 0, (2 << 4) | 2])
    else:
        c1["nm"][4321] = 2**30
    bits = H.synth.pack_unit_bits(rng.random(n) < 0.6)
    if flavour == "huge_nm":
        with pytest.raises(OverflowError):
            ctx.classify_cigar(1, c1["nm"], c1["cig_off"], c1["cig_oplen"], xs[0],
                               c2["nm"], c2["cig_off"], c2["cig_oplen"], xs[1], bits, ABSENT)
        return
    for mode in (0, 1, 2):
        code, counts = ctx.classify_cigar(mode, c1["nm"], c1["cig_off"], c1["cig_oplen"], xs[0],
                                          c2["nm"], c2["cig_off"], c2["cig_oplen"], xs[1], bits, ABSENT)
        want, want_counts = _oracle_cigar_classify(mode, c1, xs[0], c2, xs[1], bits, ABSENT)
        assert np.array_equal(code, want) and np.array_equal(counts, want_counts)
        _check_csr_dev(ctx, mode, c1, xs[0], c2, xs[1], bits, ABSENT)          # the same guards exist in K1c


def _packed_on_device(c, xs):
    import torch
    from xenomapper_amd import _ffi
    cnt, tile, ops = _ffi.cigar_pack(c["cig_off"], c["cig_oplen"])
    wcnt, wtile, wops = H.np_cigar_pack(c["cig_off"], c["cig_oplen"]) if c["nm"].shape[0] <= 200_000 else (cnt, tile, ops)
    assert np.array_equal(cnt, wcnt) and np.array_equal(tile, wtile) and np.array_equal(ops, wops)
    dev = torch.device("cuda:0")
    if ops.shape[0] == 0:
        ops = np.zeros(1, dtype=np.uint32)
    if cnt.shape[0] == 0:
        cnt = np.zeros(4, dtype=np.uint8)
    return [torch.from_numpy(a).to(dev) for a in (c["nm"], cnt, tile.view(np.int32), ops.view(np.int32), xs)]


def _check_packed_dev(ctx, mode, c1, xs1, c2, xs2, bits, mi, forms=("code", "bins4", "both")):
    """xm_classify_compact_cigar_packed_dev in its output forms against the oracle (CIGAR scores from the CSR
    columns, classify, compact)."""
    import torch
    from xenomapper_amd import _ffi
    n = c1["nm"].shape[0]
    dev = torch.device("cuda:0")
    want, want_counts = _oracle_cigar_classify(mode, c1, xs1, c2, xs2, bits, mi)
    want_idx, want_off = H.c_compact(mode, want)
    want_bins = np.full(n, 7, dtype=np.uint8)
    for b in range(7):
        want_bins[want_idx[int(want_off[b]):int(want_off[b + 1])]] = b
    d = _packed_on_device(c1, xs1) + _packed_on_device(c2, xs2) + [torch.from_numpy(bits.view(np.int64)).to(dev)]
    for form in forms:
        code = torch.full((n + 16,), 0xAA, dtype=torch.uint8, device=dev) if form != "bins4" else None
        bins4 = torch.full((_ffi.bins4_bytes(n),), 0xAA, dtype=torch.uint8, device=dev) if form != "code" else None
        idx = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
        off = torch.zeros(8, dtype=torch.int64, device=dev)
        counts = torch.zeros(64, dtype=torch.int64, device=dev)
        flag = torch.zeros(4, dtype=torch.int32, device=dev)
        ctx.classify_compact_cigar_packed_dev(mode, *d, mi, code, idx, off, counts, bins4=bins4, range_flag=flag)
        torch.cuda.synchronize()
        assert int(flag[0].item()) == 0
        if code is not None:
            assert np.array_equal(code[:n].cpu().numpy(), want), form
        if bins4 is not None:
            assert np.array_equal(_ffi.unpack_bins4(bins4.cpu().numpy(), n), want_bins), form
        assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts), form
        h_off = off.cpu().numpy().astype(np.uint64)
        assert np.array_equal(h_off, want_off), form
        assert np.array_equal(idx[:int(h_off[7])].cpu().numpy().view(np.uint32), want_idx), form
    if n and n <= 2_000_000:
        # the six-list form of the same call (xm_classify_place_cigar_packed_dev)
        lists = [torch.full((n,), -1, dtype=torch.int32, device=dev) for _ in range(6)]
        n_out = torch.zeros(8, dtype=torch.int64, device=dev)
        counts = torch.zeros(64, dtype=torch.int64, device=dev)
        bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev)
        ctx.classify_place_cigar_packed_dev(mode, *d, mi, lists, n_out, counts, bins4=bins4)
        torch.cuda.synchronize()
        h_n = n_out.cpu().numpy()
        for b in range(6):
            w = want_idx[int(want_off[b]):int(want_off[b + 1])]
            assert int(h_n[b]) == w.shape[0] and np.array_equal(lists[b][:w.shape[0]].cpu().numpy().view(np.uint32), w), b
        assert int(h_n[7]) == int(want_off[7])
        assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 255, 256, 257, 2047, 2048, 2049, 4097, 6145, 100_003])
def test_classify_cigar_packed_dev(ctx, n):
    """The packed-column kernel (K1p) through its device-resident entry point: all modes, a threshold, irregular
    unit masks (so that workgroups begin with units: the halo), every output form."""
    rng = np.random.default_rng(300 + n)
    c1, c2 = _random_cigar(rng, n, op_hi=16), _random_cigar(rng, n)
    xs = [np.where(rng.random(n) < 0.8, ABSENT, -rng.integers(0, 200, n)).astype(np.int32) for _ in range(2)]
    for mode, m in itertools.product((0, 1, 2), (NEG, -40.5)):
        flags = rng.random(n) < (0.55 if mode else 0.9)
        _check_packed_dev(ctx, mode, c1, xs[0], c2, xs[1], H.synth.pack_unit_bits(flags), H.floor_min_score(m))
    if n > 1:                                                       # strictly interleaved mates: the fast scatter path
        _check_packed_dev(ctx, 1, c1, xs[0], c2, xs[1], H.synth.interleaved_unit_bits(n), ABSENT)


@pytest.mark.parametrize("flavour", ["escapes", "escape_before_workgroup", "all_escaped_tile", "dense_254", "no_ops"])
def test_classify_cigar_packed_escapes(ctx, flavour):
    """Records with 255 operations or more (count byte 255 + trailer word) and the other shapes only the careful path
    of K1p serves: several escaped records in one tile, an escaped record right in front of a workgroup whose first
    record closes a unit (the halo reads the trailer), a tile of escaped records, records of 254 operations (the
    largest plain count; a tile of them is far beyond the 1024-op stretch), and no operations at all."""
    n = 3 * 2048 + 300
    rng = np.random.default_rng(11)
    c1, c2 = _random_cigar(rng, n, max_ops=4), _random_cigar(rng, n, max_ops=4)

    def set_ops(c, lengths):                                        # {record: number of ops}
        k = np.diff(c["cig_off"].astype(np.int64))
        for at, length in lengths.items():
            k[at] = length
        off = np.zeros(n + 1, dtype=np.uint32)
        np.cumsum(k, out=off[1:])
        c["cig_off"] = off
        c["cig_oplen"] = (rng.integers(1, 9, int(off[-1])).astype(np.uint32) << 4) | rng.integers(0, 9, int(off[-1])).astype(np.uint32)
        for at in lengths:
            c["nm"][at] = 2
    if flavour == "escapes":
        set_ops(c1, {3: 255, 4: 256, 200: 1000, 255: 300, 256: 255, 2047: 511, 4096: 70000, n - 1: 255})
        set_ops(c2, {3: 300, 1000: 255, 1001: 255, 1002: 255, n - 2: 400})
    elif flavour == "escape_before_workgroup":
        set_ops(c1, {2047: 260, 4095: 255})
        set_ops(c2, {2047: 3, 4095: 999, 6143: 255})
    elif flavour == "all_escaped_tile":
        set_ops(c1, {at: 255 + (at % 3) for at in range(2048, 2048 + 256)})
        set_ops(c2, {at: 256 for at in range(2048 + 128, 2048 + 384)})
    elif flavour == "dense_254":
        set_ops(c1, {at: 254 for at in range(512, 768)})
        set_ops(c2, {at: 254 for at in range(700, 710)})
    else:
        c1["cig_off"][:] = 0
        c1["cig_oplen"] = np.zeros(0, dtype=np.uint32)
    xs = [np.where(rng.random(n) < 0.8, ABSENT, -rng.integers(0, 200, n)).astype(np.int32) for _ in range(2)]
    flags = rng.random(n) < 0.6
    flags[[2048, 4096, 6144]] = True                                # the first record of every later workgroup closes a unit
    bits = H.synth.pack_unit_bits(flags)
    for mode in (0, 1, 2):
        _check_packed_dev(ctx, mode, c1, xs[0], c2, xs[1], bits, ABSENT, forms=("both",))
    # and through the host-buffer entry points (CSR in; they pack and run the same kernel)
    code, counts = ctx.classify_cigar(1, c1["nm"], c1["cig_off"], c1["cig_oplen"], xs[0],
                                      c2["nm"], c2["cig_off"], c2["cig_oplen"], xs[1], bits, ABSENT)
    want, want_counts = _oracle_cigar_classify(1, c1, xs[0], c2, xs[1], bits, ABSENT)
    assert np.array_equal(code, want) and np.array_equal(counts, want_counts)
    fcode, fidx, foff, fcounts = ctx.classify_compact_cigar(1, c1["nm"], c1["cig_off"], c1["cig_oplen"], xs[0],
                                                            c2["nm"], c2["cig_off"], c2["cig_oplen"], xs[1], bits, ABSENT,
                                                            want_code=False)
    want_idx, want_off = H.c_compact(1, want)
    assert fcode is None and np.array_equal(fcounts, want_counts)
    assert np.array_equal(foff, want_off) and np.array_equal(fidx, want_idx)


def test_classify_cigar_packed_inconsistent_columns_stay_in_bounds(ctx):
    """Columns that do not describe each other (random count bytes, a tile table that is not monotone or points past
    the op array, escape bytes without trailers): the results mean nothing, but the kernel neither faults nor writes
    outside its outputs -- guard words around every output stay intact and the unit bookkeeping stays consistent."""
    import torch
    from xenomapper_amd import _ffi
    n = 4 * 2048 + 77
    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")
    n_ops = 5000
    for trial in range(6):
        sp = []
        for _ in range(2):
            cnt = rng.integers(0, 256, n).astype(np.uint8) if trial % 2 == 0 else np.full(n, 255, dtype=np.uint8)
            tile = rng.integers(0, 2 * n_ops, _ffi.cigar_tiles(n) + 1).astype(np.uint32)
            if trial >= 3:
                tile.sort()
            tile[-1] = n_ops                                             # the length of the op array is honest
            ops = rng.integers(0, 2**32, n_ops, dtype=np.uint64).astype(np.uint32)
            nm = rng.integers(0, 5, n).astype(np.int32)
            xs_ = np.full(n, ABSENT, dtype=np.int32)
            sp += [torch.from_numpy(a).to(dev) for a in (nm, cnt, tile.view(np.int32), ops.view(np.int32), xs_)]
        bits = torch.from_numpy(H.synth.pack_unit_bits(rng.random(n) < 0.7).view(np.int64)).to(dev)
        guard = 64
        code = torch.full((n + 16 + guard,), 0x5A, dtype=torch.uint8, device=dev)
        bins4 = torch.full((_ffi.bins4_bytes(n) + guard,), 0x5A, dtype=torch.uint8, device=dev)
        idx = torch.full((n + guard,), 0x5A5A5A5A, dtype=torch.int32, device=dev)
        off = torch.zeros(8, dtype=torch.int64, device=dev)
        counts = torch.zeros(64, dtype=torch.int64, device=dev)
        flag = torch.zeros(4, dtype=torch.int32, device=dev)
        ctx.classify_compact_cigar_packed_dev(1, *sp, bits, ABSENT, code, idx, off, counts, bins4=bins4, range_flag=flag)
        torch.cuda.synchronize()
        h_off = off.cpu().numpy()
        assert int(h_off[7]) == int(counts.sum().item()) <= n and (np.diff(h_off) >= 0).all()
        assert bool((code[n + 16:] == 0x5A).all()) and bool((bins4[_ffi.bins4_bytes(n):] == 0x5A).all())
        assert bool((idx[n:] == 0x5A5A5A5A).all())


def test_host_buffer_calls_from_registered_memory(ctx):
    """xm_host_register / xm_host_unregister: the host-buffer entry points give the same answers from page-locked
    caller arrays (direct DMA) as from pageable ones, and again from the same arrays once they are unregistered."""
    n = 300_001
    rng = np.random.default_rng(9)
    cols = random_columns(rng, n)
    bits = H.synth.pack_unit_bits(rng.random(n) < 0.55)
    want_code, want_counts = H.c_classify(1, *cols, bits, ABSENT)
    want_idx, want_off = H.c_compact(1, want_code)
    from xenomapper_amd import _ffi
    before = _ffi.pinned_bytes()["registered"]
    with ctx.registered(*(cols + [bits])):
        assert _ffi.pinned_bytes()["registered"] == before + sum(a.nbytes for a in cols + [bits])
        for _ in range(2):
            code, idx, off, counts = ctx.classify_compact(1, *cols, bits, ABSENT)
            assert np.array_equal(code, want_code) and np.array_equal(counts, want_counts)
            assert np.array_equal(off, want_off) and np.array_equal(idx, want_idx)
        with pytest.raises(ValueError):                               # a second registration of the same memory is refused
            ctx.host_register(cols[0])
    assert _ffi.pinned_bytes()["registered"] == before                # nothing stays locked behind the block
    code, idx, off, counts = ctx.classify_compact(1, *cols, bits, ABSENT)
    assert np.array_equal(code, want_code) and np.array_equal(idx, want_idx)
    # locking fails half way (the third "array" is not memory the process owns): the two before it are unlocked again
    with pytest.raises(Exception):
        with ctx.registered(cols[0], cols[1], _FakeArray(0x10, 4096)):
            pass
    assert _ffi.pinned_bytes()["registered"] == before


class _FakeArray(object):
    """Quacks like a NumPy array for host_register: an address that is not mapped."""
    def __init__(self, address, nbytes):
        import ctypes

        class _C(object):
            data = address

            @staticmethod
            def data_as(_t):
                return ctypes.c_void_p(address)
        self.ctypes, self.nbytes = _C(), nbytes


def test_classify_cigar_range_error(ctx):
    big = np.array([((2**28 - 1) << 4) | 1] * 8, dtype=np.uint32)
    one = {"nm": np.array([0], np.int32), "off": np.array([0, 8], np.uint32)}
    none = np.array([ABSENT], np.int32)
    with pytest.raises(OverflowError):
        ctx.classify_cigar(0, one["nm"], one["off"], big, none, one["nm"], one["off"], big, none,
                           np.array([1], np.uint64), ABSENT)


@pytest.mark.parametrize("case", [c for c in H.golden("g3_end_to_end.json")["cases"]],
                         ids=lambda c: c["name"])
def test_g3_units_on_gpu(ctx, case):
    """Golden per-unit states recorded from the reference, reproduced by the kernels from columns
    (columns filled with the oracle's text-level scorers; the product's own column stripper is
    tested in test_host_api.py)."""
    import io
    t1, t2 = H.case_texts(case)
    s1, s2 = io.StringIO(t1), io.StringIO(t2)
    ORACLE.read_header(s1), ORACLE.read_header(s2)
    pairs = list(ORACLE.read_pairs(s1, s2, case["options"]["skip_repeated"]))
    scorer = {"get_tag": ORACLE.tag_score, "get_tag_with_ZS_as_XS": ORACLE.tag_score_zs,
              "get_cigarbased_AS_tag": ORACLE.cigar_score}[case["options"]["tag_func"]]
    n = len(pairs)
    cols = [np.array([scorer(p[f], tag=t) for p in pairs], dtype=np.float64) for f, t in
            ((0, "AS"), (0, "XS"), (1, "AS"), (1, "XS"))]
    names = [p[0][0] for p in pairs]
    mode = H.MODES[case["mode"]]
    if mode == 0:
        flags = np.ones(n, dtype=np.uint8)
    else:
        flags = np.array([0] + [int(names[i] == names[i - 1]) for i in range(1, n)], dtype=np.uint8)
    bits = H.synth.pack_unit_bits(flags)
    m = H.unnum(case["options"]["min_score"])
    icols = [np.where(c == NEG, ABSENT, c).astype(np.int32) for c in cols]
    code_i, counts_i = ctx.classify(mode, *icols, bits, H.floor_min_score(m))
    code_f, counts_f = ctx.classify_f64(mode, *cols, bits, m)
    assert np.array_equal(code_i, code_f) and np.array_equal(counts_i, counts_f)
    exp = case["expect"]
    unit_idx = np.flatnonzero(code_i != 0xFF)
    assert unit_idx.tolist() == exp["unit_index"]
    got_rev = "".join(str(int(c) & 7) for c in code_i[unit_idx])
    assert got_rev == exp["unit_rev"]
    if mode:
        assert "".join(str(int(c) >> 3) for c in code_i[unit_idx]) == exp["unit_fwd"]
    named = {}
    for c in np.flatnonzero(counts_i):
        key = H.STATES[c] if mode == 0 else H.STATES[c >> 3] + "|" + H.STATES[c & 7]
        named[key] = int(counts_i[c])
    assert named == exp["counts"]
    # bin sizes: lines of a bin's text = its header block (oracle) + units of the bin x records per unit, exactly
    idx, off, _ = ctx.compact(mode, code_i)
    per_unit = {0: (1, 1, 1, 1, 2, 1), 1: (2, 2, 2, 2, 4, 2), 2: (2, 2, 2, 2, 4, 2)}[mode]
    if case["options"]["header_sinks"] == "all":
        heads = [io.StringIO() for _ in range(6)]
        ORACLE.write_headers(io.StringIO(t1), io.StringIO(t2), heads)
        for b, name in enumerate(H.STATES):
            n_units = int(off[b + 1] - off[b])
            assert exp["bins"][name]["lines"] == heads[b].getvalue().count("\n") + n_units * per_unit[b], name


def test_device_entry_points_and_full_size(ctx):
    """BASELINE.json configs[1]: 50 M paired-end pairs (100 M records per species), HBM-resident,
    through the *_dev entry points; exact against the C oracle plus size-independent properties."""
    import torch
    from xenomapper_amd import _ffi
    n_pairs = 50_000_000
    cols = H.synth.score_columns(n_pairs, seed=2002)
    n = 2 * n_pairs
    dev = torch.device("cuda:0")
    d = {k: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) for k, v in cols.items()}
    code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    for mode in (_ffi.MODE_PE_LIBERAL, _ffi.MODE_PE_CONSERVATIVE):
        ctx.classify_dev(mode, d["as1"], d["xs1"], d["as2"], d["xs2"], d["unit_bits"], ABSENT, code)
        ctx.compact_dev(mode, code[:n], idx, off, counts)
        torch.cuda.synchronize()
        h_code = code[:n].cpu().numpy()
        h_off = off.cpu().numpy().astype(np.uint64)
        h_counts = counts.cpu().numpy().astype(np.uint64)
        h_idx = idx[:int(h_off[7])].cpu().numpy().view(np.uint32)
        want_code, want_counts = H.c_classify(mode, cols["as1"], cols["xs1"], cols["as2"], cols["xs2"],
                                              cols["unit_bits"], ABSENT)
        assert np.array_equal(h_code, want_code)
        assert np.array_equal(h_counts, want_counts)
        want_idx, want_off = H.c_compact(mode, want_code)
        assert np.array_equal(h_off, want_off)
        assert np.array_equal(h_idx, want_idx)
        # size-independent properties
        assert int(h_off[7]) == n_pairs == int(h_counts.sum())
        assert np.array_equal(np.sort(h_idx), np.arange(1, n, 2, dtype=np.uint32))      # a permutation of the units
        for b in range(6):
            seg = h_idx[int(h_off[b]):int(h_off[b + 1])]
            assert (np.diff(seg.astype(np.int64)) > 0).all()                              # stable within a bin
        # the fused call, in its three output forms: category bytes only, compact stream only (what bench.py times), both
        want_bins = _bins_of_codes(mode, want_code)
        bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev)
        for use_code, use_bins4 in ((True, False), (False, True), (True, True)):
            code.fill_(0x55), idx.fill_(-1), off.zero_(), counts.zero_(), bins4.fill_(0x55)
            ctx.classify_compact_dev(mode, d["as1"], d["xs1"], d["as2"], d["xs2"], d["unit_bits"], ABSENT,
                                     code if use_code else None, idx, off, counts, bins4=bins4 if use_bins4 else None)
            torch.cuda.synchronize()
            assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
            assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
            assert np.array_equal(idx[:n_pairs].cpu().numpy().view(np.uint32), want_idx)
            if use_code:
                assert np.array_equal(code[:n].cpu().numpy(), want_code)
            if use_bins4:
                assert np.array_equal(_ffi.unpack_bins4(bins4.cpu().numpy(), n), want_bins)


def _bins_of_codes(mode, code):
    """Output bin per record (6 = unit holding a state 6, 7 = closes no unit) as the oracle's split assigns it: the
    content of the compact category stream."""
    idx, off = H.c_compact(mode, code)
    bins = np.full(code.shape[0], 7, dtype=np.uint8)
    for b in range(7):
        bins[idx[int(off[b]):int(off[b + 1])]] = b
    return bins


@pytest.mark.parametrize("n", [1, 5, 2047, 2048, 2049, 8191, 70_001])
def test_fused_device_forms_small(ctx, n):
    """xm_classify_compact_dev / _f64_dev with category bytes, with the compact stream, with both -- every mode, ragged
    sizes around the 2048-record granule, irregular unit masks, and NaN (state 6) on the binary64 side."""
    import torch
    from xenomapper_amd import _ffi
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n)
    icols = random_columns(rng, n)
    fvals = np.array([NEG, NEG, float("nan"), -1.5, 0.0, 0.0, 1.0, 2.5, 3.0])
    fcols = [fvals[rng.integers(0, len(fvals), n)] for _ in range(4)]
    code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    for mode in (0, 1, 2):
        flags = rng.random(n) < (0.55 if mode else 0.9)
        bits = H.synth.pack_unit_bits(flags)
        d_bits = torch.from_numpy(bits.view(np.int64)).to(dev)
        for cols, m in ((icols, ABSENT), (fcols, 0.5)):
            want_code, want_counts = H.c_classify(mode, *cols, bits, m)
            want_idx, want_off = H.c_compact(mode, want_code)
            want_bins = _bins_of_codes(mode, want_code)
            d_cols = [torch.from_numpy(np.ascontiguousarray(c)).to(dev) for c in cols]
            for use_code, use_bins4 in ((True, False), (False, True), (True, True)):
                code.fill_(0x55), idx.fill_(-1), off.zero_(), counts.zero_(), bins4.fill_(0x55)
                ctx.classify_compact_dev(mode, *d_cols, d_bits, m, code if use_code else None, idx, off, counts,
                                         bins4=bins4 if use_bins4 else None)
                torch.cuda.synchronize()
                assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
                assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
                assert np.array_equal(idx[:int(want_off[7])].cpu().numpy().view(np.uint32), want_idx)
                if use_code:
                    assert np.array_equal(code[:n].cpu().numpy(), want_code)
                if use_bins4:
                    assert np.array_equal(_ffi.unpack_bins4(bins4.cpu().numpy(), n), want_bins)
    with pytest.raises(ValueError):                       # neither form given
        ctx.classify_compact_dev(1, *d_cols, d_bits, 0.5, None, idx, off, counts, bins4=None)


def test_full_size_cfg3_cigar(ctx):
    """BASELINE.json configs[2]: 50 M paired-end pairs on the --cigar_scores path (no AS tag; NM + CIGAR),
    device-resident, fused K3+K1 then K2; exact against the C oracle."""
    import torch
    from xenomapper_amd import _ffi
    n_pairs = 50_000_000
    n = 2 * n_pairs
    c1 = H.synth.cigar_columns(n, seed=3003)
    c2 = H.synth.cigar_columns(n, seed=3004, mapped_p=0.3)
    rng = np.random.default_rng(3005)
    xs1 = np.where(rng.random(n) < 0.95, ABSENT, -rng.integers(0, 40, n)).astype(np.int32)
    xs2 = np.full(n, ABSENT, dtype=np.int32)
    bits = H.synth.interleaved_unit_bits(n)
    dev = torch.device("cuda:0")

    def up(a):
        return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else (a.view(np.int64) if a.dtype == np.uint64 else a)).to(dev)
    d = [up(x) for x in (c1["nm"], c1["cig_off"], c1["cig_oplen"], xs1, c2["nm"], c2["cig_off"], c2["cig_oplen"], xs2, bits)]
    code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    flag = torch.zeros(4, dtype=torch.int32, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    mode = _ffi.MODE_PE_LIBERAL
    ctx.classify_cigar_dev(mode, *d, ABSENT, code, range_flag=flag)
    ctx.compact_dev(mode, code[:n], idx, off, counts)
    torch.cuda.synchronize()
    assert int(flag[0].item()) == 0
    want, want_counts = _oracle_cigar_classify(mode, c1, xs1, c2, xs2, bits, ABSENT)
    assert np.array_equal(code[:n].cpu().numpy(), want)
    assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
    want_idx, want_off = H.c_compact(mode, want)
    h_off = off.cpu().numpy().astype(np.uint64)
    assert np.array_equal(h_off, want_off)
    assert np.array_equal(idx[:int(h_off[7])].cpu().numpy().view(np.uint32), want_idx)
    k_bar = (c1["cig_oplen"].shape[0] + c2["cig_oplen"].shape[0]) / (2.0 * n)
    print("cfg3 mean CIGAR ops per record: %.3f -> algorithmic %.1f B/pair" % (k_bar, 4 * (12 + 4 * k_bar) + 6))


def test_full_size_cfg3_cigar_packed(ctx):
    """BASELINE.json configs[2] as bench.py --workload cfg3 runs it: 50 M pairs, packed CIGAR columns, ONE
    xm_classify_compact_cigar_packed_dev call (K1p counts and writes the compact stream); exact against the C oracle."""
    import torch
    from xenomapper_amd import _ffi
    n_pairs = 50_000_000
    n = 2 * n_pairs
    c1 = H.synth.cigar_columns(n, seed=3003)
    c2 = H.synth.cigar_columns(n, seed=3004, mapped_p=0.3)
    rng = np.random.default_rng(3005)
    xs1 = np.where(rng.random(n) < 0.95, ABSENT, -rng.integers(0, 40, n)).astype(np.int32)
    xs2 = np.full(n, ABSENT, dtype=np.int32)
    bits = H.synth.interleaved_unit_bits(n)
    _check_packed_dev(ctx, _ffi.MODE_PE_LIBERAL, c1, xs1, c2, xs2, bits, ABSENT, forms=("bins4",))
    k_bar = (c1["cig_oplen"].shape[0] + c2["cig_oplen"].shape[0]) / (2.0 * n)
    print("cfg3 mean CIGAR ops per record: %.3f -> algorithmic %.2f B/pair (classify), + 5 (compact)"
          % (k_bar, 4 * (9 + 1 / 64 + 4 * k_bar) + 1))


def test_full_size_cfg5_zs_conservative(ctx):
    """BASELINE.json configs[4]: 50 M pairs, HISAT-style scores (AS in [-90, 0], second-best from ZS incl. the
    AS = 0 / ZS = 0 records that exercise the `not XS` quirk), --conservative; exact against the C oracle."""
    import torch
    from xenomapper_amd import _ffi
    n_pairs = 50_000_000
    n = 2 * n_pairs
    dev = torch.device("cuda:0")
    cols = H.synth.score_columns_torch(n_pairs, seed=5005, device=dev, profile="hisat")
    code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    host = {k: v.cpu().numpy() for k, v in cols.items()}
    host["unit_bits"] = host["unit_bits"].view(np.uint64)
    assert int(((host["as1"] == 0) & (host["xs1"] == 0)).sum()) > 1000          # the quirk is exercised
    for mode, m in ((_ffi.MODE_PE_CONSERVATIVE, ABSENT), (_ffi.MODE_PE_CONSERVATIVE, -30)):
        ctx.classify_dev(mode, cols["as1"], cols["xs1"], cols["as2"], cols["xs2"], cols["unit_bits"], m, code)
        ctx.compact_dev(mode, code[:n], idx, off, counts)
        torch.cuda.synchronize()
        want, want_counts = H.c_classify(mode, host["as1"], host["xs1"], host["as2"], host["xs2"], host["unit_bits"], m)
        want_idx, want_off = H.c_compact(mode, want)
        assert np.array_equal(code[:n].cpu().numpy(), want)
        assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
        h_off = off.cpu().numpy().astype(np.uint64)
        assert np.array_equal(h_off, want_off)
        assert np.array_equal(idx[:int(h_off[7])].cpu().numpy().view(np.uint32), want_idx)
        # the FUSED call on the same columns (what bench.py --workload cfg5 and the product run): compact stream only,
        # then the six-list form of it
        bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev)
        idx.zero_(); off.zero_(); counts.zero_()
        ctx.classify_compact_dev(mode, cols["as1"], cols["xs1"], cols["as2"], cols["xs2"], cols["unit_bits"], m, None,
                                 idx, off, counts, bins4=bins4)
        torch.cuda.synchronize()
        assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
        assert np.array_equal(idx[:int(h_off[7])].cpu().numpy().view(np.uint32), want_idx)
        lists = [torch.zeros(n_pairs, dtype=torch.int32, device=dev) for _ in range(6)]
        n_out = torch.zeros(8, dtype=torch.int64, device=dev)
        counts.zero_()
        ctx.classify_place_dev(mode, cols["as1"], cols["xs1"], cols["as2"], cols["xs2"], cols["unit_bits"], m, lists, n_out,
                               counts, bins4=bins4)
        torch.cuda.synchronize()
        _check_lists(lists, n_out.cpu().numpy(), want_idx, want_off)
        assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
        del lists


def _check_lists(lists, n_out, want_idx, want_off):
    """Six (or seven) device lists + n_out[8] against the oracle's packed split."""
    for b, lst in enumerate(lists):
        want = want_idx[int(want_off[b]):int(want_off[b + 1])]
        assert int(n_out[b]) == want.shape[0], (b, n_out.tolist(), want_off.tolist())
        assert np.array_equal(lst[:want.shape[0]].cpu().numpy().view(np.uint32), want), b
    assert int(n_out[7]) == int(want_off[7])
    assert int(n_out[6]) == int(want_off[7] - want_off[6])


def test_six_list_form_on_the_device_every_mode_and_layout(ctx):
    """xm_classify_place_dev / _f64_dev (idx_out[6], n_out[6]) against the oracle: every loop, compact
    stream and category bytes as the scatter's input, interleaved and irregular unit masks, single-end input with
    staged granules, a state-6 list for binary64 columns, and lists too short for their bin (truncated, never
    overrun: guard words behind every list)."""
    import torch
    from xenomapper_amd import _ffi
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(606)
    for n, mode, p_unit in ((100_003, 0, 1.0), (100_003, 0, 0.8), (250_001, 1, None), (250_001, 2, 0.5), (6_000, 1, 0.3),
                            (2_049, 2, None), (1, 0, 1.0), (5_000_000, 1, None)):
        cols = random_columns(rng, n)
        if p_unit is None:
            bits = H.synth.interleaved_unit_bits(n)
        else:
            bits = H.synth.pack_unit_bits(rng.random(n) < p_unit)
        want_code, want_counts = H.c_classify(mode, *cols, bits, -2)
        want_idx, want_off = H.c_compact(mode, want_code)
        d_cols = [torch.from_numpy(c).to(dev) for c in cols]
        d_bits = torch.from_numpy(bits.view(np.int64)).to(dev)
        for form in ("bins4", "code"):
            cap = n
            guard = 0x7FFFFFF0
            lists = [torch.full((cap + 8,), guard, dtype=torch.int32, device=dev) for _ in range(6)]
            n_out = torch.zeros(8, dtype=torch.int64, device=dev)
            counts = torch.zeros(64, dtype=torch.int64, device=dev)
            code = torch.empty(n + 16, dtype=torch.uint8, device=dev) if form == "code" else None
            bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev) if form == "bins4" else None
            ctx.classify_place_dev(mode, *d_cols, d_bits, -2, lists, n_out, counts, code_out=code, bins4=bins4, capacity=cap)
            torch.cuda.synchronize()
            _check_lists(lists, n_out.cpu().numpy(), want_idx, want_off)
            assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
            for b, lst in enumerate(lists):
                k = int(want_off[b + 1] - want_off[b])
                assert bool((lst[k:] == guard).all()), (n, mode, form, b)
        # lists shorter than their bins: truncated at the capacity, lengths still reported in full
        cap = max(1, int((want_off[1] - want_off[0]) // 2))
        lists = [torch.full((cap + 8,), 0x7FFFFFF0, dtype=torch.int32, device=dev) for _ in range(6)]
        n_out = torch.zeros(8, dtype=torch.int64, device=dev)
        counts = torch.zeros(64, dtype=torch.int64, device=dev)
        bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev)
        ctx.classify_place_dev(mode, *d_cols, d_bits, -2, lists, n_out, counts, bins4=bins4, capacity=cap)
        torch.cuda.synchronize()
        h_n = n_out.cpu().numpy()
        for b, lst in enumerate(lists):
            want = want_idx[int(want_off[b]):int(want_off[b + 1])]
            assert int(h_n[b]) == want.shape[0]
            k = min(cap, want.shape[0])
            assert np.array_equal(lst[:k].cpu().numpy().view(np.uint32), want[:k])
            assert bool((lst[cap:] == 0x7FFFFFF0).all()), (n, mode, b)
    # binary64 columns with NaN: the seventh list
    nan = float("nan")
    n = 4099
    f = [rng.integers(-5, 6, n).astype(np.float64) for _ in range(4)]
    f[0][rng.random(n) < 0.1] = nan
    f[2][rng.random(n) < 0.05] = nan
    bits = H.synth.pack_unit_bits(rng.random(n) < 0.7)
    for mode in (0, 1, 2):
        want_code, want_counts = H.c_classify(mode, *f, bits, 0.5)
        want_idx, want_off = H.c_compact(mode, want_code)
        assert want_off[7] > want_off[6]
        d_cols = [torch.from_numpy(c).to(dev) for c in f]
        d_bits = torch.from_numpy(bits.view(np.int64)).to(dev)
        for with_list6 in (True, False):
            lists = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(7)]
            n_out = torch.zeros(8, dtype=torch.int64, device=dev)
            counts = torch.zeros(64, dtype=torch.int64, device=dev)
            code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
            ctx.classify_place_dev(mode, *d_cols, d_bits, 0.5, lists[:6], n_out, counts, code_out=code,
                                   list_state6=lists[6] if with_list6 else None)
            torch.cuda.synchronize()
            _check_lists(lists if with_list6 else lists[:6], n_out.cpu().numpy(), want_idx, want_off)
            assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
            assert np.array_equal(code[:n].cpu().numpy(), want_code)


def test_six_lists_in_chunks_equal_the_one_pass_lists(ctx, monkeypatch):
    """xm_classify_place*_dev goes over large inputs a chunk at a time (K1, K2b, K2c per chunk, running totals carried
    through the part totals: xm_api.hip place_steps).  With the chunk cut down to one part (2 M records) a 9 M-record input
    takes five chunks, the last one partial and ending off a part boundary: lists, lengths, category_counts, category
    bytes and the workspace's between-calls state must be those of the one-pass order (= the oracle's), for every loop, both
    scatter inputs, single-end staged granules, irregular unit masks and binary64 columns with the state-6 list; and a call
    in one pass right after a chunked one must find the workspace as it expects it."""
    import torch
    from xenomapper_amd import _ffi
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4242)
    n = 9_000_001
    guard = 0x7FFFFFF0
    for mode, p_unit, form in ((1, None, "bins4"), (2, 0.55, "bins4"), (0, 0.9, "bins4"), (1, 0.4, "code")):
        cols = random_columns(rng, n)
        bits = H.synth.interleaved_unit_bits(n) if p_unit is None else H.synth.pack_unit_bits(rng.random(n) < p_unit)
        want_code, want_counts = H.c_classify(mode, *cols, bits, -2)
        want_idx, want_off = H.c_compact(mode, want_code)
        d_cols = [torch.from_numpy(c).to(dev) for c in cols]
        d_bits = torch.from_numpy(bits.view(np.int64)).to(dev)
        for parts in ("1", "0"):                                    # chunked, then the one-pass order on the same context
            monkeypatch.setenv("XM_PLACE_CHUNK_PARTS", parts)
            lists = [torch.full((n + 8,), guard, dtype=torch.int32, device=dev) for _ in range(6)]
            n_out = torch.full((8,), -1, dtype=torch.int64, device=dev)
            counts = torch.full((64,), -1, dtype=torch.int64, device=dev)
            code = torch.empty(n + 16, dtype=torch.uint8, device=dev) if form == "code" else None
            bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev) if form == "bins4" else None
            ctx.classify_place_dev(mode, *d_cols, d_bits, -2, lists, n_out, counts, code_out=code, bins4=bins4, capacity=n)
            torch.cuda.synchronize()
            _check_lists(lists, n_out.cpu().numpy(), want_idx, want_off)
            assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts), (mode, parts)
            for b, lst in enumerate(lists):
                k = int(want_off[b + 1] - want_off[b])
                assert bool((lst[k:] == guard).all()), (mode, parts, b)
            if code is not None:
                assert np.array_equal(code[:n].cpu().numpy(), want_code)
            assert ctx.workspace_is_clean(), (mode, parts)
            del lists
        del d_cols, d_bits
    # binary64 with NaN: the seventh list, chunked
    nan = float("nan")
    f = [rng.integers(-5, 6, n).astype(np.float64) for _ in range(4)]
    f[0][rng.random(n) < 0.1] = nan
    f[2][rng.random(n) < 0.05] = nan
    bits = H.synth.pack_unit_bits(rng.random(n) < 0.7)
    want_code, want_counts = H.c_classify(2, *f, bits, 0.5)
    want_idx, want_off = H.c_compact(2, want_code)
    d_cols = [torch.from_numpy(c).to(dev) for c in f]
    d_bits = torch.from_numpy(bits.view(np.int64)).to(dev)
    monkeypatch.setenv("XM_PLACE_CHUNK_PARTS", "1")
    lists = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(7)]
    n_out = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev)
    ctx.classify_place_dev(2, *d_cols, d_bits, 0.5, lists[:6], n_out, counts, bins4=bins4, list_state6=lists[6])
    torch.cuda.synchronize()
    _check_lists(lists, n_out.cpu().numpy(), want_idx, want_off)
    assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
    assert ctx.workspace_is_clean()


def test_scatter_runs_of_granules_equal_one_granule_per_wave(ctx, monkeypatch):
    """K2c's second launch shape (scatter_kernel MULTI: a wave places a RUN of consecutive granules, the per-bin places carried
    from granule to granule) is what inputs beyond XM_SCATTER_WAVES granules take -- 2^31 records in the shipped build, which
    no test reaches.  The environment variable of the same name lowers the wave count: with 7 and 100 waves a 1.2 M-record
    input gives runs of 84 and 6 granules (the last run short, the last granule partial).  Packed output and six lists, every
    loop, compact stream and category bytes, staged single-end granules beside unstaged ones: all equal to the oracle's
    split, as the one-granule shape is, with guard words behind the lists."""
    import torch
    from xenomapper_amd import _ffi
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(777)
    n = 1_200_003
    guard = 0x7FFFFFF0
    for mode, p_unit in ((1, None), (2, 0.55), (0, 1.0), (0, 0.8), (1, 0.3)):
        cols = random_columns(rng, n)
        if p_unit is None:
            bits = H.synth.interleaved_unit_bits(n)
        else:
            flags = rng.random(n) < p_unit
            if mode == 0 and p_unit < 1.0:
                flags[: n // 2] = True                              # staged granules in front, unstaged ones behind
            bits = H.synth.pack_unit_bits(flags)
        want_code, want_counts = H.c_classify(mode, *cols, bits, -2)
        want_idx, want_off = H.c_compact(mode, want_code)
        d_cols = [torch.from_numpy(c).to(dev) for c in cols]
        d_bits = torch.from_numpy(bits.view(np.int64)).to(dev)
        for waves in ("7", "100", ""):
            monkeypatch.setenv("XM_SCATTER_WAVES", waves)
            for form in ("bins4", "code"):
                code = torch.empty(n + 16, dtype=torch.uint8, device=dev) if form == "code" else None
                bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev) if form == "bins4" else None
                idx = torch.full((n + 8,), guard, dtype=torch.int32, device=dev)
                off = torch.zeros(8, dtype=torch.int64, device=dev)
                counts = torch.zeros(64, dtype=torch.int64, device=dev)
                ctx.classify_compact_dev(mode, *d_cols, d_bits, -2, code, idx, off, counts, bins4=bins4)
                torch.cuda.synchronize()
                assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off), (mode, waves, form)
                assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
                assert np.array_equal(idx[:int(want_off[7])].cpu().numpy().view(np.uint32), want_idx), (mode, waves, form)
                assert bool((idx[int(want_off[7]):] == guard).all())
                lists = [torch.full((n + 8,), guard, dtype=torch.int32, device=dev) for _ in range(6)]
                n_out = torch.zeros(8, dtype=torch.int64, device=dev)
                counts.zero_()
                ctx.classify_place_dev(mode, *d_cols, d_bits, -2, lists, n_out, counts, code_out=code, bins4=bins4, capacity=n)
                torch.cuda.synchronize()
                _check_lists(lists, n_out.cpu().numpy(), want_idx, want_off)
                for b, lst in enumerate(lists):
                    assert bool((lst[int(want_off[b + 1] - want_off[b]):] == guard).all()), (mode, waves, form, b)
                assert ctx.workspace_is_clean()


def test_two_streams_on_one_context_are_ordered_by_the_library(ctx):
    """The compaction workspace belongs to the context (include/xenomapper_hip.h): fused calls issued alternately on two
    streams, without any synchronisation by the caller, must still give each call its own right answer -- the library puts
    a call on another stream behind the previous one."""
    import torch
    from xenomapper_amd import _ffi
    dev = torch.device("cuda:0")
    n = 3_000_001
    rng = np.random.default_rng(9090)
    jobs = []
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for k in range(6):
        cols = random_columns(rng, n - 1000 * k)
        nk = cols[0].shape[0]
        mode = k % 3
        bits = H.synth.pack_unit_bits(rng.random(nk) < (0.9 if mode == 0 else 0.5))
        d = dict(cols=cols, bits=bits, mode=mode, n=nk,
                 d_cols=[torch.from_numpy(c).to(dev) for c in cols], d_bits=torch.from_numpy(bits.view(np.int64)).to(dev),
                 bins4=torch.empty(_ffi.bins4_bytes(nk), dtype=torch.uint8, device=dev),
                 idx=torch.zeros(nk, dtype=torch.int32, device=dev), off=torch.zeros(8, dtype=torch.int64, device=dev),
                 counts=torch.zeros(64, dtype=torch.int64, device=dev))
        jobs.append(d)
    torch.cuda.synchronize()
    for rep in range(3):
        for k, d in enumerate(jobs):
            st = streams[k % 2]
            ctx.classify_compact_dev(d["mode"], *d["d_cols"], d["d_bits"], -1, None, d["idx"], d["off"], d["counts"],
                                     bins4=d["bins4"], stream=st)
    torch.cuda.synchronize()
    for d in jobs:
        want_code, want_counts = H.c_classify(d["mode"], *d["cols"], d["bits"], -1)
        want_idx, want_off = H.c_compact(d["mode"], want_code)
        assert np.array_equal(d["counts"].cpu().numpy().astype(np.uint64), want_counts)
        assert np.array_equal(d["off"].cpu().numpy().astype(np.uint64), want_off)
        assert np.array_equal(d["idx"][:int(want_off[7])].cpu().numpy().view(np.uint32), want_idx)


def test_counting_workspace_is_zero_between_calls_whatever_the_sequence(ctx):
    """The replicated category_counts and the per-part bin totals are added into by every counting kernel and consumed by
    the scan / scatter (include/xenomapper_hip.h): after any mix of calls -- counts only, fused compaction, stand-alone
    compaction, the six-list form, CIGAR columns, a small input after a large one, an empty one -- they must be all zero
    again, and the results right."""
    rng = np.random.default_rng(515)
    assert ctx.workspace_is_clean()
    for n, mode in ((3_000_001, 1), (5, 2), (70_000, 0), (0, 1), (2_200_000, 2), (2049, 1)):
        cols = random_columns(rng, n)
        bits = H.synth.pack_unit_bits(rng.random(n) < 0.6) if n else np.zeros(1, dtype=np.uint64)
        want_code, want_counts = H.c_classify(mode, *cols, bits, -1)
        want_idx, want_off = H.c_compact(mode, want_code)
        code, counts = ctx.classify(mode, *cols, bits, -1)                    # counting K1 + scan only
        assert np.array_equal(code, want_code) and np.array_equal(counts, want_counts)
        assert ctx.workspace_is_clean()
        _, idx, off, counts = ctx.classify_compact(mode, *cols, bits, -1, want_code=False)
        assert np.array_equal(idx, want_idx) and np.array_equal(off, want_off) and np.array_equal(counts, want_counts)
        assert ctx.workspace_is_clean()
        idx, off, counts = ctx.compact(mode, want_code)                       # K2a counts from memory
        assert np.array_equal(idx, want_idx) and np.array_equal(counts, want_counts)
        assert ctx.workspace_is_clean()
        _, lists, n_out, counts = ctx.classify_place(mode, *cols, bits, -1, want_code=False)
        assert int(n_out[7]) == int(want_off[7]) and np.array_equal(counts, want_counts)
        assert ctx.workspace_is_clean()
    c1 = H.synth.cigar_columns(300_001, seed=11)
    c2 = H.synth.cigar_columns(300_001, seed=12, mapped_p=0.4)
    xs = np.full(300_001, ABSENT, dtype=np.int32)
    bits = H.synth.interleaved_unit_bits(300_001)
    ctx.classify_compact_cigar(1, c1["nm"], c1["cig_off"], c1["cig_oplen"], xs, c2["nm"], c2["cig_off"], c2["cig_oplen"], xs, bits, ABSENT)
    assert ctx.workspace_is_clean()
    ctx.classify_cigar(1, c1["nm"], c1["cig_off"], c1["cig_oplen"], xs, c2["nm"], c2["cig_off"], c2["cig_oplen"], xs, bits, ABSENT)
    assert ctx.workspace_is_clean()


def test_large_batch_64bit_offsets(ctx):
    """300 M records per species (column byte offsets beyond 2^32, record indices beyond 2^28), ragged tail;
    exact against the C oracle."""
    import torch
    from xenomapper_amd import _ffi
    n_pairs = 150_000_001
    n = 2 * n_pairs
    dev = torch.device("cuda:0")
    cols = H.synth.score_columns_torch(n_pairs, seed=77, device=dev)
    # break the strict interleave in a few places, also far beyond 2^31 bytes into the columns
    bits = cols["unit_bits"].clone()
    for word in (5, 1_000_003, (n // 64) - 2):
        bits[word] = 0x0F0F00FF00FF0F0F
    code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    mode = _ffi.MODE_PE_CONSERVATIVE
    ctx.classify_dev(mode, cols["as1"], cols["xs1"], cols["as2"], cols["xs2"], bits, 100, code)
    ctx.compact_dev(mode, code[:n], idx, off, counts)
    torch.cuda.synchronize()
    host = {k: v.cpu().numpy() for k, v in cols.items()}
    h_bits = bits.cpu().numpy().view(np.uint64)
    want, want_counts = H.c_classify(mode, host["as1"], host["xs1"], host["as2"], host["xs2"], h_bits, 100)
    assert np.array_equal(code[:n].cpu().numpy(), want)
    assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
    want_idx, want_off = H.c_compact(mode, want)
    h_off = off.cpu().numpy().astype(np.uint64)
    assert np.array_equal(h_off, want_off)
    assert np.array_equal(idx[:int(h_off[7])].cpu().numpy().view(np.uint32), want_idx)


@pytest.mark.parametrize("mode", [0, 1])
def test_beyond_2_30_records_the_wide_scatter(ctx, mode):
    """More than 2^30 records per species in one call: byte offsets into the index lists pass 2^32 (single-end: more
    than 2^30 units), so the scatter runs its 64-bit-addressing instantiation.  The input is periodic -- record i
    (single-end) / pair k (paired, both mates) holds state i % 6 / k % 6 -- so every bin list is an arithmetic
    progression and the whole result is checked exactly on the device, chunk by chunk, without a host oracle pass."""
    import torch
    from xenomapper_amd import _ffi
    dev = torch.device("cuda:0")
    n = 12 * ((1 << 30) // 12 + 4099)                                 # > 2^30, a multiple of 12, not of 2048
    assert n % 12 == 0 and n > (1 << 30) and n % 2048
    table = torch.tensor([_REAL[k] for k in range(6)], dtype=torch.int32, device=dev)
    cols = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4)]
    step = 6 * (1 << 24)
    for a in range(0, n, step):
        z = min(n, a + step)
        i = torch.arange(a, z, device=dev, dtype=torch.int64)
        s = (i % 6) if mode == 0 else ((i // 2) % 6)
        for j in range(4):
            cols[j][a:z] = table[s, j]
        del i, s
    words = (n + 63) // 64
    bits = torch.full((words,), -1 if mode == 0 else -0x5555555555555556, dtype=torch.int64, device=dev)   # all / odd records
    if n % 64:
        bits[-1] &= (1 << (n % 64)) - 1
    code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev)
    ctx.classify_compact_dev(mode, *cols, bits, _ffi.ABSENT, code, idx, off, counts, bins4=bins4)     # K2c reads bins4
    torch.cuda.synchronize()
    units = n if mode == 0 else n // 2
    per_bin = units // 6
    assert off.cpu().tolist() == [b * per_bin for b in range(6)] + [units, units]
    want_counts = torch.zeros(64, dtype=torch.int64)
    for b in range(6):
        want_counts[b if mode == 0 else b * 9] = per_bin
    assert torch.equal(counts.cpu(), want_counts)
    stride, first = (6, 0) if mode == 0 else (12, 1)                  # bin b: records b, b+6, ... / 2b+1, 2b+13, ...
    chunk = 1 << 26
    for b in range(6):
        lo = b * per_bin
        for a in range(0, per_bin, chunk):
            z = min(per_bin, a + chunk)
            want = torch.arange(a, z, device=dev, dtype=torch.int64) * stride + (b if mode == 0 else 2 * b + first)
            got = idx[lo + a:lo + z].to(torch.int64) & 0xFFFFFFFF
            assert torch.equal(got, want), (b, a)
            del want, got
    # the stand-alone compaction of the same category bytes gives the same lists
    idx2 = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.compact_dev(mode, code[:n], idx2, off, counts)
    torch.cuda.synchronize()
    assert torch.equal(idx2[:units], idx[:units]) and torch.equal(counts.cpu(), want_counts)


def test_fused_call_is_graph_capturable(ctx):
    """The *_dev entry points only enqueue work: one fused call captured into a HIP graph and replayed on fresh inputs
    gives what the eager call gives (include/xenomapper_hip.h: "graph-capturable")."""
    import torch
    from xenomapper_amd import _ffi
    dev = torch.device("cuda:0")
    n = 300_001
    rng = np.random.default_rng(4242)
    mode = 2
    bits = H.synth.pack_unit_bits(rng.random(n) < 0.5)
    d_bits = torch.from_numpy(bits.view(np.int64)).to(dev)
    d_cols = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4)]
    code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                   # warm-up outside the capture, as torch asks for
        ctx.classify_compact_dev(mode, *d_cols, d_bits, -3, code, idx, off, counts, bins4=bins4)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        ctx.classify_compact_dev(mode, *d_cols, d_bits, -3, code, idx, off, counts, bins4=bins4)
    for seed in (1, 2, 3):
        cols = random_columns(np.random.default_rng(seed), n)
        for d, c in zip(d_cols, cols):
            d.copy_(torch.from_numpy(c))
        graph.replay()
        torch.cuda.synchronize()
        want_code, want_counts = H.c_classify(mode, *cols, bits, -3)
        want_idx, want_off = H.c_compact(mode, want_code)
        assert np.array_equal(code[:n].cpu().numpy(), want_code)
        assert np.array_equal(counts.cpu().numpy().astype(np.uint64), want_counts)
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
        assert np.array_equal(idx[:int(want_off[7])].cpu().numpy().view(np.uint32), want_idx)


def test_c_example_through_the_c_abi(tmp_path):
    """examples/classify_pairs.c (plain C99, no Python, no torch in the process): one xm_classify_compact call on three
    read pairs; its printout against the oracle's result for the same columns."""
    import subprocess
    from tests.test_host_cpu import _build_c_example
    proc = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stderr
    as1 = np.array([200, 198, 80, 90, ABSENT, ABSENT], np.int32)
    xs1 = np.array([150, ABSENT, ABSENT, 70, ABSENT, ABSENT], np.int32)
    as2 = np.array([120, ABSENT, 190, 188, ABSENT, ABSENT], np.int32)
    xs2 = np.array([ABSENT, ABSENT, 190, 188, ABSENT, ABSENT], np.int32)
    code, counts = H.c_classify(1, as1, xs1, as2, xs2, np.array([0x2A], np.uint64), ABSENT)
    idx, off = H.c_compact(1, code)
    want = []
    for c in np.flatnonzero(counts):
        want.append("count (%s, %s) = %d" % (H.STATES[c >> 3], H.STATES[c & 7], counts[c]))
    for b in range(6):
        for i in idx[int(off[b]):int(off[b + 1])]:
            want.append("bin %s: records %d and %d" % (H.STATES[b], i - 1, i))
    want += ["code[%d] = 0x%02X" % (i, c) for i, c in enumerate(code)]
    assert proc.stdout.splitlines() == want
    assert any("primary_specific" in l for l in want) and any("secondary_multi" in l for l in want)
