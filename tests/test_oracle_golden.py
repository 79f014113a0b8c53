"""Pins the oracle (oracle/xm_oracle.py, oracle/xm_oracle.c) to the reference.

Two anchors: the known-answer rows and SHA-224 digests the reference's own test-suite holds
(quoted literally below with their file:line), and golden vectors recorded by importing the
reference in the build container (tools/make_golden.py).  CPU only.
"""
import hashlib
import io
import itertools

import numpy as np
import pytest

from tests import helpers as H
from tests.helpers import ORACLE, NEG


# ---------------------------------------------------------------- reference's own KAT rows
REF_STATE_ROWS = [      # xenomapper/tests/test_xenomapper.py:165-183
    ((200, 199, 199, 198, NEG), 'primary_specific'), ((200, 200, 199, 198, NEG), 'primary_multi'),
    ((199, 198, 200, 198, NEG), 'secondary_specific'), ((199, 198, 200, 200, NEG), 'secondary_multi'),
    ((NEG, NEG, NEG, NEG, NEG), 'unassigned'), ((200, 199, 200, 198, NEG), 'unresolved'),
    ((200, 199, 199, 199, NEG), 'primary_specific'), ((200, 200, 199, 199, NEG), 'primary_multi'),
    ((199, 199, 200, 199, NEG), 'secondary_specific'), ((199, 199, 200, 200, NEG), 'secondary_multi'),
    ((9, 8, 8, 8, 10), 'unassigned'), ((200, 200, 200, 200, NEG), 'unresolved'),
    ((-6, NEG, NEG, NEG, NEG), 'primary_specific'), ((NEG, NEG, -6, NEG, NEG), 'secondary_specific'),
    ((-6, NEG, -2, NEG, NEG), 'secondary_specific'), ((0, NEG, -2, NEG, NEG), 'primary_specific'),
    ((-2, NEG, 0, NEG, NEG), 'secondary_specific'),
]


def _to_i32(v):
    return -2**31 if v == NEG else int(v)


@pytest.mark.parametrize("row,name", REF_STATE_ROWS)
def test_state_reference_rows(row, name):
    assert ORACLE.mapping_state_name(*row) == name
    c = H.c_oracle()
    assert H.STATES[c.xmo_state_f64(*[float(v) for v in row])] == name
    assert H.STATES[c.xmo_state_i32(*[_to_i32(v) for v in row[:4]], H.floor_min_score(row[4]))] == name


def test_cigar_reference_rows():
    # xenomapper/tests/test_xenomapper.py:215-232
    blank = [''] * 5
    rows = [('50M', ['NM:i:0'], 0), ('1S49M', ['NM:i:0'], -2), ('50M', ['NM:i:2'], -12),
            ('50M', ['NM:i:0', 'AS:i:100', 'XS:i:99'], 0), ('10M1I39M', ['NM:i:0'], -8),
            ('10M1D39M', ['NM:i:0'], -8), ('10M2D38M', ['NM:i:0'], -11),
            ('10M1I10M1D28M', ['NM:i:0'], -16), ('10M1234N40M', ['NM:i:0'], 0)]
    for cigar, opts, want in rows:
        assert ORACLE.cigar_score(blank + [cigar] + [''] * 5 + opts) == want
    assert ORACLE.cigar_score(blank + ['50M'] + [''] * 5 + ['NM:i:0', 'AS:i:100', 'XS:i:99'], tag='XS') == 99
    assert ORACLE.cigar_score(blank + ['*'] + [''] * 5 + ['YT:Z:UU']) == NEG


def test_tag_reference_rows():
    # xenomapper/tests/test_xenomapper.py:191-197, :203-209
    rec = [''] * 5 + ['50M'] + [''] * 5
    assert ORACLE.tag_score(rec + ['NM:i:0', 'AS:i:101', 'XS:i:99'], 'AS') == 101
    assert ORACLE.tag_score(rec + ['NM:i:0', 'AS:i:100', 'XS:i:99'], 'XS') == 99
    assert ORACLE.tag_score(rec + ['NM:i:0', 'AS:i:100', 'XS:i:99'], 'NM') == 0
    assert ORACLE.tag_score(rec + ['YT:Z:UU'], 'AS') == NEG
    assert ORACLE.tag_score_zs(rec + ['NM:i:0', 'AS:i:101', 'XS:A:+', 'ZS:i:99'], 'AS') == 101
    assert ORACLE.tag_score_zs(rec + ['NM:i:0', 'AS:i:100', 'XS:A:+', 'ZS:i:99'], 'XS') == 99
    assert ORACLE.tag_score_zs(rec + ['NM:i:0', 'AS:i:100', 'XS:A:+', 'ZS:i:99'], 'NM') == 0


def test_summary_reference_text():
    # xenomapper/tests/test_xenomapper.py:235-245
    canned = ('-' * 80 + '\n' + 'Read Count Category Summary\n\n'
              '|       Category                                     |     Count       |\n'
              '|:--------------------------------------------------:|:---------------:|\n'
              '|  bar                                               |            101  |\n'
              '|  foo                                               |              1  |\n\n')
    assert ORACLE.summary_text({'foo': 1, 'bar': 101}) == canned


# ---------------------------------------------------------------- G1
def test_g1_lattice_python_and_c():
    g = H.golden("g1_mapping_state.json")
    vals = [H.unnum(v) for v in g["lattice_values"]]
    mins = [H.unnum(m) for m in g["lattice_min_scores"]]
    want = g["lattice_states"]
    c = H.c_oracle()
    k = 0
    for m in mins:
        mi = H.floor_min_score(m)
        for a1, x1, a2, x2 in itertools.product(vals, repeat=4):
            w = int(want[k]); k += 1
            assert ORACLE.mapping_state(a1, x1, a2, x2, m) == w
            assert c.xmo_state_f64(float(a1), float(x1), float(a2), float(x2), float(m)) == w
            assert c.xmo_state_i32(_to_i32(a1), _to_i32(x1), _to_i32(a2), _to_i32(x2), mi) == w
    assert k == len(want) == 5184


def test_g1_rows():
    g = H.golden("g1_mapping_state.json")
    c = H.c_oracle()
    for row in g["rows"]:
        vals = [H.unnum(v) for v in row[:5]]
        assert ORACLE.mapping_state(*vals) == row[5]
        assert c.xmo_state_f64(*[float(v) for v in vals]) == row[5]
        integral = all(v == NEG or (float(v).is_integer() and abs(v) < 2**31) for v in vals[:4])
        if integral and vals[4] == vals[4]:
            assert c.xmo_state_i32(*[_to_i32(v) for v in vals[:4]], H.floor_min_score(vals[4])) == row[5]


def test_nan_falls_through():
    nan = float("nan")
    with pytest.raises(RuntimeError):
        ORACLE.mapping_state(nan, 1, 2, 3)
    assert H.c_oracle().xmo_state_f64(nan, 1.0, 2.0, 3.0, NEG) == 6
    assert ORACLE.mapping_state(3, nan, 2, nan) == ORACLE.PM      # `not nan` is False, 3 > nan is False


# ---------------------------------------------------------------- G2
FUNCS = {"get_tag": ORACLE.tag_score, "get_tag_with_ZS_as_XS": ORACLE.tag_score_zs,
         "get_cigarbased_AS_tag": ORACLE.cigar_score}


def test_g2_parsers():
    g = H.golden("g2_tag_parsers.json")
    assert len(g["cases"]) > 200
    for case in g["cases"]:
        fn = FUNCS[case["func"]]
        exp = case["expect"]
        if "error" in exp:
            with pytest.raises(Exception) as info:
                fn(case["fields"], tag=case["tag"])
            assert type(info.value).__name__ == exp["error"], case
        else:
            got = fn(case["fields"], tag=case["tag"])
            want = H.unnum(exp["value"])
            assert type(got).__name__ == exp["type"], case
            assert got == want or (got != got and want != want), case


# ---------------------------------------------------------------- G3 / G4
G3 = H.golden("g3_end_to_end.json")["cases"]


def run_oracle_case(case):
    t1, t2 = H.case_texts(case)
    sam1, sam2 = io.StringIO(t1), io.StringIO(t2)
    outs = [io.StringIO() for _ in range(6)]
    hdr_outs = list(outs)
    if case["options"]["header_sinks"] == "two":
        hdr_outs = [outs[0], outs[1], None, None, None, None]
    ORACLE.write_headers(sam1, sam2, hdr_outs)
    pairs = ORACLE.read_pairs(sam1, sam2, case["options"]["skip_repeated"])
    scorer = FUNCS[case["options"]["tag_func"]]
    m = H.unnum(case["options"]["min_score"])
    if case["mode"] == "se":
        res = ORACLE.run_single_end(pairs, outs, m, scorer)
    else:
        res = ORACLE.run_paired_end(pairs, outs, m, scorer, conservative=case["mode"] == "pe_conservative")
    return res, outs


@pytest.mark.parametrize("case", G3, ids=[c["name"] for c in G3])
def test_g3_end_to_end(case):
    res, outs = run_oracle_case(case)
    exp = case["expect"]
    paired = case["mode"] != "se"
    assert res.n_records == exp["n_records"]
    assert [u[0] for u in res.units] == exp["unit_index"]
    assert "".join(str(u[1]) for u in res.units) == exp["unit_fwd"]
    assert "".join(str(u[2]) for u in res.units) == exp["unit_rev"]
    named = res.named_counts(paired)
    flat = {("|".join(k) if isinstance(k, tuple) else k): v for k, v in named.items()}
    assert flat == exp["counts"]
    for b, name in enumerate(H.STATES):
        text = outs[b].getvalue()
        assert len(text) == exp["bins"][name]["len"], name
        assert hashlib.sha224(text.encode("latin-1")).hexdigest() == exp["bins"][name]["sha224"], name
    assert ORACLE.summary_text(named) == exp["summary"]


def test_reference_pinned_digests():
    """The three SHA-224 digests the reference's own tests assert (tests/test_xenomapper.py:93,
    :125, :158) -- quoted here, not read from the golden file."""
    by_name = {c["name"]: c for c in G3}
    want = {"ref_se": "381325b12dd9a9cd3afdd72eeb16b23cc92ddd16f675bb21bb21e08e",
            "ref_pe_liberal_testlayout": "64c0e24bf141c5aa3bb0993c73b34cdfe630a504ac424843f746918d",
            "ref_pe_conservative_testlayout": "c4de3de755092c8f9ff1eb2cd360a502d74ebd4c1e65ed282515ed3e"}
    for name, digest in want.items():
        _, outs = run_oracle_case(by_name[name])
        assert hashlib.sha224(outs[0].getvalue().encode("latin-1")).hexdigest() == digest


def test_g4_headers_and_reference_lengths():
    g = H.golden("g4_headers.json")
    case = {c["name"]: c for c in G3}["ref_pe_liberal"]
    t1, t2 = H.case_texts(case)
    outs = [io.StringIO() for _ in range(6)]
    ORACLE.write_headers(io.StringIO(t1), io.StringIO(t2), outs)
    for b, name in enumerate(H.STATES):
        assert outs[b].getvalue() == g[name]
    # tests/test_xenomapper.py:46-51
    lens = dict(primary_specific=695, secondary_specific=629, primary_multi=708, secondary_multi=642,
                unassigned=705, unresolved=705)
    for b, name in enumerate(H.STATES):
        assert len(outs[b].getvalue()) == lens[name]


# ---------------------------------------------------------------- column form: C oracle == Python oracle
def test_c_classify_matches_python_columns():
    rng = np.random.default_rng(7)
    n = 3000
    vals = np.array([-2**31, -7, -1, 0, 1, 3, 5], dtype=np.int64)
    cols = [vals[rng.integers(0, len(vals), n)].astype(np.int32) for _ in range(4)]
    flags = rng.random(n) < 0.6
    bits = H.synth.pack_unit_bits(flags)
    for mode in (0, 1, 2):
        for m in (NEG, 0.5, -2.0):
            fcols = [np.where(c == -2**31, NEG, c.astype(np.float64)) for c in cols]
            py_code, py_counts = ORACLE.classify_columns(mode, *[c.tolist() for c in fcols], flags.tolist(), m)
            code, counts = H.c_classify(mode, *cols, bits, H.floor_min_score(m))
            assert code.tolist() == py_code
            assert {k: int(v) for k, v in enumerate(counts) if v} == dict(py_counts)
            codef, countsf = H.c_classify(mode, *fcols, bits, m)
            assert (codef == code).all() and (countsf == counts).all()
            idx, off = H.c_compact(mode, code)
            # stable split: each bin's indices ascending, bins follow the combine rule
            for b in range(7):
                seg = idx[int(off[b]):int(off[b + 1])]
                assert (np.diff(seg.astype(np.int64)) > 0).all()
                for i in seg[:50]:
                    c = int(code[i])
                    assert ORACLE.bin_of(mode, c >> 3, c & 7) == b


def test_c_cigar_matches_python():
    cols = H.synth.cigar_columns(2000, 11)
    got, bad = H.c_cigar_scores(cols["nm"], cols["cig_off"], cols["cig_oplen"])
    assert bad == 0
    for i in range(2000):
        ops = cols["cig_oplen"][cols["cig_off"][i]:cols["cig_off"][i + 1]]
        fields = [''] * 5 + [H.synth.cigar_string(ops)] + [''] * 5
        if cols["nm"][i] != H.synth.ABSENT:
            fields.append("NM:i:%d" % cols["nm"][i])
        want = ORACLE.cigar_score(fields)
        assert (got[i] == H.synth.ABSENT and want == NEG) or got[i] == want


# ---------------------------------------------------------------- G5: malformed input
G5 = H.golden("g5_errors.json")["cases"]


@pytest.mark.parametrize("case", G5, ids=[c["name"] for c in G5])
def test_g5_error_type_and_partial_output(case):
    """Exception type and what was written before it, as recorded from the reference -- including the order in
    which the eight tags of a pair are read (all before either state, xenomapper.py:408-418)."""
    t1, t2 = case["text"]
    outs = [io.StringIO() for _ in range(6)]
    scorer = FUNCS[case["tag_func"]]
    m = H.unnum(case["min_score"])
    err = None
    try:
        pairs = ORACLE.read_pairs(io.StringIO(t1), io.StringIO(t2))
        if case["mode"] == "se":
            ORACLE.run_single_end(pairs, outs, m, scorer)
        else:
            ORACLE.run_paired_end(pairs, outs, m, scorer, conservative=case["mode"] == "pe_conservative")
    except Exception as exc:
        err = type(exc).__name__
    assert err == case["error"]
    for b, name in enumerate(H.STATES):
        assert outs[b].getvalue() == case["outputs"][name], name


def test_g7_random_corpus():
    """1000 small adversarial text pairs run through the reference (tools/make_golden.py, G7): the oracle must raise
    the same exception type, have written the same six texts by then, and return the same counts."""
    cases = H.golden("g7_random_corpus.json")["cases"]
    assert len(cases) == 1000
    for k, case in enumerate(cases):
        t1, t2 = case["text"]
        outs = [io.StringIO() for _ in range(6)]
        scorer = FUNCS[case["tag_func"]]
        m = H.unnum(case["min_score"])
        err, counts = None, None
        try:
            pairs = ORACLE.read_pairs(io.StringIO(t1, newline=None), io.StringIO(t2, newline=None), case["skip_repeated"])
            if case["mode"] == "se":
                res = ORACLE.run_single_end(pairs, outs, m, scorer)
            else:
                res = ORACLE.run_paired_end(pairs, outs, m, scorer, conservative=case["mode"] == "pe_conservative")
            counts = {("|".join(key) if isinstance(key, tuple) else key): v
                      for key, v in res.named_counts(case["mode"] != "se").items()}
        except Exception as exc:
            err = type(exc).__name__
        assert err == case["error"], k
        for b, name in enumerate(H.STATES):
            assert outs[b].getvalue() == case["outputs"][name], (k, name)
        if err is None:
            assert counts == case["counts"], k


G8 = H.golden("g8_large_runs.json")["cases"]


@pytest.mark.parametrize("case", G8, ids=[c["name"] for c in G8])
def test_g8_large_runs(case):
    """100 k-pair text twins of configs 1, 2, 3 and 5 as the reference classified them: counts, digests and line
    counts of the six outputs, summary."""
    res, outs = run_oracle_case(case)
    exp = case["expect"]
    counts = {("|".join(k) if isinstance(k, tuple) else k): v for k, v in res.named_counts(case["mode"] != "se").items()}
    assert counts == exp["counts"]
    for b, name in enumerate(H.STATES):
        text = outs[b].getvalue()
        assert (hashlib.sha224(text.encode("latin-1")).hexdigest(), len(text)) == (exp["bins"][name]["sha224"], exp["bins"][name]["len"]), name


@pytest.mark.parametrize("species", ["human", "mouse"])
def test_bam_oracle_reproduces_the_reference_sam_fixture(species):
    """Pin of oracle/bam_oracle.py: the reference's BAM fixture, decoded by the plain-Python restatement of the BAM
    layout, is the reference's SAM fixture (what `samtools view -h` gave the reference's authors)."""
    import os
    from oracle import bam_oracle
    base = os.path.join(H.GOLDEN, "ref_data", "paired_end_testdata_%s" % species)
    with open(base + ".bam", "rb") as fh:
        header, lines = bam_oracle.bam_to_sam(fh.read())
    with open(base + ".sam") as fh:
        assert header + "\n".join(lines) == fh.read()
