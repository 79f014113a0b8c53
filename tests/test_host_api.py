"""The drop-in Python interface (xenomapper_amd.xenomapper) on the GPU, driven the way the
reference's own test-suite drives the reference (tests/test_xenomapper.py), plus every golden
end-to-end case: six bin texts byte-identical (SHA-224), category_counts, summary text."""
import hashlib
import io
import os

import pytest

from tests import helpers as H
from tests.helpers import NEG

pytestmark = pytest.mark.gpu

G3 = H.golden("g3_end_to_end.json")["cases"]


def run_case(xm, case, block_records=None):
    t1, t2 = H.case_texts(case)
    sam1, sam2 = io.StringIO(t1), io.StringIO(t2)
    outs = {name: io.StringIO() for name in H.STATES}
    hdr = outs if case["options"]["header_sinks"] == "all" else \
        {k: outs[k] for k in ("primary_specific", "secondary_specific")}
    xm.process_headers(sam1, sam2, **hdr)
    loop = {"se": xm.main_single_end, "pe": xm.main_paired_end,
            "pe_conservative": xm.conservative_main_paired_end}[case["mode"]]
    tag_func = getattr(xm, case["options"]["tag_func"])
    old = xm.BLOCK_RECORDS
    if block_records:
        xm.BLOCK_RECORDS = block_records
    try:
        counts = loop(xm.getReadPairs(sam1, sam2, skip_repeated_reads=case["options"]["skip_repeated"]),
                      min_score=H.unnum(case["options"]["min_score"]), tag_func=tag_func, **outs)
    finally:
        xm.BLOCK_RECORDS = old
    return counts, outs


def check_case(xm, case, block_records=None):
    counts, outs = run_case(xm, case, block_records)
    exp = case["expect"]
    flat = {("|".join(k) if isinstance(k, tuple) else k): v for k, v in counts.items()}
    assert flat == exp["counts"]
    for name in H.STATES:
        text = outs[name].getvalue()
        assert len(text) == exp["bins"][name]["len"], name
        assert hashlib.sha224(text.encode("latin-1")).hexdigest() == exp["bins"][name]["sha224"], name
    buf = io.StringIO()
    xm.output_summary(counts, outfile=buf)
    assert buf.getvalue() == exp["summary"]


@pytest.mark.parametrize("case", G3, ids=[c["name"] for c in G3])
def test_golden_end_to_end(case):
    from xenomapper_amd import xenomapper as xm
    check_case(xm, case)


@pytest.mark.parametrize("block", [1, 2, 3, 7, 64, 257])
def test_block_boundaries_do_not_change_output(block):
    from xenomapper_amd import xenomapper as xm
    by = {c["name"]: c for c in G3}
    for name in ("all36_liberal", "all36_conservative", "all36_se", "ref_pe_conservative_cigar_min"):
        check_case(xm, by[name], block_records=block)


def test_reference_suite_rows():
    """The reference's own unit-test rows against the drop-in functions (tests/test_xenomapper.py)."""
    from xenomapper_amd import xenomapper as xm
    rows = [((200, 199, 199, 198, NEG), 'primary_specific'), ((200, 200, 199, 198, NEG), 'primary_multi'),
            ((199, 198, 200, 198, NEG), 'secondary_specific'), ((199, 198, 200, 200, NEG), 'secondary_multi'),
            ((NEG, NEG, NEG, NEG, NEG), 'unassigned'), ((200, 199, 200, 198, NEG), 'unresolved'),
            ((200, 199, 199, 199, NEG), 'primary_specific'), ((200, 200, 199, 199, NEG), 'primary_multi'),
            ((199, 199, 200, 199, NEG), 'secondary_specific'), ((199, 199, 200, 200, NEG), 'secondary_multi'),
            ((9, 8, 8, 8, 10), 'unassigned'), ((200, 200, 200, 200, NEG), 'unresolved'),
            ((-6, NEG, NEG, NEG, NEG), 'primary_specific'), ((NEG, NEG, -6, NEG, NEG), 'secondary_specific'),
            ((-6, NEG, -2, NEG, NEG), 'secondary_specific'), ((0, NEG, -2, NEG, NEG), 'primary_specific'),
            ((-2, NEG, 0, NEG, NEG), 'secondary_specific')]          # :165-183
    for inpt, outpt in rows:
        assert xm.get_mapping_state(*inpt) == outpt
    with pytest.raises(RuntimeError):
        xm.get_mapping_state(float("nan"), 1, 2, 3)
    blank = [''] * 5
    cig = [('50M', ['NM:i:0'], 0), ('1S49M', ['NM:i:0'], -2), ('50M', ['NM:i:2'], -12),
           ('50M', ['NM:i:0', 'AS:i:100', 'XS:i:99'], 0), ('10M1I39M', ['NM:i:0'], -8),
           ('10M1D39M', ['NM:i:0'], -8), ('10M2D38M', ['NM:i:0'], -11), ('10M1I10M1D28M', ['NM:i:0'], -16),
           ('10M1234N40M', ['NM:i:0'], 0)]                             # :215-227
    for cigar, opts, want in cig:
        got = xm.get_cigarbased_AS_tag(blank + [cigar] + [''] * 5 + opts)
        assert got == want and isinstance(got, int)
    assert xm.get_cigarbased_AS_tag(blank + ['*'] + [''] * 5 + ['YT:Z:UU']) == NEG


def test_g2_cigar_rows_on_gpu():
    from xenomapper_amd import xenomapper as xm
    for case in H.golden("g2_tag_parsers.json")["cases"]:
        if case["func"] != "get_cigarbased_AS_tag" or case["tag"] != "AS":
            continue
        exp = case["expect"]
        if "error" in exp:
            with pytest.raises(Exception) as info:
                xm.get_cigarbased_AS_tag(case["fields"])
            assert type(info.value).__name__ == exp["error"]
        else:
            got = xm.get_cigarbased_AS_tag(case["fields"])
            assert got == H.unnum(exp["value"]) and type(got).__name__ == exp["type"], case


def test_reference_pinned_digests():
    """The three digests the reference's tests assert (:93, :125, :158), literal."""
    from xenomapper_amd import xenomapper as xm
    by = {c["name"]: c for c in G3}
    want = {"ref_se": "381325b12dd9a9cd3afdd72eeb16b23cc92ddd16f675bb21bb21e08e",
            "ref_pe_liberal_testlayout": "64c0e24bf141c5aa3bb0993c73b34cdfe630a504ac424843f746918d",
            "ref_pe_conservative_testlayout": "c4de3de755092c8f9ff1eb2cd360a502d74ebd4c1e65ed282515ed3e"}
    for name, digest in want.items():
        _, outs = run_case(xm, by[name])
        assert hashlib.sha224(outs["primary_specific"].getvalue().encode("latin-1")).hexdigest() == digest


def test_errors_keep_partial_output():
    """Input errors surface as the reference's exception types after everything before them was written."""
    from xenomapper_amd import xenomapper as xm

    def rec(name, *opts):
        return [name, "0", "chr1", "1", "30", "10M", "*", "0", "0", "ACGTACGTAC", "IIIIIIIIII"] + list(opts)

    good = [(rec("a", "AS:i:9"), rec("a", "AS:i:3")), (rec("b", "AS:i:2"), rec("b", "AS:i:8"))]
    # duplicate tag match (RG:Z:BASS contains 'AS') -> ValueError, first two reads already written
    bad = good + [(rec("c", "AS:i:5", "RG:Z:BASS"), rec("c", "AS:i:1"))] + good
    out1, out2 = io.StringIO(), io.StringIO()
    with pytest.raises(ValueError):
        xm.main_single_end(iter(bad), primary_specific=out1, secondary_specific=out2)
    assert out1.getvalue().count("\n") == 1 and out2.getvalue().count("\n") == 1
    # name mismatch inside the loop -> AssertionError
    out1 = io.StringIO()
    with pytest.raises(AssertionError):
        xm.main_single_end(iter(good + [(rec("x", "AS:i:1"), rec("y", "AS:i:1"))]), primary_specific=out1)
    assert out1.getvalue().count("\n") == 1
    # non-numeric value -> ValueError
    with pytest.raises(ValueError):
        xm.main_single_end(iter([(rec("a", "AS:i:7", "XS:A:+"), rec("a"))]), primary_specific=io.StringIO())
    # paired: an unpaired record is never scored, so its malformed tag does not raise (ref :402-405)
    pe = [(rec("p", "AS:i:9"), rec("p")), (rec("p", "AS:i:9"), rec("p")),
          (rec("lonely", "AS:i:1", "RG:Z:BASS"), rec("lonely")),
          (rec("q", "AS:i:9"), rec("q")), (rec("q", "AS:i:9"), rec("q"))]
    out1 = io.StringIO()
    counts = xm.main_paired_end(iter(pe), primary_specific=out1)
    assert counts == {("primary_specific", "primary_specific"): 2} and out1.getvalue().count("\n") == 4
    # NaN score -> RuntimeError after the earlier pair was written
    pe = pe[:2] + [(rec("n", "AS:f:nan"), rec("n", "AS:i:2")), (rec("n", "AS:i:1"), rec("n", "AS:i:2"))]
    out1 = io.StringIO()
    with pytest.raises(RuntimeError):
        xm.main_paired_end(iter(pe), primary_specific=out1)
    assert out1.getvalue().count("\n") == 2


def test_float_and_wide_scores_use_binary64():
    from xenomapper_amd import xenomapper as xm

    def rec(name, *opts):
        return [name, "0", "chr1", "1", "30", "10M", "*", "0", "0", "ACGTACGTAC", "IIIIIIIIII"] + list(opts)
    pairs = [(rec("a", "AS:f:12.5", "XS:f:12.25"), rec("a", "AS:f:12.25")),
             (rec("b", "AS:i:3000000000"), rec("b", "AS:i:3000000001")),
             (rec("c", "AS:i:5", "XS:i:5"), rec("c", "AS:f:4.999"))]
    counts = xm.main_single_end(iter(pairs), primary_specific=io.StringIO(), min_score=12.3)
    assert counts == {"primary_specific": 1, "secondary_specific": 1, "unassigned": 1}


def test_custom_tag_func_plugin():
    """A caller-supplied tag_func (the plugin interface, ref :208-225, :310-312)."""
    from xenomapper_amd import xenomapper as xm

    def zm_scores(sam_line, tag="AS"):
        return xm.get_tag(sam_line, {"AS": "ZA", "XS": "ZM"}.get(tag, tag))

    def rec(name, *opts):
        return [name, "0", "chr1", "1", "30", "10M", "*", "0", "0", "ACGT", "IIII"] + list(opts)
    pairs = [(rec("a", "ZA:i:9", "ZM:i:9"), rec("a", "ZA:i:3")), (rec("b", "ZA:i:1"), rec("b", "ZA:i:4", "ZM:i:2"))]
    pm, ss = io.StringIO(), io.StringIO()
    counts = xm.main_single_end(iter(pairs), primary_specific=None, primary_multi=pm, secondary_specific=ss,
                                tag_func=zm_scores)
    assert counts == {"primary_multi": 1, "secondary_specific": 1}
    assert pm.getvalue().startswith("a\t") and ss.getvalue().startswith("b\t")


def test_shared_sink_keeps_reference_interleaving():
    from xenomapper_amd import xenomapper as xm
    by = {c["name"]: c for c in G3}
    case = by["all36_liberal"]
    t1, t2 = H.case_texts(case)
    s1, s2 = io.StringIO(t1), io.StringIO(t2)
    xm.get_sam_header(s1), xm.get_sam_header(s2)
    one = io.StringIO()
    xm.main_paired_end(xm.getReadPairs(s1, s2), primary_specific=one, secondary_specific=one, primary_multi=one,
                       secondary_multi=one, unassigned=one, unresolved=one)
    o1, o2 = io.StringIO(t1), io.StringIO(t2)
    H.ORACLE.read_header(o1), H.ORACLE.read_header(o2)
    ref_one = io.StringIO()
    H.ORACLE.run_paired_end(H.ORACLE.read_pairs(o1, o2), [ref_one] * 6)
    assert one.getvalue() == ref_one.getvalue()


def test_cli_main(tmp_path, capsys):
    from xenomapper_amd import xenomapper as xm
    by = {c["name"]: c for c in G3}
    case = by["ref_pe_conservative"]
    t1, t2 = H.case_texts(case)
    (tmp_path / "h.sam").write_text(t1)
    (tmp_path / "m.sam").write_text(t2)
    args = ["--primary_sam", str(tmp_path / "h.sam"), "--secondary_sam", str(tmp_path / "m.sam"),
            "--paired", "--conservative"]
    for name in H.STATES:
        args += ["--" + name, str(tmp_path / (name + ".sam"))]
    xm.main(args)
    err = capsys.readouterr().err
    assert err == case["expect"]["summary"]
    for name in H.STATES:
        text = (tmp_path / (name + ".sam")).read_text()
        assert hashlib.sha224(text.encode("latin-1")).hexdigest() == case["expect"]["bins"][name]["sha224"]


G5 = H.golden("g5_errors.json")["cases"]


@pytest.mark.parametrize("via_files", [False, True])
@pytest.mark.parametrize("case", G5, ids=[c["name"] for c in G5])
def test_g5_errors_like_the_reference(case, via_files, tmp_path):
    """Malformed input: the reference's exception type after the reference's partial output, through the iterator
    loops and through the file fast path."""
    from xenomapper_amd import xenomapper as xm
    t1, t2 = case["text"]
    outs = {name: io.StringIO() for name in H.STATES}
    m = H.unnum(case["min_score"])
    tag_func = getattr(xm, case["tag_func"])
    err = None
    try:
        if via_files:
            (tmp_path / "a.sam").write_text(t1)
            (tmp_path / "b.sam").write_text(t2)
            xm.classify_sam_files(str(tmp_path / "a.sam"), str(tmp_path / "b.sam"), paired=case["mode"] != "se",
                                  conservative=case["mode"] == "pe_conservative", min_score=m, tag_func=tag_func,
                                  skip_repeated_reads=False, **outs)
        else:
            loop = {"se": xm.main_single_end, "pe": xm.main_paired_end,
                    "pe_conservative": xm.conservative_main_paired_end}[case["mode"]]
            loop(xm.getReadPairs(io.StringIO(t1), io.StringIO(t2)), min_score=m, tag_func=tag_func, **outs)
    except Exception as exc:
        err = type(exc).__name__
    assert err == case["error"]
    for name in H.STATES:
        assert outs[name].getvalue() == case["outputs"][name], name


@pytest.mark.parametrize("name", ["all36_liberal", "all36_conservative", "ref_se", "cfg2_pe_liberal"])
def test_the_ctypes_stub_of_integration_md(name):
    """INTEGRATION.md section B shows the binding a maintainer of the reference would add.  The code block is taken
    from the document as it stands and run: its category counts and bin lists must be the golden ones."""
    import io
    import math
    import re
    from xenomapper_amd import _ffi, xenomapper as xm
    case = {c["name"]: c for c in H.golden("g3_end_to_end.json")["cases"]}[name]
    text = open(os.path.join(H.REPO, "INTEGRATION.md")).read()
    section = text[text.index("## B."):text.index("## C.")]
    code = re.search(r"```python\n(.*?)```", section, flags=re.S).group(1)
    assert '"libxenomapper_hip.so"' in code
    _ffi.lib()                                                       # one HIP runtime per process (see _ffi)
    code = code.replace('"libxenomapper_hip.so"', repr(_ffi.LIB_PATH))
    ns = {"get_tag": xm.get_tag,
          "_to_i32": lambda v: -2**31 if v == float("-inf") else int(v),
          "_floor_i32": lambda m: -2**31 if m == float("-inf") else max(-2**31, min(2**31 - 1, math.floor(m)))}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    t1, t2 = H.case_texts(case)
    s1, s2 = io.StringIO(t1), io.StringIO(t2)
    xm.get_sam_header(s1), xm.get_sam_header(s2)
    pairs = list(xm.getReadPairs(s1, s2, skip_repeated_reads=case["options"]["skip_repeated"]))
    mode = {"se": 0, "pe": 1, "pe_conservative": 2}[case["mode"]]
    counts, lists = ns["classify_block"]([p[0] for p in pairs], [p[1] for p in pairs], mode,
                                         H.unnum(case["options"]["min_score"]))
    exp = case["expect"]
    names = ["primary_specific", "secondary_specific", "primary_multi", "secondary_multi", "unresolved", "unassigned"]
    got = {}
    for c in range(64):
        if counts[c]:
            key = names[c] if mode == 0 else "|".join((names[c >> 3], names[c & 7]))
            got[key] = int(counts[c])
    assert got == exp["counts"]
    want = [[] for _ in range(6)]
    for f, r, i in zip(exp["unit_fwd"], exp["unit_rev"], exp["unit_index"]):
        want[H.ORACLE.bin_of(mode, int(f), int(r))].append(i)
    assert [l.tolist() for l in lists] == want
