#!/usr/bin/env python3
"""Record golden vectors by running the *reference* (genomematt/xenomapper v1.0.2).

Runs only in the build container, where the reference is mounted read-only at
/root/reference; it refuses to run anywhere else.  Only the recorded inputs/outputs
(tests/golden/*.json) and the data files the reference's own tests hold
(tests/golden/ref_data/*.sam) are committed -- no reference source travels.

    python tools/make_golden.py            # rewrites tests/golden/

Vectors:
  G1  get_mapping_state truth table: lattice {-inf,-7,-1,0,1,3}^4 x m in {-inf,-2,0,2}
      plus the 17 rows of the reference test (tests/test_xenomapper.py:165-183)
  G2  tag / CIGAR parser table incl. adversarial lines and error rows
  G3  end-to-end runs of the three main loops on the reference's four SAM fixtures and on
      build-generated synthetic SAM text: per-unit (index, fwd, rev, bin), category_counts,
      SHA-224 + length of each of the six bin texts, the stderr summary text
  G5  malformed inputs: exception type and the partial outputs
  G6  the xenomappability companion tool
  G7  a random corpus of small adversarial text pairs: error type, outputs and counts of the reference
  G8  100 k-pair text twins of configs 1, 2, 3, 5: counts, digests of the six outputs, summary
  G9  the command line as a child process: stdout, stderr, exit code, output files
  G10 the companion tool's command line as a child process
"""
import hashlib
import io
import itertools
import json
import os
import shutil
import sys

REF_ROOT = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")

if not os.path.isdir(os.path.join(REF_ROOT, "xenomapper")):
    sys.exit("make_golden.py: reference not mounted at %s -- golden vectors can only be "
             "recorded in the build container" % REF_ROOT)

sys.dont_write_bytecode = True
sys.path.insert(0, REF_ROOT)
sys.path.insert(0, REPO)
from xenomapper import xenomapper as ref            # noqa: E402  (the reference)
from xenomapper_amd import synth                     # noqa: E402  (the build's generator)

NEG = float("-inf")
STATES = ["primary_specific", "secondary_specific", "primary_multi", "secondary_multi",
          "unresolved", "unassigned"]
SIDX = {s: i for i, s in enumerate(STATES)}
BIN_ARGS = STATES          # keyword names of the six sinks == state names


def num(x):
    """JSON-safe number: -inf/inf/nan become strings."""
    if isinstance(x, float) and (x != x or x in (NEG, -NEG)):
        return repr(x)
    return x


# ------------------------------------------------------------------ G1
def g1():
    vals = [NEG, -7, -1, 0, 1, 3]
    mins = [NEG, -2, 0, 2]
    digits = []
    for m in mins:
        for a1, x1, a2, x2 in itertools.product(vals, repeat=4):
            digits.append(str(SIDX[ref.get_mapping_state(a1, x1, a2, x2, m)]))
    extra_in = [
        (200, 199, 199, 198, NEG), (200, 200, 199, 198, NEG), (199, 198, 200, 198, NEG),
        (199, 198, 200, 200, NEG), (NEG, NEG, NEG, NEG, NEG), (200, 199, 200, 198, NEG),
        (200, 199, 199, 199, NEG), (200, 200, 199, 199, NEG), (199, 199, 200, 199, NEG),
        (199, 199, 200, 200, NEG), (9, 8, 8, 8, 10), (200, 200, 200, 200, NEG),
        (-6, NEG, NEG, NEG, NEG), (NEG, NEG, -6, NEG, NEG), (-6, NEG, -2, NEG, NEG),
        (0, NEG, -2, NEG, NEG), (-2, NEG, 0, NEG, NEG),
        # build-authored rows: the XS == 0 quirk, fractional thresholds, large values
        (0, 0, -5, NEG, NEG), (0, 0, 0, 0, NEG), (-3, 0, -9, 0, NEG), (5, 0, 7, 0, 4),
        (10, 9, 10, 9, 9.5), (10, 9, 9, 9, 9.5), (9, 9, 10, 10, 9.5), (10, 10, 10, 10, 10),
        (2147483647, 2147483646, 2147483646, 0, NEG), (-2147483647, NEG, NEG, NEG, NEG),
        (12.5, 12.25, 12.5, NEG, NEG), (12.5, 12.5, 12.25, NEG, 12.4), (3, NEG, 4, 4, 3.999),
        (1, 1, 1, 1, -NEG), (NEG, 5, NEG, 5, NEG),
    ]
    rows = [[num(v) for v in r] + [SIDX[ref.get_mapping_state(*r)]] for r in extra_in]
    return {"lattice_values": [num(v) for v in vals], "lattice_min_scores": [num(m) for m in mins],
            "lattice_states": "".join(digits),
            "lattice_order": "for m in min_scores: for (AS1,XS1,AS2,XS2) in product(values, repeat=4)",
            "rows": rows, "row_format": "AS1,XS1,AS2,XS2,min_score,state"}


# ------------------------------------------------------------------ G2
def g2():
    unmapped = ['HWI-ST960:63:D0CYJACXX:4:1101:21264:2228', '4', '*', '0', '0', '*', '*', '0', '0',
                'TGGTAGTATTGGTTATGGTTCATTGTCCGGAGAGTATATTGTTGAAGAGG',
                'BBCBDFDDHHHGFHHIIIIIJIJJJIGJJJGIAF:CFEGHGGHEEEG@HI', 'YT:Z:UU']
    blank = [''] * 5

    def rec(cigar, *opts):
        return blank + [cigar] + [''] * 5 + list(opts)

    cases = []
    funcs = {"get_tag": ref.get_tag, "get_tag_with_ZS_as_XS": ref.get_tag_with_ZS_as_XS,
             "get_cigarbased_AS_tag": ref.get_cigarbased_AS_tag}

    def add(func, fields, tag):
        try:
            val = funcs[func](fields, tag=tag)
            out = {"value": num(val), "type": type(val).__name__}
        except Exception as exc:            # record the exception type only
            out = {"error": type(exc).__name__}
        cases.append({"func": func, "fields": fields, "tag": tag, "expect": out})

    # rows of the reference's own tests (tests/test_xenomapper.py:191-197, :203-209, :215-232)
    for tag in ("AS", "XS", "NM"):
        add("get_tag", unmapped, tag)
        add("get_tag", rec('50M', 'NM:i:0', 'AS:i:101', 'XS:i:99'), tag)
        add("get_tag", rec('50M', 'NM:i:0', 'AS:i:100', 'XS:i:99'), tag)
        add("get_tag_with_ZS_as_XS", unmapped, tag)
        add("get_tag_with_ZS_as_XS", rec('50M', 'NM:i:0', 'AS:i:101', 'XS:A:+', 'ZS:i:99'), tag)
        add("get_tag_with_ZS_as_XS", rec('50M', 'NM:i:0', 'AS:i:100', 'XS:A:+', 'ZS:i:99'), tag)
    for cigar, opts in [('*', None), ('50M', ['NM:i:0']), ('1S49M', ['NM:i:0']), ('50M', ['NM:i:2']),
                        ('50M', ['NM:i:0', 'AS:i:100', 'XS:i:99']), ('10M1I39M', ['NM:i:0']),
                        ('10M1D39M', ['NM:i:0']), ('10M2D38M', ['NM:i:0']),
                        ('10M1I10M1D28M', ['NM:i:0']), ('10M1234N40M', ['NM:i:0'])]:
        fields = unmapped if opts is None else rec(cigar, *opts)
        add("get_cigarbased_AS_tag", fields, "AS")
        add("get_cigarbased_AS_tag", fields, "XS")
    # build-authored adversarial rows
    adv = [
        rec('50M', 'AS:i:-12', 'XS:i:0'), rec('50M', 'AS:i:0', 'XS:i:-0'),
        rec('50M', 'AS:f:12.5', 'XS:f:1e2'), rec('50M', 'AS:i:7', 'RG:Z:BASS'),
        rec('50M', 'AS:i:7', 'XS:A:+'), rec('50M', 'AS:i:7', 'XS:A:+', 'ZS:i:3'),
        rec('50M', 'AS:i:7', 'XS:i:5', 'ZS:i:3', 'YS:i:9'), rec('50M', 'AS:i:'), rec('50M', 'AS'),
        rec('50M', 'xAS:i:44'), rec('50M', 'AS:i:4:5'), rec('50M', 'AS:i: 9'.strip()),
        rec('50M', 'AS:i:+9', 'XS:i:1_0'), rec('50M', 'AS:i:inf', 'XS:i:nan'),
        rec('50M', 'AS:i:3000000000', 'XS:i:-3000000000'), rec('50M', 'XS:i:4', 'XS:i:4'),
        rec('*', 'NM:i:3'), rec('5H10M2I3M1D4M6S7H', 'NM:i:4'), rec('3=2X1P4N5M', 'NM:i:1'),
        rec('10M2I3M2I3M4D1M', 'NM:i:0'), rec('0010S40M', 'NM:i:00'), rec('7S', 'XN:Z:NM', 'NM:i:2'),
        rec('10Q5S', 'NM:i:1'), rec('5S10', 'NM:i:1'), rec('M5S', 'NM:i:1'), rec('50M', 'NM:i:x'),
        rec('50M', 'NM:i:1.0'), rec('50M', 'NM:i:2', 'NM:i:3'), rec('50M', 'NM:i:-2'),
        rec('1I1D1S', 'NM:i:0', 'AS:i:5', 'XS:i:-17'), rec('50M'), blank + ['50M'] + [''] * 5,
        blank + ['50M'] + [''] * 4, rec('50M', 'NM:i:1', 'XS:i:2', 'XS:i:2'),
    ]
    for fields in adv:
        for func in funcs:
            for tag in ("AS", "XS"):
                add(func, fields, tag)
    return {"cases": cases}


# ------------------------------------------------------------------ G3
def run_reference(text1, text2, mode, tag_func_name="get_tag", min_score=NEG, skip_repeated=False,
                  sinks="all"):
    sam1, sam2 = io.StringIO(text1), io.StringIO(text2)
    outs = {name: io.StringIO() for name in BIN_ARGS}
    if sinks == "two":        # as tests/test_xenomapper.py:107 does for the headers
        hdr_outs = {k: outs[k] for k in ("primary_specific", "secondary_specific")}
    else:
        hdr_outs = outs
    ref.process_headers(sam1, sam2, **hdr_outs)
    tag_func = getattr(ref, tag_func_name)
    pairs = list(ref.getReadPairs(sam1, sam2, skip_repeated_reads=skip_repeated))
    loop = {"se": ref.main_single_end, "pe": ref.main_paired_end,
            "pe_conservative": ref.conservative_main_paired_end}[mode]
    counts = loop(iter(pairs), min_score=min_score, tag_func=tag_func, **outs)
    # per-unit record: recomputed with the reference's own functions on the same pairs
    units = []
    prev = None
    for i, (l1, l2) in enumerate(pairs):
        if mode == "se":
            s = SIDX[ref.get_mapping_state(tag_func(l1, tag='AS'), tag_func(l1, tag='XS'),
                                           tag_func(l2, tag='AS'), tag_func(l2, tag='XS'), min_score)]
            units.append([i, s, s])
        else:
            if prev is not None and prev[0][0] == l1[0]:
                f = SIDX[ref.get_mapping_state(tag_func(prev[0], tag='AS'), tag_func(prev[0], tag='XS'),
                                               tag_func(prev[1], tag='AS'), tag_func(prev[1], tag='XS'),
                                               min_score)]
                r = SIDX[ref.get_mapping_state(tag_func(l1, tag='AS'), tag_func(l1, tag='XS'),
                                               tag_func(l2, tag='AS'), tag_func(l2, tag='XS'), min_score)]
                units.append([i, f, r])
            prev = (l1, l2)
    # cross-check the recomputed units against the Counter the reference returned
    from collections import Counter
    chk = Counter()
    for _, f, r in units:
        chk[STATES[r] if mode == "se" else (STATES[f], STATES[r])] += 1
    assert chk == counts, (chk, counts)
    summary = io.StringIO()
    ref.output_summary(counts, outfile=summary)
    bins = {}
    for name in BIN_ARGS:
        text = outs[name].getvalue()
        bins[name] = {"sha224": hashlib.sha224(text.encode("latin-1")).hexdigest(), "len": len(text),
                      "lines": text.count("\n")}
    return {
        "n_records": len(pairs),
        "unit_index": [u[0] for u in units],
        "unit_fwd": "".join(str(u[1]) for u in units),
        "unit_rev": "".join(str(u[2]) for u in units),
        "counts": {("|".join(k) if isinstance(k, tuple) else k): v for k, v in sorted(counts.items())},
        "bins": bins,
        "summary": summary.getvalue(),
    }


def all36_text():
    """Hand-built pairs hitting all 36 (fwd, rev) tuples, with XS = 0 / AS = 0 / negative scores,
    a triple-QNAME run, singletons and mixed whitespace."""
    # one record realising each state: (AS1, XS1, AS2, XS2); None = tag absent
    realise = {
        0: [(10, 5, 3, None), (0, 0, -5, None), (-3, None, None, None), (7, 0, 7 - 1, 9)],
        1: [(3, None, 10, 5), (-5, None, 0, 0), (None, None, -3, None), (6, 9, 7, 0)],
        2: [(10, 10, 3, None), (-4, -2, -9, None), (5, 7, None, None)],
        3: [(3, None, 10, 10), (-9, None, -4, -2), (None, None, 5, 7)],
        4: [(10, 5, 10, 5), (0, 0, 0, 0), (-7, None, -7, -7)],
        5: [(None, None, None, None), (None, 4, None, 4)],
    }
    lines = [["@HD\tVN:1.0"], ["@HD\tVN:1.0", "@PG\tID:aligner\tPN:x"]]
    q = 0

    def emit(name, vals, k):
        for f in range(2):
            a, x = vals[2 * f], vals[2 * f + 1]
            cols = [name, str(64 + 64 * (k & 1)), "chr%d" % (f + 1), str(100 + q), "30",
                    "20M" if a is not None else "*", "=", "0", "0", "ACGTACGTACGTACGTACGT", "IIIIIIIIIIIIIIIIIIII"]
            if a is not None:
                cols.append("AS:i:%d" % a)
            if x is not None:
                cols.append("XS:i:%d" % x)
            cols.append("YT:Z:CP")
            sep = "\t" if (q + k) % 5 else " "
            lines[f].append(sep.join(cols))

    for f_state in range(6):
        for r_state in range(6):
            for rep in range(2):
                name = "pair%03d" % q
                emit(name, realise[f_state][(q + rep) % len(realise[f_state])], 0)
                emit(name, realise[r_state][(q // 2 + rep) % len(realise[r_state])], 1)
                q += 1
            if (f_state * 6 + r_state) % 7 == 3:          # a singleton
                emit("single%03d" % q, realise[r_state][0], 0)
                q += 1
            if (f_state * 6 + r_state) % 9 == 4:          # three records with one name
                name = "triple%03d" % q
                emit(name, realise[f_state][0], 0)
                emit(name, realise[r_state][0], 1)
                emit(name, realise[(f_state + r_state) % 6][0], 0)
                q += 1
    return "\n".join(lines[0]) + "\n", "\n".join(lines[1]) + "\n"


def g3():
    data_dir = os.path.join(REF_ROOT, "xenomapper", "tests", "data")
    out_dir = os.path.join(GOLDEN, "ref_data")
    os.makedirs(out_dir, exist_ok=True)
    fixtures = {}
    for name in ("test_human_in.sam", "test_mouse_in.sam",
                 "paired_end_testdata_human.sam", "paired_end_testdata_mouse.sam"):
        shutil.copyfile(os.path.join(data_dir, name), os.path.join(out_dir, name))
        os.chmod(os.path.join(out_dir, name), 0o644)
        with open(os.path.join(data_dir, name), "rt") as fh:
            fixtures[name] = fh.read()
    se = ("test_human_in.sam", "test_mouse_in.sam")
    pe = ("paired_end_testdata_human.sam", "paired_end_testdata_mouse.sam")

    cases = []

    def add(name, inputs, mode, **kw):
        if inputs[0] == "ref":
            t1, t2 = fixtures[inputs[1]], fixtures[inputs[2]]
            src = {"kind": "ref_data", "files": [inputs[1], inputs[2]]}
        elif inputs[0] == "synth":
            t1, t2, _ = synth.sam_text_pair(**inputs[1])
            src = {"kind": "synth", "args": inputs[1],
                   "sha224": [hashlib.sha224(t.encode()).hexdigest() for t in (t1, t2)]}
        else:
            t1, t2 = all36_text()
            src = {"kind": "inline", "text": [t1, t2]}
        res = run_reference(t1, t2, mode, **kw)
        opts = {"tag_func": kw.get("tag_func_name", "get_tag"), "min_score": num(kw.get("min_score", NEG)),
                "skip_repeated": kw.get("skip_repeated", False), "header_sinks": kw.get("sinks", "all")}
        cases.append({"name": name, "source": src, "mode": mode, "options": opts, "expect": res})

    # the reference's fixtures, each mode; "two" reproduces the header layout of its own tests
    add("ref_se", ("ref",) + se, "se")
    add("ref_se_skip_repeated", ("ref",) + se, "se", skip_repeated=True)
    add("ref_se_min60", ("ref",) + se, "se", min_score=60.0)
    add("ref_se_cigar", ("ref",) + se, "se", tag_func_name="get_cigarbased_AS_tag")
    add("ref_pe_liberal_testlayout", ("ref",) + pe, "pe", sinks="two")
    add("ref_pe_conservative_testlayout", ("ref",) + pe, "pe_conservative", sinks="two")
    add("ref_pe_liberal", ("ref",) + pe, "pe")
    add("ref_pe_conservative", ("ref",) + pe, "pe_conservative")
    add("ref_pe_liberal_cigar", ("ref",) + pe, "pe", tag_func_name="get_cigarbased_AS_tag")
    add("ref_pe_conservative_cigar_min", ("ref",) + pe, "pe_conservative",
        tag_func_name="get_cigarbased_AS_tag", min_score=-30.5)
    add("ref_pe_liberal_min150", ("ref",) + pe, "pe", min_score=150.0)
    add("ref_pe_conservative_min99_5", ("ref",) + pe, "pe_conservative", min_score=99.5)
    add("ref_pe_as_se", ("ref",) + pe, "se", skip_repeated=True)
    # hand-built: all 36 tuples
    add("all36_liberal", ("inline",), "pe")
    add("all36_conservative", ("inline",), "pe_conservative")
    add("all36_liberal_min0", ("inline",), "pe", min_score=0.0)
    add("all36_se", ("inline",), "se")
    add("all36_se_skip", ("inline",), "se", skip_repeated=True)
    # synthetic text twins of the BASELINE.json configs (same score model, small n)
    cfg1 = dict(n_pairs=3000, seed=1001, profile="bowtie2", paired=False, read_len=50, mixed_ws=0.2,
                irregular=0.05)
    cfg2 = dict(n_pairs=1500, seed=2002, profile="bowtie2", paired=True, read_len=150, irregular=0.02)
    cfg3 = dict(n_pairs=1200, seed=3003, profile="cigar", paired=True, read_len=150, irregular=0.02)
    cfg5 = dict(n_pairs=1500, seed=5005, profile="hisat", paired=True, read_len=150, irregular=0.02,
                mixed_ws=0.05)
    add("cfg1_se", ("synth", cfg1), "se")
    add("cfg1_se_cli", ("synth", cfg1), "se", skip_repeated=True)
    add("cfg2_pe_liberal", ("synth", cfg2), "pe")
    add("cfg2_pe_conservative_min100", ("synth", cfg2), "pe_conservative", min_score=100.0)
    add("cfg3_pe_cigar", ("synth", cfg3), "pe", tag_func_name="get_cigarbased_AS_tag")
    add("cfg3_pe_cigar_conservative", ("synth", cfg3), "pe_conservative",
        tag_func_name="get_cigarbased_AS_tag", min_score=-20.0)
    add("cfg5_pe_zs_conservative", ("synth", cfg5), "pe_conservative",
        tag_func_name="get_tag_with_ZS_as_XS")
    add("cfg5_pe_zs_liberal", ("synth", cfg5), "pe", tag_func_name="get_tag_with_ZS_as_XS")
    return {"cases": cases}


def g5():
    """Malformed inputs: the exception type the reference raises and what it had written by then."""
    def rec(name, *opts):
        return "\t".join([name, "0", "chr1", "1", "30", "10M", "*", "0", "0", "ACGT", "IIII"] + list(opts))

    def sam(lines):
        return "\n".join(lines) + "\n"
    cases = []

    def add(name, l1, l2, mode, tag_func_name="get_tag", min_score=NEG):
        t1, t2 = sam(l1), sam(l2)
        outs = {k: io.StringIO() for k in BIN_ARGS}
        loop = {"se": ref.main_single_end, "pe": ref.main_paired_end,
                "pe_conservative": ref.conservative_main_paired_end}[mode]
        err = None
        try:
            loop(ref.getReadPairs(io.StringIO(t1), io.StringIO(t2)), min_score=min_score,
                 tag_func=getattr(ref, tag_func_name), **outs)
        except Exception as exc:
            err = type(exc).__name__
        cases.append({"name": name, "text": [t1, t2], "mode": mode, "tag_func": tag_func_name, "min_score": num(min_score),
                      "error": err, "outputs": {k: outs[k].getvalue() for k in BIN_ARGS}})

    good = [rec("a", "AS:i:9"), rec("b", "AS:i:2")]
    add("se_duplicate_tag", good + [rec("c", "AS:i:5", "RG:Z:BASS"), rec("d", "AS:i:9")], good + [rec("c"), rec("d")], "se")
    add("se_name_mismatch", good + [rec("x", "AS:i:1")], good + [rec("y", "AS:i:1")], "se")
    add("se_non_numeric", good + [rec("c", "AS:i:7", "XS:A:+")], good + [rec("c")], "se")
    add("se_nan_falls_through", good + [rec("n", "AS:f:nan")], good + [rec("n", "AS:i:2")], "se")
    add("pe_unpaired_malformed_is_never_read",
        [rec("p", "AS:i:9"), rec("p", "AS:i:9"), rec("lonely", "AS:i:1", "RG:Z:BASS"), rec("q", "AS:i:9"), rec("q", "AS:i:9")],
        [rec("p"), rec("p"), rec("lonely"), rec("q"), rec("q")], "pe")
    # all eight tags are read before either state is evaluated: the duplicate of the second mate wins over the first's NaN
    add("pe_second_mate_error_before_first_mate_nan",
        [rec("p", "AS:i:9"), rec("p", "AS:i:9"), rec("r", "AS:i:nan"), rec("r", "AS:i:3")],
        [rec("p"), rec("p"), rec("r"), rec("r", "AS:i:0", "RG:Z:BASS")], "pe")
    add("pe_nan_first_mate", [rec("p", "AS:i:9"), rec("p", "AS:i:9"), rec("r", "AS:i:nan"), rec("r", "AS:i:3")],
        [rec("p"), rec("p"), rec("r"), rec("r", "AS:i:0")], "pe_conservative")
    add("pe_cigar_bad_nm", [rec("p", "NM:i:1"), rec("p", "NM:i:x")], [rec("p", "NM:i:0"), rec("p", "NM:i:0")], "pe",
        tag_func_name="get_cigarbased_AS_tag")
    add("pe_cigar_short_line", [rec("p", "NM:i:1"), "p 0 chr1 1 30 NM:i:2 x x x x x NM:i:2"],
        [rec("p", "NM:i:0"), rec("p", "NM:i:0")], "pe", tag_func_name="get_cigarbased_AS_tag")
    add("se_zs_duplicate", good + [rec("c", "AS:i:5", "ZS:i:1", "ZS:i:1")], good + [rec("c")], "se",
        tag_func_name="get_tag_with_ZS_as_XS")
    return {"cases": cases}


def g6():
    """xenomappability: outputs of the reference's mappability module on its own fixtures and on
    random tracks.  Floats are recorded as hex so that they compare bit for bit."""
    import random
    from xenomapper import mappability as refm
    data_dir = os.path.join(REF_ROOT, "xenomapper", "tests", "data")
    out_dir = os.path.join(GOLDEN, "ref_data")
    for name in ("test_from_EcoliK12DH10B.fasta", "test_from_EcoliK12DH10B_150reads.sam"):
        shutil.copyfile(os.path.join(data_dir, name), os.path.join(out_dir, name))
        os.chmod(os.path.join(out_dir, name), 0o644)
    out = {}
    buf = io.StringIO()
    refm.simulate_reads(open(os.path.join(data_dir, "test_from_EcoliK12DH10B.fasta")), readlength=150, outfile=buf)
    out["simulate_reads_150_sha224"] = hashlib.sha224(buf.getvalue().encode("latin-1")).hexdigest()
    buf = io.StringIO()
    with open(os.path.join(data_dir, "test_from_EcoliK12DH10B_150reads.sam")) as fh:
        refm.single_end_mappability_from_sam(fh, outfile=buf, chromosome_sizes={"Chromosome": 2752, "A_Repeat": 991})
    out["single_end_wiggle_sha224"] = hashlib.sha224(buf.getvalue().encode("latin-1")).hexdigest()
    out["single_end_wiggle_text"] = buf.getvalue()
    with open(os.path.join(data_dir, "paired_end_testdata_human.sam")) as fh:
        out["mate_density_sample3"] = [v.hex() for v in refm.mate_distribution_from_sam(samfile=fh, sample_size=3)]
    with open(os.path.join(data_dir, "paired_end_testdata_human.sam")) as fh:
        dens = refm.mate_distribution_from_sam(samfile=fh)
        out["mate_density_default"] = [v.hex() for v in dens]
    # paired-end mappability of the E. coli single-end track with the fixture's mate density
    buf2 = io.StringIO()
    refm.paired_end_mappability(io.StringIO(out["single_end_wiggle_text"]), dens, outfile=buf2,
                                chromosome_sizes={"Chromosome": 2752, "A_Repeat": 991})
    out["paired_wiggle_sha224"] = hashlib.sha224(buf2.getvalue().encode("latin-1")).hexdigest()
    out["paired_wiggle_len"] = len(buf2.getvalue())
    # random tracks: 0/1 tracks (what the tool produces) and fractional tracks (what from_wiggle accepts)
    rnd = random.Random(60606)
    cases = []
    for n, m, kind in ((0, 3, "bits"), (1, 1, "bits"), (7, 3, "bits"), (64, 5, "bits"), (300, 17, "bits"), (1000, 451, "bits"),
                       (257, 300, "frac"), (2000, 64, "frac"), (513, 1, "frac"), (100, 7, "ints")):
        if kind == "bits":
            track = [float(rnd.random() < 0.6) for _ in range(n)]
        elif kind == "ints":
            track = [rnd.choice([0, 1, 1, 2]) for _ in range(n)]
        else:
            track = [rnd.choice([0.0, 1.0, rnd.random(), 0.25, 1e-3]) for _ in range(n)]
        raw = [rnd.random() for _ in range(m)]
        tot = sum(raw)
        density = [x / tot for x in raw]
        mp = refm.Mappability(chromosome_sizes={"c": n})
        mp["c"] = list(track)
        res = mp.single_end_to_paired(mate_density=density)["c"]
        cases.append({"track": [float(v).hex() for v in track], "track_is_int": kind == "ints",
                      "density": [v.hex() for v in density], "expect": [float(v).hex() for v in res]})
    out["single_end_to_paired"] = cases
    out["smoothed_list"] = [repr(v) for v in refm.smoothed_list([1, 2, 3] * 10 + [100] + [1, 2, 3] * 10)]   # ints and floats
    return out


# ------------------------------------------------------------------ G7
def g7(n_cases=1000, seed=7007):
    """Random corpus: small adversarial SAM-like text pairs (odd whitespace, every newline style, colliding and
    malformed tags, odd CIGARs, repeated / mismatching names, blank lines) through the reference's three loops with
    every plugin -- what it raised, what it had written by then, and the counts it returned."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))

    def pick(seq):
        return seq[int(rng.integers(0, len(seq)))]
    ws = ["\t", "\t", "\t", " ", "\t\t", "  ", " \t", "\x0b", "\x0c", "\x1c", "\x1f"]
    names = ["r1", "r2", "r2", "read/3", "q", "AS", "x:1"]
    nums = [str(int(rng.integers(-300, 301))) for _ in range(40)] + ["0", "-0", "+7", "007", "2147483647", "2147483648",
                                                                    "-2147483648", "1.5", "1e2", "inf", "nan", "", "1_0", "x"]
    tagn = ["AS", "XS", "ZS", "NM", "YS", "XN", "MD", "RG", "xAS", "ASx"]
    fixed_tags = ["RG:Z:BASS", "XS:A:+", "AS", "NM", "YT:Z:UU", "ZS:i:4:5"]
    cigars = ["*", "50M", "10M2I3M1D4M6S", "5H10M", "0010S40M", "10Q5S", "5S10", "M5S", "3=2X1P4N5M", "268435455S", "1I1D1S"]

    def tag():
        u = rng.random()
        if u < 0.10:
            return pick(fixed_tags)
        if u < 0.35:
            return "%s:%s:%s" % (pick(tagn), pick("ifZA"), pick(nums))
        return "%s:i:%d" % (pick(["AS", "XS", "ZS", "NM", "YS"]), int(rng.integers(-60, 61)) if rng.random() < 0.8 else 0)

    def cigar():
        if rng.random() < 0.5:
            return pick(cigars)
        return "".join("%d%s" % (int(rng.integers(0, 401)), pick("MIDNSHP=XQ")) for _ in range(int(rng.integers(0, 7))))

    def line(name):
        n_fixed = pick([11, 11, 11, 11, 3, 6, 1])
        fields = [name] + ["0", "chr1", "7", "30", cigar(), "*", "0", "0", "ACGT", "IIII"][:n_fixed - 1]
        if n_fixed == 11:
            good = ["%s:i:%d" % (t, int(rng.integers(-60, 61)) if rng.random() < 0.85 else 0)
                    for t in ("AS", "XS", "ZS", "NM", "YS") if rng.random() < 0.6]
            if good and "NM" in good[-1]:
                good[-1] = "NM:i:%d" % int(rng.integers(0, 9))
            if rng.random() < 0.2:
                good += [tag() for _ in range(int(rng.integers(1, 3)))]
            fields += [good[i] for i in rng.permutation(len(good))]
        text = fields[0] + "".join(pick(ws) + f for f in fields[1:])
        if rng.random() < 0.1:
            text = pick(ws) + text + pick(ws)
        return text

    cases = []
    for k in range(n_cases):
        n = int(rng.integers(0, 13))
        nm = []
        while len(nm) < n:                                            # runs of equal names: mates and repeats
            nm += [pick(names)] * int(pick([1, 1, 2, 2, 2, 3]))
        nm = nm[:n]
        l1, l2 = [line(x) for x in nm], [line(x) for x in nm]
        if n and rng.random() < 0.125:
            l2[int(rng.integers(0, n))] = line("other")
        if rng.random() < 0.17:
            (l1 if rng.random() < 0.5 else l2).insert(int(rng.integers(0, n + 1)), pick(["", " ", "\t"]))
        nl = pick(["\n", "\n", "\r\n", "\r"])
        t1 = nl.join(l1) + (nl if l1 and rng.random() < 0.5 else "")
        t2 = nl.join(l2) + (nl if l2 and rng.random() < 0.5 else "")
        mode = pick(["se", "pe", "pe_conservative"])
        func = pick(["get_tag", "get_tag_with_ZS_as_XS", "get_cigarbased_AS_tag"])
        m = pick([NEG, 0.0, -12.5, 3.0])
        skip = bool(rng.random() < 0.5)
        outs = {name: io.StringIO() for name in BIN_ARGS}
        loop = {"se": ref.main_single_end, "pe": ref.main_paired_end, "pe_conservative": ref.conservative_main_paired_end}[mode]
        err, counts = None, None
        try:
            got = loop(ref.getReadPairs(io.StringIO(t1, newline=None), io.StringIO(t2, newline=None), skip_repeated_reads=skip),
                       min_score=m, tag_func=getattr(ref, func), **outs)
            counts = {("|".join(key) if isinstance(key, tuple) else key): v for key, v in sorted(got.items())}
        except Exception as exc:
            err = type(exc).__name__
        cases.append({"text": [t1, t2], "mode": mode, "tag_func": func, "min_score": num(m), "skip_repeated": skip,
                      "error": err, "counts": counts, "outputs": {name: outs[name].getvalue() for name in BIN_ARGS}})
    return {"cases": cases}


# ------------------------------------------------------------------ G8
def g8():
    """The reference on synthetic text twins of configs 1, 2, 3 and 5 at 100 k pairs (200 k SAM lines per file): large
    enough for many kernel tiles, chunks and several stripper windows.  Only counts, digests and the summary are kept."""
    cases = []

    def add(name, args, mode, **kw):
        t1, t2, _ = synth.sam_text_pair(**args)
        res = run_reference(t1, t2, mode, **kw)
        for key in ("unit_index", "unit_fwd", "unit_rev"):
            res.pop(key)
        opts = {"tag_func": kw.get("tag_func_name", "get_tag"), "min_score": num(kw.get("min_score", NEG)),
                "skip_repeated": kw.get("skip_repeated", False), "header_sinks": "all"}
        cases.append({"name": name, "mode": mode, "options": opts, "expect": res,
                      "source": {"kind": "synth", "args": args,
                                 "sha224": [hashlib.sha224(t.encode()).hexdigest() for t in (t1, t2)]}})
    add("cfg1_se_cli_200k", dict(n_pairs=200000, seed=1101, profile="bowtie2", paired=False, read_len=50, mixed_ws=0.02,
                                 irregular=0.01), "se", skip_repeated=True)
    add("cfg2_pe_liberal_100k", dict(n_pairs=100000, seed=2102, profile="bowtie2", paired=True, read_len=150, irregular=0.005), "pe")
    add("cfg3_pe_cigar_100k", dict(n_pairs=100000, seed=3103, profile="cigar", paired=True, read_len=150, irregular=0.005), "pe",
        tag_func_name="get_cigarbased_AS_tag")
    add("cfg5_pe_zs_conservative_100k", dict(n_pairs=100000, seed=5105, profile="hisat", paired=True, read_len=150,
                                             irregular=0.005, mixed_ws=0.01), "pe_conservative",
        tag_func_name="get_tag_with_ZS_as_XS", min_score=-40.0)
    return {"cases": cases}


# ------------------------------------------------------------------ G9
def g9():
    """The reference's command line (xenomapper.py:568-743) run as a child process on its own SAM fixtures and on a
    HISAT-style text twin: stdout, stderr, exit code and the output files for a spread of flag combinations.  {P} and
    {S} stand for the two input paths, {O}/<bin>.sam for an output path."""
    import subprocess
    import tempfile
    data_dir = os.path.join(REF_ROOT, "xenomapper", "tests", "data")
    se = ("test_human_in.sam", "test_mouse_in.sam")
    pe = ("paired_end_testdata_human.sam", "paired_end_testdata_mouse.sam")
    zs_args = dict(n_pairs=300, seed=5905, profile="hisat", paired=True, read_len=100, irregular=0.02)
    runner = ("import sys; sys.dont_write_bytecode = True; sys.path.insert(0, %r); "
              "from xenomapper import xenomapper as x; x.main()" % REF_ROOT)
    cases = []

    def add(name, inputs, flags, outputs):
        with tempfile.TemporaryDirectory() as d:
            if inputs[0] == "ref":
                paths = [os.path.join(data_dir, inputs[1]), os.path.join(data_dir, inputs[2])]
                src = {"kind": "ref_data", "files": [inputs[1], inputs[2]]}
            elif inputs[0] == "synth":
                t1, t2, _ = synth.sam_text_pair(**inputs[1])
                paths = [os.path.join(d, "p.sam"), os.path.join(d, "s.sam")]
                for path, text in zip(paths, (t1, t2)):
                    with open(path, "w") as fh:
                        fh.write(text)
                src = {"kind": "synth", "args": inputs[1], "sha224": [hashlib.sha224(t.encode()).hexdigest() for t in (t1, t2)]}
            else:
                paths, src = [], {"kind": "none"}
            argv = []
            if paths:
                argv += ["--primary_sam", paths[0], "--secondary_sam", paths[1]]
            argv += flags
            for b in outputs:
                argv += ["--" + b, os.path.join(d, b + ".sam")]
            proc = subprocess.run([sys.executable, "-c", runner] + argv, capture_output=True, text=True, cwd=d)
            files = {}
            for b in outputs:
                with open(os.path.join(d, b + ".sam")) as fh:
                    text = fh.read()
                files[b] = {"sha224": hashlib.sha224(text.encode("latin-1")).hexdigest(), "len": len(text)}
            cases.append({"name": name, "source": src, "flags": flags, "outputs": outputs, "returncode": proc.returncode,
                          "stdout": {"sha224": hashlib.sha224(proc.stdout.encode("latin-1")).hexdigest(), "len": len(proc.stdout)},
                          # a traceback quotes source lines: only its last line (exception type and message) is kept
                          "stderr": (None if inputs[0] == "none" else
                                     proc.stderr if "Traceback" not in proc.stderr else None),
                          "exception": (proc.stderr.strip().splitlines()[-1] if "Traceback" in proc.stderr else None),
                          "files": files})
    add("se_default_stdout", ("ref",) + se, [], [])
    add("se_all_outputs", ("ref",) + se, [], list(BIN_ARGS))
    add("se_min60_two_outputs", ("ref",) + se, ["--min_score", "60"], ["unresolved", "unassigned"])
    add("pe_liberal_two_outputs", ("ref",) + pe, ["--paired"], ["primary_specific", "secondary_specific"])
    add("pe_conservative_min_all", ("ref",) + pe, ["--paired", "--conservative", "--min_score", "99.5"], list(BIN_ARGS))
    add("pe_cigar_scores", ("ref",) + pe, ["--paired", "--cigar_scores"], ["primary_specific", "primary_multi"])
    add("pe_conservative_without_paired_is_single_end", ("ref",) + pe, ["--conservative"], ["primary_specific", "unresolved"])
    add("pe_use_zs_conservative", ("synth", zs_args), ["--paired", "--conservative", "--use_zs"], list(BIN_ARGS))
    add("pe_use_zs_and_cigar_scores", ("synth", zs_args), ["--paired", "--use_zs", "--cigar_scores"], ["primary_specific"])
    add("version", ("none",), ["--version"], [])
    add("no_inputs_is_a_usage_error", ("none",), [], [])
    return {"cases": cases}


# ------------------------------------------------------------------ G10
def g10():
    """The companion tool's command line (mappability.py:275-331) as a child process on the reference's E. coli
    fixtures: exit status and stdout of each of its three steps (the wiggle of step 3 is the input of step 4), the
    version flag, the usage error and the missing --sam_for_sizes error.  {D} stands for tests/golden/ref_data."""
    import subprocess
    import tempfile
    data_dir = os.path.join(REF_ROOT, "xenomapper", "tests", "data")
    runner = ("import sys; sys.dont_write_bytecode = True; sys.path.insert(0, %r); "
              "from xenomapper import mappability as m; m.main()" % REF_ROOT)
    cases = []
    with tempfile.TemporaryDirectory() as d:
        def run(name, argv_template, keep_stdout=False):
            argv = [a.replace("{D}", data_dir).replace("{T}", d) for a in argv_template]
            proc = subprocess.run([sys.executable, "-c", runner] + argv, capture_output=True, text=True, cwd=d)
            case = {"name": name, "argv": argv_template, "returncode": proc.returncode,
                    "stdout": {"sha224": hashlib.sha224(proc.stdout.encode("latin-1")).hexdigest(), "len": len(proc.stdout)},
                    "exception": (proc.stderr.strip().splitlines()[-1] if "Traceback" in proc.stderr else None)}
            if keep_stdout:
                case["stdout"]["text"] = proc.stdout
            cases.append(case)
            return proc.stdout
        run("simulate_reads_150", ["--fasta", "{D}/test_from_EcoliK12DH10B.fasta", "--readlength", "150"])
        run("simulate_reads_default_length", ["--fasta", "{D}/test_from_EcoliK12DH10B.fasta"])
        wig = run("single_end_wiggle", ["--mapped_test_data", "{D}/test_from_EcoliK12DH10B_150reads.sam"], keep_stdout=True)
        with open(os.path.join(d, "single.wig"), "w") as fh:
            fh.write(wig)
        run("paired_end_wiggle", ["--single_end_wiggle", "{T}/single.wig", "--sam_for_sizes", "{D}/paired_end_testdata_human.sam"])
        run("paired_needs_sam_for_sizes", ["--single_end_wiggle", "{T}/single.wig"])
        run("version", ["--version"])
        run("no_arguments", [])
    return {"cases": cases}


def header_golden():
    """process_headers on the PE fixtures (tests/test_xenomapper.py:29-54): full texts."""
    data_dir = os.path.join(REF_ROOT, "xenomapper", "tests", "data")
    outs = {name: io.StringIO() for name in BIN_ARGS}
    with open(os.path.join(data_dir, "paired_end_testdata_human.sam")) as s1, \
            open(os.path.join(data_dir, "paired_end_testdata_mouse.sam")) as s2:
        ref.process_headers(s1, s2, **outs)
    return {name: outs[name].getvalue() for name in BIN_ARGS}


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    payload = {"g1_mapping_state.json": g1(), "g2_tag_parsers.json": g2(), "g3_end_to_end.json": g3(),
               "g4_headers.json": header_golden(), "g5_errors.json": g5(), "g6_mappability.json": g6(),
               "g7_random_corpus.json": g7(), "g8_large_runs.json": g8(),
               "g9_cli.json": g9(), "g10_mappability_cli.json": g10()}
    for name, obj in payload.items():
        with open(os.path.join(GOLDEN, name), "wt") as fh:
            json.dump(obj, fh, indent=None, separators=(",", ":"), sort_keys=True)
            fh.write("\n")
        print(name, os.path.getsize(os.path.join(GOLDEN, name)), "bytes")
    g3c = payload["g3_end_to_end.json"]["cases"]
    seen = set()
    for c in g3c:
        if c["mode"] != "se":
            seen.update(c["expect"]["counts"])
    print("G3 cases:", len(g3c), "distinct paired tuples covered:", len(seen))


if __name__ == "__main__":
    main()
