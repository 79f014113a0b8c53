"""Live differential test of the oracle against the reference itself -- runs only where the reference is mounted
(the build container); everywhere else the recorded vectors of tests/golden/ stand in for it.  Nothing is copied:
the reference is imported from /root/reference, fed random adversarial text pairs, and the oracle must raise the same
exception type, have written the same six texts by then and return the same counts."""
import io
import os
import sys

import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from tests import helpers as H
from tests.helpers import ORACLE, NEG
from tests.test_host_fuzz import sam_pair

REF_ROOT = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF_ROOT, "xenomapper")),
                                reason="the reference is only mounted in the build container")

SCORERS = {"get_tag": ORACLE.tag_score, "get_tag_with_ZS_as_XS": ORACLE.tag_score_zs,
           "get_cigarbased_AS_tag": ORACLE.cigar_score}


def _reference():
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from xenomapper import xenomapper as ref
    return ref


@settings(max_examples=int(os.environ.get("XM_FUZZ_EXAMPLES", "300")), deadline=None, suppress_health_check=list(HealthCheck))
@given(texts=sam_pair(), mode=st.sampled_from(["se", "pe", "pe_conservative"]), func=st.sampled_from(sorted(SCORERS)),
       m=st.sampled_from([NEG, 0.0, -12.5, 3.0, float("inf")]), skip=st.booleans())
def test_oracle_behaves_like_the_reference(texts, mode, func, m, skip):
    ref = _reference()
    t1, t2 = texts
    want_outs = {name: io.StringIO() for name in H.STATES}
    want_err, want_counts = None, None
    try:
        loop = {"se": ref.main_single_end, "pe": ref.main_paired_end, "pe_conservative": ref.conservative_main_paired_end}[mode]
        got = loop(ref.getReadPairs(io.StringIO(t1, newline=None), io.StringIO(t2, newline=None), skip_repeated_reads=skip),
                   min_score=m, tag_func=getattr(ref, func), **want_outs)
        want_counts = {("|".join(k) if isinstance(k, tuple) else k): v for k, v in got.items()}
    except Exception as exc:
        want_err = type(exc).__name__
    outs = [io.StringIO() for _ in range(6)]
    err, counts = None, None
    try:
        pairs = ORACLE.read_pairs(io.StringIO(t1, newline=None), io.StringIO(t2, newline=None), skip)
        if mode == "se":
            res = ORACLE.run_single_end(pairs, outs, m, SCORERS[func])
        else:
            res = ORACLE.run_paired_end(pairs, outs, m, SCORERS[func], conservative=mode == "pe_conservative")
        counts = {("|".join(k) if isinstance(k, tuple) else k): v for k, v in res.named_counts(mode != "se").items()}
    except Exception as exc:
        err = type(exc).__name__
    assert err == want_err, (t1, t2)
    assert [o.getvalue() for o in outs] == [want_outs[name].getvalue() for name in H.STATES], (t1, t2)
    if err is None:
        assert counts == want_counts


def test_mappability_host_functions_behave_like_the_reference():
    """The text and small-list functions of the companion tool (wiggle and FASTA readers, smoothing, normalisation) side by
    side with the reference's on odd inputs: tracks without values, files without declarations, repeated chromosomes, names
    with '=' in them, nameless FASTA records, text in front of the first header."""
    _reference()                                           # skips where the reference is not mounted
    import importlib
    ref = importlib.import_module("xenomapper.mappability")
    from xenomapper_amd import mappability as ours
    dec = "fixedStep\tchrom=%s\tstart=1\tstep=1\n"
    wiggles = ["", dec % "a" + "1\n0.5\n" + dec % "b" + dec % "c=d" + "2\n", "3\n4\n", dec % "a",
               dec % "a" + "1\n" + dec % "a" + "7\n8\n", dec % "z" + "1e-3\n" + dec % "y" + dec % "x"]
    for text in wiggles:
        a, b = ref.Mappability(chromosome_sizes={}), ours.Mappability(chromosome_sizes={})
        a.from_wiggle(io.StringIO(text))
        b.from_wiggle(io.StringIO(text))
        assert dict(a) == dict(b) and a.chromosome_sizes == b.chromosome_sizes, text
    for text in ["", "junk\n>a desc\nAC\nGT\n>\nTT\n>b\n\n>c\nA\n", ">x\n", "no header\n", ">>double\nAA\n  >  sp\nCC\n"]:
        assert list(ref.parse_fasta(io.StringIO(text))) == list(ours.parse_fasta(io.StringIO(text))), text
    for values in ([1, 2, 3, 10, 0, 0, 5] * 5, [0.1, 0.25, 7.0], [3]):
        for width in (10, 2, 1):
            assert ref.smoothed_list(values, width) == ours.smoothed_list(values, width)
        assert ref.normalised_list(values) == ours.normalised_list(values)
