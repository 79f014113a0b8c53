"""Property test on the GPU: random SAM-like files through the file fast path (C++ stripper -> kernels -> C++
writer) and through the iterator-level loops must behave like the oracle's restatement of the reference:
same six outputs, same category_counts, and for malformed input the same exception type after the same
partial output."""
import io
import os
import tempfile

import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from tests import helpers as H
from tests.helpers import ORACLE, NEG
from tests.test_host_fuzz import sam_pair

pytestmark = pytest.mark.gpu

SCORERS = {"get_tag": ORACLE.tag_score, "get_tag_with_ZS_as_XS": ORACLE.tag_score_zs,
           "get_cigarbased_AS_tag": ORACLE.cigar_score}


def oracle_run(t1, t2, mode, scorer, m, skip):
    outs = [io.StringIO() for _ in range(6)]
    err = None
    counts = None
    try:
        pairs = ORACLE.read_pairs(io.StringIO(t1, newline=None), io.StringIO(t2, newline=None), skip)
        if mode == "se":
            res = ORACLE.run_single_end(pairs, outs, m, scorer)
        else:
            res = ORACLE.run_paired_end(pairs, outs, m, scorer, conservative=mode == "pe_conservative")
        counts = res.named_counts(mode != "se")
    except Exception as exc:
        err = type(exc).__name__
    return [o.getvalue() for o in outs], counts, err


@settings(max_examples=int(os.environ.get('XM_FUZZ_EXAMPLES', '250')), deadline=None, suppress_health_check=list(HealthCheck))
@given(texts=sam_pair(), mode=st.sampled_from(["se", "pe", "pe_conservative"]),
       func=st.sampled_from(sorted(SCORERS)), m=st.sampled_from([NEG, 0.0, -12.5, 3.0]), skip=st.booleans(),
       via=st.sampled_from(["files", "api", "python"]))
def test_random_inputs_behave_like_the_oracle(texts, mode, func, m, skip, via):
    from xenomapper_amd import xenomapper as xm
    t1, t2 = texts
    want_texts, want_counts, want_err = oracle_run(t1, t2, mode, SCORERS[func], m, skip)
    if want_err == "OverflowError":
        return
    outs = {name: io.StringIO() for name in H.STATES}
    got_err, counts = None, None
    with tempfile.TemporaryDirectory() as d:
        p1, p2 = os.path.join(d, "a.sam"), os.path.join(d, "b.sam")
        with open(p1, "w", newline="") as f:
            f.write(t1)
        with open(p2, "w", newline="") as f:
            f.write(t2)
        try:
            if via == "files":
                counts = xm.classify_sam_files(p1, p2, paired=mode != "se", conservative=mode == "pe_conservative",
                                               min_score=m, tag_func=getattr(xm, func), skip_repeated_reads=skip, **outs)
            else:
                loop = {"se": xm.main_single_end, "pe": xm.main_paired_end,
                        "pe_conservative": xm.conservative_main_paired_end}[mode]
                with open(p1) as f1, open(p2) as f2:
                    pairs = xm.getReadPairs(f1, f2, skip_repeated_reads=skip)    # regular files: the C++ text path
                    if via == "python":
                        pairs = (pr for pr in pairs)                             # any other iterable: split in Python
                    counts = loop(pairs, min_score=m, tag_func=getattr(xm, func), **outs)
        except OverflowError:
            return                       # a CIGAR length / NM beyond the packed columns: documented limit
        except Exception as exc:
            got_err = type(exc).__name__
    assert got_err == want_err, (t1, t2)
    got_texts = [outs[name].getvalue() for name in H.STATES]
    assert got_texts == want_texts, (t1, t2)
    if want_err is None:
        assert dict(counts) == dict(want_counts)
