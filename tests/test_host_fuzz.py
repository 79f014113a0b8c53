"""Property tests: random SAM-like text (odd whitespace, newline styles, adversarial tags and CIGARs, repeated and
mismatching names) through the C++ stripper must agree with the oracle's text-level restatement -- record
count, names, unit mask, every score the stripper vouches for, and an exception wherever it does not.  CPU only."""
import io
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from tests import helpers as H
from tests.helpers import ORACLE, NEG

ABSENT = -2**31

WS = st.sampled_from(["\t", " ", "\t\t", "  ", " \t", "\x0b", "\x0c", "\x1c", "\x1f"])
NAMES = st.sampled_from(["r1", "r2", "r2", "read/3", "q", "AS", "x:1"])
NUM = st.one_of(st.integers(-300, 300).map(str), st.sampled_from(["0", "-0", "+7", "007", "2147483647", "2147483648",
                                                                 "-2147483648", "1.5", "1e2", "inf", "nan", "", "1_0", "x"]))
TAGN = st.sampled_from(["AS", "XS", "ZS", "NM", "YS", "XN", "MD", "RG", "xAS", "ASx"])
TAG = st.builds(lambda t, ty, v: "%s:%s:%s" % (t, ty, v), TAGN, st.sampled_from(["i", "f", "Z", "A"]), NUM) | \
    st.sampled_from(["RG:Z:BASS", "XS:A:+", "AS", "NM", "YT:Z:UU", "ZS:i:4:5"])
CIGAR = st.one_of(st.sampled_from(["*", "50M", "10M2I3M1D4M6S", "5H10M", "0010S40M", "10Q5S", "5S10", "M5S", "3=2X1P4N5M",
                                   "268435456M", "268435455S", "1I1D1S"]),
                  st.lists(st.tuples(st.integers(0, 400), st.sampled_from("MIDNSHP=XQ")), max_size=6)
                  .map(lambda ops: "".join("%d%s" % o for o in ops)))


@st.composite
def line(draw, name=None):
    n_fixed = draw(st.sampled_from([11, 11, 11, 11, 3, 6, 1]))
    fields = [name if name is not None else draw(NAMES)]
    base = ["0", "chr1", "7", "30", draw(CIGAR), "*", "0", "0", "ACGT", "IIII"]
    fields += base[:n_fixed - 1]
    if n_fixed == 11:
        fields += draw(st.lists(TAG, max_size=5))
    seps = [draw(WS) for _ in fields]
    text = "".join(s + f for s, f in zip([""] + seps[1:], fields))
    if draw(st.integers(0, 9)) == 0:
        text = draw(WS) + text + draw(WS)
    return text


@st.composite
def sam_pair(draw):
    n = draw(st.integers(0, 12))
    names = [draw(NAMES) for _ in range(n)]
    l1 = [draw(line(name=nm)) for nm in names]
    l2 = [draw(line(name=nm)) for nm in names]
    if n and draw(st.integers(0, 7)) == 0:                       # a name mismatch somewhere
        l2[draw(st.integers(0, n - 1))] = draw(line(name="other"))
    if draw(st.integers(0, 5)) == 0:                             # a blank line ends the walk
        (l1 if draw(st.booleans()) else l2).insert(draw(st.integers(0, n)), draw(st.sampled_from(["", " ", "\t"])))
    nl = draw(st.sampled_from(["\n", "\r\n", "\r"]))
    tail1, tail2 = draw(st.booleans()), draw(st.booleans())
    return nl.join(l1) + (nl if tail1 and l1 else ""), nl.join(l2) + (nl if tail2 and l2 else "")


@pytest.fixture(scope="module")
def parser():
    from xenomapper_amd import _host
    p = _host.Parser(2)
    yield p
    p.close()


def oracle_walk(t1, t2, skip):
    pairs, err = [], None
    try:
        for pr in ORACLE.read_pairs(io.StringIO(t1, newline=None), io.StringIO(t2, newline=None), skip):
            pairs.append(pr)
    except AssertionError:
        err = "mismatch"
    return pairs, err


@settings(max_examples=int(os.environ.get('XM_FUZZ_EXAMPLES', '400')), deadline=None, suppress_health_check=list(HealthCheck))
@given(texts=sam_pair(), score_mode=st.sampled_from([0, 1, 2]), paired=st.booleans(), skip=st.booleans())
def test_stripper_agrees_with_oracle(parser, texts, score_mode, paired, skip):
    t1, t2 = texts
    r1 = np.frombuffer(t1.encode("ascii"), dtype=np.uint8)
    r2 = np.frombuffer(t2.encode("ascii"), dtype=np.uint8)
    block = parser.parse(r1, 0, len(r1), True, r2, 0, len(r2), True, score_mode, paired, skip, False, 1 << 20)
    pairs, err = oracle_walk(t1, t2, skip)
    assert block.n == len(pairs)
    assert (block.mismatch_at >= 0) == (err == "mismatch")
    if err != "mismatch":
        assert block.ended
    scorer = [ORACLE.tag_score, ORACLE.tag_score_zs, ORACLE.cigar_score][score_mode]
    flags = np.unpackbits(block.unit_bits.view(np.uint8), bitorder="little")[:block.n].tolist()
    names = [p[0][0] for p in pairs]
    assert flags == ([1] * len(pairs) if not paired else [int(i > 0 and names[i] == names[i - 1]) for i in range(len(names))])
    exc = {(k, c) for k, c, _ in block.exc}
    for k, (f1, f2) in enumerate(pairs):
        for f, fields in enumerate((f1, f2)):
            start = int(block.line_off[f][k])
            raw = (r1, r2)[f]
            assert bytes(raw[start:start + int(block.line_len[f][k])]).decode().split() == fields
            for c, tag in ((2 * f, "AS"), (2 * f + 1, "XS")):
                try:
                    want = scorer(fields, tag=tag)
                    failed = False
                except Exception:
                    failed = True
                if (k, c) in exc:
                    continue                                    # the stripper declined: Python resolves it
                assert not failed, (fields, tag)               # never vouch for a value the reference rejects
                if score_mode == 2 and tag == "AS":
                    nm_col, off, ops = block.csr[f]
                    ops_k = ops[int(off[k]):int(off[k + 1])]
                    got = NEG if nm_col[k] == ABSENT else -6 * int(nm_col[k]) - sum(
                        (5 + 3 * (int(v) >> 4)) if (int(v) & 15) in (1, 2) else (2 * (int(v) >> 4) if (int(v) & 15) == 4 else 0)
                        for v in ops_k)
                else:
                    got = NEG if block.cols[c][k] == ABSENT else int(block.cols[c][k])
                assert got == want, (fields, tag, got, want)
    # the writer: every record's normalised line
    if block.n:
        idx = np.arange(block.n, dtype=np.uint32)
        out = bytes(parser.emit(False, 0, idx)).decode()
        assert out == "".join("\t".join(p[0]) + "\n" for p in pairs)
        out2 = bytes(parser.emit(False, 4, idx)).decode()
        assert out2 == "".join("\t".join(p[0]) + "\n" + "\t".join(p[1]) + "\n" for p in pairs)
