"""The SAM column stripper on the GPU (include/xenomapper_strip.h) against the host stripper (xmh_parse), which the CPU
suite pins to the oracle's restatement of the reference's text rules: same records, same consumed bytes, same walk
outcome, same score columns, unit mask, line tables and flagged values -- on random adversarial text, on windows cut
in the middle of files, and on a large synthetic pair of files; then the fused classify pass on the device-resident
columns against the same pass on the host stripper's columns."""
import os
import re

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from tests.test_host_fuzz import sam_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rig():
    from xenomapper_amd import _ffi, _host
    from xenomapper_amd.xenomapper import default_context
    ctx = default_context()
    s = _ffi.Stripper(ctx)
    p = _host.Parser(4)
    yield ctx, s, p
    s.close()
    p.close()


def strip(s, slot, b1, b2, eof1, eof2, score_mode, paired, keep_halo, max_records, skip=False):
    s.reserve(slot, max(len(b1), len(b2), 1), max_records)
    for f, b in enumerate((b1, b2)):
        if len(b):
            s.staging(slot, f)[:len(b)] = np.frombuffer(b, dtype=np.uint8)
    return s.run(slot, len(b1), eof1, len(b2), eof2, score_mode, paired, skip, keep_halo, max_records)


def compare(s, p, b1, b2, eof1, eof2, score_mode, paired, keep_halo, max_records, slot=0, skip=False):
    from xenomapper_amd import _ffi, _host
    r1, r2 = np.frombuffer(b1, dtype=np.uint8), np.frombuffer(b2, dtype=np.uint8)
    got = strip(s, slot, b1, b2, eof1, eof2, score_mode, paired, keep_halo, max_records, skip)
    try:
        want = p.parse(r1, 0, len(r1), eof1, r2, 0, len(r2), eof2, score_mode, paired, skip, keep_halo, max_records)
    except _host.NonAsciiInput:
        assert got.non_ascii
        return got, None
    assert not got.non_ascii
    ctxt = (b1, b2, eof1, eof2, score_mode, paired, keep_halo, max_records, skip)
    if got.overflow:                        # the skipping walk met more lines than the tables hold: the caller's cue to use xmh_parse
        assert skip and max(b1.count(b"\n") + b1.count(b"\r"), b2.count(b"\n") + b2.count(b"\r")) >= max_records, ctxt
        return got, want
    assert got.n == want.n, ctxt
    assert (got.ended, got.starved, got.mismatch_at) == (want.ended, want.starved, want.mismatch_at), ctxt
    assert got.consumed == want.consumed, ctxt
    assert got.consumed_lines == want.consumed_lines, ctxt
    n = got.n
    for f in (0, 1):
        assert np.array_equal(got.line_off[f], want.line_off[f].astype(np.uint32)), ctxt
        assert np.array_equal(got.line_len[f], want.line_len[f]), ctxt
    assert sorted(got.exc) == sorted(want.exc), ctxt
    cols = got.cols
    for c in range(4):
        assert np.array_equal(cols[c], want.cols[c]), (c, ctxt)
    bits = lambda a: np.unpackbits(np.ascontiguousarray(a).view(np.uint8), bitorder="little")[:n]
    assert np.array_equal(bits(got.unit_bits), bits(want.unit_bits)), ctxt
    if score_mode == 2:                     # NM + the CIGAR operations: the packed columns the classify kernel reads
        for f in (0, 1):
            nm, off, ops = want.csr[f]
            cnt, tile, packed = _ffi.cigar_pack(off, ops)
            gnm, gcnt, gtile, gops = s.cigar_columns(slot, f, n)
            assert np.array_equal(gnm, nm) and np.array_equal(gcnt, cnt), ctxt
            assert np.array_equal(gtile[:_ffi.cigar_tiles(n)], tile[:_ffi.cigar_tiles(n)]), ctxt
            assert np.array_equal(gops[:packed.shape[0]], packed), ctxt
    if n:                                   # the writer on the adopted tables writes what it writes on its own
        idx = np.arange(1 if paired else 0, n, dtype=np.uint32)
        want_text = [bytes(p.emit(paired, b, idx)) for b in (0, 1, 4)]
        p.adopt_lines(r1, 0, r2, 0, n, got.tables)
        assert [bytes(p.emit(paired, b, idx)) for b in (0, 1, 4)] == want_text, ctxt
    return got, want


@settings(max_examples=int(os.environ.get("XM_FUZZ_EXAMPLES", "300")), deadline=None, suppress_health_check=list(HealthCheck))
@given(texts=sam_pair(), score_mode=st.sampled_from([0, 1, 2]), paired=st.booleans(), keep_halo=st.booleans(),
       eofs=st.tuples(st.booleans(), st.booleans()), max_records=st.sampled_from([1 << 16, 1 << 16, 3, 1]),
       cut=st.tuples(st.integers(0, 40), st.integers(0, 40)), skip=st.booleans())
def test_gpu_stripper_agrees_with_the_host_stripper_on_random_text(rig, texts, score_mode, paired, keep_halo, eofs, max_records, cut, skip):
    _ctx, s, p = rig
    b1, b2 = texts[0].encode("ascii"), texts[1].encode("ascii")
    if not eofs[0]:
        b1 = b1[:max(0, len(b1) - cut[0])]          # a window that stops somewhere inside the file
    if not eofs[1]:
        b2 = b2[:max(0, len(b2) - cut[1])]
    compare(s, p, b1, b2, eofs[0], eofs[1], score_mode, paired, keep_halo, max_records, skip=skip)


def test_non_ascii_windows_are_refused_as_a_whole(rig):
    _ctx, s, p = rig
    good = b"r1\t0\tc\t1\t9\t4M\t*\t0\t0\tACGT\tIIII\tAS:i:3\tXS:i:1\n"
    bad = good.replace(b"ACGT", b"AC\xc3\xa9")
    for b1, b2 in ((bad, good), (good, bad), (good + bad[:30], good)):
        got = strip(s, 0, b1, b2, True, True, 0, False, False, 16)
        assert got.non_ascii
    assert not strip(s, 0, good, good, True, True, 0, False, False, 16).non_ascii


_CIGARS = ["150M", "100M2I48M", "20S130M", "75M1D75M", "10S50M3I40M2D47M", "*", "150=", "5H145M", "1M1I" * 150, "3S" + "2M1D" * 49]


def _big_pair(n_pairs, seed, crlf=False, repeats=False):
    """Two files of 2 x n_pairs records in the same order: names, tags and a few oddities drawn at random.  repeats: some
    records are followed by further lines with the same name (secondary alignments), a different number in each file."""
    rng = np.random.default_rng(seed)
    out = [[], []]
    seq = "ACGT" * 37 + "AC"
    qual = "I" * 150
    for i in range(n_pairs):
        name = "read%d/%d" % (i, int(rng.integers(0, 1000)))
        for mate in (0, 1):
            if repeats:
                name = "read%d.%d" % (i, mate)
            for f in (0, 1):
              for _rep in range(1 + (int(rng.integers(0, 4)) if repeats and rng.integers(0, 5) == 0 else 0)):
                a, x = int(rng.integers(-60, 1)), int(rng.integers(-80, 1))
                tags = []
                r = int(rng.integers(0, 40))
                if r != 0:
                    tags.append("AS:i:%d" % a)
                if r % 3:
                    tags.append("XS:i:%d" % x)
                if r == 7:
                    tags.append("ZS:i:%d" % x)
                if r == 11:
                    tags.append("XS:f:1.5")              # flagged: not an integer
                if r == 13:
                    tags.append("RG:Z:BASS")             # flagged: a second field containing "AS"
                tags += ["XN:i:0", "XM:i:%d" % (r % 5)] + (["NM:i:%d" % (r % 5)] if r != 19 else []) + ["YT:Z:CP"]
                sep = " " if r == 17 else "\t"
                cigar = _CIGARS[int(rng.integers(0, len(_CIGARS)))] if r % 4 == 0 else "150M"
                fields = [name, str(83 + 16 * mate), "chr%d" % (1 + f), str(1000 + i), "42", cigar, "=", str(1200 + i),
                          "350", seq, qual] + tags
                out[f].append(sep.join(fields))
    nl = "\r\n" if crlf else "\n"
    return (nl.join(out[0]) + nl).encode(), (nl.join(out[1]) + nl).encode()


@pytest.mark.parametrize("crlf", [False, True])
def test_large_files_in_windows_with_halo_like_the_file_path(rig, crlf):
    ctx, s, p = rig
    from xenomapper_amd import _ffi
    b1, b2 = _big_pair(30000, 5, crlf)
    pos = [0, 0]
    window = 3 << 20
    blocks = 0
    totals_g, totals_h = np.zeros(64, np.uint64), np.zeros(64, np.uint64)
    while True:
        w1, w2 = b1[pos[0]:pos[0] + window], b2[pos[1]:pos[1] + window]
        e1, e2 = pos[0] + window >= len(b1), pos[1] + window >= len(b2)
        got, want = compare(s, p, w1, w2, e1, e2, 0, True, True, 1 << 20, slot=blocks & 1)
        assert got.n > 1000
        # the fused pass on the device-resident columns == the same pass on the host stripper's columns
        # (flagged values are absent in both; the file path patches them before it classifies)
        code, idx, off, counts = s.classify(blocks & 1, _ffi.MODE_PE_LIBERAL, got.n, -2**31)
        hcode, hidx, hoff, hcounts = ctx.classify_compact(_ffi.MODE_PE_LIBERAL, *want.cols, want.unit_bits, -2**31)
        assert np.array_equal(code, hcode) and np.array_equal(idx, hidx) and np.array_equal(off, hoff)
        assert np.array_equal(counts, hcounts)
        totals_g += counts
        totals_h += hcounts
        blocks += 1
        if got.ended:
            break
        assert got.consumed[0] > 0
        pos[0] += got.consumed[0]
        pos[1] += got.consumed[1]
    assert blocks >= 4 and int(totals_g.sum()) == 30000 and np.array_equal(totals_g, totals_h)


def test_zs_plugin_and_single_end_units(rig):
    _ctx, s, p = rig
    b1, b2 = _big_pair(2000, 9)
    for score_mode, paired in ((1, True), (1, False), (0, False)):
        compare(s, p, b1, b2, True, True, score_mode, paired, paired, 1 << 20)


def test_record_limit_and_empty_windows(rig):
    _ctx, s, p = rig
    b1, b2 = _big_pair(300, 3)
    for max_records in (1, 2, 63, 64, 65, 600, 601):
        compare(s, p, b1, b2, True, True, 0, True, True, max_records)
        compare(s, p, b1, b2, False, False, 0, True, False, max_records)
    compare(s, p, b"", b"", True, True, 0, True, True, 16)
    compare(s, p, b1, b"", True, True, 0, True, True, 16)
    compare(s, p, b"", b2, False, True, 0, False, False, 16)
    compare(s, p, b1[:-1], b2[:-1], True, True, 0, True, True, 1 << 20)        # last line without a terminator
    compare(s, p, b1[:-1], b2[:-1], False, False, 0, True, True, 1 << 20)
    compare(s, p, b1 + b"\n" + b1, b2 + b"\n" + b2, True, True, 0, True, True, 1 << 20)   # a blank line ends the walk


def test_skipping_walk_over_runs_of_repeated_names_in_windows(rig):
    _ctx, s, p = rig
    b1, b2 = _big_pair(6000, 21, repeats=True)
    pos, window, blocks, yielded = [0, 0], 2 << 20, 0, 0
    while True:
        w1, w2 = b1[pos[0]:pos[0] + window], b2[pos[1]:pos[1] + window]
        e1, e2 = pos[0] + window >= len(b1), pos[1] + window >= len(b2)
        got, _want = compare(s, p, w1, w2, e1, e2, 0, False, False, 1 << 20, slot=blocks & 1, skip=True)
        yielded += got.n
        blocks += 1
        if got.ended:
            break
        assert got.consumed[0] > 0
        pos[0] += got.consumed[0]
        pos[1] += got.consumed[1]
    assert blocks >= 4 and yielded == 12000
    for paired, halo in ((True, True), (True, False), (False, True)):
        compare(s, p, b1[:3 << 20], b2[:3 << 20], False, False, 1, paired, halo, 1 << 20, skip=True)
    # more lines than the tables hold: reported, not guessed
    got = strip(s, 0, b1[:1 << 20], b2[:1 << 20], False, False, 0, False, False, 100, skip=True)
    assert got.overflow


def test_cigar_plugin_columns_and_the_fused_pass_on_them(rig):
    ctx, s, p = rig
    from xenomapper_amd import _ffi
    b1, b2 = _big_pair(20000, 33)
    for skip in (False, True):
        got, want = compare(s, p, b1, b2, True, True, 2, True, True, 1 << 20, skip=skip)
        assert got.n == (20000 if skip else 40000)        # mates share a name: the skipping walk yields one of them
        code, idx, off, counts = s.classify(0, _ffi.MODE_PE_LIBERAL, got.n, -2**31)
        h = ctx.classify_compact_cigar(_ffi.MODE_PE_LIBERAL, want.csr[0][0], want.csr[0][1], want.csr[0][2], want.cols[1],
                                       want.csr[1][0], want.csr[1][1], want.csr[1][2], want.cols[3], want.unit_bits, -2**31)
        assert np.array_equal(code, h[0]) and np.array_equal(idx, h[1]) and np.array_equal(off, h[2]) and np.array_equal(counts, h[3])
    # flagged lines: a non-integer NM, an operation length of 2^28 (a line too short for its NM to count is not flagged)
    lines = [b"r1\t0\tc\t1\t9\t4M\t*\t0\t0\tACGT\tIIII\tNM:i:1.5\tXS:i:3", b"r2\t0\tc\t1\tNM:i:1",
             b"r3\t0\tc\t1\t9\t268435456M5S\t*\t0\t0\tACGT\tIIII\tNM:i:2", b"r4\t0\tc\t1\t9\t4M2S\t*\t0\t0\tA\tI\tNM:i:0"]
    text = b"\n".join(lines) + b"\n"
    got, want = compare(s, p, text, text, True, True, 2, False, False, 64)
    assert sorted(k for _, _, k in got.exc) == [1, 1, 4, 4]


_ALPHA = "ASXZNM:i0123456789-+.xyIDHP=*"
_TOKEN = st.text(alphabet=_ALPHA, min_size=1, max_size=45)
_SEP = st.sampled_from(["\t", "\t", "\t", " ", "\t\t", " \t", "\x0b", "\x1c\x1f", "   "])


@st.composite
def _long_lines(draw):
    """A few lines of many fields of random lengths, so that field boundaries fall at every position of the 8-byte words and
    32-byte load steps the parse kernel works in, with the odd separator, leading / trailing white space, tags in front of
    the eleventh field and lines that stop short of it."""
    n = draw(st.integers(1, 6))
    names = [draw(st.text(alphabet="abcdefgh/._0123", min_size=1, max_size=30)) for _ in range(n)]
    files = []
    for _f in (0, 1):
        lines = []
        for nm in names:
            toks = [nm] + [draw(_TOKEN) for _ in range(draw(st.integers(0, 18)))]
            if len(toks) > 5 and draw(st.booleans()):
                toks[5] = draw(st.sampled_from(["10M", "3S7M2I", "5M1D5M", "*", "12", "M", "1M" * 130]))
            if len(toks) > 11 and draw(st.booleans()):
                toks[draw(st.integers(11, len(toks) - 1))] = draw(st.sampled_from(["AS:i:-7", "XS:i:12", "ZS:i:0", "NM:i:3", "NM:i:x", "AS:f:1.5"]))
            seps = [draw(_SEP) for _ in toks]
            text = "".join(s + t for s, t in zip([""] + seps[1:], toks))
            if draw(st.integers(0, 7)) == 0:
                text = draw(_SEP) + text
            if draw(st.integers(0, 7)) == 0:
                text = text + draw(_SEP)
            lines.append(text)
        files.append(lines)
    nl = draw(st.sampled_from(["\n", "\r\n", "\r"]))
    pad = draw(st.integers(0, 31))                     # shifts every line against the word and load-step grid
    head = "p" * pad + nl if pad else ""
    return [(head + nl.join(lines) + nl).encode("ascii") for lines in files], bool(pad)


@settings(max_examples=int(os.environ.get("XM_FUZZ_EXAMPLES", "300")), deadline=None, suppress_health_check=list(HealthCheck))
@given(data=_long_lines(), score_mode=st.sampled_from([0, 1, 2]), paired=st.booleans(), skip=st.booleans())
def test_long_random_fields_at_every_alignment(rig, data, score_mode, paired, skip):
    _ctx, s, p = rig
    (b1, b2), _padded = data
    compare(s, p, b1, b2, True, True, score_mode, paired, False, 1 << 12, skip=skip)


def test_whole_file_path_with_either_stripper_writes_the_same_bytes(tmp_path):
    """tools/check_strip_scale.py at test size: 150 k units per mode (paired liberal, conservative + ZS, single-end with
    the skipping walk, --cigar_scores) through classify_sam_files with the GPU stripper and with the host stripper."""
    import json
    import subprocess
    import sys
    from tests import helpers as H
    out = subprocess.run([sys.executable, os.path.join(H.REPO, "tools", "check_strip_scale.py"), "--pairs", "150000", "--dir", str(tmp_path)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert len(rep) == 4 and all(v["identical"] and v["units"] == 150000 for v in rep.values())


def _unpack_cigar(cnt, ops):
    """Packed CIGAR columns -> the operations of every record (xenomapper_hip.h: a count byte of 255 means "255 or more", the
    record's operations are then followed by one trailer word n_ops << 4 | 15)."""
    out, pos = [], 0
    for c in cnt.tolist():
        n = c
        if c == 255:
            while not ((int(ops[pos + n]) & 15) == 15 and (int(ops[pos + n]) >> 4) == n):
                n += 1
        out.append([int(v) for v in ops[pos:pos + n]])
        pos += n + (1 if c == 255 else 0)
    return out


_PLAIN_INT = re.compile(r"^[+-]?[0-9]{1,10}$")


def _plain_int32(text):
    return bool(_PLAIN_INT.match(text)) and int(text) <= 2**31 - 1 and int(text) >= -(2**31 - 1)


def _flag_reason(fields, tag, score_mode):
    """Why the stripper may (and must) decline to vouch for column `tag` of this line, from the text alone -- None: it must
    deliver the value.  The plugins' rules (xenomapper.py:176-191, :193-206, :228-256): optional fields fields[11:] that CONTAIN
    the tag; two of them raise; the value is what follows the last ':'.  The kernels vouch for plain int32 literals only."""
    opt = fields[11:]
    if score_mode == 2 and tag == "AS":                               # get_cigarbased_AS_tag: NM (first match) + fields[5]
        nm = [x for x in opt if "NM" in x]
        if not nm:
            return None
        if len(fields) < 6:
            return "short"
        if not _plain_int32(nm[0].split(":")[-1]):
            return "nonint"
        if any(int(n) >= 2**28 for n, _op in re.findall(r"([0-9]+)([MIDNSHPX=])", fields[5])):
            return "biglen"
        return None
    name = "ZS" if (score_mode == 1 and tag == "XS") else tag
    hits = [x for x in opt if name in x]
    if not hits:
        return None
    if len(hits) > 1:
        return "dup"
    return None if _plain_int32(hits[0].split(":")[-1]) else "nonint"


@settings(max_examples=int(os.environ.get("XM_FUZZ_EXAMPLES", "300")), deadline=None, suppress_health_check=list(HealthCheck))
@given(texts=sam_pair(), score_mode=st.sampled_from([0, 1, 2]), paired=st.booleans(), skip=st.booleans())
def test_gpu_stripper_agrees_with_the_oracle_directly(rig, texts, score_mode, paired, skip):
    """Not through the host stripper: the oracle's restatement of getReadPairs and of the three plugins on the same text --
    record count, walk outcome, unit mask, every score the kernels vouch for (a flag wherever they do not), line spans."""
    import io
    from tests.helpers import ORACLE, NEG
    _ctx, s, _p = rig
    t1, t2 = texts
    b1, b2 = t1.encode("ascii"), t2.encode("ascii")
    got = strip(s, 0, b1, b2, True, True, score_mode, paired, False, 1 << 12, skip)
    pairs, err = [], None
    try:
        for pr in ORACLE.read_pairs(io.StringIO(t1, newline=None), io.StringIO(t2, newline=None), skip):
            pairs.append(pr)
    except AssertionError:
        err = "mismatch"
    assert got.n == len(pairs) and (got.mismatch_at >= 0) == (err == "mismatch")
    if err != "mismatch":
        assert got.ended
    names = [p[0][0] for p in pairs]
    cols = got.cols
    flags = np.unpackbits(np.ascontiguousarray(cols[4] if len(cols) > 4 else got.unit_bits).view(np.uint8), bitorder="little")[:got.n].tolist()
    assert flags == ([1] * len(pairs) if not paired else [int(i > 0 and names[i] == names[i - 1]) for i in range(len(names))])
    scorer = [ORACLE.tag_score, ORACLE.tag_score_zs, ORACLE.cigar_score][score_mode]
    exc = {(k, c) for k, c, _ in got.exc}
    cig = [None, None]
    if score_mode == 2 and got.n:
        for f in (0, 1):
            nm, cnt, _tile, ops = s.cigar_columns(0, f, got.n)
            cig[f] = (nm, _unpack_cigar(cnt, ops))
    raws = (b1, b2)
    for k, (f1, f2) in enumerate(pairs):
        for f, fields in enumerate((f1, f2)):
            st0 = int(got.line_off[f][k])
            assert raws[f][st0:st0 + int(got.line_len[f][k])].decode().split() == fields
            assert int(got.norm_len[f][k]) == len("\t".join(fields))
            for c, tag in ((2 * f, "AS"), (2 * f + 1, "XS")):
                try:
                    want, failed = scorer(fields, tag=tag), False
                except Exception:
                    want, failed = None, True
                # a flag must be JUSTIFIED by the text, and a justified flag must be there (an over-flagging stripper -- which the
                # host path would quietly absorb by re-reading the line -- fails here)
                why = _flag_reason(fields, tag, score_mode)
                assert ((k, c) in exc) == (why is not None), (fields, tag, why)
                if (k, c) in exc:
                    if why in ("dup", "short"):
                        assert failed, (fields, tag, why)               # the reference raises (ValueError :189-190 / IndexError)
                    continue
                assert not failed, (fields, tag)
                if score_mode == 2 and tag == "AS":
                    nm, ops = cig[f][0][k], cig[f][1][k]
                    have = NEG if nm == -2**31 else -6 * int(nm) - sum(
                        (5 + 3 * (v >> 4)) if (v & 15) in (1, 2) else (2 * (v >> 4) if (v & 15) == 4 else 0) for v in ops)
                else:
                    have = NEG if cols[c][k] == -2**31 else int(cols[c][k])
                assert have == want, (fields, tag, have, want)


def test_closing_a_stripper_leaves_no_dangling_stream_in_the_context():
    """the context orders workspace calls across streams and used to keep the handle of the last stream -- a closed
    Stripper's slot stream.  xm_strip_destroy now hands the stream back (xm_workspace_release): a classify on another stream
    afterwards must simply work."""
    import torch
    from xenomapper_amd import _ffi, synth
    from tests import helpers as H
    ctx = _ffi.Context(0)
    try:
        s = _ffi.Stripper(ctx)
        t1, t2, _ = synth.sam_text_pair(n_pairs=300, seed=3, profile="bowtie2", paired=True, read_len=50)
        bodies = []
        for t in (t1, t2):
            at = 0
            while t[at] == "@":
                at = t.index("\n", at) + 1
            bodies.append(t[at:].encode("ascii"))
        blk = strip(s, 0, bodies[0], bodies[1], True, True, 0, True, False, 1 << 12)
        code, idx, off, counts = s.classify(0, _ffi.MODE_PE_LIBERAL, blk.n, _ffi.ABSENT)     # on the slot's own stream
        want_counts = counts.copy()
        cols = s.columns(0, blk.n)
        s.close()                                                                           # destroys that stream
        dev = torch.device("cuda:0")
        d = [torch.from_numpy(c).to(dev) for c in cols[:4]] + [torch.from_numpy(cols[4].view(np.int64)).to(dev)]
        n = blk.n
        out_code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
        out_idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        out_off = torch.zeros(8, dtype=torch.int64, device=dev)
        out_counts = torch.zeros(64, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream()
        for stream in (side, None):                                                          # another stream, then torch's own
            ctx.classify_compact_dev(_ffi.MODE_PE_LIBERAL, *d, _ffi.ABSENT, out_code, out_idx, out_off, out_counts, stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(out_counts.cpu().numpy().astype(np.uint64), want_counts)
            assert np.array_equal(out_off.cpu().numpy().astype(np.uint64), off)
        assert ctx.workspace_is_clean()
    finally:
        ctx.close()


@pytest.mark.parametrize("mode", ["pe", "pe_conservative", "se"])
def test_outputs_gathered_on_the_device_equal_the_oracle_and_the_host_writer(tmp_path, monkeypatch, mode):
    """xm_strip_fetch_bins through the file path (windows of 1 MB: ~25 windows): the six outputs gathered on the device equal the
    oracle's and the host writer's (XENOMAPPER_GPU_SAM_BINS=0) byte for byte, every window took the device route for plain
    tab-separated text -- and with a sprinkling of lines whose fields are separated by mixed white space (the reference re-joins
    them with tabs, xenomapper.py:103 + the print calls) the windows that hold a wanted one fall back to the host writer while
    the others stay on the device, outputs still equal."""
    import io
    from tests import helpers as H
    from tests.test_file_fuzz_gpu import oracle_run, SCORERS
    from xenomapper_amd import xenomapper as xm
    monkeypatch.setattr(xm, "FILE_WINDOW_BYTES", 1 << 20)
    monkeypatch.delenv("XENOMAPPER_SAM_READ_AHEAD", raising=False)    # (the default first, the option below)
    paired = mode != "se"
    for mixed in (0.0, 0.0002):
        t1, t2, _info = H.synth.sam_text_pair(n_pairs=30_000, seed=17, profile="bowtie2", paired=paired, read_len=100, mixed_ws=mixed,
                                              irregular=0.01, header_pg=False)
        bodies = []
        for k, text in enumerate((t1, t2)):
            at = 0
            while text[at] == "@":
                at = text.index("\n", at) + 1
            bodies.append(text[at:])
            (tmp_path / ("g%d.sam" % k)).write_text(text)
        paths = [str(tmp_path / "g0.sam"), str(tmp_path / "g1.sam")]
        want_texts, want_counts, want_err = oracle_run(bodies[0], bodies[1], mode, SCORERS["get_tag"], H.NEG, not paired)
        assert want_err is None
        got = {}
        for bins in ("1", "0"):
            monkeypatch.setenv("XENOMAPPER_GPU_SAM_BINS", bins)
            outs = {name: io.StringIO() for name in H.STATES}
            counts = xm.classify_sam_files(paths[0], paths[1], paired=paired, conservative=mode == "pe_conservative", **outs)
            prof = dict(xm.LAST_FILE_PROFILE)
            assert dict(counts) == dict(want_counts)
            got[bins] = [outs[name].getvalue() for name in H.STATES]
            assert got[bins] == want_texts, (mode, mixed, bins)
            assert prof.get("sam_windows", 0) >= 5
            if bins == "0":
                assert prof.get("sam_windows_device_bins", 0) == 0
            elif mixed == 0.0:
                assert prof.get("sam_windows_device_bins", 0) == prof["sam_windows"], prof
            else:
                assert 0 < prof.get("sam_windows_device_bins", 0) < prof["sam_windows"], prof
            assert prof.get("sam_windows_read_ahead", 0) == 0
        # XENOMAPPER_SAM_READ_AHEAD=1: every window's bytes read, and sent over the link, while the window in front is stripped --
        # behind a gap in the other slot's buffer, the tail of the window in front put into the gap when its walk has said where
        # it stopped (xm_strip_begin_behind / xm_strip_set_lead: the first `lead` bytes of the buffer are no text).  The same
        # outputs; most windows go that way (not the first two, not those behind a window the host writer still reads).
        monkeypatch.setenv("XENOMAPPER_GPU_SAM_BINS", "1")
        monkeypatch.setenv("XENOMAPPER_SAM_READ_AHEAD", "1")
        outs = {name: io.StringIO() for name in H.STATES}
        counts = xm.classify_sam_files(paths[0], paths[1], paired=paired, conservative=mode == "pe_conservative", **outs)
        prof = dict(xm.LAST_FILE_PROFILE)
        assert dict(counts) == dict(want_counts) and [outs[name].getvalue() for name in H.STATES] == want_texts
        if mixed == 0.0:
            assert prof.get("sam_windows_read_ahead", 0) >= prof["sam_windows"] * 2 // 3, prof
        else:
            assert prof.get("sam_windows_read_ahead", 0) > 0, prof
        monkeypatch.delenv("XENOMAPPER_SAM_READ_AHEAD", raising=False)


def test_overlapping_units_outgrow_the_output_stream_and_go_to_the_host_writer(tmp_path, monkeypatch):
    """Every record carries the same name, so every record closes a unit with the one in front (xenomapper.py:402-405: `previous`
    always advances) and every line is printed twice; with equal scores in both files every unit is `unresolved` and prints from BOTH
    files (:439-444): four times the text of a window, twice what the device's output stream holds -- xm_strip_fetch_bins declines
    (status 2), the host writer takes those windows, and the outputs equal the oracle's.  The same file with only the
    primary-specific sink given fits and stays on the device."""
    import io
    from tests import helpers as H
    from tests.test_file_fuzz_gpu import oracle_run, SCORERS
    from xenomapper_amd import xenomapper as xm
    monkeypatch.setattr(xm, "FILE_WINDOW_BYTES", 1 << 20)
    monkeypatch.delenv("XENOMAPPER_SAM_READ_AHEAD", raising=False)    # (with it the buffers hold a window and the room in front of it)
    xm.release_buffers()                                             # (the process-wide stripper may have grown in a test before: this one counts on its sizes)
    line = "samename\t%d\tchr1\t%d\t30\t50M\t=\t%d\t0\t" + "ACGT" * 12 + "AC\t" + "F" * 50 + "\tAS:i:-5\tXS:i:-9\n"
    text = "".join(line % (99 if k % 2 == 0 else 147, 100 + k, 300 + k) for k in range(40_000))
    paths = []
    for k in (0, 1):
        p = tmp_path / ("o%d.sam" % k)
        p.write_text(text)
        paths.append(str(p))
    want_texts, want_counts, want_err = oracle_run(text, text, "pe", SCORERS["get_tag"], H.NEG, False)
    assert want_err is None and sum(want_counts.values()) == 39_999 and len(want_texts[H.STATES.index("unresolved")]) > 4 * len(text) - 1000
    outs = {name: io.StringIO() for name in H.STATES}
    counts = xm.classify_sam_files(paths[0], paths[1], paired=True, **outs)
    prof = dict(xm.LAST_FILE_PROFILE)
    assert dict(counts) == dict(want_counts)
    assert [outs[name].getvalue() for name in H.STATES] == want_texts
    assert prof.get("sam_windows", 0) >= 3 and prof.get("sam_windows_device_bins", 0) < prof["sam_windows"], prof
    # one sink only: nothing of `unresolved` is printed, the stream is empty and fits
    only = io.StringIO()
    counts = xm.classify_sam_files(paths[0], paths[1], primary_specific=only, paired=True)
    assert dict(counts) == dict(want_counts) and only.getvalue() == want_texts[0]
    assert xm.LAST_FILE_PROFILE.get("sam_windows_device_bins", 0) == xm.LAST_FILE_PROFILE.get("sam_windows", -1)
