"""Property tests: random SAM-like text (odd whitespace, newline styles, adversarial tags and CIGARs, repeated and
mismatching names) through the C++ stripper must agree with the oracle's text-level restatement -- record
count, names, unit mask, every score the stripper vouches for, and an exception wherever it does not.  CPU only."""
import io
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

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


# ---------------------------------------------------------------------------------------------- BAM decoder
import struct
import zlib

from oracle import bam_oracle

_INT_TYPES = {"c": (-128, 127), "C": (0, 255), "s": (-32768, 32767), "S": (0, 65535), "i": (-2**31, 2**31 - 1),
              "I": (0, 2**32 - 1)}
_FLOATS = st.sampled_from([0.0, 1.5, -2.25, 1e10, 3.0e-5, 123456.0, 1234567.0, -0.1, float("inf")])
_TAGNAME = st.text(alphabet="ABCXYZabc019", min_size=2, max_size=2)


@st.composite
def bam_tag(draw):
    tag = draw(_TAGNAME).encode()
    kind = draw(st.sampled_from(list("AcCsSiIfZHB")))
    if kind == "A":
        return tag + b"A" + draw(st.sampled_from(list("+-!~aZ"))).encode()
    if kind in _INT_TYPES:
        lo, hi = _INT_TYPES[kind]
        v = draw(st.one_of(st.integers(lo, hi), st.sampled_from([lo, hi, 0])))
        return tag + kind.encode() + struct.pack(bam_oracle._SCALAR[kind], v)
    if kind == "f":
        return tag + b"f" + struct.pack("<f", draw(_FLOATS))
    if kind == "Z":
        # (now and then a value longer than the device printer's 64-byte trips)
        return tag + b"Z" + draw(st.one_of(st.text(alphabet="abcXYZ09 :;,*", max_size=12), st.text(alphabet="abcXYZ09 :;,*", max_size=12),
                                           st.text(alphabet="abcXYZ09 :;,*", min_size=60, max_size=150))).encode() + b"\0"
    if kind == "H":
        return tag + b"H" + draw(st.text(alphabet="0123456789ABCDEF", max_size=8)).encode() + b"\0"
    sub = draw(st.sampled_from(list("cCsSiIf")))
    if sub == "f":
        vals = draw(st.lists(_FLOATS, max_size=5))
    else:
        lo, hi = _INT_TYPES[sub]
        vals = draw(st.lists(st.one_of(st.integers(lo, hi), st.sampled_from([lo, hi])), max_size=40))
    return tag + b"B" + sub.encode() + struct.pack("<I", len(vals)) + b"".join(struct.pack(bam_oracle._SCALAR[sub], v) for v in vals)


@st.composite
def bam_record(draw, n_ref):
    name = draw(st.one_of(st.text(alphabet="abcXYZ019:/._", min_size=1, max_size=30), st.text(alphabet="abcXYZ019:/._", min_size=1, max_size=30),
                          st.text(alphabet="abcXYZ019:/._", min_size=64, max_size=200))).encode() + b"\0"
    cigar = draw(st.lists(st.tuples(st.integers(0, 2**28 - 1), st.integers(0, 9)), max_size=6))
    l_seq = draw(st.sampled_from([0, 1, 2, 7, 50, 151, 127, 128, 129, 300]))
    seq = bytes(draw(st.lists(st.integers(0, 255), min_size=(l_seq + 1) // 2, max_size=(l_seq + 1) // 2)))
    if l_seq and draw(st.booleans()):
        qual = bytes([0xFF] * l_seq)
    else:
        qual = bytes(draw(st.lists(st.integers(0, 93), min_size=l_seq, max_size=l_seq)))
    tag_list = draw(st.lists(bam_tag(), max_size=6))
    if draw(st.integers(0, 5)) == 0:                                  # the long-CIGAR convention: placeholder + CG:B:I
        cigar = [(l_seq if draw(st.integers(0, 3)) else l_seq + 1, 4), (draw(st.integers(0, 5000)), 3)][:draw(st.integers(1, 2))]
        real = draw(st.lists(st.tuples(st.integers(0, 2**28 - 1), st.integers(0, 8)), max_size=9))
        sub = draw(st.sampled_from(["I", "I", "I", "i"]))
        cg = b"CGB" + sub.encode() + struct.pack("<I", len(real)) + b"".join(struct.pack("<I", (ln << 4) | op) for ln, op in real)
        tag_list.insert(draw(st.integers(0, len(tag_list))), cg)
    tags = b"".join(tag_list)
    core = struct.pack("<iiBBHHHIiii", draw(st.integers(-1, n_ref)), draw(st.integers(-1, 2**31 - 2)), len(name),
                       draw(st.integers(0, 255)), 4680, len(cigar), draw(st.integers(0, 65535)), l_seq,
                       draw(st.integers(-1, n_ref)), draw(st.integers(-1, 2**31 - 2)), draw(st.integers(-2**31, 2**31 - 1)))
    body = core + name + b"".join(struct.pack("<I", (ln << 4) | op) for ln, op in cigar) + seq + qual + tags
    return struct.pack("<I", len(body)) + body


@st.composite
def bam_image(draw):
    refs = draw(st.lists(st.text(alphabet="chrXY0123_", min_size=1, max_size=12), max_size=3))
    text = draw(st.sampled_from(["", "@HD\tVN:1.0\n", "@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:10\n@PG\tID:x\n"])).encode()
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for r in refs:
        rn = r.encode() + b"\0"
        head += struct.pack("<i", len(rn)) + rn + struct.pack("<i", 1000)
    recs = draw(st.lists(bam_record(len(refs)), max_size=25))
    payload = head + b"".join(recs) * draw(st.sampled_from([1, 1, 7]))
    out, at = [], 0
    while at < len(payload):
        step = draw(st.sampled_from([1, 5, 33, 200, 3000, 65280]))
        part = payload[at:at + step]
        comp = zlib.compressobj(draw(st.sampled_from([0, 1, 6])), zlib.DEFLATED, -15)
        body = comp.compress(part) + comp.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(body) + 25) + body +
                   struct.pack("<II", zlib.crc32(part), len(part)))
        at += step
    if draw(st.booleans()):
        out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)


@settings(max_examples=int(os.environ.get('XM_FUZZ_EXAMPLES', '300')), deadline=None, suppress_health_check=list(HealthCheck))
@given(image=bam_image(), threads=st.sampled_from([1, 3, 8]), cap=st.sampled_from([1 << 10, 1 << 14, 1 << 20]))
def test_bam_decoder_agrees_with_the_spec_restatement(image, threads, cap):
    """Random alignment records (every optional-field type, extreme integers, empty and odd-length sequences) in BGZF
    blocks of random sizes: the native decoder must print what the oracle's restatement of the BAM layout prints."""
    from xenomapper_amd import _host
    want_header, want_lines = bam_oracle.bam_to_sam(image)
    r = _host.BamReader(np.frombuffer(image, dtype=np.uint8), threads)
    buf = np.empty(cap, dtype=np.uint8)
    parts = []
    grown = 0
    while not r.eof:
        n = r.read_into(buf, 0)
        if n == 0 and not r.eof:                                   # a line longer than the buffer: the caller grows it
            buf = np.empty(2 * buf.shape[0], dtype=np.uint8)
            grown += 1
            assert grown < 20
        parts.append(bytes(buf[:n]))
    got_header = r.header()
    r.close()
    assert got_header == want_header
    assert b"".join(parts).decode("latin-1") == "".join(l + "\n" for l in want_lines)


@settings(max_examples=int(os.environ.get('XM_FUZZ_EXAMPLES', '200')), deadline=None, suppress_health_check=list(HealthCheck))
@given(image=bam_image(), threads=st.sampled_from([1, 3, 8]), data=st.data())
def test_record_printer_of_the_gpu_bam_path_prints_the_decoder_s_text(image, threads, data):
    """xmh_bam_open_header + xmh_bam_walk + xmh_bam_print -- what the GPU BAM path leaves to the host: the record chain of an
    inflated window and the SAM text of (some of) its records -- on the same random images: every record printed (dense and
    sparse) is the line the oracle's restatement of the BAM layout prints, records no sink wants are skipped with an empty
    table entry, a text buffer that is too small is reported with the size it takes and nothing is written behind it, and a
    window cut anywhere yields exactly the complete records in front of the cut.  (CPU only: this file also runs under ASan.)"""
    import gzip
    from xenomapper_amd import _host
    _want_header, want_lines = bam_oracle.bam_to_sam(image)
    raw = np.frombuffer(gzip.decompress(image), dtype=np.uint8).copy() if image else np.zeros(0, np.uint8)
    r = _host.BamReader(np.frombuffer(image, dtype=np.uint8), threads, header_only=True)
    try:
        start = r.records_start()
        rec = np.empty(raw.shape[0] // 36 + 8, dtype=np.uint32)
        n, stop = _host.bam_walk(raw.ctypes.data, raw.shape[0], start, rec)
        assert n == len(want_lines) and stop == raw.shape[0]
        # a window cut short: the complete records in front of the cut, and where the first incomplete one begins
        cut = data.draw(st.integers(min_value=start, max_value=raw.shape[0]))
        n_cut, stop_cut = _host.bam_walk(raw.ctypes.data, cut, start, rec.copy())
        ends = [int(rec[k + 1]) if k + 1 < n else raw.shape[0] for k in range(n)]
        assert n_cut == sum(1 for e in ends if e <= cut) and stop_cut == (int(rec[n_cut]) if n_cut < n else raw.shape[0])
        want = [l.encode("latin-1") for l in want_lines]
        loff, llen = np.empty(n + 1, dtype=np.uint32), np.empty(n + 1, dtype=np.uint32)
        for sparse in (False, True):
            mask = None
            if data.draw(st.booleans()) and n:
                mask = np.array(data.draw(st.lists(st.integers(0, 1), min_size=n, max_size=n)), dtype=np.uint8)
            guard = 64
            text = np.full(16, 0xEE, dtype=np.uint8)
            got = r.print_records(raw.ctypes.data, rec.ctypes.data, n, text[:text.shape[0] - guard if text.shape[0] > guard else 0],
                                  loff, llen, sparse, mask)
            if n and (mask is None or mask.any()):
                assert got < 0                                             # too small: the size it takes, nothing written
                assert (text == 0xEE).all()
                text = np.full(-got + guard, 0xEE, dtype=np.uint8)
                got = r.print_records(raw.ctypes.data, rec.ctypes.data, n, text[:-guard], loff, llen, sparse, mask)
            assert got >= 0 and (text[text.shape[0] - guard:] == 0xEE).all()
            for k in range(n):
                if mask is not None and not mask[k]:
                    assert int(llen[k]) == 0
                    continue
                line = bytes(text[int(loff[k]):int(loff[k]) + int(llen[k])])
                assert line == want[k], (k, sparse)
                assert text[int(loff[k]) + int(llen[k])] == 0x0A
    finally:
        r.close()


def test_sparse_printing_falls_back_to_two_passes_when_its_estimate_passes_4_gib():
    """xmh_bam_print(sparse) reserves every thread's worst case (5 x record bytes + per record 128 + two reference names) and
    addresses its text with 32 bits: a window whose ESTIMATE passes 4 GiB used to end in 'ValueError: xmh_bam_print' (ADVICE r5;
    a large carried tail, XENOMAPPER_BAM_WINDOW_MB >= ~800).  Here the estimate is driven up by a 600 KB reference name that no
    record uses: 4 000 records estimate 4.8 GB, their real text is 200 KB -- the binding prints in two passes instead, and the
    lines are the oracle's."""
    import gzip
    from xenomapper_amd import _host
    long_ref = "r" * 600_000
    recs = []
    for k in range(4_000):
        name = ("q%05d" % k).encode() + b"\0"
        body = struct.pack("<iiBBHHHIiii", 0, k, len(name), 30, 4680, 1, 0, 4, -1, -1, 0) + name + struct.pack("<I", (4 << 4)) + \
            bytes([0x12, 0x48]) + bytes([30, 31, 32, 33]) + b"ASi" + struct.pack("<i", -k)
        recs.append(struct.pack("<I", len(body)) + body)
    image = _bam_image_of(recs, refs=("chr1", long_ref), aligned=True)
    _header, want_lines = bam_oracle.bam_to_sam(image)
    raw = np.frombuffer(gzip.decompress(image), dtype=np.uint8).copy()
    r = _host.BamReader(np.frombuffer(image, dtype=np.uint8), 4, header_only=True)
    try:
        rec = np.empty(len(recs) + 8, dtype=np.uint32)
        n, stop = _host.bam_walk(raw.ctypes.data, raw.shape[0], r.records_start(), rec)
        assert n == len(recs) and stop == raw.shape[0]
        loff, llen = np.empty(n + 1, dtype=np.uint32), np.empty(n + 1, dtype=np.uint32)
        text = np.empty(1 << 20, dtype=np.uint8)
        got = r.print_records(raw.ctypes.data, rec.ctypes.data, n, text, loff, llen, True, None)
        assert 0 < got <= text.shape[0]
        for k in (0, 1, n // 2, n - 1):
            assert bytes(text[int(loff[k]):int(loff[k]) + int(llen[k])]) == want_lines[k].encode("latin-1")
    finally:
        r.close()


# ---- BAM line descriptions (xmh_bam_read_pre / xmh_parse_pre) against the text rules ------------------------------------

_SCORE_TAGS = st.sampled_from(["AS", "XS", "ZS", "NM", "AS", "XS", "YS", "XA", "SA", "MN", "NH"])


@st.composite
def score_tag(draw):
    """Optional fields around the four tags the plugins look for: typed integers at and beyond the int32 edges, floats and
    characters under those names, and strings that merely CONTAIN the letters (substring match, duplicate error)."""
    tag = draw(_SCORE_TAGS).encode()
    kind = draw(st.sampled_from(list("cCsSiIiiIfAZZH")))
    if kind in _INT_TYPES:
        lo, hi = _INT_TYPES[kind]
        v = draw(st.one_of(st.integers(lo, hi), st.sampled_from([lo, hi, 0, max(lo, -2**31 + 1), min(hi, 2**31 - 1)])))
        return tag + kind.encode() + struct.pack(bam_oracle._SCALAR[kind], v)
    if kind == "f":
        return tag + b"f" + struct.pack("<f", draw(st.sampled_from([0.0, 12.0, -3.5, 1e10])))
    if kind == "A":
        return tag + b"A" + draw(st.sampled_from(list("+-S7"))).encode()
    if kind == "Z":
        return tag + b"Z" + draw(st.text(alphabet="ASXZNM12:- ", max_size=10)).encode() + b"\0"
    return tag + b"H" + draw(st.text(alphabet="0123456789ABCDEF", max_size=6)).encode() + b"\0"


def _bam_image_of(records, refs=("chr1", "chrX"), aligned=False):
    """aligned: every BGZF block begins with a record (what samtools writes); else blocks of 3000 bytes that cut records."""
    head = b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", len(refs))
    for r in refs:
        rn = r.encode() + b"\0"
        head += struct.pack("<i", len(rn)) + rn + struct.pack("<i", 1000)
    payload = head + b"".join(records)
    if aligned:
        # "share": the header shares its block with the first records (the record chain then starts inside the block)
        parts, cur = ([head], b"") if aligned != "share" else ([], head)
        for rec in records:
            if cur and len(cur) + len(rec) > 3000:
                parts.append(cur)
                cur = b""
            cur += rec
        parts.append(cur)
        parts = [q for k, q in enumerate(parts) if q or k == 0] or [head]
    else:
        parts = [payload[at:at + 3000] for at in range(0, len(payload), 3000)]
    out = []
    for part in parts:
        comp = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = comp.compress(part) + comp.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(body) + 25) + body +
                   struct.pack("<II", zlib.crc32(part), len(part)))
    return b"".join(out)


@st.composite
def bam_file_pair(draw, aligned=False):
    n = draw(st.integers(0, 40))
    names = []
    for _ in range(n):
        if names and draw(st.integers(0, 2)) == 0:
            names.append(names[-1])                                  # mates / repeated names
        else:
            names.append(draw(st.text(alphabet="abXY01:/", min_size=1, max_size=8)))
    odd = draw(st.integers(0, 9)) == 0                               # now and then a name the text rules would split
    images = []
    for f in (0, 1):
        recs = []
        for k, nm in enumerate(names):
            name = ((nm + " x") if (odd and k == n // 2 and f == 0) else nm).encode() + b"\0"
            cigar = draw(st.lists(st.tuples(st.integers(0, 2**28 - 1), st.integers(0, 10)), max_size=5))
            l_seq = draw(st.sampled_from([0, 3, 20]))
            seq = bytes(draw(st.lists(st.integers(0, 255), min_size=(l_seq + 1) // 2, max_size=(l_seq + 1) // 2)))
            qual = bytes(draw(st.lists(st.integers(0, 93), min_size=l_seq, max_size=l_seq)))
            tags = b"".join(draw(st.lists(st.one_of(score_tag(), score_tag(), bam_tag()), max_size=5)))
            core = struct.pack("<iiBBHHHIiii", draw(st.integers(-1, 1)), draw(st.integers(-1, 10**6)), len(name), 30, 4680,
                               len(cigar), draw(st.integers(0, 4095)), l_seq, -1, -1, 0)
            body = core + name + b"".join(struct.pack("<I", (ln << 4) | op) for ln, op in cigar) + seq + qual + tags
            recs.append(struct.pack("<I", len(body)) + body)
        if f == 1 and recs and draw(st.integers(0, 5)) == 0:
            recs = recs[:draw(st.integers(0, len(recs)))]            # the second file ends early
        images.append(_bam_image_of(recs, aligned=aligned))
    return images


def _decode_with_descriptions(image, threads, cap):
    from xenomapper_amd import _host
    r = _host.BamReader(np.frombuffer(image, dtype=np.uint8), threads)
    buf = np.empty(1 << 22, dtype=np.uint8)
    at, pres, opss = 0, [], []
    while not r.eof:
        view = buf[:min(buf.shape[0], at + cap)]                     # small capacities: lines left pending between calls
        got, pre, ops = r.read_into_pre(view, at)
        if got == 0 and not r.eof:
            cap *= 2
            assert cap <= 1 << 22
            continue
        q = pre.copy()
        q[:, _host.PRE_OPS_AT] += np.uint32(sum(o.shape[0] for o in opss))
        pres.append(q)
        opss.append(ops.copy())
        at += got
    r.close()
    pre = np.concatenate(pres) if pres else np.zeros((0, _host.PRE_WORDS), dtype=np.uint32)
    ops = np.concatenate(opss) if opss else np.zeros(0, dtype=np.uint32)
    return buf[:at].copy(), pre, ops


def _blocks_equal(a, b, cigar):
    assert (a.n, a.consumed, a.consumed_lines, a.ended, a.starved, a.mismatch_at) == \
           (b.n, b.consumed, b.consumed_lines, b.ended, b.starved, b.mismatch_at)
    for x, y in zip(a.cols, b.cols):
        if cigar:
            x, y = x[:0], y[:0]                                      # the AS columns are not filled in CIGAR mode ...
        assert np.array_equal(x, y)
    assert np.array_equal(a.cols[1], b.cols[1]) and np.array_equal(a.cols[3], b.cols[3])      # ... the XS columns always are
    assert np.array_equal(a.unit_bits[:(a.n + 63) // 64], b.unit_bits[:(b.n + 63) // 64])
    if cigar:
        for f in (0, 1):
            for x, y in zip(a.csr[f], b.csr[f]):
                assert np.array_equal(x, y)
    for f in (0, 1):
        assert np.array_equal(a.line_off[f], b.line_off[f]) and np.array_equal(a.line_len[f], b.line_len[f])
    assert a.exc == b.exc


@settings(max_examples=int(os.environ.get('XM_FUZZ_EXAMPLES', '120')), deadline=None, suppress_health_check=list(HealthCheck))
@given(images=bam_file_pair(), threads=st.sampled_from([1, 4]), cap=st.sampled_from([300, 1 << 12, 1 << 20]),
       score_mode=st.sampled_from([0, 1, 2]), paired=st.booleans(), skip=st.booleans(), halo=st.booleans(),
       max_records=st.sampled_from([1 << 20, 7]))
def test_bam_line_descriptions_give_what_the_text_rules_give(images, threads, cap, score_mode, paired, skip, halo, max_records):
    """xmh_parse_pre (the stripper fed by the BAM decoder's own knowledge of every line: typed AS / XS / ZS / NM, CIGAR
    operations) must return exactly what xmh_parse returns after tokenising the printed text -- columns, CIGAR CSR,
    unit mask, line index, exceptions (non-integers, duplicate and substring matches), consumed bytes and lines -- or
    decline (a line marked XMH_PRE_WEIRD), and it may only decline for lines the text rules could really split."""
    from xenomapper_amd import _host
    t1, p1, o1 = _decode_with_descriptions(images[0], threads, cap)
    t2, p2, o2 = _decode_with_descriptions(images[1], threads, cap)
    v1, v2 = _host.pre_view(p1), _host.pre_view(p2)
    for text, pre in ((t1, v1), (t2, v2)):
        lines = bytes(text).split(b"\n")[:-1] if text.shape[0] else []
        assert [len(l) for l in lines] == pre["line_len"].tolist()
        for l, q in zip(lines, pre):
            risky = any(c <= 0x20 and c != 9 or c >= 0x7F for c in l) or l.startswith(b"\t") or b"\t\t" in l
            assert bool(q["flags"] & 1) == risky, l
    a = _host.Parser(threads)
    b = _host.Parser(threads)
    try:
        want = a.parse(t1, 0, t1.shape[0], True, t2, 0, t2.shape[0], True, score_mode, paired, skip, halo, max_records)
        got = b.parse_pre(t1, 0, t1.shape[0], True, p1, o1, t2, 0, t2.shape[0], True, p2, o2, score_mode, paired, skip, halo,
                          max_records)
        if got is None:
            assert bool((v1["flags"] & 1).any() or (v2["flags"] & 1).any())
        else:
            assert not bool((v1["flags"] & 1).any() or (v2["flags"] & 1).any())
            _blocks_equal(got, want, score_mode == 2)
        # a window that ends inside a line: the same lines are seen complete
        if t1.shape[0] > 5 and t2.shape[0] > 5 and got is not None:
            c1, c2 = t1.shape[0] - 3, t2.shape[0] - 3
            want = a.parse(t1, 0, c1, False, t2, 0, c2, False, score_mode, paired, skip, halo, max_records)
            got = b.parse_pre(t1, 0, c1, False, p1, o1, t2, 0, c2, False, p2, o2, score_mode, paired, skip, halo, max_records)
            _blocks_equal(got, want, score_mode == 2)
    finally:
        a.close()
        b.close()
