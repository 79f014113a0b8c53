"""The C++ SAM column stripper / line writer (libxenomapper_host.so) against the oracle's text-level
restatement, on every golden end-to-end input plus newline / whitespace / error edge cases.  CPU only."""
import io
import os

import numpy as np
import pytest

from tests import helpers as H
from tests.helpers import ORACLE, NEG

G3 = H.golden("g3_end_to_end.json")["cases"]
ABSENT = -2**31
SCORERS = {"get_tag": (0, ORACLE.tag_score), "get_tag_with_ZS_as_XS": (1, ORACLE.tag_score_zs),
           "get_cigarbased_AS_tag": (2, ORACLE.cigar_score)}


@pytest.fixture(scope="module")
def parser():
    from xenomapper_amd import _host, build
    build.build_host()
    p = _host.Parser(3)
    yield p
    p.close()


def body_of(text):
    """Bytes of a SAM text after its header lines."""
    from xenomapper_amd import xenomapper as xm
    raw = np.frombuffer(text.encode("ascii"), dtype=np.uint8)
    return raw, xm._record_start(raw)


def parse_all(parser, t1, t2, score_mode, paired, skip, keep_halo=False, max_records=1 << 22):
    r1, p1 = body_of(t1)
    r2, p2 = body_of(t2)
    return parser.parse(r1, p1, len(r1) - p1, True, r2, p2, len(r2) - p2, True, score_mode, paired, skip, keep_halo,
                        max_records), (r1, p1), (r2, p2)


def oracle_pairs(t1, t2, skip):
    s1, s2 = io.StringIO(t1), io.StringIO(t2)
    ORACLE.read_header(s1), ORACLE.read_header(s2)
    return list(ORACLE.read_pairs(s1, s2, skip))


@pytest.mark.parametrize("case", G3, ids=[c["name"] for c in G3])
def test_columns_and_lines_match_oracle(parser, case):
    t1, t2 = H.case_texts(case)
    score_mode, scorer = SCORERS[case["options"]["tag_func"]]
    paired = case["mode"] != "se"
    skip = case["options"]["skip_repeated"]
    block, (r1, p1), (r2, p2) = parse_all(parser, t1, t2, score_mode, paired, skip)
    pairs = oracle_pairs(t1, t2, skip)
    assert block.n == len(pairs) == case["expect"]["n_records"]
    assert block.ended and block.mismatch_at == -1
    names = [p[0][0] for p in pairs]
    flags = np.unpackbits(block.unit_bits.view(np.uint8), bitorder="little")[:block.n]
    want_flags = [1] * len(pairs) if not paired else [0] + [int(names[i] == names[i - 1]) for i in range(1, len(names))]
    assert flags.tolist() == want_flags
    exc = {(k, c) for k, c, _ in block.exc}
    for k, (f1, f2) in enumerate(pairs):
        for c, (fields, tag) in enumerate(((f1, "AS"), (f1, "XS"), (f2, "AS"), (f2, "XS"))):
            if (k, c) in exc:
                continue
            if score_mode == 2 and tag == "AS":
                nm_col, off, ops = block.csr[c // 2]
                ops_k = ops[int(off[k]):int(off[k + 1])]
                score = ABSENT if nm_col[k] == ABSENT else -6 * int(nm_col[k]) - sum(
                    (5 + 3 * (int(v) >> 4)) if (int(v) & 15) in (1, 2) else (2 * (int(v) >> 4) if (int(v) & 15) == 4 else 0)
                    for v in ops_k)
                want = scorer(fields, tag="AS")
                assert (score == ABSENT and want == NEG) or score == want, (k, fields)
            else:
                want = scorer(fields, tag=tag)
                got = int(block.cols[c][k])
                assert (got == ABSENT and want == NEG) or got == want, (k, c, fields)
        # the line table addresses the record's text
        for f, (raw, pos, fields) in enumerate(((r1, p1, f1), (r2, p2, f2))):
            start = pos + int(block.line_off[f][k])
            assert bytes(raw[start:start + int(block.line_len[f][k])]).decode().split() == fields


@pytest.mark.parametrize("name", ["ref_se", "ref_pe_liberal", "ref_pe_conservative", "all36_liberal", "all36_se",
                                   "cfg1_se", "cfg5_pe_zs_conservative"])
def test_emit_reproduces_reference_bins(parser, name):
    """xmh_emit output for the golden units equals the reference's bin texts (hash of header + body)."""
    import hashlib
    case = {c["name"]: c for c in G3}[name]
    t1, t2 = H.case_texts(case)
    score_mode, _ = SCORERS[case["options"]["tag_func"]]
    paired = case["mode"] != "se"
    block, _, _ = parse_all(parser, t1, t2, score_mode, paired, case["options"]["skip_repeated"])
    exp = case["expect"]
    mode = H.MODES[case["mode"]]
    bins = [ORACLE.bin_of(mode, int(f), int(r)) for f, r in zip(exp["unit_fwd"], exp["unit_rev"])]
    heads = [io.StringIO() for _ in range(6)]
    ORACLE.write_headers(io.StringIO(t1), io.StringIO(t2), heads)
    for b, state in enumerate(H.STATES):
        idx = np.array([i for i, bb in zip(exp["unit_index"], bins) if bb == b], dtype=np.uint32)
        body = bytes(parser.emit(paired, b, idx)).decode("ascii")
        text = heads[b].getvalue() + body
        assert hashlib.sha224(text.encode("latin-1")).hexdigest() == exp["bins"][state]["sha224"], state


def sam(lines, nl="\n", tail=True):
    return nl.join(lines) + (nl if tail else "")


def rec(name, *opts, sep="\t"):
    return sep.join([name, "0", "chr1", "1", "30", "10M", "*", "0", "0", "ACGT", "IIII"] + list(opts))


def parse_text(parser, t1, t2, score_mode=0, paired=False, skip=False):
    r1 = np.frombuffer(t1.encode("latin-1"), dtype=np.uint8)
    r2 = np.frombuffer(t2.encode("latin-1"), dtype=np.uint8)
    return parser.parse(r1, 0, len(r1), True, r2, 0, len(r2), True, score_mode, paired, skip, False, 1 << 20)


@pytest.mark.parametrize("nl,tail", [("\n", True), ("\n", False), ("\r\n", True), ("\r\n", False), ("\r", True)])
def test_newline_conventions(parser, nl, tail):
    lines = [rec("a", "AS:i:5"), rec("b", "AS:i:7", "XS:i:3"), rec("c")]
    t = sam(lines, nl, tail)
    block = parse_text(parser, t, t)
    want = list(ORACLE.read_pairs(io.StringIO(t, newline=None), io.StringIO(t, newline=None)))
    assert block.n == len(want) == 3
    assert block.cols[0].tolist() == [5, 7, ABSENT] and block.cols[1].tolist() == [ABSENT, 3, ABSENT]


def test_blank_line_and_shorter_file_end_the_walk(parser):
    a = sam([rec("a"), rec("b"), "", rec("c")])
    b = sam([rec("a"), rec("b"), rec("c"), rec("d")])
    assert parse_text(parser, a, b).n == 2
    assert parse_text(parser, b, a).n == 2
    assert parse_text(parser, sam([rec("a"), "   \t "]), b).n == 1          # whitespace-only line splits to []
    assert parse_text(parser, sam([rec("a")]), b).n == 1                      # EOF of the shorter file
    assert parse_text(parser, "", b).n == 0


def test_name_mismatch_reported(parser):
    a = sam([rec("a"), rec("b"), rec("c")])
    b = sam([rec("a"), rec("x"), rec("c")])
    block = parse_text(parser, a, b)
    assert block.mismatch_at == 1 and block.n == 1


def test_skip_repeated_matches_oracle(parser):
    a = sam([rec("a"), rec("a"), rec("a"), rec("b"), rec("c"), rec("c")])
    b = sam([rec("a"), rec("b"), rec("b"), rec("c")])
    block = parse_text(parser, a, b, skip=True)
    want = list(ORACLE.read_pairs(io.StringIO(a), io.StringIO(b), True))
    assert block.n == len(want) == 3


def test_mixed_whitespace_and_normalisation(parser):
    messy = rec("a", "AS:i:5", sep=" ") + "  "
    t = sam([messy, rec("b", "AS:i:9")])
    block = parse_text(parser, t, t)
    assert block.cols[0].tolist() == [5, 9]
    out = bytes(parser.emit(False, 0, np.array([0, 1], dtype=np.uint32))).decode()
    assert out == "\t".join(messy.split()) + "\n" + rec("b", "AS:i:9") + "\n"


def test_exceptions_are_reported_not_guessed(parser):
    from xenomapper_amd import _host
    lines = [rec("a", "AS:f:12.5"), rec("b", "AS:i:7", "RG:Z:BASS"), rec("c", "AS:i:3000000000"),
             rec("d", "AS:i:1_0"), rec("e", "AS:i:", "XS:A:+"), rec("f", "AS:i:-2147483647"), rec("g", "xAS:i:44")]
    t = sam(lines)
    block = parse_text(parser, t, t)
    kinds = {(k, c): kind for k, c, kind in block.exc}
    assert kinds[(0, 0)] == _host.EX_NONINT and kinds[(1, 0)] == _host.EX_DUP and kinds[(2, 0)] == _host.EX_NONINT
    assert kinds[(3, 0)] == _host.EX_NONINT and kinds[(4, 0)] == _host.EX_NONINT and kinds[(4, 1)] == _host.EX_NONINT
    assert (5, 0) not in kinds and block.cols[0][5] == -2147483647
    assert (6, 0) not in kinds and block.cols[0][6] == 44           # substring match, like the reference
    # CIGAR mode
    lines = [rec("a", "NM:i:1"), "b 0 chr1 1 30 NM:i:2 x x x x x NM:i:2", rec("c", "NM:i:x"), rec("d", "NM:i:2", "NM:i:3")]
    t = sam(lines)
    block = parse_text(parser, t, t, score_mode=2)
    kinds = {(k, c): kind for k, c, kind in block.exc}
    assert kinds.get((2, 0)) == _host.EX_NONINT and (3, 0) not in kinds and block.csr[0][0][3] == 2
    assert (0, 0) not in kinds and block.csr[0][0][0] == 1


def test_non_ascii_is_refused(parser):
    from xenomapper_amd import _host
    t = sam([rec("a", "AS:i:5"), rec("b c", "AS:i:5")])
    with pytest.raises(_host.NonAsciiInput):
        parse_text(parser, t, t)


def test_windowed_walk_equals_whole_file(parser):
    """Drive the parser the way _run_files does, with tiny windows (halo kept for paired input)."""
    case = {c["name"]: c for c in G3}["cfg2_pe_liberal"]
    t1, t2 = H.case_texts(case)
    r1, p1 = body_of(t1)
    r2, p2 = body_of(t2)
    whole = parser.parse(r1, p1, len(r1) - p1, True, r2, p2, len(r2) - p2, True, 0, True, False, False, 1 << 22)
    whole_as1 = whole.cols[0].copy()
    whole_flags = np.unpackbits(whole.unit_bits.view(np.uint8), bitorder="little")[:whole.n].copy()
    pos = [p1, p2]
    got_as1, got_flags = [], []
    window = 4096
    for _ in range(100000):
        lens = [min(window, len(r1) - pos[0]), min(window, len(r2) - pos[1])]
        eofs = [pos[0] + lens[0] >= len(r1), pos[1] + lens[1] >= len(r2)]
        b = parser.parse(r1, pos[0], lens[0], eofs[0], r2, pos[1], lens[1], eofs[1], 0, True, False, True, 1 << 22)
        flags = np.unpackbits(b.unit_bits.view(np.uint8), bitorder="little")[:b.n]
        first = 1 if got_as1 else 0                       # record 0 of later windows is the halo
        got_as1 += b.cols[0][first:].tolist()
        got_flags += flags[first:].tolist()
        if first:
            assert flags[0] == 0
        if b.ended:
            break
        assert b.consumed[0] > 0 and b.consumed[1] > 0
        pos = [pos[0] + b.consumed[0], pos[1] + b.consumed[1]]
    assert got_as1 == whole_as1.tolist()
    assert got_flags == whole_flags.tolist()


def _repeated_name_texts(n_names, seed):
    """Two single-end SAM bodies over the same names, each name repeated 1-4 times independently per file."""
    rng = np.random.default_rng(seed)
    texts = []
    for f in range(2):
        reps = rng.integers(1, 5, size=n_names)
        lines = []
        for k in range(n_names):
            for j in range(int(reps[k])):
                lines.append("r%d\t0\tchr1\t%d\t30\t10M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\tAS:i:%d\tXS:i:%d\n"
                             % (k, 100 + k, -(k % 50) - 10 * j - f, -(k % 7) - 60))
        texts.append("".join(lines))
    return texts


@pytest.mark.parametrize("threads", [1, 7])
def test_skip_repeated_in_parallel_and_in_windows(threads):
    """skip_repeated_reads on > 4096 lines (every parallel slice boundary falls somewhere inside a run of equal
    names), whole file and in 3000-byte windows, against the oracle's lock-step reader."""
    from xenomapper_amd import _host
    t1, t2 = _repeated_name_texts(6000, 5)
    want = [int(ORACLE.tag_score(p[0], "AS")) for p in oracle_pairs(t1, t2, True)]
    assert len(want) == 6000
    prs = _host.Parser(threads)
    r1, r2 = (np.frombuffer(t.encode("ascii"), dtype=np.uint8) for t in (t1, t2))
    whole = prs.parse(r1, 0, len(r1), True, r2, 0, len(r2), True, 0, False, True, False, 1 << 22)
    assert whole.n == 6000 and whole.ended and whole.mismatch_at == -1
    assert whole.cols[0].tolist() == want
    pos, got = [0, 0], []
    for _ in range(100000):
        lens = [min(3000, len(r1) - pos[0]), min(3000, len(r2) - pos[1])]
        eofs = [pos[0] + lens[0] >= len(r1), pos[1] + lens[1] >= len(r2)]
        b = prs.parse(r1, pos[0], lens[0], eofs[0], r2, pos[1], lens[1], eofs[1], 0, False, True, False, 1 << 22)
        got += b.cols[0].tolist()
        if b.ended:
            break
        assert b.starved and (b.consumed[0] > 0 or b.consumed[1] > 0)
        pos = [pos[0] + b.consumed[0], pos[1] + b.consumed[1]]
    assert got == want
    # a mismatch and a blank line are found at the right pair, also behind a slice boundary
    bad = t2.replace("r5000\t", "rX\t")
    rb = np.frombuffer(bad.encode("ascii"), dtype=np.uint8)
    blk = prs.parse(r1, 0, len(r1), True, rb, 0, len(rb), True, 0, False, True, False, 1 << 22)
    assert blk.mismatch_at == 5000 and blk.n == 5000
    cut = t1.index("r4500\t")
    blank = t1[:cut] + "\n" + t1[cut:]
    rc = np.frombuffer(blank.encode("ascii"), dtype=np.uint8)
    blk = prs.parse(rc, 0, len(rc), True, r2, 0, len(r2), True, 0, False, True, False, 1 << 22)
    assert blk.ended and blk.n == 4500 and blk.mismatch_at == -1
    prs.close()


def test_worker_count_does_not_change_the_result():
    """Enough lines for every parallel phase to split (> 4096), parsed and written with 1, 3 and 7 workers."""
    import os
    from xenomapper_amd import _host, synth
    t1, t2, _ = synth.sam_text_pair(n_pairs=5000, seed=77, profile="bowtie2", paired=True, read_len=50)
    want = None
    for threads in (1, 3, 7):
        prs = _host.Parser(threads)
        block, _, _ = parse_all(prs, t1, t2, 2, True, False)
        flags = np.unpackbits(block.unit_bits.view(np.uint8), bitorder="little")[:block.n]
        idx = np.flatnonzero(flags).astype(np.uint32)
        fresh = bytes(prs.emit(True, 4, idx))
        assert bytes(prs.emit(True, 4, idx, reuse=True)) == fresh
        assert bytes(prs.emit(True, 4, idx[:10], reuse=True)) == bytes(prs.emit(True, 4, idx[:10]))
        got = ([c.tolist() for c in block.cols], [[a.tolist() for a in f] for f in block.csr], flags.tolist(),
               [o.tolist() for o in block.line_off], [l.tolist() for l in block.line_len], block.exc, fresh)
        prs.close()
        if want is None:
            want = got
            assert block.n == 10000 and len(fresh) > 0
        else:
            assert got == want
    n = _host.lib().xmh_default_threads()
    assert 1 <= n <= min(64, len(os.sched_getaffinity(0)))


def test_library_exports_header_symbols():
    import ctypes, os, re
    from xenomapper_amd import _host, build
    build.build_host()
    text = open(os.path.join(H.REPO, "include", "xenomapper_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(xmh_[a-z0-9_]+)\s*\(", text)))
    L = ctypes.CDLL(_host.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_host.EXPORTED) == names


# ---------------------------------------------------------------------------------------------- BAM input
def _decode_bam(path, chunk=1 << 20):
    from xenomapper_amd import _host
    data = np.fromfile(path, dtype=np.uint8)
    r = _host.BamReader(data, 3)
    hdr = r.header()
    buf = np.empty(chunk, dtype=np.uint8)
    parts = []
    while not r.eof:
        n = r.read_into(buf, 0)
        parts.append(bytes(buf[:n]).decode("ascii"))
        assert n > 0 or r.eof
    r.close()
    return hdr, "".join(parts)


@pytest.mark.parametrize("species", ["human", "mouse"])
@pytest.mark.parametrize("chunk", [1 << 20, 700])
def test_bam_decoder_reproduces_the_sam_fixture(species, chunk):
    """The reference's BAM fixtures hold the same alignments as its SAM fixtures; the native decoder must print
    them exactly as the SAM text (which is what `samtools view -h` gives the reference, ref :48-64)."""
    import os
    base = os.path.join(H.GOLDEN, "ref_data", "paired_end_testdata_%s" % species)
    hdr, body = _decode_bam(base + ".bam", chunk)
    with open(base + ".sam") as fh:
        sam = fh.read()
    assert hdr + body == sam + "\n"                      # the SAM fixture has no trailing newline


def _reblocked_bam(path, chunk, copies):
    """The alignments of a BAM file, `copies` times over, in BGZF blocks of `chunk` uncompressed bytes (records then
    straddle blocks, worker ranges and batches at arbitrary places)."""
    import gzip, struct, zlib
    raw = gzip.decompress(open(path, "rb").read())
    l_text, = struct.unpack_from("<i", raw, 4)
    at = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, at)
    at += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, at)
        at += 4 + l_name + 4
    payload = raw[:at] + raw[at:] * copies
    out = []
    for lo in range(0, len(payload), chunk):
        part = payload[lo:lo + chunk]
        comp = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = comp.compress(part) + comp.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" +
                   struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(part), len(part)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return np.frombuffer(b"".join(out), dtype=np.uint8)


@pytest.mark.parametrize("chunk,copies,threads,cap", [(37, 1, 5, 1 << 20), (300, 3, 16, 5000), (4096, 40, 7, 1 << 22),
                                                      (65280, 40, 16, 1 << 20), (65280, 3, 1, 1 << 16)])
def test_bam_decoder_with_records_across_blocks_and_workers(chunk, copies, threads, cap):
    import os
    from xenomapper_amd import _host
    base = os.path.join(H.GOLDEN, "ref_data", "paired_end_testdata_human")
    data = _reblocked_bam(base + ".bam", chunk, copies)
    r = _host.BamReader(data, threads)
    buf = np.empty(cap, dtype=np.uint8)
    parts = []
    while not r.eof:
        n = r.read_into(buf, 0)
        assert n > 0 or r.eof
        parts.append(bytes(buf[:n]))
    r.close()
    with open(base + ".sam") as fh:
        lines = [l for l in (fh.read() + "\n").splitlines(True) if not l.startswith("@")]
    assert b"".join(parts).decode("ascii") == "".join(lines) * copies


def test_bam_decoder_reports_a_file_that_ends_inside_a_record():
    import os
    from xenomapper_amd import _host
    data = _reblocked_bam(os.path.join(H.GOLDEN, "ref_data", "paired_end_testdata_human.bam"), 1000, 1)
    raw = data.tobytes()
    cut = raw.rfind(b"\x1f\x8b\x08\x04", 0, len(raw) - 28)       # drop the last data block and the end marker
    r = _host.BamReader(np.frombuffer(raw[:cut], dtype=np.uint8), 3)
    buf = np.empty(1 << 20, dtype=np.uint8)
    with pytest.raises(ValueError):
        while not r.eof:
            r.read_into(buf, 0)


@pytest.mark.parametrize("window,head", [(1500, 64), (1500, 4096), (50_000, 1 << 20), (1 << 22, 1 << 20), (333, 0),
                                         (9_999, 17), (70_000, 300)])
def test_bam_source_windows_reassemble_the_text(window, head, monkeypatch):
    """The file path's BAM source (three buffers, decoder thread running ahead) driven the way _run_files drives it:
    take a window, consume its whole lines, ask again; a window without a complete line is asked for again, larger."""
    import os
    from xenomapper_amd import xenomapper as xm
    monkeypatch.setattr(xm, "FILE_WINDOW_BYTES", window)
    monkeypatch.setattr(xm._BamSource, "HEAD", head)
    base = os.path.join(H.GOLDEN, "ref_data", "paired_end_testdata_mouse")
    src = xm._BamSource(base + ".bam", 3)
    out, want, prev = [], window, None
    for _ in range(1_000_000):
        buf, start, n, eof = src.window(want)
        if prev is not None:            # the writer gathers window k's lines while window k+1 is decoded and parsed
            assert bytes(prev[0][prev[1]:prev[1] + len(prev[2])]) == prev[2]
        view = bytes(buf[start:start + n])
        if eof:
            out.append(view)
            break
        cut = view.rfind(b"\n") + 1
        if cut == 0:
            want *= 2
            continue
        held = bytes(buf[start:start + cut])
        out.append(held)
        src.advance(cut)
        want = window
        prev = (buf, start, held)
    src.close()
    with open(base + ".sam") as fh:
        lines = [l for l in (fh.read() + "\n").splitlines(True) if not l.startswith("@")]
    assert b"".join(out).decode("ascii") == "".join(lines)


def test_bam_decoder_zlib_and_libdeflate_agree():
    """The decoder prefers libdeflate when the shared library is installed; XMH_NO_LIBDEFLATE=1 forces zlib.
    Both must print the fixture identically (the switch is read once per process, hence the child process)."""
    import os, subprocess, sys
    base = os.path.join(H.GOLDEN, "ref_data", "paired_end_testdata_mouse")
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from tests.test_host_parser import _decode_bam\n"
            "hdr, body = _decode_bam(%r, 1 << 16)\n"
            "print(hashlib.md5((hdr + body).encode()).hexdigest())" % (H.REPO, base + ".bam"))
    digests = []
    for force_zlib in (False, True):
        env = dict(os.environ)
        env.pop("XMH_NO_LIBDEFLATE", None)
        if force_zlib:
            env["XMH_NO_LIBDEFLATE"] = "1"
        digests.append(subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip())
    import hashlib
    with open(base + ".sam") as fh:
        want = hashlib.md5((fh.read() + "\n").encode()).hexdigest()
    assert digests == [want, want]


def test_bam_decoder_checks_the_block_crc():
    import os
    from xenomapper_amd import _host
    data = np.fromfile(os.path.join(H.GOLDEN, "ref_data", "paired_end_testdata_human.bam"), dtype=np.uint8).copy()
    bsize = int(data[16]) | (int(data[17]) << 8)                  # first block: flip one bit of its CRC-32 trailer
    data[bsize + 1 - 8] ^= 1
    with pytest.raises(ValueError):
        r = _host.BamReader(data)
        buf = np.empty(1 << 20, dtype=np.uint8)
        while not r.eof:
            r.read_into(buf, 0)


def test_bam_decoder_rejects_garbage():
    from xenomapper_amd import _host
    with pytest.raises(ValueError):
        _host.BamReader(np.frombuffer(b"this is not a BAM file at all, not even gzip....", dtype=np.uint8))
    import os
    data = np.fromfile(os.path.join(H.GOLDEN, "ref_data", "paired_end_testdata_human.bam"), dtype=np.uint8)
    with pytest.raises(ValueError):
        r = _host.BamReader(data[:20000].copy())
        buf = np.empty(1 << 20, dtype=np.uint8)
        while not r.eof:
            r.read_into(buf, 0)


def test_bam_optional_field_formats():
    """Hand-assembled BAM record covering every optional-field type."""
    import struct, zlib
    from xenomapper_amd import _host

    def bgzf(payload):
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        cdata = comp.compress(payload) + comp.flush()
        bsize = len(cdata) + 25
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + cdata +
                struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))
    text = b"@HD\tVN:1.0\n@SQ\tSN:chrA\tLN:1000\n@SQ\tSN:chrB\tLN:2000\n"
    head = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 2)
    for name, ln in ((b"chrA\x00", 1000), (b"chrB\x00", 2000)):
        head += struct.pack("<i", len(name)) + name + struct.pack("<i", ln)
    tags = (b"XAAq" + b"XBc" + struct.pack("<b", -5) + b"XCC" + struct.pack("<B", 200) + b"XDs" + struct.pack("<h", -300) +
            b"XES" + struct.pack("<H", 60000) + b"ASi" + struct.pack("<i", -70000) + b"XFI" + struct.pack("<I", 4000000000) +
            b"XGf" + struct.pack("<f", 1.5) + b"XHZhello world\x00" + b"XIH1AE3\x00" +
            b"XJBs" + struct.pack("<i", 3) + struct.pack("<hhh", 1, -2, 3) + b"XKBf" + struct.pack("<i", 2) + struct.pack("<ff", 0.25, 1e10))
    seq = bytes([0x12, 0x48, 0x10])                         # A C G T A (5 bases)
    qual = bytes([30, 31, 32, 33, 34])
    cigar = struct.pack("<II", (3 << 4) | 0, (2 << 4) | 4)  # 3M2S
    rname = b"read/1\x00"
    core = struct.pack("<iiBBHHHIiii", 0, 99, len(rname), 42, 4680, 2, 99, 5, 1, 199, -250)
    rec = core + rname + cigar + seq + qual + tags
    unm = struct.pack("<iiBBHHHIiii", -1, -1, 2, 0, 4680, 0, 4, 0, -1, -1, 0) + b"u\x00"
    payload = head + struct.pack("<i", len(rec)) + rec + struct.pack("<i", len(unm)) + unm
    image = bgzf(payload[:150]) + bgzf(payload[150:]) + bgzf(b"")
    r = _host.BamReader(np.frombuffer(image, dtype=np.uint8), 2)
    assert r.header() == text.decode()
    buf = np.empty(4096, dtype=np.uint8)
    n = r.read_into(buf, 0)
    lines = bytes(buf[:n]).decode().split("\n")
    assert lines[0] == ("read/1\t99\tchrA\t100\t42\t3M2S\tchrB\t200\t-250\tACGTA\t?@ABC\tXA:A:q\tXB:i:-5\tXC:i:200\tXD:i:-300"
                        "\tXE:i:60000\tAS:i:-70000\tXF:i:4000000000\tXG:f:1.5\tXH:Z:hello world\tXI:H:1AE3\tXJ:B:s,1,-2,3"
                        "\tXK:B:f,0.25,1e+10")
    assert lines[1] == "u\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*" and r.eof


def test_bam_long_cigar_convention():
    """A CIGAR of more than 65535 operations is stored as "<l_seq>S<ref_len>N" plus a CG:B:I field; htslib (sam.c,
    bam_tag2cigar) moves it back on reading, so `samtools view` prints the real operations and no CG field -- only for
    a mapped record whose first operation soft-clips the whole read and whose CG field is of type B,I."""
    import struct, zlib
    from xenomapper_amd import _host

    def record(ref_id, pos, cigar, cg_sub=b"I", l_seq=5, extra_before=b"", extra_after=b"NMC\x02"):
        real = struct.pack("<III", (3 << 4) | 0, (1 << 4) | 1, (1 << 4) | 0)                  # 3M1I1M
        cg = b"CGB" + cg_sub + struct.pack("<I", 3) + real
        name = b"r\x00"
        core = struct.pack("<iiBBHHHIiii", ref_id, pos, len(name), 7, 4680, len(cigar), 0, l_seq, -1, -1, 0)
        body = (core + name + b"".join(struct.pack("<I", (ln << 4) | op) for ln, op in cigar) + bytes([0x12, 0x48, 0x10]) +
                bytes([30] * 5) + extra_before + cg + extra_after)
        return struct.pack("<I", len(body)) + body
    head = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chrA\x00" + struct.pack("<i", 9)
    recs = [record(0, 9, [(5, 4), (4, 3)]),                                  # the convention: rewritten
            record(0, 9, [(5, 4), (4, 3)], extra_before=b"XXZhi\x00"),       # CG behind another field
            record(-1, 9, [(5, 4), (4, 3)]),                                 # unmapped: left alone
            record(0, 9, [(4, 4), (4, 3)]),                                  # the clip is not the whole read
            record(0, 9, [(5, 4), (4, 3)], cg_sub=b"i"),                     # CG is not B,I
            record(0, -1, [(5, 4), (4, 3)])]                                 # no position
    payload = head + b"".join(recs)
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(payload) + comp.flush()
    image = (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25) + cdata +
             struct.pack("<II", zlib.crc32(payload), len(payload)))
    r = _host.BamReader(np.frombuffer(image, dtype=np.uint8), 2)
    buf = np.empty(4096, dtype=np.uint8)
    lines = bytes(buf[:r.read_into(buf, 0)]).decode().split("\n")
    tail = "\t*\t0\t0\tACGTA\t?????"
    assert lines[0] == "r\t0\tchrA\t10\t7\t3M1I1M" + tail + "\tNM:i:2"
    assert lines[1] == "r\t0\tchrA\t10\t7\t3M1I1M" + tail + "\tXX:Z:hi\tNM:i:2"
    assert lines[2] == "r\t0\t*\t10\t7\t5S4N" + tail + "\tCG:B:I,48,17,16\tNM:i:2"
    assert lines[3] == "r\t0\tchrA\t10\t7\t4S4N" + tail + "\tCG:B:I,48,17,16\tNM:i:2"
    assert lines[4] == "r\t0\tchrA\t10\t7\t5S4N" + tail + "\tCG:B:i,48,17,16\tNM:i:2"
    assert lines[5] == "r\t0\tchrA\t0\t7\t5S4N" + tail + "\tCG:B:I,48,17,16\tNM:i:2"
    # and the plain-Python restatement says the same
    from oracle import bam_oracle
    assert bam_oracle.bam_to_sam(image)[1] == lines[:6]


def _one_record_bam(l_seq, name=b"longread\0"):
    """A minimal BAM image: no references, one unmapped alignment with an l_seq-base sequence, BGZF blocks of 65280 bytes."""
    import struct
    import zlib
    head = b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", 0)
    seq = bytes([0x12]) * ((l_seq + 1) // 2)                     # "AC" repeated
    qual = bytes([30]) * l_seq
    core = struct.pack("<iiBBHHHIiii", -1, -1, len(name), 0, 4680, 0, 4, l_seq, -1, -1, 0)
    body = core + name + seq + qual
    payload = head + struct.pack("<I", len(body)) + body
    out = []
    for at in range(0, len(payload), 65280):
        part = payload[at:at + 65280]
        comp = zlib.compressobj(1, zlib.DEFLATED, -15)
        blk = comp.compress(part) + comp.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(blk) + 25) + blk +
                   struct.pack("<II", zlib.crc32(part), len(part)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)


def test_bam_lines_grows_its_buffer_for_a_line_longer_than_8_mib(tmp_path):
    """One alignment whose SAM text (SEQ + QUAL of 5 M bases) is longer than bam_lines' first buffer: the generator
    must deliver it (it used to spin without progress).  Through a real file (memory-mapped) and through BytesIO."""
    import io
    from xenomapper_amd import xenomapper as xm
    l_seq = 5_000_000
    image = _one_record_bam(l_seq)
    path = tmp_path / "long.bam"
    path.write_bytes(image)
    for source in (open(path, "rb"), io.BytesIO(image)):
        with source:
            lines = list(xm.bam_lines(source))
        assert len(lines) == 1
        fields = lines[0].rstrip("\n").split("\t")
        assert fields[0] == "longread" and len(fields[9]) == l_seq and len(fields[10]) == l_seq
        assert fields[9][:4] == "ACAC" and fields[10][:2] == "??"
    with open(path, "rb") as fh:
        assert xm.get_bam_header(fh) == []


def test_bam_reader_maps_the_handle_it_was_given(tmp_path):
    """A binary handle without a usable name (open(fd, 'rb')), a handle that has been read from, and a file that was
    renamed and replaced after it was opened: the decoder reads the file the HANDLE refers to (it used to pass an int
    to np.memmap, and to reopen the path)."""
    import os
    import shutil
    from xenomapper_amd import xenomapper as xm
    base = os.path.join(H.GOLDEN, "ref_data", "paired_end_testdata_human")
    with open(base + ".sam") as fh:
        want = [line for line in fh.read().split("\n") if line and not line.startswith("@")]
    path = tmp_path / "a.bam"
    shutil.copy(base + ".bam", path)
    fd = os.open(path, os.O_RDONLY)
    with open(fd, "rb") as fh:                                   # fh.name is the int fd
        assert isinstance(fh.name, int)
        got = [line.rstrip("\n") for line in xm.bam_lines(fh)]
    assert got == want
    with open(path, "rb") as fh:
        fh.read(10)                                              # position moved: the whole file is decoded all the same
        os.rename(path, tmp_path / "moved.bam")
        path.write_bytes(b"not a bam file")                      # the old name now points at something else
        got = [line.rstrip("\n") for line in xm.bam_lines(fh)]
        assert fh.tell() == 10
    assert got == want


def _long_cigar_tool():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("make_bam_long_cigar_fixture",
                                                  os.path.join(H.REPO, "tools", "make_bam_long_cigar_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bam_long_cigar_fixture_from_the_specification():
    """tests/golden/long_cigar_cg.bam holds alignments with 70 000 and 65 537 CIGAR operations, written by
    tools/make_bam_long_cigar_fixture.py from the SAM/BAM specification's rule (SAMv1 section 4.2.2: real CIGAR in
    CG:B,I, CIGAR field = kSmN), next to a record with exactly 65 535 operations and an ordinary one; the .sam beside
    it is the text those records stand for.  The decoder must print exactly that text -- the one BAM rule the
    reference's own fixtures do not exercise."""
    import os
    from xenomapper_amd import xenomapper as xm
    base = os.path.join(H.GOLDEN, "long_cigar_cg")
    with open(base + ".sam") as fh:
        want = fh.read()
    hdr, body = _decode_bam(base + ".bam", 1 << 20)
    assert hdr + body == want
    with open(base + ".bam", "rb") as fh:
        lines = list(xm.bam_lines(fh))
    assert "".join(lines) == "".join(line + "\n" for line in want.split("\n") if line and not line.startswith("@"))
    assert [len(line.split("\t")[5]) for line in lines] == [140000, 131074, 3, 131070]      # CIGAR texts: 2 characters per one-digit operation
    assert all("CG:B" not in line for line in lines)
    # the committed fixture is what the committed script writes
    tool = _long_cigar_tool()
    recs = tool.logical_records()
    assert "".join(tool.sam_line(r) + "\n" for r in recs) == "".join(lines)
    assert [len(r[4]) for r in recs] == [70000, 65537, 1, 65535]


def test_writer_fills_the_mapped_output_file_directly(tmp_path, monkeypatch):
    """_emit_into_file: the bin's text gathered by the writer's threads straight into the extended, mapped output file
    must be byte for byte what the buffered path writes, land after what the sink already holds, and leave the sink
    usable; sinks that are not regular files at their end are declined."""
    import io
    import os
    from xenomapper_amd import _host, synth, xenomapper as xm
    t1, t2, _ = synth.sam_text_pair(n_pairs=3000, seed=5, profile="bowtie2", paired=True, read_len=150)

    def head_end(t):
        e = 0
        while t[e] == "@":
            e = t.index("\n", e) + 1
        return e
    a1, a2 = np.frombuffer(t1.encode(), dtype=np.uint8), np.frombuffer(t2.encode(), dtype=np.uint8)
    o1, o2 = head_end(t1), head_end(t2)
    parser = _host.Parser(4)
    blk = parser.parse(a1, o1, len(a1) - o1, True, a2, o2, len(a2) - o2, True, _host.SCORE_AS_XS, True, False, True, 1 << 22)
    idx = np.arange(1, blk.n, 2, dtype=np.uint32)
    want = bytes(parser.emit(True, 0, idx))
    assert len(want) > 100_000
    monkeypatch.setattr(xm, "MMAP_EMIT_MIN_BYTES", 1000)
    out = tmp_path / "bin.sam"
    with open(out, "wt") as sink:
        sink.write("@CO\twhat process_headers wrote\n")
        assert xm._emit_into_file(parser, True, 0, idx, sink) is True
        assert xm._emit_into_file(parser, True, 0, idx[:1], sink) is False         # too little: the caller writes it
        sink.write("after\n")
        assert xm._emit_into_file(parser, True, 0, idx, sink) is True             # page-unaligned start
    assert out.read_bytes() == b"@CO\twhat process_headers wrote\n" + want + b"after\n" + want
    with open(os.devnull, "wt") as sink:
        assert xm._emit_into_file(parser, True, 0, idx, sink) is False             # not a regular file
    assert xm._emit_into_file(parser, True, 0, idx, io.StringIO()) is False
    with open(out, "r+") as sink:                                                  # not at the end of the file
        assert xm._emit_into_file(parser, True, 0, idx, sink) is False
    monkeypatch.setenv("XENOMAPPER_MMAP_EMIT", "0")
    with open(tmp_path / "off.sam", "wt") as sink:
        assert xm._emit_into_file(parser, True, 0, idx, sink) is False
    parser.close()


@pytest.mark.parametrize("cap", [1 << 16, 1 << 20, 1 << 24])
def test_bam_line_descriptions_over_many_batches_and_pending_lines(cap, tmp_path):
    """xmh_bam_read_pre on an input of many BGZF batches read through small and large buffers (lines left pending between
    calls, batches in which some workers own no record): every call's descriptions must describe exactly the lines it
    wrote -- same text as xmh_bam_read, line lengths adding up, names and CIGAR operations in place."""
    import sys
    sys.path.insert(0, os.path.join(H.REPO, "tools"))
    import bench_bam
    from xenomapper_amd import _host
    path = str(tmp_path / "tiled.bam")
    bench_bam.tiled_bam(os.path.join(H.GOLDEN, "ref_data", "paired_end_testdata_human.bam"), path, 60)
    data = np.fromfile(path, dtype=np.uint8)
    plain = _host.BamReader(data, 8)
    big = np.empty(1 << 25, dtype=np.uint8)
    n = 0
    while not plain.eof:
        n += plain.read_into(big, n)
    plain.close()
    want = bytes(big[:n])
    r = _host.BamReader(data, 8)
    buf = np.empty(cap, dtype=np.uint8)
    got, lines = [], 0
    while not r.eof:
        w, pre, ops = r.read_into_pre(buf, 0)
        pre = _host.pre_view(pre)
        assert w > 0 or r.eof
        chunk = bytes(buf[:w])
        assert int(pre["line_len"].sum(dtype=np.int64)) + pre.shape[0] == w
        at = 0
        for q in pre[:50]:                                           # spot checks: the name and the CIGAR of the first lines
            line = chunk[at:at + int(q["line_len"])]
            f = line.split(b"\t")
            assert len(f[0]) == int(q["name_len"]) and not q["flags"]
            cig = [] if f[5] == b"*" else [(int(a), b"MIDNSHP=X".index(o)) for a, o in __import__("re").findall(rb"([0-9]+)([MIDNSHPX=])", f[5])]
            assert [(int(v) >> 4, int(v) & 15) for v in ops[int(q["ops_at"]):int(q["ops_at"]) + int(q["n_ops"])]] == cig
            tags = {t[:2]: t for t in f[11:]}
            for key, col in ((b"AS", "as"), (b"XS", "xs"), (b"NM", "nm")):
                assert int(q[col]) == (int(tags[key].split(b":")[-1]) if key in tags else -2**31)
            at += int(q["line_len"]) + 1
        got.append(chunk)
        lines += pre.shape[0]
    r.close()
    assert b"".join(got) == want and lines == want.count(b"\n")


def test_pread_and_copy_by_the_parser_threads_and_adopted_line_tables(tmp_path):
    """The host half of the GPU stripper's file path: windows read (or copied) into a buffer by the parser's threads, and
    xmh_emit on line tables that were made elsewhere (here: by another parser) writes what that parser writes itself."""
    import ctypes
    from xenomapper_amd import _host
    rng = np.random.default_rng(4)
    blob = rng.integers(0, 256, size=(5 << 20) + 12345, dtype=np.uint8)
    path = tmp_path / "blob.bin"
    path.write_bytes(blob.tobytes())
    p = _host.Parser(4)
    dst = np.zeros(blob.shape[0], dtype=np.uint8)
    fd = os.open(path, os.O_RDONLY)
    try:
        p.pread(fd, 1000, dst.ctypes.data, blob.shape[0] - 1000)
        assert np.array_equal(dst[:blob.shape[0] - 1000], blob[1000:])
        with pytest.raises(OSError):
            p.pread(fd, 1000, dst.ctypes.data, blob.shape[0])            # the file ends first
    finally:
        os.close(fd)
    dst[:] = 0
    p.copy(dst.ctypes.data + 7, blob, 11, 3 << 20)
    assert np.array_equal(dst[7:7 + (3 << 20)], blob[11:11 + (3 << 20)]) and not dst[:7].any()
    # adopted tables
    lines = ["r%d\t0\tc\t1\t9\t4M\t*\t0\t0\tACGT\tIIII\tAS:i:%d" % (i // 2, -i) for i in range(400)]
    lines[7] = lines[7].replace("\t", " ", 2)                               # one line that needs normalising
    text = ("\n".join(lines) + "\n").encode()
    raw = np.frombuffer(text, dtype=np.uint8)
    blk = p.parse(raw, 0, len(raw), True, raw, 0, len(raw), True, 0, True, False, False, 1 << 20)
    idx = np.arange(1, blk.n, 2, dtype=np.uint32)
    want = [bytes(p.emit(True, b, idx)) for b in (0, 1, 4)]
    q = _host.Parser(2)
    u32 = lambda a: np.ascontiguousarray(a, dtype=np.uint32)
    norm = [np.array([len("\t".join(l.split())) for l in lines], dtype=np.uint32)] * 2
    flags = [np.array([1 if "\t".join(l.split()) == l else 0 for l in lines], dtype=np.uint8)] * 2
    keep = [u32(blk.line_off[0]), u32(blk.line_len[0]), norm[0], flags[0], u32(blk.line_off[1]), u32(blk.line_len[1]), norm[1], flags[1]]
    q.adopt_lines(raw, 0, raw, 0, blk.n, [a.ctypes.data for a in keep])
    assert [bytes(q.emit(True, b, idx)) for b in (0, 1, 4)] == want
    p.close()
    q.close()
