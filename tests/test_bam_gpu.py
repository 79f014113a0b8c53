"""BAM input decoded ON THE GPU (include/xenomapper_bgzf.h: xm_bamdev_* = inflate + CRC + record chain + stripper + pair kernel)
against the text rules of the reference's plugins applied to the printed SAM text (the oracle's tag_score on the lines
`samtools view` would print, restated by oracle/bam_oracle.py / the host decoder), and the whole file path with either
BAM front end byte for byte.  The reference code this path replaces: getBamReadPairs / bam_lines, xenomapper.py:56-93, with
get_tag / get_tag_with_ZS_as_XS :176-206 on the lines."""
import ctypes
import gzip
import io
import os
import re
import sys

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from tests import helpers as H
from tests.test_host_fuzz import bam_file_pair

pytestmark = pytest.mark.gpu

DATA = os.path.join(H.REPO, "tests", "golden", "ref_data")
sys.path.insert(0, os.path.join(H.REPO, "tools"))
ABSENT = -2**31


@pytest.fixture(scope="module")
def ctx():
    from xenomapper_amd import xenomapper as xm
    return xm.default_context()


def host_text(image):
    """The SAM text of a BAM image through the host decoder (pinned to oracle/bam_oracle.py in test_host_fuzz)."""
    from xenomapper_amd import _host
    r = _host.BamReader(np.frombuffer(image, dtype=np.uint8), 2)
    buf = np.empty(max(1 << 16, 8 * len(image) + (1 << 16)), dtype=np.uint8)
    at = 0
    while not r.eof:
        got = r.read_into(buf, at)
        assert got or r.eof
        at += got
    r.close()
    return bytes(buf[:at])


def host_text_header_refs(image):
    """The reference names of a BAM image, in order (SAM specification 4.2)."""
    import struct
    raw = gzip.decompress(image)
    l_text, = struct.unpack_from("<i", raw, 4)
    at = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, at)
    at += 4
    names = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, at)
        names.append(raw[at + 4:at + 4 + l_name].split(b"\0", 1)[0])
        at += 4 + l_name + 4
    return names


def expected_from_text(line, tag, first_decides=False):
    """What the device must report for one record and tag, from the printed line: (value or ABSENT, flag 0 / 1 NONINT / 2 DUP).
    The plugins match a tag as a substring of any optional field (xenomapper.py:186) and raise on two matches (:189-190); the
    device vouches for a value only when the ONE matching field is the tag's own integer-typed field inside int32.
    first_decides: NM as get_cigarbased_AS_tag reads it (:247-250) -- the first matching field, however many follow."""
    fields = line.split(b"\t")[11:]
    hits = [x for x in fields if tag in x]
    if not hits:
        return ABSENT, 0
    if len(hits) > 1 and not first_decides:
        return ABSENT, 2
    parts = hits[0].split(b":")
    if hits[0][:2] == tag and len(parts) == 3 and parts[1] == b"i":
        v = int(parts[2])
        if -(2**31 - 1) <= v <= 2**31 - 1:
            return v, 0
    return ABSENT, 1


def run_whole_files(dev, images, score_mode, paired, halo=False, skip_repeated=False):
    from xenomapper_amd import _ffi, _host
    inputs, readers = [], []
    for f, image in enumerate(images):
        data = np.frombuffer(image, dtype=np.uint8)
        reader = _host.BamReader(data, 2)
        at = reader.records_start()
        blocks, crc, nxt, total = _ffi.bgzf_index(data)
        ends = blocks["out_off"] + blocks["isize"]
        j = int(np.searchsorted(ends, at, side="right"))
        blocks, crc = blocks[j:].copy(), crc[j:]
        skip = 0
        comp_len = 0
        if len(blocks):
            skip = at - int(blocks["out_off"][0])
            blocks["out_off"] -= blocks["out_off"][0]
            c0 = int(blocks["cdata_off"][0])
            comp_len = int(blocks["cdata_off"][-1]) + int(blocks["cdata_len"][-1]) - c0
            blocks["cdata_off"] -= np.uint64(c0)
        inputs.append((c0 if len(blocks) else 0, comp_len, {"comp_len": comp_len, "blocks": blocks, "crc": crc, "eof": True, "skip": skip}))
        readers.append(reader)
    raw_cap = max(1 << 16, max(int(x[2]["blocks"]["isize"].sum()) if len(x[2]["blocks"]) else 0 for x in inputs) + 64)
    dev.reserve(0, raw_cap + 4096, raw_cap, 4096, 1 << 16)
    for f, (c0, comp_len, _x) in enumerate(inputs):
        if comp_len:
            dev.staging(0, f)[:comp_len] = np.frombuffer(images[f], dtype=np.uint8)[c0:c0 + comp_len]
    blk = dev.run(0, [x[2] for x in inputs], score_mode, paired, halo, 1 << 16, skip_repeated=skip_repeated)
    return blk, readers


@settings(max_examples=int(os.environ.get("XM_FUZZ_EXAMPLES", "80")), deadline=None, suppress_health_check=list(HealthCheck))
@given(images=st.one_of(bam_file_pair(aligned=True), bam_file_pair(aligned="share")), score_mode=st.sampled_from([0, 1, 2]), paired=st.booleans(), skip=st.booleans())
def test_device_stripper_reports_what_the_text_rules_read(ctx, images, score_mode, paired, skip):
    """Record by record: value and flag of AS and XS (or ZS) exactly as the printed text dictates (a flag must be justified and
    a justified flag must be raised), names (unit mask, first mismatch), the walk's outcome, and `weird` exactly when a
    line holds a byte the text rules could split at.  score_mode 2 (--cigar_scores): NM in place of AS by the first-field rule,
    and the packed CIGAR columns the device makes of the records' CIGAR words against xm_cigar_pack of the printed CIGARs.
    skip: the skipping walk (getReadPairs :110-117) -- pair k is the first record of the k-th run of equal names of either file."""
    from xenomapper_amd import _ffi
    dev = _ffi.BamDev(ctx)
    try:
        blk, readers = run_whole_files(dev, images, score_mode, paired, skip_repeated=skip)
        assert not blk.bad_block and not blk.unaligned
        every = [host_text(im).split(b"\n")[:-1] for im in images]
        assert blk.n_rec == (len(every[0]), len(every[1]))
        odd = lambda l: any(c <= 0x20 and c != 9 or c >= 0x7F for c in l) or l.startswith(b"\t") or b"\t\t" in l   # noqa: E731
        if skip:
            # the records the walk can yield: the first of every run of equal names (a record the text rules might read
            # differently could cut the runs differently wherever it lies: `weird` looks at every record then)
            lines = [[l for i, l in enumerate(ls) if i == 0 or l.split(b"\t")[0] != ls[i - 1].split(b"\t")[0]] for ls in every]
            risky = min(len(every[0]), len(every[1])) > 0 and any(odd(l) for ls in every for l in ls)
        else:
            lines = every
            risky = any(odd(l) for ls in lines for l in ls[:min(len(lines[0]), len(lines[1]))])
        n_pairs = min(len(lines[0]), len(lines[1]))
        names = [[l.split(b"\t")[0] for l in ls] for ls in lines]
        mism = next((k for k in range(n_pairs) if names[0][k] != names[1][k]), -1)
        assert blk.mismatch_at == mism
        n = mism if mism >= 0 else n_pairs
        assert blk.n == n
        assert blk.weird == risky
        if mism < 0:
            assert blk.ended and not blk.starved                   # whole files in one window
        xtag = b"ZS" if score_mode == 1 else b"XS"
        cols = blk.cols
        flags = blk.line_flags
        for k in range(n):
            for f in (0, 1):
                va, fa = expected_from_text(lines[f][k], b"NM", True) if score_mode == 2 else expected_from_text(lines[f][k], b"AS")
                vx, fx = expected_from_text(lines[f][k], xtag)
                assert (int(cols[2 * f][k]), (int(flags[f][k]) >> 2) & 7) == (va, fa), (k, f, lines[f][k])
                assert (int(cols[2 * f + 1][k]), (int(flags[f][k]) >> 5) & 3) == (vx, fx), (k, f, lines[f][k])
        bits = np.unpackbits(blk.unit_bits.view(np.uint8), bitorder="little")[:n] if n else np.zeros(0, np.uint8)
        want_bits = [(1 if (k > 0 and names[0][k] == names[0][k - 1]) else 0) if paired else 1 for k in range(n)]
        assert bits.tolist() == want_bits
        assert blk.n_exceptions == sum(1 for k in range(n) if (int(flags[0][k]) | int(flags[1][k])) & 0x7C)
        if score_mode == 2 and n:
            letters = {c: k for k, c in enumerate("MIDNSHP=X")}
            for f in (0, 1):
                nm, cnt, tile, ops = dev.cigar_columns(0, f, n)
                assert (nm == cols[2 * f][:n]).all()
                assert int(cnt.max()) < 255                         # (records with 255 operations and more: the test below)
                ends = np.cumsum(cnt.astype(np.int64))
                assert int(ends[-1]) == ops.shape[0] == int(tile[-1])
                assert (tile[:-1] == (ends - cnt)[::256]).all()
                for k in range(n):
                    # operation codes above 8 print as letters the reference's pattern (:251) skips; the device keeps such words
                    # and the kernel scores I, D and S only
                    mine = [int(w) for w in ops[int(ends[k]) - int(cnt[k]):int(ends[k])] if (int(w) & 15) <= 8]
                    text = [(int(ln) << 4) | letters[op]
                            for ln, op in re.findall(r"([0-9]+)([MIDNSHPX=])", lines[f][k].split(b"\t")[5].decode("latin-1"))]
                    assert mine == text, (f, k, lines[f][k])
        # the printed text of the records is the host decoder's text
        for f in (0, 1):
            text = np.empty(max(1 << 16, 8 * len(images[f])), dtype=np.uint8)
            loff, llen = np.empty(n + 1, dtype=np.uint32), np.empty(n + 1, dtype=np.uint32)
            got = readers[f].print_records(blk.raw_addr[f], blk.rec_off_addr[f], n, text, loff, llen)
            assert bytes(text[:got]) == b"".join(l + b"\n" for l in lines[f][:n])
            assert [bytes(text[int(loff[k]):int(loff[k]) + int(llen[k])]) for k in range(n)] == lines[f][:n]
        # the device printer (xm_bamdev_fetch_text) on the same records: every line it prints is the host printer's; it declines
        # (status 1) exactly when a record a sink takes holds a floating-point field
        if n:
            for f in (0, 1):
                dev.set_refs(f, host_text_header_refs(images[f]))
            try:
                dev.classify(0, _ffi.MODE_PE_LIBERAL if paired else _ffi.MODE_SE, n, -2**31)
                status, dtext, dloff, dllen = dev.fetch_text(0, n, paired, 0b111111)
            except OverflowError:                                    # --cigar_scores: a synthesised score left int32
                status = -1
            floats = re.compile(rb"\t[^\t]{2}:(f|d|B:f)[:,]?")
            if status == -1:
                pass
            elif status == 0:
                dev.raw_wait(0)
                for f in (0, 1):
                    for k in range(n):
                        if int(dllen[f][k]) or int(dloff[f][k]):
                            a = int(dloff[f][k])
                            assert bytes(dtext[f][a:a + int(dllen[f][k]) + 1]) == lines[f][k] + b"\n", (f, k, lines[f][k])
            else:
                assert status == 1 and any(floats.search(l) for f in (0, 1) for l in lines[f][:n])
        for r in readers:
            r.close()
    finally:
        dev.close()


def run_path(paths, gpu, conservative=False, tag="AS", paired=True, skip=None):
    """tag: "AS" get_tag, "ZS" get_tag_with_ZS_as_XS, "NM" get_cigarbased_AS_tag; skip None: as the command line (not paired)."""
    """classify_sam_files(bam=True) with either BAM front end -> (six texts, counts) or the exception it raises."""
    from xenomapper_amd import xenomapper as xm
    os.environ["XENOMAPPER_GPU_BAM"] = "1" if gpu else "0"
    sinks = [io.StringIO() for _ in range(6)]
    try:
        counts = xm.classify_sam_files(paths[0], paths[1], *sinks, paired=paired, conservative=conservative, bam=True, skip_repeated_reads=skip,
                                       tag_func={"AS": xm.get_tag, "ZS": xm.get_tag_with_ZS_as_XS, "NM": xm.get_cigarbased_AS_tag}[tag])
    except Exception as exc:                                        # noqa: BLE001
        return type(exc).__name__, [s.getvalue() for s in sinks]
    finally:
        os.environ.pop("XENOMAPPER_GPU_BAM", None)
    return dict(counts), [s.getvalue() for s in sinks]


def oracle_path(images, conservative=False, tag="AS", paired=True, skip=None):
    """What the REFERENCE would do with two BAM images: `samtools view` text (oracle/bam_oracle.py, pinned to the reference's BAM
    fixtures) -> getBamReadPairs (xenomapper.py:66-93 = the oracle's read_pairs on lines cut at \n only) -> the main loop with
    the plugin.  Same shape as run_path's result: (counts or exception name, six texts in run_path's sink order)."""
    from oracle import bam_oracle
    texts = []
    for im in images:
        _header, lines = bam_oracle.bam_to_sam(im)
        texts.append("".join(l + "\n" for l in lines))
    scorer = {"AS": H.ORACLE.tag_score, "ZS": H.ORACLE.tag_score_zs, "NM": H.ORACLE.cigar_score}[tag]
    if skip is None:
        skip = not paired
    outs = [io.StringIO() for _ in range(6)]                        # the oracle's order: PS, SS, PM, SM, unresolved, unassigned
    order = [0, 1, 2, 3, 5, 4]                                      # run_path's: ..., unassigned, unresolved
    try:
        pairs = H.ORACLE.read_pairs(io.StringIO(texts[0], newline="\n"), io.StringIO(texts[1], newline="\n"), skip)
        if paired:
            res = H.ORACLE.run_paired_end(pairs, outs, H.NEG, scorer, conservative=conservative)
        else:
            res = H.ORACLE.run_single_end(pairs, outs, H.NEG, scorer)
    except Exception as exc:                                        # noqa: BLE001
        return type(exc).__name__, [outs[k].getvalue() for k in order]
    return dict(res.named_counts(paired)), [outs[k].getvalue() for k in order]


@pytest.mark.parametrize("bins", ["1", "0"])
@pytest.mark.parametrize("aligned", [True, False])
def test_file_path_many_windows_byte_identical_with_either_front_end(tmp_path, monkeypatch, aligned, bins):
    """The tiled fixtures through the whole file path: GPU BAM front end (small windows: many windows, carried tails, halo
    records) against the host decoder; record-aligned blocks (device record chain) and blocks that cut records (reported,
    chain followed on the host).  bins: the six outputs gathered on the device (xm_bamdev_fetch_bins, the default) or the
    records' text printed on the device and gathered by the host (xm_bamdev_fetch_text)."""
    import bench_bam
    from xenomapper_amd import xenomapper as xm
    monkeypatch.setenv("XENOMAPPER_GPU_BAM_BINS", bins)
    paths = []
    for tag in ("human", "mouse"):
        p = str(tmp_path / ("%s.bam" % tag))
        bench_bam.tiled_bam(os.path.join(DATA, "paired_end_testdata_%s.bam" % tag), p, 40, aligned=aligned)
        paths.append(p)
    want = run_path(paths, gpu=False)
    monkeypatch.setattr(xm, "BAM_GPU_WINDOW_BYTES", 1 << 20)
    for conservative, tag, paired in ((False, "AS", True), (True, "AS", True), (False, "NM", True), (False, "AS", False), (False, "NM", False)):
        # (not paired: the skipping walk, as the command line runs single-end input -- the mates' runs of two cross the windows)
        ref = run_path(paths, gpu=False, conservative=conservative, tag=tag, paired=paired) if (conservative or tag != "AS" or not paired) else want
        got = run_path(paths, gpu=True, conservative=conservative, tag=tag, paired=paired)
        assert got[0] == ref[0]
        assert got[1] == ref[1]
        prof = xm.LAST_FILE_PROFILE
        assert prof.get("strip_kernels_ms", 0) > 0 or not aligned   # the device path really ran
        if aligned:
            assert (prof.get("bam_windows_device_bins", 0) > 0) == (bins == "1"), dict(prof)
            assert prof.get("bam_windows_device_text", 0) == prof["bam_windows"] - prof.get("bam_windows_raw", 0), dict(prof)
    assert sum(want[0].values()) == 40 * 238


def test_uncompressed_bam_outgrows_the_staging_once_and_whole_windows_are_pinned_only_when_asked_for(tmp_path, monkeypatch):
    """The page-locked memory a slot holds: (1) the staging for compressed bytes is reserved for half the inflated size -- `samtools
    view -u` output (stored DEFLATE blocks: as large as the records) outgrows it, the engine reserves for the worst case and stages
    the window again, outputs equal to the host decoder's; (2) the host copies of whole inflated windows (xm_bamdev_fetch_raw) exist
    only from the first window that needs them: none after runs whose windows all went the device's way, both after a window with
    blocks that cut records."""
    import bench_bam
    from xenomapper_amd import xenomapper as xm
    xm.release_buffers()
    monkeypatch.setattr(xm, "BAM_GPU_WINDOW_BYTES", 2 << 20)         # (the staging: half of 2 + 1 MB and the carried tail, + 64 KB)
    monkeypatch.setattr(xm, "FILE_WINDOW_BYTES", 2 << 20)

    def files(level, aligned=True):
        paths = []
        for tag in ("human", "mouse"):
            p = str(tmp_path / ("%s_l%d_%d.bam" % (tag, level, aligned)))
            bench_bam.tiled_bam(os.path.join(DATA, "paired_end_testdata_%s.bam" % tag), p, 800, aligned=aligned, level=level)
            paths.append(p)
        return paths

    def raw_pinned():
        dev = xm.default_bamdev()
        return [bool(dev._L.xm_bamdev_raw(dev._h, slot, f)) for slot in (0, 1) for f in (0, 1)]

    compressed, stored = files(6), files(0)
    want = run_path(compressed, gpu=False)
    got = run_path(compressed, gpu=True)
    assert got == want
    assert xm.LAST_FILE_PROFILE.get("bam_staging_regrown", 0) == 0 and xm.LAST_FILE_PROFILE["bam_windows"] >= 6
    assert raw_pinned() == [False] * 4
    got = run_path(stored, gpu=True)
    assert got == want                                               # the same records, stored
    assert xm.LAST_FILE_PROFILE.get("bam_staging_regrown", 0) == 1 and xm.LAST_FILE_PROFILE.get("bam_windows_raw", 0) == 0
    assert raw_pinned() == [False] * 4
    assert run_path(stored, gpu=True) == want                        # (the buffers are large enough now)
    assert xm.LAST_FILE_PROFILE.get("bam_staging_regrown", 0) == 0
    cut = files(6, aligned=False)                                    # blocks that cut records: whole windows come back for the host to walk
    got = run_path(cut, gpu=True)
    assert xm.LAST_FILE_PROFILE.get("bam_windows_raw", 0) > 0 and any(raw_pinned())
    assert got == run_path(cut, gpu=False)
    xm.release_buffers()


@settings(max_examples=int(os.environ.get("XM_FUZZ_EXAMPLES", "60")), deadline=None, suppress_health_check=list(HealthCheck))
@given(images=st.one_of(bam_file_pair(aligned=True), bam_file_pair(aligned="share"), bam_file_pair()), tag=st.sampled_from(["AS", "ZS", "NM"]),
       conservative=st.booleans(), walk=st.sampled_from(["paired", "paired", "single", "paired skipping"]))
def test_adversarial_bam_pairs_through_the_file_path(tmp_path_factory, images, tag, conservative, walk):
    """Typed tags at the int32 edges, floats and characters under the tags' names, strings that merely contain the letters,
    names with blanks, second files that end early: outputs, counts and exception types of the GPU front end equal the host
    decoder's (which is pinned to the reference through the text rules) AND the oracle's own reading of the same two images."""
    d = tmp_path_factory.mktemp("bam")
    paths = []
    for f, im in enumerate(images):
        p = str(d / ("f%d.bam" % f))
        with open(p, "wb") as fh:
            fh.write(im)
        paths.append(p)
    kw = {"paired": walk != "single", "skip": True if walk == "paired skipping" else None}
    want = run_path(paths, gpu=False, conservative=conservative, tag=tag, **kw)
    got = run_path(paths, gpu=True, conservative=conservative, tag=tag, **kw)
    assert got == want
    # and with the oracle directly (VERDICT r5 #6a): the GPU front end against the reference's own rules, not only against the
    # product's host decoder.  The one documented deviation (--cigar_scores values beyond the packed columns: OverflowError,
    # tests/test_file_fuzz_gpu.py) is the only difference allowed.
    ref = oracle_path(images, conservative=conservative, tag=tag, **kw)
    if got[0] == "OverflowError" and tag == "NM" and ref[0] != "OverflowError":
        return
    assert got[0] == ref[0]
    assert got[1] == ref[1]


@pytest.mark.parametrize("bins", ["0", "1"])
def test_more_text_than_the_buffers_hold_goes_to_the_host_printer(tmp_path, monkeypatch, bins):
    """The same file on both sides: every pair is unresolved, every record of both files is wanted, and their SAM text (1.6 x the
    records) does not fit the slot's packed-record buffers in windows of 8 MB -- xm_bamdev_fetch_text declines (status 2) and the
    window is printed by the host from the packed records; a smaller mask of sinks fits and is printed on the device.  Outputs equal
    the host decoder's either way."""
    import bench_bam
    from xenomapper_amd import xenomapper as xm
    # bins "0": the text printed per file (xm_bamdev_fetch_text), what the paragraph above describes; "1": the outputs gathered on
    # the device (xm_bamdev_fetch_bins), whose one stream may use both files' buffers -- here it fits, so nothing is declined
    monkeypatch.setenv("XENOMAPPER_GPU_BAM_BINS", bins)
    p = str(tmp_path / "same.bam")
    bench_bam.tiled_bam(os.path.join(DATA, "paired_end_testdata_human.bam"), p, 100)
    monkeypatch.setattr(xm, "BAM_GPU_WINDOW_BYTES", 8 << 20)
    monkeypatch.setattr(xm, "FILE_WINDOW_BYTES", 8 << 20)
    want = run_path([p, p], gpu=False)
    if xm._bamdev is not None:                                       # a front end of its own: the process-wide one keeps the
        xm._bamdev.close()                                           # largest buffers any earlier run asked for
    monkeypatch.setattr(xm, "_bamdev", None)
    got = run_path([p, p], gpu=True)
    xm._bamdev.close()
    assert isinstance(want[0], dict) and got == want
    assert len(got[1][4]) > 0 and not any(got[1][b] for b in (0, 1, 2, 3))     # all unresolved (or unassigned)
    if bins == "0":
        assert xm.LAST_FILE_PROFILE.get("bam_print", 0) > 0                    # the host printed
    else:
        assert xm.LAST_FILE_PROFILE.get("bam_windows_device_bins", 0) + xm.LAST_FILE_PROFILE.get("bam_print", 0) > 0


def test_runs_of_equal_names_longer_than_a_window(tmp_path, monkeypatch):
    """The skipping walk over files whose runs of equal names differ between the files and cross the windows -- one run longer
    than several windows (the window has to grow), runs that end exactly with a BGZF block, a file whose last run reaches its
    end: GPU front end against the host decoder, single-end (the command line's walk) and paired with skipping."""
    import struct
    from tests.test_host_fuzz import _bam_image_of
    from xenomapper_amd import xenomapper as xm
    rng = np.random.default_rng(3)
    names = 400
    lens = [rng.integers(1, 6, size=names), rng.integers(1, 6, size=names)]
    lens[0][37], lens[1][37] = 9000, 3                               # ~ 450 KB of one name in file 1
    lens[1][200] = 2500
    lens[0][names - 1] = 40                                          # the last run reaches the end of the file
    images = []
    for f in (0, 1):
        recs = []
        for k in range(names):
            for j in range(int(lens[f][k])):
                name = b"read%d\0" % k
                tags = b"ASc" + struct.pack("<b", -int(rng.integers(0, 40))) + (b"XSc" + struct.pack("<b", -int(rng.integers(0, 60))) if (k + j) % 3 else b"")
                core = struct.pack("<iiBBHHHIiii", 0, 100 + k, len(name), 30, 4680, 1, 0, 4, -1, -1, 0)
                body = core + name + struct.pack("<I", (4 << 4) | 0) + b"\x12\x48" + b"\x1e\x1e\x1e\x1e" + tags
                recs.append(struct.pack("<I", len(body)) + body)
        images.append(_bam_image_of(recs, aligned=True))
    paths = []
    for f, im in enumerate(images):
        paths.append(str(tmp_path / ("r%d.bam" % f)))
        with open(paths[-1], "wb") as fh:
            fh.write(im)
    monkeypatch.setattr(xm, "BAM_GPU_WINDOW_BYTES", 1 << 16)
    monkeypatch.setattr(xm, "FILE_WINDOW_BYTES", 1 << 16)
    for paired, skip in ((False, None), (True, True)):
        want = run_path(paths, gpu=False, paired=paired, skip=skip)
        got = run_path(paths, gpu=True, paired=paired, skip=skip)
        assert isinstance(want[0], dict) and sum(want[0].values()) == (names if not paired else 0)
        assert got == want
        assert xm.LAST_FILE_PROFILE.get("strip_kernels_ms", 0) > 0   # the device path really ran


def _cigar_bam(n, seed, n_ops_of):
    """n records, mates r0 r0 r1 r1 .., with NM and XS tags; record k has n_ops_of(k) CIGAR operations (M / I / D / S / N mixed)."""
    import struct
    from tests.test_host_fuzz import _bam_image_of
    rng = np.random.default_rng(seed)
    recs = []
    for k in range(n):
        name = b"r%d\0" % (k // 2)
        c = n_ops_of(k)
        words = (rng.integers(1, 40, size=c).astype(np.uint32) << 4) | rng.choice(np.array([0, 0, 1, 2, 4, 3], dtype=np.uint32), size=c)
        tags = b""
        if k % 7 != 3:
            tags += b"NMC" + bytes([int(rng.integers(0, 9))])
        if k % 5 == 1:
            tags += b"XSc" + struct.pack("<b", int(rng.integers(-60, 0)))
        core = struct.pack("<iiBBHHHIiii", 0, 100 + k, len(name), 30, 4680, c, 0, 0, -1, -1, 0)
        body = core + name + words.astype("<u4").tobytes() + tags
        recs.append(struct.pack("<I", len(body)) + body)
    return _bam_image_of(recs, aligned=True)


def test_cigar_columns_made_on_the_device_with_255_operations_and_more(ctx, tmp_path):
    """--cigar_scores on the GPU BAM front end: records with 0, 254, 255, 256 and 3000 CIGAR operations (count byte 255 + trailer
    word), more than one tile of 256 records: the device's packed CIGAR columns equal xm_cigar_pack of the files' own CIGAR words,
    and the whole file path equals the host decoder's (which is pinned to the reference's text rule, xenomapper.py:228-256)."""
    from xenomapper_amd import _ffi
    import struct
    sizes = {5: 254, 6: 255, 7: 256, 300: 3000, 511: 255, 512: 0, 513: 700}
    images = [_cigar_bam(700, 11, lambda k: sizes.get(k, k % 6)), _cigar_bam(700, 12, lambda k: sizes.get(k + 1, (k + 2) % 5))]
    dev = _ffi.BamDev(ctx)
    try:
        blk, readers = run_whole_files(dev, images, 2, False)
        assert not blk.bad_block and not blk.unaligned and not blk.weird and blk.n == 700 and blk.n_exceptions == 0
        for f in (0, 1):
            raw = __import__("gzip").decompress(images[f])
            at = readers[f].records_start()
            per, nms = [], []
            while at < len(raw):
                size, = struct.unpack_from("<I", raw, at)
                l_name, n_cig = raw[at + 12], struct.unpack_from("<H", raw, at + 16)[0]
                p = at + 36 + l_name
                per.append(np.frombuffer(raw, dtype="<u4", count=n_cig, offset=p))
                tags = raw[p + 4 * n_cig:at + 4 + size]
                nms.append(tags[3] if tags[:2] == b"NM" else ABSENT)
                at += 4 + size
            off = np.cumsum([0] + [x.shape[0] for x in per]).astype(np.uint32)
            w_cnt, w_tile, w_ops = _ffi.cigar_pack(off, np.concatenate(per).astype(np.uint32))
            nm, cnt, tile, ops = dev.cigar_columns(0, f, blk.n)
            assert nm.tolist() == nms
            assert (cnt == w_cnt).all() and (tile == w_tile).all() and (ops == w_ops).all()
        for r in readers:
            r.close()
    finally:
        dev.close()
    paths = []
    for f, im in enumerate(images):
        paths.append(str(tmp_path / ("c%d.bam" % f)))
        with open(paths[-1], "wb") as fh:
            fh.write(im)
    for conservative in (False, True):
        want = run_path(paths, gpu=False, conservative=conservative, tag="NM")
        got = run_path(paths, gpu=True, conservative=conservative, tag="NM")
        assert isinstance(want[0], dict) and sum(want[0].values()) == 350
        assert got == want
    from xenomapper_amd import xenomapper as xm
    assert xm.LAST_FILE_PROFILE.get("strip_kernels_ms", 0) > 0      # the device path really ran


def test_bytes_sent_ahead_are_the_bytes_the_run_would_have_sent(ctx, tmp_path):
    """xm_bamdev_upload: part of a window's compressed bytes goes to the device before xm_bamdev_run (what the file path's read-ahead
    thread does while the GPU has the other slot); the run told so (`uploaded`) must give what a run that sends everything itself
    gives -- also when only a prefix went ahead, when nothing did, and after the staging buffers have grown in between (what was sent
    is forgotten then: claiming it is refused)."""
    import bench_bam
    from xenomapper_amd import _ffi
    paths = []
    for tag in ("human", "mouse"):
        p = str(tmp_path / ("%s.bam" % tag))
        bench_bam.tiled_bam(os.path.join(DATA, "paired_end_testdata_%s.bam" % tag), p, 12)
        paths.append(p)
    images = [open(p, "rb").read() for p in paths]
    dev = _ffi.BamDev(ctx)
    try:
        base, readers = run_whole_files(dev, images, 0, True)
        want = (base.n, base.n_rec, base.mismatch_at, [c[:base.n].copy() for c in base.cols], base.unit_bits.copy())
        for r in readers:
            r.close()
        # the same windows again, their compressed bytes (or a prefix) sent ahead
        def again(fractions, grow=False):
            inputs = []
            for f, image in enumerate(images):
                data = np.frombuffer(image, dtype=np.uint8)
                from xenomapper_amd import _host
                reader = _host.BamReader(data, 1, header_only=True)
                at = reader.records_start()
                reader.close()
                blocks, crc, _nxt, _total = _ffi.bgzf_index(data)
                ends = blocks["out_off"] + blocks["isize"]
                j = int(np.searchsorted(ends, at, side="right"))
                blocks, crc = blocks[j:].copy(), crc[j:]
                skip = at - int(blocks["out_off"][0])
                blocks["out_off"] -= blocks["out_off"][0]
                c0 = int(blocks["cdata_off"][0])
                comp_len = int(blocks["cdata_off"][-1]) + int(blocks["cdata_len"][-1]) - c0
                blocks["cdata_off"] -= np.uint64(c0)
                dev.staging(0, f)[:comp_len] = data[c0:c0 + comp_len]
                sent = int(comp_len * fractions[f])
                dev.upload(0, f, sent)
                inputs.append({"comp_len": comp_len, "blocks": blocks, "crc": crc, "eof": True, "skip": skip, "uploaded": sent})
            if grow:
                cap = dev.capacity(0)
                dev.reserve(0, cap[0] * 2, cap[1], cap[2], cap[3])
            return dev.run(0, inputs, 0, True, False, 1 << 16)
        for fractions in ((1.0, 1.0), (0.5, 0.25), (0.0, 1.0)):
            blk = again(fractions)
            assert not blk.bad_block and not blk.unaligned
            assert (blk.n, blk.n_rec, blk.mismatch_at) == want[:3]
            assert all(np.array_equal(blk.cols[c][:blk.n], want[3][c]) for c in range(4))
            assert np.array_equal(blk.unit_bits, want[4])
        with pytest.raises(ValueError):
            again((1.0, 1.0), grow=True)                                   # the buffers grew: nothing is on the device any more
    finally:
        dev.close()


@pytest.mark.parametrize("paired,mode", [(True, 1), (True, 2), (False, 0)])
def test_only_the_records_a_sink_takes_come_back(ctx, tmp_path, paired, mode):
    """xm_bamdev_fetch_wanted: with the bins on the device, the records that go back to the host are exactly those of units whose
    bin has a sink -- file 1's records for bins 0, 2, 5, file 2's for bins 1, 3, both for bin 4 (xenomapper.py:423-448; a paired
    unit = records i - 1 and i) -- packed next to each other, byte for byte the records of the whole window, in input order; every
    other record's table entry says so.  All sink masks with one or two sinks missing, and none / all."""
    import struct
    import bench_bam
    from xenomapper_amd import _ffi
    paths = []
    for tag in ("human", "mouse"):
        p = str(tmp_path / ("%s.bam" % tag))
        bench_bam.tiled_bam(os.path.join(DATA, "paired_end_testdata_%s.bam" % tag), p, 6)
        paths.append(p)
    images = [open(p, "rb").read() for p in paths]
    dev = _ffi.BamDev(ctx)
    try:
        blk, readers = run_whole_files(dev, images, 0, paired)           # the whole windows on the host as well (wait_raw)
        n = blk.n
        assert n > 1000 and not blk.n_exceptions
        if mode == 1:                                                     # the device printer needs the reference names
            dev.classify(0, mode, n, -2**31)
            with pytest.raises(ValueError):
                dev.fetch_text(0, n, paired, 0b111111)
        # the host printer's text of every record, and the files' reference names for the device printer
        host_lines = []
        for f in (0, 1):
            text = np.empty(8 * len(images[f]) + (1 << 16), dtype=np.uint8)
            loff, llen = np.empty(n + 1, dtype=np.uint32), np.empty(n + 1, dtype=np.uint32)
            readers[f].print_records(blk.raw_addr[f], blk.rec_off_addr[f], n, text, loff, llen)
            host_lines.append([bytes(text[int(loff[k]):int(loff[k]) + int(llen[k])]) for k in range(n)])
            head = host_text_header_refs(images[f])
            dev.set_refs(f, head)
        for r in readers:
            r.close()
        raws = [np.ctypeslib.as_array((ctypes.c_uint8 * blk.raw_len[f]).from_address(blk.raw_addr[f])).copy() for f in (0, 1)]
        rec = [np.ctypeslib.as_array((ctypes.c_uint32 * blk.n_rec[f]).from_address(blk.rec_off_addr[f])).copy() for f in (0, 1)]
        code, idx, off, _counts = dev.classify(0, mode, n, -2**31)
        idx, off = idx.copy(), off.copy()
        masks = [0, 0b111111, 0b000001, 0b010010, 0b101101, 0b011111, 0b110111]
        for sink_mask in masks:
            want = [np.zeros(n, dtype=bool), np.zeros(n, dtype=bool)]
            for b in range(6):
                if not (sink_mask >> b) & 1:
                    continue
                seg = idx[int(off[b]):int(off[b + 1])].astype(np.int64)
                for f in ((0,) if b in (0, 2, 5) else (1,) if b in (1, 3) else (0, 1)):
                    want[f][seg] = True
                    if paired:
                        want[f][seg - 1] = True
            addrs, places, nbytes = dev.fetch_wanted(0, n, paired, sink_mask)
            dev.raw_wait(0)
            for f in (0, 1):
                place = places[f].copy()
                assert np.array_equal(place != 0xFFFFFFFF, want[f]), (sink_mask, f)
                packed = np.ctypeslib.as_array((ctypes.c_uint8 * max(nbytes[f], 1)).from_address(addrs[f]))[:nbytes[f]]
                at = 0
                for i in np.flatnonzero(want[f]).tolist():
                    o = int(rec[f][i])
                    size = 4 + struct.unpack_from("<I", raws[f], o)[0]
                    assert int(place[i]) == at, (sink_mask, f, i)
                    assert bytes(packed[at:at + size]) == bytes(raws[f][o:o + size])
                    at += size
                assert at == nbytes[f]
            # (c) xm_bamdev_fetch_text: the same records as SAM text printed on the device = the host printer's lines, next to
            # each other in input order, with the line table the writer gathers from
            status, text, loff, llen = dev.fetch_text(0, n, paired, sink_mask)
            assert status == 0
            dev.raw_wait(0)
            for f in (0, 1):
                at = 0
                lo, ll = loff[f].copy(), llen[f].copy()
                for i in range(n):
                    if not want[f][i]:
                        assert lo[i] == 0 and ll[i] == 0
                        continue
                    assert int(lo[i]) == at and bytes(text[f][at:at + int(ll[i]) + 1]) == host_lines[f][i] + b"\n", (sink_mask, f, i)
                    at += int(ll[i]) + 1
    finally:
        dev.close()


def _tagged_records(n_pairs, float_at=(), string_at=(), seed=0):
    """Two BAM record lists (the same reads, interleaved mates, integer AS / XS) with, at the given PAIR numbers, a value the
    device does not vouch for in file 1: AS typed `f` (float_at), or a Z string that holds the letters 'XS' (string_at: a second
    match of the plugin's substring rule -> the reference raises ValueError there)."""
    import struct
    rng = np.random.default_rng(seed)
    recs = [[], []]
    float_at, string_at = set(float_at), set(string_at)
    for k in range(n_pairs):
        for mate in (0, 1):
            name = ("read%07d" % k).encode() + b"\0"
            for f in (0, 1):
                a, x = int(rng.integers(-40, 1)), int(rng.integers(-60, 1))
                if f == 0 and mate == 0 and k in float_at:
                    tags = b"ASf" + struct.pack("<f", a + 0.5) + b"XSi" + struct.pack("<i", x)
                else:
                    tags = b"ASi" + struct.pack("<i", a) + (b"XSi" + struct.pack("<i", x) if rng.random() < 0.6 else b"")
                if f == 0 and mate == 1 and k in string_at:
                    tags += b"coZ" + b"seeXS1" + b"\0"
                l_seq = 20
                seq = bytes(rng.integers(0, 256, (l_seq + 1) // 2, dtype=np.uint8))
                qual = bytes(rng.integers(0, 60, l_seq, dtype=np.uint8))
                core = struct.pack("<iiBBHHHIiii", 0, 100 + k, len(name), 30, 4680, 1, 64 if mate == 0 else 128, l_seq, 0, 300 + k, 0)
                body = core + name + struct.pack("<I", (l_seq << 4) | 0) + seq + qual + tags
                recs[f].append(struct.pack("<I", len(body)) + body)
    return recs


def test_windows_with_exception_records_between_windows_without(tmp_path, monkeypatch):
    """Two threads, one context: the helper thread runs the fused pass of window k + 1 on the process-wide context
    while the main thread settles window k -- which classifies AGAIN, through the host-buffer call on the same context, when the
    window held a value the device does not vouch for (here: AS typed `f` in every other stretch of the files).  With windows of
    256 KB the 24 000 pairs below are ~20 windows per file, alternating between the two kinds; every output and the counts must
    equal the oracle's reading of the same images, and the host decoder's."""
    from tests.test_host_fuzz import _bam_image_of
    from xenomapper_amd import xenomapper as xm
    n_pairs = 24_000
    stretch = 1_500                                                   # pairs per stretch: about one window
    float_at = [k for k in range(n_pairs) if (k // stretch) % 2 == 0 and k % 97 == 5]
    recs = _tagged_records(n_pairs, float_at=float_at, seed=11)
    images = [_bam_image_of(r, aligned=True) for r in recs]
    paths = []
    for f, im in enumerate(images):
        p = str(tmp_path / ("w%d.bam" % f))
        with open(p, "wb") as fh:
            fh.write(im)
        paths.append(p)
    monkeypatch.setattr(xm, "BAM_GPU_WINDOW_BYTES", 256 << 10)
    monkeypatch.setattr(xm, "FILE_WINDOW_BYTES", 256 << 10)
    for conservative in (False, True):
        ref = oracle_path(images, conservative=conservative)
        assert isinstance(ref[0], dict) and sum(ref[0].values()) == n_pairs
        got = run_path(paths, gpu=True, conservative=conservative)
        prof = dict(xm.LAST_FILE_PROFILE)
        assert got[0] == ref[0]
        assert got[1] == ref[1]
        assert prof.get("strip_kernels_ms", 0) > 0
        assert run_path(paths, gpu=False, conservative=conservative) == got


def test_a_float_tag_takes_one_window_to_the_host_printer_not_the_run(tmp_path, monkeypatch):
    """Golden row for floating-point optional fields (VERDICT r5 #6c, #8; `samtools view` prints them with %g, which the reference
    then reads as text, xenomapper.py:56-64): a file pair with ONE `de:f` field in one record.  Outputs equal the oracle's; the
    profile says which printer ran -- the device's for every window (it prints %g itself since round 6), the host's for none."""
    import struct
    from tests.test_host_fuzz import _bam_image_of
    from xenomapper_amd import xenomapper as xm
    recs = _tagged_records(6_000, seed=5)
    for f in (0, 1):                                                  # pair 1 000, mate 0: one more field, typed f
        k = 2 * 1_000
        body = recs[f][k][4:] + b"def" + struct.pack("<f", 0.0123)
        recs[f][k] = struct.pack("<I", len(body)) + body
    images = [_bam_image_of(r, aligned=True) for r in recs]
    paths = []
    for f, im in enumerate(images):
        p = str(tmp_path / ("d%d.bam" % f))
        with open(p, "wb") as fh:
            fh.write(im)
        paths.append(p)
    monkeypatch.setattr(xm, "BAM_GPU_WINDOW_BYTES", 256 << 10)
    monkeypatch.setattr(xm, "FILE_WINDOW_BYTES", 256 << 10)
    ref = oracle_path(images)
    assert isinstance(ref[0], dict) and any("de:f:0.0123" in t for t in ref[1])
    got = run_path(paths, gpu=True)
    prof = dict(xm.LAST_FILE_PROFILE)
    assert got[0] == ref[0]
    assert got[1] == ref[1]
    assert prof.get("bam_windows", 0) >= 4
    assert prof.get("bam_windows_device_text", 0) == prof["bam_windows"] and prof.get("bam_windows_host_text", 0) == 0


@pytest.mark.parametrize("damage", ["trailer", "payload"])
def test_a_block_that_fails_its_crc_is_an_error_with_either_front_end(tmp_path, monkeypatch, damage):
    """BGZF carries a CRC-32 per block (SAM specification 4.1; htslib checks it, so the reference -- which reads BAM through
    `samtools view` -- never sees bytes that fail it).  A file whose ONE damaged block still inflates (stored DEFLATE blocks: a
    flipped payload byte is still a valid stream; or the CRC word of the trailer itself is flipped) must end the run with
    ValueError on the GPU front end (crc32_kernel against the member trailers) as with the host decoder, and nothing of the
    damaged window may have been written; the windows in front of it are (several windows: 256 KB each)."""
    import bench_bam
    from xenomapper_amd import xenomapper as xm
    raw = gzip.decompress(open(os.path.join(DATA, "paired_end_testdata_human.bam"), "rb").read())
    l_text, = __import__("struct").unpack_from("<i", raw, 4)
    at = 8 + l_text
    n_ref, = __import__("struct").unpack_from("<i", raw, at)
    at += 4
    for _ in range(n_ref):
        l_name, = __import__("struct").unpack_from("<i", raw, at)
        at += 4 + l_name + 4
    head = bench_bam.bgzf_blocks(raw[:at])
    records = bench_bam.record_aligned_blocks(raw[at:], level=0)                  # stored blocks: every payload byte is literal
    copies = 12
    good = head + records * copies + bench_bam.BGZF_EOF
    # the damaged copy is the ninth: behind several whole windows
    where = len(head) + 8 * len(records)
    first_len = __import__("struct").unpack_from("<H", good, where + 16)[0] + 1   # BSIZE + 1 of the first member of that copy
    bad = bytearray(good)
    if damage == "trailer":
        bad[where + first_len - 8] ^= 0x01                                        # the CRC-32 word of the member trailer
    else:
        bad[where + 18 + 5 + 200] ^= 0x20                                         # a byte of the stored payload (18 header + 5 stored-block bytes in front)
    paths = []
    for name, image in (("ok.bam", good), ("bad.bam", bytes(bad))):
        p = str(tmp_path / name)
        with open(p, "wb") as fh:
            fh.write(image)
        paths.append(p)
    monkeypatch.setattr(xm, "BAM_GPU_WINDOW_BYTES", 256 << 10)
    monkeypatch.setattr(xm, "FILE_WINDOW_BYTES", 256 << 10)
    clean = run_path([paths[0], paths[0]], gpu=True)
    assert isinstance(clean[0], dict) and sum(clean[0].values()) == copies * 238
    for gpu in (True, False):
        for pair in ([paths[1], paths[0]], [paths[0], paths[1]]):
            got = run_path(pair, gpu=gpu)
            assert got[0] == "ValueError", (gpu, damage, got[0])
            # whatever was written is a prefix of the clean run's outputs, whole lines only, and stops short of the damaged copy
            for b in range(6):
                assert clean[1][b].startswith(got[1][b]) and (got[1][b] == "" or got[1][b].endswith("\n"))
            assert sum(t.count("\n") for t in got[1]) <= 2 * 9 * 238
