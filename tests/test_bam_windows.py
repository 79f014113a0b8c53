"""The BAM window machinery of the file path without a GPU: _BamSource (three rotating buffers, a decode-ahead thread,
line descriptions carried from window to window) + Parser.parse_pre, driven exactly as _run_files drives them -- many
windows, 16 decoder / parser threads, the next window parsed by a helper thread while the current one is "settled".
Every block is checked against the plain text parse of the same windows and, at the end, the records against the oracle's
reading of the same BAM (oracle/bam_oracle.py -> text -> oracle.read_pairs).

Why this file exists: `python tools/bench_bam.py --copies 8000` dumped core twice on a GPU box in round 4 while this path
was being written (gpurun_out/r4/bam2.log; DESIGN.md section 8 f-3 has what is known).  tests/test_host_sanitizers.py
runs this file against the ASan + UBSan build of the host library; XM_BAM_WINDOWS_COPIES=8000 XENOMAPPER_WINDOW_MB=128
reproduces the shape of the run that crashed."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from tests import helpers as H

sys.path.insert(0, os.path.join(H.REPO, "tools"))

DATA = os.path.join(H.REPO, "tests", "golden", "ref_data")
COPIES = int(os.environ.get("XM_BAM_WINDOWS_COPIES", "300"))


def _tiled(tmp_path, copies):
    import bench_bam
    paths = []
    for tag in ("human", "mouse"):
        p = str(tmp_path / ("%s.bam" % tag))
        bench_bam.tiled_bam(os.path.join(DATA, "paired_end_testdata_%s.bam" % tag), p, copies)
        paths.append(p)
    return paths


def _walk(paths, window, threads, score_mode, paired, skip_repeated, use_pre, check_text=True):
    """The window loop of _run_files (parse_next in a helper thread, advance, next), classify and emit left out.
    -> (records, [name of record k], per-column arrays)"""
    from xenomapper_amd import _host, xenomapper as xm
    os.environ["XENOMAPPER_BAM_PRE"] = "1" if use_pre else "0"
    sources = [xm._BamSource(p, threads) for p in paths]
    parsers = [_host.Parser(threads), _host.Parser(threads)]
    checker = _host.Parser(threads)                               # the text parse a description block is compared with
    pool = ThreadPoolExecutor(max_workers=1)
    n_total, cols, names = 0, [[] for _ in range(4)], []
    stats = {"windows": 0, "pre_blocks": 0}

    def parse_next(which, want):
        wins = [src.window(want) for src in sources]
        blk = None
        if all(src.pre_ok for src in sources):
            pw = [src.pre_window() for src in sources]
            blk = parsers[which].parse_pre(wins[0][0], wins[0][1], wins[0][2], wins[0][3], pw[0][0], pw[0][1],
                                           wins[1][0], wins[1][1], wins[1][2], wins[1][3], pw[1][0], pw[1][1],
                                           score_mode, paired, skip_repeated, paired, xm.FILE_MAX_RECORDS)
            if blk is not None:
                stats["pre_blocks"] += 1
                if check_text:                                    # the same windows through the text rules: identical block
                    ref = checker.parse(wins[0][0], wins[0][1], wins[0][2], wins[0][3], wins[1][0], wins[1][1],
                                                   wins[1][2], wins[1][3], score_mode, paired, skip_repeated, paired,
                                                   xm.FILE_MAX_RECORDS)
                    assert (ref.n, ref.consumed, ref.consumed_lines, ref.ended, ref.starved, ref.mismatch_at) == \
                           (blk.n, blk.consumed, blk.consumed_lines, blk.ended, blk.starved, blk.mismatch_at)
                    for c in range(4):
                        assert np.array_equal(np.asarray(ref.cols[c][:ref.n]), np.asarray(blk.cols[c][:blk.n])), c
                    assert np.array_equal(ref.unit_bits[:(ref.n + 63) // 64], blk.unit_bits[:(blk.n + 63) // 64])
                    for f in (0, 1):
                        assert np.array_equal(ref.line_off[f][:ref.n], blk.line_off[f][:blk.n])
                        assert np.array_equal(ref.line_len[f][:ref.n], blk.line_len[f][:blk.n])
        if blk is None:
            blk = parsers[which].parse(wins[0][0], wins[0][1], wins[0][2], wins[0][3], wins[1][0], wins[1][1], wins[1][2],
                                       wins[1][3], score_mode, paired, skip_repeated, paired, xm.FILE_MAX_RECORDS)
        return blk, [w[0] for w in wins], [w[1] for w in wins], [w[3] for w in wins]

    which, future = 0, None
    try:
        while True:
            parsed = future.result() if future is not None else parse_next(which, window)
            future = None
            block, raws, pos, eofs = parsed
            stats["windows"] += 1
            progressed = block.consumed[0] > 0 or block.consumed[1] > 0
            if block.starved and not progressed and not (eofs[0] and eofs[1]):
                window *= 2
                continue
            assert block.mismatch_at < 0
            last = block.ended or (eofs[0] and eofs[1] and not progressed)
            if not last:
                for f in (0, 1):
                    sources[f].advance(block.consumed[f], block.consumed_lines[f])
                future = pool.submit(parse_next, which ^ 1, window)
            # "settle": what the writer and the classifier read of this block while the next window is parsed
            n = block.n
            halo = 1 if (paired and n_total and n) else 0         # keep_halo: record 0 is the previous window's last one
            for c in range(4):
                cols[c].append(np.array(block.cols[c][halo:n]))
            off, ln = block.line_off[0][:n], block.line_len[0][:n]
            raw = raws[0]
            for k in (halo, n // 2, n - 1) if n > halo else ():
                line = bytes(raw[pos[0] + int(off[k]):pos[0] + int(off[k]) + int(ln[k])])
                names.append((n_total + k - halo, line.split(b"\t", 1)[0]))
            n_total += n - halo
            if last:
                break
            which ^= 1
    finally:
        if future is not None:
            future.result()
        pool.shutdown(wait=True)
        for p in parsers + [checker]:
            p.close()
        for s in sources:
            s.close()
    return n_total, names, [np.concatenate(c) if c else np.zeros(0, np.int32) for c in cols], stats


@pytest.mark.parametrize("mode", ["as_xs", "cigar"])
def test_many_windows_sixteen_threads_decode_ahead(tmp_path, monkeypatch, mode):
    from xenomapper_amd import _host, xenomapper as xm
    window = int(os.environ.get("XENOMAPPER_WINDOW_MB", "0")) << 20 or (3 << 20)
    monkeypatch.setattr(xm, "FILE_WINDOW_BYTES", window)
    paths = _tiled(tmp_path, COPIES)
    score_mode = _host.SCORE_CIGAR if mode == "cigar" else _host.SCORE_AS_XS
    n, names, cols, stats = _walk(paths, window, 16, score_mode, True, False, True, check_text=COPIES <= 1000)
    assert n == COPIES * 476                                      # 238 pairs = 476 records per copy of the fixture
    assert stats["pre_blocks"] >= 2 and stats["windows"] >= 3     # several windows, and the description path took them
    # against the text path of the same run shape (decoder without descriptions, tokenising stripper)
    n2, names2, cols2, _ = _walk(paths, window, 16, score_mode, True, False, False)
    assert n2 == n and names2 == names
    for c in range(4):
        assert np.array_equal(cols[c], cols2[c]), c
    # and against the oracle's reading of the fixture itself: every copy repeats it
    import io
    from oracle import bam_oracle
    texts = []
    for tag in ("human", "mouse"):
        with open(os.path.join(DATA, "paired_end_testdata_%s.bam" % tag), "rb") as fh:
            header, lines = bam_oracle.bam_to_sam(fh.read())
        texts.append("".join(l + "\n" for l in lines))
    pairs = list(H.ORACLE.read_pairs(io.StringIO(texts[0]), io.StringIO(texts[1]), False))
    assert len(pairs) == 476
    if mode == "as_xs":
        want = [np.array([-2**31 if v == H.NEG else int(v) for v in (H.ORACLE.tag_score(p[f], tag=t) for p in pairs)], dtype=np.int32)
                for f in (0, 1) for t in ("AS", "XS")]
        for c in range(4):
            assert np.array_equal(cols[c], np.tile(want[c], COPIES)), c
    for k, name in names:
        assert name == pairs[k % 476][0][0].encode()


def test_growing_windows_and_a_window_larger_than_the_head_room(tmp_path, monkeypatch):
    """A first window far smaller than a line forces the 'same window again, larger' branch; a tail longer than HEAD the
    'grown' branch of _BamSource.window."""
    from xenomapper_amd import _host, xenomapper as xm
    monkeypatch.setattr(xm, "FILE_WINDOW_BYTES", 1 << 16)
    monkeypatch.setattr(xm._BamSource, "HEAD", 1 << 10)
    paths = _tiled(tmp_path, 40)
    n, names, cols, stats = _walk(paths, 64, 16, _host.SCORE_AS_XS, True, False, True)
    assert n == 40 * 476 and stats["windows"] > 10
    n2, names2, cols2, _ = _walk(paths, 1 << 16, 3, _host.SCORE_AS_XS, True, False, False)
    assert n2 == n
    for c in range(4):
        assert np.array_equal(cols[c], cols2[c])


def test_descriptions_that_point_outside_their_operation_array_are_refused(tmp_path):
    """xmh_parse_pre checks ops_at / n_ops of every line against the operation array it is given (the arrays are rebased in
    three places on their way here and arrive as raw pointers): a description past the array is an error code, not a read
    out of bounds.  AS/XS mode never looks at the operations and is unaffected."""
    from xenomapper_amd import _host, xenomapper as xm
    paths = _tiled(tmp_path, 3)
    os.environ["XENOMAPPER_BAM_PRE"] = "1"
    sources = [xm._BamSource(p, 4) for p in paths]
    parser = _host.Parser(4)
    try:
        wins = [src.window(1 << 24) for src in sources]
        pw = [src.pre_window() for src in sources]
        args = lambda pre1, ops1: (wins[0][0], wins[0][1], wins[0][2], wins[0][3], pre1, ops1,              # noqa: E731
                                   wins[1][0], wins[1][1], wins[1][2], wins[1][3], pw[1][0], pw[1][1])
        good = parser.parse_pre(*args(pw[0][0], pw[0][1]), _host.SCORE_CIGAR, True, False, True, xm.FILE_MAX_RECORDS)
        assert good is not None and good.n == 3 * 476
        for how in ("past_end", "short_array", "backwards"):
            pre = pw[0][0].copy()
            ops = pw[0][1]
            if how == "past_end":
                pre[700, _host.PRE_OPS_AT] = ops.shape[0] + 5
            elif how == "short_array":
                ops = ops[:ops.shape[0] // 2].copy()
            else:
                pre[0, _host.PRE_OPS_AT] = pre[900, _host.PRE_OPS_AT]
            with pytest.raises(RuntimeError):
                parser.parse_pre(*args(pre, ops), _host.SCORE_CIGAR, True, False, True, xm.FILE_MAX_RECORDS)
            ok = parser.parse_pre(*args(pre, ops), _host.SCORE_AS_XS, True, False, True, xm.FILE_MAX_RECORDS)
            assert ok is not None and ok.n == 3 * 476
    finally:
        parser.close()
        for s_ in sources:
            s_.close()
