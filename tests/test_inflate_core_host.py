"""The DEFLATE decoder's logic (xenomapper_amd/csrc/xm_inflate_core.h) on this CPU-only container: the same source compiled
for the host as a chain of one lane (tests/inflate_core_host.cpp), under ASan + UBSan, against zlib -- every BGZF block of the
BAM fixtures (CRC-32 of the member trailers) and a few hundred raw-DEFLATE streams written by zlib at all levels and
strategies at shifted alignments, plus damaged streams that must end with a status inside their own block.  The GPU
build of the same source (GS lanes per chain, cross-lane cooperation) is tests/test_inflate_gpu.py; nothing in the
product calls the host build."""
import glob
import os
import subprocess

import numpy as np

from tests import helpers as H


def test_decoder_logic_against_zlib_on_the_host(tmp_path):
    exe = str(tmp_path / "inflate_core_host")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-Wno-unknown-pragmas",
                           os.path.join(H.REPO, "tests", "inflate_core_host.cpp"), "-o", exe, "-lz"])
    bams = sorted(glob.glob(os.path.join(H.GOLDEN, "*.bam")) + glob.glob(os.path.join(H.GOLDEN, "ref_data", "*.bam")))
    assert len(bams) >= 3
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    proc = subprocess.run([exe] + bams + ["--fuzz", "400", "11"], capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0, (proc.stdout + proc.stderr)[-2000:]
    assert "fuzz: 400 streams, 0 failures" in proc.stdout and proc.stdout.count(" 0 bad") == len(bams)
    assert "runtime error" not in proc.stderr and "AddressSanitizer" not in proc.stderr, proc.stderr[-2000:]


def test_bgzf_index_walks_the_member_headers():
    """xm_bgzf_index (host only): block table of the fixtures against a plain-Python walk of the gzip members; windows
    (max_out) continue where the previous call stopped; a non-BGZF image is refused."""
    import gzip
    import struct
    from xenomapper_amd import _ffi
    path = os.path.join(H.GOLDEN, "long_cigar_cg.bam")
    image = np.fromfile(path, dtype=np.uint8)
    blocks, crc, nxt, total = _ffi.bgzf_index(image)
    raw = image.tobytes()
    p, want = 0, []
    while p < len(raw):
        xlen, = struct.unpack_from("<H", raw, p + 10)
        bsize, = struct.unpack_from("<H", raw, p + 16)
        c, isize = struct.unpack_from("<II", raw, p + bsize + 1 - 8)
        want.append((p + 12 + xlen, bsize + 1 - 12 - xlen - 8, isize, c))
        p += bsize + 1
    assert nxt == len(raw) and len(blocks) == len(want) and total == len(gzip.decompress(raw))
    assert [(int(b["cdata_off"]), int(b["cdata_len"]), int(b["isize"])) for b in blocks] == [w[:3] for w in want]
    assert crc.tolist() == [w[3] for w in want]
    assert np.array_equal(blocks["out_off"], np.cumsum(blocks["isize"]) - blocks["isize"])
    got, at = [], 0
    while at < len(raw):
        part, _, at, out = _ffi.bgzf_index(image, at, 200_000)
        assert 0 < len(part) and (out >= 200_000 or at == len(raw)) and int(part["out_off"][0]) == 0
        got += part["cdata_off"].tolist()
    assert got == [w[0] for w in want]
    import pytest
    with pytest.raises(ValueError):
        _ffi.bgzf_index(np.frombuffer(gzip.compress(b"plain gzip, no BC field"), dtype=np.uint8))
    with pytest.raises(ValueError):
        _ffi.bgzf_index(image[:len(raw) - 5])
