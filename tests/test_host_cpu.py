"""CPU-only checks of the product's host side: the C ABI library loads and exports every symbol the
headers declare, the text-level host functions match the golden vectors, and the classification
path refuses to run (loudly) without the HIP device -- there is no CPU fallback."""
import ctypes
import io
import os
import re

import pytest

from tests import helpers as H
from tests.helpers import NEG


def _declared(header):
    text = open(os.path.join(H.REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xm[hs]?_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    from xenomapper_amd import _ffi, build
    build.build_hip()
    names = sorted(_declared("xenomapper_hip.h") + _declared("xenomapper_strip.h") + _declared("xenomapper_bgzf.h"))
    assert len(names) >= 17 + 10 + 12
    L = ctypes.CDLL(_ffi.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_ffi.EXPORTED) == names
    assert _ffi.lib().xm_abi_version() == 6 and _ffi.lib().xms_abi_version() == 3
    assert b"gfx950" in _ffi.lib().xm_strerror(-2)


def test_code_object_is_gfx950_only():
    from xenomapper_amd import _ffi, build
    build.build_hip()
    blob = open(_ffi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback_without_device():
    from xenomapper_amd import xenomapper as xm
    with pytest.raises(RuntimeError, match="gfx950"):
        xm.get_mapping_state(200, 199, 199, 198)
    pairs = [(["r1"] + [""] * 10 + ["AS:i:5"], ["r1"] + [""] * 10 + ["AS:i:3"])]
    with pytest.raises(RuntimeError, match="gfx950"):
        xm.main_single_end(iter(pairs), primary_specific=io.StringIO())
    with pytest.raises(RuntimeError, match="gfx950"):
        xm.get_cigarbased_AS_tag([""] * 5 + ["50M"] + [""] * 5 + ["NM:i:1"])


def test_text_level_tag_functions_match_golden():
    """get_tag / get_tag_with_ZS_as_XS and the non-AS branch of get_cigarbased_AS_tag are host text
    work and need no device."""
    from xenomapper_amd import xenomapper as xm
    funcs = {"get_tag": xm.get_tag, "get_tag_with_ZS_as_XS": xm.get_tag_with_ZS_as_XS,
             "get_cigarbased_AS_tag": xm.get_cigarbased_AS_tag}
    n = 0
    for case in H.golden("g2_tag_parsers.json")["cases"]:
        if case["func"] == "get_cigarbased_AS_tag" and case["tag"] == "AS":
            continue
        n += 1
        exp = case["expect"]
        if "error" in exp:
            with pytest.raises(Exception) as info:
                funcs[case["func"]](case["fields"], tag=case["tag"])
            assert type(info.value).__name__ == exp["error"], case
        else:
            got = funcs[case["func"]](case["fields"], tag=case["tag"])
            want = H.unnum(exp["value"])
            assert type(got).__name__ == exp["type"]
            assert got == want or (got != got and want != want), case
    assert n > 150


def test_cigar_column_parser_matches_oracle():
    from xenomapper_amd import xenomapper as xm
    for case in H.golden("g2_tag_parsers.json")["cases"]:
        if case["func"] != "get_cigarbased_AS_tag" or case["tag"] != "AS":
            continue
        exp = case["expect"]
        if "error" in exp:
            with pytest.raises(Exception) as info:
                xm._cigar_columns(case["fields"])
            assert type(info.value).__name__ == exp["error"]
            continue
        nm, ops = xm._cigar_columns(case["fields"])
        if nm is None:
            assert H.unnum(exp["value"]) == NEG
            continue
        score = -6 * nm
        for v in ops:
            if v & 15 in (1, 2):
                score -= 5 + 3 * (v >> 4)
            elif v & 15 == 4:
                score -= 2 * (v >> 4)
        assert score == exp["value"], case


def test_headers_summary_and_reader():
    from xenomapper_amd import xenomapper as xm
    g3 = {c["name"]: c for c in H.golden("g3_end_to_end.json")["cases"]}
    t1, t2 = H.case_texts(g3["ref_pe_liberal"])
    outs = {name: io.StringIO() for name in H.STATES}
    s1, s2 = io.StringIO(t1), io.StringIO(t2)
    xm.process_headers(s1, s2, **outs)
    want = H.golden("g4_headers.json")
    for name in H.STATES:
        assert outs[name].getvalue() == want[name]
    # reference test: header block lengths (tests/test_xenomapper.py:46-51)
    assert [len(outs[n].getvalue()) for n in ("primary_specific", "secondary_specific", "primary_multi",
                                               "secondary_multi", "unassigned", "unresolved")] == \
        [695, 629, 708, 642, 705, 705]
    pairs = list(xm.getReadPairs(s1, s2))
    o1, o2 = io.StringIO(t1), io.StringIO(t2)
    H.ORACLE.read_header(o1), H.ORACLE.read_header(o2)
    assert pairs == list(H.ORACLE.read_pairs(o1, o2))
    assert len(pairs) == 476
    # skip_repeated_reads and the mixed-whitespace single-end fixture
    t1, t2 = H.case_texts(g3["all36_se_skip"])
    s1, s2 = io.StringIO(t1), io.StringIO(t2)
    xm.get_sam_header(s1), xm.get_sam_header(s2)
    assert len(list(xm.getReadPairs(s1, s2, skip_repeated_reads=True))) == g3["all36_se_skip"]["expect"]["n_records"]
    # summary (tests/test_xenomapper.py:235-245)
    buf = io.StringIO()
    xm.output_summary({'foo': 1, 'bar': 101}, outfile=buf)
    assert buf.getvalue() == H.ORACLE.summary_text({'foo': 1, 'bar': 101})
    # name mismatch -> AssertionError from the reader
    a = io.StringIO("r1\t0\n")
    b = io.StringIO("r2\t0\n")
    with pytest.raises(AssertionError):
        list(xm.getReadPairs(a, b))
    with pytest.raises(IndexError):
        xm.get_sam_header(io.StringIO("@HD\tVN:1.0\n"))


def test_write_bytes_reaches_every_kind_of_sink(tmp_path):
    """Bin texts reach text files (through the binary buffer, in order with text written around them), sinks whose
    encoding is not an ASCII superset, and StringIO unchanged."""
    import io
    import numpy as np
    from xenomapper_amd import xenomapper as xm
    data = np.random.default_rng(3).integers(32, 127, size=100_003, dtype=np.uint8)
    body = data.tobytes()
    path = tmp_path / "out.sam"
    with open(path, "wt") as f:
        f.write("@HD\tVN:1.0\n")
        xm._write_bytes(f, data)
        f.write("between\n")
        xm._write_bytes(f, data[:100])
        xm._write_bytes(f, data[:0])
        f.write("end\n")
    assert path.read_bytes() == b"@HD\tVN:1.0\n" + body + b"between\n" + body[:100] + b"end\n"
    with open(path, "at") as f:
        xm._write_bytes(f, data)
    assert path.read_bytes().endswith(b"end\n" + body)
    with open(path, "wt", encoding="utf-16") as f:
        xm._write_bytes(f, data[:50])
    assert path.read_text(encoding="utf-16") == body[:50].decode("ascii")
    s = io.StringIO()
    xm._write_bytes(s, data[:77])
    assert s.getvalue() == body[:77].decode("ascii")


def test_usage_error_prints_the_reference_help(capsys, monkeypatch):
    """No inputs: the error line, the help text and exit status 1, byte for byte what the reference's command line
    prints (G9 `no_inputs_is_a_usage_error`, recorded from the reference run as a child process).  Needs no GPU."""
    import hashlib
    from xenomapper_amd import xenomapper as xm
    case = {c["name"]: c for c in H.golden("g9_cli.json")["cases"]}["no_inputs_is_a_usage_error"]
    monkeypatch.setenv("COLUMNS", "80")
    with pytest.raises(SystemExit) as exc:
        xm.main([])
    assert exc.value.code == case["returncode"] == 1
    out = capsys.readouterr().out
    assert out.startswith("ERROR: You must provide --primary_sam and --secondary_sam")
    assert (hashlib.sha224(out.encode("latin-1")).hexdigest(), len(out)) == (case["stdout"]["sha224"], case["stdout"]["len"])


def test_public_headers_are_plain_c(tmp_path):
    """include/*.h are the drop-in boundary: they must compile as C99 (no C++-isms outside the extern "C" guards), with
    warnings as errors, and as C++17; a C program links every declared symbol of the HIP library without a GPU."""
    import subprocess
    src = tmp_path / "use_headers.c"
    names = _declared("xenomapper_hip.h")
    src.write_text('#include "xenomapper_hip.h"\n#include "xenomapper_host.h"\n'
                   "typedef void (*fn)(void);\n"
                   "static const fn table[] = {" + ", ".join("(fn)%s" % n for n in names) + "};\n"
                   "int main(void) { return (XM_ABI_VERSION == xm_abi_version() && table[0] && "
                   "XM_BINS4_BYTES(2049) == 2048u && XM_UNIQUE_ID_BYTES == 128) ? 0 : 1; }\n")
    inc = os.path.join(H.REPO, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-c", str(src),
                           "-o", str(tmp_path / "c99.o")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", inc, "-x", "c++", "-c", str(src),
                           "-o", str(tmp_path / "cxx.o")])
    from xenomapper_amd import _ffi, build
    build.build_hip()
    exe = tmp_path / "use_headers"
    libdir = os.path.dirname(_ffi.LIB_PATH)
    subprocess.check_call(["gcc", str(tmp_path / "c99.o"), "-o", str(exe), "-L", libdir, "-l:libxenomapper_hip.so",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L", "/opt/rocm/lib"])
    assert subprocess.call([str(exe)]) == 0


def _build_c_example(tmp_path):
    import subprocess
    from xenomapper_amd import _ffi, build
    build.build_hip()
    exe = tmp_path / "classify_pairs"
    libdir = os.path.dirname(_ffi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(H.REPO, "include"),
                           os.path.join(H.REPO, "examples", "classify_pairs.c"), "-L", libdir, "-l:libxenomapper_hip.so",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L", "/opt/rocm/lib", "-o", str(exe)])
    return str(exe)


@pytest.mark.skipif(_gpu_present(), reason="checks the no-GPU behaviour")
def test_c_example_refuses_without_a_device(tmp_path):
    """examples/classify_pairs.c is plain C99 against include/xenomapper_hip.h; without a gfx950 device xm_ctx_create
    fails and the program says so -- no CPU fallback behind the C ABI either."""
    import subprocess
    proc = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True)
    assert proc.returncode == 2 and "gfx950" in proc.stderr and proc.stdout == ""


class _FakeParser(object):
    """Stands in for the C++ writer in the mapped-file tests: `need` bytes of 'x' per call."""

    def __init__(self, need):
        self.need, self.calls = need, 0

    def emit_size(self, paired, b, seg):
        return None, self.need

    def emit_to(self, paired, b, idx, addr, size):
        import ctypes
        self.calls += 1
        ctypes.memset(addr, ord("x"), size)
        return size


@pytest.mark.parametrize("err", ["ENOSPC", "EDQUOT", "EFBIG"])
def test_mapped_writer_does_not_map_what_it_could_not_allocate(tmp_path, monkeypatch, err):
    """fallocate failing for lack of space (or quota, or the file size limit) must not be papered over with a
    sparse ftruncate -- storing into pages that cannot be backed dies with SIGBUS.  The range is given back and the
    caller's ordinary write raises the real error, as the reference's print() would."""
    import errno
    import os
    from xenomapper_amd import xenomapper as xm
    path = tmp_path / "bin.sam"
    with open(path, "wt") as sink:
        sink.write("@HD\tVN:1.0\n")

        def no_space(fd, off, length):
            raise OSError(getattr(errno, err), os.strerror(getattr(errno, err)))
        monkeypatch.setattr(xm, "_fallocate", no_space)
        fake = _FakeParser(4 << 20)
        assert xm._emit_into_file(fake, True, 0, None, sink) is False
        assert fake.calls == 0
    assert path.read_text() == "@HD\tVN:1.0\n"              # not extended, nothing left behind


def test_mapped_writer_falls_back_to_a_sparse_extension_only_where_preallocation_is_unsupported(tmp_path, monkeypatch):
    import errno
    import os
    from xenomapper_amd import xenomapper as xm
    path = tmp_path / "bin.sam"
    with open(path, "wt") as sink:
        sink.write("@HD\n")

        def unsupported(fd, off, length):
            raise OSError(errno.EOPNOTSUPP, os.strerror(errno.EOPNOTSUPP))
        monkeypatch.setattr(xm, "_fallocate", unsupported)
        fake = _FakeParser(2 << 20)
        assert xm._emit_into_file(fake, True, 0, None, sink) is True
        sink.write("tail\n")
    data = path.read_bytes()
    assert data[:4] == b"@HD\n" and data[4:4 + (2 << 20)] == b"x" * (2 << 20) and data[-5:] == b"tail\n"


def test_mapped_writer_extends_files_ahead_and_cuts_them_back(tmp_path, monkeypatch):
    """With the run's `ahead` table the output file stays longer than its content between calls (allocated by a helper
    thread); ordinary writes in between land at the content's end, and finish() leaves exactly the content."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from xenomapper_amd import xenomapper as xm
    path = tmp_path / "bin.sam"
    monkeypatch.setattr(xm, "AHEAD_FACTOR", 2.0)
    ahead, pool = {}, ThreadPoolExecutor(max_workers=1)
    with open(path, "wt") as sink:
        sink.write("@HD\n")
        fake = _FakeParser(2 << 20)
        assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool) is True
        state = ahead[id(sink)]
        assert state.settle(4 + 3 * (2 << 20)) == 4 + 3 * (2 << 20) == os.path.getsize(path)      # content + twice the last call, ahead
        sink.write("small\n")                                                    # a bin too small for the mapping
        assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool) is True  # fits what is there already
        fake.need = 9 << 20
        assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool) is True  # does not fit: extended on the spot
        sink.write("tail\n")
        state.finish(sink)
    pool.shutdown()
    data = path.read_bytes()
    want = b"@HD\n" + b"x" * (2 << 20) + b"small\n" + b"x" * (2 << 20) + b"x" * (9 << 20) + b"tail\n"
    assert data == want


def test_fallocate_is_the_system_call_and_reports_what_it_cannot_do(tmp_path):
    """_fallocate is fallocate(2) itself: it allocates on a file system that can (tmp_path), and on one that cannot it must
    fail with an errno instead of writing zeros block by block as glibc's posix_fallocate does (/proc: EBADF / EOPNOTSUPP /
    ENODEV ... -- anything but success)."""
    import os
    from xenomapper_amd import xenomapper as xm
    path = tmp_path / "f"
    fd = os.open(path, os.O_RDWR | os.O_CREAT)
    try:
        xm._fallocate(fd, 0, 1 << 20)
        assert os.fstat(fd).st_size == 1 << 20
    finally:
        os.close(fd)
    fd = os.open("/proc/self/status", os.O_RDONLY)
    try:
        with pytest.raises(OSError):
            xm._fallocate(fd, 0, 4096)
    finally:
        os.close(fd)


def test_small_bins_wait_for_the_extension_running_ahead(tmp_path, monkeypatch):
    """A bin below the mapping threshold is written through the sink by the caller: _emit_into_file must first have stopped the
    extension that is running further down the same file, and waited for its piece in flight."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from xenomapper_amd import xenomapper as xm
    path = tmp_path / "bin.sam"
    ahead, pool = {}, ThreadPoolExecutor(max_workers=1)
    gate = threading.Event()
    with open(path, "wt") as sink:
        sink.write("@HD\n")
        fake = _FakeParser(2 << 20)
        assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool) is True
        state = ahead[id(sink)]
        state.settle()
        released, real, entered = [], xm._fallocate, threading.Event()

        def slow(fd, off, length):
            entered.set()
            gate.wait(5)
            released.append((off, length))
            return real(fd, off, length)
        monkeypatch.setattr(xm, "_fallocate", slow)
        state.extend_later(pool, state.size + (3 << 20))              # an extension that is still under way
        assert entered.wait(5)                                        # (... really under way: the helper thread is inside fallocate)
        fake.need = 100                                               # too small for the mapping
        threading.Timer(0.2, gate.set).start()
        assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool) is False
        assert len(released) == 1 and not state.busy                 # it waited for the piece in flight; no further piece follows
        sink.write("small\n")
        state.finish(sink)
    pool.shutdown()
    assert path.read_bytes() == b"@HD\n" + b"x" * (2 << 20) + b"small\n"


def test_files_are_extended_in_pieces_towards_the_size_the_run_predicts(tmp_path, monkeypatch):
    """With the fraction of the input read so far, the file is extended towards content / fraction (+ 2 %), piece by piece (each a
    job of its own: several files take turns in the pool), never further than AHEAD_MOST past the content; a writer that needs
    bytes the extension has reached does not wait for the rest; at the end of the input nothing is added; finish() cuts back."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from xenomapper_amd import xenomapper as xm
    monkeypatch.setattr(xm, "AHEAD_PIECE", 1 << 20)
    monkeypatch.setattr(xm, "AHEAD_MOST", 16 << 20)
    calls, real = [], xm._fallocate

    def counted(fd, off, length):
        calls.append((off, length))
        return real(fd, off, length)
    monkeypatch.setattr(xm, "_fallocate", counted)
    path = tmp_path / "bin.sam"
    ahead, pool = {}, ThreadPoolExecutor(max_workers=2)
    with open(path, "wt") as sink:
        fake = _FakeParser(2 << 20)
        assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool, progress=0.25) is True
        state = ahead[id(sink)]
        want = int((2 << 20) / 0.25 * 1.02)
        assert state.settle(want) == want == os.path.getsize(path)
        assert calls[0] == (0, 2 << 20) and all(n <= (1 << 20) for _off, n in calls[1:]) and len(calls) >= 7
        assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool, progress=0.26) is True      # predicted 16.1 MB: the 8.4 there do
        n_before = len(calls)
        assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool, progress=0.001) is True     # predicted 6 GB: capped
        assert state.settle((6 << 20) + (16 << 20)) == (6 << 20) + (16 << 20)
        assert len(calls) > n_before
        assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool, progress=1.0) is True       # the end: nothing more ahead
        size = state.settle()
        assert size == (6 << 20) + (16 << 20) == os.path.getsize(path)
        state.finish(sink)
    pool.shutdown()
    assert path.read_bytes() == b"x" * (8 << 20)


def test_the_writer_keeps_one_long_mapping_per_file_and_survives_a_fill_that_fails(tmp_path, monkeypatch):
    """One mapping per output file, MAP_AHEAD bytes long from where it was made, serves the calls until the content runs off its
    end (then the next one is made and the old one unmapped by the helper pool); a fill that raises leaves nothing of its bin
    behind, drops the mapping, and the next call maps again; finish() unmaps and cuts the file to its content."""
    from concurrent.futures import ThreadPoolExecutor
    from xenomapper_amd import xenomapper as xm
    monkeypatch.setattr(xm, "MAP_AHEAD", 5 << 20)
    made, gone, real_map, real_unmap = [], [], xm._map_file, xm._unmap_file
    monkeypatch.setattr(xm, "_map_file", lambda fd, length, offset: made.append((offset, length)) or real_map(fd, length, offset))
    monkeypatch.setattr(xm, "_unmap_file", lambda addr, length: gone.append(length) or real_unmap(addr, length))
    path = tmp_path / "bin.sam"
    ahead, pool = {}, ThreadPoolExecutor(max_workers=1)
    with open(path, "wt") as sink:
        sink.write("@HD\n")
        fake = _FakeParser(2 << 20)
        for _ in range(4):                                           # 8 MB: the first call's own mapping, then two long ones
            assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool) is True
        state = ahead[id(sink)]
        assert [m[1] for m in made] == [4 + (2 << 20), 5 << 20, 5 << 20] and state.mapped is not None

        class Boom(RuntimeError):
            pass

        def failing(paired, b, idx, dst, cap):
            raise Boom()
        good = fake.emit_to
        fake.emit_to = failing
        with pytest.raises(Boom):
            xm._emit_into_file(fake, True, 0, None, sink, ahead, pool)
        assert state.mapped is None and sink.buffer.tell() == 4 + (8 << 20)
        fake.emit_to = good
        assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool) is True     # maps again, writes behind the content
        assert len(made) == 4 and state.mapped is not None
        sink.write("tail\n")
        state.finish(sink, pool)
        assert state.mapped is None
    pool.shutdown()
    assert len(gone) == len(made)                                    # every mapping was taken down
    assert path.read_bytes() == b"@HD\n" + b"x" * (10 << 20) + b"tail\n"


def test_mapped_writer_never_extends_an_append_mode_file_ahead(tmp_path):
    """O_APPEND writes go to the end of the FILE: such a sink must never be longer than its content."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from xenomapper_amd import xenomapper as xm
    path = tmp_path / "bin.sam"
    path.write_text("@HD\n")
    ahead, pool = {}, ThreadPoolExecutor(max_workers=1)
    with open(path, "at") as sink:
        fake = _FakeParser(2 << 20)
        assert xm._emit_into_file(fake, True, 0, None, sink, ahead, pool) is True
        assert not ahead and os.path.getsize(path) == 4 + (2 << 20)
        sink.write("tail\n")
    pool.shutdown()
    assert path.read_bytes() == b"@HD\n" + b"x" * (2 << 20) + b"tail\n"


def test_header_only_bam_handle_knows_the_header_and_refuses_to_read(tmp_path):
    """xmh_bam_open_header (the GPU BAM path's handle: the file's blocks are not indexed): same header text and same start
    of the records as the full handle -- also when the header spans more BGZF blocks than the first indexing step takes --
    records printed from it equal the full reader's text, and reading through it is refused."""
    import struct
    import sys
    import numpy as np
    from xenomapper_amd import _host
    sys.path.insert(0, os.path.join(H.REPO, "tools"))
    import bench_bam
    src = os.path.join(H.REPO, "tests", "golden", "ref_data", "paired_end_testdata_human.bam")
    image = np.frombuffer(open(src, "rb").read(), dtype=np.uint8)
    # a second image whose header text fills ~600 small blocks (the first indexing step takes 256)
    import gzip
    raw = gzip.decompress(image.tobytes())
    l_text, = struct.unpack_from("<i", raw, 4)
    filler = ("@CO\t" + "x" * 60 + "\n") * 3000
    text = raw[8:8 + l_text].rstrip(b"\0") + filler.encode("ascii")
    big = b"BAM\x01" + struct.pack("<i", len(text)) + text + raw[8 + l_text:]
    big_image = np.frombuffer(bench_bam.bgzf_blocks(big, chunk=320) + bench_bam.BGZF_EOF, dtype=np.uint8)
    for im in (image, big_image):
        full, head = _host.BamReader(im, 2), _host.BamReader(im, 2, header_only=True)
        try:
            assert head.header() == full.header()
            assert head.records_start() == full.records_start()
            buf = np.empty(1 << 16, dtype=np.uint8)
            with pytest.raises((ValueError, RuntimeError)):
                head.read_into(buf, 0)
            assert full.read_into(np.empty(1 << 22, dtype=np.uint8), 0) > 0
        finally:
            full.close()
            head.close()


def test_gpu_bam_cursor_reads_the_reference_names(tmp_path):
    """_GpuBamFile (the GPU BAM path's cursor over a file's BGZF blocks) reads the reference names the device printer writes in
    RNAME / RNEXT from the inflated bytes in front of the first record: the names of the reference's fixtures as the SAM header
    lists them (@SQ SN), also when the header spans hundreds of small blocks, and None -- the host printer then -- for a file
    whose header it cannot follow."""
    import gzip
    import struct
    import sys
    from xenomapper_amd import xenomapper as xm
    sys.path.insert(0, os.path.join(H.REPO, "tools"))
    import bench_bam
    for tag in ("human", "mouse"):
        src = os.path.join(H.REPO, "tests", "golden", "ref_data", "paired_end_testdata_%s.bam" % tag)
        raw = gzip.decompress(open(src, "rb").read())
        l_text, = struct.unpack_from("<i", raw, 4)
        text = raw[8:8 + l_text]
        want = [f[3:] for line in text.split(b"\n") if line.startswith(b"@SQ") for f in line.split(b"\t") if f.startswith(b"SN:")]
        assert len(want) > 1
        g = xm._GpuBamFile(src, 2)
        try:
            assert g.ref_names == want
        finally:
            g.close()
        # the same file with a header text that fills hundreds of small blocks
        filler = ("@CO\t" + "x" * 60 + "\n") * 2000
        big_text = text.rstrip(b"\0") + filler.encode("ascii")
        big = b"BAM\x01" + struct.pack("<i", len(big_text)) + big_text + raw[8 + l_text:]
        path = str(tmp_path / ("big_%s.bam" % tag))
        with open(path, "wb") as fh:
            fh.write(bench_bam.bgzf_blocks(big, chunk=320) + bench_bam.BGZF_EOF)
        g = xm._GpuBamFile(path, 2)
        try:
            assert g.ref_names == want
        finally:
            g.close()
