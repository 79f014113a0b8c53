"""The C++ host library under AddressSanitizer + UndefinedBehaviorSanitizer: the host-parser test files and the BAM
window test are re-run in a subprocess against a sanitized build (CPU only; the GPU pool has no ASan)."""
import os
import subprocess
import sys

import pytest

from tests import helpers as H


def _asan_runtime():
    try:
        path = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    except Exception:
        return None
    return path if os.path.isabs(path) and os.path.exists(path) else None


@pytest.mark.skipif(_asan_runtime() is None, reason="no libasan in this toolchain")
def test_host_parser_under_asan_ubsan(tmp_path):
    from xenomapper_amd import build
    lib = build.build_host_sanitized(str(tmp_path / "libxenomapper_host_asan.so"))
    env = dict(os.environ)
    env.update({"XENOMAPPER_HOST_LIB": lib, "LD_PRELOAD": _asan_runtime(),
                "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1", "PYTHONDONTWRITEBYTECODE": "1"})
    proc = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                           os.path.join(H.REPO, "tests", "test_host_parser.py"),
                           os.path.join(H.REPO, "tests", "test_host_fuzz.py"),
                           # _BamSource + xmh_parse_pre over many windows, 16 threads, decode-ahead: the shape of the round-4
                           # crash (XM_BAM_WINDOWS_COPIES=8000 XENOMAPPER_WINDOW_MB=128 is the run itself; clean, 21 s here)
                           os.path.join(H.REPO, "tests", "test_bam_windows.py")],
                          cwd=H.REPO, env=env, capture_output=True, text=True, timeout=900)
    tail = (proc.stdout + proc.stderr)[-3000:]
    assert proc.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
