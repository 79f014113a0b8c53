"""printf("%g") for binary32 as the device SAM printer does it (xenomapper_amd/csrc/xm_fmtg.h: optional fields of type f and
B:f, which `samtools view` prints with %g and the reference then reads as text, xenomapper.py:56-64) -- the same source compiled
for the host against the C library's snprintf on ~14 M bit patterns: every exponent, binade edges, the exact integers around and
above 10^6 (ties), halves and eighths at the 6-digit boundary, decimal literals, subnormals, zeros, infinities and NaNs of both
signs, random patterns.  The device build is exercised by tests/test_bam_gpu.py."""
import os
import subprocess

from tests import helpers as H


def test_fmt_g_f32_equals_snprintf(tmp_path):
    exe = str(tmp_path / "fmtg_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fsanitize=undefined", "-Wall", "-Wextra", "-Werror", "-Wno-unknown-pragmas",
                           os.path.join(H.REPO, "tests", "fmtg_host.cpp"), "-o", exe])
    proc = subprocess.run([exe, "16411", "500000", "7", "13"], capture_output=True, text=True, timeout=600,
                          env=dict(os.environ, UBSAN_OPTIONS="halt_on_error=1"))
    assert proc.returncode == 0, (proc.stdout + proc.stderr)[-2000:]
    assert " 0 mismatches" in proc.stdout and "runtime error" not in proc.stderr
