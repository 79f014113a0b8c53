import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# Should the process die inside the HIP runtime again (round 5: one SIGABRT in tensor.to(device), no message), it must say why:
# ROCclr's errors (level 1 = errors only) go to a file that travels back from the GPU box, glibc's own fatal messages to stderr
# instead of the controlling terminal.  Set before anything loads the runtime; harmless where there is no GPU.
_LOG_DIR = os.path.join(REPO, "gpurun_out")
try:
    os.makedirs(_LOG_DIR, exist_ok=True)
    os.environ.setdefault("AMD_LOG_LEVEL", "1")
    os.environ.setdefault("AMD_LOG_LEVEL_FILE", os.path.join(_LOG_DIR, "amd_log_tests.txt"))
except OSError:                                                       # a read-only checkout: the runtime keeps its defaults
    pass
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")


def _install_abort_trace():
    """Native frames at SIGABRT (tests/abort_trace.c), to the real stderr and to gpurun_out/abort_trace_<pid>.txt.  Installed
    in pytest_configure, i.e. while output capture is suspended and before pytest's faulthandler plugin chains to it."""
    import ctypes
    import subprocess
    import tempfile
    try:
        os.makedirs(_LOG_DIR, exist_ok=True)
        so = os.path.join(tempfile.gettempdir(), "xm_abort_trace_%d.so" % os.getuid())
        src = os.path.join(REPO, "tests", "abort_trace.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", src, "-o", so + ".tmp%d" % os.getpid()])
            os.replace(so + ".tmp%d" % os.getpid(), so)
        lib = ctypes.CDLL(so)
        fd_err = os.dup(2)
        fd_file = os.open(os.path.join(_LOG_DIR, "abort_trace_%d.txt" % os.getpid()), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        lib.xm_install_abort_trace(fd_err, fd_file)
    except Exception as exc:                                         # noqa: BLE001 -- a diagnostic aid must never fail the session
        sys.stderr.write("abort trace not installed: %r\n" % (exc,))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if _has_gpu():
        _install_abort_trace()


def pytest_unconfigure(config):
    # an abort_trace file that stayed empty says nothing: remove it
    path = os.path.join(_LOG_DIR, "abort_trace_%d.txt" % os.getpid())
    try:
        if os.path.exists(path) and os.path.getsize(path) == 0:
            os.unlink(path)
    except OSError:
        pass


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="module", autouse=True)
def _front_ends_released_between_modules(request):
    """GPU sessions: when a test module ends, the process-wide GPU front ends give their buffers back
    (xenomapper.release_buffers()) and nothing may still be page-locked through xm_host_register -- so that no module runs on
    top of the several GB of page-locked memory an earlier module's file runs left behind (round 5's abort happened in exactly
    that state), and a leak shows up in the module that made it."""
    yield
    xm = sys.modules.get("xenomapper_amd.xenomapper")
    ffi = sys.modules.get("xenomapper_amd._ffi")
    if xm is None or ffi is None or not _has_gpu():
        return
    left = xm.release_buffers()
    assert left["registered"] == 0, "memory still page-locked through xm_host_register: %r" % (left,)
