"""In-tree build of the native parts (hipcc cross-compiles gfx950 without a GPU).

    python -m xenomapper_amd.build        # libxenomapper_hip.so (+ host parser when present)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HIP_LIB = os.path.join(PKG, "libxenomapper_hip.so")
HOST_LIB = os.path.join(PKG, "libxenomapper_host.so")

HIP_SOURCES = ["xm_kernels.hip", "xm_api.hip", "xm_strip.hip", "xm_inflate.hip", "xm_bamdev.hip"]
HOST_SOURCES = ["xm_sam.cpp", "xm_bam.cpp"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, "xm_kernels.h"), os.path.join(CSRC, "xm_inflate_core.h"),
                   os.path.join(CSRC, "xm_bamrec.h"), os.path.join(CSRC, "xm_fmtg.h"), os.path.join(CSRC, "xm_pinned.h"), os.path.join(CSRC, "xm_gather.h"),
                   os.path.join(REPO, "include", "xenomapper_hip.h"), os.path.join(REPO, "include", "xenomapper_strip.h"),
                   os.path.join(REPO, "include", "xenomapper_bgzf.h")]
    if not force and not _stale(HIP_LIB, deps):
        return HIP_LIB
    # XENOMAPPER_HIPCC_FLAGS: extra flags for tuning builds (e.g. -DXM_CIGAR_BLOCK=256); not used by the tests
    # One object per source, compiled side by side (the five translation units share no device code), objects kept under
    # build/obj and reused while neither their source nor any header is newer; then one link.
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-I", os.path.join(REPO, "include")] + \
        os.environ.get("XENOMAPPER_HIPCC_FLAGS", "").split()
    obj_dir = os.path.join(REPO, "build", "obj")
    os.makedirs(obj_dir, exist_ok=True)
    tag = os.path.join(obj_dir, "flags.txt")
    same_flags = os.path.exists(tag) and open(tag).read() == " ".join(flags)
    headers = deps[len(srcs):]
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not same_flags or _stale(obj, [src] + headers):
            cmd = [_hipcc()] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((cmd, subprocess.Popen(cmd)))
    failed = [cmd for cmd, proc in jobs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    with open(tag, "w") as fh:
        fh.write(" ".join(flags))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", HIP_LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return HIP_LIB


def build_host(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    if not all(os.path.exists(s) for s in srcs):
        return None
    deps = srcs + [os.path.join(REPO, "include", "xenomapper_host.h"), os.path.join(CSRC, "xm_pool.h")]
    if not force and not _stale(HOST_LIB, deps):
        return HOST_LIB
    cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra",
           "-I", os.path.join(REPO, "include")] + srcs + ["-o", HOST_LIB, "-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return HOST_LIB


def build_host_sanitized(out_path, verbose=False):
    """ASan + UBSan build of the host library, for CPU-side test runs (GPU ASan is not available on the pool)."""
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fsanitize=address,undefined",
           "-fno-omit-frame-pointer", "-I", os.path.join(REPO, "include")] + srcs + ["-o", out_path, "-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out_path


def build_all(force=False, verbose=False):
    out = [build_hip(force, verbose)]
    host = build_host(force, verbose)
    if host:
        out.append(host)
    return out


if __name__ == "__main__":
    print("\n".join(build_all(force="--force" in sys.argv, verbose=True)))
