"""The process-wide GPU front ends give their memory back (VERDICT r5 #1a, #7): xenomapper_amd.xenomapper.release_buffers()
destroys default_stripper() / default_bamdev(); the library's own book of page-locked bytes (xm_pinned_bytes, ABI 6) and the
device's free memory (hipMemGetInfo) return to where they were before the file runs that made the front ends allocate."""
import io
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

DATA = os.path.join(H.REPO, "tests", "golden", "ref_data")


def _free_hbm():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def _run_both_front_ends(xm, tmp_path):
    import sys
    sys.path.insert(0, os.path.join(H.REPO, "tools"))
    import bench_bam
    bams = []
    for tag in ("human", "mouse"):
        p = str(tmp_path / ("%s.bam" % tag))
        if not os.path.exists(p):
            bench_bam.tiled_bam(os.path.join(DATA, "paired_end_testdata_%s.bam" % tag), p, 200, aligned=True)
        bams.append(p)
    sinks = [io.StringIO() for _ in range(6)]
    counts = xm.classify_sam_files(bams[0], bams[1], *sinks, paired=True, bam=True)
    assert sum(counts.values()) == 200 * 238 and xm.LAST_FILE_PROFILE.get("strip_kernels_ms", 0) > 0
    sams = []
    for k, text in enumerate(H.synth.sam_text_pair(n_pairs=20_000, seed=3, profile="bowtie2", paired=True, read_len=150)[:2]):
        p = tmp_path / ("s%d.sam" % k)
        p.write_text(text)                                            # (the file path starts behind the header lines itself)
        sams.append(str(p))
    sinks = [io.StringIO() for _ in range(6)]
    counts = xm.classify_sam_files(sams[0], sams[1], *sinks, paired=True)
    assert sum(counts.values()) == 20_000 and xm.LAST_FILE_PROFILE.get("strip_kernels_ms", 0) > 0


def test_release_buffers_returns_pinned_and_device_memory(tmp_path):
    from xenomapper_amd import _ffi, xenomapper as xm
    xm.release_buffers()
    base = _ffi.pinned_bytes()
    assert base["registered"] == 0
    _run_both_front_ends(xm, tmp_path)                                # first run: the context's own workspace grows to its size
    held = _ffi.pinned_bytes()
    assert held["allocated"] > base["allocated"] + (1 << 20)          # the front ends keep their staging / text / table buffers
    after = xm.release_buffers()
    assert after["allocated"] == base["allocated"] and after["registered"] == 0
    assert xm._bamdev is None and xm._stripper is None
    free0 = _free_hbm()
    _run_both_front_ends(xm, tmp_path)                                # second run: fresh front ends, same answers
    assert _ffi.pinned_bytes()["allocated"] == held["allocated"]      # and the same buffers as the first time
    assert _free_hbm() < free0 - (1 << 20)
    after = xm.release_buffers()
    assert after["allocated"] == base["allocated"]
    assert _free_hbm() >= free0 - (8 << 20)                           # device memory of the front ends is back as well
    assert after["peak"] >= held["allocated"]


def test_a_closed_front_end_holds_nothing():
    """A stripper / BAM front end of one's own: reserve() allocates, close() gives all of it back (the book counts both)."""
    from xenomapper_amd import _ffi, xenomapper as xm
    ctx = xm.default_context()
    base = _ffi.pinned_bytes()["allocated"]
    strip = _ffi.Stripper(ctx)
    strip.reserve(0, 8 << 20, 100_000)
    strip.reserve(1, 4 << 20, 50_000)
    mid = _ffi.pinned_bytes()["allocated"]
    assert mid >= base + 2 * (8 << 20) + 2 * (4 << 20)                # two files per slot: the staging windows alone
    strip.close()
    assert _ffi.pinned_bytes()["allocated"] == base
    dev = _ffi.BamDev(ctx)
    dev.reserve(0, 4 << 20, 16 << 20, 4096, 1 << 16)
    assert _ffi.pinned_bytes()["allocated"] >= base + 2 * (4 << 20) + 2 * (16 << 20)
    dev.close()
    assert _ffi.pinned_bytes()["allocated"] == base
