#!/usr/bin/env python3
"""End-to-end BAM -> six SAM files throughput of the file fast path (native BGZF/BAM decoder -> stripper -> GPU -> writer).

    python tools/bench_bam.py --copies 4000

Input: the reference's two paired-end BAM fixtures (tests/golden/ref_data/), their alignment records repeated
`--copies` times.  The record section is deflated once into BGZF blocks and the same blocks are written again for every
copy, so building a multi-gigabyte input takes seconds.  Reports read-pairs/s and GB/s of BAM and of decoded SAM text.
"""
import argparse
import gzip
import json
import os
import struct
import sys
import time
import zlib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
DATA = os.path.join(REPO, "tests", "golden", "ref_data")


def bgzf_blocks(payload, level=6, chunk=0xff00):
    out = []
    for at in range(0, len(payload), chunk):
        part = payload[at:at + chunk]
        comp = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = comp.compress(part) + comp.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" +
                   struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(part), len(part)))
    return b"".join(out)


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def record_aligned_blocks(payload, level=6, limit=0xff00):
    """BGZF blocks that each begin with an alignment record, as htslib / samtools write them (bgzf_flush_try in front of every
    record): the payload is cut at the last record boundary that keeps a block within `limit` bytes."""
    cuts, at, start = [], 0, 0
    while at < len(payload):
        size, = struct.unpack_from("<i", payload, at)
        nxt = at + 4 + size
        if nxt - start > limit and at > start:
            cuts.append((start, at))
            start = at
        at = nxt
    cuts.append((start, len(payload)))
    return b"".join(bgzf_blocks(payload[a:b], level, chunk=limit) for a, b in cuts)


def tiled_bam(src, dst, copies, aligned=True, level=6):
    """aligned: record-aligned blocks (what samtools writes; the GPU BAM path's record walk is parallel over such blocks);
    False: blocks of a fixed size that cut through records (what htsjdk writes; the device reports them and the host
    walks the chain).  level: zlib's (0: stored blocks, `samtools view -u`)."""
    raw = gzip.decompress(open(src, "rb").read())
    assert raw[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", raw, 4)
    at = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, at)
    at += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, at)
        at += 4 + l_name + 4
    head = bgzf_blocks(raw[:at])
    records = record_aligned_blocks(raw[at:], level) if aligned else bgzf_blocks(raw[at:], level)
    with open(dst, "wb") as fh:
        fh.write(head)
        for _ in range(copies):
            fh.write(records)
        fh.write(BGZF_EOF)
    return len(raw) - at


def run(copies=4000, threads=0, workdir="/dev/shm", cigar_scores=False, single_end=False, to_files=False):
    """One warm-up and one timed pass over the tiled fixtures; returns the result record (also bench.py's `e2e.bam`).
    cigar_scores: the --cigar_scores plugin (AS made of NM + CIGAR) instead of the AS / XS tags.  single_end: the files as
    single-end input, as the command line runs it: the skipping walk (every run of equal names yields its first record)."""
    import types
    a = types.SimpleNamespace(copies=copies, threads=threads, dir=workdir)
    from xenomapper_amd import _host, xenomapper as xm
    paths = []
    for tag in ("human", "mouse"):
        path = os.path.join(a.dir, "xm_bam_%s_%d.bam" % (tag, os.getpid()))
        tiled_bam(os.path.join(DATA, "paired_end_testdata_%s.bam" % tag), path, a.copies)
        paths.append(path)
    size = sum(os.path.getsize(p) for p in paths)
    names = ("primary_specific", "secondary_specific", "primary_multi", "secondary_multi", "unassigned", "unresolved")
    out_paths = [os.path.join(a.dir, "xm_bam_out_%s_%d.sam" % (k, os.getpid())) for k in names] if to_files else []
    sinks = {k: open(os.devnull, "wt") for k in names}
    try:
        xm.default_context()
        first = None
        for _warm in (True, False):
            if to_files:                                              # real files (tmpfs): every pass starts from empty ones
                for s_ in sinks.values():
                    s_.close()
                sinks = {k: open(p_, "wt") for k, p_ in zip(names, out_paths)}
            t0 = time.perf_counter()
            counts = xm.classify_sam_files(paths[0], paths[1], paired=not single_end, n_threads=a.threads, bam=True,
                                           tag_func=xm.get_cigarbased_AS_tag if cigar_scores else xm.get_tag, **sinks)
            el = time.perf_counter() - t0
            first = el if first is None else first
        units = sum(counts.values())
        return {"metric": "end-to-end %s/s (BAM in, six SAM files out)" % ("reads" if single_end else "read-pairs"), "value": units / el,
                "plugin": "get_cigarbased_AS_tag" if cigar_scores else "get_tag", "units": units, "seconds": el, "first_run_seconds": first, "bam_bytes": size, "bam_GBps": size / el / 1e9,
                "threads": a.threads or _host.lib().xmh_default_threads(),
                "phases": {k: round(v, 4) for k, v in xm.LAST_FILE_PROFILE.items()}}
    finally:
        for s_ in sinks.values():
            s_.close()
        for p in paths + out_paths:
            if os.path.exists(p):
                os.unlink(p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=4000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--cigar_scores", action="store_true")
    ap.add_argument("--single_end", action="store_true")
    ap.add_argument("--files", action="store_true", help="six real output files in --dir instead of /dev/null")
    a = ap.parse_args()
    print(json.dumps(run(a.copies, a.threads, a.dir, a.cigar_scores, a.single_end, a.files)))


if __name__ == "__main__":
    main()
