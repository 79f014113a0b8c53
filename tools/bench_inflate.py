#!/usr/bin/env python3
"""GB/s of inflated bytes of the GPU BGZF decoder (xm_bgzf_inflate_dev) on a tiled BAM fixture, HIP events on the launch
stream, compressed image and output resident in HBM; every block's CRC-32 (GPU kernel) against the member trailers, and a
sample of blocks byte for byte against zlib.  Beside it: zlib / libdeflate-free host inflate on one core of this box.

    python tools/bench_inflate.py --out-gb 1.0
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
DATA = os.path.join(REPO, "tests", "golden", "ref_data")


def run(out_gb=1.0, reps=5, workdir=None, ctx=None):
    """-> the result record (also bench.py's `e2e.inflate`)."""
    import types
    a = types.SimpleNamespace(out_gb=out_gb, reps=reps, dir=workdir or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"))
    import torch
    import bench_bam
    from xenomapper_amd import _ffi
    path = os.path.join(a.dir, "xm_inflate_%d.bam" % os.getpid())
    per_copy = 119_000                                    # inflated record bytes of one copy of the fixture, roughly
    copies = max(1, int(a.out_gb * 1e9 / per_copy))
    bench_bam.tiled_bam(os.path.join(DATA, "paired_end_testdata_human.bam"), path, copies)
    image = np.fromfile(path, dtype=np.uint8)
    os.unlink(path)
    blocks, crc, nxt, total = _ffi.bgzf_index(image)
    dev = torch.device("cuda:0")
    own = ctx is None
    if own:
        ctx = _ffi.Context(0)
    comp = torch.zeros(image.shape[0] + _ffi.BGZF_COMP_PAD, dtype=torch.uint8, device=dev)
    comp[:image.shape[0]] = torch.from_numpy(image).to(dev)
    d_blocks = torch.from_numpy(blocks.view(np.uint8)).to(dev)
    out = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    status = torch.zeros(len(blocks), dtype=torch.int32, device=dev)
    work = torch.zeros(1, dtype=torch.int32, device=dev)
    d_crc = torch.zeros(len(blocks), dtype=torch.int32, device=dev)
    ms, ms_crc = [], []
    for it in range(a.reps + 1):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        ctx.bgzf_inflate_dev(comp, d_blocks, out, status, work)
        e1.record()
        ctx.bgzf_crc32_dev(out, d_blocks, d_crc)
        e2.record()
        torch.cuda.synchronize()
        if it:
            ms.append(e0.elapsed_time(e1))
            ms_crc.append(e1.elapsed_time(e2))
    st = status.cpu().numpy()
    ok = bool((st == 0).all()) and bool(np.array_equal(d_crc.cpu().numpy().view(np.uint32), crc))
    # a sample of blocks byte for byte against zlib, and zlib's own rate on one core
    sample = np.linspace(0, len(blocks) - 1, 64).astype(int)
    h_out = out.cpu().numpy()
    t0 = time.perf_counter()
    cpu_bytes = 0
    for b in sample:
        o, n = int(blocks["out_off"][b]), int(blocks["isize"][b])
        c0, cl = int(blocks["cdata_off"][b]), int(blocks["cdata_len"][b])
        raw = zlib.decompress(image[c0:c0 + cl].tobytes(), -15)
        cpu_bytes += len(raw)
        ok &= len(raw) == n and raw == h_out[o:o + n].tobytes()
    cpu_s = time.perf_counter() - t0
    med = sorted(ms)[len(ms) // 2]
    rec = {"metric": "GB/s of inflated bytes (BGZF blocks decoded on the GPU)", "value": total / (med * 1e-3) / 1e9,
           "ms": med, "ms_all": [round(x, 3) for x in ms], "crc_ms": sorted(ms_crc)[len(ms_crc) // 2],
           "blocks": int(len(blocks)), "inflated_bytes": int(total), "compressed_bytes": int(image.shape[0]),
           "ratio": total / image.shape[0], "verified": ok,
           "bad_blocks": [(int(b), _ffi.bgzf_strerror(s)) for b, s in enumerate(st) if s][:4],
           "zlib_one_core_GBps": cpu_bytes / cpu_s / 1e9,
           "what": "xm_bgzf_inflate_dev on the reference's human BAM fixture tiled to ~%.1f GB of records, compressed image and output "
                   "resident in HBM, HIP events on the launch stream, median of %d; every block's CRC-32 (GPU kernel) against the member "
                   "trailers and 64 blocks byte for byte against zlib; zlib alone on one core of this box beside it" % (out_gb, reps)}
    del comp, out, d_blocks
    torch.cuda.empty_cache()
    if own:
        ctx.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-gb", type=float, default=1.0, help="inflated bytes per launch")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dir", default="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    a = ap.parse_args()
    print(json.dumps(run(a.out_gb, a.reps, a.dir)))


if __name__ == "__main__":
    main()
