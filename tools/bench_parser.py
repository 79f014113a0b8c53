#!/usr/bin/env python3
"""Host-only throughput of the C++ stripper and writer (no GPU): xmh_parse + xmh_emit on a synthetic 2x150 bp SAM twin.

    XMH_PROFILE=1 python tools/bench_parser.py --mb 128 --threads 0 --reps 5

Every unit is emitted to bin 0 (file 1's lines), which is the writer's worst case for one bin.
"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=128, help="window bytes per file")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--score-mode", type=int, default=0, help="0 AS/XS, 1 AS/ZS, 2 NM + CIGAR")
    ap.add_argument("--single-end", action="store_true", help="single-end input walked with skip_repeated_reads")
    a = ap.parse_args()
    from xenomapper_amd import _host, synth
    t1, t2, _ = synth.sam_text_pair(n_pairs=20_000, seed=2002, profile="bowtie2", paired=not a.single_end, read_len=150)
    bodies = []
    for text in (t1, t2):
        body = "".join(l for l in text.splitlines(True) if not l.startswith("@")).encode("ascii")
        reps = max(1, (a.mb << 20) // len(body))
        bodies.append(np.frombuffer(body * reps, dtype=np.uint8).copy())
    parser = _host.Parser(a.threads)
    total = bodies[0].shape[0] + bodies[1].shape[0]
    for rep in range(a.reps):
        t0 = time.perf_counter()
        blk = parser.parse(bodies[0], 0, bodies[0].shape[0], True, bodies[1], 0, bodies[1].shape[0], True,
                           a.score_mode, not a.single_end, a.single_end, not a.single_end, 1 << 22)
        t1_ = time.perf_counter()
        flags = np.unpackbits(blk.unit_bits.view(np.uint8), bitorder="little")[:blk.n]
        idx = np.flatnonzero(flags).astype(np.uint32)
        t2_ = time.perf_counter()
        out = parser.emit(not a.single_end, 0, idx)
        t3 = time.perf_counter()
        print("rep %d: n=%d  parse %.1f ms (%.2f GB/s)  emit %.1f ms (%.2f GB/s of output, %d MB)" % (
            rep, blk.n, (t1_ - t0) * 1e3, total / (t1_ - t0) / 1e9, (t3 - t2_) * 1e3, len(out) / (t3 - t2_) / 1e9,
            len(out) >> 20))


if __name__ == "__main__":
    main()
