#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-buffer C ABI: xm_classify_compact on NumPy arrays in host memory (H2D of the
four score columns and the unit mask, the fused pass, D2H of the index lists and the counts; --code adds the category
bytes, --two-calls is round 1's xm_classify + xm_compact).  No SAM parsing, no output text -- that is tools/bench_e2e.py.

    python tools/bench_hostabi.py --pairs 25000000
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=25_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--two-calls", action="store_true", help="xm_classify + xm_compact instead of the fused xm_classify_compact")
    ap.add_argument("--code", action="store_true", help="also bring the category bytes back (1 B per record)")
    a = ap.parse_args()
    from xenomapper_amd import _ffi, synth
    n = 2 * a.pairs
    cols = synth.score_columns(n_pairs=a.pairs, seed=2002, profile="bowtie2")
    as1, xs1, as2, xs2, bits = cols["as1"], cols["xs1"], cols["as2"], cols["xs2"], cols["unit_bits"]
    ctx = _ffi.Context(int(os.environ.get("XENOMAPPER_DEVICE", "0")))
    best = None
    for _ in range(a.reps):
        t0 = time.perf_counter()
        if a.two_calls:      # round 1's shape: category bytes down and up again between the two calls
            code, counts = ctx.classify(_ffi.MODE_PE_LIBERAL, as1, xs1, as2, xs2, bits, _ffi.ABSENT)
            idx, off, _ = ctx.compact(_ffi.MODE_PE_LIBERAL, code)
        else:
            code, idx, off, counts = ctx.classify_compact(_ffi.MODE_PE_LIBERAL, as1, xs1, as2, xs2, bits, _ffi.ABSENT,
                                                          want_code=a.code)
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    units = int(off[7])
    moved = 16 * n + n // 8 + 4 * units + (3 * n if a.two_calls else (n if a.code else 0))
    print(json.dumps({"metric": "read-pairs/s through the host-buffer C ABI (H2D + kernels + D2H)", "value": units / best,
                      "pairs": a.pairs, "records": n, "seconds": best, "bytes_over_pcie": moved,
                      "pcie_GBps": moved / best / 1e9}))


if __name__ == "__main__":
    main()
