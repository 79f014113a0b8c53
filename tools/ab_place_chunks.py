#!/usr/bin/env python3
"""A/B of the chunked six-list step (xm_api.hip place_steps) on ONE box, one process: bench.py's sharded-input workload at
N = 1 (configs[3]'s 400 M read pairs as one read block, xm_classify_place_dev) timed with the chunk size set to each of
XM_PLACE_CHUNK_PARTS = 0 (one pass), 16, 32, 64, 128, in rotation, three rounds.

    python tools/ab_place_chunks.py [total_pairs] > profiles/rNN_ab_place_chunks.txt
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import bench
    from xenomapper_amd import _ffi
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
    total -= total % (bench.Workload.PARTS * 32)
    dev = torch.device("cuda:0")
    ctx = _ffi.Context(0)
    wl = bench.Workload("cfg2", ctx, dev, total, 0, None, 10, 0.0, (total, 1), form="place")

    def fence():
        torch.cuda.synchronize()

    print("# %s; %d read pairs in one read block, %d records" % (wl.call_name(), total, wl.n))
    settings = ["0", "16", "32", "64", "128"]
    rows = {s: [] for s in settings}
    for rnd in range(3):
        for s in settings:
            os.environ["XM_PLACE_CHUNK_PARTS"] = s
            el, _t, tma = bench.time_steps(ctx, wl, 10, 3, fence)
            k = {name: v["ms"] / 10 for name, v in tma.items() if v["launches"]}
            n_launch = sum(v["launches"] for v in tma.values()) / 10
            rows[s].append(1e3 * el / 10)
            print("round %d  chunk parts %4s  ms/step %.4f  kernels per step: classify %.4f scan %.4f scatter %.4f  launches/step %g" % (
                rnd, s, 1e3 * el / 10, k.get("classify", 0), k.get("scan", 0), k.get("scatter", 0), n_launch), flush=True)
    os.environ["XM_PLACE_CHUNK_PARTS"] = "32"
    wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ok = wl.verify()
    print("# verified against the oracle (chunk parts 32): %s (%.0f s)" % (ok, time.perf_counter() - t0))
    for s in settings:
        v = sorted(rows[s])
        print("chunk parts %4s  ms/step min %.4f median %.4f max %.4f" % (s, v[0], v[len(v) // 2], v[-1]))


if __name__ == "__main__":
    main()
