"""On the GPU box, with a -DXM_PLACE_TRACE build (XENOMAPPER_HIP_LIB): per-workgroup timeline of the placing kernel."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import numpy as np
    import torch
    import bench
    from xenomapper_amd import _ffi
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    os.environ["XM_BENCH_PLACE"] = "1"
    ctx = _ffi.Context(0)
    wl = bench.Workload("cfg2", ctx, torch.device("cuda:0"), pairs, 0)
    ng = (wl.n + 2047) // 2048
    lag = min(ng, int(os.environ.get("XM_PLACE_LAG", "2560")))
    ng += lag                                        # workgroups of the launch
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    ctx.place_debug_trace(ng)
    wl.step()
    torch.cuda.synchronize()
    t = ctx.place_debug_trace(ng, fetch=True).astype(np.int64)
    t0 = t[:, 0].min()
    us = lambda a: (a - t0) / 100.0                                        # noqa: E731
    start, loaded, pub, res, end = (us(t[:, k]) for k in range(5))
    print("kernel span %.1f us; granules %d" % (us(t[:, 4]).max(), ng))
    start, pub, res, end = start[lag:-lag], pub[lag:-lag], res[lag:-lag], end[lag:-lag]      # workgroups that place AND classify
    for name, a in (("start -> look-back begins", pub - start),
                    ("look-back", res - pub), ("look-back done -> end", end - res), ("lifetime", end - start)):
        print("%-32s mean %7.2f  p50 %7.2f  p90 %7.2f  p99 %7.2f  max %7.2f" % (name, a.mean(), np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99), a.max()))
    print("polls beyond first: mean %.2f" % t[:, 5].mean(), " picks", np.bincount(np.clip(t[:, 6], -1, 6) + 1).tolist())
    # residency: how many workgroups are alive at sample times
    for q in (0.1, 0.3, 0.5, 0.7, 0.9):
        ts = q * end.max()
        print("t=%7.1f us alive %d (before look-back %d, in look-back %d)  lowest unresolved %d highest started %d" % (
            ts, int(((start <= ts) & (end > ts)).sum()), int(((start <= ts) & (pub > ts)).sum()), int(((pub <= ts) & (res > ts)).sum()),
            int(np.argmax(res > ts)), int(np.nonzero(start <= ts)[0].max())))
    k = len(start) // 2
    print("workgroup start  lookback resolved end   polls pick   (around the middle)")
    for g in range(k, k + 24):
        print("%7d %7.2f %7.2f %7.2f %7.2f %3d %3d" % (g + lag, start[g], pub[g], res[g], end[g], t[g + lag, 5], t[g + lag, 6]))
    # dispatch order: is start monotone in g?
    inv = int((np.diff(start) < -0.5).sum())
    print("start-time inversions > 0.5 us between consecutive granules: %d" % inv)


main()
