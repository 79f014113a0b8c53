"""On the GPU box, tuning build with -DXM_ONEPASS=1 -DXM_OP_TRACE (XENOMAPPER_HIP_LIB): per-workgroup timeline of the
single-kernel experiment (xm_onepass.inc)."""
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
    os.environ["XM_BENCH_PLACE"] = "2"
    ctx = _ffi.Context(0)
    wl = bench.Workload("cfg2", ctx, torch.device("cuda:0"), pairs, 0)
    ng = (wl.n + 2047) // 2048
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    ctx.onepass_debug(ng)
    wl.step()
    torch.cuda.synchronize()
    ctl, t = ctx.onepass_debug(ng, fetch=True)
    t = t.astype(np.int64)
    t0 = t[:, 0].min()
    us = lambda a: (a - t0) / 100.0                                        # noqa: E731
    start, loaded, pub, res, end = (us(t[:, k]) for k in range(5))
    print("kernel span %.1f us; granules %d; gave up: %d" % (end.max(), ng, ctl[2]))
    for name, a in (("load (start -> states)", loaded - start), ("ranks, sort, publish (-> look-back)", pub - loaded),
                    ("look-back", res - pub), ("copy-out", end - res), ("lifetime of wave 0", end - start),
                    ("eight-wave phase", pub - start)):
        print("%-36s mean %7.2f  p50 %7.2f  p90 %7.2f  p99 %7.2f  max %7.2f" % (name, a.mean(), np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99), a.max()))
    print("polls beyond the first: mean %.2f max %d   picks %s" % (t[:, 5].mean(), t[:, 5].max(), np.bincount(np.clip(t[:, 6], -1, 12) + 1).tolist()))
    for q in (0.1, 0.3, 0.5, 0.7, 0.9):
        ts = q * end.max()
        print("t=%7.1f us  eight-wave phase %4d  waiting %5d  copying %3d   lowest unresolved %d highest started %d" % (
            ts, int(((start <= ts) & (pub > ts)).sum()), int(((pub <= ts) & (res > ts)).sum()), int(((res <= ts) & (end > ts)).sum()),
            int(np.argmax(res > ts)), int(np.nonzero(start <= ts)[0].max())))
    k = ng // 2
    print("granule  start   states  lookback resolved end   polls pick")
    for g in range(k, k + 20):
        print("%7d %7.2f %7.2f %7.2f %7.2f %7.2f %3d %3d" % (g, start[g], loaded[g], pub[g], res[g], end[g], t[g, 5], t[g, 6]))
    print("ok=%s" % wl.verify())


main()
