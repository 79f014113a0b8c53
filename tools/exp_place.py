"""On the GPU box: time the single-pass placing kernel (and the three-launch step beside it) on one bench workload with
the library named by XENOMAPPER_HIP_LIB, and print the poll statistics a -DXM_PLACE_STATS build collects.
    python tools/exp_place.py [workload] [pairs] [--no-verify]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfg2"
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 50_000_000
    import torch
    import bench
    from xenomapper_amd import _ffi
    dev = torch.device("cuda:0")
    ctx = _ffi.Context(0)
    out = []
    for place in (("0", "1") if "--place-only" not in sys.argv else ("1",)):
        os.environ["XM_BENCH_PLACE"] = place
        wl = bench.Workload(name, ctx, dev, pairs, 0)
        for _ in range(5):
            wl.step()
        torch.cuda.synchronize()
        if place == "1":
            ctx.place_debug_stats(reset=True)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                wl.step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 10 * 1e3)
        line = "%-12s %s pairs=%d  ms_per_step min %.4f med %.4f" % ("place" if place == "1" else "three-launch", name, pairs, min(ts), sorted(ts)[2])
        if place == "1":
            st = ctx.place_debug_stats()
            n_wg = (wl.n + 2047) // 2048 * 50
            line += "  polls/wg %.2f max %d picks %s gave_up %d" % (st[4] / max(n_wg, 1), st[5], st[8:16].tolist(), st[2])
        if "--no-verify" not in sys.argv:
            line += "  ok=%s" % wl.verify()
        print(line, flush=True)
        del wl
        torch.cuda.empty_cache()


main()
