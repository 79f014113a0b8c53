"""cProfile of the HELPER thread (parse_next: staging, the device front end's calls, the fused pass, fetch_bins) of a BAM -> /dev/null
run: the Python around the C calls of the thread the GPU waits for.   python tools/probe_helper_profile.py [sam]"""
import cProfile, pstats, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
from concurrent.futures import ThreadPoolExecutor
import xenomapper_amd.xenomapper as x
prof = cProfile.Profile()
real_submit = ThreadPoolExecutor.submit
armed = {"on": False}


def submit(self, fn, *a, **k):
    if armed["on"] and getattr(fn, "__name__", "") == "parse_next":
        def run(*aa, **kk):
            prof.enable()
            try:
                return fn(*aa, **kk)
            finally:
                prof.disable()
        return real_submit(self, run, *a, **k)
    return real_submit(self, fn, *a, **k)


ThreadPoolExecutor.submit = submit
if len(sys.argv) > 1 and sys.argv[1] == "sam":
    import bench_e2e
    r = bench_e2e.run(pairs=4000000)            # warm
    armed["on"] = True
    r = bench_e2e.run(pairs=4000000)
else:
    import bench_bam
    r = bench_bam.run(copies=48000)             # (its own warm-up pass is profiled too: halve the numbers)
    armed["on"] = True
    r = bench_bam.run(copies=48000)
print("%.2f M pairs/s" % (r["value"] / 1e6), file=sys.stderr)
pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(24)
