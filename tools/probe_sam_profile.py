"""cProfile of the main thread of a SAM text -> six files run (tools/bench_e2e.py's workload): where the writer's time goes."""
import cProfile, pstats, os, sys, io
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import bench_e2e
import xenomapper_amd.xenomapper as x
args = ["--pairs", "4000000", "--out-dir", "/dev/shm"]
sys.argv = ["bench_e2e.py"] + args
real = x.classify_sam_files
state = {"n": 0}


def wrapped(*a, **k):
    state["n"] += 1
    if state["n"] < 2:                      # the warm-up pass unprofiled
        return real(*a, **k)
    pr = cProfile.Profile()
    pr.enable()
    try:
        return real(*a, **k)
    finally:
        pr.disable()
        st = pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime")
        st.print_stats(22)


x.classify_sam_files = wrapped
bench_e2e.main()
