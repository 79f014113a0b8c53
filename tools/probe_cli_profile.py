"""cProfile of xenomapper.main() on the tiled BAM fixtures (six files on /dev/shm): where the time outside the file path goes."""
import cProfile, pstats, os, sys, io
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import bench_bam
import xenomapper_amd.xenomapper as x
paths = []
for tag in ("human", "mouse"):
    p = "/dev/shm/xm_clipp_%s.bam" % tag
    bench_bam.tiled_bam(os.path.join(bench_bam.DATA, "paired_end_testdata_%s.bam" % tag), p, 48000)
    paths.append(p)
names = ("primary_specific", "secondary_specific", "primary_multi", "secondary_multi", "unassigned", "unresolved")
argv = ["--primary_bam", paths[0], "--secondary_bam", paths[1], "--paired"]
for n in names:
    argv += ["--" + n, "/dev/shm/xm_clipp_out_" + n + ".sam"]
pr = cProfile.Profile()
real_stderr = sys.stderr
sys.stderr = io.StringIO()
pr.enable()
x.main(argv)
pr.disable()
sys.stderr = real_stderr
st = pstats.Stats(pr, stream=sys.stdout).sort_stats("cumulative")
st.print_stats(28)
st.print_callees("_emit_into_file")
for n in names:
    f = "/dev/shm/xm_clipp_out_" + n + ".sam"
    if os.path.exists(f): os.unlink(f)
for q in paths: os.unlink(q)
