import os, sys, time, subprocess, json
REPO = os.getcwd()
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import bench_bam
paths = []
for tag in ("human", "mouse"):
    p = "/dev/shm/xm_clip_%s.bam" % tag
    bench_bam.tiled_bam(os.path.join(bench_bam.DATA, "paired_end_testdata_%s.bam" % tag), p, 48000)
    paths.append(p)
names = ("primary_specific", "secondary_specific", "primary_multi", "secondary_multi", "unassigned", "unresolved")
prog = r'''
import time, sys, os
t0 = time.perf_counter()
import xenomapper_amd.xenomapper as x
t1 = time.perf_counter()
argv = ["--primary_bam", sys.argv[1], "--secondary_bam", sys.argv[2], "--paired"]
for n in %r:
    argv += ["--" + n, "/dev/shm/xm_clip_out_" + n + ".sam"]
x.main(argv)
t2 = time.perf_counter()
print("import %%.3f main %%.3f" %% (t1 - t0, t2 - t1), {k: round(v, 3) for k, v in x.LAST_FILE_PROFILE.items() if isinstance(v, float) and abs(v) > 0.01}, file=sys.stderr)
sys.stderr.flush()
t3 = time.perf_counter()
x.release_buffers()
print("release %%.3f" %% (time.perf_counter() - t3), file=sys.stderr)
''' % (names,)
t0 = time.perf_counter()
p = subprocess.run([sys.executable, "-c", prog] + paths, cwd=REPO, capture_output=True, text=True)
el = time.perf_counter() - t0
print("process %.3f s" % el)
print(p.stderr[-1500:])
for n in names:
    f = "/dev/shm/xm_clip_out_" + n + ".sam"
    if os.path.exists(f): os.unlink(f)
for q in paths: os.unlink(q)
