import os, sys, time, json
REPO = "/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd()
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import bench_bam
from xenomapper_amd import xenomapper as xm, _ffi
paths = []
for tag in ("human", "mouse"):
    path = "/dev/shm/xm_cold_%s.bam" % tag
    bench_bam.tiled_bam(os.path.join(bench_bam.DATA, "paired_end_testdata_%s.bam" % tag), path, 48000)
    paths.append(path)
names = ("primary_specific", "secondary_specific", "primary_multi", "secondary_multi", "unassigned", "unresolved")
t0 = time.perf_counter()
xm.default_context()
t1 = time.perf_counter()
print("context %.3f s" % (t1 - t0))
for k in range(3):
    sinks = {k_: open(os.devnull, "wt") for k_ in names}
    t0 = time.perf_counter()
    counts = xm.classify_sam_files(paths[0], paths[1], paired=True, bam=True, **sinks)
    el = time.perf_counter() - t0
    p = {k_: round(v, 3) for k_, v in xm.LAST_FILE_PROFILE.items() if isinstance(v, float) and v > 0.005}
    print("run %d: %.3f s" % (k, el), p, "pinned", _ffi.pinned_bytes() if hasattr(_ffi, "pinned_bytes") else "")
for p in paths: os.unlink(p)
