"""How the BAM path's rate depends on the streams the process made BEFORE the front end made its own: the HIP runtime deals streams
onto a few hardware queues (GPU_MAX_HW_QUEUES, four by default), and two of the library's streams that work side by side can land
on one.   python tools/probe_queue_collision.py N   -> N streams made (and used once) first, then tools/bench_bam.py's run."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import bench_bam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
keep = []
for _ in range(n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        keep.append(torch.zeros(16, device="cuda") + 1)
    keep.append(s)
torch.cuda.synchronize()
r = bench_bam.run(copies=48000)
p = r["phases"]
print("%d streams made first: %.2f M pairs/s  %.3f s | strip %.3f wait_raw %.3f" % (n, r["value"] / 1e6, r["seconds"], p.get("strip", 0), p.get("bam_wait_raw", 0)))
