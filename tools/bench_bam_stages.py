"""Stage rates of the BAM input path on this host: BGZF/BAM -> SAM text (xmh_bam_read), then the stripper on that text.
    python tools/bench_bam_stages.py [copies] [threads]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np
import bench_bam
from xenomapper_amd import _host
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
paths = []
for tag in ("human", "mouse"):
    p = "/dev/shm/xm_stage_%s.bam" % tag
    bench_bam.tiled_bam(os.path.join(bench_bam.DATA, "paired_end_testdata_%s.bam" % tag), p, copies)
    paths.append(p)
texts = []
for p in paths:
    data = np.fromfile(p, dtype=np.uint8)
    r = _host.BamReader(data, thr)
    out = np.zeros(max(1 << 30, 400 * copies * 476), dtype=np.uint8)          # prefaulted; ~330 B of text per record
    t0 = time.perf_counter(); at = 0
    while not r.eof:
        got = r.read_into(out, at)
        if got == 0:
            break
        at += got
    el = time.perf_counter() - t0
    print("decode %s: bam %.3f GB -> text %.3f GB in %.3f s = %.2f GB/s text, %.2f GB/s bam" % (p, data.shape[0]/1e9, at/1e9, el, at/1e9/el, data.shape[0]/1e9/el))
    texts.append(out[:at].copy()); r.close(); del out
P = _host.Parser(thr)
t0 = time.perf_counter()
pos = [0, 0]; n = 0
while True:
    b = P.parse(texts[0], pos[0], texts[0].shape[0]-pos[0], True, texts[1], pos[1], texts[1].shape[0]-pos[1], True, 0, True, False, False, 1 << 22)
    n += b.n
    pos[0] += b.consumed[0]; pos[1] += b.consumed[1]
    if b.ended or b.n == 0: break
el = time.perf_counter() - t0
print("parse: %d records, %.3f s = %.2f GB/s text (both files), %.2f M records/s" % (n, el, (texts[0].shape[0]+texts[1].shape[0])/1e9/el, n/1e6/el))
for p in paths: os.unlink(p)
