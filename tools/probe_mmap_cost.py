"""What mmap / first touch / munmap of a window of a tmpfs file cost in a process with and without a HIP context and page-locked buffers
(the writer maps every bin's range of every window: tools/probe_cli_profile.py shows 2.5 ms per call outside the copy)."""
import mmap, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)


def probe(tag):
    p = "/dev/shm/xm_mm_probe_%d" % os.getpid()
    fd = os.open(p, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    os.posix_fallocate(fd, 0, 1 << 30)
    rows = []
    for k in range(6):
        t0 = time.perf_counter()
        mm = mmap.mmap(fd, 150 << 20, offset=k * (150 << 20), access=mmap.ACCESS_WRITE)
        t1 = time.perf_counter()
        v = np.frombuffer(mm, dtype=np.uint8)
        v[::4096] = 1
        t2 = time.perf_counter()
        del v
        mm.close()
        t3 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2))
    os.close(fd); os.unlink(p)
    m = np.median(np.array(rows), axis=0)
    print("%-44s mmap %.4f s  touch 150 MB %.4f s  munmap %.4f s" % (tag, m[0], m[1], m[2]))


probe("plain process:")
from xenomapper_amd import xenomapper as x
ctx = x.default_context()
probe("with a HIP context:")
b = x.default_bamdev()
b.reserve(0, 300 << 20, 600 << 20, 20000, 4 << 20)
b.reserve(1, 300 << 20, 600 << 20, 20000, 4 << 20)
probe("... and 3 GB of page-locked buffers:")
