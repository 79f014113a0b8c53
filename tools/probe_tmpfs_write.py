#!/usr/bin/env python3
"""How fast six output files on tmpfs take 3.4 GB that already sits in memory as six contiguous ranges (what the file path has in hand
once the outputs are gathered on the device): positional writes from several threads per file against posix_fallocate + mmap +
threads copying into the mapping (what _emit_into_file does).  python tools/probe_tmpfs_write.py [GB]"""
import ctypes
import mmap
import os
import sys
import threading
import time

import numpy as np

N = int(float(sys.argv[1]) * 1e9) if len(sys.argv) > 1 else 3_400_000_000
frac = [0.33, 0.33, 0.08, 0.08, 0.13, 0.05]
src = np.random.randint(0, 255, N, dtype=np.uint8)
offs = np.cumsum([0] + [int(N * f) for f in frac])
paths = ["/dev/shm/xm_wt_%d_%d" % (os.getpid(), k) for k in range(6)]
libc = ctypes.CDLL(None, use_errno=True)


def clean():
    for p in paths:
        if os.path.exists(p):
            os.unlink(p)


def run_threads(jobs):
    ts = [threading.Thread(target=fn, args=args) for fn, args in jobs]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return time.perf_counter() - t0


def pwrite_split(parts):
    fds = [os.open(p, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644) for p in paths]
    jobs = []

    def w(k, a, n):
        mv = memoryview(src[offs[k] + a: offs[k] + a + n])
        done = 0
        while done < n:
            done += os.pwrite(fds[k], mv[done:], a + done)
    for k in range(6):
        n = int(offs[k + 1] - offs[k])
        step = (n + parts - 1) // parts
        for a in range(0, n, step):
            jobs.append((w, (k, a, min(step, n - a))))
    el = run_threads(jobs)
    for fd in fds:
        os.close(fd)
    return el


def fallocate_mmap(threads):
    fds = [os.open(p, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644) for p in paths]
    t0 = time.perf_counter()
    for k in range(6):                                                # as the product: bin after bin
        n = int(offs[k + 1] - offs[k])
        libc.posix_fallocate(fds[k], ctypes.c_long(0), ctypes.c_long(n))
        mm = mmap.mmap(fds[k], n, access=mmap.ACCESS_WRITE)
        dst = np.frombuffer(mm, dtype=np.uint8)
        step = (n + threads - 1) // threads

        def cp(a, m, dst=dst, k=k):
            dst[a:a + m] = src[offs[k] + a: offs[k] + a + m]
        run_threads([(cp, (a, min(step, n - a))) for a in range(0, n, step)])
        del dst, cp
        try:
            mm.close()
        except BufferError:
            pass
    el = time.perf_counter() - t0
    for fd in fds:
        os.close(fd)
    return el


for name, fn in [("pwrite, %d per file" % p, (lambda p=p: pwrite_split(p))) for p in (1, 2, 4, 8, 16)] + \
        [("fallocate + mmap, %d copiers" % t, (lambda t=t: fallocate_mmap(t))) for t in (8, 16)]:
    for rep in range(2):
        clean()
        el = fn()
        print("%-32s %.3f s  %5.1f GB/s" % (name, el, N / el / 1e9), flush=True)
clean()
