#!/usr/bin/env python3
"""How fast the CPU reads page-locked (hipHostMalloc) memory against ordinary memory -- the GPU BAM path prints SAM text from the
inflated windows where the D2H copy left them.  One thread and 16 threads, 512 MB."""
import time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch

n = 512 << 20
plain = np.ones(n, dtype=np.uint8)
pinned_t = torch.empty(n, dtype=torch.uint8).pin_memory()
pinned = pinned_t.numpy()
pinned[:] = 1
dst = np.empty(n, dtype=np.uint8)


def best(fn, reps=3):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t)


def par(src, k=16):
    step = n // k
    with ThreadPoolExecutor(k) as ex:
        list(ex.map(lambda i: np.copyto(dst[i * step:(i + 1) * step], src[i * step:(i + 1) * step]), range(k)))


def par_sum(src, k=16):
    step = n // k
    with ThreadPoolExecutor(k) as ex:
        list(ex.map(lambda i: int(src[i * step:(i + 1) * step].view(np.uint64).sum()), range(k)))


for name, src in (("ordinary", plain), ("page-locked", pinned)):
    print("%-12s copy out of it: 1 thread %.1f GB/s, 16 threads %.1f GB/s; read only (sum): 16 threads %.1f GB/s" % (
        name, n / best(lambda: np.copyto(dst, src)) / 1e9, n / best(lambda: par(src)) / 1e9, n / best(lambda: par_sum(src)) / 1e9))
