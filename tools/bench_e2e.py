#!/usr/bin/env python3
"""End-to-end SAM -> six SAM files throughput of the file fast path (C++ stripper -> GPU -> C++ writer).

    python tools/bench_e2e.py --pairs 2000000 --threads 0

Input: a synthetic 2x150 bp paired SAM text twin (50 k pairs, seed 2002) tiled to the requested size in
/dev/shm.  Reports read-pairs/s including parsing, H2D/D2H through the host-buffer C ABI and writing all six
bins (to /dev/null).  This is the PCIe- and parser-inclusive number; bench.py's `value` is the HBM-resident one.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def run(pairs=2_000_000, threads=0, mode="liberal", workdir="/dev/shm", out_dir=None):
    """One warm-up and one timed pass; returns the result record (also used by bench.py's `e2e.sam_text`)."""
    from xenomapper_amd import _host, synth, xenomapper as xm
    base = 50_000
    t1, t2, _ = synth.sam_text_pair(n_pairs=base, seed=2002, profile="bowtie2", paired=mode != "se", read_len=150)
    reps = max(1, pairs // base)
    paths = []
    for tag, text in (("p", t1), ("s", t2)):
        head_end = 0
        while text[head_end] == "@":
            head_end = text.index("\n", head_end) + 1
        path = os.path.join(workdir, "xm_e2e_%s_%d.sam" % (tag, os.getpid()))
        with open(path, "wt") as fh:
            fh.write(text[:head_end])
            body = text[head_end:]
            for _ in range(reps):
                fh.write(body)
        paths.append(path)
    size = sum(os.path.getsize(p) for p in paths)
    names = ("primary_specific", "secondary_specific", "primary_multi", "secondary_multi", "unassigned", "unresolved")
    out_paths = [os.path.join(out_dir, "xm_e2e_out_%s_%d.sam" % (k, os.getpid())) for k in names] if out_dir else []
    sinks = {k: open(out_paths[i] if out_dir else os.devnull, "wt") for i, k in enumerate(names)}
    try:
        xm.default_context()
        for warm in (True, False):
            for sink in sinks.values():
                if out_dir:
                    sink.seek(0)
                    sink.truncate()
            t0 = time.perf_counter()
            counts = xm.classify_sam_files(paths[0], paths[1], paired=mode != "se", conservative=mode == "conservative",
                                           n_threads=threads, **sinks)
            for sink in sinks.values():
                sink.flush()
            el = time.perf_counter() - t0
        units = sum(counts.values())
        return {"metric": "end-to-end read-pairs/s (SAM text in, six SAM files out)", "value": units / el,
                "units": units, "seconds": el, "input_bytes": size, "input_GBps": size / el / 1e9,
                "threads": threads or _host.lib().xmh_default_threads(), "mode": mode,
                "outputs": "files" if out_dir else "/dev/null",
                "output_bytes": sum(os.path.getsize(p) for p in out_paths),
                # wall seconds per phase of the timed pass: window + parse run in a helper thread beside classify + emit + write
                "phases": {k: round(v, 4) for k, v in xm.LAST_FILE_PROFILE.items()}}
    finally:
        for sink in sinks.values():
            sink.close()
        for p in paths + out_paths:
            if os.path.exists(p):
                os.unlink(p)


def host_ceilings(workdir="/dev/shm", mb=512, parse_mb=64):
    """What the host side of the file path can do at best on this box (the file path's own rooflines): one-core and
    all-core memory copy rates, one write(2) stream into a tmpfs file against all threads filling the mapped file, and
    the stripper's parse rate against its thread count."""
    import mmap
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from xenomapper_amd import _host, synth
    n_thr = _host.lib().xmh_default_threads()
    src = np.ones(mb << 20, dtype=np.uint8)
    dst = np.empty_like(src)
    out = {"threads_granted": n_thr, "buffer_MB": mb}

    def best(fn, reps=3):
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        return min(t)

    def par_copy(target):
        step = (src.shape[0] + n_thr - 1) // n_thr
        with ThreadPoolExecutor(n_thr) as ex:
            list(ex.map(lambda k: np.copyto(target[k * step:(k + 1) * step], src[k * step:(k + 1) * step]), range(n_thr)))
    out["memcpy_1_thread_GBps"] = src.nbytes / best(lambda: np.copyto(dst, src)) / 1e9
    out["memcpy_all_threads_GBps"] = src.nbytes / best(lambda: par_copy(dst)) / 1e9
    path = os.path.join(workdir, "xm_ceiling_%d.bin" % os.getpid())
    try:
        def one_stream():
            with open(path, "wb", buffering=0) as fh:
                mv = memoryview(src)
                for at in range(0, src.nbytes, 64 << 20):
                    fh.write(mv[at:at + (64 << 20)])

        def allocate():
            with open(path, "w+b") as fh:
                os.posix_fallocate(fh.fileno(), 0, src.nbytes)
        out["tmpfs_one_write_stream_GBps"] = src.nbytes / best(one_stream) / 1e9
        out["tmpfs_fallocate_GBps"] = src.nbytes / best(allocate) / 1e9          # the kernel allocating (and zeroing) the pages, one thread
        # what feeds the PCIe link in the SAM text path: pread(2) of a tmpfs file by k threads (xmh_pread, as the stage loop
        # calls it) into page-locked memory when a GPU is here (else ordinary memory) -- compare e2e.pcie_ceiling.h2d_pinned_GBps
        one_stream()
        target, kind = dst, "ordinary memory"
        try:
            import torch
            if torch.cuda.is_available():
                pinned = torch.empty(src.nbytes, dtype=torch.uint8).pin_memory()
                target, kind = pinned.numpy(), "page-locked memory"
        except Exception:                                            # noqa: BLE001 -- the ordinary buffer then
            pass
        fd = os.open(path, os.O_RDONLY)
        try:
            rates = {}
            for k in sorted(set(t for t in (1, 4, 8, 16, 32, n_thr) if t <= 2 * max(n_thr, 1))):
                reader = _host.Parser(k)
                rates[str(k)] = round(src.nbytes / best(lambda: reader.pread(fd, 0, target.ctypes.data, src.nbytes)) / 1e9, 2)
                reader.close()
            out["tmpfs_pread_GBps_by_threads"] = rates
            out["tmpfs_pread_into"] = kind
        finally:
            os.close(fd)
    finally:
        if os.path.exists(path):
            os.unlink(path)
    t1, t2, _ = synth.sam_text_pair(n_pairs=20_000, seed=2002, profile="bowtie2", paired=True, read_len=150)
    bodies = []
    for text in (t1, t2):
        body = "".join(line for line in text.splitlines(True) if not line.startswith("@")).encode("ascii")
        bodies.append(np.frombuffer(body * max(1, (parse_mb << 20) // len(body)), dtype=np.uint8).copy())
    total = bodies[0].shape[0] + bodies[1].shape[0]
    scaling = {}
    for k in sorted(set(t for t in (1, 2, 4, 8, 16, n_thr) if t <= max(n_thr, 1))):
        parser = _host.Parser(k)
        el = best(lambda: parser.parse(bodies[0], 0, bodies[0].shape[0], True, bodies[1], 0, bodies[1].shape[0], True,
                                       0, True, False, True, 1 << 22))
        scaling[str(k)] = round(total / el / 1e9, 2)
        parser.close()
    out["stripper_GBps_of_text_by_threads"] = scaling
    out["stripper_input"] = "2 x %d MB of 2x150 bp SAM text" % parse_mb
    # the writer itself, as the file path uses it: the lines of ALL units of the parsed block gathered by the writer's own
    # threads straight into the pages of an extended, mapped tmpfs file (posix_fallocate + mmap + xmh_emit), against the
    # same text gathered into ordinary memory
    from xenomapper_amd import xenomapper as xm
    parser = _host.Parser(n_thr)
    try:
        blk = parser.parse(bodies[0], 0, bodies[0].shape[0], True, bodies[1], 0, bodies[1].shape[0], True, 0, True, False, True, 1 << 22)
        flags = np.unpackbits(blk.unit_bits.view(np.uint8), bitorder="little")[:blk.n].astype(bool)
        idx = np.flatnonzero(flags).astype(np.uint32)
        _, need = parser.emit_size(True, 0, idx)

        def into_file():
            with open(path, "wt") as sink:
                assert xm._emit_into_file(parser, True, 0, idx, sink)

        def into_memory():
            parser.emit(True, 0, idx, reuse=True)
        out["writer_into_mapped_tmpfs_file_GBps"] = need / best(into_file) / 1e9
        out["writer_into_memory_GBps"] = need / best(into_memory) / 1e9
        out["writer_input"] = "%d MB of lines (the file-1 lines of every unit of the stripper input)" % (need >> 20)
    finally:
        parser.close()
        if os.path.exists(path):
            os.unlink(path)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2_000_000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--mode", default="liberal", choices=("liberal", "conservative", "se"))
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--out-dir", default=None, help="write the six outputs to real files here (default: /dev/null)")
    ap.add_argument("--ceilings", action="store_true", help="print the host-side ceilings of this box instead")
    a = ap.parse_args()
    if a.ceilings:
        print(json.dumps(host_ceilings(a.dir)))
        return
    print(json.dumps(run(a.pairs, a.threads, a.mode, a.dir, a.out_dir)))


if __name__ == "__main__":
    main()
