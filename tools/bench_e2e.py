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
                "output_bytes": sum(os.path.getsize(p) for p in out_paths)}
    finally:
        for sink in sinks.values():
            sink.close()
        for p in paths + out_paths:
            if os.path.exists(p):
                os.unlink(p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2_000_000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--mode", default="liberal", choices=("liberal", "conservative", "se"))
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--out-dir", default=None, help="write the six outputs to real files here (default: /dev/null)")
    a = ap.parse_args()
    print(json.dumps(run(a.pairs, a.threads, a.mode, a.dir, a.out_dir)))


if __name__ == "__main__":
    main()
