#!/usr/bin/env python3
"""The command line as a user runs it, a process per job: python -m xenomapper_amd.xenomapper --primary_bam ... --paired
--primary_specific ... (six real output files), wall-clock time of the whole process -- interpreter start, imports, the context,
the front end's page-locked buffers, the run, the files closed.  Inputs: the reference's BAM fixtures tiled (tools/bench_bam.py)
or a 2 x 150 bp SAM text twin (tools/bench_e2e.py).     python tools/bench_cli.py [--copies N] [--sam-pairs N] [--dir /dev/shm]"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
NAMES = ("primary_specific", "secondary_specific", "primary_multi", "secondary_multi", "unassigned", "unresolved")


def run_cli(inputs, flags, workdir, tag):
    outs = [os.path.join(workdir, "xm_cli_%s_%s_%d.sam" % (tag, n, os.getpid())) for n in NAMES]
    cmd = [sys.executable, "-m", "xenomapper_amd.xenomapper"] + inputs + flags
    for n, p in zip(NAMES, outs):
        cmd += ["--" + n, p]
    try:
        t0 = time.perf_counter()
        proc = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=900)
        el = time.perf_counter() - t0
        if proc.returncode != 0:
            raise RuntimeError(proc.stderr[-2000:])
        size = sum(os.path.getsize(p) for p in outs)
        return el, size, proc.stderr
    finally:
        for p in outs:
            if os.path.exists(p):
                os.unlink(p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=48000)
    ap.add_argument("--sam-pairs", type=int, default=4000000)
    ap.add_argument("--dir", default="/dev/shm")
    a = ap.parse_args()
    import bench_bam
    out = {}
    paths = []
    try:
        for tag in ("human", "mouse"):
            p = os.path.join(a.dir, "xm_cli_%s_%d.bam" % (tag, os.getpid()))
            bench_bam.tiled_bam(os.path.join(bench_bam.DATA, "paired_end_testdata_%s.bam" % tag), p, a.copies)
            paths.append(p)
        pairs = a.copies * 238
        for rep in ("first", "second"):
            el, size, _err = run_cli(["--primary_bam", paths[0], "--secondary_bam", paths[1]], ["--paired"], a.dir, "bam")
            out["bam_" + rep] = {"seconds": round(el, 3), "read_pairs_per_s": round(pairs / el), "output_bytes": size}
        out["bam_input_bytes"] = sum(os.path.getsize(p) for p in paths)
        out["bam_pairs"] = pairs
    finally:
        for p in paths:
            if os.path.exists(p):
                os.unlink(p)
    t0 = time.perf_counter()
    proc = subprocess.run([sys.executable, "-c", "import xenomapper_amd.xenomapper as x; x.default_context()"], cwd=REPO, capture_output=True, text=True)
    out["interpreter_imports_context_seconds"] = round(time.perf_counter() - t0, 3)
    out["what"] = ("wall-clock time of `python -m xenomapper_amd.xenomapper --primary_bam a --secondary_bam b --paired --<six outputs> ...` as a child "
                   "process, six output files on " + a.dir + "; first and second process on the same inputs (page cache warm either time)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
