#!/usr/bin/env python3
"""Whole-pipeline parity at scale, on the GPU box: the same pair of SAM files (the 50 k-pair twin tiled) through the file
path with the stripper on the GPU and with the host stripper -- the six outputs must be byte-identical and the counters
equal, for the paired loops, the single-end loop (skipping walk) and --cigar_scores.

    python tools/check_strip_scale.py [--pairs 1000000]
"""
import argparse
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--dir", default="/dev/shm")
    a = ap.parse_args()
    from xenomapper_amd import synth, xenomapper as xm
    base = 50_000
    names = ("primary_specific", "secondary_specific", "primary_multi", "secondary_multi", "unassigned", "unresolved")
    report = {}
    for label, paired, conservative, tag_func in (("pe_liberal", True, False, xm.get_tag), ("pe_conservative_zs", True, True, xm.get_tag_with_ZS_as_XS),
                                                  ("se_skipping", False, False, xm.get_tag), ("pe_cigar", True, False, xm.get_cigarbased_AS_tag)):
        t1, t2, _ = synth.sam_text_pair(n_pairs=base, seed=2002, profile="bowtie2", paired=paired, read_len=150)
        paths = []
        for tag, text in (("p", t1), ("s", t2)):
            head_end = 0
            while text[head_end] == "@":
                head_end = text.index("\n", head_end) + 1
            path = os.path.join(a.dir, "xm_chk_%s_%d.sam" % (tag, os.getpid()))
            with open(path, "wt") as fh:
                fh.write(text[:head_end])
                for _ in range(max(1, a.pairs // base)):
                    fh.write(text[head_end:])
            paths.append(path)
        got = {}
        try:
            for strip in ("1", "0"):
                os.environ["XENOMAPPER_GPU_STRIP"] = strip
                outs = [os.path.join(a.dir, "xm_chk_out_%s_%s_%d.sam" % (k, strip, os.getpid())) for k in names]
                sinks = {k: open(outs[i], "wt") for i, k in enumerate(names)}
                try:
                    counts = xm.classify_sam_files(paths[0], paths[1], paired=paired, conservative=conservative, tag_func=tag_func, **sinks)
                finally:
                    for s in sinks.values():
                        s.close()
                got[strip] = ([sha(p) for p in outs], [os.path.getsize(p) for p in outs], sorted((str(k), v) for k, v in counts.items()))
                for p in outs:
                    os.unlink(p)
        finally:
            for p in paths:
                os.unlink(p)
        same = got["1"] == got["0"]
        report[label] = {"identical": same, "output_bytes": sum(got["1"][1]), "units": sum(v for _, v in got["1"][2])}
        if not same:
            print(json.dumps(report))
            sys.exit(1)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
