#!/usr/bin/env python3
"""Condense rocprofv3 CSV output to the xm:: kernels (the torch generator kernels are noise).

    python tools/summarize_prof.py --stats gpurun_out/prof_stats/r01_kernel_stats.csv \
        --fetch gpurun_out/prof_fetch/r01_counter_collection.csv \
        --write gpurun_out/prof_write/r01_counter_collection.csv --pairs 50000000 --out profiles/r01

Writes <out>_kernel_stats.csv (rows of the --stats summary for xm:: kernels), and
<out>_pmc.json with per-launch HBM traffic of each xm kernel: FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes for wide coalesced streams, so the read
side is doubled (guides/MI355X_MICROARCH.md, section HBM).
"""
import argparse
import csv
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"xm::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--pairs", type=int, default=50_000_000)
    ap.add_argument("--out", required=True)
    ap.add_argument("--traffic-json", help="profiles/pmc_traffic.json: add/replace this workload's entry (read by bench.py), "
                                           "stamped with the hash of the kernel sources in the tree")
    ap.add_argument("--workload", help="bench.py workload name of this profile (cfg2, cfg3, cfg5 ...)")
    a = ap.parse_args()
    if a.stats:
        rows = list(csv.DictReader(open(a.stats)))
        keep = [r for r in rows if short(r["Name"])]
        with open(a.out + "_kernel_stats.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
            for r in keep:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"],
                            r["MaxNs"], r["StdDev"]])
        for r in keep:
            print("%-40s calls %4s avg %10.1f us" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3))
    pmc = {}
    for label, path in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
        if not path:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == label:
                acc[k].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            pmc.setdefault(k, {})[label + "_KiB_per_launch"] = sum(v) / len(v)
            pmc[k]["launches_" + label] = len(v)
    if pmc:
        for k, d in pmc.items():
            rd = d.get("FETCH_SIZE_KiB_per_launch")
            wr = d.get("WRITE_SIZE_KiB_per_launch")
            if rd is not None:
                d["read_bytes_per_launch_corrected"] = 2.0 * rd * 1024        # gfx950: FETCH_SIZE reads 1/2
            if wr is not None:
                d["write_bytes_per_launch"] = wr * 1024
            if rd is not None and wr is not None:
                d["hbm_bytes_per_launch"] = d["read_bytes_per_launch_corrected"] + d["write_bytes_per_launch"]
                d["hbm_bytes_per_pair"] = d["hbm_bytes_per_launch"] / a.pairs
        json.dump(pmc, open(a.out + "_pmc.json", "w"), indent=1, sort_keys=True)
        if a.traffic_json and a.workload:
            import os
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            import kernel_hash
            sha = kernel_hash.kernel_src_sha256()
            try:
                rec = json.load(open(a.traffic_json))
            except (OSError, ValueError):
                rec = {}
            if rec.get("kernel_src_sha256") != sha:             # entries measured on other sources are void
                rec = {"kernel_src_sha256": sha, "workloads": {}}
            rec["source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 bench.py --workload W`; "
                             "FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md section HBM; kernel_src_sha256 = "
                             "tools/kernel_hash.py over xm_kernels.hip, xm_kernels.h, xm_api.hip, xenomapper_hip.h (+ extra hipcc flags)")
            per = {k: d["hbm_bytes_per_launch"] for k, d in pmc.items()
                   if "hbm_bytes_per_launch" in d and not k.startswith("stream_probe")}    # the ceiling probe is no part of a step
            cls = [v for k, v in per.items() if k.startswith("classify")]
            if cls:
                rec["workloads"][a.workload] = {"pairs": a.pairs, "classify_hbm_bytes_per_launch": max(cls),
                                                "step_hbm_bytes": sum(per.values()), "kernels": per,
                                                "profile": "profiles/" + os.path.basename(a.out) + "_pmc.json"}
                json.dump(rec, open(a.traffic_json, "w"), indent=1, sort_keys=True)
        print(json.dumps(pmc, indent=1, sort_keys=True))


if __name__ == "__main__":
    sys.exit(main())
