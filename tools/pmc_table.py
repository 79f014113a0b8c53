#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters for the xm:: kernels.
    python tools/pmc_table.py gpurun_out/prof_sq/x_counter_collection.csv"""
import csv, re, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        m = re.search(r"xm::(\w+)(<[^>]*>)?", r["Kernel_Name"])
        if m:
            acc[m.group(1) + (m.group(2) or "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
