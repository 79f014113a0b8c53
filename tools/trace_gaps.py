#!/usr/bin/env python3
"""What the GPU did over a rocprofv3 --kernel-trace run: busy time (union of the kernels' intervals), the share of each kernel,
how much of it overlapped another kernel, and the longest idle gaps with the kernels on either side.

    python tools/trace_gaps.py <..._kernel_trace.csv> [--from-kernel NAME] [--top 12]

--from-kernel: measure from the LAST cluster of launches of that kernel's first occurrence (skips a warm-up pass): the trace is cut
at the largest gap in front of the second half of that kernel's launches."""
import argparse
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--second-pass", action="store_true", help="only the second half of the trace by launches of inflate_kernel")
    ap.add_argument("--timeline", type=int, default=0, help="also list the first N launches of at least --min-us, with start and end")
    ap.add_argument("--min-us", type=float, default=200.0)
    a = ap.parse_args()
    rows = []
    with open(a.trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    if a.second_pass:
        infl = [k for k, r in enumerate(rows) if r[2].startswith("inflate_kernel")]
        if len(infl) >= 2:
            rows = rows[infl[len(infl) // 2]:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
    last_name = rows[0][2]
    for s, e, name in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0, last_name, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = name
    busy += cur_e - cur_s
    per = {}
    for s, e, name in rows:
        d = per.setdefault(name, [0, 0])
        d[0] += e - s
        d[1] += 1
    print("span %.1f ms, GPU busy %.1f ms (%.0f %%), idle %.1f ms in %d gaps; sum of kernel times %.1f ms" %
          ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), (t1 - t0 - busy) / 1e6, len(gaps), sum(v[0] for v in per.values()) / 1e6))
    for name, (ns, calls) in sorted(per.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print("  %-48s %5d calls %9.2f ms  %5.1f %% of busy  mean %8.1f us" % (name, calls, ns / 1e6, 100.0 * ns / busy, ns / calls / 1e3))
    if a.timeline:
        print("timeline (ms from the first launch; launches of >= %.0f us):" % a.min_us)
        shown = 0
        for s_, e_, name in rows:
            if (e_ - s_) / 1e3 >= a.min_us:
                print("  %9.2f .. %9.2f  %8.2f ms  %s" % ((s_ - t0) / 1e6, (e_ - t0) / 1e6, (e_ - s_) / 1e6, name))
                shown += 1
                if shown >= a.timeline:
                    break
    print("longest idle gaps:")
    for g, at, before, after in sorted(gaps, reverse=True)[:a.top]:
        print("  %8.2f ms at %8.1f ms  after %-32s before %s" % (g / 1e6, at / 1e6, before, after))


if __name__ == "__main__":
    main()
