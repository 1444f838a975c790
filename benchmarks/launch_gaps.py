#!/usr/bin/env python3
"""Gaps between dependent launches of a forward, from a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp of every dispatch):
gap(n) = start(n + 1) - end(n) on the main stream's queue, grouped by the kernel that FOLLOWS the gap and by the one in front of it.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --mode bf16 --steps 1 --warmup 1 ...
    python benchmarks/launch_gaps.py DIR/**/t_kernel_trace.csv
"""
import csv
import glob
import re
import statistics
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name.replace("kd::", "")[:60]


def main(paths):
    rows = []
    for p in paths:
        for f in glob.glob(p, recursive=True):
            with open(f) as fh:
                rows += list(csv.DictReader(fh))
    if not rows:
        raise SystemExit("no rows")
    by_queue = defaultdict(list)
    for r in rows:
        by_queue[(r.get("Queue_Id"), r.get("Stream_Id", ""))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    q = max(by_queue.values(), key=len)          # the main chain's queue
    q.sort()
    gaps_after, gaps_before, durs = defaultdict(list), defaultdict(list), defaultdict(list)
    allg = []
    for (s0, e0, n0), (s1, e1, n1) in zip(q, q[1:]):
        g = (s1 - e0) / 1e3
        durs[short(n0)].append((e0 - s0) / 1e3)
        if -5 < g < 50:                           # (host-side pauses between passes are not launch gaps)
            gaps_after[short(n0)].append(g)
            gaps_before[short(n1)].append(g)
            allg.append(g)
    print(f"{len(q)} dispatches on the main queue; gap between consecutive dispatches: median {statistics.median(allg):.2f} us, mean {statistics.mean(allg):.2f} us, "
          f"sum {sum(allg) / 1e3:.2f} ms of {(q[-1][1] - q[0][0]) / 1e6:.2f} ms")
    print(f"{'kernel':62s} {'n':>6s} {'dur us':>8s} {'gap BEHIND it':>14s} {'gap IN FRONT':>13s}")
    for k in sorted(durs, key=lambda k: -sum(durs[k])):
        a, b = gaps_after.get(k, [0]), gaps_before.get(k, [0])
        print(f"{k:62s} {len(durs[k]):6d} {statistics.mean(durs[k]):8.2f} {statistics.median(a):14.2f} {statistics.median(b):13.2f}")


if __name__ == "__main__":
    main(sys.argv[1:])
