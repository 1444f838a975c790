#!/usr/bin/env python3
"""Per-POSITION kernel durations of a forward from rocprofv3 --kernel-trace CSVs (benchmarks/ab_trace.sh).

A forward of the bf16 path is a fixed chain of launches that starts with patchin4_kernel; launch j of every forward is the
same kernel with the same shapes, so averaging by position separates "the first layer of a level" from the following ones
(profiles/r02_level_entry.md).  Several traces (A/B passes of one gpurun call) are printed side by side.

    python profiles/summarize_trace_positions.py gpurun_out/ab_0/t_kernel_trace.csv gpurun_out/ab_1/t_kernel_trace.csv ...
        [--first 60 --last 95]   forwards averaged (the first ones include warm-up and weight packing)
"""
import argparse
import collections
import csv


def positions(path, first, last, anchor="patchin4"):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    if len(starts) < 2:
        raise SystemExit(f"{path}: no forwards found (anchor kernel {anchor!r})")
    pos, names = collections.defaultdict(list), {}
    for f in range(min(first, len(starts) - 2), min(last, len(starts) - 1)):
        queue = rows[starts[f]]["Queue_Id"]
        chain = [r for r in rows[starts[f]:starts[f + 1]] if r["Queue_Id"] == queue]       # main stream only
        for j, r in enumerate(chain):
            pos[j].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
            names[j] = r["Kernel_Name"].replace("void ", "").replace("kd::b16::", "").replace("kd::", "").split("(")[0]
    return {j: sum(v) / len(v) for j, v in pos.items()}, names, len(starts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("traces", nargs="+")
    ap.add_argument("--first", type=int, default=60)
    ap.add_argument("--last", type=int, default=95)
    a = ap.parse_args()
    res = [positions(p, a.first, a.last) for p in a.traces]
    names = res[0][1]
    print("| position | kernel | " + " | ".join(f"trace {i} (µs)" for i in range(len(res))) + " |")
    print("|---|---|" + "---|" * len(res))
    for j in sorted(res[0][0]):
        print(f"| {j} | `{names[j]}` | " + " | ".join(f"{r[0].get(j, float('nan')):.1f}" for r in res) + " |")
    print("| | **sum per forward** | " + " | ".join(f"**{sum(r[0].values()):.0f}**" for r in res) + " |")
    print("forwards in the traces:", ", ".join(str(r[2]) for r in res))


if __name__ == "__main__":
    main()
