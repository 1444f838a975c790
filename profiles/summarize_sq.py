#!/usr/bin/env python3
"""Summarise the three `benchmarks/profile_sq.sh` passes (rocprofv3 --kernel-trace --pmc ..., one counter group per pass) into
per-kernel-shape figures per launch.

    python profiles/summarize_sq.py gpurun_out profiles/rNN_sq_counters.md

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES
counts cycles summed over SIMDs (32 per v_mfma_f32_32x32x16_bf16); SQ_BUSY_CYCLES is per shader engine (x8 on this part are
active); GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (values / duration are ~8 x 2.4 GHz), so the kernel's length in
shader clocks is GRBM_GUI_ACTIVE / 8 and MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).
Counter collection serialises and slows the kernels (durations here are 10-25 % above the kernel-trace ones): ratios, not times.
"""
import collections
import csv
import os
import sys


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (r["Kernel_Name"], int(r["Grid_Size"]))
            a = acc[key][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                d = dur[key]
                d[0] += 1
                d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return acc, dur


def short(name):
    name = name.replace("void ", "").replace("kd::", "")
    return name.split("(")[0][:46]


def main():
    root, out = sys.argv[1], sys.argv[2]
    a, dur = load(os.path.join(root, "sq_a", "a_counter_collection.csv"))
    b, _ = load(os.path.join(root, "sq_b", "b_counter_collection.csv"))
    c, durc = load(os.path.join(root, "sq_c", "c_counter_collection.csv"))
    rows = []
    for key in a:
        if "kd::" not in key[0] and "kd::" not in key[0].replace("void ", ""):
            continue
        n, t = dur[key]
        us = t / n / 1e3
        if us < 15:
            continue
        per = lambda src, cn: (src[key][cn][1] / src[key][cn][0]) if key in src and cn in src[key] else float("nan")
        gui = per(c, "GRBM_GUI_ACTIVE") / 8.0
        usc = durc[key][1] / durc[key][0] / 1e3 if key in durc else float("nan")
        mfma_busy = per(a, "SQ_VALU_MFMA_BUSY_CYCLES")
        rows.append((t, short(key[0]), key[1], n, us, gui / usc / 1e3, mfma_busy / (1024 * gui) if gui == gui else float("nan"),
                     per(c, "SQ_INSTS_MFMA") / 1e6, per(a, "SQ_INSTS_VALU") / 1e6, per(b, "SQ_INSTS_SALU") / 1e6, per(b, "SQ_INSTS_LDS") / 1e6,
                     per(b, "SQ_WAIT_INST_ANY") / per(a, "SQ_WAVE_CYCLES"), per(b, "SQ_ACTIVE_INST_ANY") / per(a, "SQ_WAVE_CYCLES")))
    rows.sort(reverse=True)
    with open(out, "w") as f:
        f.write("| kernel | grid | launches | us (under PMC) | clock GHz | MFMA busy / (SIMDs x clocks) | MFMA insts M | VALU insts M | SALU M | LDS M | "
                "WAIT_INST_ANY / WAVE_CYCLES | ACTIVE_INST_ANY / WAVE_CYCLES |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| `{r[1]}` | {r[2]} | {r[3]} | {r[4]:.1f} | {r[5]:.2f} | {r[6]:.3f} | {r[7]:.2f} | {r[8]:.2f} | {r[9]:.2f} | {r[10]:.2f} | {r[11]:.2f} | {r[12]:.2f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
