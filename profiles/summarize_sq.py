#!/usr/bin/env python3
"""Summarise the three `benchmarks/profile_sq.sh` passes (rocprofv3 --kernel-trace --pmc ..., one counter group per pass) into
per-kernel-shape figures per launch.

    python profiles/summarize_sq.py gpurun_out profiles/rNN_sq_counters.md

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles summed over waves (their RATIOS are what is
read here: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES); SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's
1024 SIMDs (32 per v_mfma_f32_32x32x16_bf16).  MFMA-pipe utilisation is therefore MFMA_BUSY / (1024 x kernel clocks); the kernel's
length in clocks is taken as (End - Start timestamp) x f, and since the shader clock under load is not known per dispatch the
column is given for f = 2.4 GHz (the part's maximum: a LOWER bound of the utilisation) -- round 1's "clock GHz" column derived
from GRBM_GUI_ACTIVE read above 2.4 GHz and was wrong (that counter is not one tick per shader clock per XCD here); it is gone.
Counter collection serialises and slows the kernels (durations 10-25 % above the kernel-trace ones): ratios, not times.
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
    name = name.replace("void ", "").replace("kd::", "").replace("b16::", "").replace("x3t::", "").replace("x3a::", "").replace("x3::", "")
    return name.split("(")[0][:44]


def main():
    root, out = sys.argv[1], sys.argv[2]
    a, dur = load(os.path.join(root, "sq_a", "a_counter_collection.csv"))
    b, _ = load(os.path.join(root, "sq_b", "b_counter_collection.csv"))
    c, durc = load(os.path.join(root, "sq_c", "c_counter_collection.csv"))
    rows = []
    for key in a:
        if "kd::" not in key[0]:
            continue
        n, t = dur[key]
        us = t / n / 1e3
        if us < 12:
            continue
        per = lambda src, cn: (src[key][cn][1] / src[key][cn][0]) if key in src and cn in src[key] else float("nan")
        wave = per(a, "SQ_WAVE_CYCLES")
        mfma_busy = per(a, "SQ_VALU_MFMA_BUSY_CYCLES")
        n_mfma, n_valu, n_lds = per(c, "SQ_INSTS_MFMA"), per(a, "SQ_INSTS_VALU"), per(b, "SQ_INSTS_LDS")
        rows.append((t, short(key[0]), key[1], n, us, mfma_busy / (1024 * us * 2400.0), n_mfma / 1e6, n_valu / 1e6, n_lds / 1e6,
                     (n_valu - n_mfma) / n_mfma if n_mfma else float("nan"),
                     per(a, "SQ_WAIT_ANY") / wave, per(b, "SQ_WAIT_INST_ANY") / wave, per(b, "SQ_ACTIVE_INST_ANY") / wave, per(b, "SQ_WAIT_INST_LDS") / wave,
                     per(c, "SQ_LDS_BANK_CONFLICT") / per(c, "SQ_LDS_IDX_ACTIVE") if per(c, "SQ_LDS_IDX_ACTIVE") else float("nan")))
    rows.sort(reverse=True)
    with open(out, "w") as f:
        f.write("| kernel | grid | launches | us (under PMC) | MFMA busy / (1024 SIMDs x us x 2.4 GHz) | MFMA insts M | VALU insts M (incl. MFMA) | LDS insts M | VALU : MFMA | "
                "WAIT_ANY / WAVE | WAIT_INST_ANY / WAVE | ACTIVE_INST_ANY / WAVE | WAIT_INST_LDS / WAVE | LDS bank conflict / LDS active |\n"
                "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| `{r[1]}` | {r[2]} | {r[3]} | {r[4]:.1f} | {r[5]:.3f} | {r[6]:.2f} | {r[7]:.2f} | {r[8]:.2f} | {r[9]:.1f} | {r[10]:.2f} | {r[11]:.2f} | {r[12]:.2f} | {r[13]:.3f} | {r[14]:.3f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
