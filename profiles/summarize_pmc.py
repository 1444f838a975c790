#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as the MI355X guide prescribes)
into per-kernel HBM traffic per launch.

    python profiles/summarize_pmc.py gpurun_out/pmc_fetch/f_counter_collection.csv \
           gpurun_out/pmc_write/w_counter_collection.csv profiles/rNN_pmc_summary.json [profiles/rNN_pmc_traffic.json]

Units / corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 this
rocprofv3 reports exactly 1/2 of the bytes of a wide coalesced streaming read (128-B requests tallied at
64 B), so FETCH_SIZE is doubled; WRITE_SIZE is used as reported (uncalibrated).  Dispatches are keyed by
(kernel name, grid size) so that the different GEMM shapes of one kernel stay apart.
"""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            key = (row["Kernel_Name"], int(row["Grid_Size"]))
            acc[key][0] += 1
            acc[key][1] += float(row["Counter_Value"])
    return acc


def family_of(symbol):
    """rocprof kernel symbol -> the family name bench.py reports (LaunchScope names in csrc/)."""
    if "x3::ffn_x3" in symbol:          # ffn_x3_kernel<NC, OUTP> / ffn_x3h_kernel<OUTP>: with the out projection fused in front or not
        outp = ("ffn_x3h_kernel<true" in symbol or "ffn_x3h_kernel<1" in symbol or ", true>" in symbol or ", 1>" in symbol
                or "(bool)1" in symbol)
        return "ffn_x3+out" if outp else "ffn_x3"
    for needle, name in (("x3::gemm_x3_astat_kernel", "gemm_x3_astat"), ("x3::gemm_x3h_kernel", "gemm_x3_astat"), ("x3r::gemm_x3r_kernel", "gemm_x3r"),
                         ("x3a::attn_na2d_x3_kernel", "attn_na2d_x3"), ("x3a::attn_global_x3_kernel", "attn_global_x3"), ("x3t::gemm_x3_tiled_kernel", "gemm_x3_tiled"), ("norm_split_kernel", "norm_split_f32"),
                         ("b16::ffn_kernel", "ffn_bf16"), ("b16::unpatch4_kernel", "gemm_bf16_unpatch4"), ("b16::patchin4_kernel", "gemm_bf16_patchin4"),
                         ("b16::gemm_wstat_kernel", "gemm_bf16_wstat"), ("b16::gemm_astat_kernel", "gemm_bf16_astat"), ("b16::gemm_tiled_kernel", "gemm_bf16_tiled"),
                         ("b16::gemm_generic_bf16_kernel", "gemm_bf16_generic"), ("b16::attn_na2d_bf16_kernel", "attn_na2d_bf16"),
                         ("b16::attn_block_bf16_kernel", "attn_block_bf16"), ("mx8::gemm_mx8_astat_kernel", "gemm_mx8_astat"), ("mx8::gemm_mx8_tiled_kernel", "gemm_mx8_tiled"), ("b16::proj_block_bf16_kernel", "proj_block_bf16"), ("b16::attn_dense_bf16_kernel<0", "attn_global_bf16"), ("b16::attn_dense_bf16_kernel", "attn_window_bf16"),
                         ("b16::attn_long_bf16_kernel", "attn_global_bf16"), ("gemm_astat_kernel", "gemm_astat"), ("attn_na2d_kernel", "attn_na2d"), ("attn_global_split_kernel", "attn_global_bf16x3"), ("attn_dense_kernel<0", "attn_global_f32"),
                         ("attn_dense_kernel<1", "attn_window_f32"), ("sampler_step_kernel", "sampler_step_f32")):
        if needle in symbol:
            return name
    if "gemm_skinny_kernel" in symbol:
        return "gemm_skinny"
    if "attn_global_long_kernel" in symbol:
        return "attn_global_bf16x3"
    if "gemm_kernel<" in symbol:        # gemm_kernel<AMODE, NORM, EPI, PREC[, KS]>: PREC 1 = split-bf16x3
        args = [a.strip() for a in symbol.split("gemm_kernel<", 1)[1].split(">", 1)[0].split(",")]
        return "gemm_bf16x3" if len(args) >= 4 and args[3] == "1" else "gemm_f32"
    return None


def main():
    fetch, write, out = sys.argv[1:4]
    fe, wr = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    rows = []
    for key in sorted(set(fe) | set(wr)):
        n_f, kib_f = fe.get(key, (0, 0.0))
        n_w, kib_w = wr.get(key, (0, 0.0))
        if not key[0].startswith(("void kd::", "kd::")):
            continue
        f_bytes = 2.0 * 1024.0 * kib_f / max(n_f, 1)
        w_bytes = 1024.0 * kib_w / max(n_w, 1)
        rows.append({"kernel": key[0], "grid_size": key[1], "dispatches": n_f, "fetch_bytes_per_launch": round(f_bytes),
                     "write_bytes_per_launch": round(w_bytes), "hbm_bytes_per_launch": round(f_bytes + w_bytes)})
    json.dump({"note": "FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE x1, KiB -> bytes; per launch averages", "kernels": rows},
              open(out, "w"), indent=1)
    # per kernel FAMILY (bench.py's grouping: all shapes of one kernel), average over every launch of the profiled run
    fam = {}
    for r in rows:
        name = family_of(r["kernel"])
        if name:
            f = fam.setdefault(name, {"launches": 0, "fetch": 0.0, "write": 0.0})
            f["launches"] += r["dispatches"]
            f["fetch"] += r["fetch_bytes_per_launch"] * r["dispatches"]
            f["write"] += r["write_bytes_per_launch"] * r["dispatches"]
    table = {k: {"launches": v["launches"], "fetch_bytes_per_launch": round(v["fetch"] / v["launches"]),
                 "write_bytes_per_launch": round(v["write"] / v["launches"]),
                 "hbm_bytes_per_launch": round((v["fetch"] + v["write"]) / v["launches"])} for k, v in fam.items() if v["launches"]}
    if len(sys.argv) > 4:
        json.dump(table, open(sys.argv[4], "w"), indent=1)
    for r in sorted(rows, key=lambda r: -r["hbm_bytes_per_launch"])[:25]:
        print(f'{r["kernel"][:60]:60s} grid={r["grid_size"]:8d} n={r["dispatches"]:4d} fetch={r["fetch_bytes_per_launch"]/1e6:9.1f} MB write={r["write_bytes_per_launch"]/1e6:9.1f} MB')


if __name__ == "__main__":
    main()
