#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy table of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).

    python benchmarks/resource_usage.py k-diffusion_amd/csrc/attn_x3.hip [filter]

Runs without a GPU (cross-compile).  The attention files take the Makefile's extra flag (-mllvm -amdgpu-mfma-vgpr-form=1) automatically."""
import os
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    if os.path.basename(src).startswith(("attn_", "block_")):
        flags += ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
    if os.path.basename(src) == "elementwise.hip":
        flags += ["-ffp-contract=off"]
    out = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *flags, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: (?:[^:]*:\d+:\d+: )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip() or v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" ")[0]] = v
    print(f"{'SGPR':>5} {'VGPR':>5} {'AGPR':>5} {'scr':>5} {'occ':>4} {'LDS':>7}  kernel")
    for r in rows:
        if "kd_text_pad" in r["name"] or flt not in r["name"]:
            continue
        print(f"{r.get('TotalSGPRs', '?'):>5} {r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('ScratchSize', '?'):>5} {r.get('Occupancy', '?'):>4} "
              f"{r.get('LDS', '?'):>7}  {r['name'][:150]}")


if __name__ == "__main__":
    main()
