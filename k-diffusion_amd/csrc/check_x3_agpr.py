#!/usr/bin/env python3
"""Build-time audit of gemm_x3.hip's K = 512 kernels (named-AccVGPR path).

Those kernels keep a row's activation fragments in AccVGPRs a0..a255 that the kernel names literally inside inline asm.  That is only
sound while the COMPILER keeps nothing of its own there and spills nothing (a spill reload is a vector-memory load inside the ring's
counted vmcnt waits).  This script compiles the file to assembly and fails the build if, in any gemm_x3_astat_kernel<32, *>, an AccVGPR
or a scratch access appears outside an ;;#ASMSTART / ;;#ASMEND block (also every ffn_x3_kernel of ffn_x3.hip).

    python check_x3_agpr.py gemm_x3.hip
"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def main(src):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "x3.s")
        subprocess.run([HIPCC, "--offload-arch=" + os.environ.get("ARCH", "gfx950"), "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-S",
                        "--cuda-device-only", src, "-o", out], check=True, capture_output=True)
        txt = open(out).read()
    pos = [(m.start(), m.group(1)) for m in re.finditer(r"\n(_ZN2kd2x3[A-Za-z0-9_]+):", txt)]
    checked = 0
    for (st, name), (en, _) in zip(pos, pos[1:] + [(len(txt), "")]):
        if "gemm_x3_astat_kernelILi32E" not in name and "ffn_x3" not in name and "gemm_x3h_kernel" not in name:
            continue
        body = txt[st:en].split(".end_amdhsa_kernel")[0]
        if not re.search(r"\.vgpr_spill_count:\s*0", txt[st:]) and "vgpr_spill_count" in txt[st:en]:
            raise SystemExit(f"check_x3_agpr: {name}: spills")
        inasm, n_mfma = False, 0
        for line in body.split("\n"):
            t = line.strip()
            if ";;#ASMSTART" in t:
                inasm = True
            elif ";;#ASMEND" in t:
                inasm = False
            elif not t or t.startswith((";", ".")):
                continue
            elif inasm:
                n_mfma += "v_mfma" in t
            elif re.search(r"\ba\[?\d", t) or "accvgpr" in t or "scratch_" in t:
                raise SystemExit(f"check_x3_agpr: {name}: the compiler touches an AccVGPR / scratch outside the kernel's asm: {t}")
        if n_mfma == 0:
            raise SystemExit(f"check_x3_agpr: {name}: no asm MFMA found (did the kernel change?)")
        checked += 1
    if checked == 0:
        raise SystemExit("check_x3_agpr: no named-AccVGPR kernel found in " + src)
    print(f"check_x3_agpr: {os.path.basename(src)}: {checked} named-AccVGPR kernels: a0..a255 are the kernel's alone, no scratch: ok")


if __name__ == "__main__":
    for f in sys.argv[1:]:
        main(f)
