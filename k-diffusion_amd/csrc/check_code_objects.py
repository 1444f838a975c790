#!/usr/bin/env python3
"""Build-time check of the device code objects (kd_common.h, "code warm-up").

Kernels that warm their own code read up to KD_CODE_WARM_MAX bytes behind their entry point as data.  That range must stay
inside the loaded image of the code object, which is guaranteed by layout: every object file that uses the warm-up ends its
executable segment with the `.kd_text_pad` section (kd_text_pad_kernel: >= KD_CODE_WARM_MAX + 4 KiB of s_nop), directly behind
`.text`.  This script extracts the gfx950 code object of each given object file and verifies exactly that; the Makefile and
__graft_entry__.build() run it, and a violation fails the build.

    python check_code_objects.py gemm_bf16.o ffn_bf16.o ...
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("KD_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
# toolchains on which the warm-up was validated on hardware (kd_common.h: KD_CODE_WARM_DEFAULT is 8 only for these; any other
# compiler builds with the warm-up OFF by default).  The layout check below runs -- and must pass -- regardless.
VALIDATED_CLANG = ("AMD clang version 22.0.0git", "roc-7.2.0")
WARM_MAX = 32768          # KD_CODE_WARM_MAX
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def check(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fatbin"), os.path.join(tmp, "co")
        run("objcopy", f"--dump-section=.hip_fatbin={fat}", obj)
        run(os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}")
        sections = run(os.path.join(LLVM, "llvm-readelf"), "-S", "--wide", co)
        segments = run(os.path.join(LLVM, "llvm-readelf"), "-l", "--wide", co)
    sec = {}
    for m in re.finditer(r"\]\s+(\S+)\s+\S+\s+([0-9a-f]{16})\s+[0-9a-f]+\s+([0-9a-f]+)\s", sections):
        sec[m.group(1)] = (int(m.group(2), 16), int(m.group(3), 16))
    if ".text" not in sec:
        raise SystemExit(f"{obj}: no .text in its gfx950 code object")
    if ".kd_text_pad" not in sec:
        raise SystemExit(f"{obj}: no .kd_text_pad section: put KD_TEXT_PAD(tag) at the end of the source file")
    (t0, tn), (p0, pn) = sec[".text"], sec[".kd_text_pad"]
    if p0 < t0 + tn or p0 - (t0 + tn) > 4096:
        raise SystemExit(f"{obj}: .kd_text_pad (0x{p0:x}) does not follow .text (0x{t0:x} + 0x{tn:x})")
    if pn < WARM_MAX + 4096:
        raise SystemExit(f"{obj}: .kd_text_pad holds {pn} bytes, need >= {WARM_MAX + 4096}")
    # both in ONE loadable segment (mapped as one block by the loader)
    ok = False
    for m in re.finditer(r"LOAD\s+0x[0-9a-f]+\s+0x([0-9a-f]+)\s+0x[0-9a-f]+\s+0x[0-9a-f]+\s+0x([0-9a-f]+)\s+R E", segments):
        v0, vn = int(m.group(1), 16), int(m.group(2), 16)
        ok |= v0 <= t0 and p0 + pn <= v0 + vn
    if not ok:
        raise SystemExit(f"{obj}: .text and .kd_text_pad are not inside one executable segment")
    return tn, pn


def main(objs):
    try:
        ver = run(os.path.join(LLVM, "clang"), "--version").splitlines()[0]
    except Exception as e:  # noqa: BLE001
        ver = f"unknown ({e})"
    if all(v in ver for v in VALIDATED_CLANG):
        print(f"check_code_objects: toolchain validated for the code warm-up: {ver}")
    else:
        print(f"check_code_objects: WARNING: code warm-up not validated on this toolchain ({ver}): it builds with the warm-up OFF by "
              "default (kd_common.h KD_CODE_WARM_DEFAULT); re-run profiles/r02_level_entry.md's A/B before enabling it")
    for obj in objs:
        tn, pn = check(obj)
        print(f"check_code_objects: {os.path.basename(obj)}: .text {tn} bytes, pad {pn} bytes behind it: ok")


if __name__ == "__main__":
    main(sys.argv[1:])
