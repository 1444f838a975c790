#!/usr/bin/env python3
"""fp32-parity attention cores on operands stored split by the qkv projection (prep = "packed"): the round-3 cores (csrc/attn_x3.hip)
against the round-1 cores (option attn_x3 = 0) at the headline config's levels.     python benchmarks/attn_x3_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KDIFF_GEMM"] = "split3"
import k_diffusion_amd as K  # noqa: E402

nat, ops = K._native, K.ops
dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def split_stored(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    sh = x.shape[:-1]
    return torch.cat([hi.view(*sh, 16, 4), lo.view(*sh, 16, 4)], dim=-1).contiguous().view(torch.float32).view(*sh, 64)


for name, B, H, W, nh, kind in [("L0", 32, 64, 64, 2, "na"), ("L1", 32, 32, 32, 4, "na"), ("L2", 32, 16, 16, 8, "global")]:
    g = torch.Generator().manual_seed(1)
    qkv = split_stored((torch.randn(B, H, W, 3 * nh, 64, generator=g) * 0.5).to(dev)).view(B, H, W, 3 * nh * 64)
    out = torch.empty(B, H, W, nh * 64, device=dev)
    if kind == "na":
        fn = lambda: ops.attn_na2d(qkv, nh, 7, prep="packed", out=out)  # noqa: E731
    else:
        fn = lambda: ops.attn_global(qkv.view(B, H * W, -1), nh, prep="packed", out=out.view(B, H * W, -1))  # noqa: E731
    byts = 16.0 * B * H * W * nh * 64
    line = f"{name} {kind:6s} {H}x{W} nh={nh}"
    for opt in (1, 0):
        nat.set_option("attn_x3", opt)
        us = timed(fn)
        line += f" | attn_x3={opt}: {us:7.1f} us {byts / us * 1e-3:5.0f} GB/s"
    nat.set_option("attn_x3", 1)
    print(line)
