#!/usr/bin/env python3
"""Probe: HBM rate of 128-byte pieces at a 768-byte stride (how the neighbourhood core reads K / V of one head out of the [token][3][heads][64]
bf16 qkv rows of level 0) against the same bytes contiguous (a head-major plane layout).  torch's own copy kernels on both sides: a proxy for the
DRAM page locality of the two layouts, not for our kernels."""
import torch

dev = "cuda"
N = 32 * 4096 * 4                       # tokens (x4: 400 MB of rows, beyond the 256 MiB Infinity Cache)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for pieces, label in ((6, "level 0: 2 heads, 768-byte rows"), (12, "level 1: 4 heads, 1536-byte rows")):
    rows = torch.randn(N // (pieces // 6), pieces, 64, device=dev).to(torch.bfloat16)
    n = rows.shape[0]
    out = torch.empty(n, 64, device=dev, dtype=torch.bfloat16)
    plane = rows[:, 2, :].contiguous()
    us_s = t(lambda: out.copy_(rows[:, 2, :]))
    us_c = t(lambda: out.copy_(plane))
    by = 2 * n * 128
    print(f"{label}: strided pieces {by / us_s / 1e6:6.2f} TB/s (read + write), contiguous plane {by / us_c / 1e6:6.2f} TB/s", flush=True)
    # all pieces of the rows read in one pass (what a fused consumer of whole rows would see)
    out6 = torch.empty_like(rows)
    us_a = t(lambda: out6.copy_(rows))
    print(f"   whole rows: {2 * rows.numel() * 2 / us_a / 1e6:6.2f} TB/s", flush=True)
