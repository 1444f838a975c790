#!/usr/bin/env python3
"""fp8 mode (csrc/gemm_mx8.hip): the norm -> projection kernels at the headline shapes against the bf16 kernels they replace, with workgroup
0's in-kernel time line (kd_prof_clock_buffer: row prologue, first tile's K loop, first tile's epilogue, the rest).

    python benchmarks/mx8_bench.py [iters]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KDIFF_GEMM"] = "bf16"
import k_diffusion_amd as K  # noqa: E402

nat, ops = K._native, K.ops
dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30


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


def timeline(fn):
    clk = torch.zeros(64, dtype=torch.int64, device=dev)
    nat.lib().kd_prof_clock_buffer(C.c_void_p(clk.data_ptr()))
    fn()
    torch.cuda.synchronize()
    nat.lib().kd_prof_clock_buffer(None)
    c = clk.cpu().tolist()
    total, real = c[2] - c[0], (c[3] - c[1]) / 100.0          # s_memrealtime: 100 MHz
    mhz = total / real if real > 0 else 0.0
    return (f"workgroup 0: {total} clocks = {real:.1f} us at {mhz:.0f} MHz: rows {c[4] - c[0]}, first tile K loop {c[5] - c[4]}, "
            f"its epilogue {c[6] - c[5]}, the other {c[7] - 1} tiles {c[2] - c[6]}")


for name, B, T, d, dff in [("L1", 32, 1024, 256, 768), ("L2", 32, 256, 512, 1536)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, d, generator=g).to(dev).to(torch.bfloat16)
    scale = (1 + 0.1 * torch.randn(B, d, generator=g)).to(dev)
    wg = (torch.randn(2 * dff, d, generator=g) * d ** -0.5).to(dev)
    wq = (torch.randn(3 * d, d, generator=g) * d ** -0.5).to(dev)
    nh = d // 64
    H = W = int(T ** 0.5)
    import numpy as np
    rope = K.models.axial_rope
    qk = (torch.linspace(5.0, 12.0, nh).to(dev), rope.make_axial_pos(H, W).reshape(T, 2).contiguous().to(dev), (rope.rope_freqs(32, nh) / (2 * np.pi)).contiguous().to(dev), nh)
    cases = [("GEGLU", dict(epi=nat.EPI_GEGLU), wg), ("qkv", dict(epi=nat.EPI_QKV, qk=qk), wq)]
    for what, kw, w in cases:
        f16 = lambda: ops.norm_linear(x, scale, w, rows_per_sample=T, **kw)  # noqa: E731
        f8 = lambda: ops.norm_linear(x, scale, w, rows_per_sample=T, mx8=True, **kw)  # noqa: E731
        line = f"{name} {what:6s} M={B * T} K={d}: bf16 {timed(f16):6.1f} us | mx8 {timed(f8):6.1f} us"
        if what == "GEGLU":
            f8c = lambda: ops.norm_linear(x, scale, w, rows_per_sample=T, epi=nat.EPI_GEGLU, mx8=True, c_fp8=True)  # noqa: E731
            line += f" | mx8 -> e4m3 {timed(f8c):6.1f} us"
            print(line, flush=True)
            print("   ", timeline(f8c), flush=True)
        else:
            print(line, flush=True)
            print("   ", timeline(f8), flush=True)
