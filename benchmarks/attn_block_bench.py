#!/usr/bin/env python3
"""bf16 mode, the global-attention block in front of its out projection (csrc/block_bf16.hip): the one-launch form (kd_attn_block_bf16)
against the two launches it replaces, with workgroup 0's in-kernel time line.

    python benchmarks/attn_block_bench.py [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KDIFF_GEMM"] = "bf16"
import k_diffusion_amd as K  # noqa: E402
from oracle import hdit  # noqa: E402  (positions / frequencies only)

nat, ops = K._native, K.ops
dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for B, nh in ((32, 8), (64, 4)):
    Kw, T = nh * 64, 256
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, Kw, generator=g).to(dev).to(torch.bfloat16)
    scale = (1 + 0.2 * torch.randn(B, Kw, generator=g)).to(dev)
    w = (torch.randn(3 * Kw, Kw, generator=g) * Kw ** -0.5).to(dev)
    qk = (torch.linspace(5., 12., nh).to(dev), hdit.axial_pos(16, 16).reshape(T, 2).contiguous().to(dev),
          (hdit.rope_freqs(nh) / (2 * np.pi)).contiguous().to(dev), nh)
    qkv = torch.empty(B, T, 3 * Kw, device=dev, dtype=torch.bfloat16)
    att2, att1 = torch.empty_like(x), torch.empty_like(x)
    wo = (torch.randn(Kw, Kw, generator=g) * 0.5 * Kw ** -0.5).to(dev)
    x3a, x3b = x.clone(), x.clone()

    def three():
        ops.norm_linear(x3a, scale, w, rows_per_sample=T, epi=nat.EPI_QKV, qk=qk, out=qkv)
        ops.attn_global(qkv, nh, out=att2)
        ops.gemm(att2, wo, x3a, M=B * T, N=Kw, K=Kw, epi=nat.EPI_RESIDUAL, residual=x3a, precision=nat.PREC_BF16)

    def one_out():
        ops.attn_block(x3b, scale, w, rows_per_sample=T, qk=qk, out=att1, w_out=wo)

    def two():
        ops.norm_linear(x, scale, w, rows_per_sample=T, epi=nat.EPI_QKV, qk=qk, out=qkv)
        ops.attn_global(qkv, nh, out=att2)

    def one():
        ops.attn_block(x, scale, w, rows_per_sample=T, qk=qk, out=att1)
    t2, t1 = timed(two), timed(one)
    t3, t1o = timed(three), timed(one_out)
    print(f"B={B} K={Kw} nh={nh}: with the out projection: three launches {t3:6.1f} us, one launch {t1o:6.1f} us "
          f"({(2.0 * B * T * 4 * Kw * Kw + 4.0 * B * nh * T * T * 64) / t1o * 1e-6 / 2500:.3f} of the bf16 MFMA peak)", flush=True)
    flops = 2.0 * B * T * 3 * Kw * Kw + 4.0 * B * nh * T * T * 64
    print(f"B={B} K={Kw} nh={nh}: two launches {t2:6.1f} us, one launch {t1:6.1f} us ({flops / t1 * 1e-6 / 2500:.3f} of the bf16 MFMA peak), "
          f"identical: {bool(torch.equal(att1, att2))}", flush=True)
    grid = B * nh
    clk = torch.zeros(32 + 3 * grid, dtype=torch.int64, device=dev)
    clk[15] = 0x4b44
    for _ in range(3):                    # (the last launch counts: warm clocks, warm caches)
        nat.lib().kd_prof_clock_buffer(clk.data_ptr())
        one_out()
        torch.cuda.synchronize()
        nat.lib().kd_prof_clock_buffer(None)
    c = clk[:16].cpu().tolist()
    wg = clk[32:].cpu().view(grid, 3).double()
    t0 = float(wg[:, 0].min())
    start, end = (wg[:, 0] - t0) / 100, (wg[:, 1] - t0) / 100
    dur = end - start
    qq = lambda t, f: float(torch.quantile(t, f))
    print(f"   {grid} workgroups: first entry -> last exit {float(end.max()):.1f} us; starts median {qq(start, 0.5):.1f}, last {float(start.max()):.1f} us; "
          f"duration min / median / p90 / max {float(dur.min()):.1f} / {qq(dur, 0.5):.1f} / {qq(dur, 0.9):.1f} / {float(dur.max()):.1f} us", flush=True)
    mhz = (c[2] - c[0]) / max(c[3] - c[1], 1) * 100.0
    rel = lambda i: c[i] - c[0]
    print(f"   workgroup 0 (clocks, {mhz:.0f} MHz): rows normalised {rel(4)}, k pass {rel(5) - rel(4)}, v pass {rel(6) - rel(5)}, q pass {rel(8) - rel(6)}, "
          f"scores + softmax {rel(9) - rel(8)}, PV + store {rel(10) - rel(9)}, rendezvous {rel(11) - rel(10)}, out projection {rel(2) - rel(11)}, "
          f"total {rel(2)} = {rel(2) / mhz:.1f} us", flush=True)

# ---- where does the one-launch form pay?  Batch sweep at the level-2 shape with the library's defaults (few-rows kernels on) ----------------
print("batch sweep, K = 512, nh = 8 (two launches: the kernels the planner would pick at that row count)")
for B in (1, 2, 4, 8, 16, 32, 64):
    nh, Kw, T = 8, 512, 256
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, Kw, generator=g).to(dev).to(torch.bfloat16)
    scale = (1 + 0.2 * torch.randn(B, Kw, generator=g)).to(dev)
    w = (torch.randn(3 * Kw, Kw, generator=g) * Kw ** -0.5).to(dev)
    qk = (torch.linspace(5., 12., nh).to(dev), hdit.axial_pos(16, 16).reshape(T, 2).contiguous().to(dev),
          (hdit.rope_freqs(nh) / (2 * np.pi)).contiguous().to(dev), nh)
    qkv = torch.empty(B, T, 3 * Kw, device=dev, dtype=torch.bfloat16)
    att2, att1 = torch.empty_like(x), torch.empty_like(x)

    def two():
        ops.norm_linear(x, scale, w, rows_per_sample=T, epi=nat.EPI_QKV, qk=qk, out=qkv)
        ops.attn_global(qkv, nh, out=att2)

    def one():
        ops.attn_block(x, scale, w, rows_per_sample=T, qk=qk, out=att1)
    t2, t1 = timed(two), timed(one)
    print(f"   B={B:3d}: two launches {t2:6.1f} us, one launch {t1:6.1f} us, max |diff| {float((att1.float() - att2.float()).abs().max()):.2e}", flush=True)
