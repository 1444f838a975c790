#!/usr/bin/env python3
"""Fused fp32-parity feed-forward block (kd_ffn_f32) at the headline config's levels 0 and 1: time per launch and the time line of
workgroup 0 (kd_prof_clock_buffer): prologue, the third d_ff tile's up projection / GEGLU / down k-steps, epilogue.
    python benchmarks/ffn_x3_bench.py [iters]"""
import ctypes as C
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


for name, B, T, Kd, dff in [("L0", 32, 4096, 128, 384), ("L1", 32, 1024, 256, 768)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, Kd, generator=g).to(dev)
    scale = (1 + 0.2 * torch.randn(B, Kd, generator=g)).to(dev)
    wu = (torch.randn(2 * dff, Kd, generator=g) * Kd ** -0.5).to(dev)
    wd = (torch.randn(Kd, dff, generator=g) * dff ** -0.5).to(dev)
    y = torch.empty_like(x)
    fn = lambda: ops.ffn(x, scale, wu, wd, out=y, rows_per_sample=T)  # noqa: E731
    if Kd == 128:
        nat.set_option("ffn_x3_half", 0)
        print(f"{name} one workgroup per CU, 64-feature tiles: {timed(fn):7.1f} us")
        nat.set_option("ffn_x3_half", 1)
    us = timed(fn)
    flops = 2.0 * B * T * 3 * dff * Kd
    print(f"{name} ffn_x3 M={B * T} K={Kd} d_ff={dff}: {us:7.1f} us, {3 * flops / us * 1e-6:6.0f} TF/s executed ({3 * flops / us * 1e-6 / 2500:.2f} of the bf16 MFMA peak)")
    # with the attention block's out projection fused in front, against the out projection as its own launch + the block
    att = torch.randn(B, T, Kd, generator=g).to(dev)
    wo = (torch.randn(Kd, Kd, generator=g) * Kd ** -0.5).to(dev)
    x1 = torch.empty_like(x)
    fused = timed(lambda: ops.ffn(x, scale, wu, wd, out=y, rows_per_sample=T, attn=att, w_out=wo))

    def pair():
        ops.gemm(att, wo, x1, M=B * T, N=Kd, K=Kd, epi=nat.EPI_RESIDUAL, residual=x)
        ops.ffn(x1, scale, wu, wd, out=y, rows_per_sample=T)
    print(f"  + out projection: fused {fused:7.1f} us, as two launches {timed(pair):7.1f} us")
    clk = torch.zeros(16, dtype=torch.int64, device=dev)
    nat.lib().kd_prof_clock_buffer(C.c_void_p(clk.data_ptr()))
    fn()
    torch.cuda.synchronize()
    nat.lib().kd_prof_clock_buffer(None)
    c = clk.cpu().tolist()
    ghz = (c[2] - c[0]) / max(c[3] - c[1], 1) * 0.1
    half = Kd == 128                 # two workgroups per CU, half tiles of 32 hidden features (ffn_x3h_kernel)
    nt = dff // (32 if half else 64)
    mf_up, mf_dn = 3 * (2 if half else 4) * (Kd // 16), 3 * (Kd // 32) * (2 if half else 4)
    print(f"  wg 5/8 {c[2] - c[0]} clk @ {ghz:.2f} GHz: prologue {c[4] - c[0]}, tiles {c[12] - c[4]} ({nt} x {(c[12] - c[4]) // nt}), epilogue {c[2] - c[12]}")
    for label, f2 in (("fused out projection", lambda: ops.ffn(x, scale, wu, wd, out=y, rows_per_sample=T, attn=att, w_out=wo)),):
        clk2 = torch.zeros(16, dtype=torch.int64, device=dev)
        nat.lib().kd_prof_clock_buffer(C.c_void_p(clk2.data_ptr()))
        f2()
        torch.cuda.synchronize()
        nat.lib().kd_prof_clock_buffer(None)
        c2 = clk2.cpu().tolist()
        print(f"  {label}: wg 5/8 {c2[2] - c2[0]} clk: prologue + out projection + norm {c2[4] - c2[0]} (to the first weight request) ..., tiles {c2[12] - c2[4]}, epilogue {c2[2] - c2[12]}")
    print(f"  {'half tile 4' if half else 'tile 2'}: up {c[9] - c[8]} ({mf_up} MFMAs: floor {32 * mf_up}), GEGLU {c[10] - c[9]}, down {c[11] - c[10]} ({mf_dn} MFMAs: floor {32 * mf_dn})")
