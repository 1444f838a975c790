#!/usr/bin/env python3
"""Per-shape timing of the fp32-parity (split3) projections of the headline config through the C ABI, with the in-kernel time line of
workgroup 0 (kd_prof_clock_buffer: prologue / first tile's K loop / its epilogue, shader clocks) and an A/B against the round-1
kernels (option x3 = 0).     python benchmarks/x3_bench.py [iters]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KDIFF_GEMM"] = "split3"
import k_diffusion_amd as K  # noqa: E402

nat, ops = K._native, K.ops
dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20

# (name, batch, H, W, nh, K, d_ff)
LEVELS = [("L0", 32, 64, 64, 2, 128, 384), ("L1", 32, 32, 32, 4, 256, 768), ("L2", 32, 16, 16, 8, 512, 1536)]


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
    clk = torch.zeros(16, dtype=torch.int64, device=dev)
    nat.lib().kd_prof_clock_buffer(C.c_void_p(clk.data_ptr()))
    fn()
    torch.cuda.synchronize()
    nat.lib().kd_prof_clock_buffer(None)
    c = clk.cpu().tolist()
    if c[3] <= c[1] or not c[4]:
        return ""
    ghz = (c[2] - c[0]) / (c[3] - c[1]) * 0.1
    if c[8] and c[10]:       # half-tile kernels (two workgroups per CU): a workgroup of a later round, its second half tile
        return (f"  wg 5/8 {c[2] - c[0]} clk @ {ghz:.2f} GHz: prologue {c[4] - c[0]}, half tile 1: K loop {c[9] - c[8]} (96 MFMAs: floor 3072), "
                f"epilogue {c[10] - c[9]}, stages {c[7]}")
    return (f"  wg0 {c[2] - c[0]} clk @ {ghz:.2f} GHz: prologue {c[4] - c[0]}, tile0 K loop {c[5] - c[4]}, tile0 epilogue {c[6] - c[5]}, stages {c[7]}")


# out patch projection of the headline config (out_norm -> 128 -> 48, unpatch, c_out / c_skip)
if True:
    B, Hh, Ww, Kd = 32, 64, 64, 128
    g = torch.Generator().manual_seed(2)
    tok = torch.randn(B, Hh, Ww, Kd, generator=g).to(dev)
    gain = (1 + 0.1 * torch.randn(Kd, generator=g)).to(dev)
    wpo = (torch.randn(48, Kd, generator=g) * Kd ** -0.5).to(dev)
    img = torch.randn(B, 3, 4 * Hh, 4 * Ww, generator=g).to(dev)
    sig = torch.rand(B, generator=g).to(dev) + 0.1
    outi = torch.empty_like(img)
    line = "patch-out M=131072 N=48 K=128"
    for opt in (1, 0):
        nat.set_option("x3_unpatch", opt)
        us = timed(lambda: ops.patch_out(tok, gain, wpo, (4, 4), 3, x_in=img, sigma=sig, sigma_data=0.5, out=outi))
        line += f" | x3_unpatch={opt}: {us:7.1f} us ({(tok.numel() + 2 * img.numel()) * 4 / us * 1e-3:5.0f} GB/s)"
    nat.set_option("x3_unpatch", 1)
    print(line)

for name, B, H, W, nh, Kd, dff in LEVELS:
    T, d = H * W, nh * 64
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, Kd, generator=g).to(dev)
    scale = (1 + 0.2 * torch.randn(B, Kd, generator=g)).to(dev)
    wq = (torch.randn(3 * d, Kd, generator=g) * Kd ** -0.5).to(dev)
    wg = (torch.randn(2 * dff, Kd, generator=g) * Kd ** -0.5).to(dev)
    qs = torch.linspace(5.0, 12.0, nh).to(dev)
    rope = K.models.axial_rope                  # the product's own position / frequency helpers (nothing outside tests/ imports the oracle)
    pos, freqs = rope.make_axial_pos(H, W).reshape(T, 2), rope.rope_freqs(32, nh)
    cos_t, sin_t = rope.rope_tables(pos.reshape(H, W, 2), freqs)
    qk = (qs, cos_t.to(dev), sin_t.to(dev), nh, pos.contiguous().to(dev), (freqs / (2 * np.pi)).contiguous().to(dev))
    oq, og = torch.empty(B, T, 3 * d, device=dev), torch.empty(B, T, dff, device=dev)
    cases = {
        "qkv": lambda: ops.norm_linear(x, scale, wq, rows_per_sample=T, epi=nat.EPI_QKV, qk=qk, qkv_packed=True, out=oq),
        "geglu": lambda: ops.norm_linear(x, scale, wg, rows_per_sample=T, epi=nat.EPI_GEGLU, out=og),
    }
    # the pre-split operand path: norm -> planes (one launch) + tiled GEMM (gemm_x3t.hip); down projection on hidden planes vs on fp32
    xh, xl = ops.norm_split(x, scale, rows_per_sample=T)
    hh, hl = (torch.empty(B, T, dff, device=dev, dtype=torch.bfloat16) for _ in range(2))
    wd = (torch.randn(Kd, dff, generator=g) * dff ** -0.5).to(dev)
    res, yo = torch.randn(B, T, Kd, generator=g).to(dev), torch.empty(B, T, Kd, device=dev)
    hid32 = torch.randn(B, T, dff, generator=g).to(dev)

    def qkv_tiled():
        a = ops.norm_split(x, scale, rows_per_sample=T)
        ops.gemm(None, wq, oq, M=B * T, N=3 * d, K=Kd, epi=nat.EPI_QKV, rows_per_sample=T, qk=qk, qkv_packed=True, a_planes=a)

    def geglu_tiled():
        a = ops.norm_split(x, scale, rows_per_sample=T)
        ops.gemm(None, wg, None, M=B * T, N=dff, K=Kd, epi=nat.EPI_GEGLU, a_planes=a, c_planes=(hh, hl))
    extra = {
        "norm_split": (lambda: ops.norm_split(x, scale, rows_per_sample=T), 0.0, 8.0 * B * T * Kd),
        "qkv tiled (+split)": (qkv_tiled, 2.0 * B * T * 3 * d * Kd, 4.0 * (B * T * Kd + B * T * 3 * d)),
        "geglu tiled (+split)": (geglu_tiled, 2.0 * B * T * 2 * dff * Kd, 4.0 * (B * T * Kd + B * T * dff)),
        "qkv tiled alone": (lambda: ops.gemm(None, wq, oq, M=B * T, N=3 * d, K=Kd, epi=nat.EPI_QKV, rows_per_sample=T, qk=qk, qkv_packed=True, a_planes=(xh, xl)),
                            2.0 * B * T * 3 * d * Kd, 4.0 * (B * T * Kd + B * T * 3 * d)),
        "geglu tiled alone": (lambda: ops.gemm(None, wg, None, M=B * T, N=dff, K=Kd, epi=nat.EPI_GEGLU, a_planes=(xh, xl), c_planes=(hh, hl)),
                              2.0 * B * T * 2 * dff * Kd, 4.0 * (B * T * Kd + B * T * dff)),
        "down planes": (lambda: ops.gemm(None, wd, yo, M=B * T, N=Kd, K=dff, epi=nat.EPI_RESIDUAL, residual=res, a_planes=(hh, hl)),
                        2.0 * B * T * Kd * dff, 4.0 * (B * T * dff + 2 * B * T * Kd)),
        "down fp32 (round 1)": (lambda: ops.gemm(hid32, wd, yo, M=B * T, N=Kd, K=dff, epi=nat.EPI_RESIDUAL, residual=res),
                                2.0 * B * T * Kd * dff, 4.0 * (B * T * dff + 2 * B * T * Kd)),
    }
    if Kd <= 256:
        extra["geglu fused -> planes"] = (lambda: ops.gemm(x, wg, None, M=B * T, N=dff, K=Kd, epi=nat.EPI_GEGLU, norm_scale=scale, scale_stride=Kd,
                                                           rows_per_sample=T, c_planes=(hh, hl)), 2.0 * B * T * 2 * dff * Kd, 4.0 * (B * T * Kd + B * T * dff))
    if Kd <= 256:
        wup2, wdn2 = wg, wd
        xf = x.clone()
        extra["ffn fused (x3)"] = (lambda: ops.ffn(xf, scale, wup2, wdn2, out=xf, rows_per_sample=T), 2.0 * B * T * 3 * dff * Kd, 8.0 * B * T * Kd)

        def ffn_pair():
            h = ops.norm_linear(x, scale, wg, rows_per_sample=T, epi=nat.EPI_GEGLU, out=og)
            ops.gemm(h, wd, yo, M=B * T, N=Kd, K=dff, epi=nat.EPI_RESIDUAL, residual=res)
        extra["ffn as two GEMMs"] = (ffn_pair, 2.0 * B * T * 3 * dff * Kd, 8.0 * B * T * Kd)
    for ename, (fn, flops, byts) in extra.items():
        us = timed(fn)
        print(f"{name} {ename:22s} {us:7.1f} us {flops / us * 1e-6:6.1f} TF/s (x3 executed {3 * flops / us * 1e-6 / 2500:.2f} of peak) {byts / us * 1e-3:5.0f} GB/s")
        if ename.startswith("ffn fused"):
            tl = timeline(fn)
            if tl:
                print(tl.replace("tile0 K loop", "tile0 up").replace("tile0 epilogue", "tile0 GEGLU + down"))
    # the residual projection behind the attention core (out = x + att W_o^T): round-3 A-stationary kernel vs the round-1 tile kernel
    wo = (torch.randn(Kd, Kd, generator=g) * Kd ** -0.5).to(dev)
    att = torch.randn(B, T, Kd, generator=g).to(dev)
    line = f"{name} out-proj M={B * T:6d} N={Kd:3d} K={Kd:3d}"
    for opt in (1, 0):
        nat.set_option("x3_res", opt)
        us = timed(lambda: ops.gemm(att, wo, yo, M=B * T, N=Kd, K=Kd, epi=nat.EPI_RESIDUAL, residual=res))
        line += f" | x3_res={opt}: {us:7.1f} us {12.0 * B * T * Kd / us * 1e-3:5.0f} GB/s (x3 executed {6.0 * B * T * Kd * Kd / us * 1e-6 / 2500:.2f} of peak)"
    nat.set_option("x3_res", 1)
    print(line)
    # projections without a norm in front (gemm_x3r.hip) against the round-1 tile kernel: out projection, down projection, token merge
    nat.set_option("x3_res", 0)
    shapes = [("out-proj", B * T, Kd, Kd, att, wo, res, False), ("down", B * T, Kd, dff, hid32, wd, res, False)]
    if name != "L2":
        wm = (torch.randn(2 * Kd, 4 * Kd, generator=g) * (4 * Kd) ** -0.5).to(dev)
        shapes.append(("merge", B * T // 4, 2 * Kd, 4 * Kd, x, wm, None, True))
    for what, M_, N_, K_, a_, w_, r_, mg in shapes:
        line = f"{name} {what:8s} M={M_:6d} N={N_:4d} K={K_:4d}"
        outb = torch.empty(M_, N_, device=dev)
        for opt in (2, 0):
            nat.set_option("x3r", opt)
            if mg:
                f = lambda: ops.gemm(a_, w_, outb, M=M_, N=N_, K=K_, a_mode=nat.A_MERGE2x2, grid=(H // 2, W // 2))  # noqa: E731
            else:
                f = lambda: ops.gemm(a_, w_, outb, M=M_, N=N_, K=K_, epi=nat.EPI_RESIDUAL, residual=r_)  # noqa: E731
            us = timed(f)
            line += f" | x3r={opt}: {us:7.1f} us (x3 executed {6.0 * M_ * N_ * K_ / us * 1e-6 / 2500:.2f} of peak, {4.0 * (M_ * K_ + (2 if r_ is not None else 1) * M_ * N_) / us * 1e-3:5.0f} GB/s)"
        nat.set_option("x3r", 1)
        print(line)
        clk = torch.zeros(16, dtype=torch.int64, device=dev)
        nat.lib().kd_prof_clock_buffer(C.c_void_p(clk.data_ptr()))
        f()
        torch.cuda.synchronize()
        nat.lib().kd_prof_clock_buffer(None)
        c = clk.cpu().tolist()
        if c[12]:
            print(f"  wg 5/8: K loop {c[2] - c[0]} clk for {c[7]} stages; stage 8: chunk 0 {c[9] - c[8]}, wait {c[10] - c[9]}, barrier {c[11] - c[10]}, chunk 1 {c[12] - c[11]}")
    nat.set_option("x3_res", 1)
    for cname, fn in cases.items():
        nw = 3 * d if cname == "qkv" else 2 * dff
        flops = 2.0 * B * T * nw * Kd
        byts = 4.0 * (B * T * Kd + B * T * (3 * d if cname == "qkv" else dff))
        line = f"{name} {cname:6s} M={B * T:6d} N={nw:4d} K={Kd:3d}"
        for opt in (1, 0):
            nat.set_option("x3", opt)
            us = timed(fn)
            line += f" | x3={opt}: {us:7.1f} us {flops / us * 1e-6:6.1f} TF/s (x3 executed {3 * flops / us * 1e-6 / 2500:.2f} of peak) {byts / us * 1e-3:5.0f} GB/s"
            if opt == 1:
                tl = timeline(fn)
        nat.set_option("x3", 1)
        if Kd == 256:                # two workgroups per CU over half tiles (default) vs one workgroup per CU
            nat.set_option("x3_half", 0)
            line += f" | x3_half=0: {timed(fn):7.1f} us"
            nat.set_option("x3_half", 1)
        print(line)
        if tl:
            print(tl)
