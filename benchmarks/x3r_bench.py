#!/usr/bin/env python3
"""Projections without a norm in front (csrc/gemm_x3r.hip: out / down projection + residual, token merges) of the headline config through
the C ABI: loader-wave form (round 4, option x3r_lw = 1) vs staging requests inside the compute waves' K loop (x3r_lw = 0) vs the
round-1 tile kernel (x3r = 0), with the in-kernel time line of one workgroup's stage 8 and a check of every variant against fp64.

    python benchmarks/x3r_bench.py [iters]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KDIFF_GEMM"] = "split3"
import k_diffusion_amd as K  # noqa: E402

nat, ops = K._native, K.ops
dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30

# (name, batch, H, W, width, d_ff)
LEVELS = [("L0", 32, 64, 64, 128, 384), ("L1", 32, 32, 32, 256, 768), ("L2", 32, 16, 16, 512, 1536)]


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


def stamps(fn, tiles=0):
    clk = torch.zeros(32 + 3 * max(tiles, 1), dtype=torch.int64, device=dev)
    clk[15] = 0x4b44                      # extended time line: entry / exit ticks (100 MHz) of every workgroup
    for _ in range(2):                    # (the second launch: warm clocks)
        nat.lib().kd_prof_clock_buffer(C.c_void_p(clk.data_ptr()))
        fn()
        torch.cuda.synchronize()
        nat.lib().kd_prof_clock_buffer(None)
    c = clk.cpu().tolist()
    wg = ""
    if tiles:
        w = torch.tensor(c[32:32 + 3 * tiles]).view(tiles, 3)
        t0 = int(w[:, 0].min())
        dur = (w[:, 1] - w[:, 0]).float() / 100
        wg = (f"\n        all {tiles} workgroups: start skew {(int(w[:, 0].max()) - t0) / 100:.1f} us, duration min / median / max {float(dur.min()):.1f} / "
              f"{float(dur.median()):.1f} / {float(dur.max()):.1f} us, last exit {(int(w[:, 1].max()) - t0) / 100:.1f} us after the first entry"
              + (f", last loader exit {(int(w[:, 2].max()) - t0) / 100:.1f} us" if int(w[:, 2].max()) else ""))
    if not c[12]:
        return wg
    ghz = (c[2] - c[0]) / max(c[3] - c[1], 1) * 0.1
    pre = f"entry -> loop {c[0] - c[13]}, loop end -> stores out {c[14] - c[2]}; " if c[13] and c[14] else ""
    return (f"      wg 5/8 @ {ghz:.2f} GHz: {pre}K loop {c[2] - c[0]} clk for {c[7]} stages = {(c[2] - c[0]) / max(c[7], 1):.0f} per stage (24 MFMAs: floor 768); "
            f"stage 8: chunk 0 {c[9] - c[8]}, wait {c[10] - c[9]}, barrier {c[11] - c[10]}, chunk 1 {c[12] - c[11]}") + wg


for name, B, H, W, d, dff in LEVELS:
    T = H * W
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, d, generator=g).to(dev)
    att = torch.randn(B, T, d, generator=g).to(dev)
    hid = torch.randn(B, T, dff, generator=g).to(dev)
    res = torch.randn(B, T, d, generator=g).to(dev)
    wo = (torch.randn(d, d, generator=g) * d ** -0.5).to(dev)
    wd = (torch.randn(d, dff, generator=g) * dff ** -0.5).to(dev)
    shapes = [("out-proj", B * T, d, d, att, wo, res, False), ("down", B * T, d, dff, hid, wd, res, False)]
    if name != "L2":
        wm = (torch.randn(2 * d, 4 * d, generator=g) * (4 * d) ** -0.5).to(dev)
        shapes.append(("merge", B * T // 4, 2 * d, 4 * d, x, wm, None, True))
    if name != "L2":
        # TokenSplit back into this level (image_transformer_v2.py:610-621): coarse tokens [B, H/2, W/2, 2d] -> 4 d columns -> lerp with the skip
        xc = torch.randn(B, H // 2, W // 2, 2 * d, generator=g).to(dev)
        ws = (torch.randn(4 * d, 2 * d, generator=g) * (2 * d) ** -0.5).to(dev)
        skip = torch.randn(B, H, W, d, generator=g).to(dev)
        facs = torch.tensor([0.37]).to(dev)
        shapes.append(("split", B * T // 4, 4 * d, 2 * d, xc, ws, skip, "split"))
    for what, M_, N_, K_, a_, w_, r_, mg in shapes:
        outb = torch.empty(M_, N_, device=dev)
        if mg == "split":
            outs = torch.empty_like(skip)
            f = lambda: ops.token_split_lerp(xc, ws, skip, facs, out=outs)  # noqa: E731
            fac = 0.37
            rows = torch.randperm(M_, generator=g)[:256]
            y = (xc.reshape(M_, K_)[rows.to(dev)].double() @ ws.double().T).view(-1, 2, 2, d)          # [rows, nh, nw, e]
            bb, rr = rows // ((H // 2) * (W // 2)), rows % ((H // 2) * (W // 2))
            hh, wwc = (rr // (W // 2)).to(dev), (rr % (W // 2)).to(dev)
            bb = bb.to(dev)
            sk = torch.stack([torch.stack([skip[bb, 2 * hh + i, 2 * wwc + j] for j in range(2)], 1) for i in range(2)], 1).double()
            ref_split = sk + fac * (y - sk)

            def check_split():
                got = torch.stack([torch.stack([outs[bb, 2 * hh + i, 2 * wwc + j] for j in range(2)], 1) for i in range(2)], 1).double()
                return float((got - ref_split).abs().max() / ref_split.abs().max())
        elif mg:
            f = lambda: ops.gemm(a_, w_, outb, M=M_, N=N_, K=K_, a_mode=nat.A_MERGE2x2, grid=(H // 2, W // 2))  # noqa: E731
            a2 = a_.view(B, H // 2, 2, W // 2, 2, d).permute(0, 1, 3, 2, 4, 5).reshape(M_, K_)
        else:
            f = lambda: ops.gemm(a_, w_, outb, M=M_, N=N_, K=K_, epi=nat.EPI_RESIDUAL, residual=r_)  # noqa: E731
            a2 = a_.reshape(M_, K_)
        if mg != "split":
            rows = torch.randperm(M_, generator=g)[:256].to(dev)            # fp64 check on a row sample
            ref = a2[rows].double() @ w_.double().T + (r_.reshape(M_, N_)[rows].double() if r_ is not None else 0)
        line = f"{name} {what:8s} M={M_:6d} N={N_:4d} K={K_:4d} tiles={-(-M_ // 128) * (N_ // 128):5d}"
        tl = ""
        for label, x3r, lw in (("loader waves", 2, 1), ("in-loop requests", 2, 0), ("round-1 tile kernel", 0, 0)):
            nat.set_option("x3r", x3r)
            nat.set_option("x3r_lw", lw)
            us = timed(f)
            err = check_split() if mg == "split" else float(((outb[rows].double() - ref).abs().max() / ref.abs().max()))
            line += f" | {label}: {us:6.1f} us ({6.0 * M_ * N_ * K_ / us * 1e-6 / 2500:.2f} of peak, err {err:.1e})"
            assert err < 1e-4, (name, what, label, err)
            if x3r:
                t = stamps(f, -(-M_ // 128) * (N_ // 128))
                tl += ("\n" + t.replace("wg 5/8", f"{label}: wg 5/8")) if t else ""
        nat.set_option("x3r", 1)
        nat.set_option("x3r_lw", 1)
        print(line + tl, flush=True)
