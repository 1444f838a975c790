#!/usr/bin/env python3
"""bf16 mode, projections without a norm in front (csrc/gemm_bf16.hip: gemm_tiled_kernel) at the headline shapes: the loader-wave form of
round 4 (option tiled_lw = 1: grids of at most one tile per CU, i.e. the level-2 shapes) against the round-2 form, checked against fp64.

    python benchmarks/tiled_bf16_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KDIFF_GEMM"] = "bf16"
import k_diffusion_amd as K  # noqa: E402

nat, ops = K._native, K.ops
dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
BF = nat.PREC_BF16


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


# (name, batch, H, W, width, d_ff)
for name, B, H, W, d, dff in [("L1", 32, 32, 32, 256, 768), ("L2", 32, 16, 16, 512, 1536)]:
    T = H * W
    g = torch.Generator().manual_seed(1)
    bf = lambda t: t.to(dev).to(torch.bfloat16)  # noqa: E731
    x, att, hid, res = (bf(torch.randn(B, T, n, generator=g)) for n in (d, d, dff, d))
    wo = (torch.randn(d, d, generator=g) * d ** -0.5).to(dev)
    wd = (torch.randn(d, dff, generator=g) * dff ** -0.5).to(dev)
    shapes = [("out-proj", B * T, d, d, att, wo, res, "res"), ("down", B * T, d, dff, hid, wd, res, "res")]
    if name != "L2":
        wm = (torch.randn(2 * d, 4 * d, generator=g) * (4 * d) ** -0.5).to(dev)
        shapes.append(("merge", B * T // 4, 2 * d, 4 * d, x, wm, None, "merge"))
        xc = bf(torch.randn(B, H // 2, W // 2, 2 * d, generator=g))
        ws = (torch.randn(4 * d, 2 * d, generator=g) * (2 * d) ** -0.5).to(dev)
        skip = bf(torch.randn(B, H, W, d, generator=g))
        shapes.append(("split", B * T // 4, 4 * d, 2 * d, xc, ws, skip, "split"))
    for what, M_, N_, K_, a_, w_, r_, kind in shapes:
        outb = torch.empty(M_, N_, device=dev, dtype=torch.bfloat16)
        w16 = w_.to(torch.bfloat16).double()
        rows = torch.randperm(M_, generator=g)[:128].to(dev)
        if kind == "split":
            outs = torch.empty_like(r_)
            facs = torch.tensor([0.37]).to(dev)
            f = lambda: ops.token_split_lerp(a_, w_, r_, facs, out=outs)  # noqa: E731
            ref = None
        elif kind == "merge":
            f = lambda: ops.gemm(a_, w_, outb, M=M_, N=N_, K=K_, a_mode=nat.A_MERGE2x2, grid=(H // 2, W // 2), precision=BF)  # noqa: E731
            a2 = a_.view(B, H // 2, 2, W // 2, 2, d).permute(0, 1, 3, 2, 4, 5).reshape(M_, K_)
            ref = a2[rows].double() @ w16.T
        else:
            f = lambda: ops.gemm(a_, w_, outb, M=M_, N=N_, K=K_, epi=nat.EPI_RESIDUAL, residual=r_, precision=BF)  # noqa: E731
            ref = a_.reshape(M_, K_)[rows].double() @ w16.T + r_.reshape(M_, N_)[rows].double()
        line = f"{name} {what:8s} M={M_:6d} N={N_:4d} K={K_:4d} tiles={-(-M_ // 128) * (N_ // 128):5d}"
        outs_by = {}
        for label, lw in (("loader waves", 1), ("round-2 form", 0)):
            nat.set_option("tiled_lw", lw)
            us = timed(f)
            got = (outs if kind == "split" else outb).clone()
            outs_by[lw] = got
            err = "" if ref is None else f", err {float((got[rows].double() - ref).abs().max() / ref.abs().max()):.1e}"
            line += f" | {label}: {us:6.1f} us ({2.0 * M_ * N_ * K_ / us * 1e-6 / 2500:.2f} of peak{err})"
        nat.set_option("tiled_lw", 1)
        same = bool(torch.equal(outs_by[0], outs_by[1]))
        print(line + f" | identical results: {same}", flush=True)
