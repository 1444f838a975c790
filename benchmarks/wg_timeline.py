#!/usr/bin/env python3
"""Per-workgroup time line of the heavy fp32-parity (split3) kernels at the headline shapes (batch 32): every workgroup's first thread stamps
its entry and exit (s_memrealtime, x3_common.h: wg_stamp_begin) into the kd_prof_clock_buffer, and this script turns the stamps into launch
ramp / rounds / tail figures: how long a workgroup lives, how the starts are spread, how much of the launch is a tail with idle CUs.

    python benchmarks/wg_timeline.py [iters]"""
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
MAXWG = 8192


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
    clk = torch.zeros(32 + 3 * MAXWG, dtype=torch.int64, device=dev)
    clk[15] = 0x4b44
    for _ in range(3):                    # (the last launch counts: warm clocks, warm caches)
        clk[32:] = 0
        nat.lib().kd_prof_clock_buffer(C.c_void_p(clk.data_ptr()))
        fn()
        torch.cuda.synchronize()
        nat.lib().kd_prof_clock_buffer(None)
    w = clk[32:].cpu().view(MAXWG, 3)
    n = int((w[:, 0] != 0).sum())
    if not n:
        return "      (no workgroup stamps)"
    w = w[:n].double()
    t0 = float(w[:, 0].min())
    start, end = (w[:, 0] - t0) / 100, (w[:, 1] - t0) / 100          # us
    dur = end - start
    total = float(end.max())
    busy = float(dur.sum()) / total                                     # average number of resident workgroups
    order = torch.argsort(start)
    first = float(start[order[min(n - 1, 255)]])
    q = lambda t, f: float(torch.quantile(t, f))
    # time at which half / 90 % of the workgroups have finished: the tail is what follows
    return (f"      {n} workgroups: launch -> last exit {total:.1f} us; starts: first 256 within {first:.1f} us, median {q(start, 0.5):.1f}, last {float(start.max()):.1f} us; "
            f"duration min / median / p90 / max {float(dur.min()):.1f} / {q(dur, 0.5):.1f} / {q(dur, 0.9):.1f} / {float(dur.max()):.1f} us; "
            f"resident on average {busy:.0f}; exits: 50 % by {q(end, 0.5):.1f}, 90 % by {q(end, 0.9):.1f} us")


LEVELS = [("L0", 32, 64, 64, 2, 128, 384), ("L1", 32, 32, 32, 4, 256, 768), ("L2", 32, 16, 16, 8, 512, 1536)]
for name, B, H, W, nh, Kd, dff in LEVELS:
    T, d = H * W, nh * 64
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, Kd, generator=g).to(dev)
    scale = (1 + 0.2 * torch.randn(B, Kd, generator=g)).to(dev)
    wq = (torch.randn(3 * d, Kd, generator=g) * Kd ** -0.5).to(dev)
    wg = (torch.randn(2 * dff, Kd, generator=g) * Kd ** -0.5).to(dev)
    wd = (torch.randn(Kd, dff, generator=g) * dff ** -0.5).to(dev)
    wo = (torch.randn(Kd, Kd, generator=g) * Kd ** -0.5).to(dev)
    att = torch.randn(B, T, Kd, generator=g).to(dev)
    qs = torch.linspace(5.0, 12.0, nh).to(dev)
    rope = K.models.axial_rope                  # the product's own position / frequency helpers (nothing outside tests/ imports the oracle)
    pos, freqs = rope.make_axial_pos(H, W).reshape(T, 2), rope.rope_freqs(32, nh)
    cos_t, sin_t = rope.rope_tables(pos.reshape(H, W, 2), freqs)
    qk = (qs, cos_t.to(dev), sin_t.to(dev), nh, pos.contiguous().to(dev), (freqs / (2 * np.pi)).contiguous().to(dev))
    oq, og = torch.empty(B, T, 3 * d, device=dev), torch.empty(B, T, dff, device=dev)
    cases = [("qkv (norm -> projection + cosine-sim + RoPE)", lambda: ops.norm_linear(x, scale, wq, rows_per_sample=T, epi=nat.EPI_QKV, qk=qk, qkv_packed=True, out=oq))]
    if Kd <= 256:
        xf = x.clone()
        cases.append(("fused FF block", lambda: ops.ffn(xf, scale, wg, wd, out=xf, rows_per_sample=T)))
        if Kd == 128:
            xg = x.clone()
            cases.append(("out projection + fused FF block", lambda: ops.ffn(xg, scale, wg, wd, out=xg, rows_per_sample=T, attn=att, w_out=wo)))
    else:
        cases.append(("GEGLU up projection", lambda: ops.norm_linear(x, scale, wg, rows_per_sample=T, epi=nat.EPI_GEGLU, out=og)))
    qkv_p = oq.view(B, H, W, 3 * d)
    if name != "L2":
        cases.append(("neighbourhood attention 7x7", lambda: ops.attn_na2d(qkv_p, nh, 7, prep="packed")))
    else:
        cases.append(("global attention", lambda: ops.attn_global(oq, nh, prep="packed")))
    for what, fn in cases:
        print(f"{name} {what}: {timed(fn):.1f} us per launch", flush=True)
        print(timeline(fn), flush=True)
