#!/usr/bin/env python3
"""Socket power and shader clock per KERNEL: each heavy fp32-parity (split3) kernel of the headline config runs back to back for ~2.5 s while
rocm-smi is sampled from a side thread.  Which kernels make the chip throttle (the whole path runs at ~2.0 GHz / ~1.2 kW: profiles/r04_power_clock.log)?

    python benchmarks/kernel_power.py"""
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KDIFF_GEMM"] = "split3"
import k_diffusion_amd as K  # noqa: E402

nat, ops = K._native, K.ops
dev = "cuda"
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            break
        p = re.search(r"Socket Graphics Package Power \(W\): ([0-9.]+)", txt)
        c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
        if p and c:
            out.append((float(p.group(1)), int(c.group(1))))
        time.sleep(0.15)


def measure(name, fn, flops, byts):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    t0 = time.perf_counter()
    th.start()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    body = out[len(out) // 3:] or out                      # (the first samples still see the previous kernel / the ramp)
    pw = sum(p for p, _ in body) / max(len(body), 1)
    ck = sum(c for _, c in body) / max(len(body), 1)
    print(f"{name:44s} {us:7.1f} us  {3 * flops / us * 1e-6:6.0f} TF/s executed  {byts / us * 1e-3:5.0f} GB/s  | {pw:6.0f} W  {ck:5.0f} MHz  ({len(body)} samples)", flush=True)


B = 32
for name, H, W, nh, Kd, dff in [("L0", 64, 64, 2, 128, 384), ("L1", 32, 32, 4, 256, 768), ("L2", 16, 16, 8, 512, 1536)]:
    T, d = H * W, nh * 64
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, Kd, generator=g).to(dev)
    scale = (1 + 0.2 * torch.randn(B, Kd, generator=g)).to(dev)
    wq = (torch.randn(3 * d, Kd, generator=g) * Kd ** -0.5).to(dev)
    wg = (torch.randn(2 * dff, Kd, generator=g) * Kd ** -0.5).to(dev)
    wd = (torch.randn(Kd, dff, generator=g) * dff ** -0.5).to(dev)
    wo = (torch.randn(Kd, Kd, generator=g) * Kd ** -0.5).to(dev)
    att = torch.randn(B, T, Kd, generator=g).to(dev)
    hid = torch.randn(B, T, dff, generator=g).to(dev)
    res = torch.randn(B, T, Kd, generator=g).to(dev)
    qs = torch.linspace(5.0, 12.0, nh).to(dev)
    rope = K.models.axial_rope                  # the product's own position / frequency helpers (nothing outside tests/ imports the oracle)
    pos, freqs = rope.make_axial_pos(H, W).reshape(T, 2), rope.rope_freqs(32, nh)
    cos_t, sin_t = rope.rope_tables(pos.reshape(H, W, 2), freqs)
    qk = (qs, cos_t.to(dev), sin_t.to(dev), nh, pos.contiguous().to(dev), (freqs / (2 * np.pi)).contiguous().to(dev))
    oq, og, yo = torch.empty(B, T, 3 * d, device=dev), torch.empty(B, T, dff, device=dev), torch.empty(B, T, Kd, device=dev)
    M = B * T
    measure(f"{name} qkv projection", lambda: ops.norm_linear(x, scale, wq, rows_per_sample=T, epi=nat.EPI_QKV, qk=qk, qkv_packed=True, out=oq),
            2.0 * M * 3 * d * Kd, 4.0 * (M * Kd + M * 3 * d))
    if Kd <= 256:
        xf = x.clone()
        measure(f"{name} fused FF block", lambda: ops.ffn(xf, scale, wg, wd, out=xf, rows_per_sample=T), 2.0 * M * 3 * dff * Kd, 8.0 * M * Kd)
    else:
        measure(f"{name} GEGLU up projection", lambda: ops.norm_linear(x, scale, wg, rows_per_sample=T, epi=nat.EPI_GEGLU, out=og),
                2.0 * M * 2 * dff * Kd, 4.0 * (M * Kd + M * dff))
        measure(f"{name} down projection (gemm_x3r)", lambda: ops.gemm(hid, wd, yo, M=M, N=Kd, K=dff, epi=nat.EPI_RESIDUAL, residual=res),
                2.0 * M * Kd * dff, 4.0 * (M * dff + 2 * M * Kd))
    qkv_p = oq.view(B, H, W, 3 * d)
    if name != "L2":
        measure(f"{name} neighbourhood attention 7x7", lambda: ops.attn_na2d(qkv_p, nh, 7, prep="packed"), 2.0 * M * nh * 64 * 2 * 49, 16.0 * M * d)
    else:
        measure(f"{name} global attention", lambda: ops.attn_global(oq, nh, prep="packed"), 4.0 * B * nh * T * T * 64, 16.0 * M * d)
# for scale: a pure streaming kernel (the fused solver step over the batch of images)
img = torch.randn(B, 3, 256, 256, device=dev)
den = torch.randn_like(img)
measure("solver step (pure streaming)", lambda: ops.sampler_step(nat.STEP_EULER, img, den, c0=1.0, c1=0.5), 0.0, 12.0 * img.numel())
