#!/usr/bin/env python3
"""Per-kernel micro-benchmarks at the flowers (BASELINE config 3/4) shapes, B=32: GEMM families, attention cores.
Times with torch.cuda events on the launch stream (the kernels are launched on torch's current stream).

    python benchmarks/bench_kernels.py [--what gemm,na,global,window] [--iters 20] [--out file.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k_diffusion_amd as K  # noqa: E402
from k_diffusion_amd import _native as nat  # noqa: E402

ops = K.ops
DEV = "cuda"


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3        # us


def _rope(h, nh):
    """cos / sin tables [h*h, nh, 16] of the axial RoPE, on the device (the package's own table builder)."""
    rope = K.models.axial_rope
    cos_t, sin_t = rope.rope_tables(rope.make_axial_pos(h, h).view(h, h, 2), rope.rope_freqs(32, nh))
    return cos_t.to(DEV), sin_t.to(DEV)


def gemm_cases(B):
    T = [B * 4096, B * 1024, B * 256]
    d = [128, 256, 512]
    out = []
    for lv in range(3):
        M, w = T[lv], d[lv]
        out += [(f"L{lv} qkv  norm", M, 3 * w, w, "norm"), (f"L{lv} qkv  norm+prep", M, 3 * w, w, "qkv"), (f"L{lv} out  +res", M, w, w, "res"),
                (f"L{lv} up   norm+geglu", M, 3 * w, w, "geglu"), (f"L{lv} down +res", M, w, 3 * w, "res")]
    return out


def run_gemm(args, res):
    B = args.batch
    for name, M, N, Kd, kind in gemm_cases(B):
        x = torch.randn(M, Kd, device=DEV)
        if kind == "geglu":
            w = torch.randn(2 * N, Kd, device=DEV) / Kd ** 0.5
            scale = 1 + 0.1 * torch.randn(B, Kd, device=DEV)
            out = torch.empty(M, N, device=DEV)
            fn = lambda: ops.norm_linear(x, scale, w, rows_per_sample=M // B, epi=nat.EPI_GEGLU, out=out)
            n_eff = 2 * N
        elif kind == "qkv":
            nh, T = Kd // 64, M // B
            h = int(T ** 0.5)
            cos_t, sin_t = _rope(h, nh)
            qk = (10.0 * torch.ones(nh, device=DEV), cos_t, sin_t, nh)
            w = torch.randn(N, Kd, device=DEV) / Kd ** 0.5
            scale = 1 + 0.1 * torch.randn(B, Kd, device=DEV)
            out = torch.empty(M, N, device=DEV)
            fn = lambda: ops.norm_linear(x, scale, w, rows_per_sample=T, epi=nat.EPI_QKV, out=out, qk=qk)
            n_eff = N
        elif kind == "norm":
            w = torch.randn(N, Kd, device=DEV) / Kd ** 0.5
            scale = 1 + 0.1 * torch.randn(B, Kd, device=DEV)
            out = torch.empty(M, N, device=DEV)
            fn = lambda: ops.norm_linear(x, scale, w, rows_per_sample=M // B, out=out)
            n_eff = N
        else:
            w = torch.randn(N, Kd, device=DEV) / Kd ** 0.5
            r = torch.randn(M, N, device=DEV)
            out = torch.empty(M, N, device=DEV)
            fn = lambda: ops.linear(x, w, residual=r, out=out)
            n_eff = N
        us = timeit(fn, args.iters)
        flops = 2.0 * M * n_eff * Kd
        byts = 4.0 * (M * Kd + M * N * (2 if kind == "res" else 1))
        res[name] = {"us": round(us, 1), "tflops": round(flops / us / 1e6, 1), "gbs": round(byts / us / 1e3, 0), "M": M, "N": N, "K": Kd}
        print(f"{name:24s} M={M:6d} N={N:5d} K={Kd:5d}  {us:8.1f} us  {flops / us / 1e6:7.1f} TF/s  {byts / us / 1e3:7.0f} GB/s", flush=True)


def run_attn(args, res, what):
    B = args.batch
    for lv, (h, nh) in enumerate([(64, 2), (32, 4), (16, 8)]):
        qkv = torch.randn(B, h, h, 3 * nh * 64, device=DEV)
        prep = (10.0 * torch.ones(nh, device=DEV), *_rope(h, nh))
        out = torch.empty(B, h, h, nh * 64, device=DEV)
        byts = 4.0 * B * h * h * nh * 64 * 4
        if what == "na" and lv < 2:
            us = timeit(lambda: ops.attn_na2d(qkv, nh, 7, prep=prep, out=out), args.iters)
            us0 = timeit(lambda: ops.attn_na2d(qkv, nh, 7, out=out), args.iters)
            print(f"na (q,k prepared by the qkv GEMM) L{lv}  {us0:8.1f} us  {byts / us0 / 1e3:7.0f} GB/s", flush=True)
            res[f"na_noprep L{lv}"] = {"us": round(us0, 1), "gbs": round(byts / us0 / 1e3, 0)}
        elif what == "window" and lv < 2:
            us = min(timeit(lambda: ops.attn_window(qkv, nh, 8, s, prep=prep, out=out), args.iters) for s in (0, 4))
        elif what == "global" and lv == 2:
            q2 = qkv.view(B, h * h, -1)
            us = timeit(lambda: ops.attn_global(q2, nh, prep=prep, out=out.view(B, h * h, -1)), args.iters)
        else:
            continue
        res[f"{what} L{lv}"] = {"us": round(us, 1), "gbs": round(byts / us / 1e3, 0)}
        print(f"{what:7s} L{lv} {h}x{h} nh={nh}  {us:8.1f} us  {byts / us / 1e3:7.0f} GB/s", flush=True)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--what", default="gemm,na,global,window")
    p.add_argument("--iters", type=int, default=20)
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--out", default=None)
    args = p.parse_args()
    res = {"mode": os.environ.get("KDIFF_GEMM", "split3")}
    for w in args.what.split(","):
        if w == "gemm":
            run_gemm(args, res)
        else:
            run_attn(args, res, w)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
