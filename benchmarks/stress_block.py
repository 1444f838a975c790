#!/usr/bin/env python3
"""Race screen of the round-5 block kernels (csrc/block_bf16.hip): kd_attn_block_bf16 (with and without the fused out projection and its
per-sample rendezvous) and kd_proj_block_bf16 run the same products in the same order as the launches they replace, so every launch must
reproduce those bit for bit.  Each shape is launched `reps` times with fresh random inputs every few launches, interleaved with a memory-bound
kernel on a second stream (uneven load, workgroups that start late), and every word is compared; the rendezvous counters must stay zero.

    python benchmarks/stress_block.py [reps]"""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
side = torch.cuda.Stream()
junk_a, junk_b = torch.randn(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
bad = 0


def noise():
    with torch.cuda.stream(side):
        junk_b.copy_(junk_a)


def rnd(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


for B, T, Kw in ((32, 256, 512), (5, 256, 512), (64, 256, 256), (8, 1024, 256)):
    nh = Kw // 64
    H, W = T // 16, 16
    qk = (torch.linspace(5., 12., nh).to(dev), hdit.axial_pos(H, W).reshape(T, 2).contiguous().to(dev), (hdit.rope_freqs(nh) / (2 * np.pi)).contiguous().to(dev), nh)
    wq, wo, wg = rnd(1, 3 * Kw, Kw, scale=Kw ** -0.5), rnd(2, Kw, Kw, scale=0.5 * Kw ** -0.5), rnd(3, 6 * Kw, Kw, scale=Kw ** -0.5)
    mism = {"attn": 0, "attn+out": 0, "geglu": 0, "qkv": 0}
    flagged = 0
    for r in range(reps):
        if r % 10 == 0:
            x = rnd(100 + r, B, T, Kw).to(torch.bfloat16)
            sc = 1 + 0.2 * rnd(200 + r, B, Kw)
            qkv = ops.norm_linear(x, sc, wq, rows_per_sample=T, epi=nat.EPI_QKV, qk=qk)
            hid = ops.norm_linear(x, sc, wg, rows_per_sample=T, epi=nat.EPI_GEGLU)
            if T == 256:
                att = ops.attn_global(qkv, nh)
                x_ref = x.clone()
                ops.gemm(att, wo, x_ref, M=B * T, N=Kw, K=Kw, epi=nat.EPI_RESIDUAL, residual=x_ref, precision=nat.PREC_BF16)
        if r % 3 == 0:
            noise()
        mism["geglu"] += int(not torch.equal(ops.proj_block(x, sc, wg, rows_per_sample=T), hid))
        mism["qkv"] += int(not torch.equal(ops.proj_block(x, sc, wq, rows_per_sample=T, epi=nat.EPI_QKV, qk=qk), qkv))
        if T == 256:
            mism["attn"] += int(not torch.equal(ops.attn_block(x, sc, wq, rows_per_sample=T, qk=qk), att))
            x1 = x.clone()
            a1, _, sync = ops.attn_block(x1, sc, wq, rows_per_sample=T, qk=qk, w_out=wo)
            mism["attn+out"] += int(not (torch.equal(a1, att) and torch.equal(x1, x_ref)))
            flagged += int(bool(sync.any()))
    torch.cuda.synchronize()
    print(f"B={B:3d} T={T:4d} K={Kw}: {reps} launches each, mismatching: {mism}, rendezvous counters non-zero after a launch: {flagged}", flush=True)
    bad += sum(mism.values()) + flagged
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
