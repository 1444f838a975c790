#!/usr/bin/env python3
"""Race screen of the loader-wave kernels and of the few-rows latency kernels (round 4): the loader-wave form and the form with the staging requests inside the compute waves run the SAME
products in the SAME order, so their results must be bit-identical on every launch.  Each shape is launched `reps` times in both forms with fresh random
inputs every few launches, interleaved with a memory-bound kernel on a second stream (uneven load), and every word is compared.

    python benchmarks/stress_lw.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k_diffusion_amd as K  # noqa: E402

nat, ops = K._native, K.ops
dev = "cuda"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
side = torch.cuda.Stream()
junk_a, junk_b = torch.randn(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
bad = 0


def noise():
    with torch.cuda.stream(side):
        junk_b.copy_(junk_a)


def screen(name, make, run, option):
    """make(seed) -> inputs; run(inputs) -> output tensor; option: the loader-wave switch (1 = loader waves)."""
    global bad
    mism = 0
    inp = None
    for r in range(reps):
        if r % 10 == 0:
            inp = make(r)
        nat.set_option(option, 1)
        if r % 3 == 0:
            noise()
        a = run(inp).clone()
        nat.set_option(option, 0)
        b = run(inp).clone()
        if not torch.equal(a, b):
            mism += 1
    nat.set_option(option, 1)
    torch.cuda.synchronize()
    print(f"{name:58s} {reps} launch pairs, mismatching: {mism}", flush=True)
    bad += mism


def rnd(seed, *shape, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dev).to(dtype)


# fp32-parity mode: gemm_x3r (residual projection, merge, split + lerp)
os.environ["KDIFF_GEMM"] = "split3"
nat.set_option("x3r", 2)
for M, N, Kd in ((8192, 512, 1536), (8192, 512, 512), (32768, 256, 256), (1000, 256, 768), (256, 512, 1536)):
    w = rnd(7, N, Kd) * Kd ** -0.5
    out = torch.empty(M, N, device=dev)
    screen(f"gemm_x3r residual M={M} N={N} K={Kd}", lambda s: (rnd(s, M, Kd), rnd(s + 1, M, N)),
           lambda i: ops.gemm(i[0], w, out, M=M, N=N, K=Kd, epi=nat.EPI_RESIDUAL, residual=i[1]), "x3r_lw")
for B, H, W, C in ((32, 32, 32, 256), (32, 64, 64, 128), (3, 24, 40, 64)):
    wm = rnd(8, 2 * C, 4 * C) * (4 * C) ** -0.5
    screen(f"gemm_x3r merge [{B},{H},{W},{C}] -> {2 * C}", lambda s: (rnd(s, B, H, W, C),), lambda i: ops.token_merge(i[0], wm), "x3r_lw")
for B, h, w_, Kd, C in ((32, 16, 16, 512, 256), (32, 32, 32, 256, 128), (3, 20, 12, 256, 128)):
    ws = rnd(9, 4 * C, Kd) * Kd ** -0.5
    fac = torch.tensor([0.37], device=dev)
    screen(f"gemm_x3r split [{B},{h},{w_},{Kd}] -> {C}", lambda s: (rnd(s, B, h, w_, Kd), rnd(s + 1, B, 2 * h, 2 * w_, C)),
           lambda i: ops.token_split_lerp(i[0], ws, i[1], fac), "x3r_lw")
nat.set_option("x3r", 1)

# bf16 mode: tiled kernel at one tile per CU
os.environ["KDIFF_GEMM"] = "bf16"
BF = nat.PREC_BF16
for M, N, Kd in ((8192, 512, 1536), (8192, 512, 512), (4096, 256, 768), (1000, 512, 64)):
    w = rnd(7, N, Kd) * Kd ** -0.5
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    screen(f"gemm_bf16_tiled residual M={M} N={N} K={Kd}", lambda s: (rnd(s, M, Kd, dtype=torch.bfloat16), rnd(s + 1, M, N, dtype=torch.bfloat16)),
           lambda i: ops.gemm(i[0], w, out, M=M, N=N, K=Kd, epi=nat.EPI_RESIDUAL, residual=i[1], precision=BF), "tiled_lw")
for B, H, W, C in ((32, 32, 32, 256),):
    wm = rnd(8, 2 * C, 4 * C) * (4 * C) ** -0.5
    screen(f"gemm_bf16_tiled merge [{B},{H},{W},{C}] -> {2 * C}", lambda s: (rnd(s, B, H, W, C, dtype=torch.bfloat16),), lambda i: ops.token_merge(i[0], wm), "tiled_lw")


def screen_repeat(name, make, run):
    """A kernel with no second form to compare with: the same launch twice (noise on the side stream in between) must give the same bits."""
    global bad
    mism = 0
    inp = None
    for r in range(reps):
        if r % 10 == 0:
            inp = make(r)
        a = run(inp).clone()
        if r % 3 == 0:
            noise()
        b = run(inp).clone()
        if not torch.equal(a, b):
            mism += 1
    torch.cuda.synchronize()
    print(f"{name:58s} {reps} launch pairs, mismatching: {mism}", flush=True)
    bad += mism


# the few-rows latency kernels (gemm_x3s.hip / gemm_b16s.hip): K split over 8 waves, partial sums reduced through LDS in wave order
os.environ["KDIFF_GEMM"] = "split3"
nat.set_option("x3s_max_wgs", 1 << 20)
nat.set_option("b16s_max_wgs", 1 << 20)
for M, N, Kd in ((256, 512, 1536), (1024, 256, 768), (196, 256, 256), (4096, 128, 384)):
    w = rnd(7, N, Kd) * Kd ** -0.5
    out = torch.empty(M, N, device=dev)
    screen_repeat(f"gemm_x3s residual M={M} N={N} K={Kd}", lambda s: (rnd(s, M, Kd), rnd(s + 1, M, N)),
                  lambda i: ops.gemm(i[0], w, out, M=M, N=N, K=Kd, epi=nat.EPI_RESIDUAL, residual=i[1]))
for M, N, Kd in ((256, 1536, 512), (1024, 768, 256)):
    wg = rnd(8, 2 * N, Kd) * Kd ** -0.5
    sc = 1 + 0.1 * rnd(9, 1, Kd)
    screen_repeat(f"gemm_x3s norm -> GEGLU M={M} d_ff={N} K={Kd}", lambda s: (rnd(s, 1, M, Kd),),
                  lambda i: ops.norm_linear(i[0], sc, wg, rows_per_sample=M, epi=nat.EPI_GEGLU))
os.environ["KDIFF_GEMM"] = "bf16"
for M, N, Kd in ((256, 512, 1536), (1024, 256, 768), (4096, 128, 384)):
    w = rnd(7, N, Kd) * Kd ** -0.5
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    screen_repeat(f"gemm_bf16_few_rows residual M={M} N={N} K={Kd}", lambda s: (rnd(s, M, Kd, dtype=torch.bfloat16), rnd(s + 1, M, N, dtype=torch.bfloat16)),
                  lambda i: ops.gemm(i[0], w, out, M=M, N=N, K=Kd, epi=nat.EPI_RESIDUAL, residual=i[1], precision=BF))
for M, N, Kd in ((256, 1536, 512),):
    wg = rnd(8, 2 * N, Kd) * Kd ** -0.5
    sc = 1 + 0.1 * rnd(9, 1, Kd)
    screen_repeat(f"gemm_bf16_few_rows norm -> GEGLU M={M} d_ff={N} K={Kd}", lambda s: (rnd(s, 1, M, Kd, dtype=torch.bfloat16),),
                  lambda i: ops.norm_linear(i[0], sc, wg, rows_per_sample=M, epi=nat.EPI_GEGLU))
print("TOTAL mismatching launch pairs:", bad)
sys.exit(1 if bad else 0)
