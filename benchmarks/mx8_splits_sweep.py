#!/usr/bin/env python3
"""fp8 mode: the norm -> GEGLU kernel (kd_gemm_mx8, e4m3 output) at the headline shapes for every n-split of a row panel (library option
mx8_splits) against the launcher's own choice (rounds x (row prologue + tiles per split), 0).  Round 6, one box:
L1 (12 n-tiles): {0: 32.4, 1: 37.1, 2: 32.9, 3: 34.4, 4: 35.4, 6: 38.8, 12: 51.6} us; L2 (24): {0: 28.4, 1: 74.1, 2: 43.5, 3: 33.5, 4: 28.6, 6: 30.2,
8: 27.0, 12: 31.8, 24: 42.4} -- the cost model lands on the measured minimum (2 and 8).

    python benchmarks/mx8_splits_sweep.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["KDIFF_GEMM"] = "bf16"
import k_diffusion_amd as K
nat, ops = K._native, K.ops
dev = "cuda"
def timed(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for name, B, T, d, dff in [("L1", 32, 1024, 256, 768), ("L2", 32, 256, 512, 1536)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, d, generator=g).to(dev).to(torch.bfloat16)
    scale = (1 + 0.1 * torch.randn(B, d, generator=g)).to(dev)
    wg = (torch.randn(2 * dff, d, generator=g) * d ** -0.5).to(dev)
    f = lambda: ops.norm_linear(x, scale, wg, rows_per_sample=T, epi=nat.EPI_GEGLU, mx8=True, c_fp8=True)
    n_tiles = dff // 64
    res = {}
    for sp in [s for s in (0, 1, 2, 3, 4, 6, 8, 12, 24) if s == 0 or n_tiles % s == 0]:
        nat.set_option("mx8_splits", sp)
        res[sp] = round(timed(f), 1)
    nat.set_option("mx8_splits", 0)
    print(name, "GEGLU c8, n_tiles", n_tiles, "us by forced n-splits (0 = cost model):", res, flush=True)
