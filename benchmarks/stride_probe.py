"""Probe: does the row stride of A / C (K*4, N*4 bytes) matter?  (L2-channel hot-spotting on power-of-two-ish strides)"""
import sys, torch
sys.path.insert(0, ".")
import k_diffusion_amd as K
ops = K.ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, Kd in [(8192, 512, 1536), (8192, 512, 1568), (8192, 512, 1600), (8192, 528, 1536), (8192, 528, 1568),
                 (8192, 512, 512), (8192, 512, 544), (8192, 528, 544),
                 (32768, 256, 768), (32768, 256, 800), (32768, 272, 800), (32768, 256, 256), (32768, 272, 288),
                 (131072, 128, 384), (131072, 128, 416), (131072, 144, 416)]:
    x = torch.randn(M, Kd, device="cuda"); w = torch.randn(N, Kd, device="cuda") / Kd ** 0.5; r = torch.randn(M, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    us = t(lambda: ops.linear(x, w, residual=r, out=out))
    print(f"M={M:6d} N={N:4d} K={Kd:5d}  {us:7.1f} us   {2.0*M*N*Kd/us/1e6:7.1f} TF/s alg   {(M*Kd+2*M*N)*4/us/1e3:7.0f} GB/s", flush=True)
