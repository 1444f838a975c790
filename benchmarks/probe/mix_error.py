"""Per-GEMM error of the ways to carry a fp32-grade product on the bf16 / fp16 / e4m3 matrix instructions, emulated in torch on the CPU against an fp64 product
(the matrix-time side is mix_probe.cpp):  python benchmarks/probe/mix_error.py
  split3     bf16 hi / lo, hi*hi + hi*lo + lo*hi (the fp32-parity mode)
  f16+fp8    fp16 hi*hi + e4m3 (x_lo*w_hi + x_hi*w_lo), per-(row, 32-k block) power-of-two scales, the lo blocks' scale = the hi blocks' * 2^-11
  bf16+fp8   the same with a bf16 main term (lo scale 2^-8)
  bf16       one bf16 term
"""
import torch
torch.manual_seed(0)
def scale_of(amax):
    r=(amax/448.0).float().contiguous()
    byte=((r.view(torch.int32)+0x7FFFFF)>>23).clamp(1,253)
    return torch.ldexp(torch.ones_like(r), byte-127)
def q8(u, s):
    return (u/s).to(torch.float8_e4m3fn).float()*s
def blocks(x): return x.reshape(*x.shape[:-1], x.shape[-1]//32, 32)
for K in (128,256,512,1536):
    M,N=512,256
    x=torch.randn(M,K)*torch.rand(M,1)*3; w=torch.randn(N,K)*0.05
    ref=(x.double()@w.double().T)
    den=ref.abs().max()
    # split3
    xh=x.bfloat16().float(); xl=(x-xh).bfloat16().float(); wh=w.bfloat16().float(); wl=(w-wh).bfloat16().float()
    s3=(xh@wh.T+xh@wl.T+xl@wh.T)
    # bf16 one term
    b1=xh@wh.T
    # f16 + fp8 corrections
    xh16=x.half().float(); xl16=x-xh16; wh16=w.half().float(); wl16=w-wh16
    sx=scale_of(blocks(xh16).abs().amax(-1,keepdim=True)); sw=scale_of(blocks(wh16).abs().amax(-1,keepdim=True))
    qxh=q8(blocks(xh16),sx).reshape(M,K); qxl=q8(blocks(xl16),sx*2.0**-11).reshape(M,K)
    qwh=q8(blocks(wh16),sw).reshape(N,K); qwl=q8(blocks(wl16),sw*2.0**-11).reshape(N,K)
    mix=xh16@wh16.T + (qxl@qwh.T + qxh@qwl.T)
    # bf16 main + fp8 corrections (the first idea)
    sxb=scale_of(blocks(xh).abs().amax(-1,keepdim=True)); swb=scale_of(blocks(wh).abs().amax(-1,keepdim=True))
    xlb=x-xh; wlb=w-wh
    mixb=xh@wh.T + (q8(blocks(xlb),sxb*2.0**-8).reshape(M,K)@q8(blocks(wh),swb).reshape(N,K).T + q8(blocks(xh),sxb).reshape(M,K)@q8(blocks(wlb),swb*2.0**-8).reshape(N,K).T)
    f32=(x@w.T)
    e=lambda y: ((y.double()-ref).abs().max()/den).item()
    r=lambda y: ((y.double()-ref).pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()).item()
    print(f"K={K:5d}  max-norm rel err: fp32 {e(f32):.2e}  split3 {e(s3):.2e}  f16+fp8 {e(mix):.2e}  bf16+fp8 {e(mixb):.2e}  bf16 {e(b1):.2e}   | rms: split3 {r(s3):.2e} f16+fp8 {r(mix):.2e} bf16+fp8 {r(mixb):.2e} bf16 {r(b1):.2e}")
