"""Quick GPU check of the bf16 arithmetic mode against the fp32 goldens (and the autocast goldens when present),
plus a forward / sampler timing at the headline shape.  python benchmarks/bf16_check.py"""
import os
import sys
import time

import torch
from safetensors.torch import load_file

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["KDIFF_GEMM"] = sys.argv[1] if len(sys.argv) > 1 else "bf16"
import k_diffusion_amd as KD  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.helpers import relerr  # noqa: E402

DEV = "cuda"
gd = cases.GOLDEN_DIR
gold = load_file(os.path.join(gd, "forward.safetensors"))
gold16 = load_file(os.path.join(gd, "forward_bf16.safetensors")) if os.path.exists(os.path.join(gd, "forward_bf16.safetensors")) else {}


def build(name):
    cfg = KD.config.load_config(cases.raw_config(name))
    model = KD.config.make_model(cfg).eval().requires_grad_(False)
    model.load_state_dict(KD.synth.synth_state_dict(model.state_dict(), seed=cases.WEIGHT_SEED))
    return cfg, model.to(DEV)


for case, cfgname, batch, sigmas in cases.FORWARD_CASES:
    try:
        cfg, model = build(cfgname)
        x, sigma, cls = cases.forward_inputs(cfg, batch, sigmas)
        kw = {"class_cond": cls.to(DEV)} if cls is not None else {}
        den = KD.Denoiser(model, sigma_data=cfg["model"]["sigma_data"])
        yd = den(x.to(DEV), sigma.to(DEV), **kw)
        torch.cuda.synchronize()
        e32 = relerr(yd, gold[case + ".denoised"])
        e16 = relerr(yd, gold16[case + ".denoised"]) if case + ".denoised" in gold16 else float("nan")
        ref_gap = relerr(gold16[case + ".denoised"], gold[case + ".denoised"]) if case + ".denoised" in gold16 else float("nan")
        print(f"{case:20s} vs fp32 ref {e32:.3e}   vs autocast ref {e16:.3e}   (autocast ref vs fp32 ref {ref_gap:.3e})  nan={bool(torch.isnan(yd).any())}", flush=True)
    except Exception as e:  # noqa
        print(f"{case:20s} FAILED: {type(e).__name__}: {e}", flush=True)

for cfgname in ("flowers_na", "flowers_sw", "cifar"):
    cfg, model = build(cfgname)
    mc = cfg["model"]
    B = 64 if cfgname == "cifar" else 32
    den = KD.Denoiser(model, sigma_data=mc["sigma_data"])
    x, cls = cases.sample_inputs(cfg, B)
    x = x.to(DEV)
    extra = {"class_cond": cls.to(DEV)} if cls is not None else {}
    sig = torch.full((B,), 2.5, device=DEV)
    for _ in range(3):
        den(x, sig, **extra)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        den(x, sig, **extra)
    torch.cuda.synchronize()
    fwd = (time.time() - t0) / 20
    sigmas = KD.sampling.get_sigmas_karras(50, mc["sigma_min"], mc["sigma_max"], rho=7.).to(DEV)
    sampler = KD.sampling.sample_heun if cfgname == "cifar" else KD.sampling.sample_dpmpp_2m
    sampler(den, x, sigmas, extra_args=extra, disable=True)
    torch.cuda.synchronize()
    t0 = time.time()
    y = sampler(den, x, sigmas, extra_args=extra, disable=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"{cfgname}: B={B} forward {fwd * 1e3:.3f} ms   50-step {sampler.__name__} {dt * 1e3:.1f} ms = {B / dt:.1f} img/s  |y|max {y.abs().max().item():.3f} nan={bool(torch.isnan(y).any())}", flush=True)
