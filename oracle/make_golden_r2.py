#!/usr/bin/env python3
"""Round-2 additions to the golden vectors, recorded from the REAL reference (build container only):

    python -m oracle.make_golden_r2 [--kat-only]

  tests/golden/kat2.json               known answers of the samplers round 1 left without one (sample_dpm_2, sample_dpm_2_ancestral,
                                       sample_dpmpp_2s_ancestral; recorded noise through ``noise_sampler=``)
  tests/golden/forward_b32.safetensors fp32 forwards of the two 256x256 configs at the full per-GPU batch (32), images B32_KEEP
  tests/golden/forward_bf16.safetensors, samples_bf16.safetensors
                                       the same forward / sampling cases as forward.safetensors / samples.safetensors with the
                                       reference running under torch.autocast("cpu", dtype=torch.bfloat16) -- what the bf16 arithmetic
                                       mode (KDIFF_GEMM=bf16) is compared with besides the fp32 goldens
The round-1 files are not touched (oracle/make_golden.py regenerates them bit-for-bit).
"""
import json
import os
import sys
import time

import torch
from safetensors.torch import save_file

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import make_golden as mg  # noqa: E402  (imports the reference)
from tests.golden import cases  # noqa: E402

K = mg.K
S = K.sampling


def kat2():
    kat = {}
    toy = lambda x, sigma, **kw: torch.tanh(x) / (1 + sigma.view(-1, 1, 1, 1))
    g = torch.Generator().manual_seed(3)
    xt = torch.randn(2, 3, 4, 4, generator=g) * 80
    sig20 = S.get_sigmas_karras(20, 1e-2, 80)
    noise = cases.recorded_noise(tuple(xt.shape), 64, seed=77)
    kat["solver_toy_dpm_2"] = mg.hexf(S.sample_dpm_2(toy, xt, sig20, disable=True))
    it = iter(noise)
    kat["solver_toy_dpm_2_ancestral_recorded"] = mg.hexf(S.sample_dpm_2_ancestral(toy, xt, sig20, disable=True, noise_sampler=lambda a, b: next(it)))
    it = iter(noise)
    kat["solver_toy_dpm_2_ancestral_eta0.4_recorded"] = mg.hexf(
        S.sample_dpm_2_ancestral(toy, xt, sig20, disable=True, eta=0.4, s_noise=0.9, noise_sampler=lambda a, b: next(it)))
    it = iter(noise)
    kat["solver_toy_dpmpp_2s_ancestral_recorded"] = mg.hexf(S.sample_dpmpp_2s_ancestral(toy, xt, sig20, disable=True, noise_sampler=lambda a, b: next(it)))
    it = iter(noise)
    kat["solver_toy_dpmpp_2s_ancestral_eta0.4_recorded"] = mg.hexf(
        S.sample_dpmpp_2s_ancestral(toy, xt, sig20, disable=True, eta=0.4, s_noise=0.9, noise_sampler=lambda a, b: next(it)))
    torch.manual_seed(321)
    kat["solver_toy_dpm_2_churn"] = mg.hexf(S.sample_dpm_2(toy, xt, sig20, disable=True, s_churn=8.0))
    torch.manual_seed(321)
    kat["solver_toy_heun_churn_window"] = mg.hexf(S.sample_heun(toy, xt, sig20, disable=True, s_churn=8.0, s_tmin=0.5, s_tmax=20.0))
    kat["rng_after_heun_churn_window"] = mg.hexf(torch.randn(4))      # the reference draws eps on EVERY step (sampling.py:124,165)
    return kat


def forward_b32():
    out = {}
    for case, cfgname, batch in cases.FORWARD_B32_CASES:
        t0 = time.time()
        cfg, model = mg.build_reference_model(cfgname)
        x, sigma, cls = cases.forward_inputs(cfg, batch, cases.b32_sigmas(batch))
        den = K.Denoiser(model, sigma_data=cfg["model"]["sigma_data"])
        y = den(x, sigma, **({"class_cond": cls} if cls is not None else {}))
        out[case + ".denoised"] = y[cases.B32_KEEP]
        print(f"{case}: |y|max {y.abs().max():.4f}  {time.time() - t0:.1f}s", flush=True)
    return {k: v.detach().contiguous() for k, v in out.items()}


def forward_bf16():
    out = {}
    for case, cfgname, batch, sigmas in cases.FORWARD_CASES:
        t0 = time.time()
        cfg, model = mg.build_reference_model(cfgname)
        x, sigma, cls = cases.forward_inputs(cfg, batch, sigmas)
        den = K.Denoiser(model, sigma_data=cfg["model"]["sigma_data"])
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y = den(x, sigma, **({"class_cond": cls} if cls is not None else {}))
        out[case + ".denoised"] = y.float()
        print(f"{case} [autocast bf16]: |y|max {y.abs().max():.4f}  {time.time() - t0:.1f}s", flush=True)
    return {k: v.detach().contiguous() for k, v in out.items()}


def samples_bf16():
    out = {}
    for case, cfgname, sampler, steps, batch in cases.SAMPLE_CASES:
        if case not in cases.BF16_SAMPLE_CASES:
            continue
        t0 = time.time()
        cfg, model = mg.build_reference_model(cfgname)
        mc = cfg["model"]
        den = K.Denoiser(model, sigma_data=mc["sigma_data"])
        x, cls = cases.sample_inputs(cfg, batch)
        extra = {"class_cond": cls} if cls is not None else {}
        sigmas = S.get_sigmas_karras(steps, mc["sigma_min"], mc["sigma_max"], rho=7.)

        def den_bf16(xx, ss, **kw):
            with torch.autocast("cpu", dtype=torch.bfloat16):
                return den(xx, ss, **kw).float()
        torch.manual_seed(0)
        y = getattr(S, sampler)(den_bf16, x, sigmas, extra_args=extra, disable=True)
        out[case] = y
        print(f"{case} [autocast bf16]: |y|max {y.abs().max():.4f}  {time.time() - t0:.1f}s", flush=True)
    return {k: v.detach().contiguous() for k, v in out.items()}


def main():
    torch.set_num_threads(os.cpu_count())
    gd = cases.GOLDEN_DIR
    meta = {"generator": "oracle/make_golden_r2.py", "torch": torch.__version__,
            "reference": "crowsonkb/k-diffusion @ /root/reference (v0.2.0.dev0)"}
    json.dump({"meta": meta, **kat2()}, open(os.path.join(gd, "kat2.json"), "w"), indent=1)
    if "--kat-only" in sys.argv:
        return
    save_file(forward_b32(), os.path.join(gd, "forward_b32.safetensors"), metadata=meta)
    save_file(forward_bf16(), os.path.join(gd, "forward_bf16.safetensors"), metadata=meta)
    save_file(samples_bf16(), os.path.join(gd, "samples_bf16.safetensors"), metadata=meta)
    for f in sorted(os.listdir(gd)):
        print(f, os.path.getsize(os.path.join(gd, f)))


if __name__ == "__main__":
    main()
