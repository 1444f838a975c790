#!/usr/bin/env python3
"""One denoiser call at a batch whose tensors pass 2^31 elements (level-0 qkv: batch x 4096 tokens x 384 features), per arithmetic mode: the
first, a middle and the LAST sample against the same samples run alone -- index arithmetic that wraps at 32 bits shows up in the last ones.
    python benchmarks/big_batch.py [batch = 1536] [config]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import k_diffusion_amd as K  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
    path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "configs/config_oxford_flowers.json")
    cfg = K.config.load_config(path)
    mc = cfg["model"]
    model = K.config.make_model(cfg).eval().requires_grad_(False)
    model.load_state_dict(K.synth.synth_state_dict(model.state_dict(), seed=1234))
    model = model.to("cuda")
    den = K.Denoiser(model, sigma_data=mc["sigma_data"])
    shape = (mc["input_channels"], *mc["input_size"])
    g = torch.Generator(device="cuda").manual_seed(1)
    sig = (torch.rand(B, device="cuda", generator=g) * 6 - 3).exp()
    x = torch.randn(B, *shape, device="cuda", generator=g) * sig[:, None, None, None]
    nc = cfg["dataset"].get("num_classes") or 0
    cls = (torch.arange(B, device="cuda") % nc) if nc else None
    picks = [0, B // 2 + 1, B - 1]
    bad = 0
    for mode, band in (("split3", 5e-5), ("bf16", 6e-2), ("fp8", 2.5e-1)):
        os.environ["KDIFF_GEMM"] = mode
        torch.cuda.reset_peak_memory_stats()
        kw = {"class_cond": cls} if cls is not None else {}
        y = den(x, sig, **kw)
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(y).all())
        errs = []
        for i in picks:
            kw1 = {"class_cond": cls[i:i + 1]} if cls is not None else {}
            y1 = den(x[i:i + 1].contiguous(), sig[i:i + 1].contiguous(), **kw1)
            errs.append(float((y[i:i + 1] - y1).abs().max() / y1.abs().max()))
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        print(f"{mode:7s} batch {B}: finite {ok}, samples {picks} against themselves run alone: {['%.2e' % e for e in errs]} (band {band:.0e}), peak {peak:.1f} GiB")
        bad += (not ok) or any(not e < band for e in errs)
        del y
        model._drop_plans()
        torch.cuda.empty_cache()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
