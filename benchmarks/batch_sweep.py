#!/usr/bin/env python3
"""Every batch size 1 .. N of a config through one denoiser call in each arithmetic mode, against the exact-fp32-MFMA mode of the same build on
the same inputs: the kernel a projection gets changes with its row count (latency forms, one / two workgroups per CU, fused blocks, fp8
projections, panel splits), and this walks all of them.  Prints the worst per-sample max-norm relative distance per mode and the batch it
occurred at; exits 1 if a mode leaves its band.   python benchmarks/batch_sweep.py [config.json] [N]
(tests/test_model_gpu.py::test_batch_sizes_between_the_tested_ones_agree_across_modes is the seeded sample of this that the suite runs.)"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import k_diffusion_amd as K  # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "configs/config_oxford_flowers.json")
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    cfg = K.config.load_config(path)
    mc = cfg["model"]
    model = K.config.make_model(cfg).eval().requires_grad_(False)
    model.load_state_dict(K.synth.synth_state_dict(model.state_dict(), seed=1234))
    model = model.to("cuda")
    den = K.Denoiser(model, sigma_data=mc["sigma_data"])
    g = torch.Generator().manual_seed(3)
    sig = (torch.rand(top, generator=g) * 6 - 3).exp()
    shape = (mc["input_channels"], *mc["input_size"])
    x = (K.synth.synth_noise_batch(shape, 5, 0, top, 1.0) * sig[:, None, None, None]).cuda()
    sig = sig.cuda()
    nc = cfg["dataset"].get("num_classes") or 0
    cls = (torch.arange(top) % nc).cuda() if nc else None
    bands = {"split3": 5e-4, "bf16": 6e-2, "fp8": 2.5e-1}
    worst = {m: (0.0, 0) for m in bands}
    bad = []
    for B in range(1, top + 1):
        kw = {"class_cond": cls[:B]} if cls is not None else {}
        os.environ["KDIFF_GEMM"] = "exact"
        ref = den(x[:B], sig[:B], **kw)
        for m, band in bands.items():
            os.environ["KDIFF_GEMM"] = m
            got = den(x[:B], sig[:B], **kw)
            e = float(((got - ref).flatten(1).abs().amax(1) / ref.flatten(1).abs().amax(1)).max())
            if not e < band:
                bad.append((B, m, e))
            if e > worst[m][0]:
                worst[m] = (e, B)
    print(f"{os.path.basename(path)}: batch 1 .. {top}, per-sample max-norm distance to the exact-fp32 mode, worst over samples and batches")
    for m, (e, B) in worst.items():
        print(f"  {m:7s} {e:.3e} (batch {B}; band {bands[m]:.1e})")
    print("  outside the band:", bad if bad else "none")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
