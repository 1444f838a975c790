#!/usr/bin/env python3
"""Round-5 additions to the golden vectors, recorded from the REAL reference (build container only):

    python -m oracle.make_golden_r5

  tests/golden/samples_r5.safetensors
      smp64_cifar_heun50        BASELINE configs[1] at its STATED batch: config_cifar10_transformer.json, sample_heun x 50 (99 model calls),
                                batch 64, fp32 -- the reference's own Denoiser + sample_heun on the CPU.  (samples.safetensors holds the
                                same case at batch 2; kernel selection depends on the row count, so the batch the benchmark runs at gets
                                its own golden.)  Images cases.B64_KEEP kept.
      smp64_cifar_heun50_bf16   the same run with the reference's denoiser under torch.autocast("cpu", dtype=torch.bfloat16): what the
                                bf16 arithmetic mode ("sample_heun 50 steps bf16" in BASELINE.json) is gated against.
The earlier files are not touched.
"""
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


def b64_samples():
    out = {}
    case, cfgname, sampler, steps, batch = cases.SAMPLE_B64_CASE
    cfg, model = mg.build_reference_model(cfgname)
    mc = cfg["model"]
    den = K.Denoiser(model, sigma_data=mc["sigma_data"])
    x, cls = cases.sample_inputs(cfg, batch)
    extra = {"class_cond": cls} if cls is not None else {}
    sigmas = S.get_sigmas_karras(steps, mc["sigma_min"], mc["sigma_max"], rho=7.)

    def den_bf16(xx, ss, **kw):
        with torch.autocast("cpu", dtype=torch.bfloat16):
            return den(xx, ss, **kw).float()
    for tag, fn in (("", den), ("_bf16", den_bf16)):
        t0 = time.time()
        y = getattr(S, sampler)(fn, x, sigmas, extra_args=extra, disable=True)
        out[case + tag] = y[cases.B64_KEEP]
        print(f"{case}{tag}: |y|max {y.abs().max():.4f}  {time.time() - t0:.1f}s", flush=True)
    return out


def main():
    torch.set_num_threads(os.cpu_count())
    gd = cases.GOLDEN_DIR
    meta = {"generator": "oracle/make_golden_r5.py", "torch": torch.__version__,
            "reference": "crowsonkb/k-diffusion @ /root/reference (v0.2.0.dev0)"}
    out = b64_samples()
    save_file({k: v.detach().contiguous() for k, v in out.items()}, os.path.join(gd, "samples_r5.safetensors"), metadata=meta)
    for f in sorted(os.listdir(gd)):
        print(f, os.path.getsize(os.path.join(gd, f)))


if __name__ == "__main__":
    main()
