#!/usr/bin/env python3
"""Round-4 additions to the golden vectors, recorded from the REAL reference (build container only):

    python -m oracle.make_golden_r4

  tests/golden/samples_r4.safetensors
      smp32_flowers_na_2m5_bf16 /   the two batch-32, 5-step DPM++2M cases of samples_r3.safetensors (cases.SAMPLE_B32_CASES) with the
      smp32_flowers_sw_2m5_bf16     reference's denoiser under torch.autocast("cpu", dtype=torch.bfloat16): what the bf16 arithmetic mode's
                                    batch-32 run is gated against (tests/test_model_gpu.py::test_b32_sampling_bf16) instead of a
                                    hand-set distance to the fp32 run.  Run at the full batch 32; images cases.B32_KEEP kept.
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


def b32_samples_bf16():
    out = {}
    for case, cfgname, sampler, steps, batch in cases.SAMPLE_B32_CASES:
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
        y = getattr(S, sampler)(den_bf16, x, sigmas, extra_args=extra, disable=True)
        out[case + "_bf16"] = y[cases.B32_KEEP]
        print(f"{case} [autocast bf16]: |y|max {y.abs().max():.4f}  {time.time() - t0:.1f}s", flush=True)
    return out


def main():
    torch.set_num_threads(os.cpu_count())
    gd = cases.GOLDEN_DIR
    meta = {"generator": "oracle/make_golden_r4.py", "torch": torch.__version__,
            "reference": "crowsonkb/k-diffusion @ /root/reference (v0.2.0.dev0)"}
    out = b32_samples_bf16()
    save_file({k: v.detach().contiguous() for k, v in out.items()}, os.path.join(gd, "samples_r4.safetensors"), metadata=meta)
    for f in sorted(os.listdir(gd)):
        print(f, os.path.getsize(os.path.join(gd, f)))


if __name__ == "__main__":
    main()
