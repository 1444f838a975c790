"""Round-2 addendum to the recorded goldens: the reference's OWN bf16-autocast outputs for the kept images of the batch-32
256x256 forwards (tests/golden/forward_b32.safetensors holds their fp32 outputs).

  tests/golden/forward_b32_bf16.safetensors   K.Denoiser(image_transformer_v2) under torch.autocast("cpu", torch.bfloat16) on images
                                              B32_KEEP of the batch-32 inputs (the reference treats every image independently, so the
                                              kept images are run as a batch of their own)

The per-image distance between this file and forward_b32.safetensors is the reference's own reduced-precision distance at each
kept sigma: it calibrates the per-image gates of tests/test_model_gpu.py::test_full_batch_forward_bf16.

Run in the build container (imports /root/reference):  python oracle/make_golden_r2b.py
"""
import os
import sys
import time

import torch
from safetensors.torch import save_file, load_file

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import make_golden as mg  # noqa: E402  (imports the reference)
from tests.golden import cases  # noqa: E402

K = mg.K


def main():
    torch.set_num_threads(os.cpu_count())
    out = {}
    ref32 = load_file(os.path.join(cases.GOLDEN_DIR, "forward_b32.safetensors"))
    for case, cfgname, batch in cases.FORWARD_B32_CASES:
        t0 = time.time()
        cfg, model = mg.build_reference_model(cfgname)
        x, sigma, cls = cases.forward_inputs(cfg, batch, cases.b32_sigmas(batch))
        keep = cases.B32_KEEP
        den = K.Denoiser(model, sigma_data=cfg["model"]["sigma_data"])
        kw = {"class_cond": cls[keep]} if cls is not None else {}
        y32 = den(x[keep], sigma[keep], **kw)
        assert torch.allclose(y32, ref32[case + ".denoised"], atol=1e-4, rtol=1e-4), "kept images are not batch independent?"
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y = den(x[keep], sigma[keep], **kw).float()
        out[case + ".denoised"] = y.detach().contiguous()
        for j, g in enumerate(keep):
            d = (y[j] - y32[j]).norm() / y32[j].norm()
            print(f"{case} image {g} sigma {float(sigma[g]):.4g}: autocast vs fp32 reference {float(d):.3e}", flush=True)
        print(f"{case}: {time.time() - t0:.1f}s", flush=True)
    save_file(out, os.path.join(cases.GOLDEN_DIR, "forward_b32_bf16.safetensors"),
              metadata={"generator": "oracle/make_golden_r2b.py", "torch": torch.__version__})


if __name__ == "__main__":
    main()
