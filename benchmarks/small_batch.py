"""Launch-bound regime: full sampling runs at small batch sizes, launch lists issued directly (KDIFF_GRAPH=0) against hipGraph
replay of the main and conditioning chains (KDIFF_GRAPH=1), through the public sampler API.  One JSON line per case.

    python benchmarks/small_batch.py [config batch sampler steps]...   (default: the cases below)
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KDIFF_GEMM", "bf16")
import k_diffusion_amd as K  # noqa: E402

CASES = [("configs/config_oxford_flowers.json", 1, "sample_dpmpp_2m", 50),
         ("configs/config_oxford_flowers.json", 2, "sample_dpmpp_2m", 50),
         ("configs/config_oxford_flowers.json", 4, "sample_dpmpp_2m", 50),
         ("configs/config_oxford_flowers.json", 8, "sample_dpmpp_2m", 50),
         ("configs/config_oxford_flowers_shifted_window.json", 1, "sample_dpmpp_2m", 50),
         ("configs/config_cifar10_transformer.json", 64, "sample_heun", 50),
         ("configs/config_cifar10_transformer.json", 256, "sample_heun", 50),
         ("configs/config_mnist_transformer.json", 4, "sample_euler", 10),
         ("configs/config_mnist_transformer.json", 64, "sample_euler", 10)]


def run(cfg_path, batch, sampler, steps, reps=3):
    cfg = K.config.load_config(cfg_path)
    mc = cfg["model"]
    model = K.config.make_model(cfg).eval().requires_grad_(False)
    model.load_state_dict(K.synth.synth_state_dict(model.state_dict(), seed=0))
    den = K.Denoiser(model.to("cuda"), sigma_data=mc["sigma_data"])
    size = mc["input_size"]
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(batch, mc["input_channels"], size[0], size[1], generator=g) * mc["sigma_max"]).to("cuda")
    extra = {}
    if cfg["dataset"].get("num_classes", 0):
        extra["class_cond"] = (torch.arange(batch) % cfg["dataset"]["num_classes"]).to("cuda")
    sigmas = K.sampling.get_sigmas_karras(steps, mc["sigma_min"], mc["sigma_max"], device="cuda")
    fn = getattr(K.sampling, sampler)
    res, outs = {}, {}
    for mode in ("0", "1"):
        os.environ["KDIFF_GRAPH"] = mode
        for _ in range(2):
            outs[mode] = fn(den, x, sigmas, extra_args=extra, disable=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn(den, x, sigmas, extra_args=extra, disable=True)
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t) / reps
    nfe = steps * 2 - 1 if sampler == "sample_heun" else steps
    return {"config": os.path.basename(cfg_path), "batch": batch, "sampler": sampler, "steps": steps,
            "direct_ms_per_run": round(res["0"] * 1e3, 2), "graph_ms_per_run": round(res["1"] * 1e3, 2),
            "direct_ms_per_forward": round(res["0"] * 1e3 / nfe, 4), "graph_ms_per_forward": round(res["1"] * 1e3 / nfe, 4),
            "direct_images_per_s": round(batch / res["0"], 2), "graph_images_per_s": round(batch / res["1"], 2),
            "tokens_level0": batch * (size[0] // mc["patch_size"][0]) * (size[1] // mc["patch_size"][1]),
            "bit_identical": bool(torch.equal(outs["0"], outs["1"]))}


def main():
    args = sys.argv[1:]
    cases = [(args[i], int(args[i + 1]), args[i + 2], int(args[i + 3])) for i in range(0, len(args), 4)] if args else CASES
    for c in cases:
        try:
            print(json.dumps(run(*c)), flush=True)
        except Exception as e:  # keep going: one failing case must not hide the others
            print(json.dumps({"config": c[0], "batch": c[1], "error": repr(e)}), flush=True)


if __name__ == "__main__":
    main()
