"""Shared definitions of the golden-vector cases.

``oracle/make_golden.py`` (run in the build container, where /root/reference exists) feeds these
inputs to the REAL reference and stores its outputs in tests/golden/*.safetensors; the tests
rebuild the same inputs from the same seeds and compare the oracle / the HIP path to the stored
outputs.  Nothing here touches /root/reference.
"""
import importlib
import json
import os

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
synth = importlib.import_module("k-diffusion_amd.synth")


def _v2(model, num_classes=0):
    cfg = {"model": {"type": "image_transformer_v2", **model}}
    if num_classes:
        cfg["dataset"] = {"num_classes": num_classes}
    return cfg


G, NA7, SW8 = {"type": "global", "d_head": 64}, {"type": "neighborhood", "d_head": 64, "kernel_size": 7}, \
    {"type": "shifted-window", "d_head": 64, "window_size": 8}

# name -> raw config (dict or path relative to the repo)
CONFIGS = {
    "tiny_global": _v2(dict(input_channels=3, input_size=[16, 16], patch_size=[2, 2], depths=[2], widths=[128],
                            self_attns=[G], sigma_data=0.5, sigma_min=1e-2, sigma_max=80)),
    "tiny_sw": _v2(dict(input_channels=3, input_size=[32, 32], patch_size=[2, 2], depths=[2, 1], widths=[64, 128],
                        self_attns=[SW8, G], sigma_data=0.5, sigma_min=1e-2, sigma_max=80), num_classes=10),
    "tiny_na": _v2(dict(input_channels=1, input_size=[32, 32], patch_size=[2, 2], depths=[1, 1, 1],
                        widths=[128, 128, 256], self_attns=[NA7, NA7, G], sigma_data=0.6, sigma_min=1e-2,
                        sigma_max=80)),
    "tiny_odd": _v2(dict(input_channels=1, input_size=[28, 28], patch_size=[4, 4], depths=[2], widths=[128],
                         self_attns=[G], sigma_data=0.6162, sigma_min=1e-2, sigma_max=80), num_classes=10),
    "mnist": "configs/config_mnist_transformer.json",
    "cifar": "configs/config_cifar10_transformer.json",
    "flowers_sw": "configs/config_oxford_flowers_shifted_window.json",
    "flowers_na": "configs/config_oxford_flowers.json",
}

WEIGHT_SEED = 1

# single-forward cases: (case name, config name, batch, sigmas)
FORWARD_CASES = [
    ("fwd_tiny_global", "tiny_global", 3, [0.05, 1.5, 60.0]),
    ("fwd_tiny_sw", "tiny_sw", 2, [0.3, 20.0]),
    ("fwd_tiny_na", "tiny_na", 2, [0.02, 4.0]),
    ("fwd_tiny_odd", "tiny_odd", 3, [0.1, 1.0, 79.0]),
    ("fwd_mnist", "mnist", 4, [0.01, 0.5, 5.0, 80.0]),
    ("fwd_cifar", "cifar", 2, [0.7, 33.0]),
    ("fwd_flowers_sw", "flowers_sw", 1, [2.5]),
    ("fwd_flowers_na", "flowers_na", 1, [2.5]),
]

# full sampling cases: (case, config, sampler, steps, batch)
SAMPLE_CASES = [
    ("smp_tiny_global_euler", "tiny_global", "sample_euler", 8, 2),
    ("smp_tiny_sw_heun", "tiny_sw", "sample_heun", 6, 2),
    ("smp_tiny_na_2m", "tiny_na", "sample_dpmpp_2m", 8, 2),
    ("smp_tiny_odd_lms", "tiny_odd", "sample_lms", 8, 2),
    ("smp_tiny_sw_sde", "tiny_sw", "sample_dpmpp_sde", 6, 2),
    ("smp_mnist_euler10", "mnist", "sample_euler", 10, 4),          # BASELINE config 1
    ("smp_cifar_heun50", "cifar", "sample_heun", 50, 2),            # BASELINE config 2 (batch reduced)
    ("smp_flowers_sw_2m50", "flowers_sw", "sample_dpmpp_2m", 50, 1),  # BASELINE config 3 (batch reduced)
    ("smp_flowers_na_2m50", "flowers_na", "sample_dpmpp_2m", 50, 1),  # BASELINE config 4 (batch reduced)
]


# round 2 ------------------------------------------------------------------------------------------------
# full-batch forwards of the two 256x256 configs (the batch at which the n-split / wide-panel kernels are selected); the golden
# file keeps the reference's outputs for the images B32_KEEP only (the reference is per-sample, the inputs are rebuilt from seeds)
FORWARD_B32_CASES = [
    ("fwd32_flowers_sw", "flowers_sw", 32),
    ("fwd32_flowers_na", "flowers_na", 32),
]
B32_KEEP = [0, 17, 31]


def b32_sigmas(batch):
    return [float(v) for v in torch.logspace(-1.7, 2.1, batch)]


# cases recorded under torch.autocast("cpu", torch.bfloat16) (the reference's reduced-precision mode) for the bf16 arithmetic mode
BF16_SAMPLE_CASES = ["smp_tiny_global_euler", "smp_tiny_sw_heun", "smp_tiny_na_2m", "smp_mnist_euler10", "smp_cifar_heun50",
                     "smp_flowers_sw_2m50", "smp_flowers_na_2m50"]


# round 3 ------------------------------------------------------------------------------------------------
# BASELINE configs[4]: flowers-NA, sample_dpmpp_sde x 50 with Brownian-tree noise (batch reduced to 1 for the CPU reference).
# Recorded by oracle/make_golden_r3.py from the reference's own sampler + BrownianTreeNoiseSampler over this package's tree stream.
SDE_FULL_CASE = ("smp_flowers_na_sde50", "flowers_na", "sample_dpmpp_sde", 50, 1)
SDE_SEED = 5
# 5-step DPM++2M at the full per-GPU batch (the batch the benchmark runs at); the golden file keeps images B32_KEEP
SAMPLE_B32_CASES = [
    ("smp32_flowers_na_2m5", "flowers_na", "sample_dpmpp_2m", 5, 32),
    ("smp32_flowers_sw_2m5", "flowers_sw", "sample_dpmpp_2m", 5, 32),
]


# round 5 ------------------------------------------------------------------------------------------------
# BASELINE configs[1] at its stated batch: CIFAR-10 config, sample_heun x 50, batch 64 (fp32 and the reference under autocast(bfloat16)).
# Recorded by oracle/make_golden_r5.py; the golden file keeps images B64_KEEP.
SAMPLE_B64_CASE = ("smp64_cifar_heun50", "cifar", "sample_heun", 50, 64)
B64_KEEP = [0, 13, 37, 63]


# round 6 ------------------------------------------------------------------------------------------------
# The headline at its own size: config_oxford_flowers.json, sample_dpmpp_2m x 50, batch 32 (fp32 and autocast(bfloat16)); images B32_KEEP kept.
# Recorded by oracle/make_golden_r6.py (samples_r6.safetensors; its metadata holds the reference's own wall time on the build container's cores).
HEADLINE_CASE = ("smp32_flowers_na_2m50", "flowers_na", "sample_dpmpp_2m", 50, 32)
# The second caller of the path (train.py:333-369): sample_dpmpp_2m_sde(make_cfg_model_fn(model), ..., eta=0, solver_type='heun') with
# classifier-free guidance on class-conditional models: (case, config, steps, batch).  samples_r6_demo.safetensors.
DEMO_CFG_SCALE = 3.0
DEMO_CASES = [
    ("demo_tiny_sw_cfg3_2msde_heun8", "tiny_sw", 8, 2),
    ("demo_cifar_cfg3_2msde_heun50", "cifar", 50, 4),
]
DEMO_FWD_SIGMAS = [2.0, 0.3]
# The fp8 arithmetic mode's goldens (forward_fp8.safetensors): the reference with the quantising hook of oracle/make_golden_r6.py on the inputs /
# weights of the projections the mode takes.  Same inputs as the fp32 cases of the same names.
FP8_FORWARD_CASES = ["fwd_cifar", "fwd_flowers_na"]
FP8_SAMPLE_CASES = ["smp_cifar_heun50"]
FP8_SDE = True            # + SDE_FULL_CASE (configs[4]) on the fp8-stored weights: "smp_flowers_na_sde50_fp8"


def sde_brownian_seeds(batch, seed=SDE_SEED):
    """One Brownian-tree seed per global image index: the rule of sample.py --seed (sample.brownian_seeds), restated here so that
    the golden script and the tests do not import the CLI."""
    return [((int(seed) * 0x9E3779B97F4A7C15) ^ (int(g) * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D)) & 0x7FFFFFFFFFFFFFFF for g in range(batch)]


def raw_config(name):
    c = CONFIGS[name]
    if isinstance(c, str):
        return json.loads(open(os.path.join(REPO, c)).read())
    return json.loads(json.dumps(c))


def num_classes_of(cfg):
    return cfg.get("dataset", {}).get("num_classes", 0)


def forward_inputs(cfg, batch, sigmas, seed=11):
    """x ~ N(0, 1 + sigma^2)-ish noised input, sigma, class ids for a single-forward case."""
    m = cfg["model"]
    g = torch.Generator().manual_seed(seed)
    sigma = torch.tensor(sigmas, dtype=torch.float32)
    x = torch.randn(batch, m["input_channels"], *m["input_size"], generator=g)
    x = x * (sigma.view(-1, 1, 1, 1) ** 2 + 0.25).sqrt()
    nc = num_classes_of(cfg)
    cls = (torch.arange(batch) * 3 + 1) % (nc + 1) if nc else None   # includes the "uncond" id nc
    return x, sigma, cls


def sample_inputs(cfg, batch, seed=5):
    """Initial noise (per global sample index) and class ids for a sampling case."""
    m = cfg["model"]
    shape = (m["input_channels"], *m["input_size"])
    x = torch.stack([synth.synth_noise(shape, seed, g, m["sigma_max"]) for g in range(batch)])
    nc = num_classes_of(cfg)
    cls = torch.arange(batch) % 10 if nc else None
    return x, cls


def recorded_noise(shape, n_queries, seed=77):
    """Deterministic unit-variance noise tensors injected through ``noise_sampler=`` so that the
    SDE solver arithmetic can be pinned without torchsde."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(shape, generator=g) for _ in range(n_queries)]
