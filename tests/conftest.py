import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "few_rows: runs with the few-rows latency kernel (csrc/gemm_x3s.hip) at its default threshold")


@pytest.fixture(scope="session")
def golden():
    from safetensors.torch import load_file
    import json
    from tests.golden import cases
    gd = cases.GOLDEN_DIR
    out = {"kat": json.load(open(os.path.join(gd, "kat.json")))}
    out["kat2"] = json.load(open(os.path.join(gd, "kat2.json")))
    for name in ("ops", "forward", "samples", "forward_b32", "forward_b32_bf16", "forward_bf16", "samples_bf16", "samples_r3", "samples_r4", "samples_r5", "samples_r6", "samples_r6_demo", "forward_fp8"):
        out[name] = load_file(os.path.join(gd, name + ".safetensors"))
    return out


@pytest.fixture(scope="session")
def KD():
    import k_diffusion_amd
    return k_diffusion_amd
