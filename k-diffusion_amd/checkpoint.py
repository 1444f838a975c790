"""On-disk format of the sampling path: slim inference checkpoints.

The reference's sample.py reads a safetensors file whose tensors are the EMA weights and whose ``__metadata__`` carries
the model config as a JSON string under "config" (convert_for_inference.py:28-45 writes it, config.py:113-115 and
sample.py:33,44 read it).  These helpers write / read that format; the CLIs at the repository root
(convert_for_inference.py, config_from_inference.py) keep the reference's flags.
"""
import json
from pathlib import Path

import torch

from . import utils

DTYPES = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def write_inference_checkpoint(state_dict, config, path, dtype="fp16"):
    """state_dict (EMA weights) + config dict -> safetensors with the config in its metadata.  ``dtype`` is the storage
    type of the floating-point tensors (the HIP path computes in fp32 and upcasts on load)."""
    import safetensors.torch as safetorch
    if dtype not in DTYPES:
        raise ValueError(f"dtype must be one of {sorted(DTYPES)}")
    if config is None:
        raise ValueError("No configuration found in checkpoint and no override provided")
    out = {}
    for name, t in state_dict.items():
        t = t.detach().cpu()
        out[name] = (t.to(DTYPES[dtype]) if t.is_floating_point() else t).contiguous()
    safetorch.save_file(out, str(path), metadata={"config": json.dumps(config, indent=4)})
    return Path(path)


def convert_training_checkpoint(checkpoint, output=None, config_override=None, dtype="fp16", unsafe=False):
    """``.pth`` training checkpoint ({'config', 'model_ema', ...}, train.py:397-423) -> slim inference checkpoint.
    The file is read with ``weights_only=True`` (tensors, plain containers and scalars: everything the conversion needs); a
    checkpoint that only loads through the full unpickler (arbitrary code execution from an untrusted file) needs
    ``unsafe=True`` / ``--unsafe``."""
    checkpoint = Path(checkpoint)
    try:
        ckpt = torch.load(checkpoint, map_location="cpu", weights_only=True)
    except Exception as e:  # noqa: BLE001 -- pickle.UnpicklingError and friends
        if not unsafe:
            raise RuntimeError(f"{checkpoint} does not load with weights_only=True ({type(e).__name__}: {e}); "
                               "pass unsafe=True / --unsafe to unpickle it fully (only for files you trust)") from e
        ckpt = torch.load(checkpoint, map_location="cpu", weights_only=False)
    config = ckpt.get("config") if config_override is None else config_override
    weights = ckpt["model_ema"]
    del ckpt
    return write_inference_checkpoint(weights, config, output or checkpoint.with_suffix(".safetensors"), dtype)


def read_config(checkpoint):
    """The JSON config text stored in an inference checkpoint's metadata."""
    meta = utils.get_safetensors_metadata(checkpoint)
    if "config" not in meta:
        raise ValueError("No configuration found in checkpoint")
    return meta["config"]
