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

DTYPES = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16, "fp8": torch.float8_e4m3fn}

# ---- fp8 weight storage (BASELINE.json configs[4]; the reference's converter stops at fp16 / bf16, convert_for_inference.py:23) --
# Every 2-D projection weight W [N, K] of the transformer (nn.Linear.weight: qkv / out / up / down projections, token merges and
# splits, patch-in / patch-out, the mapping network, the AdaRMSNorm linears) is stored as OCP e4m3 ("fn": no inf, max 448) with
# one POWER-OF-TWO scale per output channel n:
#       scale[n] = 2^ceil(log2(max_k |W[n, k]| / 448)),      q[n, k] = e4m3_rne(W[n, k] / scale[n]),      W8[n, k] = q[n, k] * scale[n]
# Because the scale is a power of two and e4m3 carries 4 significand bits, W8 is EXACTLY representable in bf16 (8 bits): the
# bf16 arithmetic mode (kd_pack_weight_bf16) runs on the fp8 weights bit for bit, with bf16 activations and fp32 accumulation --
# the product the matrix core would form from an fp8 operand, without quantising activations.  Embeddings, norm scales, Fourier
# features and every 1-D tensor stay fp32 (they are not GEMM operands).
FP8_MAX = 448.0
FP8_SCALE_SUFFIX = ".fp8_scale"


def is_fp8_weight(name, t):
    """Which tensors of an image_transformer_v2 state dict are stored as fp8: the 2-D ``.weight`` of linear layers."""
    return t.ndim == 2 and t.is_floating_point() and name.endswith(".weight") and "emb" not in name.rsplit(".", 2)[-2]


def quantize_fp8(w):
    """W [N, K] fp32 -> (q float8_e4m3fn [N, K], scale fp32 [N], a power of two per output channel)."""
    w = w.detach().to(torch.float32)
    amax = w.abs().amax(dim=1).clamp_min(2.0 ** -100)
    scale = torch.exp2(torch.ceil(torch.log2(amax / FP8_MAX)))
    # guard the ceil against log2 rounding: the scaled row must fit
    scale = torch.where(amax / scale > FP8_MAX, scale * 2, scale)
    q = (w / scale[:, None]).to(torch.float8_e4m3fn)
    return q, scale


def dequantize_fp8(q, scale):
    return q.to(torch.float32) * scale.to(torch.float32)[:, None]


def fake_quantize_fp8(w):
    """fp32 -> the fp32 value of its fp8 storage (what the oracle runs on to pin the fp8 path)."""
    return dequantize_fp8(*quantize_fp8(w))


def fp8_state_dict(state_dict):
    """Replace every projection weight by its fp8-stored value (fp32 tensors; bit-for-bit what a --dtype fp8 checkpoint loads to)."""
    return {k: (fake_quantize_fp8(v) if is_fp8_weight(k, v) else v) for k, v in state_dict.items()}


def write_inference_checkpoint(state_dict, config, path, dtype="fp16"):
    """state_dict (EMA weights) + config dict -> safetensors with the config in its metadata.  ``dtype`` is the storage
    type of the floating-point tensors (upcast on load); "fp8" stores the projection weights as e4m3 + per-channel scales."""
    import safetensors.torch as safetorch
    if dtype not in DTYPES:
        raise ValueError(f"dtype must be one of {sorted(DTYPES)}")
    if config is None:
        raise ValueError("No configuration found in checkpoint and no override provided")
    out = {}
    for name, t in state_dict.items():
        t = t.detach().cpu()
        if dtype == "fp8":
            if is_fp8_weight(name, t):
                out[name], out[name + FP8_SCALE_SUFFIX] = (x.contiguous() for x in quantize_fp8(t))
            else:
                out[name] = (t.to(torch.float32) if t.is_floating_point() else t).contiguous()
        else:
            out[name] = (t.to(DTYPES[dtype]) if t.is_floating_point() else t).contiguous()
    meta = {"config": json.dumps(config, indent=4)}
    if dtype == "fp8":
        meta["weight_format"] = "e4m3fn, power-of-two scale per output channel in <name>" + FP8_SCALE_SUFFIX
    safetorch.save_file(out, str(path), metadata=meta)
    return Path(path)


def load_inference_checkpoint(path):
    """safetensors inference checkpoint -> fp32-loadable state dict.  fp16 / bf16 tensors load as stored (load_state_dict
    upcasts); fp8 weights are expanded with their per-channel scales (exact in bf16, see above)."""
    import safetensors.torch as safetorch
    raw = safetorch.load_file(str(path))
    out = {}
    for name, t in raw.items():
        if name.endswith(FP8_SCALE_SUFFIX):
            continue
        if t.dtype == torch.float8_e4m3fn:
            if name + FP8_SCALE_SUFFIX not in raw:
                raise ValueError(f"{path}: fp8 tensor {name} has no {name + FP8_SCALE_SUFFIX}")
            out[name] = dequantize_fp8(t, raw[name + FP8_SCALE_SUFFIX])
        else:
            out[name] = t
    return out


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
