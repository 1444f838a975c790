"""Deterministic synthetic ("random-init") weights for ImageTransformerDenoiserModelV2.

The reference zero-initialises every residual-branch output projection, every AdaRMSNorm
projection and the final un-patch projection (image_transformer_v2.py:159, 365, 410, 458, 485,
558, 706), so a freshly constructed network outputs exactly zero and says nothing about the
attention / feed-forward kernels.  For benchmarking and parity work there is no network access
to real checkpoints either, so this module defines ONE recipe -- a function of the parameter
*name*, *shape* and a seed only -- that both the HIP model, the CPU oracle and the golden-vector
generator (which loads the result into the real reference model) use.

Every tensor is drawn on the CPU from its own ``torch.Generator`` seeded by
``crc32(name) ^ seed``, so the values do not depend on parameter order, device or world size.
"""
import math
import zlib

import torch

_RESIDUAL_OUT = ("out_proj.weight", "down_proj.weight")


def _gen(name, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFFFFFFFFFF)
    return g


def synth_tensor(name, shape, seed=0, template=None):
    """One synthetic fp32 tensor for parameter/buffer ``name``."""
    g = _gen(name, seed)
    rn = lambda: torch.randn(shape, generator=g, dtype=torch.float32)
    if name.endswith("pos_emb.freqs"):
        # deterministic buffer (image_transformer_v2.py:237-240): keep the model's own values
        if template is None:
            raise ValueError("pos_emb.freqs needs the constructed buffer as template")
        return template.detach().to(torch.float32).clone()
    if name in ("time_emb.weight", "aug_emb.weight"):       # FourierFeatures buffers, std 1
        return rn()
    if name == "class_emb.weight":                           # nn.Embedding default N(0, 1)
        return rn()
    if name.endswith("self_attn.scale"):                     # per-head cosine-sim scale, > 0
        return 10.0 * torch.exp(0.2 * rn())
    if name.endswith(".scale"):                              # RMSNorm gains
        return 1.0 + 0.1 * rn()
    if name.endswith(".fac"):                                # TokenSplit lerp factor
        return 0.5 + 0.1 * rn()
    if name.endswith("norm.linear.weight"):                  # AdaRMSNorm cond projection (zero-init in ref)
        return 0.03 * rn()
    if len(shape) == 2:
        fan_in = shape[1]
        gain = 0.5 if name.endswith(_RESIDUAL_OUT) else 1.0
        return rn() * (gain / math.sqrt(fan_in))
    raise ValueError(f"no synthetic-weight rule for {name} {tuple(shape)}")


def synth_state_dict(template_state_dict, seed=0):
    """Synthetic state dict with the same keys / shapes as ``template_state_dict``."""
    return {k: synth_tensor(k, tuple(v.shape), seed, template=v) for k, v in template_state_dict.items()}


def synth_noise(shape_chw, seed, global_index, sigma_max):
    """Initial noise for the sample with global index ``global_index`` (independent of the GPU
    count, unlike sample.py:59's rank-local torch.randn): randn(C,H,W) * sigma_max on the CPU."""
    g = torch.Generator(device="cpu")
    g.manual_seed(((seed << 32) + global_index) & 0x7FFFFFFFFFFFFFFF)
    return torch.randn(shape_chw, generator=g, dtype=torch.float32) * sigma_max


def synth_noise_batch(shape_chw, seed, first_index, count, sigma_max):
    """[count, C, H, W]: ``synth_noise`` of the global indices first_index .. first_index + count - 1 (one generator per image, so
    a sample's noise does not depend on the batch / rank it is drawn in), written into one pinned-size buffer."""
    out = torch.empty((count, *shape_chw), dtype=torch.float32)
    g = torch.Generator(device="cpu")
    for i in range(count):
        g.manual_seed(((seed << 32) + first_index + i) & 0x7FFFFFFFFFFFFFFF)
        torch.randn(shape_chw, generator=g, dtype=torch.float32, out=out[i])
    return out.mul_(sigma_max)
