"""Test helpers shared by CPU and GPU tests."""
import struct

import torch


def state_dict_shapes(cfg):
    """Key -> shape map of ImageTransformerDenoiserModelV2.state_dict() for a merged config
    (the weight-format contract; image_transformer_v2.py:667-706)."""
    m = cfg["model"]
    nc = cfg.get("dataset", {}).get("num_classes", 0)
    widths, depths = m["widths"], m["depths"]
    d_ffs = m.get("d_ffs") or [3 * w for w in widths]
    specs = m["self_attns"]
    mw = m.get("mapping_width", 256)
    mdff = m.get("mapping_d_ff") or 3 * mw
    ph, pw = m["patch_size"]
    c = m["input_channels"]
    s = {"patch_in.proj.weight": (widths[0], c * ph * pw), "time_emb.weight": (mw // 2, 1),
         "time_in_proj.weight": (mw, mw), "aug_emb.weight": (mw // 2, 9), "aug_in_proj.weight": (mw, mw)}
    if nc:
        s["class_emb.weight"] = (nc + 1, mw)
    if m.get("mapping_cond_dim", 0):
        s["mapping_cond_in_proj.weight"] = (mw, m["mapping_cond_dim"])
    s["mapping.in_norm.scale"] = (mw,)
    for i in range(m.get("mapping_depth", 2)):
        s[f"mapping.blocks.{i}.norm.scale"] = (mw,)
        s[f"mapping.blocks.{i}.up_proj.weight"] = (2 * mdff, mw)
        s[f"mapping.blocks.{i}.down_proj.weight"] = (mw, mdff)
    s["mapping.out_norm.scale"] = (mw,)

    def layer(prefix, d, d_ff, spec):
        if spec["type"] != "none":
            nh = d // spec.get("d_head", 64)
            s[prefix + "self_attn.scale"] = (nh,)
            s[prefix + "self_attn.norm.linear.weight"] = (d, mw)
            s[prefix + "self_attn.qkv_proj.weight"] = (3 * d, d)
            s[prefix + "self_attn.pos_emb.freqs"] = (nh, spec.get("d_head", 64) // 8)
            s[prefix + "self_attn.out_proj.weight"] = (d, d)
        s[prefix + "ff.norm.linear.weight"] = (d, mw)
        s[prefix + "ff.up_proj.weight"] = (2 * d_ff, d)
        s[prefix + "ff.down_proj.weight"] = (d, d_ff)
    n = len(widths)
    for li in range(n - 1):
        for i in range(depths[li]):
            layer(f"down_levels.{li}.{i}.", widths[li], d_ffs[li], specs[li])
        for i in range(depths[li]):
            layer(f"up_levels.{li}.{i}.", widths[li], d_ffs[li], specs[li])
    for i in range(depths[-1]):
        layer(f"mid_level.{i}.", widths[-1], d_ffs[-1], specs[-1])
    for li in range(n - 1):
        s[f"merges.{li}.proj.weight"] = (widths[li + 1], 4 * widths[li])
        s[f"splits.{li}.proj.weight"] = (4 * widths[li], widths[li + 1])
        s[f"splits.{li}.fac"] = (1,)
    s["out_norm.scale"] = (widths[0],)
    s["patch_out.proj.weight"] = (c * ph * pw, widths[0])
    return s


def unhex(lst):
    return torch.tensor([struct.unpack(">f", bytes.fromhex(h))[0] for h in lst], dtype=torch.float32)


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int32)


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
