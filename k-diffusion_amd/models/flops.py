"""Analytic cost model of one ImageTransformerDenoiserModelV2 forward.

Stands in for k_diffusion/models/flops.py (a hook-based counter that needs a live forward): the same convention
(:40-54: multiply-accumulates, no x2; Linear / attention / natten only), evaluated from the model config alone.
Used by bench.py to report algorithmic TFLOP/s next to the executed-MFMA roofline.
"""


def forward_cost_mac(mcfg, image_hw=None):
    """Multiply-accumulates per image for one forward, in the reference's own convention
    (k_diffusion/models/flops.py:40-54: MACs, no x2; Linear / attention / natten only)."""
    widths, depths, d_ffs, specs = mcfg["widths"], mcfg["depths"], mcfg["d_ffs"], mcfg["self_attns"]
    ph, pw = mcfg["patch_size"]
    H, W = image_hw if image_hw is not None else mcfg["input_size"]
    c = mcfg["input_channels"]
    mw = mcfg.get("mapping_width", 256)
    mdff = mcfg.get("mapping_d_ff") or 3 * mw
    h, w = H // ph, W // pw
    total = {"linear": 0, "attn": 0}
    total["linear"] += h * w * (c * ph * pw) * widths[0]                   # patch_in
    total["linear"] += h * w * widths[0] * (c * ph * pw)                   # patch_out
    total["linear"] += 2 * mw * mw                                         # time_in_proj, aug_in_proj
    total["linear"] += mcfg.get("mapping_depth", 2) * (mw * 2 * mdff + mdff * mw)

    def layer_cost(t_h, t_w, d, d_ff, spec):
        t = t_h * t_w
        lin = mw * d * 2                                                   # two AdaRMSNorm projections
        lin += t * d * d_ff * 2 + t * d_ff * d                             # GEGLU up (2*d_ff) + down
        att = 0
        if spec["type"] != "none":
            lin += t * d * 3 * d + t * d * d
            nh, e = d // spec.get("d_head", 64), spec.get("d_head", 64)
            if spec["type"] == "global":
                att = nh * t * t * 2 * e
            elif spec["type"] == "neighborhood":
                att = t * nh * 2 * e * spec.get("kernel_size", 7) ** 2
            else:
                ws = spec["window_size"]
                att = nh * (t // (ws * ws)) * (ws * ws) ** 2 * 2 * e
        else:
            lin -= mw * d
        return lin, att
    for li, (d, depth, d_ff, spec) in enumerate(zip(widths, depths, d_ffs, specs)):
        reps = depth * (2 if li < len(widths) - 1 else 1)
        lin, att = layer_cost(h, w, d, d_ff, spec)
        total["linear"] += reps * lin
        total["attn"] += reps * att
        if li < len(widths) - 1:
            total["linear"] += (h // 2) * (w // 2) * 4 * d * widths[li + 1]      # merge
            total["linear"] += (h // 2) * (w // 2) * widths[li + 1] * 4 * d      # split
            h, w = h // 2, w // 2
    total["total"] = total["linear"] + total["attn"]
    return total
