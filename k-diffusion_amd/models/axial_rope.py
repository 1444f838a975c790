"""Axial position grids and rotary tables for the HDiT denoiser (host side, computed once).

Stands in for k_diffusion/models/axial_rope.py:31-68 (``make_axial_pos`` chain) and
image_transformer_v2.py:52-54 (``downscale_pos``), :234-248 (``AxialRoPE``).  The reference
recomputes theta, cos and sin on the device for every attention block of every forward; here the
angles are evaluated once per (grid, freqs) on the CPU in fp32 -- with the same torch ops as the
reference's CPU path, so the tables are bit-identical to what it rotates with -- and uploaded.
"""
import math

import torch


def make_axial_pos(h, w, pixel_aspect_ratio=1.0, align_corners=False, dtype=None, device=None):
    """Cell-centre (or corner-aligned) coordinates of an h x w grid inside the [-1, 1]^2 bounding
    box that preserves the aspect ratio.  Returns [h * w, 2] = (y, x), row-major."""
    ratio = w / (h * pixel_aspect_ratio)
    ylim, xlim = (1.0, 1.0)
    if ratio > 1:
        ylim = 1 / ratio
    elif ratio < 1:
        xlim = ratio

    def axis(lim, n):
        if align_corners:
            return torch.linspace(-lim, lim, n, dtype=dtype, device=device)
        edges = torch.linspace(-lim, lim, n + 1, dtype=dtype, device=device)
        return (edges[:-1] + edges[1:]) / 2
    ys, xs = axis(ylim, h), axis(xlim, w)
    return torch.stack([ys[:, None].expand(h, w), xs[None, :].expand(h, w)], dim=-1).reshape(h * w, 2)


def downscale_pos(pos):
    """[..., 2h, 2w, 2] -> [..., h, w, 2]: mean over each 2x2 block, elements visited in the
    reference's (nh, nw) order so the fp32 sum rounds identically."""
    *b, H, W, e = pos.shape
    blocks = pos.reshape(*b, H // 2, 2, W // 2, 2, e).movedim(-4, -3).reshape(*b, H // 2, W // 2, 4, e)
    return torch.mean(blocks, dim=-2)


def rope_freqs(dim, n_heads):
    """AxialRoPE(dim, n_heads).freqs: [n_heads, dim // 4]; head h gets every n_heads-th entry of a
    log-spaced ladder from pi to 10*pi (exclusive)."""
    n = n_heads * dim // 4
    ladder = torch.linspace(math.log(math.pi), math.log(10.0 * math.pi), n + 1)[:-1].exp()
    return ladder.view(dim // 4, n_heads).T.contiguous()


def rope_tables(pos, freqs):
    """pos [h, w, 2], freqs [nh, F] (CPU fp32) -> (cos, sin) tables [h*w, nh, 2F]: angles are
    (y * freqs, x * freqs) concatenated (AxialRoPE.forward)."""
    pos = pos.to(torch.float32).cpu()
    freqs = freqs.to(torch.float32).cpu()
    th = pos[..., None, 0:1] * freqs
    tw = pos[..., None, 1:2] * freqs
    theta = torch.cat((th, tw), dim=-1).reshape(-1, freqs.shape[0], 2 * freqs.shape[1])
    return torch.cos(theta).contiguous(), torch.sin(theta).contiguous()
