"""The three attention operators the reference's model calls into third-party libraries for, under THEIR names, signatures and tensor
layouts, on the HIP attention cores of this package -- the operator-level seam of SURVEY.md section 8b:

    natten.functional.na2d(q, k, v, kernel_size, scale=1.0)            image_transformer_v2.py:428     q, k, v [n, h, w, nh, e]
    flash_attn.flash_attn_qkvpacked_func(qkv, softmax_scale=1.0)       image_transformer_v2.py:383     qkv [n, s, 3, nh, e]
    F.scaled_dot_product_attention(q, k, v, scale=1.0)                 image_transformer_v2.py:392     q, k, v [n, nh, s, e]

A maintainer of the reference who wants only the attention cores swaps the import; the model of this package does not
come through here -- its qkv projection writes the packed operand the cores read, with q and k already scaled and rotated.  What these
wrappers add around a core is layout plumbing on the device (one concatenation into the packed [.., 3 * nh * e] operand, the softmax scale
folded into q); the arithmetic is the core's: fp32 tensors take the fp32-parity cores (``KDIFF_GEMM`` = split3 / exact decides which),
bf16 tensors the bf16 cores.  Limits of the cores, stated as errors: head dimension 64, dilation 1, no mask, no dropout, not causal.
"""
import torch

from . import _native as nat
from . import ops

D_HEAD = 64


def _split_stored(x):
    """fp32 [..., C] -> the operand format of the split-bf16x3 attention cores (KdGemm.qkv_packed, include/kdiff_hip.h): every 4 values' 16
    bytes hold [hi: 4 x bf16][lo: 4 x bf16], hi = bf16(x), lo = bf16(x - hi).  In the model the qkv projection's epilogue writes this; here it
    is three elementwise torch ops on the device."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.to(torch.float32)).to(torch.bfloat16)
    both = torch.cat([hi.view(*x.shape[:-1], -1, 4), lo.view(*x.shape[:-1], -1, 4)], dim=-1)        # [..., C / 4, 8] bf16
    return both.view(torch.float32).view(x.shape)


def _check(q, what):
    if q.shape[-1] != D_HEAD:
        raise NotImplementedError(f"{what}: the HIP attention cores take head dimension {D_HEAD} (got {q.shape[-1]})")
    if q.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"{what}: float32 or bfloat16 tensors (got {q.dtype})")
    if not q.is_cuda:
        raise RuntimeError(f"{what}: the HIP path needs tensors on a ROCm device (got {q.device}); there is no CPU fallback")


def _pack(q, k, v, scale, default_scale):
    """q, k, v [..., nh, e] -> the cores' packed operand [..., 3 * nh * e] (all q heads, all k heads, all v heads), q times the softmax scale."""
    scale = default_scale if scale is None else float(scale)
    if scale != 1.0:
        q = q * scale
    return torch.cat([q.flatten(-2), k.flatten(-2), v.flatten(-2)], dim=-1)


def na2d(q, k, v, kernel_size, dilation=1, scale=None):
    """``natten.functional.na2d``: fused 2-D neighbourhood attention, q / k / v ``[n, h, w, heads, 64]`` -> ``[n, h, w, heads, 64]``;
    every query attends the ``kernel_size`` x ``kernel_size`` keys of its clamped window; ``scale`` defaults to NATTEN's ``64 ** -0.5``
    (the reference passes 1.0: its q and k are cosine-sim scaled).  Odd kernel sizes 3 .. 13."""
    _check(q, "na2d")
    if q.dim() != 5 or q.shape != k.shape or q.shape != v.shape:
        raise ValueError(f"na2d: q, k, v must share the shape [n, h, w, heads, {D_HEAD}] (got {tuple(q.shape)}, {tuple(k.shape)}, {tuple(v.shape)})")
    if isinstance(kernel_size, (tuple, list)):
        if len(set(kernel_size)) != 1:
            raise NotImplementedError("na2d: square neighbourhoods only")
        kernel_size = kernel_size[0]
    if dilation not in (1, (1, 1), [1, 1]):
        raise NotImplementedError("na2d: dilation 1 only")
    n, h, w, nh, e = q.shape
    packed = _pack(q, k, v, scale, D_HEAD ** -0.5)
    if q.dtype == torch.float32 and ops._prec_of(packed) == nat.PREC_SPLIT3:
        # the fp32-parity mode's neighbourhood core (every kernel size) reads split-stored operands
        return ops.attn_na2d(_split_stored(packed), nh, int(kernel_size), prep="packed").view(n, h, w, nh, e)
    if q.dtype == torch.float32 and int(kernel_size) != 7:
        raise NotImplementedError("na2d: the exact-fp32 neighbourhood core takes kernel_size 7 only (KDIFF_GEMM=split3 and bf16 tensors: 3 .. 13)")
    return ops.attn_na2d(packed, nh, int(kernel_size)).view(n, h, w, nh, e)


def flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False):
    """``flash_attn.flash_attn_qkvpacked_func``: dense attention over the sequence, qkv ``[n, s, 3, heads, 64]`` -> ``[n, s, heads, 64]``;
    ``softmax_scale`` defaults to ``64 ** -0.5`` (the reference passes 1.0)."""
    _check(qkv, "flash_attn_qkvpacked_func")
    if qkv.dim() != 5 or qkv.shape[2] != 3:
        raise ValueError(f"flash_attn_qkvpacked_func: qkv must be [n, s, 3, heads, {D_HEAD}] (got {tuple(qkv.shape)})")
    if dropout_p or causal:
        raise NotImplementedError("flash_attn_qkvpacked_func: no dropout, not causal (the sampling path uses neither)")
    n, s, _, nh, e = qkv.shape
    scale = D_HEAD ** -0.5 if softmax_scale is None else float(softmax_scale)
    if scale == 1.0:
        packed = qkv.reshape(n, s, 3 * nh * e)                   # already the cores' operand: "(t nh e)" with t outermost
    else:
        packed = _pack(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], scale, 1.0)
    return ops.attn_global(packed.contiguous(), nh).view(n, s, nh, e)


def scaled_dot_product_attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None):
    """``torch.nn.functional.scaled_dot_product_attention`` for the global-attention call site: q / k / v ``[n, heads, s, 64]`` ->
    ``[n, heads, s, 64]``; ``scale`` defaults to ``1 / sqrt(64)`` (the reference passes 1.0).  The masked call of the shifted-window block
    (image_transformer_v2.py:333) is not this operator's job here: that block's windowing, roll and mask are inside ``ops.attn_window``."""
    _check(query, "scaled_dot_product_attention")
    if attn_mask is not None or dropout_p or is_causal:
        raise NotImplementedError("scaled_dot_product_attention: no mask, no dropout, not causal (shifted windows: ops.attn_window)")
    if query.dim() != 4 or query.shape != key.shape or query.shape != value.shape:
        raise ValueError(f"scaled_dot_product_attention: q, k, v must share the shape [n, heads, s, {D_HEAD}]")
    n, nh, s, e = query.shape
    packed = _pack(query.transpose(1, 2), key.transpose(1, 2), value.transpose(1, 2), scale, D_HEAD ** -0.5)
    return ops.attn_global(packed, nh).view(n, s, nh, e).transpose(1, 2)
