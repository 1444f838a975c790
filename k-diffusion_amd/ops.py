"""Tensor-level wrappers over the C ABI (include/kdiff_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation below is a HIP
kernel from libkdiff_hip.so launched on ``torch.cuda.current_stream()``.  Tensors must be contiguous,
on a ROCm device and fp32 -- or, for the activation tensors of the bf16 arithmetic mode (``precision=PREC_BF16`` /
``KDIFF_GEMM=bf16``), bf16 -- anything else raises (there is no eager fallback).

The op names mirror the reference functions they replace
(k_diffusion/models/image_transformer_v2.py): ``rms_norm`` (:98), ``linear_geglu`` (:89),
``scale_for_cosine_sim``+``apply_rotary_emb_`` -> ``qk_prep_`` (:106, :230), SDPA / flash-attn ->
``attn_global`` (:383,:392), ``apply_window_attention`` -> ``attn_window`` (:319),
``natten.functional.na2d`` -> ``attn_na2d`` (:428).
"""
import ctypes as C
import os
import weakref

import math

import torch

from . import _native as nat


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the HIP path needs a tensor on a ROCm device (got {t.device}); there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_packed = {}


def pack_weight(W, N, K, geglu, cache=True, bf16=False):
    """Packed image of a weight (uint8 tensor): the split-bf16 image of KD_PREC_SPLIT3, or with ``bf16=True`` the plain bf16
    image of KD_PREC_BF16, or with ``bf16="mx8"`` the e4m3 image + channel scales of kd_gemm_mx8 (``geglu`` 0 / 1 only).  ``geglu``: 0 / False plain, 1 / True GEGLU rows, 2 the k order of the fused FF block's down projection.  Weights are static while sampling, so the image is cached per tensor OBJECT (weak reference +
    version counter: a new tensor that happens to reuse the address of a freed one never hits a stale image)."""
    geglu = int(geglu)
    cache = cache and not W.is_inference()        # (no version counter to tell a rewritten tensor by: never cached)
    key = (id(W), bf16 if bf16 == "mx8" else bool(bf16), geglu)
    ent = _packed.get(key) if cache else None
    if ent is not None:
        ref, version, meta, img = ent
        if ref() is W and version == W._version and meta == (tuple(W.shape), N, K, geglu, W.data_ptr()):
            return img
    _chk(W, "W")
    lib = nat.lib()
    if bf16 == "mx8":
        size = lib.kd_packed_weight_bytes_mx8(N, K, int(geglu))
        if size <= 0:
            raise ValueError(f"kd_pack_weight_mx8: N={N} K={K} is not taken (K must be a multiple of 128)")
        img = torch.empty(size, device=W.device, dtype=torch.uint8)
        nat.check(lib.kd_pack_weight_mx8(_p(W), _p(img), N, K, int(geglu), _stream()), "kd_pack_weight_mx8")
    elif bf16:
        img = torch.empty(lib.kd_packed_weight_bytes_bf16(N, K, int(geglu)), device=W.device, dtype=torch.uint8)
        nat.check(lib.kd_pack_weight_bf16(_p(W), _p(img), N, K, int(geglu), _stream()), "kd_pack_weight_bf16")
    else:
        img = torch.empty(lib.kd_packed_weight_bytes(N, K, int(geglu)), device=W.device, dtype=torch.uint8)
        nat.check(lib.kd_pack_weight_bf16x3(_p(W), _p(img), N, K, int(geglu), _stream()), "kd_pack_weight_bf16x3")
    if cache:
        if len(_packed) > 512:
            for k in [k for k, e in _packed.items() if e[0]() is None]:
                del _packed[k]
            if len(_packed) > 512:
                _packed.clear()
        def gone(ref, key=key):               # the weight died: its image goes with it (plans pack per-plan concatenations of weights)
            ent = _packed.get(key)
            if ent is not None and ent[0] is ref:
                del _packed[key]
        _packed[key] = (weakref.ref(W, gone), W._version, (tuple(W.shape), N, K, geglu, W.data_ptr()), img)
    return img


def gemm(A, W, out, *, M, N, K, a_mode=nat.A_PLAIN, epi=nat.EPI_STORE, norm_scale=None, scale_stride=0,
         rows_per_sample=0, residual=None, grid=(0, 0), patch=(0, 0, 0), eps=1e-6, out_add=0.0,
         sigma=None, sigma_data=1.0, fac=None, scale_ptr=None, precision=None, qk=None, qkv_packed=False, per_row=False,
         a_planes=None, c_planes=None, launch=True, mx8=False):
    """Fused GEMM (see KdGemm in include/kdiff_hip.h).  ``norm_scale`` may be a tensor or, with
    ``scale_ptr``, a raw device address inside a larger scale table.  ``precision``: nat.PREC_EXACT /
    nat.PREC_SPLIT3 / nat.PREC_BF16 (default: KDIFF_GEMM env, split3).  In bf16 mode A, out and residual are bf16 tensors
    (except the fp32 image side of the patch modes) and ``qk`` = (scale_h, rope_pos [T, 2], rope_freq [nh, 8], nh).
    ``per_row`` (fp32 modes): the per-row FMA kernel of the conditioning chain whatever M (KdGemm.per_row).
    ``a_planes`` / ``c_planes`` (split3): (hi, lo) bf16 tensors instead of the fp32 ``A`` / ``out`` (KdGemm.a_split / c_split; ``A`` / ``out``
    are then ignored and may be None).  ``launch=False``: only build and return the descriptor (for entry points that take one, e.g.
    kd_attn_block_bf16).  ``mx8`` (bf16 tensors, norm -> store / qkv / GEGLU at K = 256 / 512): the product on the block-scaled fp8 matrix
    instruction (kd_gemm_mx8: e4m3 weights with power-of-two channel scales, activations quantised per 32-k block)."""
    d = nat.KdGemm()
    d.per_row = 1 if per_row else 0
    d.precision = nat.kernel_precision() if precision is None else precision
    bf = d.precision == nat.PREC_BF16
    act = torch.bfloat16 if bf else torch.float32
    a_dt = torch.float32 if a_mode == nat.A_PATCH_NCHW else act
    c_dt = torch.float32 if epi == nat.EPI_UNPATCH_NCHW else act
    if mx8:
        if not bf:
            raise TypeError("mx8: bf16 activations only (precision=PREC_BF16)")
        d.Wp = pack_weight(W, N, K, epi == nat.EPI_GEGLU, bf16="mx8").data_ptr()
    elif bf:
        d.Wp = pack_weight(W, N, K, epi == nat.EPI_GEGLU, bf16=True).data_ptr()
    elif d.precision == nat.PREC_SPLIT3:
        d.Wp = pack_weight(W, N, K, epi == nat.EPI_GEGLU).data_ptr()
    d.M, d.N, d.K = M, N, K
    d.a_mode, d.epi = a_mode, epi
    d.norm = 1 if (norm_scale is not None or scale_ptr is not None) else 0
    d.rows_per_sample, d.scale_stride = rows_per_sample, scale_stride
    d.gh, d.gw = grid
    d.ph, d.pw, d.chan = patch
    d.eps, d.out_add, d.sigma_data = eps, out_add, sigma_data
    d.W = _chk(W, "W").data_ptr()
    if a_planes is not None and mx8:          # (e4m3 rows [M, K], E8M0 block-scale bytes [M, K / 32]): kd_gemm_mx8's tiled form
        d.a_split, d.A, d.A_lo = 1, _chk(a_planes[0], "A e4m3", torch.uint8).data_ptr(), _chk(a_planes[1], "A scales", torch.uint8).data_ptr()
    elif a_planes is not None:
        d.a_split, d.A, d.A_lo = 1, _chk(a_planes[0], "A hi", torch.bfloat16).data_ptr(), _chk(a_planes[1], "A lo", torch.bfloat16).data_ptr()
    else:
        d.A = _chk(A, "A", a_dt).data_ptr()
    if c_planes is not None and mx8:          # the GEGLU result as (e4m3 rows [M, N], scale bytes [M, N / 32])
        d.c_split, d.C, d.C_lo = 1, _chk(c_planes[0], "C e4m3", torch.uint8).data_ptr(), _chk(c_planes[1], "C scales", torch.uint8).data_ptr()
        out = c_planes
    elif c_planes is not None:
        d.c_split, d.C, d.C_lo = 1, _chk(c_planes[0], "C hi", torch.bfloat16).data_ptr(), _chk(c_planes[1], "C lo", torch.bfloat16).data_ptr()
        out = c_planes
    else:
        d.C = _chk(out, "C", c_dt).data_ptr()
    d.R = None if residual is None else _chk(residual, "R", c_dt).data_ptr()
    d.scale = scale_ptr if scale_ptr is not None else (None if norm_scale is None else _chk(norm_scale, "scale").data_ptr())
    d.sigma = None if sigma is None else _chk(sigma, "sigma").data_ptr()
    d.fac = None if fac is None else _chk(fac, "fac").data_ptr()
    if qk is not None and bf:   # EPI_QKV, bf16 mode: (scale_h [nh], rope_pos [T, 2], rope_freq [nh, 8] in revolutions, nh)
        d.qk_scale, d.rope_pos, d.rope_freq = (_chk(t, n).data_ptr() for t, n in zip(qk[:3], ("qk_scale", "rope_pos", "rope_freq")))
        d.n_heads = qk[3]
    elif qk is not None:        # EPI_QKV: (scale_h [nh], cos [T, nh, 16], sin [T, nh, 16], nh)
        d.qk_scale, d.rope_cos, d.rope_sin = (_chk(t, n).data_ptr() for t, n in zip(qk[:3], ("qk_scale", "cos", "sin")))
        d.n_heads = qk[3]
        if len(qk) >= 6:        # + (rope_pos [T, 2], rope_freq [nh, 8] in revolutions): the round-3 split3 kernel evaluates the angles itself
            d.rope_pos, d.rope_freq = _chk(qk[4], "rope_pos").data_ptr(), _chk(qk[5], "rope_freq").data_ptr()
        d.qkv_packed = 1 if qkv_packed else 0      # q, k, v stored as split-bf16 chunks for the attention cores (prep="packed")
    if not launch:
        return d
    if mx8:
        nat.check(nat.lib().kd_gemm_mx8(C.byref(d), _stream()), "kd_gemm_mx8")
    elif bf:
        nat.check(nat.lib().kd_gemm_bf16(C.byref(d), _stream()), "kd_gemm_bf16")
    else:
        nat.check(nat.lib().kd_gemm_f32(C.byref(d), _stream()), "kd_gemm_f32")
    return out


def _prec_of(x):
    """Arithmetic mode implied by an activation tensor: bf16 tensors run the bf16 kernels, fp32 ones the KDIFF_GEMM fp32 mode."""
    if x.dtype == torch.bfloat16:
        return nat.PREC_BF16
    p = nat.kernel_precision()
    return nat.PREC_SPLIT3 if p == nat.PREC_BF16 else p


def linear(x, weight, residual=None, out=None, out_add=0.0):
    """x[..., K] @ weight[N, K]^T (+ residual)  --  Linear (:126-129) with the skip add fused."""
    K = x.shape[-1]
    M = x.numel() // K
    Nn = weight.shape[0]
    out = torch.empty(*x.shape[:-1], Nn, device=x.device, dtype=x.dtype) if out is None else out
    return gemm(x, weight, out, M=M, N=Nn, K=K, epi=nat.EPI_RESIDUAL if residual is not None else nat.EPI_STORE,
                residual=residual, out_add=out_add, precision=_prec_of(x))


def linear_geglu(x, weight, out=None):
    """linear_geglu (:89-95): value * gelu(gate), value = first half of the output features."""
    K = x.shape[-1]
    M = x.numel() // K
    d_ff = weight.shape[0] // 2
    out = torch.empty(*x.shape[:-1], d_ff, device=x.device, dtype=x.dtype) if out is None else out
    return gemm(x, weight, out, M=M, N=d_ff, K=K, epi=nat.EPI_GEGLU, precision=_prec_of(x))


def norm_split(x, scale=None, *, rows_per_sample=0, eps=1e-6):
    """AdaRMSNorm / RMSNorm (:98-103, :155-166) of fp32 rows -> (hi, lo) bf16 planes, the A operand of an ``a_planes`` GEMM.
    ``scale``: [B, K] per-sample, [K] shared, or None (plain split of ``x``)."""
    K = x.shape[-1]
    M = x.numel() // K
    hi, lo = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16), torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    stride = 0 if scale is None or scale.dim() == 1 else K
    nat.check(nat.lib().kd_norm_split_f32(_p(_chk(x, "x")), None if scale is None else _p(_chk(scale, "scale")), stride, rows_per_sample or M,
                                          _p(hi), _p(lo), M, K, eps, _stream()), "kd_norm_split_f32")
    return hi, lo


def rms_norm(x, scale, eps=1e-6, out=None):
    """rms_norm (:98-103) with a shared gain (RMSNorm :142-152)."""
    d = x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    nat.check(nat.lib().kd_rmsnorm_f32(_p(_chk(x, "x")), _p(_chk(scale, "scale")), _p(_chk(out, "y")), x.numel() // d, d, eps, _stream()),
            "kd_rmsnorm_f32")
    return out


def norm_linear(x, scale, weight, *, rows_per_sample, epi=nat.EPI_STORE, out=None, eps=1e-6, qk=None, qkv_packed=False, mx8=False, c_fp8=False):
    """AdaRMSNorm/RMSNorm (:155-166) fused into the following Linear / LinearGEGLU.
    ``scale``: [B, K] per-sample scales (AdaRMSNorm: Linear(cond) + 1) or [K] shared gain.
    ``epi=EPI_QKV`` with ``qk=(scale_h, cos, sin, nh)``: qkv projection whose q, k come out prepared
    (scale_for_cosine_sim + apply_rotary_emb_, :106-121, :187-231), ready for the attention cores with prep=None."""
    K = x.shape[-1]
    M = x.numel() // K
    Nn = weight.shape[0] // (2 if epi == nat.EPI_GEGLU else 1)
    if c_fp8:            # mx8 + GEGLU: the result as (e4m3 rows [..., N] uint8, E8M0 scale bytes [..., N / 32] uint8) -- linear_mx8's A operand
        planes = (torch.empty(*x.shape[:-1], Nn, device=x.device, dtype=torch.uint8), torch.empty(*x.shape[:-1], Nn // 32, device=x.device, dtype=torch.uint8))
        return gemm(x, weight, None, M=M, N=Nn, K=K, epi=epi, norm_scale=scale, scale_stride=K if scale.dim() == 2 else 0,
                    rows_per_sample=rows_per_sample, eps=eps, precision=_prec_of(x), mx8=True, c_planes=planes)
    out = torch.empty(*x.shape[:-1], Nn, device=x.device, dtype=x.dtype) if out is None else out
    return gemm(x, weight, out, M=M, N=Nn, K=K, epi=epi, norm_scale=scale, scale_stride=K if scale.dim() == 2 else 0,
                rows_per_sample=rows_per_sample, eps=eps, qk=qk, qkv_packed=qkv_packed, precision=_prec_of(x), mx8=mx8)


def linear_mx8(a8, a_scales, weight, residual=None, out=None):
    """(e4m3 rows [..., K] uint8, E8M0 block-scale bytes [..., K / 32] uint8) @ e4m3(weight[N, K])^T (+ residual, bf16) -> bf16 [..., N]:
    kd_gemm_mx8's tiled form (the fp8 mode's down projection; both operands e4m3 on the block-scaled matrix instruction)."""
    K = a8.shape[-1]
    M = a8.numel() // K
    Nn = weight.shape[0]
    out = torch.empty(*a8.shape[:-1], Nn, device=a8.device, dtype=torch.bfloat16) if out is None else out
    return gemm(None, weight, out, M=M, N=Nn, K=K, epi=nat.EPI_RESIDUAL if residual is not None else nat.EPI_STORE, residual=residual,
                precision=nat.PREC_BF16, mx8=True, a_planes=(a8, a_scales))


def attn_block(x, scale, weight, *, rows_per_sample, qk, out=None, eps=1e-6):
    """The global-attention block as one launch (kd_attn_block_bf16; image_transformer_v2.py:370-392): x [B, T, K] bf16, ``scale`` [B, K]
    AdaRMSNorm scales, ``weight`` the qkv projection [3 K, K], ``qk`` = (scale_h [nh], rope_pos [T, 2], rope_freq [nh, 8] in revolutions, nh)
    -> attention output [B, T, K] bf16 (``out``).  256 tokens per sample, K = 64 nh in {256, 512}."""
    K = x.shape[-1]
    M = x.numel() // K
    out = torch.empty_like(x) if out is None else out
    d = gemm(x, weight, out, M=M, N=3 * K, K=K, epi=nat.EPI_QKV, norm_scale=scale, scale_stride=K, rows_per_sample=rows_per_sample, eps=eps,
             qk=qk, precision=nat.PREC_BF16, launch=False)
    nat.check(nat.lib().kd_attn_block_bf16(C.byref(d), _stream()), "kd_attn_block_bf16")
    return out


def proj_block(x, scale, weight, *, rows_per_sample, epi=nat.EPI_GEGLU, qk=None, out=None, eps=1e-6):
    """AdaRMSNorm -> wide projection in the attention block's form (kd_proj_block_bf16): the FF block's up projection + GEGLU
    (image_transformer_v2.py:487-491; ``weight`` [2 d_ff, K] -> [.., d_ff]) or, with ``epi=EPI_QKV`` and ``qk`` as in ``norm_linear``, the qkv
    projection with cosine-sim scale + RoPE (``weight`` [3 K, K] -> [.., 3 K]).  x [B, T, K] bf16 with T a multiple of 256, K in {256, 512};
    bit-identical to ``norm_linear(..., epi=epi)``."""
    K = x.shape[-1]
    M = x.numel() // K
    N = weight.shape[0] // (2 if epi == nat.EPI_GEGLU else 1)
    out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=x.dtype) if out is None else out
    d = gemm(x, weight, out, M=M, N=N, K=K, epi=epi, norm_scale=scale, scale_stride=K, rows_per_sample=rows_per_sample, eps=eps, qk=qk,
             precision=nat.PREC_BF16, launch=False)
    nat.check(nat.lib().kd_proj_block_bf16(C.byref(d), _stream()), "kd_proj_block_bf16")
    return out


def token_merge(x, weight, out=None):
    """TokenMerge (:586-595) with the 2x2 space-to-depth folded into the GEMM's A addressing.  x: [B, 2h, 2w, C]."""
    B, H2, W2, Cc = x.shape
    h, w = H2 // 2, W2 // 2
    Nn = weight.shape[0]
    out = torch.empty(B, h, w, Nn, device=x.device, dtype=x.dtype) if out is None else out
    return gemm(x, weight, out, M=B * h * w, N=Nn, K=4 * Cc, a_mode=nat.A_MERGE2x2, grid=(h, w), precision=_prec_of(x))


def token_split_lerp(x, weight, skip, fac, out=None):
    """TokenSplit (:610-621): Linear -> depth-to-space -> lerp(skip, x, fac), fused.  x: [B, h, w, K]."""
    B, h, w, K = x.shape
    Nn = weight.shape[0]
    out = torch.empty_like(skip) if out is None else out
    return gemm(x, weight, out, M=B * h * w, N=Nn, K=K, epi=nat.EPI_SPLIT_LERP, residual=skip, fac=fac, grid=(h, w), precision=_prec_of(x))


def patch_in(image, weight, patch, sigma=None, sigma_data=1.0, out=None, precision=None):
    """NCHW->NHWC (:723) + TokenMerge(patch) (:672,:724) (+ Denoiser's x * c_in, layers.py:90).  The image is fp32; the tokens come
    out in the activation type of the arithmetic mode (bf16 for PREC_BF16)."""
    B, Cc, H, W = image.shape
    ph, pw = patch
    h, w = H // ph, W // pw
    Nn = weight.shape[0]
    precision = nat.kernel_precision() if precision is None else precision
    act = torch.bfloat16 if precision == nat.PREC_BF16 else torch.float32
    out = torch.empty(B, h, w, Nn, device=image.device, dtype=act) if out is None else out
    return gemm(image, weight, out, M=B * h * w, N=Nn, K=Cc * ph * pw, a_mode=nat.A_PATCH_NCHW, grid=(h, w),
                patch=(ph, pw, Cc), sigma=sigma, sigma_data=sigma_data, precision=precision)


def patch_out(x, norm_scale, weight, patch, channels, x_in=None, sigma=None, sigma_data=1.0, out=None, eps=1e-6):
    """out_norm (:758) + TokenSplitWithoutSkip (:759) + NHWC->NCHW (:760)
    (+ Denoiser's F * c_out + x * c_skip, layers.py:90).  x: [B, h, w, K]."""
    B, h, w, K = x.shape
    ph, pw = patch
    out = torch.empty(B, channels, h * ph, w * pw, device=x.device, dtype=torch.float32) if out is None else out
    return gemm(x, weight, out, M=B * h * w, N=channels * ph * pw, K=K, epi=nat.EPI_UNPATCH_NCHW, norm_scale=norm_scale,
                scale_stride=0, rows_per_sample=h * w, grid=(h, w), patch=(ph, pw, channels), residual=x_in, sigma=sigma,
                sigma_data=sigma_data, eps=eps, precision=_prec_of(x))


def ffn_supported(M, K, d_ff, bf16=True):
    """Does the fused feed-forward kernel take this shape (bf16 mode, or with ``bf16=False`` the fp32-parity split3 mode)?
    Otherwise use the up / down pair of ``gemm`` calls."""
    fn = nat.lib().kd_ffn_bf16_supported if bf16 else nat.lib().kd_ffn_f32_supported
    return bool(fn(int(M), int(K), int(d_ff)))


def ffn(x, norm_scale, w_up, w_down, out=None, scale_stride=None, rows_per_sample=None, eps=1e-6, attn=None, w_out=None):
    """FeedForwardBlock.forward (image_transformer_v2.py:487-493) in one kernel:
    out = x + down_proj(GEGLU(up_proj(rms_norm(x) * norm_scale))).  x: bf16 or fp32 [..., K]; norm_scale: fp32 [B, K] (one row per
    sample; ``rows_per_sample`` tokens each) ; w_up: fp32 [2 d_ff, K]; w_down: fp32 [K, d_ff].  ``out`` may be x.
    With ``attn`` ([..., K], the attention core's output, same dtype as x) and ``w_out`` ([K, K]) the attention block's out projection
    runs in front of the block in the same kernel: x' = x + attn @ w_out.T, out = x' + ff(x') (:473-476, :487-493); widths 128 (bf16,
    fp32) and 256 (fp32)."""
    K = x.shape[-1]
    M = x.numel() // K
    d_ff = w_down.shape[1]
    if x.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("ffn: bf16 activations (KDIFF_GEMM=bf16) or fp32 activations (the fp32-parity split3 mode)")
    bf = x.dtype == torch.bfloat16
    out = torch.empty_like(x) if out is None else out
    d = nat.KdFfn()
    d.x, d.out, d.scale = _p(_chk(x, "x", x.dtype)), _p(_chk(out, "out", x.dtype)), _p(_chk(norm_scale, "norm_scale"))
    d.scale_stride = norm_scale.shape[-1] if scale_stride is None else scale_stride
    d.rows_per_sample = (M // max(norm_scale.numel() // norm_scale.shape[-1], 1)) if rows_per_sample is None else rows_per_sample
    d.eps = eps
    fused_out = attn is not None
    if fused_out and w_out is None:
        raise ValueError("ffn: attn and w_out go together")
    up_img, down_img = pack_weight(w_up, d_ff, K, 3 if fused_out else 1, bf16=bf), pack_weight(w_down, K, d_ff, 2, bf16=bf)
    d.Wp_up, d.Wp_down = _p(up_img), _p(down_img)
    d.M, d.K, d.d_ff = M, K, d_ff
    if fused_out:
        out_img = pack_weight(w_out, K, K, 0, bf16=bf)
        d.attn, d.Wp_out = _p(_chk(attn, "attn", x.dtype)), _p(out_img)
    if bf:
        nat.check(nat.lib().kd_ffn_bf16(C.byref(d), _stream()), "kd_ffn_bf16")
    else:
        nat.check(nat.lib().kd_ffn_f32(C.byref(d), _stream()), "kd_ffn_f32")
    return out


def fourier_sigma(sigma, weight, out=None):
    half = weight.shape[0]
    out = torch.empty(sigma.shape[0], 2 * half, device=sigma.device, dtype=torch.float32) if out is None else out
    nat.check(nat.lib().kd_fourier_sigma_f32(_p(_chk(sigma, "sigma")), _p(_chk(weight, "weight")), _p(_chk(out, "ff")), sigma.shape[0], half,
                                         _stream()), "kd_fourier_sigma_f32")
    return out


def fourier_features(x, weight, out=None):
    """FourierFeatures (k_diffusion/layers.py:285-293)."""
    half, in_dim = weight.shape
    B = x.numel() // in_dim
    out = torch.empty(*x.shape[:-1], 2 * half, device=x.device, dtype=torch.float32) if out is None else out
    nat.check(nat.lib().kd_fourier_f32(_p(_chk(x, "x")), _p(_chk(weight, "weight")), _p(_chk(out, "ff")), B, in_dim, half, _stream()), "kd_fourier_f32")
    return out


def cond_sum(a, b, emb=None, ids=None, c=None, out=None):
    """time_emb + aug_emb + class_emb + mapping_emb (:740)."""
    B, d = a.shape
    out = torch.empty_like(a) if out is None else out
    if emb is not None:
        _chk(ids, "ids", torch.int64)
    nat.check(nat.lib().kd_cond_sum_f32(_p(_chk(out, "out")), _p(_chk(a, "a")), _p(_chk(b, "b")), 1 if b.dim() == 2 else 0,
                                    _p(None if emb is None else _chk(emb, "emb")), _p(ids), _p(None if c is None else _chk(c, "c")),
                                    B, d, _stream()), "kd_cond_sum_f32")
    return out


def _qkv_dims(qkv, nh):
    if qkv.shape[-1] != 3 * nh * 64:
        raise ValueError(f"qkv last dim {qkv.shape[-1]} != 3*{nh}*64 (head dim is fixed at 64)")


def qk_prep_(qkv, scale_h, cos_t, sin_t, nh, eps=1e-6):
    """In place on qkv [B, T, 3*nh*64]: scale_for_cosine_sim (:106-114) + apply_rotary_emb_ (:230)."""
    _qkv_dims(qkv, nh)
    B = qkv.shape[0]
    T = qkv.numel() // (B * 3 * nh * 64)
    nat.check(nat.lib().kd_qk_prep_f32(_p(_chk(qkv, "qkv")), _p(_chk(scale_h, "scale")), _p(_chk(cos_t, "cos")), _p(_chk(sin_t, "sin")),
                                   B, T, nh, eps, _stream()), "kd_qk_prep_f32")
    return qkv


def _prep_args(prep):
    if prep is None:
        return 0, None, None, None, 1e-6
    if isinstance(prep, str):
        if prep != "packed":
            raise ValueError("prep: None (q, k prepared), (scale_h, cos, sin[, eps]) or 'packed' (prepared and stored split by the qkv GEMM)")
        return 2, None, None, None, 1e-6
    scale_h, cos_t, sin_t = prep[:3]
    eps = prep[3] if len(prep) > 3 else 1e-6
    return 1, _p(_chk(scale_h, "scale")), _p(_chk(cos_t, "cos")), _p(_chk(sin_t, "sin")), eps


def _bf16_prep(prep):
    if prep is not None:
        raise ValueError("the bf16 attention cores take q, k already prepared by the qkv GEMM's epilogue (prep=None)")


def attn_global(qkv, nh, prep=None, out=None):
    """qkv: [B, T, 3*nh*64] -> [B, T, nh*64].  ``prep=(scale_h, cos, sin[, eps])`` applies the q/k
    preparation on the fly; ``None`` means q,k are already prepared."""
    _qkv_dims(qkv, nh)
    B = qkv.shape[0]
    T = qkv.numel() // (B * 3 * nh * 64)
    out = torch.empty(*qkv.shape[:-1], nh * 64, device=qkv.device, dtype=qkv.dtype) if out is None else out
    if qkv.dtype == torch.bfloat16:
        _bf16_prep(prep)
        nat.check(nat.lib().kd_attn_global_bf16(_p(_chk(qkv, "qkv", torch.bfloat16)), _p(_chk(out, "out", torch.bfloat16)), B, T, nh, _stream()),
                  "kd_attn_global_bf16")
        return out
    f, s, c, sn, eps = _prep_args(prep)
    nat.check(nat.lib().kd_attn_global_f32(_p(_chk(qkv, "qkv")), _p(_chk(out, "out")), B, T, nh, f, s, c, sn, eps, _prec_of(qkv), _stream()), "kd_attn_global_f32")
    return out


def attn_window(qkv, nh, window_size, shift, prep=None, out=None):
    """qkv: [B, H, W, 3*nh*64] -> [B, H, W, nh*64]; apply_window_attention (:319-337)."""
    _qkv_dims(qkv, nh)
    B, H, W, _ = qkv.shape
    out = torch.empty(B, H, W, nh * 64, device=qkv.device, dtype=qkv.dtype) if out is None else out
    if qkv.dtype == torch.bfloat16:
        _bf16_prep(prep)
        nat.check(nat.lib().kd_attn_window_bf16(_p(_chk(qkv, "qkv", torch.bfloat16)), _p(_chk(out, "out", torch.bfloat16)), B, H, W, nh, window_size, shift,
                                                _stream()), "kd_attn_window_bf16")
        return out
    f, s, c, sn, eps = _prep_args(prep)
    nat.check(nat.lib().kd_attn_window_f32(_p(_chk(qkv, "qkv")), _p(_chk(out, "out")), B, H, W, nh, window_size, shift, f, s, c, sn, eps, _prec_of(qkv),
                                           _stream()), "kd_attn_window_f32")
    return out


def attn_na2d(qkv, nh, kernel_size, prep=None, out=None):
    """qkv: [B, H, W, 3*nh*64] -> [B, H, W, nh*64]; natten.functional.na2d(q, k, v, ks, scale=1.0) (:428)."""
    _qkv_dims(qkv, nh)
    B, H, W, _ = qkv.shape
    out = torch.empty(B, H, W, nh * 64, device=qkv.device, dtype=qkv.dtype) if out is None else out
    if qkv.dtype == torch.bfloat16:
        _bf16_prep(prep)
        nat.check(nat.lib().kd_attn_na2d_bf16(_p(_chk(qkv, "qkv", torch.bfloat16)), _p(_chk(out, "out", torch.bfloat16)), B, H, W, nh, kernel_size, _stream()),
                  "kd_attn_na2d_bf16")
        return out
    f, s, c, sn, eps = _prep_args(prep)
    nat.check(nat.lib().kd_attn_na2d_f32(_p(_chk(qkv, "qkv")), _p(_chk(out, "out")), B, H, W, nh, kernel_size, f, s, c, sn, eps, _prec_of(qkv), _stream()),
            "kd_attn_na2d_f32")
    return out


def sampler_step(op, x, den, in2=None, out=None, aux=None, c0=0.0, c1=0.0, c2=1.0, c3=0.0):
    out = torch.empty_like(den) if out is None else out
    for name, t in (("x", x), ("den", den), ("in2", in2), ("out", out), ("aux", aux)):
        if t is not None:
            _chk(t, name)
            if t.numel() != den.numel():
                raise ValueError(f"sampler_step: {name} has {t.numel()} elements, expected {den.numel()}")
    nat.check(nat.lib().kd_sampler_step_f32(op, _p(x), _p(den), _p(in2), _p(out), _p(aux), float(c0), float(c1), float(c2), float(c3),
                                        den.numel(), _stream()), "kd_sampler_step_f32")
    return out


def precond_in(x, sigma, sigma_data, out=None):
    out = torch.empty_like(x) if out is None else out
    B = x.shape[0]
    nat.check(nat.lib().kd_precond_in_f32(_p(_chk(x, "x")), _p(_chk(sigma, "sigma")), _p(_chk(out, "y")), float(sigma_data), B, x.numel() // B, _stream()),
            "kd_precond_in_f32")
    return out


def precond_out(f, x, sigma, sigma_data, out=None):
    out = torch.empty_like(x) if out is None else out
    B = x.shape[0]
    nat.check(nat.lib().kd_precond_out_f32(_p(_chk(f, "f")), _p(_chk(x, "x")), _p(_chk(sigma, "sigma")), _p(_chk(out, "y")), float(sigma_data), B,
                                       x.numel() // B, _stream()), "kd_precond_out_f32")
    return out


def rows_affine(f, a, x=None, c=None, out=None):
    """y[b] = f[b] * a[b] (+ x[b] * c[b]): per-sample scalars over image-sized tensors (external.py forward()s)."""
    out = torch.empty_like(f) if out is None else out
    B = f.shape[0]
    nat.check(nat.lib().kd_rows_affine_f32(_p(_chk(f, "f")), _p(None if x is None else _chk(x, "x")), _p(_chk(a, "a")),
                                       _p(None if c is None else _chk(c, "c")), _p(_chk(out, "y")), B, f.numel() // B, _stream()),
            "kd_rows_affine_f32")
    return out


def sigma_to_t(sigma, log_sigmas, quantize):
    out = torch.empty(sigma.shape, device=sigma.device, dtype=torch.float32)
    nat.check(nat.lib().kd_sigma_to_t_f32(_p(_chk(sigma.contiguous(), "sigma")), _p(_chk(log_sigmas, "log_sigmas")), _p(out), sigma.numel(),
                                      log_sigmas.numel(), int(bool(quantize)), _stream()), "kd_sigma_to_t_f32")
    return out


def t_to_sigma(t, log_sigmas):
    out = torch.empty(t.shape, device=t.device, dtype=torch.float32)
    nat.check(nat.lib().kd_t_to_sigma_f32(_p(_chk(t.contiguous(), "t")), _p(_chk(log_sigmas, "log_sigmas")), _p(out), t.numel(), log_sigmas.numel(),
                                      _stream()), "kd_t_to_sigma_f32")
    return out


def dpm_eps(x, denoised, sigma, out=None):
    """(x - denoised) / sigma -- DPMSolver.eps (sampling.py:354), the reference's rounding order."""
    out = torch.empty_like(x) if out is None else out
    nat.check(nat.lib().kd_dpm_eps_f32(_p(_chk(out, "out")), _p(_chk(x, "x")), _p(_chk(denoised, "denoised")), float(sigma), x.numel(), _stream()),
              "kd_dpm_eps_f32")
    return out


def dpm_combine(x, eps, a, eps_r=None, b=0.0, out=None):
    """x - a * eps [- b * (eps_r - eps)]: every DPM-Solver state (sampling.py:363-387), the reference's rounding order."""
    out = torch.empty_like(x) if out is None else out
    nat.check(nat.lib().kd_dpm_combine_f32(_p(_chk(out, "out")), _p(_chk(x, "x")), _p(_chk(eps, "eps")), None if eps_r is None else _p(_chk(eps_r, "eps_r")),
                                           float(a), float(b), x.numel(), _stream()), "kd_dpm_combine_f32")
    return out


def dpm_error(x_low, x_high, x_prev, atol, rtol):
    """Adaptive DPM-Solver local error (sampling.py:464-465) as a python float: sqrt(sum(((lo - hi) / delta)^2) / numel).
    One launch writes fixed-grid partial sums; their total is taken on the host (the controller needs the value there)."""
    part = torch.empty(nat.lib().kd_dpm_error_partials(), device=x_low.device, dtype=torch.float32)
    nat.check(nat.lib().kd_dpm_error_f32(_p(_chk(x_low, "x_low")), _p(_chk(x_high, "x_high")), _p(_chk(x_prev, "x_prev")), float(atol), float(rtol),
                                         x_low.numel(), _p(part), _stream()), "kd_dpm_error_f32")
    return math.sqrt(float(part.double().sum().item()) / x_low.numel())


def brownian(out, seeds, T0, T1, t0, t1, mult, depth=36):
    """out[b, ...] = (W_b(t1) - W_b(t0)) * mult, one virtual Brownian tree per batch item (seeds: uint64 [B])."""
    B = out.shape[0]
    _chk(seeds, "seeds", torch.int64)
    nat.check(nat.lib().kd_brownian_f32(_p(_chk(out, "out")), _p(seeds), B, out.numel() // B, float(T0), float(T1), float(t0), float(t1),
                                    float(mult), depth, _stream()), "kd_brownian_f32")
    return out


def brownian_cached(out, w0, have0, w1, have1, seeds, T0, T1, t0, t1, mult, depth=36):
    """As ``brownian`` with the end-point tensors W(t0) / W(t1) kept by the caller: ``have*`` says the
    buffer already holds that end point (read, no descent); otherwise it is computed and stored."""
    B = out.shape[0]
    _chk(seeds, "seeds", torch.int64)
    for w in (w0, w1):
        if w is not None and _chk(w, "w").numel() != out.numel():
            raise ValueError("end-point buffers must match out")
    nat.check(nat.lib().kd_brownian_cached_f32(_p(_chk(out, "out")), _p(w0) if w0 is not None else None, _p(w1) if w1 is not None else None,
                                           int(bool(have0)), int(bool(have1)), _p(seeds), B, out.numel() // B, float(T0), float(T1),
                                           float(t0), float(t1), float(mult), depth, _stream()), "kd_brownian_cached_f32")
    return out


def randn_indexed(out, seeds, draw=0, scale=1.0):
    """out[b, ...] = scale * standard normals that are a function of (seeds[b], draw, element index) only (seeds: int64 [B] on the
    device, ``draw`` numbers the calls of one run): the device-side form of ``torch.randn(...) * sigma_max`` (sample.py:59) /
    ``randn_like`` (sampling.py:61-62) for jobs whose image i must not depend on the batch or rank it is drawn in."""
    B = out.shape[0]
    _chk(seeds, "seeds", torch.int64)
    if seeds.numel() != B:
        raise ValueError(f"one seed per batch item: {seeds.numel()} seeds for batch {B}")
    nat.check(nat.lib().kd_randn_f32(_p(_chk(out, "out")), _p(seeds), B, out.numel() // B, int(draw), float(scale), _stream()), "kd_randn_f32")
    return out


def to_uint8(x, out=None):
    out = torch.empty(x.shape, device=x.device, dtype=torch.uint8) if out is None else out
    if x.numel() == 0:       # an empty shard (more ranks than images in a round): nothing to launch, and the other ranks still gather
        _chk(x, "x"), _chk(out, "y", torch.uint8)
        return out
    nat.check(nat.lib().kd_to_uint8(_p(_chk(x, "x")), _p(_chk(out, "y", torch.uint8)), x.numel(), _stream()), "kd_to_uint8")
    return out
