"""CPU oracle (test infrastructure): functional restatement of the HDiT denoiser forward pass.

Restates /root/reference/k_diffusion/models/image_transformer_v2.py as pure functions over a
``state_dict`` (no nn.Module), in plain torch fp32 on the CPU.  Each function cites the reference
lines it follows.  Pinned against outputs of the real reference by tests/test_oracle_vs_golden.py
-- except ``na2d`` (NATTEN is third-party and absent: **parity unpinned**, semantics defined here
per SURVEY.md App. A.9: 7x7 window clamped inside the image, dilation 1, no bias).

All attention variants use the heads-last layout ``[n, h, w, nh, e]``
(image_transformer_v2.py:422); q/k/v are views of the qkv projection output whose feature index
is ``t*(nh*e) + head*e + e_idx`` (``:386, :422, :431, :467``).
"""
import contextlib
import math

import torch
import torch.nn.functional as F

EPS = 1e-6


# ----------------------------------------------------------------------------- small ops

def rms_norm(x, scale):
    """image_transformer_v2.py:98-103: x * (scale * rsqrt(mean(x^2) + eps)), statistics in fp32."""
    mean_sq = x.float().square().mean(dim=-1, keepdim=True)
    return x * (scale.float() * torch.rsqrt(mean_sq + EPS)).to(x.dtype)


def ada_rms_norm(x, cond, w):
    """:155-166: scale = Linear(cond) + 1, broadcast over (h, w)."""
    scale = (cond @ w.T)[:, None, None, :] + 1
    return rms_norm(x, scale)


def linear_geglu(x, w):
    """:89-95: first half of the output features is the value, second half the (erf-)GELU gate."""
    h = x @ w.T
    d = h.shape[-1] // 2
    return h[..., :d] * F.gelu(h[..., d:])


def cosine_sim_scale(q, k, scale):
    """:106-114.  q, k: [..., nh, e]; scale: [nh]."""
    s = torch.sqrt(scale.float())[:, None]
    q = q * (s * torch.rsqrt(q.float().square().sum(-1, keepdim=True) + EPS))
    k = k * (s * torch.rsqrt(k.float().square().sum(-1, keepdim=True) + EPS))
    return q, k


def axial_pos(h, w):
    """axial_rope.py:31-68 (align_corners=False, square pixel aspect): centres of an h x w grid
    laid over the bounding box; returns [h, w, 2] = (y, x)."""
    ar = w / h
    y_min, y_max, x_min, x_max = -1.0, 1.0, -1.0, 1.0
    if ar > 1:
        y_min, y_max = -1 / ar, 1 / ar
    elif ar < 1:
        x_min, x_max = -ar, ar

    def centres(lo, hi, n):
        edges = torch.linspace(lo, hi, n + 1)
        return (edges[:-1] + edges[1:]) / 2
    ys, xs = centres(y_min, y_max, h), centres(x_min, x_max, w)
    return torch.stack([ys[:, None].expand(h, w), xs[None, :].expand(h, w)], dim=-1)


def downscale_pos(pos):
    """image_transformer_v2.py:52-54: 2x2 mean of the position grid."""
    h, w, _ = pos.shape
    return pos.view(h // 2, 2, w // 2, 2, 2).permute(0, 2, 1, 3, 4).reshape(h // 2, w // 2, 4, 2).mean(dim=-2)


def rope_freqs(n_heads, d_head=64):
    """:234-240: AxialRoPE(dim=d_head//2, n_heads).freqs -> [nh, d_head//8]."""
    dim = d_head // 2
    f = torch.linspace(math.log(math.pi), math.log(10.0 * math.pi), n_heads * dim // 4 + 1)[:-1].exp()
    return f.view(dim // 4, n_heads).T.contiguous()


def rope_theta(pos, freqs):
    """:245-248: theta[h, w, nh, 2*F] = cat(pos_y * freqs, pos_x * freqs)."""
    th = pos[..., None, 0:1] * freqs
    tw = pos[..., None, 1:2] * freqs
    return torch.cat((th, tw), dim=-1)


def apply_rope(x, theta):
    """:187-199: rotate dims [0,d) against [d,2d) (half-split pairing), rest untouched.
    x: [n, h, w, nh, e], theta: [h, w, nh, d]."""
    d = theta.shape[-1]
    x1, x2, x3 = x[..., :d], x[..., d:2 * d], x[..., 2 * d:]
    cos, sin = torch.cos(theta), torch.sin(theta)
    return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin, x3), dim=-1)


def split_qkv(qkv, n_heads):
    """[n, h, w, 3*nh*e] -> q, k, v each [n, h, w, nh, e] (":... (t nh e)" with t outermost)."""
    n, h, w, c = qkv.shape
    e = c // (3 * n_heads)
    t = qkv.view(n, h, w, 3, n_heads, e)
    return t[..., 0, :, :], t[..., 1, :, :], t[..., 2, :, :]


# ----------------------------------------------------------------------------- attention cores

def attn_global(q, k, v, scale=1.0):
    """:385-393: dense softmax attention over all h*w tokens per (sample, head)."""
    n, h, w, nh, e = q.shape
    qs = q.reshape(n, h * w, nh, e).permute(0, 2, 1, 3)
    ks = k.reshape(n, h * w, nh, e).permute(0, 2, 1, 3)
    vs = v.reshape(n, h * w, nh, e).permute(0, 2, 1, 3)
    p = torch.softmax((qs @ ks.transpose(-1, -2)) * scale, dim=-1)
    return (p @ vs).permute(0, 2, 1, 3).reshape(n, h, w, nh, e)


def na2d_window_start(length, kernel_size):
    """Start index of the clamped neighbourhood along one axis (NATTEN semantics, dilation 1):
    start(i) = clamp(i - kernel_size // 2, 0, length - kernel_size)."""
    if length < kernel_size:
        raise ValueError(f"neighbourhood kernel {kernel_size} larger than axis {length}")
    i = torch.arange(length)
    return (i - kernel_size // 2).clamp(0, length - kernel_size)


def na2d(q, k, v, kernel_size, scale=1.0):
    """Restated natten.functional.na2d (call site image_transformer_v2.py:428), heads-last.
    PARITY UNPINNED (NATTEN absent): every query attends the kernel_size^2 keys of its clamped
    window; softmax over them; no bias."""
    n, h, w, nh, e = q.shape
    ks = kernel_size
    ih = na2d_window_start(h, ks)[:, None] + torch.arange(ks)[None, :]    # [h, ks]
    iw = na2d_window_start(w, ks)[:, None] + torch.arange(ks)[None, :]    # [w, ks]
    out = torch.empty_like(q)
    rows = max(1, 4096 // w)
    for r0 in range(0, h, rows):
        r1 = min(h, r0 + rows)
        # keys for queries in rows [r0, r1): [n, r, w, ks, ks, nh, e]
        kk = k[:, ih[r0:r1][:, None, :, None], iw[None, :, None, :]]
        vv = v[:, ih[r0:r1][:, None, :, None], iw[None, :, None, :]]
        logits = torch.einsum("nrwhe,nrwabhe->nrwhab", q[:, r0:r1], kk) * scale
        p = torch.softmax(logits.flatten(-2), dim=-1).view_as(logits)
        out[:, r0:r1] = torch.einsum("nrwhab,nrwabhe->nrwhe", p, vv)
    return out


def na2d_shifted(q, k, v, kernel_size, scale=1.0):
    """The same op as ``na2d`` evaluated one window offset at a time (kernel_size^2 passes over [n, h, w, nh, e] tensors instead
    of one gather that materialises kernel_size^2 copies of K and V): several times faster on the CPU at 64x64 tokens.  Used by
    bench.py's cpu_baseline only -- the golden vectors were recorded with ``na2d`` and its summation order (the two agree to
    fp32 rounding, tests/test_oracle_vs_golden.py::test_na2d_shifted_matches)."""
    n, h, w, nh, e = q.shape
    ks = kernel_size
    sh, sw = na2d_window_start(h, ks), na2d_window_start(w, ks)
    logits = q.new_empty(n, h, w, nh, ks * ks)
    for a in range(ks):
        ka = k[:, sh + a]
        for b in range(ks):
            logits[..., a * ks + b] = (q * ka[:, :, sw + b]).sum(-1)
    p = torch.softmax(logits * scale, dim=-1)
    out = torch.zeros_like(q)
    for a in range(ks):
        va = v[:, sh + a]
        for b in range(ks):
            out += p[..., a * ks + b, None] * va[:, :, sw + b]
    return out


def window_token_index(h, w, ws, shift):
    """Token coordinates (in the un-rolled image) of every window slot after the reference's
    ``torch.roll(x, (shift, shift), dims=(w, h))`` + 8x8 tiling (:253-276).
    rolled[i, j] = orig[(i - shift) % h, (j - shift) % w].  Returns flat indices [nwin, ws*ws]."""
    wi, wj = torch.arange(h // ws), torch.arange(w // ws)
    a, b = torch.arange(ws), torch.arange(ws)
    ri = (wi[:, None, None, None] * ws + a[None, None, :, None] - shift) % h
    rj = (wj[None, :, None, None] * ws + b[None, None, None, :] - shift) % w
    return (ri * w + rj).reshape(-1, ws * ws)


def window_mask(h, w, ws, shift):
    """:285-316 restated: a (query, key) pair is allowed iff both lie in the same wrapped region.
    Region id along rows is (a < shift) only for windows in the top window-row, along columns
    (b < shift) only for the left window-column.  Returns bool [nwin, ws*ws, ws*ws]."""
    nwh, nww = h // ws, w // ws
    a = torch.arange(ws)
    rid_row = torch.zeros(nwh, ws, dtype=torch.long)
    rid_row[0] = (a < shift).long()
    rid_col = torch.zeros(nww, ws, dtype=torch.long)
    rid_col[0] = (a < shift).long()
    rid = rid_row[:, None, :, None] * 2 + rid_col[None, :, None, :]        # [nwh, nww, ws, ws]
    rid = rid.reshape(nwh * nww, ws * ws)
    return rid[:, :, None] == rid[:, None, :]


def attn_shifted_window(q, k, v, window_size, shift, scale=1.0):
    """:319-337: shifted-window attention, heads-last in and out."""
    n, h, w, nh, e = q.shape
    idx = window_token_index(h, w, window_size, shift)                      # [nwin, T]
    mask = window_mask(h, w, window_size, shift)                            # [nwin, T, T]
    flat = lambda t: t.reshape(n, h * w, nh, e)[:, idx]                     # [n, nwin, T, nh, e]
    qw, kw, vw = flat(q), flat(k), flat(v)
    logits = torch.einsum("nwqhe,nwkhe->nwhqk", qw, kw) * scale
    logits = logits.masked_fill(~mask[None, :, None], float("-inf"))
    p = torch.softmax(logits, dim=-1)
    ow = torch.einsum("nwhqk,nwkhe->nwqhe", p, vw)
    out = torch.empty(n, h * w, nh, e, dtype=q.dtype)
    out[:, idx.reshape(-1)] = ow.reshape(n, -1, nh, e)
    return out.view(n, h, w, nh, e)


# ----------------------------------------------------------------------------- fp8 arithmetic mode (no reference counterpart)
# Restates the arithmetic csrc/gemm_mx8.hip defines (kd_gemm_mx8 / kd_pack_weight_mx8): OCP e4m3 values with power-of-two scales -- one per
# output channel for weights (checkpoint.quantize_fp8's rule), one per (row, 32-k block) for activations (OCP microscaling, E8M0 scale byte).

FP8_MAX = 448.0
MX8_WIDTHS = (256, 512)     # K of the norm -> projection products the fp8 mode takes


def mx8_scale(amax):
    """Smallest power of two >= amax / 448 (exponent byte clamped to [1, 253]), from the bits of the fp32 quotient like the kernel."""
    r = (amax.to(torch.float32) / FP8_MAX).contiguous()
    byte = ((r.view(torch.int32) + 0x7FFFFF) >> 23).clamp(1, 253)
    return torch.ldexp(torch.ones_like(r), byte - 127)


def mx8_quantize_rows(u):
    """u [..., K] fp32, K % 32 == 0 -> the fp32 value of its microscaled e4m3 storage: per 32-k block, scale = mx8_scale(max |u|)."""
    shape = u.shape
    b = u.to(torch.float32).reshape(*shape[:-1], shape[-1] // 32, 32)
    s = mx8_scale(b.abs().amax(dim=-1, keepdim=True))
    return ((b / s).to(torch.float8_e4m3fn).to(torch.float32) * s).reshape(shape)


def mx8_quantize_weight(w):
    """W [N, K] fp32 -> the fp32 value of its e4m3 storage with one power-of-two scale per output channel."""
    s = mx8_scale(w.abs().amax(dim=1, keepdim=True))
    return (w.to(torch.float32) / s).to(torch.float8_e4m3fn).to(torch.float32) * s


MX8 = False     # inside ``with mx8_arithmetic():`` the norm -> qkv / norm -> GEGLU products of the MX8_WIDTHS levels use the arithmetic above


@contextlib.contextmanager
def mx8_arithmetic(on=True):
    global MX8
    saved, MX8 = MX8, bool(on)
    try:
        yield
    finally:
        MX8 = saved


def norm_linear(x, cond, w_norm, w):
    """AdaRMSNorm -> Linear (:370-372, :487-489).  With MX8 on and the width taken by the fp8 mode: the kernel's order of operations --
    u = x * scale quantised per block, exact products with the quantised weight, the RMS row factor of the UNQUANTISED row afterwards."""
    if not (MX8 and x.shape[-1] in MX8_WIDTHS):
        return ada_rms_norm(x, cond, w_norm) @ w.T
    scale = (cond @ w_norm.T)[:, None, None, :] + 1
    rs = torch.rsqrt(x.float().square().mean(dim=-1, keepdim=True) + EPS)
    return (mx8_quantize_rows(x * scale) @ mx8_quantize_weight(w).T) * rs


# ----------------------------------------------------------------------------- blocks

def self_attention_block(sd, prefix, spec, layer_index, x, pos, cond):
    """:370-396 / :415-443 / :463-476: AdaRMSNorm -> qkv -> cos-sim scale -> RoPE(q, k) ->
    attention(scale=1.0) -> out_proj -> + skip."""
    skip = x
    n_heads = x.shape[-1] // spec.get("d_head", 64)
    qkv = norm_linear(x, cond, sd[prefix + "norm.linear.weight"], sd[prefix + "qkv_proj.weight"])
    q, k, v = split_qkv(qkv, n_heads)
    q, k = cosine_sim_scale(q, k, sd[prefix + "scale"])
    theta = rope_theta(pos, sd[prefix + "pos_emb.freqs"])
    q, k = apply_rope(q, theta), apply_rope(k, theta)
    kind = spec["type"]
    if kind == "global":
        o = attn_global(q, k, v, 1.0)
    elif kind == "neighborhood":
        o = na2d(q, k, v, spec.get("kernel_size", 7), 1.0)
    elif kind == "shifted-window":
        ws = spec["window_size"]
        o = attn_shifted_window(q, k, v, ws, ws // 2 if layer_index % 2 == 1 else 0, 1.0)   # :523
    else:
        raise ValueError(kind)
    o = o.reshape(*o.shape[:3], -1)
    return o @ sd[prefix + "out_proj.weight"].T + skip


def feed_forward_block(sd, prefix, x, cond):
    """:487-493."""
    h = norm_linear(x, cond, sd[prefix + "norm.linear.weight"], sd[prefix + "up_proj.weight"])
    d = h.shape[-1] // 2
    h = h[..., :d] * F.gelu(h[..., d:])                       # linear_geglu (:89-95) on the projection above
    if MX8 and x.shape[-1] in MX8_WIDTHS and d % 128 == 0:     # fp8 mode: the hidden activation and the down projection's weight are e4m3 too
        return mx8_quantize_rows(h) @ mx8_quantize_weight(sd[prefix + "down_proj.weight"]).T + x
    return h @ sd[prefix + "down_proj.weight"].T + x


def level(sd, prefix, spec, depth, index_offset, x, pos, cond):
    """:496-547; shifted-window layers get index i (down / mid) or i + depth (up) (:696-697)."""
    for i in range(depth):
        p = f"{prefix}{i}."
        if spec["type"] != "none":
            x = self_attention_block(sd, p + "self_attn.", spec, i + index_offset, x, pos, cond)
        x = feed_forward_block(sd, p + "ff.", x, cond)
    return x


def token_merge(x, w, ph, pw):
    """:586-595: "(h nh) (w nw) e -> h w (nh nw e)" then Linear."""
    n, H, W, e = x.shape
    x = x.view(n, H // ph, ph, W // pw, pw, e).permute(0, 1, 3, 2, 4, 5).reshape(n, H // ph, W // pw, ph * pw * e)
    return x @ w.T


def token_split(x, w, ph, pw):
    """:598-607: Linear then "h w (nh nw e) -> (h nh) (w nw) e"."""
    x = x @ w.T
    n, h, ww, c = x.shape
    e = c // (ph * pw)
    return x.view(n, h, ww, ph, pw, e).permute(0, 1, 3, 2, 4, 5).reshape(n, h * ph, ww * pw, e)


def fourier_features(x, weight):
    """layers.py:285-293."""
    f = 2 * math.pi * x @ weight.T
    return torch.cat([f.cos(), f.sin()], dim=-1)


def mapping_network(sd, x, depth):
    """image_transformer_v2.py:552-581."""
    x = rms_norm(x, sd["mapping.in_norm.scale"])
    for i in range(depth):
        p = f"mapping.blocks.{i}."
        h = rms_norm(x, sd[p + "norm.scale"])
        h = linear_geglu(h, sd[p + "up_proj.weight"])
        x = h @ sd[p + "down_proj.weight"].T + x
    return rms_norm(x, sd["mapping.out_norm.scale"])


def conditioning(sd, mcfg, sigma, class_cond=None, aug_cond=None, mapping_cond=None):
    """:729-740."""
    c_noise = torch.log(sigma) / 4
    emb = fourier_features(c_noise[:, None], sd["time_emb.weight"]) @ sd["time_in_proj.weight"].T
    aug = torch.zeros(sigma.shape[0], 9) if aug_cond is None else aug_cond
    emb = emb + fourier_features(aug, sd["aug_emb.weight"]) @ sd["aug_in_proj.weight"].T
    if "class_emb.weight" in sd:
        if class_cond is None:
            raise ValueError("class_cond must be specified if num_classes > 0")
        emb = emb + sd["class_emb.weight"][class_cond]
    if "mapping_cond_in_proj.weight" in sd:
        if mapping_cond is None:
            raise ValueError("mapping_cond must be specified if mapping_cond_dim > 0")
        emb = emb + mapping_cond @ sd["mapping_cond_in_proj.weight"].T
    return mapping_network(sd, emb, mcfg.get("mapping_depth", 2))


def forward(sd, mcfg, x, sigma, class_cond=None, aug_cond=None, mapping_cond=None, taps=None):
    """ImageTransformerDenoiserModelV2.forward (:721-762).

    sd: state dict (fp32 CPU tensors); mcfg: the merged ``config['model']`` dict
    (widths, depths, d_ffs, self_attns, patch_size, mapping_depth); x: [n, c, H, W]; sigma: [n].
    ``taps``: optional dict that receives named intermediate activations (for per-stage parity).
    """
    widths, depths, specs = mcfg["widths"], mcfg["depths"], mcfg["self_attns"]
    ph, pw = mcfg["patch_size"]
    tap = (lambda k, v: taps.__setitem__(k, v.clone())) if taps is not None else (lambda k, v: None)

    x = token_merge(x.movedim(-3, -1).contiguous(), sd["patch_in.proj.weight"], ph, pw)
    tap("patch_in", x)
    pos = axial_pos(x.shape[-3], x.shape[-2])
    cond = conditioning(sd, mcfg, sigma, class_cond, aug_cond, mapping_cond)
    tap("cond", cond)

    skips, poses = [], []
    n_levels = len(widths)
    for li in range(n_levels - 1):
        x = level(sd, f"down_levels.{li}.", specs[li], depths[li], 0, x, pos, cond)
        tap(f"down{li}", x)
        skips.append(x)
        poses.append(pos)
        x = token_merge(x, sd[f"merges.{li}.proj.weight"], 2, 2)
        pos = downscale_pos(pos)
    x = level(sd, "mid_level.", specs[-1], depths[-1], 0, x, pos, cond)
    tap("mid", x)
    for li in reversed(range(n_levels - 1)):
        up = token_split(x, sd[f"splits.{li}.proj.weight"], 2, 2)
        x = torch.lerp(skips[li], up, sd[f"splits.{li}.fac"])                       # :621
        x = level(sd, f"up_levels.{li}.", specs[li], depths[li], depths[li], x, poses[li], cond)
        tap(f"up{li}", x)
    x = rms_norm(x, sd["out_norm.scale"])
    x = token_split(x, sd["patch_out.proj.weight"], ph, pw)
    return x.movedim(-1, -3).contiguous()


from importlib import import_module as _im

forward_cost_mac = _im("k_diffusion_amd").models.flops.forward_cost_mac     # lives in the package (bench.py reports it); re-exported for the tests
