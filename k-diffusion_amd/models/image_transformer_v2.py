"""Hourglass diffusion transformer ("image_transformer_v2") denoiser, MI355X-native forward.

Drop-in for ``k_diffusion.models.ImageTransformerDenoiserModelV2``
(k_diffusion/models/image_transformer_v2.py:667-762): same constructor, same ``forward(x, sigma,
aug_cond, class_cond, mapping_cond)``, same ``state_dict`` keys and shapes (the checkpoint
contract), same error behaviour for missing conditioning.

What differs is *how* the forward runs.  The reference is a tree of nn.Modules issuing ~5 400
ATen / Triton / NATTEN / flash-attn launches per forward.  Here the module tree only *holds*
weights; the forward is a flat, pre-planned list of ~90 launches of hand-written gfx950 kernels
(csrc/, C ABI in include/kdiff_hip.h) over a token-major fp32 workspace:

  conditioning : FourierFeatures kernel -> GEMMs (mapping network) -> ONE GEMM producing every
                 AdaRMSNorm scale of the network ([B, sum(d)] table, "+1" folded into the epilogue)
  patch_in     : NCHW gather + patch + Linear (+ Karras c_in) in one GEMM
  each layer   : [AdaRMSNorm -> qkv GEMM] -> attention core with cosine-sim scaling and axial RoPE
                 applied on the fly -> [out_proj GEMM + residual] ;
                 [AdaRMSNorm -> up GEMM -> GEGLU] -> [down GEMM + residual]
  merge/split  : 2x2 space-to-depth / depth-to-space folded into GEMM addressing; lerp in the epilogue
  patch_out    : [RMSNorm -> GEMM -> NHWC->NCHW scatter (+ Karras c_out, c_skip)] in one GEMM

The plan (workspace + prebuilt launch descriptors) is cached per input shape.  There is no eager /
CPU path: calling ``forward`` without a ROCm device or without the built library raises.
"""
import ctypes as C
import os
import math
from dataclasses import dataclass
from typing import Union

import torch
from torch import nn

from .. import _native as nat
from .. import ops
from . import axial_rope

D_HEAD = 64
# Launch-bound regime: below this many level-0 tokens per forward the 60-odd kernels of the main chain are shorter than the host
# needs to issue them one by one, so the chain is captured once per (scale table, preconditioning) combination and replayed as a
# hipGraph (one host call per forward).  Above it the host stays ahead anyway and replay measured 2 % slower.
# KDIFF_GRAPH=0 / 1 forces never / always.
GRAPH_AUTO_MAX_TOKENS = 16384


def _graph_policy(tokens):
    mode = os.environ.get("KDIFF_GRAPH", "auto").lower()
    if mode in ("0", "off", "never"):
        return False
    if mode in ("1", "on", "always"):
        return True
    if mode != "auto":
        raise ValueError(f"KDIFF_GRAPH={mode!r}: expected 0, 1 or auto")
    return tokens <= GRAPH_AUTO_MAX_TOKENS


# ---------------------------------------------------------------------------------- configuration

@dataclass
class GlobalAttentionSpec:
    d_head: int


@dataclass
class NeighborhoodAttentionSpec:
    d_head: int
    kernel_size: int


@dataclass
class ShiftedWindowAttentionSpec:
    d_head: int
    window_size: int


@dataclass
class NoAttentionSpec:
    pass


@dataclass
class LevelSpec:
    depth: int
    width: int
    d_ff: int
    self_attn: Union[GlobalAttentionSpec, NeighborhoodAttentionSpec, ShiftedWindowAttentionSpec, NoAttentionSpec]
    dropout: float


@dataclass
class MappingSpec:
    depth: int
    width: int
    d_ff: int
    dropout: float


# ---------------------------------------------------------------------------------- weight holders

class _Holder(nn.Module):
    """Names a group of parameters / sub-holders so that state_dict keys match the reference."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            if isinstance(v, nn.Parameter):
                self.register_parameter(k, v)
            elif isinstance(v, torch.Tensor):
                self.register_buffer(k, v)
            else:
                self.add_module(k, v)


def _linear_weight(out_f, in_f, zero=False):
    """nn.Linear's default init (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(in), 1/sqrt(in))), or the
    reference's zero_init (image_transformer_v2.py:37-41)."""
    w = torch.zeros(out_f, in_f)
    if not zero:
        bound = 1.0 / math.sqrt(in_f)
        w.uniform_(-bound, bound)
    return _Holder(weight=nn.Parameter(w))


def _layer(spec: LevelSpec, cond_features: int):
    d, parts = spec.width, {}
    if not isinstance(spec.self_attn, NoAttentionSpec):
        nh = d // spec.self_attn.d_head
        parts["self_attn"] = _Holder(
            scale=nn.Parameter(torch.full([nh], 10.0)),
            norm=_Holder(linear=_linear_weight(d, cond_features, zero=True)),
            qkv_proj=_linear_weight(3 * d, d),
            pos_emb=_Holder(freqs=axial_rope.rope_freqs(spec.self_attn.d_head // 2, nh)),
            out_proj=_linear_weight(d, d, zero=True))
    parts["ff"] = _Holder(norm=_Holder(linear=_linear_weight(d, cond_features, zero=True)),
                          up_proj=_linear_weight(2 * spec.d_ff, d),
                          down_proj=_linear_weight(d, spec.d_ff, zero=True))
    return _Holder(**parts)


def _rms_scale(n):
    return _Holder(scale=nn.Parameter(torch.ones(n)))


# ---------------------------------------------------------------------------------- the plan

class _Launch:
    __slots__ = ("fn", "args", "what")

    def __init__(self, fn, args, what):
        self.fn, self.args, self.what = fn, args, what


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class _Plan:
    """Workspace + prebuilt launch list for one (batch, H, W, conditioning-kinds) combination."""

    def __init__(self, model, B, H, W, has_aug, has_class, has_mapping_cond, device):
        lib = nat.lib()
        m = model
        precision = nat.default_precision()
        bf = precision == nat.PREC_BF16
        cond_precision = nat.PREC_SPLIT3 if bf else precision      # the per-sample conditioning chain stays fp32 in every mode
        self.keep = []           # descriptors and tensors that must outlive the plan
        self.launches = []
        f32 = dict(device=device, dtype=torch.float32)
        act = dict(device=device, dtype=torch.bfloat16 if bf else torch.float32)   # residual stream, qkv, attention out, FF hidden
        ph, pw = m.patch_size
        if H % ph or W % pw:
            raise ValueError(f"input {H}x{W} not divisible by the patch size {ph}x{pw}")
        levels = m.level_specs
        n_lv = len(levels)
        grids = [(H // ph, W // pw)]
        for _ in range(n_lv - 1):
            gh, gw = grids[-1]
            if gh % 2 or gw % 2:
                raise ValueError(f"token grid {gh}x{gw} cannot be merged 2x2")
            grids.append((gh // 2, gw // 2))
        self.B, self.grids = B, grids
        self.out_shape = (B, m.out_channels, H, W)
        self.use_graph = _graph_policy(B * grids[0][0] * grids[0][1])
        self.graphs, self.cond_graphs = {}, {}              # captured main chains / conditioning chains (see replay())
        self.g_x = self.g_out = self.capture_stream = None   # their fixed input / output images
        self.direct_runs = self.direct_cond_runs = 0
        self.graph_epoch = nat.option_epoch
        mw, mdff = m.mapping_spec.width, m.mapping_spec.d_ff

        # ---- static buffers -----------------------------------------------------------------
        self.sigma = torch.empty(B, **f32)                  # read by the patch-in / patch-out preconditioning of the MAIN chain
        # inputs of the CONDITIONING chain (its own copies: the chain of the next solver step may run on the side
        # stream while the main chain of the current step is still in flight)
        self.c_sigma = torch.empty(B, **f32)
        self.class_ids = torch.zeros(B, device=device, dtype=torch.int64)
        self.aug_in = torch.zeros(B, 9, **f32) if has_aug else None
        self.map_in = torch.zeros(B, m.mapping_cond_dim, **f32) if has_mapping_cond else None
        xs = [torch.empty(B, gh, gw, lv.width, **act) for (gh, gw), lv in zip(grids, levels)]
        toks = [B * gh * gw for gh, gw in grids]
        qkv = torch.empty(max(t * 3 * lv.width for t, lv in zip(toks, levels)), **act)
        att = torch.empty(max(t * lv.width for t, lv in zip(toks, levels)), **act)
        hid = torch.empty(max(t * lv.d_ff for t, lv in zip(toks, levels)), **act)
        ff, temb, emb, mres, cond = (torch.empty(B, mw, **f32) for _ in range(5))
        mh = torch.empty(B, mdff, **f32)
        norm_mods = m._ada_norm_modules()
        offsets, total = {}, 0
        for name, mod in norm_mods:
            offsets[name] = total
            total += mod.linear.weight.shape[0]
        wcat = torch.cat([mod.linear.weight.detach() for _, mod in norm_mods], dim=0).contiguous()
        # AdaRMSNorm scale tables, ping-pong: the main chain reads one while the next step's table is being written
        self.scales = [torch.empty(B, total, **f32), torch.empty(B, total, **f32)]
        self.norm_descs = []                                # (descriptor, byte offset into a scale table)
        self.last_buf, self.prefetched = 1, None            # prefetched: (identity of the conditioning tensors, table, done event)
        self.side_stream = torch.cuda.Stream(device=device)
        self.main_entry = torch.cuda.Event()
        self.cond_launches = []
        self.keep += [xs, qkv, att, hid, ff, temb, emb, mres, cond, mh, wcat]
        self.xs = xs

        def gemm(what, A, Wt, Cc, M, N, K, a_mode=nat.A_PLAIN, epi=nat.EPI_STORE, scale_ptr=None, scale_stride=0,
                 rows_per_sample=0, R=None, grid=(0, 0), patch=(0, 0, 0), out_add=0.0, sigma=None, fac=None, qk=None):
            d = nat.KdGemm()
            d.M, d.N, d.K, d.a_mode, d.epi = M, N, K, a_mode, epi
            main = target is self.launches
            d.precision = precision if main else cond_precision
            if d.precision == nat.PREC_BF16:
                d.Wp = m._packed_image(Wt, N, K, epi == nat.EPI_GEGLU, bf16=True).data_ptr()
            elif d.precision == nat.PREC_SPLIT3:
                d.Wp = m._packed_image(Wt, N, K, epi == nat.EPI_GEGLU).data_ptr()
            d.norm = 1 if scale_ptr is not None else 0
            d.rows_per_sample, d.scale_stride = rows_per_sample, scale_stride
            d.gh, d.gw = grid
            d.ph, d.pw, d.chan = patch
            d.eps, d.out_add, d.sigma_data = 1e-6, out_add, 1.0
            d.A = None if A is None else A.data_ptr()
            d.W, d.C = Wt.data_ptr(), (None if Cc is None else Cc.data_ptr())
            d.R = None if R is None else R.data_ptr()
            d.scale = scale_ptr if not isinstance(scale_ptr, tuple) else None
            if isinstance(scale_ptr, tuple):                # ("table", byte offset): patched per run to the live scale table
                self.norm_descs.append((d, scale_ptr[1]))
            d.sigma = None if sigma is None else sigma.data_ptr()
            d.fac = None if fac is None else fac.data_ptr()
            if qk is not None and d.precision == nat.PREC_BF16:      # (scale_h, rope_pos, rope_freq, nh)
                d.qk_scale, d.rope_pos, d.rope_freq, d.n_heads = qk[0].data_ptr(), qk[1].data_ptr(), qk[2].data_ptr(), qk[3]
            elif qk is not None:
                d.qk_scale, d.rope_cos, d.rope_sin, d.n_heads = qk[0].data_ptr(), qk[1].data_ptr(), qk[2].data_ptr(), qk[3]
            self.keep.append(d)
            target.append(_Launch(lib.kd_gemm_bf16 if d.precision == nat.PREC_BF16 else lib.kd_gemm_f32, (C.byref(d),), what))
            return d

        def call(what, fn, *args):
            target.append(_Launch(fn, args, what))

        target = self.cond_launches

        # ---- conditioning (image_transformer_v2.py:734-740, :569-581) ------------------------------
        call("fourier_sigma", lib.kd_fourier_sigma_f32, _ptr(self.c_sigma), _ptr(m.time_emb.weight), _ptr(ff), B, mw // 2)
        gemm("time_in_proj", ff, m.time_in_proj.weight, temb, B, mw, mw)
        if has_aug:
            aug_ff, aug_proj = torch.empty(B, mw, **f32), torch.empty(B, mw, **f32)
            self.keep += [aug_ff, aug_proj]
            call("fourier_aug", lib.kd_fourier_f32, _ptr(self.aug_in), _ptr(m.aug_emb.weight), _ptr(aug_ff), B, 9, mw // 2)
            gemm("aug_in_proj", aug_ff, m.aug_in_proj.weight, aug_proj, B, mw, mw)
            aug_term, aug_rows = aug_proj, 1
        else:
            # aug_cond = zeros  =>  FourierFeatures = [cos 0, sin 0] = [1..1, 0..0]: a constant vector
            z_ff, aug_const = torch.empty(1, mw, **f32), torch.empty(1, mw, **f32)
            self.keep += [z_ff, aug_const]
            zeros9 = torch.zeros(1, 9, **f32)
            self.keep.append(zeros9)
            call("fourier_aug0", lib.kd_fourier_f32, _ptr(zeros9), _ptr(m.aug_emb.weight), _ptr(z_ff), 1, 9, mw // 2)
            gemm("aug_in_proj0", z_ff, m.aug_in_proj.weight, aug_const, 1, mw, mw)
            aug_term, aug_rows = aug_const, 0
        map_term = None
        if has_mapping_cond:
            map_term = torch.empty(B, mw, **f32)
            self.keep.append(map_term)
            gemm("mapping_cond_in_proj", self.map_in, m.mapping_cond_in_proj.weight, map_term, B, mw, m.mapping_cond_dim)
        call("cond_sum", lib.kd_cond_sum_f32, _ptr(emb), _ptr(temb), _ptr(aug_term), aug_rows,
             _ptr(m.class_emb.weight) if has_class else None, _ptr(self.class_ids) if has_class else None,
             None if map_term is None else _ptr(map_term), B, mw)
        call("mapping.in_norm", lib.kd_rmsnorm_f32, _ptr(emb), _ptr(m.mapping.in_norm.scale), _ptr(mres), B, mw, C.c_float(1e-6))
        for blk in m.mapping.blocks:
            gemm("mapping.up_proj", mres, blk.up_proj.weight, mh, B, mdff, mw, epi=nat.EPI_GEGLU,
                 scale_ptr=blk.norm.scale.data_ptr(), scale_stride=0, rows_per_sample=B)
            gemm("mapping.down_proj", mh, blk.down_proj.weight, mres, B, mw, mdff, epi=nat.EPI_RESIDUAL, R=mres)
        call("mapping.out_norm", lib.kd_rmsnorm_f32, _ptr(mres), _ptr(m.mapping.out_norm.scale), _ptr(cond), B, mw, C.c_float(1e-6))
        self.d_scales = gemm("ada_norm_scales", cond, wcat, None, B, total, mw, out_add=1.0)

        # ---- hourglass ------------------------------------------------------------------------
        target = self.launches
        self.d_patch_in = gemm("patch_in", None, m.patch_in.proj.weight, xs[0], toks[0], levels[0].width, m.in_channels * ph * pw,
                               a_mode=nat.A_PATCH_NCHW, grid=grids[0], patch=(ph, pw, m.in_channels))

        def scale_ptr(name):
            return ("table", 4 * offsets[name])

        packed_qkv = precision == nat.PREC_SPLIT3 and os.environ.get("KDIFF_QKV_PACKED", "1") != "0"

        def add_layer(li, prefix, mod, index):
            lv, (gh, gw), T = levels[li], grids[li], toks[li]
            d, x = lv.width, xs[li]
            rps = gh * gw
            if hasattr(mod, "self_attn"):
                sa, spec = mod.self_attn, lv.self_attn
                nh = d // spec.d_head
                if bf:
                    # bf16 mode: the qkv epilogue evaluates the RoPE angles itself (hardware sin / cos) from the token's axial
                    # position and the head's frequencies in revolutions -- two tiny tables instead of cos / sin per (token, head)
                    cos_t, sin_t = m._rope_pos_freq(li, grids, sa, device)
                else:
                    cos_t, sin_t = m._rope_tables(li, grids, sa, device)
                self.keep += [cos_t, sin_t]
                # q, k leave the qkv GEMM already prepared (cosine-sim scale + RoPE in its epilogue): every halo /
                # window / key tile of the attention cores would otherwise redo that work per use
                # ... and, for the split-bf16x3 cores, already SPLIT (hi / lo bf16 chunks in the fp32 slots): the cores take
                # their operands as stored instead of converting every halo / window / key-block element again
                dq = gemm(prefix + "qkv_proj", x, sa.qkv_proj.weight, qkv, T, 3 * d, d, epi=nat.EPI_QKV,
                          scale_ptr=scale_ptr(prefix + "self_attn.norm"), scale_stride=total, rows_per_sample=rps,
                          qk=(sa.scale, cos_t, sin_t, nh))
                dq.qkv_packed = 1 if packed_qkv else 0
                prep = (2 if packed_qkv else 0, None, None, None, C.c_float(1e-6), precision)
                shift = 0
                if isinstance(spec, ShiftedWindowAttentionSpec):
                    shift = spec.window_size // 2 if index % 2 == 1 else 0          # :523
                if bf and isinstance(spec, GlobalAttentionSpec):
                    call(prefix + "attn_global", lib.kd_attn_global_bf16, _ptr(qkv), _ptr(att), B, gh * gw, nh)
                elif bf and isinstance(spec, NeighborhoodAttentionSpec):
                    call(prefix + "attn_na2d", lib.kd_attn_na2d_bf16, _ptr(qkv), _ptr(att), B, gh, gw, nh, spec.kernel_size)
                elif bf:
                    call(prefix + "attn_window", lib.kd_attn_window_bf16, _ptr(qkv), _ptr(att), B, gh, gw, nh, spec.window_size, shift)
                elif isinstance(spec, GlobalAttentionSpec):
                    call(prefix + "attn_global", lib.kd_attn_global_f32, _ptr(qkv), _ptr(att), B, gh * gw, nh, *prep)
                elif isinstance(spec, NeighborhoodAttentionSpec):
                    call(prefix + "attn_na2d", lib.kd_attn_na2d_f32, _ptr(qkv), _ptr(att), B, gh, gw, nh, spec.kernel_size, *prep)
                else:
                    call(prefix + "attn_window", lib.kd_attn_window_f32, _ptr(qkv), _ptr(att), B, gh, gw, nh, spec.window_size, shift, *prep)
                gemm(prefix + "out_proj", att, sa.out_proj.weight, x, T, d, d, epi=nat.EPI_RESIDUAL, R=x)
            if bf and target is self.launches and lib.kd_ffn_bf16_supported(T, d, lv.d_ff):
                # the whole FeedForwardBlock (:487-493) in one kernel: the d_ff-wide hidden activation stays on the chip
                fd = nat.KdFfn()
                fd.x = fd.out = x.data_ptr()
                fd.scale_stride, fd.rows_per_sample, fd.eps = total, rps, 1e-6
                fd.Wp_up = m._packed_image(mod.ff.up_proj.weight, lv.d_ff, d, 1, bf16=True).data_ptr()
                fd.Wp_down = m._packed_image(mod.ff.down_proj.weight, d, lv.d_ff, 2, bf16=True).data_ptr()
                fd.M, fd.K, fd.d_ff = T, d, lv.d_ff
                self.norm_descs.append((fd, scale_ptr(prefix + "ff.norm")[1]))
                self.keep.append(fd)
                target.append(_Launch(lib.kd_ffn_bf16, (C.byref(fd),), prefix + "ff"))
            else:
                gemm(prefix + "up_proj", x, mod.ff.up_proj.weight, hid, T, lv.d_ff, d, epi=nat.EPI_GEGLU,
                     scale_ptr=scale_ptr(prefix + "ff.norm"), scale_stride=total, rows_per_sample=rps)
                gemm(prefix + "down_proj", hid, mod.ff.down_proj.weight, x, T, d, lv.d_ff, epi=nat.EPI_RESIDUAL, R=x)

        for li in range(n_lv - 1):
            for i, mod in enumerate(m.down_levels[li]):
                add_layer(li, f"down_levels.{li}.{i}.", mod, i)
            gemm(f"merges.{li}", xs[li], m.merges[li].proj.weight, xs[li + 1], toks[li + 1], levels[li + 1].width, 4 * levels[li].width,
                 a_mode=nat.A_MERGE2x2, grid=grids[li + 1])
        for i, mod in enumerate(m.mid_level):
            add_layer(n_lv - 1, f"mid_level.{i}.", mod, i)
        for li in reversed(range(n_lv - 1)):
            gemm(f"splits.{li}", xs[li + 1], m.splits[li].proj.weight, xs[li], toks[li + 1], 4 * levels[li].width, levels[li + 1].width,
                 epi=nat.EPI_SPLIT_LERP, R=xs[li], fac=m.splits[li].fac, grid=grids[li + 1])
            for i, mod in enumerate(m.up_levels[li]):
                add_layer(li, f"up_levels.{li}.{i}.", mod, i + levels[li].depth)                      # :697
        self.d_patch_out = gemm("patch_out", xs[0], m.patch_out.proj.weight, None, toks[0], m.out_channels * ph * pw, levels[0].width,
                                epi=nat.EPI_UNPATCH_NCHW, scale_ptr=m.out_norm.scale.data_ptr(), scale_stride=0,
                                rows_per_sample=grids[0][0] * grids[0][1], grid=grids[0], patch=(ph, pw, m.out_channels))

    def run_cond(self, buf, stream):
        """Conditioning chain (FourierFeatures -> mapping network -> every AdaRMSNorm scale of the network) into scale
        table ``buf`` on ``stream``; its inputs ``c_sigma`` / ``class_ids`` / ``aug_in`` / ``map_in`` were filled by the
        caller on the same stream."""
        self.d_scales.C = self.scales[buf].data_ptr()
        for ln in self.cond_launches:
            rc = ln.fn(*ln.args, stream)
            if rc:
                nat.check(rc, ln.what)

    def run(self, x, out, sigma_data, buf):
        """Main chain on the current stream.  x: input image (read by patch_in and, when preconditioning, by patch_out);
        out: result image; ``self.sigma`` was filled by the caller; scale table ``buf`` holds this step's scales."""
        base = self.scales[buf].data_ptr()
        for d, off in self.norm_descs:
            d.scale = base + off
        pin, pout = self.d_patch_in, self.d_patch_out
        pin.A = x.data_ptr()
        pout.C = out.data_ptr()
        if sigma_data is None:
            pin.sigma, pout.sigma, pout.R = None, None, None
        else:
            sp = self.sigma.data_ptr()
            pin.sigma, pout.sigma, pout.R = sp, sp, x.data_ptr()
            pin.sigma_data = pout.sigma_data = float(sigma_data)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for ln in self.launches:
            rc = ln.fn(*ln.args, stream)
            if rc:
                nat.check(rc, ln.what)

    # ---- hipGraph replay (launch-bound batch sizes only: ``use_graph``) ---------------------------------------------------
    def _capture(self, issue):
        # capture_begin / capture_end rather than the torch.cuda.graph context: that one synchronises the device, runs the
        # garbage collector and empties the allocator cache first, none of which a capture of pure kernel launches needs
        # (the library allocates nothing and never syncs; nothing executes during capture)
        g = torch.cuda.CUDAGraph()
        if self.capture_stream is None:
            self.capture_stream = torch.cuda.Stream(device=self.sigma.device)
        with torch.cuda.stream(self.capture_stream):
            g.capture_begin(capture_error_mode="thread_local")
            try:
                issue()
            finally:
                g.capture_end()
        return g

    def _graph_epoch(self):
        """Kernel selection inside the library follows its options (kd_set_option): graphs captured under other settings are dropped."""
        if self.graph_epoch != nat.option_epoch:
            self.graphs, self.cond_graphs, self.graph_epoch = {}, {}, nat.option_epoch
            self.direct_runs = self.direct_cond_runs = 0          # another kernel family may see its first launch now

    def replay(self, x, sigma_data, buf):
        """``run`` through a captured graph.  Kernel arguments are frozen at capture, so the chain reads a fixed input image
        and writes a fixed output image (two small copies per forward at these sizes) and there is one graph per
        (scale table, preconditioning) combination.  The first forward of a plan is issued directly (one-time kernel
        attribute set-up must not happen under capture)."""
        if self.g_x is None:
            self.g_x = torch.empty_like(x)
            self.g_out = torch.empty(self.out_shape, device=x.device, dtype=torch.float32)
        self.g_x.copy_(x, non_blocking=True)
        self._graph_epoch()
        key = (buf, None if sigma_data is None else float(sigma_data))
        g = self.graphs.get(key)
        if g is None and self.direct_runs == 0:
            self.direct_runs += 1
            self.run(self.g_x, self.g_out, sigma_data, buf)
        else:
            if g is None:
                g = self.graphs[key] = self._capture(lambda: self.run(self.g_x, self.g_out, sigma_data, buf))
            g.replay()
        return self.g_out.clone()

    def replay_cond(self, buf):
        """``run_cond`` on the current stream, replayed from a graph after the first direct pass."""
        self._graph_epoch()
        g = self.cond_graphs.get(buf)
        if g is None and self.direct_cond_runs == 0:
            self.direct_cond_runs += 1
            self.run_cond(buf, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            return
        if g is None:
            g = self.cond_graphs[buf] = self._capture(lambda: self.run_cond(buf, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        g.replay()


# ---------------------------------------------------------------------------------- the model

class ImageTransformerDenoiserModelV2(nn.Module):
    def __init__(self, levels, mapping, in_channels, out_channels, patch_size, num_classes=0, mapping_cond_dim=0):
        super().__init__()
        for lv in levels:
            sa = lv.self_attn
            if not isinstance(sa, (GlobalAttentionSpec, NeighborhoodAttentionSpec, ShiftedWindowAttentionSpec, NoAttentionSpec)):
                raise ValueError(f"unsupported self attention spec {sa}")
            if not isinstance(sa, NoAttentionSpec) and sa.d_head != D_HEAD:
                raise ValueError(f"the HIP attention cores are built for d_head == {D_HEAD} (got {sa.d_head})")
            if not isinstance(sa, NoAttentionSpec) and lv.width % sa.d_head:
                raise ValueError(f"width {lv.width} is not a multiple of d_head {sa.d_head}")
        self.level_specs, self.mapping_spec = list(levels), mapping
        self.in_channels, self.out_channels = in_channels, out_channels
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.num_classes, self.mapping_cond_dim = num_classes, mapping_cond_dim
        ph, pw = self.patch_size
        mw = mapping.width

        self.patch_in = _Holder(proj=_linear_weight(levels[0].width, in_channels * ph * pw))
        self.time_emb = _Holder(weight=torch.randn(mw // 2, 1))
        self.time_in_proj = _linear_weight(mw, mw)
        self.aug_emb = _Holder(weight=torch.randn(mw // 2, 9))
        self.aug_in_proj = _linear_weight(mw, mw)
        self.class_emb = _Holder(weight=nn.Parameter(torch.randn(num_classes, mw))) if num_classes else None
        self.mapping_cond_in_proj = _linear_weight(mw, mapping_cond_dim) if mapping_cond_dim else None
        self.mapping = _Holder(
            in_norm=_rms_scale(mw),
            blocks=nn.ModuleList([_Holder(norm=_rms_scale(mw), up_proj=_linear_weight(2 * mapping.d_ff, mw),
                                          down_proj=_linear_weight(mw, mapping.d_ff, zero=True)) for _ in range(mapping.depth)]),
            out_norm=_rms_scale(mw))
        self.down_levels, self.up_levels = nn.ModuleList(), nn.ModuleList()
        for i, lv in enumerate(levels):
            if i < len(levels) - 1:
                self.down_levels.append(nn.ModuleList([_layer(lv, mw) for _ in range(lv.depth)]))
                self.up_levels.append(nn.ModuleList([_layer(lv, mw) for _ in range(lv.depth)]))
            else:
                self.mid_level = nn.ModuleList([_layer(lv, mw) for _ in range(lv.depth)])
        self.merges = nn.ModuleList([_Holder(proj=_linear_weight(b.width, 4 * a.width)) for a, b in zip(levels[:-1], levels[1:])])
        self.splits = nn.ModuleList([_Holder(proj=_linear_weight(4 * a.width, b.width), fac=nn.Parameter(torch.ones(1) * 0.5))
                                     for a, b in zip(levels[:-1], levels[1:])])
        self.out_norm = _rms_scale(levels[0].width)
        self.patch_out = _Holder(proj=_linear_weight(out_channels * ph * pw, levels[0].width, zero=True))
        self._plans, self._fingerprint, self._packed = {}, None, {}

    # ---- bookkeeping ---------------------------------------------------------------------------
    def _ada_norm_modules(self):
        out = []

        def visit(prefix, layers):
            for i, mod in enumerate(layers):
                if hasattr(mod, "self_attn"):
                    out.append((f"{prefix}{i}.self_attn.norm", mod.self_attn.norm))
                out.append((f"{prefix}{i}.ff.norm", mod.ff.norm))
        for li, lvl in enumerate(self.down_levels):
            visit(f"down_levels.{li}.", lvl)
        for li, lvl in enumerate(self.up_levels):
            visit(f"up_levels.{li}.", lvl)
        visit("mid_level.", self.mid_level)
        return out

    def _rope_tables(self, li, grids, sa, device):
        h0, w0 = grids[0]
        pos = axial_rope.make_axial_pos(h0, w0).view(h0, w0, 2)
        for _ in range(li):
            pos = axial_rope.downscale_pos(pos)
        cos_t, sin_t = axial_rope.rope_tables(pos, sa.pos_emb.freqs)
        return cos_t.to(device), sin_t.to(device)

    def _rope_pos_freq(self, li, grids, sa, device):
        """bf16 mode: ([tokens, 2] axial positions (y, x) of the level's grid, [nh, 8] frequencies in revolutions)."""
        h0, w0 = grids[0]
        pos = axial_rope.make_axial_pos(h0, w0).view(h0, w0, 2)
        for _ in range(li):
            pos = axial_rope.downscale_pos(pos)
        freq = sa.pos_emb.freqs.detach().to(torch.float32).cpu() / (2.0 * math.pi)
        return pos.reshape(-1, 2).to(torch.float32).contiguous().to(device), freq.contiguous().to(device)

    def _packed_image(self, W, N, K, geglu, bf16=False):
        """Packed image of a weight (split-bf16, or plain bf16 for the bf16 mode), shared by all plans of this model.  The
        entry keeps the source tensor alive, so its address cannot be recycled under the cached image; the dict is dropped
        with the plans whenever the weights change (``_weights_fingerprint``)."""
        key = (id(W), N, K, int(geglu), bool(bf16))
        ent = self._packed.get(key)
        if ent is None:
            ent = self._packed[key] = (W, ops.pack_weight(W, N, K, geglu, cache=False, bf16=bf16))
        return ent[1]

    def _weights_fingerprint(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def param_groups(self, base_lr=5e-4, mapping_lr_scale=1 / 3):
        raise NotImplementedError("training is outside this package's scope (sampling hot path only)")

    # ---- forward ---------------------------------------------------------------------------------
    def forward(self, x, sigma, aug_cond=None, class_cond=None, mapping_cond=None):
        """Inner model F(x, sigma): [B, C, H, W] fp32 on a ROCm device -> [B, C, H, W]."""
        return self._run(x, sigma, aug_cond, class_cond, mapping_cond, None)

    def forward_preconditioned(self, x, sigma, sigma_data, aug_cond=None, class_cond=None, mapping_cond=None):
        """Denoiser D(x, sigma) = F(x * c_in, sigma) * c_out + x * c_skip (k_diffusion/layers.py:88-90)
        with c_in folded into the patch gather and c_out / c_skip into the un-patch scatter."""
        return self._run(x, sigma, aug_cond, class_cond, mapping_cond, sigma_data)

    @torch.no_grad()
    def _run(self, x, sigma, aug_cond, class_cond, mapping_cond, sigma_data):
        if class_cond is None and self.class_emb is not None:
            raise ValueError("class_cond must be specified if num_classes > 0")
        if mapping_cond is None and self.mapping_cond_in_proj is not None:
            raise ValueError("mapping_cond must be specified if mapping_cond_dim > 0")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected input [B, {self.in_channels}, H, W], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("ImageTransformerDenoiserModelV2 runs on the HIP path only: move the model and inputs "
                               "to a ROCm device (there is no CPU fallback)")
        if x.dtype != torch.float32:
            raise TypeError(f"fp32 inputs only (got {x.dtype})")
        x = x.contiguous()
        B, _, H, W = x.shape
        fp = self._weights_fingerprint()
        if fp != self._fingerprint:
            self._plans, self._fingerprint, self._packed = {}, fp, {}
        has_class = self.class_emb is not None
        key = (B, H, W, aug_cond is not None, has_class, self.mapping_cond_in_proj is not None, x.device, nat.default_precision(),
               os.environ.get("KDIFF_QKV_PACKED", "1"), os.environ.get("KDIFF_GRAPH", "auto"))
        plan = self._plans.get(key)
        if plan is None:
            if self.patch_in.proj.weight.device != x.device:
                raise RuntimeError(f"model weights are on {self.patch_in.proj.weight.device}, input on {x.device}")
            plan = self._plans[key] = _Plan(self, B, H, W, key[3], has_class, key[5], x.device)
            if has_class:
                # nn.Embedding raises on an out-of-range id; the HIP kernel would read past the table.  Checked once per plan
                # (one device->host read), not per call: the per-step path stays sync-free.
                lo, hi = int(class_cond.min()), int(class_cond.max())
                if lo < 0 or hi >= self.class_emb.weight.shape[0]:
                    del self._plans[key]
                    raise IndexError(f"class_cond ids must lie in [0, {self.class_emb.weight.shape[0] - 1}] (got {lo}..{hi})")
        cur = torch.cuda.current_stream()
        graphed = plan.use_graph and not nat.prof_active and not torch.cuda.is_current_stream_capturing()
        ident = self._cond_identity(sigma, aug_cond, class_cond, mapping_cond)
        pre, plan.prefetched = plan.prefetched, None
        if pre is not None and pre[0] == ident:
            buf = pre[1]                          # this step's scale table was computed ahead of time on the side stream
            cur.wait_event(pre[2])
        else:
            buf = 1 - plan.last_buf
            if pre is not None:
                cur.wait_event(pre[2])            # an unused prefetch still owns the conditioning workspace: order behind it
            self._fill_cond_inputs(plan, B, sigma, aug_cond, class_cond, mapping_cond)
            if graphed:
                plan.replay_cond(buf)
            else:
                plan.run_cond(buf, C.c_void_p(cur.cuda_stream))
        plan.last_buf = buf
        plan.sigma.copy_(sigma.reshape(-1).expand(B) if sigma.numel() == 1 else sigma.reshape(B), non_blocking=True)
        plan.main_entry.record(cur)               # everything before this step's main chain (incl. an inline conditioning chain)
        if graphed:
            return plan.replay(x, sigma_data, buf)
        out = torch.empty(B, self.out_channels, H, W, device=x.device, dtype=torch.float32)
        plan.run(x, out, sigma_data, buf)
        return out

    # ---- conditioning ahead of time ---------------------------------------------------------------
    def _cond_identity(self, sigma, aug_cond, class_cond, mapping_cond):
        return tuple(None if t is None else (t.data_ptr(), t._version, tuple(t.shape)) for t in (sigma, aug_cond, class_cond, mapping_cond))

    def _fill_cond_inputs(self, plan, B, sigma, aug_cond, class_cond, mapping_cond):
        plan.c_sigma.copy_(sigma.reshape(-1).expand(B) if sigma.numel() == 1 else sigma.reshape(B), non_blocking=True)
        if self.class_emb is not None:
            plan.class_ids.copy_(class_cond.reshape(B), non_blocking=True)
        if plan.aug_in is not None:
            plan.aug_in.copy_(aug_cond.reshape(B, 9), non_blocking=True)
        if plan.map_in is not None:
            plan.map_in.copy_(mapping_cond.reshape(B, self.mapping_cond_dim), non_blocking=True)

    @torch.no_grad()
    def prefetch_conditioning(self, x_like, sigma, aug_cond=None, class_cond=None, mapping_cond=None):
        """Hint from the solver loop: the NEXT model call will use exactly these conditioning tensors.  The conditioning
        chain (it depends on sigma / class / aug / mapping_cond only, never on x) then runs on a side HIP stream,
        concurrently with the main chain of the step in flight, into the other scale table; the next ``forward`` with the
        same tensors just waits for its event.  A hint that is not followed costs nothing but the side-stream work."""
        if not x_like.is_cuda:
            return
        B, _, H, W = x_like.shape
        key = (B, H, W, aug_cond is not None, self.class_emb is not None, self.mapping_cond_in_proj is not None, x_like.device,
               nat.default_precision(), os.environ.get("KDIFF_QKV_PACKED", "1"), os.environ.get("KDIFF_GRAPH", "auto"))
        plan = self._plans.get(key)
        if plan is None or plan.prefetched is not None or self._weights_fingerprint() != self._fingerprint:
            return
        if (class_cond is None and self.class_emb is not None) or (mapping_cond is None and self.mapping_cond_in_proj is not None):
            return
        buf = 1 - plan.last_buf
        side = plan.side_stream
        side.wait_event(plan.main_entry)          # table `buf` and the conditioning workspace are free once the main chain of
        with torch.cuda.stream(side):             # the step in flight has started (its predecessors are complete in stream order)
            self._fill_cond_inputs(plan, B, sigma, aug_cond, class_cond, mapping_cond)
            if plan.use_graph and not nat.prof_active:
                plan.replay_cond(buf)
            else:
                plan.run_cond(buf, C.c_void_p(side.cuda_stream))
            done = torch.cuda.Event()
            done.record(side)
        # the record keeps the hinted tensors alive: while it is pending their storage cannot be freed and handed to another
        # tensor, so a later call whose (address, version, shape) key matches really is the same data (a sampler aborted
        # mid-loop leaves a record behind; the next run's schedule then necessarily lives at other addresses)
        plan.prefetched = (self._cond_identity(sigma, aug_cond, class_cond, mapping_cond), buf, done, (sigma, aug_cond, class_cond, mapping_cond))
