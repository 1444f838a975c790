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
import gc
import itertools
import operator
import weakref
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
# Largest [steps, B, scale_width] scale table kept per sigma schedule (prefetch_schedule); longer schedules use the per-step chain.
SCHEDULE_TABLE_MAX_BYTES = 1 << 30
SCHEDULES_KEPT = 4            # a run of a two-stage solver hints two tables; older records (and their tensors) are dropped
SCHEDULE_CHAINS_KEPT = 2      # conditioning workspaces kept per plan, by schedule length (least recently used dropped)
# environment switches read while a plan is built (name, default): part of the plan key
MAX_PLANS = 16     # cached launch plans (one per batch / size / conditioning kinds / device / arithmetic mode / switches) per model: least recently used beyond that
PLAN_SWITCHES = (("KDIFF_ATTN_BLOCK", "1"), ("KDIFF_PROJ_BLOCK", "1"), ("KDIFF_FFN_OUT", "all"), ("KDIFF_RUN_LIST", "1"))
CLASS_IDS_KEPT = 4            # range-checked class_cond tensors remembered per plan (cond / uncond pairs of a guidance wrapper)


_untracked = itertools.count(-1, -1)

def _ver(t):
    """Version counter of a tensor, or a value that never repeats for tensors made under torch.inference_mode() (they carry no
    counter, so an in-place change cannot be seen: such a tensor is never recognised as 'the same data as last time')."""
    return next(_untracked) if t.is_inference() else t._version


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
    __slots__ = ("fn", "args", "what", "enc")

    def __init__(self, fn, args, what, enc=None):
        # enc: (entry point name, its arguments without the stream) for a kd_run_list entry; None = only callable from Python
        self.fn, self.args, self.what, self.enc = fn, args, what, enc


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class _CondChain:
    """Launch list + buffers of the conditioning chain for a fixed number of rows (one row per sample and model call).
    Every kernel of the chain works row by row (the products go through the per-row fp32 FMA kernel, KdGemm.per_row), so
    a row's scales do not depend on how many rows share the launches: B rows per solver step or steps x B rows at once."""

    def __init__(self, rows):
        self.rows, self.launches, self.keep = rows, [], []
        self.c_sigma = self.class_ids = self.aug_in = self.map_in = self.d_scales = None

    def fill(self, sigma, aug_cond, class_cond, mapping_cond, repeat=1):
        """Inputs of the chain on the current stream.  ``sigma``: [rows] (or one value); the other tensors are per sample
        ([rows / repeat, ...]) and repeated for every step of a schedule."""
        rows = self.rows
        self.c_sigma.copy_(sigma.reshape(-1).expand(rows) if sigma.numel() == 1 else sigma.reshape(rows), non_blocking=True)
        B = rows // repeat
        if self.class_ids is not None and class_cond is not None:
            self.class_ids.view(repeat, B).copy_(class_cond.reshape(1, B).expand(repeat, B), non_blocking=True)
        if self.aug_in is not None:
            self.aug_in.view(repeat, B, 9).copy_(aug_cond.reshape(1, B, 9).expand(repeat, B, 9), non_blocking=True)
        if self.map_in is not None:
            self.map_in.view(repeat, B, -1).copy_(mapping_cond.reshape(1, B, -1).expand(repeat, B, -1), non_blocking=True)

    def run(self, table_ptr, stream):
        """Scales of every AdaRMSNorm of the network for every row -> [rows, scale_width] fp32 at ``table_ptr``."""
        self.d_scales.C = table_ptr
        for ln in self.launches:
            rc = ln.fn(*ln.args, stream)
            if rc:
                nat.check(rc, ln.what)


class _Schedule:
    """Scale tables of a whole sigma schedule: rows of ``sigma_table`` ([n, B], one row per model call) -> tables[i]."""

    def __init__(self, sigma_table, others, ident_others, tables, done):
        self.base, self.version, self.n, self.B = sigma_table.data_ptr(), _ver(sigma_table), sigma_table.shape[0], sigma_table.shape[1]
        self.ident_others, self.tables, self.done = ident_others, tables, done
        self.waited = set()       # streams (handles) that have waited for ``done`` once: everything they run later is ordered behind it
        # the record keeps the hinted tensors alive, so their addresses cannot be handed to other tensors while it exists
        self.keep = (sigma_table, others)

    def row_of(self, sigma):
        """Index of the table row ``sigma`` is a view of, or None."""
        if sigma.dtype != torch.float32 or sigma.numel() != self.B or not sigma.is_contiguous() or _ver(sigma) != self.version:
            return None
        off = sigma.data_ptr() - self.base
        if off < 0 or off % (4 * self.B) or off // (4 * self.B) >= self.n:
            return None
        return off // (4 * self.B)


class _Plan:
    """Workspace + prebuilt launch list for one (batch, H, W, conditioning-kinds) combination."""

    def release(self):
        """Give the workspaces back NOW: a plan's launch lists and conditioning chains close over the plan and over each other (reference
        cycles), so dropping the last outside reference would leave ~20 MB per image allocated until Python's cycle collector happens to run.
        The callers (eviction, ``_drop_plans``: rare events) run the collector once behind this for the helper objects' own cycles."""
        self.__dict__.clear()

    def __init__(self, model, B, H, W, has_aug, has_class, has_mapping_cond, device):
        lib = nat.lib()
        m = model
        precision = nat.default_precision()
        # fp8 mode: the bf16 plan with the norm -> qkv / norm -> GEGLU projections of the K = 256 / 512 levels on the fp8 matrix instruction
        fp8 = precision == nat.PREC_FP8
        precision = nat.PREC_BF16 if fp8 else precision
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
        self.class_checked = {}                             # identities of the range-checked class_cond tensors (_plan_for) -> the tensor
        mw, mdff = m.mapping_spec.width, m.mapping_spec.d_ff

        # ---- static buffers -----------------------------------------------------------------
        self.sigma = torch.empty(B, **f32)                  # preconditioning sigmas of the MAIN chain when the caller's cannot be read in place
        self.sigma_ptr = self.sigma.data_ptr()
        xs = [torch.empty(B, gh, gw, lv.width, **act) for (gh, gw), lv in zip(grids, levels)]
        toks = [B * gh * gw for gh, gw in grids]
        qkv = torch.empty(max(t * 3 * lv.width for t, lv in zip(toks, levels)), **act)
        att = torch.empty(max(t * lv.width for t, lv in zip(toks, levels)), **act)
        hid = torch.empty(max(t * lv.d_ff for t, lv in zip(toks, levels)), **act)
        # fp32-parity mode, round 3: GEMM operands that a producer can split once travel as two bf16 planes (hi, lo: the same bytes as
        # fp32) and the consumer GEMM moves them by LDS-DMA (csrc/gemm_x3t.hip).  Planes of the FF hidden activation live in `hid`
        # (hi in its first half, lo in the second); the normalised rows of the levels whose width exceeds the fused norm -> projection
        # kernel's register budget (> 256) get planes of their own (`xn`, written by kd_norm_split_f32).
        planes = precision == nat.PREC_SPLIT3
        def prepass(width):                                  # widths the fused norm -> projection kernel (gemm_x3.hip) does not take
            return planes and width not in (128, 256, 512) and width > 256 and width % 128 == 0 and width <= 2048    # (tiles of 128 features)
        wide = [t * lv.width for t, lv in zip(toks, levels) if prepass(lv.width)]
        xn = torch.empty(max(wide), **f32) if wide else None
        self.keep.append(xn)
        norm_mods = m._ada_norm_modules()
        offsets, total = {}, 0
        for name, mod in norm_mods:
            offsets[name] = total
            total += mod.linear.weight.shape[0]
        # one concatenation of the AdaRMSNorm projections per MODEL, not per plan: it is a function of the weights only, and the packed-image
        # cache keeps its source alive -- made per plan, every batch size ever seen left 2 x 7.5 MB behind (256 x 256 configs) until the
        # weights changed.  (model._packed goes with the plans whenever the weights do.)
        wcat_ent = m._packed.get("ada_norm_wcat")
        if wcat_ent is None or wcat_ent.device != device:
            wcat_ent = m._packed["ada_norm_wcat"] = torch.cat([mod.linear.weight.detach() for _, mod in norm_mods], dim=0).contiguous()
        wcat = wcat_ent
        # AdaRMSNorm scale tables, ping-pong: the main chain reads one while the next step's table is being written
        self.scales = [torch.empty(B, total, **f32), torch.empty(B, total, **f32)]
        self.norm_descs = []                                # (descriptor, byte offset into a scale table)
        self.last_buf, self.prefetched = 1, None            # prefetched: (identity of the conditioning tensors, table, done event)
        self.side_stream = torch.cuda.Stream(device=device)
        self.main_entry = torch.cuda.Event()
        self.schedules, self.schedule_chains = [], {}       # conditioning of whole sigma schedules (prefetch_schedule); chains by length
        self.keep += [xs, qkv, att, hid, wcat]
        self.xs = xs

        def gemm(what, A, Wt, Cc, M, N, K, a_mode=nat.A_PLAIN, epi=nat.EPI_STORE, scale_ptr=None, scale_stride=0,
                 rows_per_sample=0, R=None, grid=(0, 0), patch=(0, 0, 0), out_add=0.0, sigma=None, fac=None, qk=None,
                 a_planes=None, c_planes=None, mx8=False):
            d = nat.KdGemm()
            d.M, d.N, d.K, d.a_mode, d.epi = M, N, K, a_mode, epi
            main = target is self.launches
            d.precision = precision if main else cond_precision
            d.per_row = 0 if main else 1                    # conditioning products: one row per sample, see _CondChain
            if mx8:
                d.Wp = m._packed_image(Wt, N, K, epi == nat.EPI_GEGLU, bf16="mx8").data_ptr()
            elif d.precision == nat.PREC_BF16:
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
            if a_planes is not None:                        # (hi address, lo address): pre-split bf16 planes instead of fp32 A
                d.a_split, d.A, d.A_lo = 1, a_planes[0], a_planes[1]
            if c_planes is not None:
                d.c_split, d.C, d.C_lo = 1, c_planes[0], c_planes[1]
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
                if len(qk) >= 6:                                     # split3: positions / frequencies for the round-3 kernels (gemm_x3*.hip)
                    d.rope_pos, d.rope_freq = qk[4].data_ptr(), qk[5].data_ptr()
            (self.keep if main else chain.keep).append(d)
            name = "kd_gemm_mx8" if mx8 else ("kd_gemm_bf16" if d.precision == nat.PREC_BF16 else "kd_gemm_f32")
            target.append(_Launch(getattr(lib, name), (C.byref(d),), what + ("(mx8)" if mx8 else ""), enc=(name, (d,))))
            return d

        def mx8_ok(M, N, K, epi):
            """fp8 mode: this norm -> projection of the main chain goes to the block-scaled fp8 matrix instruction (kd_gemm_mx8)."""
            # (from 4 096 rows on -- library option mx8_min_rows: below that the few-rows bf16 kernels are ahead -- batch 1: 0.512 against 0.552 ms
            # per forward, profiles/r06_bench_detail_full.json)
            return fp8 and target is self.launches and M >= lib.kd_get_option(b"mx8_min_rows", 4096) and bool(lib.kd_gemm_mx8_supported(M, N, K, epi, 1))

        def call(what, fn, *args):
            target.append(_Launch(fn, args, what, enc=(fn.__name__, args)))

        def build_cond(rows):
            """The conditioning chain (image_transformer_v2.py:734-740, :569-581) for ``rows`` rows: FourierFeatures ->
            mapping network -> every AdaRMSNorm scale of the network, with its own input and work buffers."""
            nonlocal target, chain
            ch = _CondChain(rows)
            saved, target, chain = target, ch.launches, ch
            try:
                ch.c_sigma = torch.empty(rows, **f32)
                ch.class_ids = torch.zeros(rows, device=device, dtype=torch.int64)
                ch.aug_in = torch.zeros(rows, 9, **f32) if has_aug else None
                ch.map_in = torch.zeros(rows, m.mapping_cond_dim, **f32) if has_mapping_cond else None
                ff, temb, emb, mres, cond = (torch.empty(rows, mw, **f32) for _ in range(5))
                mh = torch.empty(rows, mdff, **f32)
                ch.keep += [ff, temb, emb, mres, cond, mh]
                call("fourier_sigma", lib.kd_fourier_sigma_f32, _ptr(ch.c_sigma), _ptr(m.time_emb.weight), _ptr(ff), rows, mw // 2)
                gemm("time_in_proj", ff, m.time_in_proj.weight, temb, rows, mw, mw)
                if has_aug:
                    aug_ff, aug_proj = torch.empty(rows, mw, **f32), torch.empty(rows, mw, **f32)
                    ch.keep += [aug_ff, aug_proj]
                    call("fourier_aug", lib.kd_fourier_f32, _ptr(ch.aug_in), _ptr(m.aug_emb.weight), _ptr(aug_ff), rows, 9, mw // 2)
                    gemm("aug_in_proj", aug_ff, m.aug_in_proj.weight, aug_proj, rows, mw, mw)
                    aug_term, aug_rows = aug_proj, 1
                else:
                    # aug_cond = zeros  =>  FourierFeatures = [cos 0, sin 0] = [1..1, 0..0]: a constant vector
                    z_ff, aug_const = torch.empty(1, mw, **f32), torch.empty(1, mw, **f32)
                    zeros9 = torch.zeros(1, 9, **f32)
                    ch.keep += [z_ff, aug_const, zeros9]
                    call("fourier_aug0", lib.kd_fourier_f32, _ptr(zeros9), _ptr(m.aug_emb.weight), _ptr(z_ff), 1, 9, mw // 2)
                    gemm("aug_in_proj0", z_ff, m.aug_in_proj.weight, aug_const, 1, mw, mw)
                    aug_term, aug_rows = aug_const, 0
                map_term = None
                if has_mapping_cond:
                    map_term = torch.empty(rows, mw, **f32)
                    ch.keep.append(map_term)
                    gemm("mapping_cond_in_proj", ch.map_in, m.mapping_cond_in_proj.weight, map_term, rows, mw, m.mapping_cond_dim)
                call("cond_sum", lib.kd_cond_sum_f32, _ptr(emb), _ptr(temb), _ptr(aug_term), aug_rows,
                     _ptr(m.class_emb.weight) if has_class else None, _ptr(ch.class_ids) if has_class else None,
                     None if map_term is None else _ptr(map_term), rows, mw)
                call("mapping.in_norm", lib.kd_rmsnorm_f32, _ptr(emb), _ptr(m.mapping.in_norm.scale), _ptr(mres), rows, mw, C.c_float(1e-6))
                for blk in m.mapping.blocks:
                    gemm("mapping.up_proj", mres, blk.up_proj.weight, mh, rows, mdff, mw, epi=nat.EPI_GEGLU,
                         scale_ptr=blk.norm.scale.data_ptr(), scale_stride=0, rows_per_sample=rows)
                    gemm("mapping.down_proj", mh, blk.down_proj.weight, mres, rows, mw, mdff, epi=nat.EPI_RESIDUAL, R=mres)
                call("mapping.out_norm", lib.kd_rmsnorm_f32, _ptr(mres), _ptr(m.mapping.out_norm.scale), _ptr(cond), rows, mw, C.c_float(1e-6))
                ch.d_scales = gemm("ada_norm_scales", cond, wcat, None, rows, total, mw, out_add=1.0)
            finally:
                target, chain = saved, None
            return ch

        target = chain = None
        self.build_cond = build_cond
        self.scale_width = total
        self.step_chain = build_cond(B)                     # the per-step chain (inline, or one step ahead on the side stream)

        # ---- hourglass ------------------------------------------------------------------------
        target = self.launches
        self.d_patch_in = gemm("patch_in", None, m.patch_in.proj.weight, xs[0], toks[0], levels[0].width, m.in_channels * ph * pw,
                               a_mode=nat.A_PATCH_NCHW, grid=grids[0], patch=(ph, pw, m.in_channels))

        def scale_ptr(name):
            return ("table", 4 * offsets[name])

        class _ScaleRef:                                    # patched per run like the descriptors in norm_descs
            def __init__(self):
                self._scale, self._call, self._index = None, None, 0

            def bind_call(self, call, index):               # (nat.encode_call: this pointer lives in a kd_run_list entry too)
                self._call, self._index = call, index

            @property
            def scale(self):
                return self._scale

            @scale.setter
            def scale(self, v):
                self._scale = v
                if self._call is not None:
                    self._call.p[self._index] = v

        def norm_split(what, x_t, table_off, T_, d_, rps_):
            """AdaRMSNorm of the fp32 rows of ``x_t`` -> (hi, lo) bf16 planes in ``xn`` (kd_norm_split_f32)."""
            ref = _ScaleRef()
            self.norm_descs.append((ref, table_off))
            hi_p, lo_p = xn.data_ptr(), xn.data_ptr() + 2 * T_ * d_
            xp = x_t.data_ptr()
            target.append(_Launch(lambda stream, ref=ref: lib.kd_norm_split_f32(xp, ref.scale, total, rps_, hi_p, lo_p, T_, d_, 1e-6, stream),
                                  (), what + " (split)", enc=("kd_norm_split_f32", (xp, ref, total, rps_, hi_p, lo_p, T_, d_, 1e-6))))
            return (hi_p, lo_p)

        packed_qkv = precision == nat.PREC_SPLIT3

        def add_layer(li, prefix, mod, index):
            lv, (gh, gw), T = levels[li], grids[li], toks[li]
            d, x = lv.width, xs[li]
            rps = gh * gw
            ffn_x3 = precision == nat.PREC_SPLIT3 and target is self.launches and lib.kd_ffn_f32_supported(T, d, lv.d_ff)     # (library option ffn_x3)
            # out projection fused into the FF kernel: width 128 (+3.5 % images/s in round 3) and, since round 5, width 256 too: in round 3
            # that measured level (176.1 vs 176.0: one wave per SIMD there); with the round-4 / 5 kernels around it the removed launch + the
            # attention rows' HBM round trip are worth +1.5 .. +3.0 % on two boxes (same box, back to back: 207.9 / 207.9 / 208.7 vs 214.3;
            # 186.9 / 186.9 / 187.0 vs 189.4 / 190.0).  KDIFF_FFN_OUT: 0 never, 1 = width 128 only, all (default) = every width the kernel
            # takes, or ONE width
            fo = os.environ.get("KDIFF_FFN_OUT", "all")
            ffn_bf = bf and target is self.launches and bool(lib.kd_ffn_bf16_supported(T, d, lv.d_ff))
            fuse_out = hasattr(mod, "self_attn") and ((ffn_x3 and d in (128, 256)) or (ffn_bf and d == 128)) \
                and (d == 128 if fo == "1" else fo in ("all", str(d)))
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
                qk = (sa.scale, cos_t, sin_t, nh)
                if precision == nat.PREC_SPLIT3:
                    # the split3 projections of round 3 evaluate the angles like the bf16 ones (no table loads beside their LDS-DMA ring);
                    # the tables stay for the round-1 kernels they fall back to (ragged shapes) and for the exact mode
                    pos_t, freq_t = m._rope_pos_freq(li, grids, sa, device)
                    self.keep += [pos_t, freq_t]
                    qk += (pos_t, freq_t)
                # q, k leave the qkv GEMM already prepared (cosine-sim scale + RoPE in its epilogue): every halo /
                # window / key tile of the attention cores would otherwise redo that work per use
                # ... and, for the split-bf16x3 cores, already SPLIT (hi / lo bf16 chunks in the fp32 slots): the cores take
                # their operands as stored instead of converting every halo / window / key-block element again
                # (the tiled qkv epilogue keeps its per-head constants in a 16-head LDS table, csrc/gemm_x3t.hip: wider levels -- 1152 =
                # 18 heads and up -- take the fp32-A norm -> projection kernels of round 1 like every shape the fused kernels refuse)
                # bf16 mode, global attention at 256 tokens per sample: norm -> qkv projection of a head -> cosine-sim + RoPE -> attention in
                # ONE launch per layer (csrc/block_bf16.hip: attn_block_bf16_kernel; q, k, v never reach HBM).  The descriptor is the qkv
                # projection's, its C the attention output
                # From 32 (sample, head) workgroups on: below that (batch 1 - 2 at level 2) the few-rows projection + the dense core are faster
                # (0.514 against 0.543 ms per forward at batch 1; from batch 4 on the one-launch form wins: profiles/r05_attn_block.md)
                fused_block = bf and isinstance(spec, GlobalAttentionSpec) and target is self.launches and T % 256 == 0 \
                    and os.environ.get("KDIFF_ATTN_BLOCK", "1") != "0" and bool(lib.kd_attn_block_bf16_supported(rps, d, nh)) \
                    and (B * nh >= 32 or os.environ.get("KDIFF_ATTN_BLOCK", "1") == "force")
                # fp8 mode: the qkv projection on the fp8 matrix instruction -- except where the one-launch bf16 attention block takes the layer
                # (level 2 of the 256 x 256 configs: 25.5 us against 25.6 + 13.6 us for fp8 projection + dense core, profiles/r06_fp8_mode.md)
                qkv_mx8 = not fused_block and mx8_ok(T, 3 * d, d, nat.EPI_QKV)
                if fused_block:
                    dq = gemm(prefix + "attn_block", x, sa.qkv_proj.weight, att, T, 3 * d, d, epi=nat.EPI_QKV,
                              scale_ptr=scale_ptr(prefix + "self_attn.norm"), scale_stride=total, rows_per_sample=rps, qk=qk)
                    target.pop()
                    target.append(_Launch(lib.kd_attn_block_bf16, (C.byref(dq),), prefix + "attn_block", enc=("kd_attn_block_bf16", (dq,))))
                elif xn is not None and prepass(d) and nh <= 16:
                    # AdaRMSNorm -> planes once, then a GEMM whose two operands both move by LDS-DMA
                    xn_planes = norm_split(prefix + "self_attn.norm", x, scale_ptr(prefix + "self_attn.norm")[1], T, d, rps)
                    dq = gemm(prefix + "qkv_proj", None, sa.qkv_proj.weight, qkv, T, 3 * d, d, epi=nat.EPI_QKV, rows_per_sample=rps, qk=qk,
                              a_planes=xn_planes)
                else:
                    dq = gemm(prefix + "qkv_proj", x, sa.qkv_proj.weight, qkv, T, 3 * d, d, epi=nat.EPI_QKV,
                              scale_ptr=scale_ptr(prefix + "self_attn.norm"), scale_stride=total, rows_per_sample=rps, qk=qk, mx8=qkv_mx8)
                    # bf16 mode, K = 256 / 512 with an attention core of its own (neighbourhood / window levels): the projection in the block
                    # form (kd_proj_block_bf16: a workgroup per (256-row group, 6 head vectors), rows normalised once) for one-round grids (192 .. 256)
                    if bf and not qkv_mx8 and target is self.launches and os.environ.get("KDIFF_PROJ_BLOCK", "1") != "0" \
                            and bool(lib.kd_proj_block_bf16_supported(rps, d, 3 * d, nat.EPI_QKV)) and 192 <= (T // 256) * (3 * d // 384) <= 256:
                        target[-1] = _Launch(lib.kd_proj_block_bf16, (C.byref(dq),), prefix + "qkv_proj(block)", enc=("kd_proj_block_bf16", (dq,)))
                dq.qkv_packed = 1 if packed_qkv else 0
                prep = (2 if packed_qkv else 0, None, None, None, C.c_float(1e-6), precision)
                shift = 0
                if isinstance(spec, ShiftedWindowAttentionSpec):
                    shift = spec.window_size // 2 if index % 2 == 1 else 0          # :523
                if fused_block:
                    pass
                elif bf and isinstance(spec, GlobalAttentionSpec):
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
                if not fuse_out:
                    gemm(prefix + "out_proj", att, sa.out_proj.weight, x, T, d, d, epi=nat.EPI_RESIDUAL, R=x)
            if ffn_x3:
                # fp32-parity mode: the whole FeedForwardBlock in one kernel (csrc/ffn_x3.hip), hidden activation on the chip; at widths 128 / 256
                # the attention block's out projection runs in front of it in the same kernel (x + att W_out^T never crosses HBM)
                fd = nat.KdFfn()
                fd.x = fd.out = x.data_ptr()
                fd.scale_stride, fd.rows_per_sample, fd.eps = total, rps, 1e-6
                fd.Wp_up = m._packed_image(mod.ff.up_proj.weight, lv.d_ff, d, 3 if fuse_out else 1).data_ptr()
                fd.Wp_down = m._packed_image(mod.ff.down_proj.weight, d, lv.d_ff, 2).data_ptr()
                if fuse_out:
                    fd.attn = att.data_ptr()
                    fd.Wp_out = m._packed_image(mod.self_attn.out_proj.weight, d, d, 0).data_ptr()
                fd.M, fd.K, fd.d_ff = T, d, lv.d_ff
                self.norm_descs.append((fd, scale_ptr(prefix + "ff.norm")[1]))
                self.keep.append(fd)
                target.append(_Launch(lib.kd_ffn_f32, (C.byref(fd),), prefix + "ff", enc=("kd_ffn_f32", (fd,))))
            elif ffn_bf:
                # the whole FeedForwardBlock (:487-493) in one kernel: the d_ff-wide hidden activation stays on the chip (width 128: with
                # the attention block's out projection in front of it)
                fd = nat.KdFfn()
                fd.x = fd.out = x.data_ptr()
                fd.scale_stride, fd.rows_per_sample, fd.eps = total, rps, 1e-6
                fd.Wp_up = m._packed_image(mod.ff.up_proj.weight, lv.d_ff, d, 3 if fuse_out else 1, bf16=True).data_ptr()
                fd.Wp_down = m._packed_image(mod.ff.down_proj.weight, d, lv.d_ff, 2, bf16=True).data_ptr()
                if fuse_out:
                    fd.attn = att.data_ptr()
                    fd.Wp_out = m._packed_image(mod.self_attn.out_proj.weight, d, d, 0, bf16=True).data_ptr()
                fd.M, fd.K, fd.d_ff = T, d, lv.d_ff
                self.norm_descs.append((fd, scale_ptr(prefix + "ff.norm")[1]))
                self.keep.append(fd)
                target.append(_Launch(lib.kd_ffn_bf16, (C.byref(fd),), prefix + "ff", enc=("kd_ffn_bf16", (fd,))))
            else:
                hid8 = None
                if xn is not None and target is self.launches and prepass(d) and lv.d_ff % 64 == 0:
                    xn_planes = norm_split(prefix + "ff.norm", x, scale_ptr(prefix + "ff.norm")[1], T, d, rps)
                    gemm(prefix + "up_proj", None, mod.ff.up_proj.weight, hid, T, lv.d_ff, d, epi=nat.EPI_GEGLU, a_planes=xn_planes)
                else:
                    up_mx8 = mx8_ok(T, lv.d_ff, d, nat.EPI_GEGLU)
                    # fp8 mode: the hidden activation leaves the GEGLU epilogue as e4m3 rows + one power-of-two scale per (row, 32 features) -- in
                    # the first half of `hid`, the scale bytes behind them -- and the down projection takes both operands as e4m3 by LDS-DMA
                    down_mx8 = up_mx8 and lv.d_ff % 128 == 0 and bool(lib.kd_gemm_mx8_supported(T, d, lv.d_ff, nat.EPI_RESIDUAL, 0))
                    hid8 = (hid.data_ptr(), hid.data_ptr() + T * lv.d_ff) if down_mx8 else None
                    du = gemm(prefix + "up_proj", x, mod.ff.up_proj.weight, hid, T, lv.d_ff, d, epi=nat.EPI_GEGLU,
                              scale_ptr=scale_ptr(prefix + "ff.norm"), scale_stride=total, rows_per_sample=rps, mx8=up_mx8, c_planes=hid8)
                    # bf16 mode, rows per sample a multiple of 256: the projection in the attention block's form (a workgroup per (256-row group,
                    # 192-output slice), rows normalised once; csrc/block_bf16.hip: proj_block_bf16_kernel) for grids that fill ONE round of the
                    # chip's 256 CUs (192 .. 256 workgroups): two rounds measured level with the A-stationary kernel (42.3 against 41.5 us at
                    # level 1), and a workgroup's six passes are a serial chain -- at 32 - 128 workgroups the A-stationary kernel, which splits
                    # the same work over up to 512 slots, is faster (batch 4: 0.645 against 0.759 ms per forward; batch 16: level); same bits
                    if bf and not up_mx8 and target is self.launches and os.environ.get("KDIFF_PROJ_BLOCK", "1") != "0" \
                            and bool(lib.kd_proj_block_bf16_supported(rps, d, lv.d_ff, nat.EPI_GEGLU)) and 192 <= (T // 256) * (lv.d_ff // 192) <= 256:
                        target[-1] = _Launch(lib.kd_proj_block_bf16, (C.byref(du),), prefix + "up_proj(block)", enc=("kd_proj_block_bf16", (du,)))
                if hid8 is not None:
                    gemm(prefix + "down_proj", None, mod.ff.down_proj.weight, x, T, d, lv.d_ff, epi=nat.EPI_RESIDUAL, R=x, a_planes=hid8, mx8=True)
                else:
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
        self._encode_launches()

    def run_cond(self, buf, stream):
        """Per-step conditioning chain into scale table ``buf`` on ``stream``; its inputs were filled by the caller on the
        same stream (``step_chain.fill``)."""
        self.step_chain.run(self.scales[buf].data_ptr(), stream)

    def run(self, x, out, sigma_data, base):
        """Main chain on the current stream.  x: input image (read by patch_in and, when preconditioning, by patch_out);
        out: result image; ``self.sigma_ptr`` was set by the caller; the [B, scale_width] table at ``base`` holds this
        step's AdaRMSNorm scales."""
        for d, off in self.norm_descs:
            d.scale = base + off
        pin, pout = self.d_patch_in, self.d_patch_out
        pin.A = x.data_ptr()
        pout.C = out.data_ptr()
        if sigma_data is None:
            pin.sigma, pout.sigma, pout.R = None, None, None
        else:
            sp = self.sigma_ptr
            pin.sigma, pout.sigma, pout.R = sp, sp, x.data_ptr()
            pin.sigma_data = pout.sigma_data = float(sigma_data)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if self.calls is not None:
            # the whole list in one host call (kd_run_list): ~6.5 us of Python / ctypes per launch otherwise, which at batch 1 - 2 is more
            # than the kernels of the bf16 mode take
            rc = self.run_list(self.calls, len(self.launches), stream, C.byref(self.failed))
            if rc:
                nat.check(rc, self.launches[self.failed.value].what)
            return
        for ln in self.launches:
            rc = ln.fn(*ln.args, stream)
            if rc:
                nat.check(rc, ln.what)

    def _encode_launches(self):
        """The main chain as a kd_run_list array, if every launch of it can be named there (and KDIFF_RUN_LIST is not 0)."""
        self.calls, self.failed, self.run_list = None, C.c_int(0), nat.lib().kd_run_list
        if os.environ.get("KDIFF_RUN_LIST", "1") == "0" or not self.launches or any(ln.enc is None for ln in self.launches):
            return
        calls = (nat.KdCall * len(self.launches))()
        for call, ln in zip(calls, self.launches):
            if not nat.encode_call(call, *ln.enc):
                return
        self.calls = calls


def _weak_epoch_bump(model):
    """load_state_dict post-hook that bumps ``model``'s weights epoch without holding the model alive."""
    ref = weakref.ref(model)

    def hook(*_args):
        m = ref()
        if m is not None:
            m._fp_epoch += 1
    return hook


# ---------------------------------------------------------------------------------- the model

class ImageTransformerDenoiserModelV2(nn.Module):
    def __init__(self, levels, mapping, in_channels, out_channels, patch_size, num_classes=0, mapping_cond_dim=0):
        super().__init__()
        for lv in levels:
            sa = lv.self_attn
            if not isinstance(sa, (GlobalAttentionSpec, NeighborhoodAttentionSpec, ShiftedWindowAttentionSpec, NoAttentionSpec)):
                raise ValueError(f"unsupported self attention spec {sa}")
            if not isinstance(sa, NoAttentionSpec) and sa.d_head != D_HEAD:
                raise ValueError(f"d_head must be {D_HEAD} (got {sa.d_head}): the HIP kernels of both arithmetic modes -- qkv epilogue (cosine-sim "
                                 f"norm + RoPE over {D_HEAD // 2} dims), attention cores, qkv layout [tokens, 3, n_heads, {D_HEAD}] of "
                                 f"include/kdiff_hip.h -- are built for {D_HEAD}-dim heads, as every shipped config uses (config.py:137-138); "
                                 f"the reference's other head sizes (image_transformer_v2.py:355-363) are not supported")
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
        self._plans, self._fingerprint, self._packed, self._plans_epoch = {}, None, {}, None
        self._fp_dicts, self._fp_names, self._fp_objs, self._fp_tensors, self._fp_tracked, self._fp_epoch = (), (), (), (), (), 0
        self._fp_hooked = weakref.WeakSet()
        self._fp_sized, self._fp_sizes = (), ()

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

    def _drop_plans(self):
        """All cached plans go, workspaces at once (``_Plan.release``).  The device is idle first: a plan's side-stream work may still be
        reading buffers the allocator would hand out again."""
        if self._plans:
            dev = next(iter(self._plans))[6]
            if getattr(dev, "type", "cpu") == "cuda":
                torch.cuda.synchronize(dev)
            for plan in self._plans.values():
                plan.release()
            self._plans = {}
            gc.collect()
        self._plans = {}

    def invalidate(self):
        """Drop the plans and packed weight images at the next call.  Needed only after an IN-PLACE edit of weights that were created
        under torch.inference_mode() outside load_state_dict / .to(): such tensors carry no version counter, so the edit leaves no trace
        (_weights_fingerprint sees everything else by itself)."""
        self._fp_epoch += 1

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .half(): the same Parameter objects with new .data (seen through their addresses) or, with
        # torch.__future__.set_overwrite_module_params_on_conversion(True), new Parameters in the dicts (seen through the slot check);
        # the epoch covers inference-mode tensors converted in place
        out = super()._apply(fn, *args, **kwargs)
        self._fp_epoch += 1
        return out

    def _packed_image(self, W, N, K, geglu, bf16=False):
        """Packed image of a weight (split-bf16, or plain bf16 for the bf16 mode), shared by all plans of this model.  The
        entry keeps the source tensor alive, so its address cannot be recycled under the cached image; the dict is dropped
        with the plans whenever the weights change (``_weights_fingerprint``)."""
        key = (id(W), N, K, int(geglu), bf16 if bf16 == "mx8" else bool(bf16))
        ent = self._packed.get(key)
        if ent is None:
            ent = self._packed[key] = (W, ops.pack_weight(W, N, K, geglu, cache=False, bf16=bf16))
        return ent[1]

    def _weights_fingerprint(self):
        """(address, version) of every parameter and buffer (behind a per-model epoch): a changed entry drops the plans and the packed
        weight images.  Read on every model call, so the LIST of tensors is kept -- torch's module traversal (parameters() / buffers() over
        ~90 sub-modules) took 0.3 - 0.4 ms per call, more than the launches of a batch-1 forward.  What makes the kept list safe is a
        per-call identity check of every SLOT the tree has -- each (module._parameters | _buffers | _modules dict, name) still holds the
        object it held when the list was built, and each of those dicts still has the size it had (a parameter / buffer / sub-module ADDED
        to an existing module) -- which is ~300 dict reads in C (a few microseconds), needs no traversal and sees the ways a tensor can be
        swapped: attribute assignment, register_*, del + re-register, a replaced sub-module, and direct writes into
        ``module._parameters[name]`` (torch.func.functional_call / stateless._reparametrize_module swap parameters that way, past every
        registration hook).  In-place edits move the tensors' version counters; .to() moves their addresses; load_state_dict / _apply /
        invalidate() bump the epoch (which also covers inference-mode tensors, whose edits leave no version trace).  No process-wide
        hooks: only this model's own tree is looked at."""
        if not (self._fp_objs and all(map(operator.is_, map(dict.get, self._fp_dicts, self._fp_names), self._fp_objs))
                and tuple(map(len, self._fp_sized)) == self._fp_sizes):      # (sizes: a slot ADDED to a recorded dict is no recorded slot)
            dicts, names, objs, ts, sized = [], [], [], [], []
            for mod in self.modules():
                if mod not in self._fp_hooked:       # a (sub-)module's load_state_dict rewrites weights in place: bump the epoch (inference-mode tensors)
                    self._fp_hooked.add(mod)
                    mod.register_load_state_dict_post_hook(_weak_epoch_bump(self))      # (weak: a sub-module shared with another model does not pin this one)
                for d, is_tensor in ((mod._parameters, True), (mod._buffers, True), (mod._modules, False)):
                    sized.append(d)
                    for name, obj in d.items():
                        dicts.append(d), names.append(name), objs.append(obj)
                        if is_tensor and obj is not None:
                            ts.append(obj)
            seen, uniq = set(), []
            for t in ts:                     # tied tensors once, like parameters() / buffers()
                if id(t) not in seen:
                    seen.add(id(t))
                    uniq.append(t)
            self._fp_dicts, self._fp_names, self._fp_objs = tuple(dicts), tuple(names), tuple(objs)
            self._fp_sized, self._fp_sizes = tuple(sized), tuple(map(len, sized))
            self._fp_tensors, self._fp_tracked = tuple(uniq), tuple(not t.is_inference() for t in uniq)
        return (self._fp_epoch, *[(t.data_ptr(), t._version if tr else 0) for t, tr in zip(self._fp_tensors, self._fp_tracked)])

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
        plan = self._plan_for(x, aug_cond, class_cond, create=True)
        cur = torch.cuda.current_stream()
        ident = self._cond_identity(sigma, aug_cond, class_cond, mapping_cond)
        table, from_schedule = None, False
        for sch in plan.schedules:                # this call's scales were computed with its whole sigma schedule
            i = sch.row_of(sigma) if sch.ident_others == ident[1:] else None
            if i is not None:
                # the table was computed once, ahead of the loop, on the side stream: ONE wait per (schedule, stream) orders every later
                # forward on that stream behind it (a wait packet per forward cost ~3 us of idle queue each: profiles/r06_launch_gaps_*.txt)
                if cur.cuda_stream not in sch.waited:
                    cur.wait_event(sch.done)
                    sch.waited.add(cur.cuda_stream)
                table = sch.tables[i].data_ptr()
                from_schedule = True
                break
        if table is None:
            pre, plan.prefetched = plan.prefetched, None
            if pre is not None and pre[0] == ident:
                buf = pre[1]                      # this step's scale table was computed ahead of time on the side stream
                cur.wait_event(pre[2])
            else:
                buf = 1 - plan.last_buf
                if pre is not None:
                    cur.wait_event(pre[2])        # an unused prefetch still owns the conditioning workspace: order behind it
                plan.step_chain.fill(sigma, aug_cond, class_cond, mapping_cond)
                plan.run_cond(buf, C.c_void_p(cur.cuda_stream))
            plan.last_buf = buf
            table = plan.scales[buf].data_ptr()
        if sigma_data is not None:                # per-sample sigma of the preconditioning folded into patch-in / patch-out
            if sigma.numel() != B or sigma.dtype != torch.float32 or not sigma.is_contiguous() or sigma.device != x.device:
                plan.sigma.copy_(sigma.reshape(-1).expand(B) if sigma.numel() == 1 else sigma.reshape(B), non_blocking=True)
                plan.sigma_ptr = plan.sigma.data_ptr()
            else:
                plan.sigma_ptr = sigma.data_ptr()  # read in place: freed storage is not reused before this stream's work is done
        if not from_schedule:                     # (only the per-step side-stream chain of prefetch_conditioning waits on this event)
            plan.main_entry.record(cur)           # everything before this step's main chain (incl. an inline conditioning chain)
        out = torch.empty(B, self.out_channels, H, W, device=x.device, dtype=torch.float32)
        plan.run(x, out, sigma_data, table)
        return out

    def _plan_for(self, x, aug_cond, class_cond, create):
        """The launch plan of this (batch, size, conditioning kinds, device, arithmetic mode) combination."""
        B, _, H, W = x.shape
        fp = self._weights_fingerprint()
        if fp != self._fingerprint:
            if not create:
                return None
            self._drop_plans()
            self._fingerprint, self._packed = fp, {}
        has_class = self.class_emb is not None
        # Kernel selection is fixed when a plan is built: by the arithmetic mode, by the environment switches read in _Plan and by
        # library options (kd_ffn_f32_supported follows "ffn_x3").  The switches are part of the key; a change of any library option
        # (nat.option_epoch, bumped by set_option and by a changed KDIFF_OPTIONS / KDIFF_* variable) drops the cached plans, so an
        # A/B run on ONE model object really compares two plans.
        nat.lib()                                  # (syncs the environment-mapped options -> option_epoch)
        if self._plans_epoch != nat.option_epoch:
            if not create:
                return None
            self._drop_plans()
            self._plans_epoch = nat.option_epoch
        key = (B, H, W, aug_cond is not None, has_class, self.mapping_cond_in_proj is not None, x.device, nat.default_precision()) \
            + tuple(os.environ.get(k, d) for k, d in PLAN_SWITCHES)
        plan = self._plans.get(key)
        if plan is not None and len(self._plans) > 1:
            self._plans[key] = self._plans.pop(key)                # most recently used last (dicts keep insertion order)
        if plan is None and create:
            if self.patch_in.proj.weight.device != x.device:
                raise RuntimeError(f"model weights are on {self.patch_in.proj.weight.device}, input on {x.device}")
            if len(self._plans) >= MAX_PLANS:
                # a plan owns the workspaces of its shape (~20 MB per 256 x 256 image in the fp32 modes): a caller that walks through
                # batch sizes would otherwise keep them all.  The least recently used one goes; its side-stream work (conditioning
                # prefetch) may still be in flight in buffers the allocator would hand out again at once, hence the device-wide wait
                # (rare: only when a NEW shape arrives with MAX_PLANS shapes cached).
                torch.cuda.synchronize(x.device)
                self._plans.pop(next(iter(self._plans))).release()
                gc.collect()
            with torch.inference_mode(False):     # (workspaces made under inference_mode could not be written in place outside it later)
                plan = self._plans[key] = _Plan(self, B, H, W, key[3], has_class, key[5], x.device)
        if plan is not None and has_class and class_cond is not None:
            # nn.Embedding raises on an out-of-range id (on every call); the HIP kernel would read past the table.  Checked whenever
            # the ids are a tensor this plan has not seen in this state (address, version): one device->host read per new id
            # tensor -- a sampling run passes the same tensor to every step, so the per-step path stays sync-free.
            # (tensors made under torch.inference_mode() carry no version counter: they cannot be recognised as unchanged and are
            # re-checked on every call.  A few identities are remembered, so alternating id tensors -- cond / uncond calls of a
            # guidance wrapper -- stay sync-free as well.)
            tracked = not class_cond.is_inference()
            ident = (class_cond.data_ptr(), _ver(class_cond), tuple(class_cond.shape))
            if not tracked or ident not in plan.class_checked:
                lo, hi = (int(class_cond.min()), int(class_cond.max())) if class_cond.numel() else (0, 0)
                if lo < 0 or hi >= self.class_emb.weight.shape[0]:
                    raise IndexError(f"class_cond ids must lie in [0, {self.class_emb.weight.shape[0] - 1}] (got {lo}..{hi})")
                if tracked:
                    plan.class_checked[ident] = class_cond        # (kept alive: the address cannot be recycled under the record)
                    while len(plan.class_checked) > CLASS_IDS_KEPT:
                        del plan.class_checked[next(iter(plan.class_checked))]
        return plan

    # ---- conditioning ahead of time ---------------------------------------------------------------
    def _cond_identity(self, sigma, aug_cond, class_cond, mapping_cond):
        return tuple(None if t is None else (t.data_ptr(), _ver(t), tuple(t.shape)) for t in (sigma, aug_cond, class_cond, mapping_cond))

    def _hint_usable(self, x_like, class_cond, mapping_cond):
        if not x_like.is_cuda or x_like.dim() != 4:
            return False
        return not ((class_cond is None and self.class_emb is not None) or (mapping_cond is None and self.mapping_cond_in_proj is not None))

    @torch.no_grad()
    def prefetch_schedule(self, x_like, sigma_table, aug_cond=None, class_cond=None, mapping_cond=None):
        """Hint from the solver loop: model calls of this run will pass ROWS of ``sigma_table`` ([n, B] fp32 on the device,
        one row per call) as their sigma, together with exactly these other conditioning tensors.  The conditioning chain
        depends on sigma / class / aug / mapping_cond only, never on x, so it runs here once for all n x B rows (the same
        row-by-row kernels as the per-step chain: bit-identical scales) and every such call finds its [B, scale_width]
        table ready.  Calls that pass anything else fall back to the per-step chain.  Returns True when taken."""
        if not self._hint_usable(x_like, class_cond, mapping_cond) or sigma_table.dim() != 2 or not sigma_table.is_cuda:
            return False
        n, B = sigma_table.shape
        if B != x_like.shape[0] or n == 0 or sigma_table.dtype != torch.float32 or not sigma_table.is_contiguous():
            return False
        plan = self._plan_for(x_like, aug_cond, class_cond, create=True)
        if n * B * plan.scale_width * 4 > SCHEDULE_TABLE_MAX_BYTES:
            return False
        chain = plan.schedule_chains.pop(n, None)
        if chain is None:
            with torch.inference_mode(False):
                chain = plan.build_cond(n * B)
        plan.schedule_chains[n] = chain                   # most recently used last; older lengths (and their workspaces) are dropped
        for old_n in list(plan.schedule_chains)[:-SCHEDULE_CHAINS_KEPT]:
            del plan.schedule_chains[old_n]
        cur = torch.cuda.current_stream()
        for sch in plan.schedules:                # an earlier schedule's chain of the same length shares the work buffers
            cur.wait_event(sch.done)
        tables = torch.empty(n, B, plan.scale_width, device=sigma_table.device, dtype=torch.float32)
        chain.fill(sigma_table, aug_cond, class_cond, mapping_cond, repeat=n)
        chain.run(tables.data_ptr(), C.c_void_p(cur.cuda_stream))
        done = torch.cuda.Event()
        done.record(cur)
        others = (aug_cond, class_cond, mapping_cond)
        plan.schedules.append(_Schedule(sigma_table, others, self._cond_identity(None, *others)[1:], tables, done))
        del plan.schedules[:-SCHEDULES_KEPT]
        while len(plan.schedules) > 1 and sum(sc.tables.numel() * 4 for sc in plan.schedules) > SCHEDULE_TABLE_MAX_BYTES:
            del plan.schedules[0]                          # all tables of a plan together stay below the bound of a single one
        return True

    def release_schedules(self):
        """Drop the scale tables (and the hinted tensors they keep alive) of finished runs: up to SCHEDULE_TABLE_MAX_BYTES of HBM
        per plan otherwise stay allocated until the next run replaces them.  Safe at any time: calls no table covers use the
        per-step chain, with the same bits."""
        for plan in self._plans.values():
            plan.schedules.clear()
            plan.schedule_chains.clear()

    @torch.no_grad()
    def prefetch_conditioning(self, x_like, sigma, aug_cond=None, class_cond=None, mapping_cond=None):
        """Hint from the solver loop: the NEXT model call will use exactly these conditioning tensors.  Unless a schedule
        hint already covers that call, the conditioning chain then runs on a side HIP stream, concurrently with the main
        chain of the step in flight, into the other scale table; the next ``forward`` with the same tensors just waits for
        its event.  A hint that is not followed costs nothing but the side-stream work."""
        if not self._hint_usable(x_like, class_cond, mapping_cond):
            return
        plan = self._plan_for(x_like, aug_cond, class_cond, create=False)
        if plan is None or plan.prefetched is not None:
            return
        others = self._cond_identity(None, aug_cond, class_cond, mapping_cond)[1:]
        if any(sch.ident_others == others and sch.row_of(sigma) is not None for sch in plan.schedules):
            return
        buf = 1 - plan.last_buf
        side = plan.side_stream
        side.wait_event(plan.main_entry)          # table `buf` and the conditioning workspace are free once the main chain of
        with torch.cuda.stream(side):             # the step in flight has started (its predecessors are complete in stream order)
            plan.step_chain.fill(sigma, aug_cond, class_cond, mapping_cond)
            plan.run_cond(buf, C.c_void_p(side.cuda_stream))
            done = torch.cuda.Event()
            done.record(side)
        # the record keeps the hinted tensors alive: while it is pending their storage cannot be freed and handed to another
        # tensor, so a later call whose (address, version, shape) key matches really is the same data (a sampler aborted
        # mid-loop leaves a record behind; the next run's schedule then necessarily lives at other addresses)
        plan.prefetched = (self._cond_identity(sigma, aug_cond, class_cond, mapping_cond), buf, done, (sigma, aug_cond, class_cond, mapping_cond))
