"""ctypes binding of csrc/libkdiff_hip.so (C ABI declared in include/kdiff_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent, the first use raises.
The library is built in-tree by ``__graft_entry__.build()`` / ``make -C k-diffusion_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KDIFF_HIP_LIB") or os.path.join(_HERE, "csrc", "libkdiff_hip.so")

# enums (include/kdiff_hip.h)
A_PLAIN, A_MERGE2x2, A_PATCH_NCHW = 0, 1, 2
EPI_STORE, EPI_RESIDUAL, EPI_GEGLU, EPI_SPLIT_LERP, EPI_UNPATCH_NCHW, EPI_QKV = 0, 1, 2, 3, 4, 5
PREC_EXACT, PREC_SPLIT3, PREC_BF16 = 0, 1, 2
# PREC_FP8 is a mode of the NETWORK, not a KdGemm.precision value: the bf16 mode with the AdaRMSNorm -> wide projections of the K = 256 / 512
# levels on the block-scaled fp8 matrix instruction (kd_gemm_mx8); every descriptor of such a plan says KD_PREC_BF16
PREC_FP8 = 3
_MODES = {"exact": PREC_EXACT, "split3": PREC_SPLIT3, "bf16": PREC_BF16, "fp8": PREC_FP8}


def default_precision():
    """Arithmetic mode of the network (KDIFF_GEMM):
    exact   fp32-input MFMA, bit-for-bit an fmaf chain (fp32 activations);
    split3  (default) every fp32 operand split into two bf16, 3 bf16 MFMAs per product, fp32 accumulate (fp32 activations):
            the fp32-parity mode, inside the 1e-3 tolerance of the reference's fp32 path;
    bf16    bf16 activations in HBM, one bf16 MFMA per product, fp32 accumulate / statistics / softmax: the arithmetic of the
            reference under torch.autocast(bfloat16);
    fp8     the bf16 mode with the norm -> qkv / norm -> GEGLU projections of the K = 256 / 512 levels as e4m3 x e4m3 products with
            power-of-two block scales (weights per output channel, activations per 32-k block) on the fp8 matrix instruction."""
    mode = os.environ.get("KDIFF_GEMM", "split3").lower()
    if mode not in _MODES:
        raise ValueError(f"KDIFF_GEMM={mode!r}: expected one of {sorted(_MODES)}")
    return _MODES[mode]


def kernel_precision():
    """KdGemm.precision of the default mode: the fp8 mode's descriptors are bf16 descriptors (see PREC_FP8)."""
    p = default_precision()
    return PREC_BF16 if p == PREC_FP8 else p


def precision_name(p):
    return {v: k for k, v in _MODES.items()}[p]
(STEP_EULER, STEP_HEUN_PRED, STEP_HEUN_CORR, STEP_DPMPP_2M1, STEP_DPMPP_2M2, STEP_ADD_NOISE, STEP_LERP2, STEP_AXPY,
 STEP_EULER_FROM, STEP_AXPBY, STEP_ADD_DIFF, STEP_TO_D) = range(12)


class KdGemm(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("a_mode", C.c_int), ("epi", C.c_int), ("norm", C.c_int),
        ("rows_per_sample", C.c_int), ("scale_stride", C.c_int),
        ("gh", C.c_int), ("gw", C.c_int), ("ph", C.c_int), ("pw", C.c_int), ("chan", C.c_int),
        ("eps", C.c_float), ("out_add", C.c_float), ("sigma_data", C.c_float),
        ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("R", C.c_void_p),
        ("scale", C.c_void_p), ("sigma", C.c_void_p), ("fac", C.c_void_p),
        ("precision", C.c_int), ("Wp", C.c_void_p),
        ("n_heads", C.c_int), ("qk_scale", C.c_void_p), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
        ("qkv_packed", C.c_int), ("rope_pos", C.c_void_p), ("rope_freq", C.c_void_p),
        ("a_split", C.c_int), ("c_split", C.c_int), ("A_lo", C.c_void_p), ("C_lo", C.c_void_p),
        ("per_row", C.c_int),
    ]


class KdFfn(C.Structure):
    """include/kdiff_hip.h KdFfn: the fused feed-forward block (FeedForwardBlock.forward, image_transformer_v2.py:487-493)."""
    _fields_ = [
        ("x", C.c_void_p), ("out", C.c_void_p), ("scale", C.c_void_p),
        ("scale_stride", C.c_int), ("rows_per_sample", C.c_int), ("eps", C.c_float),
        ("Wp_up", C.c_void_p), ("Wp_down", C.c_void_p),
        ("M", C.c_int), ("K", C.c_int), ("d_ff", C.c_int),
        ("attn", C.c_void_p), ("Wp_out", C.c_void_p),
    ]


class KdCall(C.Structure):
    """include/kdiff_hip.h KdCall: one launch of a kd_run_list list (entry point + its arguments)."""
    _fields_ = [("op", C.c_int), ("f", C.c_float), ("p", C.c_void_p * 5), ("i", C.c_int * 8)]


_vp, _i, _f, _ll, _d = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_double

# name -> argtypes; every symbol declared in include/kdiff_hip.h
SIGNATURES = {
    "kd_version": [],
    "kd_last_error": [],
    "kd_set_option": [C.c_char_p, _i],
    "kd_get_option": [C.c_char_p, _i],
    "kd_gemm_f32": [C.POINTER(KdGemm), _vp],
    "kd_gemm_bf16": [C.POINTER(KdGemm), _vp],
    "kd_packed_weight_bytes_bf16": [_i, _i, _i],
    "kd_pack_weight_bf16": [_vp, _vp, _i, _i, _i, _vp],
    "kd_ffn_bf16_supported": [_i, _i, _i],
    "kd_ffn_bf16": [_vp, _vp],
    "kd_ffn_f32_supported": [_i, _i, _i],
    "kd_ffn_f32": [_vp, _vp],
    "kd_attn_global_bf16": [_vp, _vp, _i, _i, _i, _vp],
    "kd_attn_window_bf16": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "kd_attn_na2d_bf16": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "kd_attn_block_bf16_supported": [_i, _i, _i],
    "kd_attn_block_bf16": [C.POINTER(KdGemm), _vp],
    "kd_proj_block_bf16_supported": [_i, _i, _i, _i],
    "kd_proj_block_bf16": [C.POINTER(KdGemm), _vp],
    "kd_packed_weight_bytes_mx8": [_i, _i, _i],
    "kd_pack_weight_mx8": [_vp, _vp, _i, _i, _i, _vp],
    "kd_gemm_mx8_supported": [_i, _i, _i, _i, _i],
    "kd_gemm_mx8": [C.POINTER(KdGemm), _vp],
    "kd_packed_weight_bytes": [_i, _i, _i],
    "kd_pack_weight_bf16x3": [_vp, _vp, _i, _i, _i, _vp],
    "kd_rmsnorm_f32": [_vp, _vp, _vp, _i, _i, _f, _vp],
    "kd_fourier_sigma_f32": [_vp, _vp, _vp, _i, _i, _vp],
    "kd_fourier_f32": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "kd_cond_sum_f32": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "kd_qk_prep_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "kd_attn_global_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _vp],
    "kd_attn_window_f32": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _vp],
    "kd_attn_na2d_f32": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _vp],
    "kd_sampler_step_f32": [_i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _ll, _vp],
    "kd_precond_in_f32": [_vp, _vp, _vp, _f, _i, _ll, _vp],
    "kd_precond_out_f32": [_vp, _vp, _vp, _vp, _f, _i, _ll, _vp],
    "kd_rows_affine_f32": [_vp, _vp, _vp, _vp, _vp, _i, _ll, _vp],
    "kd_sigma_to_t_f32": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "kd_t_to_sigma_f32": [_vp, _vp, _vp, _i, _i, _vp],
    "kd_dpm_eps_f32": [_vp, _vp, _vp, _f, _ll, _vp],
    "kd_dpm_combine_f32": [_vp, _vp, _vp, _vp, _f, _f, _ll, _vp],
    "kd_dpm_error_partials": [],
    "kd_dpm_error_f32": [_vp, _vp, _vp, _f, _f, _ll, _vp, _vp],
    "kd_brownian_f32": [_vp, _vp, _i, _ll, _d, _d, _d, _d, _f, _i, _vp],
    "kd_brownian_cached_f32": [_vp, _vp, _vp, _i, _i, _vp, _i, _ll, _d, _d, _d, _d, _f, _i, _vp],
    "kd_randn_f32": [_vp, _vp, _i, _ll, C.c_ulonglong, _f, _vp],
    "kd_to_uint8": [_vp, _vp, _ll, _vp],
    "kd_norm_split_f32": [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _f, _vp],
    "kd_prof_enable": [_i],
    "kd_prof_count": [],
    "kd_prof_get": [_i, C.c_char_p, _i, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "kd_prof_reset": [],
    "kd_prof_clock_buffer": [_vp],
    "kd_run_list": [C.POINTER(KdCall), _i, _vp, C.POINTER(C.c_int)],
}

# entry points a kd_run_list entry can name (include/kdiff_hip.h: KD_OP_*)
RUN_LIST_OPS = {"kd_gemm_f32": 0, "kd_gemm_bf16": 1, "kd_ffn_f32": 2, "kd_ffn_bf16": 3, "kd_attn_global_f32": 4, "kd_attn_window_f32": 5,
                "kd_attn_na2d_f32": 6, "kd_attn_global_bf16": 7, "kd_attn_window_bf16": 8, "kd_attn_na2d_bf16": 9, "kd_norm_split_f32": 10,
                "kd_attn_block_bf16": 11, "kd_proj_block_bf16": 12, "kd_gemm_mx8": 13}


def encode_call(call, name, args):
    """Fill the KdCall ``call`` from an entry point's name and its positional arguments WITHOUT the trailing stream: pointers (None, int
    addresses, c_void_p, or a ctypes Structure = its address) go to ``p`` in order, ints to ``i`` in order, the float to ``f`` -- the rule
    kd_run_list unpacks by.  An argument with a ``bind_call(call, index)`` method (a pointer the caller patches per run) is told where it
    lives.  Returns False if the entry point cannot be named in a list."""
    op = RUN_LIST_OPS.get(name)
    if op is None:
        return False
    kinds = SIGNATURES[name][:-1]
    if len(kinds) != len(args):
        raise ValueError(f"{name}: {len(args)} arguments for {len(kinds)} parameters")
    call.op = op
    n_p = n_i = 0
    for kind, a in zip(kinds, args):
        if kind is _i:
            call.i[n_i] = int(a)
            n_i += 1
        elif kind is _f:
            call.f = a.value if isinstance(a, C.c_float) else float(a)
        else:
            if hasattr(a, "bind_call"):
                a.bind_call(call, n_p)
                v = a.scale
            elif isinstance(a, C.Structure):
                v = C.addressof(a)
            elif isinstance(a, C.c_void_p):
                v = a.value
            else:
                v = a
            call.p[n_p] = v
            n_p += 1
    return True

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """The loaded shared library (loads on first use; raises loudly if it is not there)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C k-diffusion_amd/csrc). "
                "There is no CPU / PyTorch fallback for the sampling hot path.")
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.argtypes = argtypes
            fn.restype = C.c_char_p if name == "kd_last_error" else (C.c_longlong if name.startswith("kd_packed_weight_bytes") else C.c_int)
        _lib = handle
    _sync_options(_lib)
    return _lib


# The library reads no environment variables; KDIFF_OPTIONS="name=value,..." is mapped onto kd_set_option here (re-read whenever it changes,
# so that tests can flip options inside one process).  (Rounds 2 - 5 also mapped eight per-option variables -- KDIFF_SKINNY, KDIFF_ASTAT_WAVES,
# ... -- onto the same calls; every one of them is reachable through KDIFF_OPTIONS by its library name and they were removed in round 6.)
_ENV_OPTIONS = {}
_env_applied = None       # (value of every mapped variable, {name: value} parsed from KDIFF_OPTIONS) as last applied
_programmatic = set()     # option names set through set_option(): the environment sync leaves them alone
option_epoch = 0          # bumped whenever a library option may have changed: captured launch graphs are bound to one epoch


def _parse_options(text):
    out = {}
    for item in (text or "").split(","):
        if item.strip():
            name, _, val = item.partition("=")
            out[name.strip()] = int(val)
    return out


def _sync_options(handle):
    """KDIFF_* variables -> kd_set_option; KDIFF_OPTIONS="name=value,..." sets any library option by name (A-B runs of bench.py).
    Only what CHANGED since the last call is re-applied: a variable that changed, a KDIFF_OPTIONS entry that changed or appeared,
    and -- reset to the library default -- an entry that disappeared.  Options set through ``set_option`` are not touched unless
    the environment names them anew."""
    global _env_applied, option_epoch
    raw = tuple(os.environ.get(k) for k in _ENV_OPTIONS) + (os.environ.get("KDIFF_OPTIONS"),)
    if _env_applied is not None and raw == _env_applied[0]:
        return
    named = _parse_options(raw[-1])
    old_raw, old_named = _env_applied[1:] if _env_applied is not None else ((None,) * len(_ENV_OPTIONS), {})
    for (env, (name, dflt)), val, was in zip(_ENV_OPTIONS.items(), raw, old_raw):
        if val != was and (val not in (None, "") or was not in (None, "")) and (name not in _programmatic or val not in (None, "")):
            handle.kd_set_option(name.encode(), dflt if val in (None, "") else int(val))
            _programmatic.discard(name)
    for name, val in named.items():
        if old_named.get(name) != val:
            if handle.kd_set_option(name.encode(), val) != 0:
                raise ValueError(f"KDIFF_OPTIONS: unknown library option {name!r}")
            _programmatic.discard(name)
    for name in old_named:
        if name not in named and name not in _programmatic:
            handle.kd_set_option(name.encode(), -0x7FFFFFFF - 1)           # INT_MIN: back to the built-in default (kd_set_option)
    _env_applied = (raw, raw[:-1], named)
    option_epoch += 1


prof_active = False


def prof_enable(on):
    """Per-launch HIP-event timing of every kernel (kd_prof_*).  While it is on, the model issues its launch lists directly
    (a captured graph has no per-kernel events to read back)."""
    global prof_active
    check(lib().kd_prof_enable(1 if on else 0), "kd_prof_enable")
    prof_active = bool(on)


def set_option(name, value):
    """Tuning / A-B switch of the library (include/kdiff_hip.h: kd_set_option)."""
    global option_epoch
    check(lib().kd_set_option(name.encode(), int(value)), "kd_set_option")
    _programmatic.add(name)
    option_epoch += 1


def check(code, what="libkdiff_hip"):
    if code != 0:
        msg = lib().kd_last_error()
        raise RuntimeError(f"{what} failed ({code}): {msg.decode() if msg else '?'}")
