"""Parity of every HIP kernel (through the C ABI) against the CPU oracle and the reference's golden
op outputs.  Needs a real MI355X:  pytest -m gpu.

Tolerances (max-norm relative unless stated): GEMM-class ops 2e-5 in the exact-fp32 MFMA mode (FMA-chain
order only) and 1e-4 in the default split-bf16x3 mode (per-product error <= ~2^-15), attention 2e-5, elementwise solver steps BIT-EXACT, index maps implied bit-exact by the value checks
on asymmetric random data."""
import numpy as np
import pytest
import torch

from oracle import brownian as obrown
from oracle import hdit, solvers
from tests.helpers import bits, relerr

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops(KD):
    return KD.ops


@pytest.fixture(params=["exact", "split3"])
def gtol(request, monkeypatch):
    """Runs a GEMM-class test in both arithmetic modes; returns the mode's tolerance."""
    monkeypatch.setenv("KDIFF_GEMM", request.param)
    return 2e-5 if request.param == "exact" else 1e-4


@pytest.fixture(autouse=True)
def throughput_kernels(request, KD):
    """The tests of this file were written against the throughput kernels, also at their ragged / few-row shapes: the few-rows latency
    kernel (round 4, csrc/gemm_x3s.hip: projections of at most 1024 rows) is switched off for them and has its own tests
    (marked `few_rows`, which run with the library's defaults)."""
    if request.node.get_closest_marker("few_rows"):
        yield
        return
    KD._native.set_option("x3s_max_rows", 0)
    KD._native.set_option("b16s_max_rows", 0)                 # (its bf16 sibling, csrc/gemm_b16s.hip)
    try:
        yield
    finally:
        KD._native.set_option("x3s_max_rows", -2 ** 31)       # back to the built-in defaults
        KD._native.set_option("b16s_max_rows", -2 ** 31)


def _prof_names(nat):
    """Names of the launches recorded since kd_prof_reset (kd_prof_enable(1)): which kernel served each call."""
    import ctypes as C
    lib, out = nat.lib(), []
    name, ms, fl, by = C.create_string_buffer(128), C.c_float(), C.c_double(), C.c_double()
    for i in range(lib.kd_prof_count()):
        nat.check(lib.kd_prof_get(i, name, 128, C.byref(ms), C.byref(fl), C.byref(by)), "kd_prof_get")
        out.append(name.value.decode())
    return out


def rn(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def g(t):
    return t.to(DEV).contiguous()


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (131, 200, 96), (32, 768, 256), (1000, 48, 128), (5, 7 * 4, 12), (384, 512, 1536),
                                   (257, 129, 48), (64, 3072, 512), (300, 130, 256), (2048, 512, 512), (1000, 256, 1536)])
def test_gemm_plain_and_residual(ops, gtol, M, N, K):
    x, w, r = rn(M, K, seed=1), rn(N, K, seed=2) / K ** 0.5, rn(M, N, seed=3)
    assert relerr(ops.linear(g(x), g(w)), x @ w.T) < gtol
    assert relerr(ops.linear(g(x), g(w), residual=g(r)), x @ w.T + r) < gtol
    assert relerr(ops.linear(g(x), g(w), out_add=1.0), x @ w.T + 1) < gtol


@pytest.mark.parametrize("M,N,K", [(1, 256, 256), (7, 12, 12), (64, 260, 772), (65, 8, 256), (128, 52, 128), (32, 768, 256)])
def test_skinny_gemm_rows_per_sample(ops, gtol, M, N, K):
    """<= 128 rows (one per sample: mapping network, embeddings, AdaRMSNorm scale projection) are served by the
    fp32 skinny kernel in both precision modes: every epilogue / prologue it takes, against torch."""
    x, w, r = rn(M, K, seed=1), rn(N, K, seed=2) / K ** 0.5, rn(M, N, seed=3)
    w2, gain = rn(2 * N, K, seed=4) / K ** 0.5, 1 + 0.2 * rn(K, seed=5)
    assert relerr(ops.linear(g(x), g(w)), x @ w.T) < 2e-5
    assert relerr(ops.linear(g(x), g(w), residual=g(r)), x @ w.T + r) < 2e-5
    assert relerr(ops.linear(g(x), g(w), out_add=1.0), x @ w.T + 1) < 2e-5
    val, gate = (x @ w2.T).chunk(2, dim=-1)
    assert relerr(ops.linear_geglu(g(x), g(w2)), val * torch.nn.functional.gelu(gate)) < 2e-5
    xn = x * torch.rsqrt(x.square().mean(-1, keepdim=True) + 1e-6) * gain
    assert relerr(ops.norm_linear(g(x), g(gain), g(w), rows_per_sample=M), xn @ w.T) < 2e-5
    val, gate = (xn @ w2.T).chunk(2, dim=-1)
    assert relerr(ops.norm_linear(g(x), g(gain), g(w2), rows_per_sample=M, epi=__import__("k_diffusion_amd")._native.EPI_GEGLU), val * torch.nn.functional.gelu(gate)) < 2e-5
    # per-sample scale vectors are not the skinny kernel's case: the tile kernels keep serving them
    scales = 1 + 0.2 * rn(M, K, seed=6)
    xs = x * torch.rsqrt(x.square().mean(-1, keepdim=True) + 1e-6) * scales
    assert relerr(ops.norm_linear(g(x), g(scales), g(w), rows_per_sample=1), xs @ w.T) < gtol


def test_split3_error_is_bounded_and_asymmetric_safe(ops, monkeypatch):
    """The split-bf16x3 product against an fp64 reference on wide-dynamic-range data: error stays ~2^-15-class
    relative to sum|a*b| (a transposed / permuted operand would show up as O(1))."""
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    M, N, K = 300, 260, 512
    x = rn(M, K, seed=7) * torch.logspace(-3, 3, K)[None, :]
    w = rn(N, K, seed=8) * torch.logspace(2, -2, N)[:, None]
    ref = x.double() @ w.double().T
    bound = x.abs().double() @ w.abs().double().T
    got = ops.linear(g(x), g(w)).cpu().double()
    assert ((got - ref).abs() / bound).max().item() < 2.0 ** -14
    monkeypatch.setenv("KDIFF_GEMM", "exact")
    got = ops.linear(g(x), g(w)).cpu().double()
    assert ((got - ref).abs() / bound).max().item() < 2.0 ** -20


def test_gemm_rejects_bad_shapes(ops):
    x, w = g(rn(8, 6)), g(rn(4, 6))
    with pytest.raises(RuntimeError, match="K % 4"):
        ops.linear(x, w)
    with pytest.raises(RuntimeError):
        ops.linear(rn(8, 8), g(rn(4, 8)))          # CPU tensor: no fallback


def test_norm_linear_and_geglu(ops, gtol, golden):
    o = golden["ops"]
    x, cond, wl, wg = o["rms_norm.x"], o["adarms.cond"], o["adarms.w"], o["geglu.w"]
    # stand-alone rms_norm vs the reference's output
    assert relerr(ops.rms_norm(g(x), g(o["rms_norm.scale"])), o["rms_norm.y"]) < 2e-6
    # AdaRMSNorm scale = Linear(cond) + 1 via the "+const" epilogue, then fused norm -> GEGLU
    scale = ops.linear(g(cond), g(wl), out_add=1.0)
    y = ops.norm_linear(g(x), scale, g(wg), rows_per_sample=64, epi=2)
    ref = hdit.linear_geglu(o["adarms.y"], wg)
    assert relerr(y, ref) < gtol
    # plain GEGLU against the reference's own linear_geglu output
    assert relerr(ops.linear_geglu(g(x), g(wg)), o["geglu.y"]) < gtol
    # fused norm -> plain linear with shared gain
    w = rn(96, 128, seed=5) / 128 ** 0.5
    y = ops.norm_linear(g(x), g(o["rms_norm.scale"]), g(w), rows_per_sample=64)
    assert relerr(y, o["rms_norm.y"] @ w.T) < gtol


def test_qkv_epilogue_prepares_q_and_k(ops, gtol):
    """AdaRMSNorm -> qkv projection -> [cosine-sim scale + axial RoPE on q, k] in ONE GEMM (EPI_QKV), against the
    oracle's separate steps (image_transformer_v2.py:370-392 up to the attention call)."""
    B, H, W, nh = 2, 16, 8, 2
    d = nh * 64
    x, scale = rn(B, H, W, d, seed=1), 1 + 0.1 * rn(B, d, seed=2)
    w = rn(3 * d, d, seed=3) / d ** 0.5
    sh = torch.tensor([9.0, 12.5])
    cos, sin = _tables(H, W, nh)
    y = ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=H * W, epi=5, qk=(g(sh), g(cos), g(sin), nh))
    qkv = hdit.rms_norm(x, scale[:, None, None, :]) @ w.T
    q, k, v = hdit.split_qkv(qkv, nh)
    q, k = hdit.cosine_sim_scale(q, k, sh)
    theta = hdit.rope_theta(hdit.axial_pos(H, W), hdit.rope_freqs(nh))
    ref = torch.stack([hdit.apply_rope(q, theta), hdit.apply_rope(k, theta), v], dim=3).reshape(B, H, W, 3 * d)
    assert relerr(y, ref) < gtol
    with pytest.raises(RuntimeError, match="qkv epilogue"):
        ops.norm_linear(g(x), g(scale), g(w[: 2 * d]), rows_per_sample=H * W, epi=5, qk=(g(sh), g(cos), g(sin), nh))


@pytest.mark.parametrize("mode", ["exact", "split3"])
def test_per_row_products_do_not_depend_on_the_row_count(KD, ops, monkeypatch, mode):
    """KdGemm.per_row: the conditioning products (one row per sample) go through the per-row fp32 FMA kernel whatever M, so
    the conditioning of a whole sigma schedule (steps x batch rows in one launch) is bit-identical with computing it
    32 rows at a time -- and agrees with fp64 to fp32 rounding."""
    monkeypatch.setenv("KDIFF_GEMM", mode)
    nat = KD._native
    g = torch.Generator().manual_seed(5)
    M, K = 1000, 256
    A = torch.randn(M, K, generator=g).to(DEV)
    scale = (1 + 0.1 * torch.randn(K, generator=g)).to(DEV)
    for N, epi, norm in [(256, nat.EPI_STORE, False), (768, nat.EPI_GEGLU, True), (256, nat.EPI_RESIDUAL, False), (1792, nat.EPI_STORE, False)]:
        W = (torch.randn((2 if epi == nat.EPI_GEGLU else 1) * N, K, generator=g) / K ** 0.5).to(DEV)
        R = torch.randn(M, N, generator=g).to(DEV) if epi == nat.EPI_RESIDUAL else None
        kw = dict(N=N, K=K, epi=epi, norm_scale=scale if norm else None, rows_per_sample=M if norm else 0, out_add=1.0 if epi == nat.EPI_STORE else 0.0)
        whole = ops.gemm(A, W, torch.empty(M, N, device=DEV), M=M, residual=R, per_row=True, **kw)
        parts = []
        for m0 in range(0, M, 32):
            m1 = min(M, m0 + 32)
            kw["rows_per_sample"] = (m1 - m0) if norm else 0
            parts.append(ops.gemm(A[m0:m1].contiguous(), W, torch.empty(m1 - m0, N, device=DEV), M=m1 - m0,
                                  residual=None if R is None else R[m0:m1].contiguous(), **kw))
        assert torch.equal(whole, torch.cat(parts)), (N, epi, norm)
        a = A.double().cpu()
        if norm:
            a = a * scale.double().cpu() * torch.rsqrt(a.pow(2).mean(1, keepdim=True) + 1e-6)
        y = a @ W.double().cpu().T
        if epi == nat.EPI_GEGLU:
            y = y[:, :N] * torch.nn.functional.gelu(y[:, N:])
        y = y + (1.0 if epi == nat.EPI_STORE else 0.0) + (0 if R is None else R.double().cpu())
        assert relerr(whole, y) < 1e-5


@pytest.mark.parametrize("B,T,d_ff", [(8, 4096, 384), (5, 4096, 448), (1, 16500, 64), (2, 300, 128)])
def test_bf16_fused_ffn(ops, B, T, d_ff):
    """kd_ffn_bf16 (the whole FeedForwardBlock, image_transformer_v2.py:487-493, hidden activation on-chip) against the fp32
    oracle on the bf16-rounded input and against the two-kernel form (GEGLU GEMM + residual GEMM) it replaces: same roundings
    (bf16 normalised input, bf16 hidden, bf16 output), so the two agree to a couple of bf16 ulps."""
    from k_diffusion_amd import _native as nat
    K = 128
    x, scale = rn(B, T, K, seed=14), 1 + 0.2 * rn(B, K, seed=15)
    wu, wd = rn(2 * d_ff, K, seed=16, scale=K ** -0.5), rn(K, d_ff, seed=17, scale=d_ff ** -0.5)
    xb = _bf(x)
    ref = _rt(x) + hdit.linear_geglu(hdit.rms_norm(_rt(x), scale[:, None, :]), _rt(wu)) @ _rt(wd).T
    y = ops.ffn(xb, g(scale), g(wu), g(wd), rows_per_sample=T)
    assert y.dtype == BF and y.shape == xb.shape and relerr(y, ref) < 8e-3
    hid = ops.norm_linear(xb, g(scale), g(wu), rows_per_sample=T, epi=nat.EPI_GEGLU)
    y2 = ops.linear(hid, g(wd), residual=xb)
    assert relerr(y, y2.float()) < 6e-3
    z = xb.clone()
    ops.ffn(z, g(scale), g(wu), g(wd), rows_per_sample=T, out=z)             # in place (how the model uses it)
    assert torch.equal(z, y)
    assert ops.ffn_supported(B * T, K, d_ff) == (B * T >= 16384)
    assert not ops.ffn_supported(1 << 20, 256, 768) and not ops.ffn_supported(1 << 20, 128, 100)
    with pytest.raises(RuntimeError):
        ops.ffn(_bf(rn(B, 64, 512, seed=1)), g(1 + 0.2 * rn(B, 512, seed=2)), g(rn(2 * 64, 512, seed=3)), g(rn(512, 64, seed=4)), rows_per_sample=64)


@pytest.mark.parametrize("B,T,d_ff", [(3, 1024, 768), (2, 77, 192), (1, 130, 64)])
def test_bf16_fused_ffn_width_256(ops, B, T, d_ff):
    """The width-256 form of kd_ffn_bf16 (half-unit weight ring, one wave per SIMD; off by default because it is not faster than the
    two-kernel form) against the fp32 oracle and the two-kernel form."""
    from k_diffusion_amd import _native as nat
    K = 256
    x, scale = rn(B, T, K, seed=24), 1 + 0.2 * rn(B, K, seed=25)
    wu, wd = rn(2 * d_ff, K, seed=26, scale=K ** -0.5), rn(K, d_ff, seed=27, scale=d_ff ** -0.5)
    xb = _bf(x)
    ref = _rt(x) + hdit.linear_geglu(hdit.rms_norm(_rt(x), scale[:, None, :]), _rt(wu)) @ _rt(wd).T
    y = ops.ffn(xb, g(scale), g(wu), g(wd), rows_per_sample=T)
    assert relerr(y, ref) < 8e-3
    hid = ops.norm_linear(xb, g(scale), g(wu), rows_per_sample=T, epi=nat.EPI_GEGLU)
    assert relerr(y, ops.linear(hid, g(wd), residual=xb).float()) < 6e-3
    nat.set_option("ffn_fused_256", 1)
    try:
        assert ops.ffn_supported(1 << 20, 256, 768)
    finally:
        nat.set_option("ffn_fused_256", 0)


@pytest.mark.parametrize("H,W,nh,B,K", [(16, 16, 2, 2, 128), (32, 32, 2, 1, 128), (20, 24, 4, 1, 256), (8, 8, 8, 3, 512)])
def test_split_stored_qkv_feeds_the_attention_cores(ops, monkeypatch, H, W, nh, B, K):
    """qkv_packed: the qkv GEMM stores q, k, v as split-bf16 chunks and the split cores (prep='packed') take them as stored.
    Same split of the same values, so every core must return BIT-identical results to the fp32-qkv path."""
    from k_diffusion_amd import _native as nat
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    T, N = H * W, 3 * nh * 64
    x, w = rn(B * T, K, seed=1), rn(N, K, seed=3) / K ** 0.5
    scale = 1 + 0.1 * rn(B, K, seed=2)
    sh = torch.linspace(6.0, 12.0, nh)
    cos, sin = _tables(H, W, nh)
    qk = (g(sh), g(cos), g(sin), nh)
    plain = ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=T, epi=5, qk=qk).view(B, H, W, N)
    packed = ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=T, epi=5, qk=qk, qkv_packed=True).view(B, H, W, N)
    assert not torch.equal(plain, packed)
    # the packed words are (hi, hi, lo, lo) bf16 pairs of the fp32 values
    pw = packed.view(torch.int32).view(-1, 4)
    hi = torch.stack([(pw[:, 0] << 16), (pw[:, 0] & -65536), (pw[:, 1] << 16), (pw[:, 1] & -65536)], dim=1).view(torch.float32)
    lo = torch.stack([(pw[:, 2] << 16), (pw[:, 2] & -65536), (pw[:, 3] << 16), (pw[:, 3] & -65536)], dim=1).view(torch.float32)
    assert relerr((hi + lo).view_as(plain), plain) < 2.0 ** -15
    assert torch.equal(hi.view_as(plain), plain.to(torch.bfloat16).to(torch.float32))
    if H >= 7 and W >= 7:
        # (the split-stored operands go to the round-3 core, attn_x3.hip: same products, another summation order and exp2 softmax)
        assert relerr(ops.attn_na2d(packed, nh, 7, prep="packed"), ops.attn_na2d(plain, nh, 7)) < 2e-5
        nat.set_option("attn_x3", 0)
        try:
            assert torch.equal(ops.attn_na2d(packed, nh, 7, prep="packed"), ops.attn_na2d(plain, nh, 7))
        finally:
            nat.set_option("attn_x3", 1)
    # T <= 256 and streaming (T = 64 / 128 / 256 split-stored: the round-3 core, other summation order; the round-1 core is bit-identical)
    assert relerr(ops.attn_global(packed.view(B, T, N), nh, prep="packed"), ops.attn_global(plain.view(B, T, N), nh)) < 2e-5
    nat.set_option("attn_x3", 0)
    try:
        assert torch.equal(ops.attn_global(packed.view(B, T, N), nh, prep="packed"), ops.attn_global(plain.view(B, T, N), nh))
    finally:
        nat.set_option("attn_x3", 1)
    if H % 8 == 0 and W % 8 == 0:
        for shift in (0, 4):
            assert torch.equal(ops.attn_window(packed, nh, 8, shift, prep="packed"), ops.attn_window(plain, nh, 8, shift))
    if H % 4 == 0 and W % 4 == 0:
        assert torch.equal(ops.attn_window(packed, nh, 4, 2, prep="packed"), ops.attn_window(plain, nh, 4, 2))
    monkeypatch.setenv("KDIFF_GEMM", "exact")
    with pytest.raises(RuntimeError, match="split"):
        ops.attn_global(packed.view(B, T, N), nh, prep="packed")
    with pytest.raises(RuntimeError, match="qkv_packed"):
        ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=T, epi=5, qk=qk, qkv_packed=True)


@pytest.mark.parametrize("M,K,N,kind", [(1024, 128, 384, "qkv"), (1000, 128, 768, "geglu"), (4096, 256, 768, "qkv"), (640, 256, 1536, "geglu"),
                                        (8192, 512, 1536, "qkv"), (2048, 512, 3072, "geglu"), (768, 512, 512, "store"), (520, 128, 256, "store"),
                                        # >= 65536 rows at K = 128: the 8-wave / 256-row-panel form (ragged last panel included)
                                        (65536, 128, 384, "qkv"), (65736, 128, 768, "geglu"), (65736, 128, 256, "store"), (131072, 128, 384, "geglu")])
def test_wide_projections_a_stationary(ops, monkeypatch, M, K, N, kind):
    """The A-stationary kernel (gemm_astat.hip: K in {128, 256, 512}, >= 2 n-tiles, M >= 512) against the oracle at
    shapes that exercise its n-split grids (few panels), a ragged last panel, per-sample and shared norm scales, and all
    three epilogues (store, GEGLU, qkv with q/k preparation)."""
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    if M >= 65536:
        monkeypatch.setenv("KDIFF_OPTIONS", "astat_waves=8")       # the optional 8-wave / 256-row-panel form
    B = 4 if M % 4 == 0 and (M // 4) % 128 == 0 else 1
    T = M // B
    x = rn(M, K, seed=1)
    scale = 1 + 0.1 * rn(B, K, seed=2) if B > 1 else 1 + 0.1 * rn(K, seed=2)
    xn = hdit.rms_norm(x.view(B, T, K), scale.view(B, 1, K) if B > 1 else scale).view(M, K)
    if kind == "geglu":
        w = rn(2 * N, K, seed=3) / K ** 0.5
        y = ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=T, epi=2)
        ref = hdit.linear_geglu(xn, w)
    elif kind == "store":
        w = rn(N, K, seed=3) / K ** 0.5
        y = ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=T)
        ref = xn @ w.T
    else:
        nh = N // 192
        w = rn(N, K, seed=3) / K ** 0.5
        sh = torch.linspace(6.0, 12.0, nh)
        h = 1
        while h * h < T:
            h += 1
        assert h * h == T or T % 128 == 0
        hh, ww = (h, h) if h * h == T else (T // 16, 16)
        theta = hdit.rope_theta(hdit.axial_pos(hh, ww), hdit.rope_freqs(nh))
        cos, sin = torch.cos(theta).reshape(T, nh, 16), torch.sin(theta).reshape(T, nh, 16)
        y = ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=T, epi=5, qk=(g(sh), g(cos), g(sin), nh))
        q, k, v = hdit.split_qkv((xn @ w.T).view(B, hh, ww, N), nh)
        q, k = hdit.cosine_sim_scale(q, k, sh)
        ref = torch.stack([hdit.apply_rope(q, theta), hdit.apply_rope(k, theta), v], dim=3).reshape(M, N)
    assert relerr(y, ref) < 1e-4


@pytest.mark.parametrize("M,K,N", [(1024, 256, 256), (2048, 512, 512), (700, 128, 256)])
def test_residual_in_place_multi_tile(ops, monkeypatch, M, K, N):
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    x, w, r = rn(M, K, seed=1), rn(N, K, seed=2) / K ** 0.5, rn(M, N, seed=3)
    assert relerr(ops.linear(g(x), g(w), residual=g(r)), x @ w.T + r) < 1e-4
    xin = g(r.clone())                                   # in place, as the model uses it (C == R)
    assert relerr(ops.linear(g(x), g(w), residual=xin, out=xin), x @ w.T + r) < 1e-4


@pytest.mark.parametrize("B,h,w,K,C", [(2, 16, 16, 256, 128), (1, 32, 16, 512, 256), (3, 16, 12, 128, 64), (3, 20, 12, 256, 128), (5, 16, 16, 512, 256)])
def test_token_split_multi_tile(ops, monkeypatch, B, h, w, K, C):
    """TokenSplit (Linear -> depth-to-space -> lerp with the skip) at multi-tile shapes (several m- and n-tiles; a ragged last row panel).
    Round 4: with C % 128 == 0 and K >= 256 the split runs on gemm_x3r.hip (loader waves; the scatter is the row base of its store runs, the skip
    operand is read ahead of the K loop): checked against the reference formula AND against the round-1 kernel it replaces (x3r_split = 0)."""
    from k_diffusion_amd import _native as nat
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    x, wt, skip = rn(B, h, w, K, seed=1), rn(4 * C, K, seed=2) / K ** 0.5, rn(B, 2 * h, 2 * w, C, seed=3)
    for fac in (0.3, 0.75):
        ref = torch.lerp(skip, hdit.token_split(x, wt, 2, 2), torch.tensor([fac]))
        y = ops.token_split_lerp(g(x), g(wt), g(skip), g(torch.tensor([fac])))
        assert relerr(y, ref) < 1e-4
        nat.set_option("x3r_split", 0)
        try:
            old = ops.token_split_lerp(g(x), g(wt), g(skip), g(torch.tensor([fac])))
        finally:
            nat.set_option("x3r_split", 1)
        assert relerr(y, old) < 1e-4
        xin = g(skip.clone())                             # in place, as the model uses it (the result overwrites the skip)
        assert torch.equal(ops.token_split_lerp(g(x), g(wt), xin, g(torch.tensor([fac])), out=xin), y)


def test_token_merge_split(ops, gtol, golden):
    o = golden["ops"]
    x = o["rms_norm.x"]
    assert relerr(ops.token_merge(g(x), g(o["merge.w"])), o["merge.y"]) < gtol
    for fac in (0.37, 0.5, 0.8):
        skip = o["split.skip"]
        ref = torch.lerp(skip, hdit.token_split(x, o["split.w"], 2, 2), torch.tensor([fac]))
        y = ops.token_split_lerp(g(x), g(o["split.w"]), g(skip), g(torch.tensor([fac])))
        assert relerr(y, ref) < gtol
    y = ops.token_split_lerp(g(x), g(o["split.w"]), g(o["split.skip"]), g(torch.tensor([0.37])))
    assert relerr(y, o["split.y"]) < gtol


@pytest.mark.parametrize("C,H,W", [(3, 128, 128), (3, 72, 88), (1, 128, 64), (4, 64, 64)])
def test_split3_patch_out_round3(KD, ops, monkeypatch, C, H, W):
    """out_norm + out patch projection + NHWC -> NCHW + c_out * y + c_skip * x_in (image_transformer_v2.py:758-760, layers.py:90) on the
    round-3 A-stationary kernel (W rows read in (py, channel, px) order: 16-byte image accesses) against the oracle and the round-1
    kernel; 1, 3 and 4 channels, a ragged last row panel, with and without the Karras scalings."""
    from k_diffusion_amd import _native as nat
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    B, p, d = 3, 4, 128
    img = rn(B, C, H, W, seed=1)
    sigma = torch.tensor([0.05, 1.3, 70.0])
    c_skip, c_out, _ = solvers.karras_scalings(sigma, 0.5)
    x, scale, w_out = rn(B, H // p, W // p, d, seed=3), 1 + 0.1 * rn(d, seed=4), rn(C * p * p, d, seed=5) / d ** 0.5
    inner = hdit.token_split(hdit.rms_norm(x, scale), w_out, p, p).movedim(-1, 1)
    ref_d = inner * c_out.view(-1, 1, 1, 1) + img * c_skip.view(-1, 1, 1, 1)
    y0 = ops.patch_out(g(x), g(scale), g(w_out), (p, p), C)
    y1 = ops.patch_out(g(x), g(scale), g(w_out), (p, p), C, x_in=g(img), sigma=g(sigma), sigma_data=0.5)
    assert relerr(y0, inner) < 1e-4 and relerr(y1, ref_d) < 1e-4
    nat.set_option("x3_unpatch", 0)
    try:
        o0 = ops.patch_out(g(x), g(scale), g(w_out), (p, p), C)
        o1 = ops.patch_out(g(x), g(scale), g(w_out), (p, p), C, x_in=g(img), sigma=g(sigma), sigma_data=0.5)
    finally:
        nat.set_option("x3_unpatch", 1)
    assert relerr(y0, o0) < 1e-4 and relerr(y1, o1) < 1e-4


@pytest.mark.parametrize("C,H,W,p,d", [(3, 16, 16, 2, 128), (1, 28, 28, 4, 64), (3, 32, 64, 4, 128)])
def test_patch_in_out(ops, gtol, C, H, W, p, d):
    B = 3
    img, w_in = rn(B, C, H, W, seed=1), rn(d, C * p * p, seed=2)
    sigma = torch.tensor([0.05, 1.3, 70.0])
    ref = hdit.token_merge(img.movedim(1, -1).contiguous(), w_in, p, p)
    assert relerr(ops.patch_in(g(img), g(w_in), (p, p)), ref) < gtol
    c_skip, c_out, c_in = solvers.karras_scalings(sigma, 0.5)
    ref_c = hdit.token_merge((img * c_in.view(-1, 1, 1, 1)).movedim(1, -1).contiguous(), w_in, p, p)
    assert relerr(ops.patch_in(g(img), g(w_in), (p, p), sigma=g(sigma), sigma_data=0.5), ref_c) < gtol
    x, scale, w_out = rn(B, H // p, W // p, d, seed=3), 1 + 0.1 * rn(d, seed=4), rn(C * p * p, d, seed=5) / d ** 0.5
    inner = hdit.token_split(hdit.rms_norm(x, scale), w_out, p, p).movedim(-1, 1)
    assert relerr(ops.patch_out(g(x), g(scale), g(w_out), (p, p), C), inner) < gtol
    ref_d = inner * c_out.view(-1, 1, 1, 1) + img * c_skip.view(-1, 1, 1, 1)
    y = ops.patch_out(g(x), g(scale), g(w_out), (p, p), C, x_in=g(img), sigma=g(sigma), sigma_data=0.5)
    assert relerr(y, ref_d) < gtol


def _tables(h, w, nh):
    theta = hdit.rope_theta(hdit.axial_pos(h, w), hdit.rope_freqs(nh)).reshape(h * w, nh, 16)
    return torch.cos(theta), torch.sin(theta)


def _pack(q, k, v):
    # heads-last q,k,v [n,h,w,nh,e] -> qkv [n,h,w,3*nh*e]
    return torch.stack([q, k, v], dim=3).reshape(*q.shape[:3], -1).contiguous()


def test_qk_prep_inplace(ops, golden):
    o = golden["ops"]
    qkv = g(_pack(o["qk.q"], o["qk.k"], o["qk.v"]))
    cos, sin = _tables(16, 16, 2)
    ops.qk_prep_(qkv, g(o["qk.scale"]), g(cos), g(sin), 2)
    out = qkv.cpu().view(2, 16, 16, 3, 2, 64)
    assert relerr(out[..., 0, :, :], o["qk.q_out"]) < 2e-6
    assert relerr(out[..., 1, :, :], o["qk.k_out"]) < 2e-6
    assert torch.equal(out[..., 2, :, :], o["qk.v"])          # v untouched (:379, :389-390)


@pytest.mark.parametrize("fused", [False, True])
def test_attention_vs_reference_golden(ops, golden, gtol, fused):
    o = golden["ops"]
    cos, sin = _tables(16, 16, 2)
    prep = (g(o["qk.scale"]), g(cos), g(sin))
    if fused:
        qkv = g(_pack(o["qk.q"], o["qk.k"], o["qk.v"]))
        kw = dict(prep=prep)
    else:
        qkv = g(_pack(o["qk.q_out"], o["qk.k_out"], o["qk.v"]))
        kw = {}
    y = ops.attn_global(qkv.view(2, 256, -1), 2, **kw).view(2, 16, 16, 2, 64)
    assert relerr(y, o["attn_global.o"]) < gtol        # exact: fp32 MFMA core; split3: bf16x3 core
    for shift in (0, 4):
        y = ops.attn_window(qkv, 2, 8, shift, **kw).view(2, 16, 16, 2, 64)
        assert relerr(y, o[f"attn_window{shift}.o"]) < gtol, shift
    ref = hdit.na2d(o["qk.q_out"], o["qk.k_out"], o["qk.v"], 7, 1.0)
    y = ops.attn_na2d(qkv, 2, 7, **kw).view(2, 16, 16, 2, 64)
    assert relerr(y, ref) < 1e-4            # split-bf16x3 MFMA products (K, Q, P, V each hi + lo)


def test_window_attention_rect(ops, gtol, golden):
    o = golden["ops"]
    qkv = g(_pack(o["attn_window_rect.q"], o["attn_window_rect.k"], o["attn_window_rect.v"]))
    y = ops.attn_window(qkv, 1, 8, 4).view(1, 8, 24, 1, 64)
    assert relerr(y, o["attn_window_rect.o"]) < gtol


@pytest.mark.parametrize("tag,ws,shift", [("w4s0", 4, 0), ("w4s2", 4, 2), ("w16s8", 16, 8), ("w16s0", 16, 0)])
def test_window_attention_other_sizes(ops, gtol, golden, tag, ws, shift):
    """4x4 and 16x16 windows (one and eight waves per window) against the reference's apply_window_attention."""
    o = golden["ops"]
    q = o[f"attn_{tag}.q"]
    y = ops.attn_window(g(_pack(q, o[f"attn_{tag}.k"], o[f"attn_{tag}.v"])), q.shape[3], ws, shift).view(*q.shape)
    assert relerr(y, o[f"attn_{tag}.o"]) < gtol
    with pytest.raises(RuntimeError, match="window_size"):
        ops.attn_window(g(rn(1, 12, 12, 192)), 1, 6, 0)


@pytest.mark.parametrize("T,nh,B", [(49, 4, 3), (64, 8, 2), (100, 1, 2), (256, 2, 2), (7, 1, 1)])
def test_attn_global_sizes(ops, gtol, T, nh, B):
    q, k, v = (rn(B, 1, T, nh, 64, seed=s, scale=sc) for s, sc in ((1, 0.6), (2, 0.6), (3, 1.0)))
    ref = hdit.attn_global(q, k, v, 1.0)
    y = ops.attn_global(g(_pack(q, k, v)).view(B, T, -1), nh).view(B, 1, T, nh, 64)
    assert relerr(y, ref) < gtol


@pytest.mark.parametrize("H,W,nh,B", [(20, 15, 2, 2), (32, 32, 2, 1), (17, 16, 1, 3), (24, 40, 1, 1)])
def test_attn_global_long_sequences(ops, monkeypatch, H, W, nh, B):
    """T > 256 tokens: the streaming core (128-key blocks, online softmax), with and without fused q/k preparation."""
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    T = H * W
    q, k, v = (rn(B, 1, T, nh, 64, seed=s, scale=sc) for s, sc in ((1, 0.6), (2, 0.6), (3, 1.0)))
    y = ops.attn_global(g(_pack(q, k, v)).view(B, T, -1), nh).view(B, 1, T, nh, 64)
    assert relerr(y, hdit.attn_global(q, k, v, 1.0)) < 1e-4
    scale = torch.linspace(5.0, 12.0, nh)
    cos, sin = _tables(H, W, nh)
    theta = hdit.rope_theta(hdit.axial_pos(H, W), hdit.rope_freqs(nh))
    qs, ks = hdit.cosine_sim_scale(q.view(B, H, W, nh, 64), k.view(B, H, W, nh, 64), scale)
    ref = hdit.attn_global(hdit.apply_rope(qs, theta).reshape(B, 1, T, nh, 64), hdit.apply_rope(ks, theta).reshape(B, 1, T, nh, 64), v, 1.0)
    y = ops.attn_global(g(_pack(q, k, v)).view(B, T, -1), nh, prep=(g(scale), g(cos), g(sin))).view(B, 1, T, nh, 64)
    assert relerr(y, ref) < 1e-4
    # a peaked row (one dominant key far from the first block) exercises the running-max rescale
    k2 = k.clone()
    k2[:, :, T - 3] = q[:, :, 5] * 4.0
    y = ops.attn_global(g(_pack(q, k2, v)).view(B, T, -1), nh).view(B, 1, T, nh, 64)
    assert relerr(y, hdit.attn_global(q, k2, v, 1.0)) < 5e-4          # logits of order 100: 3e-5 relative on them
    monkeypatch.setenv("KDIFF_GEMM", "exact")
    with pytest.raises(RuntimeError, match="streaming"):
        ops.attn_global(g(_pack(q, k, v)).view(B, T, -1), nh)


def _split_stored(x):
    """fp32 [..., 64] -> the qkv_packed storage: every 4 dims as 16 bytes [hi: 4 x bf16][lo: 4 x bf16], viewed as fp32 [..., 64]."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    sh = x.shape[:-1]
    return torch.cat([hi.view(*sh, 16, 4), lo.view(*sh, 16, 4)], dim=-1).contiguous().view(torch.float32).view(*sh, 64)


@pytest.mark.parametrize("H,W,nh,B", [(7, 7, 1, 2), (8, 8, 2, 1), (9, 12, 1, 2), (16, 16, 2, 2), (20, 13, 1, 1), (32, 32, 4, 1), (64, 64, 2, 1), (14, 22, 1, 1), (15, 40, 2, 1)])
def test_attn_na2d_split_stored_round3(ops, monkeypatch, H, W, nh, B):
    """The round-3 neighbourhood core (csrc/attn_x3.hip: halo rows by LDS-DMA, transposing V reads, K and V through one image) on operands
    stored split, against the restated na2d and against the round-1 core on the same operands; images smaller than the 14 x 22 halo,
    ragged tiles, border clamping."""
    from k_diffusion_amd import _native as nat
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    q, k, v = (rn(B, H, W, nh, 64, seed=s, scale=sc) for s, sc in ((1, 0.5), (2, 0.5), (3, 1.0)))
    ref = hdit.na2d(q, k, v, 7, 1.0)
    packed = g(_pack(_split_stored(q), _split_stored(k), _split_stored(v)))
    y = ops.attn_na2d(packed, nh, 7, prep="packed").view(B, H, W, nh, 64)
    assert relerr(y, ref) < 1e-4
    nat.set_option("attn_x3", 0)
    try:
        old = ops.attn_na2d(packed, nh, 7, prep="packed").view(B, H, W, nh, 64)
    finally:
        nat.set_option("attn_x3", 1)
    assert relerr(y, old) < 2e-5


@pytest.mark.parametrize("ks,B,H,W,nh", [(3, 2, 9, 12, 2), (3, 1, 3, 5, 1), (5, 2, 20, 13, 1), (5, 1, 32, 32, 4), (9, 2, 20, 33, 2), (9, 1, 9, 9, 1), (9, 1, 64, 64, 2),
                                          (11, 2, 24, 40, 2), (11, 1, 11, 13, 1), (13, 1, 32, 32, 2), (13, 2, 13, 21, 1), (13, 1, 45, 19, 1)])
def test_attn_na2d_split_stored_kernel_sizes(ops, monkeypatch, ks, B, H, W, nh):
    """Neighbourhood attention of the fp32-parity mode for every kernel size the reference's interface takes here (image_transformer_v2.py:399-410;
    odd sizes 3 .. 13): 3 / 5 / 9 on the kernel of the shipped size 7 with other constants, 11 / 13 on the densely packed patch form
    (csrc/attn_x3.hip), operands stored split, against the restated na2d; grids smaller than the halo, ragged tiles, border clamping."""
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    q, k, v = (rn(B, H, W, nh, 64, seed=s, scale=sc) for s, sc in ((1, 0.5), (2, 0.5), (3, 1.0)))
    ref = hdit.na2d(q, k, v, ks, 1.0)
    packed = g(_pack(_split_stored(q), _split_stored(k), _split_stored(v)))
    y = ops.attn_na2d(packed, nh, ks, prep="packed").view(B, H, W, nh, 64)
    assert relerr(y, ref) < 1e-4, (ks, H, W)
    if ks == 5 and H == 20:
        # fp32 (not split-stored) operands exist for the shipped size only: refused by name, not misread
        with pytest.raises(RuntimeError, match="split-stored"):
            ops.attn_na2d(g(_pack(q, k, v)), nh, ks)
        with pytest.raises(RuntimeError, match="kernel_size"):
            ops.attn_na2d(packed, nh, 15, prep="packed")
        with pytest.raises(RuntimeError, match="kernel_size"):
            ops.attn_na2d(packed, nh, 6, prep="packed")


@pytest.mark.parametrize("T,nh,B", [(256, 8, 3), (128, 2, 2), (64, 1, 5), (256, 1, 1)])
def test_attn_global_split_stored_round3(ops, monkeypatch, T, nh, B):
    """The round-3 global core (csrc/attn_x3.hip) on operands stored split, T = 64 / 128 / 256, against the oracle and the round-1 core."""
    from k_diffusion_amd import _native as nat
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    q, k, v = (rn(B, 1, T, nh, 64, seed=s, scale=sc) for s, sc in ((4, 0.5), (5, 0.5), (6, 1.0)))
    ref = hdit.attn_global(q, k, v, 1.0).view(B, T, nh, 64)
    packed = g(_pack(_split_stored(q), _split_stored(k), _split_stored(v))).view(B, T, -1)
    y = ops.attn_global(packed, nh, prep="packed").view(B, T, nh, 64)
    assert relerr(y, ref) < 1e-4
    nat.set_option("attn_x3", 0)
    try:
        old = ops.attn_global(packed, nh, prep="packed").view(B, T, nh, 64)
    finally:
        nat.set_option("attn_x3", 1)
    assert relerr(y, old) < 2e-5


@pytest.mark.parametrize("H,W,nh,B", [(7, 7, 1, 2), (8, 8, 2, 1), (9, 12, 1, 2), (16, 16, 2, 2), (20, 13, 1, 1), (32, 32, 4, 1)])
def test_attn_na2d_sizes(ops, H, W, nh, B):
    q, k, v = (rn(B, H, W, nh, 64, seed=s, scale=sc) for s, sc in ((1, 0.5), (2, 0.5), (3, 1.0)))
    ref = hdit.na2d(q, k, v, 7, 1.0)
    y = ops.attn_na2d(g(_pack(q, k, v)), nh, 7).view(B, H, W, nh, 64)
    assert relerr(y, ref) < 1e-4
    # fused preparation path against oracle-prepared q, k
    scale = torch.linspace(5.0, 12.0, nh)
    cos, sin = _tables(H, W, nh)
    theta = hdit.rope_theta(hdit.axial_pos(H, W), hdit.rope_freqs(nh))
    qs, ks = hdit.cosine_sim_scale(q, k, scale)
    ref = hdit.na2d(hdit.apply_rope(qs, theta), hdit.apply_rope(ks, theta), v, 7, 1.0)
    y = ops.attn_na2d(g(_pack(q, k, v)), nh, 7, prep=(g(scale), g(cos), g(sin))).view(B, H, W, nh, 64)
    assert relerr(y, ref) < 1e-4
    if H == 7:
        with pytest.raises(RuntimeError, match="smaller"):
            ops.attn_na2d(g(_pack(q, k, v))[:, :6].contiguous(), nh, 7)


def test_sampler_steps_bit_exact(ops):
    n = 2 * 3 * 17 * 5 + 3                                     # exercises the scalar tail
    x, d, o2, a = (rn(n, seed=s, scale=sc) for s, sc in ((1, 30.0), (2, 1.0), (3, 1.0), (4, 2.0)))
    c = [torch.tensor(v, dtype=torch.float32) for v in (0.731, -1.37, 1.618, 0.618)]
    f = [float(v) for v in c]
    X, D, O2, A = g(x), g(d), g(o2), g(a)
    N = __import__("k_diffusion_amd")._native
    cases = {
        N.STEP_EULER: x + ((x - d) / c[0]) * c[1],
        N.STEP_DPMPP_2M1: c[0] * x - c[1] * d,
        N.STEP_DPMPP_2M2: c[0] * x - c[1] * (c[2] * d - c[3] * o2),
        N.STEP_ADD_NOISE: x + d * f[0] * c[1] * c[2],
        N.STEP_EULER_FROM: x + ((o2 - d) / c[0]) * c[1],
        N.STEP_AXPBY: c[0] * x + c[1] * d,
        N.STEP_ADD_DIFF: x + c[0] * (d - o2),
        N.STEP_TO_D: (x - d) / c[0],
        N.STEP_LERP2: c[0] * d + c[1] * o2,
        N.STEP_AXPY: x + d * c[0],
    }
    for op, ref in cases.items():
        y = ops.sampler_step(op, X, D, in2=O2, c0=f[0], c1=f[1], c2=f[2], c3=f[3])
        assert torch.equal(bits(y), bits(ref)), op
    aux = torch.empty_like(X)
    y = ops.sampler_step(N.STEP_HEUN_PRED, X, D, aux=aux, c0=f[0], c1=f[1])
    dref = (x - d) / c[0]
    assert torch.equal(bits(aux), bits(dref)) and torch.equal(bits(y), bits(x + dref * c[1]))
    y = ops.sampler_step(N.STEP_HEUN_CORR, X, D, in2=O2, aux=A, c0=f[0], c1=f[1])
    ref = x + ((a + (o2 - d) / c[0]) / 2) * c[1]
    assert torch.equal(bits(y), bits(ref))
    # in-place use (out aliases x), as the samplers do
    Xc = X.clone()
    ops.sampler_step(N.STEP_DPMPP_2M1, Xc, D, out=Xc, c0=f[0], c1=f[1])
    assert torch.equal(bits(Xc), bits(cases[N.STEP_DPMPP_2M1]))


def test_dpm_solver_kernels(ops):
    n = 2 * 3 * 17 * 5 + 3
    x, d, e2 = (rn(n, seed=s, scale=sc) for s, sc in ((1, 30.0), (2, 1.0), (3, 5.0)))
    k = [torch.tensor(v, dtype=torch.float32) for v in (0.731, 1.37, 1.618)]
    f = [float(v) for v in k]
    eps = (x - d) / k[0]
    assert torch.equal(bits(ops.dpm_eps(g(x), g(d), f[0])), bits(eps))
    assert torch.equal(bits(ops.dpm_combine(g(x), g(eps), f[1])), bits(x - k[1] * eps))
    assert torch.equal(bits(ops.dpm_combine(g(x), g(eps), f[1], g(e2), f[2])), bits(x - k[1] * eps - k[2] * (e2 - eps)))
    with pytest.raises(RuntimeError, match="sigma"):
        ops.dpm_eps(g(x), g(d), 0.0)
    # adaptive-solver error norm (sampling.py:464-465), reproducible to the bit run to run
    lo, hi, prev = rn(3, 5, 7, 11, seed=5, scale=4.0), rn(3, 5, 7, 11, seed=6, scale=4.0), rn(3, 5, 7, 11, seed=7, scale=9.0)
    delta = torch.maximum(torch.tensor(0.0078), torch.tensor(0.05) * torch.maximum(lo.abs(), prev.abs()))
    ref = (torch.linalg.norm(((lo - hi) / delta).double()) / lo.numel() ** 0.5).item()
    got = ops.dpm_error(g(lo), g(hi), g(prev), 0.0078, 0.05)
    assert abs(got - ref) < 1e-6 * ref
    assert got == ops.dpm_error(g(lo), g(hi), g(prev), 0.0078, 0.05)


def test_preconditioner_generic(ops):
    x, fx = rn(3, 3, 8, 8, seed=1, scale=20.0), rn(3, 3, 8, 8, seed=2)
    sigma = torch.tensor([0.01, 2.0, 160.0])
    c_skip, c_out, c_in = (c.view(-1, 1, 1, 1) for c in solvers.karras_scalings(sigma, 0.5))
    assert torch.equal(bits(ops.precond_in(g(x), g(sigma), 0.5)), bits(x * c_in))
    assert torch.equal(bits(ops.precond_out(g(fx), g(x), g(sigma), 0.5)), bits(fx * c_out + x * c_skip))


def test_conditioning_front_end(ops):
    sigma = torch.tensor([0.01, 0.3, 7.0, 160.0])
    w = rn(128, 1, seed=1)
    ref = hdit.fourier_features((torch.log(sigma) / 4)[:, None], w)
    assert (ops.fourier_sigma(g(sigma), g(w)).cpu() - ref).abs().max() < 2e-5
    a9, w9 = rn(4, 9, seed=2, scale=0.3), rn(128, 9, seed=3)
    assert (ops.fourier_features(g(a9), g(w9)).cpu() - hdit.fourier_features(a9, w9)).abs().max() < 5e-5
    a, b, emb, c = rn(4, 256, seed=4), rn(256, seed=5), rn(11, 256, seed=6), rn(4, 256, seed=7)
    ids = torch.tensor([10, 0, 3, 3])
    assert torch.allclose(ops.cond_sum(g(a), g(b), g(emb), g(ids), g(c)).cpu(), a + b + emb[ids] + c, atol=1e-6)
    assert torch.allclose(ops.cond_sum(g(a), g(c)).cpu(), a + c, atol=1e-6)


def test_brownian_vs_oracle(ops):
    seeds = [12345, 2 ** 63 - 7, 0]
    per = 3 * 5 * 7
    out = torch.empty(3, 3, 5, 7, device=DEV)
    ops.brownian(out, g(torch.tensor(seeds, dtype=torch.int64)), 0.01, 80.0, 0.5, 3.25, 1.0 / (3.25 - 0.5) ** 0.5)
    ref = obrown.brownian_increment(seeds, per, 0.01, 80.0, 0.5, 3.25, 1.0 / (3.25 - 0.5) ** 0.5)
    assert np.abs(out.cpu().numpy().reshape(3, per) - ref).max() < 1e-4       # hardware log2/sqrt/cos vs numpy
    # odd depth (the last level uses only the node's own deviate) and a shallow tree
    for depth in (1, 7, 12):
        ops.brownian(out, g(torch.tensor(seeds, dtype=torch.int64)), 0.01, 80.0, 0.5, 3.25, 1.0, depth)
        ref = obrown.brownian_increment(seeds, per, 0.01, 80.0, 0.5, 3.25, 1.0, depth)
        assert np.abs(out.cpu().numpy().reshape(3, per) - ref).max() < 1e-4, depth
    # path consistency: W(a,c) == W(a,b) + W(b,c)
    ab, bc, ac = (torch.empty(3, 3, 5, 7, device=DEV) for _ in range(3))
    s = g(torch.tensor(seeds, dtype=torch.int64))
    ops.brownian(ab, s, 0.01, 80.0, 1.0, 2.0, 1.0)
    ops.brownian(bc, s, 0.01, 80.0, 2.0, 5.5, 1.0)
    ops.brownian(ac, s, 0.01, 80.0, 1.0, 5.5, 1.0)
    assert (ab + bc - ac).abs().max() < 1e-5
    # unit variance of the normalised increment, independent elements
    big = torch.empty(4, 1 << 16, device=DEV)
    ops.brownian(big, g(torch.arange(4, dtype=torch.int64) + 99), 0.01, 160.0, 0.02, 0.03, 1.0 / 0.01 ** 0.5)
    assert abs(big.var().item() - 1.0) < 0.02 and abs(big.mean().item()) < 0.01
    assert abs(torch.corrcoef(big[:2])[0, 1].item()) < 0.02


def test_index_addressed_normals_vs_oracle(ops):
    """kd_randn_f32 (the device noise source of a seeded job: sample.py --noise device) against oracle.brownian.randn_indexed: integer
    Philox bit-exact, Box-Muller on the hardware log2 / sqrt / cos (1e-5 absolute per unit of scale); a sample's values do not depend
    on the batch it is drawn in or on the launch size; ragged lengths; moments of a 25 MB draw (the job's batch)."""
    seeds = [12345, 2 ** 63 - 7, 0]
    for per, shape in ((3 * 6 * 8, (3, 3, 6, 8)), (3 * 5 * 7, (3, 3, 5, 7)), (1, (3, 1))):       # a multiple of 4, a ragged tail, one element
        for draw, scale in ((0, 1.0), (3, 160.0)):
            out = torch.full(shape, float("nan"), device=DEV)
            ops.randn_indexed(out, g(torch.tensor(seeds, dtype=torch.int64)), draw=draw, scale=scale)
            ref = obrown.randn_indexed(seeds, per, draw=draw, scale=scale)
            assert np.abs(out.cpu().numpy().reshape(3, per) - ref).max() < 1e-5 * scale, (per, draw)
    big = torch.empty(32, 3, 256, 256, device=DEV)
    keys = torch.arange(32, dtype=torch.int64) * 977 + 5
    ops.randn_indexed(big, g(keys), draw=0, scale=1.0)
    assert abs(big.mean().item()) < 2e-3 and abs(big.var().item() - 1) < 2e-3 and abs((big ** 4).mean().item() - 3) < 2e-2
    assert abs(torch.corrcoef(big[:2].reshape(2, -1))[0, 1].item()) < 1e-2
    sub = torch.empty(2, 3, 256, 256, device=DEV)                      # images 7 and 30 drawn alone: the same values
    ops.randn_indexed(sub, g(keys[[7, 30]]), draw=0, scale=1.0)
    assert torch.equal(sub[0], big[7]) and torch.equal(sub[1], big[30])
    ref = obrown.randn_indexed([int(keys[7])], 4096)
    assert np.abs(big[7].reshape(-1)[:4096].cpu().numpy() - ref[0]).max() < 1e-5
    with pytest.raises(ValueError):
        ops.randn_indexed(big, g(keys[:3]))
    with pytest.raises(RuntimeError):
        ops.randn_indexed(torch.empty(1, 4), keys[:1])               # host tensors: no CPU fallback


def test_brownian_cached_endpoints(ops):
    """kd_brownian_cached_f32: stored end points reproduce the uncached increments bit for bit."""
    seeds = g(torch.tensor([7, 8], dtype=torch.int64))
    shape = (2, 3 * 11 * 13)
    plain = lambda a, b, m: ops.brownian(torch.empty(shape, device=DEV), seeds, 0.01, 80.0, a, b, m)
    w = [torch.full(shape, float("nan"), device=DEV) for _ in range(3)]
    out = torch.empty(shape, device=DEV)
    ops.brownian_cached(out, w[0], False, w[1], False, seeds, 0.01, 80.0, 1.0, 2.0, 0.7)          # fills W(1), W(2)
    assert torch.equal(out, plain(1.0, 2.0, 0.7)) and torch.isfinite(w[0]).all() and torch.isfinite(w[1]).all()
    ops.brownian_cached(out, w[1], True, w[2], False, seeds, 0.01, 80.0, 2.0, 5.5, 1.3)           # reads W(2), fills W(5.5)
    assert torch.equal(out, plain(2.0, 5.5, 1.3))
    ops.brownian_cached(out, w[0], True, w[2], True, seeds, 0.01, 80.0, 1.0, 5.5, -2.0)           # both read
    assert torch.equal(out, plain(1.0, 5.5, -2.0))
    ops.brownian_cached(out, None, False, w[2], True, seeds, 0.01, 80.0, 0.3, 5.5, 1.0)           # computed, not stored
    assert torch.equal(out, plain(0.3, 5.5, 1.0))
    with pytest.raises(RuntimeError):
        ops.brownian_cached(out, None, True, w[2], True, seeds, 0.01, 80.0, 0.3, 5.5, 1.0)


def test_to_uint8(ops):
    x = torch.linspace(-1.5, 1.5, 1001)
    ref = (((x.clamp(-1, 1) + 1) / 2) * 255).to(torch.uint8)
    assert torch.equal(ops.to_uint8(g(x)).cpu(), ref)


def test_brownian_multi_scale_statistics(ops):
    """The virtual tree behaves like Brownian motion at every scale: Var[W(t1) - W(t0)] = t1 - t0 across four decades of
    interval length (short intervals deep in the tree, long ones near its root), increments over disjoint intervals are
    uncorrelated (also when they are siblings under one tree node), and a coarse increment is the sum of its parts."""
    seeds = g(torch.arange(8, dtype=torch.int64) * 7919 + 5)
    n = 1 << 15
    inc = lambda a, b: ops.brownian(torch.empty(8, n, device=DEV), seeds, 0.01, 160.0, a, b, 1.0)
    for a, b in ((0.0100, 0.0101), (0.02, 0.03), (0.5, 0.9), (3.0, 11.0), (20.0, 150.0), (0.01, 160.0)):
        w = inc(a, b)
        assert abs(w.var().item() / (b - a) - 1.0) < 0.03, (a, b)
        assert abs(w.mean().item()) < 0.02 * (b - a) ** 0.5, (a, b)
    pairs = (((0.5, 0.9), (0.9, 1.7)), ((0.02, 0.03), (40.0, 41.0)), ((79.99, 80.0), (80.0, 80.02)), ((1.0, 2.0), (2.0, 2.0001)))
    for (a, b), (c, d) in pairs:
        x, y = inc(a, b).flatten(), inc(c, d).flatten()
        assert abs(torch.corrcoef(torch.stack([x, y]))[0, 1].item()) < 0.01, ((a, b), (c, d))
    parts = inc(1.0, 1.5) + inc(1.5, 4.0) + inc(4.0, 32.0)
    assert (parts - inc(1.0, 32.0)).abs().max() < 1e-4
    # different samples (seeds) and different elements are independent streams
    w = inc(0.3, 7.0)
    assert abs(torch.corrcoef(w[:4])[0, 1:].abs().max().item()) < 0.03
    assert abs(torch.corrcoef(torch.stack([w[0, :-1], w[0, 1:]]))[0, 1].item()) < 0.03


# ---- bf16 arithmetic mode (KD_PREC_BF16): the same ops on bf16 activations ------------------------------------------------
BF = torch.bfloat16


def _bf(t):
    return t.to(BF).to(DEV).contiguous()


def _rt(t):
    """what a bf16 kernel sees of an fp32 tensor"""
    return t.to(BF).float()


@pytest.mark.parametrize("M,N,K", [(4096, 128, 128), (4100, 256, 384), (2048, 256, 256), (1000, 96, 192), (300, 160, 64), (8192, 512, 1536), (70, 32, 12)])
def test_bf16_gemm_plain_and_residual(ops, M, N, K):
    """kd_gemm_bf16 through the W-stationary, tiled and generic kernels (the shape picks the kernel): bf16 in / out, fp32 accumulate."""
    a, w, r = rn(M, K, seed=1), rn(N, K, seed=2, scale=K ** -0.5), rn(M, N, seed=3)
    ref = _rt(a) @ _rt(w).T
    y = ops.linear(_bf(a), g(w))
    assert y.dtype == BF and relerr(y, ref) < 8e-3
    y = ops.linear(_bf(a), g(w), residual=_bf(r))
    assert relerr(y, ref + _rt(r)) < 8e-3
    x = _bf(r)
    ops.linear(_bf(a), g(w), residual=x, out=x)                   # in place (how the model uses it)
    assert relerr(x, ref + _rt(r)) < 8e-3


@pytest.mark.parametrize("B,T,K,d_ff", [(2, 4096, 128, 384), (3, 1024, 256, 768), (32, 64, 512, 1536), (4, 49, 256, 768), (2, 50, 100, 96)])
def test_bf16_norm_linear_and_geglu(ops, B, T, K, d_ff):
    """AdaRMSNorm -> Linear / LinearGEGLU in bf16 mode (W-stationary at K = 128, A-stationary ring at K = 256 / 512, generic otherwise)
    against the fp32 oracle on the bf16-rounded input."""
    x, scale = rn(B, T, K, seed=4), 1 + 0.2 * rn(B, K, seed=5)
    w, wg = rn(d_ff, K, seed=6, scale=K ** -0.5), rn(2 * d_ff, K, seed=7, scale=K ** -0.5)
    xn = hdit.rms_norm(_rt(x), scale[:, None, :])
    y = ops.norm_linear(_bf(x), g(scale), g(w), rows_per_sample=T)
    assert y.dtype == BF and relerr(y, xn @ _rt(w).T) < 1.2e-2
    from k_diffusion_amd import _native as nat
    y = ops.norm_linear(_bf(x), g(scale), g(wg), rows_per_sample=T, epi=nat.EPI_GEGLU)
    assert relerr(y, hdit.linear_geglu(xn, _rt(wg))) < 1.5e-2


@pytest.mark.parametrize("H,W,nh,B,K", [(64, 64, 2, 2, 128), (32, 32, 4, 2, 256), (16, 16, 8, 4, 512), (7, 7, 4, 3, 256)])
def test_bf16_qkv_epilogue(ops, H, W, nh, B, K):
    """The bf16 qkv epilogue (cosine-sim scale + RoPE with hardware sin / cos from the position / frequency tables) against
    the oracle's scale_for_cosine_sim + apply_rotary_emb on the fp32 projection."""
    from k_diffusion_amd import _native as nat
    T, d = H * W, nh * 64
    x, scale = rn(B, T, K, seed=8), 1 + 0.2 * rn(B, K, seed=9)
    w = rn(3 * d, K, seed=10, scale=K ** -0.5)
    qs = torch.linspace(5.0, 12.0, nh)
    pos, freqs = hdit.axial_pos(H, W).reshape(T, 2), hdit.rope_freqs(nh)
    qkv = ops.norm_linear(_bf(x), g(scale), g(w), rows_per_sample=T, epi=nat.EPI_QKV,
                          qk=(g(qs), g(pos.contiguous()), g((freqs / (2 * np.pi)).contiguous()), nh)).float().cpu().view(B, H, W, 3, nh, 64)
    ref = (hdit.rms_norm(_rt(x), scale[:, None, :]) @ _rt(w).T).view(B, H, W, 3, nh, 64)
    theta = hdit.rope_theta(hdit.axial_pos(H, W), freqs)
    q_ref, k_ref = hdit.cosine_sim_scale(ref[..., 0, :, :], ref[..., 1, :, :], qs)
    assert relerr(qkv[..., 0, :, :], hdit.apply_rope(q_ref, theta)) < 1.2e-2
    assert relerr(qkv[..., 1, :, :], hdit.apply_rope(k_ref, theta)) < 1.2e-2
    assert relerr(qkv[..., 2, :, :], ref[..., 2, :, :]) < 1.2e-2


@pytest.mark.parametrize("B,T,K,d_ff", [(32, 256, 512, 1536), (3, 256, 512, 1536), (8, 1024, 256, 768), (5, 256, 256, 384), (2, 512, 512, 192), (3, 768, 256, 768)])
def test_proj_block_bf16_matches_the_projection_kernel(KD, ops, B, T, K, d_ff):
    """kd_proj_block_bf16 (round 5: AdaRMSNorm -> up projection + GEGLU, and AdaRMSNorm -> qkv projection + cosine-sim scale + RoPE, as a
    workgroup per (256-row group, six 64-row half blocks of the packed weight) with the group's rows normalised once) against kd_gemm_bf16 on
    the same descriptor: same products in the same order, BIT-IDENTICAL; and against the oracle at the bf16 mode's tolerance.  One and
    several row groups per sample, group counts that are and are not multiples of 8 (the XCD-aware and the plain workgroup order)."""
    from k_diffusion_amd import _native as nat
    assert nat.lib().kd_proj_block_bf16_supported(T, K, d_ff, nat.EPI_GEGLU) == 1 and nat.lib().kd_proj_block_bf16_supported(T, K, 3 * K, nat.EPI_QKV) == 1
    x, scale = rn(B, T, K, seed=8) * (1 + rn(B, T, 1, seed=3).abs()), 1 + 0.2 * rn(B, K, seed=9)
    wg = rn(2 * d_ff, K, seed=7, scale=K ** -0.5)
    xb, sc, wd = _bf(x), g(scale), g(wg)
    two = ops.norm_linear(xb, sc, wd, rows_per_sample=T, epi=nat.EPI_GEGLU)
    one = ops.proj_block(xb, sc, wd, rows_per_sample=T)
    assert one.dtype == BF and one.shape == (B, T, d_ff)
    print(f"proj_block GEGLU B={B} T={T} K={K} d_ff={d_ff}: {int((one != two).sum())} of {one.numel()} outputs differ")
    assert torch.equal(one, two)
    assert relerr(one.float().cpu(), hdit.linear_geglu(hdit.rms_norm(_rt(x), scale[:, None, :]), _rt(wg))) < 1.5e-2
    # the qkv form: q, k prepared (cosine-sim scale + RoPE from the token's position inside ITS sample), v scaled by the row factor
    nh = K // 64
    H, W = T // 16, 16
    w = rn(3 * K, K, seed=10, scale=K ** -0.5)
    qk = (g(torch.linspace(5.0, 12.0, nh)), g(hdit.axial_pos(H, W).reshape(T, 2).contiguous()), g((hdit.rope_freqs(nh) / (2 * np.pi)).contiguous()), nh)
    q2 = ops.norm_linear(xb, sc, g(w), rows_per_sample=T, epi=nat.EPI_QKV, qk=qk)
    q1 = ops.proj_block(xb, sc, g(w), rows_per_sample=T, epi=nat.EPI_QKV, qk=qk)
    print(f"proj_block QKV B={B} T={T} K={K}: {int((q1 != q2).sum())} of {q1.numel()} outputs differ")
    assert q1.shape == (B, T, 3 * K) and torch.equal(q1, q2)
    with pytest.raises(RuntimeError):
        ops.proj_block(_bf(rn(2, 64, K, seed=1)), g(scale[:2]), wd, rows_per_sample=64)


@pytest.mark.parametrize("B,T,K,d_ff", [(32, 256, 512, 1536), (3, 256, 512, 1536), (8, 1024, 256, 768), (5, 256, 256, 384), (2, 100, 512, 192), (1, 333, 256, 768)])
def test_gemm_mx8_vs_the_restated_arithmetic(KD, ops, B, T, K, d_ff):
    """kd_gemm_mx8 (round 6, the fp8 arithmetic mode: AdaRMSNorm -> projection on v_mfma_scale_f32_32x32x64_f8f6f4 -- e4m3 weights with one
    power-of-two scale per output channel, activations quantised per (row, 32-k block) with a power-of-two block scale) against the oracle's
    restatement of exactly that arithmetic (oracle/hdit.py: mx8_quantize_rows / mx8_quantize_weight; products exact, fp32 sums): all three
    epilogues, full and ragged panels, one and several n-splits.  The quantised operands are the same values on both sides, so what remains
    is fp32 summation order + the bf16 rounding of the output -- far inside the distance between this arithmetic and the unquantised one,
    which is asserted to be visible (the kernel really quantises)."""
    from k_diffusion_amd import _native as nat
    lib = nat.lib()
    M = B * T
    assert lib.kd_gemm_mx8_supported(M, d_ff, K, nat.EPI_GEGLU, 1) == 1 and lib.kd_gemm_mx8_supported(M, 3 * K, K, nat.EPI_QKV, 1) == 1
    assert lib.kd_gemm_mx8_supported(M, d_ff, 384, nat.EPI_GEGLU, 1) == 0 and lib.kd_gemm_mx8_supported(M, d_ff, K, nat.EPI_GEGLU, 0) == 0
    x, scale = rn(B, T, K, seed=8) * (1 + rn(B, T, 1, seed=3).abs()), 1 + 0.2 * rn(B, K, seed=9)
    x[0, 0, 32:64] = 0.0                                           # an all-zero block: scale byte at its floor, zeros out
    x[0, 1, 5] = 300.0                                             # one outlier in a block: its neighbours lose their bits, nobody saturates
    xb, sc = _bf(x), g(scale)
    xr = _rt(x)
    rs = torch.rsqrt(xr.square().mean(-1, keepdim=True) + 1e-6)
    uq = hdit.mx8_quantize_rows(xr * scale[:, None, :])
    assert relerr(uq, xr * scale[:, None, :]) > 1e-3               # (3 mantissa bits)
    # plain store
    w = rn(2 * K, K, seed=11, scale=K ** -0.5)
    w[3] = 0.0                                                     # an all-zero channel
    got = ops.norm_linear(xb, sc, g(w), rows_per_sample=T, mx8=True)
    ref = (uq @ hdit.mx8_quantize_weight(w).T) * rs
    plain = hdit.rms_norm(xr, scale[:, None, :]) @ w.T
    e, gap = relerr(got.float().cpu(), ref), relerr(ref, plain)
    print(f"mx8 store B={B} T={T} K={K}: vs restated arithmetic {e:.3e} (that arithmetic vs unquantised: {gap:.3e})")
    assert got.dtype == BF and got.shape == (B, T, 2 * K) and e < 6e-3 and gap > 2 * e
    assert not got[..., 3].any()
    # GEGLU
    wg = rn(2 * d_ff, K, seed=7, scale=K ** -0.5)
    got = ops.norm_linear(xb, sc, g(wg), rows_per_sample=T, epi=nat.EPI_GEGLU, mx8=True)
    h = (uq @ hdit.mx8_quantize_weight(wg).T) * rs
    ref = h[..., :d_ff] * torch.nn.functional.gelu(h[..., d_ff:])
    e = relerr(got.float().cpu(), ref)
    print(f"mx8 GEGLU d_ff={d_ff}: vs restated arithmetic {e:.3e}")
    assert got.shape == (B, T, d_ff) and e < 8e-3
    # qkv: cosine-sim scale + RoPE of q, k in the epilogue, v scaled by the row factor
    nh = K // 64
    H, W = (T // 16, 16) if T % 16 == 0 else (T, 1)
    wq = rn(3 * K, K, seed=10, scale=K ** -0.5)
    qs, freqs = torch.linspace(5.0, 12.0, nh), hdit.rope_freqs(nh)
    qk = (g(qs), g(hdit.axial_pos(H, W).reshape(T, 2).contiguous()), g((freqs / (2 * np.pi)).contiguous()), nh)
    got = ops.norm_linear(xb, sc, g(wq), rows_per_sample=T, epi=nat.EPI_QKV, qk=qk, mx8=True).float().cpu().view(B, H, W, 3, nh, 64)
    r = ((uq @ hdit.mx8_quantize_weight(wq).T) * rs).view(B, H, W, 3, nh, 64)
    theta = hdit.rope_theta(hdit.axial_pos(H, W), freqs)
    q_ref, k_ref = hdit.cosine_sim_scale(r[..., 0, :, :], r[..., 1, :, :], qs)
    q_ref, k_ref = hdit.apply_rope(q_ref, theta), hdit.apply_rope(k_ref, theta)
    e = max(relerr(got[..., 0, :, :], q_ref), relerr(got[..., 1, :, :], k_ref), relerr(got[..., 2, :, :], r[..., 2, :, :]))
    print(f"mx8 qkv: vs restated arithmetic {e:.3e}")
    assert e < 8e-3
    # the hidden activation as the next product's operand (c_split): e4m3 rows + E8M0 scale bytes must be EXACTLY the restated quantiser applied
    # to the kernel's own fp32 GEGLU values -- checked through the values they decode to, against the bf16 result of the same launch shape --
    # and the tiled form (both operands e4m3, down projection + residual) against exact products of those decoded operands
    h8, hs = ops.norm_linear(xb, sc, g(wg), rows_per_sample=T, epi=nat.EPI_GEGLU, mx8=True, c_fp8=True)
    assert h8.dtype == torch.uint8 and h8.shape == (B, T, d_ff) and hs.shape == (B, T, d_ff // 32)
    dec = h8.cpu().view(torch.float8_e4m3fn).float().view(B, T, d_ff // 32, 32) * torch.ldexp(torch.ones(()), hs.cpu().to(torch.int32) - 127)[..., None]
    dec = dec.view(B, T, d_ff)
    e = relerr(dec, hdit.mx8_quantize_rows(ref))
    print(f"mx8 GEGLU -> e4m3 hidden: decoded vs the restated quantiser on the restated hidden {e:.3e}")
    assert e < 7e-2 and relerr(dec, ref) < 7e-2                   # (a last-bit difference of a hidden value moves its e4m3 code by one step: 2^-3 of the block maximum at worst)
    assert torch.equal(hdit.mx8_quantize_rows(dec), dec)          # what was stored IS a point of the quantiser's grid (block scale minimal, values representable)
    if lib.kd_gemm_mx8_supported(M, K, d_ff, nat.EPI_RESIDUAL, 0):          # down-projection lengths the tiled form takes: 256, 512, 768, 1536
        wdn = rn(K, d_ff, seed=13, scale=d_ff ** -0.5)
        got = ops.linear_mx8(h8, hs, g(wdn), residual=xb)
        refd = dec @ hdit.mx8_quantize_weight(wdn).T + xr
        e = relerr(got.float().cpu(), refd)
        print(f"mx8 tiled down projection K={d_ff} N={K}: vs exact products of the decoded operands {e:.3e}")
        assert got.dtype == BF and got.shape == (B, T, K) and e < 6e-3
        assert relerr(ops.linear_mx8(h8, hs, g(wdn)).float().cpu(), dec @ hdit.mx8_quantize_weight(wdn).T) < 6e-3
    # an fp8 checkpoint's weight (already e4m3 x power-of-two channel scale: checkpoint.quantize_fp8) enters bit for bit: the packed image of W
    # and of its fp8-stored value are the same image
    wq8 = KD.checkpoint.fake_quantize_fp8(w)
    assert torch.equal(hdit.mx8_quantize_weight(w), wq8) and torch.equal(hdit.mx8_quantize_weight(wq8), wq8)
    assert torch.equal(ops.norm_linear(xb, sc, g(wq8), rows_per_sample=T, mx8=True), ops.norm_linear(xb, sc, g(w), rows_per_sample=T, mx8=True))
    # shapes it does not take are refused
    with pytest.raises(RuntimeError):
        ops.norm_linear(_bf(rn(1, 64, K, seed=1)), g(scale[:1]), g(w), rows_per_sample=64, mx8=True)


@pytest.mark.parametrize("M,K,N", [(8192, 1536, 512), (32768, 768, 256), (40000, 256, 256), (300, 512, 512), (4111, 768, 384), (128, 1536, 128)])
def test_gemm_mx8_tiled_form(KD, ops, M, K, N):
    """kd_gemm_mx8's second form (norm = 0, a_split = 1: e4m3 rows + E8M0 block scales x a kd_pack_weight_mx8 image, both operands by LDS-DMA;
    the fp8 mode's down projection) on operands quantised by the oracle's own quantiser on the host: every K it takes (256, 512, 768, 1536),
    grids of at most one tile per CU (the 4-slot ring) and larger ones (two workgroups per CU), ragged last row tile, with and without the
    residual.  The products of e4m3 values are exact and sums are fp32: what remains is summation order + the bf16 rounding of the output."""
    from k_diffusion_amd import _native as nat
    assert nat.lib().kd_gemm_mx8_supported(M, N, K, nat.EPI_RESIDUAL, 0) == 1 and nat.lib().kd_gemm_mx8_supported(M, N, 384, nat.EPI_RESIDUAL, 0) == 0
    u = rn(M, K, seed=4) * torch.logspace(-2, 2, K // 32).repeat_interleave(32)[None, :]       # block maxima over four decades
    u[5, 64:96] = 0.0
    blocks = u.view(M, K // 32, 32)
    s = hdit.mx8_scale(blocks.abs().amax(-1, keepdim=True))
    q8 = (blocks / s).to(torch.float8_e4m3fn)
    a8 = q8.view(torch.uint8).reshape(M, K).contiguous()
    sb = (torch.log2(s).round().to(torch.int32) + 127).to(torch.uint8).reshape(M, K // 32).contiguous()
    dec = (q8.float() * s).reshape(M, K)
    assert torch.equal(dec, hdit.mx8_quantize_rows(u))
    w = rn(N, K, seed=6, scale=K ** -0.5)
    res = rn(M, N, seed=2)
    ref = dec.double() @ hdit.mx8_quantize_weight(w).double().T
    got = ops.linear_mx8(g(a8), g(sb), g(w))
    got_r = ops.linear_mx8(g(a8), g(sb), g(w), residual=_bf(res))
    e0, e1 = relerr(got.float().cpu(), ref.float()), relerr(got_r.float().cpu(), (ref + _rt(res).double()).float())
    print(f"mx8 tiled M={M} K={K} N={N}: store {e0:.3e}, + residual {e1:.3e}")
    assert got.dtype == BF and got.shape == (M, N) and e0 < 6e-3 and e1 < 6e-3
    with pytest.raises(RuntimeError):                                # bf16 rows are not this form's operand
        ops.gemm(_bf(u), g(w), torch.empty(M, N, device=DEV, dtype=BF), M=M, N=N, K=K, precision=nat.PREC_BF16, mx8=True)


@pytest.mark.parametrize("nh,B,K", [(8, 32, 512), (8, 3, 512), (4, 16, 256), (4, 5, 256)])
def test_attn_block_bf16_matches_two_launches(KD, ops, nh, B, K):
    """kd_attn_block_bf16 (round 5: AdaRMSNorm -> qkv projection of a head -> cosine-sim + RoPE -> dense attention in one launch per
    (sample, head); q, k, v never reach HBM) against the two launches it replaces (kd_gemm_bf16 EPI_QKV + kd_attn_global_bf16): the same
    arithmetic operation for operation, so BIT-IDENTICAL; and against the oracle's separate steps at the bf16 mode's tolerance.  Batches
    that are and are not multiples of 8 (the XCD-aware and the plain workgroup order)."""
    from k_diffusion_amd import _native as nat
    H = W = 16
    T, d = H * W, nh * 64
    assert d == K and nat.lib().kd_attn_block_bf16_supported(T, K, nh) == 1
    x, scale = rn(B, T, K, seed=8) * (1 + rn(B, T, 1, seed=3).abs()), 1 + 0.2 * rn(B, K, seed=9)
    w = rn(3 * d, K, seed=10, scale=K ** -0.5)
    qs = torch.linspace(5.0, 12.0, nh)
    pos, freqs = hdit.axial_pos(H, W).reshape(T, 2), hdit.rope_freqs(nh)
    qk = (g(qs), g(pos.contiguous()), g((freqs / (2 * np.pi)).contiguous()), nh)
    xb, sc, wd = _bf(x), g(scale), g(w)
    qkv = ops.norm_linear(xb, sc, wd, rows_per_sample=T, epi=nat.EPI_QKV, qk=qk)
    two = ops.attn_global(qkv, nh)
    one = ops.attn_block(xb, sc, wd, rows_per_sample=T, qk=qk)
    assert one.dtype == BF and one.shape == (B, T, K)
    ndiff = int((one != two).sum())
    print(f"attn_block B={B} K={K}: {ndiff} of {one.numel()} outputs differ, max |diff| {float((one.float() - two.float()).abs().max()):.3e}")
    assert torch.equal(one, two)
    # the oracle: norm -> projection -> cosine-sim + RoPE -> softmax attention, fp32 on the bf16-rounded operands
    ref = (hdit.rms_norm(_rt(x), scale[:, None, :]) @ _rt(w).T).view(B, H, W, 3, nh, 64)
    theta = hdit.rope_theta(hdit.axial_pos(H, W), freqs)
    q_ref, k_ref = hdit.cosine_sim_scale(ref[..., 0, :, :], ref[..., 1, :, :], qs)
    q_ref, k_ref, v_ref = hdit.apply_rope(q_ref, theta), hdit.apply_rope(k_ref, theta), ref[..., 2, :, :]
    tohead = lambda t: t.reshape(B, T, nh, 64).transpose(1, 2)
    att = torch.softmax(tohead(q_ref) @ tohead(k_ref).transpose(-1, -2), dim=-1) @ tohead(v_ref)
    assert relerr(one.float().cpu(), att.transpose(1, 2).reshape(B, T, K)) < 2e-2
    # shapes it does not take are refused, not approximated
    with pytest.raises(RuntimeError):
        ops.attn_block(_bf(rn(2, 64, K, seed=1)), g(scale[:2]), wd, rows_per_sample=64, qk=qk)


@pytest.mark.parametrize("H,W,nh,B,K", [(64, 64, 2, 2, 128), (32, 32, 4, 2, 256), (32, 16, 4, 3, 256), (24, 24, 2, 1, 128), (20, 20, 2, 2, 128), (16, 16, 8, 4, 512),
                                        (20, 20, 4, 3, 256), (12, 20, 8, 3, 512)])
def test_split3_projections_round3(KD, ops, monkeypatch, H, W, nh, B, K):
    """The round-3 fp32-parity projections (gemm_x3.hip / gemm_x3t.hip: lane-owns-row epilogues, RoPE angles from positions /
    frequencies, qkv operands stored split for the attention cores) against the oracle's separate steps, and against the
    round-1 kernels they replace (option x3 = 0) at split-bf16x3 accuracy.  Full panels, ragged panels (24 x 24 tokens) and rows of
    several samples inside one 128-row panel."""
    from k_diffusion_amd import _native as nat
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    T, d = H * W, nh * 64
    x, scale = rn(B, T, K, seed=8), 1 + 0.2 * rn(B, K, seed=9)
    w = rn(3 * d, K, seed=10, scale=K ** -0.5)
    qs = torch.linspace(5.0, 12.0, nh)
    pos, freqs = hdit.axial_pos(H, W).reshape(T, 2), hdit.rope_freqs(nh)
    cos, sin = _tables(H, W, nh)
    qk = (g(qs), g(cos), g(sin), nh, g(pos.contiguous()), g((freqs / (2 * np.pi)).contiguous()))
    qkv = ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=T, epi=nat.EPI_QKV, qk=qk).cpu().view(B, H, W, 3, nh, 64)
    ref = (hdit.rms_norm(x, scale[:, None, :]) @ w.T).view(B, H, W, 3, nh, 64)
    theta = hdit.rope_theta(hdit.axial_pos(H, W), freqs)
    q_ref, k_ref = hdit.cosine_sim_scale(ref[..., 0, :, :], ref[..., 1, :, :], qs)
    assert relerr(qkv[..., 0, :, :], hdit.apply_rope(q_ref, theta)) < 1e-4
    assert relerr(qkv[..., 1, :, :], hdit.apply_rope(k_ref, theta)) < 1e-4
    assert relerr(qkv[..., 2, :, :], ref[..., 2, :, :]) < 1e-4
    # the split-stored form is the same numbers as (hi, hi, lo, lo) bf16 pairs
    packed = ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=T, epi=nat.EPI_QKV, qk=qk, qkv_packed=True)
    pw = packed.view(torch.int32).view(-1, 4)
    hi = torch.stack([(pw[:, 0] << 16), (pw[:, 0] & -65536), (pw[:, 1] << 16), (pw[:, 1] & -65536)], dim=1).view(torch.float32)
    lo = torch.stack([(pw[:, 2] << 16), (pw[:, 2] & -65536), (pw[:, 3] << 16), (pw[:, 3] & -65536)], dim=1).view(torch.float32)
    assert relerr((hi + lo).cpu().view_as(qkv), qkv) < 2.0 ** -15
    # GEGLU and plain store
    wg = rn(2 * 3 * K, K, seed=11, scale=K ** -0.5)
    y = ops.norm_linear(g(x), g(scale), g(wg), rows_per_sample=T, epi=nat.EPI_GEGLU)
    assert relerr(y, hdit.linear_geglu(hdit.rms_norm(x, scale[:, None, :]), wg)) < 1e-4
    ws = rn(256, K, seed=12, scale=K ** -0.5)
    y = ops.norm_linear(g(x), g(scale[0].contiguous()), g(ws), rows_per_sample=T)            # shared gain
    assert relerr(y, hdit.rms_norm(x, scale[0]) @ ws.T) < 1e-4
    # the pre-split operand path (csrc/gemm_x3t.hip): norm -> bf16 hi / lo planes once, then GEMMs that move both operands by LDS-DMA
    xh, xl = ops.norm_split(g(x), g(scale), rows_per_sample=T)
    xn = hdit.rms_norm(x, scale[:, None, :])
    assert torch.equal(xh.float().cpu(), xn.to(torch.bfloat16).float()) or relerr(xh.float(), xn) < 2.0 ** -8
    assert relerr(xh.float().cpu() + xl.float().cpu(), xn) < 2.0 ** -15
    o = torch.empty(B, T, 3 * d, device="cuda")
    ops.gemm(None, g(w), o, M=B * T, N=3 * d, K=K, epi=nat.EPI_QKV, rows_per_sample=T, qk=qk, a_planes=(xh, xl))
    assert relerr(o.cpu().view(B, H, W, 3, nh, 64), qkv) < 1e-4
    hh, hl = torch.empty(B, T, 3 * K, device="cuda", dtype=torch.bfloat16), torch.empty(B, T, 3 * K, device="cuda", dtype=torch.bfloat16)
    ops.gemm(None, g(wg), None, M=B * T, N=3 * K, K=K, epi=nat.EPI_GEGLU, a_planes=(xh, xl), c_planes=(hh, hl))
    hid_ref = hdit.linear_geglu(xn, wg)
    assert relerr(hh.float().cpu() + hl.float().cpu(), hid_ref) < 1e-4
    if K <= 256 and B * T >= 512:        # the fused norm -> GEGLU kernel writing planes itself
        h2, l2 = torch.empty_like(hh), torch.empty_like(hl)
        ops.gemm(g(x), g(wg), None, M=B * T, N=3 * K, K=K, epi=nat.EPI_GEGLU, norm_scale=g(scale), scale_stride=K, rows_per_sample=T, c_planes=(h2, l2))
        assert relerr(h2.float().cpu() + l2.float().cpu(), hid_ref) < 1e-4
    wd = rn(128, 3 * K, seed=13, scale=(3 * K) ** -0.5)
    res = rn(B, T, 128, seed=14)
    y = torch.empty(B, T, 128, device="cuda")
    ops.gemm(None, g(wd), y, M=B * T, N=128, K=3 * K, epi=nat.EPI_RESIDUAL, residual=g(res), a_planes=(hh, hl))
    assert relerr(y, hid_ref @ wd.T + res) < 1e-4
    ops.gemm(None, g(wd), y, M=B * T, N=128, K=3 * K, a_planes=(hh, hl))
    assert relerr(y, hid_ref @ wd.T) < 1e-4
    # against the round-1 kernels
    nat.set_option("x3", 0)
    try:
        old = ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=T, epi=nat.EPI_QKV, qk=qk).cpu().view(B, H, W, 3, nh, 64)
    finally:
        nat.set_option("x3", 1)
    assert relerr(qkv, old) < 1e-4 and (B * T < 512 or K >= 512 or not torch.equal(qkv, old))      # (another kernel really ran)


@pytest.mark.few_rows
@pytest.mark.parametrize("H,W,nh,B,K", [(16, 16, 8, 1, 512), (7, 7, 4, 4, 256), (32, 32, 4, 1, 256), (9, 12, 2, 3, 128), (16, 16, 18, 2, 384), (6, 6, 1, 4, 64)])
def test_split3_few_rows_latency_kernel(KD, ops, monkeypatch, request, H, W, nh, B, K):
    """The few-rows form of the fp32-parity projections (round 4, csrc/gemm_x3s.hip: 32 rows x one half tile per workgroup, K split over
    its 8 waves, operands straight into registers, partial sums reduced in wave order) at batch-1 shapes of the headline config, the
    MNIST shape (49 tokens per sample: a 32-row group spans samples), ragged row counts, 18 heads and K = 64: every epilogue against the
    oracle at split-bf16x3 accuracy, against the throughput kernels it stands in for, and bit-identical from run to run."""
    from k_diffusion_amd import _native as nat
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    T, d, M = H * W, nh * 64, B * H * W
    assert 128 < M <= 1024
    x, scale = rn(B, T, K, seed=8), 1 + 0.2 * rn(B, K, seed=9)
    xn = hdit.rms_norm(x, scale[:, None, :])
    w = rn(3 * d, K, seed=10, scale=K ** -0.5)
    qs = torch.linspace(5.0, 12.0, nh)
    pos, freqs = hdit.axial_pos(H, W).reshape(T, 2), hdit.rope_freqs(nh)
    cos, sin = _tables(H, W, nh)
    qk = (g(qs), g(cos), g(sin), nh, g(pos.contiguous()), g((freqs / (2 * np.pi)).contiguous()))
    theta = hdit.rope_theta(hdit.axial_pos(H, W), freqs)
    wg = rn(2 * 3 * K, K, seed=11, scale=K ** -0.5)
    ws = rn(256, K, seed=12, scale=K ** -0.5)
    wd = rn(K, 3 * K, seed=13, scale=(3 * K) ** -0.5)
    hid, res = rn(B, T, 3 * K, seed=15), rn(B, T, K, seed=14)

    def run_all():
        out = {}
        out["qkv"] = ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=T, epi=nat.EPI_QKV, qk=qk)
        out["packed"] = ops.norm_linear(g(x), g(scale), g(w), rows_per_sample=T, epi=nat.EPI_QKV, qk=qk, qkv_packed=True)
        out["geglu"] = ops.norm_linear(g(x), g(scale), g(wg), rows_per_sample=T, epi=nat.EPI_GEGLU)
        out["shared"] = ops.norm_linear(g(x), g(scale[0].contiguous()), g(ws), rows_per_sample=T)
        out["plain"] = ops.gemm(g(hid), g(wd), torch.empty(M, K, device=DEV), M=M, N=K, K=3 * K, out_add=0.5)
        out["res"] = ops.gemm(g(hid), g(wd), torch.empty(M, K, device=DEV), M=M, N=K, K=3 * K, epi=nat.EPI_RESIDUAL, residual=g(res))
        inplace = g(res).clone()
        ops.gemm(g(hid), g(wd), inplace, M=M, N=K, K=3 * K, epi=nat.EPI_RESIDUAL, residual=inplace)
        out["inplace"] = inplace
        return out

    nat.set_option("x3s_max_wgs", 1 << 20)        # (by default the kernel takes a shape where its cost estimate wins: here every shape)
    request.addfinalizer(lambda: nat.set_option("x3s_max_wgs", -2 ** 31))
    nat.lib().kd_prof_reset()
    nat.lib().kd_prof_enable(1)
    try:
        got = run_all()
        torch.cuda.synchronize()
        names = _prof_names(nat)
    finally:
        nat.lib().kd_prof_enable(0)
        nat.lib().kd_prof_reset()
    assert sum(n.startswith("gemm_x3s") for n in names) == 7, names           # every one of them ran on the few-rows kernel
    qkv = got["qkv"].cpu().view(B, H, W, 3, nh, 64)
    ref = (xn @ w.T).view(B, H, W, 3, nh, 64)
    q_ref, k_ref = hdit.cosine_sim_scale(ref[..., 0, :, :], ref[..., 1, :, :], qs)
    assert relerr(qkv[..., 0, :, :], hdit.apply_rope(q_ref, theta)) < 1e-4
    assert relerr(qkv[..., 1, :, :], hdit.apply_rope(k_ref, theta)) < 1e-4
    assert relerr(qkv[..., 2, :, :], ref[..., 2, :, :]) < 1e-4
    pw = got["packed"].view(torch.int32).view(-1, 4)
    hi = torch.stack([(pw[:, 0] << 16), (pw[:, 0] & -65536), (pw[:, 1] << 16), (pw[:, 1] & -65536)], dim=1).view(torch.float32)
    lo = torch.stack([(pw[:, 2] << 16), (pw[:, 2] & -65536), (pw[:, 3] << 16), (pw[:, 3] & -65536)], dim=1).view(torch.float32)
    assert relerr((hi + lo).cpu().view_as(qkv), qkv) < 2.0 ** -15
    assert relerr(got["geglu"], hdit.linear_geglu(xn, wg)) < 1e-4
    assert relerr(got["shared"], hdit.rms_norm(x, scale[0]) @ ws.T) < 1e-4
    assert relerr(got["plain"], hid.view(M, -1) @ wd.T + 0.5) < 1e-4
    assert relerr(got["res"], hid.view(M, -1) @ wd.T + res.view(M, -1)) < 1e-4
    assert torch.equal(got["inplace"].view(M, K), got["res"])
    # GEGLU result as bf16 hi / lo planes (the pre-split operand of gemm_x3t.hip)
    hh, hl = torch.empty(B, T, 3 * K, device=DEV, dtype=torch.bfloat16), torch.empty(B, T, 3 * K, device=DEV, dtype=torch.bfloat16)
    ops.gemm(g(x), g(wg), None, M=M, N=3 * K, K=K, epi=nat.EPI_GEGLU, norm_scale=g(scale), scale_stride=K, rows_per_sample=T, c_planes=(hh, hl))
    assert relerr(hh.float().cpu() + hl.float().cpu(), hdit.linear_geglu(xn, wg)) < 1e-4
    # run to run: the reduction over the 8 waves is in wave order
    again = run_all()
    assert all(torch.equal(got[k], again[k]) for k in got)
    # the scale vector through LDS (option x3s_scale_lds; taken where a workgroup's 32 rows share it): the same products in the same order
    nat.set_option("x3s_scale_lds", 1)
    try:
        lds = run_all()
    finally:
        nat.set_option("x3s_scale_lds", -2 ** 31)
    assert all(torch.equal(got[k], lds[k]) for k in got)
    # against the throughput kernels (another summation order: close, not equal)
    nat.set_option("x3s_max_rows", 0)
    try:
        old = run_all()
    finally:
        nat.set_option("x3s_max_rows", -2 ** 31)
    for k in got:
        a, b = (got[k], old[k]) if k != "packed" else (got["qkv"], old["qkv"])
        assert relerr(a, b) < 1e-4, k
    assert not torch.equal(got["qkv"], old["qkv"])


@pytest.mark.few_rows
@pytest.mark.parametrize("H,W,nh,B,K", [(16, 16, 8, 1, 512), (7, 7, 4, 4, 256), (32, 32, 4, 1, 256), (9, 12, 2, 3, 128), (6, 6, 1, 4, 64), (16, 16, 18, 2, 384)])
def test_bf16_few_rows_latency_kernel(KD, ops, request, H, W, nh, B, K):
    """The few-rows form of the bf16 projections (round 4, csrc/gemm_b16s.hip: the bf16 sibling of gemm_x3s.hip) at batch-1 shapes of the
    headline config, 49 tokens per sample (a 32-row group spans samples), ragged row counts, 18 heads, K = 64: every epilogue against the
    oracle on bf16-rounded operands at the bf16 kernels' tolerances, against the throughput kernels (same roundings, another summation
    order: within a bf16 step), and bit-identical from run to run."""
    from k_diffusion_amd import _native as nat
    T, d, M = H * W, nh * 64, B * H * W
    x, scale = rn(B, T, K, seed=8), 1 + 0.2 * rn(B, K, seed=9)
    xn = hdit.rms_norm(_rt(x), scale[:, None, :])
    w = rn(3 * d, K, seed=10, scale=K ** -0.5)
    qs = torch.linspace(5.0, 12.0, nh)
    pos, freqs = hdit.axial_pos(H, W).reshape(T, 2), hdit.rope_freqs(nh)
    qk = (g(qs), g(pos.contiguous()), g((freqs / (2 * np.pi)).contiguous()), nh)
    theta = hdit.rope_theta(hdit.axial_pos(H, W), freqs)
    wg = rn(2 * 3 * K, K, seed=11, scale=K ** -0.5)
    ws = rn(256, K, seed=12, scale=K ** -0.5)
    wd = rn(K, 3 * K, seed=13, scale=(3 * K) ** -0.5)
    hid, res = rn(B, T, 3 * K, seed=15), rn(B, T, K, seed=14)
    PB = nat.PREC_BF16

    def run_all():
        out = {}
        out["qkv"] = ops.norm_linear(_bf(x), g(scale), g(w), rows_per_sample=T, epi=nat.EPI_QKV, qk=qk)
        out["geglu"] = ops.norm_linear(_bf(x), g(scale), g(wg), rows_per_sample=T, epi=nat.EPI_GEGLU)
        out["shared"] = ops.norm_linear(_bf(x), g(scale[0].contiguous()), g(ws), rows_per_sample=T)
        out["plain"] = ops.gemm(_bf(hid), g(wd), torch.empty(M, K, device=DEV, dtype=BF), M=M, N=K, K=3 * K, precision=PB)
        out["res"] = ops.gemm(_bf(hid), g(wd), torch.empty(M, K, device=DEV, dtype=BF), M=M, N=K, K=3 * K, epi=nat.EPI_RESIDUAL, residual=_bf(res), precision=PB)
        inplace = _bf(res).clone()
        ops.gemm(_bf(hid), g(wd), inplace, M=M, N=K, K=3 * K, epi=nat.EPI_RESIDUAL, residual=inplace, precision=PB)
        out["inplace"] = inplace
        return out

    nat.set_option("b16s_max_wgs", 1 << 20)        # (by default the kernel takes grids of one round, two behind a norm at few rows: here every shape)
    request.addfinalizer(lambda: nat.set_option("b16s_max_wgs", -2 ** 31))
    nat.lib().kd_prof_reset()
    nat.lib().kd_prof_enable(1)
    try:
        got = run_all()
        torch.cuda.synchronize()
        names = _prof_names(nat)
    finally:
        nat.lib().kd_prof_enable(0)
        nat.lib().kd_prof_reset()
    assert sum(n.startswith("gemm_bf16_few_rows") for n in names) == 6, names
    qkv = got["qkv"].float().cpu().view(B, H, W, 3, nh, 64)
    ref = (xn @ _rt(w).T).view(B, H, W, 3, nh, 64)
    q_ref, k_ref = hdit.cosine_sim_scale(ref[..., 0, :, :], ref[..., 1, :, :], qs)
    assert relerr(qkv[..., 0, :, :], hdit.apply_rope(q_ref, theta)) < 1.2e-2
    assert relerr(qkv[..., 1, :, :], hdit.apply_rope(k_ref, theta)) < 1.2e-2
    assert relerr(qkv[..., 2, :, :], ref[..., 2, :, :]) < 1.2e-2
    assert relerr(got["geglu"], hdit.linear_geglu(xn, _rt(wg))) < 1.5e-2
    assert relerr(got["shared"], hdit.rms_norm(_rt(x), scale[0]) @ _rt(ws).T) < 1.2e-2
    assert relerr(got["plain"], _rt(hid).view(M, -1) @ _rt(wd).T) < 1.2e-2
    assert relerr(got["res"], _rt(hid).view(M, -1) @ _rt(wd).T + _rt(res).view(M, -1)) < 1.2e-2
    assert torch.equal(got["inplace"].view(M, K), got["res"])
    again = run_all()
    assert all(torch.equal(got[k], again[k]) for k in got)
    nat.set_option("b16s_max_rows", 0)
    try:
        old = run_all()
    finally:
        nat.set_option("b16s_max_rows", -2 ** 31)
    for k in got:
        assert relerr(got[k], old[k]) < 1e-2, k


@pytest.mark.parametrize("B,H,W,C,N", [(2, 32, 32, 128, 256), (3, 16, 24, 256, 512), (1, 48, 40, 64, 128)])
def test_split3_token_merge_round3(KD, ops, monkeypatch, B, H, W, C, N):
    """TokenMerge (image_transformer_v2.py:586-595) on gemm_x3r.hip (the 2 x 2 gather as address arithmetic of its staging requests)
    against the oracle and the round-1 kernel; a ragged last row panel (1 x 24 x 20 coarse tokens)."""
    from k_diffusion_amd import _native as nat
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    x, w = rn(B, H, W, C, seed=51), rn(N, 4 * C, seed=52, scale=(4 * C) ** -0.5)
    ref = hdit.token_merge(x, w, 2, 2)
    nat.set_option("x3r", 2)
    try:
        y = ops.token_merge(g(x), g(w))
    finally:
        nat.set_option("x3r", 1)
    assert relerr(y, ref) < 1e-4
    nat.set_option("x3r", 0)
    try:
        old = ops.token_merge(g(x), g(w))
    finally:
        nat.set_option("x3r", 1)
    assert relerr(y, old) < 1e-4
    nat.set_option("x3r", 2)
    nat.set_option("x3r_lw", 0)               # the form without loader waves: the same products in the same order
    try:
        assert torch.equal(ops.token_merge(g(x), g(w)), y)
    finally:
        nat.set_option("x3r", 1)
        nat.set_option("x3r_lw", 1)


@pytest.mark.parametrize("M,K,N", [(1024, 512, 512), (1000, 512, 512), (8192, 512, 512), (2048, 256, 256), (640, 512, 256), (4096, 1536, 512), (520, 768, 256)])
def test_split3_residual_projection_round3(KD, ops, monkeypatch, M, K, N):
    """out = residual + A W^T (the projection behind the attention core) on the round-3 kernels (gemm_x3r.hip where the tiles fit one
    round of the chip; the accumulators start from the residual); full and ragged row panels, in place (out is the residual) and not,
    and against the A-stationary form (x3_res), the round-1 kernel (x3r = 0) and gemm_x3r forced (x3r = 2)."""
    from k_diffusion_amd import _native as nat
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    a, res = rn(M, K, seed=21), rn(M, N, seed=22)
    w = rn(N, K, seed=23, scale=K ** -0.5)
    ref = res + a @ w.T
    out = torch.empty(M, N, device="cuda")
    ops.gemm(g(a), g(w), out, M=M, N=N, K=K, epi=nat.EPI_RESIDUAL, residual=g(res))
    assert relerr(out, ref) < 1e-4
    inplace = g(res).clone()
    ops.gemm(g(a), g(w), inplace, M=M, N=N, K=K, epi=nat.EPI_RESIDUAL, residual=inplace)
    assert torch.equal(inplace, out)
    # the same projection on the A-stationary kernel (option x3_res, K = 512) and on the round-1 tile kernel (x3r = 0)
    # ... and with the staging requests inside the compute waves' K loop instead of on loader waves (round 4: x3r_lw = 0)
    for name, val, back in (("x3_res", 1, 0), ("x3r", 0, 1), ("x3r", 2, 1), ("x3r_lw", 0, 1)):
        nat.set_option(name, val)
        try:
            old = torch.empty_like(out)
            ops.gemm(g(a), g(w), old, M=M, N=N, K=K, epi=nat.EPI_RESIDUAL, residual=g(res))
        finally:
            nat.set_option(name, back)
        assert relerr(old, ref) < 1e-4 and relerr(out, old) < 1e-4, (name, val)


@pytest.mark.parametrize("H,W,B,K,dff", [(64, 64, 2, 128, 384), (32, 32, 4, 256, 768), (48, 40, 2, 128, 320), (30, 30, 3, 256, 448)])
def test_fused_feed_forward_split3(KD, ops, monkeypatch, H, W, B, K, dff):
    """kd_ffn_f32 (csrc/ffn_x3.hip): the whole FeedForwardBlock (image_transformer_v2.py:487-493) in fp32-parity arithmetic against the
    oracle's separate steps, and against the two-GEMM form it replaces.  Full and ragged 128-row panels, rows of two samples in one wave
    (30 x 30 tokens), d_ff that is not a multiple of 128."""
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    T = H * W
    x, scale = rn(B, T, K, seed=21), 1 + 0.2 * rn(B, K, seed=22)
    wu, wd = rn(2 * dff, K, seed=23, scale=K ** -0.5), rn(K, dff, seed=24, scale=dff ** -0.5)
    # kd_ffn_f32_supported says where the fused form is the FASTER one: width 128 from 16 row panels on, width 256 (one workgroup per CU, ~120 us whatever
    # the grid) only from a chip-filling grid on; the kernel itself takes every shape below
    assert ops.ffn_supported(B * T, K, dff, bf16=False) == (K == 128)
    assert ops.ffn_supported(32 * 1024, 256, 768, bf16=False) and not ops.ffn_supported(16 * 1024, 256, 768, bf16=False)
    y = ops.ffn(g(x), g(scale), g(wu), g(wd), rows_per_sample=T)
    ref = x + hdit.linear_geglu(hdit.rms_norm(x, scale[:, None, :]), wu) @ wd.T
    assert relerr(y, ref) < 1e-4
    hid = ops.norm_linear(g(x), g(scale), g(wu), rows_per_sample=T, epi=KD._native.EPI_GEGLU)
    two = ops.linear(hid, g(wd), residual=g(x))
    assert relerr(y, two) < 1e-4
    xi = g(x).clone()                                   # in place, as the model runs it
    ops.ffn(xi, g(scale), g(wu), g(wd), out=xi, rows_per_sample=T)
    assert torch.equal(xi, y)
    with pytest.raises(RuntimeError, match="kd_ffn_f32"):
        ops.ffn(g(rn(4, 96, seed=1)), g(rn(1, 96, seed=2)), g(rn(2 * 64, 96, seed=3)), g(rn(96, 64, seed=4)), rows_per_sample=4)


@pytest.mark.parametrize("H,W,B,K,dff", [(64, 64, 2, 128, 384), (48, 40, 2, 128, 320), (30, 30, 3, 128, 192), (32, 32, 4, 256, 768), (30, 30, 3, 256, 448)])
def test_fused_out_projection_and_feed_forward_split3(KD, ops, monkeypatch, H, W, B, K, dff):
    """kd_ffn_f32 with the attention block's out projection fused in front (KdFfn.attn / Wp_out): x' = x + attn W_out^T (:473-476), then
    x' + ff(x') (:487-493), against the oracle's separate steps and against out projection + fused block as two calls."""
    monkeypatch.setenv("KDIFF_GEMM", "split3")
    T = H * W
    x, att, scale = rn(B, T, K, seed=31), rn(B, T, K, seed=32), 1 + 0.2 * rn(B, K, seed=33)
    wo = rn(K, K, seed=34, scale=K ** -0.5)
    wu, wd = rn(2 * dff, K, seed=35, scale=K ** -0.5), rn(K, dff, seed=36, scale=dff ** -0.5)
    y = ops.ffn(g(x), g(scale), g(wu), g(wd), rows_per_sample=T, attn=g(att), w_out=g(wo))
    x1 = x + att @ wo.T
    ref = x1 + hdit.linear_geglu(hdit.rms_norm(x1, scale[:, None, :]), wu) @ wd.T
    assert relerr(y, ref) < 1e-4
    x1g = ops.linear(g(att), g(wo), residual=g(x))
    two = ops.ffn(x1g, g(scale), g(wu), g(wd), rows_per_sample=T)
    assert relerr(y, two) < 1e-4
    xi = g(x).clone()                                   # in place, as the model runs it
    ops.ffn(xi, g(scale), g(wu), g(wd), out=xi, rows_per_sample=T, attn=g(att), w_out=g(wo))
    assert torch.equal(xi, y)


@pytest.mark.parametrize("H,W,B,dff", [(64, 64, 4, 384), (48, 40, 9, 320), (30, 30, 19, 192)])
def test_fused_out_projection_and_feed_forward_bf16(KD, ops, monkeypatch, H, W, B, dff):
    """kd_ffn_bf16 with the attention block's out projection fused in front (KdFfn.attn / Wp_out, width 128) against the oracle's
    separate steps on the bf16-rounded operands and against out projection + fused block as two calls."""
    monkeypatch.setenv("KDIFF_GEMM", "bf16")
    K, T = 128, H * W
    x, att, scale = rn(B, T, K, seed=41), rn(B, T, K, seed=42), 1 + 0.2 * rn(B, K, seed=43)
    wo = rn(K, K, seed=44, scale=K ** -0.5)
    wu, wd = rn(2 * dff, K, seed=45, scale=K ** -0.5), rn(K, dff, seed=46, scale=dff ** -0.5)
    assert ops.ffn_supported(B * T, K, dff)
    y = ops.ffn(_bf(x), g(scale), g(wu), g(wd), rows_per_sample=T, attn=_bf(att), w_out=g(wo)).float().cpu()
    x1 = _rt(x) + _rt(att) @ _rt(wo).T
    ref = x1 + hdit.linear_geglu(hdit.rms_norm(x1, scale[:, None, :]), _rt(wu)) @ _rt(wd).T
    assert relerr(y, ref) < 1.2e-2
    x1g = ops.linear(_bf(att), g(wo), residual=_bf(x))
    two = ops.ffn(x1g, g(scale), g(wu), g(wd), rows_per_sample=T).float().cpu()
    assert relerr(y, two) < 1.2e-2
    xi = _bf(x).clone()                                 # in place, as the model runs it
    ops.ffn(xi, g(scale), g(wu), g(wd), out=xi, rows_per_sample=T, attn=_bf(att), w_out=g(wo))
    assert torch.equal(xi.float().cpu(), y)


def test_bf16_attention_cores(ops, golden):
    """bf16 global / window / neighbourhood cores against the reference's op goldens (prepared q, k) and the oracle."""
    o = golden["ops"]
    qkv = _bf(_pack(o["qk.q_out"], o["qk.k_out"], o["qk.v"]))
    y = ops.attn_global(qkv.view(2, 256, -1), 2).float().cpu().view(2, 16, 16, 2, 64)
    assert relerr(y, o["attn_global.o"]) < 1.2e-2
    for shift in (0, 4):
        y = ops.attn_window(qkv, 2, 8, shift).float().cpu().view(2, 16, 16, 2, 64)
        assert relerr(y, o[f"attn_window{shift}.o"]) < 1.2e-2
    for tag, ws, shift in (("w4s0", 4, 0), ("w4s2", 4, 2), ("w16s8", 16, 8), ("w16s0", 16, 0)):
        q, k, v = o[f"attn_{tag}.q"], o[f"attn_{tag}.k"], o[f"attn_{tag}.v"]
        y = ops.attn_window(_bf(_pack(q, k, v)), q.shape[3], ws, shift).float().cpu().view(q.shape)
        assert relerr(y, o[f"attn_{tag}.o"]) < 1.2e-2, tag
    for ks, (B, H, W, nh) in ((3, (2, 9, 12, 2)), (5, (2, 20, 13, 1)), (7, (1, 32, 32, 4)), (7, (2, 7, 7, 1)), (9, (2, 20, 33, 2)), (9, (1, 9, 9, 1)),
                              (11, (2, 24, 40, 2)), (11, (1, 11, 13, 1)), (13, (1, 32, 32, 2)), (13, (2, 13, 21, 1)), (13, (1, 45, 19, 1))):
        q, k, v = (rn(B, H, W, nh, 64, seed=s, scale=sc) for s, sc in ((1, 0.5), (2, 0.5), (3, 1.0)))
        y = ops.attn_na2d(_bf(_pack(q, k, v)), nh, ks).float().cpu().view(B, H, W, nh, 64)
        assert relerr(y, hdit.na2d(_rt(q), _rt(k), _rt(v), ks, 1.0)) < 1.2e-2, (ks, H, W)
    for T, nh, B in ((49, 4, 2), (100, 2, 3), (960, 2, 1), (1024, 1, 2)):                # whole-in-LDS and streamed forms
        q, k, v = (rn(B, T, 1, nh, 64, seed=s, scale=sc) for s, sc in ((1, 0.4), (2, 0.4), (3, 1.0)))
        y = ops.attn_global(_bf(_pack(q, k, v)).view(B, T, -1), nh).float().cpu().view(B, T, nh, 64)
        ref = hdit.attn_global(_rt(q), _rt(k), _rt(v)).view(B, T, nh, 64)
        assert relerr(y, ref) < 1.2e-2, T
    with pytest.raises(ValueError, match="prepared"):
        ops.attn_global(qkv.view(2, 256, -1), 2, prep=(g(o["qk.scale"]), None, None))
    with pytest.raises(RuntimeError, match="kernel_size"):
        ops.attn_na2d(qkv, 2, 15)
    with pytest.raises(RuntimeError, match="kernel_size"):
        ops.attn_na2d(qkv, 2, 6)


def test_bf16_merge_split_patch(ops, golden):
    from k_diffusion_amd import _native as nat
    o = golden["ops"]
    x = o["rms_norm.x"]                                                      # [2, 8, 8, 128]
    assert relerr(ops.token_merge(_bf(x), g(o["merge.w"])), hdit.token_merge(_rt(x), _rt(o["merge.w"]), 2, 2)) < 1e-2
    y = ops.token_split_lerp(_bf(x), g(o["split.w"]), _bf(o["split.skip"]), g(torch.tensor([0.37])))
    ref = torch.lerp(_rt(o["split.skip"]), hdit.token_split(_rt(x), _rt(o["split.w"]), 2, 2), 0.37)
    assert relerr(y, ref) < 1e-2
    # big enough for the tiled kernel (merge gather / split scatter through global_load_lds addressing)
    xb, wm = rn(2, 32, 32, 128, seed=11), rn(256, 512, seed=12, scale=512 ** -0.5)
    assert relerr(ops.token_merge(_bf(xb), g(wm)), hdit.token_merge(_rt(xb), _rt(wm), 2, 2)) < 1e-2
    xs, ws_, skip = rn(2, 16, 16, 256, seed=13), rn(512, 256, seed=14, scale=1 / 16), rn(2, 32, 32, 128, seed=15)
    y = ops.token_split_lerp(_bf(xs), g(ws_), _bf(skip), g(torch.tensor([0.37])))
    assert relerr(y, torch.lerp(_rt(skip), hdit.token_split(_rt(xs), _rt(ws_), 2, 2), 0.37)) < 1e-2
    # patch in (fp32 image -> bf16 tokens) / patch out (bf16 tokens -> fp32 image) with the Karras scalings
    img, sigma = rn(2, 3, 32, 32, seed=16, scale=3.0), torch.tensor([0.3, 9.0])
    w_in, w_out, gain = rn(128, 48, seed=17, scale=48 ** -0.5), rn(48, 128, seed=18, scale=128 ** -0.5), 1 + 0.1 * rn(128, seed=19)
    c_skip, c_out, c_in = solvers.karras_scalings(sigma, 0.5)
    t = ops.patch_in(g(img), g(w_in), (4, 4), sigma=g(sigma), sigma_data=0.5, precision=nat.PREC_BF16)
    assert t.dtype == BF
    ref_t = hdit.token_merge((img * c_in.view(-1, 1, 1, 1)).movedim(1, -1).contiguous(), w_in, 4, 4)
    assert relerr(t, ref_t) < 1.5e-2
    tok = rn(2, 8, 8, 128, seed=20)
    out = ops.patch_out(_bf(tok), g(gain), g(w_out), (4, 4), 3, x_in=g(img), sigma=g(sigma), sigma_data=0.5)
    assert out.dtype == torch.float32
    f = hdit.token_split(hdit.rms_norm(_rt(tok), gain), _rt(w_out), 4, 4).movedim(-1, 1)
    assert relerr(out, f * c_out.view(-1, 1, 1, 1) + img * c_skip.view(-1, 1, 1, 1)) < 5e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_third_party_operator_signatures_on_the_hip_cores(KD, gtol, dtype):
    """compat.na2d / flash_attn_qkvpacked_func / scaled_dot_product_attention: the operators the reference's model calls
    (image_transformer_v2.py:428, :383, :392), with their own signatures, layouts and default scales, against the oracle's restatement
    (fp32 on the CPU) -- at the reference's scale = 1.0 and at the libraries' defaults."""
    C = KD.compat
    tol = gtol if dtype == torch.float32 else 2e-2
    g = torch.Generator().manual_seed(4)
    n, h, w, nh, e = 2, 12, 20, 3, 64
    q, k, v = (torch.randn(n, h, w, nh, e, generator=g) * 0.7 for _ in range(3))
    dq, dk, dv = (t.to(DEV, dtype) for t in (q, k, v))
    # natten.functional.na2d: [n, h, w, nh, e], default scale e ** -0.5
    import os
    exact = dtype == torch.float32 and os.environ["KDIFF_GEMM"] == "exact"
    if exact:
        with pytest.raises(NotImplementedError):                      # the exact-fp32 neighbourhood core: kernel size 7 only
            C.na2d(dq, dk, dv, 5)
    for ks, scale in ((7, 1.0), (7, None)) if exact else ((7, 1.0), (5, None), ((3, 3), 0.3), (11, 1.0)):
        want = hdit.na2d(q, k, v, ks if isinstance(ks, int) else ks[0], e ** -0.5 if scale is None else scale)
        got = C.na2d(dq, dk, dv, ks) if scale is None else C.na2d(dq, dk, dv, kernel_size=ks, scale=scale)
        assert got.shape == want.shape and got.dtype == dtype and relerr(got.float(), want) < tol, (ks, scale, relerr(got.float(), want))
    # flash_attn_qkvpacked_func: [n, s, 3, nh, e] -> [n, s, nh, e]
    s = h * w
    packed = torch.stack([q.reshape(n, s, nh, e), k.reshape(n, s, nh, e), v.reshape(n, s, nh, e)], dim=2)
    for scale in (1.0, None):
        want = hdit.attn_global(q, k, v, e ** -0.5 if scale is None else scale).reshape(n, s, nh, e)
        got = C.flash_attn_qkvpacked_func(packed.to(DEV, dtype), softmax_scale=scale)
        assert got.shape == (n, s, nh, e) and relerr(got.float(), want) < tol, scale
    # F.scaled_dot_product_attention: [n, nh, s, e]
    hq, hk, hv = (t.reshape(n, s, nh, e).permute(0, 2, 1, 3) for t in (q, k, v))
    for scale in (1.0, None):
        want = hdit.attn_global(q, k, v, e ** -0.5 if scale is None else scale).reshape(n, s, nh, e).permute(0, 2, 1, 3)
        got = C.scaled_dot_product_attention(hq.to(DEV, dtype), hk.to(DEV, dtype), hv.to(DEV, dtype), scale=scale)
        assert got.shape == (n, nh, s, e) and relerr(got.float(), want) < tol, scale
        if dtype == torch.float32:      # and against torch's own operator on the device
            ref = torch.nn.functional.scaled_dot_product_attention(hq.to(DEV), hk.to(DEV), hv.to(DEV), scale=scale)
            assert relerr(got, ref) < 1e-4
    # what the cores do not do is an error, not a silent approximation
    with pytest.raises(NotImplementedError):
        C.na2d(dq, dk, dv, 7, dilation=2)
    with pytest.raises(NotImplementedError):
        C.scaled_dot_product_attention(hq.to(DEV), hk.to(DEV), hv.to(DEV), attn_mask=torch.ones(s, s, dtype=torch.bool, device=DEV))
    with pytest.raises(NotImplementedError):
        C.flash_attn_qkvpacked_func(packed.to(DEV), causal=True)
    with pytest.raises(NotImplementedError):
        C.na2d(dq[..., :32], dk[..., :32], dv[..., :32], 7)
    with pytest.raises(RuntimeError):
        C.na2d(q, k, v, 7)                                            # CPU tensors: no fallback
