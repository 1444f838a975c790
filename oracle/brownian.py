"""CPU oracle (test infrastructure): the counter-based virtual Brownian tree behind the HIP
``BrownianTreeNoiseSampler``.

**Parity unpinned** with respect to the reference: k_diffusion/sampling.py:65-114 delegates to
``torchsde.BrownianTree`` (third-party, un-pinned in requirements.txt:14, absent here and not
installable), whose random stream is generator- and device-dependent.  What the reference's call
sites rely on, and what the tests check, is (a) unit-variance normalised increments, (b) path
consistency over nested/adjacent intervals (sample_dpmpp_sde queries (sigma_i, sigma_mid) then
(sigma_i, sigma_next), sampling.py:572,580), (c) determinism per seed, (d) sign/sort handling
(sampling.py:82-89).  This file restates, in numpy, the algorithm the HIP kernel implements
(k-diffusion_amd/csrc/brownian.hip): integer Philox4x32-10 (bit-exact; one block per two tree
levels) + float32 Box-Muller and bridge arithmetic.  The device evaluates log2 / sqrt / cos with the
hardware v_log_f32 / v_sqrt_f32 / v_cos_f32 (about 1e-6 absolute), so the comparison tolerance is
1e-4 absolute on values of order sqrt(T1 - T0).
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(key, c0, c1, c2, c3):
    """key: python int (64 bit); c*: uint32 arrays.  Returns the four output words."""
    k0, k1 = key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & _MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & _MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _radius(w):
    """sqrt(-2 ln u), u = (w >> 8 + 1) / 2^24 in (0, 1]; float32 with the kernel's op order (log2 * -2 ln 2)."""
    u = ((w >> np.uint32(8)).astype(np.float32) + np.float32(1)) * np.float32(2.0 ** -24)
    return np.sqrt(np.float32(-1.3862943611198906) * np.log2(u)).astype(np.float32)


def _unit24(w):
    return (w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def _cos_rev(rev):
    """cos of `rev` revolutions (the device uses v_cos_f32, whose argument is in revolutions)."""
    return np.cos(np.float64(2.0 * np.pi) * rev.astype(np.float64)).astype(np.float32)


def _philox(key, elem, node):
    elem = np.asarray(elem, dtype=np.uint64)
    node = np.broadcast_to(np.asarray(node, dtype=np.uint64), elem.shape)
    return philox4x32_10(key, elem & _MASK, elem >> np.uint64(32), node & _MASK, node >> np.uint64(32))


def brownian_w(key, n_elem, t, T0, T1, depth=36):
    """W(t) for elements 0..n_elem-1 of the tree keyed by ``key``.

    One Philox block serves two levels: the even-level node (heap index ``node``) takes
    r(x0)*cos(x1); its left child r(x0)*sin(x1) (= cos(x1 - 1/4 turn)), its right child r(x2)*cos(x3)."""
    elem = np.arange(n_elem, dtype=np.uint64)
    ta, tb = float(T0), float(T1)
    sd0 = np.sqrt(np.float32(T1 - T0))
    x0, x1, _, _ = _philox(key, elem, 0)
    wa = np.zeros(n_elem, dtype=np.float32)
    wb = (sd0 * _radius(x0) * _cos_rev(_unit24(x1))).astype(np.float32)
    hs = np.float32(0.5) * sd0
    r2 = np.float32(0.70710678118654752)
    node = 1
    for lv in range(0, depth, 2):
        x0, x1, x2, x3 = _philox(key, elem, node)
        tm = 0.5 * (ta + tb)
        wm = np.float32(0.5) * (wa + wb) + hs * (_radius(x0) * _cos_rev(_unit24(x1)))
        hs = hs * r2
        right0 = not (t < tm)
        if right0:
            ta, wa = tm, wm
        else:
            tb, wb = tm, wm
        if lv + 1 >= depth:
            break
        rad = _radius(x2 if right0 else x0)
        rev = _unit24(x3) if right0 else _unit24(x1) - np.float32(0.25)
        tm = 0.5 * (ta + tb)
        wm = np.float32(0.5) * (wa + wb) + hs * (rad * _cos_rev(rev))
        hs = hs * r2
        right1 = not (t < tm)
        if right1:
            ta, wa = tm, wm
        else:
            tb, wb = tm, wm
        node = 4 * node + (2 if right0 else 0) + (1 if right1 else 0)
    frac = np.float32((t - ta) / (tb - ta))
    return (wa + frac * (wb - wa)).astype(np.float32)


def brownian_increment(seeds, per_sample, T0, T1, t0, t1, mult=1.0, depth=36):
    """[len(seeds), per_sample] array of (W(t1) - W(t0)) * mult."""
    out = np.empty((len(seeds), per_sample), dtype=np.float32)
    for b, s in enumerate(seeds):
        out[b] = (brownian_w(int(s), per_sample, t1, T0, T1, depth) - brownian_w(int(s), per_sample, t0, T0, T1, depth)) * np.float32(mult)
    return out


def randn_indexed(seeds, per_sample, draw=0, scale=1.0):
    """[len(seeds), per_sample] float32: the index-addressed normals of ``kd_randn_f32`` (k-diffusion_amd/csrc/brownian.hip).  One
    Philox block (key seeds[b], counter (e >> 2, draw | 2^63)) per four consecutive elements, two Box-Muller pairs per block:
    r(x0) cos(x1), r(x0) sin(x1), r(x2) cos(x3), r(x2) sin(x3) with sin(a) evaluated as cos(a - 1/4 turn), each radius scaled first.
    Stands where the reference draws ``torch.randn(...) * sigma_max`` from rank-local generator state (sample.py:59): there is no
    reference stream to match, so -- like the tree above -- what is checked is this restatement, the N(0, 1) statistics and the
    independence of batch / rank."""
    quads = (per_sample + 3) // 4
    q = np.arange(quads, dtype=np.uint64)
    node = int(draw) | (1 << 63)
    out = np.empty((len(seeds), quads * 4), dtype=np.float32)
    s = np.float32(scale)
    for b, key in enumerate(seeds):
        x0, x1, x2, x3 = _philox(int(key) & 0xFFFFFFFFFFFFFFFF, q, node)
        r1, r2 = s * _radius(x0), s * _radius(x2)
        a1, a2 = _unit24(x1), _unit24(x3)
        z = np.stack([r1 * _cos_rev(a1), r1 * _cos_rev(a1 - np.float32(0.25)), r2 * _cos_rev(a2), r2 * _cos_rev(a2 - np.float32(0.25))], axis=1)
        out[b] = z.reshape(-1).astype(np.float32)
    return out[:, :per_sample]


class OracleBrownianTree:
    """Stands in for ``torchsde.BrownianTree(t0, w0, t1, entropy=seed)`` (the constructor and call surface
    k_diffusion/sampling.py:80,88 uses) with THIS package's stream: ``tree(ta, tb)`` -> W(tb) - W(ta) shaped like ``w0``,
    element i of the flattened sample = element i of the virtual tree keyed by ``entropy``.  With it installed as
    ``torchsde.BrownianTree`` the reference's own BatchedBrownianTree / BrownianTreeNoiseSampler (sampling.py:65-114: sorting,
    signs, one tree per batch item, the 1 / sqrt|t1 - t0| normalisation) run unchanged on the CPU -- which is how
    oracle/make_golden_r3.py records BASELINE configs[4] (flowers-NA, sample_dpmpp_sde, Brownian noise) from the reference."""

    def __init__(self, t0, w0, t1, entropy=None, depth=36, **kwargs):
        import torch
        self.T0, self.T1, self.key, self.depth = float(t0), float(t1), int(entropy) & 0x7FFFFFFFFFFFFFFF, int(depth)
        self.shape, self._torch = tuple(w0.shape), torch
        self._points = {}

    def _w(self, t):
        t = min(max(float(t), self.T0), self.T1)
        if t not in self._points:
            if len(self._points) > 4:
                self._points.pop(next(iter(self._points)))
            n = int(np.prod(self.shape))
            self._points[t] = brownian_w(self.key, n, t, self.T0, self.T1, self.depth)
        return self._points[t]

    def __call__(self, ta, tb):
        return self._torch.from_numpy((self._w(tb) - self._w(ta)).reshape(self.shape).copy())
