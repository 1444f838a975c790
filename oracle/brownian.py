"""CPU oracle (test infrastructure): the counter-based virtual Brownian tree behind the HIP
``BrownianTreeNoiseSampler``.

**Parity unpinned** with respect to the reference: k_diffusion/sampling.py:65-114 delegates to
``torchsde.BrownianTree`` (third-party, un-pinned in requirements.txt:14, absent here and not
installable), whose random stream is generator- and device-dependent.  What the reference's call
sites rely on, and what the tests check, is (a) unit-variance normalised increments, (b) path
consistency over nested/adjacent intervals (sample_dpmpp_sde queries (sigma_i, sigma_mid) then
(sigma_i, sigma_next), sampling.py:572,580), (c) determinism per seed, (d) sign/sort handling
(sampling.py:82-89).  This file restates, in numpy, the algorithm the HIP kernel implements
(k-diffusion_amd/csrc/brownian.hip): integer Philox4x32-10 (bit-exact) + float32 Box-Muller and
bridge arithmetic (tolerance 2e-5 absolute against the device's libm).
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


def philox_normal(key, elem, node):
    """Standard normal for (key, element index array, node index array)."""
    elem = np.asarray(elem, dtype=np.uint64)
    node = np.broadcast_to(np.asarray(node, dtype=np.uint64), elem.shape)
    x0, x1, _, _ = philox4x32_10(key, elem & _MASK, elem >> np.uint64(32), node & _MASK, node >> np.uint64(32))
    u1 = ((x0 >> np.uint32(8)).astype(np.float32) + np.float32(1)) * np.float32(2.0 ** -24)
    u2 = (x1 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)).astype(np.float32)


def brownian_w(key, n_elem, t, T0, T1, depth=36):
    """W(t) for elements 0..n_elem-1 of the tree keyed by ``key``."""
    elem = np.arange(n_elem, dtype=np.uint64)
    ta, tb = float(T0), float(T1)
    wa = np.zeros(n_elem, dtype=np.float32)
    wb = np.float32(np.sqrt(np.float32(T1 - T0))) * philox_normal(key, elem, 0)
    node = 1
    for _ in range(depth):
        tm = 0.5 * (ta + tb)
        wm = np.float32(0.5) * (wa + wb) + np.float32(0.5) * np.sqrt(np.float32(tb - ta)) * philox_normal(key, elem, node)
        if t < tm:
            tb, wb, node = tm, wm, 2 * node
        else:
            ta, wa, node = tm, wm, 2 * node + 1
    frac = np.float32((t - ta) / (tb - ta))
    return (wa + frac * (wb - wa)).astype(np.float32)


def brownian_increment(seeds, per_sample, T0, T1, t0, t1, mult=1.0, depth=36):
    """[len(seeds), per_sample] array of (W(t1) - W(t0)) * mult."""
    out = np.empty((len(seeds), per_sample), dtype=np.float32)
    for b, s in enumerate(seeds):
        out[b] = (brownian_w(int(s), per_sample, t1, T0, T1, depth) - brownian_w(int(s), per_sample, t0, T0, T1, depth)) * np.float32(mult)
    return out
